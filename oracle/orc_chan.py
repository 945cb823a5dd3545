"""ctypes bindings of oracle/ems_channel_oracle.c -- TEST INFRASTRUCTURE ONLY.

Constellation channel front ends of the reference (channel.c:112-929).  `Channel` bundles one
of the three shipped front ends (or one of ModelChannel's switched-off compile-time variants) with its constellation table built from the fixture's raw tables exactly as the
reference builds it (rotation, normalisation), and splits it the way the product is split: draw() = harness side
(RNG, modulation, fading, erasures), llr() = the hot-path part the GPU kernel replaces."""
import ctypes as C
import os

import numpy as np

import orc

# draw kinds of orc_chan_draw: ModelChannel_AWGN_64 / ModelChannel as shipped / ModelChannel_AWGN_256QAM_4D, then ModelChannel's
# compile-time variants (channel.c:20-26): APSK + rayleigh_fading, plain QAM without fading, rotated QAM + SSD without erasures
KINDS = {"apsk64": 0, "qamr_ssd": 1, "qam4d": 2, "apsk_rf": 3, "qam_awgn": 0, "qamr_ssd_ne": 4}
MODE_SUM, MODE_SSD, MODE_NEG = 0, 1, 2


def _f(a): return a.ctypes.data_as(C.POINTER(C.c_float))
def _i(a): return a.ctypes.data_as(C.POINTER(C.c_int))


def lib():
    L = orc.lib()
    L.orc_chan_sigma.restype = C.c_float
    L.orc_chan_sigma.argtypes = [C.c_float]
    return L


class Channel:
    def __init__(self, kind, q, raw):
        """raw: the fixture tests/golden/channel_const.npz (raw tables of the reference's include/channel.h)."""
        L = lib()
        self.kind, self.q = kind, q
        self.gf = orc.GF()
        assert L.orc_gf_init(C.byref(self.gf), q) == 0
        self.idx2bin = np.frombuffer(bytes(self.gf.idx2bin), dtype=np.uint8)[:q].copy()
        if kind == "apsk64":
            assert q == 64
            self.D, self.mode = 2, MODE_SUM
            t = np.zeros((64, 2), np.float32)
            L.orc_const_apsk64(raw["apsk64_radius"].astype(np.float64).ctypes.data_as(C.POINTER(C.c_double)),
                               _i(raw["apsk64_num"].astype(np.int32)), _i(raw["apsk64_den"].astype(np.int32)), _f(t))
        elif kind in ("qamr_ssd", "qamr_ssd_ne"):
            self.D, self.mode = 2, MODE_SSD
            src = np.ascontiguousarray(raw["qam%d" % q], np.float32)
            t = np.zeros((q, 2), np.float32)
            L.orc_const_rotate(_f(src), q, _f(t))
        elif kind == "qam_awgn":                         # channel.c:377-389 / :502-510 / :539-552, LLR :717
            self.D, self.mode = 2, MODE_SUM
            t = np.ascontiguousarray(raw["qam%d" % q], np.float32).copy()
        elif kind == "apsk_rf":                          # channel.c:361-376 / :417-501, LLR :696-700
            self.D, self.mode = 2, MODE_SUM
            t = np.zeros((q, 2), np.float32)
            if q == 256:
                L.orc_const_apsk256(_f(np.ascontiguousarray(raw["apsk256_radius"], np.float32)), _f(t))
            else:
                assert q == 64
                L.orc_const_apsk64(raw["apsk64mc_radius"].astype(np.float64).ctypes.data_as(C.POINTER(C.c_double)),
                                   _i(raw["apsk64mc_num"].astype(np.int32)), _i(raw["apsk64mc_den"].astype(np.int32)), _f(t))
        elif kind == "qam4d":
            assert q == 256
            self.D, self.mode = 4, MODE_SSD
            t = np.ascontiguousarray(raw["qam256_4d_16qam_r"], np.float32).copy()
        elif kind == "v2grid":                           # the older generation's ModelChannelAWGN (v2.c:1167): no normalisation pass
            assert q == 256
            self.D, self.mode = 2, MODE_NEG
            t = np.zeros((256, 2), np.float32)
            L.orc_v2_table(_f(t))
        else:
            raise ValueError(kind)
        if kind != "v2grid":
            L.orc_const_normalise(_f(t), q, self.D)
        self.table = t                                   # [q][D], indexed by the symbol's binary image

    def sigma(self, EbN):
        return float(lib().orc_chan_sigma(C.c_float(EbN)))

    def draw(self, rng, codeword, sigma):
        """One frame: (obs [N][D], att [N][D]) from the drand48 stream `rng` (orc.Rng), as the reference draws them."""
        N = len(codeword)
        images = self.idx2bin[np.asarray(codeword)].astype(np.int32)
        obs = np.zeros((N, self.D), np.float32)
        att = np.zeros((N, self.D), np.float32)
        if self.kind == "v2grid":
            lib().orc_v2_chan_draw(C.byref(rng), _f(self.table), N, _i(images), C.c_float(sigma), _f(obs), _f(att))
            return obs, att
        lib().orc_chan_draw(C.byref(rng), KINDS[self.kind], _f(self.table), self.D, N, _i(images), C.c_float(sigma), _f(obs), _f(att))
        return obs, att

    def llr(self, obs, att, sigma, mode=None):
        N = obs.shape[0]
        out = np.zeros((N, self.q), np.float32)
        obs = np.ascontiguousarray(obs, np.float32)
        att = None if att is None else np.ascontiguousarray(att, np.float32)
        lib().orc_const_llr(C.byref(self.gf), _f(self.table), self.q, self.D, self.mode if mode is None else mode, _f(obs),
                            _f(att) if att is not None else None, N, C.c_float(sigma), _f(out))
        return out

    def sort_likelihood(self, like):
        """Decreasing order with symbols (decoder.likelihood / likelihood_index of the older generation, v2.c:1483-1497)."""
        N = like.shape[0]
        sl, si = np.zeros((N, self.q), np.float32), np.zeros((N, self.q), np.int32)
        lib().orc_v2_sort_likelihood(_f(np.ascontiguousarray(like, np.float32)), N, self.q, _f(sl), _i(si))
        return sl, si
