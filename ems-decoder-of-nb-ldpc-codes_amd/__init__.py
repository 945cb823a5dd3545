"""Host-side Python mirror of the MI355X EMS NB-LDPC decoder library (libems_hip.so).

This package is plumbing over the C ABI of ``include/ems_abi.h`` / ``include/nbldpc_ref_api.h``: it
loads the shared library with ctypes and exposes the reference's operator names (``LoadCode``,
``LoadTables``, ``ModelChannel_AWGN_BPSK``, ``Decoding``, ``CheckPassLogEMS``, ``ElementaryStep``,
``Decision``, ``Syndrom``) plus the batched ``Decoder`` the GPU path really uses.  All compute runs in
the hand-written HIP kernels; nothing here (or in the library) falls back to the CPU, and nothing here
imports ``oracle/``.
"""
import ctypes as C
import os
import subprocess

import numpy as np

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_PATH = os.environ.get("EMS_LIB") or os.path.join(PKG_DIR, "libems_hip.so")  # EMS_LIB: developer override (ablation builds)
SIM_PATH = os.path.join(PKG_DIR, "ems_sim")
SIM_V2_PATH = os.path.join(PKG_DIR, "ems_sim_v2")   # the older generation's harness (csrc/harness/ems_sim_v2.c)

LLR_SUM, LLR_SSD, LLR_NEG = 0, 1, 2   # ems_abi.h: EMS_LLR_SUM / EMS_LLR_SSD / EMS_LLR_NEG
EMS_OK, EMS_EINVAL, EMS_ENODEV, EMS_EHIP, EMS_EIO, EMS_ENOMEM, EMS_EUNSUPPORTED = 0, -1, -2, -3, -4, -5, -6


class EmsError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("ems error %d: %s" % (code, msg))
        self.code = code


def build(force=False):
    """Compile the HIP kernels + C ABI for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    args = ["make", "-s", "-C", CSRC]
    if force:
        args.append("-B")
    subprocess.check_call(args)
    return LIB_PATH


class CodeDesc(C.Structure):
    _fields_ = [("N", C.c_int), ("M", C.c_int), ("GF", C.c_int), ("logGF", C.c_int), ("nbBranch", C.c_int),
                ("rowDegree", C.POINTER(C.c_int)), ("col", C.POINTER(C.c_int)), ("coef", C.POINTER(C.c_int))]


class Params(C.Structure):
    _fields_ = [("nm", C.c_int), ("nbOper", C.c_int), ("offset", C.c_float), ("NbIterMax", C.c_int),
                ("early_exit", C.c_int)]


class Tuning(C.Structure):
    """ems_abi.h: ems_tuning -- launch-shape overrides, -1 = the library's choice."""
    _fields_ = [("dual", C.c_int), ("waves", C.c_int), ("occ4", C.c_int), ("twin", C.c_int)]


_lib = None


def lib():
    """Load libems_hip.so.  Raises if it has not been built: there is no Python/CPU fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise EmsError(EMS_ENODEV, "libems_hip.so is not built (run build() / make -C csrc); no fallback exists")
        L = C.CDLL(LIB_PATH)
        L.ems_last_error.restype = C.c_char_p
        L.ems_version.restype = C.c_char_p
        L.ems_last_llr_ms.restype = C.c_float
        L.ems_last_decode_ms.restype = C.c_float
        L.ems_last_llr_ms.argtypes = [C.c_void_p]
        L.ems_last_decode_ms.argtypes = [C.c_void_p]
        L.ems_noisy_dev_ptr.restype = C.c_void_p
        L.ems_noisy_dev_ptr.argtypes = [C.c_void_p]
        L.ems_stream.restype = C.c_void_p
        L.ems_stream.argtypes = [C.c_void_p]
        L.ems_create.argtypes = [C.POINTER(C.c_void_p), C.POINTER(CodeDesc), C.POINTER(Params), C.c_int, C.c_int]
        L.ems_destroy.argtypes = [C.c_void_p]
        L.ems_set_params.argtypes = [C.c_void_p, C.POINTER(Params)]
        L.ems_llr_init.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_int]
        L.ems_llr_init_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_int]
        L.ems_load_llr.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.ems_llr_init_constellation.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_int]
        L.ems_decode.argtypes = [C.c_void_p, C.c_int]
        L.ems_get_results.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.ems_decode_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.ems_count_errors.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.ems_set_codewords.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.ems_count_errors_resident.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.ems_source_attach.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.ems_source_draw.argtypes = [C.c_void_p, C.c_ulonglong, C.c_ulonglong, C.c_int, C.c_float]
        L.ems_source_get.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.ems_source_draw_reference.argtypes = [C.c_void_p, C.c_ulonglong, C.c_int, C.c_float, C.POINTER(C.c_int)]
        L.ems_get_app.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.ems_get_ctov.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.ems_get_schedule.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.c_void_p, C.c_void_p]
        L.ems_synchronize.argtypes = [C.c_void_p]
        L.ems_level_schedule.argtypes = [C.POINTER(CodeDesc), C.POINTER(C.c_int), C.c_void_p, C.c_void_p]
        L.ems_get_sorted_llr.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.ems_load_code_file.argtypes = [C.c_char_p, C.c_int, C.POINTER(CodeDesc)]
        L.ems_free_code_desc.argtypes = [C.POINTER(CodeDesc)]
        L.ems_gf_tables.argtypes = [C.c_int, C.c_void_p, C.c_void_p]
        L.ems_elementary_step.argtypes = [C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 6 + [C.c_int]
        L.ems_check_node.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_float] + [C.c_void_p] * 4 + [C.c_int]
        L.ems_decision.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        L.ems_syndrome.argtypes = [C.POINTER(CodeDesc), C.c_void_p, C.POINTER(C.c_int), C.c_int]
        L.ems_last_kernel_name.restype = C.c_char_p
        L.ems_last_kernel_name.argtypes = [C.c_void_p]
        L.ems_last_workgroup_waves.argtypes = [C.c_void_p]
        L.ems_device_cu_count.argtypes = [C.c_void_p]
        L.ems_set_tuning.argtypes = [C.c_void_p, C.POINTER(Tuning)]
        L.ems_get_tuning.argtypes = [C.c_void_p, C.POINTER(Tuning)]
        L.ems_suggest_batch.argtypes = [C.POINTER(CodeDesc), C.POINTER(Params), C.c_int, C.c_int, C.POINTER(C.c_int)]
        _lib = L
    return _lib


def _check(rc):
    if rc != 0:
        raise EmsError(rc, lib().ems_last_error().decode("utf-8", "replace"))


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class Code:
    """A parity-check matrix as ``LoadCode`` (init.c:143) reads it; both on-disk layouts are accepted."""

    def __init__(self, path, fmt=0):
        self.desc = CodeDesc()
        _check(lib().ems_load_code_file(os.fsencode(path), fmt, C.byref(self.desc)))
        d = self.desc
        self.N, self.M, self.K, self.GF, self.logGF, self.E = d.N, d.M, d.N - d.M, d.GF, d.logGF, d.nbBranch
        self.rate = np.float32(self.K) / np.float32(self.N)          # (float)(N-M)/N, init.c:170
        self.rowDegree = np.ctypeslib.as_array(d.rowDegree, (self.M,))
        self.col = np.ctypeslib.as_array(d.col, (self.E,))
        self.coef = np.ctypeslib.as_array(d.coef, (self.E,))
        self.dc = int(self.rowDegree[0])

    def __del__(self):
        try:
            lib().ems_free_code_desc(C.byref(self.desc))
        except Exception:
            pass


def level_schedule(code):
    """(level_ptr, order) of the layered sweep, computed on the host (no device needed)."""
    nl = C.c_int(0)
    lp = np.zeros(code.M + 1, np.int32)
    order = np.zeros(code.M, np.int32)
    _check(lib().ems_level_schedule(C.byref(code.desc), C.byref(nl), _ptr(lp), _ptr(order)))
    return lp[: nl.value + 1].copy(), order


def pair_schedule(code):
    """Pairs of check nodes [(a, b), ...] (b = -1: single) in the order the two-visits-per-wave kernels claim them."""
    n = C.c_int(0)
    buf = np.zeros(2 * code.M, np.int32)
    _check(lib().ems_pair_schedule(C.byref(code.desc), C.byref(n), _ptr(buf)))
    return buf[: 2 * n.value].reshape(-1, 2).copy()


def suggest_batch(code, nm, nbOper, offset, NbIterMax, device=0, handles=1):
    """Frames per launch that fill the device with the kernel the library picks for this code (ems_suggest_batch)."""
    prm = Params(nm, nbOper, offset, NbIterMax, 1)
    b = C.c_int(0)
    _check(lib().ems_suggest_batch(C.byref(code.desc), C.byref(prm), device, handles, C.byref(b)))
    return b.value


def gf_tables(q):
    """(idx2bin, bin2idx) as LoadTables (init.c:427) derives them from struct.h's binary images."""
    a = np.zeros(q, np.uint8)
    b = np.zeros(q, np.uint8)
    _check(lib().ems_gf_tables(q, _ptr(a), _ptr(b)))
    return a, b


class Decoder:
    """Batched EMS decoder for one code on one GPU (``ems_create`` ... ``ems_destroy``)."""

    def __init__(self, code, nm, nbOper, offset, NbIterMax, max_batch, device=0, early_exit=True):
        self.code = code
        self.max_batch = max_batch
        self.prm = Params(nm, nbOper, offset, NbIterMax, int(bool(early_exit)))
        self.h = C.c_void_p()
        _check(lib().ems_create(C.byref(self.h), C.byref(code.desc), C.byref(self.prm), max_batch, device))

    def close(self):
        if self.h:
            lib().ems_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_params(self, **kw):
        for k, v in kw.items():
            setattr(self.prm, k, v)
        _check(lib().ems_set_params(self.h, C.byref(self.prm)))

    # ---- hot path ----
    def llr_init(self, noisy, sigma):
        noisy = np.ascontiguousarray(noisy, np.float32)
        B = noisy.shape[0]
        assert noisy.shape == (B, self.code.N, self.code.logGF)
        _check(lib().ems_llr_init(self.h, _ptr(noisy), np.float32(sigma), B))
        return B

    def llr_init_constellation(self, obs, att, table, mode, sigma):
        """Constellation front ends (channel.c:112-929): obs / att [B][N][D], table [GF][D] by binary image,
        mode LLR_SUM (att may be None) or LLR_SSD."""
        obs = np.ascontiguousarray(obs, np.float32)
        B, N, D = obs.shape
        assert N == self.code.N
        att = None if att is None else np.ascontiguousarray(att, np.float32)
        table = np.ascontiguousarray(table, np.float32)
        assert table.shape == (self.code.GF, D) and (att is None or att.shape == obs.shape)
        _check(lib().ems_llr_init_constellation(self.h, _ptr(obs), _ptr(att) if att is not None else None, _ptr(table), D, mode,
                                                np.float32(sigma), B))
        return B

    def llr_init_dev(self, dev_ptr, sigma, B):
        _check(lib().ems_llr_init_dev(self.h, C.c_void_p(dev_ptr), np.float32(sigma), B))

    def load_llr(self, llr):
        llr = np.ascontiguousarray(llr, np.float32)
        B = llr.shape[0]
        assert llr.shape == (B, self.code.N, self.code.GF)
        _check(lib().ems_load_llr(self.h, _ptr(llr), B))
        return B

    def decode(self, B):
        _check(lib().ems_decode(self.h, B))

    def results(self, B, want_decide=True):
        decide = np.zeros((B, self.code.N), np.int32) if want_decide else None
        iters = np.zeros(B, np.int32)
        synd = np.zeros(B, np.int32)
        _check(lib().ems_get_results(self.h, B, _ptr(decide) if want_decide else None, _ptr(iters), _ptr(synd)))
        return decide, iters, synd

    def decode_batch(self, noisy, sigma):
        B = self.llr_init(noisy, sigma)
        self.decode(B)
        return self.results(B)

    # ---- v2 flavour (include/ems_abi.h: ems_v2_*) ----
    def v2_load_likelihood(self, like, like_idx):
        like = np.ascontiguousarray(like, np.float32)
        like_idx = np.ascontiguousarray(like_idx, np.int32)
        B = like.shape[0]
        assert like.shape == (B, self.code.N, self.code.GF) and like_idx.shape == like.shape
        _check(lib().ems_v2_load_likelihood(self.h, _ptr(like), _ptr(like_idx), B))
        return B

    def v2_channel_likelihood(self, obs, att, table, sigma):
        """The older generation's channel front end, LLR part (v2.c:1472-1497): obs / att [B][N][2], table [GF][2]."""
        obs = np.ascontiguousarray(obs, np.float32)
        att = np.ascontiguousarray(att, np.float32)
        table = np.ascontiguousarray(table, np.float32)
        B = obs.shape[0]
        assert obs.shape == (B, self.code.N, 2) and att.shape == obs.shape and table.shape == (self.code.GF, 2)
        L = lib()
        L.ems_v2_channel_likelihood.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_int]
        _check(L.ems_v2_channel_likelihood(self.h, _ptr(obs), _ptr(att), _ptr(table), np.float32(sigma), B))
        return B

    def v2_sorted_likelihood(self, frame):
        """decoder.likelihood / likelihood_index of one frame (v2.c:1483-1497), sorted on the GPU."""
        like = np.zeros((self.code.N, self.code.GF), np.float32)
        idx = np.zeros((self.code.N, self.code.GF), np.int32)
        L = lib()
        L.ems_v2_get_sorted_likelihood.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        _check(L.ems_v2_get_sorted_likelihood(self.h, frame, _ptr(like), _ptr(idx)))
        return like, idx

    def v2_decode(self, B):
        _check(lib().ems_v2_decode(self.h, B))

    def v2_reinit(self, B):
        _check(lib().ems_v2_reinit(self.h, B))

    def v2_decode_batch(self, like, like_idx):
        B = self.v2_load_likelihood(like, like_idx)
        self.v2_decode(B)
        return self.results(B)

    def v2_enable_snapshots(self, on=True):
        """ems_v2_enable_snapshots: the BER samples of the older generation's main() (v2.c:2174-2243)."""
        L = lib()
        L.ems_v2_enable_snapshots.argtypes = [C.c_void_p, C.c_int]
        _check(L.ems_v2_enable_snapshots(self.h, int(bool(on))))

    def v2_snapshots(self, B):
        """[B][5] bit errors of the decisions in force when iterations 9, 19, 29, 39, 49 begin (0 where the frame had stopped)."""
        snap = np.zeros((B, 5), np.int32)
        L = lib()
        L.ems_v2_get_snapshots.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        _check(L.ems_v2_get_snapshots(self.h, B, _ptr(snap)))
        return snap

    def v2_state(self, frame):
        E, nm, N, q = self.code.desc.nbBranch, self.prm.nm, self.code.N, self.code.GF
        st = dict(vtoc=np.zeros((E, nm), np.float32), vtoci=np.zeros((E, nm), np.int32), ctov=np.zeros((E, nm), np.float32),
                  ctovi=np.zeros((E, nm), np.int32), app=np.zeros((N, q), np.float32))
        _check(lib().ems_v2_get_state(self.h, frame, _ptr(st["vtoc"]), _ptr(st["vtoci"]), _ptr(st["ctov"]), _ptr(st["ctovi"]), _ptr(st["app"])))
        return st

    def count_errors(self, codewords):
        cw = np.ascontiguousarray(codewords, np.int32)
        B = cw.shape[0]
        out = np.zeros(B, np.int32)
        _check(lib().ems_count_errors(self.h, _ptr(cw), B, _ptr(out)))
        return out

    def set_codewords(self, codewords):
        cw = np.ascontiguousarray(codewords, np.int32)
        _check(lib().ems_set_codewords(self.h, _ptr(cw), cw.shape[0]))

    # ---- device-side frame source (include/ems_abi.h: ems_source_*) ----
    def source_attach(self, session):
        """`session`: a refapi.RefSession built with need_encoder=True (GaussianElimination's matUT / Perm)."""
        c = session.code
        if not c.matUT or not c.Perm:
            raise ValueError("session has no encoder (need_encoder=True)")
        _check(lib().ems_source_attach(self.h, c.matUT[0], c.Perm))

    def source_draw(self, seed, first_frame, B, sigma):
        """Frames [first_frame, first_frame+B) generated in HBM (Philox stream `seed`), LLRs initialised."""
        _check(lib().ems_source_draw(self.h, int(seed), int(first_frame), int(B), float(sigma)))

    def source_draw_reference(self, first_frame, B, sigma):
        """Frames [first_frame, first_frame+B) of the REFERENCE's drand48 stream generated in HBM, bit for bit (ems_source_draw_reference);
        returns how many samples the host recomputed (their float rounding was not decidable on the device)."""
        n = C.c_int(0)
        _check(lib().ems_source_draw_reference(self.h, int(first_frame), int(B), float(sigma), C.byref(n)))
        return n.value

    def source_get(self, B):
        cw = np.zeros((B, self.code.N), np.int32)
        noisy = np.zeros((B, self.code.N, self.code.logGF), np.float32)
        _check(lib().ems_source_get(self.h, int(B), _ptr(cw), _ptr(noisy)))
        return cw, noisy

    def count_errors_resident(self, B):
        out = np.zeros(B, np.int32)
        _check(lib().ems_count_errors_resident(self.h, B, _ptr(out)))
        return out

    # ---- introspection ----
    def app(self, frame):
        a = np.zeros((self.code.N, self.code.GF), np.float32)
        _check(lib().ems_get_app(self.h, frame, _ptr(a)))
        return a

    def ctov(self, frame):
        a = np.zeros((self.code.E, self.code.GF), np.float32)
        _check(lib().ems_get_ctov(self.h, frame, _ptr(a)))
        return a

    def sorted_llr(self, frame):
        """decoder_t.intrinsic_LLR / intrinsic_GF of one frame (channel.c:78-91), sorted on the GPU."""
        sl = np.zeros((self.code.N, self.code.GF), np.float32)
        sg = np.zeros((self.code.N, self.code.GF), np.int32)
        _check(lib().ems_get_sorted_llr(self.h, frame, _ptr(sl), _ptr(sg)))
        return sl, sg

    def schedule(self):
        nl = C.c_int(0)
        lp = np.zeros(self.code.M + 1, np.int32)
        order = np.zeros(self.code.M, np.int32)
        _check(lib().ems_get_schedule(self.h, C.byref(nl), _ptr(lp), _ptr(order)))
        return lp[: nl.value + 1].copy(), order

    def synchronize(self):
        _check(lib().ems_synchronize(self.h))

    def set_tuning(self, dual=-1, waves=-1, occ4=0, twin=-1):
        """Launch-shape overrides (ems_set_tuning); -1 = the library's choice."""
        t = Tuning(dual, waves, occ4, twin)
        _check(lib().ems_set_tuning(self.h, C.byref(t)))

    @property
    def tuning(self):
        t = Tuning()
        _check(lib().ems_get_tuning(self.h, C.byref(t)))
        return dict(dual=t.dual, waves=t.waves, occ4=t.occ4, twin=t.twin)

    @property
    def last_kernel(self):
        """Name of the decode kernel the last decode() / v2_decode() launched (template arguments included)."""
        return lib().ems_last_kernel_name(self.h).decode()

    @property
    def last_workgroup_waves(self):
        return int(lib().ems_last_workgroup_waves(self.h))

    @property
    def n_cu(self):
        return int(lib().ems_device_cu_count(self.h))

    @property
    def last_decode_ms(self):
        return float(lib().ems_last_decode_ms(self.h))

    @property
    def last_llr_ms(self):
        return float(lib().ems_last_llr_ms(self.h))

    @property
    def noisy_dev_ptr(self):
        return lib().ems_noisy_dev_ptr(self.h)


# ---- single-node operators on the device (bubble_decoder.c / tools.c shapes) ----
def ElementaryStep(Input1, Input2, IndiceInput1, IndiceInput2, GF, nbMax, nbOper, device=0):
    """bubble_decoder.c:316 -- returns (Output, IndiceOut)."""
    a = np.ascontiguousarray(Input1, np.float32); b = np.ascontiguousarray(Input2, np.float32)
    ia = np.ascontiguousarray(IndiceInput1, np.int32); ib = np.ascontiguousarray(IndiceInput2, np.int32)
    out = np.zeros(nbMax, np.float32); io = np.zeros(nbMax, np.int32)
    _check(lib().ems_elementary_step(GF, nbMax, nbOper, _ptr(a), _ptr(b), _ptr(ia), _ptr(ib), _ptr(out), _ptr(io), device))
    return out, io


def CheckPassLogEMS(coef, M_VtoC_LLR, M_VtoC_GF, GF, NbOper, offset, device=0):
    """bubble_decoder.c:72 -- inputs [dc][nm] (sorted, normalised, unrotated); returns M_CtoV_LLR [dc][GF]."""
    coef = np.ascontiguousarray(coef, np.int32)
    vl = np.ascontiguousarray(M_VtoC_LLR, np.float32); vg = np.ascontiguousarray(M_VtoC_GF, np.int32)
    dc, nm = vl.shape
    out = np.zeros((dc, GF), np.float32)
    _check(lib().ems_check_node(GF, dc, nm, NbOper, np.float32(offset), _ptr(coef), _ptr(vl), _ptr(vg), _ptr(out), device))
    return out


def Decision(APP, device=0):
    """tools.c:312"""
    app = np.ascontiguousarray(APP, np.float32)
    d = np.zeros(app.shape[0], np.int32)
    _check(lib().ems_decision(app.shape[1], app.shape[0], _ptr(app), _ptr(d), device))
    return d


def Syndrom(code, decide, device=0):
    """tools.c:284"""
    d = np.ascontiguousarray(decide, np.int32)
    s = C.c_int(0)
    _check(lib().ems_syndrome(C.byref(code.desc), _ptr(d), C.byref(s), device))
    return s.value
