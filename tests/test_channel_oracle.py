"""Constellation channel front ends (channel.c:112-929): the oracle restatement
(oracle/ems_channel_oracle.c) against outputs of the REAL reference functions -- the committed fixture
tests/golden/channel_const.npz everywhere, and live against oracle/_ref where the reference is built."""
import ctypes as C
import importlib
import os
import sys

import numpy as np
import pytest

from conftest import GOLD, MAT, ROOT, gold

CASES = [("apsk64_n96", "apsk64", "ModelChannel_AWGN_64", "N96_K48_GF64"),
         ("qamr64_n96", "qamr_ssd", "ModelChannel", "N96_K48_GF64"),
         ("qamr256_big", "qamr_ssd", "ModelChannel", "Ahmed_64800_R12_GF256"),
         ("qam4d_big", "qam4d", "ModelChannel_AWGN_256QAM_4D", "Ahmed_64800_R12_GF256"),
         ("qamr16_big", "qamr_ssd", "ModelChannel", "Ahmed_64800_R34_GF16")]          # ModelChannel's GF(16) branch (channel.c:520)
# ModelChannel compiled with the switches the reference ships disabled (channel.c:20-26; oracle/Makefile CHAN_VARIANTS):
# the oracle kind is also the name of the variant library
VARIANTS = [("apskrf256_big", "apsk_rf", "Ahmed_64800_R12_GF256"),
            ("apskrf64_n96", "apsk_rf", "N96_K48_GF64"),
            ("qamawgn64_n96", "qam_awgn", "N96_K48_GF64"),
            ("qamawgn16_big", "qam_awgn", "Ahmed_64800_R34_GF16"),
            ("qamrssdne256_big", "qamr_ssd_ne", "Ahmed_64800_R12_GF256")]


def field_order(matrix):
    return 16 if "GF16" in matrix else (64 if "GF64" in matrix else 256)


@pytest.fixture(scope="module")
def chan(orc):
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    return importlib.import_module("orc_chan")


def oracle_frame(orc, chan, raw, tag, kind, matrix):
    ch = chan.Channel(kind, field_order(matrix), raw)
    cw = raw[tag + "_cw"].astype(np.int32)
    sigma = ch.sigma(float(raw[tag + "_ebn"]))
    obs, att = ch.draw(orc.Rng(0), cw, sigma)
    return ch, obs, att, sigma, ch.llr(obs, att, sigma)


@pytest.mark.parametrize("tag,kind,fn,matrix", CASES)
def test_oracle_channels_reproduce_reference_outputs(orc, chan, tag, kind, fn, matrix):
    """Same codeword, same drand48 stream: every LLR and the sorted (LLR, symbol) rows the reference function left in
    decoder.intrinsic_LLR / intrinsic_GF, bit for bit (head rows in full, CRC-32 of the whole blocks)."""
    raw = gold("channel_const.npz")
    ch, obs, att, sigma, llr = oracle_frame(orc, chan, raw, tag, kind, matrix)
    o = orc.Oracle(os.path.join(MAT, matrix))
    sl, sg = o.sort_llr(llr)
    head = raw[tag + "_llr_head"].shape[0]
    assert sl[:head].tobytes() == raw[tag + "_llr_head"].tobytes()
    assert (sg[:head] == raw[tag + "_gf_head"]).all()
    assert orc.crc32(sl) == int(raw[tag + "_crc"][0]) and orc.crc32(sg.astype(np.int32)) == int(raw[tag + "_crc"][1])
    if kind != "apsk64":                                  # fading + erasures are really exercised
        assert (att == 0).mean() > 0.05 and (att > 0).mean() > 0.8


@pytest.mark.parametrize("tag,kind,matrix", VARIANTS)
def test_oracle_channel_variants_reproduce_reference_outputs(orc, chan, tag, kind, matrix):
    """ModelChannel with the other constellation / fading switches: the fixture holds the outputs of reference builds made
    by toggling the #defines on the way into the compiler (nothing of the reference is modified or copied)."""
    raw = gold("channel_const.npz")
    ch, obs, att, sigma, llr = oracle_frame(orc, chan, raw, tag, kind, matrix)
    o = orc.Oracle(os.path.join(MAT, matrix))
    sl, sg = o.sort_llr(llr)
    head = raw[tag + "_llr_head"].shape[0]
    assert sl[:head].tobytes() == raw[tag + "_llr_head"].tobytes()
    assert (sg[:head] == raw[tag + "_gf_head"]).all()
    assert orc.crc32(sl) == int(raw[tag + "_crc"][0]) and orc.crc32(sg.astype(np.int32)) == int(raw[tag + "_crc"][1])
    if kind == "qam_awgn":
        assert (att == 1).all()
    elif kind == "apsk_rf":
        assert (att[:, 0] == att[:, 1]).all() and len(np.unique(att[:, 0])) > len(att) // 2     # one draw per node
    else:
        assert (att > 0).all() and (att[:, 0] != att[:, 1]).mean() > 0.99                       # per axis, nothing erased


def test_erased_axes_are_skipped_and_fully_erased_nodes_give_zero_rows(orc, chan):
    raw = gold("channel_const.npz")
    ch = chan.Channel("qamr_ssd", 64, raw)
    obs = np.array([[0.3, -0.2], [0.1, 0.4], [1.0, 1.0]], np.float32)
    att = np.array([[0.0, 0.0], [0.0, 1.2], [0.7, 0.0]], np.float32)
    llr = ch.llr(obs, att, 0.2)
    assert (llr[0] == 0).all()
    t = ch.table[ch.idx2bin]                                # rows by symbol index
    e1 = (obs[1, 1] - att[1, 1] * t[:, 1]).astype(np.float32)
    assert llr[1].tobytes() == ((e1 * e1).astype(np.float64) / (2.0 * float(np.float32(0.2) * np.float32(0.2)))).astype(np.float32).tobytes()


LIVE = [(t, k, f, m, "") for t, k, f, m in CASES[:3] + CASES[4:]] + [(t, k, "ModelChannel", m, k) for t, k, m in VARIANTS]


@pytest.mark.parametrize("tag,kind,fn,matrix,variant", LIVE)
def test_oracle_channels_vs_live_reference(orc, chan, tag, kind, fn, matrix, variant):
    """Other codewords / SNRs than the fixture holds, against the real functions of oracle/_ref."""
    if not orc.ref_available() or not os.path.exists(os.path.join(orc.REF_DIR, "libref_ubs_apsk_rf.so")):
        pytest.skip("oracle/_ref not built (reference tree not mounted)")
    sys.path.insert(0, GOLD)
    mg = importlib.import_module("make_golden_channels")
    raw = gold("channel_const.npz")
    ch = chan.Channel(kind, field_order(matrix), raw)
    o = orc.Oracle(os.path.join(MAT, matrix))
    for ebn, seed in ((8.0, 11), (17.5, 12)):
        cw, sl, sg = mg.run_reference(fn, matrix, ebn, seed, variant)
        sigma = ch.sigma(ebn)
        obs, att = ch.draw(orc.Rng(0), cw, sigma)
        s2, g2 = o.sort_llr(ch.llr(obs, att, sigma))
        assert s2.tobytes() == sl.tobytes() and (g2 == sg).all()


def test_older_generation_channel_reproduces_reference_output(orc, chan):
    """ModelChannelAWGN of EMS_HS_L-BubbleCheck_UBS_decoder_v2.c (:1167-1499; grid constellation, one fading draw per
    node, max-is-best likelihoods, decreasing sort) on the all-zero codeword with srand48(5): the oracle's draws + mode-2
    LLR + sort against decoder.likelihood / likelihood_index of the real function (fixture; live where oracle/_ref is built)."""
    raw = gold("channel_const.npz")
    ch = chan.Channel("v2grid", 256, None)
    ebn, seed = float(raw["v2chan_ebn"]), int(raw["v2chan_seed"])
    sigma = ch.sigma(ebn)
    N = 8100
    obs, att = ch.draw(orc.Rng((seed << 16) | 0x330E), np.zeros(N, np.int32), sigma)          # srand48(seed)
    like = ch.llr(obs, att, sigma)
    assert (like <= 0).all()
    sl, si = ch.sort_likelihood(like)
    assert sl[:16].tobytes() == raw["v2chan_like_head"].tobytes() and (si[:16] == raw["v2chan_idx_head"]).all()
    assert orc.crc32(sl) == int(raw["v2chan_crc"][0]) and orc.crc32(si) == int(raw["v2chan_crc"][1])
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    v2 = importlib.import_module("orc_v2")
    if os.path.exists(v2.REF_V2):
        r = v2.RefV2(os.path.join(MAT, "Ahmed_64800_R12_GF256"), 20)
        for ebn2, seed2 in ((7.0, 9), (15.0, 10)):
            like_r, idx_r = r.channel_zero_codeword(ebn2, seed2)
            s2 = ch.sigma(ebn2)
            o2, a2 = ch.draw(orc.Rng((seed2 << 16) | 0x330E), np.zeros(N, np.int32), s2)
            l2, i2 = ch.sort_likelihood(ch.llr(o2, a2, s2))
            assert l2.tobytes() == like_r.tobytes() and (i2 == idx_r).all()
        r.close()


def test_sum_mode_with_fading(orc, chan):
    """EMS_LLR_SUM with attenuations (the `rayleigh_fading` form of channel.c:681-684): one rounding of the double sum,
    0 for a node whose attenuation is <= 0; mode 2 is its negation except for the sign of an exact zero."""
    raw = gold("channel_const.npz")
    ch = chan.Channel("qamr_ssd", 64, raw)
    rs = np.random.RandomState(4)
    obs = rs.randn(50, 2).astype(np.float32)
    a = np.abs(rs.randn(50)).astype(np.float32)
    a[::7] = 0.0
    att = np.stack([a, a], 1)
    s0 = ch.llr(obs, att, 0.4, mode=chan.MODE_SUM)
    s2 = ch.llr(obs, att, 0.4, mode=chan.MODE_NEG)
    assert (s0[::7] == 0).all() and (s0[1] > 0).all()
    assert (s2 == -s0).all()
    t = ch.table[ch.idx2bin]
    e0 = (obs[1, 0] - t[:, 0] * a[1]).astype(np.float32); e1 = (obs[1, 1] - a[1] * t[:, 1]).astype(np.float32)
    den = 2.0 * float(np.float32(0.4) * np.float32(0.4))
    assert s0[1].tobytes() == ((e0 * e0).astype(np.float64) / den + (e1 * e1).astype(np.float64) / den).astype(np.float32).tobytes()


# ---- whole simulations of the reference's main() behind its other channel front ends (NB_LDPC.c:262-264 switched in; round 5) ----
KIND_OF_CHAN = {"apsk": "apsk64", "qam": "qamr_ssd", "4d": "qam4d"}


def oracle_simulation(orc, chan, k):
    """The loop of NB_LDPC.c:250-506 in oracle calls: RandomBinaryGenerator, Encoding, the channel's draws and LLR rows, the decode,
    the bit-error count; returns the cumulative trace the executable prints."""
    raw = np.load(os.path.join(GOLD, "channel_const.npz"))
    o = orc.Oracle(os.path.join(MAT, k["matrix"]))
    ch = chan.Channel(KIND_OF_CHAN[k["chan"]], o.q, raw)
    L = o.L
    rng = orc.Rng(0)
    sigma = ch.sigma(k["EbN"])
    rec = []
    errf = 0
    for _ in range(k["frames"]):
        ks = np.zeros(o.K, np.int32)
        cw = np.zeros(o.N, np.int32)
        L.orc_random_info(C.byref(rng), o.code_p, C.byref(o.gf), ks.ctypes.data_as(C.POINTER(C.c_int)))
        L.orc_encode(C.c_void_p(o.enc), ks.ctypes.data_as(C.POINTER(C.c_int)), cw.ctypes.data_as(C.POINTER(C.c_int)))
        obs, att = ch.draw(rng, cw, sigma)
        res = o.decode(ch.llr(obs, att if ch.mode != chan.MODE_SUM else None, sigma), k["nm"], k["NbOper"], k["offset"], k["NbIterMax"])
        nb = o.bit_errors(res["decide"], cw)
        rec.append((nb, res["synd"], res["iters"]))
        errf += nb != 0
        if errf == 40:
            break
    return [list(t) for t in orc.trace_from_records(np.array(rec, np.int32), o.K, o.logq)]


@pytest.mark.parametrize("name", ["Q1", "Q2", "Q3", "Q4", "Q5"])
def test_oracle_reproduces_the_executables_with_the_other_channels(orc, chan, kat, name):
    """tests/golden/kat_traces.json Q1..Q5: the REAL executable with ModelChannel_AWGN_64 / ModelChannel / ModelChannel_AWGN_256QAM_4D
    switched into its main() (oracle/_ref/essai_ubs_apsk / _qam / _4d), frame for frame: draws, LLR rows, decode and counting together."""
    k = kat[name]
    assert oracle_simulation(orc, chan, k) == [list(t) for t in k["trace"]]
