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
         ("qam4d_big", "qam4d", "ModelChannel_AWGN_256QAM_4D", "Ahmed_64800_R12_GF256")]


@pytest.fixture(scope="module")
def chan(orc):
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    return importlib.import_module("orc_chan")


def oracle_frame(orc, chan, raw, tag, kind, matrix):
    q = 64 if "N96" in matrix else 256
    ch = chan.Channel(kind, q, raw)
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


@pytest.mark.parametrize("tag,kind,fn,matrix", CASES[:3])
def test_oracle_channels_vs_live_reference(orc, chan, tag, kind, fn, matrix):
    """Other codewords / SNRs than the fixture holds, against the real functions of oracle/_ref."""
    if not orc.ref_available():
        pytest.skip("oracle/_ref not built (reference tree not mounted)")
    sys.path.insert(0, GOLD)
    mg = importlib.import_module("make_golden_channels")
    raw = gold("channel_const.npz")
    q = 64 if "N96" in matrix else 256
    ch = chan.Channel(kind, q, raw)
    o = orc.Oracle(os.path.join(MAT, matrix))
    for ebn, seed in ((8.0, 11), (17.5, 12)):
        cw, sl, sg = mg.run_reference(fn, matrix, ebn, seed)
        sigma = ch.sigma(ebn)
        obs, att = ch.draw(orc.Rng(0), cw, sigma)
        s2, g2 = o.sort_llr(ch.llr(obs, att, sigma))
        assert s2.tobytes() == sl.tobytes() and (g2 == sg).all()
