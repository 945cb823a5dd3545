"""GPU parity of the constellation channel front ends (channel.c:112-929): k_llr_const through
ems_llr_init_constellation / ModelChannelLLR against the oracle and against outputs of the REAL reference functions
(tests/golden/channel_const.npz), then whole decodes fed by these front ends."""
import importlib
import os
import sys

import numpy as np
import pytest

from conftest import MAT, ROOT, gold

pytestmark = pytest.mark.gpu
CASES = [("apsk64_n96", "apsk64", "N96_K48_GF64"), ("qamr64_n96", "qamr_ssd", "N96_K48_GF64"),
         ("qamr256_big", "qamr_ssd", "Ahmed_64800_R12_GF256"), ("qam4d_big", "qam4d", "Ahmed_64800_R12_GF256"),
         ("qamr16_big", "qamr_ssd", "Ahmed_64800_R34_GF16"),
         # ModelChannel's switched-off compile-time variants (fixture outputs from reference builds with the #defines toggled)
         ("apskrf256_big", "apsk_rf", "Ahmed_64800_R12_GF256"), ("apskrf64_n96", "apsk_rf", "N96_K48_GF64"),
         ("qamawgn64_n96", "qam_awgn", "N96_K48_GF64"), ("qamawgn16_big", "qam_awgn", "Ahmed_64800_R34_GF16"),
         ("qamrssdne256_big", "qamr_ssd_ne", "Ahmed_64800_R12_GF256")]
NO_ATT = ("apsk64", "qam_awgn")                            # front ends without attenuations (plain AWGN)


def field_order(matrix):
    return 16 if "GF16" in matrix else (64 if "GF64" in matrix else 256)


@pytest.fixture(scope="module")
def chan(orc):
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    return importlib.import_module("orc_chan")


def _frames(orc, ch, o, sigma, n, seed=5):
    rng = orc.Rng(seed)
    out = []
    for _ in range(n):
        cw, _noisy = o.draw_frame(rng, 0.7)               # a codeword of the code (the BPSK samples are not used)
        obs, att = ch.draw(rng, cw, sigma)
        out.append((cw, obs, att))
    return out


@pytest.mark.parametrize("tag,kind,matrix", CASES)
def test_llr_matches_reference_outputs(pkg, orc, chan, tag, kind, matrix):
    """The fixture's frame (same codeword, same drand48 stream as the reference function saw): every LLR the kernel
    writes, and the sorted rows ems_get_sorted_llr builds from them, equal decoder.intrinsic_LLR / intrinsic_GF of the
    real function (head rows in full, CRC-32 of the whole blocks)."""
    raw = gold("channel_const.npz")
    ch = chan.Channel(kind, field_order(matrix), raw)
    sigma = ch.sigma(float(raw[tag + "_ebn"]))
    obs, att = ch.draw(orc.Rng(0), raw[tag + "_cw"].astype(np.int32), sigma)
    dec = pkg.Decoder(pkg.Code(os.path.join(MAT, matrix)), 8, 16, 0.3, 4, max_batch=1)
    dec.llr_init_constellation(obs[None], None if kind in NO_ATT else att[None], ch.table, ch.mode, sigma)
    dec.synchronize()
    assert dec.app(0).tobytes() == ch.llr(obs, att, sigma).tobytes()
    sl, sg = dec.sorted_llr(0)
    head = raw[tag + "_llr_head"].shape[0]
    assert sl[:head].tobytes() == raw[tag + "_llr_head"].tobytes() and (sg[:head] == raw[tag + "_gf_head"]).all()
    assert orc.crc32(sl) == int(raw[tag + "_crc"][0]) and orc.crc32(sg.astype(np.int32)) == int(raw[tag + "_crc"][1])
    dec.close()


@pytest.mark.parametrize("kind,matrix,ebn,nm,B", [("apsk64", "N96_K48_GF64", 9.0, 16, 40), ("qamr_ssd", "N96_K48_GF64", 15.0, 16, 40),
                                                  ("qamr_ssd", "Ahmed_64800_R12_GF256", 23.0, 20, 3), ("qam4d", "Ahmed_64800_R12_GF256", 13.0, 20, 3),
                                                  ("qamr_ssd", "Ahmed_64800_R34_GF16", 18.0, 8, 2),      # ModelChannel's GF(16) branch
                                                  ("apsk_rf", "N96_K48_GF64", 12.0, 16, 40), ("qam_awgn", "Ahmed_64800_R34_GF16", 11.0, 8, 2)])
def test_decode_behind_constellation_front_ends(pkg, orc, chan, kind, matrix, ebn, nm, B):
    """Batches through the front end and the layered decode: APP rows of the init, then decisions, iterations,
    syndromes and the final APP against the oracle fed with the oracle's LLRs."""
    raw = gold("channel_const.npz")
    path = os.path.join(MAT, matrix)
    o = orc.Oracle(path)
    ch = chan.Channel(kind, o.q, raw)
    sigma = ch.sigma(ebn)
    fr = _frames(orc, ch, o, sigma, B)
    obs = np.stack([f[1] for f in fr]); att = np.stack([f[2] for f in fr])
    dec = pkg.Decoder(pkg.Code(path), nm, 2 * nm, 0.3, 8, max_batch=B)
    dec.llr_init_constellation(obs, None if kind in NO_ATT else att, ch.table, ch.mode, sigma)
    dec.decode(B)
    d, it, sy = dec.results(B)
    nconv = 0
    for b in range(B):
        ref = o.decode(ch.llr(obs[b], att[b], sigma), nm, 2 * nm, 0.3, 8, want_state=True)
        assert (d[b] == ref["decide"]).all() and it[b] == ref["iters"] and sy[b] == ref["synd"], "frame %d" % b
        assert dec.app(b).tobytes() == ref["app"].tobytes()
        nconv += int(sy[b] == 0 and (d[b] == fr[b][0]).all())
    assert nconv >= 1                                      # the operating points are decodable: the chain is not vacuous
    dec.close()


def test_erasures_edge_cases(pkg, orc, chan):
    """Every axis of a node erased (all-zero LLR row: the top-nm selection falls back to its literal path), one axis
    erased, and a batch position that differs only in its attenuations."""
    raw = gold("channel_const.npz")
    path = os.path.join(MAT, "N96_K48_GF64")
    o = orc.Oracle(path)
    ch = chan.Channel("qamr_ssd", 64, raw)
    sigma = ch.sigma(16.0)
    (cw, obs, att), = _frames(orc, ch, o, sigma, 1, seed=9)
    att2 = att.copy(); att2[3] = 0.0; att2[7, 0] = 0.0; att2[8, 1] = 0.0
    att3 = np.zeros_like(att)
    dec = pkg.Decoder(pkg.Code(path), 16, 32, 0.3, 6, max_batch=3)
    dec.llr_init_constellation(np.stack([obs, obs, obs]), np.stack([att, att2, att3]), ch.table, ch.mode, sigma)
    dec.decode(3)
    d, it, sy = dec.results(3)
    for b, a in enumerate((att, att2, att3)):
        llr = ch.llr(obs, a, sigma)
        if b == 2:
            assert (llr == 0).all()
        ref = o.decode(llr, 16, 32, 0.3, 6, want_state=True)
        assert (d[b] == ref["decide"]).all() and it[b] == ref["iters"] and sy[b] == ref["synd"]
        assert dec.app(b).tobytes() == ref["app"].tobytes()
    dec.close()


def test_reference_shaped_model_channel_llr_then_decoding(pkg, orc, chan, refapi):
    """ModelChannelLLR -> Decoding() on the reference's own structs (what the patched channel.c of INTEGRATION.md
    section 5 calls), frame after frame."""
    raw = gold("channel_const.npz")
    path = os.path.join(MAT, "N96_K48_GF64")
    o = orc.Oracle(path)
    s = refapi.RefSession(path, 16)
    for kind, ebn in (("apsk64", 10.0), ("qamr_ssd", 15.0)):
        ch = chan.Channel(kind, 64, raw)
        sigma = ch.sigma(ebn)
        for cw, obs, att in _frames(orc, ch, o, sigma, 4, seed=21):
            sl, sg = s.model_channel_llr(obs, None if ch.mode == chan.MODE_SUM else att, ch.table, ch.mode, sigma)
            llr = ch.llr(obs, att, sigma)
            osl, osg = o.sort_llr(llr)
            assert sl.tobytes() == osl.tobytes() and (sg == osg).all()
            it, decide, synd = s.decoding(10, 32, 0.3)
            ref = o.decode(llr, 16, 32, 0.3, 10, want_state=True)
            assert it == ref["iters"] and synd == ref["synd"] and (decide == ref["decide"]).all()
            assert s.APP.tobytes() == ref["app"].tobytes()
    s.close()


def test_reference_shaped_model_channel_llr_gf16(pkg, orc, chan, refapi):
    """The same hoisted symbol on ModelChannel's GF(16) branch (channel.c:520-557) and on one of its switched-off variants."""
    raw = gold("channel_const.npz")
    path = os.path.join(MAT, "Ahmed_64800_R34_GF16")
    o = orc.Oracle(path)
    s = refapi.RefSession(path, 8)
    for kind, ebn in (("qamr_ssd", 18.0), ("qam_awgn", 11.0)):
        ch = chan.Channel(kind, 16, raw)
        sigma = ch.sigma(ebn)
        (cw, obs, att), = _frames(orc, ch, o, sigma, 1, seed=31)
        sl, sg = s.model_channel_llr(obs, None if kind in NO_ATT else att, ch.table, ch.mode, sigma)
        llr = ch.llr(obs, att, sigma)
        osl, osg = o.sort_llr(llr)
        assert sl.tobytes() == osl.tobytes() and (sg == osg).all()
        it, decide, synd = s.decoding(8, 16, 0.3)
        ref = o.decode(llr, 8, 16, 0.3, 8, want_state=True)
        assert it == ref["iters"] and synd == ref["synd"] and (decide == ref["decide"]).all()
        assert s.APP.tobytes() == ref["app"].tobytes()
    s.close()


def test_front_end_argument_checks(pkg):
    dec = pkg.Decoder(pkg.Code(os.path.join(MAT, "N96_K48_GF64")), 16, 32, 0.3, 6, max_batch=1)
    obs = np.zeros((1, 16, 2), np.float32); tab = np.zeros((64, 2), np.float32)
    with pytest.raises(pkg.EmsError) as e:
        dec.llr_init_constellation(obs, None, tab, pkg.LLR_SSD, 0.3)          # SSD needs attenuations
    assert e.value.code == pkg.EMS_EINVAL
    with pytest.raises(pkg.EmsError):
        dec.llr_init_constellation(obs, None, tab, pkg.LLR_SUM, 0.0)          # sigma
    with pytest.raises(pkg.EmsError):
        dec.llr_init_constellation(np.zeros((1, 16, 4), np.float32), None, np.zeros((64, 4), np.float32), pkg.LLR_SUM, 0.3)   # SUM is 2-D
    dec.close()
    d16 = pkg.Decoder(pkg.Code(os.path.join(MAT, "Ahmed_64800_R34_GF16")), 8, 16, 0.3, 4, max_batch=1)
    d16.llr_init_constellation(np.zeros((1, d16.code.N, 2), np.float32), None, np.zeros((16, 2), np.float32), pkg.LLR_SUM, 0.3)   # GF(16): channel.c:520
    d16.synchronize()
    assert (d16.app(0) == 0).all()
    d16.close()


# ---------------------------------------------------------------------------------------------------
# the older generation's own channel front end (EMS_HS_L-BubbleCheck_UBS_decoder_v2.c:1167-1499)
# ---------------------------------------------------------------------------------------------------
def test_older_generation_channel_front_end(pkg, orc, chan):
    """ems_v2_channel_likelihood against the real ModelChannelAWGN (fixture: zero codeword, srand48(5)) and, for random
    codewords, against the oracle; then a whole decode started from it must equal the decode started from the sorted
    likelihood arrays (ems_v2_load_likelihood) and the v2 oracle's, every message array included."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    v2 = importlib.import_module("orc_v2")
    raw = gold("channel_const.npz")
    path = os.path.join(MAT, "Ahmed_64800_R12_GF256")
    ch = chan.Channel("v2grid", 256, None)
    o = v2.V2Oracle(path)
    N = o.N
    ebn, seed = float(raw["v2chan_ebn"]), int(raw["v2chan_seed"])
    sigma = ch.sigma(ebn)
    obs0, att0 = ch.draw(orc.Rng((seed << 16) | 0x330E), np.zeros(N, np.int32), sigma)
    rs = np.random.RandomState(8)
    cw1 = rs.randint(0, 256, N).astype(np.int32)                # not a codeword: the front end does not care
    s1 = ch.sigma(14.0)
    obs1, att1 = ch.draw(orc.Rng(77), cw1, s1)
    att1[5] = 0.0                                               # a node the fading wiped out: all-zero likelihood row
    code = pkg.Code(path)
    dec = pkg.Decoder(code, 20, 40, 0.3, 8, max_batch=2)
    # frame by frame (each has its own sigma)
    for b, (ob, at, sg) in enumerate(((obs0, att0, sigma), (obs1, att1, s1))):
        dec.v2_channel_likelihood(ob[None], at[None], ch.table, sg)
        like, idx = dec.v2_sorted_likelihood(0)
        ol, oi = ch.sort_likelihood(ch.llr(ob, at, sg))
        assert like.tobytes() == ol.tobytes() and (idx == oi).all()
        if b == 0:
            assert like[:16].tobytes() == raw["v2chan_like_head"].tobytes() and (idx[:16] == raw["v2chan_idx_head"]).all()
            assert orc.crc32(like) == int(raw["v2chan_crc"][0]) and orc.crc32(idx) == int(raw["v2chan_crc"][1])
        else:
            assert (like[5] == 0).all()
        dec.v2_decode(1)
        d, it, sy = dec.results(1)
        st = dec.v2_state(0)
        ref = o.decode(ol, oi, 20, 40, 0.3, 8)
        assert it[0] == ref["iters"] and sy[0] == ref["synd"] and (d[0] == ref["decide"]).all()
        for k in ("vtoc", "vtoci", "ctov", "ctovi", "app"):
            assert st[k].tobytes() == ref[k].tobytes(), k
        d2, it2, sy2 = dec.v2_decode_batch(ol[None], oi[None])   # the same frame through the sorted-array input
        assert it2[0] == it[0] and sy2[0] == sy[0] and (d2[0] == d[0]).all()
        dec.v2_channel_likelihood(ob[None], at[None], ch.table, sg)
        dec.v2_decode(1); dec.v2_reinit(1); dec.v2_decode(1)     # reinit re-runs the initialisation from the rows
        d3, it3, sy3 = dec.results(1)
        assert it3[0] == it[0] and (d3[0] == d[0]).all()
    dec.close()


def test_sum_mode_with_fading_on_gpu(pkg, orc, chan):
    """EMS_LLR_SUM with attenuations (channel.c:681-684 form) against the oracle, erased nodes included."""
    raw = gold("channel_const.npz")
    path = os.path.join(MAT, "N96_K48_GF64")
    ch = chan.Channel("qamr_ssd", 64, raw)
    rs = np.random.RandomState(12)
    obs = rs.randn(3, 16, 2).astype(np.float32)
    a = np.abs(rs.randn(3, 16)).astype(np.float32)
    a[:, ::5] = 0.0
    att = np.stack([a, a], 2)
    dec = pkg.Decoder(pkg.Code(path), 16, 32, 0.3, 4, max_batch=3)
    dec.llr_init_constellation(obs, att, ch.table, pkg.LLR_SUM, 0.4)
    dec.synchronize()
    for b in range(3):
        assert dec.app(b).tobytes() == ch.llr(obs[b], att[b], 0.4, mode=chan.MODE_SUM).tobytes()
    dec.close()
