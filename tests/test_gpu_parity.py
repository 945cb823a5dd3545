"""Parity tests proper (run on the MI355X with -m gpu): the HIP path, called through the C ABI, against
the golden vectors generated from the reference and against the CPU oracle on the same seeded inputs.
Bar: bit-exact for everything on the decode path (hard decisions, syndrome, iteration counts, APP and
CtoV floats -- the path uses only f32 add/sub/compare, so the tolerance is 0 ulp); the channel LLR is
also required bit-exact (tolerance 0: the kernel keeps the reference's float/double promotion order)."""
import json
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import GOLD, MAT, FieldOracle, es_case_fields, gold

pytestmark = pytest.mark.gpu


# ---------------------------------------------------------------------------------------------------
# check-node level (bubble_decoder.c) through ElementaryStep / CheckPassLogEMS
# ---------------------------------------------------------------------------------------------------
def test_elementary_step_golden(pkg):
    cases = gold("es_cases.npz")["cases"]
    short = 0
    for row in cases:
        q, nm, nop, a, b, ia, ib, eo, eio = es_case_fields(row)
        out, io = pkg.ElementaryStep(a, b, ia, ib, q, nm, nop)
        assert out.tobytes() == eo.tobytes() and (io == eio).all()
        short += int((eio < 0).any())
    assert short >= 40


def test_check_node_golden(pkg):
    g = gold("cn_cases.npz")
    code = pkg.Code(os.path.join(MAT, "N96_K48_GF64"))
    coefs = code.coef.reshape(code.M, 4)
    for vl, vg, node, nop, out in zip(g["vl"], g["vg"], g["node"], g["nop"], g["out"]):
        got = pkg.CheckPassLogEMS(coefs[node], vl, vg, 64, int(nop), float(g["offset"]))
        assert got.tobytes() == out.tobytes()


@pytest.mark.parametrize("q,nm,dc", [(64, 16, 4), (256, 20, 4), (16, 8, 8), (64, 12, 3), (64, 12, 6), (256, 10, 5), (256, 32, 4),
                                     (64, 8, 2), (16, 5, 4), (256, 20, 20), (64, 16, 17), (16, 8, 33), (256, 12, 10)])
def test_check_node_random_vs_oracle(pkg, orc, q, nm, dc):
    o = FieldOracle(orc, q)
    rs = np.random.RandomState(q + nm + dc)
    for it in range(30):
        vl = np.zeros((dc, nm), np.float32)
        vg = np.zeros((dc, nm), np.int32)
        for t in range(dc):
            vl[t] = np.sort((rs.randint(0, 60, nm) * [0.25, 1.0][it % 2]).astype(np.float32))
            vl[t, 0] = 0
            vg[t] = rs.permutation(q)[:nm]
        coef = rs.randint(1, q, dc).astype(np.int32)
        nop = int(rs.choice([nm, 2 * nm]))
        assert pkg.CheckPassLogEMS(coef, vl, vg, q, nop, 0.3).tobytes() == o.check_node(coef, vl, vg, nop, 0.3).tobytes()


@pytest.mark.parametrize("dc", [8, 5, 12])
def test_check_node_gf16_first_occurrence_form_vs_oracle(pkg, orc, dc):
    """GF(16) with nm = 8 takes the closed form of the elementary step (es_stage_q16: one LDS minimum per cell into a table of 16
    symbols, a rank of the 16 first occurrences, closed-form stop tests) and redoes a stage by the merge form when a tie or the
    NbOper limit could matter.  Continuous values (no ties: the closed form decides alone), coarse ones (ties everywhere: the
    redo path), NbOper from nm (the limit binds) to 3 nm, lists with -1 tails -- dense C->V rows against the oracle, bit for bit."""
    q, nm = 16, 8
    o = FieldOracle(orc, q)
    rs = np.random.RandomState(100 + dc)
    for it in range(120):
        vl = np.zeros((dc, nm), np.float32)
        vg = np.zeros((dc, nm), np.int32)
        for t in range(dc):
            if it % 3 == 0:
                x = rs.uniform(0.0, 12.0, nm)                          # continuous
            elif it % 3 == 1:
                x = rs.randint(0, 24, nm) * 0.5                        # massive ties
            else:
                x = np.round(rs.uniform(0.0, 6.0, nm), 1)              # a few ties
            vl[t] = np.sort(x.astype(np.float32))
            vl[t, 0] = 0
            vg[t] = rs.permutation(q)[:nm]
            if it % 7 == 3 and t % 2:                                  # a short list: (1e5, -1) behind its valid entries
                k = int(rs.randint(5, nm))
                vl[t, k:] = 1e5
                vg[t, k:] = -1
        coef = rs.randint(1, q, dc).astype(np.int32)
        nop = int(rs.choice([nm, nm + 3, 2 * nm, 3 * nm]))
        got = pkg.CheckPassLogEMS(coef, vl, vg, q, nop, 0.3)
        assert got.tobytes() == o.check_node(coef, vl, vg, nop, 0.3).tobytes(), (dc, it, nop)


def test_decision_and_syndrome_units(pkg, orc):
    path = os.path.join(MAT, "Ahmed_64800_R34_GF16")
    o = orc.Oracle(path)
    code = pkg.Code(path)
    rs = np.random.RandomState(5)
    app = (rs.randint(0, 50, (o.N, o.q)) * 0.5).astype(np.float32)     # many ties -> lowest symbol
    app[7] = 2e5                                                       # nothing below 1e5 -> symbol 0
    d = pkg.Decision(app)
    assert (d == o.decision(app)).all() and d[7] == 0
    assert pkg.Syndrom(code, d) == o.syndrome(d)
    cw, _ = o.draw_frame(orc.Rng(0), 0.5)
    assert pkg.Syndrom(code, cw) == 0
    cw[100] ^= 3
    assert pkg.Syndrom(code, cw) == o.syndrome(cw) != 0


# ---------------------------------------------------------------------------------------------------
# channel LLR (channel.c:66-91)
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("tag,mat,nm", [("n96", "N96_K48_GF64", 16), ("gf16", "Ahmed_64800_R34_GF16", 8), ("gf256", "Ahmed_64800_R12_GF256", 20)])
def test_channel_llr_golden(pkg, orc, tag, mat, nm):
    g = gold("channel.npz")
    path = os.path.join(MAT, mat)
    o = orc.Oracle(path)
    sigma = o.sigma(float(g[tag + "_ebn"]))
    _cw, noisy = o.draw_frame(orc.Rng(0), sigma)       # host-side noise stays in the harness (drand48)
    dec = pkg.Decoder(pkg.Code(path), nm, 2 * nm, 0.3, 10, max_batch=1)
    dec.llr_init(noisy[None], sigma)
    dec.synchronize()
    assert dec.app(0).tobytes() == o.llr(noisy, sigma).tobytes()       # every LLR of the frame, 0 ulp
    sl, sg = dec.sorted_llr(0)
    n = g[tag + "_llr"].shape[0]
    assert sl[:n].tobytes() == g[tag + "_llr"].tobytes() and (sg[:n] == g[tag + "_gf"]).all()
    dec.close()


# ---------------------------------------------------------------------------------------------------
# frame decode (NB_LDPC.c:266-472)
# ---------------------------------------------------------------------------------------------------
def _decode_vs_oracle(pkg, orc, name, nm, nop, off, itmax, ebn, B, state=True, early_exit=True):
    path = os.path.join(MAT, name)
    o = orc.Oracle(path)
    sigma = o.sigma(ebn)
    rng = orc.Rng(0)
    frames = [o.draw_frame(rng, sigma) for _ in range(B)]
    noisy = np.stack([f[1] for f in frames])
    cws = np.stack([f[0] for f in frames])
    dec = pkg.Decoder(pkg.Code(path), nm, nop, off, itmax, max_batch=B, early_exit=early_exit)
    decide, iters, synd = dec.decode_batch(noisy, sigma)
    nerr = dec.count_errors(cws)
    for b in range(B):
        ref = o.decode(o.llr(noisy[b], sigma), nm, nop, off, itmax, early_exit=early_exit, want_state=state)
        assert (decide[b] == ref["decide"]).all(), "frame %d: hard decisions differ" % b
        assert iters[b] == ref["iters"] and synd[b] == ref["synd"], "frame %d: iterations / syndrome differ" % b
        assert nerr[b] == o.bit_errors(ref["decide"], cws[b])
        if state:
            assert dec.app(b).tobytes() == ref["app"].tobytes(), "frame %d: APP differs" % b
            assert dec.ctov(b).tobytes() == ref["ctov"].tobytes(), "frame %d: CtoV differs" % b
    dec.close()
    return iters


def test_decode_n96_vs_oracle(pkg, orc):
    it = _decode_vs_oracle(pkg, orc, "N96_K48_GF64", 16, 32, 0.3, 10, 2.0, 96)
    assert list(it[:3]) == [4, 10, 10]                 # K1 frames 1..3: 4 sweeps, then two non-converged


def test_decode_n96_low_snr_fixed_work(pkg, orc):
    _decode_vs_oracle(pkg, orc, "N96_K48_GF64", 16, 32, 0.3, 30, -3.0, 32, early_exit=False)


def test_decode_n96_other_parameters(pkg, orc):
    _decode_vs_oracle(pkg, orc, "KN_N96_K48_GF64.txt", 12, 18, 1.0, 8, 1.0, 16)     # NbOper = 1.5 nm, offset 1
    _decode_vs_oracle(pkg, orc, "N96_K48_GF64", 30, 45, 0.3, 5, 3.0, 8)             # start.sh parameters
    _decode_vs_oracle(pkg, orc, "Mat24_N192_M96", 24, 48, 1.0, 20, 1.75, 24)        # NB_LDPC.c:55-57


def test_decode_irregular_check_degrees_vs_oracle(pkg, orc):
    """SURVEY.md 8(f) f4, second half: check degrees 3 / 4 / 5 in one code (tests/golden/make_irregular_matrix.py; row 0 carries the
    largest, where the reference itself is well defined -- kat_traces.json I1 / I2 are the real executable's traces and the oracle
    reproduces them, tests/test_oracle_golden.py).  The generic one-visit kernel laid out for the largest degree, every visit with its
    own degree: whole-frame state (hard decisions, iterations, syndrome, bit errors, APP, every C->V row) against the oracle."""
    _decode_vs_oracle(pkg, orc, "Irr345_N60_M24_GF64", 12, 24, 0.3, 12, 3.0, 24)
    _decode_vs_oracle(pkg, orc, "Irr345_N60_M24_GF64", 16, 19, 1.0, 6, 1.0, 8, early_exit=False)
    _decode_vs_oracle(pkg, orc, "Irr345_N60_M24_GF256", 16, 32, 0.5, 10, 3.5, 16)
    _decode_vs_oracle(pkg, orc, "Irr345_N60_M24_GF256", 20, 40, 0.3, 8, 2.0, 300)          # more frames than one wave of workgroups
    code = pkg.Code(os.path.join(MAT, "Irr345_N60_M24_GF256"))
    dec = pkg.Decoder(code, 20, 40, 0.3, 8, max_batch=2)
    with pytest.raises(pkg.EmsError):                       # the older generation's flavour is built for regular codes
        dec.v2_load_likelihood(np.zeros((1, code.N, 256), np.float32), np.zeros((1, code.N, 256), np.int32))
    dec.close()


def test_decode_messages_longer_than_32_entries_vs_oracle(pkg, orc):
    """nm up to 40 (the reference caps nm nowhere; kat_traces.json N40 / N36 are its own traces at nm = 40 and 36): beyond the 32
    entries the merge form and the threshold selection hold, the literal forms run -- same bits."""
    _decode_vs_oracle(pkg, orc, "N96_K48_GF64", 40, 60, 0.3, 10, 2.0, 24)
    _decode_vs_oracle(pkg, orc, "N96_K48_GF64", 33, 66, 0.5, 6, 1.0, 8, early_exit=False)
    _decode_vs_oracle(pkg, orc, "KN_N128_K64_GF256.txt", 40, 80, 0.3, 8, 2.5, 12)
    _decode_vs_oracle(pkg, orc, "Irr345_N60_M24_GF256", 36, 72, 0.3, 8, 3.0, 8)
    _decode_vs_oracle(pkg, orc, "N96_K48_GF64", 64, 128, 0.3, 10, 2.0, 16)                        # nm = GF (EMS_MAX_NM = 64 since round 4)
    _decode_vs_oracle(pkg, orc, "KN_N128_K64_GF256.txt", 56, 70, 0.3, 8, 2.5, 8)
    _decode_vs_oracle(pkg, orc, "Wide_N80_M8_GF256", 48, 96, 0.5, 5, 4.0, 4)                      # ... with check degrees above 16


def test_decode_generic_kernels_other_degrees(pkg, orc):
    """Codes outside the specialised (dc, nm) instantiations: dc = 6, 8 and 12 over GF(64) (rows of 8 and 4
    lanes), GF(256) with dc = 4 at a non-default nm, GF(256) with dc = 8."""
    _decode_vs_oracle(pkg, orc, "Mat26_N48_M16", 12, 24, 0.5, 12, 3.5, 24)            # dc = 6
    _decode_vs_oracle(pkg, orc, "Mat28_N72_M18", 16, 32, 0.3, 10, 4.0, 16)            # dc = 8
    _decode_vs_oracle(pkg, orc, "Mat212_N96_M16", 12, 24, 0.3, 8, 5.0, 12)            # dc = 12
    _decode_vs_oracle(pkg, orc, "KN_N576_K480_GF64.txt", 30, 45, 0.3, 10, 4.0, 6)     # start.sh:8-13 parameters
    _decode_vs_oracle(pkg, orc, "KN_N128_K64_GF256.txt", 24, 40, 0.3, 15, 2.5, 32)    # GF(256), dc = 4, nm = 24
    _decode_vs_oracle(pkg, orc, "KN_N128_K64_GF256.txt", 20, 22, 0.3, 15, 2.5, 8)     # NbOper < nm + 3: generic kernel


def test_decode_random_parameter_sweep(pkg, orc):
    """Seeded random (nm, NbOper, offset, NbIterMax, Eb/N0, early exit) over the small codes of every kernel family:
    whole-frame state against the oracle each time (fast paths, literal fallbacks, short lists, early stops)."""
    rs = np.random.RandomState(2026)
    codes = [("N96_K48_GF64", 64), ("Mat24_N192_M96", 64), ("Mat26_N48_M16", 64), ("Mat28_N72_M18", 64), ("Mat212_N96_M16", 64),
             ("KN_N128_K64_GF256.txt", 256), ("KN_N576_K480_GF64.txt", 64)]
    for trial in range(28):
        name, q = codes[trial % len(codes)]
        nm = int(rs.choice([5, 6, 8, 12, 16, 20, 24, 30, 32]))
        nm = min(nm, q)
        nop = int(rs.choice([max(1, nm - 2), nm, nm + 1, nm + 3, 2 * nm, 3 * nm, 70]))
        off = float(rs.choice([0.0, 0.3, 0.5, 1.0]))
        itmax = int(rs.randint(2, 9))
        ebn = float(rs.choice([-1.0, 1.0, 2.5, 4.0, 6.0]))
        _decode_vs_oracle(pkg, orc, name, nm, nop, off, itmax, ebn, 3, state=True, early_exit=bool(trial % 3))


def test_decode_gf256_dc8_vs_oracle(pkg, orc):
    _decode_vs_oracle(pkg, orc, "Ahmed_64800_R34_GF256", 20, 40, 0.3, 12, 3.2, 1, state=True)


def test_decode_gf256_dc12_vs_oracle(pkg, orc):
    """Rate 5/6 GF(256) code of the reference's matrices/ (dc = 12): rows of 4 lanes with 64 symbols per lane."""
    _decode_vs_oracle(pkg, orc, "Ahmed_64800_R56_GF256", 20, 40, 0.3, 6, 4.6, 1, state=True)


def test_decode_check_degrees_above_16_vs_oracle(pkg, orc):
    """Check degrees above 16 (k_decode_wide: the edges of a visit in passes of 64 / LPE rows, Mvc recomputed for the output pass): the
    reference caps the degree nowhere (init.c:320, bubble_decoder.c:166-227).  Small irregular codes with degrees 17..24 over GF(64) /
    GF(256) and 25..40 over GF(16) (tests/golden/make_wide_matrix.py; kat_traces.json W1..W3 are the REAL executable's traces on them):
    whole-frame state -- decisions, iterations, syndrome, bit errors, APP, every C->V row -- against the oracle, fast and literal forms."""
    _decode_vs_oracle(pkg, orc, "Wide_N80_M8_GF64", 16, 32, 0.3, 10, 4.5, 24)
    _decode_vs_oracle(pkg, orc, "Wide_N80_M8_GF64", 12, 15, 1.0, 6, 3.0, 8, early_exit=False)      # NbOper < nm + 3
    _decode_vs_oracle(pkg, orc, "Wide_N80_M8_GF256", 20, 40, 0.3, 12, 4.5, 16)
    _decode_vs_oracle(pkg, orc, "Wide_N80_M8_GF256", 36, 72, 0.3, 6, 4.0, 6)                       # literal forms (nm > 32)
    _decode_vs_oracle(pkg, orc, "Wide_N120_M6_GF16", 8, 16, 0.3, 10, 5.0, 16)                      # three passes, first-occurrence form
    _decode_vs_oracle(pkg, orc, "Wide_N120_M6_GF16", 12, 24, 0.5, 8, 4.0, 8, early_exit=False)


def test_decode_repeated_variable_nodes_vs_oracle(pkg, orc):
    """Check nodes that list one variable node twice (tests/golden/make_repeat_matrix.py; kat_traces.json R1..R3 are the REAL executable's
    traces): both edges read the same APP row (NB_LDPC.c:329-337), the later edge's APP update and decision stay (:444-450).  The generic
    one-visit kernel (dc = 4) and k_decode_wide (dc = 18: repeats inside a pass and across the two), whole-frame state against the oracle,
    fast and literal forms, early exit on and off."""
    it = _decode_vs_oracle(pkg, orc, "Rep_N40_M20_GF64", 12, 24, 0.3, 10, 3.0, 24)
    _decode_vs_oracle(pkg, orc, "Rep_N40_M20_GF64", 16, 32, 0.5, 6, 6.0, 16, early_exit=False)
    _decode_vs_oracle(pkg, orc, "Rep_N40_M20_GF64", 40, 60, 0.3, 5, 4.0, 4)                          # literal forms (nm > 32)
    _decode_vs_oracle(pkg, orc, "RepWide_N54_M6_GF64", 16, 32, 0.3, 8, 5.0, 24)
    _decode_vs_oracle(pkg, orc, "RepWide_N54_M6_GF64", 12, 15, 1.0, 6, 3.0, 5, early_exit=False)
    assert it is not None


def test_decode_gf256_dc20_vs_oracle(pkg, orc):
    """The reference's rate-9/10 GF(256) matrix (Ahmed_64800_R910_GF256, dc = 20; kat_traces.json D20 / D20b from the real executable):
    a frame that converges and one that does not, whole state."""
    it = _decode_vs_oracle(pkg, orc, "Ahmed_64800_R910_GF256", 20, 40, 0.3, 30, 4.0, 2, state=True)
    assert list(it) == [6, 6]                                                                      # D20b: 6.00, 6.00, 6.67
    _decode_vs_oracle(pkg, orc, "Ahmed_64800_R910_GF256", 20, 40, 0.3, 4, 3.7, 1, state=True)      # D20's first frame, first sweeps


def test_decode_gf256_dc10_vs_oracle(pkg, orc):
    """dc = 10 over GF(256) (Ahmed_64800_R45_GF256; the dc 9..16 path was tested at dc = 12 only); traces D10 / D10b."""
    _decode_vs_oracle(pkg, orc, "Ahmed_64800_R45_GF256", 20, 40, 0.3, 10, 3.0, 1, state=True)


def test_decode_full_alist_code_vs_oracle(pkg, orc):
    """A code that only exists in the full-alist layout (rate 3/4, GF(256), dc = 8): loader + generic dc <= 8 kernel."""
    _decode_vs_oracle(pkg, orc, "KN_N64800_K48600_GF256_fullalist.txt", 20, 40, 0.3, 8, 3.4, 1, state=True)


def test_isolated_variable_node_is_decided_from_its_channel_llr(pkg, orc, tmp_path):
    """A column of degree 0: no check-node visit writes its decision; Decision (tools.c:312) reads its channel APP."""
    src = open(os.path.join(MAT, "N96_K48_GF64")).read().split()
    N, M, q = map(int, src[:3])
    T = src[3:]
    p = tmp_path / "iso.txt"
    p.write_text("%d %d %d\n%s\n%s\n%s\n" % (N + 1, M, q, " ".join(T[:N] + ["0"]), " ".join(T[N:N + M]), " ".join(T[N + M:])))
    o = orc.Oracle(str(p))
    sigma = o.sigma(2.0)
    rng = orc.Rng(0)
    frames = [o.draw_frame(rng, sigma) for _ in range(6)]
    noisy = np.stack([f[1] for f in frames])
    dec = pkg.Decoder(pkg.Code(str(p)), 16, 32, 0.3, 10, max_batch=6)
    decide, iters, synd = dec.decode_batch(noisy, sigma)
    for b in range(6):
        ref = o.decode(o.llr(noisy[b], sigma), 16, 32, 0.3, 10)
        assert (decide[b] == ref["decide"]).all() and iters[b] == ref["iters"] and synd[b] == ref["synd"]
    dec.close()


def test_set_params_switches_kernels_and_keeps_the_record_layout(pkg, orc, monkeypatch):
    """ems_set_params after ems_create: lowering nm (and NbOper, offset) takes the handle off the nm = 20 two-visit build onto
    the generic kernel; records stay strided by the create-time nm while their fields follow the current one, so APP, the
    expanded C->V rows (ems_get_ctov) and the decisions must still equal the oracle's; raising nm back returns to the
    two-visit build."""
    monkeypatch.setenv("EMS_DUAL", "2")
    path = os.path.join(MAT, "KN_N128_K64_GF256.txt")
    o = orc.Oracle(path)
    sigma = o.sigma(2.5)
    rng = orc.Rng(0)
    noisy = np.stack([o.draw_frame(rng, sigma)[1] for _ in range(6)])
    dec = pkg.Decoder(pkg.Code(path), 20, 40, 0.3, 8, max_batch=6)
    for nm, nop, off in ((20, 40, 0.3), (12, 20, 0.5), (20, 64, 0.0), (5, 9, 1.0), (20, 23, 0.3)):
        dec.set_params(nm=nm, nbOper=nop, offset=off)
        d, it, sy = dec.decode_batch(noisy, sigma)
        for b in range(6):
            ref = o.decode(o.llr(noisy[b], sigma), nm, nop, off, 8, want_state=True)
            assert (d[b] == ref["decide"]).all() and it[b] == ref["iters"] and sy[b] == ref["synd"], (nm, b)
            assert dec.app(b).tobytes() == ref["app"].tobytes() and dec.ctov(b).tobytes() == ref["ctov"].tobytes(), (nm, b)
    with pytest.raises(pkg.EmsError):
        dec.set_params(nm=24)                                   # above the create-time value
    dec.close()


def test_unsupported_configurations_fail_loudly(pkg):
    code = pkg.Code(os.path.join(MAT, "Mat212_N96_M16"))
    with pytest.raises(pkg.EmsError) as e:
        pkg.Decoder(code, 65, 80, 0.3, 10, max_batch=1)          # nm > 64 (EMS_MAX_NM)
    assert e.value.code == pkg.EMS_EINVAL


def test_decode_gf16_vs_oracle(pkg, orc):
    it = _decode_vs_oracle(pkg, orc, "Ahmed_64800_R34_GF16", 8, 16, 0.3, 30, 3.0, 4)
    assert list(it) == [6, 5, 5, 5]                    # K3


def test_decode_gf16_one_visit_per_wave_kernel_vs_oracle(pkg, orc, monkeypatch):
    """dc = 8 codes take the two-visits-per-wave kernel by default; EMS_DUAL=0 keeps the one-visit kernel reachable."""
    monkeypatch.setenv("EMS_DUAL", "0")
    it = _decode_vs_oracle(pkg, orc, "Ahmed_64800_R34_GF16", 8, 16, 0.3, 30, 3.0, 4, state=True)
    assert list(it) == [6, 5, 5, 5]


def test_decode_gf16_two_visits_per_wave_kernel_vs_oracle(pkg, orc, monkeypatch):
    """Round 2/3's GF(16) kernel (k_decode_dual<16, 8, 8>) stays reachable (EMS_DUAL=<waves> / tuning.dual): full state, the NbOper cut."""
    monkeypatch.setenv("EMS_DUAL", "16")
    _decode_vs_oracle(pkg, orc, "Ahmed_64800_R34_GF16", 8, 16, 0.3, 30, 3.0, 2, state=True)
    _decode_vs_oracle(pkg, orc, "Ahmed_64800_R34_GF16", 8, 16, 0.3, 6, 0.0, 2, state=True, early_exit=False)
    for nop in (11, 13, 15, 24):
        _decode_vs_oracle(pkg, orc, "Ahmed_64800_R34_GF16", 8, nop, 0.3, 5, 0.0, 1, state=True, early_exit=False)


@pytest.mark.parametrize("trav", ["0", "1"])
def test_decode_gf16_state_and_fixed_work(pkg, orc, monkeypatch, trav):
    """The GF(16) configuration on its default kernel (round 4: the lane-serial k_decode_q16t; the closed forms of ems_q16t_lane.h with the
    merge-form redo of the check nodes they do not decide): full APP / C->V state, and a non-converging fixed-work run.  Round 5: also on the
    opt-in travelling layout (per edge a record and a slot for the APP row of its node; the state comes back through k_q16t_export)."""
    monkeypatch.setenv("EMS_Q16T_TRAV", trav)
    _decode_vs_oracle(pkg, orc, "Ahmed_64800_R34_GF16", 8, 16, 0.3, 30, 3.0, 2, state=True)
    _decode_vs_oracle(pkg, orc, "Ahmed_64800_R34_GF16", 8, 16, 0.3, 6, 0.0, 2, state=True, early_exit=False)
    # NbOper = nm + 3 (the scan is cut short in one task of eight: the closed form of that cut in the 8-lane stages) and 3 nm (never)
    _decode_vs_oracle(pkg, orc, "Ahmed_64800_R34_GF16", 8, 11, 0.3, 5, 0.0, 2, state=True, early_exit=False)
    _decode_vs_oracle(pkg, orc, "Ahmed_64800_R34_GF16", 8, 24, 0.3, 4, 0.0, 2, state=True, early_exit=False)


@pytest.mark.parametrize("nop", [12, 13, 14, 15, 17, 20, 32])
def test_decode_gf16_nboper_sweep_vs_oracle(pkg, orc, nop):
    """The GF(16) default kernel under every NbOper around the message size: the closed form of the scan cut by NbOper (the prefix of the
    rank order with at most NbOper cells; q16t::es_lane) against the oracle's literal loop -- one non-converging frame, four sweeps,
    full APP / C->V state."""
    _decode_vs_oracle(pkg, orc, "Ahmed_64800_R34_GF16", 8, nop, 0.3, 5, 0.0, 1, state=True, early_exit=False)


def test_decode_gf256_vs_oracle(pkg, orc):
    it = _decode_vs_oracle(pkg, orc, "Ahmed_64800_R12_GF256", 20, 40, 0.3, 30, 1.6, 4)
    assert list(it) == [30, 30, 30, 25]                # K4: frame 4 converges in 25 sweeps


def test_trajectory_n96_golden(pkg, orc):
    """Sweep-by-sweep APP of the reference-derived K1 trajectories: run the GPU for s sweeps, compare."""
    g = gold("traj_n96.npz")
    code = pkg.Code(os.path.join(MAT, "KN_N96_K48_GF64.txt"))
    for f in range(3):
        apps, decs, synds = g["f%d_app" % f], g["f%d_dec" % f], g["f%d_synd" % f]
        for s in range(1, len(synds) + 1):
            dec = pkg.Decoder(code, 16, 32, 0.3, s + 1, max_batch=1, early_exit=False)
            dec.load_llr(g["f%d_llr" % f][None])
            dec.decode(1)
            d, _it, sy = dec.results(1)
            assert dec.app(0).tobytes() == apps[s - 1].tobytes()
            assert (d[0] == decs[s - 1]).all() and sy[0] == synds[s - 1]
            dec.close()


@pytest.mark.parametrize("tag,mat,nm,nop,sweeps", [("gf16", "Ahmed_64800_R34_GF16", 8, 16, (1, 3, 6)),
                                                    ("gf256", "Ahmed_64800_R12_GF256", 20, 40, (1, 2, 10, 29))])
def test_trajectory_big_golden(pkg, orc, tag, mat, nm, nop, sweeps):
    g = gold("traj_big.npz")
    path = os.path.join(MAT, mat)
    o = orc.Oracle(path)
    sigma = o.sigma(float(g[tag + "_ebn"]))
    _cw, noisy = o.draw_frame(orc.Rng(0), sigma)
    code = pkg.Code(path)
    for s in sweeps:
        if s > len(g[tag + "_crc"]):
            continue
        dec = pkg.Decoder(code, nm, nop, 0.3, s + 1, max_batch=1, early_exit=False)
        d, _it, sy = dec.decode_batch(noisy[None], sigma)
        assert orc.crc32(dec.app(0)) == int(g[tag + "_crc"][s - 1])
        assert (d[0].astype(np.uint8) == g[tag + "_dec"][s - 1]).all() and sy[0] == int(g[tag + "_synd"][s - 1])
        dec.close()


# ---------------------------------------------------------------------------------------------------
# edge cases the domain offers
# ---------------------------------------------------------------------------------------------------
def test_noiseless_frames_massive_ties(pkg, orc):
    """y = +-1 exactly: only logGF+1 distinct LLR values per node (huge tie groups in the top-nm
    selection, resolved by lowest symbol), decode must return the codeword after one sweep."""
    for name, nm in (("N96_K48_GF64", 16), ("Ahmed_64800_R12_GF256", 20)):
        path = os.path.join(MAT, name)
        o = orc.Oracle(path)
        rng = orc.Rng(0)
        cw, _ = o.draw_frame(rng, 0.5)
        bits = (o.idx2bin[cw][:, None] >> np.arange(o.logq)) & 1
        noisy = (1.0 - 2.0 * bits).astype(np.float32)
        dec = pkg.Decoder(pkg.Code(path), nm, 2 * nm, 0.3, 10, max_batch=1)
        d, it, sy = dec.decode_batch(noisy[None], 0.7)
        ref = o.decode(o.llr(noisy, 0.7), nm, 2 * nm, 0.3, 10, want_state=True)
        assert (d[0] == cw).all() and it[0] == 1 and sy[0] == 0
        assert (d[0] == ref["decide"]).all() and dec.app(0).tobytes() == ref["app"].tobytes()
        dec.close()


def test_saturated_and_negative_llrs(pkg, orc):
    """LLR rows containing values >= 1e5 (never selectable, NB_LDPC.c:362) and negative values."""
    path = os.path.join(MAT, "N96_K48_GF64")
    o = orc.Oracle(path)
    rs = np.random.RandomState(11)
    llr = (rs.rand(4, o.N, o.q) * 40).astype(np.float32)
    llr[0, :, 20:] = 1e5            # fewer than nm selectable symbols
    llr[1] -= 15.0                  # negative entries
    llr[2, 3, :] = 2e5              # a node with nothing selectable
    llr[3] = np.round(llr[3])       # ties
    dec = pkg.Decoder(pkg.Code(path), 16, 32, 0.3, 6, max_batch=4)
    dec.load_llr(llr)
    dec.decode(4)
    d, it, sy = dec.results(4)
    for b in range(4):
        ref = o.decode(llr[b], 16, 32, 0.3, 6, want_state=True)
        assert (d[b] == ref["decide"]).all() and it[b] == ref["iters"] and sy[b] == ref["synd"]
        assert dec.app(b).tobytes() == ref["app"].tobytes()
    dec.close()


def test_gf64_network_selection_edge_inputs(pkg, orc, monkeypatch):
    """GF(64) / nm = 16 on the two-visit and two-frame kernels, whose top-16 selection is a sorting network over 32-bit keys (value bits
    with the symbol in the low six): rows it must hand to the threshold / literal selection -- values within 64 ulp of each other
    around the 16th place and inside the top 16, exact ties, negative values, values >= 1e5, fewer than nm selectable symbols --
    next to continuous rows it decides alone.  Whole-frame state against the oracle."""
    path = os.path.join(MAT, "N96_K48_GF64")
    o = orc.Oracle(path)
    rs = np.random.RandomState(23)
    llr = (rs.rand(8, o.N, o.q) * 40).astype(np.float32)
    one = np.float32(5.0)
    ulp = np.spacing(one)
    llr[0, :, :24] = one + ulp * rs.randint(0, 200, size=(o.N, 24)).astype(np.float32)    # a cluster a few ulp wide holds the whole top 16
    llr[1, :, 10:20] = llr[1, :, 10:11] + ulp * np.arange(10, dtype=np.float32)          # ten values one ulp apart, somewhere in the order
    srt = np.sort(llr[2], axis=1)
    llr[2] = np.where(llr[2] == srt[:, 16:17], srt[:, 15:16] + np.spacing(srt[:, 15:16]), llr[2])   # the 17th one ulp above the 16th
    llr[3] = np.round(llr[3])                                                            # exact ties
    llr[4] -= 15.0                                                                       # negative entries
    llr[5, :, 20:] = 1e5                                                                 # fewer than nm selectable symbols
    llr[5, 3, :] = 2e5                                                                   # ... and a node with none
    llr[6, ::3] = np.round(llr[6, ::3] * 4) / 4                                          # every third node on a grid, the rest continuous
    # (frame 7: continuous)
    refs = [o.decode(llr[b], 16, 32, 0.3, 6, want_state=True) for b in range(8)]
    for knob, val in (("EMS_DUAL", "1"), ("EMS_DUAL", "2"), ("EMS_TWIN", "1")):
        monkeypatch.setenv(knob, val)
        dec = pkg.Decoder(pkg.Code(path), 16, 32, 0.3, 6, max_batch=8)
        dec.load_llr(llr)
        dec.decode(8)
        d, it, sy = dec.results(8)
        assert ("dual" in dec.last_kernel or "twin" in dec.last_kernel), dec.last_kernel
        for b in range(8):
            ref = refs[b]
            assert (d[b] == ref["decide"]).all() and it[b] == ref["iters"] and sy[b] == ref["synd"], (knob, val, b)
            assert dec.app(b).tobytes() == ref["app"].tobytes(), (knob, val, b)
            assert dec.ctov(b).tobytes() == ref["ctov"].tobytes(), (knob, val, b)
        dec.close()
        monkeypatch.delenv(knob)


def test_gf16_coarse_llrs_take_the_merge_form(pkg, orc, monkeypatch):
    """GF(16) / nm = 8 on the two-visits-per-wave kernel with LLRs on a coarse grid: equal first occurrences in nearly every elementary
    step, i.e. the stage leaves the first-occurrence loop and the rest of the visit runs in the merge form through the out-of-line
    call (5 wave-stages in 100 000 on the bench workload, nearly all of them here); half-integer and integer grids, full state."""
    path = os.path.join(MAT, "Ahmed_64800_R34_GF16")
    o = orc.Oracle(path)
    rs = np.random.RandomState(5)
    llr = (rs.rand(3, o.N, o.q) * 14).astype(np.float32)
    llr[0] = np.round(llr[0])                       # integers: ties everywhere
    llr[1] = np.round(llr[1] * 2) / 2               # half-integers
    llr[2, ::2] = np.round(llr[2, ::2])             # every other node coarse, the rest continuous
    # the default kernel since round 4 is the lane-serial one (k_decode_q16t): on these inputs nearly every check node reports a tie its
    # 32-bit keys do not decide and is redone by the merge-form visit (q16t_redo_cn) -- the fallback under load; then the two-visit kernel
    # (round 5: also on its opt-in travelling layout, where the redo runs on a staging block: q16t_redo_trav)
    for forced, kernel in ((None, "k_decode_q16t<true, false>"), ("trav", "k_decode_q16t<true, true>"), (16, "k_decode_dual<16, 8, 8>")):
        if forced == "trav":
            monkeypatch.setenv("EMS_Q16T_TRAV", "1")
            forced = None
        else:
            monkeypatch.delenv("EMS_Q16T_TRAV", raising=False)
        dec = pkg.Decoder(pkg.Code(path), 8, 16, 0.3, 4, max_batch=3, early_exit=False)
        if forced is not None:
            dec.set_tuning(dual=forced)
        dec.load_llr(llr)
        dec.decode(3)
        assert dec.last_kernel == kernel
        d, it, sy = dec.results(3)
        for b in range(3):
            ref = o.decode(llr[b], 8, 16, 0.3, 4, early_exit=False, want_state=True)
            assert (d[b] == ref["decide"]).all() and it[b] == ref["iters"] and sy[b] == ref["synd"]
            assert dec.app(b).tobytes() == ref["app"].tobytes() and dec.ctov(b).tobytes() == ref["ctov"].tobytes()
        dec.close()


def test_batch_position_invariance_and_repeatability(pkg, orc):
    path = os.path.join(MAT, "N96_K48_GF64")
    o = orc.Oracle(path)
    sigma = o.sigma(1.5)
    rng = orc.Rng(0)
    noisy = np.stack([o.draw_frame(rng, sigma)[1] for _ in range(40)])
    dec = pkg.Decoder(pkg.Code(path), 16, 32, 0.3, 10, max_batch=40)
    a = dec.decode_batch(noisy, sigma)
    perm = np.random.RandomState(0).permutation(40)
    b = dec.decode_batch(noisy[perm], sigma)
    c = dec.decode_batch(noisy[:7], sigma)            # ragged: fewer frames than max_batch
    for x, y in zip(a, b):
        assert (x[perm] == y).all()
    for x, y in zip(a, c):
        assert (x[:7] == y).all()
    dec.close()


# ---------------------------------------------------------------------------------------------------
# BASELINE.json full sizes: size-independent properties
# ---------------------------------------------------------------------------------------------------
def test_full_size_headline_batch256_properties(pkg, orc):
    """Ahmed_64800_R12_GF256, nm=20, 30 iterations, batch 256: encode -> AWGN -> decode round trip at
    high SNR recovers every codeword (zero syndrome, zero bit errors); duplicated frames decode
    identically; two of the frames are checked in full against the oracle."""
    path = os.path.join(MAT, "Ahmed_64800_R12_GF256")
    o = orc.Oracle(path)
    sigma = o.sigma(3.0)
    rng = orc.Rng(0)
    base = [o.draw_frame(rng, sigma) for _ in range(8)]
    idx = np.arange(256) % 8
    noisy = np.stack([base[i][1] for i in idx])
    cws = np.stack([base[i][0] for i in idx])
    dec = pkg.Decoder(pkg.Code(path), 20, 40, 0.3, 30, max_batch=256)
    d, it, sy = dec.decode_batch(noisy, sigma)
    assert (sy == 0).all() and (d == cws).all() and (dec.count_errors(cws) == 0).all()
    for b in range(8, 256):
        assert it[b] == it[b % 8]
    for b in (0, 5):
        ref = o.decode(o.llr(noisy[b], sigma), 20, 40, 0.3, 30)
        assert it[b] == ref["iters"] and (d[b] == ref["decide"]).all()
    dec.close()


@pytest.mark.parametrize("waves,occ4", [(4, False), (4, True), (2, False)])
def test_decode_gf256_small_workgroup_shapes_vs_oracle(pkg, orc, monkeypatch, waves, occ4):
    """The shapes the library picks for batches of several frames per CU (4-wave workgroups, and the
    128-VGPR build of the kernel for four of them per CU), forced here on a batch the oracle can follow."""
    monkeypatch.setenv("EMS_WAVES", str(waves))
    if occ4:
        monkeypatch.setenv("EMS_OCC4", "1")
    it = _decode_vs_oracle(pkg, orc, "Ahmed_64800_R12_GF256", 20, 40, 0.3, 30, 1.6, 4)
    assert list(it) == [30, 30, 30, 25]


@pytest.mark.parametrize("waves", [2, 4, 1])
def test_decode_gf256_two_visits_per_wave_vs_oracle(pkg, orc, monkeypatch, waves):
    """k_decode_dual (two check nodes per wave, the shape picked for >= 5 frames per CU), forced on a batch the oracle
    can follow: hard decisions, iterations, syndromes, the whole APP block and every C->V row, bit for bit."""
    monkeypatch.setenv("EMS_DUAL", str(waves))
    it = _decode_vs_oracle(pkg, orc, "Ahmed_64800_R12_GF256", 20, 40, 0.3, 30, 1.6, 4, state=True)
    assert list(it) == [30, 30, 30, 25]
    _decode_vs_oracle(pkg, orc, "Ahmed_64800_R12_GF256", 20, 40, 0.3, 5, 0.0, 2, state=True, early_exit=False)


def test_two_visits_per_wave_kernel_on_other_dc4_gf256_inputs(pkg, orc, monkeypatch):
    """The literal fall-backs inside the dual kernel: noiseless frames (huge tie groups -> select_rounds) and saturated /
    negative LLRs (the guard that sends a visit to the literal bubble loop), against the oracle."""
    monkeypatch.setenv("EMS_DUAL", "2")
    path = os.path.join(MAT, "Ahmed_64800_R12_GF256")
    o = orc.Oracle(path)
    rs = np.random.RandomState(5)
    llr = np.zeros((3, o.N, 256), np.float32)
    llr[0] = o.llr(o.draw_frame(orc.Rng(7), 0.0)[1], 0.5)                        # noiseless: few distinct values per row
    llr[1] = (rs.randint(0, 4, size=(o.N, 256)) * 2.5).astype(np.float32)       # massive ties
    llr[2] = rs.uniform(-50.0, 2.0e4, size=(o.N, 256)).astype(np.float32)       # negative and > 1e4 values
    llr[2, 3, :] = 2e5                                                           # a node with nothing selectable
    llr[2, 9, 12:] = 1e5                                                         # fewer than nm selectable symbols
    dec = pkg.Decoder(pkg.Code(path), 20, 40, 0.3, 4, max_batch=3, early_exit=False)
    dec.load_llr(llr)
    dec.decode(3)
    d, it, sy = dec.results(3)
    for b in range(3):
        ref = o.decode(llr[b], 20, 40, 0.3, 4, early_exit=False, want_state=True)
        assert (d[b] == ref["decide"]).all() and sy[b] == ref["synd"] and it[b] == ref["iters"], "frame %d" % b
        assert dec.app(b).tobytes() == ref["app"].tobytes(), "frame %d: APP differs" % b
    dec.close()


@pytest.mark.parametrize("B", [1536, 700])
def test_full_size_headline_dual_shapes_agree(pkg, orc, B):
    """The bench's batch (1536 frames = 6 per CU, fixed work at 1.0 dB) takes the two-visits-per-wave kernel in 2-wave
    workgroups, 700 frames (3 per CU) the same kernel in 4-wave workgroups: frame for frame they must give what the
    one-frame-per-CU launch gives, and the oracle's result on a sampled frame."""
    path = os.path.join(MAT, "Ahmed_64800_R12_GF256")
    o = orc.Oracle(path)
    sigma = o.sigma(1.0)
    rng = orc.Rng(0)
    base = [o.draw_frame(rng, sigma) for _ in range(8)]
    idx = np.arange(B) % 8
    noisy = np.stack([base[i][1] for i in idx])
    code = pkg.Code(path)
    big = pkg.Decoder(code, 20, 40, 0.3, 6, max_batch=B, early_exit=False)
    d, it, sy = big.decode_batch(noisy, sigma)
    small = pkg.Decoder(code, 20, 40, 0.3, 6, max_batch=8, early_exit=False)
    d8, it8, sy8 = small.decode_batch(noisy[:8], sigma)
    for b in range(B):
        assert (d[b] == d8[b % 8]).all() and sy[b] == sy8[b % 8] and it[b] == it8[b % 8]
    for b in (0, 5, 517, B - 1):
        assert big.app(b).tobytes() == small.app(b % 8).tobytes()
        assert big.ctov(b).tobytes() == small.ctov(b % 8).tobytes()
    ref = o.decode(o.llr(noisy[3], sigma), 20, 40, 0.3, 6, early_exit=False, want_state=True)
    assert big.app(3).tobytes() == ref["app"].tobytes() and (d[3] == ref["decide"]).all()
    big.close(); small.close()


def test_full_size_headline_batch1024_shapes_agree(pkg, orc):
    """BASELINE batch of the bench (1024 frames = 4 per CU, fixed work at 1.0 dB): the 128-VGPR 4-wave
    launch must give, frame for frame, what the one-frame-per-CU launch gives (hard decisions, syndromes,
    APP and C->V state of sampled frames); 8 distinct frames, so copies must also agree among themselves."""
    path = os.path.join(MAT, "Ahmed_64800_R12_GF256")
    o = orc.Oracle(path)
    sigma = o.sigma(1.0)
    rng = orc.Rng(0)
    base = [o.draw_frame(rng, sigma) for _ in range(8)]
    idx = np.arange(1024) % 8
    noisy = np.stack([base[i][1] for i in idx])
    code = pkg.Code(path)
    big = pkg.Decoder(code, 20, 40, 0.3, 6, max_batch=1024, early_exit=False)
    d, it, sy = big.decode_batch(noisy, sigma)
    small = pkg.Decoder(code, 20, 40, 0.3, 6, max_batch=8, early_exit=False)
    d8, it8, sy8 = small.decode_batch(noisy[:8], sigma)
    for b in range(1024):
        assert (d[b] == d8[b % 8]).all() and sy[b] == sy8[b % 8] and it[b] == it8[b % 8]
    for b in (0, 5, 517, 1023):
        assert big.app(b).tobytes() == small.app(b % 8).tobytes()
        assert big.ctov(b).tobytes() == small.ctov(b % 8).tobytes()
    ref = o.decode(o.llr(noisy[3], sigma), 20, 40, 0.3, 6, early_exit=False, want_state=True)
    assert big.app(3).tobytes() == ref["app"].tobytes() and (d[3] == ref["decide"]).all()
    big.close(); small.close()


def test_decode_n96_two_visits_per_wave_vs_oracle(pkg, orc, monkeypatch):
    """The GF(64) build of k_decode_dual (picked for the N96 code from 4 frames per CU on), forced on a small batch:
    one-wave and two-wave workgroups, full state."""
    for waves in ("1", "2"):
        monkeypatch.setenv("EMS_DUAL", waves)
        _decode_vs_oracle(pkg, orc, "N96_K48_GF64", 16, 32, 0.3, 10, 2.0, 48, state=True)
        _decode_vs_oracle(pkg, orc, "KN_N96_K48_GF64.txt", 16, 32, 0.3, 30, -3.0, 8, state=True, early_exit=False)


def test_decode_n96_two_frames_per_wave_vs_oracle(pkg, orc, monkeypatch):
    """k_decode_twin (one wave = two frames, check nodes in reference order; picked by itself from 4 frames per CU on for codes
    whose pairs would be mostly singles): forced here at small batches -- even and odd, frames that stop at different sweeps
    sharing a wave, fixed work, the other on-disk layout -- with the full state against the oracle."""
    monkeypatch.setenv("EMS_TWIN", "1")
    it = _decode_vs_oracle(pkg, orc, "N96_K48_GF64", 16, 32, 0.3, 10, 2.0, 96, state=True)
    assert len(set(it.tolist())) > 2
    _decode_vs_oracle(pkg, orc, "N96_K48_GF64", 16, 32, 0.3, 30, 1.0, 47, state=True)                       # odd batch: the last wave has one frame
    _decode_vs_oracle(pkg, orc, "KN_N96_K48_GF64.txt", 16, 32, 0.3, 30, -3.0, 9, state=True, early_exit=False)
    _decode_vs_oracle(pkg, orc, "N96_K48_GF64", 16, 19, 1.0, 12, 3.0, 1, state=True)                       # one frame, NbOper = nm + 3
    monkeypatch.setenv("EMS_TWIN", "0")
    _decode_vs_oracle(pkg, orc, "N96_K48_GF64", 16, 32, 0.3, 10, 2.0, 8, state=True)                       # (and the knob that forbids it)


def test_two_visit_kernels_random_parameter_sweep(pkg, orc, monkeypatch):
    """Seeded random (NbOper, offset, NbIterMax, Eb/N0, early exit, workgroup shape) through the two-visits-per-wave builds on
    small codes of their three families -- GF(256) dc 4 nm 20, GF(64) dc 4 nm 16, and (one frame of the long code) GF(16) dc 8
    nm 8: whole-frame state against the oracle each time.  NbOper from nm+3 (first pass only) to 64 (long continuations),
    low SNR for short outputs and ties, codes with single (unpaired) check nodes."""
    rs = np.random.RandomState(77)
    fam = [("KN_N128_K64_GF256.txt", 20), ("Mat24_N192_M96", 16), ("N96_K48_GF64", 16), ("KN_N96_K48_GF64.txt", 16), ("Mat24_N192_M96", 24)]
    for trial in range(30):
        name, nm = fam[trial % len(fam)]
        nop = min(64, int(rs.choice([nm + 3, nm + 4, nm + 8, 2 * nm, 3 * nm, 64])))
        off = float(rs.choice([0.0, 0.3, 0.7, 1.0]))
        itmax = int(rs.randint(2, 13))
        ebn = float(rs.choice([-2.0, 0.5, 2.0, 3.5, 6.0]))
        monkeypatch.setenv("EMS_DUAL", str(int(rs.choice([1, 2, 4]))))
        _decode_vs_oracle(pkg, orc, name, nm, nop, off, itmax, ebn, 5, state=True, early_exit=bool(trial % 3))
    for nop, off, itmax, ebn, ee in ((11, 0.3, 5, 1.0, False), (16, 0.0, 7, 2.6, True), (40, 1.0, 4, 0.0, False)):
        monkeypatch.setenv("EMS_DUAL", "4")
        _decode_vs_oracle(pkg, orc, "Ahmed_64800_R34_GF16", 8, nop, off, itmax, ebn, 1, state=True, early_exit=ee)


def test_full_size_n96_batch8192(pkg, orc):
    path = os.path.join(MAT, "N96_K48_GF64")
    o = orc.Oracle(path)
    sigma = o.sigma(2.0)
    rng = orc.Rng(0)
    fr = [o.draw_frame(rng, sigma) for _ in range(8192)]
    noisy = np.stack([f[1] for f in fr])
    cws = np.stack([f[0] for f in fr])
    dec = pkg.Decoder(pkg.Code(path), 16, 32, 0.3, 30, max_batch=8192)
    d, it, sy = dec.decode_batch(noisy, sigma)
    nerr = dec.count_errors(cws)
    rec = o.sim(orc.Rng(0), 2.0, 16, 32, 0.3, 30, 8192)
    assert (nerr == rec[:, 0]).all() and (sy == rec[:, 1]).all() and (it == rec[:, 2]).all()
    dec.close()


# ---------------------------------------------------------------------------------------------------
# the reference-shaped boundary and the C harness
# ---------------------------------------------------------------------------------------------------
def test_reference_shaped_api_frame_by_frame(pkg, orc, refapi):
    """LoadCode ... ModelChannel_AWGN_BPSK -> Decoding() in the order main() calls them, vs the oracle."""
    path = os.path.join(MAT, "N96_K48_GF64")
    s = refapi.RefSession(path, 16)
    o = orc.Oracle(path)
    rng = orc.Rng(0)
    refapi.seed48(0)
    sg = o.sigma(2.0)
    for f in range(6):
        cw = s.draw_codeword()
        sl, sgf = s.model_channel(2.0)
        ocw, onoisy = o.draw_frame(rng, sg)
        osl, osg = o.sort_llr(o.llr(onoisy, sg))
        assert (cw == ocw).all() and sl.tobytes() == osl.tobytes() and (sgf == osg).all()
        it, decide, synd = s.decoding(10, 32, 0.3)
        ref = o.decode(o.llr(onoisy, sg), 16, 32, 0.3, 10, want_state=True)
        assert it == ref["iters"] and synd == ref["synd"] and (decide == ref["decide"]).all()
        assert s.APP.tobytes() == ref["app"].tobytes() and s.CtoV.tobytes() == ref["ctov"].tobytes()
    s.close()


def _run_ems_sim(pkg, args, batch, cwd):
    os.makedirs(os.path.join(cwd, "data"), exist_ok=True)
    out = subprocess.run([pkg.SIM_PATH] + [str(a) for a in args], cwd=cwd, env=dict(os.environ, EMS_BATCH=str(batch)),
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    txt = out.stdout.decode("latin-1")
    tr = [(int(m.group(1)), int(m.group(3)), m.group(4)) for m in
          re.finditer(r"<\d+> FER= (\d+) / (\d+) = [\d.]+ BER= (\d+) / x = [\d.]+\s+avr_it=([\d.]+)", txt)]
    return tr


@pytest.mark.parametrize("name,batch", [("K1", 5), ("K1_ubs", 64), ("K2", 64), ("K3", 4), ("K4", 4), ("K5", 128), ("I1", 16), ("I2", 7), ("N40", 32), ("N36", 8), ("N64", 24), ("N56", 16),
                                        ("D20", 3), ("D20b", 2), ("D10", 3), ("D10b", 3), ("W1", 64), ("W2", 50), ("W3", 33), ("R1", 16), ("R2", 7), ("R3", 32)])
def test_c_harness_reproduces_reference_traces(pkg, kat, tmp_path, name, batch):
    """The C Monte-Carlo harness (same argv as the reference) on the GPU prints the reference's trace."""
    k = kat[name]
    matrix = ("KN_" + k["matrix"]) if k["fmt"] == "kn" and not k["matrix"].startswith("KN_") else k["matrix"]
    if name in ("K1", "K2"):
        matrix = "KN_N96_K48_GF64.txt"
    tr = _run_ems_sim(pkg, [k["frames"], k["NbIterMax"], os.path.join(MAT, matrix), k["EbN"], k["nm"], k["offset"], k["NbOper"]],
                      batch, str(tmp_path))
    assert [list(t) for t in tr] == [list(t) for t in k["trace"]]


def test_python_driver_single_rank_gpu(harness, kat):
    k = kat["K2"]
    cnt, _ = harness.monte_carlo(os.path.join(MAT, "KN_N96_K48_GF64.txt"), k["frames"], k["NbIterMax"], k["EbN"], k["nm"], k["offset"],
                                 k["NbOper"], batch=32)
    assert [list(t) for t in cnt.trace] == [list(t) for t in k["trace"]]


def test_c_harness_first_frame_offset(pkg, kat, tmp_path):
    """EMS_FIRST_FRAME=f: the process simulates frames f.. of the reference's stream (one process per GPU on disjoint
    ranges); frames 3..11 of K1 give K1's trace minus its first three frames, with either generator."""
    k = kat["K1"]
    full = [list(t) for t in k["trace"]]                     # cumulative (errFrames, bitErr, "avr_it")
    sum_it = [round(float(t[2]) * (i + 1)) for i, t in enumerate(full)]
    want = []
    for i in range(3, len(full)):
        n = i - 2
        want.append([full[i][0] - full[2][0], full[i][1] - full[2][1], "%.2f" % ((sum_it[i] - sum_it[2]) / n)])
    for threads in ("1", "4"):
        os.makedirs(os.path.join(str(tmp_path), "data"), exist_ok=True)
        out = subprocess.run([pkg.SIM_PATH, str(len(full) - 3), str(k["NbIterMax"]), os.path.join(MAT, "KN_N96_K48_GF64.txt"), str(k["EbN"]),
                              str(k["nm"]), str(k["offset"]), str(k["NbOper"])], cwd=str(tmp_path),
                             env=dict(os.environ, EMS_BATCH="4", EMS_FIRST_FRAME="3", EMS_THREADS=threads), stdout=subprocess.PIPE,
                             stderr=subprocess.PIPE, timeout=600)
        assert out.returncode == 0, out.stderr.decode()[-2000:]
        tr = [[int(m.group(1)), int(m.group(3)), m.group(4)] for m in
              re.finditer(r"<\d+> FER= (\d+) / (\d+) = [\d.]+ BER= (\d+) / x = [\d.]+\s+avr_it=([\d.]+)", out.stdout.decode("latin-1"))]
        assert tr == want


@pytest.mark.parametrize("case,batch", [("L4", 512), ("L4b", 256), ("L4c", 500), ("L3a", 128), ("L3b", 64), ("L1", 100)])
def test_c_harness_full_operating_point_run(pkg, tmp_path, case, batch):
    """Complete Monte-Carlo points of the 64800-bit codes (several hundred full-size frames each; L4 runs to the 40th
    erroneous frame of the headline code at 2.0 dB): the GPU harness prints the trace the real reference executable
    printed (tests/golden/kat_long.json, generated by tests/golden/make_kat_long.py)."""
    path = os.path.join(GOLD, "kat_long.json")
    if not os.path.exists(path) or case not in json.load(open(path)):
        pytest.skip("tests/golden/kat_long.json has no case " + case)
    k = json.load(open(path))[case]
    tr = _run_ems_sim(pkg, [k["frames"], k["NbIterMax"], os.path.join(MAT, k["matrix"]), k["EbN"], k["nm"], k["offset"], k["NbOper"]],
                      batch, str(tmp_path))
    assert len(tr) == len(k["trace"]) and (tr[-1][0] == 40 or len(tr) == k["frames"])
    assert [list(t) for t in tr] == [list(t) for t in k["trace"]]


@pytest.mark.parametrize("inflight", [1, 2, 4])
def test_c_harness_trace_independent_of_handles_in_flight(pkg, kat, tmp_path, inflight):
    """EMS_INFLIGHT only changes how many batches overlap on the GPU, never the reported trace (K2, batch 16)."""
    k = kat["K2"]
    os.makedirs(os.path.join(str(tmp_path), "data"), exist_ok=True)
    out = subprocess.run([pkg.SIM_PATH] + [str(a) for a in (k["frames"], k["NbIterMax"], os.path.join(MAT, "KN_N96_K48_GF64.txt"), k["EbN"],
                                                          k["nm"], k["offset"], k["NbOper"])], cwd=str(tmp_path),
                         env=dict(os.environ, EMS_BATCH="16", EMS_INFLIGHT=str(inflight)), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                         timeout=600)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    tr = [[int(m.group(1)), int(m.group(3)), m.group(4)] for m in
          re.finditer(r"<\d+> FER= (\d+) / (\d+) = [\d.]+ BER= (\d+) / x = [\d.]+\s+avr_it=([\d.]+)", out.stdout.decode("latin-1"))]
    assert tr == [list(t) for t in k["trace"]]
