"""GPU parity of the v2 flavour through the C ABI (ems_v2_*): against the oracle
restatement (oracle/ems_v2_oracle.c, pinned to the real v2 functions) on seeded likelihoods, and against the
reference-generated golden frames of tests/golden/v2_frames.npz."""
import importlib
import os
import sys

import numpy as np
import pytest

from conftest import MAT, ROOT, gold

pytestmark = pytest.mark.gpu
STATE = ("vtoc", "vtoci", "ctov", "ctovi", "app")


@pytest.fixture(scope="module")
def v2(orc):
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    return importlib.import_module("orc_v2")


def _check_frames(pkg, v2, mat, q, logq, nm, nop, off, itmax, cases, early_exit=True, full_state=True, kernel=None):
    path = os.path.join(MAT, mat)
    o = v2.V2Oracle(path)
    pairs = [v2.synthetic_likelihood(o.N, q, logq, o.o.idx2bin, sg, seed) for sg, seed in cases]
    like = np.stack([p[0] for p in pairs])
    idx = np.stack([p[1] for p in pairs])
    dec = pkg.Decoder(pkg.Code(path), nm, nop, off, itmax, max_batch=len(cases), early_exit=early_exit)
    decide, iters, synd = dec.v2_decode_batch(like, idx)
    assert kernel is None or dec.last_kernel.startswith(kernel), dec.last_kernel
    out = []
    for b in range(len(cases)):
        ref = o.decode(like[b], idx[b], nm, nop, off, itmax, early_exit=early_exit)
        assert iters[b] == ref["iters"] and synd[b] == ref["synd"], (b, iters[b], ref["iters"], synd[b], ref["synd"])
        assert (decide[b] == ref["decide"]).all(), "frame %d: hard decisions differ" % b
        if full_state:
            st = dec.v2_state(b)
            for k in STATE:
                assert st[k].tobytes() == ref[k].tobytes(), "frame %d: %s differs" % (b, k)
        out.append(ref)
    dec.close()
    return out


def test_v2_small_code_vs_oracle(pkg, v2):
    _check_frames(pkg, v2, "N96_K48_GF64", 64, 6, 12, 24, 0.5, 12, [(0.7, 1), (0.95, 2), (1.3, 3), (0.85, 4), (1.0, 5), (0.6, 6)])


def test_v2_small_code_other_parameters(pkg, v2):
    _check_frames(pkg, v2, "N96_K48_GF64", 64, 6, 16, 48, 0.3, 6, [(0.9, 11), (1.2, 12)])
    _check_frames(pkg, v2, "N96_K48_GF64", 64, 6, 8, 8, 1.0, 5, [(0.9, 13), (0.75, 14)], early_exit=False)
    _check_frames(pkg, v2, "N96_K48_GF64", 64, 6, 12, 80, 0.5, 6, [(0.9, 15), (1.1, 16)])       # NbOper > 64: literal elementary steps
    _check_frames(pkg, v2, "N96_K48_GF64", 64, 6, 5, 6, 0.5, 6, [(0.9, 17), (1.1, 18)])         # NbOper < nm + 3


def test_v2_random_parameter_sweep(pkg, v2):
    """Seeded random (nm, NbOper, offset, NbIterMax, noise, early exit) on the small dc = 4, dv = 2 codes."""
    rs = np.random.RandomState(77)
    codes = [("N96_K48_GF64", 64, 6), ("KN_N128_K64_GF256.txt", 256, 8), ("Mat24_N192_M96", 64, 6)]
    for trial in range(24):
        name, q, lq = codes[trial % len(codes)]
        nm = int(rs.choice([5, 6, 8, 12, 16, 20, 24, 32]))
        nop = int(rs.choice([max(2, nm - 2), nm, nm + 3, 2 * nm, 3 * nm, 70]))
        off = float(rs.choice([0.0, 0.3, 0.5, 1.0]))
        itmax = int(rs.randint(2, 8))
        cases = [(float(rs.choice([0.6, 0.8, 1.0, 1.3])), int(rs.randint(1, 10 ** 6))) for _ in range(2)]
        _check_frames(pkg, v2, name, q, lq, nm, nop, off, itmax, cases, early_exit=bool(trial % 3))


def test_v2_two_visit_kernel_parameter_sweep(pkg, v2, monkeypatch):
    """Seeded random (NbOper 23..64, offset, NbIterMax, noise, early exit, workgroup shape) through k_decode_v2_dual on the
    small GF(256) dc = 4 code (8 check nodes: pairs and singles), every message array against the oracle."""
    rs = np.random.RandomState(99)
    for trial in range(12):
        nop = int(rs.choice([23, 24, 30, 40, 60, 64]))
        off = float(rs.choice([0.0, 0.3, 0.5, 1.0]))
        itmax = int(rs.randint(2, 10))
        cases = [(float(rs.choice([0.5, 0.7, 0.9, 1.2])), int(rs.randint(1, 10 ** 6))) for _ in range(3)]
        monkeypatch.setenv("EMS_DUAL", str(int(rs.choice([1, 2, 4]))))
        _check_frames(pkg, v2, "KN_N128_K64_GF256.txt", 256, 8, 20, nop, off, itmax, cases, early_exit=bool(trial % 3))


def test_v2_check_degree_8(pkg, v2):
    """dc = 8 (rows of 8 lanes): GF(256) rate 3/4 (UBS and full-alist files), GF(64) small code."""
    _check_frames(pkg, v2, "Ahmed_64800_R34_GF256", 256, 8, 20, 40, 0.3, 6, [(0.5, 21), (0.62, 22)])
    _check_frames(pkg, v2, "KN_N64800_K48600_GF256_fullalist.txt", 256, 8, 16, 32, 0.5, 4, [(0.55, 23)])
    _check_frames(pkg, v2, "Mat28_N72_M18", 64, 6, 12, 24, 0.3, 8, [(0.6, 24), (0.9, 25), (1.2, 26)])
    _check_frames(pkg, v2, "Mat26_N48_M16", 64, 6, 10, 30, 0.3, 8, [(0.7, 27), (1.0, 28)])          # dc = 6


def test_v2_check_degree_12(pkg, v2):
    """dc = 12 (rows of 4 lanes): the rate-5/6 GF(256) code and a small GF(64) one."""
    _check_frames(pkg, v2, "Ahmed_64800_R56_GF256", 256, 8, 20, 40, 0.3, 4, [(0.42, 31)])
    _check_frames(pkg, v2, "Mat212_N96_M16", 64, 6, 12, 24, 0.3, 6, [(0.5, 32), (0.8, 33)])


def test_v2_check_degrees_above_16_and_irregular_codes(pkg, v2):
    """Round 5 (k_decode_v2_wide): the older generation caps the check degree nowhere (v2.c:1756-1852 loops over rowDegree[node], :750 /
    :794-800 size the scratch by rowDegree[0]).  An irregular (2, dc) code with degrees 2..12, dc = 22 over GF(256), dc = 18 over GF(64)
    (tests/golden/make_v2_degree_matrices.py) and the reference's own rate-9/10 matrix (dc = 20): every message array against the oracle,
    which tests/test_v2_oracle_vs_reference.py holds to the REAL functions on the same codes.  Fast and literal elementary steps."""
    K = "k_decode_v2_wide"
    _check_frames(pkg, v2, "V2Irr_M24_GF256", 256, 8, 20, 40, 0.3, 8, [(0.55, 41), (0.8, 42), (1.1, 43), (0.65, 48)], kernel=K)
    _check_frames(pkg, v2, "V2Irr_M24_GF256", 256, 8, 8, 11, 1.0, 5, [(0.7, 44), (0.9, 49)], early_exit=False, kernel=K)
    _check_frames(pkg, v2, "V2Irr_M24_GF256", 256, 8, 12, 80, 0.5, 6, [(0.7, 50)], kernel=K)            # NbOper > 64: literal elementary steps
    _check_frames(pkg, v2, "V2Irr_M24_GF256", 256, 8, 36, 40, 0.3, 4, [(0.7, 51)], kernel=K)            # nm > 32: literal steps, literal selection
    _check_frames(pkg, v2, "V2Wide22_M10_GF256", 256, 8, 20, 40, 0.3, 6, [(0.4, 45), (0.6, 46), (0.5, 52)], kernel=K)
    _check_frames(pkg, v2, "V2Wide18_M8_GF64", 64, 6, 12, 24, 0.5, 8, [(0.45, 53), (0.6, 54), (0.8, 55)], kernel=K)
    _check_frames(pkg, v2, "Ahmed_64800_R910_GF256", 256, 8, 20, 40, 0.3, 4, [(0.3, 47), (0.33, 56)], kernel=K)


def test_v2_big_code_vs_oracle(pkg, v2):
    _check_frames(pkg, v2, "Ahmed_64800_R12_GF256", 256, 8, 20, 40, 0.3, 8, [(0.62, 7), (0.78, 11)])


@pytest.mark.parametrize("waves,occ4", [(4, False), (4, True)])
def test_v2_big_code_small_workgroup_shapes(pkg, v2, monkeypatch, waves, occ4):
    """The launch shapes picked for big batches (4-wave workgroups; the 128-VGPR build), forced on two frames."""
    monkeypatch.setenv("EMS_WAVES", str(waves))
    if occ4:
        monkeypatch.setenv("EMS_OCC4", "1")
    _check_frames(pkg, v2, "Ahmed_64800_R12_GF256", 256, 8, 20, 40, 0.3, 8, [(0.62, 7), (0.78, 11)])


@pytest.mark.parametrize("waves", [2, 4, 1])
def test_v2_big_code_two_visits_per_wave(pkg, v2, monkeypatch, waves):
    """k_decode_v2_dual (two check nodes per wave; the shape picked from 5 frames per CU on), forced on frames the
    oracle can follow: decisions, iterations, syndromes and every message array, bit for bit."""
    monkeypatch.setenv("EMS_DUAL", str(waves))
    _check_frames(pkg, v2, "Ahmed_64800_R12_GF256", 256, 8, 20, 40, 0.3, 8, [(0.62, 7), (0.78, 11), (0.95, 3)])


@pytest.mark.parametrize("B", [600, 1024, 1536])
def test_v2_full_size_big_batches_shapes_agree(pkg, v2, B):
    """600 frames (three 4-wave workgroups per CU of k_decode_v2_dual), 1024 (four 4-wave workgroups per CU, 128-VGPR
    build of k_decode_v2) and the bench's 1536 (six 2-wave workgroups per CU of k_decode_v2_dual) against the one-frame-per-CU
    launch on 8 distinct frames tiled: decisions, iteration counts, syndromes of every frame and the full state of
    sampled frames agree; one frame is also checked against the oracle."""
    path = os.path.join(MAT, "Ahmed_64800_R12_GF256")
    o = v2.V2Oracle(path)
    base = [v2.synthetic_likelihood(o.N, 256, 8, o.o.idx2bin, sg, seed) for sg, seed in
            ((0.6, 1), (0.7, 2), (0.8, 3), (0.9, 4), (1.0, 5), (0.75, 6), (0.85, 7), (0.95, 8))]
    like = np.stack([base[i % 8][0] for i in range(B)])
    idx = np.stack([base[i % 8][1] for i in range(B)])
    code = pkg.Code(path)
    big = pkg.Decoder(code, 20, 40, 0.3, 9, max_batch=B)
    d, it, sy = big.v2_decode_batch(like, idx)
    small = pkg.Decoder(code, 20, 40, 0.3, 9, max_batch=8)
    d8, it8, sy8 = small.v2_decode_batch(like[:8], idx[:8])
    assert len(set(it8.tolist())) > 1                      # frames that stop at different iterations share the CUs
    for b in range(B):
        assert it[b] == it8[b % 8] and sy[b] == sy8[b % 8] and (d[b] == d8[b % 8]).all(), b
    for b in (0, 5, 517, B - 1):
        sb, ss = big.v2_state(b), small.v2_state(b % 8)
        for k in STATE:
            assert sb[k].tobytes() == ss[k].tobytes(), (b, k)
    ref = o.decode(like[3], idx[3], 20, 40, 0.3, 9)
    assert it[3] == ref["iters"] and (d[3] == ref["decide"]).all() and big.v2_state(3)["app"].tobytes() == ref["app"].tobytes()
    big.close(); small.close()


def test_v2_golden_frames_from_the_reference(pkg, orc, v2):
    g = gold("v2_frames.npz")
    path = os.path.join(MAT, "Ahmed_64800_R12_GF256")
    o = v2.V2Oracle(path)
    code = pkg.Code(path)
    for i, (sg, seed, itmax) in enumerate(g["big_cases"]):
        like, idx = v2.synthetic_likelihood(o.N, 256, 8, o.o.idx2bin, float(sg), int(seed))
        dec = pkg.Decoder(code, 20, 40, 0.3, int(itmax), max_batch=1)
        d, it, sy = dec.v2_decode_batch(like[None], idx[None])
        meta = g["big%d_meta" % i]
        assert it[0] == meta[0] and sy[0] == meta[1] and (d[0].astype(np.uint8) == g["big%d_decide" % i]).all()
        st = dec.v2_state(0)
        assert [orc.crc32(st[k]) for k in STATE] == [int(x) for x in meta[2:]]
        dec.close()
    path = os.path.join(MAT, "N96_K48_GF64")
    o = v2.V2Oracle(path)
    code = pkg.Code(path)
    for i, (sg, seed) in enumerate(g["small_cases"]):
        like, idx = v2.synthetic_likelihood(o.N, 64, 6, o.o.idx2bin, float(sg), int(seed))
        dec = pkg.Decoder(code, 12, 24, 0.5, 12, max_batch=1)
        d, it, sy = dec.v2_decode_batch(like[None], idx[None])
        assert it[0] == g["small%d_meta" % i][0] and sy[0] == g["small%d_meta" % i][1]
        st = dec.v2_state(0)
        for k in STATE:
            assert st[k].tobytes() == g["small%d_%s" % (i, k)].tobytes(), (i, k)
        dec.close()


def test_v2_rejects_what_it_does_not_build(pkg, v2):
    path = os.path.join(MAT, "Ahmed_64800_R34_GF16")            # GF(16)
    dec = pkg.Decoder(pkg.Code(path), 8, 16, 0.3, 5, max_batch=1)
    o = v2.V2Oracle(path)
    like, idx = v2.synthetic_likelihood(o.N, 16, 4, o.o.idx2bin, 0.8, 1)
    with pytest.raises(pkg.EmsError):
        dec.v2_load_likelihood(like[None], idx[None])
    dec.close()


def test_v2_long_golden_run_from_the_reference(pkg, orc, v2):
    """48 full-size frames (easy to non-converging) decoded by the real v2 functions (tests/golden/v2_frames_long.npz,
    make_golden_v2_long.py): iterations, syndrome, hard decisions and every state array (CRC-32) of one batched GPU
    decode."""
    p = os.path.join(ROOT, "tests", "golden", "v2_frames_long.npz")
    if not os.path.exists(p):
        pytest.skip("tests/golden/v2_frames_long.npz not generated")
    g = np.load(p)
    nm, nop, off, itmax = int(g["params"][0]), int(g["params"][1]), float(g["params"][2]), int(g["params"][3])
    path = os.path.join(MAT, "Ahmed_64800_R12_GF256")
    o = v2.V2Oracle(path)
    cases = g["cases"]
    B = len(cases)
    dec = pkg.Decoder(pkg.Code(path), nm, nop, off, itmax, max_batch=B)
    like = np.zeros((B, o.N, 256), np.float32)
    idx = np.zeros((B, o.N, 256), np.int32)
    for b, (sg, seed) in enumerate(cases):
        like[b], idx[b] = v2.synthetic_likelihood(o.N, 256, 8, o.o.idx2bin, float(sg), int(seed))
    d, it, sy = dec.v2_decode_batch(like, idx)
    for b in range(B):
        m = [int(x) for x in g["meta"][b]]
        assert (int(it[b]), int(sy[b]), orc.crc32(d[b].astype(np.uint8))) == (m[0], m[1], m[2]), (b, cases[b], int(it[b]), m[0])
        st = dec.v2_state(b)
        assert [orc.crc32(st[k]) for k in STATE] == m[3:], (b, cases[b])
    dec.close()


# ---------------------------------------------------------------------------------------------------
# the older generation's main(): BER samples at iterations 9/19/29/39/49, "> 12 bit errors", printed lines
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dual", [0, 2])
def test_v2_snapshots_vs_oracle(pkg, v2, monkeypatch, dual):
    """ems_v2_get_snapshots against the oracle: the sample taken when iteration 10 j + 9 begins is the bit error count of the
    decisions after iterations 0 .. 10 j + 8, i.e. of an oracle decode with NbIterMax = 10 j + 8 -- or 0 when the frame had stopped
    before (v2.c:2174-2243).  Small GF(256) code, all-zero codeword, frames from easy to hopeless; both kernel families."""
    monkeypatch.setenv("EMS_DUAL", str(dual))
    path = os.path.join(MAT, "KN_N128_K64_GF256.txt")
    o = v2.V2Oracle(path)
    cases = [(0.55, 31), (0.8, 32), (0.9, 33), (1.0, 34), (1.1, 35), (1.3, 36), (0.95, 37), (1.05, 38)]
    pairs = [v2.synthetic_likelihood(o.N, 256, 8, o.o.idx2bin, sg, seed) for sg, seed in cases]
    like, idx = np.stack([p[0] for p in pairs]), np.stack([p[1] for p in pairs])
    B, itmax, nm, nop, off = len(cases), 52, 20, 40, 0.3
    dec = pkg.Decoder(pkg.Code(path), nm, nop, off, itmax, max_batch=B)
    dec.v2_enable_snapshots(True)
    dec.set_codewords(np.zeros((B, o.N), np.int32))
    decide, iters, synd = dec.v2_decode_batch(like, idx)
    snap = dec.v2_snapshots(B)
    K = o.N - o.M
    popc = np.array([bin(int(x)).count("1") for x in o.o.idx2bin])
    seen = set()
    for b in range(B):
        for j in range(5):
            it_j = 10 * j + 9                                     # the sample is taken iff iteration it_j is run: iters counts iterations run
            want = 0
            if iters[b] > it_j:
                ref = o.decode(like[b], idx[b], nm, nop, off, it_j - 1)       # iterations 0 .. it_j - 1
                want = int(popc[ref["decide"][:K]].sum())
                seen.add(j)
            assert snap[b, j] == want, (b, j, iters[b], snap[b].tolist(), want)
    assert seen == {0, 1, 2, 3, 4} and (iters < 10).any()          # every sample exercised, and frames that stop before the first
    dec.close()


@pytest.mark.parametrize("case,batch", [("V1", 24), ("V2", 5), ("V3", 3), ("V4", 8), ("V5", 7), ("V6", 40)])     # V4..V6: dc = 20 (the reference's rate-9/10 matrix), irregular (2, dc), dc = 22
def test_v2_harness_prints_the_lines_of_the_v2_executable(pkg, tmp_path, case, batch):
    """csrc/harness/ems_sim_v2.c (host draws in the reference's drand48 order, GPU likelihoods + decode + BER samples) against every
    statistics line the REAL EMS_HS_L-BubbleCheck_UBS_decoder_v2.c executable printed (tests/golden/v2_traces.json, generated by
    tests/golden/make_golden_v2_trace.py from oracle/_ref/essai_v2): five lines per frame, frame after frame."""
    import json
    import re
    import subprocess
    path = os.path.join(ROOT, "tests", "golden", "v2_traces.json")
    if not os.path.exists(path) or case not in json.load(open(path)):
        pytest.skip("tests/golden/v2_traces.json has no case " + case)
    k = json.load(open(path))[case]
    out = subprocess.run([pkg.SIM_V2_PATH] + [str(a) for a in (k["frames"], k["NbIterMax"], os.path.join(MAT, k["matrix"]), k["EbN"], k["nm"], k["offset"], k["NbOper"])],
                         cwd=str(tmp_path), env=dict(os.environ, EMS_BATCH=str(batch)), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    lines = re.findall(r"<\d+> FER= \d+/ \d+ = \S+ BER= \d+/ \d+ = \S+", out.stdout.decode("latin-1"))
    assert lines == k["lines"]


def test_v2_more_than_255_iterations(pkg, v2, monkeypatch):
    """NbIterMax = 300, early exit off: the done-flag tags of the persistent v2 kernels wrap (see tests/test_gpu_fuzz_codes.py)."""
    _check_frames(pkg, v2, "N96_K48_GF64", 64, 6, 12, 24, 0.5, 300, [(1.3, 1), (1.0, 2)], early_exit=False)
    monkeypatch.setenv("EMS_DUAL", "2")
    _check_frames(pkg, v2, "KN_N128_K64_GF256.txt", 256, 8, 20, 40, 0.3, 300, [(1.2, 3)], early_exit=False)


@pytest.mark.parametrize("block", range(int(os.environ.get("EMS_FUZZ_V2_HARNESS", "2"))))
def test_v2_harness_on_random_codes_equals_the_real_executable(pkg, tmp_path, block):
    """Live on the GPU box: the REAL older-generation executable (oracle/_ref/essai_v2, built from /root/reference where it lies and carried
    along as a binary) and csrc/harness/ems_sim_v2.c with the same argv on seeded random (2, dc) GF(256) codes -- regular with dc 3..24 and (round 5)
    irregular with the largest degree in row 0 -- every statistics
    line (BER samples at iterations 9 / 19 / 29 / 39 / 49, the "> 12 bit errors" frame rule), frame after frame, at SNRs where frames fail."""
    import re
    import subprocess
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from make_irregular_matrix import write
    exe = os.path.join(ROOT, "oracle", "_ref", "essai_v2")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/essai_v2 is not built (make -C oracle ref, needs /root/reference)")
    rs = np.random.RandomState(7700 + block)
    pat = r"<\d+> FER= \d+/ \d+ = \S+ BER= \d+/ \d+ = \S+"
    for case in range(6):
        M = 2 * int(rs.randint(4, 24))
        if case < 4:                                             # (2, dc)-regular, lane-row kernels up to 16, k_decode_v2_wide above
            degs = [int(rs.choice([3, 4, 6, 8, 12, 17, 20, 24]))] * M
        else:                                                    # irregular (2, dc): the largest degree in row 0, where the executable sizes its scratch
            top = int(rs.choice([8, 12, 18, 21]))
            while True:
                degs = [top] + [int(rs.randint(2, top + 1)) for _ in range(M - 1)]
                if sum(degs) % 2 == 0:
                    break
        N = sum(degs) // 2
        while True:
            slots = [int(c) for c in rs.permutation(np.repeat(np.arange(N), 2))]
            rows, k0 = [], 0
            for d in degs:
                rows.append(slots[k0:k0 + d]); k0 += d
            if all(len(set(r)) == len(r) for r in rows):
                break
        coefs = [[int(x) for x in rs.randint(1, 256, d)] for d in degs]
        path = str(tmp_path / ("v2h_%d_%d" % (block, case)))
        write(path, N, M, 256, degs, rows, coefs)
        nm = int(rs.choice([8, 12, 20]))
        argv = [str(a) for a in (int(rs.randint(6, 15)), int(rs.choice([12, 25, 50])), path, float(rs.choice([4.0, 6.0, 7.5, 9.0])), nm, float(rs.choice([0.3, 1.0])),
                                 int(rs.choice([nm + 3, 2 * nm])))]
        ref = subprocess.run([exe] + argv, cwd=str(tmp_path), stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=300).stdout.decode("latin-1")
        want = re.findall(pat, ref)
        if not want:                                             # (a rank-deficient random matrix: the executable stops before the first frame)
            continue
        out = subprocess.run([pkg.SIM_V2_PATH] + argv, cwd=str(tmp_path), env=dict(os.environ, EMS_BATCH=str(2 + case)), stdout=subprocess.PIPE,
                             stderr=subprocess.PIPE, timeout=300)
        assert out.returncode == 0, (argv, out.stderr.decode()[-1500:])
        assert re.findall(pat, out.stdout.decode("latin-1")) == want, (block, case, argv)
