"""Seeded random codes through the library against the oracle, whole-frame state: check degrees from 2 to 40 (regular and mixed, the largest
in row 0 as the reference's scratch sizing wants it, init.c:320), GF(16) / GF(64) / GF(256), variable nodes of any degree including 0
(never visited: decided once from the channel) and check nodes that list a variable node twice, random nm / NbOper / offset / NbIterMax,
early exit on and off, LLR rows that are continuous, on a grid (ties) or partly saturated.  The matrices are written in the reference's
block alist layout (init.c:195-207) into a temporary directory; the decode input is the LLR block itself (no encoder in the loop), so a
rank-deficient random matrix is as good as any.  Every kernel family the shape policy can pick for such codes is exercised: the generic
one-visit kernels (rows of 16 / 8 / 4 lanes), k_decode_wide, the (dc, nm)-specialised ones when a regular code happens to fit them."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from make_irregular_matrix import write  # noqa: E402

pytestmark = pytest.mark.gpu
SEED = int(os.environ.get("EMS_FUZZ_SEED", "0"))          # shifts every seeded stream of this file (soak runs)

DEGREE_SETS = [[2], [3], [4], [2, 3], [3, 4, 5], [5, 6, 7, 8], [8], [9, 10, 12], [13, 16], [17, 18, 20], [24, 19, 17], [2, 9, 17], [33, 20, 40]]


def _random_code(rs, path):
    q = int(rs.choice([16, 64, 256]))
    degs = DEGREE_SETS[int(rs.randint(len(DEGREE_SETS)))]
    M = int(rs.randint(3, 9))
    deg = sorted((int(rs.choice(degs)) for _ in range(M)), reverse=True)            # the largest degree in row 0
    E = sum(deg)
    N = int(max(max(deg) + 1, M + 1, E // int(rs.choice([1, 2, 3]))))                  # average column degree about 1, 2 or 3 (and K = N - M > 0)
    N += int(rs.randint(0, 3))                                                         # ... and now and then a variable node in no check
    rows = []
    for d in deg:
        r = [int(c) for c in rs.choice(N, d, replace=False)]
        if d >= 3 and rs.rand() < 0.25:                                                # a variable node listed twice
            i, j = (int(x) for x in rs.choice(d, 2, replace=False))
            r[j] = r[i]
        rows.append(r)
    coefs = [[int(x) for x in rs.randint(1, q, d)] for d in deg]
    write(path, N, M, q, deg, rows, coefs)
    return q, N, deg, any(len(set(r)) < len(r) for r in rows)


@pytest.mark.parametrize("block", range(int(os.environ.get("EMS_FUZZ_BLOCKS", "12"))))      # (EMS_FUZZ_BLOCKS=300: a one-minute soak of 4800 codes)
def test_random_codes_vs_oracle(pkg, orc, tmp_path, block):
    rs = np.random.RandomState(4100 + block + SEED)
    ran = 0
    kernels = set()
    for case in range(16):
        path = str(tmp_path / ("fuzz_%d_%d" % (block, case)))
        q, N, deg, rep = _random_code(rs, path)
        nm = int(rs.randint(5, min(q, 40) + 1))
        if rs.rand() < 0.15:
            nm = min(q, int(rs.randint(41, 65)))                                       # the long-message builds
        nop = int(rs.choice([nm, nm + 3, 2 * nm, 3 * nm, 70]))
        off = float(rs.choice([0.0, 0.3, 1.0]))
        itmax = int(rs.randint(2, 6))
        ee = bool(rs.randint(2))
        B = 3
        llr = (rs.rand(B, N, q) * 30).astype(np.float32)
        llr[1] = np.round(llr[1] * 2) / 2                                              # ties
        llr[2, ::2, q // 2:] = 1e5                                                     # half the symbols of every other node not selectable
        llr -= llr.min(axis=2, keepdims=True)                                          # the channel's normalisation: 0 at the best symbol
        llr[2, ::2, q // 2:] = 1e5
        o = orc.Oracle(path)
        try:
            dec = pkg.Decoder(pkg.Code(path), nm, nop, off, itmax, max_batch=B, early_exit=ee)
        except pkg.EmsError as e:                                                      # a combination the library states it does not build
            assert e.code == pkg.EMS_EUNSUPPORTED, (deg, q, nm, str(e))
            continue
        dec.load_llr(llr)
        dec.decode(B)
        d, it, sy = dec.results(B)
        kernels.add(dec.last_kernel.split("(")[0])
        for b in range(B):
            ref = o.decode(llr[b], nm, nop, off, itmax, early_exit=ee, want_state=True)
            ctx = (block, case, q, deg, nm, nop, off, itmax, ee, rep, dec.last_kernel, b)
            assert (d[b] == ref["decide"]).all() and it[b] == ref["iters"] and sy[b] == ref["synd"], ctx
            assert dec.app(b).tobytes() == ref["app"].tobytes(), ctx
            assert dec.ctov(b).tobytes() == ref["ctov"].tobytes(), ctx
        if case % 3 == 0 and nm > 6:                                                   # ems_set_params on the same handle: shorter messages, other NbOper / offset
            nm2 = int(rs.randint(5, nm))
            nop2 = int(rs.choice([nm2, nm2 + 3, 2 * nm2]))
            dec.set_params(nm=nm2, nbOper=nop2, offset=0.5)
            dec.load_llr(llr)
            dec.decode(B)
            d, it, sy = dec.results(B)
            for b in range(B):
                ref = o.decode(llr[b], nm2, nop2, 0.5, itmax, early_exit=ee, want_state=True)
                ctx = (block, case, "set_params", q, deg, nm, nm2, nop2, itmax, ee, rep, dec.last_kernel, b)
                assert (d[b] == ref["decide"]).all() and it[b] == ref["iters"] and sy[b] == ref["synd"], ctx
                assert dec.app(b).tobytes() == ref["app"].tobytes() and dec.ctov(b).tobytes() == ref["ctov"].tobytes(), ctx
        dec.close()
        ran += 1
    print("block %d: %d of 16 cases ran, kernels %s" % (block, ran, sorted(kernels)))
    assert ran >= 10, ran                                                              # (most combinations are built)
    assert len(kernels) >= 2, kernels


def _random_regular_code(rs, path, q, dc):
    """M checks of degree dc over N variable nodes, no variable node twice in a check, column degrees 1..4 (mostly 2)."""
    M = int(rs.randint(5, 41))
    E = M * dc
    N = max(M + 1, dc + 1, E // 2 + int(rs.randint(-2, 3)))
    while True:
        slots = [int(c) for c in rs.permutation(N)] + [int(c) for c in rs.randint(0, N, max(0, E - N))]
        slots = slots[:E]
        rs.shuffle(slots)
        rows = [slots[m * dc:(m + 1) * dc] for m in range(M)]
        if all(len(set(r)) == dc for r in rows):
            break
    coefs = [[int(x) for x in rs.randint(1, q, dc)] for _ in range(M)]
    write(path, N, M, q, [dc] * M, rows, coefs)
    return N, M


FAMILIES = [  # q, dc, nm, the knobs that force each kernel family built for this shape ({} = the library's own choice)
    (256, 4, 20, [{}, {"EMS_DUAL": "1"}, {"EMS_DUAL": "2"}, {"EMS_DUAL": "4"}, {"EMS_TWIN": "1"}, {"EMS_WAVES": "4", "EMS_OCC4": "1"}]),
    (64, 4, 16, [{}, {"EMS_DUAL": "1"}, {"EMS_DUAL": "2"}, {"EMS_TWIN": "1"}]),
    (64, 4, 24, [{}, {"EMS_DUAL": "2"}]),
    (16, 8, 8, [{}, {"EMS_Q16T": "0"}, {"EMS_DUAL": "2"}, {"EMS_Q16T_WAVES": "3"}, {"EMS_Q16T_TRAV": "1"}]),
]


@pytest.mark.parametrize("fam", range(len(FAMILIES)))
def test_random_regular_codes_through_the_specialised_kernels(pkg, orc, tmp_path, monkeypatch, fam):
    """Random REGULAR graphs (their level structure, pairs and groups of 16 differ from code to code: build_pairs / build_groups and the
    dependency flags of the persistent kernels see topologies the fixed matrices do not have) through every kernel family built for the
    shape -- one visit per wave, two visits, two frames, the lane-serial GF(16) kernel -- whole-frame state against the oracle."""
    q, dc, nm, knobs = FAMILIES[fam]
    rs = np.random.RandomState(5200 + fam + SEED)
    seen = set()
    for case in range(int(os.environ.get("EMS_FUZZ_REGULAR", "14"))):
        path = str(tmp_path / ("reg_%d_%d" % (fam, case)))
        N, M = _random_regular_code(rs, path, q, dc)
        nop = int(rs.choice([nm + 3, 2 * nm, 2 * nm, 3 * nm]))
        off = float(rs.choice([0.0, 0.3, 1.0]))
        itmax = int(rs.randint(2, 7))
        ee = bool(rs.randint(2))
        B = int(rs.randint(3, 10))
        llr = (rs.rand(B, N, q) * 30).astype(np.float32)
        llr[1] = np.round(llr[1] * 2) / 2
        llr[2, ::3, q // 2:] = 1e5
        llr -= llr.min(axis=2, keepdims=True)
        llr[2, ::3, q // 2:] = 1e5
        o = orc.Oracle(path)
        refs = [o.decode(llr[b], nm, nop, off, itmax, early_exit=ee, want_state=True) for b in range(B)]
        for kn in knobs:
            for k, v in kn.items():
                monkeypatch.setenv(k, v)
            dec = pkg.Decoder(pkg.Code(path), nm, nop, off, itmax, max_batch=B, early_exit=ee)
            dec.load_llr(llr)
            dec.decode(B)
            d, it, sy = dec.results(B)
            seen.add(dec.last_kernel.split("(")[0])
            for b in range(B):
                ref = refs[b]
                ctx = (fam, case, M, N, nop, off, itmax, ee, kn, dec.last_kernel, b)
                assert (d[b] == ref["decide"]).all() and it[b] == ref["iters"] and sy[b] == ref["synd"], ctx
                assert dec.app(b).tobytes() == ref["app"].tobytes(), ctx
                assert dec.ctov(b).tobytes() == ref["ctov"].tobytes(), ctx
            dec.close()
            for k in kn:
                monkeypatch.delenv(k)
    print("family", (q, dc, nm), sorted(seen))
    assert len(seen) >= 2, seen


def _random_dv2_code(rs, path, q, dc):
    """A random (2, dc)-regular graph: every variable node on exactly two DIFFERENT check nodes (what the v2 flavour decodes, v2.c:2157)."""
    M = 2 * int(rs.randint(3, 25))
    N = M * dc // 2
    while True:
        slots = [int(c) for c in rs.permutation(np.repeat(np.arange(N), 2))]
        rows = [slots[m * dc:(m + 1) * dc] for m in range(M)]
        if all(len(set(r)) == dc for r in rows):
            break
    coefs = [[int(x) for x in rs.randint(1, q, dc)] for _ in range(M)]
    write(path, N, M, q, [dc] * M, rows, coefs)
    return N, M


@pytest.mark.parametrize("q,logq,dc,nm", [(256, 8, 4, 20), (64, 6, 4, 12), (64, 6, 3, 8), (256, 8, 6, 16), (64, 6, 8, 10), (256, 8, 12, 20)])
def test_random_dv2_codes_through_the_v2_flavour(pkg, orc, tmp_path, monkeypatch, q, logq, dc, nm):
    """The v2 flavour (ems_v2_*) on random (2, dc)-regular graphs -- its destination-edge tables, its last-visitor flags and the pair tables
    of k_decode_v2_dual depend on the topology -- every message array, APP and decision against the v2 oracle."""
    import importlib
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    v2 = importlib.import_module("orc_v2")
    rs = np.random.RandomState(6300 + q + dc + SEED)
    seen = set()
    for case in range(int(os.environ.get("EMS_FUZZ_V2", "6"))):
        path = str(tmp_path / ("v2_%d_%d_%d" % (q, dc, case)))
        N, M = _random_dv2_code(rs, path, q, dc)
        nop = int(rs.choice([nm, nm + 3, 2 * nm, 3 * nm]))
        off = float(rs.choice([0.0, 0.3, 0.5, 1.0]))
        itmax = int(rs.randint(2, 7))
        ee = bool(rs.randint(2))
        o = v2.V2Oracle(path)
        cases = [(float(rs.choice([0.6, 0.8, 1.0, 1.3])), int(rs.randint(1, 10 ** 6))) for _ in range(3)]
        pairs = [v2.synthetic_likelihood(o.N, q, logq, o.o.idx2bin, sg, seed) for sg, seed in cases]
        like = np.stack([p[0] for p in pairs])
        idx = np.stack([p[1] for p in pairs])
        refs = [o.decode(like[b], idx[b], nm, nop, off, itmax, early_exit=ee) for b in range(3)]
        for kn in ({}, {"EMS_DUAL": "1"}, {"EMS_DUAL": "2"}):
            if kn and not (dc == 4 and nm == 20 and nop >= 23):
                continue                                                               # (the two-visit v2 kernel is built for dc 4 / nm 20)
            for k, v in kn.items():
                monkeypatch.setenv(k, v)
            dec = pkg.Decoder(pkg.Code(path), nm, nop, off, itmax, max_batch=3, early_exit=ee)
            decide, iters, synd = dec.v2_decode_batch(like, idx)
            seen.add(dec.last_kernel.split("(")[0])
            for b in range(3):
                ref = refs[b]
                ctx = (q, dc, case, M, nop, off, itmax, ee, kn, dec.last_kernel, b)
                assert iters[b] == ref["iters"] and synd[b] == ref["synd"] and (decide[b] == ref["decide"]).all(), ctx
                st = dec.v2_state(b)
                for k in ("vtoc", "vtoci", "ctov", "ctovi", "app"):
                    assert st[k].tobytes() == ref[k].tobytes(), ctx + (k,)
            dec.close()
            for k in kn:
                monkeypatch.delenv(k)
    print("v2", (q, dc, nm), sorted(seen))


@pytest.mark.parametrize("block", range(int(os.environ.get("EMS_FUZZ_HARNESS", "4"))))
def test_c_harness_on_random_codes_equals_the_oracle_simulation(pkg, orc, tmp_path, block):
    """The C Monte-Carlo harness (the reference's argv; frame source, encoder, channel, decode on the GPU, statistics) on the seeded random
    codes of tests/fuzz_codes.py prints the trace of the oracle's simulation -- which tests/test_oracle_fuzz_vs_reference.py holds to the REAL
    executable on the same codes where that executable exists.  A rank-deficient matrix ends both before the first frame (tools.c:181-185)."""
    import re
    import subprocess
    from fuzz_codes import simulation_case
    ran = 0
    for seed in range(block * 10, block * 10 + 10):
        path = str(tmp_path / ("rnd_%d" % seed))
        frames, it, eb, nm, off, nop = simulation_case(seed, path)
        o = orc.Oracle(path)
        cwd = str(tmp_path)
        os.makedirs(os.path.join(cwd, "data"), exist_ok=True)
        out = subprocess.run([pkg.SIM_PATH] + [str(a) for a in (frames, it, path, eb, nm, off, nop)], cwd=cwd,
                             env=dict(os.environ, EMS_BATCH=str(3 + seed % 7)), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
        txt = out.stdout.decode("latin-1")
        tr = [(int(m.group(1)), int(m.group(3)), m.group(4)) for m in
              re.finditer(r"<\d+> FER= (\d+) / (\d+) = [\d.]+ BER= (\d+) / x = [\d.]+\s+avr_it=([\d.]+)", txt)]
        try:
            rec = o.sim(orc.Rng(0), eb, nm, nop, off, it, frames)
        except RuntimeError as e:
            assert "not full rank" in str(e) and "not full rank" in txt and not tr, (seed, str(e), txt[-300:])
            continue
        assert out.returncode == 0, (seed, out.stderr.decode()[-1000:])
        mine = orc.trace_from_records(rec, o.K, o.logq)
        assert [list(t) for t in tr] == [list(t) for t in mine], (seed, nm, nop, off, it, eb)
        if os.path.exists(orc.ref_binary("ubs")):                                      # ... and, live, the REAL executable's (its binary travels with oracle/_ref)
            real = orc.run_reference_binary("ubs", frames, it, path, eb, nm, off, nop, timeout=120)
            assert [list(t) for t in real] == [list(t) for t in tr], (seed, "real executable", nm, nop, off, it, eb)
        ran += 1
    assert ran >= 7, ran


@pytest.mark.parametrize("q,deg,nm,nop", [(256, [40, 40, 33, 20, 20], 52, 156), (64, [64, 50, 40, 64], 32, 64), (64, [64, 33, 64], 40, 43), (256, [64, 64, 17], 31, 62),
                                          (16, [64, 64, 40], 16, 32)])
def test_list_areas_beyond_64_kib(pkg, orc, tmp_path, q, deg, nm, nop):
    """Check nodes whose message lists take more than 64 KiB of LDS (4 dc - 6 lists of nm + 2 entries: dc = 40 with nm >= 52, dc = 64 with
    nm >= 31): the elementary-step task descriptors carried BYTE offsets in 16 bits until a seeded random code of the soak
    (EMS_FUZZ_BLOCKS=300, block 53) found them wrapping; they carry entry offsets now.  Whole-frame state against the oracle."""
    rs = np.random.RandomState(len(deg) * 1000 + nm)
    N = max(deg) + 12
    rows = [[int(c) for c in rs.choice(N, d, replace=False)] for d in deg]
    rows[1][3] = rows[1][0]                                                            # (and a repeated variable node)
    coefs = [[int(x) for x in rs.randint(1, q, d)] for d in deg]
    path = str(tmp_path / "big_lists")
    write(path, N, len(deg), q, deg, rows, coefs)
    B = 2
    llr = (rs.rand(B, N, q) * 30).astype(np.float32)
    llr[1] = np.round(llr[1] * 2) / 2
    llr -= llr.min(axis=2, keepdims=True)
    o = orc.Oracle(path)
    try:
        dec = pkg.Decoder(pkg.Code(path), nm, nop, 0.3, 3, max_batch=B, early_exit=False)
    except pkg.EmsError as e:                                                          # (only "does not fit the LDS" is an answer here)
        assert e.code == pkg.EMS_EUNSUPPORTED and "LDS" in str(e), str(e)
        pytest.skip(str(e))
    dec.load_llr(llr)
    dec.decode(B)
    d, it, sy = dec.results(B)
    assert "k_decode_wide" in dec.last_kernel
    for b in range(B):
        ref = o.decode(llr[b], nm, nop, 0.3, 3, early_exit=False, want_state=True)
        assert (d[b] == ref["decide"]).all() and it[b] == ref["iters"] and sy[b] == ref["synd"], b
        assert dec.app(b).tobytes() == ref["app"].tobytes() and dec.ctov(b).tobytes() == ref["ctov"].tobytes(), b
    dec.close()


def test_degree_two_checks_with_a_symbol_named_twice_in_a_list(pkg, orc, tmp_path):
    """dc = 2: a check node's output for one edge IS the other edge's input list (bubble_decoder.c:155-163, :214-222), and when fewer than nm
    symbols of a node are selectable the entries behind them all carry symbol 0 (NB_LDPC.c:362) -- the reference's in-order scatter lets the
    last of them stay (bubble_decoder.c:262-265).  Found by the soak (EMS_FUZZ_BLOCKS=300, block 68): lanes scattering such entries at once
    let whichever the LDS / memory system serves last win (usually the right one, hence a soak to find it); the record now gives every entry
    the value of the last one with its symbol.  GF(16) nm = 11 with 8 selectable symbols per node, GF(64) nm = 40 with 20, GF(256) 20 / 12."""
    for q, nm, sel in ((16, 11, 8), (64, 40, 20), (256, 20, 12)):
        rs = np.random.RandomState(q + nm)
        N, M = 9, 6
        rows = [[int(c) for c in rs.choice(N, 2, replace=False)] for _ in range(M)]
        coefs = [[int(x) for x in rs.randint(1, q, 2)] for _ in range(M)]
        path = str(tmp_path / ("dc2_%d" % q))
        write(path, N, M, q, [2] * M, rows, coefs)
        B = 4
        llr = (rs.rand(B, N, q) * 30).astype(np.float32)
        llr -= llr.min(axis=2, keepdims=True)
        for b in range(B):
            for n in range(N):
                llr[b, n, rs.permutation(q)[sel:]] = 1e5                               # q - sel symbols of every node not selectable
        o = orc.Oracle(path)
        for ee in (False, True):
            dec = pkg.Decoder(pkg.Code(path), nm, nm, 0.3, 4, max_batch=B, early_exit=ee)
            dec.load_llr(llr)
            dec.decode(B)
            d, it, sy = dec.results(B)
            for b in range(B):
                ref = o.decode(llr[b], nm, nm, 0.3, 4, early_exit=ee, want_state=True)
                assert (d[b] == ref["decide"]).all() and it[b] == ref["iters"] and sy[b] == ref["synd"], (q, ee, b)
                assert dec.app(b).tobytes() == ref["app"].tobytes() and dec.ctov(b).tobytes() == ref["ctov"].tobytes(), (q, ee, b)
            dec.close()


@pytest.mark.parametrize("block", range(int(os.environ.get("EMS_FUZZ_MID", "3"))))
def test_random_mid_size_codes_vs_oracle(pkg, orc, tmp_path, monkeypatch, block):
    """Random codes with 60..400 check nodes (many dependency levels, every wave of the persistent workgroups busy, done flags and claims across
    waves) at random workgroup sizes, frames drawn by the oracle's encoder at an SNR where some converge (early exit at different sweeps in
    one batch) and some do not; whole-frame state against the oracle."""
    rs = np.random.RandomState(8800 + block + SEED)
    ran = 0
    for case in range(4):
        q = int(rs.choice([16, 64, 256]))
        dcs = [[3], [4], [3, 4, 5], [6], [8], [5, 6, 7, 8], [12, 9]][int(rs.randint(7))]
        M = int(rs.randint(60, 401))
        deg = sorted((int(rs.choice(dcs)) for _ in range(M)), reverse=True)
        E = sum(deg)
        N = max(2 * M, E // 2)
        while True:
            slots = ([int(c) for c in rs.permutation(N)] + [int(c) for c in rs.randint(0, N, max(0, E - N))])[:E]
            rs.shuffle(slots)
            rows, k, ok = [], 0, True
            for d in deg:
                r = slots[k:k + d]; k += d
                if len(set(r)) != d:
                    if rs.rand() < 0.5:                                                # now and then a repeated variable node stays
                        rows.append(r); continue
                    ok = False; break
                rows.append(r)
            if ok:
                break
        coefs = [[int(x) for x in rs.randint(1, q, d)] for d in deg]
        path = str(tmp_path / ("mid_%d_%d" % (block, case)))
        write(path, N, M, q, deg, rows, coefs)
        nm = min(q, int(rs.choice([8, 12, 16, 20])))
        nop = int(rs.choice([nm + 3, 2 * nm]))
        itmax = int(rs.randint(3, 9))
        ee = bool(rs.randint(2))
        B = int(rs.randint(2, 6))
        o = orc.Oracle(path)
        try:
            sigma = o.sigma(float(rs.choice([3.0, 5.0, 7.0])))
            rng = orc.Rng(block * 10 + case)
            frames = [o.draw_frame(rng, sigma) for _ in range(B)]
        except RuntimeError:                                                           # rank-deficient: no encoder
            continue
        noisy = np.stack([f[1] for f in frames])
        w = int(rs.choice([0, 1, 2, 3, 5]))
        if w:
            monkeypatch.setenv("EMS_WAVES", str(w))
        dec = pkg.Decoder(pkg.Code(path), nm, nop, 0.3, itmax, max_batch=B, early_exit=ee)
        decide, iters, synd = dec.decode_batch(noisy, sigma)
        for b in range(B):
            ref = o.decode(o.llr(noisy[b], sigma), nm, nop, 0.3, itmax, early_exit=ee, want_state=True)
            ctx = (block, case, q, M, N, deg[0], deg[-1], nm, nop, itmax, ee, w, dec.last_kernel, dec.last_workgroup_waves, b)
            assert (decide[b] == ref["decide"]).all() and iters[b] == ref["iters"] and synd[b] == ref["synd"], ctx
            assert dec.app(b).tobytes() == ref["app"].tobytes() and dec.ctov(b).tobytes() == ref["ctov"].tobytes(), ctx
        dec.close()
        if w:
            monkeypatch.delenv("EMS_WAVES")
        ran += 1
    assert ran >= 2, ran


def test_more_than_255_sweeps(pkg, orc, tmp_path, monkeypatch):
    """NbIterMax = 300 with early exit off: the persistent kernels tag their per-task done flags with (sweep + 1) & 0xFF (the lane-serial GF(16)
    kernel counts sweeps per group instead) -- the tags wrap, the order of the visits must not.  N96 on the one-visit, two-visit and two-frame
    kernels, a GF(16) dc = 8 code on the lane-serial and two-visit kernels, whole-frame state against the oracle after 300 sweeps."""
    MAT = os.path.join(ROOT, "tests", "golden", "matrices")

    def run(name, nm, nop, B, ebn):
        path = os.path.join(MAT, name)
        o = orc.Oracle(path)
        sigma = o.sigma(ebn)
        rng = orc.Rng(5)
        frames = [o.draw_frame(rng, sigma) for _ in range(B)]
        noisy = np.stack([f[1] for f in frames])
        dec = pkg.Decoder(pkg.Code(path), nm, nop, 0.3, 300, max_batch=B, early_exit=False)
        d, it, sy = dec.decode_batch(noisy, sigma)
        for b in range(B):
            ref = o.decode(o.llr(noisy[b], sigma), nm, nop, 0.3, 300, early_exit=False, want_state=True)
            assert (d[b] == ref["decide"]).all() and it[b] == ref["iters"] and sy[b] == ref["synd"], (name, dec.last_kernel, b)
            assert dec.app(b).tobytes() == ref["app"].tobytes() and dec.ctov(b).tobytes() == ref["ctov"].tobytes(), (name, dec.last_kernel, b)
        k = dec.last_kernel
        dec.close()
        return k

    seen = {run("N96_K48_GF64", 16, 32, 3, -3.0)}
    for knob, val in (("EMS_DUAL", "1"), ("EMS_TWIN", "1")):
        monkeypatch.setenv(knob, val)
        seen.add(run("N96_K48_GF64", 16, 32, 3, -3.0))
        monkeypatch.delenv(knob)
    # GF(16) / dc = 8 / nm = 8 on a random regular graph, LLR block as the input
    rs = np.random.RandomState(300)
    path = str(tmp_path / "gf16_dc8")
    N, M = _random_regular_code(rs, path, 16, 8)
    llr = (rs.rand(2, N, 16) * 20).astype(np.float32)
    llr -= llr.min(axis=2, keepdims=True)
    o = orc.Oracle(path)
    refs = [o.decode(llr[b], 8, 16, 0.3, 300, early_exit=False, want_state=True) for b in range(2)]
    for kn in ({}, {"EMS_Q16T": "0"}):
        for k, v in kn.items():
            monkeypatch.setenv(k, v)
        dec = pkg.Decoder(pkg.Code(path), 8, 16, 0.3, 300, max_batch=2, early_exit=False)
        dec.load_llr(llr)
        dec.decode(2)
        d, it, sy = dec.results(2)
        for b in range(2):
            assert (d[b] == refs[b]["decide"]).all() and it[b] == refs[b]["iters"] and sy[b] == refs[b]["synd"], (dec.last_kernel, b)
            assert dec.app(b).tobytes() == refs[b]["app"].tobytes() and dec.ctov(b).tobytes() == refs[b]["ctov"].tobytes(), (dec.last_kernel, b)
        seen.add(dec.last_kernel)
        dec.close()
        for k in kn:
            monkeypatch.delenv(k)
    print(sorted(seen))
    assert len(seen) >= 5, seen


def test_gf16_travelling_layout_on_random_codes(pkg, orc, tmp_path, monkeypatch):
    """k_decode_q16t on its opt-in travelling layout (EMS_Q16T_TRAV=1; kernels/ems_q16t.inc: per edge a record and a slot for the APP row of
    its variable node, written by the node's other edge) on random (2, 8)-regular graphs over GF(16) -- who visits a node first, which
    edge is the other one and the groups of 16 differ from code to code -- with continuous, half-integer (the merge-form redo on its
    staging block) and partly saturated LLRs, early exit on and off: whole-frame state, through k_q16t_export, against the oracle; then
    a second decode on the same handle without a new LLR load (the exported state is its start) against the plain layout's."""
    monkeypatch.setenv("EMS_Q16T_TRAV", "1")
    rs = np.random.RandomState(7700 + SEED)
    for case in range(int(os.environ.get("EMS_FUZZ_REGULAR", "14"))):
        path = str(tmp_path / ("trav_%d" % case))
        N, M = _random_dv2_code(rs, path, 16, 8)
        nop = int(rs.choice([11, 16, 16, 24]))
        off = float(rs.choice([0.0, 0.3, 1.0]))
        itmax = int(rs.randint(2, 8))
        ee = bool(rs.randint(2))
        B = int(rs.randint(3, 8))
        llr = (rs.rand(B, N, 16) * 30).astype(np.float32)
        llr[1] = np.round(llr[1] * 2) / 2
        llr[2, ::3, 8:] = 1e5
        llr -= llr.min(axis=2, keepdims=True)
        llr[2, ::3, 8:] = 1e5
        o = orc.Oracle(path)
        dec = pkg.Decoder(pkg.Code(path), 8, nop, off, itmax, max_batch=B, early_exit=ee)
        dec.load_llr(llr)
        dec.decode(B)
        assert dec.last_kernel.startswith("k_decode_q16t<") and dec.last_kernel.endswith(", true>"), dec.last_kernel
        d, it, sy = dec.results(B)
        for b in range(B):
            ref = o.decode(llr[b], 8, nop, off, itmax, early_exit=ee, want_state=True)
            ctx = (case, M, N, nop, off, itmax, ee, b)
            assert (d[b] == ref["decide"]).all() and it[b] == ref["iters"] and sy[b] == ref["synd"], ctx
            assert dec.app(b).tobytes() == ref["app"].tobytes() and dec.ctov(b).tobytes() == ref["ctov"].tobytes(), ctx
        if case % 4 == 0:                              # decode again from the state the first decode left (APP as the new LLRs, C->V = 0)
            dec.load_llr(llr)
            dec.decode(B); dec.decode(B)
            a2 = [dec.app(b).copy() for b in range(B)]
            d2 = dec.results(B)[0].copy()
            monkeypatch.setenv("EMS_Q16T_TRAV", "0")
            pl = pkg.Decoder(pkg.Code(path), 8, nop, off, itmax, max_batch=B, early_exit=ee)
            monkeypatch.setenv("EMS_Q16T_TRAV", "1")
            pl.load_llr(llr)
            pl.decode(B); pl.decode(B)
            assert pl.last_kernel.endswith(", false>"), pl.last_kernel
            assert (pl.results(B)[0] == d2).all(), case
            for b in range(B):
                assert pl.app(b).tobytes() == a2[b].tobytes(), (case, b)
            pl.close()
        dec.close()
