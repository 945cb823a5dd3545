"""CPU-side tests of the product library: the C ABI loads and exports every declared symbol, the host
logic (loader, field tables, level schedule, encoder, RNG positioning, sharding) is right, and compute
entry points fail loudly -- never fall back -- when no GPU is present."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import GOLD, MAT, ROOT, gold


def _declared_symbols():
    names = set()
    for hdr in ("ems_abi.h", "nbldpc_ref_api.h"):
        txt = open(os.path.join(ROOT, "include", hdr)).read()
        txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
        names |= set(re.findall(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\([^;{}]*\)\s*;", txt))
    names -= {"defined"}
    return sorted(names)


def test_library_exports_every_declared_symbol(pkg):
    L = pkg.lib()
    syms = _declared_symbols()
    assert len(syms) > 40
    for s in syms:
        assert hasattr(L, s), "libems_hip.so does not export " + s
    assert b"gfx950" in L.ems_version()


def test_gf_tables_via_abi(pkg):
    t = gold("tables.npz")
    for q, lq in ((16, 4), (64, 6), (256, 8)):
        i2b, b2i = pkg.gf_tables(q)
        img = (t["bin%d" % q].astype(np.int32) << np.arange(lq)).sum(1)
        assert (i2b == img).all() and (b2i[i2b] == np.arange(q)).all()
    with pytest.raises(pkg.EmsError) as e:
        pkg.gf_tables(32)
    assert e.value.code == pkg.EMS_EUNSUPPORTED


def test_reference_table_symbols(pkg, refapi):
    """LoadTables / Table_*_GF of the drop-in boundary reproduce the reference's q x q tables."""
    t = gold("tables.npz")
    L = pkg.lib()
    for q, lq in ((16, 4), (64, 6), (256, 8)):
        tb = refapi.table_t()
        L.LoadTables(C.byref(tb), q, lq)
        assert (refapi.block(tb.BINGF, q, lq, np.int32) == t["bin%d" % q]).all()
        assert (refapi.block(tb.ADDGF, q, q, np.int32) == t["add%d" % q]).all()
        assert (refapi.block(tb.MULGF, q, q, np.int32) == t["mul%d" % q]).all()
        assert (refapi.block(tb.DIVGF, q, q, np.int32) == t["div%d" % q]).all()
        L.FreeTable(C.byref(tb))


def test_loader_both_layouts_via_abi(pkg):
    g = gold("loader.npz")
    for name in ("N96_K48_GF64", "KN_N96_K48_GF64.txt"):
        c = pkg.Code(os.path.join(MAT, name))
        assert (c.N, c.M, c.GF, c.logGF, c.E, c.dc) == (16, 8, 64, 6, 32, 4)
        assert (c.col.reshape(8, 4) == g["ubs_mat"]).all() and (c.coef.reshape(8, 4) == g["ubs_val"]).all()
    # forcing the wrong layout must not silently mis-parse into the same code
    wrong = pkg.Code(os.path.join(MAT, "N96_K48_GF64"), fmt=2)
    assert not (wrong.col.reshape(8, 4) == g["ubs_mat"]).all()
    with pytest.raises(pkg.EmsError) as e:
        pkg.Code(os.path.join(MAT, "does_not_exist"))
    assert e.value.code == pkg.EMS_EIO


@pytest.mark.parametrize("name,levels", [("N96_K48_GF64", 4), ("Ahmed_64800_R34_GF16", 21), ("Ahmed_64800_R12_GF256", 441)])
def test_level_schedule_properties(pkg, name, levels):
    c = pkg.Code(os.path.join(MAT, name))
    lp, order = pkg.level_schedule(c)
    assert len(lp) - 1 == levels                      # SURVEY 0.4 [measured]
    assert sorted(order) == list(range(c.M)) and lp[0] == 0 and lp[-1] == c.M
    col = c.col.reshape(c.M, c.dc)
    level_of = np.zeros(c.M, int)
    for l in range(levels):
        cns = order[lp[l]: lp[l + 1]]
        level_of[cns] = l
        vns = col[cns].ravel()
        assert len(set(vns)) == len(vns), "two check nodes of a level share a variable node"
    last = {}
    for m in range(c.M):                               # every reference-order dependency goes upward
        for v in col[m]:
            if v in last:
                assert level_of[last[v]] < level_of[m]
            last[v] = m


@pytest.mark.parametrize("name,nm,nop,ebn", [("N96_K48_GF64", 16, 32, 2.0), ("Ahmed_64800_R34_GF16", 8, 16, 3.0)])
def test_level_order_equals_sequential_order(pkg, orc, name, nm, nop, ebn):
    """Visiting check nodes level by level (reversed inside a level) is the reference sweep, bit for bit."""
    path = os.path.join(MAT, name)
    c = pkg.Code(path)
    lp, order = pkg.level_schedule(c)
    rev = np.concatenate([order[lp[l]: lp[l + 1]][::-1] for l in range(len(lp) - 1)])
    o = orc.Oracle(path)
    sigma = o.sigma(ebn)
    _cw, noisy = o.draw_frame(orc.Rng(0), sigma)
    llr = o.llr(noisy, sigma)
    a = o.decode(llr, nm, nop, 0.3, 6, want_state=True)
    b = o.decode(llr, nm, nop, 0.3, 6, order=rev, want_state=True)
    assert a["iters"] == b["iters"] and a["synd"] == b["synd"] and (a["decide"] == b["decide"]).all()
    assert a["app"].tobytes() == b["app"].tobytes() and a["ctov"].tobytes() == b["ctov"].tobytes()


def test_host_source_encoder_noise_match_oracle(pkg, orc, refapi):
    """LoadCode/LoadTables/GaussianElimination/RandomBinaryGenerator/Encoding/noise of the product host
    code draw the very frames the reference does (same drand48 stream, X0 = 0)."""
    for name, ebn in (("N96_K48_GF64", 2.0), ("Mat24_N192_M96", 1.75)):
        path = os.path.join(MAT, name)
        s = refapi.RefSession(path, 16)
        o = orc.Oracle(path)
        rng = orc.Rng(0)
        refapi.seed48(0)
        sg = o.sigma(ebn)
        for _ in range(5):
            cw = s.draw_codeword()
            sigma, noisy = s.noise(ebn)
            ocw, onoisy = o.draw_frame(rng, sg)
            assert (cw == ocw).all() and sigma == sg and noisy.tobytes() == onoisy.tobytes()
            assert o.syndrome(cw) == 0
        s.close()


def test_lcg_jump_and_sharding(harness, orc):
    r = orc.Rng(0)
    L = orc.lib()
    for _ in range(1000):
        L.orc_drand48(orc.C.byref(r))
    assert harness.lcg_jump(0, 1000) == r.x
    assert harness.lcg_jump(harness.lcg_jump(0, 123456789), 987654321) == harness.lcg_jump(0, 123456789 + 987654321)
    assert harness.draws_per_frame(4050, 8100, 8) == 162000
    seen = []
    for rnd in range(3):
        for rk in range(4):
            st, cnt = harness.batch_of_rank(rnd, rk, 4, 8, 70)
            seen += list(range(st, st + cnt))
    assert seen == list(range(70))


def test_frame_source_random_access(harness, orc):
    """Positioning the stream at frame f gives the frame the sequential harness would draw there."""
    path = os.path.join(MAT, "N96_K48_GF64")
    src = harness.FrameSource(path, 16)
    o = orc.Oracle(path)
    rng = orc.Rng(0)
    sg = o.sigma(2.0)
    seq = [o.draw_frame(rng, sg) for _ in range(7)]
    for f in (5, 0, 3):
        src.position(f)
        cws, noisy, sigma = src.draw(2, 2.0)
        for i in range(2):
            assert (cws[i] == seq[f + i][0]).all() and noisy[i].tobytes() == seq[f + i][1].tobytes()
    src.close()


@pytest.mark.parametrize("mat,nm,count,threads,first", [("N96_K48_GF64", 16, 7, 3, 11), ("Mat24_N192_M96", 24, 5, 8, 0),
                                                        ("Ahmed_64800_R34_GF16", 8, 3, 2, 2), ("Ahmed_64800_R12_GF256", 20, 3, 3, 1)])
def test_threaded_frame_generation_equals_reference_call_sequence(harness, mat, nm, count, threads, first):
    """ems_draw_frames (frame-parallel, own LCG with jump-ahead, chunked back-substitution) against the
    reference's own call sequence RandomBinaryGenerator -> Encoding -> noise draw on libc's drand48."""
    path = os.path.join(MAT, mat)
    par = harness.FrameSource(path, nm, threads=threads)
    par.position(first)
    cw_p, ny_p, sg_p = par.draw(count, 1.5)
    par.position(first)
    par.threads = 1
    cw_s, ny_s, sg_s = par.draw(count, 1.5)
    assert sg_p == sg_s and (cw_p == cw_s).all() and ny_p.tobytes() == ny_s.tobytes()
    cw_n, ny_n, _ = par.draw(1, 1.5)                    # the stream continues where the batch ended ...
    par.threads = threads
    par.position(first + count)
    cw_m, ny_m, _ = par.draw(1, 1.5)                    # ... for either path
    assert (cw_n == cw_m).all() and ny_n.tobytes() == ny_m.tobytes()
    par.close()


def test_full_alist_layout(pkg, orc, tmp_path):
    """the full non-binary alist of matrices/KN/N64800_*.txt (dvmax dcmax line, per-column AND per-row
    sections), which neither loader of the reference reads.  Product loader == oracle loader == an independent parse
    of the row section; a file whose column section contradicts its row section is rejected by both."""
    path = os.path.join(MAT, "KN_N64800_K48600_GF256_fullalist.txt")
    tok = np.array(open(path).read().split(), np.int64)
    N, M, q = (int(x) for x in tok[:3])
    cdeg, rdeg = tok[5:5 + N], tok[5 + N:5 + N + M]
    E = int(rdeg.sum())
    assert tok[3] == cdeg.max() and tok[4] == rdeg.max() and len(tok) == 5 + N + M + 4 * E
    rsec = tok[5 + N + M + 2 * E:].reshape(E, 2)
    o = orc.Oracle(path)
    assert (o.N, o.M, o.q, o.E, o.dc) == (N, M, q, E, 8)
    assert (o.col == rsec[:, 0] - 1).all() and (o.coef == rsec[:, 1] + 1).all() and (o.row_deg == rdeg).all()
    c = pkg.Code(path)
    d = c.desc
    assert (d.N, d.M, d.GF, d.nbBranch) == (N, M, q, E)
    assert (np.ctypeslib.as_array(d.col, (E,)) == o.col).all() and (np.ctypeslib.as_array(d.coef, (E,)) == o.coef).all()
    assert (np.ctypeslib.as_array(d.rowDegree, (M,)) == rdeg).all()
    bad = tok.copy()
    bad[5 + N + M + 1] += 1                                  # first exponent of the column section
    p2 = tmp_path / "bad_alist.txt"
    p2.write_text(" ".join(str(int(x)) for x in bad))
    with pytest.raises(IOError):
        orc.Oracle(str(p2))
    with pytest.raises(pkg.EmsError):
        pkg.Code(str(p2))


def test_compute_entry_points_fail_loudly_without_gpu(pkg):
    """No CPU fallback: on a box without a HIP device every compute call reports EMS_ENODEV."""
    c = pkg.Code(os.path.join(MAT, "N96_K48_GF64"))
    try:
        d = pkg.Decoder(c, 16, 32, 0.3, 10, max_batch=2)
    except pkg.EmsError as e:
        assert e.code == pkg.EMS_ENODEV and "no CPU fallback" in str(e)
        with pytest.raises(pkg.EmsError) as e2:
            pkg.ElementaryStep(np.zeros(8, np.float32), np.zeros(8, np.float32), np.arange(8), np.arange(8), 16, 8, 16)
        assert e2.value.code == pkg.EMS_ENODEV
        with pytest.raises(pkg.EmsError):
            pkg.Decision(np.zeros((4, 16), np.float32))
    else:
        d.close()  # a GPU is present: the gpu-marked tests exercise the device path


def test_parameter_validation(pkg):
    c = pkg.Code(os.path.join(MAT, "N96_K48_GF64"))
    for nm in (4, 41, 65):
        with pytest.raises(pkg.EmsError) as e:
            pkg.Decoder(c, nm, 32, 0.3, 10, max_batch=2)
        assert e.value.code in (pkg.EMS_EINVAL, pkg.EMS_ENODEV)


def test_unit_entry_points_validate_symbol_indices(pkg):
    """ElementaryStep / CheckPassLogEMS index q-entry device tables with the caller's symbols: anything outside
    {-1, 0..GF-1} is refused on the host (before a device is even looked for)."""
    z = np.zeros(8, np.float32)
    for bad in (16, 300, -2):
        idx = np.arange(8)
        idx[3] = bad
        with pytest.raises(pkg.EmsError) as e:
            pkg.ElementaryStep(z, z, idx, np.arange(8), 16, 8, 16)
        assert e.value.code == pkg.EMS_EINVAL
        with pytest.raises(pkg.EmsError) as e:
            pkg.ElementaryStep(z, z, np.arange(8), idx, 16, 8, 16)
        assert e.value.code == pkg.EMS_EINVAL
        g = np.tile(np.arange(8), (4, 1))
        g[2, 5] = bad
        with pytest.raises(pkg.EmsError) as e:
            pkg.CheckPassLogEMS(np.array([1, 2, 3, 4]), np.zeros((4, 8), np.float32), g, 16, 16, 0.3)
        assert e.value.code == pkg.EMS_EINVAL


@pytest.mark.parametrize("name", ["N96_K48_GF64", "Ahmed_64800_R12_GF256", "Ahmed_64800_R34_GF16", "Ahmed_64800_R34_GF256", "Mat24_N192_M96"])
def test_pair_schedule_properties(pkg, name):
    """The pairs the two-visits-per-wave kernels claim: every check node exactly once; the members of a pair share no
    variable node; for every variable node its visitors come in reference order, in different pairs, the earlier pair
    first -- so visiting the pairs in order, both members at once, is the sequential sweep of NB_LDPC.c:320."""
    c = pkg.Code(os.path.join(MAT, name))
    pairs = pkg.pair_schedule(c)
    cns = pairs[pairs >= 0]
    assert sorted(cns.tolist()) == list(range(c.M)) and (pairs[:, 0] >= 0).all()
    col = c.col.reshape(c.M, c.dc)
    pair_of = np.zeros(c.M, int)
    for i, (a, b) in enumerate(pairs):
        pair_of[a] = i
        if b >= 0:
            pair_of[b] = i
            assert not set(col[a]) & set(col[b])
    last = {}
    for m in range(c.M):                                    # reference order
        for v in col[m]:
            if v in last:
                assert pair_of[last[v]] < pair_of[m], "variable node %d: check node %d must come in an earlier pair than %d" % (v, last[v], m)
            last[v] = m
    assert len(pairs) <= (c.M + 1) // 2 + max(4, c.M // 50)        # nearly everything is paired
