"""The v2 decode core: oracle/ems_v2_oracle.c against the REAL functions of
EMS_HS_L-BubbleCheck_UBS_decoder_v2.c compiled into oracle/_ref/libref_v2.so (oracle/Makefile `ref`,
driver oracle/ref_v2_shim.c).  Skipped when oracle/_ref is absent (the GPU box); tests/golden/v2_*.npz
carries reference-generated outputs for the same inputs."""
import importlib
import os
import sys

import numpy as np
import pytest

from conftest import MAT, ROOT

GF256 = os.path.join(MAT, "Ahmed_64800_R12_GF256")
GF64 = os.path.join(MAT, "N96_K48_GF64")


@pytest.fixture(scope="module")
def v2(orc):
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    return importlib.import_module("orc_v2")


class quiet:
    """The reference printf()s on every short message (v2.c:1654, :1782); keep the test log readable."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        self.null = os.open(os.devnull, os.O_WRONLY)
        os.dup2(self.null, 1)

    def __exit__(self, *exc):
        import ctypes
        ctypes.CDLL(None).fflush(None)
        os.dup2(self.saved, 1)
        os.close(self.saved)
        os.close(self.null)


@pytest.fixture(scope="module")
def ref256(v2):
    if not os.path.exists(v2.REF_V2):
        pytest.skip("oracle/_ref/libref_v2.so not built (reference tree not mounted)")
    with quiet():
        r = v2.RefV2(GF256, 20)
    yield r
    r.close()


def _msg(rs, q, nm, short, quant):
    """A message as the v2 decoder carries it: values in decreasing order, the last valid one 0, then (-1000, -1)."""
    n = nm - short
    v = np.sort((rs.randint(0, 60, nm) * quant).astype(np.float32))[::-1].copy()
    v = (v - v[n - 1]).astype(np.float32)
    g = rs.permutation(q)[:nm].astype(np.int32)
    v[n:] = -1000.0
    g[n:] = -1
    return v, g


def test_field_tables_and_graph_match_the_reference(v2, ref256):
    o = v2.V2Oracle(GF256)
    add, mul, div, binm = ref256.tables()
    assert (add == o.add).all() and (mul == o.mul).all() and (div == o.div).all()
    img = (binm * (1 << np.arange(8))[None, :]).sum(axis=1)
    assert (img == o.o.idx2bin).all()
    ntob, inter = ref256.graph()
    _cd, ntob2, inter2 = o.graph()
    assert (ntob == ntob2).all() and (inter == inter2).all()


def test_elementary_step_random(v2, ref256):
    o = v2.V2Oracle(GF256)
    rs = np.random.RandomState(42)
    with quiet():
        for it in range(4000):
            nm = 20
            a, ia = _msg(rs, 256, nm, rs.randint(0, nm - 1) if it % 3 == 0 else 0, [0.25, 1.0, 0.37][it % 3])
            b, ib = _msg(rs, 256, nm, rs.randint(0, nm - 1) if it % 5 == 0 else 0, [0.25, 1.0, 0.37][it % 3])
            if it % 7 == 0:
                ib[: nm // 2] = ia[: nm // 2]                                  # many repeated sums
            nop = int(rs.choice([3, nm, 2 * nm]))
            o1, i1, _s = o.elementary_step(a, b, ia, ib, nop)
            o2, i2 = ref256.elementary_step(a, b, ia, ib, nop)
            assert o1.tobytes() == o2.tobytes() and (i1 == i2).all(), it


def test_data_pass_random(v2, ref256):
    o = v2.V2Oracle(GF256)
    rs = np.random.RandomState(43)
    with quiet():
        for it in range(400):
            like, idx = v2.synthetic_likelihood(1, 256, 8, o.o.idx2bin, [0.5, 0.8, 1.2][it % 3], it)
            cv, icv = _msg(rs, 256, 20, rs.randint(1, 19) if it % 4 == 0 else 0, [0.25, 1.0][it % 2])
            a1, b1 = o.data_pass(like[0], idx[0], cv, icv, 0.3)
            a2, b2 = ref256.data_pass(like[0], idx[0], cv, icv, 0.3)
            assert a1.tobytes() == a2.tobytes() and (b1 == b2).all(), it


def test_check_node_random(v2, ref256):
    o = v2.V2Oracle(GF256)
    rs = np.random.RandomState(44)
    with quiet():
        for it in range(400):
            node = int(rs.randint(0, o.M))
            vc, ivc = np.zeros((4, 20), np.float32), np.zeros((4, 20), np.int32)
            for t in range(4):
                vc[t], ivc[t] = _msg(rs, 256, 20, 0, [0.5, 1.0][it % 2])   # (V->C messages are always full: DataPassLog emits nm entries)
            e0 = int(o.o.row_ptr[node])
            c1, i1 = o.check_node(o.coef[e0:e0 + 4], vc, ivc, [40, 20, 6][it % 3])
            c2, i2 = ref256.check_node(node, vc, ivc, [40, 20, 6][it % 3])
            assert c1.tobytes() == c2.tobytes() and (i1 == i2).all(), it


@pytest.mark.parametrize("sigma,itmax,seed", [(0.62, 8, 7), (0.82, 20, 11), (1.05, 4, 3)])
def test_frame_decode_synthetic_likelihoods(v2, ref256, sigma, itmax, seed):
    """Whole-frame decode (all state bit for bit): easy, slow-converging and non-converging frames."""
    o = v2.V2Oracle(GF256)
    like, idx = v2.synthetic_likelihood(o.N, 256, 8, o.o.idx2bin, sigma, seed)
    a = o.decode(like, idx, 20, 40, 0.3, itmax)
    with quiet():
        b = ref256.decode(like, idx, 40, 0.3, itmax)
    assert a["iters"] == b["iters"] and a["synd"] == b["synd"] and (a["decide"] == b["decide"]).all()
    for k in ("vtoc", "vtoci", "ctov", "ctovi", "app"):
        assert a[k].tobytes() == b[k].tobytes(), k


def test_frame_decode_reference_channel(v2, ref256):
    """Likelihoods from the reference's own channel (256-point constellation + fading, v2.c:1167-1499)."""
    o = v2.V2Oracle(GF256)
    with quiet():
        like, idx = ref256.channel_zero_codeword(11.0, 5)
        b = ref256.decode(like, idx, 40, 0.3, 10)
    a = o.decode(like, idx, 20, 40, 0.3, 10)
    assert a["iters"] == b["iters"] and a["synd"] == b["synd"] and (a["decide"] == b["decide"]).all()
    for k in ("vtoc", "vtoci", "ctov", "ctovi", "app"):
        assert a[k].tobytes() == b[k].tobytes(), k


def test_frame_decode_check_degree_8(v2):
    """Rate 3/4 GF(256) code (dc = 8): twelve elementary steps per check node instead of six."""
    if not os.path.exists(v2.REF_V2):
        pytest.skip("oracle/_ref/libref_v2.so not built")
    path = os.path.join(MAT, "Ahmed_64800_R34_GF256")
    o = v2.V2Oracle(path)
    with quiet():
        r = v2.RefV2(path, 20)
    for sigma, seed in ((0.5, 21), (0.62, 22)):
        like, idx = v2.synthetic_likelihood(o.N, 256, 8, o.o.idx2bin, sigma, seed)
        a = o.decode(like, idx, 20, 40, 0.3, 6)
        with quiet():
            b = r.decode(like, idx, 40, 0.3, 6)
        assert a["iters"] == b["iters"] and a["synd"] == b["synd"] and (a["decide"] == b["decide"]).all()
        for k in ("vtoc", "vtoci", "ctov", "ctovi", "app"):
            assert a[k].tobytes() == b[k].tobytes(), (seed, k)
    r.close()


def test_frame_decode_gf64_with_supplied_tables(v2):
    """The core functions are field-agnostic given the tables (the reference fills BINGF for GF(256) only,
    v2.c:972-978): GF(64) N=96 code, other nm / NbOper / offset."""
    if not os.path.exists(v2.REF_V2):
        pytest.skip("oracle/_ref/libref_v2.so not built")
    o = v2.V2Oracle(GF64)
    with quiet():
        r = v2.RefV2(GF64, 12, tables=(o.add, o.mul, o.div))
    for seed, sigma in ((1, 0.7), (2, 0.95), (3, 1.3)):
        like, idx = v2.synthetic_likelihood(o.N, 64, 6, o.o.idx2bin, sigma, seed)
        a = o.decode(like, idx, 12, 24, 0.5, 12)
        with quiet():
            b = r.decode(like, idx, 24, 0.5, 12)
        assert a["iters"] == b["iters"] and a["synd"] == b["synd"] and (a["decide"] == b["decide"]).all()
        for k in ("vtoc", "vtoci", "ctov", "ctovi", "app"):
            assert a[k].tobytes() == b[k].tobytes(), (seed, k)
    r.close()


@pytest.mark.parametrize("name,q,nm,nop,itmax,cases", [
    ("V2Irr_M24_GF256", 256, 20, 40, 8, ((0.55, 41), (0.8, 42), (1.1, 43))),       # irregular (2, dc): degrees 2..12, the largest in row 0
    ("V2Irr_M24_GF256", 256, 8, 11, 5, ((0.7, 44),)),
    ("V2Wide22_M10_GF256", 256, 20, 40, 6, ((0.4, 45), (0.6, 46))),                # dc = 22
    ("Ahmed_64800_R910_GF256", 256, 20, 40, 3, ((0.3, 47),)),                      # the reference's rate-9/10 matrix, dc = 20
])
def test_frame_decode_other_check_degrees(v2, name, q, nm, nop, itmax, cases):
    """Round 5: check degrees above 16 and irregular (2, dc) codes (v2.c:1756-1852 loops over rowDegree[node]; :750 / :794-800 size the
    scratch by rowDegree[0]) -- the oracle against the REAL functions, all message state."""
    if not os.path.exists(v2.REF_V2):
        pytest.skip("oracle/_ref/libref_v2.so not built")
    path = os.path.join(MAT, name)
    o = v2.V2Oracle(path)
    with quiet():
        r = v2.RefV2(path, nm)
    for sigma, seed in cases:
        like, idx = v2.synthetic_likelihood(o.N, q, 8, o.o.idx2bin, sigma, seed)
        a = o.decode(like, idx, nm, nop, 0.3, itmax)
        with quiet():
            b = r.decode(like, idx, nop, 0.3, itmax)
        assert a["iters"] == b["iters"] and a["synd"] == b["synd"] and (a["decide"] == b["decide"]).all()
        for k in ("vtoc", "vtoci", "ctov", "ctovi", "app"):
            assert a[k].tobytes() == b[k].tobytes(), (seed, k)
    r.close()
