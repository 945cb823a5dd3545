"""Live differential tests: the oracle restatement against the REAL reference compiled into oracle/_ref
(recipe: oracle/Makefile).  Skipped when oracle/_ref is absent; the committed golden fixtures cover the
same ground without it."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import MAT, FieldOracle


@pytest.fixture(scope="module")
def ref(orc):
    if not orc.ref_available():
        pytest.skip("oracle/_ref not built (reference tree not mounted)")
    return orc.ref_lib("ubs")


def _rand_list(rs, q, nm, short, quant):
    v = np.sort((rs.randint(0, 80, nm) * quant).astype(np.float32))
    v[0] = 0
    g = rs.permutation(q)[:nm].astype(np.int32)
    if short:
        v[nm - short:] = 1e5
        g[nm - short:] = -1
    return v, g


@pytest.mark.parametrize("q,lq,nm", [(16, 4, 8), (64, 6, 16), (256, 8, 20), (256, 8, 5), (64, 6, 30)])
def test_elementary_step_random(orc, ref, q, lq, nm, capfd):
    t = orc.RefTable()
    ref.LoadTables(C.byref(t), q, lq)
    o = FieldOracle(orc, q)
    rs = np.random.RandomState(q * 100 + nm)
    nm = min(nm, q)
    for it in range(1500):
        a, ia = _rand_list(rs, q, nm, rs.randint(0, min(5, nm - 1)) if it % 4 == 0 else 0, [0.25, 1.0, 0.03125][it % 3])
        b, ib = _rand_list(rs, q, nm, rs.randint(0, min(5, nm - 1)) if it % 6 == 0 else 0, [0.25, 1.0, 0.03125][it % 3])
        nop = int(rs.choice([nm, nm + 2, 2 * nm]))
        out = np.zeros(nm, np.float32)
        io = np.zeros(nm, np.int32)
        ref.ElementaryStep(a.ctypes, b.ctypes, ia.ctypes, ib.ctypes, out.ctypes, io.ctypes, t.ADDGF, q, nm, nop)
        got, gio = o.elementary_step(a, b, ia, ib, nm, nop)
        assert got.tobytes() == out.tobytes() and (gio == io).all()
    capfd.readouterr()  # drop the reference's "out of bounds" chatter


def test_channel_llr_full_frame(orc, ref):
    """Every LLR of one frame of the headline code, bit for bit (float/double promotion order)."""
    path = os.path.join(MAT, "Ahmed_64800_R12_GF256")
    code, table, dec = orc.RefCode(), orc.RefTable(), orc.RefDecoder()
    ref.LoadCode(os.fsencode(path), C.byref(code))
    ref.LoadTables(C.byref(table), code.GF, code.logGF)
    dec.nbMax = 20
    ref.AllocateDecoder(C.byref(code), C.byref(dec))
    o = orc.Oracle(path)
    rs = np.random.RandomState(3)
    cw = rs.randint(0, 256, code.N).astype(np.int32)
    PI = C.POINTER(C.c_int)
    rows = [(C.c_int * 8)(*[(int(o.idx2bin[s]) >> l) & 1 for l in range(8)]) for s in cw]
    NBIN = (PI * code.N)(*[C.cast(r, PI) for r in rows])
    C.CDLL("libc.so.6").seed48((C.c_ushort * 3)(0, 0, 0))
    idum = C.c_int(-1)
    ref.ModelChannel_AWGN_BPSK(C.byref(code), C.byref(dec), C.byref(table), NBIN, C.c_float(1.0), C.byref(idum))
    rng = orc.Rng(0)
    sigma = o.sigma(1.0)
    noisy = np.zeros((o.N, o.logq), np.float32)
    o.L.orc_awgn_bpsk(C.byref(rng), o.code_p, C.byref(o.gf), cw.ctypes.data_as(PI), C.c_float(sigma),
                      noisy.ctypes.data_as(C.POINTER(C.c_float)))
    llr = o.llr(noisy, sigma)
    sl = np.ctypeslib.as_array(dec.intrinsic_LLR[0], (o.N, o.q))
    sg = np.ctypeslib.as_array(dec.intrinsic_GF[0], (o.N, o.q))
    app = np.zeros_like(llr)
    np.put_along_axis(app, sg.astype(np.int64), sl, axis=1)
    assert app.tobytes() == llr.tobytes()
    n = 300
    s2, g2 = o.sort_llr(llr[:n])
    assert s2.tobytes() == sl[:n].tobytes() and (g2 == sg[:n]).all()
