"""Device-side frame source: the CPU statement oracle/ems_source_oracle.c.

* Philox4x32-10 against the known-answer vectors published with the generator (Random123 kat_vectors);
* the encoder against the REAL reference Encoding (tools.c:232) of oracle/_ref on the same information symbols
  (skipped where oracle/_ref is absent) and against the parity-check matrix (zero syndrome);
* the symbol / noise mapping: uniformity and the first four moments -- the stream is this repo's own, parity
  with the reference's drand48 frames is statistical by construction."""
import ctypes as C
import importlib
import os
import sys

import numpy as np
import pytest

from conftest import MAT, ROOT

N96 = os.path.join(MAT, "N96_K48_GF64")
MAT24 = os.path.join(MAT, "Mat24_N192_M96")
PI = C.POINTER(C.c_int)


@pytest.fixture(scope="module")
def osrc(orc):
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    return importlib.import_module("orc_src")


def encoder_of(refapi, path):
    s = refapi.RefSession(path, 16, need_encoder=True)
    matUT = np.ctypeslib.as_array(s.code.matUT[0], shape=(s.M * s.N,)).reshape(s.M, s.N).copy()
    perm = np.ctypeslib.as_array(s.code.Perm, shape=(s.N,)).copy()
    return s, matUT, perm


def test_philox_known_answers(osrc):
    kat = [([0, 0, 0, 0], [0, 0], "6627e8d5 e169c58d bc57ac4c 9b00dbd8"),
           ([0xFFFFFFFF] * 4, [0xFFFFFFFF] * 2, "408f276d 41c83b0e a20bc7c6 6d5451fd"),
           ([0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], [0xA4093822, 0x299F31D0], "d16cfe09 94fdcceb 5001e420 24126ea1")]
    for ctr, key, want in kat:
        assert " ".join("%08x" % x for x in osrc.philox(ctr, key)) == want


@pytest.mark.parametrize("path", [N96, MAT24])
def test_encoder_matches_real_reference(orc, osrc, refapi, path):
    """srco_encode == the reference's own GaussianElimination + Encoding on the same information symbols."""
    if not orc.ref_available():
        pytest.skip("oracle/_ref not built (no /root/reference here)")
    R = orc.ref_lib("ubs")
    code, table = orc.RefCode(), orc.RefTable()
    R.LoadCode(os.fsencode(path), C.byref(code))
    R.LoadTables(C.byref(table), code.GF, code.logGF)
    R.GaussianElimination(C.byref(code), C.byref(table))
    N, M, K, logq = code.N, code.M, code.K, code.logGF
    matUT = np.ctypeslib.as_array(code.matUT[0], shape=(M * N,)).reshape(M, N).copy()
    perm = np.ctypeslib.as_array(code.Perm, shape=(N,)).copy()
    s, our_ut, our_perm = encoder_of(refapi, path)          # the product's host GaussianElimination gives the same matrices
    assert (our_ut == matUT).all() and (our_perm == perm).all()
    s.close()
    so = osrc.SourceOracle(path, matUT, perm)
    rows = [(C.c_int * logq)() for _ in range(N)]
    NBIN = (PI * N)(*[C.cast(r, PI) for r in rows])
    for f in range(6):
        ks = so.info(11, f)
        cw_ref = np.zeros(N, np.int32)
        ksc = (C.c_int * K)(*[int(x) for x in ks])
        R.Encoding(C.byref(code), C.byref(table), cw_ref.ctypes.data_as(PI), NBIN, ksc)
        cw = so.encode(ks)
        assert (cw == cw_ref).all()
        assert so.o.syndrome(cw) == 0


def test_frames_are_codewords_and_keyed_by_frame_and_seed(osrc, refapi):
    s, matUT, perm = encoder_of(refapi, N96)
    so = osrc.SourceOracle(N96, matUT, perm)
    seen = set()
    for seed, f in ((0, 0), (0, 1), (1, 0), (7, (1 << 40) + 3), (1 << 33, 5)):
        cw, noisy = so.frame(seed, f, 0.7)
        assert so.o.syndrome(cw) == 0
        seen.add(cw.tobytes() + noisy.tobytes())
    assert len(seen) == 5                                   # every (seed, frame) is its own frame
    a, b = so.frame(3, 9, 0.5), so.frame(3, 9, 0.5)
    assert (a[0] == b[0]).all() and a[1].tobytes() == b[1].tobytes()
    s.close()


def test_symbol_uniformity_and_noise_moments(osrc, refapi):
    s, matUT, perm = encoder_of(refapi, N96)
    so = osrc.SourceOracle(N96, matUT, perm)
    q, K, N, logq = so.q, so.K, so.N, so.logq
    F = 4000
    counts = np.zeros(q, np.int64)
    z = np.zeros((F, N * logq), np.float64)
    sigma = 0.8
    for f in range(F):
        ks = so.info(5, f)
        counts += np.bincount(ks, minlength=q)
        cw = so.encode(ks)
        tx = 1.0 - 2.0 * ((so.i2b[cw][:, None] >> np.arange(logq)[None, :]) & 1)
        z[f] = ((so.noise(5, f, cw, sigma).astype(np.float64) - tx) / sigma).ravel()
    n = F * K
    chi2 = float(((counts - n / q) ** 2 / (n / q)).sum())     # q-1 = 63 degrees of freedom: mean 63, sd 11.2
    assert chi2 < 63 + 5 * 11.3, chi2
    z = z.ravel()                                              # 384000 samples
    m = len(z)
    assert abs(z.mean()) < 5 / np.sqrt(m)
    assert abs(z.var() - 1.0) < 5 * np.sqrt(2.0 / m)
    assert abs((z ** 3).mean()) < 5 * np.sqrt(15.0 / m)
    assert abs((z ** 4).mean() - 3.0) < 5 * np.sqrt(96.0 / m)
    assert 4.0 < np.abs(z).max() < 6.5                          # tails reach where 4e5 Gaussian samples should
    # no correlation between neighbouring samples (the two halves of one Philox block, consecutive blocks)
    assert abs(np.corrcoef(z[:-1], z[1:])[0, 1]) < 5 / np.sqrt(m)
    s.close()
