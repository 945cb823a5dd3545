"""Device-side frame source through the C ABI (ems_source_*), against
oracle/ems_source_oracle.c: information symbols and codewords bit for bit (the encoder is the reference's,
tools.c:232), noisy samples within the float tolerance of the device's logf/cosf (the stream is this repo's
own: statistical parity with the reference's drand48 frames), and the decoder fed from it against the decoder
fed the same frames through the host path."""
import importlib
import os
import sys

import numpy as np
import pytest

from conftest import MAT, ROOT

pytestmark = pytest.mark.gpu
NOISE_ATOL = 4e-6          # |sigma * (gauss_gpu - gauss_cpu)|: a few ulp of logf / cosf at |gauss| <= 6, sigma <= 1


@pytest.fixture(scope="module")
def osrc(orc):
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    return importlib.import_module("orc_src")


def _open(pkg, refapi, osrc, mat, nm, nop, itmax, B, early_exit=True):
    path = os.path.join(MAT, mat)
    s = refapi.RefSession(path, nm, need_encoder=True)
    matUT = np.ctypeslib.as_array(s.code.matUT[0], shape=(s.M * s.N,)).reshape(s.M, s.N).copy()
    perm = np.ctypeslib.as_array(s.code.Perm, shape=(s.N,)).copy()
    so = osrc.SourceOracle(path, matUT, perm)
    dec = pkg.Decoder(pkg.Code(path), nm, nop, 0.3, itmax, max_batch=B, early_exit=early_exit)
    dec.source_attach(s)
    return s, so, dec


@pytest.mark.parametrize("mat,nm,seed,first,B,check", [
    ("N96_K48_GF64", 16, 0, 0, 64, range(64)),
    ("N96_K48_GF64", 16, 0x1234567890ABCDEF, (1 << 40) + 7, 100, range(0, 100, 7)),   # 64-bit seed and frame index, ragged batch
    ("Mat24_N192_M96", 16, 3, 5, 33, range(33)),
    ("Ahmed_64800_R34_GF16", 8, 9, 2, 65, (0, 63, 64)),                                # frames in the second wavefront of the solver
    ("Ahmed_64800_R12_GF256", 20, 1, 1000, 70, (0, 1, 69)),
])
def test_source_frames_match_oracle(pkg, refapi, osrc, mat, nm, seed, first, B, check):
    s, so, dec = _open(pkg, refapi, osrc, mat, nm, 2 * nm, 5, B)
    sigma = 0.77
    dec.source_draw(seed, first, B, sigma)
    cw, noisy = dec.source_get(B)
    for b in check:
        ocw, onoisy = so.frame(seed, first + b, sigma)
        assert (cw[b] == ocw).all(), "frame %d: codeword differs" % b
        assert np.abs(noisy[b] - onoisy).max() <= NOISE_ATOL, (b, float(np.abs(noisy[b] - onoisy).max()))
    for b in range(B):                                      # every frame is a codeword of H
        assert so.o.syndrome(cw[b]) == 0
    dec.close(); s.close()


def test_source_is_batch_and_position_independent(pkg, refapi, osrc):
    s, so, dec = _open(pkg, refapi, osrc, "N96_K48_GF64", 16, 32, 5, 64)
    dec.source_draw(42, 0, 64, 0.6)
    cw0, ny0 = dec.source_get(64)
    dec.source_draw(42, 10, 4, 0.6)
    cw1, ny1 = dec.source_get(4)
    assert (cw1 == cw0[10:14]).all() and ny1.tobytes() == ny0[10:14].tobytes()
    dec.source_draw(43, 10, 4, 0.6)
    cw2, _ = dec.source_get(4)
    assert not (cw2 == cw1).all()
    dec.close(); s.close()


def test_decode_from_source_equals_decode_from_host_copy(pkg, refapi, osrc, orc):
    """source_draw leaves the handle as set_codewords + llr_init on the same frames would: same decisions, same
    error counts, and both equal the oracle's decode of those frames."""
    B = 48
    s, so, dec = _open(pkg, refapi, osrc, "N96_K48_GF64", 16, 32, 10, B)
    sigma = so.o.sigma(2.0)
    dec.source_draw(77, 500, B, sigma)
    dec.decode(B)
    d1, it1, sy1 = dec.results(B)
    e1 = dec.count_errors_resident(B)
    cw, noisy = dec.source_get(B)
    d2, it2, sy2 = dec.decode_batch(noisy, sigma)
    e2 = dec.count_errors(cw)
    assert (d1 == d2).all() and (it1 == it2).all() and (sy1 == sy2).all() and (e1 == e2).all()
    for b in range(0, B, 5):
        ref = so.o.decode(so.o.llr(noisy[b], sigma), 16, 32, 0.3, 10)
        assert it1[b] == ref["iters"] and sy1[b] == ref["synd"] and (d1[b] == ref["decide"]).all()
        assert e1[b] == so.o.bit_errors(ref["decide"], cw[b])
    dec.close(); s.close()


def test_source_statistics_on_device(pkg, refapi, osrc):
    """Uniform information symbols and unit-variance Gaussian noise over one large draw."""
    B = 4096
    s, so, dec = _open(pkg, refapi, osrc, "N96_K48_GF64", 16, 32, 5, B)
    sigma = 0.9
    dec.source_draw(2024, 0, B, sigma)
    cw, noisy = dec.source_get(B)
    q, logq = so.q, so.logq
    tx = 1.0 - 2.0 * ((so.i2b[cw][:, :, None] >> np.arange(logq)[None, None, :]) & 1)
    z = ((noisy.astype(np.float64) - tx) / sigma).ravel()
    m = len(z)
    assert abs(z.mean()) < 5 / np.sqrt(m) and abs(z.var() - 1.0) < 5 * np.sqrt(2.0 / m)
    assert abs((z ** 3).mean()) < 5 * np.sqrt(15.0 / m) and abs((z ** 4).mean() - 3.0) < 5 * np.sqrt(96.0 / m)
    counts = np.bincount(cw.ravel(), minlength=q).astype(np.float64)          # codeword symbols of a linear code are uniform too
    n = cw.size
    assert ((counts - n / q) ** 2 / (n / q)).sum() < 63 + 6 * 11.3
    dec.close(); s.close()


def test_frame_error_rate_agrees_with_reference_stream(pkg, refapi, osrc, harness):
    """Statistical parity: FER of the N96 code at 3.0 dB from the device stream against the reference's
    drand48 stream (K2 operating point), 40000 frames each, difference within 4.5 standard deviations."""
    path = os.path.join(MAT, "N96_K48_GF64")
    F, B = 40000, 8192
    s, so, dec = _open(pkg, refapi, osrc, "N96_K48_GF64", 16, 32, 10, B)
    sigma = so.o.sigma(3.0)
    err_dev = 0
    for first in range(0, F, B):
        n = min(B, F - first)
        dec.source_draw(1, first, n, sigma)
        dec.decode(n)
        err_dev += int((dec.count_errors_resident(n) != 0).sum())
    src = harness.FrameSource(path, 16)
    err_ref = 0
    for first in range(0, F, B):
        n = min(B, F - first)
        cws, noisy, sg = src.draw(n, 3.0)
        assert sg == sigma
        dec.decode_batch(noisy, sg)
        err_ref += int((dec.count_errors(cws) != 0).sum())
    src.close()
    p = (err_dev + err_ref) / (2.0 * F)
    sd = np.sqrt(2.0 * p * (1 - p) / F)
    assert 0.005 < p < 0.05, (err_dev, err_ref)
    assert abs(err_dev - err_ref) / F < 4.5 * sd, (err_dev, err_ref, sd)
    dec.close(); s.close()


def test_monte_carlo_driver_with_device_source(pkg, harness, kat):
    """harness.monte_carlo(source="device"): same frames whatever the batch size; FER at the K2 operating point
    in the range the reference stream gives (3/200)."""
    path = os.path.join(MAT, "N96_K48_GF64")
    k = kat["K2"]
    a, _ = harness.monte_carlo(path, 3000, k["NbIterMax"], k["EbN"], k["nm"], k["offset"], k["NbOper"], batch=512, source="device", seed=5)
    b, _ = harness.monte_carlo(path, 3000, k["NbIterMax"], k["EbN"], k["nm"], k["offset"], k["NbOper"], batch=77, source="device", seed=5,
                               inflight=1)
    assert [list(t) for t in a.trace] == [list(t) for t in b.trace]
    assert a.err_frames == 40 or a.nb == 3000
    assert 0.004 < a.err_frames / a.nb < 0.04, (a.err_frames, a.nb)


def test_c_harness_with_device_source(pkg, tmp_path):
    import re
    import subprocess
    os.makedirs(os.path.join(str(tmp_path), "data"), exist_ok=True)
    outs = []
    for batch in (256, 100):
        out = subprocess.run([pkg.SIM_PATH, "2000", "10", os.path.join(MAT, "N96_K48_GF64"), "3.0", "16", "0.3", "32"], cwd=str(tmp_path),
                             env=dict(os.environ, EMS_BATCH=str(batch), EMS_GPU_SOURCE="5"), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                             timeout=600)
        assert out.returncode == 0, out.stderr.decode()[-2000:]
        txt = out.stdout.decode("latin-1")
        outs.append(re.findall(r"<\d+> FER= (\d+) / (\d+) = [\d.]+ BER= (\d+) / x = [\d.]+\s+avr_it=([\d.]+)", txt))
    assert outs[0] == outs[1] and len(outs[0]) > 100
    ef, nb = int(outs[0][-1][0]), int(outs[0][-1][1])
    assert ef == 40 or nb == 2000
    assert 0.004 < ef / nb < 0.04


def test_source_error_paths(pkg, refapi):
    path = os.path.join(MAT, "N96_K48_GF64")
    s = refapi.RefSession(path, 16, need_encoder=True)
    dec = pkg.Decoder(pkg.Code(path), 16, 32, 0.3, 5, max_batch=8)
    with pytest.raises(pkg.EmsError) as e:                  # not attached yet
        dec.source_draw(1, 0, 4, 0.5)
    assert e.value.code == pkg.EMS_EINVAL
    N, M = s.N, s.M
    matUT = np.ctypeslib.as_array(s.code.matUT[0], shape=(M * N,)).reshape(M, N).copy()
    perm = np.ctypeslib.as_array(s.code.Perm, shape=(N,)).copy()
    L = pkg.lib()
    bad = perm.copy(); bad[0] = bad[1]                      # not a permutation
    assert L.ems_source_attach(dec.h, pkg._ptr(matUT), pkg._ptr(bad)) == pkg.EMS_EINVAL
    low = matUT.copy(); low[3, 1] = 5                       # below the diagonal
    assert L.ems_source_attach(dec.h, pkg._ptr(low), pkg._ptr(perm)) == pkg.EMS_EINVAL
    zp = matUT.copy(); zp[2, 2] = 0                         # zero pivot
    assert L.ems_source_attach(dec.h, pkg._ptr(zp), pkg._ptr(perm)) == pkg.EMS_EINVAL
    assert L.ems_source_attach(dec.h, pkg._ptr(matUT), pkg._ptr(perm)) == 0
    with pytest.raises(pkg.EmsError):                       # batch above the handle's capacity
        dec.source_draw(1, 0, 9, 0.5)
    dec.source_draw(1, 0, 8, 0.5)
    dec.close(); s.close()


@pytest.mark.parametrize("block", range(int(os.environ.get("EMS_FUZZ_SOURCE", "2"))))
def test_source_on_random_codes(pkg, refapi, osrc, orc, tmp_path, block):
    """The device-side encoder / noise source on the seeded random codes of tests/fuzz_codes.py (mixed check degrees, three fields, repeated
    variable nodes, column degrees 0..n): codewords bit for bit against the source oracle, noisy samples within the float tolerance, and
    every frame in the null space the reference's encoder sees (one coefficient per (row, column), tools.c:143-230)."""
    from fuzz_codes import simulation_case
    ran = 0
    for seed in range(block * 10, block * 10 + 10):
        path = str(tmp_path / ("rnd_%d" % seed))
        frames, it, eb, nm, off, nop = simulation_case(seed, path)
        try:                                                     # a rank-deficient matrix makes GaussianElimination exit() as the reference's does
            orc.Oracle(path).draw_frame(orc.Rng(0), 0.5)         # (tools.c:181-185): ask the oracle first
        except RuntimeError as e:
            assert "not full rank" in str(e), (seed, str(e))
            continue
        s = refapi.RefSession(path, nm, need_encoder=True)
        matUT = np.ctypeslib.as_array(s.code.matUT[0], shape=(s.M * s.N,)).reshape(s.M, s.N).copy()
        perm = np.ctypeslib.as_array(s.code.Perm, shape=(s.N,)).copy()
        so = osrc.SourceOracle(path, matUT, perm)
        B = 5 + seed % 9
        dec = pkg.Decoder(pkg.Code(path), nm, nop, off, it, max_batch=B)
        dec.source_attach(s)
        sigma = 0.5 + 0.05 * (seed % 7)
        dec.source_draw(1000 + seed, 3 * seed, B, sigma)
        cw, noisy = dec.source_get(B)
        for b in range(B):
            ocw, onoisy = so.frame(1000 + seed, 3 * seed + b, sigma)
            assert (cw[b] == ocw).all(), (seed, b)
            assert np.abs(noisy[b] - onoisy).max() <= NOISE_ATOL, (seed, b)
        dec.close(); s.close()
        ran += 1
    assert ran >= 7, ran


# ---- round 5: the REFERENCE's own drand48 stream generated on the device (ems_source_draw_reference) ----
@pytest.mark.parametrize("mat,nm,first,B", [
    ("N96_K48_GF64", 16, 0, 2048),
    ("N96_K48_GF64", 16, 123457, 300),
    ("Mat24_N192_M96", 16, 5, 333),
    ("Ahmed_64800_R34_GF16", 8, 2, 130),
    ("Ahmed_64800_R12_GF256", 20, 0, 96),
    ("Ahmed_64800_R12_GF256", 20, 968, 70),                     # where kat_long.json L4 ends
    ("Ahmed_64800_R910_GF256", 20, 3, 40),
])
def test_reference_stream_on_the_device_equals_the_host_stream(pkg, refapi, osrc, harness, mat, nm, first, B):
    """Codewords and every noisy sample of frames [first, first + B) bit for bit against ems_draw_frames (the host generator that the
    reference traces pin): exact 48-bit LCG with jump-ahead, the reference's float / double promotions, device-side Box-Muller in double with
    the undecidable float roundings recomputed by the host's libm."""
    s, so, dec = _open(pkg, refapi, osrc, mat, nm, 2 * nm, 5, B)
    src = harness.FrameSource(os.path.join(MAT, mat), nm)
    src.position(first)
    cw_h, noisy_h, sigma = src.draw(B, 2.0)
    n_host = dec.source_draw_reference(first, B, sigma)
    cw, noisy = dec.source_get(B)
    assert (cw == cw_h).all(), "codewords differ"
    assert noisy.tobytes() == noisy_h.tobytes(), "noisy samples differ in %d places (host recomputed %d)" % (int((noisy.view(np.uint32) != noisy_h.view(np.uint32)).sum()), n_host)
    assert n_host <= 16 + noisy.size // 20000                   # ~1.5e-5 of the samples are undecidable on the device (mostly samples near 0, where floats are dense)
    src.close(); dec.close(); s.close()


def test_reference_stream_on_the_device_feeds_the_decoder(pkg, refapi, osrc, harness):
    """... and leaves the handle as ems_set_codewords + ems_llr_init would: the decode of device-drawn frames = the decode of the host's."""
    mat, nm, B = "N96_K48_GF64", 16, 200
    s, so, dec = _open(pkg, refapi, osrc, mat, nm, 32, 10, B)
    src = harness.FrameSource(os.path.join(MAT, mat), nm)
    src.position(40)
    cw_h, noisy_h, sigma = src.draw(B, 2.5)
    dec.source_draw_reference(40, B, sigma)
    dec.decode(B)
    d1, it1, sy1 = dec.results(B)
    nb1 = dec.count_errors_resident(B)
    d2, it2, sy2 = dec.decode_batch(noisy_h, sigma)
    nb2 = dec.count_errors(cw_h)
    assert (d1 == d2).all() and (it1 == it2).all() and (sy1 == sy2).all() and (nb1 == nb2).all() and nb1.sum() > 0
    src.close(); dec.close(); s.close()


def test_monte_carlo_driver_with_the_reference_stream_on_the_device(pkg, harness, kat):
    """harness.monte_carlo(source="device_reference"): the traces of the REAL executable (K2 on N = 96; K3 on the 64800-bit GF(16) code) with
    the frames generated on the GPU."""
    for name, batch in (("K2", 64), ("K3", 3)):
        k = kat[name]
        c, _ = harness.monte_carlo(os.path.join(MAT, k["matrix"].replace(".txt", "") if k["fmt"] == "kn" else k["matrix"]), k["frames"], k["NbIterMax"], k["EbN"],
                                   k["nm"], k["offset"], k["NbOper"], batch=batch, source="device_reference")
        assert [list(t) for t in c.trace] == [list(t) for t in k["trace"]], name
