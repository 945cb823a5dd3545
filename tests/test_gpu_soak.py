"""Opt-in soak (EMS_SOAK=<distinct frames per case>, -m gpu): many distinct frames over a range of operating points, tiled
to the batch sizes that select every kernel shape of the headline configurations, against the CPU oracle on all host
cores.  Not part of the default suite (minutes of oracle time); the numbers of the last run are in DESIGN.md §2."""
import importlib
import os
import sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

from conftest import MAT

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not os.environ.get("EMS_SOAK"), reason="opt-in: EMS_SOAK=<frames per case>")]


@pytest.fixture(scope="module")
def v2(orc):
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    return importlib.import_module("orc_v2")


CASES = [  # name, nm, NbOper, offset, NbIterMax, Eb/N0 points, batches (each selects another kernel shape)
    ("Ahmed_64800_R12_GF256", 20, 40, 0.3, 30, (1.45, 1.6, 1.75, 1.9), (256, 640, 1024, 1536)),
    ("Ahmed_64800_R34_GF16", 8, 16, 0.3, 30, (1.8, 2.3, 2.6, 2.9, 3.1), (256,)),     # 1.8 / 2.3 dB: frames that run all 29 sweeps
    ("N96_K48_GF64", 16, 32, 0.3, 30, (0.0, 2.0, 3.5), (512, 8192)),
]


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_soak_decode_vs_oracle(pkg, orc, case):
    name, nm, nop, off, itmax, ebns, batches = case
    F = int(os.environ["EMS_SOAK"])
    path = os.path.join(MAT, name)
    o = orc.Oracle(path)
    rng = orc.Rng(12345)
    noisy, sig = [], []
    for i in range(F):
        s = o.sigma(ebns[i % len(ebns)])
        noisy.append(o.draw_frame(rng, s)[1])
        sig.append(s)
    noisy = np.stack(noisy)
    threads = os.cpu_count() or 8

    def one(i):
        r = o.decode(o.llr(noisy[i], sig[i]), nm, nop, off, itmax, early_exit=True, want_state=False)
        return r["decide"], r["iters"], r["synd"]

    with ThreadPoolExecutor(threads) as ex:
        ref = list(ex.map(one, range(F)))
    code = pkg.Code(path)
    # one sigma per launch: frames are grouped by operating point, each group tiled to the batch
    for B in batches:
        dec = pkg.Decoder(code, nm, nop, off, itmax, max_batch=B, early_exit=True)
        for k, e in enumerate(ebns):
            ids = [i for i in range(F) if i % len(ebns) == k]
            if not ids:
                continue
            tile = [ids[j % len(ids)] for j in range(B)]
            decide, iters, synd = dec.decode_batch(noisy[tile], sig[ids[0]])
            for j, i in enumerate(tile):
                assert iters[j] == ref[i][1] and synd[j] == ref[i][2] and (decide[j] == ref[i][0]).all(), (name, B, e, j, i)
        dec.close()
    its = np.array([r[1] for r in ref])
    print("soak %s: %d distinct frames, iterations min/median/max %d/%d/%d, %d not converged" %
          (name, F, its.min(), int(np.median(its)), its.max(), int(sum(r[2] != 0 for r in ref))))


def test_soak_v2_flavour_vs_oracle(pkg, v2):
    """The older generation's decode: min(EMS_SOAK, 96) distinct frames (synthetic likelihoods at four noise levels), tiled to the
    batches of its three kernel shapes."""
    F = min(int(os.environ["EMS_SOAK"]), 96)
    path = os.path.join(MAT, "Ahmed_64800_R12_GF256")
    o = v2.V2Oracle(path)
    sigmas = (0.6, 0.7, 0.8, 0.9)
    frames = [v2.synthetic_likelihood(o.N, 256, 8, o.o.idx2bin, sigmas[i % 4], 1000 + i) for i in range(F)]
    with ThreadPoolExecutor(os.cpu_count() or 8) as ex:
        ref = list(ex.map(lambda fr: o.decode(fr[0], fr[1], 20, 40, 0.3, 29), frames))
    code = pkg.Code(path)
    for B in (256, 600, 1024, 1536):
        dec = pkg.Decoder(code, 20, 40, 0.3, 29, max_batch=B)
        like = np.stack([frames[j % F][0] for j in range(B)])
        idx = np.stack([frames[j % F][1] for j in range(B)])
        d, it, sy = dec.v2_decode_batch(like, idx)
        del like, idx
        for j in range(B):
            r = ref[j % F]
            assert it[j] == r["iters"] and (d[j] == r["decide"]).all(), (B, j)
        for j in (1, B // 2, B - 1):
            assert dec.v2_state(j)["app"].tobytes() == ref[j % F]["app"].tobytes(), (B, j)
        dec.close()
    its = np.array([r["iters"] for r in ref])
    print("soak v2: %d distinct frames, iterations min/median/max %d/%d/%d" % (F, its.min(), int(np.median(its)), its.max()))
