"""oracle/ems_v2_oracle.c against outputs of the REAL v2 reference stored in tests/golden/v2_frames.npz
(tests/golden/make_golden_v2.py).  Runs anywhere (no reference tree, no GPU)."""
import importlib
import os
import sys

import numpy as np

from conftest import MAT, ROOT, gold

STATE = ("vtoc", "vtoci", "ctov", "ctovi", "app")


def _v2():
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    return importlib.import_module("orc_v2")


def test_v2_big_code_frames(orc):
    v2 = _v2()
    g = gold("v2_frames.npz")
    o = v2.V2Oracle(os.path.join(MAT, "Ahmed_64800_R12_GF256"))
    for i, (sg, seed, itmax) in enumerate(g["big_cases"]):
        like, idx = v2.synthetic_likelihood(o.N, 256, 8, o.o.idx2bin, float(sg), int(seed))
        a = o.decode(like, idx, 20, 40, 0.3, int(itmax))
        meta = g["big%d_meta" % i]
        assert a["iters"] == meta[0] and a["synd"] == meta[1]
        assert (a["decide"].astype(np.uint8) == g["big%d_decide" % i]).all()
        assert [orc.crc32(a[k]) for k in STATE] == [int(x) for x in meta[2:]]


def test_v2_small_code_frames_full_state(orc):
    v2 = _v2()
    g = gold("v2_frames.npz")
    o = v2.V2Oracle(os.path.join(MAT, "N96_K48_GF64"))
    for i, (sg, seed) in enumerate(g["small_cases"]):
        like, idx = v2.synthetic_likelihood(o.N, 64, 6, o.o.idx2bin, float(sg), int(seed))
        a = o.decode(like, idx, 12, 24, 0.5, 12)
        assert a["iters"] == g["small%d_meta" % i][0] and a["synd"] == g["small%d_meta" % i][1]
        assert (a["decide"].astype(np.uint8) == g["small%d_decide" % i]).all()
        for k in STATE:
            assert a[k].tobytes() == g["small%d_%s" % (i, k)].tobytes(), (i, k)
