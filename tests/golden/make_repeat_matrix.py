"""Generates tests/golden/matrices/Rep_N40_M20_GF64: a small dc = 4 code in which some check nodes list one variable node TWICE
(two edges, two coefficients), in the reference's block ("UBS") alist layout (init.c:195-207).  None of the reference's matrices does
this, but nothing in it forbids it: both edges read the same APP row (NB_LDPC.c:329-337), both carry a message of their own through
the check node, and the APP row is updated once per edge in edge order (NB_LDPC.c:444-450) -- the later edge's update is the one that
stays.  kat_traces.json R1 is the real executable's trace on it (make_golden.py).  Seeded.

    python tests/golden/make_repeat_matrix.py          (run from the repo root)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
from make_irregular_matrix import write  # noqa: E402


def build(N, M, q, dc, repeats, seed):
    """A (2, dc)-regular-ish graph by dealing a shuffled double list of the variable nodes to the checks, then `repeats` checks get one of
    their variable nodes a second time in place of another one (adjacent or not, so both positions of the pair occur)."""
    rs = np.random.RandomState(seed)
    while True:
        cols = [int(c) for c in rs.permutation(np.repeat(np.arange(N), 2))]
        rows = [cols[m * dc:(m + 1) * dc] for m in range(M)]
        if any(len(set(r)) != dc for r in rows):
            continue
        for m in rs.choice(M, repeats, replace=False):
            i, j = (int(x) for x in rs.choice(dc, 2, replace=False))
            rows[m][j] = rows[m][i]
        if len(set(c for r in rows for c in r)) != N:          # (every variable node still in some check)
            continue
        coefs = [[int(x) for x in rs.randint(1, q, dc)] for _ in range(M)]
        return [dc] * M, rows, coefs


if __name__ == "__main__":
    deg, rows, coefs = build(40, 20, 64, 4, 5, 7)
    path = os.path.join(ROOT, "tests", "golden", "matrices", "Rep_N40_M20_GF64")
    write(path, 40, 20, 64, deg, rows, coefs)
    print(path, [r for r in rows if len(set(r)) < 4])
    # check degree 18 (k_decode_wide: two passes of 16 edge rows over GF(64)): repeats inside the first pass, inside the second, and across the two
    deg, rows, coefs = build(54, 6, 64, 18, 0, 11)
    for m, (i, j) in ((1, (3, 4)), (2, (2, 17)), (4, (16, 17)), (5, (0, 16))):
        rows[m][j] = rows[m][i]
    assert len(set(c for r in rows for c in r)) == 54
    path = os.path.join(ROOT, "tests", "golden", "matrices", "RepWide_N54_M6_GF64")
    write(path, 54, 6, 64, deg, rows, coefs)
    print(path, [[c for c in set(r) if r.count(c) > 1] for r in rows])
