"""Seeded random small codes with an encoder in the loop (shared by tests/test_oracle_fuzz_vs_reference.py and the harness case of
tests/test_gpu_fuzz_codes.py): regular and mixed check degrees from 2 to 20 with the largest in row 0 (the reference sizes its scratch by
rowDegree[0], init.c:320), GF(16) / GF(64) / GF(256), now and then a check node that lists a variable node twice, N >= 2 M so that the random
matrix has full rank with near certainty.  Written in the reference's block alist layout (init.c:195-207)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
from make_irregular_matrix import write  # noqa: E402

DEG = [[3], [4], [3, 4, 5], [6, 8], [2, 3], [10, 12], [17, 18, 20], [8], [2], [5, 9, 13]]


def simulation_case(seed, path):
    """Writes the matrix of case `seed` to `path`; returns the argv of a simulation on it: (frames, NbIterMax, EbN, nm, offset, NbOper)."""
    rs = np.random.RandomState(70000 + seed)
    q = int(rs.choice([16, 64, 256]))
    degs = DEG[seed % len(DEG)]
    M = int(rs.randint(4, 10))
    deg = sorted((int(rs.choice(degs)) for _ in range(M)), reverse=True)
    N = max(max(deg) + 2, 2 * M, sum(deg) // 2)
    rows = []
    for d in deg:
        r = [int(c) for c in rs.choice(N, d, replace=False)]
        if d >= 3 and rs.rand() < 0.3:
            i, j = (int(x) for x in rs.choice(d, 2, replace=False))
            r[j] = r[i]
        rows.append(r)
    coefs = [[int(x) for x in rs.randint(1, q, d)] for d in deg]
    write(path, N, M, q, deg, rows, coefs)
    nm = min(q, int(rs.choice([6, 8, 12, 16, 24])))
    nop = int(rs.choice([nm + 3, 2 * nm, 3 * nm]))
    return 25, int(rs.randint(3, 9)), float(rs.choice([2.0, 4.0, 6.0])), nm, float(rs.choice([0.0, 0.3, 1.0])), nop
