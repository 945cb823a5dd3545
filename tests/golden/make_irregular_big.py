"""Generates tests/golden/matrices/Irr345_N8100_M4050_GF256: a 64800-bit GF(256) code with IRREGULAR check degrees 3 / 4 / 5 (1000 / 2050 /
1000 rows, row 0 has degree 5 so that the reference stays inside the scratch it sizes by rowDegree[0]; every variable node in exactly two
checks, like the reference's 64800-bit matrices) in the block ("UBS") alist layout -- the full-size irregular workload the bench line
times beside the BASELINE configurations.  Constructive (no rejection loop), seeded; full rank is checked with the
oracle's encoder.      python tests/golden/make_irregular_big.py          (run from the repo root, ~1 minute)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import orc  # noqa: E402
from make_irregular_matrix import write  # noqa: E402

N, M, Q = 8100, 4050, 256


def build(seed):
    rs = np.random.RandomState(seed)
    deg = np.array([5] + [3] * 1000 + [4] * 2050 + [5] * 999)
    deg[1:] = deg[1:][rs.permutation(M - 1)]
    slots = np.repeat(np.arange(N), 2)[rs.permutation(2 * N)]
    ptr = np.concatenate([[0], np.cumsum(deg)])
    rows = [list(slots[ptr[m]:ptr[m + 1]]) for m in range(M)]
    for m in range(M):                                   # a check never lists a variable node twice: swap the repeat with a later row's entry
        for i in range(len(rows[m])):
            tries = 0
            while rows[m].count(rows[m][i]) > 1:
                m2 = int(rs.randint(0, M)); j = int(rs.randint(0, len(rows[m2])))
                a, b = rows[m][i], rows[m2][j]
                if m2 != m and a not in rows[m2] and b not in rows[m]:
                    rows[m][i], rows[m2][j] = b, a
                tries += 1
                assert tries < 1000
    rows = [sorted(int(c) for c in r) for r in rows]
    coefs = [[int(x) for x in rs.randint(1, Q, len(r))] for r in rows]
    return [int(d) for d in deg], rows, coefs


if __name__ == "__main__":
    orc.build()
    path = os.path.join(ROOT, "tests", "golden", "matrices", "Irr345_N8100_M4050_GF256")
    for seed in range(5):
        deg, rows, coefs = build(4242 + seed)
        write(path, N, M, Q, deg, rows, coefs)
        try:
            o = orc.Oracle(path)
            o.enc
        except Exception as e:
            print("seed", 4242 + seed, "rank deficient", e)
            continue
        print(path, "E", sum(deg), "seed", 4242 + seed)
        break
