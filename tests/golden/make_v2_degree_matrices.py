"""Generates the (2, dc) matrices of the v2 flavour's round-5 degree tests in the reference's block ("UBS") alist layout:

  V2Irr_M24_GF256       irregular check degrees 2..12 (every variable node has exactly two edges, as v2.c:2157 requires), the largest
                        degree in row 0 -- the older generation sizes its scratch by rowDegree[0] (v2.c:750, :794-800) and loops over
                        rowDegree[node] (:1756-1852), so its own executable decodes such a code inside its buffers
  V2Wide22_M10_GF256  regular check degree 22 (above the 16 lane rows of process_cn_v2: k_decode_v2_wide walks it in three passes)
  V2Wide18_M8_GF64    regular check degree 18 over GF(64) (decode core only: the older generation's main() is GF(256))

tests/golden/v2_traces.json V5 / V6 are the REAL executable's statistics on the first two (make_golden_v2_trace.py).  Seeded; full rank
is checked with the oracle's encoder (the executable stops on a rank-deficient matrix, v2.c Gaussian elimination).

    python tests/golden/make_v2_degree_matrices.py          (run from the repo root)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import orc  # noqa: E402
from make_irregular_matrix import write  # noqa: E402


def two_edge_code(rs, deg, q):
    """Rows of the given degrees over N = sum(deg) / 2 variable nodes, every node in exactly two different checks."""
    E = sum(deg)
    assert E % 2 == 0
    N = E // 2
    while True:
        slots = [int(c) for c in rs.permutation(np.repeat(np.arange(N), 2))]
        rows, k, ok = [], 0, True
        for d in deg:
            r = slots[k:k + d]
            k += d
            if len(set(r)) != d:
                ok = False
                break
            rows.append(sorted(r))
        if ok:
            return N, rows, [[int(x) for x in rs.randint(1, q, d)] for d in deg]


def make(name, q, deg_fn):
    path = os.path.join(ROOT, "tests", "golden", "matrices", name)
    for seed in range(1000):
        rs = np.random.RandomState(9100 + seed)
        deg = deg_fn(rs)
        N, rows, coefs = two_edge_code(rs, deg, q)
        write(path, N, len(deg), q, deg, rows, coefs)
        try:
            o = orc.Oracle(path)
            o.enc                                        # Gaussian elimination: raises on a rank-deficient matrix
        except RuntimeError:
            continue
        print(name, "N", N, "M", len(deg), "degrees", sorted(set(deg)), "seed", seed)
        return
    raise SystemExit("no full-rank matrix found for " + name)


if __name__ == "__main__":
    orc.build()

    def irregular(rs):
        while True:
            deg = [12, 2, 3] + [int(rs.choice([2, 3, 4, 5, 6, 8, 10, 12])) for _ in range(21)]
            if sum(deg) % 2 == 0:
                return deg
    make("V2Irr_M24_GF256", 256, irregular)
    make("V2Wide22_M10_GF256", 256, lambda rs: [22] * 10)
    make("V2Wide18_M8_GF64", 64, lambda rs: [18] * 8)
