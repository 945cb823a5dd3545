"""Generates tests/golden/matrices/Irr345_N60_M24_GF64 (+ the GF(256) twin Irr345_N60_M24_GF256): small codes with IRREGULAR check
degrees 3 / 4 / 5 in the reference's block ("UBS") alist layout (init.c:195-207), for SURVEY.md 8(f) row f4.  Row 0 carries the largest
degree, so the reference itself -- which sizes its per-check-node scratch by rowDegree[0] and only warns about a varying degree
(init.c:320, NB_LDPC.c:166-183) -- decodes them without leaving its buffers: tests/golden/kat_traces.json gets their traces from the
real executable (make_golden.py), and the GPU kernels are held to those.  Seeded; full rank is checked with the oracle's encoder.

    python tests/golden/make_irregular_matrix.py          (run from the repo root)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import orc  # noqa: E402


def build(N, M, q, seed):
    rs = np.random.RandomState(seed)
    while True:
        deg = [5] + [int(rs.choice([3, 4, 5])) for _ in range(M - 1)]
        E = sum(deg)
        # columns: every variable node in at least one check, at most three; a check never lists a variable node twice
        cols = list(range(N)) + [int(c) for c in rs.randint(0, N, E - N)] if E >= N else None
        if cols is None:
            continue
        rs.shuffle(cols)
        rows, ok, k = [], True, 0
        for d in deg:
            r = cols[k:k + d]
            k += d
            if len(set(r)) != d:
                ok = False
                break
            rows.append(sorted(r))
        if not ok or max(np.bincount(cols, minlength=N)) > 3:
            continue
        coefs = [[int(x) for x in rs.randint(1, q, d)] for d in deg]
        return deg, rows, coefs


def write(path, N, M, q, deg, rows, coefs):
    cdeg = np.bincount([c for r in rows for c in r], minlength=N)
    with open(path, "w") as f:
        f.write("%d %d %d \n" % (N, M, q))
        f.write(" ".join(str(int(x)) for x in cdeg) + " \n")
        f.write(" ".join(str(d) for d in deg) + " \n")
        for r in rows:
            f.write(" ".join(str(c) for c in r) + " \n")
        for cf in coefs:
            f.write(" ".join(str(c) for c in cf) + " \n")


if __name__ == "__main__":
    orc.build()
    for q in (64, 256):
        path = os.path.join(ROOT, "tests", "golden", "matrices", "Irr345_N60_M24_GF%d" % q)
        for seed in range(1000):
            deg, rows, coefs = build(60, 24, q, 17 * q + seed)
            write(path, 60, 24, q, deg, rows, coefs)
            try:
                o = orc.Oracle(path)
                o.enc                               # raises when H is rank deficient
            except Exception:
                continue
            print(path, "degrees", sorted(set(deg)), "E", sum(deg), "seed", 17 * q + seed)
            break
