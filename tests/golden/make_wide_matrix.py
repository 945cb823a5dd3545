"""Generates tests/golden/matrices/Wide_N80_M8_GF64, Wide_N80_M8_GF256 and Wide_N120_M6_GF16: small codes whose check degrees lie
ABOVE 16 (17..24 over GF(64) / GF(256), 17..40 over GF(16)), irregular, in the reference's block ("UBS") alist layout
(init.c:195-207).  The reference caps the check degree nowhere (scratch sized by rowDegree[0], init.c:320; bubble_decoder.c:166-227
loops over any dc) and ships a dc = 20 matrix (Ahmed_64800_R910_GF256); these small ones let the tests compare WHOLE frame state of
k_decode_wide -- one, two and three passes over the edge rows, shadow rows, per-visit degrees -- with the oracle in seconds.  Row 0
carries the largest degree, so the real executable decodes them inside its buffers: kat_traces.json W1..W3 (make_golden.py).
Seeded; full rank is checked with the oracle's encoder.

    python tests/golden/make_wide_matrix.py          (run from the repo root)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import orc  # noqa: E402
from make_irregular_matrix import write  # noqa: E402


def build(N, M, q, dmax, choices, seed):
    rs = np.random.RandomState(seed)
    for _ in range(200):
        deg = [dmax] + [int(rs.choice(choices)) for _ in range(M - 1)]
        E = sum(deg)
        if E < N:
            continue
        # every variable node in at least one check (the first N slots are a permutation), the rest drawn at random; slots are dealt to
        # the checks round-robin, a check never lists a variable node twice
        slots = [int(c) for c in rs.permutation(N)] + [int(c) for c in rs.randint(0, N, E - N)]
        rows = [[] for _ in deg]
        ok = True
        m = 0
        for c in slots:
            for _try in range(M):
                if len(rows[m]) < deg[m] and c not in rows[m]:
                    rows[m].append(c)
                    break
                m = (m + 1) % M
            else:
                ok = False
                break
            m = (m + 1) % M
        if not ok or any(len(r) != d for r, d in zip(rows, deg)):
            continue
        rows = [sorted(r) for r in rows]
        if max(np.bincount([c for r in rows for c in r], minlength=N)) > 4:
            continue
        coefs = [[int(x) for x in rs.randint(1, q, d)] for d in deg]
        return deg, rows, coefs
    return None


if __name__ == "__main__":
    orc.build()
    for name, N, M, q, dmax, choices in (("Wide_N80_M8_GF64", 80, 8, 64, 24, [17, 18, 19, 20, 22, 24]),
                                         ("Wide_N80_M8_GF256", 80, 8, 256, 24, [17, 18, 19, 20, 22, 24]),
                                         ("Wide_N120_M6_GF16", 120, 6, 16, 40, [17, 25, 32, 33, 36, 40])):
        path = os.path.join(ROOT, "tests", "golden", "matrices", name)
        for seed in range(1000):
            b = build(N, M, q, dmax, choices, 31 * q + seed)
            if b is None:
                continue
            deg, rows, coefs = b
            write(path, N, M, q, deg, rows, coefs)
            try:
                o = orc.Oracle(path)
                o.enc                               # raises when H is rank deficient
            except Exception:
                continue
            print(path, "degrees", deg, "E", sum(deg), "seed", 31 * q + seed)
            break
