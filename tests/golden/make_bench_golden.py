"""Generates tests/golden/bench_golden.npz + bench_counters.json: what the CPU oracle (oracle/ems_oracle.c, pinned against the
real reference by tests/test_oracle_vs_reference.py) gives on the EXACT workloads bench.py times --

    gf256  Ahmed_64800_R12_GF256  nm 20 NbOper 40 offset 0.3 NbIterMax 30, 1.0 dB, fixed work, frames 0..1535 of the drand48 stream
    gf16   Ahmed_64800_R34_GF16   nm 8  NbOper 16 offset 0.3 NbIterMax 30, 0.0 dB, fixed work, frames 0..255
    n96    N96_K48_GF64           nm 16 NbOper 32 offset 0.3 NbIterMax 30, -3.0 dB, fixed work, frames 0..8191

(BASELINE.json configs[3] / [2] / [1] at the batch the bench runs them).  Per frame: bit errors over the first K symbols
(NB_LDPC.c:479-485), the value Syndrom returns, CRC-32 of the hard decisions (as bytes) and CRC-32 of the whole APP block
after the last sweep; per workload the five counters of the harness (frames, erroneous frames, bit errors, undetected,
sum of iterations; NB_LDPC.c:474-497) that bench.py compares its timed run with (`counters_match_golden`).

    python tests/golden/make_bench_golden.py [gf256] [gf16] [n96]     (run from the repo root; gf256 takes ~15 min on 8 cores)
    python tests/golden/make_bench_golden.py --ranks 1 2 ... 7          (the headline counters of ranks 1.. of a multi-GPU bench)
    python tests/golden/make_bench_golden.py gf256_v2                   (the v2 flavour's step: its 8 distinct frames, rank 0)

gf256_v2 (bench.py's `other_workloads.gf256_v2`): the v2 oracle (oracle/ems_v2_oracle.c, pinned against the real v2 translation
unit by tests/test_v2_oracle_vs_reference.py) on the 8 seeded synthetic likelihood frames the bench tiles over its batch
(seeds 0..7 = rank 0), NbIterMax 28 = 29 iterations, fixed work: per frame the syndrome value, the iteration count and the CRC-32
of the hard decisions (as u8).
"""
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import orc  # noqa: E402

WORKLOADS = {   # name: (matrix, nm, NbOper, offset, NbIterMax, Eb/N0, frames) -- bench.py's WORKLOADS
    "gf256": ("Ahmed_64800_R12_GF256", 20, 40, 0.3, 30, 1.0, 1536),
    "gf16": ("Ahmed_64800_R34_GF16", 8, 16, 0.3, 30, 0.0, 256),
    "n96": ("N96_K48_GF64", 16, 32, 0.3, 30, -3.0, 8192),
    "irr345": ("Irr345_N8100_M4050_GF256", 20, 40, 0.3, 30, 1.0, 256),        # side lines: irregular check degrees, check degree 20
    "dc20": ("Ahmed_64800_R910_GF256", 20, 40, 0.3, 30, 3.0, 256),
}


LCG_A, LCG_C, LCG_MASK = 0x5DEECE66D, 0xB, (1 << 48) - 1


def lcg_jump(x0, n):
    """State of glibc's drand48 LCG after n draws from x0 (the affine map raised to the n-th power)."""
    a, c, ba, bc = 1, 0, LCG_A, LCG_C
    while n > 0:
        if n & 1:
            a, c = (ba * a) & LCG_MASK, (ba * c + bc) & LCG_MASK
        ba, bc = (ba * ba) & LCG_MASK, (ba * bc + bc) & LCG_MASK
        n >>= 1
    return (a * x0 + c) & LCG_MASK


def generate(name, threads, rank=0):
    mat, nm, nop, off, itmax, ebn, F = WORKLOADS[name]
    o = orc.Oracle(os.path.join(ROOT, "tests", "golden", "matrices", mat))
    sigma = o.sigma(ebn)
    # rank r of a multi-GPU bench owns frames [r F, (r + 1) F) of the stream: K logq source draws + 2 N logq channel draws per frame
    rng = orc.Rng(lcg_jump(0, rank * F * (o.K * o.logq + 2 * o.N * o.logq)))
    frames = [o.draw_frame(rng, sigma) for _ in range(F)]          # the reference's stream, frame after frame (NB_LDPC.c:255-261)

    def one(i):
        cw, noisy = frames[i]
        r = o.decode(o.llr(noisy, sigma), nm, nop, off, itmax, early_exit=False, want_state=True)
        return (o.bit_errors(r["decide"], cw), r["synd"], r["iters"], orc.crc32(r["decide"].astype(np.uint8)), orc.crc32(r["app"]))

    t0 = time.time()
    with ThreadPoolExecutor(threads) as ex:
        rec = np.array(list(ex.map(one, range(F))), dtype=np.int64)
    nerr, synd, iters = rec[:, 0], rec[:, 1], rec[:, 2]
    counters = [F, int((nerr != 0).sum()), int(nerr.sum()), int(((nerr != 0) & (synd == 0)).sum()), int(iters.sum())]
    print("%s: %d frames in %.0f s; counters %s" % (name, F, time.time() - t0, counters), flush=True)
    return rec, counters


V2 = ("Ahmed_64800_R12_GF256", 20, 40, 0.3, 28, 0.95, 8)          # bench.py's WORKLOADS["gf256_v2"] + its sigma and distinct frames


def generate_v2(threads):
    import orc_v2
    mat, nm, nop, off, itmax, sigma, F = V2
    o = orc_v2.V2Oracle(os.path.join(ROOT, "tests", "golden", "matrices", mat))

    def one(s):
        like, idx = orc_v2.synthetic_likelihood(o.N, o.q, o.logq, o.o.idx2bin, sigma, s)
        r = o.decode(like, idx, nm, nop, off, itmax, early_exit=False)
        return [int(r["synd"]), int(r["iters"]), int(orc.crc32(r["decide"].astype(np.uint8)))]

    t0 = time.time()
    with ThreadPoolExecutor(threads) as ex:
        rec = list(ex.map(one, range(F)))
    print("gf256_v2: %d frames in %.0f s: %s" % (F, time.time() - t0, rec), flush=True)
    return {"matrix": mat, "nm": nm, "NbOper": nop, "offset": off, "NbIterMax": itmax, "sigma": sigma, "distinct_frames": F, "seeds": list(range(F)),
            "synd_iters_crc32decisions": rec}


def add_ranks(ranks, threads):
    """bench_counters.json["gf256"]["ranks"][r]: the counters of rank r's batch of a `bench.py --gpus N` run (r >= 1; counters only)."""
    cj = os.path.join(ROOT, "tests", "golden", "bench_counters.json")
    counters = json.load(open(cj))
    rk = counters["gf256"].setdefault("ranks", {"0": counters["gf256"]["frames_errframes_biterrs_undetected_sumiter"]})
    for r in ranks:
        if str(r) in rk:
            continue
        _rec, cnt = generate("gf256", threads, rank=r)
        rk[str(r)] = cnt
        json.dump(counters, open(cj, "w"), indent=1)


if __name__ == "__main__":
    orc.build()
    if len(sys.argv) > 1 and sys.argv[1] == "--ranks":              # e.g. --ranks 1 2 3 4 5 6 7 (18 minutes of 8 cores each)
        add_ranks([int(x) for x in sys.argv[2:]], int(os.environ.get("GOLDEN_THREADS", str(os.cpu_count() or 8))))
        sys.exit(0)
    which = sys.argv[1:] or ["gf256", "gf16", "n96"]
    threads = int(os.environ.get("GOLDEN_THREADS", str(os.cpu_count() or 8)))
    gdir = os.path.join(ROOT, "tests", "golden")
    npz = os.path.join(gdir, "bench_golden.npz")
    cj = os.path.join(gdir, "bench_counters.json")
    arrays = dict(np.load(npz)) if os.path.exists(npz) else {}
    counters = json.load(open(cj)) if os.path.exists(cj) else {}
    for name in which:
        if name == "gf256_v2":
            counters[name] = generate_v2(threads)
            json.dump(counters, open(cj, "w"), indent=1)
            continue
        rec, cnt = generate(name, threads)
        mat, nm, nop, off, itmax, ebn, F = WORKLOADS[name]
        arrays[name] = rec.astype(np.int64)         # columns: bit errors, syndrome, iterations, CRC-32(decisions as u8), CRC-32(APP)
        counters[name] = {"matrix": mat, "nm": nm, "NbOper": nop, "offset": off, "NbIterMax": itmax, "EbN": ebn, "batch": F,
                          "first_frame": 0, "frames_errframes_biterrs_undetected_sumiter": cnt}
        np.savez_compressed(npz, **arrays)
        json.dump(counters, open(cj, "w"), indent=1)
