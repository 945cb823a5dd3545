#!/usr/bin/env python3
"""bench.py -- decoded frames/s of the MI355X EMS NB-LDPC hot path on the headline configuration.

  python bench.py --gpus N --steps K --warmup W

N > 1 works two ways: under `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` (RANK / WORLD_SIZE in
the environment: this process IS a rank), or from a plain shell -- the process then only starts N fresh rank processes of
itself (launch_ranks below; it never imports torch or touches HIP), relays rank 0's JSON line, and exits non-zero as soon
as any rank fails or the run exceeds EMS_BENCH_LAUNCH_TIMEOUT seconds (default 1500), after killing every rank it started.

A "step" is one pass of the hot path over one batch of synthetic frames per GPU: GF(q) channel-LLR
init -> 29 layered EMS sweeps with per-sweep hard decision + syndrome -> bit-error count, inputs
(noisy BPSK samples + transmitted codewords) already resident in HBM.  Workload at every N is
BASELINE.json configs[3]: Ahmed_64800_R12_GF256, nm=20, NbIterMax=30 (29 sweeps, NB_LDPC.c:314),
offset 0.3, NbOper 40, batch 1536 per GPU (6 frames per CU, the shape of the two-visits-per-wave kernel, see DESIGN.md
sections 4.1 and 5), early exit disabled (fixed work),
Eb/N0 = 1.0 dB frames drawn
from the reference harness' own drand48 stream (each rank jumps the LCG to its frame range).  Frames are
independent, so ranks share nothing on the data path ("scaling": "weak"); per step the ranks all-reduce
the five error counters over RCCL, as the Monte-Carlo harness does.

Prints ONE JSON line (rank 0) with `roofline` (dominant kernel = the persistent decode kernel, timed
with HIP events on the stream it runs on; its name and workgroup shape are asked from the library) and
`cpu_baseline` (the oracle, i.e. the CPU port of the reference path, and the reference's own executable,
on a bounded sample of the same workload; N=1 only).  The line checks itself: `counters_match_golden`
says whether the error counters of the timed run equal the CPU oracle's on the same frames
(tests/golden/bench_counters.json).  At N=1 `config.other_workloads` carries short runs of BASELINE
configs[1] (N96, batch 8192), configs[2] (GF(16), batch 256) and the v2 flavour of the headline code (batch 1536), each with its
own `roofline` and its own check against an oracle golden.
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (matrix, nm, NbOper, offset, NbIterMax, EbN0 for fixed work, default batch)
    "gf256": ("Ahmed_64800_R12_GF256", 20, 40, 0.3, 30, 1.0, 1536),
    "gf16": ("Ahmed_64800_R34_GF16", 8, 16, 0.3, 30, 0.0, 256),
    "n96": ("N96_K48_GF64", 16, 32, 0.3, 30, -3.0, 8192),
    # full-size codes outside the BASELINE list, on the generic kernels (side lines): irregular check degrees 3 / 4 / 5 (SURVEY.md 8(f) f4;
    # tests/golden/make_irregular_big.py) and the reference's dc = 20 matrix (k_decode_wide)
    "irr345": ("Irr345_N8100_M4050_GF256", 20, 40, 0.3, 30, 1.0, 256),
    "dc20": ("Ahmed_64800_R910_GF256", 20, 40, 0.3, 30, 3.0, 256),
    # v2 flavour: same code and message size, NbIterMax 28 = 29 iterations (iter = 0..NbIterMax,
    # v2.c:2119); inputs are seeded synthetic max-is-best likelihoods (synthetic_likelihood below, sigma 0.95)
    "gf256_v2": ("Ahmed_64800_R12_GF256", 20, 40, 0.3, 28, None, 1536),
}
def kernel_sources():
    """Every file the decode kernels are compiled from: ems_kernels.hip, ems_plan.h and EVERYTHING under csrc/kernels/ (the .inc
    files and the lane-code header of k_decode_q16t), so that a new kernel file can never be left out of the hash."""
    csrc = os.path.join(ROOT, "ems-decoder-of-nb-ldpc-codes_amd", "csrc")
    return ["ems_kernels.hip", "ems_plan.h"] + sorted("kernels/" + f for f in os.listdir(os.path.join(csrc, "kernels"))
                                                      if f.endswith((".inc", ".h")))


def kernel_source_hash():
    """sha256 (first 16 hex digits) over the sources the decode kernels are compiled from (names included).  scripts/pmc_summary.py
    stores it in profiles/traffic_*.json next to the PMC measurement; `roofline.traffic` and `roofline.valu` are only printed while the
    two agree -- a committed measurement of a kernel that has since changed is refused instead of going stale silently."""
    import hashlib
    h = hashlib.sha256()
    for f in kernel_sources():
        h.update(f.encode())
        with open(os.path.join(ROOT, "ems-decoder-of-nb-ldpc-codes_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def committed_traffic(name, kernel, B):
    """(bytes per launch of B frames, provenance) from profiles/traffic_<name>.json, or (None, why not)."""
    tr = os.path.join(ROOT, "profiles", "traffic_%s.json" % name)
    if not os.path.exists(tr):
        return None, None
    try:
        tj = json.load(open(tr))
        if tj["kernel"].split("::")[-1].split("(")[0] != kernel:
            return None, "profiles/traffic_%s.json measures another kernel (%s): not printed" % (name, tj["kernel"].split("::")[-1].split("(")[0])
        if tj.get("kernel_source_sha16") != kernel_source_hash():
            return None, "profiles/traffic_%s.json was measured on other kernel sources (sha16 %s, now %s): refused, re-run scripts/pmc_passes.sh" % (
                name, tj.get("kernel_source_sha16"), kernel_source_hash())
        return int(tj["hbm_bytes_per_frame"] * B), ("profiles/traffic_%s.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this kernel (same kernel sources, sha16 %s) at %d "
                                                    "frames per launch, scaled per frame (committed measurement, not this run)" % (name, tj["kernel_source_sha16"], tj["frames_per_launch"]))
    except Exception as e:                                      # a broken file is not worth a failed bench
        return None, "profiles/traffic_%s.json unreadable: %s" % (name, e)


HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


def algorithmic_bytes_per_frame_decode(E, q, nm, sweeps):
    """per sweep every edge reads+writes one dense fp32 APP row and one compact C->V
    message (nm (f32,u8) pairs + a default)."""
    return sweeps * (2 * E * q * 4 + 2 * E * min(q * 4, nm * 5 + 4))


def algorithmic_bytes_per_frame_decode_v2(N, E, q, nm, iters):
    """v2 flavour, per iteration: every edge reads its V->C message and its node's likelihood row (all q values feed
    the top-nm selection), writes its C->V message and the other edge's V->C message (nm (f32,u8) pairs each); the
    hard decision of a node needs, beyond what its last visit already holds, the node's other C->V message, and
    writes one byte."""
    return iters * (E * (q * 4 + 3 * nm * 5) + N * (nm * 5 + 1))


def cpu_baseline_v2(v2, matrix, nm, nop, off, itmax, sigma, threads):
    from concurrent.futures import ThreadPoolExecutor
    o = v2.V2Oracle(matrix)
    frames = [v2.synthetic_likelihood(o.N, o.q, o.logq, o.o.idx2bin, sigma, 100 + i) for i in range(threads)]
    t0 = time.time()
    with ThreadPoolExecutor(threads) as ex:
        list(ex.map(lambda fr: o.decode(fr[0], fr[1], nm, nop, off, itmax, early_exit=False)["iters"], frames))
    dt = time.time() - t0
    return {"value": round(threads / dt, 4), "unit": "frames/s", "cores": threads, "kind": "port",
            "sample": "%d frames (1 per thread), v2 oracle, fixed work %d iterations, %.1f s wall" % (threads, itmax + 1, dt)}


def synthetic_likelihood(N, q, logq, idx2bin, sigma, seed):
    """Bench input generator for the v2 flavour (the product path never touches oracle/): max-is-best log-likelihoods
    of an all-zero BPSK codeword over AWGN, per node the q values in decreasing order with their symbols -- the layout
    of decoder.likelihood / likelihood_index (v2.c:1483-1497)."""
    rs = np.random.RandomState(seed)
    y = (1.0 + sigma * rs.randn(N, logq)).astype(np.float32)
    bits = ((idx2bin.astype(np.int64)[:, None] >> np.arange(logq)[None, :]) & 1).astype(np.float32)
    d = (y[:, None, :] - (1.0 - 2.0 * bits)[None, :, :]) ** 2
    llr = (-d.sum(axis=2) / np.float32(2.0 * sigma * sigma)).astype(np.float32)
    order = np.argsort(-llr, axis=1, kind="stable")
    return np.ascontiguousarray(np.take_along_axis(llr, order, axis=1), np.float32), np.ascontiguousarray(order.astype(np.int32))


def run_v2_workload(a, pkg, ge, torch, dist, rank, world, local, B, steps, warmup):
    """Timed fixed-work steps of the v2 flavour on this rank's GPU (init-from-staged-likelihoods + k_decode_v2 + results); returns what
    rank 0 prints.  Self-check: syndrome, iteration count and CRC-32 of the decisions of the 8 distinct frames against the v2 oracle's
    (tests/golden/bench_counters.json["gf256_v2"], tests/golden/make_bench_golden.py) -- oracle/ itself is not touched here."""
    import zlib
    mat, nm, nop, off, itmax, _e, _B0 = WORKLOADS["gf256_v2"]
    matrix = os.path.join(ge.MATRICES, mat)
    code = pkg.Code(matrix)
    idx2bin, _b2i = pkg.gf_tables(code.GF)
    sigma = 0.95
    ND = 8
    t_setup = time.time()
    base = [synthetic_likelihood(code.N, code.GF, code.logGF, idx2bin, sigma, 1000 * rank + s) for s in range(ND)]
    like = np.stack([base[i % ND][0] for i in range(B)])
    idx = np.stack([base[i % ND][1] for i in range(B)])
    t_setup = time.time() - t_setup
    dec = pkg.Decoder(code, nm, nop, off, itmax, max_batch=B, device=local, early_exit=False)
    dec.v2_load_likelihood(like, idx)                  # staged in HBM before the timed region
    dec.synchronize()
    counters = torch.zeros(3, dtype=torch.int64, device="cuda")
    last = [None, None]

    def step():
        dec.v2_reinit(B)
        dec.v2_decode(B)
        it = np.zeros(B, np.int32); sy = np.zeros(B, np.int32)
        pkg._check(pkg.lib().ems_get_results(dec.h, B, None, pkg._ptr(it), pkg._ptr(sy)))
        last[0], last[1] = it, sy
        counters.copy_(torch.tensor([B, int((sy != 0).sum()), int(it.sum())], dtype=torch.int64))
        if dist is not None:
            dist.all_reduce(counters)
        return dec.last_decode_ms, dec.last_llr_ms

    def fence():
        dec.synchronize(); torch.cuda.synchronize()
        if dist is not None:
            dist.barrier(); torch.cuda.synchronize()

    for _ in range(warmup):
        step()
    fence()
    t0 = time.perf_counter()
    kms, lms = [], []
    for _ in range(steps):
        k, l = step(); kms.append(k); lms.append(l)
    fence()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    # ---- self-check against the v2 oracle's golden (rank 0's seeds; every copy of a distinct frame must agree) ----
    gold = load_bench_golden("gf256_v2")
    match = None
    if gold is not None and rank == 0 and B >= ND and [gold["nm"], gold["NbOper"], gold["NbIterMax"], gold["distinct_frames"]] == [nm, nop, itmax, ND]:
        dcd = np.zeros((ND, code.N), np.int32)
        pkg._check(pkg.lib().ems_get_results(dec.h, ND, pkg._ptr(dcd), None, None))
        it, sy = last
        match = True
        for s_, (gs, gi, gc) in enumerate(gold["synd_iters_crc32decisions"]):
            match = match and bool((sy[s_::ND] == gs).all()) and bool((it[s_::ND] == gi).all()) and zlib.crc32(dcd[s_].astype(np.uint8).tobytes()) == gc
    alg = algorithmic_bytes_per_frame_decode_v2(code.N, code.E, code.GF, nm, itmax + 1) * B
    kavg = float(np.mean(kms)) * 1e-3
    res = {"value": round(world * B * steps / dt, 2), "ms_per_step": round(dt / steps * 1e3, 3),
           "data": "synthetic: seeded max-is-best likelihoods of an all-zero BPSK/AWGN codeword, sigma %.2f, %d distinct frames tiled" % (sigma, ND),
           "workload": "%s v2 flavour nm=%d NbIterMax=%d (%d iterations) offset=%.1f NbOper=%d, fixed work (early exit off)" % (mat, nm, itmax, itmax + 1, off, nop),
           "batch_per_gpu": B, "counters": counters.cpu().tolist(), "counters_match_golden": match,
           "roofline": {"bound": "hbm", "kernel": "%s (persistent, 1 workgroup per frame, %d waves per workgroup, %d frames per launch on %d CUs)" % (dec.last_kernel, dec.last_workgroup_waves, B, dec.n_cu),
                        "achieved": round(alg / kavg / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(alg / kavg / 1e9 / HBM_PEAK_GBS, 5),
                        "traffic": None, "traffic_source": None, "algorithmic_bytes_per_launch": alg, "avg_launch_ms": round(kavg * 1e3, 3), "init_avg_ms": round(float(np.mean(lms)), 3)},
           "host_setup_s": round(t_setup, 1), "matrix": matrix, "params": (nm, nop, off, itmax, sigma)}
    res["roofline"]["traffic"], res["roofline"]["traffic_source"] = committed_traffic("gf256_v2", dec.last_kernel, B)
    add_valu_bound(res["roofline"], "gf256_v2", dec.last_kernel, code.M * (itmax + 1) * B, kavg, dec.n_cu)
    dec.close()
    return res


def bench_v2(a, pkg, ge, torch, dist, rank, world, local):
    """`--workload gf256_v2`: the v2 flavour's own line, same contract.  (oracle/ is imported only by the cpu_baseline leg.)"""
    B = a.batch or WORKLOADS["gf256_v2"][6]
    r = run_v2_workload(a, pkg, ge, torch, dist, rank, world, local, B, a.steps, a.warmup)
    if rank == 0:
        nm, nop, off, itmax, sigma = r["params"]
        out = {"metric": "decoded frames/sec, v2 flavour (truncated messages both ways) @ N=64800 GF(256) R=1/2, nm=20, 29 iterations",
               "value": r["value"], "unit": "frames/s", "n_gpus": world, "n_ranks_seen": dist.get_world_size() if dist is not None else 1,
               "steps": a.steps, "warmup": a.warmup,
               "ms_per_step": r["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "f32", "data": r["data"],
               "config": {"workload": r["workload"], "batch_per_gpu": B, "sharding": "independent frames",
                          "collective": "all-reduce of 3 int64 counters per step" if world > 1 else "none",
                          "frames_nonzero_syndromes_sumiter": r["counters"]},
               "counters_match_golden": r["counters_match_golden"],
               "roofline": r["roofline"], "host_setup_s": r["host_setup_s"]}
        if world == 1 and not a.no_cpu_baseline:
            import importlib as il
            ge.load_oracle()
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            out["cpu_baseline"] = cpu_baseline_v2(il.import_module("orc_v2"), r["matrix"], nm, nop, off, itmax, sigma,
                                                   max(1, min(os.cpu_count() or 1, 16)))
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier(); dist.destroy_process_group()


def cpu_baseline(orc, matrix, nm, nop, off, itmax, ebn, threads):
    """The oracle (CPU port of the reference path) on `threads` frames, one per thread, fixed work."""
    from concurrent.futures import ThreadPoolExecutor
    o = orc.Oracle(matrix)
    sigma = o.sigma(ebn)
    rng = orc.Rng(0)
    frames = [o.draw_frame(rng, sigma)[1] for _ in range(threads)]

    def one(noisy):
        return o.decode(o.llr(noisy, sigma), nm, nop, off, itmax, early_exit=False)["iters"]

    t0 = time.time()
    with ThreadPoolExecutor(threads) as ex:
        list(ex.map(one, frames))
    dt = time.time() - t0
    out = {"value": round(threads / dt, 4), "unit": "frames/s", "cores": threads, "kind": "port",
           "sample": "%d frames (1 per thread), same workload, fixed work, LLR init + 29 sweeps, %.1f s wall" % (threads, dt)}
    ref = reference_executable_rate(orc, matrix, nm, nop, off, itmax, ebn, threads)
    if ref:
        out["reference_executable"] = ref
    return out


def reference_executable_rate(orc, matrix, nm, nop, off, itmax, ebn, procs, frames_each=2):
    """The REAL reference (oracle/_ref/essai_ubs, compiled from /root/reference by oracle/Makefile; travels to the GPU box as a
    built binary) timed beside the port: `procs` copies of the executable, one per core, `frames_each` frames each of the
    same workload (its whole per-frame path: source, encoder, channel, decode).  The one-off set-up of a process (loader,
    tables, Gaussian elimination) is timed by a 0-frame run of the same `procs` copies and subtracted, so `value` is the
    steady-state rate; both walls are in `sample`.  The printed avr_it must say that every frame ran all sweeps.  None when
    the binary is not there or anything goes wrong."""
    import re
    import subprocess
    exe = orc.ref_binary("ubs")
    if not os.path.exists(exe):
        return None
    run = os.path.join(os.path.dirname(exe), "run")
    os.makedirs(os.path.join(run, "data"), exist_ok=True)

    import tempfile

    def wall(frames):
        """Wall time of `procs` copies run side by side.  Every copy writes its progress lines (one per frame, \\r-terminated) to a file of
        its own: drained one after the other through pipes, a copy could fill its 64 KiB pipe and stall inside the wall being measured."""
        cmd = [exe, str(frames), str(itmax), matrix, str(ebn), str(nm), str(off), str(nop)]
        with tempfile.TemporaryDirectory() as td:
            files = [open(os.path.join(td, "out%d.txt" % i), "wb") for i in range(procs)]
            t0 = time.time()
            ps = [subprocess.Popen(cmd, cwd=run, stdin=subprocess.PIPE, stdout=f, stderr=subprocess.DEVNULL) for f in files]
            ok = True
            for p in ps:
                try:
                    p.communicate(b"\n", timeout=300)
                except subprocess.TimeoutExpired:
                    ok = False
                    break
            dt = time.time() - t0
            if not ok:
                for p in ps:                            # no orphans: every copy is killed and reaped
                    p.kill()
                for p in ps:
                    try:
                        p.communicate(timeout=10)
                    except Exception:
                        pass
            for f in files:
                f.close()
            if not ok or any(p.returncode != 0 for p in ps):
                return None, None
            return dt, [open(f.name, "rb").read().decode("latin-1") for f in files]

    t_setup, _ = wall(0)
    t_setup2, _ = wall(0)                               # the set-up run twice: the smaller wall is subtracted (page cache, clocks)
    if t_setup is not None and t_setup2 is not None:
        t_setup = min(t_setup, t_setup2)
    t_all, outs = wall(frames_each)
    if t_setup is None or t_all is None or t_all <= t_setup:
        return None
    its = [re.findall(r"avr_it=([\d.]+)", o) for o in outs]
    full = all(it and abs(float(it[-1]) - itmax) < 1e-6 for it in its)   # a non-converged frame counts NbIterMax (NB_LDPC.c:474)
    return {"value": round(procs * frames_each / (t_all - t_setup), 4), "unit": "frames/s", "cores": procs, "kind": "reference",
            "value_with_setup": round(procs * frames_each / t_all, 4),      # nothing subtracted: a lower bound that cannot be overstated
            "all_sweeps_ran": bool(full),
            "sample": "%d processes x %d frames of the reference executable at %.1f dB (early exit on; avr_it printed = %s): %.1f s wall, of which "
                      "%.1f s set-up (0-frame run) subtracted" % (procs, frames_each, ebn, its[0][-1] if its and its[0] else "?", t_all, t_setup)}


class c_stdout_to_stderr:
    """The reference-shaped host functions printf() their banner (LoadCode, init.c:262-264); keep fd 1
    clean so that stdout carries exactly one JSON line."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        import ctypes
        ctypes.CDLL(None).fflush(None)
        os.dup2(self.saved, 1)
        os.close(self.saved)


def load_bench_golden(workload):
    """tests/golden/bench_counters.json (tests/golden/make_bench_golden.py): the five harness counters the CPU oracle gives on
    this workload's first batch of the drand48 stream."""
    try:
        return json.load(open(os.path.join(ROOT, "tests", "golden", "bench_counters.json"))).get(workload)
    except Exception:
        return None


def run_decode_workload(a, pkg, harness, ge, torch, dist, rank, world, local, workload, B, steps, warmup, side_figures):
    """Timed fixed-work steps of one workload of the main decode path on this rank's GPU: returns what rank 0 prints."""
    mat, nm, nop, off, itmax, ebn, _B0 = WORKLOADS[workload]
    ebn_overridden = bool(os.environ.get("EMS_BENCH_EBN"))
    if ebn_overridden:                                        # developer knob: the fixed-work step at another operating point
        ebn = float(os.environ["EMS_BENCH_EBN"])
    matrix = os.path.join(ge.MATRICES, mat)
    code = pkg.Code(matrix)
    dec = pkg.Decoder(code, nm, nop, off, itmax, max_batch=B, device=local, early_exit=False)
    # developer knob: a single process decodes the batch rank r of a multi-GPU run would own (checks the ranks' goldens on one GPU)
    frames_of = int(os.environ.get("EMS_BENCH_FRAMES_OF_RANK", rank))

    # ---- synthetic inputs, staged in HBM before the timed region ----
    t_setup = time.time()
    if a.synthetic_zero_codeword:
        rs = np.random.RandomState(1234 + rank)
        sigma = float(np.float32(np.sqrt(1.0 / (2.0 * float(code.rate) * 10 ** (ebn / 10.0)))))
        noisy = (1.0 + sigma * rs.randn(B, code.N, code.logGF)).astype(np.float32)
        cws = np.zeros((B, code.N), np.int32)
        data = "synthetic: all-zero codeword + AWGN (numpy), Eb/N0 %.1f dB" % ebn
    else:
        with c_stdout_to_stderr():
            src = harness.FrameSource(matrix, nm)
            src.position(frames_of * B)             # rank r owns frames [r*B, (r+1)*B) of the stream
            cws, noisy, sigma = src.draw(B, ebn)
            src.close()
        data = "synthetic: random codewords + BPSK/AWGN from the reference harness' drand48 stream, Eb/N0 %.1f dB" % ebn
    t_setup = time.time() - t_setup
    pkg._check(pkg.lib().ems_llr_init(dec.h, pkg._ptr(noisy), np.float32(sigma), B))  # uploads into the staging buffer
    dec.set_codewords(cws)
    dec.synchronize()
    noisy_dev = dec.noisy_dev_ptr
    counters = torch.zeros(5, dtype=torch.int64, device="cuda")
    local_counts = [None]

    def step():
        dec.llr_init_dev(noisy_dev, sigma, B)
        dec.decode(B)
        nberr = dec.count_errors_resident(B)                      # syncs the decoder's stream
        it = np.zeros(B, np.int32); sy = np.zeros(B, np.int32)
        pkg._check(pkg.lib().ems_get_results(dec.h, B, None, pkg._ptr(it), pkg._ptr(sy)))
        local_counts[0] = [B, int((nberr != 0).sum()), int(nberr.sum()), int(((nberr != 0) & (sy == 0)).sum()), int(it.sum())]
        counters.copy_(torch.tensor(local_counts[0], dtype=torch.int64))
        if dist is not None:
            dist.all_reduce(counters)                             # RCCL, 40 bytes
        return dec.last_decode_ms, dec.last_llr_ms

    def fence():
        dec.synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(warmup):
        step()
    fence()
    t0 = time.perf_counter()
    kms, lms = [], []
    for _ in range(steps):
        k, l = step()
        kms.append(k); lms.append(l)
    fence()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    tot = counters.cpu().tolist()
    kernel = dec.last_kernel                                       # what the library launched, from the library
    wg_waves = dec.last_workgroup_waves
    n_cu = dec.n_cu

    # ---- self-check: this rank's counters of the timed run against the oracle's (committed golden; rank 0's frame range) ----
    gold = load_bench_golden(workload)
    comparable = (gold is not None and not a.synthetic_zero_codeword and not ebn_overridden and gold["batch"] == B
                  and [gold["nm"], gold["NbOper"], gold["NbIterMax"]] == [nm, nop, itmax])
    match, ranks_checked = None, 0
    if comparable:
        # every rank checks ITS batch (frames [rank B, (rank + 1) B) of the stream) where the golden has it: rank 0 always, ranks 1.. of the
        # headline workload when tests/golden/make_bench_golden.py --ranks has been run; the verdicts are combined over the ranks
        per_rank = gold.get("ranks", {"0": gold["frames_errframes_biterrs_undetected_sumiter"]})
        mine = per_rank.get(str(frames_of))
        flags = torch.tensor([0 if mine is None else 1, 0 if (mine is None or local_counts[0] == mine) else 1], dtype=torch.int64, device="cuda")
        if dist is not None:
            dist.all_reduce(flags)                                  # [ranks that had a golden, ranks whose counters differ]
        ranks_checked, bad = int(flags[0].item()), int(flags[1].item())
        if ranks_checked > 0:
            match = bad == 0
        if "ranks" in gold:                                         # (the headline workload: the golden covers ranks 0..7)
            check_ranks_against_golden(world, ranks_checked, bad)

    # ---- side figure (never `value`): BASELINE.json configs[3] to the letter, the same code at batch = 256 (one frame per CU) ----
    b256 = None
    if side_figures and rank == 0 and world == 1 and workload == "gf256" and B > 256:
        def step256():
            dec.llr_init_dev(noisy_dev, sigma, 256)
            dec.decode(256)
            dec.count_errors_resident(256)
            return dec.last_decode_ms
        step256()
        t1 = time.perf_counter()
        k256 = [step256() for _ in range(3)]
        b256 = {"frames_per_s": round(3 * 256 / (time.perf_counter() - t1), 1), "avg_launch_ms": round(float(np.mean(k256)), 3),
                "kernel": dec.last_kernel, "workgroup_waves": dec.last_workgroup_waves,
                "note": "configs[3] as written (batch 256 = one frame per CU): bound by the frame's dependency chain, see DESIGN.md 4.1"}
    dec.close()     # frees its stream: ROCm multiplexes streams onto GPU_MAX_HW_QUEUES (4) hardware queues, the side run below wants 3 of its own

    # ---- side figure (never `value`): early exit enabled at the operating point of SURVEY.md 8(d) ----
    oper = None
    OPER_EBN = {"gf256": 2.0, "gf16": 3.0, "n96": 3.0}.get(workload, 3.0)
    if side_figures and rank == 0 and world == 1 and not a.synthetic_zero_codeword:
        # Every frame distinct: the reference's own drand48 stream from frame 2^20 on, generated in HBM (ems_source_draw_reference, DESIGN.md
        # 4.5b).  Rounds 2-4 tiled 256 host-drawn frames: workgroups go to the XCDs round-robin, so the period-256 pattern gave every XCD
        # the SAME frames in every tile -- the XCD whose 32 frames need the most sweeps set the time (0.74 of the fixed-work sweep rate at
        # 12288 frames; 0.89 with the same frames shuffled, scripts/gpu_oper.py OPER_ANALYSE=1).
        Bo = 4 * B                                  # several waves of workgroups: finished frames free their CU slots for queued ones
        OH = 3                                      # handles (own streams) launched round-robin as harness.monte_carlo / ems_sim
        ods = []                                    # do: the tail of one launch overlaps the head of the next ones
        with c_stdout_to_stderr():
            src = harness.FrameSource(matrix, nm)
        osigma = float(np.float32(np.sqrt(1.0 / (2.0 * float(src.s.code.rate) * 10.0 ** (float(np.float32(OPER_EBN)) / 10.0)))))   # channel.c:51
        for k in range(OH):
            od = pkg.Decoder(code, nm, nop, off, itmax, max_batch=Bo, device=local, early_exit=True)
            od.source_attach(src.s)
            od.source_draw_reference((1 << 20) + k * Bo, Bo, osigma)     # leaves codewords, noisy samples and LLRs resident
            od.synchronize()
            ods.append(od)
        src.close()

        def olaunch(od):
            od.llr_init_dev(od.noisy_dev_ptr, osigma, Bo)
            od.decode(Bo)

        def ocollect(od):
            nb = od.count_errors_resident(Bo)
            it = np.zeros(Bo, np.int32); sy = np.zeros(Bo, np.int32)
            pkg._check(pkg.lib().ems_get_results(od.h, Bo, None, pkg._ptr(it), pkg._ptr(sy)))
            return nb, it

        for od in ods:
            olaunch(od)
        for od in ods:
            ocollect(od)
        OK = 3 * OH
        t1 = time.perf_counter()
        for k in range(OK):
            if k >= OH:
                onb, oit = ocollect(ods[k % OH])
            olaunch(ods[k % OH])
        for k in range(OK, OK + OH):
            onb, oit = ocollect(ods[k % OH])
        odt = (time.perf_counter() - t1) / OK
        oper = {"frames_per_s": round(Bo / odt, 1), "EbN0_dB": OPER_EBN, "batch": Bo, "distinct_frames": OH * Bo,
                "frames": "the reference's drand48 stream, frames 2^20.. (generated on the device, bit-exact)",
                "avg_iterations": round(float(oit.mean()), 2), "frame_errors": int((onb != 0).sum()),
                "handles": OH,
                "note": "early exit on (NB_LDPC.c:470); reported beside, never instead of, the fixed-work value"}
        for od in ods:
            od.close()

    sweeps = itmax - 1
    alg = algorithmic_bytes_per_frame_decode(code.E, code.GF, nm, sweeps) * B
    kavg = float(np.mean(kms)) * 1e-3
    achieved = alg / kavg / 1e9
    res = {
        "value": round(world * B * steps / dt, 2), "ms_per_step": round(dt / steps * 1e3, 3), "data": data,
        "workload": "%s nm=%d NbIterMax=%d (%d sweeps) offset=%.1f NbOper=%d, fixed work (early exit off)" % (mat, nm, itmax, sweeps, off, nop),
        "batch_per_gpu": B, "counters": tot, "counters_match_golden": match, "golden_ranks_checked": ranks_checked,
        "roofline": {"bound": "hbm",
                     "kernel": "%s (persistent, %s, %d waves per workgroup, %d frames per launch on %d CUs)"
                               % (kernel, "1 wave per 2 frames" if "twin" in kernel else "1 workgroup per frame", wg_waves, B, n_cu),
                     "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None, "traffic_source": None,
                     "algorithmic_bytes_per_launch": alg, "avg_launch_ms": round(kavg * 1e3, 3),
                     "llr_init_avg_ms": round(float(np.mean(lms)), 3)},
        "host_setup_s": round(t_setup, 1), "oper": oper, "b256": b256,
        "matrix": matrix, "params": (nm, nop, off, itmax, ebn),
    }
    # HBM bytes per launch from the PMC passes kept under profiles/ (scripts/pmc_passes.sh on an earlier box, same kernel AND same kernel
    # sources): a committed measurement scaled to this batch, NOT a measurement of this run -- labelled as such, refused when stale
    res["roofline"]["traffic"], res["roofline"]["traffic_source"] = committed_traffic(workload, kernel, B)
    add_valu_bound(res["roofline"], workload, kernel, code.M * sweeps * B, kavg, n_cu, cache_resident=(workload == "n96"))
    return res


VALU_CLOCK_GHZ = 2.4        # MI355X peak engine clock (MI355X_MICROARCH.md); a wave64 VALU instruction occupies its SIMD for 4 cycles


def add_valu_bound(rf, name, kernel, visits_per_launch, launch_s, n_cu, cache_resident=False):
    """Every decode kernel of this library is closer to the VALU issue limit than to the HBM one, so the line
    carries BOTH bounds: `roofline.hbm` = algorithmic bytes / launch time against the 8 TB/s pin rate (what the contract defines, and what
    the top-level achieved / peak / frac keep saying), and `roofline.valu` = wave-level VALU instructions per second against CUs x 4 SIMDs x
    clock / 4 cycles per wave64 instruction, where the instruction count is SQ_INSTS_VALU per check-node visit from the committed SQ pass of
    the SAME kernel sources (profiles/traffic_<name>.json) x visits per launch / the launch time measured by THIS run.  `bound` names the
    larger fraction.  cache_resident (N96, BASELINE configs[1]: a frame's whole state lives in the caches, measured HBM traffic is BELOW the
    algorithmic bytes): the HBM figure bounds nothing there, the top-level numbers are the VALU ones."""
    rf["hbm"] = {k: rf[k] for k in ("achieved", "peak", "unit", "frac")}
    per_visit, why = None, "no SQ pass of the current kernel sources under profiles/ (re-run scripts/pmc_passes.sh)"
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "traffic_%s.json" % name)))
        if tj["kernel"].split("::")[-1].split("(")[0] != kernel:
            why = "profiles/traffic_%s.json measures another kernel (%s)" % (name, tj["kernel"].split("::")[-1].split("(")[0])
        elif tj.get("kernel_source_sha16") != kernel_source_hash():
            why = "profiles/traffic_%s.json was measured on other kernel sources (sha16 %s, now %s): refused" % (name, tj.get("kernel_source_sha16"), kernel_source_hash())
        else:
            per_visit = float(tj["sq_per_visit"]["SQ_INSTS_VALU"])
            why = ("profiles/traffic_%s.json: SQ_INSTS_VALU per check-node visit = %.1f (committed rocprofv3 --pmc pass, same kernel sources) x %d visits "
                   "/ this run's launch time" % (name, per_visit, visits_per_launch))
    except Exception as e:
        why = "profiles/traffic_%s.json: %s" % (name, e)
    peak = n_cu * 4 * VALU_CLOCK_GHZ / 4.0                     # G wave-instructions / s
    ach = None if per_visit is None else per_visit * visits_per_launch / launch_s / 1e9
    rf["valu"] = {"achieved": None if ach is None else round(ach, 2), "peak": round(peak, 2), "unit": "G wave-instr/s (VALU)",
                  "frac": None if ach is None else round(ach / peak, 5), "insts_per_visit": None if per_visit is None else round(per_visit, 1),
                  "source": why}
    vf, hf = rf["valu"]["frac"], rf["hbm"]["frac"]
    if cache_resident:
        rf.update({"bound": "valu", "achieved": rf["valu"]["achieved"], "peak": rf["valu"]["peak"], "unit": rf["valu"]["unit"], "frac": vf,
                   "valu_source": why})
    elif vf is not None:
        rf["bound"] = "valu" if vf > hf else "hbm"
        rf["bound_note"] = ("top-level achieved / peak / frac are the HBM figures the bench contract defines (algorithmic bytes / launch time); "
                            "`bound` names the larger of valu.frac and hbm.frac")


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(n, argv):
    """`python bench.py --gpus N` from a plain shell (no RANK / WORLD_SIZE): start N FRESH rank processes of this script, one per
    GPU, with the rendezvous variables torch.distributed expects.  This parent never imports torch and never touches HIP (a process
    that has initialised the GPU must not be replaced or forked on this pool); it relays rank 0's JSON line and fails loudly:
    the first rank that exits non-zero, or the deadline, kills every rank (each in its own session, by the exact process
    groups started here) and the parent exits non-zero.  The harness counters and the stop the all-reduce serves:
    NB_LDPC.c:491-506."""
    import signal
    import subprocess
    import threading
    timeout = float(os.environ.get("EMS_BENCH_LAUNCH_TIMEOUT", "1500"))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC only on this pool (RCCL across processes)
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
    cmd = [sys.executable, os.path.abspath(__file__)] + list(argv)
    procs = []
    for r in range(n):
        procs.append(subprocess.Popen(cmd, env=dict(env, RANK=str(r), LOCAL_RANK=str(r)), stdin=subprocess.DEVNULL,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, start_new_session=True))
    lines = []
    reader = threading.Thread(target=lambda: lines.extend(l.decode("utf-8", "replace") for l in procs[0].stdout), daemon=True)
    reader.start()

    def kill_all():
        for sig, grace in ((signal.SIGTERM, 5.0), (signal.SIGKILL, 5.0)):
            for p in procs:
                if p.poll() is None:
                    try:
                        os.killpg(p.pid, sig)                    # its own session: pgid == pid, nothing else is in it
                    except ProcessLookupError:
                        pass
            t_end = time.time() + grace
            while time.time() < t_end and any(p.poll() is None for p in procs):
                time.sleep(0.05)

    deadline = time.time() + timeout
    why = None
    try:
        while why is None:
            rcs = [p.poll() for p in procs]
            bad = [(r, rc) for r, rc in enumerate(rcs) if rc not in (None, 0)]
            if bad:
                why = "rank %d exited with code %d" % bad[0]
            elif all(rc == 0 for rc in rcs):
                break
            elif time.time() > deadline:
                why = "no result after %.0f s (EMS_BENCH_LAUNCH_TIMEOUT)" % timeout
            else:
                time.sleep(0.1)
    except BaseException:
        kill_all()
        raise
    if why is not None:
        kill_all()
        sys.stderr.write("bench.py --gpus %d: %s; every rank was stopped\n" % (n, why))
        raise SystemExit(1)
    reader.join(timeout=10)
    js = [l for l in lines if l.lstrip().startswith("{")]
    if len(js) != 1:
        sys.stderr.write("bench.py --gpus %d: rank 0 printed %d JSON lines instead of one\n" % (n, len(js)))
        raise SystemExit(1)
    sys.stdout.write(js[0] if js[0].endswith("\n") else js[0] + "\n")
    sys.stdout.flush()


def bench_fake(a, rank, world):
    """Launcher rehearsal (EMS_BENCH_FAKE_DECODE=1; tests/test_bench_launcher.py): the rank skeleton of run_decode_workload -- process
    group, barrier-fenced timed steps, the per-step all-reduce of the five counters, the per-rank golden check -- on the gloo backend
    with a stand-in for the decode that just reports the golden counters of the rank's frame range.  Nothing here is a measurement:
    the line says so in `metric`, `data` and `config`, and carries value 0."""
    import torch
    import torch.distributed as dist
    die, hang, corrupt = (int(os.environ.get("EMS_BENCH_FAKE_" + k, "-1")) for k in ("DIE_RANK", "HANG_RANK", "CORRUPT_RANK"))
    if world > 1:
        dist.init_process_group("gloo")
        if dist.get_world_size() != a.gpus:
            raise SystemExit("--gpus %d but the process group has %d ranks" % (a.gpus, dist.get_world_size()))
    gold = load_bench_golden("gf256")
    mine = list(gold["ranks"][str(rank)])
    if rank == corrupt:
        mine[2] += 1
    counters = torch.zeros(5, dtype=torch.int64)
    for s in range(a.warmup + a.steps):
        if s == a.warmup:
            if world > 1:
                dist.barrier()
            t0 = time.perf_counter()
        if rank == die and s == a.warmup:
            os._exit(3)
        if rank == hang and s == a.warmup:
            time.sleep(3600)
        time.sleep(0.01)
        counters.copy_(torch.tensor(mine, dtype=torch.int64))
        if world > 1:
            dist.all_reduce(counters)
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    flags = torch.tensor([1, 0 if mine == gold["ranks"][str(rank)] else 1], dtype=torch.int64)
    if world > 1:
        dist.all_reduce(flags)
    ranks_checked, bad = int(flags[0]), int(flags[1])
    check_ranks_against_golden(world, ranks_checked, bad)
    if rank == 0:
        print(json.dumps({"metric": "launcher rehearsal (EMS_BENCH_FAKE_DECODE: no decode, not a measurement)", "value": 0.0, "unit": "frames/s",
                          "n_gpus": world, "n_ranks_seen": dist.get_world_size() if world > 1 else 1, "steps": a.steps, "warmup": a.warmup,
                          "ms_per_step": round(dt / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                          "dtype": "none", "data": "fake", "config": {"workload": "fake decode", "frames_errframes_biterrs_undetected_sumiter": counters.tolist()},
                          "counters_match_golden": bad == 0, "golden_ranks_checked": ranks_checked}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def check_ranks_against_golden(world, ranks_checked, bad):
    """A multi-rank run of the committed workload must have EVERY rank's counters checked against the CPU oracle's
    (tests/golden/bench_counters.json holds ranks 0..7) and equal: anything else ends the run non-zero on every rank (the
    two numbers are all-reduced, so the ranks agree) instead of printing an aggregate nobody verified."""
    if world > 1 and (ranks_checked != world or bad != 0):
        sys.stderr.write("bench.py: %d of %d ranks were checked against the golden counters, %d differ: refusing to report\n" % (ranks_checked, world, bad))
        raise SystemExit(2)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="gf256", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-operational", action="store_true",
                    help="skip the side measurements (early exit enabled, batch 256; SURVEY.md 8(d) 'operational' figure)")
    ap.add_argument("--no-other-workloads", action="store_true",
                    help="skip the short runs of the other BASELINE configurations reported under config.other_workloads")
    ap.add_argument("--synthetic-zero-codeword", action="store_true",
                    help="skip the host encoder: all-zero codeword + numpy AWGN (faster set-up, same work)")
    a = ap.parse_args()
    if a.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(a.gpus, sys.argv[1:])               # plain shell: this process only starts and watches the N ranks

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (a.gpus, world))
    if os.environ.get("EMS_BENCH_FAKE_DECODE"):
        return bench_fake(a, rank, world)
    import torch  # first: its bundled libamdhip64 must be the one the decoder library binds to
    import __graft_entry__ as ge
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the decoder library has no CPU fallback")
    torch.cuda.set_device(local)
    dist = None
    if world > 1 or os.environ.get("EMS_BENCH_FORCE_DIST"):      # the env knob rehearses the RCCL path on one GPU
        import torch.distributed as dist
        with c_stdout_to_stderr():                  # RCCL prints its version banner on stdout when the communicator comes up
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
            dist.barrier()
            torch.cuda.synchronize()
        if dist.get_world_size() != a.gpus:         # a half-formed group would silently report a wrong aggregate
            raise SystemExit("--gpus %d but the process group has %d ranks" % (a.gpus, dist.get_world_size()))

    pkg = ge.load_package()
    if a.workload == "gf256_v2":
        return bench_v2(a, pkg, ge, torch, dist, rank, world, local)
    harness = importlib.import_module(ge.PKG_NAME + ".harness")
    B = a.batch or WORKLOADS[a.workload][6]
    main_res = run_decode_workload(a, pkg, harness, ge, torch, dist, rank, world, local, a.workload, B, a.steps, a.warmup,
                                   side_figures=not a.no_operational)
    # ---- the other BASELINE configurations (configs[1] and [2]) at their own batch, a few steps each: driver-timed side lines ----
    others = {}
    if a.workload == "gf256" and world == 1 and not a.no_other_workloads:
        for w in ("gf16", "n96", "irr345", "dc20"):
            nst, nwu = (5, 2) if w in ("gf16", "n96") else (2, 1)
            r = run_decode_workload(a, pkg, harness, ge, torch, None, rank, 1, local, w, WORKLOADS[w][6], nst, nwu, side_figures=False)
            others[w] = {"value": r["value"], "unit": "frames/s", "ms_per_step": r["ms_per_step"], "steps": nst, "warmup": nwu, "workload": r["workload"],
                         "batch": r["batch_per_gpu"], "frames_errframes_biterrs_undetected_sumiter": r["counters"],
                         "counters_match_golden": r["counters_match_golden"], "roofline": r["roofline"]}
        r = run_v2_workload(a, pkg, ge, torch, None, rank, 1, local, WORKLOADS["gf256_v2"][6], 3, 1)
        others["gf256_v2"] = {"value": r["value"], "unit": "frames/s", "ms_per_step": r["ms_per_step"], "steps": 3, "warmup": 1, "workload": r["workload"],
                              "batch": r["batch_per_gpu"], "frames_nonzero_syndromes_sumiter": r["counters"],
                              "counters_match_golden": r["counters_match_golden"], "roofline": r["roofline"], "data": r["data"]}
    if rank == 0:
        nm, nop, off, itmax, ebn = main_res["params"]
        out = {
            "metric": "decoded frames/sec @ N=64800 GF(256) R=1/2, nm=20, 30 iters; 1/2/4/8 GPU" if a.workload == "gf256"
                      else "decoded frames/sec (%s)" % a.workload,
            "value": main_res["value"], "unit": "frames/s", "n_gpus": world,
            "n_ranks_seen": dist.get_world_size() if dist is not None else 1, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": main_res["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": main_res["data"],
            "config": {"workload": main_res["workload"],
                       "batch_per_gpu": B, "sharding": "independent frames, contiguous batches per rank",
                       "collective": "all-reduce of 5 int64 error counters per step" if world > 1 else "none",
                       "frames_errframes_biterrs_undetected_sumiter": main_res["counters"]},
            # the counters of the timed run == the CPU oracle's on the same frames (tests/golden/bench_counters.json), rank by rank for the
            # ranks the golden covers (`golden_ranks_checked`); null when the run is not the committed workload (other batch,
            # --synthetic-zero-codeword, EMS_BENCH_EBN)
            "counters_match_golden": main_res["counters_match_golden"],
            "golden_ranks_checked": main_res["golden_ranks_checked"],
            "roofline": main_res["roofline"],
            "host_setup_s": main_res["host_setup_s"],
        }
        if main_res["oper"] is not None:
            out["config"]["operational_early_exit"] = main_res["oper"]
        if main_res["b256"] is not None:
            out["config"]["batch256_fixed_work"] = main_res["b256"]
        if others:
            out["config"]["other_workloads"] = others
        if world == 1 and not a.no_cpu_baseline:
            orc = ge.load_oracle()
            threads = max(1, min(os.cpu_count() or 1, 16))
            out["cpu_baseline"] = cpu_baseline(orc, main_res["matrix"], nm, nop, off, itmax, ebn, threads)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
