"""Batched / multi-GPU Monte-Carlo driver around the GPU decoder.

Mirrors the reference's main() loop (NB_LDPC.c:250-507) for N processes, one per GPU:
  * frames are independent, so the global frame index f (1-based as in NB_LDPC.c:250) is sharded in
    contiguous batches: batch j = frames [j*B, (j+1)*B) goes to rank j mod R;
  * every rank positions the drand48 stream by LCG jump-ahead to draw offset
    (f-1) * (K*logGF + 2*N*logGF), so each frame sees exactly the uniforms the single-process reference
    would have drawn for it (source -> encoder -> channel order, NB_LDPC.c:255-261);
  * there is no data-path collective; per round the ranks exchange only counters:
    one all-reduce(sum) of {frames, erroneous frames, bit errors, undetected, sum of iterations}
    (5 x int64) drives the global ">= 40 erroneous frames" stop, and an all-gather of the per-frame
    (bit errors, syndrome, iterations) triples (12 B/frame) lets rank 0 print the reference's progress
    line in frame order and truncate at precisely the 40th erroneous frame (NB_LDPC.c:506).
Collectives go through torch.distributed (backend "nccl" == RCCL over xGMI on the GPU node, "gloo" in
the CPU tests); without an initialised process group everything degenerates to one rank.
"""
import ctypes as C
import os

import numpy as np

from . import Code, Decoder
# monte_carlo keeps `inflight` decoder handles busy, one HIP stream each, next to torch's and RCCL's streams; ROCm
# multiplexes all streams of a process onto GPU_MAX_HW_QUEUES (default 4) hardware queues and streams that share
# one serialise.  Only effective when this module is imported before the process's first HIP call.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

from .refapi import RefSession, seed48

LCG_A = 0x5DEECE66D
LCG_C = 0xB
LCG_MASK = (1 << 48) - 1


def lcg_jump(x0, n):
    """State of glibc's drand48 LCG after n draws from state x0 (affine map raised to the n-th power)."""
    a, c = 1, 0            # identity map x -> a*x + c
    ba, bc = LCG_A, LCG_C  # one step
    while n > 0:
        if n & 1:
            a, c = (ba * a) & LCG_MASK, (ba * c + bc) & LCG_MASK
        ba, bc = (ba * ba) & LCG_MASK, (ba * bc + bc) & LCG_MASK
        n >>= 1
    return (a * x0 + c) & LCG_MASK


def draws_per_frame(K, N, logGF):
    """K*logGF source bits (tools.c:124) + 2 uniforms per transmitted bit (channel.c:56-57)."""
    return K * logGF + 2 * N * logGF


def batch_of_rank(round_idx, rank, world, batch, n_frames):
    """(first frame index 0-based, count) of the batch `rank` decodes in `round_idx`; count may be 0."""
    j = round_idx * world + rank
    start = j * batch
    return start, max(0, min(batch, n_frames - start))


class FrameSource:
    """Draws frames exactly as the reference harness does, from any starting frame index.

    `threads` > 1 (default: the host's cores, at most 16) generates a batch with the library's
    `ems_draw_frames` (frame-parallel, LCG jump-ahead per frame, include/nbldpc_ref_api.h); `threads=1`
    replays the reference's own call sequence on libc's drand48.  Both give the same bits
    (tests/test_multi_rank_cpu.py)."""

    def __init__(self, matrix, nbMax, threads=None):
        self.s = RefSession(matrix, nbMax, need_encoder=True)
        self.dpf = draws_per_frame(self.s.K, self.s.N, self.s.logGF)
        self.threads = max(1, min(os.cpu_count() or 1, 16)) if threads is None else int(threads)
        self.next_frame = 0

    def position(self, frame0):
        self.next_frame = int(frame0)
        seed48(lcg_jump(0, frame0 * self.dpf))

    def draw(self, count, EbN):
        cws = np.zeros((count, self.s.N), np.int32)
        noisy = np.zeros((count, self.s.N, self.s.logGF), np.float32)
        sigma = 0.0
        if self.threads > 1 and count > 0:
            sg = C.c_float(0.0)
            L = self.s.L
            L.ems_draw_frames.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_int, C.c_float, C.c_int, C.c_void_p, C.c_void_p,
                                          C.POINTER(C.c_float)]
            rc = L.ems_draw_frames(C.byref(self.s.code), C.byref(self.s.table), self.next_frame, count, C.c_float(EbN),
                                   self.threads, cws.ctypes.data_as(C.c_void_p), noisy.ctypes.data_as(C.c_void_p), C.byref(sg))
            if rc != 0:
                raise RuntimeError("ems_draw_frames failed (%d)" % rc)
            sigma = float(sg.value)
            self.position(self.next_frame + count)      # keep libc's stream where the serial path would be
            return cws, noisy, sigma
        for b in range(count):
            cws[b] = self.s.draw_codeword()
            sigma, noisy[b] = self.s.noise(EbN)
        self.next_frame += count
        return cws, noisy, sigma

    def close(self):
        self.s.close()


class Counters:
    """The reference's running totals (NB_LDPC.c:73-74,186) and its progress line (:498)."""

    def __init__(self, K, logGF):
        self.K, self.logGF = K, logGF
        self.nb = self.err_frames = self.bit_errors = self.undetected = self.sum_it = 0
        self.trace = []

    def add(self, nberr, synd, iters):
        self.nb += 1
        self.sum_it += int(iters)
        self.bit_errors += int(nberr)
        if nberr != 0:
            self.err_frames += 1
            if synd == 0:
                self.undetected += 1
        self.trace.append((self.err_frames, self.bit_errors, "%.2f" % (self.sum_it / self.nb)))

    def line(self):
        return "<%d> FER= %d / %d = %f BER= %d / x = %f  avr_it=%.2f" % (
            self.undetected, self.err_frames, self.nb, self.err_frames / self.nb, self.bit_errors,
            self.bit_errors / (self.nb * self.K * self.logGF), self.sum_it / self.nb)


def _dist():
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist
    except Exception:
        pass
    return None


def monte_carlo(matrix, NbMonteCarlo, NbIterMax, EbN, nm, offset, NbOper, batch=64, device=0, decode_fn=None,
                max_err_frames=40, tensor_device="cpu", verbose=False, inflight=3, source="host", seed=0):
    """Run the Monte-Carlo loop over all ranks of the current process group (or one process).

    decode_fn(noisy, sigma, cws) -> (nberr[B], synd[B], iters[B]); default = the GPU decoder.  The CPU
    tests inject a stand-in so the sharding / reduction / ordering logic runs without a GPU.
    `inflight` decoder handles (each with its own stream) are used round-robin: round r+1 is launched before
    the results of round r are fetched, so the few frames of a round that run all iterations (its tail)
    share the GPU with the head of the next round.  Rounds are still tallied strictly in frame order; a
    round launched ahead of the 40-error stop is dropped, exactly like a frame drawn ahead of it.
    source="host" draws the reference's own frames (drand48 stream, bit-exact traces); source="device" lets
    the GPU generate frame f from the counter-based stream `seed` (ems_source_draw: no host work, no PCIe
    traffic; statistical parity only) -- frame f is the same on any rank count or batch;
    source="device_reference" (round 5) generates the REFERENCE's drand48 frames on the GPU, bit for bit (ems_source_draw_reference):
    the traces of source="host" with no host generation.
    Returns (Counters on every rank -- identical everywhere --, number of decode rounds)."""
    import torch
    dist = _dist()
    rank = dist.get_rank() if dist else 0
    world = dist.get_world_size() if dist else 1
    src = FrameSource(matrix, nm)
    decs = []
    if decode_fn is None:
        code = Code(matrix)
        inflight = max(1, int(inflight))
        decs = [Decoder(code, nm, NbOper, offset, NbIterMax, max_batch=batch, device=device) for _ in range(inflight)]

        if source in ("device", "device_reference"):
            for d in decs:
                d.source_attach(src.s)
            sigma_dev = float(np.float32(np.sqrt(1.0 / (2.0 * float(src.s.code.rate) * 10.0 ** (float(np.float32(EbN)) / 10.0)))))   # channel.c:51

        def submit(slot, noisy, sigma, cws):           # asynchronous: upload + ems_decode on the handle's stream
            d = decs[slot]
            B = d.llr_init(noisy, sigma)
            d.decode(B)

            def collect():
                _d, iters, synd = d.results(B, want_decide=False)     # the error count reads the decisions in HBM
                return d.count_errors(cws), synd, iters
            return collect

        def submit_device(slot, start, count):          # frames [start, start+count) generated and decoded in HBM
            d = decs[slot]
            if source == "device_reference":
                d.source_draw_reference(start, count, sigma_dev)
            else:
                d.source_draw(seed, start, count, sigma_dev)
            d.decode(count)

            def collect():
                _d, iters, synd = d.results(count, want_decide=False)
                return d.count_errors_resident(count), synd, iters
            return collect
    else:
        inflight = max(1, int(inflight))               # the stand-in decodes at once; its results are still tallied `inflight` rounds later

        def submit(slot, noisy, sigma, cws):
            res = decode_fn(noisy, sigma, cws)
            return lambda: res

    cnt = Counters(src.s.K, src.s.logGF)
    rounds = 0
    stop = False
    n_rounds = -(-NbMonteCarlo // (batch * world))

    on_device = source in ("device", "device_reference")
    if source not in ("host", "device", "device_reference") or (on_device and not decs):
        raise ValueError("source must be 'host', 'device' or 'device_reference' (the device sources need the GPU decoder)")

    def draw_round(r):                                  # this rank's frames of round r (host work)
        start, count = batch_of_rank(r, rank, world, batch, NbMonteCarlo)
        if count <= 0 or on_device:
            return None
        src.position(start)
        return src.draw(count, EbN)

    from concurrent.futures import ThreadPoolExecutor
    pool = ThreadPoolExecutor(1)                        # the next round is drawn while this one decodes
    pending = pool.submit(draw_round, 0)
    def tally(r, count, collect):                       # results of round r -> counters (collective when distributed)
        nonlocal rounds, stop
        triples = np.full((batch, 3), -1, np.int64)
        if count > 0:
            nberr, synd, iters = collect()
            triples[:count, 0], triples[:count, 1], triples[:count, 2] = nberr, synd, iters
        local = np.array([count, int((triples[:count, 0] != 0).sum()), int(triples[:count, 0].sum()),
                          int(((triples[:count, 0] != 0) & (triples[:count, 1] == 0)).sum()), int(triples[:count, 2].sum())],
                         np.int64)
        if dist:
            t = torch.from_numpy(local).to(tensor_device)
            dist.all_reduce(t, op=dist.ReduceOp.SUM)                 # the one collective of the data path
            g = torch.from_numpy(triples).to(tensor_device)
            gathered = [torch.empty_like(g) for _ in range(world)]
            dist.all_gather(gathered, g)                              # 12 B/frame, reporting only
            all_tr = [x.cpu().numpy() for x in gathered]
        else:
            all_tr = [triples]
        rounds += 1
        for rk in range(world):                                       # global frame order
            _s, c = batch_of_rank(r, rk, world, batch, NbMonteCarlo)
            for b in range(c):
                cnt.add(*all_tr[rk][b])
                if verbose and rank == 0:
                    print("\r" + cnt.line(), end="", flush=True)
                if cnt.err_frames == max_err_frames:                  # NB_LDPC.c:506
                    stop = True
                    return

    queue = []                                          # rounds launched, not yet tallied (oldest first)
    for r in range(n_rounds):
        start, count = batch_of_rank(r, rank, world, batch, NbMonteCarlo)
        frames = pending.result()
        if r + 1 < n_rounds:
            pending = pool.submit(draw_round, r + 1)
        collect = None
        if count > 0 and on_device:
            collect = submit_device(r % inflight, start, count)
        elif count > 0:
            cws, noisy, sigma = frames
            collect = submit(r % inflight, noisy, sigma, cws)
        queue.append((r, count, collect))
        if len(queue) == inflight:
            tally(*queue.pop(0))
            if stop:
                break
    while queue and not stop:
        tally(*queue.pop(0))
    for _r, _c, collect in queue:                       # launched ahead of the stop: wait, then drop
        if collect is not None:
            collect()
    pool.shutdown(wait=True)                            # (a round drawn ahead of the stop is simply dropped)
    src.close()
    for d in decs:
        d.close()
    return cnt, rounds
