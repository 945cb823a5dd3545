"""Early-exit (operational) throughput: one handle vs. two handles ping-ponged on their own streams, so
that the tail of one launch (its few non-converging frames) overlaps the head of the next.  GPU box only."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
import __graft_entry__ as ge
import bench

pkg = ge.load_package()
harness = importlib.import_module(ge.PKG_NAME + ".harness")
mat, nm, nop, off, itmax, ebn, B0 = bench.WORKLOADS["gf256"]
matrix = os.path.join(ge.MATRICES, mat)
code = pkg.Code(matrix)
EBN = float(os.environ.get("OPER_EBN", "2.0"))
nd = 256
with bench.c_stdout_to_stderr():
    src = harness.FrameSource(matrix, nm)
    src.position(1 << 20)
    ocw, onoisy, osigma = src.draw(nd, EBN)
    src.close()


def mk(Bo):
    reps = -(-Bo // nd)
    cw = np.tile(ocw, (reps, 1))[:Bo]; no = np.tile(onoisy, (reps, 1, 1))[:Bo]
    if os.environ.get("OPER_SHUFFLE"):                                        # break the period-256 pattern of the tiled frames
        pm = np.random.RandomState(5).permutation(Bo)
        cw = np.ascontiguousarray(cw[pm]); no = np.ascontiguousarray(no[pm])
    od = pkg.Decoder(code, nm, nop, off, itmax, max_batch=Bo, device=0, early_exit=True)
    pkg._check(pkg.lib().ems_llr_init(od.h, pkg._ptr(no), np.float32(osigma), Bo))
    od.set_codewords(cw)
    od.synchronize()
    return od


def launch(od, Bo):
    od.llr_init_dev(od.noisy_dev_ptr, osigma, Bo)
    od.decode(Bo)


def collect(od, Bo):
    nb = od.count_errors_resident(Bo)
    it = np.zeros(Bo, np.int32); sy = np.zeros(Bo, np.int32)
    pkg._check(pkg.lib().ems_get_results(od.h, Bo, None, pkg._ptr(it), pkg._ptr(sy)))
    return nb, it


if os.environ.get("OPER_ANALYSE"):
    # where the early-exit mode loses against the fixed-work sweep rate: kernel time of one launch against (sum of the frames' sweeps) / (sweeps/s)
    Bf = 1536
    fd = pkg.Decoder(code, nm, nop, off, itmax, max_batch=Bf, device=0, early_exit=False)
    no = np.tile(onoisy, (Bf // nd, 1, 1))
    for _ in range(2):
        fd.llr_init(no, osigma); fd.decode(Bf); fd.synchronize()
    rate = Bf * (itmax - 1) / (fd.last_decode_ms * 1e-3)
    print("fixed work: %s %.1f ms per %d frames x %d sweeps = %.4g frame-sweeps/s" % (fd.last_kernel, fd.last_decode_ms, Bf, itmax - 1, rate), flush=True)
    fd.close()
    prev = None
    for im in (3, 6, 9, 12, 16, 21, 30):                                  # marginal cost of a sweep by its index (fixed work, same frames)
        fd = pkg.Decoder(code, nm, nop, off, im, max_batch=Bf, device=0, early_exit=False)
        for _ in range(2):
            fd.llr_init(no, osigma); fd.decode(Bf); fd.synchronize()
        ms = fd.last_decode_ms
        print("fixed work NbIterMax %2d: %.1f ms%s" % (im, ms, "" if prev is None else "  -> %.2f ms per sweep over sweeps %d..%d" % ((ms - prev[1]) / (im - prev[0]), prev[0], im - 1)), flush=True)
        prev = (im, ms)
        fd.close()
    for Bo in (1536, 3072, 6144, 12288):
        od = mk(Bo)
        for _ in range(2):
            launch(od, Bo); nb, it = collect(od, Bo)
        ms = od.last_decode_ms
        print("early exit B=%5d: %s x %d waves kernel %.1f ms = %.1f frames/s; sum of sweeps %d -> ideal %.1f ms (%.3f); avg it %.2f, max %d" %
              (Bo, od.last_kernel, od.last_workgroup_waves, ms, Bo / ms * 1e3, it.sum(), it.sum() / rate * 1e3, it.sum() / rate * 1e3 / ms, it.mean(), it.max()), flush=True)
        cls = np.array([it[x::8].sum() for x in range(8)], float)
        print("      sweeps of the frames = x mod 8 (one XCD each, if workgroups go round-robin): max / mean = %.3f  %s" % (cls.max() / cls.mean(), cls.astype(int).tolist()), flush=True)
        od.close()
    sys.exit(0)

for Bo in [int(x) for x in os.environ.get("OPER_B", "1024,2048,4096,8192").split(",")]:
    od = mk(Bo)
    launch(od, Bo); collect(od, Bo)
    t = time.perf_counter()
    for _ in range(3):
        launch(od, Bo); nb, it = collect(od, Bo)
    dt = (time.perf_counter() - t) / 3
    print("one handle  B=%5d: %7.1f frames/s  avg it %.2f  FE %d" % (Bo, Bo / dt, it.mean(), (nb != 0).sum()), flush=True)
    for H in (2, 3):
        hs = [od] + [mk(Bo) for _ in range(H - 1)]
        for h in hs:
            launch(h, Bo)
        for h in hs:
            collect(h, Bo)
        K = 9
        t = time.perf_counter()
        for k in range(K):
            if k >= H:
                collect(hs[k % H], Bo)
            launch(hs[k % H], Bo)
        for k in range(K, K + H):
            collect(hs[k % H], Bo)
        dt = (time.perf_counter() - t) / K
        print("%d handles   B=%5d: %7.1f frames/s" % (H, Bo, Bo / dt), flush=True)
        for h in hs[1:]:
            h.close()
    od.close()
