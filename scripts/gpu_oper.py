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
