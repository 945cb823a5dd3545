"""Developer probe (GPU box): k_decode_q16t against the oracle frame by frame, then its fixed-work time at batch 256 next to the two-visit kernel."""
import os, sys, time, signal
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
pkg = ge.load_package(); orc = ge.load_oracle()
def say(*a): print(*a, flush=True)
path = os.path.join(ge.MATRICES, "Ahmed_64800_R34_GF16")
o = orc.Oracle(path); code = pkg.Code(path)
for ebn, itmax, B, ee in ((3.0, 30, 2, True), (0.0, 4, 2, False), (1.5, 12, 3, False)):
    sigma = o.sigma(ebn); rng = orc.Rng(0)
    frames = [o.draw_frame(rng, sigma) for _ in range(B)]
    noisy = np.stack([f[1] for f in frames])
    dec = pkg.Decoder(code, 8, 16, 0.3, itmax, max_batch=B, early_exit=ee)
    t0 = time.time()
    dec.llr_init(noisy, sigma); dec.decode(B); say("launched", ebn, itmax, B)
    decide, iters, synd = dec.results(B)
    say("  done in %.3f s kernel %s iters %s synd %s" % (time.time() - t0, dec.last_kernel, iters.tolist(), synd.tolist()))
    for b in range(B):
        ref = o.decode(o.llr(noisy[b], sigma), 8, 16, 0.3, itmax, early_exit=ee, want_state=True)
        say("  frame", b, "decide", bool((decide[b] == ref["decide"]).all()), "iters", iters[b] == ref["iters"], "synd", synd[b] == ref["synd"],
            "app", dec.app(b).tobytes() == ref["app"].tobytes(), "ctov", dec.ctov(b).tobytes() == ref["ctov"].tobytes())
    dec.close()
if len(sys.argv) > 1 and sys.argv[1] == "time":
    B = 256
    rs = np.random.RandomState(1)
    sigma = o.sigma(0.0)
    noisy = (1.0 + sigma * rs.randn(B, code.N, code.logGF)).astype(np.float32)
    for env, wv in (("1", "12"), ("1", "10"), ("1", "8"), ("0", "0")):
        os.environ["EMS_Q16T"] = env
        os.environ["EMS_Q16T_WAVES"] = wv
        dec = pkg.Decoder(code, 8, 16, 0.3, 30, max_batch=B, early_exit=False)
        ts = []
        for rep in range(3):
            dec.llr_init(noisy, sigma); dec.decode(B); dec.synchronize(); ts.append(dec.last_decode_ms)
        say("EMS_Q16T=%s waves %d kernel %s: decode ms %s -> %.1f frames/s" % (env, dec.last_workgroup_waves, dec.last_kernel, ["%.2f" % t for t in ts], B / min(ts) * 1e3))
        dec.close()
