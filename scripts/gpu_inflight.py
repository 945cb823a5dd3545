"""Developer probe: two batches of 256 frames in flight on one GPU (two handles, two streams)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package()
name, nm, nop, itmax, ebn, B = "Ahmed_64800_R12_GF256", 20, 40, 30, 1.0, 256
code = pkg.Code(os.path.join(ge.MATRICES, name))
rs = np.random.RandomState(1)
sigma = float(np.sqrt(1.0 / (2 * float(code.rate) * 10 ** (ebn / 10))))
ninf = int(sys.argv[1]) if len(sys.argv) > 1 else 2
decs = []
for i in range(ninf):
    noisy = (1.0 + sigma * rs.randn(B, code.N, code.logGF)).astype(np.float32)
    d = pkg.Decoder(code, nm, nop, 0.3, itmax, max_batch=B, early_exit=False)
    d.llr_init(noisy, sigma); d.synchronize()
    decs.append(d)
for rep in range(3):
    t0 = time.perf_counter()
    for d in decs:
        if os.environ.get('WITH_LLR'): d.llr_init_dev(d.noisy_dev_ptr, sigma, B)
        d.decode(B)
    for d in decs: d.synchronize()
    dt = time.perf_counter() - t0
    print("inflight=%d waves=%s: %.1f ms wall -> %.1f frames/s (kernel ms %s)" % (ninf, os.environ.get("EMS_WAVES", "default"), dt * 1e3, ninf * B / dt,
          ["%.1f" % d.last_decode_ms for d in decs]), flush=True)
