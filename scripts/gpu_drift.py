"""Developer probe: decode time of 150 back-to-back launches (does sustained load move the clock?)."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package()
code = pkg.Code(os.path.join(ge.MATRICES, "Ahmed_64800_R12_GF256"))
B = 1024
rs = np.random.RandomState(1)
sigma = float(np.sqrt(1.0 / (2 * float(code.rate) * 10 ** (1.0 / 10))))
noisy = (1.0 + sigma * rs.randn(B, code.N, code.logGF)).astype(np.float32)
dec = pkg.Decoder(code, 20, 40, 0.3, 30, max_batch=B, early_exit=False)
dec.llr_init(noisy, sigma)
ts = []
for rep in range(150):
    dec.llr_init_dev(dec.noisy_dev_ptr, sigma, B); dec.decode(B); dec.synchronize(); ts.append(dec.last_decode_ms)
    if rep % 10 == 9:
        print("launch %3d: last 10 avg %.2f ms" % (rep + 1, np.mean(ts[-10:])), flush=True)
