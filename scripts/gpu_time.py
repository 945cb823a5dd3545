"""Developer timing probe: fixed-work decode time of one workload (EMS_LIB may point at an ablation build)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package()
W = {"gf256": ("Ahmed_64800_R12_GF256", 20, 40, 30, 1.0, 256), "gf16": ("Ahmed_64800_R34_GF16", 8, 16, 30, 0.0, 256),
     "n96": ("N96_K48_GF64", 16, 32, 30, -3.0, 8192), "gf256r56": ("Ahmed_64800_R56_GF256", 20, 40, 30, 3.0, 256),
     "gf256r34": ("Ahmed_64800_R34_GF256", 20, 40, 30, 2.0, 256)}
for w in (sys.argv[1:] or ["gf256"]):
    B = None
    if ":" in w: w, B = w.split(":"); B = int(B)
    name, nm, nop, itmax, ebn, B0 = W[w]; B = B or B0
    code = pkg.Code(os.path.join(ge.MATRICES, name))
    rs = np.random.RandomState(1)
    sigma = float(np.sqrt(1.0 / (2 * float(code.rate) * 10 ** (ebn / 10))))
    noisy = (1.0 + sigma * rs.randn(B, code.N, code.logGF)).astype(np.float32)
    dec = pkg.Decoder(code, nm, nop, 0.3, itmax, max_batch=B, early_exit=False)
    ts = []
    for rep in range(3):
        dec.llr_init(noisy, sigma); dec.decode(B); dec.synchronize(); ts.append(dec.last_decode_ms)
    print("%s lib=%s B=%d decode ms %s -> %.1f frames/s" % (w, os.path.basename(pkg.LIB_PATH), B, ["%.2f" % t for t in ts], B / min(ts) * 1e3), flush=True)
    dec.close()
