"""Developer probe: decode time of non-BASELINE parameter sets (generic-nm kernels), EMS_DUAL taken from the environment."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package()
mat, nm, nop, B = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
code = pkg.Code(os.path.join(ge.MATRICES, mat))
rs = np.random.RandomState(0)
sigma = 0.9
noisy = (1.0 + sigma * rs.randn(B, code.N, code.logGF)).astype(np.float32)
dec = pkg.Decoder(code, nm, nop, 0.3, 30, max_batch=B, early_exit=False)
dec.llr_init(noisy, sigma); dec.decode(B); dec.synchronize()
ms = []
for _ in range(3):
    dec.llr_init(noisy, sigma); dec.decode(B); dec.synchronize(); ms.append(dec.last_decode_ms)
print("%s nm=%d nbOper=%d B=%d EMS_DUAL=%s: %.1f ms -> %.0f frames/s" % (mat, nm, nop, B, os.environ.get("EMS_DUAL"), np.median(ms), B / np.median(ms) * 1e3))
dec.close()
