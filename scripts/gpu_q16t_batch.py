"""Developer probe (GPU box): k_decode_q16t fixed-work rate over batch sizes and workgroup sizes (is a frame bound by its dependency chain?)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package()
code = pkg.Code(os.path.join(ge.MATRICES, "Ahmed_64800_R34_GF16"))
rs = np.random.RandomState(1)
sigma = 0.84
for B, waves in ((256, 12), (512, 12), (512, 6), (768, 4), (1024, 6), (1024, 3), (1536, 4), (1536, 2), (2048, 3)):
    os.environ["EMS_Q16T_WAVES"] = str(waves)
    noisy = (1.0 + sigma * rs.randn(B, code.N, code.logGF)).astype(np.float32)
    dec = pkg.Decoder(code, 8, 16, 0.3, 30, max_batch=B, early_exit=False)
    ts = []
    for rep in range(3):
        dec.llr_init(noisy, sigma); dec.decode(B); dec.synchronize(); ts.append(dec.last_decode_ms)
    print("B %4d waves %2d (%s): %.2f ms -> %.0f frames/s" % (B, dec.last_workgroup_waves, dec.last_kernel, min(ts), B / min(ts) * 1e3), flush=True)
    dec.close()
