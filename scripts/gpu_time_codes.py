"""Developer probe (GPU box): fixed-work decode time at batch 256 of the codes outside the BASELINE configurations -- check degrees 10 / 12
(with and without EMS_WIDE) / 20, the full-size irregular code -- with the kernel the library picked."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
for mat, wide in (("Irr345_N8100_M4050_GF256", "0"), ("Ahmed_64800_R910_GF256", "0"), ("Ahmed_64800_R45_GF256", "0"), ("Ahmed_64800_R45_GF256", "1"),
                  ("Ahmed_64800_R56_GF256", "0"), ("Ahmed_64800_R56_GF256", "1"), ("Ahmed_64800_R34_GF256", "0")):
    os.environ["EMS_WIDE"] = wide
    code = pkg.Code(os.path.join(ge.MATRICES, mat))
    rs = np.random.RandomState(0)
    sigma = 0.75
    noisy = (1.0 + sigma * rs.randn(B, code.N, code.logGF)).astype(np.float32)
    dec = pkg.Decoder(code, 20, 40, 0.3, 30, max_batch=B, early_exit=False)
    ms = []
    for _ in range(3):
        dec.llr_init(noisy, sigma); dec.decode(B); dec.synchronize(); ms.append(dec.last_decode_ms)
    print("%-28s EMS_WIDE=%s %-34s %2d waves: %8.1f ms per %d frames -> %7.0f frames/s" % (mat, wide, dec.last_kernel, dec.last_workgroup_waves, min(ms), B, B / min(ms) * 1e3), flush=True)
    dec.close()
