"""Developer probe (GPU box): fixed-work decode time of any matrix at a batch size, whatever kernel the library picks (or EMS_* forces).
usage: python scripts/gpu_time_codes.py <matrix> <nm> <batch> [sweeps=29] [EbN0=2.0]   -> kernel, ms per launch, frames/s
       EMS_V2=1 python scripts/gpu_time_codes.py ...   the same for the v2 flavour (seeded synthetic likelihoods, sigma 0.95)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package()
mat, nm, B = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
sweeps = int(sys.argv[4]) if len(sys.argv) > 4 else 29
ebn = float(sys.argv[5]) if len(sys.argv) > 5 else 2.0
code = pkg.Code(os.path.join(ge.MATRICES, mat))
if os.environ.get("EMS_V2"):
    import bench
    i2b, _ = pkg.gf_tables(code.GF)
    base = [bench.synthetic_likelihood(code.N, code.GF, code.logGF, i2b, 0.95, s) for s in range(4)]
    like = np.stack([base[i % 4][0] for i in range(B)]); idx = np.stack([base[i % 4][1] for i in range(B)])
    dec = pkg.Decoder(code, nm, 2 * nm, 0.3, sweeps - 1, max_batch=B, early_exit=False)
    ts = []
    for rep in range(3):
        dec.v2_load_likelihood(like, idx); dec.v2_decode(B); dec.synchronize(); ts.append(dec.last_decode_ms)
    print("v2 %s nm=%d B=%d %s x %d waves: %s ms -> %.1f frames/s" % (mat, nm, B, dec.last_kernel, dec.last_workgroup_waves, ["%.1f" % t for t in ts], B / min(ts) * 1e3), flush=True)
    dec.close()
    sys.exit(0)
sigma = float(np.sqrt(1.0 / (2 * float(code.rate) * 10 ** (ebn / 10))))
noisy = (1.0 + sigma * np.random.RandomState(1).randn(min(B, 16), code.N, code.logGF)).astype(np.float32)
noisy = np.tile(noisy, (-(-B // noisy.shape[0]), 1, 1))[:B]
dec = pkg.Decoder(code, nm, 2 * nm, 0.3, sweeps + 1, max_batch=B, early_exit=False)
ts = []
for rep in range(3):
    dec.llr_init(noisy, sigma); dec.decode(B); dec.synchronize(); ts.append(dec.last_decode_ms)
print("%s nm=%d B=%d %s x %d waves: %s ms -> %.1f frames/s" % (mat, nm, B, dec.last_kernel, dec.last_workgroup_waves, ["%.1f" % t for t in ts], B / min(ts) * 1e3), flush=True)
dec.close()
