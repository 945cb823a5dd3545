"""Developer timing probe: fixed-work decode time of the v2 flavour (Ahmed_64800_R12_GF256, nm=20, 29 iterations)."""
import os, sys, importlib, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package()
import bench
path = os.path.join(ge.MATRICES, os.environ.get("EMS_V2_MATRIX", "Ahmed_64800_R12_GF256"))
code = pkg.Code(path)
i2b, _ = pkg.gf_tables(256)
for B in [int(x) for x in (sys.argv[1:] or ["256"])]:
    base = [bench.synthetic_likelihood(code.N, 256, 8, i2b, 0.95, s) for s in range(4)]
    like = np.stack([base[i % 4][0] for i in range(B)]); idx = np.stack([base[i % 4][1] for i in range(B)])
    dec = pkg.Decoder(code, 20, 40, 0.3, 28, max_batch=B, early_exit=False)
    ts = []
    for rep in range(3):
        dec.v2_load_likelihood(like, idx); dec.v2_decode(B); dec.synchronize(); ts.append(dec.last_decode_ms)
    d, it, sy = dec.results(B)
    print("v2 gf256 B=%d decode ms %s -> %.1f frames/s (iters %d, synd0 %d)" % (B, ["%.1f" % t for t in ts], B / min(ts) * 1e3, it[0], int((sy == 0).sum())), flush=True)
    dec.close()
