"""Developer probe: duration and HBM rate of the channel-LLR kernels (k_llr_init, k_llr_const) at the bench batch."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1536
code = pkg.Code(os.path.join(ge.MATRICES, "Ahmed_64800_R12_GF256"))
dec = pkg.Decoder(code, 20, 40, 0.3, 30, max_batch=B, early_exit=False)
rs = np.random.RandomState(0)
N, q = code.N, code.GF
out_bytes = B * N * q * 4
def run(name, fn, in_bytes):
    fn(); dec.synchronize()
    ms = []
    for _ in range(5):
        fn(); dec.synchronize(); ms.append(dec.last_llr_ms)
    m = float(np.median(ms))
    print("%-28s %8.3f ms  %7.1f GB/s (%.2f GB out + %.3f GB in)" % (name, m, (out_bytes + in_bytes) / m / 1e6, out_bytes / 1e9, in_bytes / 1e9))
noisy = (1.0 + 0.8 * rs.randn(B, N, 8)).astype(np.float32)
run("k_llr_init (BPSK)", lambda: dec.llr_init(noisy, 0.8), noisy.nbytes)
for D, mode, nm in ((2, pkg.LLR_SUM, "k_llr_const SUM D=2"), (2, pkg.LLR_SSD, "k_llr_const SSD D=2"), (4, pkg.LLR_SSD, "k_llr_const SSD D=4")):
    obs = rs.randn(B, N, D).astype(np.float32)
    att = np.abs(rs.randn(B, N, D)).astype(np.float32) * (rs.rand(B, N, D) > 0.1)
    tab = rs.randn(q, D).astype(np.float32)
    run(nm, lambda: dec.llr_init_constellation(obs, att, tab, mode, 0.3), obs.nbytes * 2)
dec.close()
