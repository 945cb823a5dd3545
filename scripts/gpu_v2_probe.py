"""Developer probe: one small v2 decode with progress prints."""
import os, sys, importlib, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package(); orc = ge.load_oracle(); orc.build()
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
v2 = importlib.import_module("orc_v2")
mat, q, lq, nm, nop, off = (sys.argv[1] if len(sys.argv) > 1 else "N96_K48_GF64"), 64, 6, 12, 24, 0.5
if "256" in mat: q, lq, nm, nop, off = 256, 8, 20, 40, 0.3
itmax = int(sys.argv[2]) if len(sys.argv) > 2 else 0
path = os.path.join(ge.MATRICES, mat)
o = v2.V2Oracle(path)
like, idx = v2.synthetic_likelihood(o.N, q, lq, o.o.idx2bin, 0.9, 1)
print("oracle...", flush=True)
ref = o.decode(like, idx, nm, nop, off, itmax, early_exit=False)
print("ref iters", ref["iters"], "synd", ref["synd"], flush=True)
dec = pkg.Decoder(pkg.Code(path), nm, nop, off, itmax, max_batch=1, early_exit=False)
print("load...", flush=True)
dec.v2_load_likelihood(like[None], idx[None])
print("decode...", flush=True)
t = time.time(); dec.v2_decode(1); print("launched %.3fs" % (time.time() - t), flush=True); dec.synchronize(); print("synced %.3fs" % (time.time() - t), flush=True); d, it, sy = dec.results(1)
print("gpu iters", it, "synd", sy, "%.3fs" % (time.time() - t), flush=True)
st = dec.v2_state(0)
print("decide eq", (d[0] == ref["decide"]).all(), {k: st[k].tobytes() == ref[k].tobytes() for k in ("vtoc", "vtoci", "ctov", "ctovi", "app")}, flush=True)
for k in ("ctov", "ctovi", "vtoc", "vtoci"):
    if st[k].tobytes() != ref[k].tobytes():
        bad = np.argwhere(st[k] != ref[k])
        print(k, "first diffs", bad[:5].tolist(), st[k][tuple(bad[0])], ref[k][tuple(bad[0])], "count", len(bad))
