import os, sys, importlib
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package(); orc = ge.load_oracle()
path = os.path.join(ge.MATRICES, "N96_K48_GF64")
o = orc.Oracle(path); code = pkg.Code(path)
sigma = o.sigma(2.0); rng = orc.Rng(0)
frames = [o.draw_frame(rng, sigma)[1] for _ in range(4)]
mode = sys.argv[1]
if mode == "toggle":
    dec = pkg.Decoder(code, 16, 32, 0.0, 2, max_batch=1)
    for f in range(4):
        dec.set_params(NbIterMax=2, offset=0.0); dec.llr_init(frames[f][None], sigma); dec.synchronize(); print("llr", f, flush=True)
        dec.set_params(NbIterMax=10, offset=0.3); dec.decode(1); print("decode", f, dec.results(1)[1:], flush=True)
elif mode == "plain":
    dec = pkg.Decoder(code, 16, 32, 0.3, 10, max_batch=1)
    for f in range(4):
        print("decode", f, dec.decode_batch(frames[f][None], sigma)[1:], flush=True)
elif mode == "getters":
    dec = pkg.Decoder(code, 16, 32, 0.3, 10, max_batch=1)
    for f in range(4):
        import ctypes; pkg.lib().ems_debug_dump.argtypes=[ctypes.c_void_p]; pkg.lib().ems_debug_dump(dec.h)
        dec.llr_init(frames[f][None], sigma); sl, sg = dec.sorted_llr(0); print("sorted", f, flush=True)
        dec.decode(1); print("decode", f, dec.results(1)[1:], flush=True)
        a = dec.app(0); c = dec.ctov(0); print("state", f, flush=True)
