import os, sys, importlib
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package(); refapi = importlib.import_module(ge.PKG_NAME + ".refapi")
path = os.path.join(ge.MATRICES, "N96_K48_GF64")
s = refapi.RefSession(path, 16)
refapi.seed48(0)
for f in range(3):
    cw = s.draw_codeword(); print("cw", cw[:4], flush=True)
    sl, sg = s.model_channel(2.0); print("channel ok", sl[0, :2], flush=True)
    pkg.lib().ems_synchronize  # noqa
    it, d, sy = s.decoding(10, 32, 0.3); print("decoding ok", it, sy, flush=True)
s.close(); print("done")
