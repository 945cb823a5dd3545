"""Developer probe (GPU box): time of the three frame sources for one batch of the headline code -- the reference's drand48 stream on the
device (ems_source_draw_reference), the same stream on the host cores (ems_draw_frames, 16 threads) and the counter-based device stream."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package()
import importlib
harness = importlib.import_module(ge.PKG_NAME + ".harness")
refapi = importlib.import_module(ge.PKG_NAME + ".refapi")
mat = os.path.join(ge.MATRICES, sys.argv[1] if len(sys.argv) > 1 else "Ahmed_64800_R12_GF256")
nm = int(sys.argv[2]) if len(sys.argv) > 2 else 20
B = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
s = refapi.RefSession(mat, nm, need_encoder=True)
dec = pkg.Decoder(pkg.Code(mat), nm, 2 * nm, 0.3, 30, max_batch=B)
dec.source_attach(s)
sigma = 0.7
for rep in range(3):
    t0 = time.perf_counter(); n = dec.source_draw_reference(rep * B, B, sigma); dec.synchronize(); t1 = time.perf_counter()
    print("device drand48 stream: %d frames in %.1f ms (%d samples recomputed on the host)" % (B, (t1 - t0) * 1e3, n), flush=True)
for rep in range(2):
    t0 = time.perf_counter(); dec.source_draw(7, rep * B, B, sigma); dec.synchronize(); t1 = time.perf_counter()
    print("device counter-based stream: %d frames in %.1f ms" % (B, (t1 - t0) * 1e3), flush=True)
src = harness.FrameSource(mat, nm)
t0 = time.perf_counter(); src.draw(B, 2.0); t1 = time.perf_counter()
print("host drand48 stream (%d threads): %d frames in %.1f ms" % (src.threads, B, (t1 - t0) * 1e3), flush=True)
