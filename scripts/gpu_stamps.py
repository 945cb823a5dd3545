"""Developer probe: per-phase cycle shares of the decode kernel (needs build/libems_stamps.so, -DEMS_STAMPS)."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("EMS_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "build", "libems_stamps.so"))
import __graft_entry__ as ge
pkg = ge.load_package()
W = {"gf256": ("Ahmed_64800_R12_GF256", 20, 40, 30, 1.0, 256), "gf16": ("Ahmed_64800_R34_GF16", 8, 16, 30, 0.0, 256),
     "n96": ("N96_K48_GF64", 16, 32, 30, -3.0, 8192)}
names = ["load+Mvc", "select", "ES stages", "expand+store+decide", "level barrier wait", "syndrome", "wave idle (no CN)", "sweep-end barrier"]
for w in (sys.argv[1:] or ["gf256"]):
    B = None
    if ":" in w: w, B = w.split(":"); B = int(B)
    name, nm, nop, itmax, ebn, B0 = W[w]; B = B or B0
    code = pkg.Code(os.path.join(ge.MATRICES, name))
    rs = np.random.RandomState(1)
    sigma = float(np.sqrt(1.0 / (2 * float(code.rate) * 10 ** (ebn / 10))))
    noisy = (1.0 + sigma * rs.randn(B, code.N, code.logGF)).astype(np.float32)
    dec = pkg.Decoder(code, nm, nop, 0.3, itmax, max_batch=B, early_exit=False)
    L = pkg.lib(); L.ems_debug_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    dec.llr_init(noisy, sigma); dec.decode(B); dec.synchronize()
    L.ems_debug_stamps(dec.h, np.zeros(24, np.uint64).ctypes.data_as(C.c_void_p), 1)
    dec.llr_init(noisy, sigma); dec.decode(B); dec.synchronize()
    st = np.zeros(24, np.uint64); L.ems_debug_stamps(dec.h, st.ctypes.data_as(C.c_void_p), 0)
    ms = dec.last_decode_ms
    print("%s B=%d decode %.2f ms (stamped build)" % (w, B, ms))
    for grp, lab in ((0, "waves 0-8"), (8, "waves 9-15")):
        tot = float(st[grp:grp + 8].sum()) or 1.0
        print("  %s: " % lab + ", ".join("%s %.1f%%" % (names[i], 100.0 * float(st[grp + i]) / tot) for i in range(8)))
    sub = ["ES prologue (+stage overhead)", "ES merge loop", "ES tie check", "ES post-pass + continuation"]
    if float(st[8:12].sum()) > 0:   # the two-visit kernel reports everything in slots 0..7 and ES sub-phases in 8..11
        tot = float(st[:16].sum())
        print("  two-visit kernel, share of all wave time: " + ", ".join("%s %.1f%%" % (names[i], 100.0 * float(st[i]) / tot) for i in range(8))
              + " | " + ", ".join("%s %.1f%%" % (sub[i], 100.0 * float(st[8 + i]) / tot) for i in range(4)))
    print("  share of all wave time: " + ", ".join("%s %.1f%%" % (names[i], 100.0 * float(st[i] + st[8 + i]) / float(st[:16].sum())) for i in range(8)))
    w0 = st[0:8].astype(np.float64); nw = int(os.environ.get("EMS_WAVES", "9")) * B if os.environ.get("EMS_WAVES") else 9 * B
    per_wave_cycles = w0.sum() / nw
    visits = code.M * (itmax - 1) * B
    print("  cycles per visit by phase: " + ", ".join("%s %.0f" % (names[i], float(st[i] + st[8 + i]) / visits) for i in range(8)))
    print("  cycles per wave (waves 0-8 avg): %.3e ; => clock ~ %.2f GHz if the kernel took %.1f ms" % (per_wave_cycles, per_wave_cycles / (ms * 1e-3) / 1e9, ms))
    d = st[16:24].astype(np.float64)
    print("  selections(wave) %d, threshold iterations/selection %.2f, slow-path fraction %.4f; ES stages %d, continuation fraction %.4f, tie-fallback fraction %.4f, long-continuation fraction %.4f" % (d[0], d[1] / max(d[0], 1), d[2] / max(d[0], 1), d[3], d[4] / max(d[3], 1), d[5] / max(d[3], 1), d[6] / max(d[3], 1)))
    a = int(st[23])
    if a: print("  sorting-network selections %d, out-of-range fraction %.4f, near-tie fraction %.4f" % (a & 0x1FFFFF, ((a >> 21) & 0x1FFFFF) / max(a & 0x1FFFFF, 1), (a >> 42) / max(a & 0x1FFFFF, 1)))
    dec.close()
