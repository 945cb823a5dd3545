"""Developer probe (GPU box): k_decode_wide on the small wide-degree codes, step by step, with progress lines."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
pkg = ge.load_package(); orc = ge.load_oracle()
def say(*a):
    print(*a, flush=True)
import signal
def on_alarm(*a):
    say("  TIMEOUT: still waiting after 40 s; giving up on the process"); os._exit(3)
signal.signal(signal.SIGALRM, on_alarm)
which = int(sys.argv[1]) if len(sys.argv) > 1 else -1
for ci, (name, nm, nop, itmax, ebn, B) in enumerate((("Wide_N80_M8_GF64", 16, 32, 2, 4.5, 1), ("Wide_N80_M8_GF64", 16, 32, 10, 4.5, 4),
                                      ("Wide_N80_M8_GF256", 20, 40, 3, 4.5, 2), ("Wide_N120_M6_GF16", 8, 16, 3, 5.0, 2), ("Ahmed_64800_R910_GF256", 20, 40, 3, 4.0, 1))):
    if which >= 0 and ci != which:
        continue
    signal.alarm(40)
    path = os.path.join(ge.MATRICES, name)
    o = orc.Oracle(path)
    sigma = o.sigma(ebn)
    rng = orc.Rng(0)
    frames = [o.draw_frame(rng, sigma) for _ in range(B)]
    noisy = np.stack([f[1] for f in frames])
    say("create", name, nm, itmax, B)
    code = pkg.Code(path)
    say("  code loaded")
    dec = pkg.Decoder(code, nm, nop, 0.3, itmax, max_batch=B, early_exit=True)
    say("  decoder created")
    t0 = time.time()
    try:
        dec.llr_init(noisy, sigma); dec.synchronize(); say("  llr done")
        dec.decode(B); say("  decode launched")
        decide, iters, synd = dec.results(B)
    except Exception as e:
        say("  decode raised:", e, "after %.2f s" % (time.time() - t0)); continue
    say("  decode done in %.3f s, kernel %s, iters %s synd %s" % (time.time() - t0, dec.last_kernel, iters.tolist(), synd.tolist()))
    for b in range(B):
        ref = o.decode(o.llr(noisy[b], sigma), nm, nop, 0.3, itmax, want_state=True)
        say("  frame", b, "decide", bool((decide[b] == ref["decide"]).all()), "iters", iters[b] == ref["iters"], "app", dec.app(b).tobytes() == ref["app"].tobytes(),
            "ctov", dec.ctov(b).tobytes() == ref["ctov"].tobytes())
    dec.close()
