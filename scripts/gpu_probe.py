"""Developer probe (runs on the GPU box): unit + decode parity vs the oracle and first timings."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package(); orc = ge.load_oracle()
MAT = ge.MATRICES
what = sys.argv[1:] or ["es", "cn", "n96", "gf16", "gf256", "time"]

def rand_list(rs, q, nm, short=0):
    v = np.sort(rs.randint(0, 60, nm).astype(np.float32) * 0.25); v[0] = 0
    g = rs.permutation(q)[:nm].astype(np.int32)
    if short:
        v[nm - short:] = 1e5; g[nm - short:] = -1
    return v, g

if "es" in what:
    bad = 0; n = 0
    for q, nm in ((64, 16), (256, 20), (16, 8), (256, 8)):
        o = orc.Oracle(os.path.join(MAT, "N96_K48_GF64")); o.L.orc_gf_init(orc.C.byref(o.gf), q); o.q = q
        rs = np.random.RandomState(q + nm)
        for it in range(60):
            a, ia = rand_list(rs, q, nm, short=rs.randint(0, 4) if it % 3 == 0 else 0)
            b, ib = rand_list(rs, q, nm, short=rs.randint(0, 4) if it % 5 == 0 else 0)
            nop = int(rs.choice([nm, nm + 3, 2 * nm]))
            r0 = o.elementary_step(a, b, ia, ib, nm, nop)
            r1 = pkg.ElementaryStep(a, b, ia, ib, q, nm, nop)
            n += 1
            if not (r0[0].tobytes() == r1[0].tobytes() and (r0[1] == r1[1]).all()):
                bad += 1
                if bad < 3: print("ES mismatch", q, nm, nop, "\n", r0, "\n", r1)
    print("ES unit: %d cases, %d mismatches" % (n, bad), flush=True)

if "cn" in what:
    bad = 0; n = 0
    for q, nm, dc in ((64, 16, 4), (256, 20, 4), (16, 8, 8), (64, 12, 3), (64, 12, 6), (256, 10, 5), (64, 8, 2)):
        o = orc.Oracle(os.path.join(MAT, "N96_K48_GF64")); o.L.orc_gf_init(orc.C.byref(o.gf), q); o.q = q
        rs = np.random.RandomState(q + nm + dc)
        for it in range(25):
            vl = np.zeros((dc, nm), np.float32); vg = np.zeros((dc, nm), np.int32)
            for t in range(dc): vl[t], vg[t] = rand_list(rs, q, nm)
            coef = rs.randint(1, q, dc).astype(np.int32)
            nop = int(rs.choice([nm, 2 * nm]))
            r0 = o.check_node(coef, vl, vg, nop, 0.3)
            r1 = pkg.CheckPassLogEMS(coef, vl, vg, q, nop, 0.3)
            n += 1
            if r0.tobytes() != r1.tobytes():
                bad += 1
                if bad < 3: print("CN mismatch", q, nm, dc, nop, np.argwhere(r0 != r1)[:5])
    print("CN unit: %d cases, %d mismatches" % (n, bad), flush=True)

def decode_parity(name, nm, nop, off, itmax, ebn, B, check_state=True):
    path = os.path.join(MAT, name)
    o = orc.Oracle(path); code = pkg.Code(path)
    sigma = o.sigma(ebn); rng = orc.Rng(0)
    frames = [o.draw_frame(rng, sigma) for _ in range(B)]
    noisy = np.stack([f[1] for f in frames]); cws = np.stack([f[0] for f in frames])
    dec = pkg.Decoder(code, nm, nop, off, itmax, max_batch=B)
    t = time.time(); decide, iters, synd = dec.decode_batch(noisy, sigma); tg = time.time() - t
    nerr = dec.count_errors(cws)
    bad = 0; t = time.time()
    for b in range(B):
        llr = o.llr(noisy[b], sigma)
        ref = o.decode(llr, nm, nop, off, itmax, want_state=check_state)
        ok = (decide[b] == ref["decide"]).all() and iters[b] == ref["iters"] and synd[b] == ref["synd"]
        ok = ok and nerr[b] == o.bit_errors(ref["decide"], cws[b])
        if check_state:
            ok = ok and dec.app(b).tobytes() == ref["app"].tobytes() and dec.ctov(b).tobytes() == ref["ctov"].tobytes()
        if not ok:
            bad += 1
            if bad < 3:
                print("  frame", b, "gpu iters/synd", iters[b], synd[b], "ref", ref["iters"], ref["synd"],
                      "decide diff", int((decide[b] != ref["decide"]).sum()))
                if check_state:
                    a = dec.app(b); print("   app diffs", int((a != ref["app"]).sum()), "of", a.size)
    print("%s: B=%d gpu %.3fs (kernel %.2f ms, llr %.2f ms) oracle %.1fs; mismatching frames: %d; iters %s"
          % (name, B, tg, dec.last_decode_ms, dec.last_llr_ms, time.time() - t, bad, iters[:8].tolist()), flush=True)
    dec.close()

if "n96" in what: decode_parity("N96_K48_GF64", 16, 32, 0.3, 10, 2.0, 64)
if "gf16" in what: decode_parity("Ahmed_64800_R34_GF16", 8, 16, 0.3, 30, 3.0, 2)
if "gf256" in what: decode_parity("Ahmed_64800_R12_GF256", 20, 40, 0.3, 30, 1.6, 2)

if "time" in what:
    for name, nm, nop, itmax, ebn, B in (("N96_K48_GF64", 16, 32, 30, -3.0, 8192), ("Ahmed_64800_R34_GF16", 8, 16, 30, 0.0, 256),
                                         ("Ahmed_64800_R12_GF256", 20, 40, 30, 1.0, 256)):
        path = os.path.join(MAT, name); code = pkg.Code(path)
        rs = np.random.RandomState(1)
        sigma = float(np.sqrt(1.0 / (2 * float(code.rate) * 10 ** (ebn / 10))))
        noisy = (1.0 + sigma * rs.randn(B, code.N, code.logGF)).astype(np.float32)   # all-zero codeword + AWGN
        dec = pkg.Decoder(code, nm, nop, 0.3, itmax, max_batch=B, early_exit=False)
        for rep in range(2):
            dec.llr_init(noisy, sigma); dec.decode(B); dec.synchronize()
            print("%s B=%d fixed-work: decode %.2f ms, llr %.2f ms -> %.1f frames/s" %
                  (name, B, dec.last_decode_ms, dec.last_llr_ms, B / (dec.last_decode_ms + dec.last_llr_ms) * 1e3), flush=True)
        dec.close()
