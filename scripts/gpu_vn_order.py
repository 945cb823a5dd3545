"""Developer probe: does the placement of the APP rows in HBM matter?  The variable nodes of the headline code are renumbered
(identity / random / along an Euler tour of the check-node graph, so that every check node's rows form two adjacent pairs /
by first visit in the pair schedule) before the handle is created -- the decoder then simply works on a relabelled code, the
dependency structure and the work are the same, only the rows move -- and the fixed-work decode is timed.  GPU box only."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package()
name, nm, nop, itmax, ebn = "Ahmed_64800_R12_GF256", 20, 40, 30, 1.0
B = int(os.environ.get("VN_B", "1536"))


def euler_order(col, M, N):
    """VNs = edges of the multigraph on check nodes (dv = 2): order them along Euler circuits (Hierholzer)."""
    ends = [[] for _ in range(N)]
    for c in range(M):
        for v in col[c]:
            ends[v].append(c)
    adj = [[] for _ in range(M)]                       # (vn, other check)
    for v in range(N):
        a, b = ends[v]
        adj[a].append((v, b)); adj[b].append((v, a))
    used = np.zeros(N, bool); ptr = [0] * M; order = []
    for s in range(M):
        if ptr[s] >= len(adj[s]):
            continue
        stack, path = [(s, -1)], []
        while stack:
            c, via = stack[-1]
            while ptr[c] < len(adj[c]) and used[adj[c][ptr[c]][0]]:
                ptr[c] += 1
            if ptr[c] == len(adj[c]):
                stack.pop()
                if via >= 0:
                    path.append(via)
            else:
                v, o = adj[c][ptr[c]]
                used[v] = True
                stack.append((o, v))
        order.extend(path[::-1])
    assert len(order) == N and len(set(order)) == N
    pi = np.zeros(N, np.int32); pi[np.array(order)] = np.arange(N, dtype=np.int32)
    return pi


def first_visit_order(col, M, N, pairs):
    seen = np.full(N, -1, np.int64); k = 0
    for a, b in pairs:
        for c in (a, b):
            if c < 0:
                continue
            for v in col[c]:
                if seen[v] < 0:
                    seen[v] = k; k += 1
    assert k == N
    return seen.astype(np.int32)


code0 = pkg.Code(os.path.join(ge.MATRICES, name))
col0 = code0.col.reshape(code0.M, 4).copy()
pairs = pkg.pair_schedule(code0)
perms = {"identity": np.arange(code0.N, dtype=np.int32),
         "random": np.random.RandomState(3).permutation(code0.N).astype(np.int32),
         "euler": euler_order(col0, code0.M, code0.N),
         "first-visit": first_visit_order(col0, code0.M, code0.N, pairs)}
adjf = lambda c: float((np.diff(np.sort(c, axis=1), axis=1) == 1).mean())
rs = np.random.RandomState(1)
sigma = float(np.sqrt(1.0 / (2 * float(code0.rate) * 10 ** (ebn / 10))))
noisy = (1.0 + sigma * rs.randn(B, code0.N, code0.logGF)).astype(np.float32)
for rnd in range(2):
    for tag, pi in perms.items():
        code = pkg.Code(os.path.join(ge.MATRICES, name))
        code.col[:] = pi[col0.reshape(-1)]
        dec = pkg.Decoder(code, nm, nop, 0.3, itmax, max_batch=B, early_exit=False)
        dec.llr_init(noisy, sigma); dec.decode(B); dec.synchronize()
        ms = []
        for _ in range(3):
            dec.decode(B); dec.synchronize(); ms.append(dec.last_decode_ms)
        print("%-12s adjacent-row fraction %.3f  decode %.1f ms" % (tag, adjf(code.col.reshape(code.M, 4)), float(np.mean(ms))), flush=True)
        dec.close()
