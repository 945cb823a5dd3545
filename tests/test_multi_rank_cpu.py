"""N>1 path on CPU: two gloo ranks shard the frames, all-reduce the counters and rebuild the reference's
frame-ordered progress trace; it must equal the single-process trace and the reference's own K1/K2."""
import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import MAT, ROOT

WORKER = os.path.join(ROOT, "tests", "dist_worker.py")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(world, args):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), WORLD_SIZE=str(world), OMP_NUM_THREADS="1")
    procs = [subprocess.Popen([sys.executable, WORKER] + [str(a) for a in args], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE) for r in range(world)]
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se.decode()[-2000:]
    line = [l for l in outs[0][0].decode().splitlines() if l.startswith("TRACE ")][0]
    return json.loads(line[6:])


@pytest.mark.parametrize("name,batch", [("K1", 4), ("K2", 16)])
def test_two_ranks_reproduce_reference_trace(kat, name, batch):
    k = kat[name]
    args = [os.path.join(MAT, "KN_N96_K48_GF64.txt"), k["frames"], k["NbIterMax"], k["EbN"], k["nm"], k["offset"], k["NbOper"], batch, 40]
    two = _run(2, args)
    assert [list(t) for t in two["trace"]] == [list(t) for t in k["trace"]]
    assert two["rounds"] == -(-k["frames"] // (2 * batch))
    for inflight in (1, 2):                                  # rounds launched ahead of their tally (default 3) change nothing
        assert _run(2, args + [inflight])["trace"] == two["trace"]


def test_truncation_at_error_budget_matches_single_rank():
    """The '40 erroneous frames' stop (here 3) truncates at the same frame whatever the rank count."""
    args = [os.path.join(MAT, "KN_N96_K48_GF64.txt"), 200, 10, 1.0, 16, 0.3, 32, 8, 3]
    one = _run(1, args)
    two = _run(2, args)
    assert one["trace"] == two["trace"] and one["nb"] == two["nb"] < 200
    assert one["trace"][-1][0] == 3
    assert _run(2, args + [1])["trace"] == two["trace"]     # a round launched ahead of the stop is dropped
