"""`python bench.py --gpus N` from a plain shell (no torch.distributed.run around it): the parent starts N fresh ranks of itself,
relays rank 0's line and fails loudly.  Rehearsed on the CPU with EMS_BENCH_FAKE_DECODE=1 (gloo, a stand-in that reports the golden
counters of each rank's frame range): one JSON line with n_gpus = 2; a rank that dies, hangs or disagrees with the golden makes the
parent exit non-zero within its deadline, with no rank left behind."""
import json
import os
import subprocess
import sys
import time

import pytest

from conftest import ROOT

BENCH = os.path.join(ROOT, "bench.py")


def _run(n, extra_env, timeout=240, launch_timeout=None, under_torchrun=False):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(EMS_BENCH_FAKE_DECODE="1", **extra_env)
    if launch_timeout is not None:
        env["EMS_BENCH_LAUNCH_TIMEOUT"] = str(launch_timeout)
    cmd = [sys.executable, BENCH, "--gpus", str(n), "--steps", "3", "--warmup", "1"]
    if under_torchrun:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
               "--master-port", "29611", BENCH, "--gpus", str(n), "--steps", "3", "--warmup", "1"]
    t0 = time.time()
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)
    return p.returncode, p.stdout.decode(), p.stderr.decode(), time.time() - t0


def _no_ranks_left():
    out = subprocess.run(["ps", "-eo", "pid,args"], stdout=subprocess.PIPE).stdout.decode()
    return [l for l in out.splitlines() if "bench.py --gpus" in l and "pytest" not in l]


@pytest.mark.parametrize("n", [2, 3])
def test_plain_shell_launch_prints_one_line(n):
    rc, out, err, _ = _run(n, {})
    assert rc == 0, err[-2000:]
    lines = [l for l in out.splitlines() if l.strip()]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == n and j["n_ranks_seen"] == n and j["golden_ranks_checked"] == n and j["counters_match_golden"] is True
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "bench_counters.json")))["gf256"]["ranks"]
    want = [sum(gold[str(r)][i] for r in range(n)) for i in range(5)]          # the all-reduced counters of NB_LDPC.c:491-497
    assert j["config"]["frames_errframes_biterrs_undetected_sumiter"] == want
    assert j["value"] == 0.0 and "not a measurement" in j["metric"]             # a rehearsal never looks like a result


def test_single_rank_needs_no_launcher():
    rc, out, err, _ = _run(1, {})
    assert rc == 0, err[-2000:]
    assert json.loads(out.strip())["n_gpus"] == 1


def test_same_path_under_torchrun():
    rc, out, err, _ = _run(2, {}, under_torchrun=True)
    assert rc == 0, err[-2000:]
    j = json.loads([l for l in out.splitlines() if l.startswith("{")][0])
    assert j["n_gpus"] == 2 and j["n_ranks_seen"] == 2


def test_a_rank_that_dies_fails_the_run():
    rc, out, err, dt = _run(2, {"EMS_BENCH_FAKE_DIE_RANK": "1"}, launch_timeout=60)
    assert rc != 0 and not out.strip()
    assert "rank 1 exited with code 3" in err and dt < 50
    assert not _no_ranks_left()


def test_a_rank_that_hangs_hits_the_deadline():
    rc, out, err, dt = _run(2, {"EMS_BENCH_FAKE_HANG_RANK": "1"}, launch_timeout=8)
    assert rc != 0 and not out.strip()
    assert "EMS_BENCH_LAUNCH_TIMEOUT" in err and dt < 40
    assert not _no_ranks_left()


def test_a_rank_off_the_golden_fails_the_run():
    rc, out, err, _ = _run(2, {"EMS_BENCH_FAKE_CORRUPT_RANK": "1"}, launch_timeout=60)
    assert rc != 0 and not out.strip()
    assert "refusing to report" in err
