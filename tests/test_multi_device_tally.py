"""The C harness' multi-GPU Monte-Carlo loop (csrc/harness/ems_multi.h + ems_tally.h; SURVEY.md 8(e), north_star: "sharded across
the 8 GPUs of one node with only an RCCL all-reduce of frame/bit-error counters").

CPU part: tests/ctest/test_tally.c runs the very loop ems_sim runs -- one thread per device, two batches in flight per device, one
all-reduce of the five counters per round, ordered tally -- with a fake N-device decode and a barrier-based all-reduce, under
AddressSanitizer + UBSan, and compares the printed trace with the sequential single-device tally: same line after every frame,
truncated at exactly the 40th erroneous frame, the same all-reduced totals on every device, batches launched ahead of the stop
drained and dropped.  GPU part (-m gpu): `EMS_DEVICES=0 ems_sim ...` -- the same code with the real RCCL communicator
(ncclCommInitAll + ncclAllReduce of 5 x ncclInt64) on the one GPU of the box -- reproduces the reference's traces."""
import json
import os
import re
import subprocess

import pytest

from conftest import GOLD, MAT, ROOT

HARNESS = os.path.join(ROOT, "ems-decoder-of-nb-ldpc-codes_amd", "csrc", "harness")


@pytest.fixture(scope="module")
def tally_exe(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("ctest") / "test_tally")
    cmd = ["gcc", "-O1", "-g", "-Wall", "-Wextra", "-Werror", "-pthread", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-I" + HARNESS, "-o", exe,
           os.path.join(ROOT, "tests", "ctest", "test_tally.c")]
    out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert out.returncode == 0, out.stdout.decode()[-3000:]
    return exe


@pytest.mark.parametrize("ndev,batch,frames,rate", [(1, 7, 500, 30), (2, 64, 5000, 30), (8, 16, 100000, 3), (3, 5, 200, 0), (8, 1536, 20000, 4),
                                                     (5, 1, 300, 200), (4, 100, 250, 1000), (8, 256, 300, 50), (7, 33, 1000, 0), (2, 1, 1, 1000)])
@pytest.mark.parametrize("lagged", [False, True])
def test_fake_n_device_run_equals_sequential_tally(tally_exe, ndev, batch, frames, rate, lagged):
    """lagged: the split all-reduce of ems_multi.h (what ems_sim gives RCCL): issued when a round is collected, waited for a round later, so
    that a collective kernel queued behind the next round's persistent decode workgroups never stalls the host -- the ctest logs every
    callback and asserts launch(r+1) < collect(r) < end(r-1) < begin(r) on every device; the trace must not change."""
    env = dict(os.environ, EMS_TALLY_LAGGED="1") if lagged else {k: v for k, v in os.environ.items() if k != "EMS_TALLY_LAGGED"}
    out = subprocess.run([tally_exe, str(ndev), str(batch), str(frames), str(rate)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300, env=env)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    lines = out.stdout.decode().splitlines()
    assert len(lines) <= frames
    if lines:
        last = re.match(r"<(\d+)> FER= (\d+) / (\d+) ", lines[-1])
        assert int(last.group(2)) <= 40 and (int(last.group(2)) == 40 or int(last.group(3)) == frames)


@pytest.mark.parametrize("ndev,batch,frames,dev,kind,at", [(4, 8, 4000, 2, "launch", 0), (4, 8, 4000, 0, "launch", 3), (8, 16, 100000, 5, "collect", 7),
                                                            (3, 5, 2000, 2, "collect", 0), (2, 64, 5000, 1, "launch", 1), (8, 4, 100000, 7, "launch", 40)])
@pytest.mark.parametrize("lagged", [False, True])
def test_a_failing_device_ends_the_run_instead_of_hanging_it(tally_exe, ndev, batch, frames, dev, kind, at, lagged):
    """One device's launch or collect fails in some round while the others are rounds ahead or behind: every thread must take the
    same decision for that round (the failure flag is latched between the round's two barriers) and leave; a thread left waiting
    in the all-reduce or a barrier would show as the timeout here."""
    for _ in range(5):                                           # thread interleavings differ from run to run
        env = dict(os.environ, EMS_TALLY_LAGGED="1") if lagged else {k: v for k, v in os.environ.items() if k != "EMS_TALLY_LAGGED"}
        out = subprocess.run([tally_exe, str(ndev), str(batch), str(frames), "3", str(dev), kind, str(at)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=60, env=env)
        assert out.returncode == 4, (out.returncode, out.stderr.decode()[-2000:])


def _trace(stdout):
    return [[int(m.group(1)), int(m.group(3)), m.group(4)] for m in
            re.finditer(r"<\d+> FER= (\d+) / (\d+) = [\d.]+ BER= (\d+) / x = [\d.]+\s+avr_it=([\d.]+)", stdout.decode("latin-1"))]


@pytest.mark.gpu
@pytest.mark.parametrize("case,batch", [("K2", 16), ("K3", 2), ("L4", 256)])
def test_c_harness_multi_device_path_with_rccl_on_one_gpu(pkg, kat, tmp_path, case, batch):
    """EMS_DEVICES=0: ncclCommInitAll over one device, the per-round ncclAllReduce, the ordered tally -- against the traces of the
    real reference executable (K2 / K3: tests/golden/kat_traces.json; L4: 969 full-size frames to the 40th erroneous one)."""
    k = kat[case] if case in kat else json.load(open(os.path.join(GOLD, "kat_long.json")))[case]
    matrix = ("KN_" + k["matrix"]) if k.get("fmt") == "kn" and not k["matrix"].startswith("KN_") else k["matrix"]
    os.makedirs(os.path.join(str(tmp_path), "data"), exist_ok=True)
    out = subprocess.run([pkg.SIM_PATH] + [str(a) for a in (k["frames"], k["NbIterMax"], os.path.join(MAT, matrix), k["EbN"], k["nm"], k["offset"], k["NbOper"])],
                         cwd=str(tmp_path), env=dict(os.environ, EMS_DEVICES="0", EMS_BATCH=str(batch)), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    assert b"RCCL all-reduce" in out.stderr
    assert _trace(out.stdout) == [list(t) for t in k["trace"]]


@pytest.mark.gpu
def test_c_harness_multi_device_path_with_the_device_side_source(pkg, tmp_path):
    """EMS_GPU_SOURCE (frames generated in HBM from the counter-based stream, indexed by frame) through the multi-device loop: the
    frames do not depend on which device or batch draws them, so `EMS_DEVICES=0` must print the trace of the single-device harness."""
    args = [str(a) for a in (600, 10, os.path.join(MAT, "KN_N96_K48_GF64.txt"), 2.5, 16, 0.3, 32)]
    os.makedirs(os.path.join(str(tmp_path), "data"), exist_ok=True)
    tr = []
    for env in (dict(EMS_BATCH="64"), dict(EMS_BATCH="48", EMS_DEVICES="0")):
        out = subprocess.run([pkg.SIM_PATH] + args, cwd=str(tmp_path), env=dict(os.environ, EMS_GPU_SOURCE="7", **env), stdout=subprocess.PIPE,
                             stderr=subprocess.PIPE, timeout=600)
        assert out.returncode == 0, out.stderr.decode()[-2000:]
        tr.append(_trace(out.stdout))
    assert len(tr[0]) > 100 and tr[0] == tr[1]
