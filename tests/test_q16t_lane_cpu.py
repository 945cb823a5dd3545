"""The lane-serial GF(16) check-node code (csrc/kernels/ems_q16t_lane.h) on the CPU: the SAME functions k_decode_q16t runs per lane,
compiled for the host by tests/ctest/test_q16t.cpp and compared with the oracle -- sorting networks exhaustively, 400 000 elementary
steps (continuous / coarse / short lists, several NbOper), and whole 29-sweep decodes of the 64800-bit GF(16) code where every check
node goes through select_lane -> 18 x es_lane -> output_lane (a check node that reports `bad` takes the oracle's, as the kernel falls
back to the merge-form visit).  Everything bit-exact; the fallback rate must stay under 2 % of the visits (measured: 0.03-0.08 %)."""
import os
import subprocess

import pytest

from conftest import MAT, ROOT

KER = os.path.join(ROOT, "ems-decoder-of-nb-ldpc-codes_amd", "csrc", "kernels")
ORC = os.path.join(ROOT, "oracle")


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    d = tmp_path_factory.mktemp("q16t")
    exe = str(d / "test_q16t")
    for cmd in (["g++", "-O2", "-ffp-contract=off", "-std=c++17", "-Wall", "-Wno-unknown-pragmas", "-I" + KER, "-I" + ORC, "-c", os.path.join(ROOT, "tests", "ctest", "test_q16t.cpp"), "-o", str(d / "t.o")],
                ["gcc", "-O2", "-ffp-contract=off", "-c", os.path.join(ORC, "ems_oracle.c"), "-o", str(d / "o.o")],
                ["g++", str(d / "t.o"), str(d / "o.o"), "-lm", "-o", exe]):
        out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        assert out.returncode == 0, out.stdout.decode()[-3000:]
    return exe


def test_networks_and_elementary_steps(exe):
    out = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    assert out.returncode == 0, out.stdout.decode()[-2000:]
    assert b"exhaustive 0-1 check passed" in out.stdout and b"bit-exact where decided" in out.stdout


@pytest.mark.parametrize("frames,ebn,sweeps", [(1, 0.0, 29), (2, 2.3, 29), (2, 3.0, 8)])
def test_whole_decodes_equal_the_oracle(exe, frames, ebn, sweeps):
    out = subprocess.run([exe, os.path.join(MAT, "Ahmed_64800_R34_GF16"), str(frames), str(ebn), str(sweeps)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    assert out.returncode == 0, out.stdout.decode()[-2000:]
    assert out.stdout.count(b"bit-exact") >= frames + 1
