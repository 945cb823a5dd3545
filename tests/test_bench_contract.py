"""bench.py's arithmetic against SURVEY.md 8(d) (no GPU): the algorithmic bytes the roofline figure is computed from,
the workload table against BASELINE.json's configs, and the command-line contract."""
import json
import os
import subprocess
import sys

from conftest import ROOT

sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_algorithmic_bytes_match_the_survey_table():
    # cfg 4 36 547 200 B per sweep, cfg 3 6 998 400 B, cfg 2 21 760 B
    assert bench.algorithmic_bytes_per_frame_decode(16200, 256, 20, 1) == 36_547_200
    assert bench.algorithmic_bytes_per_frame_decode(32400, 16, 8, 1) == 6_998_400
    assert bench.algorithmic_bytes_per_frame_decode(32, 64, 16, 1) == 21_760
    per_frame = bench.algorithmic_bytes_per_frame_decode(16200, 256, 20, 29)       # decode kernel only (LLR init is its own kernel)
    assert per_frame == 29 * 36_547_200
    with_init = per_frame + 8100 * 8 * 4 + 8100 * 256 * 4 + 8100
    assert abs(with_init / 1e6 - 1068.4) < 0.1                                      # the survey's per-frame figure
    assert abs(8000e9 / with_init - 7490) < 10                                      # its ceiling at 8 TB/s
    # v2 flavour: DESIGN.md 4.4
    assert bench.algorithmic_bytes_per_frame_decode_v2(8100, 16200, 256, 20, 1) == 16200 * (1024 + 300) + 8100 * 101


def test_workloads_cover_the_baseline_configs():
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert "frames" in base["metric"] or "frames" in base.get("unit", "")
    w = bench.WORKLOADS
    assert w["gf256"][0] == "Ahmed_64800_R12_GF256" and w["gf256"][1] == 20 and w["gf256"][4] == 30      # the headline config
    assert w["gf16"][0] == "Ahmed_64800_R34_GF16" and w["n96"][0] == "N96_K48_GF64"
    assert bench.HBM_PEAK_GBS == 8000.0


def test_command_line_contract():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert out.returncode == 0
    txt = out.stdout.decode()
    for flag in ("--gpus", "--steps", "--warmup"):
        assert flag in txt
