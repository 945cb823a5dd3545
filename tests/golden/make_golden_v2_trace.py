"""Generates tests/golden/v2_traces.json: every statistics line the REAL older-generation executable prints
(oracle/_ref/essai_v2 = EMS_HS_L-BubbleCheck_UBS_decoder_v2.c compiled where it lies by oracle/Makefile, its own main():
RandomBinaryGenerator + Encoding + ModelChannelAWGN draws from libc's drand48, decode, the BER samples at iterations
9/19/29/39/49 and the "> 12 bit errors" frame rule, v2.c:1980-2311) for a few operating points of the headline code, so that the
GPU harness csrc/harness/ems_sim_v2.c can be held to it line for line.

    python tests/golden/make_golden_v2_trace.py [case ...]        (needs /root/reference; run from the repo root; minutes per case)
"""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
EXE = os.path.join(ROOT, "oracle", "_ref", "essai_v2")
CASES = {   # across the waterfall of the Rayleigh-faded 256-point channel (8 dB: nothing converges; 9 dB: everything by iteration ~15)
    "V1": dict(matrix="Ahmed_64800_R12_GF256", frames=24, NbIterMax=50, EbN=8.7, nm=20, offset=0.3, NbOper=40),
    "V2": dict(matrix="Ahmed_64800_R12_GF256", frames=16, NbIterMax=30, EbN=9.0, nm=20, offset=0.3, NbOper=40),
    "V3": dict(matrix="Ahmed_64800_R12_GF256", frames=10, NbIterMax=22, EbN=8.5, nm=20, offset=1.0, NbOper=25),
}

if __name__ == "__main__":
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), os.path.join(ROOT, "oracle", "_ref", "essai_v2")])
    dst = os.path.join(ROOT, "tests", "golden", "v2_traces.json")
    out = json.load(open(dst)) if os.path.exists(dst) else {}
    for name, c in CASES.items():
        if sys.argv[1:] and name not in sys.argv[1:]:
            continue
        path = os.path.join(ROOT, "tests", "golden", "matrices", c["matrix"])
        txt = subprocess.run([EXE, str(c["frames"]), str(c["NbIterMax"]), path, str(c["EbN"]), str(c["nm"]), str(c["offset"]), str(c["NbOper"])],
                             stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=6 * 3600).stdout.decode("latin-1")
        lines = re.findall(r"<\d+> FER= \d+/ \d+ = \S+ BER= \d+/ \d+ = \S+", txt)
        assert len(lines) % 5 == 0 and lines
        out[name] = dict(c, lines=lines)
        print(name, len(lines) // 5, "frames; last:", lines[-5:], flush=True)
        json.dump(out, open(dst, "w"), indent=0)
