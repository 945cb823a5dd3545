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
    # round 5: check degrees the lane-row kernels do not hold -- the reference's rate-9/10 matrix (dc = 20; v2.c:1756-1852 loops over
    # rowDegree[node] with no cap), an irregular (2, dc) code with degrees 2..12 and the largest in row 0 (v2.c:750, :794-800 size the
    # scratch by rowDegree[0]), a small dc = 22 code (make_v2_degree_matrices.py); in the waterfall of each.  (Much above it the executable
    # writes TMP[-1]: a likelihood below -1e5 leaves likelihood_index at -1, v2.c:1489-1496.)
    "V4": dict(matrix="Ahmed_64800_R910_GF256", frames=8, NbIterMax=30, EbN=20.5, nm=20, offset=0.3, NbOper=40),
    "V5": dict(matrix="V2Irr_M24_GF256", frames=40, NbIterMax=25, EbN=14.0, nm=20, offset=0.3, NbOper=40),
    "V6": dict(matrix="V2Wide22_M10_GF256", frames=40, NbIterMax=25, EbN=20.5, nm=20, offset=0.3, NbOper=40),
}

if __name__ == "__main__":
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), os.path.join(ROOT, "oracle", "_ref", "essai_v2")])
    dst = os.path.join(ROOT, "tests", "golden", "v2_traces.json")
    out = json.load(open(dst)) if os.path.exists(dst) else {}
    for name, c in CASES.items():
        if (sys.argv[1:] and name not in sys.argv[1:]) or (not sys.argv[1:] and name in out):
            continue                                     # (no argument: only the cases the file does not hold yet)
        path = os.path.join(ROOT, "tests", "golden", "matrices", c["matrix"])
        txt = subprocess.run([EXE, str(c["frames"]), str(c["NbIterMax"]), path, str(c["EbN"]), str(c["nm"]), str(c["offset"]), str(c["NbOper"])],
                             stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=6 * 3600).stdout.decode("latin-1")
        lines = re.findall(r"<\d+> FER= \d+/ \d+ = \S+ BER= \d+/ \d+ = \S+", txt)
        assert len(lines) % 5 == 0 and lines
        out[name] = dict(c, lines=lines)
        print(name, len(lines) // 5, "frames; last:", lines[-5:], flush=True)
        json.dump(out, open(dst, "w"), indent=0)
