"""Generates tests/golden/kat_long.json: the trace the REAL reference executable (oracle/_ref/essai_ubs, built from
/root/reference by oracle/Makefile) prints for a complete operating-point run of the headline code -- every frame
up to the 40th erroneous one (NB_LDPC.c:506) at 2.0 dB -- so that the GPU harness can be held to the reference
over several hundred full-size frames, not a handful.  Takes the reference about half an hour on one core.

    python tests/golden/make_kat_long.py            (needs /root/reference; run from the repo root)
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import orc  # noqa: E402

CASES = {
    "L4": dict(matrix="Ahmed_64800_R12_GF256", fmt="ubs", frames=2000, NbIterMax=30, EbN=2.0, nm=20, offset=0.3, NbOper=40),
    "L3a": dict(matrix="Ahmed_64800_R34_GF16", fmt="ubs", frames=400, NbIterMax=30, EbN=3.0, nm=8, offset=0.3, NbOper=16),
    "L3b": dict(matrix="Ahmed_64800_R34_GF16", fmt="ubs", frames=150, NbIterMax=30, EbN=2.3, nm=8, offset=0.3, NbOper=16),
    "L4b": dict(matrix="Ahmed_64800_R12_GF256", fmt="ubs", frames=2000, NbIterMax=30, EbN=1.8, nm=20, offset=0.3, NbOper=40),
    "L4c": dict(matrix="Ahmed_64800_R12_GF256", fmt="ubs", frames=500, NbIterMax=30, EbN=2.2, nm=20, offset=0.3, NbOper=40),
    "L1": dict(matrix="KN_N96_K48_GF64.txt", fmt="kn", frames=20000, NbIterMax=10, EbN=2.5, nm=16, offset=0.3, NbOper=32),
}

if __name__ == "__main__":
    orc.build()
    dst = os.path.join(ROOT, "tests", "golden", "kat_long.json")
    out = json.load(open(dst)) if os.path.exists(dst) else {}
    only = sys.argv[1:]                                 # optional: case names (several processes may run side by side,
    for name, c in CASES.items():                       #           each writing tests/golden/kat_long.<name>.json to merge)
        if only and name not in only:
            continue
        if name in out and {k: out[name][k] for k in c} == c:
            continue                                    # already generated with these parameters
        path = os.path.join(ROOT, "tests", "golden", "matrices", c["matrix"])
        tr = orc.run_reference_binary(c["fmt"], c["frames"], c["NbIterMax"], path, c["EbN"], c["nm"], c["offset"], c["NbOper"],
                                      timeout=6 * 3600)
        out[name] = dict(c, trace=[list(t) for t in tr])
        print(name, len(tr), "frames, last", tr[-1], flush=True)
        if only:
            json.dump({name: out[name]}, open(dst.replace(".json", ".%s.json" % name), "w"))
        else:
            json.dump(out, open(dst, "w"))
