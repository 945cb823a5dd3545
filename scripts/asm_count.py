"""Developer probe: static instruction mix of line ranges of a gfx950 assembly listing."""
import sys, re
lines = open(sys.argv[1]).read().split("\n")
for rng in sys.argv[2:]:
    a, b = map(int, rng.split("-"))
    c = {"valu": 0, "salu": 0, "lds": 0, "vmem": 0, "nop": 0, "wait": 0, "br": 0}
    for l in lines[a - 1:b]:
        t = l.strip().split(" ")[0] if l.strip() else ""
        if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"): continue
        if t.startswith("s_nop"): c["nop"] += 1
        elif t.startswith("s_waitcnt"): c["wait"] += 1
        elif t.startswith("s_cbranch") or t.startswith("s_branch"): c["br"] += 1
        elif t.startswith("v_"): c["valu"] += 1
        elif t.startswith("s_"): c["salu"] += 1
        elif t.startswith("ds_"): c["lds"] += 1
        elif t.startswith("global_") or t.startswith("buffer_") or t.startswith("flat_"): c["vmem"] += 1
    print(rng, c)
