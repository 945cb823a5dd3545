"""Developer probe: instruction mix between the `; Q16T_MARK <name>` comments of one kernel (build with -DQ16T_MARKS).
usage: python scripts/asm_marks.py k.s <mangled-name substring>"""
import re, sys
s = open(sys.argv[1]).read().split("\n")
key = sys.argv[2]
i0 = next(i for i, l in enumerate(s) if l.startswith("_ZN") and key in l and l.rstrip().endswith(":") or (l.startswith("_ZN") and key in l and "@" in l))
i1 = next(i for i in range(i0, len(s)) if s[i].strip().startswith("s_endpgm"))
cur, acc, order = "start", {}, []
for l in s[i0:i1]:
    m = re.search(r"; Q16T_MARK (\w+)", l)
    if m:
        cur = m.group(1) + "@%d" % len(order)
        order.append(cur); acc[cur] = {"valu": 0, "salu": 0, "lds": 0, "vmem": 0, "nop": 0, "wait": 0}
        continue
    t = l.strip().split(" ")[0] if l.strip() else ""
    if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"): continue
    c = acc.setdefault(cur, {"valu": 0, "salu": 0, "lds": 0, "vmem": 0, "nop": 0, "wait": 0})
    if cur not in order: order.append(cur)
    if t.startswith("s_nop"): c["nop"] += 1
    elif t.startswith("s_waitcnt"): c["wait"] += 1
    elif t.startswith("v_"): c["valu"] += 1
    elif t.startswith("s_"): c["salu"] += 1
    elif t.startswith("ds_"): c["lds"] += 1
    elif t.startswith("global_") or t.startswith("buffer_") or t.startswith("flat_"): c["vmem"] += 1
for k in order: print("%-22s %s" % (k, acc[k]))
