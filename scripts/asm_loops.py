"""Developer probe: the loops (backward branches) of one kernel in a gfx950 assembly listing, with the instructions that decide them.
usage: python scripts/asm_loops.py k.s <mangled-name substring> [context lines]"""
import re, sys
s = open(sys.argv[1]).read().split("\n")
key = sys.argv[2]
ctx = int(sys.argv[3]) if len(sys.argv) > 3 else 6
i0 = next(i for i, l in enumerate(s) if l.startswith("_ZN") and key in l and l.split(":")[0].endswith("EmsPlan") or (l.startswith("_ZN") and key in l and ":" in l and "@" in l))
i1 = next(i for i in range(i0, len(s)) if s[i].strip().startswith("s_endpgm"))
labels = {}
for i in range(i0, i1):
    m = re.match(r"^(\.LBB\d+_\d+):", s[i])
    if m:
        labels[m.group(1)] = i
print("function lines", i0, i1)
for i in range(i0, i1):
    m = re.search(r"(s_cbranch_\w+|s_branch)\s+(\.LBB\d+_\d+)", s[i])
    if m and m.group(2) in labels and labels[m.group(2)] < i:
        a = labels[m.group(2)]
        print("==== loop %s: lines %d..%d (%d instr lines), closes with %s" % (m.group(2), a - i0, i - i0, i - a, m.group(1)))
        for l in s[max(a, i - ctx):i + 1]:
            print("     ", l.strip())
