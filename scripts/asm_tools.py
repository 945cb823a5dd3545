"""Developer probe: static views of a gfx950 assembly listing of the kernels
   (hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off --cuda-device-only -S [-DEMS_MARKS] -o k.s csrc/ems_kernels.hip).
usage: python scripts/asm_tools.py regs  k.s [name filter]            VGPR / SGPR / scratch of every kernel
       python scripts/asm_tools.py marks k.s <mangled-name substring>  instruction mix between the `; Q16T_MARK <name>` comments (-DEMS_MARKS)
       python scripts/asm_tools.py count k.s <a>-<b> [...]             instruction mix of line ranges
       python scripts/asm_tools.py loops k.s <mangled-name substring> [context]   the backward branches of one kernel"""
import re, subprocess, sys

KINDS = ("valu", "salu", "lds", "vmem", "nop", "wait", "br")


def kind(line):
    t = line.strip().split(" ")[0] if line.strip() else ""
    if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"): return None
    if t.startswith("s_nop"): return "nop"
    if t.startswith("s_waitcnt"): return "wait"
    if t.startswith("s_cbranch") or t.startswith("s_branch"): return "br"
    if t.startswith("v_"): return "valu"
    if t.startswith("s_"): return "salu"
    if t.startswith("ds_"): return "lds"
    if t.startswith(("global_", "buffer_", "flat_", "scratch_")): return "vmem"
    return None


def mix(lines):
    c = dict.fromkeys(KINDS, 0)
    for l in lines:
        k = kind(l)
        if k: c[k] += 1
    return c


def function(s, key):
    i0 = next(i for i, l in enumerate(s) if l.startswith("_ZN") and key in l and ":" in l)
    return i0, next(i for i in range(i0, len(s)) if s[i].strip().startswith("s_endpgm"))


cmd, path = sys.argv[1], sys.argv[2]
text = open(path).read()
s = text.split("\n")
if cmd == "regs":
    flt = sys.argv[3] if len(sys.argv) > 3 else "k_decode"
    for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", text, re.S):
        g = lambda k: (re.search(r"\.amdhsa_" + k + r" (\d+)", m.group(2)) or [None, None])[1]
        dn = subprocess.run(["c++filt", m.group(1)], stdout=subprocess.PIPE).stdout.decode().strip()
        if flt in dn:
            print("%-60s vgpr %s sgpr %s scratch %s" % (dn.split("(anonymous namespace)::")[-1][:60], g("next_free_vgpr"), g("next_free_sgpr"), g("private_segment_fixed_size")))
elif cmd == "marks":
    i0, i1 = function(s, sys.argv[3])
    cur, acc, order = "start", {"start": []}, ["start"]
    for l in s[i0:i1]:
        m = re.search(r"; Q16T_MARK (\w+)", l)
        if m:
            cur = m.group(1) + "@%d" % len(order); order.append(cur); acc[cur] = []
        else:
            acc[cur].append(l)
    for k in order: print("%-26s %s" % (k, mix(acc[k])))
elif cmd == "count":
    for rng in sys.argv[3:]:
        a, b = map(int, rng.split("-"))
        print(rng, mix(s[a - 1:b]))
elif cmd == "loops":
    i0, i1 = function(s, sys.argv[3])
    ctx = int(sys.argv[4]) if len(sys.argv) > 4 else 6
    labels = {m.group(1): i for i in range(i0, i1) for m in [re.match(r"^(\.LBB\d+_\d+):", s[i])] if m}
    for i in range(i0, i1):
        m = re.search(r"(s_cbranch_\w+|s_branch)\s+(\.LBB\d+_\d+)", s[i])
        if m and m.group(2) in labels and labels[m.group(2)] < i:
            a = labels[m.group(2)]
            print("==== loop %s: lines %d..%d (%d lines), closes with %s" % (m.group(2), a - i0, i - i0, i - a, m.group(1)))
            for l in s[max(a, i - ctx):i + 1]: print("     ", l.strip())
