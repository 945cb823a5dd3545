"""Developer probe: VGPR / SGPR / scratch / LDS of every decode kernel in a gfx950 assembly listing
(hipcc ... --cuda-device-only -S -o k.s ems_kernels.hip).  usage: python scripts/kernel_regs.py k.s [filter]"""
import re, subprocess, sys
s = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else "k_decode"
for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", s, re.S):
    name, body = m.group(1), m.group(2)
    g = lambda k: (re.search(r"\.amdhsa_" + k + r" (\d+)", body) or [None, None])[1]
    dn = subprocess.run(["c++filt", name], stdout=subprocess.PIPE).stdout.decode().strip()
    if flt in dn:
        print("%-60s vgpr %s sgpr %s scratch %s" % (dn.split("(anonymous namespace)::")[-1][:60], g("next_free_vgpr"), g("next_free_sgpr"), g("private_segment_fixed_size")))
