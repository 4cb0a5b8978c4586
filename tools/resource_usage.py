"""Parse `hipcc -Rpass-analysis=kernel-resource-usage` remarks into one line per kernel
(VGPRs, AGPRs, SGPRs, scratch bytes per lane, waves per SIMD, static LDS).  Usage: resource_usage.py <remarks.txt> [filter]"""
import re
import sys

txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
seen = set()
for b in re.split(r"remark: [^\n]*Function Name: ", txt)[1:]:
    name = b.split("\n")[0].split(" [-Rpass")[0].strip()
    if name in seen or flt not in name:
        continue
    seen.add(name)

    def g(k):
        m = re.search(k + r": (\S+)", b)
        return m.group(1) if m else "?"

    print("%-100s vgpr %3s agpr %2s sgpr %3s scratch %4s occ %s lds %s" % (
        name[:100], g("VGPRs"), g("AGPRs"), g("SGPRs"), g(r"ScratchSize \[bytes/lane\]"),
        g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]")))
