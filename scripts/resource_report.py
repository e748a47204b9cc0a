"""Per-kernel register / occupancy / LDS table from hipcc's -Rpass-analysis=kernel-resource-usage output.
Usage: hipcc ... -Rpass-analysis=kernel-resource-usage -c file.hip -o x.o > log 2>&1; python scripts/resource_report.py log"""
import re, subprocess, sys
txt = open(sys.argv[1]).read()
OCC, LDS = r"Occupancy \[waves/SIMD\]", r"LDS Size \[bytes/block\]"
for b in re.split(r"remark: [^\n]*Function Name: ", txt)[1:]:
    name = b.split()[0]
    def g(k):
        m = re.search(k + r": (\d+)", b)
        return m.group(1) if m else "?"
    dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    dn = dn.replace("(anonymous namespace)::", "").split("(")[0][:90]
    print("%-90s V=%s A=%s occ=%s spill=%s lds=%s" % (dn, g("VGPRs"), g("AGPRs"), g(OCC), g("VGPRs Spill"), g(LDS)))
