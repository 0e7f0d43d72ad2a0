"""CPU only: table of the kernels' register / scratch / LDS use from `hipcc -Rpass-analysis=kernel-resource-usage` logs.
usage: for v in mi_degensac mi_degensac_t256 mi_degensac_t128; do hipcc <FLAGS of the Makefile> -Rpass-analysis=kernel-resource-usage -c $v.hip -o /tmp/$v.o 2> /tmp/ru_$v.txt; done
       python tools/resource_usage.py /tmp/ru_mi_degensac.txt /tmp/ru_mi_degensac_t256.txt /tmp/ru_mi_degensac_t128.txt > profiles/r6_resource_usage.txt"""
import re, subprocess, sys
print("# hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Rpass-analysis=kernel-resource-usage on the three translation units")
print("# kernel<threads, placement>: placement 1 = points + pool in LDS, 2 = pool in LDS, 0 = HBM workspace")
print("%-48s %5s %7s %7s %8s %4s %8s" % ("kernel", "VGPR", "spillV", "spillS", "scratchB", "occ", "LDS B"))
for f in sys.argv[1:]:
    txt = open(f).read()
    for b in re.split(r"remark: Function Name: ", txt)[1:]:
        name = subprocess.run(["c++filt", b.split()[0]], capture_output=True, text=True).stdout.strip().split("(")[0].replace("void ", "")
        g = lambda k: (re.search(k + r": (\S+)", b) or [None, "?"])[1]
        print("%-48s %5s %7s %7s %8s %4s %8s" % (name, g("VGPRs"), g("VGPRs Spill"), g("SGPRs Spill"), g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]")))
