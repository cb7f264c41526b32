"""VGPR / SGPR / scratch / occupancy of every kernel in one HIP source (hipcc -Rpass-analysis=kernel-resource-usage).
usage: python tools/kernel_resources.py magicdance_amd/csrc/igemm.hip [extra hipcc flags]"""
import re
import subprocess
import sys

src = sys.argv[1]
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", src, "-o", "/dev/null",
       "-Rpass-analysis=kernel-resource-usage"] + sys.argv[2:]
txt = subprocess.run(cmd, capture_output=True, text=True).stderr
out = []
for b in re.split(r"remark: [^\n]*Function Name: ", txt)[1:]:
    name = b.split("\n")[0].strip()

    def g(k):
        m = re.search(k + r": (\d+)", b)
        return int(m.group(1)) if m else -1
    m = re.search(r"(\w+_kernel\w*|gn_\w+|\w+)I(.*?)EEv", name)
    d = name
    if m:
        d = m.group(1)[-24:] + "<" + ",".join(a[1] for a in re.findall(r"L([ib])(\d+)E", m.group(2))) + ">"
    out.append("%-64s v%4d a%4d s%4d scratch %4d occ %d" % (d[:64], g("VGPRs"), g("AGPRs"), g("SGPRs"),
                                                            g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]")))
print("\n".join(sorted(out)))
