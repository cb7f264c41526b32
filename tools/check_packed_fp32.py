"""Static check behind csrc/build.sh's packed-fp32 rule (round 5): compiles every translation unit to gfx950 assembly
with the flags build.sh uses and counts the packed-fp32 VALU instructions per kernel -- in particular the form the non-repeatable
LayerNorm fold was made of: a packed op whose LOW result half selects the HIGH register of a source pair (an `op_sel:[..1..]` bit).
The GEMM units must contain no packed fp32 at all; the others are listed.  usage: python tools/check_packed_fp32.py   (CPU only, ~2 min)"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "magicdance_amd", "csrc")
sh = open(os.path.join(CSRC, "build.sh")).read()
nopk = set(re.search(r'case "\$f" in ([a-z_|]+)\) EXTRA="\$EXTRA -Xclang -target-feature -Xclang -packed-fp32-ops"', sh).group(1).split("|"))
units = re.search(r"for f in ([a-z_ ]+); do", sh).group(1).split()
procs, bad = {}, 0
tmp = tempfile.mkdtemp()
for u in units:
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-S", "--cuda-device-only"]
    if u == "attention":
        flags += ["-mllvm", "-amdgpu-mfma-vgpr-form"]
    if u in nopk:
        flags += ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]
    procs[u] = subprocess.Popen(["/opt/rocm/bin/hipcc"] + flags + ["-c", os.path.join(CSRC, u + ".hip"), "-o", os.path.join(tmp, u + ".s")],
                                stderr=subprocess.DEVNULL)
for u, p in procs.items():
    if p.wait() != 0:
        print(f"{u}: compile failed")
        bad += 1
        continue
    name, per = None, {}
    for line in open(os.path.join(tmp, u + ".s")):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            name = m.group(1)
        elif re.search(r"\bv_pk_[a-z]+_f32\b", line):
            c = per.setdefault(name, [0, 0])
            c[0] += 1
            c[1] += bool(re.search(r"op_sel:\[[01,]*1", line))
    tot, cross = sum(c[0] for c in per.values()), sum(c[1] for c in per.values())
    print(f"{u}: {tot} packed-fp32 instructions, {cross} of them with a low half that reads a HIGH register"
          + (" (built without packed fp32)" if u in nopk else ""))
    for k, c in per.items():
        if c[1]:
            print(f"    {c[1]:3d}  {k}")
    if u in nopk and tot:
        bad += 1
sys.exit(1 if bad else 0)
