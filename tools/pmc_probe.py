"""One md_igemm launch with a forced (config, k-groups), for running under `rocprofv3 --pmc ...` (GPU box only):
which kernel variants does the counter tool survive?   python tools/pmc_probe.py <cfg> <kg> [ksize]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from magicdance_amd import ops  # noqa: E402

cfg, kg = int(sys.argv[1]), int(sys.argv[2])
ks = int(sys.argv[3]) if len(sys.argv) > 3 else 1
dev = torch.device("cuda:0")
b, h, c, n = 2, 32, 640, 640
x = torch.randn(b, h * h, c, device=dev).half()
w = (torch.randn(n, ks * ks * c, device=dev) * 0.02).half()
y = torch.empty(b, h * h, n, dtype=torch.float16, device=dev)
ws = torch.zeros(64 << 20, dtype=torch.uint8, device=dev)
for _ in range(3):
    ops.igemm(x, w, n, batch=b, hin=h, win=h, hout=h, wout=h, c0=c, ksize=ks, out=y, ws=ws, force_cfg=cfg, force_kg=kg)
torch.cuda.synchronize()
print("ok", cfg, kg, float(y.float().abs().mean()))
