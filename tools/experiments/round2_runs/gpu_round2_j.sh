#!/bin/bash
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "igemm" 2>&1 | tail -8 > gpurun_out/r2j_igemm_tests.log
timeout 300 python - > gpurun_out/r2j_m32_probe.txt 2>&1 <<'PY'
import sys, os, torch
sys.path.insert(0, os.getcwd())
from magicdance_amd import ops
dev = torch.device("cuda:0"); F16 = torch.float16; side = torch.cuda.Stream()
ws = torch.zeros(512 << 20, dtype=torch.uint8, device=dev)
SH = [(16, 64, 64, 320, 320, 3), (16, 64, 64, 640, 320, 3), (16, 32, 32, 640, 640, 3), (16, 16, 16, 1280, 1280, 3), (16, 32, 32, 1920, 640, 3),
      (2, 64, 64, 320, 320, 3), (2, 32, 32, 640, 640, 3), (16, 1, 4096, 1280, 320, 1), (16, 1, 1024, 2560, 640, 1), (16, 1, 256, 5120, 1280, 1)]
for (b, h, w, cin, cout, k) in SH:
    x = torch.randn(b, h * w, cin, device=dev).to(F16)
    M, K = b * h * w, k * k * cin
    ncopy = max(2, min(12, (320 << 20) // (cout * K * 2) + 1))
    wts = [(torch.randn(cout, K, device=dev) * 0.02).to(F16) for _ in range(ncopy)]
    bias = torch.randn(cout, device=dev); out = torch.empty(b, h * w, cout, dtype=F16, device=dev)
    res = []
    for cfg in (-1, 12, 25, 34, 35, 36, 37):
        if cfg in (25, 35, 37) and cout % 160: continue
        for sp in ((1,) if M > 4096 else (1, 2, 4)):
            def run(i): ops.igemm(x, wts[i % ncopy], cout, batch=b, hin=h, win=w, hout=h, wout=w, c0=cin, ksize=k, bias=bias, out=out, ws=ws, force_cfg=cfg, force_splitk=(0 if cfg < 0 else sp))
            with torch.cuda.stream(side):
                run(0); side.synchronize()
                g = ops.Graph(); g.begin()
                for i in range(12): run(i)
                g.end(); g.launch(); side.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(side); g.launch(); e1.record(side); side.synchronize()
                us = e0.elapsed_time(e1) / 12 * 1e3; g.destroy()
            res.append(f"c{cfg}/s{sp}:{us:.1f}us({2.0 * M * cout * K / us / 1e6:.0f}TF)")
            if cfg < 0: break
    print(f"M={M} N={cout} K={K} ks={k}: " + "  ".join(res), flush=True)
PY
cat gpurun_out/r2j_igemm_tests.log | tail -4; cat gpurun_out/r2j_m32_probe.txt
