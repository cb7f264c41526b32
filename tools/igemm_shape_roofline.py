"""Per-shape roofline of md_igemm over the tuned shape table (GPU box only): every (M, N, K, ksize, stride, ups) of
magicdance_amd/csrc/igemm_tuned.inc with the (config, split) the launcher picks, cold weights (>= 320 MB of weight copies in
rotation, as in a real step where every layer streams its own weights), timed from a captured HIP graph of dependent launches.
usage: python tools/igemm_shape_roofline.py > gpurun_out/igemm_shape_roofline.txt"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from magicdance_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
F16 = torch.float16
rows = []
for line in open(os.path.join(ROOT, "magicdance_amd", "csrc", "igemm_tuned.inc")):
    m = re.match(r"\s*\{(\d+),\s*(\d+),\s*(\d+),\s*(\d+),\s*(\d+),\s*(\d+),\s*(\d+),\s*(\d+)\}", line)
    if m:
        rows.append(tuple(int(v) for v in m.groups()))
ws = torch.zeros(512 << 20, dtype=torch.uint8, device=dev)
side = torch.cuda.Stream()
REPS = 12
out = []
for (M, N, K, ks, st, up, cfg, split) in rows:
    cin = K // (ks * ks)
    if ks == 1:
        B, h, w = 1, 1, M
        ho, wo = h, w
    else:
        # output spatial side: the largest of 64 / 32 / 16 / 8 whose square divides M (batch = M / side^2)
        side_o = next((s_ for s_ in (64, 32, 16, 8, 4) if M % (s_ * s_) == 0 and M // (s_ * s_) <= 64), None)
        if side_o is None:
            continue
        B = M // (side_o * side_o)
        ho = wo = side_o
        h = w = side_o // 2 if up else (side_o * 2 if st == 2 else side_o)
    if cin % 8:
        continue
    x = torch.randn(B, h * w, cin, device=dev).to(F16)
    ncopy = max(2, min(64, (320 << 20) // (N * K * 2) + 1))
    wts = [(torch.randn(N, K, device=dev) * 0.02).to(F16) for _ in range(ncopy)]
    bias = torch.randn(N, device=dev)
    y = torch.empty(B, ho * wo, N, dtype=F16, device=dev)

    def run(i):
        ops.igemm(x, wts[i % ncopy], N, batch=B, hin=h, win=w, hout=ho, wout=wo, c0=cin, ksize=ks, stride=st, ups=up, bias=bias,
                  out=y, ws=ws)
    try:
        with torch.cuda.stream(side):
            run(0)
            side.synchronize()
            g = ops.Graph()
            g.begin()
            for i in range(REPS):
                run(i)
            g.end()
            g.launch()
            side.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(side)
            g.launch()
            e1.record(side)
            side.synchronize()
            us = e0.elapsed_time(e1) / REPS * 1e3
            g.destroy()
    except Exception as ex:  # noqa: BLE001
        print("# ERR", M, N, K, ks, ex, flush=True)
        continue
    flops = 2.0 * M * N * K
    byts = 2.0 * (M * cin * (1 if ks == 1 else 1) + N * K + M * N)
    out.append((flops / us / 1e6, us, M, N, K, ks, st, up, cfg, split, byts / us / 1e3))
    del wts
tot_f = sum(r[0] * r[1] for r in out)
tot_t = sum(r[1] for r in out)
print(f"# md_igemm per-shape roofline, {len(out)} tuned shapes (one launch each, cold weights): aggregate {tot_f / tot_t:.0f} TFLOP/s = "
      f"{tot_f / tot_t / 25:.1f} % of 2500")
print("# TFLOP/s  %peak     us        M      N      K ks st up cfg split   GB/s(algorithmic)")
for r in sorted(out, key=lambda r: -r[1]):
    print(f"{r[0]:8.0f} {r[0] / 25:6.1f} {r[1]:8.1f} {r[2]:8d} {r[3]:6d} {r[4]:6d} {r[5]:2d} {r[6]:2d} {r[7]:2d} {r[8]:3d} {r[9]:5d} {r[10]:8.0f}")
