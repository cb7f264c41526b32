"""Run md_ff_block (and the md_igemm launches it replaces) a few times in isolation, for rocprofv3 --pmc; GPU box only."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
from magicdance_amd import ops  # noqa: E402
import test_gpu_ffblock as T  # noqa: E402
dev = torch.device("cuda:0")
F16 = torch.float16
for c, b, n, bm in ((320, 16, 4096, 128), (320, 16, 4096, 64), (320, 2, 4096, 32), (320, 3, 4096, 64)):
    m = b * n
    pk = T.make_params(c, 20, dev)["packed"]
    x16 = (T._rand((m, c), 1, dev) + 0.3).to(F16)
    lo16 = T._rand((m, c), 3, dev, 1e-3).to(F16)
    att16 = T._rand((m, c), 2, dev).to(F16)
    out, out_lo = torch.empty((m, c), dtype=F16, device=dev), torch.empty((m, c), dtype=F16, device=dev)
    for _ in range(3):
        ops.ff_block(x16, out, m=m, c=c, w1=pk["w1"], s1=pk["s1"], s0=pk["s0"], w2=pk["w2"], b2=pk["b2"], x_lo=lo16, out_lo=out_lo,
                     attn=att16, wo=pk["wo"], bo=pk["bo"], force_bm=bm)
    torch.cuda.synchronize()
