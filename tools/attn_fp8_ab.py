"""fp8 attention (BASELINE configs[4]): the pipelined loop on e4m3 operands with the piecewise-linear exp2 (default since round 4)
against the plain 2-stage fp8 kernel (MD_FP8_V2=1) and the fp16 kernel -- accuracy vs fp32 attention over the same e4m3 operands,
then interleaved timing.  GPU box only.
NOTE: MD_FP8_V2 (the 2-stage fp8 kernel for every shape) existed only on the day of the run (gpurun r4t).
"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from attn_ab import *   # noqa: F401,F403
E4 = torch.float8_e4m3fn


def make8(shape, seed=0):
    b, heads, nq, n0, n1, n1b, d = shape
    g = torch.Generator(device="cpu").manual_seed(seed)
    c = heads * d
    q = torch.randn(b, nq, c, generator=g).to(dev).to(F16)
    k0 = torch.randn(b, n0, c, generator=g).to(dev).to(E4); v0 = torch.randn(b, n0, c, generator=g).to(dev).to(E4)
    ld0 = (n0 + 15) // 16 * 16
    vt0 = torch.zeros(b, c, ld0, dtype=E4, device=dev); vt0[:, :, :n0] = v0.transpose(1, 2)
    kw, k1, v1 = {}, None, None
    if n1:
        k1 = torch.randn(1, n1, c, generator=g).to(dev).to(E4); v1 = torch.randn(1, n1, c, generator=g).to(dev).to(E4)
        ld1 = (n1 + 15) // 16 * 16
        vt1 = torch.zeros(1, c, ld1, dtype=E4, device=dev); vt1[:, :, :n1] = v1.transpose(1, 2)
        kw = dict(k1=k1.view(torch.uint8), vt1=vt1.view(torch.uint8), n1=n1, ld_k1=c, ld_vt1=ld1, k1_bs=0, vt1_bs=0, n1_batches=n1b)
    out = torch.empty(b, nq, c, dtype=F16, device=dev)
    def run():
        ops.attention(q, k0.view(torch.uint8), vt0.view(torch.uint8), out, batch=b, heads=heads, nq=nq, d=d, n0=n0, ld_q=c, ld_k0=c,
                      ld_vt0=ld0, ld_out=c, q_bs=nq * c, k0_bs=n0 * c, vt0_bs=c * ld0, out_bs=nq * c, kv_fp8=True, **kw)
    f = lambda t: None if t is None else t.float().to(F16)   # noqa: E731  (e4m3 values are exact in fp16)
    return run, out, (q, f(k0), f(v0), f(k1), f(v1))


SH = {"d40 64^2 B=2": (2, 8, 4096, 4096, 4096, 1, 40), "d40 64^2 B=16": (16, 8, 4096, 4096, 4096, 8, 40), "d40 96^2 B=2": (2, 8, 9216, 9216, 9216, 1, 40),
      "d80 32^2 B=2": (2, 8, 1024, 1024, 1024, 1, 80), "d80 48^2 B=2": (2, 8, 2304, 2304, 2304, 1, 80), "d40 64^2 B=1": (1, 8, 4096, 4096, 4096, 1, 40)}
for name, shape in SH.items():
    run8, out8, ten = make8(shape, seed=4)
    run16, out16, _ = make(shape, seed=4)
    line = f"{name}:"
    if shape[0] <= 2 and shape[2] <= 4096:
        ref = reference(shape, ten)
        for tag, env in (("fp8 pipelined", None), ("fp8 2-stage", "1")):
            if env: os.environ["MD_FP8_V2"] = env
            else: os.environ.pop("MD_FP8_V2", None)
            out8.zero_(); run8(); torch.cuda.synchronize()
            e = out8.float() - ref
            line += f"  {tag}: max abs {float(e.abs().max()):.2e} rms {float(e.pow(2).mean().sqrt()):.2e} |"
        line += f" (ref max {float(ref.abs().max()):.3f} rms {float(ref.pow(2).mean().sqrt()):.3f})"
        del ref
    print(line, flush=True)
    res = {"new": [], "old": [], "f16": []}
    for rnd in range(3):
        os.environ.pop("MD_FP8_V2", None); res["new"].append(time_us(run8))
        os.environ["MD_FP8_V2"] = "1"; res["old"].append(time_us(run8))
        res["f16"].append(time_us(run16))
    os.environ.pop("MD_FP8_V2", None)
    print(f"   fp8 pipelined {min(res['new']):.1f} us ({tf(shape, min(res['new'])):.0f} TF)   fp8 2-stage {min(res['old']):.1f} us ({tf(shape, min(res['old'])):.0f} TF)   "
          f"fp16 {min(res['f16']):.1f} us ({tf(shape, min(res['f16'])):.0f} TF)", flush=True)
