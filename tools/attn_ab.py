"""Round-4 attention A/B (GPU box only): the round-3 steady loop (MD_ATTN_V=0) against the round-4 form (MD_ATTN_V=1) and its
ablations (MD_ATTN_ABL bit mask: 1 no exp2, 2 no row-max chain, 4 no lane-group exchange, 8 no loads / barrier, 16 s_setprio around
the MFMA blocks, 32 no PV MFMAs, 64 no QK^T MFMAs) -- correctness at the production shapes (128-row query blocks are not reached by
the small unit-test shapes), then interleaved timing rounds in ONE process.
NOTE: the MD_ATTN_V / MD_ATTN_ABL switches this script flips existed only on the day of the run (gpurun r4i): the round-4 loop is
the only loop now, so every 'variant' below times the same kernel.  Kept as the record of how profiles/round4_attention_loop_ab.txt was
produced and for its helpers (make / reference / time_us / tf), which tools/attn_fp8_ab.py imports.
"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from magicdance_amd import ops
dev = torch.device("cuda:0"); F16 = torch.float16; side = torch.cuda.Stream()
SHAPES = {"b2": (2, 8, 4096, 4096, 4096, 1, 40), "b1": (1, 8, 4096, 4096, 0, 0, 40), "b16": (16, 8, 4096, 4096, 4096, 8, 40),
          "b3": (3, 8, 4096, 4096, 4096, 1, 40), "b24": (24, 8, 4096, 4096, 4096, 8, 40)}


def make(shape, seed=0, spike=False):
    b, heads, nq, n0, n1, n1b, d = shape
    g = torch.Generator(device="cpu").manual_seed(seed)
    c = heads * d
    q = torch.randn(b, nq, c, generator=g).to(dev).to(F16); k0 = torch.randn(b, n0, c, generator=g).to(dev).to(F16)
    v0 = torch.randn(b, n0, c, generator=g).to(dev).to(F16)
    if spike:   # one key dominates from a late tile on, for two query rows of different 16-row fragments
        k0[:, 3000] = (q[:, 17] * 4).to(F16); k0[:, 3500] = (q[:, 100] * 6).to(F16)
    vt0 = v0.transpose(1, 2).contiguous()
    kw, k1, v1 = {}, None, None
    if n1:
        k1 = torch.randn(1, n1, c, generator=g).to(dev).to(F16); v1 = torch.randn(1, n1, c, generator=g).to(dev).to(F16)
        if spike:
            k1[:, 2000] = (q[0, 33] * 5).to(F16)
        vt1 = v1.transpose(1, 2).contiguous()
        kw = dict(k1=k1, vt1=vt1, n1=n1, ld_k1=c, ld_vt1=n1, k1_bs=0, vt1_bs=0, n1_batches=n1b)
    out = torch.empty(b, nq, c, dtype=F16, device=dev)
    def run():
        ops.attention(q, k0, vt0, out, batch=b, heads=heads, nq=nq, d=d, n0=n0, ld_q=c, ld_k0=c, ld_vt0=n0, ld_out=c,
                      q_bs=nq * c, k0_bs=n0 * c, vt0_bs=c * n0, out_bs=nq * c, **kw)
    return run, out, (q, k0, v0, k1, v1)


def reference(shape, ten):
    b, heads, nq, n0, n1, n1b, d = shape
    q, k0, v0, k1, v1 = ten
    c = heads * d
    sp = lambda t: t.float().reshape(t.shape[0], t.shape[1], heads, d).permute(0, 2, 1, 3)  # noqa: E731
    ref = torch.empty(b, nq, c, device=dev)
    for i in range(b):
        kk, vv = k0[i:i + 1], v0[i:i + 1]
        if n1 and i < n1b:
            kk, vv = torch.cat([kk, k1], 1), torch.cat([vv, v1], 1)
        s = torch.einsum("bhid,bhjd->bhij", sp(q[i:i + 1]), sp(kk)) * d ** -0.5
        ref[i] = torch.einsum("bhij,bhjd->bhid", s.softmax(-1), sp(vv)).permute(0, 2, 1, 3).reshape(nq, c)
    return ref


def setv(v, abl=0):
    os.environ["MD_ATTN_V"] = str(v); os.environ["MD_ATTN_ABL"] = str(abl)


def time_us(run, reps=10):
    with torch.cuda.stream(side):
        run(); side.synchronize()
        g = ops.Graph(); g.begin()
        for _ in range(reps): run()
        g.end(); g.launch(); side.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(side); g.launch(); e1.record(side); side.synchronize()
        us = e0.elapsed_time(e1) / reps * 1e3
        g.destroy()
    return us


def tf(shape, us):
    b, heads, nq, n0, n1, n1b, d = shape
    return 4.0 * heads * nq * (b * n0 + min(n1b, b) * n1) * d / us / 1e6


def main():
    print("== correctness (max abs error against fp32 softmax attention; outputs are O(0.1))", flush=True)
    for name, spike in (("b2", False), ("b2", True), ("b3", True)):
        run, out, ten = make(SHAPES[name], seed=1, spike=spike)
        ref = reference(SHAPES[name], ten)
        for v in (0, 1):
            setv(v); out.zero_(); run(); torch.cuda.synchronize()
            err = float((out.float() - ref).abs().max())
            print(f"  {name} spike={spike} V={v}: max abs err {err:.3e}  (max |ref| {float(ref.abs().max()):.3f})  {'OK' if err <= 4e-3 else 'FAIL'}", flush=True)
        del ref

    print("== timing, interleaved rounds (us per launch, TFLOP/s of 4 B H Nq Nkv d)", flush=True)
    for name in ("b2", "b16", "b24"):
        run, out, ten = make(SHAPES[name])
        res = {0: [], 1: []}
        for rnd in range(4):
            for v in (0, 1):
                setv(v); res[v].append(time_us(run))
        for v in (0, 1):
            best = min(res[v]); med = sorted(res[v])[len(res[v]) // 2]
            print(f"  {name} V={v}: min {best:.1f} us ({tf(SHAPES[name], best):.0f} TF)  median {med:.1f} us  all {[round(x, 1) for x in res[v]]}", flush=True)

    print("== ablations of the round-4 loop (results are wrong by construction; time only)", flush=True)
    for name in ("b16", "b2"):
        run, out, ten = make(SHAPES[name])
        for abl in (0, 1, 2, 4, 3, 8, 11, 16, 32, 64, 99):
            setv(1, abl); us = min(time_us(run), time_us(run))
            print(f"  {name} ABL={abl:3d}: {us:.1f} us", flush=True)
    setv(0)



if __name__ == "__main__":
    main()
