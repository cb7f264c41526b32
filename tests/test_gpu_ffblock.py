"""GPU parity of md_ff_block (the fused transformer-block tail: attn2.to_out + residual, norm3, GEGLU, feed-forward output +
residual; ldm/modules/attention.py:318-319, 50-77) against plain PyTorch fp32 of the same ops, and against the md_igemm launches it
replaces.  Tolerances are written next to each check."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

F16, F32 = torch.float16, torch.float32


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from magicdance_amd import _lib
    _lib.load()  # fails loudly if the HIP extension is missing
    return torch.device("cuda:0")


def _rand(shape, seed, dev, scale=1.0):
    return (torch.randn(shape, generator=torch.Generator(device="cpu").manual_seed(seed)) * scale).to(dev)


def _il(v):
    """the 16-row a | gate interleave of the GEGLU projection (engine.pack_geglu)"""
    half = v.shape[0] // 2
    return torch.stack([v[:half].reshape(half // 16, 16, *v.shape[1:]), v[half:].reshape(half // 16, 16, *v.shape[1:])], 1).reshape(v.shape).contiguous()


def make_params(c, seed, dev):
    """fp32 masters of one block tail + their packed kernel forms"""
    from magicdance_amd import engine
    p = dict(wo=_rand((c, c), seed + 1, dev, c ** -0.5), bo=_rand((c,), seed + 2, dev, 0.1),
             gamma=1.0 + _rand((c,), seed + 3, dev, 0.2), beta=_rand((c,), seed + 4, dev, 0.2),
             w1=_rand((8 * c, c), seed + 5, dev, c ** -0.5), b1=_rand((8 * c,), seed + 6, dev, 0.1),
             w2=_rand((c, 4 * c), seed + 7, dev, (4 * c) ** -0.5), b2=_rand((c,), seed + 8, dev, 0.1))
    wl, s1, s0 = engine.fold_layernorm(p["w1"], p["b1"], p["gamma"], p["beta"], dev)
    p["packed"] = dict(w1=engine.tile_w(_il(wl)), s1=_il(s1), s0=_il(s0), w2=engine.tile_w(p["w2"].to(F16).contiguous()), b2=p["b2"].contiguous(),
                       wo=engine.tile_w(p["wo"].to(F16).contiguous()), bo=p["bo"].contiguous())
    return p


def reference(p, x16, lo16, att16, head):
    """fp32 arithmetic on the fp16 operands the kernel sees (weights rounded to fp16 as the kernel multiplies with them)"""
    t = x16.float() + (0 if lo16 is None else lo16.float())
    if head:
        t = t + F.linear(att16.float(), p["wo"].to(F16).float(), p["bo"])
    c = t.shape[-1]
    xn = F.layer_norm(t.to(F16).float(), (c,), p["gamma"], p["beta"], 1e-5)
    y = F.linear(xn, p["w1"], p["b1"])
    h = y[..., :4 * c] * F.gelu(y[..., 4 * c:])
    return F.linear(h, p["w2"].to(F16).float(), p["b2"]) + t


CASES = [
    # c, m, bm, head, with_lo
    (320, 256, 32, 0, 1), (320, 256, 64, 0, 1), (320, 256, 128, 0, 1),
    (320, 256, 32, 1, 1), (320, 256, 64, 1, 1), (320, 256, 128, 1, 1),
    (320, 200, 64, 1, 0), (320, 72, 128, 0, 0), (320, 1000, 32, 1, 1),       # ragged row counts, single-term stream
    (320, 8192, 0, 1, 1), (320, 12288, 0, 0, 1),                              # the one-frame 64 x 64 level, automatic tile
    (640, 256, 32, 0, 1), (640, 256, 64, 1, 1), (640, 2048, 0, 1, 1), (640, 200, 64, 0, 0),
    # the other step schedules of a tile height (force_bm = rows + 1000 * variant; variant 0 is what the launcher picks)
    (320, 320, 1032, 1, 1), (320, 320, 2032, 1, 1), (320, 320, 3032, 0, 1), (320, 320, 1064, 1, 1), (320, 320, 2064, 0, 1),
]


@pytest.mark.parametrize("case", CASES, ids=[f"c{c}_m{m}_bm{bm}_h{h}_lo{lo}" for c, m, bm, h, lo in CASES])
def test_ff_block_matches_fp32_torch(dev, case):
    from magicdance_amd import ops
    c, m, bm, head, with_lo = case
    p = make_params(c, 10, dev)
    x = _rand((m, c), 1, dev) * 1.5 + 0.4            # non-zero row mean: the folded LayerNorm's mu s1 correction matters
    x16 = x.to(F16)
    lo16 = (x - x16.float()).to(F16) if with_lo else None
    att16 = _rand((m, c), 2, dev).to(F16) if head else None
    ref = reference(p, x16, lo16, att16, head)
    out = torch.full((m, c), float("nan"), dtype=F16, device=dev)
    out_lo = torch.full((m, c), float("nan"), dtype=F16, device=dev)
    pk = p["packed"]
    kw = dict(attn=att16, wo=pk["wo"], bo=pk["bo"]) if head else {}
    ops.ff_block(x16, out, m=m, c=c, w1=pk["w1"], s1=pk["s1"], s0=pk["s0"], w2=pk["w2"], b2=pk["b2"], x_lo=lo16, out_lo=out_lo,
                 force_bm=bm, **kw)
    torch.cuda.synchronize()
    scale = max(1.0, float(ref.abs().max()))
    assert torch.isfinite(out.float()).all() and torch.isfinite(out_lo.float()).all()
    # hi term: fp16 rounding of an O(scale) value + the fp16 operand roundings of h (K = 4c) and of the normalised rows
    assert float((out.float() - ref).abs().max()) <= 4e-3 * scale, case
    # hi + lo: the two-term value carries what the fp16 store dropped
    assert float((out.float() + out_lo.float() - ref).abs().max()) <= 3e-3 * scale, case


@pytest.mark.parametrize("c,m,m_split,bm", [(320, 768, 512, 64), (320, 768, 512, 128), (320, 12288, 8192, 64), (320, 576, 320, 32), (640, 384, 256, 64)])
@pytest.mark.parametrize("head", [0, 1])
def test_ff_block_second_parameter_set(dev, c, m, m_split, bm, head):
    """rows >= m_split use the second parameter set (the pose ControlNet's samples of a merged pass): one launch must equal, bit for
    bit, two launches on the two row ranges (same tile height: the automatic choice depends on the row count of a launch)"""
    from magicdance_amd import ops
    pa, pb = make_params(c, 10, dev)["packed"], make_params(c, 50, dev)["packed"]
    x16 = (_rand((m, c), 1, dev) + 0.2).to(F16)
    lo16 = _rand((m, c), 3, dev, 1e-3).to(F16)
    att16 = _rand((m, c), 2, dev).to(F16) if head else None

    def run(x, lo, att, mm, pk, set2=None, split=0):
        out = torch.empty((mm, c), dtype=F16, device=dev)
        out_lo = torch.empty((mm, c), dtype=F16, device=dev)
        kw = dict(attn=att, wo=pk["wo"], bo=pk["bo"]) if head else {}
        ops.ff_block(x, out, m=mm, c=c, w1=pk["w1"], s1=pk["s1"], s0=pk["s0"], w2=pk["w2"], b2=pk["b2"], x_lo=lo, out_lo=out_lo,
                     force_bm=bm, set2=set2, m_split=split, **kw)
        return out, out_lo
    o, ol = run(x16, lo16, att16, m, pa, set2=pb, split=m_split)
    o1, ol1 = run(x16[:m_split], lo16[:m_split], None if att16 is None else att16[:m_split], m_split, pa)
    o2, ol2 = run(x16[m_split:], lo16[m_split:], None if att16 is None else att16[m_split:], m - m_split, pb)
    torch.cuda.synchronize()
    assert torch.equal(o, torch.cat([o1, o2])) and torch.equal(ol, torch.cat([ol1, ol2]))


@pytest.mark.parametrize("c,b,n", [(320, 2, 4096), (320, 3, 1024), (640, 2, 1024)])
def test_ff_block_agrees_with_the_md_igemm_launches_it_replaces(dev, c, b, n):
    """engine level: the same transformer-block tail through md_ff_block and through {to_out, folded-LN GEGLU, feed-forward output}
    md_igemm launches -- both against fp32 torch, and against each other within the fp16 rounding of the intermediate stores"""
    from magicdance_amd import ops, engine
    m = b * n
    p = make_params(c, 20, dev)
    pk = p["packed"]
    x = _rand((m, c), 1, dev) + 0.3
    x16 = x.to(F16)
    lo16 = (x - x16.float()).to(F16)
    att16 = _rand((m, c), 2, dev).to(F16)
    ref = reference(p, x16, lo16, att16, True)
    fused, fused_lo = torch.empty((m, c), dtype=F16, device=dev), torch.empty((m, c), dtype=F16, device=dev)
    ops.ff_block(x16, fused, m=m, c=c, w1=pk["w1"], s1=pk["s1"], s0=pk["s0"], w2=pk["w2"], b2=pk["b2"], x_lo=lo16, out_lo=fused_lo,
                 attn=att16, wo=pk["wo"], bo=pk["bo"])
    ws = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
    t2, t2_lo = torch.empty((m, c), dtype=F16, device=dev), torch.empty((m, c), dtype=F16, device=dev)
    ops.igemm(att16, pk["wo"], c, batch=b, hin=1, win=n, hout=1, wout=n, c0=c, bias=pk["bo"], res=x16, ld_res=c, res_lo=lo16, out=t2,
              out_lo=t2_lo, ws=ws, w_tiled=True)
    hid = torch.empty((m, 4 * c), dtype=F16, device=dev)
    ops.igemm(t2, pk["w1"], 8 * c, batch=b, hin=1, win=n, hout=1, wout=n, c0=c, out=hid, ld_out=4 * c, act=ops.MD_ACT_GEGLU,
              ln=(pk["s1"], pk["s0"], 1e-5), ws=ws, w_tiled=True)
    un, un_lo = torch.empty((m, c), dtype=F16, device=dev), torch.empty((m, c), dtype=F16, device=dev)
    ops.igemm(hid, pk["w2"], c, batch=b, hin=1, win=n, hout=1, wout=n, c0=4 * c, bias=pk["b2"], res=t2, ld_res=c, res_lo=t2_lo, out=un,
              out_lo=un_lo, ws=ws, w_tiled=True)
    torch.cuda.synchronize()
    scale = max(1.0, float(ref.abs().max()))
    e_f = float((fused.float() + fused_lo.float() - ref).abs().max())
    e_u = float((un.float() + un_lo.float() - ref).abs().max())
    assert e_f <= 3e-3 * scale and e_u <= 3e-3 * scale, (e_f, e_u)
    assert float((fused.float() - un.float()).abs().max()) <= 4e-3 * scale
    # repeated launches are bit-identical (fixed summation order, no atomics)
    again, again_lo = torch.empty_like(fused), torch.empty_like(fused_lo)
    ops.ff_block(x16, again, m=m, c=c, w1=pk["w1"], s1=pk["s1"], s0=pk["s0"], w2=pk["w2"], b2=pk["b2"], x_lo=lo16, out_lo=again_lo,
                 attn=att16, wo=pk["wo"], bo=pk["bo"])
    torch.cuda.synchronize()
    assert torch.equal(again, fused) and torch.equal(again_lo, fused_lo)


@pytest.mark.parametrize("bm", [0, 32, 64, 128, 1032, 2032, 3032, 1064, 2064])
def test_ff_block_is_bit_stable_under_load(dev, bm):
    """the counted waits of the weight ring (compile-time immediates) must never let a piece be read before it landed: repeated
    launches -- cold weight sets in rotation, a copy engine hogging HBM on a second stream, workgroups in both parameter sets -- stay
    bit-identical to the first result (a wait that is one load short shows up here, not on an idle chip)"""
    from magicdance_amd import ops
    c, m, m_split = 320, 12288, 8192
    sets = [make_params(c, 10 + 40 * i, dev)["packed"] for i in range(3)]
    x16 = (_rand((m, c), 1, dev) + 0.2).to(F16)
    lo16 = _rand((m, c), 3, dev, 1e-3).to(F16)
    att16 = _rand((m, c), 2, dev).to(F16)
    hog_src = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
    hog_dst = torch.empty_like(hog_src)
    side = torch.cuda.Stream()

    def run(i):
        pa, pb = sets[i % 3], sets[(i + 1) % 3]
        out, out_lo = torch.empty((m, c), dtype=F16, device=dev), torch.empty((m, c), dtype=F16, device=dev)
        ops.ff_block(x16, out, m=m, c=c, w1=pa["w1"], s1=pa["s1"], s0=pa["s0"], w2=pa["w2"], b2=pa["b2"], x_lo=lo16, out_lo=out_lo,
                     attn=att16, wo=pa["wo"], bo=pa["bo"], set2=pb, m_split=m_split, force_bm=bm)
        return out, out_lo
    refs = [run(i) for i in range(3)]
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        for _ in range(8):
            hog_dst.copy_(hog_src)
    for rep in range(5):
        for i in range(3):
            o, ol = run(i)
            assert torch.equal(o, refs[i][0]) and torch.equal(ol, refs[i][1]), (bm, rep, i)
    torch.cuda.synchronize()
