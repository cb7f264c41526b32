"""md_igemm_params.gn (ABI v10): the GroupNorm that consumes a conv's output, run inside the conv's split-K reduction.  The fused
form must be BIT-IDENTICAL to the two launches it replaces (md_igemm, then md_groupnorm on its output): same slab order, same
epilogue arithmetic, same statistics code on the same thread mapping.  Checked per kernel on the shapes of the 8x8 / 16x16 levels
of a DDIM step (ResBlock conv1 -> GroupNorm -> SiLU with the per-sample time-embedding bias; conv2 + skip -> the next block's
GroupNorm with the two-term residual stream; the merged UNet + ControlNet pass with two parameter sets), on the cases the library
must decline (no split-K, large slices), on the descriptor checks, and end to end: one full-width sampler run with the
producer-side GroupNorm on and off gives the same latent bit for bit."""
import ctypes as C

import pytest
import torch

from tests import helpers as H

pytestmark = pytest.mark.gpu

F16, F32 = torch.float16, torch.float32


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    return torch.device("cuda:0")


def _run(dev, *, b, side, cin, n, k, split, cfg, res, lo, per_sample_bias, silu, eps, dual, fused, seed=0):
    """conv (+ epilogue) -> GroupNorm, either through md_igemm_params.gn or as the two launches.  Returns (out, out_lo, normed, done)."""
    from magicdance_amd import ops
    gen = torch.Generator(device="cpu").manual_seed(seed)
    rnd = lambda *s, scale=1.0: (torch.randn(*s, generator=gen) * scale)   # noqa: E731
    hw, kk = side * side, k * k * cin
    x = rnd(b, hw, cin).to(dev, F16)
    w = rnd(n, kk, scale=kk ** -0.5).to(dev, F16)
    w2 = rnd(n, kk, scale=kk ** -0.5).to(dev, F16)
    bias = rnd(b if per_sample_bias else 1, n).to(dev, F32)
    bias2 = rnd(1, n).to(dev, F32)
    r = rnd(b, hw, n).to(dev, F16) if res else None
    rl = (rnd(b, hw, n) * 1e-3).to(dev, F16) if res and lo else None
    gamma, beta = (1 + 0.2 * rnd(n)).to(dev, F32), (0.2 * rnd(n)).to(dev, F32)
    gamma2, beta2 = (1 + 0.2 * rnd(n)).to(dev, F32), (0.2 * rnd(n)).to(dev, F32)
    out = torch.full((b, hw, n), float("nan"), dtype=F16, device=dev)
    out_lo = torch.full((b, hw, n), float("nan"), dtype=F16, device=dev) if lo else None
    hn = torch.full((b, hw, n), float("nan"), dtype=F16, device=dev)
    ws = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
    gws = torch.empty(1 << 20, dtype=torch.uint8, device=dev)
    b2 = b - 1 if dual else None
    assert not (dual and per_sample_bias)   # bias_batch_stride must be 0 with a second parameter set
    set2 = (b2, w2, bias2[0], None) if dual else None
    gp = ops.groupnorm_params(out, gamma, beta, hn, gws, batch=b, hw=hw, c0=n, groups=32, eps=eps, silu=silu,
                              set2=(b2, gamma2, beta2) if dual else None)
    kw = dict(batch=b, hin=side, win=side, hout=side, wout=side, c0=cin, ksize=k, bias=bias if per_sample_bias else bias[0],
              bias_batch_stride=n if per_sample_bias else 0, res=r, ld_res=n if res else 0, res_lo=rl, out=out, out_lo=out_lo, ws=ws,
              force_cfg=cfg, force_splitk=split, set2=set2,
              force_kg=0 if ops.igemm_config_info(cfg)["ring"] else 1)   # (a forced split stays a split: no in-workgroup k-groups)
    done = None
    if fused:
        done = ops.igemm(x, w, n, gn=gp, **kw)
        if not done:
            ops.groupnorm_launch(gp)
    else:
        ops.igemm(x, w, n, **kw)
        ops.groupnorm_launch(gp)
    torch.cuda.synchronize()
    return out, out_lo, hn, done


# (b, side, cin, n, k, split, cfg): the 8x8 / 16x16 convs of a one-frame step with their tuned (config, split), an 8-frame 8x8 one,
# a 1x1 with split-K, and a narrow test geometry (cpg = 2: four groups per block)
SHAPES = [
    (3, 8, 1280, 1280, 3, 4, 65), (2, 8, 2560, 1280, 3, 8, 66), (3, 16, 1280, 1280, 3, 2, 66), (2, 16, 2560, 1280, 3, 3, 66),
    (2, 16, 1920, 1280, 3, 3, 28), (24, 8, 1280, 1280, 3, 4, 12), (2, 16, 5120, 1280, 1, 4, 29), (2, 8, 64, 64, 3, 2, 15),
    (3, 16, 640, 1280, 3, 4, 15),
]


@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "b{}_{}x{}_c{}_n{}_k{}_s{}_cfg{}".format(s[0], s[1], s[1], *s[2:]))
@pytest.mark.parametrize("mode", ["conv1", "conv2", "attn_norm", "dual"])
def test_reduction_with_groupnorm_is_bit_identical_to_the_two_launches(dev, shape, mode):
    b, side, cin, n, k, split, cfg = shape
    opt = dict(conv1=dict(res=False, lo=False, per_sample_bias=True, silu=True, eps=1e-5, dual=False),      # openaimodel.py:238-252
               conv2=dict(res=True, lo=True, per_sample_bias=False, silu=True, eps=1e-5, dual=False),       # :295 -> next block :221-225
               attn_norm=dict(res=True, lo=False, per_sample_bias=False, silu=False, eps=1e-6, dual=False),  # attention.py:89-90
               dual=dict(res=True, lo=True, per_sample_bias=False, silu=True, eps=1e-5, dual=True))[mode]   # merged UNet + ControlNet pass
    if opt["dual"] and b < 2:
        pytest.skip("two parameter sets need two samples")
    a = _run(dev, b=b, side=side, cin=cin, n=n, k=k, split=split, cfg=cfg, fused=True, **opt)
    r = _run(dev, b=b, side=side, cin=cin, n=n, k=k, split=split, cfg=cfg, fused=False, **opt)
    assert a[3] is True, "the library declined a split-K small-slice call"
    assert torch.isfinite(r[2].float()).all() and torch.isfinite(r[0].float()).all()

    def same(u, v, what):
        if not torch.equal(u, v):
            d = (u.float() - v.float()).abs()
            bad = (d > 0) | torch.isnan(d)
            idx = bad.nonzero()
            raise AssertionError(f"{what} differs: {int(bad.sum())} of {bad.numel()} elements, max |diff| {float(d[~torch.isnan(d)].max()) if (~torch.isnan(d)).any() else 'nan'}, "
                                 f"nan {int(torch.isnan(d).sum())}, first {idx[:4].tolist()}, last {idx[-2:].tolist()}")
    same(a[0], r[0], "conv output")
    if opt["lo"]:
        same(a[1], r[1], "second term of the residual stream")
    same(a[2], r[2], "normalised output")


@pytest.mark.parametrize("case", ["no_split", "large_slice"])
def test_calls_the_library_must_decline(dev, case):
    """no split-K (the reduction does not exist) / a slice too large for the single-launch GroupNorm: *gn_done = 0, the normalised
    tensor is not written by md_igemm, and the caller's md_groupnorm launch produces it"""
    from magicdance_amd import ops
    if case == "no_split":
        kw = dict(b=2, side=8, cin=1280, n=1280, k=3, split=1, cfg=28)
    else:
        kw = dict(b=2, side=64, cin=320, n=320, k=3, split=2, cfg=15)
    opt = dict(res=True, lo=True, per_sample_bias=False, silu=True, eps=1e-5, dual=False)
    gen_a = _run(dev, fused=True, **kw, **opt)
    gen_r = _run(dev, fused=False, **kw, **opt)
    assert gen_a[3] is False
    assert torch.equal(gen_a[0], gen_r[0]) and torch.equal(gen_a[2], gen_r[2])
    # the declined call itself leaves gn->out alone
    b, side, n = kw["b"], kw["side"], kw["n"]
    x = torch.randn(b, side * side, kw["cin"], device=dev).to(F16)
    w = (torch.randn(n, 9 * kw["cin"], device=dev) * 0.01).to(F16)
    out = torch.empty(b, side * side, n, dtype=F16, device=dev)
    hn = torch.full((b, side * side, n), 7.0, dtype=F16, device=dev)
    ws = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
    g = torch.ones(n, device=dev)
    gp = ops.groupnorm_params(out, g, g, hn, torch.empty(1 << 20, dtype=torch.uint8, device=dev), batch=b, hw=side * side, c0=n)
    assert ops.igemm(x, w, n, batch=b, hin=side, win=side, hout=side, wout=side, c0=kw["cin"], ksize=3, out=out, ws=ws,
                     force_cfg=kw["cfg"], force_splitk=kw["split"], force_kg=1, gn=gp) is False
    torch.cuda.synchronize()
    assert (hn == 7.0).all()


def test_descriptor_that_does_not_describe_the_output_is_refused(dev):
    from magicdance_amd import _lib, ops
    b, side, cin, n = 2, 8, 64, 64
    x = torch.randn(b, side * side, cin, device=dev).to(F16)
    w = (torch.randn(n, 9 * cin, device=dev) * 0.05).to(F16)
    out = torch.empty(b, side * side, n, dtype=F16, device=dev)
    other = torch.empty_like(out)
    hn = torch.empty_like(out)
    ws = torch.empty(8 << 20, dtype=torch.uint8, device=dev)
    gws = torch.empty(1 << 20, dtype=torch.uint8, device=dev)
    g = torch.ones(n, device=dev)
    kw = dict(batch=b, hin=side, win=side, hout=side, wout=side, c0=cin, ksize=3, out=out, ws=ws)
    bad = [ops.groupnorm_params(other, g, g, hn, gws, batch=b, hw=side * side, c0=n),            # x0 is not this call's output
           ops.groupnorm_params(out, g, g, out, gws, batch=b, hw=side * side, c0=n),             # normalises in place
           ops.groupnorm_params(out, g, g, hn, gws, batch=b, hw=side * side // 2, c0=n),         # another geometry
           ops.groupnorm_params(out, g, g, hn, gws, batch=b, hw=side * side, c0=n // 2, x1=other, c1=n // 2),   # two sources
           ops.groupnorm_params(out, g, g, hn, gws, batch=b, hw=side * side, c0=n, set2=(1, g, g))]   # a second set the GEMM does not have
    for gp in bad:
        with pytest.raises(_lib.MagicDanceHipError, match="MD_ERR_BAD_ARG"):
            ops.igemm(x, w, n, gn=gp, **kw)
    # gn without the host flag
    p = _lib.IgemmParams()
    good = ops.groupnorm_params(out, g, g, hn, gws, batch=b, hw=side * side, c0=n)
    p.a0, p.c0, p.batch, p.hin, p.win, p.hout, p.wout, p.ksize, p.stride = x.data_ptr(), cin, b, side, side, side, side, 3, 1
    p.w, p.n, p.out, p.ld_out, p.n_tr_begin, p.force_cfg = w.data_ptr(), n, out.data_ptr(), n, n, -1
    p.gn = C.cast(C.pointer(good), C.c_void_p)
    assert _lib.load().md_igemm(C.byref(p), ops.stream_ptr()) == -1   # MD_ERR_BAD_ARG
    assert ops.igemm(x, w, n, gn=good, **kw) in (True, False)   # the same descriptor with the flag is served


def test_sampler_with_and_without_producer_side_groupnorm_is_bit_identical(dev, monkeypatch):
    """full SD-1.5 width, one frame, 4 DDIM steps of the fused route (merged UNet + ControlNet pass, captured graph): conv(gn_next=)
    on (the default) and off give the same latent bit for bit, and the on run did absorb GroupNorm launches into reductions"""
    from magicdance_amd import engine, ops
    g = H.load_golden("c1_b1_s50")
    inp = H.case_inputs(g)
    mv = lambda d: {k: ([t.to(dev) for t in v] if isinstance(v, list) else v) for k, v in d.items()}  # noqa: E731
    zs, absorbed = {}, {}
    for on in (True, False):
        monkeypatch.setattr(engine, "_GN_NEXT", 3 if on else 0)
        model = H.build_hip_model(320, 8, seed=0, device=dev, image_size=64)
        n_done = [0]
        orig = ops.igemm

        def counting(*a, _o=orig, **kw):
            r = _o(*a, **kw)
            if "gn" in kw and r is True:
                n_done[0] += 1
            return r
        monkeypatch.setattr(ops, "igemm", counting)
        z, _ = model.sample_log(cond=mv(inp["c"]), batch_size=1, ddim=True, ddim_steps=4, eta=0.0, unconditional_guidance_scale=7,
                                unconditional_conditioning=mv(inp["uc"]), inpaint=None, x_T=inp["x_T"].to(dev))
        torch.cuda.synchronize()
        monkeypatch.setattr(ops, "igemm", orig)
        zs[on], absorbed[on] = z.clone(), n_done[0]
        del model
    assert absorbed[True] >= 10 and absorbed[False] == 0, absorbed
    assert torch.isfinite(zs[True]).all() and torch.equal(zs[True], zs[False])
