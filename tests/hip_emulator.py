"""TEST INFRASTRUCTURE ONLY: torch-CPU emulation of the C ABI (include/magicdance_hip.h) with the same pointer /
leading-dimension semantics, so that the HOST logic of the product (weight packing, NHWC layouts, fused-QKV and
V^T stores, GEGLU interleave, bank read/write plumbing, arena reuse, sampler control flow) can be checked against
the reference goldens in the CPU test tier.  It is installed by monkeypatching ``magicdance_amd.ops`` inside a test;
nothing in the product imports it, and it is never what a GPU test or the bench measures."""
import contextlib
import math

import torch
import torch.nn.functional as F

F16, F32 = torch.float16, torch.float32


def _mem(t, size, stride):
    """pointer semantics: ``t``'s first element is the base address; view memory from there."""
    return t.as_strided(size, stride, t.storage_offset())


def _off(t, elems):
    """pointer + elems (in elements of t's dtype)"""
    return t.as_strided((1,), (1,), t.storage_offset() + elems)


def igemm(a0, w, n, *, batch, hin, win, hout, wout, c0, ksize=1, stride=1, ups=0, a1=None, c1=0, bias=None,
          bias_batch_stride=0, res=None, ld_res=0, act=0, out=None, ld_out=None, out_f32=False, out_t=None,
          n_tr_begin=None, ld_t=0, ws=None, force_cfg=-1, force_splitk=0, asym_pad=False, ln=None, res_lo=None, out_lo=None,
          col_scale=None, k8=None, vt_fp8=False, set2=None, gn_part=None, force_kg=0, w_tiled=False, gn=None):
    if gn is not None:   # md_igemm_params.gn: the emulated library "fuses" the consumer's GroupNorm at <= 64 pixels per sample, so the
        # CPU tier walks both branches of the caller (normalised by this call / the caller's own launch still due)
        assert gn["x0"] is out and gn["x1"] is None and gn["c0"] == n and gn["batch"] == batch and gn["hw"] == hout * wout
        assert (gn["set2"] is None) == (set2 is None) and (set2 is None or gn["set2"][0] == set2[0])
        igemm(a0, w, n, batch=batch, hin=hin, win=win, hout=hout, wout=wout, c0=c0, ksize=ksize, stride=stride, ups=ups, a1=a1, c1=c1,
              bias=bias, bias_batch_stride=bias_batch_stride, res=res, ld_res=ld_res, act=act, out=out, ld_out=ld_out, out_f32=out_f32,
              out_t=out_t, n_tr_begin=n_tr_begin, ld_t=ld_t, ws=ws, asym_pad=asym_pad, ln=ln, res_lo=res_lo, out_lo=out_lo,
              col_scale=col_scale, k8=k8, vt_fp8=vt_fp8, set2=set2, gn_part=gn_part, w_tiled=w_tiled)
        if hout * wout > 64:
            return False
        groupnorm_launch(gn)
        return True
    if w_tiled:   # tiled weight storage (md_igemm_params.w_tiled): back to row-major, then as below
        from magicdance_amd.ops import untile_weights
        kk_ = ksize * ksize * (c0 + c1)
        w = untile_weights(_mem(w, (n, kk_), (kk_, 1)), ksize)
        if set2 is not None:
            set2 = (set2[0], untile_weights(_mem(set2[1], (n, kk_), (kk_, 1)), ksize)) + tuple(set2[2:])
    if set2 is not None:   # two parameter sets: samples >= batch2 use (w2, bias2, ln2) -- two plain calls on the two sample ranges
        b2, w2, bias2, ln2 = set2
        assert 0 < b2 < batch and bias_batch_stride == 0
        tok = hout * wout
        ntr0_ = n if n_tr_begin is None else n_tr_begin
        ldo = (n if ld_out is None else ld_out)
        kw = dict(hin=hin, win=win, hout=hout, wout=wout, c0=c0, ksize=ksize, stride=stride, ups=ups, c1=c1, ld_res=ld_res, act=act,
                  ld_out=ld_out, out_f32=out_f32, n_tr_begin=n_tr_begin, ld_t=ld_t, ws=ws, asym_pad=asym_pad, col_scale=col_scale,
                  vt_fp8=vt_fp8)
        igemm(a0, w, n, batch=b2, a1=a1, bias=bias, res=res, out=out, out_t=out_t, ln=ln, res_lo=res_lo, out_lo=out_lo, k8=k8,
              gn_part=gn_part, **kw)
        o = lambda t, e: None if t is None else _off(t, e)  # noqa: E731
        igemm(_off(a0, b2 * hin * win * c0), w2, n, batch=batch - b2, a1=o(a1, b2 * hin * win * c1), bias=bias2,
              res=o(res, b2 * tok * ld_res), out=_off(out, b2 * tok * ldo), out_t=o(out_t, b2 * (n - ntr0_) * ld_t),
              ln=None if ln is None else (ln2[0], ln2[1], ln[2]), res_lo=o(res_lo, b2 * tok * ld_res),
              out_lo=o(out_lo, b2 * tok * ldo), k8=None if k8 is None else (_off(k8[0], b2 * tok * k8[3]),) + tuple(k8[1:]),
              gn_part=o(gn_part, (b2 * tok // 64) * 2 * n), **kw)
        return out
    cin = c0 + c1
    xs = [_mem(a0, (batch, hin, win, c0), (hin * win * c0, win * c0, c0, 1)).float()]
    if a1 is not None:
        xs.append(_mem(a1, (batch, hin, win, c1), (hin * win * c1, win * c1, c1, 1)).float())
    x = torch.cat(xs, -1).permute(0, 3, 1, 2)
    if ups:
        x = F.interpolate(x, scale_factor=2, mode="nearest")
    kk = ksize * ksize * cin
    wt = _mem(w, (n, ksize, ksize, cin), (kk, ksize * cin, cin, 1)).float().permute(0, 3, 1, 2)
    if asym_pad:
        y = F.conv2d(F.pad(x, (0, 1, 0, 1)), wt, stride=stride, padding=0)
    else:
        y = F.conv2d(x, wt, stride=stride, padding=ksize // 2)
    assert y.shape[2] == hout and y.shape[3] == wout, (y.shape, hout, wout)
    tokens = hout * wout
    y = y.permute(0, 2, 3, 1).reshape(batch, tokens, n)
    if ln is not None:   # folded LayerNorm: out = rstd (acc - mu s1) + s0 with the row statistics of the (fp16) A rows
        s1, s0, eps = ln
        assert ksize == 1 and a1 is None and bias is None
        rows = xs[0].reshape(batch, tokens, c0)
        mu = rows.mean(-1, keepdim=True)
        rstd = torch.rsqrt(rows.var(-1, unbiased=False, keepdim=True) + eps)
        y = rstd * (y - mu * _mem(s1, (n,), (1,))) + _mem(s0, (n,), (1,))
    if bias is not None:
        if bias_batch_stride:
            y = y + _mem(bias, (batch, 1, n), (bias_batch_stride, 0, 1))
        else:
            y = y + _mem(bias, (n,), (1,))
    if col_scale is not None:
        y = torch.cat([y[..., :col_scale[1]] * col_scale[0], y[..., col_scale[1]:]], -1)
    ld_out = n if ld_out is None else ld_out
    if act == 2:  # GEGLU on 16-row interleaved packing
        y = y.reshape(batch, tokens, n // 32, 2, 16)
        y = (y[..., 0, :] * F.gelu(y[..., 1, :])).reshape(batch, tokens, n // 2)
        _mem(out, (batch, tokens, n // 2), (tokens * ld_out, ld_out, 1)).copy_(y)
        return out
    if act == 1:
        y = F.silu(y)
    ntr0 = n if n_tr_begin is None else n_tr_begin
    if ntr0 < n:
        ntr = n - ntr0
        yt = y[..., ntr0:].transpose(1, 2)
        if vt_fp8:
            _mem(out_t, (batch, ntr, tokens), (ntr * ld_t, ld_t, 1)).copy_(_to_e4m3(yt))
        else:
            _mem(out_t, (batch, ntr, tokens), (ntr * ld_t, ld_t, 1)).copy_(yt)
        y = y[..., :ntr0]
    if k8 is not None:
        kt, k0_, k1_, ldk = k8
        _mem(kt, (batch, tokens, k1_ - k0_), (tokens * ldk, ldk, 1)).copy_(_to_e4m3(y[..., k0_:k1_]))
        assert k0_ == 0 or k1_ == ntr0, "emulator: the fp8 columns are a prefix or a suffix of the row-major part"
        if k0_ == 0:
            if k1_ == ntr0:
                return out
            raise NotImplementedError
        y, ntr0 = y[..., :k0_], k0_
    if res is not None:
        y = y + _mem(res, (batch, tokens, ntr0), (tokens * ld_res, ld_res, 1)).float()
        if res_lo is not None:
            y = y + _mem(res_lo, (batch, tokens, ntr0), (tokens * ld_res, ld_res, 1)).float()
    o = _mem(out, (batch, tokens, ntr0), (tokens * ld_out, ld_out, 1))
    o.copy_(y)
    if gn_part is not None:   # per 64-row granule and column: sum / sum of squares of the STORED (rounded) values
        assert tokens % 64 == 0 and ntr0 == n and not out_f32 and k8 is None
        v = o.float().reshape(batch * tokens // 64, 64, n)
        pt = _mem(gn_part, (batch * tokens // 64, 2, n), (2 * n, n, 1))
        pt[:, 0].copy_(v.sum(1))
        pt[:, 1].copy_((v * v).sum(1))
    if out_lo is not None:   # what the fp16 store dropped (two-term residual stream)
        _mem(out_lo, (batch, tokens, ntr0), (tokens * ld_out, ld_out, 1)).copy_(y - o.float())
    return out


def ff_block_supported(m, c):
    return c % 64 == 0 and m > 0   # (the emulation is generic; the kernel serves 320 / 640)


def ff_block(x, out, *, m, c, w1, s1, s0, w2, b2, ln_eps=1e-5, x_lo=None, out_lo=None, attn=None, wo=None, bo=None, set2=None,
             m_split=0, force_bm=0):
    """md_ff_block: [to_out + residual ->] LayerNorm-folded GEGLU projection -> feed-forward output + residual"""
    from magicdance_amd.ops import untile_weights
    if set2 is not None and 0 < m_split < m:
        o = lambda t, e: None if t is None else _off(t, e)  # noqa: E731
        ff_block(x, out, m=m_split, c=c, w1=w1, s1=s1, s0=s0, w2=w2, b2=b2, ln_eps=ln_eps, x_lo=x_lo, out_lo=out_lo, attn=attn, wo=wo, bo=bo)
        e = m_split * c
        ff_block(_off(x, e), _off(out, e), m=m - m_split, c=c, w1=set2["w1"], s1=set2["s1"], s0=set2["s0"], w2=set2["w2"],
                 b2=set2["b2"], ln_eps=ln_eps, x_lo=o(x_lo, e), out_lo=o(out_lo, e), attn=o(attn, e), wo=set2.get("wo"), bo=set2.get("bo"))
        return out
    t = _mem(x, (m, c), (c, 1)).float()
    if x_lo is not None:
        t = t + _mem(x_lo, (m, c), (c, 1)).float()
    if attn is not None:
        wo_ = untile_weights(_mem(wo, (c, c), (c, 1))).float()
        t = t + _mem(attn, (m, c), (c, 1)).float() @ wo_.t() + _mem(bo, (c,), (1,))
    t16 = t.to(F16).float()   # the fp16 rows the kernel multiplies with (and takes the LayerNorm statistics of)
    mu = t16.mean(-1, keepdim=True)
    rstd = torch.rsqrt(t16.var(-1, unbiased=False, keepdim=True) + ln_eps)
    w1_ = untile_weights(_mem(w1, (8 * c, c), (c, 1))).float()
    y = rstd * (t16 @ w1_.t() - mu * _mem(s1, (8 * c,), (1,))) + _mem(s0, (8 * c,), (1,))
    y = y.reshape(m, 8 * c // 32, 2, 16)
    h = (y[:, :, 0, :] * F.gelu(y[:, :, 1, :])).reshape(m, 4 * c).to(F16).float()
    w2_ = untile_weights(_mem(w2, (c, 4 * c), (4 * c, 1))).float()
    v = h @ w2_.t() + _mem(b2, (c,), (1,)) + t
    o = _mem(out, (m, c), (c, 1))
    o.copy_(v)
    if out_lo is not None:
        _mem(out_lo, (m, c), (c, 1)).copy_(v - o.float())
    return out


def _to_e4m3(t):
    """fp32 -> OCP e4m3 bytes (saturating, as the hardware conversion)"""
    return t.float().clamp(-448.0, 448.0).to(torch.float8_e4m3fn).view(torch.uint8)


def _from_e4m3(t):
    return t.view(torch.float8_e4m3fn).float()


def attention(q, k0, vt0, out, *, batch, heads, nq, d, n0, ld_q, ld_k0, ld_vt0, ld_out, q_bs, k0_bs, vt0_bs, out_bs,
              k1=None, vt1=None, n1=0, ld_k1=0, ld_vt1=0, k1_bs=0, vt1_bs=0, n1_batches=0, scale=None, q_prescaled=False, kv_fp8=False,
              causal=False):
    assert not causal or (nq == n0 and k1 is None and not kv_fp8)
    scale = d ** -0.5 if scale is None else scale
    if q_prescaled:   # q carries scale * log2(e): q.k is the log2-domain logit
        scale = math.log(2.0)
    if kv_fp8:   # e4m3 K / V^T bytes; q and P are rounded to e4m3 too, row sums use the unrounded P (as the kernel does)
        dec = lambda t, size, stride: _from_e4m3(_mem(t, size, stride))  # noqa: E731
        qq = _mem(q, (batch, heads, nq, d), (q_bs, d, ld_q, 1)).float() * (scale / math.log(2.0))
        qq = _from_e4m3(_to_e4m3(qq))
        o = _mem(out, (batch, heads, nq, d), (out_bs, d, ld_out, 1))
        for b in range(batch):
            kb = dec(k0, (batch, heads, n0, d), (k0_bs, d, ld_k0, 1))[b]
            vb = dec(vt0, (batch, heads, n0, d), (vt0_bs, d * ld_vt0, 1, ld_vt0))[b]
            if k1 is not None and b < n1_batches:   # (the bank segment holds n1_batches samples, or one shared one: stride 0)
                kb = torch.cat([kb, dec(k1, (n1_batches, heads, n1, d), (k1_bs, d, ld_k1, 1))[b]], 1)
                vb = torch.cat([vb, dec(vt1, (n1_batches, heads, n1, d), (vt1_bs, d * ld_vt1, 1, ld_vt1))[b]], 1)
            s2 = torch.einsum("hid,hjd->hij", qq[b], kb)
            p = torch.exp2(s2 - s2.max(-1, keepdim=True).values)
            o[b].copy_(torch.einsum("hij,hjd->hid", _from_e4m3(_to_e4m3(p)), vb) / p.sum(-1, keepdim=True))
        return out
    qq = _mem(q, (batch, heads, nq, d), (q_bs, d, ld_q, 1)).float()
    kk = _mem(k0, (batch, heads, n0, d), (k0_bs, d, ld_k0, 1)).float()
    vv = _mem(vt0, (batch, heads, n0, d), (vt0_bs, d * ld_vt0, 1, ld_vt0)).float()
    o = _mem(out, (batch, heads, nq, d), (out_bs, d, ld_out, 1))
    for b in range(batch):
        kb, vb = kk[b], vv[b]
        if k1 is not None and b < n1_batches:
            k1b = _mem(k1, (n1_batches, heads, n1, d), (k1_bs, d, ld_k1, 1)).float()[b]   # only n1_batches samples exist there
            v1b = _mem(vt1, (n1_batches, heads, n1, d), (vt1_bs, d * ld_vt1, 1, ld_vt1)).float()[b]
            kb, vb = torch.cat([kb, k1b], 1), torch.cat([vb, v1b], 1)
        s = torch.einsum("hid,hjd->hij", qq[b], kb) * scale
        if causal:
            s = s.masked_fill(torch.ones(nq, n0, dtype=torch.bool).triu(1), float("-inf"))
        o[b].copy_(torch.einsum("hij,hjd->hid", s.softmax(-1), vb))
    return out


def groupnorm_ws_bytes(batch, hw, groups=32):
    return 1024


# the emulated md_groupnorm "wants partials" wherever they are usable (hw % 64 == 0), so that the CPU tier exercises the
# producer -> consumer plumbing of the partial statistics at the small test geometries too
def groupnorm_wants_partials(batch, hw, c, groups=32):
    return hw % 64 == 0 and c % groups == 0


def groupnorm_params(x0, gamma, beta, out, ws, **kw):
    return dict(x0=x0, gamma=gamma, beta=beta, out=out, ws=ws, **{"x1": None, "set2": None, **kw})


def groupnorm_launch(p, out=None):
    return groupnorm(**p)


def groupnorm(x0, gamma, beta, out, ws, *, batch, hw, c0, x1=None, c1=0, groups=32, eps=1e-5, silu=False, set2=None, part0=None,
              part1=None):
    if set2 is not None:
        b2, gamma2, beta2 = set2
        assert 0 < b2 < batch
        kw = dict(hw=hw, c0=c0, c1=c1, groups=groups, eps=eps, silu=silu)
        o = lambda t, e: None if t is None else _off(t, e)  # noqa: E731
        groupnorm(x0, gamma, beta, out, ws, batch=b2, x1=x1, part0=part0, part1=part1, **kw)
        groupnorm(_off(x0, b2 * hw * c0), gamma2, beta2, _off(out, b2 * hw * (c0 + c1)), ws, batch=batch - b2,
                  x1=None if x1 is None else _off(x1, b2 * hw * c1), part0=o(part0, (b2 * hw // 64) * 2 * c0),
                  part1=o(part1, (b2 * hw // 64) * 2 * c1), **kw)
        return out
    xs = [_mem(x0, (batch, hw, c0), (hw * c0, c0, 1)).float()]
    if x1 is not None:
        xs.append(_mem(x1, (batch, hw, c1), (hw * c1, c1, 1)).float())
    x = torch.cat(xs, -1).transpose(1, 2)  # [B, C, HW]
    if part0 is not None and (x1 is None or part1 is not None) and hw % 64 == 0:
        # statistics from the producers' partials, NOT from x: a stale / misplaced partial buffer must show up as a wrong result
        P = hw // 64
        ps = [_mem(part0, (batch, P, 2, c0), (P * 2 * c0, 2 * c0, c0, 1))]
        if x1 is not None:
            ps.append(_mem(part1, (batch, P, 2, c1), (P * 2 * c1, 2 * c1, c1, 1)))
        pc = torch.cat(ps, -1).sum(1)                       # [B, 2, C]
        c, cpg = c0 + c1, (c0 + c1) // groups
        sg = pc.reshape(batch, 2, groups, cpg).sum(-1) / (hw * cpg)
        mu, var = sg[:, 0], (sg[:, 1] - sg[:, 0] ** 2).clamp_min(0)
        xg = x.reshape(batch, groups, cpg, hw)
        y = ((xg - mu[:, :, None, None]) * torch.rsqrt(var + eps)[:, :, None, None]).reshape(batch, c, hw)
        y = y * gamma.float()[None, :, None] + beta.float()[None, :, None]
    else:
        y = F.group_norm(x, groups, gamma.float(), beta.float(), eps=eps)
    if silu:
        y = F.silu(y)
    c = c0 + c1
    _mem(out, (batch, hw, c), (hw * c, c, 1)).copy_(y.transpose(1, 2))
    return out


def layernorm(x, gamma, beta, out, rows, c, eps=1e-5):
    y = F.layer_norm(_mem(x, (rows, c), (c, 1)).float(), (c,), gamma, beta, eps)
    _mem(out, (rows, c), (c, 1)).copy_(y)
    return out


def nchw_to_nhwc_f16(x, out, batch, c, hw, cpad):
    o = _mem(out, (batch, hw, cpad), (hw * cpad, cpad, 1))
    o.zero_()
    o[..., :c].copy_(_mem(x, (batch, c, hw), (c * hw, hw, 1)).transpose(1, 2))
    return out


def nhwc_to_nchw_f32(x, out, batch, c, hw, ld):
    _mem(out, (batch, c, hw), (c * hw, hw, 1)).copy_(_mem(x, (batch, hw, c), (hw * ld, ld, 1)).transpose(1, 2))
    return out


def add_f16(a, b, out, n, b_period=None):
    b_period = n if b_period is None else b_period
    av = _mem(a, (n // b_period, b_period), (b_period, 1)).float()
    r = av + _mem(b, (1, b_period), (0, 1)).float()
    _mem(out, (n // b_period, b_period), (b_period, 1)).copy_(r)
    return out


def image_to_u8(x, out, batch, c, hw, scale, bias):
    v = (_mem(x, (batch, c, hw), (c * hw, hw, 1)).float() * scale + bias).clamp(0, 1)
    _mem(out, (batch, hw, c), (hw * c, c, 1)).copy_((v * 255.0 + 0.5).to(torch.uint8).transpose(1, 2))
    return out


def timestep_embedding(t, out, nt, dim, max_period=10000.0):
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=F32) / half)
    args = _mem(t, (nt,), (1,))[:, None].float() * freqs[None]
    _mem(out, (nt, dim), (dim, 1)).copy_(torch.cat([torch.cos(args), torch.sin(args)], -1))
    return out


def gemv_f32(x, w, bias, y, rows, k, n, act_in=False):
    xv = _mem(x, (rows, k), (k, 1)).float()
    if act_in:
        xv = F.silu(xv)
    r = xv @ _mem(w, (n, k), (k, 1)).float().t()
    if bias is not None:
        r = r + _mem(bias, (n,), (1,))
    _mem(y, (rows, n), (n, 1)).copy_(r)
    return y


def softmax_rows(s, ld_s, p, ld_p, rows, cols, scale):
    _mem(p, (rows, cols), (ld_p, 1)).copy_((_mem(s, (rows, cols), (ld_s, 1)).float() * scale).softmax(-1))
    return p


def select_row_f32(table, counter, row_offset, dst, width, nrows=None):
    nrows = int(table.numel() // width) if nrows is None else nrows
    row = min(max((int(counter[0]) if counter is not None else 0) + row_offset, 0), nrows - 1)
    _mem(dst, (width,), (1,)).copy_(_mem(table, (row + 1, width), (width, 1))[row])
    return dst


def gather_rows(table, seg, nseg, max_row_units, counter, row_offset, dst, nrows, rows_per_block=None, block_units=0):
    rows_per_block = nrows if rows_per_block is None else rows_per_block
    row = min(max((int(counter[0]) if counter is not None else 0) + row_offset, 0), nrows - 1)
    blk, within = divmod(row, rows_per_block)
    tab, d = table.view(-1), dst.view(-1)
    u = 16 // tab.element_size()
    for off, ln, doff in seg.view(-1, 3).tolist()[:nseg]:
        src = blk * block_units + off + within * ln
        d[doff * u:(doff + ln) * u].copy_(tab[src * u:(src + ln) * u])
    return dst


def counter_add(counter, delta):
    counter += delta


def gather_frames(src, dst, idx_table, counter, window, row_bytes):
    st = min(max(int(counter[0]), 0), idx_table.shape[0] - 1)
    idx = idx_table[st, window].long()
    n = idx.numel()
    sv = src.contiguous().view(torch.uint8).reshape(-1, row_bytes)
    dst.view(torch.uint8).reshape(-1)[:n * row_bytes].copy_(sv[idx].reshape(-1))
    return dst


def cfg_scatter_add(eps_c, eps_u, ld_eps, coef, idx_table, counter, window, pred, counts, hw, c):
    st = min(max(int(counter[0]), 0), idx_table.shape[0] - 1)
    idx = idx_table[st, window].long()
    n = idx.numel()
    ec = _mem(eps_c, (n, hw, c), (hw * ld_eps, ld_eps, 1)).float()
    eu = _mem(eps_u, (n, hw, c), (hw * ld_eps, ld_eps, 1)).float()
    pv = pred.reshape(-1, hw, c)
    pv.index_add_(0, idx, eu + float(coef[4]) * (ec - eu))
    counts.index_add_(0, idx, torch.ones(n))


def window_mean(pred, counts, eps, frames, per_frame):
    eps.reshape(frames, per_frame).copy_(pred.reshape(frames, per_frame) / counts.reshape(frames, 1))
    pred.zero_()
    counts.zero_()
    return eps


def ddim_update(eps_c, eps_u, ld_eps, x, noise, coef, x_prev, pred_x0, eps_out, batch, c, hw):
    a_t, a_prev, sigma, s1m, scale = [float(v) for v in coef]
    e = _mem(eps_c, (batch, hw, c), (hw * ld_eps, ld_eps, 1)).transpose(1, 2)
    if eps_u is not None:
        eu = _mem(eps_u, (batch, hw, c), (hw * ld_eps, ld_eps, 1)).transpose(1, 2)
        e = eu + scale * (e - eu)
    xv = _mem(x, (batch, c, hw), (c * hw, hw, 1)).clone()
    px0 = (xv - s1m * e) / math.sqrt(a_t)
    xp = math.sqrt(a_prev) * px0 + math.sqrt(1.0 - a_prev - sigma ** 2) * e
    if noise is not None:
        xp = xp + sigma * _mem(noise, (batch, c, hw), (c * hw, hw, 1))
    _mem(x_prev, (batch, c, hw), (c * hw, hw, 1)).copy_(xp)
    if pred_x0 is not None:
        _mem(pred_x0, (batch, c, hw), (c * hw, hw, 1)).copy_(px0)
    if eps_out is not None:
        _mem(eps_out, (batch, c, hw), (c * hw, hw, 1)).copy_(e)
    return x_prev


class Graph:
    def begin(self):
        self.fn = None

    def end(self):
        pass

    def abort(self):
        pass

    def launch(self):
        raise RuntimeError("the emulator does not capture graphs: run FusedStepRunner with use_graph=False")

    def destroy(self):
        pass


class _Stream:
    def __init__(self, *a, **k):
        pass

    def wait_stream(self, other):
        pass

    def synchronize(self):
        pass

    def wait_event(self, ev):
        pass


class _Event:
    def __init__(self, *a, **k):
        pass

    def record(self, stream=None):
        pass


def install(monkeypatch):
    """Patch magicdance_amd.ops (and the few torch.cuda stream calls of the fused sampler) for a CPU host-logic test."""
    from magicdance_amd import ops, engine
    for name in ("igemm", "ff_block", "ff_block_supported", "attention", "groupnorm_ws_bytes", "groupnorm", "groupnorm_params", "groupnorm_launch", "groupnorm_wants_partials", "layernorm", "nchw_to_nhwc_f16",
                 "nhwc_to_nchw_f32", "add_f16", "image_to_u8", "timestep_embedding", "gemv_f32", "select_row_f32", "gather_rows", "softmax_rows", "counter_add",
                 "ddim_update", "gather_frames", "cfg_scatter_add", "window_mean", "Graph"):
        monkeypatch.setattr(ops, name, globals()[name])
    monkeypatch.setattr(engine, "_require_gpu", lambda device: None)
    monkeypatch.setattr(torch.cuda, "Stream", _Stream)
    monkeypatch.setattr(torch.cuda, "Event", _Event)
    monkeypatch.setattr(torch.cuda, "current_stream", lambda *a, **k: _Stream())
    monkeypatch.setattr(torch.cuda, "stream", lambda s: contextlib.nullcontext())
    engine._ARENAS.clear()
    engine._WS.clear()
