"""DDIM sampler of the reference-attention + pose model.

Mirror of ``DDIMSampler_ReferenceOnly`` (model_lib/ControlNet/ldm/models/diffusion/ddim.py:346-729): same
``make_schedule`` / ``sample`` / ``ddim_sampling`` / ``p_sample_ddim`` signatures and branch structure
(:537-605).  Two execution routes:

  * generic: one ``apply_model`` per branch exactly as the reference orders them (any cond layout, eta > 0,
    callbacks, masks) -- every op still runs in the HIP kernels;
  * fused (the entry points' configuration: "controlnet is more important" CFG branch :595-605, ``wonoise``,
    eta == 0): per step ONE captured HIP graph = appearance net (once, batch 1 when all reference latents are
    equal) + pose ControlNet + the UNet's cond and uncond passes batched as 2B samples (they share all weights)
    + fused CFG/DDIM update; the timestep and schedule coefficients are read from device tables indexed by a
    device-side counter, so 50 steps are 50 graph launches with no host work in between.
"""
import os
import random

import numpy as np
import torch

from . import ops
from .engine import F16, F32


def make_ddim_timesteps(ddim_discr_method, num_ddim_timesteps, num_ddpm_timesteps, verbose=True):
    """ldm/modules/diffusionmodules/util.py:45-59."""
    if ddim_discr_method == "uniform":
        c = num_ddpm_timesteps // num_ddim_timesteps
        ddim_timesteps = np.asarray(list(range(0, num_ddpm_timesteps, c)))
    elif ddim_discr_method == "quad":
        ddim_timesteps = ((np.linspace(0, np.sqrt(num_ddpm_timesteps * .8), num_ddim_timesteps)) ** 2).astype(int)
    else:
        raise NotImplementedError(f'There is no ddim discretization method called "{ddim_discr_method}"')
    return ddim_timesteps + 1


def make_ddim_sampling_parameters(alphacums, ddim_timesteps, eta, verbose=True):
    """util.py:62-73 (alphacums: fp32 numpy/torch cpu array)."""
    alphas = alphacums[ddim_timesteps]
    alphas_prev = np.asarray([alphacums[0]] + alphacums[ddim_timesteps[:-1]].tolist())
    sigmas = eta * np.sqrt((1 - alphas_prev) / (1 - alphas) * (1 - alphas / alphas_prev))
    return sigmas, alphas, alphas_prev


class DDIMSampler_ReferenceOnly(object):
    def __init__(self, model, schedule="linear", **kwargs):
        self.model = model
        self.ddpm_num_timesteps = model.num_timesteps
        self.schedule = schedule

    def make_schedule(self, ddim_num_steps, ddim_discretize="uniform", ddim_eta=0., verbose=True):
        """ddim.py:359-388 (host-side fp32 numpy; the device only sees the per-step coefficient table)."""
        self.ddim_timesteps = make_ddim_timesteps(ddim_discretize, ddim_num_steps, self.ddpm_num_timesteps, verbose)
        ac = self.model.alphas_cumprod.detach().cpu().numpy().astype(np.float32)
        assert ac.shape[0] == self.ddpm_num_timesteps, "alphas have to be defined for each timestep"
        self.ddim_sigmas, self.ddim_alphas, self.ddim_alphas_prev = make_ddim_sampling_parameters(
            ac, self.ddim_timesteps, ddim_eta, verbose)
        self.ddim_sqrt_one_minus_alphas = np.sqrt(1. - self.ddim_alphas)

    @torch.no_grad()
    def sample(self, S, batch_size, shape, conditioning=None, callback=None, normals_sequence=None, img_callback=None,
               quantize_x0=False, eta=0., mask=None, x0=None, temperature=1., noise_dropout=0., score_corrector=None,
               corrector_kwargs=None, verbose=True, x_T=None, log_every_t=100, unconditional_guidance_scale=1.,
               unconditional_conditioning=None, dynamic_threshold=None, ucg_schedule=None, inpaint=None, force_generic=False,
               **kwargs):
        """ddim.py:391-458.  ``force_generic`` (not in the reference): take the per-call route -- one apply_model per branch per
        step, exactly the reference's call structure -- even where the fused table + graph route applies (tests compare the two)."""
        if inpaint is not None or mask is not None or score_corrector is not None or quantize_x0 or dynamic_threshold:
            raise NotImplementedError("inpaint / mask / score-corrector variants are outside the pose hot path")
        self.make_schedule(ddim_num_steps=S, ddim_eta=eta, verbose=verbose)
        C, H, W = shape
        size = (batch_size, C, H, W)
        return self.ddim_sampling(conditioning, size, callback=callback, img_callback=img_callback, x_T=x_T,
                                  log_every_t=log_every_t, temperature=temperature, noise_dropout=noise_dropout,
                                  unconditional_guidance_scale=unconditional_guidance_scale,
                                  unconditional_conditioning=unconditional_conditioning, ucg_schedule=ucg_schedule,
                                  force_generic=force_generic)

    # ------------------------------------------------------------------ loop (ddim.py:461-516)
    @torch.no_grad()
    def ddim_sampling(self, cond, shape, x_T=None, callback=None, img_callback=None, log_every_t=100, temperature=1.,
                      noise_dropout=0., unconditional_guidance_scale=1., unconditional_conditioning=None,
                      ucg_schedule=None, force_generic=False):
        device = self.model.device
        b = shape[0]
        img = torch.randn(shape, device=device) if x_T is None else x_T.detach().to(device=device, dtype=F32)
        timesteps = self.ddim_timesteps
        intermediates = {"x_inter": [img], "pred_x0": [img]}
        time_range = np.flip(timesteps)
        total_steps = timesteps.shape[0]

        if (not force_generic and ucg_schedule is None and noise_dropout == 0.
                and self._fused_ok(cond, unconditional_conditioning, unconditional_guidance_scale)
                and self._table_fits(cond, unconditional_conditioning, shape)):
            return self._fused_sampling(cond, img, unconditional_guidance_scale, callback, img_callback, log_every_t,
                                        intermediates, unconditional_conditioning)

        for i, step in enumerate(time_range):
            index = total_steps - i - 1
            ts = torch.full((b,), int(step), device=device, dtype=torch.long)
            if ucg_schedule is not None:
                assert len(ucg_schedule) == len(time_range)
                unconditional_guidance_scale = ucg_schedule[i]
            img, pred_x0 = self.p_sample_ddim(img, cond, ts, index=index, temperature=temperature,
                                              noise_dropout=noise_dropout,
                                              unconditional_guidance_scale=unconditional_guidance_scale,
                                              unconditional_conditioning=unconditional_conditioning)
            if callback:
                callback(i)
            if img_callback:
                img_callback(pred_x0, i)
            if index % log_every_t == 0 or index == total_steps - 1:
                intermediates["x_inter"].append(img)
                intermediates["pred_x0"].append(pred_x0)
        return img, intermediates

    # ------------------------------------------------------------------ one step, generic (ddim.py:519-645)
    @torch.no_grad()
    def p_sample_ddim(self, x, c, t, index, repeat_noise=False, temperature=1., noise_dropout=0.,
                      unconditional_guidance_scale=1., unconditional_conditioning=None):
        model, device = self.model, self.model.device
        b = x.shape[0]
        reference_image_noisy = None
        if c.get("image_control") is not None:
            start = torch.cat(c["image_control"], 1)
            reference_image_noisy = start if c["wonoise"] else model.q_sample(start.to(device), t)   # :529-535
        uc = unconditional_conditioning
        eps_u = None
        if uc is None or unconditional_guidance_scale == 1.:
            eps_c = model.apply_model_nhwc(x, t, c, reference_image_noisy)                           # :537-538
        elif uc.get("image_control") is not None:                                                    # "balance" :540-567
            x_in, t_in = torch.cat([x] * 2), torch.cat([t] * 2)
            ref_in = torch.cat([reference_image_noisy] * 2)
            c_in = dict()
            for k in c:
                if isinstance(c[k], list):
                    c_in[k] = [torch.cat([uc[k][i], c[k][i]]) for i in range(len(c[k]))]
                else:
                    c_in[k] = c[k]
            eps_u, eps_c = model.apply_model_nhwc(x_in, t_in, c_in, ref_in).chunk(2)
        elif c.get("overlap_sampling"):                                                              # :569-594
            # temporal overlap sampling: windows of 16 frames, stride 12, starting at a random frame offset (python
            # ``random``, as the reference draws it); classifier-free guidance per window, predictions accumulated per frame
            # and divided by the visit count.  Stays on the device (the reference round-trips through the CPU, :570-592).
            nf = c["c_concat"][0].shape[0]
            offset = random.randint(0, nf - 1)
            hw = x.shape[2] * x.shape[3]
            pred_all = torch.zeros((nf, hw, 4), dtype=F32, device=device)
            counts = torch.zeros((nf,), dtype=F32, device=device)
            for start_idx in range(offset, offset + nf - 16 + 1 + 12, 12):
                idx = (torch.arange(start_idx, start_idx + 16) % nf).to(device)
                c_w = dict(c)
                c_w["c_concat"] = [c["c_concat"][0][idx].contiguous()]
                ref_w = reference_image_noisy if reference_image_noisy.shape[0] != nf else reference_image_noisy[idx]
                x_w, t_w = x[idx].contiguous(), (t[idx] if t.shape[0] == nf else t[:16])
                m_t = model.apply_model_nhwc(x_w, t_w, c_w, ref_w).clone()
                m_u = model.apply_model_nhwc(x_w, t_w, c_w, None, uc=True)
                pred_all.index_add_(0, idx, (m_u + unconditional_guidance_scale * (m_t - m_u))[..., :4])
                counts.index_add_(0, idx, torch.ones_like(idx, dtype=F32))
            eps_c, eps_u = (pred_all / counts.reshape(-1, 1, 1)).contiguous(), None   # guidance already applied
        else:                                                                                        # :595-605
            eps_c = model.apply_model_nhwc(x, t, c, reference_image_noisy).clone()   # arena is rewound below
            eps_u = model.apply_model_nhwc(x, t, c, None, uc=True)
        coef = torch.tensor([self.ddim_alphas[index], self.ddim_alphas_prev[index], self.ddim_sigmas[index],
                             self.ddim_sqrt_one_minus_alphas[index], unconditional_guidance_scale],
                            dtype=F32, device=device)
        sigma = float(self.ddim_sigmas[index])
        noise = None
        if sigma != 0.0:
            noise = torch.randn_like(x) if not repeat_noise else torch.randn_like(x[:1]).repeat(b, 1, 1, 1)
            noise = (noise * temperature).contiguous()
            if noise_dropout > 0.:
                noise = torch.nn.functional.dropout(noise, p=noise_dropout)
        hw = x.shape[2] * x.shape[3]
        cch = x.shape[1]
        x = x.detach().to(device=device, dtype=F32).contiguous()
        x_prev, pred_x0 = torch.empty_like(x), torch.empty_like(x)
        ops.ddim_update(eps_c, eps_u, eps_c.shape[-1], x, noise, coef, x_prev, pred_x0, None, b, cch, hw)
        return x_prev, pred_x0

    # ------------------------------------------------------------------ fused route
    def _fused_ok(self, c, uc, scale):
        """The fused (table + captured graph) route covers eta = 0 sampling of: the entry points' default -- "controlnet is more
        important" CFG (:595-605) with ``wonoise`` --, the "balance" CFG branch (:540-567: the unconditional dict carries the
        reference too), the noisy reference (``wonoise`` False, :529-535), the stage-1 model without pose ControlNet and the temporal
        ``overlap_sampling`` windows (:569-594).  Everything else (eta > 0, a void text context, guidance scale 1, overlap windows over
        per-frame references) takes the generic per-call route."""
        if uc is None or scale == 1. or not isinstance(c, dict) or not isinstance(uc, dict):
            return False
        if c.get("image_control") is None or c.get("c_crossattn_void") is not None:
            return False
        if getattr(self.model, "has_pose", True) and c.get("c_concat") is None:
            return False
        if c.get("overlap_sampling"):
            # temporal overlap windows (:569-594) inside the step graph (round 4): the entry points' form -- pose model, clean shared
            # reference, shared text, at least one full window of frames; anything else keeps the per-call route
            one = lambda ts: FusedStepRunner._same_rows(ts)  # noqa: E731
            if not (getattr(self.model, "has_pose", True) and c.get("wonoise", True) and uc.get("image_control") is None
                    and c["c_concat"][0].shape[0] >= FusedStepRunner.OV_WIN and one(c["image_control"]) and one(c["c_crossattn"])):
                return False
        if uc.get("image_control") is not None:   # balance: both halves of the 2B batch are conditional passes
            for k in c:
                if isinstance(c[k], list) and (k not in uc or len(uc[k]) != len(c[k])):
                    return False
        return float(np.abs(self.ddim_sigmas).max()) == 0.0

    # Reference-KV table budget of the fused route.  The table holds, per DDIM step and per reference sample, the projected K / V^T
    # of all 16 bank entries (46 MB fp16 at 512 x 512): 2.3 GB for the entry points' form (ONE shared reference, 50 steps).  The
    # balance form keeps one reference row per sample of the 2b batch and the noisy form one per frame, so their tables grow with
    # the batch (8 frames, balance: 37 GB).  Beyond the budget the per-call route runs instead (no table) -- same results.
    TABLE_BUDGET_BYTES = int(float(os.environ.get("MD_TABLE_BUDGET_GB", "64")) * (1 << 30))

    def _table_bytes(self, c, uc, shape):
        from . import engine
        from .nets import bank_shapes
        b, _, hh, ww = shape
        if uc is not None and uc.get("image_control") is not None:
            bref = 2 * b                                  # balance
        elif not c.get("wonoise", True):
            bref = b                                      # noisy reference: one bank per frame
        else:
            bref = 1 if FusedStepRunner._same_rows(c["image_control"]) else c["image_control"][0].shape[0]
        app = self.model.engines()[0]
        row = sum(n * ch + ch * engine.kv_ld(n) for n, ch in bank_shapes(app.cfg, (hh, ww)))
        return self.ddim_timesteps.shape[0] * bref * row * (1 if engine.ATTN_FP8 else 2)

    def _table_fits(self, c, uc, shape):
        """the table fits the configured budget AND what the device has free right now (an already allocated table of the same size
        counts as free: it is reused) -- otherwise the per-call route runs instead of an allocation failure"""
        need = self._table_bytes(c, uc, shape)
        budget = self.TABLE_BUDGET_BYTES
        dev = getattr(self.model, "device", None)
        if dev is not None and dev.type == "cuda" and torch.cuda.is_available():
            st = getattr(self.model, "_fused", None)
            held = 0 if st is None or getattr(st, "bank_table", None) is None else st.bank_table.numel() * st.bank_table.element_size()
            budget = min(budget, int(0.8 * torch.cuda.mem_get_info(dev)[0]) + held)
        return need <= budget

    def _fused_sampling(self, c, img, scale, callback, img_callback, log_every_t, intermediates, uc=None):
        model = self.model
        st = model._fused
        if st is None:
            st = model._fused = FusedStepRunner(model)
        total = self.ddim_timesteps.shape[0]
        # graph capture is illegal on the legacy default stream: the fused route runs on its own stream
        caller = torch.cuda.current_stream()
        st.stream.wait_stream(caller)
        with torch.cuda.stream(st.stream):
            st.prepare(c, img, self, scale, table_mode=True, uc=uc)

            def on_step(i):
                index = total - i - 1
                if callback:
                    callback(i)
                if img_callback:
                    img_callback(st.pred_x0.clone(), i)
                if index % log_every_t == 0 or index == total - 1:
                    intermediates["x_inter"].append(st.x.clone())
                    intermediates["pred_x0"].append(st.pred_x0.clone())
            st.run_steps(on_step=on_step)
            out = st.x.clone()
        caller.wait_stream(st.stream)
        return out, intermediates


class FusedStepRunner:
    """Owns the persistent device buffers of the fused step and its captured HIP graph (re-used across frames and
    sample_log calls while shapes / context / step count stay the same).

    With ``wonoise`` the appearance bank depends on (ref, t, ctx) only, never on the frame or on x_t, so the banks of all S
    steps are computed BEFORE the loop (the reference-KV table): the appearance net runs on batches of ``bank_chunk``
    timesteps at once (big, MFMA-efficient launches instead of S batch-1 passes) and the UNet's own to_k / to_v of every bank
    entry are applied there too.  ``bank_table`` holds, per bank entry, K [S, bref, n, c] and V^T [S, bref, c, ldv] (46 MB fp16
    per step at 512x512, 2.3 GB for 50 steps); the step graph gathers row ``counter`` into ``bank_cur`` with one
    md_gather_rows launch and contains neither the appearance net nor the bank projections.  The same table serves multi-GPU
    frame sharding (equal row blocks computed per rank and exchanged with RCCL all-gathers, magicdance_amd/parallel.py) and
    multi-frame sequences sharing one reference image.
    """

    OV_WIN, OV_STRIDE = 16, 12      # overlap_sampling: frames per window / window stride (ddim.py:577)

    def __init__(self, model):
        self.model = model
        self.key = None
        self.graph = None
        self.use_graph = True
        self.table_mode = True
        self.stream = torch.cuda.Stream(device=model.device)
        self.bank_chunk = 16      # appearance samples (timesteps x reference latents) per batched table pass
        self.table_stream = torch.cuda.Stream(device=model.device)      # the table pass overlaps the first steps of the loop
        # The pose ControlNet as extra samples of the UNet's encoder launches (NetEngine.unet_pose; default).  MD_MERGE_POSE=0: its
        # own ~110 launches on a concurrent stream (the round-1 / early round-2 form; tests check the two against each other).
        self.merge_pose = os.environ.get("MD_MERGE_POSE", "1") == "1"
        self.table_chunks = 2     # sharded: all-gathers per table (the 2nd overlaps the loop)
        self.tkey = None
        self.pose_stream = torch.cuda.Stream(device=model.device)

    # Facts about the CONTENT of a conditioning tensor (are all its rows equal? is it the tensor the caches were built from?) cost a
    # device -> host synchronisation when computed.  They are memoised on the IDENTITY of the tensor object(s) -- (id, version counter,
    # address, shape) with a strong reference held, so an id cannot be recycled -- and the entry points / the bench pass the same
    # objects for every batch of a sequence: after the first batch no call on this route blocks on the device.
    _FACTS = {}

    @staticmethod
    def _ident(ts):
        ts = ts if isinstance(ts, (list, tuple)) else [ts]
        return tuple((id(t), t._version, t.data_ptr(), tuple(t.shape), t.dtype) for t in ts)

    @classmethod
    def _memo(cls, ts, name, fn):
        key = (name,) + cls._ident(ts)
        hit = cls._FACTS.get(key)
        if hit is not None:
            return hit[0]
        if len(cls._FACTS) > 256:
            cls._FACTS.clear()
        val = fn()
        cls._FACTS[key] = (val, list(ts) if isinstance(ts, (list, tuple)) else [ts])   # (keeps the tensors alive: ids stay unique)
        return val

    @classmethod
    def _same_rows(cls, ts):
        """every row (dim 0) of cat(ts, 1) equals the first one -- memoised on the identity of ``ts`` (a tensor or a list)"""
        lst = ts if isinstance(ts, (list, tuple)) else [ts]
        if all(t.shape[0] == 1 for t in lst):
            return True
        return cls._memo(ts, "same_rows", lambda: all(bool((t[1:] == t[:1]).all().item()) for t in lst))

    def plan_table(self, S, bref, world=1, sharded=None):
        """(rows per block, blocks) of the reference-KV table.  A block = ``per`` consecutive DDIM rows = one contiguous piece
        of memory: what one batched appearance pass produces and, sharded, what one rank contributes to a chunk's all-gather
        (chunk k = blocks [k*world, (k+1)*world), block k*world + r computed by rank r).  world 1: per = the appearance batch
        (``bank_chunk`` samples); world > 1: S rows split into ``table_chunks`` all-gathers so that the second one overlaps the
        first steps of the loop."""
        sharded = world > 1 if sharded is None else sharded   # (a 1-rank group may run the sharded form: tests, --gpus 1 RCCL checks)
        if not sharded:
            per = max(1, min(S, self.bank_chunk // bref))
        else:
            per = max(1, -(-S // (world * max(1, self.table_chunks))))
        nblocks = -(-(-(-S // per)) // world) * world
        return per, nblocks

    def prepare(self, c, x_T, sampler, scale, table_mode=True, world=1, sharded=None, uc=None):
        """Per-batch buffers (x, pose features, schedule tables; keyed on the batch geometry) and the reference-KV table (keyed on
        the reference / schedule geometry only, so a sequence sampled in batches of different sizes keeps one table).

        The step's UNet batch is always 2b samples [x, x].  Forms (``uc``: the unconditional dict, None = the default form):
          default  first half cond (text c, bank, pose residuals), second half uncond (text c, neither)   ddim.py:595-605
          balance  first half text uc, second half text c; BOTH halves read the bank and take pose residuals  ddim.py:540-567
                   (appearance net / ControlNet see the same 2b-sample batch, so the bank is per sample: bref = 2b)
          noisy    default, but the appearance net sees q_sample(reference, t) with fresh noise per step and frame (:529-535): the
                   table rows are computed from per-step noisy references drawn HERE, in step order (bref = b)
          stage 1  no pose ControlNet (ControlLDMReferenceOnly)"""
        assert table_mode, "the fused route always runs from the reference-KV table"
        model, dev = self.model, self.model.device
        app, pose_e, unet = model.engines()
        nf, cch, hh, ww = x_T.shape
        # overlap_sampling (ddim.py:569-594): the state x holds all nf frames, the networks run on windows of OV_WIN frames picked by a
        # per-step index table (python ``random`` offsets drawn HERE, in step order, exactly as the per-step route draws them)
        overlap = bool(c.get("overlap_sampling"))
        b = self.OV_WIN if overlap else nf
        balance = uc is not None and uc.get("image_control") is not None
        noisy = not c.get("wonoise", True)
        has_pose = pose_e is not None
        assert not overlap or (has_pose and not balance and not noisy and nf >= self.OV_WIN)
        ref = torch.cat(c["image_control"], 1).detach().to(device=dev, dtype=F32)
        ctx = torch.cat(c["c_crossattn"], 1).detach().to(device=dev, dtype=F32)
        rep = lambda t, n: t if t.shape[0] == n else t.expand(n, *t.shape[1:])  # noqa: E731
        ref_orig = ref   # the reference at the batch the caller gave it (1, or one per frame): what q_sample draws its noise for
        if balance:
            ctx_u = torch.cat(uc["c_crossattn"], 1).detach().to(device=dev, dtype=F32)
            base = rep(ref, b)     # (:529-551: both halves see the CONDITIONAL dict's reference; uc's only switches the branch)
            ref = torch.cat([base, base], 0)                                     # x_in / ref_in = [uc half, c half]
            ctx = torch.cat([rep(ctx_u, b), rep(ctx, b)], 0)
            ctx_app = ctx_unet = ctx
            nread, n_pose = 2 * b, (2 * b if has_pose else 0)
        else:
            if noisy:
                ref = base = rep(ref, b)      # per-frame noise: one bank per frame
            elif ref.shape[0] > 1 and self._same_rows(c["image_control"]):
                ref = ref[:1]          # every frame shares the reference latent: one appearance pass, bank broadcast
            if ctx.shape[0] > 1 and self._same_rows(c["c_crossattn"]):
                ctx = ctx[:1]
            ctx_app = ctx if ctx.shape[0] in (1, ref.shape[0]) else ctx[:ref.shape[0]]
            ctx_unet = ctx if ctx.shape[0] == 1 else torch.cat([ctx, ctx], 0)
            nread, n_pose = b, (b if has_pose else 0)
        ref = ref.contiguous()
        bref = ref.shape[0]
        self.balance, self.nread, self.n_pose = balance, nread, n_pose
        if has_pose:
            # (a single pose tensor is passed on as the caller's OBJECT: hint_features recognises it by identity + version and skips
            # the content comparison -- a device -> host synchronisation -- that a fresh torch.cat copy would need)
            hint = c["c_concat"][0] if len(c["c_concat"]) == 1 else torch.cat(c["c_concat"], 1)
            if balance:
                hint = torch.cat([rep(torch.cat(uc["c_concat"], 1), b), rep(hint, b)], 0)
        else:
            hint = torch.zeros((0,), device=dev)
        S = sampler.ddim_timesteps.shape[0]
        # appearance passes batch ``per_pass`` timesteps x bref references: a per-sample text context repeats with the references
        per_pass = max(1, self.bank_chunk // bref)
        if ctx_app.shape[0] > 1:
            ctx_app = ctx_app.repeat(per_pass, 1, 1)
        # the context K/V caches are keyed on the tensor passed in; keep stable tensors across calls
        # (identity first: the same conditioning objects as last time -> nothing to compare, no synchronisation; fresh objects ->
        # one content comparison, and only a different content rebuilds the K / V caches)
        ident = (self._ident(c["c_crossattn"]), self._ident(uc["c_crossattn"]) if balance else None, balance, tuple(ctx_app.shape))
        ckey = None
        same = getattr(self, "_ctx_ident", None) == ident
        if not same and getattr(self, "_ctx_src", None) is not None and self._ctx_app.shape == ctx_app.shape:
            ckey = torch.cat([ctx_unet.reshape(-1), ctx_app.reshape(-1)[:1]])
            same = self._ctx_src.shape == ckey.shape and torch.equal(self._ctx_src, ckey)
        self._ctx_ident, self._ctx_keep = ident, (list(c["c_crossattn"]), list(uc["c_crossattn"]) if balance else None)
        if not same:
            self._ctx_src = (torch.cat([ctx_unet.reshape(-1), ctx_app.reshape(-1)[:1]]) if ckey is None else ckey).clone()
            self._ctx_app, self._ctx_unet = ctx_app.contiguous().clone(), ctx_unet.contiguous().clone()
            self._ctx_pose = (self._ctx_unet if balance else (ctx if ctx.shape[0] in (1, b) else ctx[:b])).contiguous().clone()
        self.kv_app = app.context_kv(self._ctx_app)
        self.kv_unet = unet.context_kv(self._ctx_unet)
        if has_pose:
            self.kv_pose = pose_e.context_kv(self._ctx_pose)
            self.kv_merged = unet.merged_context_kv(self.kv_unet, self.kv_pose, b, n_pose) if self.merge_pose else None
        else:
            self.kv_pose = self.kv_merged = None
        self.kv_unet_uc = self.kv_unet if self._ctx_unet.shape[0] == 1 else [
            (k[:b], vt[:b], b, tk, ldv) for (k, vt, bc, tk, ldv) in self.kv_unet]  # per-sample text: first half of the 2B batch
        key = (b, nf, overlap, cch, hh, ww, S, bref, tuple(hint.shape), balance, noisy, self.kv_app[0][0].data_ptr(),
               0 if self.kv_pose is None else self.kv_pose[0][0].data_ptr(), self.kv_unet[0][0].data_ptr())
        if key != self.key:
            self._allocate(key, b, cch, hh, ww, S, bref, True, nf=nf, overlap=overlap)
        per, nblocks = self.plan_table(S, bref, world, sharded)
        from . import engine as _eng
        tkey = (cch, hh, ww, S, bref, per, nblocks, _eng.ATTN_FP8)
        if tkey != self.tkey:
            self._allocate_table(tkey, hh, ww, S, bref, per, nblocks)
        self.ref.copy_(ref)
        self.x.copy_(x_T)
        if has_pose:
            hf = pose_e.hint_features(hint)
            if overlap:   # features of ALL frames stay resident; a window's rows are gathered into hint_feat inside the step
                if self.hint_all is None or self.hint_all.t.shape != hf.t.shape:
                    self.hint_all = type(hf)(torch.empty_like(hf.t), hf.b, hf.h, hf.w, hf.c)
                    self.hint_feat = type(hf)(torch.empty((b,) + tuple(hf.t.shape[1:]), dtype=hf.t.dtype, device=dev), b, hf.h, hf.w, hf.c)
                    self._drop_graph()
                self.hint_all.t.copy_(hf.t)
            else:
                if self.hint_feat is None or self.hint_feat.t.shape != hf.t.shape:
                    self.hint_feat = type(hf)(torch.empty_like(hf.t), hf.b, hf.h, hf.w, hf.c)
                    self._drop_graph()
                self.hint_feat.t.copy_(hf.t)
        if overlap:
            offs = [random.randint(0, nf - 1) for _ in range(S)]                                   # :572, one draw per step
            starts = [list(range(o, o + nf - self.OV_WIN + 1 + self.OV_STRIDE, self.OV_STRIDE)) for o in offs]   # :577
            tab = np.array([[[(s0 + j) % nf for j in range(self.OV_WIN)] for s0 in st] for st in starts], dtype=np.int32)
            assert tab.shape == tuple(self.ov_idx.shape), (tab.shape, self.ov_idx.shape)
            self.ov_idx.copy_(torch.from_numpy(tab))
            self.ov_pred.zero_()
            self.ov_counts.zero_()
        # per-step tables: timestep (as float, repeated for the 2B-sample UNet batch) and DDIM coefficients
        steps = np.flip(sampler.ddim_timesteps).astype(np.float32)
        idx = np.arange(S)[::-1]
        coef = np.stack([sampler.ddim_alphas[idx], sampler.ddim_alphas_prev[idx], sampler.ddim_sigmas[idx],
                         sampler.ddim_sqrt_one_minus_alphas[idx], np.full(S, scale)], 1).astype(np.float32)
        self.ts_table.copy_(torch.from_numpy(np.repeat(steps[:, None], self.ts_table.shape[1], 1).copy()))
        self.coef_table.copy_(torch.from_numpy(coef))
        self.counter.zero_()
        # noisy reference: q_sample(reference, t) per step, the noise drawn in step order exactly as the per-step route draws it --
        # ONE randn_like(cond_image_start) per step AT THE REFERENCE'S OWN BATCH (ddpm.py:356-359 through ddim.py:529-535): a batch-1
        # reference shared by b frames gets one noise tensor per step, broadcast over the frames by the [b] timestep factors (and
        # consumes one draw of its size from the RNG, not b) -- the table rows are built from these
        self.ref_rows = None
        if noisy:
            tl = torch.from_numpy(np.flip(sampler.ddim_timesteps).copy()).to(dev).long()
            rows = [rep(model.q_sample(ref_orig.contiguous(), tl[i].expand(b)), b) for i in range(S)]
            self.ref_rows = torch.stack([torch.cat([r, r], 0) if balance else r for r in rows]).contiguous()
        # Time-embedding tables: timestep_embedding -> time_embed MLP -> every ResBlock's emb_layers depend on the step only, not on
        # x or the frame: computed here for all S steps at once (same kernels, rows = steps), so that a step reads ONE row per
        # network (md_select_row_f32) instead of running 4 dependent launches per network at the head of its critical path.
        # (the tables and the current-row buffers are PERSISTENT: the captured step graph holds their addresses)
        arena = unet.arena
        for name, eng in (("unet", unet),) + ((("pose", pose_e),) if has_pose else ()):
            arena.reset()
            tab = eng.time_embedding(self.ts_table[:, 0].contiguous(), S)
            old = getattr(self, "emb_table_" + name, None)
            if old is None or old.shape != tab.shape:
                setattr(self, "emb_table_" + name, torch.empty_like(tab))
                setattr(self, "emb_cur_" + name, torch.empty((1, tab.shape[1]), dtype=F32, device=dev))
                self._drop_graph()
            getattr(self, "emb_table_" + name).copy_(tab)

    def _drop_graph(self):
        if self.graph is not None:
            self.graph.destroy()
            self.graph = None

    def _allocate(self, key, b, cch, hh, ww, S, bref, table_mode, nf=None, overlap=False):
        dev = self.model.device
        self._drop_graph()
        nf = b if nf is None else nf
        self.key, self.table_mode = key, table_mode
        self.b, self.cch, self.hw, self.S, self.nf, self.overlap = b, cch, hh * ww, S, nf, overlap
        self.x = torch.empty((nf, cch, hh, ww), dtype=F32, device=dev)      # the DDIM state (all frames)
        self.pred_x0 = torch.empty_like(self.x)
        self.xw = self.x                                                     # what the networks see (overlap: one window of frames)
        self.hint_all = None
        if overlap:
            nwin = len(range(0, nf - self.OV_WIN + 1 + self.OV_STRIDE, self.OV_STRIDE))
            self.xw = torch.empty((b, cch, hh, ww), dtype=F32, device=dev)
            self.ov_idx = torch.zeros((S, nwin, self.OV_WIN), dtype=torch.int32, device=dev)
            self.ov_pred = torch.zeros((nf, hh * ww, cch), dtype=F32, device=dev)       # guided eps accumulated per frame (NHWC)
            self.ov_counts = torch.zeros((nf,), dtype=F32, device=dev)
            self.ov_eps = torch.empty((nf, hh * ww, cch), dtype=F32, device=dev)
        self.ref = torch.empty((bref, cch, hh, ww), dtype=F32, device=dev)
        self.hint_feat = None
        self.ts_table = torch.empty((S, 2 * b), dtype=F32, device=dev)
        self.coef_table = torch.empty((S, 5), dtype=F32, device=dev)
        self.t_cur = torch.empty((2 * b,), dtype=F32, device=dev)
        self.coef_cur = torch.empty((5,), dtype=F32, device=dev)
        self.counter = torch.zeros((1,), dtype=torch.int32, device=dev)

    def _allocate_table(self, tkey, hh, ww, S, bref, per, nblocks):
        """Reference-KV table, layout [row block][segment][per rows][row length]; segment 2e = K of bank entry e
        ([bref, n, c] per row), 2e + 1 = its V^T ([bref, c, ldv] per row, pad columns stay zero)."""
        from . import engine
        from .engine import BankKV
        from .nets import bank_shapes
        dev = self.model.device
        self._drop_graph()
        self.tkey, self.per, self.nblocks = tkey, per, nblocks
        app = self.model.engines()[0]
        tdtype = torch.uint8 if engine.ATTN_FP8 else F16      # fp8 attention path: the table holds e4m3 bytes (half the size)
        u = 16 // (1 if engine.ATTN_FP8 else 2)               # elements per 16-byte unit of md_gather_rows
        self.table_unit = u
        self.bank_geo, segs, boff, coff = [], [], 0, 0
        for n, c in bank_shapes(app.cfg, (hh, ww)):
            ldv = engine.kv_ld(n)
            lk, lv = bref * n * c, bref * c * ldv
            assert lk % u == 0 and lv % u == 0
            self.bank_geo.append((boff, boff + per * lk, coff, coff + lk, n, c, ldv))
            segs += [(boff // u, lk // u, coff // u), ((boff + per * lk) // u, lv // u, (coff + lk) // u)]
            boff += per * (lk + lv)
            coff += lk + lv
        self.block_elems = boff
        self.bank_table = torch.zeros((nblocks * boff,), dtype=tdtype, device=dev)   # zeros: the V^T pad columns stay zero
        self.bank_cur = torch.zeros((coff,), dtype=tdtype, device=dev)
        self.bank_seg = torch.tensor(segs, dtype=torch.int64, device=dev)
        self.bank_seg_max = max(sg[1] for sg in segs)
        self.bank_cur_kv = [BankKV(self.bank_cur[ck:ck + bref * n * c].view(bref, n, c),
                                   self.bank_cur[cv:cv + bref * c * ldv].view(bref, c, ldv), bref, n, c, ldv)
                            for (_, _, ck, cv, n, c, ldv) in self.bank_geo]
        self._bank_tmp = None

    def table_block(self, blk, count=1):
        """flat view of ``count`` consecutive row blocks (contiguous memory)"""
        return self.bank_table[blk * self.block_elems:(blk + count) * self.block_elems]

    def _row_views(self, e, r0, tc):
        """K [tc*bref, n, c] and V^T [tc*bref, c, ldv] of bank entry e for DDIM rows [r0, r0 + tc) (inside ONE block)."""
        koff, voff, _, _, n, c, ldv = self.bank_geo[e]
        bref = self.ref.shape[0]
        lk, lv = bref * n * c, bref * c * ldv
        blk, within = divmod(r0, self.per)
        assert within + tc <= self.per
        base = blk * self.block_elems
        return (self.bank_table[base + koff + within * lk:base + koff + (within + tc) * lk].view(tc * bref, n, c),
                self.bank_table[base + voff + within * lv:base + voff + (within + tc) * lv].view(tc * bref, c, ldv))

    def compute_bank_rows(self, rows):
        """Fill the reference-KV table for DDIM steps ``rows`` (indices into the flipped timestep order): the appearance
        net runs on up to ``bank_chunk`` timesteps per pass (sample = (step, ref) pair, per-sample time embedding), then the
        UNet's to_k / to_v project each of the 16 bank tensors for the whole pass straight into the table.
        Runs on the current stream with its own arena and split-K workspaces, so it may overlap a step graph that reads
        rows already finished."""
        from .engine import Act, get_arena
        app, _, unet = self.model.engines()
        rows = list(rows)
        bref = self.ref.shape[0]
        per_pass = max(1, self.bank_chunk // bref)
        step_arena, app.arena = app.arena, get_arena(self.x.device, "table")
        app.ws_slot, unet.ws_slot = 3, 3
        try:
            i = 0
            while i < len(rows):
                j = i + 1
                while (j < len(rows) and j - i < per_pass and rows[j] == rows[j - 1] + 1
                       and rows[j] // self.per == rows[i] // self.per):
                    j += 1
                r0, tc = rows[i], j - i
                i = j
                nb = tc * bref
                if self._bank_tmp is None or self._bank_tmp[0] < nb:
                    self._bank_tmp = (per_pass * bref, [torch.empty((per_pass * bref, n, c), dtype=F16, device=self.x.device)
                                                        for (_, _, _, _, n, c, _) in self.bank_geo])
                tmp = [Act(t[:nb], nb, 1, t.shape[1], t.shape[2]) for t in self._bank_tmp[1]]
                app.arena.reset()
                t_dev = self.ts_table[r0:r0 + tc, 0].repeat_interleave(bref).contiguous()
                if self.ref_rows is not None:
                    x = self.ref_rows[r0:r0 + tc].reshape(nb, *self.ref.shape[1:])
                else:
                    x = self.ref.repeat(tc, 1, 1, 1) if tc > 1 else self.ref
                kv = self.kv_app if self.kv_app[0][2] in (1, nb) else [(k[:nb], vt[:nb], nb, tk, ldv) for (k, vt, _, tk, ldv) in self.kv_app]
                app.appearance(x, t_dev, kv, bank_out=tmp)
                for e in range(len(self.bank_geo)):
                    k_out, vt_out = self._row_views(e, r0, tc)
                    unet.project_bank(e, tmp[e], k_out, vt_out)
        finally:
            app.arena = step_arena
            app.ws_slot, unet.ws_slot = 0, 0

    # ------------------------------------------------------------------ table chunks / step loop
    def n_chunks(self, world=1):
        return self.nblocks // world

    def chunk_rows(self, k, world=1):
        r0 = k * world * self.per
        return min(self.S, r0), min(self.S, r0 + world * self.per)

    def enqueue_chunk(self, k, rank=0, world=1, group=None, sharded=None):
        """Chunk k of the reference-KV table on the current stream: this rank's block (one batched appearance pass + the bank
        K / V^T projections), then -- sharded -- ONE RCCL all-gather of the chunk's ``world`` contiguous blocks, in place (the
        send buffer is this rank's block inside the receive buffer; every point-to-point xGMI link carries a different block at
        the same time).  Every rank issues the same collectives in the same order, whatever its own frame count."""
        blk = k * world + rank
        r0, r1 = min(self.S, blk * self.per), min(self.S, (blk + 1) * self.per)
        if r1 > r0:
            self.compute_bank_rows(range(r0, r1))
        if world > 1 if sharded is None else sharded:
            import torch.distributed as dist
            dist.all_gather_into_tensor(self.table_block(k * world, world), self.table_block(blk), group=group)

    def run_steps(self, rank=0, world=1, group=None, on_step=None, fill=True, steps=True, sharded=None):
        """The S-step loop on ``self.stream`` (current).  In table mode the table is (``fill``) produced chunk by chunk on
        ``table_stream`` AHEAD of the loop: chunk k + 1 (appearance pass and, sharded, its all-gather) is enqueued before the
        step graphs of chunk k, which wait on chunk k's event -- step i only needs row i, and the big launches of the table pass
        fill the CUs that the small launches of a step leave idle.  ``steps=False``: fill only (a rank with no frames still takes
        part in the collectives)."""
        S = self.S
        if not fill:
            for i in range(S if steps else 0):
                self.step()
                if on_step:
                    on_step(i)
            return

        def enqueue(k):
            with torch.cuda.stream(self.table_stream):
                self.enqueue_chunk(k, rank, world, group, sharded)
                ev = torch.cuda.Event()
                ev.record(self.table_stream)
            return ev

        self.table_stream.wait_stream(torch.cuda.current_stream())   # earlier steps are done with the table; inputs in place
        nch = self.n_chunks(world)
        ev = enqueue(0)
        for k in range(nch):
            ev_next = enqueue(k + 1) if k + 1 < nch else None
            if ev is not None:
                torch.cuda.current_stream().wait_event(ev)
            ev = ev_next
            r0, r1 = self.chunk_rows(k, world)
            for i in range(r0, r1 if steps else r0):
                self.step()
                if on_step:
                    on_step(i)

    def _launch_sequence(self):
        """One DDIM step as a fixed launch sequence on fixed addresses."""
        model = self.model
        app, pose_e, unet = model.engines()
        b = self.b
        ops.select_row_f32(self.ts_table, self.counter, 0, self.t_cur, 2 * b, self.S)
        ops.select_row_f32(self.coef_table, self.counter, 0, self.coef_cur, 5, self.S)
        ops.select_row_f32(self.emb_table_unet, self.counter, 0, self.emb_cur_unet, self.emb_cur_unet.shape[1], self.S)
        if pose_e is not None:
            ops.select_row_f32(self.emb_table_pose, self.counter, 0, self.emb_cur_pose, self.emb_cur_pose.shape[1], self.S)
        arena = unet.arena
        arena.reset()
        main = torch.cuda.current_stream()
        oc = unet.cfg.out_channels
        ops.gather_rows(self.bank_table, self.bank_seg, self.bank_seg.shape[0], self.bank_seg_max, self.counter, 0,
                        self.bank_cur, self.S, self.per, self.block_elems // self.table_unit)
        if self.overlap:
            # ddim.py:569-594: every window of 16 frames through the networks, guided predictions accumulated per frame, mean over the
            # visits, ONE DDIM update of all frames.  Windows run one after the other on this stream (they may share frames).
            for w in range(self.ov_idx.shape[1]):
                arena.reset()
                ops.gather_frames(self.x, self.xw, self.ov_idx, self.counter, w, self.cch * self.hw * 4)
                ops.gather_frames(self.hint_all.t, self.hint_feat.t, self.ov_idx, self.counter, w,
                                  self.hint_all.t[0].numel() * self.hint_all.t.element_size())
                eps_c, eps_u = self._networks(model, pose_e, unet, b, main)
                ops.cfg_scatter_add(eps_c, eps_u, oc, self.coef_cur, self.ov_idx, self.counter, w, self.ov_pred, self.ov_counts,
                                    self.hw, self.cch)
            ops.window_mean(self.ov_pred, self.ov_counts, self.ov_eps, self.nf, self.hw * self.cch)
            ops.ddim_update(self.ov_eps, None, self.cch, self.x, None, self.coef_cur, self.x, self.pred_x0, None, self.nf, self.cch, self.hw)
        else:
            eps_c, eps_u = self._networks(model, pose_e, unet, b, main)
            ops.ddim_update(eps_c, eps_u, oc, self.x, None, self.coef_cur, self.x, self.pred_x0, None, b, self.cch, self.hw)
        ops.counter_add(self.counter, 1)

    def _networks(self, model, pose_e, unet, b, main):
        """the network passes of one step on ``self.xw``: (eps_cond, eps_uncond), NHWC fp32"""
        banks = self.bank_cur_kv
        nread, n_pose = self.nread, self.n_pose
        if pose_e is None:
            eps = unet.unet([self.xw, self.xw], self.t_cur, self.kv_unet, banks=banks, nread=nread,
                            only_mid_control=model.only_mid_control, emb=self.emb_cur_unet)
        elif self.merge_pose:
            # the pose ControlNet rides in the UNet encoder's launches (second parameter set): no stream of its own
            eps = unet.unet_pose(pose_e, self.xw, self.hint_feat, self.kv_unet, self.kv_merged, self.emb_cur_unet,
                                 self.emb_cur_pose, banks=banks, nread=nread, n_pose=n_pose, only_mid_control=model.only_mid_control)
        else:
            # the ControlNet's own launches on a forked stream, joined before the UNet's first pose residual add (its middle block)
            s_pose = self.pose_stream
            s_pose.wait_stream(main)
            with torch.cuda.stream(s_pose):
                pose = pose_e.pose([self.xw] * (n_pose // b), self.hint_feat, self.t_cur[:n_pose], self.kv_pose, emb=self.emb_cur_pose)
            eps = unet.unet([self.xw, self.xw], self.t_cur, self.kv_unet, banks=banks, pose=pose, nread=nread,
                            only_mid_control=model.only_mid_control, emb=self.emb_cur_unet, pose_ready=s_pose)
            main.wait_stream(s_pose)
        if self.balance:
            return eps[b:], eps[:b]      # (eps_c, eps_u): the batch is [uc half, c half] (ddim.py:566)
        return eps[:b], eps[b:]

    def step(self):
        if not self.use_graph:
            self._launch_sequence()
            return
        if self.graph is None:
            # warm-up pass sizes the arena (allocation is illegal under capture); it advances x, so restore after
            x0, c0 = self.x.clone(), self.counter.clone()
            self._launch_sequence()
            torch.cuda.current_stream().synchronize()
            self.x.copy_(x0)
            self.counter.copy_(c0)
            arena = self.model.engines()[2].arena
            g = ops.Graph()
            arena.frozen = True
            try:
                g.begin()
                try:
                    self._launch_sequence()
                except BaseException:
                    g.abort()   # leave the stream out of capture mode (a failed launch must not poison every later one)
                    raise
                g.end()
            finally:
                arena.frozen = False
            self.graph = g
        self.graph.launch()
