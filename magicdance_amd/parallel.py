"""Frame-sharded sampling over the GPUs of one node: one process per GPU, ``torch.distributed`` (backend "nccl" =
RCCL over xGMI).  New in this build -- the reference never shards inference (every rank would render every frame,
dataset/tiktok_video_arnold_copy.py:128-131).

Frames are independent given (pose_f, x_T, reference latent, text context, weights), so rank r owns a contiguous
block of frames and runs them as one batch.  The only shared quantity is the appearance bank: with ``wonoise`` it
depends on (reference latent, t, ctx) only, i.e. on the DDIM step but not on the frame.  The S banks are therefore
computed once per sequence: the table is laid out [row block][segment] (ddim.FusedStepRunner._allocate_table); per chunk
of ``world`` blocks, rank r runs the appearance net for ITS block of timesteps (one batched pass) and applies the UNet's
to_k / to_v -- the "reference-image KV" of the north star, 46 MB fp16 per step at 512x512 -- and ONE in-place RCCL all-gather
exchanges the chunk's contiguous blocks (point-to-point xGMI links: every link carries a different block at the same time).
The table is split into two chunks so that the second all-gather overlaps the first steps of the loop.  The 50-step loop
itself has no collective: every rank replays its captured step graph (pose ControlNet + UNet cond/uncond + CFG/DDIM
update) reading table row ``step``.  Decoded frames (or latents) are all-gathered at the end.  Every rank issues the same
collectives in the same order whatever its own number of frames (uneven and empty shards included).
"""
import torch

from .ddim import DDIMSampler_ReferenceOnly, FusedStepRunner


class FrameShardedSampler:
    def __init__(self, model, rank=0, world=1, group=None, force_sharded=False):
        """``force_sharded``: run the sharded form (blocked table layout, RCCL all-gathers of the table chunks and of the results)
        even when ``world`` is 1 -- a 1-rank process group exercises every collective of the multi-GPU path on one GPU."""
        self.model, self.rank, self.world, self.group = model, rank, world, group
        self.sharded = world > 1 or force_sharded

    @staticmethod
    def frame_block(F, rank, world):
        """Contiguous frame block [f0, f1) of ``rank``: sizes differ by at most one frame, earlier ranks take the extra ones."""
        q, r = divmod(F, world)
        f0 = rank * q + min(rank, r)
        return f0, f0 + q + (1 if rank < r else 0)

    def _cond(self, pose, ctx, ref):
        b = pose.shape[0]
        rep = lambda t: t if t.shape[0] == b else t.expand(b, *t.shape[1:])  # noqa: E731
        c = {"c_concat": [pose], "c_crossattn": [rep(ctx)], "image_control": [rep(ref)], "wonoise": True,
             "overlap_sampling": False}
        uc = {"c_concat": [pose], "c_crossattn": [rep(ctx)], "wonoise": True, "overlap_sampling": False}
        return c, uc

    def _runner(self):
        st = self.model._fused
        if st is None:
            st = self.model._fused = FusedStepRunner(self.model)
        return st

    def _gather(self, z, counts=None):
        """all-gather of per-rank results along the frame axis; ``counts`` (frames per rank) when the shards are uneven: the
        shards are padded to the largest one for the collective and trimmed afterwards."""
        import torch.distributed as dist
        if counts is None:
            outs = [torch.empty_like(z) for _ in range(self.world)]
            dist.all_gather(outs, z, group=self.group)
            return torch.cat(outs, 0)
        m = max(counts)
        pad = torch.zeros((m,) + tuple(z.shape[1:]), dtype=z.dtype, device=z.device)
        pad[:z.shape[0]] = z
        outs = [torch.empty_like(pad) for _ in range(self.world)]
        dist.all_gather(outs, pad, group=self.group)
        return torch.cat([o[:n] for o, n in zip(outs, counts)], 0)

    @torch.no_grad()
    def sample(self, pose, ctx, ref, x_T, ddim_steps=50, scale=7.0, gather=True, decode=False):
        """pose [Bl,3,8h,8w] = this rank's frames; ctx [1,77,768]; ref [1,4,h,w]; x_T [Bl,4,h,w].
        Returns the latents -- or, with ``decode``, the first-stage-decoded frames [.,3,8h,8w] -- of ALL frames
        (rank order) when ``gather`` else this rank's.  Equal frame counts per rank (use sample_sequence otherwise)."""
        model = self.model
        c, uc = self._cond(pose, ctx, ref)
        if not self.sharded:
            z, _ = model.sample_log(cond=c, batch_size=pose.shape[0], ddim=True, ddim_steps=ddim_steps, eta=0.0,
                                    unconditional_guidance_scale=scale, unconditional_conditioning=uc, inpaint=None,
                                    x_T=x_T)
            return model.decode_first_stage(z) if decode else z
        sampler = DDIMSampler_ReferenceOnly(model)
        sampler.make_schedule(ddim_steps, ddim_eta=0.0, verbose=False)
        st = self._runner()
        caller = torch.cuda.current_stream()
        st.stream.wait_stream(caller)
        with torch.cuda.stream(st.stream):
            st.prepare(c, x_T, sampler, scale, table_mode=True, world=self.world, sharded=self.sharded)
            st.run_steps(self.rank, self.world, self.group, sharded=self.sharded)
            z = st.x.clone()
            if decode:
                z = model.decode_first_stage(z)
            if gather:   # the all-gather of decoded frames of the north star (latents when the caller decodes itself)
                z = self._gather(z)
        caller.wait_stream(st.stream)
        return z

    @torch.no_grad()
    def sample_sequence(self, pose_frames, ctx, ref, x_T, frames_per_batch=8, ddim_steps=50, scale=7.0, decode=False,
                        gather_counts=None):
        """A whole pose sequence sharing one reference image (the entry points' use case, test_any_image_pose.py:201-262:
        same ref latent, same text, same x_T for every frame).  With ``wonoise`` the appearance bank depends on the DDIM
        step only, so the reference-KV table is filled ONCE per sequence -- by the first batch's run, overlapped with its steps;
        every rank takes part in every chunk's all-gather exactly once, whatever its number of frames (a rank with no frames
        fills its blocks and steps nothing) -- and the remaining batches of ``frames_per_batch`` frames only replay the captured
        step graph.  pose_frames [F,3,8h,8w] = this rank's frames (F may be 0 and may differ between ranks); x_T [1,4,h,w].
        Returns this rank's latents / decoded frames [F,...], or those of all ranks when ``gather_counts`` (frames per rank)."""
        model = self.model
        sampler = DDIMSampler_ReferenceOnly(model)
        sampler.make_schedule(ddim_steps, ddim_eta=0.0, verbose=False)
        st = self._runner()
        F = pose_frames.shape[0]
        outs = []
        caller = torch.cuda.current_stream()
        st.stream.wait_stream(caller)
        with torch.cuda.stream(st.stream):
            if F == 0:   # empty shard: fill this rank's table blocks (collectives included), no steps
                dummy = torch.zeros((1, 3, 8 * x_T.shape[2], 8 * x_T.shape[3]), dtype=x_T.dtype, device=x_T.device)
                c, _ = self._cond(dummy, ctx, ref)
                st.prepare(c, x_T[:1].contiguous(), sampler, scale, table_mode=True, world=self.world, sharded=self.sharded)
                st.run_steps(self.rank, self.world, self.group, steps=False, sharded=self.sharded)
            for f0 in range(0, F, frames_per_batch):
                pose = pose_frames[f0:f0 + frames_per_batch].contiguous()
                b = pose.shape[0]
                c, _ = self._cond(pose, ctx, ref)
                st.prepare(c, x_T.expand(b, *x_T.shape[1:]).contiguous(), sampler, scale, table_mode=True, world=self.world,
                           sharded=self.sharded)
                st.run_steps(self.rank, self.world, self.group, fill=(f0 == 0), sharded=self.sharded)
                outs.append(model.decode_first_stage(st.x) if decode else st.x.clone())
            if outs:
                z = torch.cat(outs, 0)
            else:
                side = 8 * x_T.shape[2] if decode else x_T.shape[2]
                z = torch.zeros((0, 3 if decode else x_T.shape[1], side, side), dtype=torch.float32, device=x_T.device)
            if gather_counts is not None and self.sharded:
                z = self._gather(z, list(gather_counts))
        caller.wait_stream(st.stream)
        return z

    @torch.no_grad()
    def network_pass_times(self, pose, ctx, ref, x_T, ddim_steps=50, scale=7.0, reps=5):
        """Metric (ii) of SURVEY 8(d), "UNet ms/step": HIP-event time of ONE forward of each network at this batch size B =
        pose.shape[0] -- ``unet_read`` (ControlledUnetModelAttnPose read branch: bank attention + pose residuals), ``unet_uc``
        (plain branch), ``pose`` (ControlNet incl. its 13 zero-convs), ``unet_pose_merged`` (UNet 2B + ControlNet B as one pass, the default step), ``appearance`` (ControlNetReferenceOnly write pass, B = 1:
        the reference image is shared) and ``unet_cfg_2b`` (what a step actually launches: read + uc batched as 2B samples).
        Each pass is captured into a HIP graph and replayed ``reps`` times between two events on the launch stream."""
        from . import ops
        from .engine import Act
        model = self.model
        c, _ = self._cond(pose, ctx, ref)
        sampler = DDIMSampler_ReferenceOnly(model)
        sampler.make_schedule(ddim_steps, ddim_eta=0.0, verbose=False)
        st = self._runner()
        app, pose_e, unet = model.engines()
        b = pose.shape[0]
        out = {}
        with torch.cuda.stream(st.stream):
            st.prepare(c, x_T, sampler, scale, table_mode=True)
            st.compute_bank_rows(range(min(st.S, st.per)))
            st.counter.zero_()
            ops.select_row_f32(st.ts_table, st.counter, 0, st.t_cur, 2 * b, st.S)
            ops.gather_rows(st.bank_table, st.bank_seg, st.bank_seg.shape[0], st.bank_seg_max, st.counter, 0, st.bank_cur,
                            st.S, st.per, st.block_elems // st.table_unit)
            ops.select_row_f32(st.emb_table_unet, st.counter, 0, st.emb_cur_unet, st.emb_cur_unet.shape[1], st.S)
            ops.select_row_f32(st.emb_table_pose, st.counter, 0, st.emb_cur_pose, st.emb_cur_pose.shape[1], st.S)
            unet.arena.reset()
            pres = [Act(p.t.clone(), p.b, p.h, p.w, p.c) for p in pose_e.pose(st.x, st.hint_feat, st.t_cur[:b], st.kv_pose)]
            bref = st.ref.shape[0]
            passes = {
                "unet_read": lambda: unet.unet(st.x, st.t_cur[:b], st.kv_unet_uc, banks=st.bank_cur_kv, pose=pres, nread=b,
                                               only_mid_control=model.only_mid_control),
                "unet_uc": lambda: unet.unet(st.x, st.t_cur[:b], st.kv_unet_uc, nread=0),
                "unet_cfg_2b": lambda: unet.unet([st.x, st.x], st.t_cur, st.kv_unet, banks=st.bank_cur_kv, pose=pres, nread=b,
                                                 only_mid_control=model.only_mid_control),
                "pose": lambda: pose_e.pose(st.x, st.hint_feat, st.t_cur[:b], st.kv_pose),
                "appearance": lambda: app.appearance(st.ref, st.t_cur[:bref], st.kv_app),
            }
            if st.kv_merged is not None:   # what a step launches by default: UNet (2B samples) with the ControlNet's B samples riding
                passes["unet_pose_merged"] = lambda: unet.unet_pose(   # in its encoder launches
                    pose_e, st.x, st.hint_feat, st.kv_unet, st.kv_merged, st.emb_cur_unet, st.emb_cur_pose, banks=st.bank_cur_kv,
                    nread=b, only_mid_control=model.only_mid_control)
            for name, fn in passes.items():
                def run():
                    unet.arena.reset()
                    fn()
                run()                                   # warm: sizes the arena (allocation is illegal under capture)
                st.stream.synchronize()
                g = ops.Graph()
                unet.arena.frozen = True
                try:
                    g.begin()
                    try:
                        run()
                    except BaseException:
                        g.abort()
                        raise
                    g.end()
                finally:
                    unet.arena.frozen = False
                g.launch()
                st.stream.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(st.stream)
                for _ in range(reps):
                    g.launch()
                e1.record(st.stream)
                st.stream.synchronize()
                g.destroy()
                out[name] = e0.elapsed_time(e1) / reps
        return out

    @torch.no_grad()
    def profile_one_step(self, pose, ctx, ref, x_T, ddim_steps=50, scale=7.0, decode=False):
        """Per-kernel-family time of ONE batch of frames: the reference-KV table pass (all S rows, once), ONE DDIM
        step (x S) and (``decode``) the first-stage decode of the batch, each (a) as plain un-captured launches timed per launch by md_prof_* and (b) -- igemm / attention --
        replayed back-to-back from a captured graph between two HIP events on the launch stream.  Returns
        {family: {ms, launches, flops, bytes, graph_ms, ...}} aggregated over the batch (table + S * step + decode) with
        the parts under "table" / "step" / "decode"."""
        import ctypes as C
        import os
        from . import ops
        model = self.model
        c, _ = self._cond(pose, ctx, ref)
        sampler = DDIMSampler_ReferenceOnly(model)
        sampler.make_schedule(ddim_steps, ddim_eta=0.0, verbose=False)
        st = model._fused
        if st is None:
            st = model._fused = FusedStepRunner(model)
        table = True

        def timed(fn):
            ops.prof_enable(True)
            ops.RECORD = []
            fn()
            st.stream.synchronize()
            fam = ops.prof_collect()
            ops.prof_enable(False)
            rec, ops.RECORD = ops.RECORD, None
            sp = ops.stream_ptr()
            for name in ("igemm", "attention"):
                calls = [r for r in rec if r[0] == name]
                if not calls:
                    continue
                g = ops.Graph()
                g.begin()
                for _, fn_, p, _, _ in calls:
                    fn_(C.byref(p), sp)
                g.end()
                g.launch()
                st.stream.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                reps = 3
                e0.record(st.stream)
                for _ in range(reps):
                    g.launch()
                e1.record(st.stream)
                st.stream.synchronize()
                g.destroy()
                fam[name]["graph_ms"] = e0.elapsed_time(e1) / reps
                fam[name]["graph_launches"] = len(calls)
            return fam

        with torch.cuda.stream(st.stream):
            st.prepare(c, x_T, sampler, scale, table_mode=table)
            S = st.S
            if table:
                st.compute_bank_rows(range(S))   # warm: sizes the arenas
            st._launch_sequence()
            st.stream.synchronize()
            st.counter.zero_()
            parts = {"table": timed(lambda: st.compute_bank_rows(range(S))) if table else None,
                     "step": timed(st._launch_sequence)}
            if decode:
                z = st.x.clone()
                model.decode_first_stage(z)          # warm: sizes the arena
                parts["decode"] = timed(lambda: model.decode_first_stage(z))
        out = {}
        for name, stp in parts["step"].items():
            once = [parts[k][name] for k in ("table", "decode") if parts.get(k) is not None]
            agg = {}
            for k in ("ms", "launches", "flops", "bytes", "graph_ms"):
                if k in stp or any(k in o for o in once):
                    agg[k] = S * stp.get(k, stp["ms"] if k == "graph_ms" else 0) + \
                        sum(o.get(k, o["ms"] if k == "graph_ms" else 0) for o in once)
            agg["step"], agg["ddim_steps"] = stp, S
            agg["table"] = parts["table"][name] if parts.get("table") is not None else None
            agg["decode"] = parts["decode"][name] if parts.get("decode") is not None else None
            out[name] = agg
        return out
