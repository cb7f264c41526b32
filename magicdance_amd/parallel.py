"""Frame-sharded sampling over the GPUs of one node: one process per GPU, ``torch.distributed`` (backend "nccl" =
RCCL over xGMI).  New in this build -- the reference never shards inference (every rank would render every frame,
dataset/tiktok_video_arnold_copy.py:128-131).

Frames are independent given (pose_f, x_T, reference latent, text context, weights), so rank r owns a contiguous
block of frames and runs them as one batch.  The only shared quantity is the appearance bank: with ``wonoise`` it
depends on (reference latent, t, ctx) only, i.e. on the DDIM step but not on the frame.  The S banks are therefore
computed once per sequence: rank r runs the appearance net for a contiguous block of S/world steps (batched over those
timesteps) and applies the UNet's to_k / to_v to them -- the "reference-image KV" of the north star, 46 MB fp16 per
step at 512x512 -- and the equal row blocks are exchanged with one RCCL all-gather per table segment (point-to-point xGMI
links: every link carries a different block at the same time).  After that the 50-step loop has no collective at all:
every rank replays its captured step graph (pose ControlNet + UNet cond/uncond + CFG/DDIM update) reading table row
``step``.  Final latents are all-gathered (the decoded-frame gather of the north star, at the latent seam of this
round's scope).
"""
import torch

from .ddim import DDIMSampler_ReferenceOnly, FusedStepRunner


class FrameShardedSampler:
    def __init__(self, model, rank=0, world=1, group=None):
        self.model, self.rank, self.world, self.group = model, rank, world, group

    def _cond(self, pose, ctx, ref):
        b = pose.shape[0]
        rep = lambda t: t if t.shape[0] == b else t.expand(b, *t.shape[1:])  # noqa: E731
        c = {"c_concat": [pose], "c_crossattn": [rep(ctx)], "image_control": [rep(ref)], "wonoise": True,
             "overlap_sampling": False}
        uc = {"c_concat": [pose], "c_crossattn": [rep(ctx)], "wonoise": True, "overlap_sampling": False}
        return c, uc

    def table_rows(self, S):
        """Rows per table segment: S padded to a multiple of the world size (equal blocks -> one all-gather per segment)."""
        return -(-S // self.world) * self.world

    def row_block(self, S, rank):
        """DDIM rows [r0, r1) whose reference-KV this rank computes: the valid part of its equal block of the padded table."""
        q = self.table_rows(S) // self.world
        return min(S, rank * q), min(S, (rank + 1) * q)

    def _fill_table(self, st):
        """Reference-KV table of the whole schedule: every rank runs the appearance net (batched over its block of
        timesteps) for ~S/world rows, then ONE RCCL all-gather per table segment (K or V^T of a bank entry, 32 segments)
        exchanges the equal row blocks -- every xGMI link carries a different block at the same time, instead of world x 32
        root->peers broadcasts one after the other."""
        r0, r1 = self.row_block(st.S, self.rank)
        if r1 > r0:
            st.compute_bank_rows(range(r0, r1))
        if self.world > 1:
            import torch.distributed as dist
            q = st.table_rows // self.world
            per_rank = [st.table_slabs(src * q, (src + 1) * q) for src in range(self.world)]   # [rank][segment] views
            for seg in range(len(per_rank[0])):
                outs = [per_rank[src][seg] for src in range(self.world)]
                dist.all_gather(outs, outs[self.rank].clone(), group=self.group)

    @torch.no_grad()
    def sample(self, pose, ctx, ref, x_T, ddim_steps=50, scale=7.0, gather=True, decode=False):
        """pose [Bl,3,8h,8w] = this rank's frames; ctx [1,77,768]; ref [1,4,h,w]; x_T [Bl,4,h,w].
        Returns the latents -- or, with ``decode``, the first-stage-decoded frames [.,3,8h,8w] -- of ALL frames
        (rank order) when ``gather`` else this rank's."""
        model = self.model
        c, uc = self._cond(pose, ctx, ref)
        if self.world == 1:
            z, _ = model.sample_log(cond=c, batch_size=pose.shape[0], ddim=True, ddim_steps=ddim_steps, eta=0.0,
                                    unconditional_guidance_scale=scale, unconditional_conditioning=uc, inpaint=None,
                                    x_T=x_T)
            return model.decode_first_stage(z) if decode else z
        import torch.distributed as dist
        sampler = DDIMSampler_ReferenceOnly(model)
        sampler.make_schedule(ddim_steps, ddim_eta=0.0, verbose=False)
        st = model._fused
        if st is None:
            st = model._fused = FusedStepRunner(model)
        caller = torch.cuda.current_stream()
        st.stream.wait_stream(caller)
        with torch.cuda.stream(st.stream):
            st.prepare(c, x_T, sampler, scale, table_mode=True, table_rows=self.table_rows(ddim_steps))
            S = st.S
            self._fill_table(st)
            for _ in range(S):
                st.step()
            z = st.x.clone()
            if decode:
                z = model.decode_first_stage(z)
            if gather:   # the all-gather of decoded frames of the north star (latents when the caller decodes itself)
                outs = [torch.empty_like(z) for _ in range(self.world)]
                dist.all_gather(outs, z, group=self.group)
                z = torch.cat(outs, 0)
        caller.wait_stream(st.stream)
        return z

    @torch.no_grad()
    def sample_sequence(self, pose_frames, ctx, ref, x_T, frames_per_batch=8, ddim_steps=50, scale=7.0, decode=False):
        """A whole pose sequence sharing one reference image (the entry points' use case, test_any_image_pose.py:201-262:
        same ref latent, same text, same x_T for every frame).  With ``wonoise`` the appearance bank depends on the DDIM
        step only, so the S banks are computed ONCE per sequence (equal row blocks per rank + RCCL all-gathers when world > 1)
        and every batch of ``frames_per_batch`` frames replays the captured step graph in table mode.
        pose_frames [F,3,8h,8w] = this rank's frames; x_T [1,4,h,w].  Returns this rank's latents [F,4,h,w]."""
        model = self.model
        sampler = DDIMSampler_ReferenceOnly(model)
        sampler.make_schedule(ddim_steps, ddim_eta=0.0, verbose=False)
        st = model._fused
        if st is None:
            st = model._fused = FusedStepRunner(model)
        F = pose_frames.shape[0]
        outs = []
        caller = torch.cuda.current_stream()
        st.stream.wait_stream(caller)
        with torch.cuda.stream(st.stream):
            have_table = False
            for f0 in range(0, F, frames_per_batch):
                pose = pose_frames[f0:f0 + frames_per_batch].contiguous()
                b = pose.shape[0]
                c, _ = self._cond(pose, ctx, ref)
                old_key = st.key
                st.prepare(c, x_T.expand(b, *x_T.shape[1:]).contiguous(), sampler, scale, table_mode=True,
                           table_rows=self.table_rows(ddim_steps))
                if not have_table or st.key != old_key:   # (re)allocated buffers: the table must be filled for this geometry
                    self._fill_table(st)
                    have_table = True
                for _ in range(st.S):
                    st.step()
                outs.append(model.decode_first_stage(st.x) if decode else st.x.clone())
        caller.wait_stream(st.stream)
        return torch.cat(outs, 0)

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
        table = os.environ.get("MD_BANK_MODE", "table") != "inline"

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
