"""Frame-sharded sampling over the GPUs of one node: one process per GPU, ``torch.distributed`` (backend "nccl" =
RCCL over xGMI).  New in this build -- the reference never shards inference (every rank would render every frame,
dataset/tiktok_video_arnold_copy.py:128-131).

Frames are independent given (pose_f, x_T, reference latent, text context, weights), so rank r owns a contiguous
block of frames and runs them as one batch.  The only shared quantity is the appearance bank: with ``wonoise`` it
depends on (reference latent, t, ctx) only, i.e. on the DDIM step but not on the frame.  The S banks (23 MB fp16 each
at 512x512) are therefore computed once per sequence, round-robin over ranks (rank r runs the appearance net for the
steps i with i % world == r), and exchanged with S RCCL broadcasts (point-to-point xGMI links: each broadcast is a
direct root->peers fan-out of one 23 MB buffer).  After that the 50-step loop has no collective at all: every rank
replays its captured step graph (pose ControlNet + UNet cond/uncond + CFG/DDIM update) reading bank row ``step``.
Final latents are all-gathered (the decoded-frame gather of the north star, at the latent seam of this round's scope).
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

    @torch.no_grad()
    def sample(self, pose, ctx, ref, x_T, ddim_steps=50, scale=7.0, gather=True):
        """pose [Bl,3,8h,8w] = this rank's frames; ctx [1,77,768]; ref [1,4,h,w]; x_T [Bl,4,h,w].
        Returns the latents of ALL frames ([world*Bl,4,h,w], rank order) when ``gather`` else this rank's."""
        model = self.model
        c, uc = self._cond(pose, ctx, ref)
        if self.world == 1:
            z, _ = model.sample_log(cond=c, batch_size=pose.shape[0], ddim=True, ddim_steps=ddim_steps, eta=0.0,
                                    unconditional_guidance_scale=scale, unconditional_conditioning=uc, inpaint=None,
                                    x_T=x_T)
            return z
        import torch.distributed as dist
        sampler = DDIMSampler_ReferenceOnly(model)
        sampler.make_schedule(ddim_steps, ddim_eta=0.0, verbose=False)
        st = model._fused
        if st is None:
            st = model._fused = FusedStepRunner(model)
        caller = torch.cuda.current_stream()
        st.stream.wait_stream(caller)
        with torch.cuda.stream(st.stream):
            st.prepare(c, x_T, sampler, scale, table_mode=True)
            S = st.S
            st.compute_bank_rows([i for i in range(S) if i % self.world == self.rank])
            for i in range(S):  # reference-image bank ("ref-KV") broadcast over RCCL / xGMI
                dist.broadcast(st.bank_table[i], src=i % self.world, group=self.group)
            for _ in range(S):
                st.step()
            z = st.x.clone()
            if gather:
                outs = [torch.empty_like(z) for _ in range(self.world)]
                dist.all_gather(outs, z, group=self.group)
                z = torch.cat(outs, 0)
        caller.wait_stream(st.stream)
        return z

    @torch.no_grad()
    def sample_sequence(self, pose_frames, ctx, ref, x_T, frames_per_batch=8, ddim_steps=50, scale=7.0):
        """A whole pose sequence sharing one reference image (the entry points' use case, test_any_image_pose.py:201-262:
        same ref latent, same text, same x_T for every frame).  With ``wonoise`` the appearance bank depends on the DDIM
        step only, so the S banks are computed ONCE per sequence (round-robin over ranks + RCCL broadcast when world > 1)
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
                st.prepare(c, x_T.expand(b, *x_T.shape[1:]).contiguous(), sampler, scale, table_mode=True)
                if not have_table or st.key != old_key:   # (re)allocated buffers: the table must be filled for this geometry
                    S = st.S
                    st.compute_bank_rows([i for i in range(S) if i % self.world == self.rank])
                    if self.world > 1:
                        import torch.distributed as dist
                        for i in range(S):
                            dist.broadcast(st.bank_table[i], src=i % self.world, group=self.group)
                    have_table = True
                for _ in range(st.S):
                    st.step()
                outs.append(st.x.clone())
        caller.wait_stream(st.stream)
        return torch.cat(outs, 0)

    @torch.no_grad()
    def profile_one_step(self, pose, ctx, ref, x_T, ddim_steps=50, scale=7.0):
        """Run ONE DDIM step as plain (un-captured) launches so md_prof_* can time every kernel."""
        model = self.model
        c, _ = self._cond(pose, ctx, ref)
        sampler = DDIMSampler_ReferenceOnly(model)
        sampler.make_schedule(ddim_steps, ddim_eta=0.0, verbose=False)
        st = model._fused
        if st is None:
            st = model._fused = FusedStepRunner(model)
        import ctypes as C
        from . import ops
        with torch.cuda.stream(st.stream):
            st.prepare(c, x_T, sampler, scale, table_mode=False)
            st._launch_sequence()          # sizes the arena / warms caches
            st.stream.synchronize()
            st.counter.zero_()
            ops.prof_enable(True)          # (a) per-launch events, un-captured: includes the eager launch latency
            ops.RECORD = []
            st._launch_sequence()
            st.stream.synchronize()
            fam = ops.prof_collect()
            ops.prof_enable(False)
            rec, ops.RECORD = ops.RECORD, None
            # (b) the same launches of one family replayed back-to-back from a captured graph on this stream, bracketed
            #     by HIP events: average launch duration as the GPU sees it inside the step graph
            sp = ops.stream_ptr()
            for name in ("igemm", "attention"):
                calls = [r for r in rec if r[0] == name]
                if not calls:
                    continue
                g = ops.Graph()
                g.begin()
                for _, fn, p, _, _ in calls:
                    fn(C.byref(p), sp)
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
                fam[name]["graph_flops"] = float(sum(r[3] for r in calls))
        return fam
