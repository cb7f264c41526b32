"""bench.py -- headline benchmark of the MI355X hot path: 512x512 frames/s at 50 DDIM steps.

A "step" (driver contract) = one pass of the hot path over one batch of synthetic input = sampling ONE batch of
`--frames-per-gpu` frames: the reference-KV table pass (appearance net for all 50 timesteps, batched over timesteps,
plus the UNet's bank K/V projections) and the full 50-step DDIM loop (pose ControlNet + UNet cond/uncond + CFG/DDIM
update per step) and the first-stage (VAE) decode of the frames, latents in -> decoded frames out, inputs resident in
HBM before the timed region; nothing is cached across batches (the table is recomputed for every batch).
N=1 default workload = BASELINE.json configs[1]: single 512x512 frame, 50-step DDIM, full Appearance+Pose ControlNet,
fp16, random-init (seeded synthetic) SD-1.5-geometry weights.  N>1: one process per GPU (torch.distributed over RCCL),
frames sharded across ranks (weak scaling: `--frames-per-gpu` each), the reference-image KV table computed in equal row
blocks (one block of timesteps per rank) and exchanged with one RCCL all-gather per table segment, decoded frames
all-gathered.

Prints ONE JSON line (rank 0).  Extra objects: "roofline" (igemm = the dominant kernel family, HIP-event timed per
launch on the launch stream) and "cpu_baseline" (the CPU oracle timed on this box's host cores on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP16_TFLOPS = 2500.0  # dense MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md


def build_model(device, size):
    import magicdance_amd as M
    from magicdance_amd import synthetic
    cfg = M.cldm.load_config(M.DEFAULT_CONFIG)["model"]
    cfg["params"]["cond_stage_config"] = "__is_unconditional__"   # CLIP runs once per sequence, outside the timed path
    cfg["params"]["image_size"] = size
    with torch.device("meta"):
        model = M.instantiate_from_config(cfg)
    model = model.to_empty(device=device)
    model.register_schedule(timesteps=1000, linear_start=cfg["params"]["linear_start"], linear_end=cfg["params"]["linear_end"])
    model.logvar = torch.zeros(1000)
    model = model.to(device)
    with torch.no_grad():
        for pre, mod in (("model.diffusion_model.", model.model.diffusion_model),
                         ("appearance_control_model.", model.appearance_control_model),
                         ("pose_control_model.", model.pose_control_model),
                         ("first_stage_model.", model.first_stage_model)):
            sd = synthetic.synth_state_dict(mod, pre, seed=0, device=device)
            mod.load_state_dict({k[len(pre):]: v for k, v in sd.items()}, strict=True)
    return model.eval()


def cpu_baseline(model, inp, size, z_hip=None, img_hip=None, steps=3):
    """The CPU oracle (oracle/restatement.py + oracle/vae_restatement.py, torch fp32) on this box's host cores, bounded
    sample: the first ``steps`` of the 50 DDIM steps of the same single-frame workload (appearance net, pose ControlNet,
    UNet read pass, UNet uncond pass -- each timed -- and the CFG / DDIM update) plus the first-stage decode of one frame;
    frames/s = 1 / (50 x mean step + decode).  The first step's eps pair doubles as a full-size parity check of the HIP path."""
    import numpy as np
    from oracle import restatement as R
    from oracle import vae_restatement as V
    sd = {}
    for pre, mod in (("model.diffusion_model.", model.model.diffusion_model),
                     ("appearance_control_model.", model.appearance_control_model),
                     ("pose_control_model.", model.pose_control_model)):
        sd.update({pre + k: v.detach().float().cpu() for k, v in mod.state_dict().items()})
    cfg = R.Cfg()
    ctx, ref, pose = inp["ctx"].cpu().float(), inp["ref"].cpu().float(), inp["pose"][:1].cpu().float()
    c = {"c_concat": [pose], "c_crossattn": [ctx], "image_control": [ref], "wonoise": True, "overlap_sampling": False}
    x = inp["x_T"].cpu()
    ts = np.flip(R.make_ddim_timesteps(50))
    _, alphas, alphas_prev = R.make_ddim_sampling_parameters(R.alphas_cumprod(), R.make_ddim_timesteps(50), 0.0)
    threads = torch.get_num_threads()
    per = {"appearance": [], "pose": [], "unet_read": [], "unet_uc": []}
    parity = None
    with torch.no_grad():
        for i in range(steps):
            t = torch.full((1,), int(ts[i]), dtype=torch.long)
            index = 50 - i - 1
            t0 = time.time()
            banks = R.appearance_forward(sd, R.APP, cfg, ref, t, ctx)
            t1 = time.time()
            pr = R.pose_forward(sd, R.POSE, cfg, x, pose, t, ctx)
            t2 = time.time()
            e_c = R.unet_forward(sd, R.UNET, cfg, x, t, ctx, banks, pr, False, False)
            t3 = time.time()
            e_u = R.unet_forward(sd, R.UNET, cfg, x, t, ctx, [], None, True, False)
            t4 = time.time()
            for k, v in zip(per, (t1 - t0, t2 - t1, t3 - t2, t4 - t3)):
                per[k].append(v)
            if i == 0:
                # the oracle outputs double as a full-size parity check of the HIP path on the same inputs (checker only)
                dev = inp["x_T"].device
                cd = {k: ([v.to(dev) for v in vv] if isinstance(vv, list) else vv) for k, vv in c.items()}
                h_c = model.apply_model(inp["x_T"], t.to(dev), cd, inp["ref"]).cpu()
                h_u = model.apply_model(inp["x_T"], t.to(dev), cd, None, uc=True).cpu()
                rel = lambda a, b: float((a - b).abs().max() / b.abs().max())  # noqa: E731
                parity = {"eps_cond_rel_max_abs": rel(h_c, e_c), "eps_uncond_rel_max_abs": rel(h_u, e_u),
                          "note": f"HIP path vs fp32 CPU oracle, t={int(ts[0])}, same synthetic weights/inputs"}
            e = e_u + 7.0 * (e_c - e_u)
            a_t, a_p = float(alphas[index]), float(alphas_prev[index])
            x = (a_p ** 0.5) * (x - (1.0 - a_t) ** 0.5 * e) / (a_t ** 0.5) + (1.0 - a_p) ** 0.5 * e
        dt = sum(sum(v) for v in per.values()) / steps
        dt_vae = 0.0
        if z_hip is not None:
            pre = "first_stage_model."
            vsd = {pre + k: v.detach().float().cpu() for k, v in model.first_stage_model.state_dict().items()}
            t0 = time.time()
            img = V.vae_decode(vsd, pre, z_hip[:1].cpu() / model.scale_factor)
            dt_vae = time.time() - t0
            parity["decode_rel_max_abs"] = rel(img_hip[:1].cpu(), img)
    return {"value": 1.0 / (50.0 * dt + dt_vae), "unit": "frames/s", "cores": threads, "kind": "port",
            "sample": f"first {steps} of 50 DDIM steps (1 frame {8 * size}x{8 * size}), mean {dt:.1f} s/step, x50 extrapolated"
                      + (f", + first-stage decode of the frame = {dt_vae:.1f}s" if z_hip is not None else ""),
            "s_per_step": {k: sum(v) / len(v) for k, v in per.items()}, "s_per_step_total": dt, "decode_s": dt_vae,
            "parity_full_size": parity}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5, help="timed frame-batches (each = a full DDIM loop)")
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--frames-per-gpu", type=int, default=None,
                    help="frames per GPU per batch; default 1 at --gpus 1 (BASELINE configs[1]), 8 at --gpus N > 1 (configs[3]: "
                         "8 frames per GPU as one batch, 64 frames on 8 GPUs)")
    ap.add_argument("--no-extra", action="store_true", help="skip the extra configs[2] line (8 frames as one batch) at --gpus 1")
    ap.add_argument("--ddim-steps", type=int, default=50)
    ap.add_argument("--size", type=int, default=64, help="latent side (64 = 512x512)")
    ap.add_argument("--sequence", type=int, default=0,
                    help="non-default workload: a pose sequence of this many frames per GPU sharing one reference image "
                         "(bank table computed once per sequence), sampled in batches of --frames-per-gpu")
    ap.add_argument("--fp8-attention", action="store_true",
                    help="BASELINE configs[4] path: K / V^T / bank table as OCP e4m3, attention contractions on the fp8 MFMA")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-decode", action="store_true", help="stop at the latents (skip the first-stage decode)")
    ap.add_argument("--no-graph", action="store_true",
                    help="launch every DDIM step un-captured (same launches, no HIP graph): the form the rocprofv3 --pmc passes of "
                         "tools/run_profiles.sh run on (the counter tool does not survive graph replays of the linear step graph)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus or world == 1, "launch with torch.distributed.run --nproc-per-node == --gpus"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)

    from magicdance_amd import synthetic, ops
    from magicdance_amd import parallel, engine
    if args.fp8_attention:
        engine.ATTN_FP8 = True   # read when the engines pack their weights / allocate their K / V^T buffers
    model = build_model(dev, args.size)
    fpg = args.frames_per_gpu if args.frames_per_gpu else (1 if world == 1 else 8)
    cfg_name = {(1, 1): "configs[1]", (8, 1): "configs[2]"}.get((fpg, world), "configs[3]" if (fpg == 8 and world > 1) else "custom")
    inp = synthetic.synth_inputs((args.size, args.size), frames=fpg * world, seed=0, device=dev)
    runner = parallel.FrameShardedSampler(model, rank=rank, world=world)
    if args.no_graph:
        # counter runs: every launch un-captured AND on ONE stream (the reference-KV table pass no longer overlaps the first steps;
        # same launches, same arguments, same order per stream) -- rocprofv3's counter collection serialises kernels anyway
        st = runner._runner()
        st.use_graph = False
        st.table_stream = st.stream
    my = slice(rank * fpg, (rank + 1) * fpg)
    pose, ctx, ref, x_T = inp["pose"][my].contiguous(), inp["ctx"], inp["ref"], inp["x_T"].repeat(fpg, 1, 1, 1)

    if args.sequence:
        inp = synthetic.synth_inputs((args.size, args.size), frames=args.sequence * world, seed=0, device=dev)
        seq_pose = inp["pose"][rank * args.sequence:(rank + 1) * args.sequence].contiguous()

    def one_batch():
        if args.sequence:
            return runner.sample_sequence(seq_pose, ctx, ref, inp["x_T"], frames_per_batch=fpg, ddim_steps=args.ddim_steps,
                                          scale=7.0, decode=not args.no_decode)
        return runner.sample(pose, ctx, ref, x_T, ddim_steps=args.ddim_steps, scale=7.0, decode=not args.no_decode)

    for _ in range(args.warmup):
        one_batch()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t0 = time.time()
    for _ in range(args.steps):
        z = one_batch()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.time() - t0
    if dist is not None:
        tt = torch.tensor([dt], device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    assert bool(torch.isfinite(z).all()), "non-finite latents"
    frames = args.steps * (args.sequence if args.sequence else fpg) * world
    out = {"metric": "512x512 frames/sec @ 50 DDIM steps", "value": frames / dt, "unit": "frames/s", "n_gpus": world,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "f16 (attention operands e4m3)" if args.fp8_attention else "f16", "data": "synthetic",
           "ms_per_ddim_step": 1e3 * dt / args.steps / args.ddim_steps,
           "config": {"workload": (f"sequence of {args.sequence} frames/GPU sharing one reference (bank table once per sequence), "
                                   f"batches of {fpg}, " if args.sequence else f"{cfg_name}: {fpg} frame(s)/GPU as one batch, ") +
                                  f"{8 * args.size}x{8 * args.size}, {args.ddim_steps}-step DDIM, "
                                  "appearance + pose ControlNet + UNet cond/uncond, CFG 7, latents in -> " +
                                  ("latents out" if args.no_decode else "first-stage-decoded frames out"),
                      "frames_per_gpu": fpg, "ddim_steps": args.ddim_steps, "weights": "seeded synthetic, SD-1.5 geometry",
                      "parallelism": f"frame-shard x{world}"}}
    if world > 1:
        # the N = 1 line's `value` is configs[1] (ONE frame per batch, the headline metric); the per-GPU work of this line is the
        # N = 1 line's `extra` entry -- that is the one-GPU figure a scaling efficiency of this line is to be taken against
        out["scaling_reference"] = (f"per-GPU work = {fpg} frame(s) as one batch: compare with the N=1 line's extra['configs[2]'].value "
                                    "(8 frames per batch on one GPU), not with its value (configs[1], 1 frame per batch)")
    if rank == 0 and not args.no_roofline:
        # every kernel family over ONE batch of frames = the reference-KV table pass (once) + S x one DDIM step: per-launch
        # HIP events on un-captured launches (ms_eager_events, includes eager launch latency) and, for igemm / attention,
        # the same launches replayed from a captured graph between two HIP events on the launch stream (graph_ms)
        fam = runner.profile_one_step(pose, ctx, ref, x_T, ddim_steps=args.ddim_steps, scale=7.0, decode=not args.no_decode)
        ig = fam["igemm"]
        ig_ms = ig.get("graph_ms", ig["ms"])
        ach = ig["flops"] / (ig_ms * 1e-3) / 1e12 if ig_ms > 0 else 0.0
        # HBM bytes per igemm launch from the PMC passes (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of THIS
        # workload -- the 50-step batch --, FETCH doubled per MI355X_MICROARCH.md); committed under profiles/, null when absent
        traffic, traffic_src = None, None
        traffic_kind = "measured"
        for cand in ("round3_pmc_summary.json", "round2_pmc_summary.json", "round1_pmc_summary.json"):
            try:
                pmc = json.load(open(os.path.join(ROOT, "profiles", cand)))
                traffic, traffic_src = pmc.get("igemm_hbm_bytes_per_launch"), cand
                # round 3: the counters are collected on THIS launch mix (bench.py --no-graph: the default merged pass, un-captured).
                # An older summary (other launch mix) is re-expressed per launch of this run's mix and labelled as derived.
                # (the counter run also sees the one-time launches of the first batch -- context K / V projections, hint encoder --:
                #  a launch count within 1 % of this run's is the same mix)
                if traffic and pmc.get("igemm_launches_per_batch") and ig["launches"] and \
                        abs(pmc["igemm_launches_per_batch"] - ig["launches"]) > 0.01 * ig["launches"]:
                    traffic = traffic * pmc["igemm_launches_per_batch"] / ig["launches"]
                    traffic_kind = f"derived: bytes per batch of a {pmc['igemm_launches_per_batch']}-launch mix over this run's launches"
                if not (args.ddim_steps == 50 and fpg == 1 and args.size == 64 and not args.sequence):
                    traffic, traffic_src = None, None   # the counters were collected on configs[1] only
                break
            except Exception:  # noqa: BLE001
                pass

        def part(d):
            if d is None:
                return None
            ms = d.get("graph_ms", d["ms"])
            return {"ms": ms, "launches": d["launches"], "tflops": d["flops"] / (ms * 1e-3) / 1e12 if ms > 0 else 0.0}
        out["roofline"] = {"bound": "mfma", "kernel": "igemm_kernel (all launches of one batch: reference-KV table pass + "
                           f"{ig['ddim_steps']} DDIM steps" + ("" if args.no_decode else " + first-stage decode") + ")", "achieved": ach,
                           "peak": PEAK_FP16_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_FP16_TFLOPS, "traffic": traffic,
                           "traffic_unit": (f"bytes/launch (PMC FETCH_SIZE x 2 + WRITE_SIZE, profiles/{traffic_src}, {traffic_kind})" if traffic_src else
                                            "null: the PMC passes (profiles/round3_pmc_summary.json) cover configs[1] only"),
                           "algorithmic_bytes_per_launch": ig["bytes"] / max(ig["launches"], 1),
                           "flops_per_launch": ig["flops"] / max(ig["launches"], 1),
                           "avg_launch_us": 1e3 * ig_ms / max(ig["launches"], 1), "launches": ig["launches"], "ms": ig_ms,
                           "ms_eager_events": ig["ms"], "table_pass": part(ig["table"]), "ddim_step": part(ig["step"]),
                           "first_stage_decode": part(ig["decode"])}
        out["families_ms_per_batch"] = {k: {"ms": v["ms"], "graph_ms": v.get("graph_ms"), "launches": v["launches"],
                                            "step_ms": v["step"].get("graph_ms", v["step"]["ms"]),
                                            "table_ms": None if v["table"] is None else v["table"].get("graph_ms", v["table"]["ms"]),
                                            "decode_ms": None if v["decode"] is None else v["decode"].get("graph_ms", v["decode"]["ms"]),
                                            "tflops": (v["flops"] / (v["ms"] * 1e-3) / 1e12 if v["ms"] > 0 else 0.0),
                                            "gbps": (v["bytes"] / (v["ms"] * 1e-3) / 1e9 if v["ms"] > 0 else 0.0)}
                                        for k, v in fam.items()}
    if rank == 0 and not args.no_roofline:
        # metric (ii), "UNet ms/step": one forward of each network at B = 1 and B = 8, graph-replayed between HIP events
        nets = {}
        for bb in (1, 8):
            pi = synthetic.synth_inputs((args.size, args.size), frames=bb, seed=0, device=dev)
            nets[f"B{bb}"] = runner.network_pass_times(pi["pose"], ctx, ref, pi["x_T"].repeat(bb, 1, 1, 1), ddim_steps=args.ddim_steps)
        out["unet_ms_per_step"] = nets
    if rank == 0 and world == 1 and fpg == 1 and not args.sequence and not args.no_extra:
        # extra line: BASELINE configs[2] (8 frames as one batch on one GPU), same timing protocol, 1 warm-up + 2 timed batches
        p8 = synthetic.synth_inputs((args.size, args.size), frames=8, seed=0, device=dev)
        x8 = p8["x_T"].repeat(8, 1, 1, 1)
        runner.sample(p8["pose"], ctx, ref, x8, ddim_steps=args.ddim_steps, scale=7.0, decode=not args.no_decode)
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(2):
            runner.sample(p8["pose"], ctx, ref, x8, ddim_steps=args.ddim_steps, scale=7.0, decode=not args.no_decode)
        torch.cuda.synchronize()
        d8 = (time.time() - t0) / 2
        out["extra"] = {"configs[2]": {"workload": "8 frames as one batch, 512x512, 50-step DDIM, decoded frames out", "value": 8 / d8,
                                       "unit": "frames/s", "ms_per_step": 1e3 * d8, "ms_per_ddim_step": 1e3 * d8 / args.ddim_steps,
                                       "steps": 2, "warmup": 1}}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        z_one = runner.sample(pose[:1], ctx, ref, x_T[:1], ddim_steps=args.ddim_steps, scale=7.0) if not args.no_decode else None
        out["cpu_baseline"] = cpu_baseline(model, inp, args.size, z_one,
                                           None if z_one is None else model.decode_first_stage(z_one))
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()   # rank 0 may still have been profiling: leave the group together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
