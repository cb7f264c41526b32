"""Entry-point glue shared by test_any_image_pose.py / test_tiktok.py (repo root): the reference scripts' CLI surface
(test_any_image_pose.py:463-577, scripts/inference_any_image_pose.sh) and output layout
(``local_image_dir/{itr}/gen_images|pose_maps/%03d.jpg``, test_any_image_pose.py:175-180,256-262) on top of the MI355X
hot path, including test_tiktok.py's dataset-driven flow (magicdance_amd/tiktok.py: validation loader, ground-truth VAE round
trip into ``gt_images``, ``condition.jpg``).  When the CLIP text encoder of the YAML cannot be built in this image (its weights
are not shipped) the text context is supplied as a tensor (``--context_embedding``).  Frames of a sequence are sharded over the
ranks in contiguous blocks; JPGs are produced by a GPU uint8 conversion + asynchronous device->host copy + writer thread.
No training and no DDP wrapper (inference never needed them)."""
import argparse
import os

import numpy as np
import torch


def str2bool(v):
    if isinstance(v, bool):
        return v
    return str(v).lower() in ("yes", "true", "t", "y", "1")


def build_parser():
    """Every flag of the reference parser with its default (unused training flags are accepted and ignored)."""
    p = argparse.ArgumentParser()
    p.add_argument("--model_config", type=str, default=None)
    p.add_argument("--reinit_hint_block", action="store_true", default=False)
    p.add_argument("--image_size", type=int, default=64)
    p.add_argument("--empty_text_prob", type=float, default=0.1)
    p.add_argument("--sd_locked", type=str2bool, default=True)
    p.add_argument("--only_mid_control", type=str2bool, default=False)
    p.add_argument("--finetune_all", action="store_true", default=False)
    p.add_argument("--finetune_imagecond_unet", action="store_true", default=False)
    p.add_argument("--control_type", type=str, nargs="+", default=["pose"])
    p.add_argument("--control_dropout", type=float, default=0.0)
    p.add_argument("--depth_bg_threshold", type=float, default=0.0)
    p.add_argument("--inpaint_unet", type=str2bool, default=False)
    p.add_argument("--blank_mask_prob", type=float, default=0.0)
    p.add_argument("--mask_densepose", type=float, default=0.0)
    p.add_argument("--control_mode", type=str, default="balance")
    p.add_argument("--wonoise", action="store_true", default=False)
    p.add_argument("--mask_bg", action="store_true", default=False)
    p.add_argument("--img_bin_limit", default=29)
    p.add_argument("--num_workers", type=int, default=1)
    p.add_argument("--train_batch_size", type=int, default=16)
    p.add_argument("--val_batch_size", type=int, default=1)
    for name, tp, dv in (("--lr", float, 1e-5), ("--lr_sd", float, 1e-5), ("--weight_decay", float, 0), ("--lr_anneal_steps", float, 0),
                         ("--ema_rate", float, 0), ("--num_train_steps", int, 1000000), ("--grad_clip_norm", float, 0.5),
                         ("--gradient_accumulation_steps", int, 1), ("--seed", int, 42), ("--logging_steps", int, 100),
                         ("--logging_gen_steps", int, 1000), ("--save_steps", int, 10000), ("--save_total_limit", int, 100),
                         ("--global_step", int, 0), ("--eta", float, 0.0), ("--gif_time", float, 0.03)):
        p.add_argument(name, type=tp, default=dv)
    p.add_argument("--use_fp16", action="store_true", default=False)
    p.add_argument("--load_optimizer_state", type=str2bool, default=True)
    p.add_argument("--compile", type=str2bool, default=False)
    p.add_argument("--with_text", action="store_false", default=True)
    p.add_argument("--pose_transfer", action="store_true", default=False)
    p.add_argument("--autoreg", action="store_true", default=False)
    p.add_argument("--text_prompt", type=str, default=None)
    p.add_argument("--v4", action="store_true", default=False)
    p.add_argument("--train_dataset", type=str, default="laionhumanDs_densepose_1face_lm")
    p.add_argument("--output_dir", type=str, default=None)
    p.add_argument("--local_log_dir", type=str, default=None)
    p.add_argument("--local_image_dir", type=str, default=None, required=True)
    p.add_argument("--resume_dir", type=str, default=None)
    p.add_argument("--image_pretrain_dir", type=str, default=None)
    p.add_argument("--pose_pretrain_dir", type=str, default=None)
    p.add_argument("--init_path", type=str, default=None)
    p.add_argument("--local_cond_image_path", type=str, default=None, help="Cond image")
    p.add_argument("--local_pose_path", type=str, default=None, help="Pose maps")
    # additions of this build (the reference hard-codes 50 steps / scale 7 in visualize(), :244-245)
    p.add_argument("--ddim_steps", type=int, default=50)
    p.add_argument("--guidance_scale", type=float, default=7.0)
    p.add_argument("--frames_per_batch", type=int, default=1, help="frames sampled together (reference: 1)")
    p.add_argument("--ref_latent", type=str, default=None, help=".pt [1,4,h,w] reference latent (VAE(ref)*scale_factor) when no VAE is built")
    p.add_argument("--context_embedding", type=str, default=None, help='.pt [1,77,768] text context (CLIP("")) when no CLIP is built')
    p.add_argument("--synthetic_weights", action="store_true", help="seeded random weights when no checkpoint is given (plumbing runs)")
    p.add_argument("--tiktok_data_path", type=str, default=None, help="test_tiktok.py: frames root (reference: ./TikTok-v4/disco_test_set)")
    p.add_argument("--tiktok_pose_path", type=str, default=None, help="test_tiktok.py: pose-map root (reference: ./TikTok-v4/pose_map_disco_test_set)")
    return p


def load_square(path, normalize, size=512, return_pil=False):
    """center_crop_to_512 / center_crop_pose_to_512 (test_any_image_pose.py:46-81) and the dataset transforms
    (test_tiktok.py:441-459): torchvision RandomResizedCrop(size, scale=(1,1), ratio=(1,1), BILINEAR) on a PIL image -- its
    get_params draws w = h = round(sqrt(H W)), which fits only a square image (then the crop is the whole image); for any other
    shape all 10 attempts fail and the documented fallback applies: the CENTRED square crop of side min(H, W) -- followed by
    PIL's bilinear resize, ToTensor (+ Normalize(0.5, 0.5) for images).  Restated independently in oracle/preprocess_restatement.py."""
    from PIL import Image
    img = Image.open(path)
    if img.mode != "RGB":
        img = img.convert("RGB")
    w, h = img.size
    s = min(w, h)
    left, top = (w - s) // 2, (h - s) // 2
    crop = img.crop((left, top, left + s, top + s)).resize((size, size), Image.BILINEAR)
    t = torch.from_numpy(np.asarray(crop, dtype=np.float32) / 255.0).permute(2, 0, 1).contiguous()
    t = (t - 0.5) / 0.5 if normalize else t
    return (t, img) if return_pil else t


def _load_square_512(path, normalize):
    return load_square(path, normalize, 512)


def _save_jpg(t, path):
    from PIL import Image
    arr = (t.detach().float().clamp(0, 1).permute(1, 2, 0).cpu().numpy() * 255.0 + 0.5).astype(np.uint8)
    Image.fromarray(arr).save(path)


def _build_model(args, dev):
    import magicdance_amd as M
    from . import synthetic
    from .cldm import _Unavailable
    model = M.create_model(args.model_config or M.DEFAULT_CONFIG)
    if args.image_pretrain_dir and os.path.exists(args.image_pretrain_dir):
        # test_any_image_pose.py:371 loads strict, test_tiktok.py:392 with strict=False
        model.load_state_dict(M.load_state_dict(args.image_pretrain_dir, location="cpu"), strict=not getattr(args, "_tiktok", False))
    elif args.synthetic_weights:
        for pre, mod in (("model.diffusion_model.", model.model.diffusion_model), ("appearance_control_model.", model.appearance_control_model),
                         ("pose_control_model.", model.pose_control_model), ("first_stage_model.", model.first_stage_model)):
            if mod is None or isinstance(mod, _Unavailable):
                continue
            sd = synthetic.synth_state_dict(mod, pre, seed=0)
            mod.load_state_dict({k[len(pre):]: v for k, v in sd.items()}, strict=True)
    else:
        raise FileNotFoundError("--image_pretrain_dir checkpoint not found (use --synthetic_weights for a plumbing run)")
    model = model.to(dev).eval()
    model.only_mid_control = args.only_mid_control
    model.image_size = args.image_size
    return model


def _contexts(args, model, dev):
    """(c_cross, uc_cross): CLIP(text) and CLIP("") (test_any_image_pose.py:198,217-220) when the text encoder is built, else the
    tensor given with --context_embedding (stand-in for CLIP(""), used for both) or a seeded synthetic one."""
    from . import synthetic
    from .cldm import _Unavailable
    have_clip = model.cond_stage_model is not None and not isinstance(model.cond_stage_model, _Unavailable)
    if args.context_embedding:      # explicit tensor (stand-in for CLIP(""), used for both branches)
        ctx = torch.load(args.context_embedding).to(dev).float()
        return ctx, ctx
    if have_clip:
        text = args.text_prompt if args.text_prompt is not None else ""          # test_any_image_pose.py:181-192
        return model.get_learned_conditioning([text]).float(), model.get_unconditional_conditioning(1).float()
    print('[magicdance_amd] no CLIP in this image and no --context_embedding: using a seeded synthetic context')
    ctx = synthetic.synth_inputs((args.image_size, args.image_size), seed=0, device=dev)["ctx"]
    return ctx, ctx


def render_sequence(args, model, sampler, writer, out_dir, cond_image, poses, ctx, uc_ctx, gt_images=None, ref_latent=None):
    """visualize() of the entry scripts (test_any_image_pose.py:155-262, test_tiktok.py:191-288) for ONE reference image and its
    pose sequence: VAE-encode the reference, draw x_T once for all frames (:201-202), sample this rank's contiguous block of
    frames, decode, write ``gen_images / pose_maps (/ gt_images) / condition.jpg`` under ``out_dir``.  Returns this rank's latents."""
    from .cldm import _Unavailable
    dev = model.device
    rank, world = sampler.rank, sampler.world
    h = args.image_size
    have_vae = model.first_stage_model is not None and not isinstance(model.first_stage_model, _Unavailable)
    for sub in ("gen_images", "pose_maps", "latents") + (("gt_images",) if gt_images is not None else ()):
        os.makedirs(os.path.join(out_dir, sub), exist_ok=True)
    if have_vae and cond_image is not None:
        ref = model.get_first_stage_encoding(model.encode_first_stage(cond_image.unsqueeze(0).to(dev)))
        if rank == 0:
            writer.save(cond_image.unsqueeze(0).to(dev).float(), [os.path.join(out_dir, "condition.jpg")])
    else:
        ref = ref_latent
    x_T = torch.randn(1, model.channels, h, h, device=dev)                      # drawn ONCE for all frames (:201-202)
    f0, f1 = sampler.frame_block(len(poses), rank, world)
    my = list(range(f0, f1))
    my_poses = torch.stack([poses[i] for i in my]).to(dev) if my else torch.zeros((0, 3, 8 * h, 8 * h), device=dev)
    if args.control_mode == "controlnet_important" and args.wonoise:
        z = sampler.sample_sequence(my_poses, ctx, ref, x_T, frames_per_batch=args.frames_per_batch, ddim_steps=args.ddim_steps,
                                    scale=args.guidance_scale)
    else:  # balance mode / noisy reference: the reference's per-frame loop through the generic sampler route
        zs = []
        for j in range(len(my)):
            c = {"c_concat": [my_poses[j:j + 1]], "c_crossattn": [ctx], "image_control": [ref], "wonoise": args.wonoise, "overlap_sampling": False}
            uc = {"c_concat": [my_poses[j:j + 1]], "c_crossattn": [uc_ctx], "wonoise": args.wonoise, "overlap_sampling": False}
            if args.control_mode != "controlnet_important":
                uc["image_control"] = [ref]                                         # :221-224
            zi, _ = model.sample_log(cond=c, batch_size=1, ddim=True, ddim_steps=args.ddim_steps, eta=args.eta,
                                     unconditional_guidance_scale=args.guidance_scale, unconditional_conditioning=uc,
                                     inpaint=None, x_T=x_T)
            zs.append(zi)
        z = torch.cat(zs, 0) if zs else torch.zeros((0, model.channels, h, h), device=dev)
    # first-stage decode in batches of frames_per_batch (one VAE pass per batch instead of per frame; a frame decoded inside a batch
    # is bit-identical to the same frame decoded alone, tests/test_gpu_vae.py)
    nb = max(1, int(getattr(args, "frames_per_batch", 1) or 1))
    decoded = {}
    for j, i in enumerate(my):
        if have_vae and j % nb == 0:
            imgs = model.decode_first_stage(z[j:j + nb]).float()
            decoded = {j + k: imgs[k:k + 1] for k in range(imgs.shape[0])}
        torch.save(z[j:j + 1].cpu(), os.path.join(out_dir, "latents", "%03d.pt" % i))
        # the reference saves c_cat[:, :3].clamp(-1, 1).add(1).mul(0.5) (test_any_image_pose.py:257, test_tiktok.py:285): a [0, 1]
        # pose map lands in [0.5, 1] -- reproduced as is, the output tree is part of the drop-in surface
        writer.save(my_poses[j:j + 1].float(), [os.path.join(out_dir, "pose_maps", "%03d.jpg" % i)], value_range=(-1.0, 1.0))
        if have_vae:
            writer.save(decoded.pop(j), [os.path.join(out_dir, "gen_images", "%03d.jpg" % i)])
            if gt_images is not None:   # VAE round trip of the ground-truth frame (test_tiktok.py:273-279)
                real = gt_images[i].unsqueeze(0).to(dev)
                rec = model.decode_first_stage(model.get_first_stage_encoding(model.encode_first_stage(real)))
                writer.save(rec.float(), [os.path.join(out_dir, "gt_images", "%03d.jpg" % i)])
    print(f"[magicdance_amd] rank {rank}: frames [{f0}, {f1}) of {len(poses)} -> {out_dir}" + ("" if have_vae else " (latents only: no VAE in this image)"))
    return z


# test_any_image_pose.py:237,533 -- without --use_fp16 the reference samples in fp32.  This build has ONE arithmetic (fp16 MFMA
# operands, fp32 accumulation, two-term / fp32 residual stream) and no fp32-class kernel set: a request for the reference's fp32
# arithmetic is REFUSED rather than silently answered in fp16 (round 5; rounds 1-3 ignored the flag, round 4 warned).
FP32_REFUSAL = ("--use_fp16 is absent: the reference would sample in fp32 (test_any_image_pose.py:237), which this MI355X build does "
                "not implement -- it always computes with fp16 matrix-core operands and fp32 accumulation (within 0.6-0.9x of the "
                "deviation the reference's own --use_fp16 mode shows against its fp32 arithmetic, DESIGN.md section 2).  Pass "
                "--use_fp16 (the shipped configuration, scripts/inference_any_image_pose.sh:9) to run.")


def run(args, need_dataset=False):
    from . import parallel, synthetic, tiktok
    from .cldm import _Unavailable
    args._tiktok = need_dataset
    use_dataset = args.local_cond_image_path is None or args.local_pose_path is None
    if use_dataset and not need_dataset:
        raise ValueError("test_any_image_pose.py needs --local_cond_image_path and --local_pose_path (test_batch_data is None there, "
                         "test_any_image_pose.py:453)")
    if not args.use_fp16:
        raise ValueError(FP32_REFUSAL)
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(dev)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)
    torch.manual_seed(args.seed)
    model = _build_model(args, dev)
    h = args.image_size
    have_vae = model.first_stage_model is not None and not isinstance(model.first_stage_model, _Unavailable)
    ctx, uc_ctx = _contexts(args, model, dev)
    sampler = parallel.FrameShardedSampler(model, rank=rank, world=world)
    writer = tiktok.AsyncImageWriter(dev)
    ref_latent = None
    if not have_vae:
        ref_latent = torch.load(args.ref_latent).to(dev).float() if args.ref_latent else \
            synthetic.synth_inputs((h, h), seed=0, device=dev)["ref"]
    z = None
    try:
        if use_dataset:
            # test_tiktok.py:461-517: one subject folder per iteration, frame 0 = reference image, frames 1.. = targets
            ds = tiktok.tiktok_video_arnold_val(**({"data_path": args.tiktok_data_path, "pose_path": args.tiktok_pose_path}
                                                   if args.tiktok_data_path else {}), rank=rank, world_size=world,
                                                img_bin_limit=args.img_bin_limit, image_size=8 * h)
            it = iter(ds)
            for itr in range(min(args.num_train_steps, len(ds))):
                batch = next(it, None)
                if batch is None:
                    break
                z = render_sequence(args, model, sampler, writer, os.path.join(args.local_image_dir, str(itr)), batch["condition_image"],
                                    batch["pose_map_list"], ctx, uc_ctx, gt_images=batch["image_list"], ref_latent=ref_latent)
        else:
            pose_files = sorted(os.listdir(args.local_pose_path))
            poses = [load_square(os.path.join(args.local_pose_path, f), False, 8 * h) for f in pose_files]
            cond = load_square(args.local_cond_image_path, True, 8 * h)
            z = render_sequence(args, model, sampler, writer, os.path.join(args.local_image_dir, "0"), cond, poses, ctx, uc_ctx,
                                ref_latent=ref_latent)
    finally:
        writer.close()
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
    return z
