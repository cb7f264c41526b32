"""Entry-point glue shared by test_any_image_pose.py / test_tiktok.py (repo root): the reference scripts' CLI surface
(test_any_image_pose.py:463-577, scripts/inference_any_image_pose.sh) and output layout
(``local_image_dir/{itr}/gen_images|pose_maps/%03d.jpg``, test_any_image_pose.py:175-180,256-262) on top of the MI355X
hot path.  Image I/O, VAE and CLIP are outside this round's scope: when the VAE / CLIP of the YAML cannot be built in
this image the reference latent / text context must be supplied as tensors (``--ref_latent``, ``--context_embedding``) and
latents are written instead of JPGs.  No training, no dataset loader, no DDP wrapper (inference never needed them)."""
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
    return p


def _load_square_512(path, normalize):
    """center_crop_to_512 / center_crop_pose_to_512 (test_any_image_pose.py:46-81): RandomResizedCrop(512, scale=(1,1),
    ratio=(1,1)) degenerates to the centred square crop of side min(h, w), bilinear resize to 512, ToTensor
    (+ Normalize(0.5, 0.5) for the reference image)."""
    from PIL import Image
    img = Image.open(path)
    if img.mode != "RGB":
        img = img.convert("RGB")
    w, h = img.size
    s = min(w, h)
    left, top = (w - s) // 2, (h - s) // 2
    img = img.crop((left, top, left + s, top + s)).resize((512, 512), Image.BILINEAR)
    t = torch.from_numpy(np.asarray(img, dtype=np.float32) / 255.0).permute(2, 0, 1).contiguous()
    return (t - 0.5) / 0.5 if normalize else t


def _save_jpg(t, path):
    from PIL import Image
    arr = (t.detach().float().clamp(0, 1).permute(1, 2, 0).cpu().numpy() * 255.0 + 0.5).astype(np.uint8)
    Image.fromarray(arr).save(path)


def run(args, need_dataset=False):
    import magicdance_amd as M
    from . import parallel, synthetic
    from .cldm import _Unavailable
    if args.local_cond_image_path is None or args.local_pose_path is None:
        raise NotImplementedError("the TikTok dataset loader (dataset/tiktok_video_arnold_copy.py) is outside this build: pass "
                                  "--local_cond_image_path and --local_pose_path" + (" (test_tiktok.py falls back to the dataset)" if need_dataset else ""))
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(dev)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)
    torch.manual_seed(args.seed)
    model = M.create_model(args.model_config or M.DEFAULT_CONFIG)
    if args.image_pretrain_dir and os.path.exists(args.image_pretrain_dir):
        model.load_state_dict(M.load_state_dict(args.image_pretrain_dir, location="cpu"), strict=True)
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
    h = args.image_size
    model.image_size = h
    have_vae = model.first_stage_model is not None and not isinstance(model.first_stage_model, _Unavailable)
    have_clip = model.cond_stage_model is not None and not isinstance(model.cond_stage_model, _Unavailable)
    pose_files = sorted(os.listdir(args.local_pose_path))
    poses = torch.stack([_load_square_512(os.path.join(args.local_pose_path, f), normalize=False) for f in pose_files]).to(dev)
    if have_vae:
        ref_img = _load_square_512(args.local_cond_image_path, normalize=True).unsqueeze(0).to(dev)
        ref = model.get_first_stage_encoding(model.encode_first_stage(ref_img))
    elif args.ref_latent:
        ref = torch.load(args.ref_latent).to(dev).float()
    else:
        print("[magicdance_amd] no VAE in this image and no --ref_latent: using a seeded synthetic reference latent")
        ref = synthetic.synth_inputs((h, h), seed=0, device=dev)["ref"]
    if have_clip:
        ctx = model.get_learned_conditioning([args.text_prompt or ""])
    elif args.context_embedding:
        ctx = torch.load(args.context_embedding).to(dev).float()
    else:
        print('[magicdance_amd] no CLIP in this image and no --context_embedding: using a seeded synthetic context')
        ctx = synthetic.synth_inputs((h, h), seed=0, device=dev)["ctx"]
    x_T = torch.randn(1, model.channels, h, h, device=dev)                      # drawn ONCE for all frames (:201-202)
    my = list(range(len(pose_files)))[rank::world] if world > 1 else list(range(len(pose_files)))
    sampler = parallel.FrameShardedSampler(model, rank=rank, world=world)
    out_dir = os.path.join(args.local_image_dir, "0")
    for sub in ("gen_images", "pose_maps", "latents"):
        os.makedirs(os.path.join(out_dir, sub), exist_ok=True)
    if args.control_mode == "controlnet_important" and args.wonoise:
        z = sampler.sample_sequence(poses[my], ctx, ref, x_T, frames_per_batch=args.frames_per_batch,
                                    ddim_steps=args.ddim_steps, scale=args.guidance_scale)
    else:  # balance mode / noisy reference: the reference's per-frame loop through the generic sampler route
        zs = []
        for i in my:
            c = {"c_concat": [poses[i:i + 1]], "c_crossattn": [ctx], "image_control": [ref], "wonoise": args.wonoise, "overlap_sampling": False}
            uc = {"c_concat": [poses[i:i + 1]], "c_crossattn": [ctx], "wonoise": args.wonoise, "overlap_sampling": False}
            if args.control_mode != "controlnet_important":
                uc["image_control"] = [ref]
            zi, _ = model.sample_log(cond=c, batch_size=1, ddim=True, ddim_steps=args.ddim_steps, eta=args.eta,
                                     unconditional_guidance_scale=args.guidance_scale, unconditional_conditioning=uc,
                                     inpaint=None, x_T=x_T)
            zs.append(zi)
        z = torch.cat(zs, 0)
    for j, i in enumerate(my):
        torch.save(z[j:j + 1].cpu(), os.path.join(out_dir, "latents", "%03d.pt" % i))
        _save_jpg(poses[i], os.path.join(out_dir, "pose_maps", "%03d.jpg" % i))
        if have_vae:
            img = model.decode_first_stage(z[j:j + 1])
            _save_jpg((img[0] + 1.0) / 2.0, os.path.join(out_dir, "gen_images", "%03d.jpg" % i))
    print(f"[magicdance_amd] rank {rank}: {len(my)} frame(s) -> {out_dir}" + ("" if have_vae else " (latents only: no VAE in this image)"))
    return z
