"""Drop-in for the reference entry point test_tiktok.py: one subject folder of the TikTok validation set per iteration (first
frame = reference image, remaining frames = targets; magicdance_amd/tiktok.py mirrors dataset/tiktok_video_arnold_copy.py:217-280),
generated frames under gen_images/, the VAE round trip of the ground-truth frames under gt_images/ (test_tiktok.py:273-279),
pose_maps/ and condition.jpg.  --local_cond_image_path / --local_pose_path override the dataset exactly as in the reference
(:156-170).  From sample_log down it is the same MI355X hot path as test_any_image_pose.py."""
from magicdance_amd import entry

if __name__ == "__main__":
    entry.run(entry.build_parser().parse_args(), need_dataset=True)
