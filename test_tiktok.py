"""Drop-in for the reference entry point test_tiktok.py.  From sample_log down it is identical to test_any_image_pose.py
(SURVEY appendix A); the TikTok validation-set loader is outside this build, so the reference image / pose folder must be
given with --local_cond_image_path / --local_pose_path (the reference script accepts the same override, :156-170)."""
from magicdance_amd import entry

if __name__ == "__main__":
    entry.run(entry.build_parser().parse_args(), need_dataset=True)
