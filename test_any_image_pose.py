"""Drop-in for the reference entry point test_any_image_pose.py (1 reference image + a folder of pose maps -> frames):
same flags as scripts/inference_any_image_pose.sh, MI355X hot path underneath.  See magicdance_amd/entry.py."""
from magicdance_amd import entry

if __name__ == "__main__":
    entry.run(entry.build_parser().parse_args())
