"""Per-launch breakdown of ONE first-stage decode of the bench workload (GPU box only): un-captured launches timed by md_prof events,
grouped by kernel family + shape tag.  usage: python tools/decode_breakdown.py [frames] > gpurun_out/decode_breakdown.txt"""
import collections, os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
dump = tempfile.mktemp(suffix=".tsv")
os.environ["MD_PROF_DUMP"] = dump
import torch  # noqa: E402
import bench  # noqa: E402
from magicdance_amd import ops, _lib  # noqa: E402

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = torch.device("cuda:0")
model = bench.build_model(dev, 64)
z = torch.randn(frames, 4, 64, 64, device=dev)
for _ in range(2):
    model.decode_first_stage(z)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    model.decode_first_stage(z)
e1.record()
torch.cuda.synchronize()
wall = e0.elapsed_time(e1) / 5
if os.path.exists(dump):
    os.remove(dump)
ops.prof_enable(True)
model.decode_first_stage(z)
torch.cuda.synchronize()
ops.prof_collect()
ops.prof_enable(False)
agg = collections.OrderedDict()
for line in open(dump):
    fam, ms, fl, by, tag = line.rstrip("\n").split("\t")
    key = (_lib.FAMILIES[int(fam)], tag)
    a = agg.setdefault(key, [0, 0.0, 0.0])
    a[0] += 1
    a[1] += float(ms)
    a[2] += float(fl)
tot = sum(a[1] for a in agg.values())
print(f"first-stage decode, {frames} frame(s) 64x64 -> 512x512: {wall:.3f} ms per call (back to back, torch events); "
      f"{sum(a[0] for a in agg.values())} launches, {tot:.3f} ms (sum of per-launch events)")
fam_tot = collections.Counter()
for (fam, tag), a in agg.items():
    fam_tot[fam] += a[1]
print(" ".join(f"{k}={v:.3f}ms" for k, v in fam_tot.items()))
for (fam, tag), a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{a[1] * 1e3:9.1f} us  x{a[0]:3d}  avg {a[1] * 1e3 / a[0]:7.1f} us  {a[2] / max(a[1], 1e-9) / 1e9:7.0f} TF  {fam:10s} {tag}")
