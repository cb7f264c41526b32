"""Per-launch breakdown of ONE DDIM step of the bench workload (GPU box only): un-captured launches timed by md_prof events,
grouped by kernel family + shape tag.  usage: python tools/step_breakdown.py [frames_per_gpu] > gpurun_out/step_breakdown.txt"""
import collections, os, re, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
dump = tempfile.mktemp(suffix=".tsv")
os.environ["MD_PROF_DUMP"] = dump
import torch  # noqa: E402
import bench  # noqa: E402
from magicdance_amd import synthetic, parallel, ops, _lib  # noqa: E402

fpg = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = torch.device("cuda:0")
model = bench.build_model(dev, 64)
inp = synthetic.synth_inputs((64, 64), frames=fpg, seed=0, device=dev)
runner = parallel.FrameShardedSampler(model)
c, _ = runner._cond(inp["pose"], inp["ctx"], inp["ref"])
from magicdance_amd.ddim import DDIMSampler_ReferenceOnly, FusedStepRunner  # noqa: E402
sampler = DDIMSampler_ReferenceOnly(model)
sampler.make_schedule(50, ddim_eta=0.0, verbose=False)
st = model._fused = FusedStepRunner(model)
with torch.cuda.stream(st.stream):
    st.prepare(c, inp["x_T"].repeat(fpg, 1, 1, 1), sampler, 7.0, table_mode=True)
    st.compute_bank_rows(range(st.S))
    st._launch_sequence()
    st.stream.synchronize()
    st.counter.zero_()
    if os.path.exists(dump):
        os.remove(dump)
    ops.prof_enable(True)
    st._launch_sequence()
    st.stream.synchronize()
    ops.prof_collect()
    ops.prof_enable(False)
agg = collections.OrderedDict()
for line in open(dump):
    fam, ms, fl, by, tag = line.rstrip("\n").split("\t")
    tag = re.sub(r" split=\d+", lambda m: m.group(0), tag)
    key = (_lib.FAMILIES[int(fam)], tag)
    a = agg.setdefault(key, [0, 0.0, 0.0])
    a[0] += 1
    a[1] += float(ms)
    a[2] += float(fl)
tot = sum(a[1] for a in agg.values())
print(f"one DDIM step, {fpg} frame(s): {sum(a[0] for a in agg.values())} launches, {tot:.3f} ms (sum of per-launch events, un-captured)")
fam_tot = collections.Counter()
for (fam, tag), a in agg.items():
    fam_tot[fam] += a[1]
print(" ".join(f"{k}={v:.3f}ms" for k, v in fam_tot.items()))
for (fam, tag), a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{a[1] * 1e3:9.1f} us  x{a[0]:3d}  avg {a[1] * 1e3 / a[0]:7.1f} us  {a[2] / max(a[1], 1e-9) / 1e9:7.0f} TF  {fam:10s} {tag}")
