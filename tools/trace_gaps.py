"""In-graph timeline of ONE steady-state DDIM step from a rocprofv3 --kernel-trace CSV (GPU box, after tools/run_profiles.sh):
kernel time vs idle gaps between consecutive kernels of the step, and the average duration per kernel name inside the replayed
graph (the un-captured per-launch event times of tools/step_breakdown.py include eager launch latency).
usage: python tools/trace_gaps.py gpurun_out/<dir>/kt_kernel_trace.csv > gpurun_out/<name>.txt"""
import collections
import csv
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# DDIM steps end with the fused CFG / DDIM update kernel; take the LAST frame's steps 25..45 (the reference-KV table pass, which
# overlaps the first steps on its own stream, is over by then) -> every kernel between two ddim_update launches is one step
idx = [i for i, r in enumerate(rows) if "ddim_update" in r[2]]
if len(idx) < 50:
    print("fewer than 50 DDIM steps in the trace")
    sys.exit(0)
sel = idx[-26:-5]
tot_k = tot_gap = tot_wall = 0.0
n_k = 0
per = collections.defaultdict(lambda: [0, 0.0])
gaps = []
for a, b in zip(sel[:-1], sel[1:]):
    ks = rows[a + 1:b + 1]
    t_prev_end = rows[a][1]
    wall = ks[-1][1] - rows[a][1]
    tot_wall += wall
    for s, e, name in ks:
        tot_k += e - s
        n_k += 1
        g = s - t_prev_end
        gaps.append(g)
        tot_gap += max(g, 0)
        t_prev_end = max(t_prev_end, e)
        short = name.replace("(anonymous namespace)::", "").replace("void ", "")[:70]
        per[short][0] += 1
        per[short][1] += e - s
steps = len(sel) - 1
print(f"{steps} steady-state DDIM steps of the last frame (profiled run: kernels serialised by the tracer)")
print(f"per step: {n_k / steps:.0f} kernels, wall {tot_wall / steps / 1e3:.1f} us, kernel time {tot_k / steps / 1e3:.1f} us, "
      f"idle gaps {tot_gap / steps / 1e3:.1f} us ({100 * tot_gap / tot_wall:.1f} % of wall), mean gap {tot_gap / n_k / 1e3:.2f} us")
gaps.sort()
print("gap percentiles (us): " + " ".join(f"p{p}={gaps[int(len(gaps) * p / 100)] / 1e3:.2f}" for p in (10, 50, 90, 99)))
print("# per kernel name: launches per step, avg us, us per step")
for name, (c, t) in sorted(per.items(), key=lambda kv: -kv[1][1])[:45]:
    print(f"{c / steps:7.1f} {t / c / 1e3:9.2f} {t / steps / 1e3:9.1f}  {name}")
