"""Timeline of the LAST frame batch of a rocprofv3 --kernel-trace CSV of bench.py (GPU box, after a kernel-trace run): wall time of
the batch (from the previous batch's last kernel to this batch's last kernel), the union of kernel intervals (busy time; kernels of the
table stream and of the step stream may overlap), idle time, the largest idle gaps with their position in the batch and the kernels
either side, and the marks: first table-pass kernel, first / last DDIM update, decode.
usage: python tools/batch_timeline.py gpurun_out/<dir>/kt_kernel_trace.csv > gpurun_out/<name>.txt"""
import csv
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:60]))
rows.sort()
upd = [i for i, r in enumerate(rows) if "ddim_update" in r[2]]
S = 50
nb = len(upd) // S
print(f"{len(rows)} kernels, {len(upd)} ddim_update launches = {nb} batches of {S} steps (incl. warm-up / capture executions)")
# batch b = kernels after the last kernel of batch b-1 (its image_to_u8 / last decode kernel) up to the next one: delimit by the
# LAST ddim_update of each batch + everything until the following batch's first kernel that is not part of the decode
ends = []
for b in range(nb):
    last_upd = upd[(b + 1) * S - 1]
    nxt = upd[(b + 1) * S - 1 + 1] if (b + 1) * S < len(upd) else len(rows)
    # the decode of batch b = kernels after last_upd until the first kernel of the next batch's table pass; take the largest start
    # gap... simpler: the batch ends at the last kernel that STARTS before the next batch's first select_row / gather
    ends.append((last_upd, nxt))
b = nb - 1
lo = upd[(b - 1) * S + S - 1]          # last update of the previous batch
hi = upd[b * S + S - 1]                # last update of this batch
seg = rows[lo:hi + 1]
t0, t1 = seg[0][1], seg[-1][1]
print(f"last batch, measured between the last DDIM updates of consecutive batches: wall {1e-6 * (t1 - t0):.2f} ms, {len(seg) - 1} kernels")
# union of intervals
busy, cur_s, cur_e = 0, None, None
gaps = []
prev_end, prev_name = t0, seg[0][2]
for s, e, name in seg[1:]:
    if s > prev_end:
        gaps.append((s - prev_end, prev_end - t0, prev_name, name))
    if e > prev_end:
        busy += e - max(s, prev_end)
        prev_end, prev_name = e, name
print(f"busy (union of kernel intervals) {1e-6 * busy:.2f} ms = {100.0 * busy / (t1 - t0):.1f} % of wall; idle {1e-6 * (t1 - t0 - busy):.2f} ms in {len(gaps)} gaps")
ksum = sum(e - s for s, e, _ in seg[1:])
print(f"sum of kernel durations {1e-6 * ksum:.2f} ms (> busy where the table stream overlaps the step stream)")
first_upd = upd[b * S]
print(f"first DDIM update of the batch at +{1e-6 * (rows[first_upd][1] - t0):.2f} ms; steps 1..49 take {1e-6 * (t1 - rows[first_upd][1]):.2f} ms "
      f"= {1e-6 * (t1 - rows[first_upd][1]) / 49:.3f} ms per step")
gaps.sort(reverse=True)
print("largest idle gaps: us, at ms into the batch, kernel before -> kernel after")
for g, at, a, c in gaps[:25]:
    print(f"{g / 1e3:9.1f} us  +{at / 1e6:8.2f} ms   {a}  ->  {c}")
small = sum(g for g, _, _, _ in gaps if g < 5000)
print(f"gaps < 5 us: {1e-6 * small:.2f} ms in total; gaps >= 5 us: {1e-6 * sum(g for g, _, _, _ in gaps if g >= 5000):.2f} ms in {sum(1 for g in gaps if g[0] >= 5000)} gaps")
