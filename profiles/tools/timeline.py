"""Kernel timeline of the last bench step from a rocprofv3 --kernel-trace CSV: span, busy time, idle gaps, top kernels.
    python profiles/tools/timeline.py <kernel_trace.csv> [n_last_steps_attention_launches=33]"""
import csv, re, sys
from collections import defaultdict
rows = list(csv.DictReader(open(sys.argv[1])))
per = int(sys.argv[2]) if len(sys.argv) > 2 else 33
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: re.sub(r"^void |\(anonymous namespace\)::|\(.*$", "", r["Kernel_Name"]).strip()[:44]
att = [i for i, r in enumerate(rows) if name(r).startswith("attention_kernel")]
# the instrumented probe step is last; take the step before it: attention launches [-2*per, -per)
lo, hi = att[-2 * per], att[-per]
# widen to the step boundaries: from the first conv of that step (search backwards for nchw3_to_nhwc4)
while lo > 0 and not name(rows[lo]).startswith("nchw3_to_nhwc4"):
    lo -= 1
while hi > 0 and not name(rows[hi]).startswith("nchw3_to_nhwc4"):
    hi -= 1
step = rows[lo:hi]
t0, t1 = int(step[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in step)
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in step)
# union of intervals = time at least one kernel runs
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in step)
union, cs, ce = 0, iv[0][0], iv[0][1]
for s, e in iv[1:]:
    if s > ce:
        union += ce - cs
        cs, ce = s, e
    else:
        ce = max(ce, e)
union += ce - cs
print(f"step span {1e-6*(t1-t0):.3f} ms, {len(step)} kernels, sum of durations {1e-6*busy:.3f} ms, time with >=1 kernel running {1e-6*union:.3f} ms, idle {1e-6*(t1-t0-union):.3f} ms")
agg = defaultdict(lambda: [0, 0])
for r in step:
    a = agg[name(r)]
    a[0] += 1
    a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for k, (n, d) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:16]:
    print(f"  {k:46s} {n:4d} x {1e-3*d/n:8.1f} us = {1e-6*d:7.3f} ms")
q = defaultdict(int)
for r in step:
    q[r.get("Queue_Id", "?")] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
print("  per queue busy ms:", {k: round(1e-6 * v, 3) for k, v in q.items()})
