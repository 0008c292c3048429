"""Copies what profiles/tools/collect_round.sh left under gpurun_out/<tag>/ into profiles/ (tracked), summarises the PMC passes
and writes the per-step kernel anatomy of the traced runs.
    python profiles/tools/install_round.py r03"""
import csv, json, os, re, shutil, subprocess, sys
from collections import defaultdict

tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src, dst = os.path.join(root, "gpurun_out", tag), os.path.join(root, "profiles")
for f in sorted(os.listdir(src)):
    if f.startswith("bench_") and f.endswith(".json") and os.path.getsize(os.path.join(src, f)) > 0:
        shutil.copy(os.path.join(src, f), os.path.join(dst, f"{tag}_{f}"))
    elif f.endswith(".txt"):
        shutil.copy(os.path.join(src, f), os.path.join(dst, f"{tag}_{f}"))


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    m = re.match(r"_ZN12_GLOBAL__N_1\d+([a-z0-9_]+?_kernel)I?([0-9A-Za-z]*)", n)
    if m:
        return m.group(1) + "<" + ",".join(re.findall(r"L[ib](\d+)E", m.group(2))) + ">"
    return re.sub(r"\(.*", "", n)[:80]


def anatomy(kt_dir, name, title):
    tr = os.path.join(src, kt_dir, "kt_kernel_trace.csv")
    st = os.path.join(src, kt_dir, "kt_kernel_stats.csv")
    if not os.path.exists(tr):
        return
    shutil.copy(st, os.path.join(dst, f"{tag}_{name}_kernel_stats.csv"))
    rows = sorted(csv.DictReader(open(tr)), key=lambda r: int(r["Start_Timestamp"]))
    # one steady-state step: from the first kernel of one extraction to the next (the frame repack kernel until the fused conv1 read
    # the image itself, that kernel since)
    starts = [i for i, r in enumerate(rows) if "nchw3_to_nhwc4" in r["Kernel_Name"]]
    if len(starts) < 3:
        starts = [i for i, r in enumerate(rows) if "conv1ab_x3_kernel" in r["Kernel_Name"]]
    a, b = starts[-3], starts[-2]
    agg, busy = defaultdict(lambda: [0, 0.0]), 0.0
    for r in rows[a:b]:
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        busy += d
        k = short(r["Kernel_Name"])
        agg[k][0] += 1
        agg[k][1] += d
    wall = int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"])
    with open(os.path.join(dst, f"{tag}_{name}_step_anatomy.md"), "w") as o:
        o.write(f"# {tag} — {title}\n\nOne steady-state step cut out of `{tag}_{name}_kernel_stats.csv`'s trace (from the first kernel of one extraction to the next): "
                f"{b - a} launches, {busy / 1e6:.2f} ms of kernel time, {wall / 1e6:.2f} ms wall under the profiler.\n\n"
                "| kernel | launches | us per step | avg us | share |\n|---|---|---|---|---|\n")
        for k, (c, d) in sorted(agg.items(), key=lambda x: -x[1][1]):
            o.write(f"| {k} | {c} | {d / 1e3:.1f} | {d / c / 1e3:.1f} | {100 * d / busy:.1f} % |\n")
        fw = [k for k in agg if k.startswith("at::") or "rocprim" in k or "hipcub" in k]
        o.write(f"\nFramework kernels in the step: {len(fw)} kinds ({sum(agg[k][0] for k in fw)} launches); "
                f"`__amd_rocclr_copyBuffer` {agg.get('__amd_rocclr_copyBuffer', [0])[0]}, `__amd_rocclr_fillBufferAligned` {agg.get('__amd_rocclr_fillBufferAligned', [0])[0]}.\n")


anatomy("kt", "bench", "step anatomy of the default bench (16 queries, one batch in flight)")
anatomy("kt_latency", "latency", "step anatomy of the one-query latency mode (eager launches)")
if os.path.isdir(os.path.join(src, "pmc")):
    subprocess.run([sys.executable, os.path.join(root, "profiles", "tools", "pmc_summarize.py"), os.path.join(src, "pmc"), tag], check=False)
print("installed:", sorted(f for f in os.listdir(dst) if f.startswith(tag)))
