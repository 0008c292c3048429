"""Per-kernel means of the rocprofv3 --pmc passes written by pmc_collect.sh.
    python profiles/tools/pmc_summarize.py <pmc_dir> <tag>  ->  profiles/<tag>_pmc_summary.md, profiles/<tag>_pmc_attention.json
gfx950 corrections follow /opt/skills/guides/MI355X_MICROARCH.md (HBM / rocprofv3 section): FETCH_SIZE is in KB and
under-reports by 2x on gfx950 (hbm_read = 2 * FETCH_SIZE KB); WRITE_SIZE is KB (uncalibrated);
MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs)."""
import csv, glob, json, os, re, sys
from collections import defaultdict

src, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
vals = defaultdict(lambda: defaultdict(list))      # kernel -> counter -> values
dur = defaultdict(list)
for f in glob.glob(os.path.join(src, "*", "*counter_collection.csv")):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = re.sub(r"^void |\(anonymous namespace\)::|\(.*$", "", r["Kernel_Name"]).strip()
        mm = re.match(r"_ZN12_GLOBAL__N_1\d+([a-z0-9_]+?_kernel)(ILi(\d)ELi(\d)E)?", k)      # names the demangler gives up on (_Float16 pointers)
        if mm:
            k = mm.group(1) + (f"<{mm.group(3)}, {mm.group(4)}>" if mm.group(2) else "")
        vals[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if r["Dispatch_Id"] not in seen and "FETCH_SIZE" in f:
            seen.add(r["Dispatch_Id"])
            dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
mean = lambda v: sum(v) / len(v) if v else float("nan")
rows = []
for k, c in vals.items():
    g = lambda n: mean(c.get(n, []))
    gui = g("GRBM_GUI_ACTIVE")
    rows.append(dict(kernel=k, launches=len(c.get("FETCH_SIZE", [])), avg_us=mean(dur[k]),
                     mfma=100.0 * g("SQ_VALU_MFMA_BUSY_CYCLES") / (gui / 8 * 1024) if gui == gui and gui > 0 else float("nan"),
                     fetch_kb=g("FETCH_SIZE"), write_kb=g("WRITE_SIZE"),
                     lds=g("SQ_LDS_BANK_CONFLICT") / g("SQ_ACTIVE_INST_LDS") if g("SQ_ACTIVE_INST_LDS") > 0 else float("nan"),
                     wait=g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES") if g("SQ_WAVE_CYCLES") > 0 else float("nan"),
                     total_us=sum(dur[k])))
rows.sort(key=lambda r: -r["total_us"])
with open(os.path.join(root, "profiles", f"{tag}_pmc_summary.md"), "w") as o:
    o.write(f"# {tag} — PMC summary (per-dispatch means; `profiles/tools/pmc_collect.sh` + `pmc_summarize.py`)\n\n"
            "`rocprofv3 --kernel-trace --pmc <set> -- python bench.py --steps 1 --warmup 1 --cpu-queries 0`, one counter set per run.\n"
            "`MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 x 1024)`; `hbm_read = 2 x FETCH_SIZE KB` (gfx950 correction); "
            "`hbm_write = WRITE_SIZE KB` (uncalibrated).\n\n"
            "| kernel | launches | avg us | MfmaUtil % | FETCH_SIZE KB | hbm_read MiB | WRITE_SIZE KB | HBM GB/s (read+write) | SQ_LDS_BANK_CONFLICT / SQ_ACTIVE_INST_LDS | SQ_WAIT_INST_ANY / WAVE_CYCLES |\n|---|---|---|---|---|---|---|---|---|---|\n")
    for r in rows[:32]:
        o.write(f"| {r['kernel'][:60]} | {r['launches']} | {r['avg_us']:.1f} | {r['mfma']:.1f} | {r['fetch_kb']:.0f} | "
                f"{2 * r['fetch_kb'] / 1024:.1f} | {r['write_kb']:.0f} | {(2 * r['fetch_kb'] + r['write_kb']) * 1024 / (r['avg_us'] * 1e3):.0f} | {r['lds']:.2f} | {r['wait']:.2f} |\n")
# bench.py reads roofline.traffic from profiles/pmc_attention.json, keyed by the precision of the run (argv[3], default x3)
prec = sys.argv[3] if len(sys.argv) > 3 else "x3"
name = {"f32": "attention_kernel", "x3": "attention_x3_pipe_kernel", "f16": "attention_x3_pipe_kernel"}[prec]
att = next((r for r in rows if r["kernel"].startswith(name)), None)
if att:
    path = os.path.join(root, "profiles", "pmc_attention.json")
    js = json.load(open(path)) if os.path.exists(path) else {}
    import hashlib
    src = {"x3": "attention_x3.hip", "f32": "attention.hip", "f16": "attention_x3.hip"}[prec]
    sha = hashlib.sha256(open(os.path.join(root, "pram_amd", "csrc", src), "rb").read()).hexdigest()[:16]
    # bench.py prints roofline.traffic only while the kernel source it runs hashes to this value: a stale counter is never reported
    js[prec] = {"kernel": name, "source": f"profiles/{tag}_pmc_summary.md", "kernel_source": f"pram_amd/csrc/{src}", "kernel_source_sha16": sha,
                "hbm_bytes_per_launch": (2 * att["fetch_kb"] + att["write_kb"]) * 1024,
                "fetch_size_kb": att["fetch_kb"], "write_size_kb": att["write_kb"], "mfma_util_pct": att["mfma"], "launches": att["launches"]}
    json.dump(js, open(path, "w"), indent=1)
    print(json.dumps(js[prec]))
