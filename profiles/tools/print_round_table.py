"""Prints the numbers the README / DESIGN / profiles/README tables quote, straight from profiles/<tag>_bench_*.json.
    python profiles/tools/print_round_table.py r03"""
import json, os, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for n in ("default", "adagml", "f32", "inflight1", "secondary_512x1024", "latency", "b1_inflight3", "c4", "c5_f16", "c5_x3"):
    p = os.path.join(root, f"{tag}_bench_{n}.json")
    if not os.path.exists(p):
        continue
    j = json.load(open(p))
    r = j.get("roofline") or {}
    print(f"{n:20s} {j['value']:8.1f} q/s {j['ms_per_step']:7.2f} ms  attention {r.get('achieved')} TFLOP/s  frac {r.get('frac')}  fp16-peak {r.get('frac_fp16_peak')}  "
          f"pipe busy {r.get('mfma_pipe_busy')}  launch {r.get('avg_launch_ms')} ms  traffic {r.get('traffic')}")
    if n == "default":
        for k, v in (j.get("alt") or {}).items():
            print(f"    alt.{k:22s} {v.get('queries_per_s')} q/s {v.get('ms_per_step')} ms {v.get('error', '')}")
        c, pw = j.get("cpu_baseline") or {}, j.get("power") or {}
        print(f"    cpu {c.get('value')} / median {c.get('value_median')} {c.get('unit')} on {c.get('cores')} cores; power {pw.get('socket_w_mean')} W mean, {pw.get('sclk_mhz_mean')} MHz")
