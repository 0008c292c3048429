import sys, json
for line in sys.stdin:
    line = line.strip()
    if line.startswith("{"):
        d = json.loads(line)
        print(sys.argv[1] if len(sys.argv) > 1 else "", d["config"]["queries_per_gpu_per_step"], d["value"], d["ms_per_step"], d["roofline"]["achieved"], d["roofline"]["launches_per_step"])
