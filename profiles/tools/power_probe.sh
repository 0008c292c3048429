#!/bin/bash
# sample socket power and clocks while a kernel loop runs
cd $GRAFT_REPO_ROOT
python - <<'PY' &
import sys, time, torch
sys.path.insert(0, ".")
from pram_amd import ops
dev = torch.device("cuda:0")
def planes(t):
    s = t * 16.0; hi = s.half(); return hi.contiguous(), (s - hi.float()).half().contiguous()
x = torch.randn(65536, 512, device=dev); w = torch.randn(256, 512, device=dev) / 22; b = torch.randn(256, device=dev)
q = torch.randn(32 * 2048, 256, device=dev); pq = planes(q); vt = ops.value_planes_t(planes(q), 32, 4, 2048)
a = torch.randn(1 << 28, device=dev); c = torch.empty_like(a)
for name, f in (("idle", None), ("gemm x3 65536x512->256", lambda: ops.linear(x, w, b, precision="x3")),
                ("attention x3 B32 N2048", lambda: ops.attention_x3(pq, pq, vt, 32, 4, 2048, 2048, 0.125)),
                ("copy 1 GB", lambda: c.copy_(a))):
    t0 = time.time()
    print("PHASE", name, flush=True)
    while time.time() - t0 < 4.0:
        if f is None:
            time.sleep(0.1)
        else:
            for _ in range(50): f()
            torch.cuda.synchronize()
print("PHASE done", flush=True)
PY
PID=$!
for i in $(seq 1 40); do
  sleep 0.45
  rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk|mclk" | tr '\n' ' ' | sed 's/GPU\[0\]//g; s/\t//g; s/  */ /g'
  echo
done
wait $PID
