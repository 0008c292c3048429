"""Host enqueue time of one pipeline step vs its GPU time, and a hipGraph capture of the same step."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from pram_amd.pipeline import QueryPipeline
from pram_amd import weights as Wt

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = torch.device("cuda:0")
sfd2, seg, matcher, sds = bench.build_models(dev, "gml", 113)
pipe = QueryPipeline(sfd2, seg, matcher, max_keypoints=2048, min_keypoints=128)
images = torch.stack([Wt.synthetic_image(i) for i in range(B)]).to(dev).contiguous()
with torch.no_grad():
    ex = sfd2.extract_batched(images, pipe.cfg)
    ref = bench.make_reference_sets(ex["descriptors"], ex["keypoints"], ex["scores"], 5000)
for _ in range(3):
    out = pipe.run(images, ref)
torch.cuda.synchronize()
for ov in (0, 8):
    pipe.overlap_below = ov
    for _ in range(3):
        pipe.run(images, ref)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        out = pipe.run(images, ref)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"B={B} overlap_below={ov}: host enqueue {1e3*(t1-t0)/10:.2f} ms/step, total {1e3*(t2-t0)/10:.2f} ms/step")

# graph capture (the product's wrapper)
from pram_amd.pipeline import GraphedPipeline
pipe.overlap_below = 8
try:
    gp = GraphedPipeline(pipe, images, ref)
    for _ in range(3):
        gp.run(images, ref)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        gout = gp.run(images, ref)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"B={B} hipGraph replay: host {1e3*(t1-t0)/10:.2f} ms/step, total {1e3*(t2-t0)/10:.2f} ms/step")
    ok = torch.equal(gout["matches0"], out["matches0"]) and torch.equal(gout["prediction"], out["prediction"])
    print("graph == eager:", ok)
except Exception as e:
    print("graph capture failed:", type(e).__name__, str(e)[:300])
