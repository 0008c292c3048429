"""Which torch ops issue device copies in one pipeline step (looking for avoidable hipMemcpy / copy kernels)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from pram_amd.pipeline import QueryPipeline
from pram_amd import weights as Wt
from torch.profiler import profile, ProfilerActivity

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
dev = torch.device("cuda:0")
sfd2, seg, matcher, sds = bench.build_models(dev, "gml", 113)
pipe = QueryPipeline(sfd2, seg, matcher, max_keypoints=2048, min_keypoints=128, overlap_below=0)
images = torch.stack([Wt.synthetic_image(i) for i in range(B)]).to(dev).contiguous()
with torch.no_grad():
    ex = sfd2.extract_batched(images, pipe.cfg)
    ref = bench.make_reference_sets(ex["descriptors"], ex["keypoints"], ex["scores"], 5000)
for _ in range(2):
    pipe.run(images, ref)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    out = pipe.run(images, ref)
    rec = QueryPipeline.pack_record(out)
    torch.cuda.synchronize()
from collections import Counter
c = Counter()
for ev in prof.events():
    if ev.name in ("aten::copy_", "aten::_to_copy", "aten::contiguous", "aten::clone", "aten::cat", "aten::fill_", "aten::zero_", "aten::index", "aten::slice", "aten::empty", "aten::to"):
        st = [s for s in (ev.stack or []) if "pram_amd" in s or "bench" in s]
        c[(ev.name, str(ev.input_shapes)[:60], st[0][-70:] if st else "?")] += 1
for k, v in c.most_common(40):
    if k[0] in ("aten::copy_", "aten::cat", "aten::fill_", "aten::zero_", "aten::index", "aten::clone"):
        print(v, k)
