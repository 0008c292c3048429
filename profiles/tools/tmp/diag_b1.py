import sys, torch
sys.path.insert(0, "/root/repo")
from pram_amd import ops
from tests import helpers as H
from pram_amd.nets.gml import GML
dev = torch.device("cuda:0")
net = GML({}); net.load_state_dict(H.gml_sd(), strict=True); net = net.to(dev).eval()
ds = [H.pair_data(i, 2048, 2048, device=dev)[0] for i in (1, 2, 3, 4, 5)]
cat = {k: torch.cat([d[k] for d in ds], 0) for k in ds[0] if torch.is_tensor(ds[0][k])}
cat["image_shape0"] = cat["image_shape1"] = (1, 3, 640, 480)
for split in (True, False):
    ops.attention_split = split
    rb = net.produce_matches(cat, p=0.0)
    for i, d in enumerate(ds[:2]):
        r1 = net.produce_matches(d, p=0.0)
        dm = int((rb["matches0"][i] != r1["matches0"][0]).sum()); dsn = float((rb["matching_scores0"][i] - r1["matching_scores0"][0]).abs().max())
        print("split", split, "pair", i, "idx diff", dm, "score maxdiff", dsn)
# layer-level: self block on B=5 vs B=1 slices
