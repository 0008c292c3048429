"""On-disk / wire formats at the two edges of the hot path and the batch pair-matching
driver built on them (§8(a) a17).

Reference behaviour restated here:
  * matches file   — localization/match_features_batch.py:119-129 (writer_fn): one group per pair, named
    names_to_pair(q, r) (colmap_utils/parsers.py:79-80: '/'-joined, '/' inside a name -> '-'), datasets
    ``matches0`` int16 and ``matching_scores0`` float16 of batch element 0; an existing group is replaced.
  * features file  — localization/extract_features.py:215-238: one group per image name, ``descriptors`` stored
    TRANSPOSED as [D, N], ``keypoints`` rescaled to the original image ``(k + .5) * (orig / size) - .5``,
    ``scores``, ``image_size`` = original (w, h).
  * reading a pair — match_features_batch.py:89-113 (FeaturePairsDataset.__getitem__): every dataset -> float32
    tensor with suffix 0 / 1, descriptors transposed back to [N, D], ``image{0,1}`` = an empty tensor whose shape
    carries (1, h, w) = (1,) + image_size[::-1].
  * pair bookkeeping — parsers.py:68-80 (parse_retrieval, names_to_pair[_old]) and
    match_features_batch.py:165-186 (find_unique_new_pairs).

The container is anything with the h5py.Group subset used by the reference (``in``, ``del``, ``create_group``,
``create_dataset``, ``[]``, ``items()``): an ``h5py.File`` when h5py is installed, else :class:`DictStore`
(in memory, optionally persisted as .npz) — h5py is not part of this image, so the h5 container itself is the one
thing here that cannot be exercised; the value conversions are what the tests pin.

``match_pairs`` is the MI355X-side replacement of match_from_paths' loop (match_features_batch.py:189-232): the
reference feeds the matcher one pair at a time (DataLoader batch_size = 1) and hands results to 5 writer threads;
here pairs are padded into batches (ragged ``lens`` on the device — each pair still gets exactly its B = 1
result) and the int16 / fp16 conversion happens on the device before one D2H copy per batch.
"""
from __future__ import annotations

from collections import defaultdict
from queue import Queue
from threading import Thread
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np
import torch


# ------------------------------------------------------------------ names / pair lists
def names_to_pair(name0: str, name1: str, separator: str = "/") -> str:
    return separator.join((name0.replace("/", "-"), name1.replace("/", "-")))


def names_to_pair_old(name0: str, name1: str) -> str:
    return names_to_pair(name0, name1, separator="_")


def parse_retrieval(path) -> Dict[str, List[str]]:
    """'query ref' per line -> {query: [refs in file order]}"""
    out = defaultdict(list)
    with open(path, "r") as f:
        for line in f.read().rstrip("\n").split("\n"):
            q, r = line.split(" ")
            out[q].append(r)
    return dict(out)


def find_unique_new_pairs(pairs_all: Sequence[Tuple[str, str]], store=None) -> List[Tuple[str, str]]:
    """Drop (j, i) when (i, j) is present, and pairs already in ``store`` under either order / either naming.
    The reference builds a ``set`` (arbitrary order); here the first occurrence order is kept — same set."""
    seen, pairs = set(), []
    for i, j in pairs_all:
        if (j, i) not in seen and (i, j) not in seen:
            seen.add((i, j))
            pairs.append((i, j))
    if store is None:
        return pairs
    has = lambda k: _contains(store, k)
    return [(i, j) for i, j in pairs
            if not (has(names_to_pair(i, j)) or has(names_to_pair(j, i)) or has(names_to_pair_old(i, j)) or has(names_to_pair_old(j, i)))]


# ------------------------------------------------------------------ a minimal group store (h5py.Group subset)
class DictStore:
    """Nested groups of numpy datasets with the h5py calls the reference uses.  '/' in a key addresses sub-groups,
    as in h5py.  ``save`` / ``load`` persist to a flat .npz (keys joined with '/')."""

    def __init__(self):
        self._d: Dict[str, object] = {}

    def _walk(self, key: str, create: bool):
        parts = [p for p in key.split("/") if p]
        node = self
        for p in parts[:-1]:
            nxt = node._d.get(p)
            if nxt is None:
                if not create:
                    raise KeyError(key)
                nxt = node._d[p] = DictStore()
            node = nxt
        return node, parts[-1]

    def create_group(self, key: str) -> "DictStore":
        node, leaf = self._walk(key, True)
        if leaf in node._d:
            raise ValueError(f"group {key!r} already exists")      # h5py raises ValueError too
        g = node._d[leaf] = DictStore()
        return g

    def create_dataset(self, key: str, data) -> np.ndarray:
        node, leaf = self._walk(key, True)
        if leaf in node._d:
            raise ValueError(f"dataset {key!r} already exists")
        a = node._d[leaf] = np.array(data)
        return a

    def __contains__(self, key: str) -> bool:
        try:
            node, leaf = self._walk(key, False)
        except KeyError:
            return False
        return isinstance(node, DictStore) and leaf in node._d

    def __getitem__(self, key: str):
        node, leaf = self._walk(key, False)
        return node._d[leaf]

    def __delitem__(self, key: str):
        node, leaf = self._walk(key, False)
        del node._d[leaf]

    def items(self):
        return self._d.items()

    def keys(self):
        return self._d.keys()

    def _flat(self, prefix=""):
        for k, v in self._d.items():
            if isinstance(v, DictStore):
                yield from v._flat(prefix + k + "/")
            else:
                yield prefix + k, v

    def save(self, path):
        np.savez(path, **{k.replace("/", "\x1f"): v for k, v in self._flat()})

    @classmethod
    def load(cls, path) -> "DictStore":
        s = cls()
        with np.load(path) as z:
            for k in z.files:
                s.create_dataset(k.replace("\x1f", "/"), z[k])
        return s


def _contains(store, key: str) -> bool:
    return key in store


def open_store(path, mode: str = "a"):
    """h5py.File(path, mode) when h5py is importable (the reference's container), otherwise raises with the
    alternative spelled out — nothing silently changes format."""
    try:
        import h5py
    except ImportError as e:
        raise ImportError("h5py is not installed: use formats.DictStore() (in memory / .npz) or install h5py to read "
                          "and write the reference's .h5 files") from e
    return h5py.File(str(path), mode, libver="latest")


# ------------------------------------------------------------------ matches
def encode_matches(matches0: torch.Tensor, matching_scores0: Optional[torch.Tensor] = None) -> Dict[str, np.ndarray]:
    """writer_fn's conversions for ONE pair: indices -> int16 (``.short()``: wraps above 32767 exactly like the
    reference), scores -> float16 (round-to-nearest-even)."""
    out = {"matches0": matches0.detach().cpu().short().numpy()}
    if matching_scores0 is not None:
        out["matching_scores0"] = matching_scores0.detach().cpu().half().numpy()
    return out


def write_matches(store, pair: str, enc: Dict[str, np.ndarray]) -> None:
    if pair in store:
        del store[pair]
    grp = store.create_group(pair)
    grp.create_dataset("matches0", data=enc["matches0"])
    if "matching_scores0" in enc:
        grp.create_dataset("matching_scores0", data=enc["matching_scores0"])


def read_matches(store, name0: str, name1: str):
    """-> (matches0 int16 [M], scores float16 [M] or None); looks the pair up under both orders is the caller's job
    (the reference's readers do the same)."""
    grp = store[names_to_pair(name0, name1)]
    m = np.asarray(grp["matches0"])
    s = np.asarray(grp["matching_scores0"]) if "matching_scores0" in grp else None
    return m, s


# ------------------------------------------------------------------ features
def encode_features(pred: Dict[str, np.ndarray], image_hw: Tuple[int, int], original_size_wh) -> Dict[str, np.ndarray]:
    """extract_features.py:215-232 on the dict returned by extract_sfd2_return: descriptors [N, D] -> [D, N];
    keypoints (x, y) in the resized image -> original image ``(k + .5) * scales - .5`` with
    ``scales = (original_size / (w, h)).astype(float32)``; ``image_size`` = original (w, h)."""
    out = dict(pred)
    out["descriptors"] = np.asarray(pred["descriptors"]).transpose()
    original_size = np.asarray(original_size_wh)
    out["image_size"] = original_size
    if "keypoints" in pred:
        size = np.array(tuple(image_hw)[::-1])
        scales = (original_size / size).astype(np.float32)
        out["keypoints"] = (np.asarray(pred["keypoints"]) + .5) * scales[None] - .5
    return out


def write_features(store, name: str, enc: Dict[str, np.ndarray]) -> None:
    grp = store.create_group(name)
    for k, v in enc.items():
        grp.create_dataset(k, data=v)


def read_feature_pair(store_q, name0: str, store_r, name1: str) -> Dict[str, torch.Tensor]:
    """FeaturePairsDataset.__getitem__ (without the DataLoader's batch dimension)."""
    data = {}
    for store, name, sfx in ((store_q, name0, "0"), (store_r, name1, "1")):
        grp = store[name]
        for k, v in grp.items():
            t = torch.from_numpy(np.asarray(v).__array__()).float()
            data[k + sfx] = t.t() if k == "descriptors" else t
        data["image" + sfx] = torch.empty((1,) + tuple(int(x) for x in np.asarray(grp["image_size"]))[::-1])
    return data


# ------------------------------------------------------------------ batch pair matching
class _Writers:
    """The reference's WorkQueue (match_features_batch.py:64-85): N threads draining a bounded queue."""

    def __init__(self, fn, num_threads: int):
        self.q: Queue = Queue(num_threads)
        self.err: List[BaseException] = []
        self.threads = [Thread(target=self._run, args=(fn,), daemon=True) for _ in range(num_threads)]
        for t in self.threads:
            t.start()

    def _run(self, fn):
        item = self.q.get()
        while item is not None:
            try:
                fn(item)
            except BaseException as e:      # surfaced in join(): a lost write must not pass silently
                self.err.append(e)
            item = self.q.get()

    def put(self, item):
        self.q.put(item)

    def join(self):
        for _ in self.threads:
            self.q.put(None)
        for t in self.threads:
            t.join()
        if self.err:
            raise self.err[0]


@torch.no_grad()
def match_pairs(model, pairs: Iterable[Tuple[str, str]], store_q, store_r, match_store, batch_size: int = 16,
                device="cuda", writer_threads: int = 1) -> int:
    """Match every (query, reference) pair and store ``matches0`` (int16) / ``matching_scores0`` (fp16) under
    names_to_pair(q, r).  ``model`` is a matcher plugin (localization/matchers/{gml,adagml}.py).  Pairs are padded
    to a common length per batch and carry ``lens0`` / ``lens1``, so each pair's result is its own B = 1 result.
    Returns the number of pairs written.  (h5py objects are not thread-safe: keep writer_threads = 1 for them; the
    reference opens the file per write instead.)"""
    import threading
    pairs = list(pairs)
    lock = threading.Lock()

    def write(item):
        pair, enc = item
        with lock:
            write_matches(match_store, pair, enc)

    writers = _Writers(write, writer_threads)
    net = getattr(model, "net", model)
    if not hasattr(net, "produce_matches"):
        # descriptor-only plugins ('NNM' -> matchers/nearest_neighbor.py): no ragged-batch support, one pair per call
        # like the reference's loop.  The plugin wants channel-major descriptors [B, D, N] (nearest_neighbor.py:44-47);
        # the reference's FeaturePairsDataset hands it [N, D] (it transposes for the attention matchers), which only
        # runs there when both sets happen to have the same size — here the plugin gets the layout it is written for.
        for a, b in pairs:
            it = read_feature_pair(store_q, a, store_r, b)
            k = it["keypoints0"].shape[0]
            if k == 0 or it["keypoints1"].shape[0] == 0:
                enc = {"matches0": np.full((k,), -1, np.int16), "matching_scores0": np.zeros((k,), np.float16)}
            else:
                pred = model({"descriptors0": it["descriptors0"].t()[None].contiguous().to(device),
                              "descriptors1": it["descriptors1"].t()[None].contiguous().to(device)})
                enc = encode_matches(pred["matches0"][0], pred.get("matching_scores0", [None])[0])
            writers.put((names_to_pair(a, b), enc))
        writers.join()
        return len(pairs)
    for lo in range(0, len(pairs), batch_size):
        chunk = pairs[lo:lo + batch_size]
        items = [read_feature_pair(store_q, a, store_r, b) for a, b in chunk]
        B = len(items)
        m = max(it["keypoints0"].shape[0] for it in items)
        n = max(it["keypoints1"].shape[0] for it in items)
        if m == 0 or n == 0:
            for (a, b), it in zip(chunk, items):
                k = it["keypoints0"].shape[0]
                writers.put((names_to_pair(a, b), {"matches0": np.full((k,), -1, np.int16), "matching_scores0": np.zeros((k,), np.float16)}))
            continue

        def pad(key, L, width=None):
            shape = (B, L) if width is None else (B, L, width)
            out = torch.zeros(shape, dtype=torch.float32)
            for i, it in enumerate(items):
                out[i, :it[key].shape[0]] = it[key]
            return out.to(device, non_blocking=True)

        D = items[0]["descriptors0"].shape[1]
        data = {
            "descriptors0": pad("descriptors0", m, D), "descriptors1": pad("descriptors1", n, D),
            "keypoints0": pad("keypoints0", m, 2), "keypoints1": pad("keypoints1", n, 2),
            "scores0": pad("scores0", m), "scores1": pad("scores1", n),
            "lens0": torch.tensor([it["keypoints0"].shape[0] for it in items], dtype=torch.int32, device=device),
            "lens1": torch.tensor([it["keypoints1"].shape[0] for it in items], dtype=torch.int32, device=device),
        }
        shapes0 = {tuple(it["image0"].shape) for it in items}
        shapes1 = {tuple(it["image1"].shape) for it in items}
        if len(shapes0) == 1 and len(shapes1) == 1:
            # the reference passes image tensors whose (DataLoader-batched) shape is (1, 1, h, w)
            data["image_shape0"] = (1,) + next(iter(shapes0))
            data["image_shape1"] = (1,) + next(iter(shapes1))
            pred = net.produce_matches(data) if hasattr(net, "produce_matches") else model(data)
            m0 = pred["matches0"].short().cpu().numpy()
            s0 = pred["matching_scores0"].half().cpu().numpy()
        else:   # mixed image sizes in one batch: keypoint normalisation differs per pair -> one pair at a time
            m0 = np.full((B, m), -1, np.int16)
            s0 = np.zeros((B, m), np.float16)
            for i, it in enumerate(items):
                one = {k: (v[i:i + 1] if torch.is_tensor(v) else v) for k, v in data.items()}
                one["image_shape0"], one["image_shape1"] = (1,) + tuple(it["image0"].shape), (1,) + tuple(it["image1"].shape)
                p1 = net.produce_matches(one) if hasattr(net, "produce_matches") else model(one)
                m0[i], s0[i] = p1["matches0"][0].short().cpu().numpy(), p1["matching_scores0"][0].half().cpu().numpy()
        for i, ((a, b), it) in enumerate(zip(chunk, items)):
            k = it["keypoints0"].shape[0]
            writers.put((names_to_pair(a, b), {"matches0": m0[i, :k].copy(), "matching_scores0": s0[i, :k].copy()}))
    writers.join()
    return len(pairs)
