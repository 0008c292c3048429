"""Matcher registry (reference: localization/match_features_batch.py:17-61).

The ``confs`` table and the plugin construction are the hot-path part.  The pair-matching loop around it
(match_from_paths, FeaturePairsDataset, writer_fn: match_features_batch.py:89-129,189-232) is
``pram_amd.localization.formats.match_pairs`` — padded batches of pairs instead of one pair per call, the same
int16 / fp16 record per pair — with the file formats in the same module."""
import pram_amd.localization.matchers as matchers
from pram_amd.localization.base_model import dynamic_load

confs = {
    'gml': {'output': 'gml',
            'model': {'name': 'gml', 'weight_path': 'weights/imp_gml.920.pth', 'sinkhorn_iterations': 20}},
    'adagml': {'output': 'adagml',
               'model': {'name': 'adagml', 'weight_path': 'weights/imp_adagml.80.pth', 'sinkhorn_iterations': 20}},
}


def build_matcher(name: str, weight_path=None, device='cuda'):
    """dynamic_load(matchers, name)(conf).eval().to(device) — localization/localizer.py:39-40,
    localization/multimap3d.py:43-44."""
    conf = dict(confs[name]['model'])
    if weight_path is not None:
        conf['weight_path'] = weight_path
    Model = dynamic_load(matchers, conf['name'])
    return Model(conf).eval().to(device)


def match_from_stores(conf_name: str, pairs, store_q, store_r, match_store, weight_path=None, device='cuda', batch_size: int = 16):
    """match_from_paths (match_features_batch.py:189-232) on already opened stores (h5py.File or formats.DictStore)."""
    from pram_amd.localization import formats
    model = build_matcher(conf_name, weight_path, device)
    pairs = formats.find_unique_new_pairs(list(pairs), match_store)
    return formats.match_pairs(model, pairs, store_q, store_r, match_store, batch_size=batch_size, device=device)
