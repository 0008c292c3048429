"""Matcher registry (reference: localization/match_features_batch.py:17-61).

Only the ``confs`` table and the plugin construction are part of the hot path; the h5 pair-matching
CLI around it (DataLoader workers, writer threads, h5py) is out of scope."""
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
