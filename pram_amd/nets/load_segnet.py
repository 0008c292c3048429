"""Factory with the reference's signature (nets/load_segnet.py:12-31)."""
from .segnetvit import SegNetViT


def load_segnet(network, n_class, desc_dim, n_layers, output_dim):
    cfg = {'descriptor_dim': desc_dim, 'n_layers': n_layers, 'n_class': n_class, 'output_dim': output_dim,
           'with_score': False}
    if network == 'segnetvit':
        return SegNetViT(cfg)
    if network == 'segnet':
        raise NotImplementedError("the BN/Conv1d 'segnet' recogniser is outside the hot path "
                                  "(every shipped config selects 'segnetvit'); see DESIGN.md")
    raise ValueError('ERROR! {:s} model does not exist'.format(str(network)))
