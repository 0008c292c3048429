"""Recogniser factory with the reference's call signature (nets/load_segnet.py:12-31):
``load_segnet(network, n_class, desc_dim, n_layers, output_dim)``; the caller then does the strict
``load_state_dict(torch.load(path)['model'])`` itself (inference.py:33-39)."""
from .segnetvit import SegNetViT

_SUPPORTED = {'segnetvit': SegNetViT}


def load_segnet(network, n_class, desc_dim, n_layers, output_dim):
    if network == 'segnet':
        raise NotImplementedError("the BN/Conv1d 'segnet' recogniser is outside the hot path "
                                  "(every shipped config selects 'segnetvit'); see DESIGN.md")
    if network not in _SUPPORTED:
        raise ValueError('ERROR! {:s} model does not exist'.format(str(network)))
    return _SUPPORTED[network]({'descriptor_dim': desc_dim, 'n_layers': n_layers, 'n_class': n_class,
                                'output_dim': output_dim, 'with_score': False})
