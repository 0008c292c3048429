"""Shared body of the attention-matcher plugins: build the HIP-backed network from ``conf``, load a
``{'model': state_dict}`` checkpoint strictly (reference: localization/matchers/gml.py:37-40), run under no_grad.

The reference wrappers also carry a ``default_config`` / ``required_inputs`` pair that BaseModel never reads
(it looks at ``default_conf`` / ``required_data_keys``), so the network's own defaults apply and a missing input
surfaces as a KeyError from inside the net; both facts are preserved here by simply not declaring them."""
import torch


def init_from_checkpoint(plugin, net_cls, conf):
    plugin.net = net_cls(config=conf).eval()
    ckpt = torch.load(conf['weight_path'], map_location='cpu')
    plugin.net.load_state_dict(ckpt['model'], strict=True)


def run(plugin, data):
    with torch.no_grad():
        return plugin.net(data)
