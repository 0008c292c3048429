"""Matcher plugin 'gml' — same construction / checkpoint / forward contract as the reference's
localization/matchers/gml.py:13-45, network = pram_amd.nets.gml.GML on the HIP kernels."""
from pram_amd.localization.base_model import BaseModel
from pram_amd.localization.matchers import _plugin
from pram_amd.nets.gml import GML as _Net


class GML(BaseModel):
    def _init(self, conf):
        _plugin.init_from_checkpoint(self, _Net, conf)

    def _forward(self, data):
        return _plugin.run(self, data)
