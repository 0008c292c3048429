"""Matcher plugin 'adagml' — contract of the reference's localization/matchers/adagml.py:13-41,
network = pram_amd.nets.adagml.AdaGML (device-resident pruning / early exit)."""
from pram_amd.localization.base_model import BaseModel
from pram_amd.localization.matchers import _plugin
from pram_amd.nets.adagml import AdaGML as _Net


class AdaGML(BaseModel):
    def _init(self, conf):
        _plugin.init_from_checkpoint(self, _Net, conf)

    def _forward(self, data):
        return _plugin.run(self, data)
