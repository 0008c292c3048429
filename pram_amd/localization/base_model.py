"""BaseModel contract + dynamic_load plugin loader (reference: localization/base_model.py:7-44)."""
from abc import ABCMeta, abstractmethod
from copy import copy
import inspect

from torch import nn


class BaseModel(nn.Module, metaclass=ABCMeta):
    default_conf = {}
    required_data_keys = []

    def __init__(self, conf):
        super().__init__()
        self.conf = conf = {**self.default_conf, **conf}
        self.required_data_keys = copy(self.required_data_keys)
        self._init(conf)

    def forward(self, data):
        for key in self.required_data_keys:
            assert key in data, 'Missing key {} in data'.format(key)
        return self._forward(data)

    @abstractmethod
    def _init(self, conf):
        raise NotImplementedError

    @abstractmethod
    def _forward(self, data):
        raise NotImplementedError


def dynamic_load(root, model):
    """Exactly one BaseModel subclass per module ``root.<model>`` (base_model.py:35-44)."""
    module_path = f'{root.__name__}.{model}'
    module = __import__(module_path, fromlist=[''])
    classes = inspect.getmembers(module, inspect.isclass)
    classes = [c for c in classes if c[1].__module__ == module_path]
    classes = [c for c in classes if issubclass(c[1], BaseModel)]
    assert len(classes) == 1, classes
    return classes[0][1]
