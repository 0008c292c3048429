"""Matcher plugin contract and loader.

Behavioural mirror of the reference's localization/base_model.py:7-44 (what its call sites rely on):
  * ``BaseModel(conf)`` merges ``default_conf`` under ``conf``, stores it as ``self.conf`` and calls ``_init``;
  * ``forward(data)`` asserts ``required_data_keys`` and returns ``_forward(data)``;
  * ``dynamic_load(root, name)`` imports ``root.<name>`` and returns the single BaseModel subclass defined there.
"""
from __future__ import annotations

import importlib
import inspect
from abc import ABCMeta, abstractmethod
from typing import Any, Dict, List

from torch import nn


class BaseModel(nn.Module, metaclass=ABCMeta):
    default_conf: Dict[str, Any] = {}
    required_data_keys: List[str] = []

    def __init__(self, conf: Dict[str, Any]):
        super().__init__()
        merged = dict(self.default_conf)
        merged.update(conf)
        self.conf = merged
        self.required_data_keys = list(self.required_data_keys)
        self._init(merged)

    def forward(self, data):
        missing = [k for k in self.required_data_keys if k not in data]
        assert not missing, 'Missing key {} in data'.format(missing[0] if missing else '')
        return self._forward(data)

    @abstractmethod
    def _init(self, conf):
        ...

    @abstractmethod
    def _forward(self, data):
        ...


def dynamic_load(root, model: str):
    """Resolve plugin ``model`` inside package ``root`` to its one BaseModel subclass."""
    dotted = root.__name__ + '.' + model
    module = importlib.import_module(dotted)
    found = [cls for _, cls in inspect.getmembers(module, inspect.isclass)
             if cls.__module__ == dotted and issubclass(cls, BaseModel)]
    assert len(found) == 1, found
    return found[0]
