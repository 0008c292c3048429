"""nets/gm.py of the reference ("GM", the SuperGlue-style matcher north_star names next to adagml).

The reference's ``GM`` class cannot be constructed — its ``__init__`` calls ``KeypointEncoder`` and
``AttentionalPropagation`` with arguments those classes do not take (nets/gm.py:127-131,51 vs nets/layers.py:98,81:
``TypeError``), and no config selects it (every config names ``gml``; ``adagml`` is the alternative,
configs/config_train_7scenes_sfd2.yaml:88-89).  So there is no behaviour to reproduce for the class, and its parity is
UNPINNED by construction.  What does exist in that file are free functions that are line-for-line the ones of
nets/gml.py (nets/gm.py:17-43,249-264 == nets/gml.py:20-46,304-319): they are re-exported here from the pinned GML
implementation, and ``GM`` fails the way the reference's does, with the reason spelled out.
"""
from __future__ import annotations

from .gml import GML, dual_softmax, sink_algorithm  # noqa: F401


def compute_matches(scores, p: float = 0.2):
    """nets/gm.py:249-264 (== GML.compute_matches, nets/gml.py:304-319) on a [B, M+1, N+1] score matrix."""
    return GML.compute_matches(None, scores, p)


class GM:
    def __init__(self, config=None):
        raise TypeError("nets.gm.GM cannot be constructed in the reference either (KeypointEncoder / AttentionalPropagation are "
                        "called with arguments they do not take, nets/gm.py:127-131,51); use GML (configs' default) or AdaGML")
