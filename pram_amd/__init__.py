"""pram_amd — MI355X-native (gfx950) implementation of PRAM's per-query inference hot path:
SFD2 extract -> SegNetViT recognise -> GML / AdaGML match + Sinkhorn.

Host code mirrors the reference's model-load / forward() surface (`nets.sfd2.load_sfd2`,
`nets.load_segnet.load_segnet`, `localization.base_model.dynamic_load`); the compute is
hand-written HIP behind the C ABI in include/pram_hip.h.  There is no CPU fallback.
"""
__version__ = "0.1.0"
