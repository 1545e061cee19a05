"""bsarec_amd: the BSARec training hot path, hand-written for MI355X (gfx950).

Host API mirrors the reference (Sun-Sir/BSARec): ``BSARecModel(args)`` with ``forward`` /
``predict`` / ``calculate_loss`` and the 4 + 19 N state_dict keys, ``Trainer`` with ``iteration`` /
``train`` / ``valid`` / ``test``.  All arithmetic runs in ``libbsarec_hip.so`` (C ABI in
``include/bsarec_hip.h``); there is no CPU or PyTorch fallback.
"""
from .model import BSARecModel, SASRecModel, FMLPRecModel, DuoRecModel, MODEL_DICT, param_shapes  # noqa: F401

__all__ = ["BSARecModel", "SASRecModel", "FMLPRecModel", "DuoRecModel", "MODEL_DICT", "param_shapes"]
