# -*- coding: utf-8 -*-
"""emsanet_amd: MI355X-native EMSANet forward/backward engine (hand-written HIP, gfx950)."""
from .args import default_args, full_args          # noqa: F401
from .data import DatasetConfig, nyuv2_config      # noqa: F401


def __getattr__(name):
    # lazy: importing the model requires torch; the C-ABI library is loaded on first use
    if name == 'EMSANet':
        from .model import EMSANet
        return EMSANet
    if name == 'get_decoders':
        from .decoder import get_decoders
        return get_decoders
    raise AttributeError(name)
