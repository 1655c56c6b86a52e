"""Model classes exported under the reference's names (src/panst3r/model/__init__.py:1-4 + `from must3r.model import *`)."""
from .encoder import Dust3rEncoder
from .must3r import MUSt3R, MemoryBank
from .dino import DinoV2Encoder
from .panoptic import (InputMixer, PixelShuffleUpscaler, LoftUpUpscaler, MaskTransformer, TextEncoder, PanopticDecoder)

__all__ = ['Dust3rEncoder', 'MUSt3R', 'MemoryBank', 'DinoV2Encoder', 'InputMixer', 'PixelShuffleUpscaler', 'LoftUpUpscaler',
           'MaskTransformer', 'TextEncoder', 'PanopticDecoder']
