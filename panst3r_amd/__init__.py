"""panst3r_amd -- MI355X-native PanSt3R inference forward path (hand-written HIP/CDNA4 kernels behind a C-ABI).

Public API mirrors the reference (naver/panst3r `src/panst3r/__init__.py:1`, `model/__init__.py:1-4`):
PanSt3R, Dust3rEncoder, MUSt3R, DinoV2Encoder, PanopticDecoder, PixelShuffleUpscaler, LoftUpUpscaler, InputMixer.
Heavy imports are lazy so that `panst3r_amd.synthetic` / `panst3r_amd.flops` work without the HIP library.
"""
__version__ = '0.1.0'

_LAZY = {
    'PanSt3R': 'panst3r_amd.panst3r',
    'Dust3rEncoder': 'panst3r_amd.model', 'MUSt3R': 'panst3r_amd.model', 'DinoV2Encoder': 'panst3r_amd.model',
    'PanopticDecoder': 'panst3r_amd.model', 'PixelShuffleUpscaler': 'panst3r_amd.model',
    'LoftUpUpscaler': 'panst3r_amd.model', 'InputMixer': 'panst3r_amd.model',
}


def __getattr__(name):
    if name in _LAZY:
        import importlib
        return getattr(importlib.import_module(_LAZY[name]), name)
    raise AttributeError(name)
