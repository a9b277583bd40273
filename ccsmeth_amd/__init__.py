"""ccsmeth_amd — MI355X-native (gfx950) implementation of the ccsmeth `call_mods` attbigru2s hot path.

Layout: csrc/ (HIP kernels + the C-ABI libccsm), models.py (host mirror of ModelAttRNN), call_modifications.py
(host mirror of the batching / per-site result functions), utils/.
"""
__version__ = "0.1.0"
