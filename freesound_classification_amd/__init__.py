"""MI355X-native (gfx950) implementation of the freesound-classification audio-tagging hot
path: waveform -> STFT -> mel -> log -> frequency channel -> 2-d / 1-d resnet-block CNN with
deep-supervision global-max-pool heads -> LSEP -> gradients -> (RCCL all-reduce) -> optimizer.

The package mirrors the reference's module layout (`networks`, `ops`, `datasets`) so
`train_2d_cnn.py`-style callers switch by changing the import root.  All device arithmetic is
in `libfsc_hip.so` (C ABI: include/fsc_hip.h); there is no CPU fallback.
"""
from . import _lib

__version__ = "0.1.0"


def library_path():
    return _lib.LIB_PATH


def load_library():
    """Load libfsc_hip.so (raises if it has not been built)."""
    return _lib.load()
