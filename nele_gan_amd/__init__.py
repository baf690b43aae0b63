"""nele_gan_amd: MI355X-native NELE-GAN hot path (HIP kernels behind libnele_hip.so + the host-side
mirror of the reference's Python call surface).  Importing the package loads the HIP library and
fails loudly if it is missing: there is no CPU fallback."""
from . import _lib  # noqa: F401  (raises ImportError if libnele_hip.so has not been built)

__version__ = '0.1.0'


def load_all_bindings():
    """Import every module that declares ctypes signatures (so _lib._SIGS covers include/nele_hip.h)."""
    import importlib
    for m in ('ops', 'audio_util', 'metrics', 'eval_metrics'):
        try:
            importlib.import_module('.' + m, __name__)
        except ModuleNotFoundError:
            pass
