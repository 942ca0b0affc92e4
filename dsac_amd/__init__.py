"""dsac_amd -- MI355X (gfx950) native DSAC hypothesis-scoring engine.

The product is dsac_amd/libdsac_hip.so (hand-written HIP kernels behind the C ABI of include/dsac_hip.h);
this package is its Python host side: `capi` (ctypes binding), `engine` (mirror of the reference's
cnn_softam.h interface), `dist` (image sharding over the GPUs of a node) and `synth` (synthetic frames).
Importing the package loads the shared library and fails loudly if it is missing.
"""
from . import capi  # noqa: F401  (raises ImportError when libdsac_hip.so is absent)
from .engine import Engine  # noqa: F401

__all__ = ["Engine", "capi"]
