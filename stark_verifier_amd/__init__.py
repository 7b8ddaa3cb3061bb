"""stark_verifier_amd — MI355X (gfx950) hot path of a MiMC STARK verifier.

The product is ``libstarkhip.so`` (hand-written HIP kernels behind the C ABI of
``include/starkhip.h``): batched NTT / inverse NTT over p = 2^256 - 351*2^32 + 1 and batched
BLAKE2s Merkle-branch verification.  ``fft`` and ``merkle_tree`` mirror the reference crate's
modules of the same names on top of that ABI.
"""
from . import _lib  # noqa: F401
from ._lib import Context, InvalidArgument, StarkError, default_context  # noqa: F401
from . import fft, merkle_tree  # noqa: F401

__all__ = ["Context", "InvalidArgument", "StarkError", "default_context", "fft", "merkle_tree"]
