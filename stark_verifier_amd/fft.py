"""Host-side mirror of the reference's ``fft`` module (reference src/fft.rs) over libstarkhip.

``fft_inv(v, root_of_unity, modulus)`` keeps the reference's name, argument order and meaning
(src/fft.rs:64); values are Python integers (the reference's BigInt).  Where the reference
panics (length not a power of two, root of the wrong order) ``InvalidArgument`` (a ValueError)
is raised.  ``fft`` is the forward transform the reference only has as the private ``_fft``
(src/fft.rs:37).  The array functions work on ``[..., 8]`` uint32 little-endian limb arrays.
"""
from __future__ import annotations

import ctypes
from typing import List, Sequence

import numpy as np

from . import _lib

MODULUS = 2**256 - 351 * 2**32 + 1


def _limbs(x: int) -> np.ndarray:
    return np.frombuffer(int(x % (1 << 256)).to_bytes(32, "little"), dtype="<u4").copy()


def ints_to_limbs(xs: Sequence[int]) -> np.ndarray:
    buf = b"".join(int(x).to_bytes(32, "little") for x in xs)
    return np.frombuffer(buf, dtype="<u4").reshape(-1, 8).copy()


def limbs_to_ints(a: np.ndarray) -> List[int]:
    b = np.ascontiguousarray(a, dtype="<u4").tobytes()
    return [int.from_bytes(b[i:i + 32], "little") for i in range(0, len(b), 32)]


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(ctypes.c_void_p)


def _reduce_inputs(v: Sequence[int], modulus: int) -> List[int]:
    # the reference accepts any BigInt (negative included) and reduces as it goes
    return [int(x) % modulus for x in v]


def fft_inv(v: Sequence[int], root_of_unity: int, modulus: int = MODULUS, ctx: _lib.Context | None = None) -> List[int]:
    ctx = ctx or _lib.default_context()
    vals = ints_to_limbs(_reduce_inputs(v, modulus)) if len(v) else np.zeros((0, 8), dtype=np.uint32)
    out = np.empty_like(vals)
    r, m = _limbs(root_of_unity % modulus), _limbs(modulus)
    ctx.check(ctx.lib.stark_fft_inv(ctx.handle, _ptr(vals), len(v), _ptr(r), _ptr(m), _ptr(out)))
    return limbs_to_ints(out)


def fft(v: Sequence[int], root_of_unity: int, modulus: int = MODULUS, ctx: _lib.Context | None = None) -> List[int]:
    ctx = ctx or _lib.default_context()
    vals = ints_to_limbs(_reduce_inputs(v, modulus)) if len(v) else np.zeros((0, 8), dtype=np.uint32)
    out = np.empty_like(vals)
    r, m = _limbs(root_of_unity % modulus), _limbs(modulus)
    ctx.check(ctx.lib.stark_fft(ctx.handle, _ptr(vals), len(v), _ptr(r), _ptr(m), _ptr(out)))
    return limbs_to_ints(out)


def ntt_batch(limbs: np.ndarray, root_of_unity: int, inverse: bool = False, ctx: _lib.Context | None = None) -> np.ndarray:
    """Transform every row of a ``[batch, n, 8]`` (or ``[n, 8]``) uint32 limb array; returns a new array."""
    ctx = ctx or _lib.default_context()
    a = np.array(limbs, dtype=np.uint32, order="C", copy=True)
    if a.ndim == 2:
        a = a[None]
    if a.ndim != 3 or a.shape[2] != 8:
        raise ValueError("expected [batch, n, 8] uint32 limbs")
    batch, n = a.shape[0], a.shape[1]
    if n == 0 or n & (n - 1):
        raise _lib.InvalidArgument(_lib.STARK_ERR_INVALID, f"length {n} is not a power of two")
    r = _limbs(root_of_unity % MODULUS)
    ctx.check(ctx.lib.stark_ntt_batch(ctx.handle, _ptr(a), n.bit_length() - 1, batch, _ptr(r), int(bool(inverse))))
    return a.reshape(np.shape(limbs))


def ntt_batch_dev(d_in: int, d_out: int, log_n: int, batch: int, root_of_unity: int, inverse: bool = False,
                  ctx: _lib.Context | None = None, stream: int = 0) -> None:
    """Queue a batched transform on device memory (raw device pointers, e.g. ``tensor.data_ptr()``)."""
    ctx = ctx or _lib.default_context()
    r = _limbs(root_of_unity % MODULUS)
    ctx.check(ctx.lib.stark_ntt_batch_dev(ctx.handle, d_in, d_out, log_n, batch, _ptr(r), int(bool(inverse)), stream or None))


def field_op(op: str, a: np.ndarray, b: np.ndarray | None = None, ctx: _lib.Context | None = None) -> np.ndarray:
    ctx = ctx or _lib.default_context()
    code = {"add": 0, "sub": 1, "mul": 2, "canon": 3, "mul29": 4, "add29": 5, "sub29": 6, "lazymul29": 7}[op]
    a = np.ascontiguousarray(a, dtype=np.uint32).reshape(-1, 8)
    bb = None if b is None else np.ascontiguousarray(b, dtype=np.uint32).reshape(-1, 8)
    r = np.empty_like(a)
    ctx.check(ctx.lib.stark_field_op(ctx.handle, code, _ptr(a), None if bb is None else _ptr(bb), _ptr(r), a.shape[0]))
    return r
