"""CPU oracle — TEST INFRASTRUCTURE ONLY.

ctypes front end for ``oracle/stark_oracle.c`` (a C restatement of the reference's
``src/fft.rs`` and ``src/merkle_tree.rs``).  Only ``tests/``, ``__graft_entry__.smoke()`` and
the ``cpu_baseline`` leg of ``bench.py`` may import this package; the product package
``stark_verifier_amd`` never does.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libstark_oracle.so")

P = 2**256 - 351 * 2**32 + 1


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "stark_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "libstark_oracle.so"])
    return _SO


_lib = None


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = ctypes.CDLL(_SO)
        vp, sz, u32, u64 = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint32, ctypes.c_uint64
        L.so_fe_add.argtypes = [vp, vp, vp]
        L.so_fe_sub.argtypes = [vp, vp, vp]
        L.so_fe_mul.argtypes = [vp, vp, vp]
        L.so_fe_pow.argtypes = [vp, vp, vp]
        L.so_fe_batch.argtypes = [ctypes.c_int, vp, vp, vp, sz]
        L.so_fft_inv.argtypes = [vp, sz, vp, vp]
        L.so_fft_inv.restype = ctypes.c_int
        L.so_fft_fwd.argtypes = [vp, sz, vp, vp]
        L.so_fft_fwd.restype = ctypes.c_int
        L.so_dft_at.argtypes = [vp, sz, vp, u64, vp, vp]
        L.so_blake2s.argtypes = [vp, sz, vp]
        L.so_permute_4_index.argtypes = [u32, u32]
        L.so_permute_4_index.restype = u32
        L.so_branch_verify.argtypes = [vp, vp, u32, vp, u32, vp, u32, vp]
        L.so_branch_verify.restype = ctypes.c_int
        L.so_merkle_verify_batch.argtypes = [vp, vp, u32, vp, u32, vp, vp, vp, sz, vp]
        L.so_merkle_verify_batch.restype = ctypes.c_int
        L.so_get_pseudorandom_indices.argtypes = [vp, sz, u32, u32, vp]
        L.so_get_pseudorandom_indices.restype = ctypes.c_int
        L.so_merkle_build.argtypes = [vp, sz, vp]
        _lib = L
    return _lib


# ---- conversions -------------------------------------------------------------------------

def int_to_limbs(x: int) -> np.ndarray:
    return np.frombuffer(int(x).to_bytes(32, "little"), dtype="<u4").copy()


def limbs_to_int(a) -> int:
    return int.from_bytes(np.ascontiguousarray(a, dtype="<u4").tobytes(), "little")


def ints_to_limbs(xs) -> np.ndarray:
    buf = b"".join(int(x).to_bytes(32, "little") for x in xs)
    return np.frombuffer(buf, dtype="<u4").reshape(-1, 8).copy()


def limbs_to_ints(a) -> list:
    b = np.ascontiguousarray(a, dtype="<u4").tobytes()
    return [int.from_bytes(b[i:i + 32], "little") for i in range(0, len(b), 32)]


def _p(a: np.ndarray):
    return a.ctypes.data_as(ctypes.c_void_p)


# ---- field -------------------------------------------------------------------------------

def fe_batch(op: str, a: np.ndarray, b: np.ndarray) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.uint32).reshape(-1, 8)
    b = np.ascontiguousarray(b, dtype=np.uint32).reshape(-1, 8)
    r = np.empty_like(a)
    lib().so_fe_batch({"add": 0, "sub": 1, "mul": 2}[op], _p(a), _p(b), _p(r), a.shape[0])
    return r


def fe_pow(a: int, e: int) -> int:
    r = np.empty(8, dtype=np.uint32)
    x, y = int_to_limbs(a), int_to_limbs(e)
    lib().so_fe_pow(_p(x), _p(y), _p(r))
    return limbs_to_int(r)


# ---- NTT ---------------------------------------------------------------------------------

def fft_inv(v: np.ndarray, omega: int) -> np.ndarray:
    """src/fft.rs:64 on [n,8] uint32 limbs.  Raises ValueError where the reference panics."""
    v = np.ascontiguousarray(v, dtype=np.uint32).reshape(-1, 8)
    out = np.empty_like(v)
    w = int_to_limbs(omega)
    if lib().so_fft_inv(_p(v), v.shape[0], _p(w), _p(out)) != 0:
        raise ValueError("fft_inv: length must be a power of two equal to the order of omega")
    return out


def fft_fwd(v: np.ndarray, omega: int) -> np.ndarray:
    v = np.ascontiguousarray(v, dtype=np.uint32).reshape(-1, 8)
    out = np.empty_like(v)
    w = int_to_limbs(omega)
    if lib().so_fft_fwd(_p(v), v.shape[0], _p(w), _p(out)) != 0:
        raise ValueError("fft_fwd: length must be a power of two equal to the order of omega")
    return out


def dft_at(v: np.ndarray, omega: int, k: int, scale: int = 1) -> int:
    v = np.ascontiguousarray(v, dtype=np.uint32).reshape(-1, 8)
    out = np.empty(8, dtype=np.uint32)
    w, s = int_to_limbs(omega), int_to_limbs(scale)
    lib().so_dft_at(_p(v), v.shape[0], _p(w), k, _p(s), _p(out))
    return limbs_to_int(out)


# ---- BLAKE2s / Merkle --------------------------------------------------------------------

def blake2s(msg: bytes) -> bytes:
    m = np.frombuffer(bytes(msg), dtype=np.uint8) if len(msg) else np.zeros(1, dtype=np.uint8)
    out = np.empty(32, dtype=np.uint8)
    lib().so_blake2s(_p(m), len(msg), _p(out))
    return out.tobytes()


def permute_4_index(x: int, L: int) -> int:
    return lib().so_permute_4_index(x, L)


def branch_verify(value: bytes, sibling: bytes, witnesses, root: bytes, a: int):
    """src/merkle_tree.rs:119.  Returns (accept, recomputed_root)."""
    val = np.frombuffer(bytes(value), dtype=np.uint8)
    sib = np.frombuffer(bytes(sibling), dtype=np.uint8)
    wit = np.frombuffer(b"".join(bytes(w) for w in witnesses) or b"\0", dtype=np.uint8)
    rt = np.frombuffer(bytes(root), dtype=np.uint8)
    dig = np.empty(32, dtype=np.uint8)
    rc = lib().so_branch_verify(_p(val), _p(sib), len(value), _p(wit), len(witnesses), _p(rt), a, _p(dig))
    if rc < 0:
        raise ValueError("branch_verify: the reference panics on this shape")
    return bool(rc), dig.tobytes()


def merkle_verify_batch(values, siblings, witnesses, indices, roots, root_ids=None) -> np.ndarray:
    """values/siblings [n,vs] u8, witnesses [w,n,32] u8, indices [n] u32, roots [r,32] u8."""
    values = np.ascontiguousarray(values, dtype=np.uint8)
    siblings = np.ascontiguousarray(siblings, dtype=np.uint8)
    witnesses = np.ascontiguousarray(witnesses, dtype=np.uint8)
    indices = np.ascontiguousarray(indices, dtype=np.uint32)
    roots = np.ascontiguousarray(roots, dtype=np.uint8).reshape(-1, 32)
    n, vs = values.shape
    w = witnesses.shape[0]
    acc = np.empty(n, dtype=np.uint8)
    rid = None if root_ids is None else np.ascontiguousarray(root_ids, dtype=np.uint32)
    rc = lib().so_merkle_verify_batch(_p(values), _p(siblings), vs, _p(witnesses), w, _p(indices),
                                      _p(roots), None if rid is None else _p(rid), n, _p(acc))
    if rc != 0:
        raise ValueError("merkle_verify_batch: the reference panics on this shape")
    return acc


def get_pseudorandom_indices(seed: bytes, count: int, modulus: int, exclude: int | None = None):
    s = np.frombuffer(bytes(seed), dtype=np.uint8)
    out = np.empty(count, dtype=np.uint32)
    if lib().so_get_pseudorandom_indices(_p(s), count, modulus, exclude or 0, _p(out)) != 0:
        raise ValueError("bad modulus")
    return out


def merkle_build(leaves: np.ndarray) -> np.ndarray:
    """leaves [L,32] u8 (already in tree order) -> nodes [2L,32] u8, nodes[1] = root."""
    leaves = np.ascontiguousarray(leaves, dtype=np.uint8).reshape(-1, 32)
    L = leaves.shape[0]
    nodes = np.zeros((2 * L, 32), dtype=np.uint8)
    lib().so_merkle_build(_p(leaves), L, _p(nodes))
    return nodes
