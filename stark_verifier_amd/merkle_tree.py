"""Host-side mirror of the reference's ``merkle_tree`` module (reference src/merkle_tree.rs).

``ProofBranch`` and ``MultiProof`` keep the reference's fields and ``verify`` signatures
(src/merkle_tree.rs:11-21, :25, :119).  ``MultiProof.verify`` groups its branches by shape
(value size, witness count), flattens each group into the level-major device layout and makes
one ``stark_merkle_verify_batch`` call per group.

Error behaviour: the reference asserts on a root mismatch (src/merkle_tree.rs:165) and on a
length mismatch (:28); here a mismatch raises ``MerkleMismatch`` unless ``strict=False``, in
which case ``verify`` returns ``None`` — the reference's (unreachable) ``None`` branch
(:166-170, :38-40).
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import _lib

Value = bytes
MerkleDigest = bytes  # 32 bytes


class MerkleMismatch(AssertionError):
    pass


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(ctypes.c_void_p)


def verify_batch(values: np.ndarray, siblings: np.ndarray, witnesses: np.ndarray, indices: np.ndarray,
                 roots: np.ndarray, root_ids: Optional[np.ndarray] = None, want_digests: bool = False,
                 ctx: _lib.Context | None = None):
    """One uniform bucket: values/siblings [n, vs] u8, witnesses [w, n, 32] u8, indices [n] u32,
    roots [r, 32] u8, root_ids [n] u32 or None.  Returns accept [n] u8 (and digests [n, 32])."""
    ctx = ctx or _lib.default_context()
    values = np.ascontiguousarray(values, dtype=np.uint8)
    siblings = np.ascontiguousarray(siblings, dtype=np.uint8)
    witnesses = np.ascontiguousarray(witnesses, dtype=np.uint8)
    indices = np.ascontiguousarray(indices, dtype=np.uint32)
    roots = np.ascontiguousarray(roots, dtype=np.uint8).reshape(-1, 32)
    n = values.shape[0]
    vs = values.shape[1] if values.ndim == 2 else 0
    w = witnesses.shape[0]
    if siblings.shape != values.shape or indices.shape != (n,) or (w and witnesses.shape != (w, n, 32)):
        raise ValueError("inconsistent bucket shapes")
    rid = None if root_ids is None else np.ascontiguousarray(root_ids, dtype=np.uint32)
    accept = np.zeros(n, dtype=np.uint8)
    digests = np.zeros((n, 32), dtype=np.uint8) if want_digests else None
    ctx.check(ctx.lib.stark_merkle_verify_batch(
        ctx.handle, _ptr(values), _ptr(siblings), vs, _ptr(witnesses), w, _ptr(indices), _ptr(roots), roots.shape[0],
        None if rid is None else _ptr(rid), n, _ptr(accept), None if digests is None else _ptr(digests)))
    return (accept, digests) if want_digests else accept


@dataclass
class ProofBranch:
    witnesses: List[MerkleDigest] = field(default_factory=list)
    sibling_value: Value = b""
    value: Value = b""

    def verify(self, root: MerkleDigest, a: int, strict: bool = True, ctx: _lib.Context | None = None) -> Optional[Value]:
        res = MultiProof([self], bytes(root)).verify([a], None, strict=strict, ctx=ctx)
        return None if res is None else res[0]


@dataclass
class MultiProof:
    branches: List[ProofBranch] = field(default_factory=list)
    root: MerkleDigest = bytes(32)

    def verify(self, indices: Sequence[int], rt: Optional[MerkleDigest] = None, strict: bool = True,
               ctx: _lib.Context | None = None) -> Optional[List[Value]]:
        if len(self.branches) != len(indices):
            raise AssertionError(f"branches len {len(self.branches)} != indices len {len(indices)}")
        root = bytes(rt) if rt is not None else bytes(self.root)
        accept = self.accept_bits(indices, root, ctx=ctx)
        if not accept.all():
            if strict:
                bad = int(np.flatnonzero(accept == 0)[0])
                raise MerkleMismatch(f"values didn't match up (branch {bad})")
            return None
        return [bytes(b.value) for b in self.branches]

    def accept_bits(self, indices: Sequence[int], root: MerkleDigest, ctx: _lib.Context | None = None) -> np.ndarray:
        """Per-branch accept flags, without the reference's panic."""
        buckets: Dict[Tuple[int, int], List[int]] = {}
        for i, b in enumerate(self.branches):
            if len(b.sibling_value) != len(b.value):
                raise _lib.InvalidArgument(_lib.STARK_ERR_INVALID, "value and sibling_value differ in length")
            buckets.setdefault((len(b.value), len(b.witnesses)), []).append(i)
        accept = np.zeros(len(self.branches), dtype=np.uint8)
        rt = np.frombuffer(bytes(root), dtype=np.uint8).reshape(1, 32)
        for (vs, w), ids in buckets.items():
            brs = [self.branches[i] for i in ids]
            values = np.frombuffer(b"".join(b.value for b in brs), dtype=np.uint8).reshape(len(ids), vs)
            sibs = np.frombuffer(b"".join(b.sibling_value for b in brs), dtype=np.uint8).reshape(len(ids), vs)
            wit = np.frombuffer(b"".join(b.witnesses[l] for l in range(w) for b in brs), dtype=np.uint8).reshape(w, len(ids), 32)
            idx = np.array([int(indices[i]) & 0xFFFFFFFF for i in ids], dtype=np.uint32)
            accept[ids] = verify_batch(values, sibs, wit, idx, rt, ctx=ctx)
        return accept


def get_pseudorandom_indices(seeds, count: int, modulus: int, exclude_multiples_of: int | None = None,
                             ctx: _lib.Context | None = None) -> np.ndarray:
    """`utils::get_pseudorandom_indices` (reference src/utils.rs:53) for one 32-byte seed (returns [count]) or a
    list / [n, 32] array of seeds (returns [n, count]); hash chain and reduction run on the GPU."""
    ctx = ctx or _lib.default_context()
    single = isinstance(seeds, (bytes, bytearray))
    s = np.frombuffer(bytes(seeds), dtype=np.uint8).reshape(1, 32) if single else \
        np.ascontiguousarray(np.array([np.frombuffer(bytes(x), dtype=np.uint8) for x in seeds]) if not isinstance(seeds, np.ndarray) else seeds, dtype=np.uint8).reshape(-1, 32)
    out = np.zeros((s.shape[0], count), dtype=np.uint32)
    ctx.check(ctx.lib.stark_pseudorandom_indices(ctx.handle, s.ctypes.data, s.shape[0], count, modulus,
                                                 exclude_multiples_of or 0, out.ctypes.data))
    return out[0] if single else out


def merkle_build(leaves: np.ndarray, ctx: _lib.Context | None = None) -> np.ndarray:
    """All nodes of the BLAKE2s tree over `leaves` ([L, 32] uint8, L a power of two, already in the tree's
    leaf order): returns [2L, 32] with the root at [1] and node i = H(node 2i || node 2i+1). Not in the
    reference (its prover is external); used to build synthetic trees on the device."""
    ctx = ctx or _lib.default_context()
    lv = np.ascontiguousarray(leaves, dtype=np.uint8).reshape(-1, 32)
    L = lv.shape[0]
    if L == 0 or L & (L - 1):
        raise _lib.InvalidArgument(_lib.STARK_ERR_INVALID, f"{L} leaves: not a power of two")
    nodes = np.zeros((2 * L, 32), dtype=np.uint8)
    ctx.check(ctx.lib.stark_merkle_build(ctx.handle, lv.ctypes.data, L.bit_length() - 1, nodes.ctypes.data))
    return nodes
