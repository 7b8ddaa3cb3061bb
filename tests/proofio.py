"""Test helper: parser for the reference's serialized STARK proof.

Follows the wire format read by the reference's src/deserializer.rs:16-140 (all integers are
little-endian u32, src/utils.rs:96-101).  Used by the tests to turn tests/golden/proof.bin
(the reference's own bundled fixture) into branches, roots and query indices.
"""
from __future__ import annotations

import struct
from dataclasses import dataclass, field
from typing import List


@dataclass
class Branch:
    value: bytes
    sibling_value: bytes
    witnesses: List[bytes]


@dataclass
class MultiProof:
    branches: List[Branch] = field(default_factory=list)


@dataclass
class LDPMerkleProof:
    root2: bytes
    column_branches: MultiProof
    poly_branches: MultiProof


@dataclass
class StarkProof:
    merkle_root: bytes
    l_merkle_root: bytes
    fri_layers: List[LDPMerkleProof]
    points: bytes
    merkle_branches: MultiProof
    linear_comb_branches: MultiProof


def _u32(b: bytes, off: int) -> int:
    return struct.unpack_from("<I", b, off)[0]


def parse_multiproof(b: bytes, off: int):
    n = _u32(b, off)
    off += 4
    mp = MultiProof()
    for _ in range(n):
        vs = _u32(b, off)
        off += 4
        value = b[off:off + vs]
        off += vs
        sib = b[off:off + vs]
        off += vs
        wb = _u32(b, off)
        off += 4
        assert wb % 32 == 0
        wit = [b[off + 32 * i:off + 32 * i + 32] for i in range(wb // 32)]
        off += wb
        mp.branches.append(Branch(value, sib, wit))
    return mp, off


def parse_proof(b: bytes):
    off = 0
    merkle_root = b[0:32]
    l_merkle_root = b[32:64]
    off = 64
    layers = []
    points = b""
    while True:
        t = _u32(b, off)
        off += 4
        if t == 1:
            root2 = b[off:off + 32]
            off += 32
            col, off = parse_multiproof(b, off)
            poly, off = parse_multiproof(b, off)
            layers.append(LDPMerkleProof(root2, col, poly))
        elif t == 2:
            nb = _u32(b, off)
            off += 4
            assert nb > 0 and nb % 32 == 0
            points = b[off:off + nb]
            off += nb
            break
        else:
            raise ValueError("invalid proof element type")
    mb, off = parse_multiproof(b, off)
    lb, off = parse_multiproof(b, off)
    return StarkProof(merkle_root, l_merkle_root, layers, points, mb, lb), off


def value_offsets(b: bytes):
    """Byte offset of every branch's `value` field: {("col" | "poly", layer, i) | ("merkle", i) | ("lincomb", i): offset}."""
    out = {}

    def walk(off, key):
        n = _u32(b, off)
        off += 4
        for i in range(n):
            vs = _u32(b, off)
            out[key + (i,)] = off + 4
            off += 4 + 2 * vs
            off += 4 + _u32(b, off)
        return off

    off, layer = 64, 0
    while True:
        t = _u32(b, off)
        off += 4
        if t == 1:
            off = walk(off + 32, ("col", layer))
            off = walk(off, ("poly", layer))
            layer += 1
        elif t == 2:
            off += 4 + _u32(b, off)
            break
        else:
            raise ValueError("invalid proof element type")
    off = walk(off, ("merkle",))
    walk(off, ("lincomb",))
    return out
