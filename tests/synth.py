"""Test helper: seeded synthetic inputs (SplitMix64, as the bench uses) and Merkle trees."""
from __future__ import annotations

import numpy as np

import oracle

MASK = (1 << 64) - 1
SEED = 0x5354524B  # "STRK"


def splitmix64(seed: int, count: int, start: int = 0) -> np.ndarray:
    """Draws start..start+count-1 of the SplitMix64 stream with the given seed (uint64 array)."""
    with np.errstate(over="ignore"):
        idx = np.arange(start + 1, start + count + 1, dtype=np.uint64)
        z = np.uint64(seed & MASK) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def field_elements(seed: int, n: int, start: int = 0) -> np.ndarray:
    """[n, 8] uint32 canonical field elements: element i = draws 4i..4i+3 as LE 64-bit words, mod p."""
    z = splitmix64(seed, 4 * n, 4 * start)
    limbs = z.view("<u4").reshape(n, 8).copy()
    ints = oracle.limbs_to_ints(limbs)
    fixed = [(i, x - oracle.P) for i, x in enumerate(ints) if x >= oracle.P]
    for i, x in fixed:
        limbs[i] = oracle.int_to_limbs(x)
    return limbs


def merkle_case(seed: int, num_witnesses: int, n: int, value_size: int = 32):
    """A real tree with 2^(w+1) leaves and n random queries against it.

    Returns dict(values, siblings, witnesses[w,n,32], indices, root).  For value_size > 32 the
    tree's bottom level hashes value||sibling pairs of that size (as the reference's first step
    does), so leaves are value_size bytes and level-1 nodes are their pair hashes.
    """
    rng = np.random.default_rng(seed)
    w = num_witnesses
    L = 1 << (w + 1)
    leaves = rng.integers(0, 256, size=(L, value_size), dtype=np.uint8)
    if value_size == 32:
        nodes = oracle.merkle_build(leaves)
        lvl1 = None
    else:
        import hashlib
        pair = np.stack([np.frombuffer(hashlib.blake2s(leaves[2 * i].tobytes() + leaves[2 * i + 1].tobytes()).digest(),
                                       dtype=np.uint8) for i in range(L // 2)])
        nodes_half = oracle.merkle_build(pair)  # tree over level-1 nodes, [L, 32], root at [1]
        nodes = None
        lvl1 = nodes_half
    idx = rng.integers(0, L, size=n, dtype=np.uint32)
    ld4 = L // 4
    pos = (idx // ld4 + 4 * (idx % ld4)).astype(np.int64)
    values, sibs = leaves[pos], leaves[pos ^ 1]
    if value_size == 32:
        wit = np.stack([nodes[((L + pos) >> (k + 1)) ^ 1] for k in range(w)])
        root = nodes[1].copy()
    else:
        half = L // 2
        wit = np.stack([lvl1[((half + (pos >> 1)) >> k) ^ 1] for k in range(w)])
        root = lvl1[1].copy()
    return dict(values=values, siblings=sibs, witnesses=np.ascontiguousarray(wit), indices=idx, root=root)
