"""CPU oracle of the MiMC STARK prover — TEST INFRASTRUCTURE ONLY.

The reference holds no prover: its proofs come from an external Python program (reference
README.md:9).  This file restates the protocol that program and the reference's verifier
(src/lib.rs:31-197) share, over Python integers, and writes the wire format the reference's
parser reads (src/deserializer.rs:16-140).  Every value the verifier recomputes is produced here
from the same definitions:

  * the three column evaluations P, D, B and the linear combination L checked at src/lib.rs:163-192;
  * the trees ProofBranch::verify walks (src/merkle_tree.rs:112-171): leaves in permute_4_index
    order, first level BLAKE2s(value || sibling), then 32-byte nodes;
  * the FRI layers checked at src/lib.rs:53-92 (column = the cubic through the four row values at
    x * {1, q, q^2, q^3}, evaluated at the previous root read as an integer), sampled with
    get_pseudorandom_indices (src/utils.rs:53-94).

Pinned: with input 3, 8192 steps and the constants of src/main.rs:209-212 the output is, byte for
byte, the reference's own proof.bin (tests/test_oracle.py).
"""
from __future__ import annotations

import hashlib
import struct

P = 2**256 - 351 * 2**32 + 1
EXTENSION_FACTOR = 8
FRI_SAMPLES = 40      # src/lib.rs:56
SPOT_CHECKS = 80      # src/lib.rs:148


def blake(b: bytes) -> bytes:
    return hashlib.blake2s(b).digest()


def fft(vals, root, inv=False, p=P):
    """out[i] = sum_j vals[j] * root^(i*j); inv=True uses root^-1 and scales by 1/n (src/fft.rs:64-86)."""
    n = len(vals)
    if inv:
        root = pow(root, -1, p)
    out = _fft(list(vals), root, p)
    if inv:
        ninv = pow(n, -1, p)
        out = [x * ninv % p for x in out]
    return out


def _fft(v, root, p):
    n = len(v)
    if n == 1:
        return v
    even = _fft(v[0::2], root * root % p, p)
    odd = _fft(v[1::2], root * root % p, p)
    out = [0] * n
    w = 1
    h = n // 2
    for i in range(h):
        t = odd[i] * w % p
        out[i] = (even[i] + t) % p
        out[i + h] = (even[i] - t) % p
        w = w * root % p
    return out


def multi_inv(values, p=P):
    """Batched inverse, 0 -> 0 (src/utils.rs:169-194)."""
    partials = [1]
    for v in values:
        partials.append(partials[-1] * (v or 1) % p)
    inv = pow(partials[-1], -1, p)
    out = [0] * len(values)
    for i in range(len(values), 0, -1):
        out[i - 1] = partials[i - 1] * inv % p if values[i - 1] else 0
        inv = inv * (values[i - 1] or 1) % p
    return out


def permute4_values(values):
    """Tree order: position 4*i + j holds values[i + j*len/4] (inverse of src/merkle_tree.rs:112-116)."""
    ld4 = len(values) // 4
    out = []
    for i in range(ld4):
        out += [values[i], values[i + ld4], values[i + 2 * ld4], values[i + 3 * ld4]]
    return out


def merkelize(leaves):
    """nodes[1] = root, nodes[i] = blake(nodes[2i] || nodes[2i+1]), leaves (bytes) at nodes[L:2L] in tree order."""
    L = len(leaves)
    nodes = [b""] * L + permute4_values(leaves)
    for i in range(L - 1, 0, -1):
        nodes[i] = blake(nodes[2 * i] + nodes[2 * i + 1])
    return nodes


def branch(nodes, index):
    """(value, sibling value, witnesses bottom-up) for the natural index, as ProofBranch holds them."""
    L = len(nodes) // 2
    ld4 = L // 4
    pos = L + index // ld4 + 4 * (index % ld4)
    value, sib = nodes[pos], nodes[pos ^ 1]
    wit = []
    pos >>= 1
    while pos > 1:
        wit.append(nodes[pos ^ 1])
        pos >>= 1
    return value, sib, wit


def get_pseudorandom_indices(seed, count, modulus, exclude=None):
    # src/utils.rs:53-94
    real = modulus * (exclude - 1) // exclude if exclude else modulus
    data = seed
    while len(data) < 4 * count:
        data += blake(data[-32:])
    out = [int.from_bytes(data[i:i + 4], "big") % real for i in range(0, 4 * count, 4)]
    if exclude:
        out = [1 + x + x // (exclude - 1) for x in out]
    return out


def fold_column(values, root_of_unity, special_x, p=P):
    """column[i] = C_i(special_x), C_i the cubic with C_i(x_i * q^j) = values[i + j*n/4], x_i = root^i."""
    n = len(values)
    m = n // 4
    q = pow(root_of_unity, m, p)
    qinv = pow(q, -1, p)
    inv4 = pow(4, -1, p)
    rinv = pow(root_of_unity, -1, p)
    col = []
    xinv = 1
    for i in range(m):
        r = [values[i + j * m] for j in range(4)]
        t = special_x * xinv % p
        acc, tk = 0, 1
        for k in range(4):
            ck = sum(r[j] * pow(qinv, j * k, p) for j in range(4)) % p   # 4 * c_k * x^k
            acc = (acc + ck * tk) % p
            tk = tk * t % p
        col.append(acc * inv4 % p)
        xinv = xinv * rinv % p
    return col


def prove_low_degree(values, root_of_unity, maxdeg_plus_1, exclude, p=P, tree=None, fold_fault=None):
    """FRI layers as (root2, column branches, poly branches) plus the final values. `tree` is the tree of `values`
    when the caller has built it already; `fold_fault` = (layer, slot): see mk_mimc_proof_faulty."""
    layers = []
    m = tree or merkelize([x.to_bytes(32, "big") for x in values])
    while maxdeg_plus_1 > 16:
        special_x = int.from_bytes(m[1], "big") % p
        column = fold_column(values, root_of_unity, special_x, p)
        m2 = merkelize([x.to_bytes(32, "big") for x in column])
        if fold_fault is not None and fold_fault[0] == len(layers):
            _grind_fold_fault(column, m2, fold_fault[1], exclude, fold_fault[2], p)
        ys = get_pseudorandom_indices(m2[1], FRI_SAMPLES, len(column), exclude)
        poly_positions = [y + len(column) * j for y in ys for j in range(4)]
        layers.append((m2[1], [branch(m2, y) for y in ys], [branch(m, x) for x in poly_positions]))
        values, m = column, m2
        root_of_unity = pow(root_of_unity, 4, p)
        maxdeg_plus_1 //= 4
    return layers, values


def _multiproof_bytes(branches):
    out = [struct.pack("<I", len(branches))]
    for value, sib, wit in branches:
        out.append(struct.pack("<I", len(value)))
        out.append(value)
        out.append(sib)
        out.append(struct.pack("<I", 32 * len(wit)))
        out += wit
    return b"".join(out)


def mimc_trace(inp, steps, round_constants, p=P):
    t = [inp % p]
    for i in range(steps - 1):
        t.append((t[-1] ** 3 + round_constants[i % len(round_constants)]) % p)
    return t


def columns(inp, steps, round_constants, p=P):
    """P, D, B evaluations over the 8x extended domain and the trace output."""
    precision = steps * EXTENSION_FACTOR
    G2 = pow(7, (p - 1) // precision, p)
    G1 = pow(G2, EXTENSION_FACTOR, p)
    skips2 = steps // len(round_constants)
    xs = [1]
    for _ in range(precision - 1):
        xs.append(xs[-1] * G2 % p)
    last = xs[(steps - 1) * EXTENSION_FACTOR]
    trace = mimc_trace(inp, steps, round_constants, p)
    output = trace[-1]
    poly = fft(trace, G1, inv=True)
    p_ev = fft(poly + [0] * (precision - steps), G2)
    mini = fft(list(round_constants), pow(G1, skips2, p), inv=True)
    n_ext = precision // skips2
    mini_ext = fft(mini + [0] * (n_ext - len(mini)), pow(G2, skips2, p))
    c_ev = [(p_ev[(i + EXTENSION_FACTOR) % precision] - pow(p_ev[i], 3, p) - mini_ext[i % n_ext]) % p
            for i in range(precision)]
    z_num_inv = multi_inv([(xs[(i * steps) % precision] - 1) % p for i in range(EXTENSION_FACTOR)], p)
    d_ev = [c_ev[i] * ((xs[i] - last) % p) % p * z_num_inv[i % EXTENSION_FACTOR] % p for i in range(precision)]
    slope = (output - inp) * pow((last - 1) % p, -1, p) % p
    inv_z2 = multi_inv([(x - 1) * (x - last) % p for x in xs], p)
    b_ev = [(p_ev[i] - (inp + slope * (xs[i] - 1))) % p * inv_z2[i] % p for i in range(precision)]
    return p_ev, d_ev, b_ev, output, G2


def _trace_leaves(p_ev, d_ev, b_ev):
    return [a.to_bytes(32, "big") + b.to_bytes(32, "big") + c.to_bytes(32, "big") for a, b, c in zip(p_ev, d_ev, b_ev)]


def _linear_combination(p_ev, d_ev, b_ev, trace_root, G2, steps, p=P):
    """L = D + k1 P + k2 P x^steps + k3 B + k4 B x^steps over the extended domain (src/lib.rs:131-146,190)."""
    k1, k2, k3, k4 = (int.from_bytes(blake(trace_root + bytes([i])), "big") for i in (1, 2, 3, 4))
    g2s = pow(G2, steps, p)
    l_ev, pw = [], 1
    for i in range(len(p_ev)):
        l_ev.append((d_ev[i] + p_ev[i] * k1 + p_ev[i] * k2 * pw + b_ev[i] * k3 + b_ev[i] * pw * k4) % p)
        pw = pw * g2s % p
    return l_ev


def _serialize(mtree, l_mtree, layers, final, positions, precision):
    augmented = []
    for x in positions:
        augmented += [x, (x + EXTENSION_FACTOR) % precision]
    out = [mtree[1], l_mtree[1]]
    for root2, col, poly in layers:
        out += [struct.pack("<I", 1), root2, _multiproof_bytes(col), _multiproof_bytes(poly)]
    out += [struct.pack("<II", 2, 32 * len(final))] + [x.to_bytes(32, "big") for x in final]
    out.append(_multiproof_bytes([branch(mtree, x) for x in augmented]))
    out.append(_multiproof_bytes([branch(l_mtree, x) for x in positions]))
    return b"".join(out)


def mk_mimc_proof(inp, steps, round_constants, p=P):
    """Returns (serialized proof, output)."""
    precision = steps * EXTENSION_FACTOR
    p_ev, d_ev, b_ev, output, G2 = columns(inp, steps, round_constants, p)
    mtree = merkelize(_trace_leaves(p_ev, d_ev, b_ev))
    l_ev = _linear_combination(p_ev, d_ev, b_ev, mtree[1], G2, steps, p)
    l_mtree = merkelize([x.to_bytes(32, "big") for x in l_ev])
    positions = get_pseudorandom_indices(l_mtree[1], SPOT_CHECKS, precision, EXTENSION_FACTOR)
    layers, final = prove_low_degree(l_ev, G2, steps * 2, EXTENSION_FACTOR, p, tree=l_mtree)
    return _serialize(mtree, l_mtree, layers, final, positions, precision), output


# ---- a malicious prover (test infrastructure) -------------------------------------------------------------------
# Proofs whose trees are all consistent (every ProofBranch::verify succeeds, src/merkle_tree.rs:119-171) but whose
# committed data is wrong at exactly one sampled place, so that exactly one of the scalar asserts the reference reaches
# after the branches (src/lib.rs:85,181,187,190) fails. What is sampled depends on the roots, which depend on the data,
# so the fault is ground in: a free value is varied, only its path of the tree rehashed, until the sampler
# (src/utils.rs:53-94) puts the wrong value at the wanted place of the wanted index set.

def _set_leaf(nodes, index, leaf):
    """Replaces the leaf of natural index `index` and rehashes its path; returns the new root."""
    L = len(nodes) // 2
    ld4 = L // 4
    pos = L + index // ld4 + 4 * (index % ld4)
    nodes[pos] = leaf
    pos >>= 1
    while pos:
        nodes[pos] = blake(nodes[2 * pos] + nodes[2 * pos + 1])
        pos >>= 1
    return nodes[1]


def _index_at(seed, slot, modulus, exclude):
    """get_pseudorandom_indices(seed, ..)[slot], hashing only as far as that entry needs."""
    data = seed
    for _ in range(slot // 8):
        data = blake(data)
    x = int.from_bytes(data[4 * (slot % 8):4 * (slot % 8) + 4], "big") % (modulus * (exclude - 1) // exclude)
    return 1 + x + x // (exclude - 1)


def _grind(nodes, leaf_index, leaf_of, target, slot, count, modulus, exclude, limit=4_000_000):
    """Varies leaf `leaf_index` = leaf_of(t), t = 1, 2, .., until index set `count` drawn from the root holds `target`
    at `slot` and nowhere else. Returns t."""
    for t in range(1, limit):
        root = _set_leaf(nodes, leaf_index, leaf_of(t))
        if _index_at(root, slot, modulus, exclude) != target:
            continue
        if get_pseudorandom_indices(root, count, modulus, exclude).count(target) == 1:
            return t
    raise RuntimeError("no tweak found")


def _grind_fold_fault(column, m2, slot, exclude, where, p):
    """column[i] becomes a value the fold does not give, with i = entry `slot` of the layer's own sample."""
    i = where % len(column)
    i += (i % exclude == 0)          # multiples of `exclude` are never sampled
    true = column[i]
    t = _grind(m2, i, lambda t: ((true + t) % p).to_bytes(32, "big"), i, slot, FRI_SAMPLES, len(column), exclude)
    column[i] = (true + t) % p


def mk_mimc_proof_faulty(inp, steps, round_constants, fault, where=12345, p=P, base=None):
    """A proof whose every Merkle branch verifies and whose data is wrong in exactly one sampled place.

    fault = ("fold", layer, slot): the column committed at FRI layer `layer` differs from the fold of the previous
                                   layer at the row the verifier samples as entry `slot` (src/lib.rs:85);
            ("D", slot) / ("B", slot): the D / B value in the trace tree is wrong at the spot-check position drawn as
                                   entry `slot`, L computed from the wrong value (src/lib.rs:181 / :187);
            ("L", slot):           the committed L differs from the linear combination there (src/lib.rs:190).
    `where` picks the domain index that carries the fault. `base` = columns(inp, steps, round_constants) when the caller
    makes several proofs of one instance. Returns (serialized proof, output, index of the fault)."""
    precision = steps * EXTENSION_FACTOR
    p_ev, d_ev, b_ev, output, G2 = base or columns(inp, steps, round_constants, p)
    kind = fault[0]
    i = where % precision
    i += (i % EXTENSION_FACTOR == 0)
    if kind in ("D", "B"):
        d_ev, b_ev = list(d_ev), list(b_ev)
        col = d_ev if kind == "D" else b_ev
        col[i] = (col[i] + 1) % p
    mtree = merkelize(_trace_leaves(p_ev, d_ev, b_ev))
    l_ev = _linear_combination(p_ev, d_ev, b_ev, mtree[1], G2, steps, p)
    l_mtree = merkelize([x.to_bytes(32, "big") for x in l_ev])
    fold_fault = None
    if kind in ("D", "B"):
        # L(0) is free: position 0 is a multiple of the extension factor, which neither the spot checks nor the first
        # FRI layer's rows ever open, and the column of that layer is folded from whatever L the tree holds
        true = l_ev[0]
        t = _grind(l_mtree, 0, lambda t: ((true + t) % p).to_bytes(32, "big"), i, fault[1], SPOT_CHECKS, precision, EXTENSION_FACTOR)
        l_ev[0] = (true + t) % p
    elif kind == "L":
        true = l_ev[i]
        t = _grind(l_mtree, i, lambda t: ((true + t) % p).to_bytes(32, "big"), i, fault[1], SPOT_CHECKS, precision, EXTENSION_FACTOR)
        l_ev[i] = (true + t) % p
    elif kind == "fold":
        fold_fault = (fault[1], fault[2], where)
    else:
        raise ValueError(fault)
    positions = get_pseudorandom_indices(l_mtree[1], SPOT_CHECKS, precision, EXTENSION_FACTOR)
    layers, final = prove_low_degree(l_ev, G2, steps * 2, EXTENSION_FACTOR, p, tree=l_mtree, fold_fault=fold_fault)
    return _serialize(mtree, l_mtree, layers, final, positions, precision), output, i
