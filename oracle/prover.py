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


def prove_low_degree(values, root_of_unity, maxdeg_plus_1, exclude, p=P):
    """FRI layers as (root2, column tree, ys, poly tree, positions) plus the final values."""
    layers = []
    m = merkelize([x.to_bytes(32, "big") for x in values])
    while maxdeg_plus_1 > 16:
        special_x = int.from_bytes(m[1], "big") % p
        column = fold_column(values, root_of_unity, special_x, p)
        m2 = merkelize([x.to_bytes(32, "big") for x in column])
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


def mk_mimc_proof(inp, steps, round_constants, p=P):
    """Returns (serialized proof, output)."""
    precision = steps * EXTENSION_FACTOR
    p_ev, d_ev, b_ev, output, G2 = columns(inp, steps, round_constants, p)
    mtree = merkelize([a.to_bytes(32, "big") + b.to_bytes(32, "big") + c.to_bytes(32, "big")
                       for a, b, c in zip(p_ev, d_ev, b_ev)])
    k1, k2, k3, k4 = (int.from_bytes(blake(mtree[1] + bytes([i])), "big") for i in (1, 2, 3, 4))
    g2s = pow(G2, steps, p)
    l_ev, pw = [], 1
    for i in range(precision):
        l_ev.append((d_ev[i] + p_ev[i] * k1 + p_ev[i] * k2 * pw + b_ev[i] * k3 + b_ev[i] * pw * k4) % p)
        pw = pw * g2s % p
    l_mtree = merkelize([x.to_bytes(32, "big") for x in l_ev])
    positions = get_pseudorandom_indices(l_mtree[1], SPOT_CHECKS, precision, EXTENSION_FACTOR)
    augmented = []
    for x in positions:
        augmented += [x, (x + EXTENSION_FACTOR) % precision]
    layers, final = prove_low_degree(l_ev, G2, steps * 2, EXTENSION_FACTOR, p)
    out = [mtree[1], l_mtree[1]]
    for root2, col, poly in layers:
        out += [struct.pack("<I", 1), root2, _multiproof_bytes(col), _multiproof_bytes(poly)]
    out += [struct.pack("<II", 2, 32 * len(final))] + [x.to_bytes(32, "big") for x in final]
    out.append(_multiproof_bytes([branch(mtree, x) for x in augmented]))
    out.append(_multiproof_bytes([branch(l_mtree, x) for x in positions]))
    return b"".join(out), output
