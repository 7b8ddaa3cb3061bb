"""Test helper: the reference's protocol layer restated over Python integers.

Restates the reference's src/lib.rs:31-197 (verify_low_degree_proof, verify_mimc_proof) and the
scalar helpers of src/utils.rs it needs, with the two hot-path primitives left pluggable:

    fft_inv(values: list[int], root: int) -> list[int]                (src/fft.rs:64)
    multiproof_verify(branches, indices, root) -> list[bytes] | None  (src/merkle_tree.rs:25)
    get_indices(seed, count, modulus, exclude) -> list[int]           (src/utils.rs:53)

The tests plug in the oracle (to pin it against proof.bin) or the HIP library (GPU parity).
"""
from __future__ import annotations

import hashlib

P = 115792089237316195423570985008687907853269984665640564039457584006405596119041
EXTENSION_FACTOR = 8


def mimc(inp, steps, round_constants, p=P):
    # src/utils.rs:8-16
    out = inp
    for i in range(steps - 1):
        out = (out ** 3 + round_constants[i % len(round_constants)]) % p
    return out


def mimc_constants():
    # src/main.rs:209-212
    return [(i ** 7) ^ 42 for i in range(64)]


def eval_poly_at(poly, x, p=P):
    y, pw = 0, 1
    for c in poly:
        y += pw * c
        pw = pw * x % p
    return y % p


def interp_4(xs, ys, p=P):
    """Coefficients (low to high) of the cubic through four points: src/utils.rs:196-244."""
    out = [0, 0, 0, 0]
    for i in range(4):
        o = [xs[j] for j in range(4) if j != i]
        eq = [(-o[0] * o[1] * o[2]) % p, (o[0] * o[1] + o[0] * o[2] + o[1] * o[2]) % p,
              (-(o[0] + o[1] + o[2])) % p, 1]
        denom = eval_poly_at(eq, xs[i], p)
        f = ys[i] * pow(denom, -1, p) % p if denom else 0
        for k in range(4):
            out[k] = (out[k] + eq[k] * f) % p
    return out


def verify_low_degree_proof(merkle_root, root_of_unity, layers, exclude, hooks, p=P):
    # src/lib.rs:31-97
    return low_degree_status(merkle_root, root_of_unity, layers, exclude, hooks, p) == 0


def low_degree_status(merkle_root, root_of_unity, layers, exclude, hooks, p=P):
    """src/lib.rs:31-97 with the failing step named: 0 passes, 4 a multiproof failed (the `.unwrap()` at :58 / :68),
    5 the assert at :85."""
    test_val, rou_deg = root_of_unity, 1
    while test_val != 1:
        rou_deg *= 2
        test_val = test_val * test_val % p
    quartic = [1, pow(root_of_unity, rou_deg // 4, p), pow(root_of_unity, rou_deg // 2, p),
               pow(root_of_unity, rou_deg * 3 // 4, p)]
    assert rou_deg == 65536
    assert quartic[3] == 80127877722526290441229381276271393407378829608771736609433200039324583025757
    root = merkle_root
    for layer in layers:
        special_x = int.from_bytes(root, "big")
        ys = hooks["get_indices"](layer.root2, 40, rou_deg // 4, exclude)
        column_values = hooks["multiproof_verify"](layer.column_branches.branches, ys, layer.root2)
        if column_values is None:
            return 4
        poly_positions = [y + (rou_deg // 4) * i for y in ys for i in range(4)]
        poly_values = hooks["multiproof_verify"](layer.poly_branches.branches, poly_positions, root)
        if poly_values is None:
            return 4
        for i, y in enumerate(ys):
            x1 = pow(root_of_unity, int(y), p)
            xs = [quartic[j] * x1 % p for j in range(4)]
            rows = [int.from_bytes(poly_values[i * 4 + j], "big") for j in range(4)]
            poly = interp_4(xs, rows, p)
            # :85: the evaluation is reduced, BigInt::from_bytes_be(column value) is not
            if eval_poly_at(poly, special_x, p) != int.from_bytes(column_values[i], "big"):
                return 5
        root_of_unity = pow(root_of_unity, 4, p)
        rou_deg //= 4
        root = layer.root2
    return 0


def verify_mimc_proof(inp, num_steps, round_constants, output, proof, hooks, p=P):
    # src/lib.rs:99-197
    return verify_mimc_proof_status(inp, num_steps, round_constants, output, proof, hooks, p) == 0


def verify_mimc_proof_status(inp, num_steps, round_constants, output, proof, hooks, p=P):
    """src/lib.rs:99-197 with the first failing step named: 0 returns true; 1 returns false at :101-111; 4 a multiproof
    failed (`.unwrap()` at :58, :68, :158, :159); 5 the assert at :85; 6 at :181; 7 at :187; 8 at :190. The instance
    asserts (:120-123) and shape panics stay Python assertions."""
    if num_steps > 2 ** 32 // EXTENSION_FACTOR:
        return 1
    if num_steps & (num_steps - 1) or len(round_constants) & (len(round_constants) - 1):
        return 1
    if len(round_constants) > num_steps:
        return 1
    precision = num_steps * EXTENSION_FACTOR
    G2 = pow(7, (p - 1) // precision, p)
    skips = precision // num_steps
    skips2 = num_steps // len(round_constants)
    val = pow(G2, EXTENSION_FACTOR * skips2, p)
    assert num_steps == 8192
    assert G2 == 41913712888260089065520476180880993127517355946012995597287997778376518235852
    assert val == 56670364103764250102176604807203318908867195832872336813161821519223575486477
    constants_mini_polynomial = hooks["fft_inv"](list(round_constants), val)
    st = low_degree_status(proof.l_merkle_root, G2, proof.fri_layers, EXTENSION_FACTOR, hooks, p)
    if st:
        return st
    k = [int.from_bytes(hashlib.blake2s(proof.merkle_root + bytes([i])).digest(), "big") for i in (1, 2, 3, 4)]
    positions = [int(x) for x in hooks["get_indices"](proof.l_merkle_root, 80, precision, EXTENSION_FACTOR)]
    augmented = []
    for pos in positions:
        augmented += [pos, (pos + skips) % precision]
    values = hooks["multiproof_verify"](proof.merkle_branches.branches, augmented, proof.merkle_root)
    lvalues = hooks["multiproof_verify"](proof.linear_comb_branches.branches, positions, proof.l_merkle_root)
    if values is None or lvalues is None:
        return 4
    last_step_position = pow(G2, (num_steps - 1) * skips, p)
    for i, pos in enumerate(positions):
        x = pow(G2, pos, p)
        x_to_the_steps = pow(x, num_steps, p)
        b1, b2 = values[2 * i], values[2 * i + 1]
        l_of_x = int.from_bytes(lvalues[i], "big")
        p_of_x = int.from_bytes(b1[0:32], "big")
        p_of_g1x = int.from_bytes(b2[0:32], "big")
        d_of_x = int.from_bytes(b1[32:64], "big")
        b_of_x = int.from_bytes(b1[64:96], "big")
        z_value = (x_to_the_steps - 1) * pow((x - last_step_position) % p, -1, p) % p
        k_of_x = eval_poly_at(constants_mini_polynomial, pow(x, skips2, p), p)
        if (p_of_g1x - p_of_x ** 3 - k_of_x - z_value * d_of_x) % p != 0:
            return 6
        # interpolant through (1, inp), (last_step_position, output); zeropoly2 = (x-1)(x-last)
        slope = (output - inp) * pow((last_step_position - 1) % p, -1, p) % p
        interp = (inp + slope * (x - 1)) % p
        zero2 = (x - 1) * (x - last_step_position) % p
        if (p_of_x - b_of_x * zero2 - interp) % p != 0:
            return 7
        if (l_of_x - d_of_x - k[0] * p_of_x - k[1] * p_of_x * x_to_the_steps
                - k[2] * b_of_x - k[3] * b_of_x * x_to_the_steps) % p != 0:
            return 8
    return 0
