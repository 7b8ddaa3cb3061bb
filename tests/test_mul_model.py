"""An exact integer model of the transform kernels' multiplier (stark_verifier_amd/csrc/field29.cuh: f29_mul and the
split form behind f29_mul_uniform2), used to check what the GPU tests cannot see directly: that no 64-bit column
accumulator can overflow at the operand bounds the kernels rely on (first operand "lazy", limbs below 3 * 2^30; second
operand "tight"), in the narrow form and with wide limbs (the low register as the limb, the high register times 8 as
the next column's start), and that the result is congruent to the product and tight again.
The product semantics are the reference's `(a * b) % modulus` (src/fft.rs:28,53)."""
import math
import random

import pytest

P = 2**256 - 351 * 2**32 + 1
MASK = (1 << 29) - 1
FOLD0, FOLD1 = (1 << 29) - 32, 89855          # 2^261 == FOLD1 * 2^29 + FOLD0 (mod p)
C0, C1 = (1 << 29) - 1, 2807                  # 2^256 == C1 * 2^29 + C0 (mod p)
M64 = (1 << 64) - 1
A_MAX = 3 * 2**30 - 1


class Acc:
    """A 64-bit accumulator that refuses to wrap and remembers its peak."""

    def __init__(self):
        self.peak = 0

    def __call__(self, x):
        assert 0 <= x <= M64, hex(x)
        self.peak = max(self.peak, x)
        return x


def leave(acc, wide):
    """(limb, start of the next column)"""
    if wide:
        return acc & 0xFFFFFFFF, (acc >> 32) * 8
    return acc & MASK, acc >> 29


def second_fold(chk, r, acc):
    top = (acc << 5) | (r[8] >> 24)
    assert top <= M64
    r8 = r[8] & 0xFFFFFF
    u0, u1 = top & MASK, top >> 29
    assert u1 < 2**32
    o = [0] * 9
    acc = chk(u0 * C0 + r[0])
    o[0] = acc & MASK
    acc >>= 29
    assert acc + r[1] < 2**64
    acc = chk(u0 * C1 + acc + r[1])
    acc = chk(u1 * C0 + acc)
    o[1] = acc & MASK
    acc >>= 29
    acc = chk(u1 * C1 + acc + r[2])
    o[2] = acc & MASK
    assert (acc >> 29) < 32
    o[3] = r[3] + (acc >> 29)
    o[4:8] = r[4:8]
    o[8] = r8
    return o


def f29_mul(chk, a, b, wide):
    th = [0] * 9
    acc = 0
    for k in range(9, 17):
        for i in range(k - 8, 9):
            acc = chk(acc + a[i] * b[k - i])
        th[k - 9], acc = leave(acc, wide & 1)
    th[8] = acc
    assert th[8] < 2**32
    r = [0] * 9
    acc = 0
    for k in range(9):
        for i in range(k + 1):
            acc = chk(acc + a[i] * b[k - i])
        acc = chk(acc + th[k] * FOLD0)
        if k >= 1:
            acc = chk(acc + th[k - 1] * FOLD1)
        r[k], acc = leave(acc, (wide & 2) and k < 3)
    acc = chk(acc + th[8] * FOLD1)
    return second_fold(chk, r, acc)


HI_COLS = [[(1, 8, 0), (2, 7, 0), (3, 6, 0), (4, 5, 0), (6, 8, 1), (7, 7, 1), (8, 6, 1)],
           [(2, 8, 0), (3, 7, 0), (4, 6, 0), (7, 8, 1), (8, 7, 1)],
           [(3, 8, 0), (4, 7, 0), (8, 8, 1)],
           [(4, 8, 0)]]


def f29_mul_split(chk, a, b, c, wide):
    """a = a_lo + 2^145 a_hi (limbs 0..4, 5..8); b and c = b * 2^145 mod p"""
    bc = (b, c)
    th = [0] * 5
    acc = 0
    for j, col in enumerate(HI_COLS):
        for i, k, w in col:
            acc = chk(acc + a[i] * bc[w][k])
        th[j], acc = leave(acc, wide & 1)
    th[4] = acc
    assert th[4] < 2**32
    r = [0] * 9
    acc = 0
    for k in range(9):
        for i in range(min(k, 4) + 1):
            acc = chk(acc + a[i] * b[k - i])
        for i in range(5, 9):
            if 0 <= k - (i - 5) <= 8:
                acc = chk(acc + a[i] * c[k - (i - 5)])
        if k <= 4:
            acc = chk(acc + th[k] * FOLD0)
        if 1 <= k <= 5:
            acc = chk(acc + th[k - 1] * FOLD1)
        r[k], acc = leave(acc, (wide & 2) and k < 3)
    return second_fold(chk, r, acc)


def val(l):
    return sum(x << (29 * i) for i, x in enumerate(l))


def limbs(x):
    return [(x >> (29 * i)) & MASK for i in range(9)]


def is_tight(o):
    return all(o[i] < 2**29 for i in (0, 1, 2, 4, 5, 6, 7)) and o[3] < 2**29 + 32 and o[8] < 2**24 and val(o) < 2**256 + 2**92


A_WORST = [A_MAX] * 9
B_WORST = [MASK, MASK, MASK, MASK + 31, MASK, MASK, MASK, MASK, 2**24 - 1 + 8]   # every limb at the top of "tight"


def test_constants():
    assert (FOLD1 << 29) + FOLD0 == 351 * 2**37 - 32 and pow(2, 261, P) == ((FOLD1 << 29) + FOLD0) % P
    assert (C1 << 29) + C0 == 351 * 2**32 - 1 and pow(2, 256, P) == (C1 << 29) + C0


@pytest.mark.parametrize("wide", [0, 2, 3])
def test_full_product_columns_fit_64_bits_and_result_is_tight(wide):
    chk = Acc()
    o = f29_mul(chk, A_WORST, B_WORST, wide)
    assert val(o) % P == val(A_WORST) * val(B_WORST) % P
    assert chk.peak < 2**64 and math.log2(chk.peak) > 63            # the bound is meant to be close
    rng = random.Random(5 + wide)
    for _ in range(3000):
        a = [rng.choice([0, 1, A_MAX, A_MAX - 1, MASK, 1 << 29, rng.randrange(A_MAX + 1)]) for _ in range(9)]
        b = [rng.choice([0, 1, MASK, MASK - 1, rng.getrandbits(29)]) for _ in range(8)] + [rng.choice([0, 2**24 - 1, rng.getrandbits(24)])]
        if rng.random() < 0.3:
            b[3] += 31
        o = f29_mul(chk, a, b, wide)
        assert val(o) % P == val(a) * val(b) % P and is_tight(o), (a, b)


@pytest.mark.parametrize("wide", [0, 2, 3])
def test_split_product_columns_fit_64_bits_and_result_is_tight(wide):
    chk = Acc()
    f29_mul_split(chk, A_WORST, B_WORST, B_WORST, wide)             # bounds only: these two tables do not belong together
    assert chk.peak < 2**64
    rng = random.Random(9 + wide)
    shift = pow(2, 145, P)
    for _ in range(3000):
        a = [rng.choice([0, 1, A_MAX, A_MAX - 1, MASK, 1 << 29, rng.randrange(A_MAX + 1)]) for _ in range(9)]
        bv = rng.choice([0, 1, P - 1, P - 2, rng.getrandbits(256) % P])
        o = f29_mul_split(chk, a, limbs(bv), limbs(bv * shift % P), wide)
        assert val(o) % P == val(a) * bv % P and is_tight(o), (a, bv)


def f29_sqr(chk, a):
    """The squaring routine of field29.cuh (wide limbs in low columns 0..2): every pair i < j once against the doubled
    limb d_i = 2 a_i, the squares a_i^2 where the column index is even."""
    d = [2 * x for x in a]
    assert all(x < 2**32 for x in d[:8])
    wide = 2

    def column(acc, k):
        for i in range(max(0, k - 8), 9):
            j = k - i
            if i < j <= 8:
                acc = chk(acc + d[i] * a[j])
            elif i == j:
                acc = chk(acc + a[i] * a[i])
        return acc

    th = [0] * 9
    acc = 0
    for k in range(9, 17):
        acc = column(acc, k)
        th[k - 9], acc = leave(acc, wide & 1)
    th[8] = acc
    assert th[8] < 2**32
    r = [0] * 9
    acc = 0
    for k in range(9):
        acc = column(acc, k)
        acc = chk(acc + th[k] * FOLD0)
        if k >= 1:
            acc = chk(acc + th[k - 1] * FOLD1)
        r[k], acc = leave(acc, (wide & 2) and k < 3)
    acc = chk(acc + th[8] * FOLD1)
    return second_fold(chk, r, acc)


def test_squaring_routine_equals_the_product_and_keeps_its_bounds():
    # inputs: what the prover's trace walk feeds it, the limbs f29_tighten leaves (limbs 0..7 below 2^29, limb 8 below
    # 2^24 + 8), and tight values as f29_mul returns them (limb 3 up to 31 above)
    chk = Acc()
    worst = [MASK, MASK, MASK, MASK + 31, MASK, MASK, MASK, MASK, 2**24 + 7]
    o = f29_sqr(chk, worst)
    assert val(o) % P == val(worst) ** 2 % P and is_tight(o)
    assert chk.peak < 2**63                                          # far from the accumulator's end: both operands are tight
    rng = random.Random(21)
    for _ in range(3000):
        a = [rng.choice([0, 1, MASK, MASK - 1, rng.getrandbits(29)]) for _ in range(8)] + [rng.choice([0, 2**24 + 7, rng.getrandbits(24)])]
        if rng.random() < 0.3:
            a[3] += 31
        o = f29_sqr(chk, a)
        ref = f29_mul(Acc(), a, a, 2)
        assert o == ref and val(o) % P == val(a) ** 2 % P and is_tight(o), a
