// Host-side scalar arithmetic mod p = 2^256 - 351*2^32 + 1, used by the library to validate
// roots of unity and to derive inverse roots, n^-1 and the power ladders the table kernels
// start from. A handful of multiplications per call; the transforms themselves run on the GPU.
#pragma once
#include <stdint.h>
#include <string.h>

namespace starkhip {
namespace hostfe {

typedef unsigned __int128 u128;

struct U256 {
    uint64_t w[4];
};

static const U256 MODULUS = {{0xFFFFFEA100000001ULL, ~0ULL, ~0ULL, ~0ULL}};
static const uint64_t FOLD = 0x15EFFFFFFFFULL;  // 2^256 mod p

inline U256 from_limbs(const uint32_t* l) {
    U256 r;
    for (int i = 0; i < 4; i++) r.w[i] = (uint64_t)l[2 * i] | ((uint64_t)l[2 * i + 1] << 32);
    return r;
}

inline void to_limbs(const U256& a, uint32_t* l) {
    for (int i = 0; i < 4; i++) {
        l[2 * i] = (uint32_t)a.w[i];
        l[2 * i + 1] = (uint32_t)(a.w[i] >> 32);
    }
}

inline U256 from_u64(uint64_t x) { return U256{{x, 0, 0, 0}}; }

inline bool eq(const U256& a, const U256& b) { return memcmp(a.w, b.w, sizeof a.w) == 0; }

inline bool geq_modulus(const U256& a) {
    for (int i = 3; i >= 0; i--) {
        if (a.w[i] != MODULUS.w[i]) return a.w[i] > MODULUS.w[i];
    }
    return true;
}

// a + k (k < 2^64) with wraparound at 2^256; returns the carry out
inline uint64_t add_small(U256& a, uint64_t k) {
    u128 c = k;
    for (int i = 0; i < 4; i++) {
        c += a.w[i];
        a.w[i] = (uint64_t)c;
        c >>= 64;
    }
    return (uint64_t)c;
}

inline U256 canon(U256 a) {
    if (geq_modulus(a)) add_small(a, FOLD);  // a - p == a + (2^256 - p) mod 2^256
    return a;
}

inline U256 mul(const U256& a, const U256& b) {
    uint64_t t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
        u128 c = 0;
        for (int j = 0; j < 4; j++) {
            c += (u128)a.w[i] * b.w[j] + t[i + j];
            t[i + j] = (uint64_t)c;
            c >>= 64;
        }
        t[i + 4] = (uint64_t)c;
    }
    // low + high * FOLD, twice, then once more for a possible final wrap
    U256 r;
    u128 c = 0;
    for (int i = 0; i < 4; i++) {
        c += (u128)t[4 + i] * FOLD + t[i];
        r.w[i] = (uint64_t)c;
        c >>= 64;
    }
    u128 f = (u128)(uint64_t)c * FOLD;  // < 2^83
    uint64_t carry = add_small(r, (uint64_t)f);
    U256 hi = {{0, (uint64_t)(f >> 64), 0, 0}};
    u128 cc = 0;
    for (int i = 0; i < 4; i++) {
        cc += (u128)r.w[i] + hi.w[i];
        r.w[i] = (uint64_t)cc;
        cc >>= 64;
    }
    carry += (uint64_t)cc;
    if (carry) add_small(r, FOLD);
    return canon(r);
}

inline U256 pow_u64(U256 base, uint64_t e) {
    U256 acc = from_u64(1);
    while (e) {
        if (e & 1) acc = mul(acc, base);
        base = mul(base, base);
        e >>= 1;
    }
    return acc;
}

inline U256 pow_u256(U256 base, const U256& e) {
    U256 acc = from_u64(1);
    for (int i = 0; i < 256; i++) {
        if ((e.w[i / 64] >> (i % 64)) & 1) acc = mul(acc, base);
        base = mul(base, base);
    }
    return acc;
}

inline U256 inverse(const U256& a) {  // a^(p-2)
    U256 e = MODULUS;
    e.w[0] -= 2;
    return pow_u256(a, e);
}

}  // namespace hostfe
}  // namespace starkhip
