/*
 * CPU oracle — TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C restatement of the reference verifier's data-parallel hot path, used as the
 * checker for the HIP library (tests/, the smoke test, and the cpu_baseline leg of bench.py).
 * Nothing in the product path may call, link or load this file.
 *
 * What it restates (paths relative to the reference checkout):
 *   - src/fft.rs:18-35   simple_ft      direct DFT base case (n <= 4)
 *   - src/fft.rs:37-61   _fft           recursive radix-2 decimation in time
 *   - src/fft.rs:64-86   fft_inv        root table, reverse, n^-1 scaling
 *   - src/utils.rs:25-42 negative_to_positive == mathematical mod (fe_sub / fe_add keep [0,p))
 *   - src/merkle_tree.rs:112-116 permute_4_index
 *   - src/merkle_tree.rs:119-171 ProofBranch::verify
 *   - src/merkle_tree.rs:25-44   MultiProof::verify
 *   - src/utils.rs:53-94 get_pseudorandom_indices
 *   - BLAKE2s-256 (crate blake2 0.8.0, not vendored in the reference): RFC 7693, unkeyed.
 *   - num-bigint 0.2.2 arithmetic: exact integer arithmetic mod p = 2^256 - 351*2^32 + 1.
 *
 * Parity pinning: tests/test_oracle.py checks this file against Python big integers,
 * hashlib.blake2s, the RFC 7693 vector, the hex vector at src/merkle_tree.rs:187-204 and all
 * 1,240 branches + the full accept decision of the reference's bundled proof.bin.
 *
 * Field elements cross this API as 8 little-endian uint32 limbs (32 bytes).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;
typedef struct { uint64_t w[4]; } fe; /* little-endian 64-bit words, canonical in [0,p) */

/* p = 2^256 - 351*2^32 + 1 */
static const fe FE_P = {{0xFFFFFEA100000001ULL, 0xFFFFFFFFFFFFFFFFULL, 0xFFFFFFFFFFFFFFFFULL,
                         0xFFFFFFFFFFFFFFFFULL}};
/* 2^256 - p = 351*2^32 - 1 */
#define FE_C 0x15EFFFFFFFFULL

static int fe_geq_p(const fe *a) {
    for (int i = 3; i >= 0; i--) {
        if (a->w[i] > FE_P.w[i]) return 1;
        if (a->w[i] < FE_P.w[i]) return 0;
    }
    return 1;
}

static uint64_t raw_add(fe *r, const fe *a, const fe *b) {
    u128 c = 0;
    for (int i = 0; i < 4; i++) {
        c += (u128)a->w[i] + b->w[i];
        r->w[i] = (uint64_t)c;
        c >>= 64;
    }
    return (uint64_t)c;
}

static uint64_t raw_sub(fe *r, const fe *a, const fe *b) {
    uint64_t borrow = 0;
    for (int i = 0; i < 4; i++) {
        u128 d = (u128)a->w[i] - b->w[i] - borrow;
        r->w[i] = (uint64_t)d;
        borrow = (uint64_t)(d >> 64) & 1;
    }
    return borrow;
}

static void fe_from_limbs(fe *r, const uint32_t *l) {
    for (int i = 0; i < 4; i++) r->w[i] = (uint64_t)l[2 * i] | ((uint64_t)l[2 * i + 1] << 32);
}

static void fe_to_limbs(uint32_t *l, const fe *a) {
    for (int i = 0; i < 4; i++) {
        l[2 * i] = (uint32_t)a->w[i];
        l[2 * i + 1] = (uint32_t)(a->w[i] >> 32);
    }
}

/* any 256-bit value -> canonical residue (2^256 < 2p, so one subtraction suffices) */
static void fe_canon(fe *a) {
    if (fe_geq_p(a)) raw_sub(a, a, &FE_P);
}

static void fe_add(fe *r, const fe *a, const fe *b) {
    uint64_t c = raw_add(r, a, b);
    if (c || fe_geq_p(r)) raw_sub(r, r, &FE_P);
}

static void fe_sub(fe *r, const fe *a, const fe *b) {
    if (raw_sub(r, a, b)) raw_add(r, r, &FE_P);
}

/* r = a*b mod p: 4x4 schoolbook product, then fold the high half with 2^256 == FE_C (mod p) */
static void fe_mul(fe *r, const fe *a, const fe *b) {
    uint64_t t[8] = {0};
    for (int i = 0; i < 4; i++) {
        u128 c = 0;
        for (int j = 0; j < 4; j++) {
            c += (u128)a->w[i] * b->w[j] + t[i + j];
            t[i + j] = (uint64_t)c;
            c >>= 64;
        }
        t[i + 4] = (uint64_t)c;
    }
    /* s = lo + hi*C : five words */
    uint64_t s[5];
    u128 c = 0;
    for (int i = 0; i < 4; i++) {
        c += (u128)t[4 + i] * FE_C + t[i];
        s[i] = (uint64_t)c;
        c >>= 64;
    }
    s[4] = (uint64_t)c; /* < 2^42 */
    /* second fold: s[4]*C < 2^83 */
    u128 f = (u128)s[4] * FE_C;
    fe v;
    c = (u128)s[0] + (uint64_t)f;
    v.w[0] = (uint64_t)c;
    c >>= 64;
    c += (u128)s[1] + (uint64_t)(f >> 64);
    v.w[1] = (uint64_t)c;
    c >>= 64;
    c += s[2];
    v.w[2] = (uint64_t)c;
    c >>= 64;
    c += s[3];
    v.w[3] = (uint64_t)c;
    c >>= 64;
    if (c) { /* wrapped past 2^256 once more: the remainder is tiny, add C */
        fe cc = {{FE_C, 0, 0, 0}};
        raw_add(&v, &v, &cc);
    }
    fe_canon(&v);
    *r = v;
}

static void fe_pow(fe *r, const fe *base, const fe *e) {
    fe acc = {{1, 0, 0, 0}}, b = *base;
    for (int i = 0; i < 256; i++) {
        if ((e->w[i / 64] >> (i % 64)) & 1) fe_mul(&acc, &acc, &b);
        fe_mul(&b, &b, &b);
    }
    *r = acc;
}

static int fe_is_one(const fe *a) { return a->w[0] == 1 && !a->w[1] && !a->w[2] && !a->w[3]; }

/* ---- exported field helpers (for tests) ---- */
void so_fe_add(const uint32_t *a, const uint32_t *b, uint32_t *r) {
    fe x, y, z;
    fe_from_limbs(&x, a); fe_from_limbs(&y, b); fe_canon(&x); fe_canon(&y);
    fe_add(&z, &x, &y); fe_to_limbs(r, &z);
}
void so_fe_sub(const uint32_t *a, const uint32_t *b, uint32_t *r) {
    fe x, y, z;
    fe_from_limbs(&x, a); fe_from_limbs(&y, b); fe_canon(&x); fe_canon(&y);
    fe_sub(&z, &x, &y); fe_to_limbs(r, &z);
}
void so_fe_mul(const uint32_t *a, const uint32_t *b, uint32_t *r) {
    fe x, y, z;
    fe_from_limbs(&x, a); fe_from_limbs(&y, b); fe_canon(&x); fe_canon(&y);
    fe_mul(&z, &x, &y); fe_to_limbs(r, &z);
}
void so_fe_pow(const uint32_t *a, const uint32_t *e, uint32_t *r) {
    fe x, y, z;
    fe_from_limbs(&x, a); fe_from_limbs(&y, e); fe_canon(&x);
    fe_pow(&z, &x, &y); fe_to_limbs(r, &z);
}
/* element-wise batch versions used to check the device field kernels */
void so_fe_batch(int op, const uint32_t *a, const uint32_t *b, uint32_t *r, size_t n) {
    for (size_t i = 0; i < n; i++) {
        if (op == 0) so_fe_add(a + 8 * i, b + 8 * i, r + 8 * i);
        else if (op == 1) so_fe_sub(a + 8 * i, b + 8 * i, r + 8 * i);
        else so_fe_mul(a + 8 * i, b + 8 * i, r + 8 * i);
    }
}

/* ---- NTT, following src/fft.rs ---- */

/* src/fft.rs:18-35 — direct DFT; requires n == number of roots, n <= 4 */
static void simple_ft(const fe *vals, const fe *roots, size_t n, fe *out) {
    for (size_t i = 0; i < n; i++) {
        fe last = {{0, 0, 0, 0}};
        for (size_t j = 0; j < n; j++) {
            fe t;
            fe_mul(&t, &vals[j], &roots[(i * j) % n]);
            fe_add(&last, &last, &t);
        }
        out[i] = last;
    }
}

/* src/fft.rs:37-61 — even/odd split, every second root, recurse, butterfly */
static void fft_rec(const fe *v, const fe *roots, size_t n, fe *out) {
    if (n <= 4) {
        simple_ft(v, roots, n, out);
        return;
    }
    size_t h = n / 2;
    fe *buf = (fe *)malloc(sizeof(fe) * (h * 5));
    fe *right_vals = buf, *left_vals = buf + h, *new_roots = buf + 2 * h, *left = buf + 3 * h,
       *right = buf + 4 * h;
    for (size_t i = 0; i < h; i++) {
        left_vals[i] = v[2 * i];
        right_vals[i] = v[2 * i + 1];
        new_roots[i] = roots[2 * i];
    }
    fft_rec(left_vals, new_roots, h, left);
    fft_rec(right_vals, new_roots, h, right);
    for (size_t i = 0; i < h; i++) {
        fe y_times_root;
        fe_mul(&y_times_root, &right[i], &roots[i]);
        fe_add(&out[i], &left[i], &y_times_root);
        fe_sub(&out[i + h], &left[i], &y_times_root);
    }
    free(buf);
}

static int is_pow2(size_t n) { return n && !(n & (n - 1)); }

/*
 * src/fft.rs:64-86. Returns 0 on success.
 * The reference builds [1, w, w^2, ...] until the power returns to 1; when that table is
 * longer than the input the reference pads the ROOT table and then indexes out of bounds
 * (panic), when it is shorter simple_ft/_fft index past it (panic). Those cases and a
 * non-power-of-two length return -1 here.
 */
int so_fft_inv(const uint32_t *v_limbs, size_t n, const uint32_t *omega_limbs, uint32_t *out_limbs) {
    if (!is_pow2(n)) return -1;
    fe w;
    fe_from_limbs(&w, omega_limbs);
    fe_canon(&w);
    /* roots_of_unity = [1, w, ...] until last == 1  (fft.rs:65-72) */
    size_t cap = n + 2, len = 2;
    fe *roots = (fe *)malloc(sizeof(fe) * cap);
    roots[0] = (fe){{1, 0, 0, 0}};
    roots[1] = w;
    while (!fe_is_one(&roots[len - 1])) {
        if (len == cap) { free(roots); return -1; } /* order exceeds n */
        fe_mul(&roots[len], &roots[len - 1], &w);
        len++;
    }
    if (len != n + 1) { free(roots); return -1; } /* order below n */
    /* reverse and drop the last element (fft.rs:79-80): [w^n=1, w^(n-1), ..., w^1] */
    fe *rev = (fe *)malloc(sizeof(fe) * n);
    for (size_t i = 0; i < n; i++) rev[i] = roots[len - 1 - i];
    free(roots);
    /* invlen = n^(p-2) (fft.rs:82) */
    fe nn = {{(uint64_t)n, 0, 0, 0}}, pm2 = FE_P, invlen;
    pm2.w[0] -= 2;
    fe_pow(&invlen, &nn, &pm2);
    fe *vals = (fe *)malloc(sizeof(fe) * n), *res = (fe *)malloc(sizeof(fe) * n);
    for (size_t i = 0; i < n; i++) { fe_from_limbs(&vals[i], v_limbs + 8 * i); fe_canon(&vals[i]); }
    fft_rec(vals, rev, n, res);
    for (size_t i = 0; i < n; i++) {
        fe_mul(&res[i], &res[i], &invlen);
        fe_to_limbs(out_limbs + 8 * i, &res[i]);
    }
    free(vals); free(res); free(rev);
    return 0;
}

/*
 * Forward transform out[i] = sum_j v[j] w^(ij): the reference has no such wrapper; it is
 * _fft (fft.rs:37) applied to the un-reversed root table [w^0 .. w^(n-1)], no scaling.
 */
int so_fft_fwd(const uint32_t *v_limbs, size_t n, const uint32_t *omega_limbs, uint32_t *out_limbs) {
    if (!is_pow2(n)) return -1;
    fe w;
    fe_from_limbs(&w, omega_limbs);
    fe_canon(&w);
    fe *roots = (fe *)malloc(sizeof(fe) * n);
    roots[0] = (fe){{1, 0, 0, 0}};
    for (size_t i = 1; i < n; i++) fe_mul(&roots[i], &roots[i - 1], &w);
    fe chk;
    fe_mul(&chk, &roots[n - 1], &w);
    int bad = !fe_is_one(&chk);
    for (size_t i = 1; i < n && !bad; i++) bad = fe_is_one(&roots[i]);
    if (bad) { free(roots); return -1; }
    fe *vals = (fe *)malloc(sizeof(fe) * n), *res = (fe *)malloc(sizeof(fe) * n);
    for (size_t i = 0; i < n; i++) { fe_from_limbs(&vals[i], v_limbs + 8 * i); fe_canon(&vals[i]); }
    fft_rec(vals, roots, n, res);
    for (size_t i = 0; i < n; i++) fe_to_limbs(out_limbs + 8 * i, &res[i]);
    free(vals); free(res); free(roots);
    return 0;
}

/* Direct evaluation out = scale * sum_j v[j] * w^(j*k) for one output index k (spot checks). */
void so_dft_at(const uint32_t *v_limbs, size_t n, const uint32_t *omega_limbs, uint64_t k,
               const uint32_t *scale_limbs, uint32_t *out_limbs) {
    fe w, wk, acc = {{0, 0, 0, 0}}, cur = {{1, 0, 0, 0}}, e = {{k, 0, 0, 0}}, sc;
    fe_from_limbs(&w, omega_limbs); fe_canon(&w);
    fe_from_limbs(&sc, scale_limbs); fe_canon(&sc);
    fe_pow(&wk, &w, &e);
    for (size_t j = 0; j < n; j++) {
        fe x, t;
        fe_from_limbs(&x, v_limbs + 8 * j); fe_canon(&x);
        fe_mul(&t, &x, &cur);
        fe_add(&acc, &acc, &t);
        fe_mul(&cur, &cur, &wk);
    }
    fe_mul(&acc, &acc, &sc);
    fe_to_limbs(out_limbs, &acc);
}

/* ---- BLAKE2s-256, RFC 7693, unkeyed ---- */
static const uint32_t B2S_IV[8] = {0x6A09E667, 0xBB67AE85, 0x3C6EF372, 0xA54FF53A,
                                   0x510E527F, 0x9B05688C, 0x1F83D9AB, 0x5BE0CD19};
static const uint8_t B2S_SIGMA[10][16] = {
    {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15},
    {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
    {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4},
    {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
    {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13},
    {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
    {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11},
    {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
    {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5},
    {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0}};

static uint32_t rotr32(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }

static void b2s_compress(uint32_t h[8], const uint8_t block[64], uint64_t t, int last) {
    uint32_t v[16], m[16];
    for (int i = 0; i < 16; i++)
        m[i] = (uint32_t)block[4 * i] | ((uint32_t)block[4 * i + 1] << 8) |
               ((uint32_t)block[4 * i + 2] << 16) | ((uint32_t)block[4 * i + 3] << 24);
    for (int i = 0; i < 8; i++) { v[i] = h[i]; v[i + 8] = B2S_IV[i]; }
    v[12] ^= (uint32_t)t;
    v[13] ^= (uint32_t)(t >> 32);
    if (last) v[14] = ~v[14];
#define B2S_G(a, b, c, d, x, y)                        \
    do {                                               \
        v[a] = v[a] + v[b] + (x); v[d] = rotr32(v[d] ^ v[a], 16); \
        v[c] = v[c] + v[d];       v[b] = rotr32(v[b] ^ v[c], 12); \
        v[a] = v[a] + v[b] + (y); v[d] = rotr32(v[d] ^ v[a], 8);  \
        v[c] = v[c] + v[d];       v[b] = rotr32(v[b] ^ v[c], 7);  \
    } while (0)
    for (int r = 0; r < 10; r++) {
        const uint8_t *s = B2S_SIGMA[r];
        B2S_G(0, 4, 8, 12, m[s[0]], m[s[1]]);
        B2S_G(1, 5, 9, 13, m[s[2]], m[s[3]]);
        B2S_G(2, 6, 10, 14, m[s[4]], m[s[5]]);
        B2S_G(3, 7, 11, 15, m[s[6]], m[s[7]]);
        B2S_G(0, 5, 10, 15, m[s[8]], m[s[9]]);
        B2S_G(1, 6, 11, 12, m[s[10]], m[s[11]]);
        B2S_G(2, 7, 8, 13, m[s[12]], m[s[13]]);
        B2S_G(3, 4, 9, 14, m[s[14]], m[s[15]]);
    }
#undef B2S_G
    for (int i = 0; i < 8; i++) h[i] ^= v[i] ^ v[i + 8];
}

void so_blake2s(const uint8_t *msg, size_t len, uint8_t out[32]) {
    uint32_t h[8];
    for (int i = 0; i < 8; i++) h[i] = B2S_IV[i];
    h[0] ^= 0x01010020u; /* digest 32 bytes, no key, fanout 1, depth 1 */
    uint64_t t = 0;
    while (len > 64) {
        t += 64;
        b2s_compress(h, msg, t, 0);
        msg += 64;
        len -= 64;
    }
    uint8_t blk[64] = {0};
    memcpy(blk, msg, len);
    t += len;
    b2s_compress(h, blk, t, 1);
    for (int i = 0; i < 8; i++) {
        out[4 * i] = (uint8_t)h[i]; out[4 * i + 1] = (uint8_t)(h[i] >> 8);
        out[4 * i + 2] = (uint8_t)(h[i] >> 16); out[4 * i + 3] = (uint8_t)(h[i] >> 24);
    }
}

/* ---- Merkle branch verification, following src/merkle_tree.rs ---- */

/* src/merkle_tree.rs:112-116 with L = leaf count; wrapping u32 arithmetic */
uint32_t so_permute_4_index(uint32_t x, uint32_t L) {
    uint32_t ld4 = L / 4;
    return (x / ld4) + 4u * (x % ld4);
}

/*
 * src/merkle_tree.rs:119-171. witnesses: num_witnesses x 32 bytes, leaf side first.
 * Returns 1 when the recomputed root equals `root` (the reference returns Some(value)),
 * 0 on mismatch (the reference panics in its assert at :165), -1 for shapes on which the
 * reference itself panics (no witnesses: division by zero at :113-114; value shorter than a
 * digest: slice at :133). `digest_out` (optional) receives the recomputed root.
 */
int so_branch_verify(const uint8_t *value, const uint8_t *sibling, uint32_t value_size,
                     const uint8_t *witnesses, uint32_t num_witnesses, const uint8_t root[32],
                     uint32_t a, uint8_t *digest_out) {
    if (num_witnesses < 1 || num_witnesses > 30 || value_size < 32) return -1;
    uint32_t idx = so_permute_4_index(a, 1u << (num_witnesses + 1));
    uint64_t tree_index = (1ull << (num_witnesses + 2)) + idx;
    uint8_t *cat = (uint8_t *)malloc(2 * (size_t)value_size > 64 ? 2 * (size_t)value_size : 64);
    uint8_t res[32];
    if (tree_index % 2 != 0) {
        memcpy(cat, sibling, value_size);
        memcpy(cat + value_size, value, value_size);
    } else {
        memcpy(cat, value, value_size);
        memcpy(cat + value_size, sibling, value_size);
    }
    so_blake2s(cat, 2 * (size_t)value_size, res);
    tree_index /= 2;
    for (uint32_t i = 0; i < num_witnesses; i++) {
        const uint8_t *wit = witnesses + 32 * (size_t)i;
        if (tree_index % 2 != 0) {
            memcpy(cat, wit, 32);
            memcpy(cat + 32, res, 32);
        } else {
            memcpy(cat, res, 32);
            memcpy(cat + 32, wit, 32);
        }
        so_blake2s(cat, 64, res);
        tree_index /= 2;
    }
    free(cat);
    if (digest_out) memcpy(digest_out, res, 32);
    return memcmp(res, root, 32) == 0;
}

/*
 * src/merkle_tree.rs:25-44 over a uniform bucket in the device library's layout:
 * values/siblings [n][value_size], witnesses level-major [num_witnesses][n][32],
 * roots [num_roots][32] with root_ids[n] (NULL -> every branch uses roots[0]).
 * accept[i] = 1/0 per branch. Returns -1 for shapes the reference panics on.
 */
int so_merkle_verify_batch(const uint8_t *values, const uint8_t *siblings, uint32_t value_size,
                           const uint8_t *witnesses, uint32_t num_witnesses,
                           const uint32_t *indices, const uint8_t *roots,
                           const uint32_t *root_ids, size_t n, uint8_t *accept) {
    if (num_witnesses < 1 || num_witnesses > 30 || value_size < 32) return -1;
    uint8_t *w = (uint8_t *)malloc(32 * (size_t)num_witnesses);
    for (size_t i = 0; i < n; i++) {
        for (uint32_t l = 0; l < num_witnesses; l++)
            memcpy(w + 32 * (size_t)l, witnesses + 32 * ((size_t)l * n + i), 32);
        const uint8_t *root = roots + 32 * (size_t)(root_ids ? root_ids[i] : 0);
        accept[i] = (uint8_t)so_branch_verify(values + (size_t)value_size * i,
                                              siblings + (size_t)value_size * i, value_size, w,
                                              num_witnesses, root, indices[i], NULL);
    }
    free(w);
    return 0;
}

/* ---- src/utils.rs:53-94 ---- */
int so_get_pseudorandom_indices(const uint8_t seed[32], size_t count, uint32_t modulus,
                                uint32_t exclude /*0 = None*/, uint32_t *out) {
    uint32_t real_modulus = exclude ? (uint32_t)(modulus * (exclude - 1) / exclude) : modulus;
    if (real_modulus == 0) return -1;
    size_t need = 4 * count, len = 32;
    size_t cap = need + 64;
    uint8_t *hashes = (uint8_t *)malloc(cap);
    memcpy(hashes, seed, 32);
    while (len < need) {
        so_blake2s(hashes + len - 32, 32, hashes + len);
        len += 32;
    }
    for (size_t j = 0; j < count; j++) {
        const uint8_t *b = hashes + 4 * j;
        uint32_t x = ((uint32_t)b[0] << 24) | ((uint32_t)b[1] << 16) | ((uint32_t)b[2] << 8) | b[3];
        out[j] = x % real_modulus;
    }
    if (exclude)
        for (size_t j = 0; j < count; j++) out[j] = 1 + out[j] + out[j] / (exclude - 1);
    free(hashes);
    return 0;
}

/* ---- Merkle tree construction for synthetic test/bench inputs (not in the reference:
 * its prover is external). nodes[1] is the root, nodes[i] = H(nodes[2i] || nodes[2i+1]),
 * leaves at nodes[L + k]. `nodes` has 2L entries of 32 bytes. Leaf values are 32 bytes:
 * the bottom level of the reference's tree holds values directly (merkle_tree.rs:129-141
 * hashes value||sibling as the first step). ---- */
void so_merkle_build(const uint8_t *leaf_values, size_t L, uint8_t *nodes) {
    memcpy(nodes + 32 * L, leaf_values, 32 * L);
    for (size_t i = L - 1; i >= 1; i--) so_blake2s(nodes + 64 * i, 64, nodes + 32 * i);
}
