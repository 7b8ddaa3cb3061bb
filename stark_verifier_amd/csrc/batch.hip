// Batch driver: verify_mimc_proof (reference src/lib.rs:99-197) over many serialized proofs.
//
// Per chunk of proofs: (1) host threads parse the wire format (src/deserializer.rs:16-140), derive
// the query indices (src/utils.rs:53-94) and flatten every Merkle branch of every proof into
// structure-of-arrays buckets keyed by (value size, witness count); (2) one device call per bucket
// recomputes all branch roots (the data-parallel part: 1,240 branches per proof); (3) host threads
// run the scalar field checks of the protocol (FRI interpolation, transition / boundary / linear
// combination constraints) on the proofs whose branches all matched.
//
// accept[i] = 1 exactly when the reference's verify_mimc_proof returns true for proof i. Where the
// reference panics or returns false, accept[i] = 0 and status[i] names a check that fails (all
// branches are hashed before the scalar checks, so it need not be the first one the reference
// would reach).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <atomic>
#include <exception>
#include <map>
#include <new>
#include <thread>
#include <memory>
#include <vector>

#include "../../include/starkhip.h"
#include "host_blake2s.hpp"
#include "host_field.hpp"
#include "internal.hpp"

namespace starkhip {
namespace batch {

using hostfe::U256;

// ---- scalar field helpers on canonical values ----
static U256 fe_add(const U256& a, const U256& b) {
    U256 r;
    hostfe::u128 c = 0;
    for (int i = 0; i < 4; i++) {
        c += (hostfe::u128)a.w[i] + b.w[i];
        r.w[i] = (uint64_t)c;
        c >>= 64;
    }
    if (c) hostfe::add_small(r, hostfe::FOLD);  // wrapped past 2^256: add 2^256 mod p
    return hostfe::canon(r);
}
static U256 fe_neg(const U256& a) {
    U256 z = {{0, 0, 0, 0}};
    if (hostfe::eq(a, z)) return z;
    U256 r;
    hostfe::u128 borrow = 0;
    for (int i = 0; i < 4; i++) {
        hostfe::u128 d = (hostfe::u128)hostfe::MODULUS.w[i] - a.w[i] - (uint64_t)borrow;
        r.w[i] = (uint64_t)d;
        borrow = (d >> 64) & 1;
    }
    return r;
}
static U256 fe_sub(const U256& a, const U256& b) { return fe_add(a, fe_neg(b)); }
static U256 fe_mul(const U256& a, const U256& b) { return hostfe::mul(a, b); }
// BigInt::from_bytes_be(Sign::Plus, 32 bytes) reduced mod p
static U256 from_be(const uint8_t* b) {
    U256 r;
    for (int i = 0; i < 4; i++) {
        uint64_t w = 0;
        for (int k = 0; k < 8; k++) w = (w << 8) | b[8 * (3 - i) + k];
        r.w[i] = w;
    }
    return hostfe::canon(r);
}
// BigInt::from_bytes_be over a value of any length (values are 32 bytes in every proof seen;
// longer ones are reduced the slow way)
static U256 from_be_n(const uint8_t* b, size_t n) {
    if (n == 32) return from_be(b);
    U256 r = {{0, 0, 0, 0}};
    const U256 k = hostfe::from_u64(256);
    for (size_t i = 0; i < n; i++) r = fe_add(fe_mul(r, k), hostfe::from_u64(b[i]));
    return r;
}
static bool is_zero(const U256& a) { return (a.w[0] | a.w[1] | a.w[2] | a.w[3]) == 0; }
// BigInt::from_bytes_be(n bytes) == v as integers, v canonical (no reduction of the bytes)
static bool raw_be_equals(const uint8_t* b, size_t n, const U256& v) {
    if (n < 32) return false;
    for (size_t i = 0; i + 32 < n; i++)
        if (b[i]) return false;
    b += n - 32;
    for (int i = 0; i < 4; i++) {
        uint64_t w = 0;
        for (int k = 0; k < 8; k++) w = (w << 8) | b[8 * (3 - i) + k];
        if (w != v.w[i]) return false;
    }
    return true;
}

// ---- wire format (src/deserializer.rs) ----
struct BranchRef {
    const uint8_t *value, *sibling, *wit;
    uint32_t vs, w;
};
struct Layer {
    const uint8_t* root2;
    std::vector<BranchRef> col, poly;
};
struct Proof {
    const uint8_t *merkle_root, *l_root;
    std::vector<Layer> layers;
    std::vector<BranchRef> merkle, lincomb;
};
struct Reader {
    const uint8_t* p;
    size_t len, off = 0;
    bool ok = true;
    const uint8_t* bytes(size_t n) {  // read_exact (deserializer.rs:9-14) panics past the end
        if (!ok || n > len - off) {
            ok = false;
            return nullptr;
        }
        const uint8_t* r = p + off;
        off += n;
        return r;
    }
    uint32_t u32() {
        const uint8_t* b = bytes(4);
        return b ? (uint32_t)b[0] | ((uint32_t)b[1] << 8) | ((uint32_t)b[2] << 16) | ((uint32_t)b[3] << 24) : 0;
    }
};
static bool parse_multiproof(Reader& r, std::vector<BranchRef>& out) {  // deserializer.rs:87-140
    uint32_t n = r.u32();
    if (!r.ok) return false;
    out.clear();
    for (uint32_t i = 0; i < n; i++) {
        BranchRef b;
        b.vs = r.u32();
        b.value = r.bytes(b.vs);
        b.sibling = r.bytes(b.vs);
        uint32_t wb = r.u32();
        if (!r.ok || wb % 32 != 0) return false;  // assert at :117
        b.w = wb / 32;
        b.wit = r.bytes(wb);
        if (!r.ok) return false;
        out.push_back(b);
    }
    return true;
}
static bool parse_proof(const uint8_t* p, size_t len, Proof& out) {  // deserializer.rs:16-85
    Reader r{p, len};
    out.merkle_root = r.bytes(32);
    out.l_root = r.bytes(32);
    if (!r.ok) return false;
    for (;;) {
        uint32_t t = r.u32();
        if (!r.ok) return false;
        if (t == 1) {
            Layer l;
            l.root2 = r.bytes(32);
            if (!r.ok || !parse_multiproof(r, l.col) || !parse_multiproof(r, l.poly)) return false;
            out.layers.push_back(std::move(l));
        } else if (t == 2) {
            uint32_t nb = r.u32();
            if (!r.ok || nb == 0 || nb % 32 != 0) return false;  // asserts at :53-54
            r.bytes(nb);  // the points are read and dropped (:56-59)
            if (!r.ok) return false;
            break;
        } else {
            return false;  // panic at :61
        }
    }
    return parse_multiproof(r, out.merkle) && parse_multiproof(r, out.lincomb);  // trailing bytes are tolerated
}

// src/utils.rs:53-94. False when the reduced modulus is zero: the reference's `x % real_modulus` (:85) panics there.
static bool pseudorandom_indices(const uint8_t seed[32], size_t count, uint32_t modulus, uint32_t exclude,
                                 std::vector<uint32_t>& out) {
    const uint32_t real_modulus = exclude ? (uint32_t)(modulus * (exclude - 1) / exclude) : modulus;
    out.clear();
    if (real_modulus == 0) return false;
    std::vector<uint8_t> hashes(seed, seed + 32);
    while (hashes.size() < 4 * count) {
        uint8_t d[32];
        hostb2s::hash(hashes.data() + hashes.size() - 32, 32, d);
        hashes.insert(hashes.end(), d, d + 32);
    }
    out.resize(count);
    for (size_t j = 0; j < count; j++) {
        const uint8_t* b = hashes.data() + 4 * j;
        uint32_t x = ((uint32_t)b[0] << 24) | ((uint32_t)b[1] << 16) | ((uint32_t)b[2] << 8) | b[3];
        out[j] = x % real_modulus;
        if (exclude) out[j] = 1 + out[j] + out[j] / (exclude - 1);
    }
    return true;
}

enum Status : uint8_t {
    ST_ACCEPT = 0,
    ST_PARAMS = 1,        // verify_mimc_proof returns false (lib.rs:101-111)
    ST_PARSE = 2,         // deserializer error / panic
    ST_SHAPE = 3,         // branch count or branch shape on which MultiProof/ProofBranch::verify panics
    ST_MERKLE = 4,        // recomputed root differs (assert at merkle_tree.rs:165)
    ST_LOW_DEGREE = 5,    // lib.rs:85
    ST_TRANSITION = 6,    // lib.rs:181
    ST_BOUNDARY = 7,      // lib.rs:187
    ST_LINEAR_COMB = 8,   // lib.rs:190
    ST_INSTANCE = 9,      // the instance-size asserts at lib.rs:50-51,120-123
    ST_FAST = 0xFF,       // internal: decided by the fast path (batch_dev.hip), skipped by the general one
};

struct Params {
    // input / output of the MiMC instance of proof i: one pair for all (io_stride 0) or one per proof (8)
    const uint32_t *inputs, *outputs;
    size_t io_stride;
    U256 inp(size_t i) const { return hostfe::canon(hostfe::from_limbs(inputs + io_stride * i)); }
    U256 output(size_t i) const { return hostfe::canon(hostfe::from_limbs(outputs + io_stride * i)); }
    U256 inv_last_m1;  // 1 / (last_step - 1)
    uint64_t num_steps;
    std::vector<U256> mini_poly;  // fft_inv(round_constants, val): computed once per call on the device
    U256 g2, last_step;
    uint64_t precision, skips, skips2;
};

// One multiproof's worth of queries: which branches, which indices, which root.
struct Query {
    const std::vector<BranchRef>* branches;
    std::vector<uint32_t> indices;
    const uint8_t* root;
};

struct Bucket {
    uint32_t vs, w;
    std::vector<const BranchRef*> br;
    std::vector<uint32_t> idx, root_id, proof_id;
};

struct ProofWork {
    Proof proof;
    uint8_t status = ST_ACCEPT;
    std::vector<Query> queries;                       // 2 per FRI layer + 2
    std::vector<std::vector<uint32_t>> ys;            // per layer
    std::vector<uint32_t> positions;
    bool plain_values = true;  // every value the scalar checks read as one number is 32 bytes (device lanes copy 32)
};

// The panics MultiProof::verify / ProofBranch::verify raise on a query before any hashing: branch count
// (assert at merkle_tree.rs:28), no witnesses (division by zero at :113), value shorter than 32 bytes (slice at :133);
// more than 30 witnesses overflow the u32 tree size at :120.
static bool query_shape_ok(const Query& q) {
    if (q.branches->size() != q.indices.size()) return false;
    for (const BranchRef& b : *q.branches)
        if (b.w == 0 || b.w > 30 || b.vs < 32) return false;
    return true;
}

// Index derivation and the shape checks the reference makes before hashing, in the reference's order: layer by
// layer (lib.rs:53-92), each layer's two multiproofs checked before the next layer's indices are drawn, so a proof
// with more layers than the domain has factors of four stops where the reference panics and never reaches a zero
// modulus.
static void prepare(ProofWork& pw, const Params& P) {
    const Proof& pr = pw.proof;
    uint64_t rou_deg = P.precision;  // order of G2
    const uint8_t* root = pr.l_root;
    for (const Layer& l : pr.layers) {
        std::vector<uint32_t> ys;
        if (!pseudorandom_indices(l.root2, 40, (uint32_t)(rou_deg / 4), 8, ys)) {  // `% 0` panic at utils.rs:85
            pw.status = ST_SHAPE;
            return;
        }
        std::vector<uint32_t> pp;
        for (uint32_t y : ys)
            for (uint32_t i = 0; i < 4; i++) pp.push_back(y + (uint32_t)(rou_deg / 4) * i);
        pw.queries.push_back(Query{&l.col, ys, l.root2});
        pw.queries.push_back(Query{&l.poly, pp, root});
        if (!query_shape_ok(pw.queries[pw.queries.size() - 2]) || !query_shape_ok(pw.queries.back())) {
            pw.status = ST_SHAPE;
            return;
        }
        pw.ys.push_back(std::move(ys));
        rou_deg /= 4;
        root = l.root2;
    }
    if (!pseudorandom_indices(pr.l_root, 80, (uint32_t)P.precision, 8, pw.positions)) {
        pw.status = ST_SHAPE;
        return;
    }
    std::vector<uint32_t> aug;
    for (uint32_t p : pw.positions) {
        aug.push_back(p);
        aug.push_back((uint32_t)((p + P.skips) % P.precision));
    }
    pw.queries.push_back(Query{&pr.merkle, aug, pr.merkle_root});
    pw.queries.push_back(Query{&pr.lincomb, pw.positions, pr.l_root});
    if (!query_shape_ok(pw.queries[pw.queries.size() - 2]) || !query_shape_ok(pw.queries.back())) {
        pw.status = ST_SHAPE;
        return;
    }
    for (const Layer& l : pr.layers) {
        for (const BranchRef& b : l.col) pw.plain_values &= b.vs == 32;
        for (const BranchRef& b : l.poly) pw.plain_values &= b.vs == 32;
    }
    for (const BranchRef& b : pr.lincomb) pw.plain_values &= b.vs == 32;
    // the main trace leaves are read as three 32-byte fields (lib.rs:171-174)
    for (size_t i = 0; i < pr.merkle.size(); i += 2)
        if (pr.merkle[i].vs < 96) {
            pw.status = ST_SHAPE;
            return;
        }
}

// utils.rs:196-244 for one group of four points; returns the cubic's coefficients, low to high
static void interp_4(const U256 xs[4], const U256 ys[4], U256 out[4]) {
    const U256 one = hostfe::from_u64(1);
    for (int k = 0; k < 4; k++) out[k] = U256{{0, 0, 0, 0}};
    for (int i = 0; i < 4; i++) {
        U256 o[3];
        for (int j = 0, k = 0; j < 4; j++)
            if (j != i) o[k++] = xs[j];
        U256 eq[4];
        eq[0] = fe_neg(fe_mul(fe_mul(o[0], o[1]), o[2]));
        eq[1] = fe_add(fe_add(fe_mul(o[0], o[1]), fe_mul(o[0], o[2])), fe_mul(o[1], o[2]));
        eq[2] = fe_neg(fe_add(fe_add(o[0], o[1]), o[2]));
        eq[3] = one;
        const U256 x = xs[i], x2 = fe_mul(x, x), x3 = fe_mul(x2, x);
        const U256 denom = fe_add(fe_add(eq[0], fe_mul(eq[1], x)), fe_add(fe_mul(eq[2], x2), x3));
        const U256 f = fe_mul(ys[i], hostfe::inverse(denom));  // inv(0) = 0 as in utils.rs:144-146
        for (int k = 0; k < 4; k++) out[k] = fe_add(out[k], fe_mul(eq[k], f));
    }
}

// The scalar checks of lib.rs:31-97 and :161-192 on a proof whose branches all matched.
static uint8_t field_checks(const ProofWork& pw, const Params& P, size_t gi) {
    const Proof& pr = pw.proof;
    U256 rou = P.g2;
    uint64_t rou_deg = P.precision;
    const uint8_t* root = pr.l_root;
    for (size_t li = 0; li < pr.layers.size(); li++) {
        const Layer& l = pr.layers[li];
        const U256 quartic[4] = {hostfe::from_u64(1), hostfe::pow_u64(rou, rou_deg / 4), hostfe::pow_u64(rou, rou_deg / 2),
                                 hostfe::pow_u64(rou, rou_deg * 3 / 4)};
        const U256 special_x = from_be(root);
        const U256 sx2 = fe_mul(special_x, special_x), sx3 = fe_mul(sx2, special_x);
        for (size_t i = 0; i < pw.ys[li].size(); i++) {
            const U256 x1 = hostfe::pow_u64(rou, pw.ys[li][i]);
            U256 xs[4], rows[4], poly[4];
            for (int j = 0; j < 4; j++) {
                xs[j] = fe_mul(quartic[j], x1);
                rows[j] = from_be_n(l.poly[i * 4 + j].value, l.poly[i * 4 + j].vs);
            }
            interp_4(xs, rows, poly);
            const U256 got = fe_add(fe_add(poly[0], fe_mul(poly[1], special_x)), fe_add(fe_mul(poly[2], sx2), fe_mul(poly[3], sx3)));
            // :85 compares the reduced evaluation with BigInt::from_bytes_be(column value), which is not reduced: a
            // column value at or above p (or with non-zero bytes in front of its last 32) can never be equal
            if (!raw_be_equals(l.col[i].value, l.col[i].vs, got)) return ST_LOW_DEGREE;
        }
        rou = hostfe::pow_u64(rou, 4);
        rou_deg /= 4;
        root = l.root2;
    }
    U256 k[4];
    for (int i = 0; i < 4; i++) {
        uint8_t msg[33], d[32];
        memcpy(msg, pr.merkle_root, 32);
        msg[32] = (uint8_t)(i + 1);
        hostb2s::hash(msg, 33, d);
        k[i] = from_be(d);
    }
    const U256 one = hostfe::from_u64(1);
    const U256 inp = P.inp(gi);
    const U256 slope = fe_mul(fe_sub(P.output(gi), inp), P.inv_last_m1);
    for (size_t i = 0; i < pw.positions.size(); i++) {
        const U256 x = hostfe::pow_u64(P.g2, pw.positions[i]);
        const U256 x_steps = hostfe::pow_u64(x, P.num_steps);
        const uint8_t *b1 = pr.merkle[2 * i].value, *b2 = pr.merkle[2 * i + 1].value;
        const U256 l_of_x = from_be_n(pr.lincomb[i].value, pr.lincomb[i].vs);
        const U256 p_of_x = from_be(b1), p_of_g1x = from_be(b2), d_of_x = from_be(b1 + 32), b_of_x = from_be(b1 + 64);
        const U256 z = fe_mul(fe_sub(x_steps, one), hostfe::inverse(fe_sub(x, P.last_step)));
        // k_of_x = constants_mini_polynomial(x^skips2)
        const U256 xk = hostfe::pow_u64(x, P.skips2);
        U256 kx = {{0, 0, 0, 0}}, pwx = one;
        for (const U256& c : P.mini_poly) {
            kx = fe_add(kx, fe_mul(pwx, c));
            pwx = fe_mul(pwx, xk);
        }
        const U256 p3 = fe_mul(fe_mul(p_of_x, p_of_x), p_of_x);
        if (!is_zero(fe_sub(fe_sub(fe_sub(p_of_g1x, p3), kx), fe_mul(z, d_of_x)))) return ST_TRANSITION;
        const U256 interp = fe_add(inp, fe_mul(slope, fe_sub(x, one)));
        const U256 zero2 = fe_mul(fe_sub(x, one), fe_sub(x, P.last_step));
        if (!is_zero(fe_sub(fe_sub(p_of_x, fe_mul(b_of_x, zero2)), interp))) return ST_BOUNDARY;
        U256 lc = fe_sub(l_of_x, d_of_x);
        lc = fe_sub(lc, fe_mul(k[0], p_of_x));
        lc = fe_sub(lc, fe_mul(fe_mul(k[1], p_of_x), x_steps));
        lc = fe_sub(lc, fe_mul(k[2], b_of_x));
        lc = fe_sub(lc, fe_mul(fe_mul(k[3], b_of_x), x_steps));
        if (!is_zero(lc)) return ST_LINEAR_COMB;
    }
    return ST_ACCEPT;
}

// f(i) for every i below n on up to `threads` host threads. The threads are kept between calls (one pool per process);
// a second context calling at the same moment starts its own, as every call did before the pool existed.
static WorkerPool& host_pool() {
    static WorkerPool pool(15);
    return pool;
}
static std::mutex host_pool_mutex;

// An exception out of f (allocation failure in the parsers) is caught where it is thrown, on whichever thread, the
// remaining indices are dropped, and the first one is thrown again here once every thread has left f: nothing unwinds
// past the pool while its workers still hold references to this frame. A call made from inside f runs serially (the
// pool's mutex is never tried by the thread that holds it).
template <class F>
static void parallel_for(size_t n, unsigned threads, F f) {
    static thread_local bool inside = false;
    if (threads <= 1 || n <= 1 || inside) {
        for (size_t i = 0; i < n; i++) f(i);
        return;
    }
    std::atomic<size_t> next{0};
    std::exception_ptr first;
    std::mutex first_mutex;
    const size_t tasks = std::min<size_t>(threads, n);
    auto body = [&](size_t) {
        inside = true;
        try {
            for (size_t i; (i = next.fetch_add(1)) < n;) f(i);
        } catch (...) {
            std::lock_guard<std::mutex> g(first_mutex);
            if (!first) first = std::current_exception();
            next.store(n);  // the other threads stop at their next index
        }
        inside = false;
    };
    {
        std::unique_lock<std::mutex> lk(host_pool_mutex, std::try_to_lock);
        if (lk.owns_lock()) {
            host_pool().run(tasks, body);
        } else {
            std::vector<std::thread> pool;
            for (size_t t = 0; t < tasks; t++) pool.emplace_back([&] { body(t); });
            for (auto& th : pool) th.join();
        }
    }
    if (first) std::rethrow_exception(first);
}

// The same checks on the device: one lane per (proof, layer, sample) and per (proof, position).
static int device_field_checks(stark_ctx* ctx, std::vector<ProofWork>& work, size_t first, const Params& P,
                               const CheckParams& CP, const uint32_t* d_mini, unsigned threads) {
    const size_t m = work.size();
    std::vector<size_t> fri_off(m + 1, 0), con_off(m + 1, 0);
    for (size_t i = 0; i < m; i++) {
        size_t f = 0, c = 0;
        if (work[i].status == ST_ACCEPT && !work[i].plain_values) work[i].status = field_checks(work[i], P, first + i);  // rare: host path
        else if (work[i].status == ST_ACCEPT) {
            for (auto& ys : work[i].ys) f += ys.size();
            c = work[i].positions.size();
        }
        fri_off[i + 1] = fri_off[i] + f;
        con_off[i + 1] = con_off[i] + c;
    }
    const size_t nf = fri_off[m], nc = con_off[m];
    if (nf + nc == 0) return STARK_OK;
    std::vector<FriLane> fri(nf);
    std::vector<ConstraintLane> con(nc);
    std::vector<uint8_t> kk(CHECK_PROOF_BYTES * m, 0);  // per proof: k1..k4, input, slope
    parallel_for(m, threads, [&](size_t i) {
        ProofWork& pw = work[i];
        if (pw.status != ST_ACCEPT || !pw.plain_values) return;
        const Proof& pr = pw.proof;
        size_t o = fri_off[i];
        const uint8_t* root = pr.l_root;
        for (size_t li = 0; li < pr.layers.size(); li++) {
            const Layer& l = pr.layers[li];
            for (size_t k = 0; k < pw.ys[li].size(); k++, o++) {
                FriLane& L = fri[o];
                for (int j = 0; j < 4; j++) memcpy(L.rows[j], l.poly[k * 4 + j].value, 32);
                memcpy(L.col, l.col[k].value, 32);
                memcpy(L.special, root, 32);
                L.y = pw.ys[li][k];
                L.layer = (uint32_t)li;
                L.pad[0] = L.pad[1] = 0;
            }
            root = l.root2;
        }
        for (int k = 0; k < 4; k++) {
            uint8_t msg[33];
            memcpy(msg, pr.merkle_root, 32);
            msg[32] = (uint8_t)(k + 1);
            hostb2s::hash(msg, 33, &kk[CHECK_PROOF_BYTES * i + 32 * k]);
        }
        {
            const U256 inp = P.inp(first + i);
            uint32_t l[16];
            hostfe::to_limbs(inp, l);
            hostfe::to_limbs(fe_mul(fe_sub(P.output(first + i), inp), P.inv_last_m1), l + 8);
            memcpy(&kk[CHECK_PROOF_BYTES * i + 128], l, 64);
        }
        o = con_off[i];
        for (size_t k = 0; k < pw.positions.size(); k++, o++) {
            ConstraintLane& L = con[o];
            memcpy(L.b1, pr.merkle[2 * k].value, 96);
            memcpy(L.pg, pr.merkle[2 * k + 1].value, 32);
            memcpy(L.l, pr.lincomb[k].value, 32);
            L.pos = pw.positions[k];
            L.proof = (uint32_t)i;
            L.pad[0] = L.pad[1] = 0;
        }
    });
    const size_t bytes_f = nf * sizeof(FriLane), bytes_c = nc * sizeof(ConstraintLane), bytes_k = kk.size();
    auto up16 = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t o_c = up16(bytes_f), o_k = o_c + up16(bytes_c), o_s = o_k + up16(bytes_k), total = o_s + up16(nf + nc);
    uint8_t* d = nullptr;
    hipError_t e = hipMalloc((void**)&d, total);
    if (e != hipSuccess) return STARK_ERR_NOMEM;
    hipStream_t st = (hipStream_t)stark_ctx_stream(ctx);
    std::vector<uint8_t> res(nf + nc);
    e = hipMemcpyAsync(d, fri.data(), bytes_f, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(d + o_c, con.data(), bytes_c, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(d + o_k, kk.data(), bytes_k, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = launch_fri_check((const FriLane*)d, nf, CP, d + o_s, st);
    if (e == hipSuccess)
        e = launch_constraint_check((const ConstraintLane*)(d + o_c), nc, CP, d + o_k, d_mini, d + o_s + nf, st);
    if (e == hipSuccess) e = hipMemcpyAsync(res.data(), d + o_s, nf + nc, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipFree(d);
    if (e != hipSuccess) return STARK_ERR_HIP;
    for (size_t i = 0; i < m; i++) {
        if (work[i].status != ST_ACCEPT || !work[i].plain_values) continue;
        uint8_t stt = ST_ACCEPT;
        for (size_t k = fri_off[i]; k < fri_off[i + 1] && !stt; k++) stt = res[k];
        for (size_t k = con_off[i]; k < con_off[i + 1] && !stt; k++) stt = res[nf + k];
        work[i].status = stt;
    }
    (void)P;
    return STARK_OK;
}

}  // namespace batch
}  // namespace starkhip

using namespace starkhip;
using namespace starkhip::batch;

// Nothing may unwind across the C ABI: every entry point is a function-try-block.
#define STARK_CATCH                                              \
    catch (const std::bad_alloc&) { return STARK_ERR_NOMEM; }    \
    catch (...) { return STARK_ERR_HIP; }


// Everything verify_mimc_proof derives from (num_steps, round_constants) before it looks at a proof.
struct Setup {
    Params P;
    CheckParams CP;
    uint32_t* d_mini = nullptr;  // constants_mini_polynomial on the device (owned by the context's SetupCache)
    bool field_on_host = false, skip_merkle = false, force_flatten = false, force_general = false;
};

// What make_setup derives from (num_steps, round_constants) alone, kept in the context: two 256-bit powers, an inverse
// and a 64-point inverse transform with its upload and download are a fifth of a millisecond per call, as much as the
// kernels of 200 proofs. One parameter set is kept at a time; every call ends synchronised, so replacing it is safe.
struct SetupCache : CtxAttachment {
    uint64_t num_steps = 0;
    std::vector<uint32_t> consts;
    Params P;
    CheckParams CP;
    uint32_t* d_mini = nullptr;
    uint32_t* d_pw = nullptr;    // CheckParams::pw / kext: powers of g2 and the constants' polynomial over them
    uint32_t* d_kext = nullptr;
    ~SetupCache() override {
        (void)hipFree(d_mini);
        (void)hipFree(d_pw);
        (void)hipFree(d_kext);
    }
};
constexpr int SETUP_SLOT = 2;

// early: ST_ACCEPT when proofs have to be looked at, else the status every proof gets (lib.rs:101-111, :120-123)
static int make_setup(stark_ctx* ctx, uint64_t num_steps, const uint32_t* round_constants, size_t n_constants, Setup& S,
                      uint8_t& early, unsigned flags = 0) {
    Params& P = S.P;
    CheckParams& CP = S.CP;
    early = ST_ACCEPT;
    // parameter checks of lib.rs:101-111: the reference returns false
    auto pow2 = [](uint64_t x) { return x && !(x & (x - 1)); };
    if (num_steps > (1ull << 32) / 8 || !pow2(num_steps) || !pow2(n_constants) || n_constants > num_steps) {
        early = ST_PARAMS;
        return STARK_OK;
    }
    // where the scalar checks run: on the device unless STARKHIP_BATCH_FIELD=host (equivalent paths)
    const char* fenv = getenv("STARKHIP_BATCH_FIELD");
    S.field_on_host = (fenv && strcmp(fenv, "host") == 0) || (flags & STARK_DIAG_HOST_FIELD_CHECKS);
#ifdef STARKHIP_DIAG
    // test build only (libstarkhip_diag.so), through stark_mimc_verify_batch_diag alone: the scalar checks decide even
    // when a branch mismatched, so damaged values exercise every check. The shipped library has no such switch.
    S.skip_merkle = (flags & STARK_DIAG_IGNORE_MERKLE) != 0;
#endif
    const char* menv = getenv("STARKHIP_BATCH_MERKLE");
    S.force_flatten = menv && strcmp(menv, "flatten") == 0;
    const char* penv = getenv("STARKHIP_BATCH_PATH");
    S.force_general = (penv && strcmp(penv, "general") == 0) || (flags & STARK_DIAG_GENERAL_PATH);
    if (SetupCache* c = static_cast<SetupCache*>(ctx_get_attachment(ctx, SETUP_SLOT))) {
        if (c->num_steps == num_steps && c->consts.size() == 8 * n_constants &&
            memcmp(c->consts.data(), round_constants, 32 * n_constants) == 0) {
            S.P = c->P;
            S.CP = c->CP;
            S.d_mini = c->d_mini;
            return STARK_OK;
        }
    }
    P.num_steps = num_steps;
    P.precision = num_steps * 8;
    P.skips = 8;
    P.skips2 = num_steps / n_constants;
    // asserts at lib.rs:120 (num_steps), :123 (val, which fixes 64 round constants), :50 (rou_deg == 65536)
    if (num_steps != 8192 || n_constants != 64) {
        early = ST_INSTANCE;
        return STARK_OK;
    }
    {
        // G2 = 7^((p-1)/precision) (lib.rs:114); (p-1)/2^16 = p >> 16 because p-1 is a multiple of 2^32
        U256 e = hostfe::MODULUS;
        e.w[0] -= 1;
        unsigned sh = 0;
        while ((1ull << sh) < P.precision) sh++;
        U256 q;
        for (int i = 0; i < 4; i++) q.w[i] = (e.w[i] >> sh) | (i < 3 && sh ? e.w[i + 1] << (64 - sh) : 0);
        P.g2 = hostfe::pow_u256(hostfe::from_u64(7), q);
    }
    P.last_step = hostfe::pow_u64(P.g2, (num_steps - 1) * P.skips);
    P.inv_last_m1 = hostfe::inverse(fe_sub(P.last_step, hostfe::from_u64(1)));
    {
        // constants_mini_polynomial = fft_inv(round_constants, val) (lib.rs:118,125): on the device
        const U256 val = hostfe::pow_u64(P.g2, 8 * P.skips2);
        uint32_t root[8], mod[8];
        hostfe::to_limbs(val, root);
        hostfe::to_limbs(hostfe::MODULUS, mod);
        std::vector<uint32_t> out(8 * n_constants);
        int rc = stark_fft_inv(ctx, round_constants, n_constants, root, mod, out.data());
        if (rc != STARK_OK) return rc;
        for (size_t i = 0; i < n_constants; i++) P.mini_poly.push_back(hostfe::from_limbs(out.data() + 8 * i));
    }
    hostfe::to_limbs(P.g2, CP.g2);
    hostfe::to_limbs(hostfe::pow_u64(P.g2, P.precision / 4), CP.q4);
    hostfe::to_limbs(P.last_step, CP.last_step);
    CP.precision = (uint32_t)P.precision;
    CP.log_steps = CP.log_skips2 = 0;
    while ((1ull << CP.log_steps) < P.num_steps) CP.log_steps++;
    while ((1ull << CP.log_skips2) < P.skips2) CP.log_skips2++;
    CP.n_coeffs = (uint32_t)P.mini_poly.size();
    std::unique_ptr<SetupCache> c(new SetupCache);
    {
        std::vector<uint32_t> mp(8 * P.mini_poly.size());
        for (size_t i = 0; i < P.mini_poly.size(); i++) hostfe::to_limbs(hostfe::canon(P.mini_poly[i]), &mp[8 * i]);
        if (hipMalloc((void**)&c->d_mini, mp.size() * 4) != hipSuccess) return STARK_ERR_NOMEM;
        if (hipMemcpy(c->d_mini, mp.data(), mp.size() * 4, hipMemcpyHostToDevice) != hipSuccess) return STARK_ERR_HIP;
    }
    // tables of the scalar checks (2 MiB + 16 KiB), built on the device by the arithmetic a lane would otherwise run
    // for itself; STARKHIP_CHECK_TABLES=0 leaves them out (every lane then computes its own powers, as before)
    if (!(getenv("STARKHIP_CHECK_TABLES") && atoi(getenv("STARKHIP_CHECK_TABLES")) == 0) && CP.precision == 65536u) {
        hipStream_t st = (hipStream_t)stark_ctx_stream(ctx);
        if (hipMalloc((void**)&c->d_pw, 32 * (size_t)CP.precision) != hipSuccess ||
            hipMalloc((void**)&c->d_kext, 32 * (size_t)(CP.precision >> CP.log_skips2)) != hipSuccess)
            return STARK_ERR_NOMEM;
        if (launch_check_tables(CP, c->d_mini, c->d_pw, c->d_kext, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
            return STARK_ERR_HIP;
        CP.pw = c->d_pw;
        CP.kext = c->d_kext;
    }
    c->num_steps = num_steps;
    c->consts.assign(round_constants, round_constants + 8 * n_constants);
    c->P = P;
    c->CP = CP;
    S.d_mini = c->d_mini;
    ctx_set_attachment(ctx, SETUP_SLOT, c.release());
    return STARK_OK;
}

static int verify_impl(stark_ctx* ctx, const uint8_t* proofs, const uint64_t* offsets, size_t n_proofs, const uint32_t* inp,
                       const uint32_t* output, size_t io_stride, uint64_t num_steps, const uint32_t* round_constants,
                       size_t n_constants, uint8_t* accept, uint8_t* status, unsigned flags = 0) {
    if (!ctx) return STARK_ERR_INVALID;
    if (ctx_make_current(ctx) != hipSuccess) return STARK_ERR_HIP;
    if (n_proofs && (!proofs || !offsets || !accept)) return STARK_ERR_INVALID;
    if (((n_proofs || !io_stride) && (!inp || !output)) || (n_constants && !round_constants)) return STARK_ERR_INVALID;
    for (size_t i = 0; i < n_proofs; i++)
        if (offsets[i + 1] < offsets[i]) return STARK_ERR_INVALID;
    auto set_all = [&](uint8_t st) {
        for (size_t i = 0; i < n_proofs; i++) {
            accept[i] = st == ST_ACCEPT;
            if (status) status[i] = st;
        }
        return STARK_OK;
    };
    Setup S;
    uint8_t early;
    {
        int rc = make_setup(ctx, num_steps, round_constants, n_constants, S, early, flags);
        if (rc != STARK_OK) return rc;
    }
    if (early != ST_ACCEPT) return set_all(early);
    Params& P = S.P;
    P.inputs = inp;
    P.outputs = output;
    P.io_stride = io_stride;
    const CheckParams& CP = S.CP;
    const uint32_t* d_mini = S.d_mini;
    const bool field_on_host = S.field_on_host, skip_merkle = S.skip_merkle, force_flatten = S.force_flatten;
    unsigned threads = std::thread::hardware_concurrency();
    if (const char* s = getenv("STARKHIP_HOST_THREADS")) threads = (unsigned)atoi(s);
    threads = std::max(1u, std::min(threads, 16u));
    size_t chunk_bytes = (size_t)1 << 30;
    if (const char* s = getenv("STARKHIP_BATCH_MB")) chunk_bytes = (size_t)atoi(s) << 20;
    size_t piece_bytes = (size_t)64 << 20;  // upload granule of the pipelined fast path (one staging buffer)
    if (const char* s = getenv("STARKHIP_PIECE_MB"))
        if (atoi(s) > 0) piece_bytes = (size_t)atoi(s) << 20;

    // STARKHIP_BATCH_TIMING=1 prints per-stage wall time to stderr
    const bool timing = getenv("STARKHIP_BATCH_TIMING") != nullptr;
    double t_parse = 0, t_bucket = 0, t_flatten = 0, t_merkle = 0, t_field = 0, t_classify = 0, t_fast = 0;
    size_t n_parsed = 0;  // proofs that took the parsing path
    // STARKHIP_BATCH_PATH=general: every proof takes the parsing path (the other switches imply it)
    const bool use_fast = !S.force_general && !field_on_host && !force_flatten && !skip_merkle;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    for (size_t c0 = 0; c0 < n_proofs;) {
        size_t c1 = c0 + 1;
        // at most 65535 proofs per chunk: the fast path's kernels index proofs by grid.y
        while (c1 < n_proofs && c1 - c0 < 65535 && offsets[c1 + 1] - offsets[c0] <= chunk_bytes) c1++;
        const size_t m = c1 - c0;
        std::vector<ProofWork> work(m);
        std::vector<uint8_t> fast_status(m, 0);
        double t0 = now();
        const uint8_t* chunk_base = proofs + offsets[c0];
        const size_t chunk_len = (size_t)(offsets[c1] - offsets[c0]);
        // proofs with the fixed instance's structure words are decided on the device without being parsed here
        std::vector<uint8_t> canon(m, 0);
        auto classify = [&](size_t i0, size_t i1) {
            parallel_for((i1 - i0 + 15) / 16, threads, [&](size_t blk) {
                for (size_t i = i0 + blk * 16; i < std::min(i1, i0 + blk * 16 + 16); i++) {
                    const uint64_t a = offsets[c0 + i], b = offsets[c0 + i + 1];
                    canon[i] = fast_is_canonical(proofs + a, (size_t)(b - a));
                }
            });
        };
        // (input, slope) pairs of proofs who[k0 .. k1) into io
        auto make_io = [&](const std::vector<size_t>& who, size_t k0, size_t k1, std::vector<uint32_t>& io) {
            parallel_for((k1 - k0 + 255) / 256, threads, [&](size_t blk) {
                for (size_t k = k0 + blk * 256; k < std::min(k1, k0 + blk * 256 + 256); k++) {
                    const U256 in = P.inp(c0 + who[k]);
                    hostfe::to_limbs(in, &io[16 * k]);
                    hostfe::to_limbs(fe_mul(fe_sub(P.output(c0 + who[k]), in), P.inv_last_m1), &io[16 * k + 8]);
                }
            });
        };
        // the device reads 4-byte words: when every proof starts on a multiple of 4 inside the chunk, the chunk goes up in
        // pieces of whole proofs and the structure check of a piece, the upload of the one before and the kernels of the
        // one before that overlap
        bool aligned = use_fast;
        for (size_t i = c0; i < c1 && aligned; i++) aligned = (offsets[i] - offsets[c0]) % 4 == 0;
        bool fast_done = false;
        if (aligned) {
            std::vector<size_t> cuts{0};       // piece boundaries (proof indices inside the chunk)
            std::vector<size_t> byte_cuts{0};  // ... and as byte offsets
            size_t max_piece = 0;
            for (size_t i = 0; i < m;) {
                size_t j = i + 1;
                while (j < m && offsets[c0 + j + 1] - offsets[c0 + i] <= piece_bytes) j++;
                max_piece = std::max(max_piece, j - i);
                cuts.push_back(j);
                byte_cuts.push_back((size_t)(offsets[c0 + j] - offsets[c0]));
                i = j;
            }
            FastSession* ses = nullptr;
            int rc = fast_session_begin(ctx, chunk_base, byte_cuts.data(), cuts.size() - 1, m, max_piece, &ses);
            if (rc != STARK_OK) return rc;
            std::vector<uint64_t> bases(m);
            std::vector<size_t> who;
            who.reserve(m);
            std::vector<uint32_t> io(16 * m);
            // the helper thread's staging copies use half of the host threads
            const unsigned saved_threads = threads;
            threads = std::max(1u, threads / 2);
            try {  // nothing may unwind past the session's running upload thread
                for (size_t p = 0; p + 1 < cuts.size() && rc == STARK_OK; p++) {
                    const size_t i0 = cuts[p], i1 = cuts[p + 1], k0 = who.size();
                    double tc = now();
                    classify(i0, i1);
                    t_classify += now() - tc;
                    for (size_t i = i0; i < i1; i++)
                        if (canon[i]) {
                            bases[who.size()] = offsets[c0 + i] - offsets[c0];
                            who.push_back(i);
                        }
                    make_io(who, k0, who.size(), io);
                    rc = fast_session_piece(ctx, ses, bases.data() + k0, io.data() + 16 * k0, who.size() - k0, CP, d_mini);
                }
            } catch (...) {
                rc = STARK_ERR_NOMEM;
            }
            threads = saved_threads;
            std::vector<uint8_t> st(who.size());
            const int rc_end = fast_session_end(ctx, ses, rc == STARK_OK ? st.data() : nullptr);
            if (rc != STARK_OK) return rc;
            if (rc_end != STARK_OK) return rc_end;
            for (size_t k = 0; k < who.size(); k++) fast_status[who[k]] = st[k];
            fast_done = true;
            t_fast += now() - t0;
            t0 = now();
        } else if (use_fast) {
            classify(0, m);
        }
        t_classify += now() - t0;
        t0 = now();
        parallel_for(m, threads, [&](size_t i) {
            ProofWork& pw = work[i];
            if (canon[i]) {
                pw.status = ST_FAST;
                return;
            }
            const uint64_t a = offsets[c0 + i], b = offsets[c0 + i + 1];
            if (!parse_proof(proofs + a, (size_t)(b - a), pw.proof)) {
                pw.status = ST_PARSE;
                return;
            }
            prepare(pw, P);
        });
        t_parse += now() - t0;
        for (size_t i = 0; i < m; i++) n_parsed += !canon[i];
        t0 = now();
        // bucket every branch of the chunk by shape; roots are deduplicated per proof
        std::map<std::pair<uint32_t, uint32_t>, Bucket> buckets;
        std::vector<uint8_t> roots;
        std::vector<uint64_t> root_src;  // offset of each root's 32 bytes inside the chunk
        for (size_t i = 0; i < m; i++) {
            ProofWork& pw = work[i];
            if (pw.status != ST_ACCEPT) continue;
            for (const Query& q : pw.queries) {
                const uint32_t rid = (uint32_t)(roots.size() / 32);
                roots.insert(roots.end(), q.root, q.root + 32);
                root_src.push_back((uint64_t)(q.root - (proofs + offsets[c0])));
                for (size_t k = 0; k < q.branches->size(); k++) {
                    const BranchRef& b = (*q.branches)[k];
                    Bucket& bk = buckets[{b.vs, b.w}];
                    bk.vs = b.vs;
                    bk.w = b.w;
                    bk.br.push_back(&b);
                    bk.idx.push_back(q.indices[k]);
                    bk.root_id.push_back(rid);
                    bk.proof_id.push_back((uint32_t)i);
                }
            }
        }
        t_bucket += now() - t0;
        std::vector<std::atomic<uint8_t>> merkle_bad(m);
        for (auto& x : merkle_bad) x.store(0);
        // In-place path: upload the chunk's proof bytes once and let every lane read its branch record where
        // it lies; needs every field at a multiple of 4 bytes. Otherwise (or with STARKHIP_BATCH_MERKLE=flatten)
        // the branches are copied into structure-of-arrays buffers on the host first.
        bool in_place = !force_flatten;
        for (size_t i = c0; i < c1 && in_place; i++) in_place = (offsets[i] - offsets[c0]) % 4 == 0;
        for (auto& kv : buckets) in_place = in_place && kv.second.vs % 4 == 0;
        if (in_place && !buckets.empty()) {
            t0 = now();
            size_t total = 0;
            for (auto& kv : buckets) total += kv.second.br.size();
            std::vector<uint64_t> rec_off(total), root_off(total);
            std::vector<uint32_t> idx(total), pid(total);
            std::vector<size_t> starts;
            {
                size_t o = 0;
                for (auto& kv : buckets) {
                    starts.push_back(o);
                    o += kv.second.br.size();
                }
            }
            {
                size_t bi = 0;
                for (auto& kv : buckets) {
                    Bucket& bk = kv.second;
                    const size_t base = starts[bi++], n = bk.br.size();
                    parallel_for((n + 65535) / 65536, threads, [&](size_t blk) {
                        const size_t lo = blk * 65536, hi = std::min(n, lo + 65536);
                        for (size_t k = lo; k < hi; k++) {
                            rec_off[base + k] = (uint64_t)(bk.br[k]->value - chunk_base);
                            root_off[base + k] = (uint64_t)(&roots[32 * (size_t)bk.root_id[k]] - roots.data());  // replaced below
                            idx[base + k] = bk.idx[k];
                            pid[base + k] = bk.proof_id[k];
                        }
                    });
                }
            }
            // roots live in the proofs themselves: root ids -> offsets of those 32 bytes inside the chunk
            for (size_t k = 0; k < total; k++) root_off[k] = root_src[root_off[k] / 32];
            t_flatten += now() - t0;
            t0 = now();
            auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
            const size_t o_rec = up(chunk_len), o_root = o_rec + up(8 * total), o_idx = o_root + up(8 * total),
                         o_acc = o_idx + up(4 * total), bytes = o_acc + up(total);
            uint8_t* d = nullptr;
            if (hipMalloc((void**)&d, bytes) != hipSuccess) {
                return STARK_ERR_NOMEM;
            }
            hipStream_t st = (hipStream_t)stark_ctx_stream(ctx);
            std::vector<uint8_t> acc(total);
            hipError_t e = copy_h2d(ctx, d, chunk_base, chunk_len, st);
            if (e == hipSuccess) e = hipMemcpyAsync(d + o_rec, rec_off.data(), 8 * total, hipMemcpyHostToDevice, st);
            if (e == hipSuccess) e = hipMemcpyAsync(d + o_root, root_off.data(), 8 * total, hipMemcpyHostToDevice, st);
            if (e == hipSuccess) e = hipMemcpyAsync(d + o_idx, idx.data(), 4 * total, hipMemcpyHostToDevice, st);
            size_t bi = 0;
            for (auto& kv : buckets) {
                const size_t base = starts[bi++], n = kv.second.br.size();
                if (e == hipSuccess)
                    e = launch_merkle_gather(d, (const uint64_t*)(d + o_rec) + base, kv.second.vs, kv.second.w,
                                             (const uint32_t*)(d + o_idx) + base, (const uint64_t*)(d + o_root) + base, n,
                                             d + o_acc + base, st);
            }
            if (e == hipSuccess) e = hipMemcpyAsync(acc.data(), d + o_acc, total, hipMemcpyDeviceToHost, st);
            if (e == hipSuccess) e = hipStreamSynchronize(st);
            (void)hipFree(d);
            if (e != hipSuccess) {
                return STARK_ERR_HIP;
            }
            for (size_t k = 0; k < total; k++)
                if (!acc[k]) merkle_bad[pid[k]].store(1);
            t_merkle += now() - t0;
        } else
        for (auto& kv : buckets) {
            Bucket& bk = kv.second;
            const size_t n = bk.br.size();
            t0 = now();
            std::vector<uint8_t> values(n * bk.vs), sibs(n * bk.vs), wit((size_t)bk.w * n * 32), acc(n);
            parallel_for((n + 4095) / 4096, threads, [&](size_t blk) {
                const size_t lo = blk * 4096, hi = std::min(n, lo + 4096);
                for (size_t k = lo; k < hi; k++) {
                    const BranchRef& b = *bk.br[k];
                    memcpy(&values[k * bk.vs], b.value, bk.vs);
                    memcpy(&sibs[k * bk.vs], b.sibling, bk.vs);
                    for (uint32_t l = 0; l < bk.w; l++) memcpy(&wit[((size_t)l * n + k) * 32], b.wit + 32 * l, 32);
                }
            });
            t_flatten += now() - t0;
            t0 = now();
            int rc = stark_merkle_verify_batch(ctx, values.data(), sibs.data(), bk.vs, wit.data(), bk.w, bk.idx.data(),
                                               roots.data(), (uint32_t)(roots.size() / 32), bk.root_id.data(), n, acc.data(),
                                               nullptr);
            if (rc != STARK_OK) {
                return rc;
            }
            for (size_t k = 0; k < n; k++)
                if (!acc[k]) merkle_bad[bk.proof_id[k]].store(1);
            t_merkle += now() - t0;
        }
        t0 = now();
        for (size_t i = 0; i < m; i++)
            if (work[i].status == ST_ACCEPT && merkle_bad[i].load() && !skip_merkle) work[i].status = ST_MERKLE;
        if (field_on_host) {
            parallel_for(m, threads, [&](size_t i) {
                ProofWork& pw = work[i];
                if (pw.status == ST_ACCEPT) pw.status = field_checks(pw, P, c0 + i);
            });
        } else {
            int rc = device_field_checks(ctx, work, c0, P, CP, d_mini, threads);
            if (rc != STARK_OK) {
                return rc;
            }
        }
        t_field += now() - t0;
        t0 = now();
        if (fast_done) {
            for (size_t i = 0; i < m; i++)
                if (canon[i]) work[i].status = fast_status[i];
        } else {
            std::vector<uint64_t> bases;
            std::vector<size_t> who;
            for (size_t i = 0; i < m; i++)
                if (canon[i]) {
                    bases.push_back(offsets[c0 + i] - offsets[c0]);
                    who.push_back(i);
                }
            // proofs that do not start on a multiple of 4 inside the chunk (a neighbour's length was odd) are copied, with
            // the rest, into a packed buffer first
            const uint8_t* fast_chunk = chunk_base;
            size_t fast_len = chunk_len;
            std::vector<uint8_t> packed;
            if (!who.empty()) {
                const size_t psz = fast_proof_size();
                packed.resize(psz * who.size());
                parallel_for(who.size(), threads, [&](size_t k) { memcpy(&packed[psz * k], chunk_base + bases[k], psz); });
                for (size_t k = 0; k < who.size(); k++) bases[k] = psz * k;
                fast_chunk = packed.data();
                fast_len = packed.size();
                std::vector<uint32_t> io(16 * who.size());
                make_io(who, 0, who.size(), io);
                std::vector<uint8_t> st(who.size());
                int rc = fast_verify(ctx, fast_chunk, fast_len, bases.data(), io.data(), who.size(), CP, d_mini, st.data());
                if (rc != STARK_OK) {
                    return rc;
                }
                for (size_t k = 0; k < who.size(); k++) work[who[k]].status = st[k];
            }
        }
        t_fast += now() - t0;
        for (size_t i = 0; i < m; i++) {
            accept[c0 + i] = work[i].status == ST_ACCEPT;
            if (status) status[c0 + i] = work[i].status;
        }
        c0 = c1;
    }
    if (timing)
        fprintf(stderr, "stark_mimc_verify_batch: %zu proofs: structure check %.3f s, fast path (upload + kernels) %.3f s; general path (%zu proofs parsed on the host): parse+indices %.3f s, bucketing %.3f s, flatten %.3f s, merkle (copies+kernel) %.3f s, field checks %.3f s\n",
                n_proofs, t_classify, t_fast, n_parsed, t_parse, t_bucket, t_flatten, t_merkle, t_field);
    return STARK_OK;
}

extern "C" int stark_mimc_verify_batch(stark_ctx* ctx, const uint8_t* proofs, const uint64_t* offsets, size_t n_proofs,
                                       const uint32_t inp[8], uint64_t num_steps, const uint32_t* round_constants,
                                       size_t n_constants, const uint32_t output[8], uint8_t* accept, uint8_t* status) try {
    return verify_impl(ctx, proofs, offsets, n_proofs, inp, output, 0, num_steps, round_constants, n_constants, accept, status);
} STARK_CATCH

// Proofs already in device memory (what stark_mimc_prove_batch_dev leaves there): the structure check, the index
// sampling, the branches and the scalar checks all run where the bytes are; only proofs of another structure
// (which the reference can only reject or panic on, or whose `points` component has another length) are copied
// to the host and take the general path.
extern "C" int stark_mimc_verify_batch_dev(stark_ctx* ctx, const void* d_proofs, uint64_t proof_stride, size_t n_proofs,
                                           const void* d_inputs, uint64_t num_steps, const uint32_t* round_constants,
                                           size_t n_constants, const void* d_outputs, uint8_t* accept, uint8_t* status) try {
    if (!ctx) return STARK_ERR_INVALID;
    if (ctx_make_current(ctx) != hipSuccess) return STARK_ERR_HIP;
    if (n_proofs && (!d_proofs || !d_inputs || !d_outputs || !accept)) return STARK_ERR_INVALID;
    if (n_constants && !round_constants) return STARK_ERR_INVALID;
    if (n_proofs == 0) return STARK_OK;
    Setup S;
    uint8_t early;
    int rc = make_setup(ctx, num_steps, round_constants, n_constants, S, early);
    if (rc != STARK_OK) return rc;
    if (early != ST_ACCEPT) {
        for (size_t i = 0; i < n_proofs; i++) {
            accept[i] = 0;
            if (status) status[i] = early;
        }
        return STARK_OK;
    }
    hipStream_t st = (hipStream_t)stark_ctx_stream(ctx);
    const bool use_fast = !S.force_general && !S.field_on_host && !S.force_flatten;
    std::vector<uint8_t> canon(n_proofs, 0), stv(n_proofs, 0);
    // One pass (default): structure words and checks of every proof are queued together and answered in one copy, the
    // proofs of another structure are then redone on the general path. STARKHIP_BATCH_RESIDENT=twostep asks for the
    // structure first and runs the checks on the matching proofs only (one more round trip per call).
    const char* renv = getenv("STARKHIP_BATCH_RESIDENT");
    const bool one_pass = use_fast && !(renv && strcmp(renv, "twostep") == 0);
    if (one_pass) {
        uint32_t inv[8];
        hostfe::to_limbs(S.P.inv_last_m1, inv);
        for (size_t k0 = 0; k0 < n_proofs; k0 += 16384) {
            const size_t mc = std::min<size_t>(16384, n_proofs - k0);
            rc = fast_verify_resident_run(ctx, (const uint8_t*)d_proofs, proof_stride, k0, mc, (const uint32_t*)d_inputs,
                                          (const uint32_t*)d_outputs, inv, S.CP, S.d_mini, canon.data() + k0, stv.data() + k0);
            if (rc != STARK_OK) return rc;
        }
    } else if (use_fast && (rc = fast_classify_resident(ctx, (const uint8_t*)d_proofs, proof_stride, n_proofs, canon.data())) != STARK_OK) {
        return rc;
    }
    std::vector<uint64_t> bases;
    std::vector<uint32_t> who;
    std::vector<size_t> rest;
    for (size_t i = 0; i < n_proofs; i++) {
        if (one_pass) {
            if (!canon[i]) rest.push_back(i);
            continue;
        }
        if (canon[i]) {
            bases.push_back(proof_stride * i);
            who.push_back((uint32_t)i);
        } else {
            rest.push_back(i);
        }
    }
    // descriptors and lanes take about 83 KB per proof (kept by the context between calls): chunks of 2^14 proofs
    for (size_t k0 = 0; k0 < who.size(); k0 += 16384) {
        const size_t mc = std::min<size_t>(16384, who.size() - k0);
        auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
        const size_t o_who = up(8 * mc), o_io = o_who + up(4 * mc);
        uint8_t* d = nullptr;
        if ((rc = fast_resident_scratch(ctx, o_io + up(64 * mc), &d)) != STARK_OK) return rc;
        uint32_t inv[8];
        hostfe::to_limbs(S.P.inv_last_m1, inv);
        if (hipMemcpyAsync(d, bases.data() + k0, 8 * mc, hipMemcpyHostToDevice, st) != hipSuccess ||
            hipMemcpyAsync(d + o_who, who.data() + k0, 4 * mc, hipMemcpyHostToDevice, st) != hipSuccess)
            return STARK_ERR_HIP;
        rc = fast_make_io(ctx, (const uint32_t*)d_inputs, (const uint32_t*)d_outputs, (const uint32_t*)(d + o_who), mc, inv,
                          (uint32_t*)(d + o_io));
        std::vector<uint8_t> out(mc);
        if (rc == STARK_OK)
            rc = fast_verify_resident(ctx, (const uint8_t*)d_proofs, (const uint64_t*)d, (const uint32_t*)(d + o_io), mc, S.CP,
                                      S.d_mini, out.data());
        if (rc != STARK_OK) return rc;
        for (size_t k = 0; k < mc; k++) stv[who[k0 + k]] = out[k];
    }
    if (!rest.empty()) {
        std::vector<uint8_t> blob(rest.size() * proof_stride), a2(rest.size()), s2(rest.size());
        std::vector<uint32_t> in(8 * rest.size()), out(8 * rest.size());
        std::vector<uint64_t> offs(rest.size() + 1);
        for (size_t k = 0; k < rest.size(); k++) {
            offs[k] = proof_stride * k;
            if (hipMemcpyAsync(&blob[proof_stride * k], (const uint8_t*)d_proofs + proof_stride * rest[k], proof_stride,
                               hipMemcpyDeviceToHost, st) != hipSuccess ||
                hipMemcpyAsync(&in[8 * k], (const uint32_t*)d_inputs + 8 * rest[k], 32, hipMemcpyDeviceToHost, st) != hipSuccess ||
                hipMemcpyAsync(&out[8 * k], (const uint32_t*)d_outputs + 8 * rest[k], 32, hipMemcpyDeviceToHost, st) != hipSuccess)
                return STARK_ERR_HIP;
        }
        offs[rest.size()] = proof_stride * rest.size();
        if (hipStreamSynchronize(st) != hipSuccess) return STARK_ERR_HIP;
        rc = verify_impl(ctx, blob.data(), offs.data(), rest.size(), in.data(), out.data(), 8, num_steps, round_constants,
                         n_constants, a2.data(), s2.data());
        if (rc != STARK_OK) return rc;
        for (size_t k = 0; k < rest.size(); k++) stv[rest[k]] = s2[k];
    }
    for (size_t i = 0; i < n_proofs; i++) {
        accept[i] = stv[i] == ST_ACCEPT;
        if (status) status[i] = stv[i];
    }
    return STARK_OK;
} STARK_CATCH

extern "C" int stark_mimc_verify_batch_io(stark_ctx* ctx, const uint8_t* proofs, const uint64_t* offsets, size_t n_proofs,
                                          const uint32_t* inputs, uint64_t num_steps, const uint32_t* round_constants,
                                          size_t n_constants, const uint32_t* outputs, uint8_t* accept, uint8_t* status) try {
    return verify_impl(ctx, proofs, offsets, n_proofs, inputs, outputs, 8, num_steps, round_constants, n_constants, accept, status);
} STARK_CATCH

// Test support: stark_mimc_verify_batch_io with diagnostic switches (STARK_DIAG_*); nothing in the library calls it.
// The switch that makes the verifier overlook a Merkle mismatch exists only in the test build (-DSTARKHIP_DIAG,
// libstarkhip_diag.so): the shipped library refuses it.
extern "C" int stark_mimc_verify_batch_diag(stark_ctx* ctx, const uint8_t* proofs, const uint64_t* offsets, size_t n_proofs,
                                            const uint32_t* inputs, uint64_t num_steps, const uint32_t* round_constants,
                                            size_t n_constants, const uint32_t* outputs, unsigned flags, uint8_t* accept,
                                            uint8_t* status) try {
    if (!ctx) return STARK_ERR_INVALID;
    if (flags & ~(STARK_DIAG_IGNORE_MERKLE | STARK_DIAG_HOST_FIELD_CHECKS | STARK_DIAG_GENERAL_PATH)) {
        ctx_set_error(ctx, "stark_mimc_verify_batch_diag: unknown flag");
        return STARK_ERR_INVALID;
    }
#ifndef STARKHIP_DIAG
    if (flags & STARK_DIAG_IGNORE_MERKLE) {
        ctx_set_error(ctx, "STARK_DIAG_IGNORE_MERKLE is honoured by the test build of the library only (libstarkhip_diag.so)");
        return STARK_ERR_UNSUPPORTED;
    }
#endif
    return verify_impl(ctx, proofs, offsets, n_proofs, inputs, outputs, 8, num_steps, round_constants, n_constants, accept,
                       status, flags);
} STARK_CATCH
