// Batch driver, fast path: proofs whose structure words (branch counts, value sizes, witness counts, element
// tags) are those of an 8192-step proof are verified without being parsed on the host. Every proof the
// reference accepts has that structure (src/lib.rs:50,120 fix the instance; a branch of another depth or value
// size cannot reproduce the root, src/merkle_tree.rs:119-171), so each field sits at a fixed offset and the
// device can sample the indices (src/utils.rs:53-94), hash every branch where it lies, and pack the lanes of
// the scalar checks straight from the proof bytes. The decision per proof is what the general path of
// batch.hip makes: the same branch kernel and the same check kernels run, only their descriptors and lanes
// are written by kernels here instead of by host threads. Proofs of any other structure (including a `points`
// component of another length, which the reference reads and ignores) take the general path.
#include <functional>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <memory>
#include <thread>
#include <vector>

#include "../../include/starkhip.h"
#include "blake2s.cuh"
#include "fieldio.cuh"
#include "internal.hpp"

namespace starkhip {
namespace batch {

namespace {

constexpr uint32_t LOG_N = 16, NL = 5, FRI_SAMPLES = 40, SPOT = 80;
constexpr uint32_t N_SAMPLES = NL * FRI_SAMPLES + SPOT;      // per proof: 5 x 40 row indices, then 80 positions
constexpr uint32_t N_BRANCH = NL * 5 * FRI_SAMPLES + 3 * SPOT;  // 1,240
constexpr uint32_t N_FRI = NL * FRI_SAMPLES;

struct TBranch {
    uint32_t rec_off, root_off;  // offset of `value` / of the root inside the proof
    uint32_t samp, add, mask;    // query index = (samples[samp] + add) & mask
    uint32_t slot;               // where its descriptor goes: bucket_base * m + proof * bucket_count + slot_in_bucket
    uint32_t bucket_base, bucket_count;
};
struct TFri {
    uint32_t row_off[4], col_off, special_off, samp, layer;
};
struct TCon {
    uint32_t b1_off, pg_off, l_off, samp;
};
struct TBucket {
    uint32_t vs, w, base, count;
};

struct Template {
    size_t proof_size = 0;
    std::vector<std::pair<uint32_t, uint32_t>> words;  // (offset, expected u32): everything that fixes the structure
    std::vector<TBranch> branches;
    std::vector<TFri> fri;
    std::vector<TCon> con;
    std::vector<TBucket> buckets;
    uint32_t seed_off[NL + 1];  // root2 of layer k; the L root for the positions
    uint32_t merkle_root_off = 0;
};

// Walks the wire format (src/deserializer.rs:16-140) for the fixed instance.
Template build_template() {
    Template T;
    struct Raw {
        uint32_t rec_off, root_off, samp, add, mask, vs, w;
    };
    std::vector<Raw> raw;
    uint32_t off = 0;
    auto word = [&](uint32_t v) {
        T.words.push_back({off, v});
        off += 4;
    };
    auto branch = [&](uint32_t vs, uint32_t w, uint32_t root_off, uint32_t samp, uint32_t add, uint32_t mask) {
        word(vs);
        raw.push_back(Raw{off, root_off, samp, add, mask, vs, w});
        off += 2 * vs;
        word(32 * w);
        off += 32 * w;
        return raw.back().rec_off;
    };
    T.merkle_root_off = 0;
    const uint32_t l_root_off = 32;
    off = 64;
    uint32_t prev_root = l_root_off;
    for (uint32_t k = 0; k < NL; k++) {
        const uint32_t log_size = LOG_N - 2 * k, m = 1u << (log_size - 2);
        word(1);
        const uint32_t root2 = off;
        off += 32;
        T.seed_off[k] = root2;
        word(FRI_SAMPLES);
        uint32_t col_off[FRI_SAMPLES];
        for (uint32_t s = 0; s < FRI_SAMPLES; s++)
            col_off[s] = branch(32, log_size - 3, root2, FRI_SAMPLES * k + s, 0, m - 1);
        word(4 * FRI_SAMPLES);
        for (uint32_t s = 0; s < FRI_SAMPLES; s++) {
            TFri f;
            for (uint32_t j = 0; j < 4; j++)
                f.row_off[j] = branch(32, log_size - 1, prev_root, FRI_SAMPLES * k + s, j * m, (1u << log_size) - 1);
            f.col_off = col_off[s];
            f.special_off = prev_root;
            f.samp = FRI_SAMPLES * k + s;
            f.layer = k;
            T.fri.push_back(f);
        }
        prev_root = root2;
    }
    word(2);
    const uint32_t fin = (1u << LOG_N) >> (2 * NL);
    word(32 * fin);
    off += 32 * fin;
    T.seed_off[NL] = l_root_off;
    const uint32_t pos0 = NL * FRI_SAMPLES, full = (1u << LOG_N) - 1;
    word(2 * SPOT);
    std::vector<uint32_t> b1(SPOT), pg(SPOT);
    for (uint32_t s = 0; s < SPOT; s++) {
        b1[s] = branch(96, LOG_N - 1, T.merkle_root_off, pos0 + s, 0, full);
        pg[s] = branch(96, LOG_N - 1, T.merkle_root_off, pos0 + s, 8, full);
    }
    word(SPOT);
    for (uint32_t s = 0; s < SPOT; s++) {
        TCon c;
        c.b1_off = b1[s];
        c.pg_off = pg[s];
        c.l_off = branch(32, LOG_N - 1, l_root_off, pos0 + s, 0, full);
        c.samp = pos0 + s;
        T.con.push_back(c);
    }
    T.proof_size = off;
    // buckets by (value size, witnesses)
    for (const Raw& r : raw) {
        size_t b = 0;
        while (b < T.buckets.size() && !(T.buckets[b].vs == r.vs && T.buckets[b].w == r.w)) b++;
        if (b == T.buckets.size()) T.buckets.push_back(TBucket{r.vs, r.w, 0, 0});
        T.buckets[b].count++;
    }
    uint32_t base = 0;
    for (TBucket& b : T.buckets) {
        b.base = base;
        base += b.count;
    }
    std::vector<uint32_t> fill(T.buckets.size(), 0);
    for (const Raw& r : raw) {
        size_t b = 0;
        while (!(T.buckets[b].vs == r.vs && T.buckets[b].w == r.w)) b++;
        T.branches.push_back(TBranch{r.rec_off, r.root_off, r.samp, r.add, r.mask, fill[b]++, T.buckets[b].base, T.buckets[b].count});
    }
    return T;
}

const Template& tmpl() {
    static const Template T = build_template();
    return T;
}

struct TWord {
    uint32_t off, value;
};
struct DevTemplate : CtxAttachment {
    TBranch* branches = nullptr;
    TFri* fri = nullptr;
    TCon* con = nullptr;
    TWord* words = nullptr;
    uint32_t n_words = 0;
    // grow-only device buffers kept across calls: [0] descriptors, lanes and partial verdicts; [1] an uploaded chunk
    // with its per-proof offsets, inputs and verdicts; [2] verdicts of resident proofs; [3] offsets, positions and inputs
    // of resident proofs
    uint8_t* buf[4] = {nullptr, nullptr, nullptr, nullptr};
    size_t buf_bytes[4] = {0, 0, 0, 0};
    // pipelined host path: kernels of piece k run on `compute` (auxiliary stream 0 of the context) while piece k + 1 is
    // on the link (the context's main stream); `pinned` carries the per-proof offsets and inputs up and the verdicts
    // down without a blocking pageable copy
    hipStream_t compute = nullptr;
    uint8_t* pinned = nullptr;
    size_t pinned_bytes = 0;
    // the scalar checks of a call run on `side` (auxiliary stream 1 of the context) beside its branch launch (fork
    // after the samples, join before the verdicts): they are short, latency-bound launches that fill the gaps of the
    // long one. Both streams belong to the context, which drains and destroys them.
    hipStream_t side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    ~DevTemplate() override {
        if (ev_fork) (void)hipEventDestroy(ev_fork);
        if (ev_join) (void)hipEventDestroy(ev_join);
        if (pinned) (void)hipHostFree(pinned);
        (void)hipFree(branches);
        (void)hipFree(fri);
        (void)hipFree(con);
        (void)hipFree(words);
        for (uint8_t* b : buf) (void)hipFree(b);
    }
};
constexpr int FAST_SLOT = 1;

typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
__device__ __forceinline__ uint4 ld16(const uint8_t* p) {
    const u32x4_a4 v = *reinterpret_cast<const u32x4_a4*>(p);
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void copy32(uint8_t* dst16, const uint8_t* src4) {
    const uint4 a = ld16(src4), b = ld16(src4 + 16);
    reinterpret_cast<uint4*>(dst16)[0] = a;
    reinterpret_cast<uint4*>(dst16)[1] = b;
}

struct SeedOffs {
    uint32_t off[NL + 1];
};

// get_pseudorandom_indices for the six seeds of every proof; lane = (proof, set)
__global__ void __launch_bounds__(64)
vsample_kernel(const uint8_t* __restrict__ blob, const uint64_t* __restrict__ base, uint32_t m, SeedOffs so,
               uint32_t* __restrict__ samples) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= m * (NL + 1)) return;
    const uint32_t p = t / (NL + 1), set = t % (NL + 1);
    const uint8_t* seed = blob + base[p] + so.off[set];
    const uint4 s0 = ld16(seed), s1 = ld16(seed + 16);
    uint32_t h[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
    const uint32_t count = set < NL ? FRI_SAMPLES : SPOT;
    const uint32_t modulus = set < NL ? (1u << (LOG_N - 2 * (set + 1))) : (1u << LOG_N);
    const uint32_t real_modulus = modulus / 8 * 7;
    uint32_t* o = samples + (size_t)N_SAMPLES * p + (set < NL ? FRI_SAMPLES * set : NL * FRI_SAMPLES);
    for (uint32_t j = 0; j < count; j += 8) {
        if (j) {
            uint32_t msg[16];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                msg[k] = h[k];
                msg[8 + k] = 0;
            }
            b2s_init(h);
            b2s_compress(h, msg, 32u, true);
        }
        for (uint32_t k = 0; k < 8 && j + k < count; k++) {
            const uint32_t x = __builtin_bswap32(h[k]) % real_modulus;
            o[j + k] = 1 + x + x / 7;
        }
    }
}

// branch descriptors for merkle_gather_kernel; lane = (proof, branch)
__global__ void __launch_bounds__(256)
vdesc_kernel(const TBranch* __restrict__ tb, const uint64_t* __restrict__ base, uint32_t m, const uint32_t* __restrict__ samples,
             uint64_t* __restrict__ rec_off, uint64_t* __restrict__ root_off, uint32_t* __restrict__ idx) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x, p = blockIdx.y;
    if (b >= N_BRANCH) return;
    const TBranch t = tb[b];
    const size_t k = (size_t)t.bucket_base * m + (size_t)p * t.bucket_count + t.slot;
    rec_off[k] = base[p] + t.rec_off;
    root_off[k] = base[p] + t.root_off;
    idx[k] = (samples[(size_t)N_SAMPLES * p + t.samp] + t.add) & t.mask;
}

// lanes of the low-degree check; lane = (proof, layer*40 + sample)
__global__ void __launch_bounds__(256)
vpack_fri_kernel(const TFri* __restrict__ tf, const uint8_t* __restrict__ blob, const uint64_t* __restrict__ base, uint32_t m,
                 const uint32_t* __restrict__ samples, FriLane* __restrict__ lanes) {
    const uint32_t f = blockIdx.x * blockDim.x + threadIdx.x, p = blockIdx.y;
    if (f >= N_FRI) return;
    const TFri t = tf[f];
    const uint8_t* pr = blob + base[p];
    FriLane& L = lanes[(size_t)p * N_FRI + f];
#pragma unroll
    for (int j = 0; j < 4; j++) copy32(L.rows[j], pr + t.row_off[j]);
    copy32(L.col, pr + t.col_off);
    copy32(L.special, pr + t.special_off);
    L.y = samples[(size_t)N_SAMPLES * p + t.samp];
    L.layer = t.layer;
    L.pad[0] = L.pad[1] = 0;
}

// lanes of the constraint checks; lane = (proof, position)
__global__ void __launch_bounds__(128)
vpack_con_kernel(const TCon* __restrict__ tc, const uint8_t* __restrict__ blob, const uint64_t* __restrict__ base, uint32_t m,
                 const uint32_t* __restrict__ samples, ConstraintLane* __restrict__ lanes) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x, p = blockIdx.y;
    if (s >= SPOT) return;
    const TCon t = tc[s];
    const uint8_t* pr = blob + base[p];
    ConstraintLane& L = lanes[(size_t)p * SPOT + s];
    copy32(L.b1, pr + t.b1_off);
    copy32(L.b1 + 32, pr + t.b1_off + 32);
    copy32(L.b1 + 64, pr + t.b1_off + 64);
    copy32(L.pg, pr + t.pg_off);
    copy32(L.l, pr + t.l_off);
    L.pos = samples[(size_t)N_SAMPLES * p + t.samp];
    L.proof = p;
    L.pad[0] = L.pad[1] = 0;
}

// per proof: BLAKE2s(merkle_root || j), j = 1..4 (src/lib.rs:131-146), then the instance's input and slope; lane = (proof, j)
__global__ void __launch_bounds__(64)
vkk_kernel(const uint8_t* __restrict__ blob, const uint64_t* __restrict__ base, uint32_t m, uint32_t merkle_root_off,
           const uint32_t* __restrict__ io, uint8_t* __restrict__ kk) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 4 * m) return;
    const uint32_t p = t >> 2, j = t & 3;
    const uint8_t* root = blob + base[p] + merkle_root_off;
    const uint4 r0 = ld16(root), r1 = ld16(root + 16);
    uint32_t msg[16] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w, j + 1, 0, 0, 0, 0, 0, 0, 0};
    uint32_t h[8];
    b2s_init(h);
    b2s_compress(h, msg, 33u, true);
    uint4* o = reinterpret_cast<uint4*>(kk + CHECK_PROOF_BYTES * (size_t)p + 32 * j);
    o[0] = make_uint4(h[0], h[1], h[2], h[3]);
    o[1] = make_uint4(h[4], h[5], h[6], h[7]);
    // 64 bytes of input / slope, 16 per lane
    reinterpret_cast<uint4*>(kk + CHECK_PROOF_BYTES * (size_t)p + 128)[j] = reinterpret_cast<const uint4*>(io + 16 * (size_t)p)[j];
}

// structure words of device-resident proofs against the fixed instance's; a workgroup per proof
__global__ void __launch_bounds__(256)
vclassify_kernel(const uint8_t* __restrict__ proofs, uint64_t stride, const TWord* __restrict__ words, uint32_t n_words,
                 uint8_t* __restrict__ canonical) {
    __shared__ uint32_t bad;
    if (threadIdx.x == 0) bad = 0;
    __syncthreads();
    const uint8_t* pr = proofs + stride * blockIdx.x;
    uint32_t mine = 0;
    for (uint32_t k = threadIdx.x; k < n_words; k += blockDim.x)
        mine |= *reinterpret_cast<const uint32_t*>(pr + words[k].off) ^ words[k].value;
    if (mine) atomicOr(&bad, 1u);
    __syncthreads();
    if (threadIdx.x == 0) canonical[blockIdx.x] = bad ? 0 : 1;
}

// per selected proof: canonical input and the slope (output - input) / (last_step - 1); lane = selected proof
struct InvArg {
    uint32_t l[8];
};
__global__ void __launch_bounds__(64)
vio_kernel(const uint32_t* __restrict__ inputs, const uint32_t* __restrict__ outputs, const uint32_t* __restrict__ who, uint32_t m,
           InvArg inv_last_m1, uint32_t* __restrict__ io, uint64_t first, uint64_t stride, uint64_t* __restrict__ base_out) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= m) return;
    // selected proof k is proof who[k], or proof first + k of a run (which then also gets its byte offset written)
    const size_t i = who ? (size_t)who[k] : (size_t)(first + k);
    if (base_out) base_out[k] = stride * i;
    const F29 in = fio::norm(fio::load_limbs(inputs + 8 * i)), out = fio::norm(fio::load_limbs(outputs + 8 * i));
    Fe c;
#pragma unroll
    for (int j = 0; j < 8; j++) c.l[j] = inv_last_m1.l[j];
    fio::store_limbs(io + 16 * (size_t)k, in);
    fio::store_limbs(io + 16 * (size_t)k + 8, fio::mul_t(f29_sub(out, in), f29_from_fe(c)));
}

struct Buckets {
    uint32_t n;
    uint32_t base[8], count[8];
};

// the verdict; one wavefront per proof. 4 a branch's root differs; else the first failing scalar check (FRI lanes in
// order, then the constraint lanes); else 0
__global__ void __launch_bounds__(64)
vstatus_kernel(uint32_t m, Buckets B, const uint8_t* __restrict__ acc, const uint8_t* __restrict__ check, uint8_t* __restrict__ status) {
    const uint32_t p = blockIdx.x, lane = threadIdx.x;
    if (p >= m) return;
    uint32_t bad = 0;
    for (uint32_t b = 0; b < B.n; b++) {
        const uint8_t* a = acc + (size_t)B.base[b] * m + (size_t)p * B.count[b];
        for (uint32_t k = lane; k < B.count[b]; k += 64) bad |= a[k] ^ 1u;
    }
    // FRI lanes of all proofs come first, then the constraint lanes; key = (index, code) of a lane's first failure
    const uint8_t* f = check + (size_t)p * N_FRI;
    const uint8_t* c = check + (size_t)m * N_FRI + (size_t)p * SPOT;
    uint32_t key = 0xFFFFFFFFu;
    for (uint32_t k = lane; k < N_FRI + SPOT; k += 64) {
        const uint32_t v = k < N_FRI ? f[k] : c[k - N_FRI];
        if (v && key == 0xFFFFFFFFu) key = (k << 8) | v;
    }
    for (int off = 32; off > 0; off >>= 1) {
        const uint32_t other = (uint32_t)__shfl_xor((int)key, off, 64);
        key = other < key ? other : key;
    }
    const bool any_bad = __ballot(bad != 0) != 0;
    if (lane == 0) status[p] = any_bad ? 4 : (key == 0xFFFFFFFFu ? 0 : (uint8_t)(key & 0xFFu));
}

}  // namespace

size_t fast_proof_size() { return tmpl().proof_size; }

// true when every structure word of the proof is the fixed instance's (trailing bytes are tolerated, as the
// reference's parser tolerates them)
bool fast_is_canonical(const uint8_t* proof, size_t len) {
    const Template& T = tmpl();
    if (len < T.proof_size) return false;
    for (const auto& w : T.words) {
        uint32_t v;
        memcpy(&v, proof + w.first, 4);
        if (v != w.second) return false;
    }
    return true;
}

#define F_TRY(call)                                                                 \
    do {                                                                            \
        hipError_t e_ = (call);                                                     \
        if (e_ != hipSuccess) {                                                     \
            char b_[256];                                                           \
            snprintf(b_, sizeof b_, "%s failed: %s", #call, hipGetErrorString(e_)); \
            ctx_set_error(ctx, b_);                                                 \
            return e_ == hipErrorOutOfMemory ? STARK_ERR_NOMEM : STARK_ERR_HIP;     \
        }                                                                           \
    } while (0)

static int get_dev_template(stark_ctx* ctx, DevTemplate** out) {
    const Template& T = tmpl();
    DevTemplate* D = static_cast<DevTemplate*>(ctx_get_attachment(ctx, FAST_SLOT));
    if (!D) {
        std::unique_ptr<DevTemplate> d(new DevTemplate);
        std::vector<TWord> words;
        for (const auto& w : T.words) words.push_back(TWord{w.first, w.second});
        d->n_words = (uint32_t)words.size();
        F_TRY(hipMalloc((void**)&d->branches, T.branches.size() * sizeof(TBranch)));
        F_TRY(hipMalloc((void**)&d->fri, T.fri.size() * sizeof(TFri)));
        F_TRY(hipMalloc((void**)&d->con, T.con.size() * sizeof(TCon)));
        F_TRY(hipMalloc((void**)&d->words, words.size() * sizeof(TWord)));
        F_TRY(hipMemcpy(d->branches, T.branches.data(), T.branches.size() * sizeof(TBranch), hipMemcpyHostToDevice));
        F_TRY(hipMemcpy(d->fri, T.fri.data(), T.fri.size() * sizeof(TFri), hipMemcpyHostToDevice));
        F_TRY(hipMemcpy(d->con, T.con.data(), T.con.size() * sizeof(TCon), hipMemcpyHostToDevice));
        F_TRY(hipMemcpy(d->words, words.data(), words.size() * sizeof(TWord), hipMemcpyHostToDevice));
        D = d.get();
        ctx_set_attachment(ctx, FAST_SLOT, d.release());
    }
    *out = D;
    return STARK_OK;
}

// the kept buffer `which`, at least `need` bytes (the stream is drained before a smaller one is released)
static int kept_buffer(stark_ctx* ctx, DevTemplate* D, int which, size_t need, uint8_t** out) {
    if (D->buf_bytes[which] < need) {
        F_TRY(hipStreamSynchronize((hipStream_t)stark_ctx_stream(ctx)));
        if (D->compute) F_TRY(hipStreamSynchronize(D->compute));
        if (D->buf[which]) F_TRY(hipFree(D->buf[which]));
        D->buf[which] = nullptr;
        D->buf_bytes[which] = 0;
        F_TRY(hipMalloc((void**)&D->buf[which], need));
        D->buf_bytes[which] = need;
    }
    *out = D->buf[which];
    return STARK_OK;
}

// Everything resident: d_blob holds the proofs, d_base[i] the offset of canonical proof i inside it (a multiple of 4),
// d_io per proof 8 + 8 canonical limbs (input, slope). status_out (host) as the general path would set it.
// where the descriptors, lanes and partial verdicts of m proofs sit inside kept buffer 0
struct LaneLayout {
    size_t o_samp, o_rec, o_root, o_idx, o_acc, o_fri, o_con, o_kk, o_chk, total;
};
static LaneLayout lane_layout(size_t m) {
    LaneLayout L;
    size_t o = 0;
    auto take = [&](size_t bytes) {
        const size_t at = o;
        o += (bytes + 255) & ~(size_t)255;
        return at;
    };
    const size_t nb = (size_t)N_BRANCH * m;
    L.o_samp = take(4 * (size_t)N_SAMPLES * m);
    L.o_rec = take(8 * nb);
    L.o_root = take(8 * nb);
    L.o_idx = take(4 * nb);
    L.o_acc = take(nb);
    L.o_fri = take(sizeof(FriLane) * (size_t)N_FRI * m);
    L.o_con = take(sizeof(ConstraintLane) * (size_t)SPOT * m);
    L.o_kk = take(CHECK_PROOF_BYTES * (size_t)m);
    L.o_chk = take((size_t)(N_FRI + SPOT) * m);
    L.total = o;
    return L;
}
static size_t fast_lane_bytes(size_t m) { return lane_layout(m).total; }

// The launches alone, on stream `st`: one verdict byte per proof to d_status (device memory). The lanes and descriptors
// live in kept buffer 0, which consecutive calls on one stream may share.
// `beside` (optional): launches that depend on nothing but the caller's inputs; they are queued where the scalar checks
// go, off the path that leads to the branch launch.
static int fast_launch(stark_ctx* ctx, DevTemplate* D, const uint8_t* d_blob, const uint64_t* d_base, const uint32_t* d_io,
                       size_t mc, const CheckParams& CP, const uint32_t* d_mini, hipStream_t st, uint8_t* d_status,
                       const std::function<void(hipStream_t)>* beside = nullptr) {
    const Template& T = tmpl();
    int rc;
    const uint32_t m = (uint32_t)mc;
    const LaneLayout L = lane_layout(m);
    const size_t o_samp = L.o_samp, o_rec = L.o_rec, o_root = L.o_root, o_idx = L.o_idx, o_acc = L.o_acc, o_fri = L.o_fri,
                 o_con = L.o_con, o_kk = L.o_kk, o_chk = L.o_chk;
    uint8_t* d = nullptr;
    if ((rc = kept_buffer(ctx, D, 0, L.total, &d)) != STARK_OK) return rc;
    uint32_t* samples = (uint32_t*)(d + o_samp);
    SeedOffs so;
    memcpy(so.off, T.seed_off, sizeof so.off);
    hipLaunchKernelGGL(vsample_kernel, dim3((m * (NL + 1) + 63) / 64), dim3(64), 0, st, d_blob, d_base, m, so, samples);
    F_TRY(hipGetLastError());
    // fork: the scalar checks need the samples (and the caller's inputs, which are ordered before this point on st).
    // Which side is queued first is the host's choice, and the host is on the critical path: every launch call takes
    // it 5-8 us, so whatever is queued behind the other side's eight calls starts 50 us late. The branch launch is
    // the long pole (0.6 of 0.8 ms for 1,250 proofs), so the descriptors and the branch launch go out first, and the
    // checks, short launches that do not fill the chip, find room beside it because its residency is capped (see
    // launch_merkle_gather_groups; uncapped they waited for its workgroups to drain: five times their own duration,
    // which is why they used to go first). STARKHIP_BATCH_ORDER=checks restores that order.
    const bool overlap = !(getenv("STARKHIP_BATCH_OVERLAP") && atoi(getenv("STARKHIP_BATCH_OVERLAP")) == 0);
    const bool checks_first = getenv("STARKHIP_BATCH_ORDER") && !strcmp(getenv("STARKHIP_BATCH_ORDER"), "checks");
    hipStream_t sc = st;
    if (overlap) {
        // A plain stream of the context (ctx_aux_stream gives the rules). The fork event is recorded on st before side
        // is told to wait for it, the join event on side before st is; the call before this one on st ended with its
        // own join, so side is idle when the fork lands and one pair of events serves every call. (An earlier version
        // gave these launches a high-priority stream, after which calls of 1,250 proofs made no progress. No trace of
        // a stuck call was kept, so the cause is not established; the one thing that differed was this wait, a
        // high-priority queue waiting for an upload and a sampling kernel queued at normal priority.)
        if (!D->side) F_TRY(ctx_aux_stream(ctx, 1, &D->side));
        if (!D->ev_fork) F_TRY(hipEventCreateWithFlags(&D->ev_fork, hipEventDisableTiming));
        if (!D->ev_join) F_TRY(hipEventCreateWithFlags(&D->ev_join, hipEventDisableTiming));
        F_TRY(hipEventRecord(D->ev_fork, st));
        sc = D->side;
    }
    Buckets B = {};
    B.n = (uint32_t)T.buckets.size();
    if (B.n > 8) return STARK_ERR_UNSUPPORTED;
    auto queue_checks = [&]() -> int {
        if (sc != st) F_TRY(hipStreamWaitEvent(sc, D->ev_fork, 0));
        if (beside) (*beside)(sc);
        hipLaunchKernelGGL(vpack_fri_kernel, dim3((N_FRI + 255) / 256, m), dim3(256), 0, sc, D->fri, d_blob, d_base, m, samples,
                           (FriLane*)(d + o_fri));
        hipLaunchKernelGGL(vpack_con_kernel, dim3((SPOT + 127) / 128, m), dim3(128), 0, sc, D->con, d_blob, d_base, m, samples,
                           (ConstraintLane*)(d + o_con));
        hipLaunchKernelGGL(vkk_kernel, dim3((4 * m + 63) / 64), dim3(64), 0, sc, d_blob, d_base, m, T.merkle_root_off, d_io, d + o_kk);
        F_TRY(hipGetLastError());
        F_TRY(launch_fri_check((const FriLane*)(d + o_fri), (size_t)N_FRI * m, CP, d + o_chk, sc));
        F_TRY(launch_constraint_check((const ConstraintLane*)(d + o_con), (size_t)SPOT * m, CP, d + o_kk, d_mini,
                                      d + o_chk + (size_t)N_FRI * m, sc));
        if (sc != st) F_TRY(hipEventRecord(D->ev_join, sc));
        return STARK_OK;
    };
    auto queue_branches = [&]() -> int {
        hipLaunchKernelGGL(vdesc_kernel, dim3((N_BRANCH + 255) / 256, m), dim3(256), 0, st, D->branches, d_base, m, samples,
                           (uint64_t*)(d + o_rec), (uint64_t*)(d + o_root), (uint32_t*)(d + o_idx));
        F_TRY(hipGetLastError());
        // all buckets of branches in one launch, deepest first (a launch per bucket would pay a tail each: the last
        // wavefronts of a launch run their chain of compressions on an emptying chip); STARKHIP_BATCH_GATHER=split keeps
        // the launch per bucket
        const bool one_launch = !(getenv("STARKHIP_BATCH_GATHER") && !strcmp(getenv("STARKHIP_BATCH_GATHER"), "split"));
        uint32_t order[8];
        for (uint32_t b = 0; b < B.n; b++) order[b] = b;
        std::stable_sort(order, order + B.n, [&](uint32_t x, uint32_t y) { return T.buckets[x].w > T.buckets[y].w; });
        GatherGroups G = {};
        G.n = B.n;
        for (uint32_t b = 0; b < B.n; b++) {
            const TBucket& k = T.buckets[b];
            B.base[b] = k.base;
            B.count[b] = k.count;
            const size_t first = (size_t)k.base * m, n = (size_t)k.count * m;
            if ((first + n) >> 32) return STARK_ERR_UNSUPPORTED;
            if (!one_launch)
                F_TRY(launch_merkle_gather(d_blob, (const uint64_t*)(d + o_rec) + first, k.vs, k.w, (const uint32_t*)(d + o_idx) + first,
                                           (const uint64_t*)(d + o_root) + first, n, d + o_acc + first, st));
        }
        for (uint32_t g = 0; g < B.n; g++) {
            const TBucket& k = T.buckets[order[g]];
            G.first[g] = (uint32_t)((size_t)k.base * m);
            G.count[g] = (uint32_t)((size_t)k.count * m);
            G.value_size[g] = k.vs;
            G.num_witnesses[g] = k.w;
        }
        if (one_launch)
            F_TRY(launch_merkle_gather_groups(d_blob, (const uint64_t*)(d + o_rec), (const uint32_t*)(d + o_idx),
                                              (const uint64_t*)(d + o_root), d + o_acc, G, st));
        return STARK_OK;
    };
    if (checks_first) {
        if ((rc = queue_checks()) != STARK_OK) return rc;
        if ((rc = queue_branches()) != STARK_OK) return rc;
    } else {
        if ((rc = queue_branches()) != STARK_OK) return rc;
        if ((rc = queue_checks()) != STARK_OK) return rc;
    }
    if (sc != st) F_TRY(hipStreamWaitEvent(st, D->ev_join, 0));
    hipLaunchKernelGGL(vstatus_kernel, dim3(m), dim3(64), 0, st, m, B, d + o_acc, d + o_chk, d_status);
    F_TRY(hipGetLastError());
    return STARK_OK;
}

int fast_verify_resident(stark_ctx* ctx, const uint8_t* d_blob, const uint64_t* d_base, const uint32_t* d_io, size_t mc,
                         const CheckParams& CP, const uint32_t* d_mini, uint8_t* status_out) {
    if (mc == 0) return STARK_OK;
    hipStream_t st = (hipStream_t)stark_ctx_stream(ctx);
    DevTemplate* D;
    int rc = get_dev_template(ctx, &D);
    if (rc != STARK_OK) return rc;
    uint8_t* d_status = nullptr;
    if ((rc = kept_buffer(ctx, D, 2, (mc + 255) & ~(size_t)255, &d_status)) != STARK_OK) return rc;
    if ((rc = fast_launch(ctx, D, d_blob, d_base, d_io, mc, CP, d_mini, st, d_status)) != STARK_OK) return rc;
    F_TRY(hipMemcpyAsync(status_out, d_status, mc, hipMemcpyDeviceToHost, st));
    F_TRY(hipStreamSynchronize(st));
    return STARK_OK;
}

// ---- pipelined host path --------------------------------------------------------------------------------------
// A chunk of proofs in host memory goes up in pieces of whole proofs. A helper thread does nothing but stage and
// queue the pieces on the context's stream, back to back (the upload needs nothing from the structure check);
// the calling thread meanwhile classifies piece after piece and queues each piece's kernels on a second stream
// behind that piece's upload event, so the link never waits for the host's checks or for the kernels.
// begin: buffers for the whole chunk, start the uploads (cuts[p] .. cuts[p + 1] = byte range of piece p);
// piece: queue the checks of the `count` canonical proofs of the next piece (bases: their offsets inside the chunk,
// io: 16 limbs each); end: wait and hand the verdicts back in the order the proofs were given.
struct FastSession {
    DevTemplate* D = nullptr;
    uint8_t* d = nullptr;  // kept buffer 1: chunk bytes, then offsets, inputs and verdicts of every proof
    size_t o_base = 0, o_io = 0, o_st = 0, cap = 0, done = 0, next_piece = 0;
    std::vector<hipEvent_t> events;       // events[p]: piece p is on the device
    std::vector<size_t> cuts;
    std::atomic<size_t> queued{0};        // pieces whose upload has been queued (and its event recorded)
    std::atomic<int> upload_error{0};     // hipError_t of the helper thread
    std::thread uploader;
};

int fast_session_begin(stark_ctx* ctx, const uint8_t* chunk, const size_t* cuts, size_t n_pieces, size_t max_proofs,
                       size_t max_piece_proofs, FastSession** out) {
    auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
    std::unique_ptr<FastSession> S(new FastSession);
    int rc = get_dev_template(ctx, &S->D);
    if (rc != STARK_OK) return rc;
    DevTemplate* D = S->D;
    if (!D->compute) F_TRY(ctx_aux_stream(ctx, 0, &D->compute));
    const size_t pin_need = up(8 * max_proofs) + up(64 * max_proofs) + up(max_proofs);
    if (D->pinned_bytes < pin_need) {
        F_TRY(hipStreamSynchronize((hipStream_t)stark_ctx_stream(ctx)));
        F_TRY(hipStreamSynchronize(D->compute));
        if (D->pinned) F_TRY(hipHostFree(D->pinned));
        D->pinned = nullptr;
        D->pinned_bytes = 0;
        F_TRY(hipHostMalloc((void**)&D->pinned, pin_need, hipHostMallocDefault));
        D->pinned_bytes = pin_need;
    }
    const size_t chunk_len = cuts[n_pieces];
    S->o_base = up(chunk_len + 16);
    S->o_io = S->o_base + up(8 * max_proofs);
    S->o_st = S->o_io + up(64 * max_proofs);
    S->cap = max_proofs;
    S->cuts.assign(cuts, cuts + n_pieces + 1);
    if ((rc = kept_buffer(ctx, D, 1, S->o_st + up(max_proofs), &S->d)) != STARK_OK) return rc;
    // size the lane buffer for the largest piece now: growing it later would free memory that queued kernels use
    uint8_t* lanes;
    if ((rc = kept_buffer(ctx, D, 0, fast_lane_bytes(max_piece_proofs), &lanes)) != STARK_OK) return rc;
    S->events.resize(n_pieces, nullptr);
    for (size_t p = 0; p < n_pieces; p++) {
        hipError_t e = hipEventCreateWithFlags(&S->events[p], hipEventDisableTiming);
        if (e != hipSuccess) {
            for (size_t q = 0; q < p; q++) (void)hipEventDestroy(S->events[q]);
            F_TRY(e);
        }
    }
    FastSession* s = S.get();
    s->uploader = std::thread([ctx, s, chunk] {
        hipStream_t link = (hipStream_t)stark_ctx_stream(ctx);
        hipError_t e = ctx_make_current(ctx);  // the current device is per thread
        for (size_t p = 0; p + 1 < s->cuts.size(); p++) {
            const size_t lo = s->cuts[p], hi = s->cuts[p + 1];
            if (e == hipSuccess && hi > lo) e = copy_h2d(ctx, s->d + lo, chunk + lo, hi - lo, link);
            if (e == hipSuccess) e = hipEventRecord(s->events[p], link);
            if (e != hipSuccess) s->upload_error.store((int)e);
            s->queued.store(p + 1, std::memory_order_release);  // also after an error: nobody may wait for ever
        }
    });
    *out = S.release();
    return STARK_OK;
}

int fast_session_piece(stark_ctx* ctx, FastSession* S, const uint64_t* bases, const uint32_t* io, size_t count,
                       const CheckParams& CP, const uint32_t* d_mini) {
    auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
    DevTemplate* D = S->D;
    const size_t p = S->next_piece++;
    if (p + 1 >= S->cuts.size() || S->done + count > S->cap) return STARK_ERR_INVALID;
    if (count == 0) return STARK_OK;
    uint8_t* pin_base = D->pinned + 8 * S->done;
    uint8_t* pin_io = D->pinned + up(8 * S->cap) + 64 * S->done;
    memcpy(pin_base, bases, 8 * count);
    memcpy(pin_io, io, 64 * count);
    F_TRY(hipMemcpyAsync(S->d + S->o_base + 8 * S->done, pin_base, 8 * count, hipMemcpyHostToDevice, D->compute));
    F_TRY(hipMemcpyAsync(S->d + S->o_io + 64 * S->done, pin_io, 64 * count, hipMemcpyHostToDevice, D->compute));
    while (S->queued.load(std::memory_order_acquire) <= p) std::this_thread::yield();
    if (S->upload_error.load()) {
        ctx_set_error(ctx, "staged upload of a piece of the chunk failed");
        return STARK_ERR_HIP;
    }
    F_TRY(hipStreamWaitEvent(D->compute, S->events[p], 0));
    int rc = fast_launch(ctx, D, S->d, (const uint64_t*)(S->d + S->o_base) + S->done, (const uint32_t*)(S->d + S->o_io) + 16 * S->done,
                         count, CP, d_mini, D->compute, S->d + S->o_st + S->done);
    if (rc != STARK_OK) return rc;
    S->done += count;
    return STARK_OK;
}

// Also the way out after an error: waits for the uploads and for everything queued, then releases the session.
// status_out may be null then.
int fast_session_end(stark_ctx* ctx, FastSession* S, uint8_t* status_out) {
    auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
    std::unique_ptr<FastSession> guard(S);
    DevTemplate* D = S->D;
    if (S->uploader.joinable()) S->uploader.join();
    hipError_t e = (hipError_t)S->upload_error.load();
    if (e == hipSuccess && status_out && S->done) {
        uint8_t* pin_st = D->pinned + up(8 * S->cap) + up(64 * S->cap);
        e = hipMemcpyAsync(pin_st, S->d + S->o_st, S->done, hipMemcpyDeviceToHost, D->compute);
        if (e == hipSuccess) e = hipStreamSynchronize(D->compute);
        if (e == hipSuccess) memcpy(status_out, pin_st, S->done);
    } else {
        hipError_t e1 = hipStreamSynchronize(D->compute);
        if (e == hipSuccess) e = e1;
    }
    hipError_t e2 = hipStreamSynchronize((hipStream_t)stark_ctx_stream(ctx));
    for (hipEvent_t ev : S->events) (void)hipEventDestroy(ev);
    F_TRY(e);
    F_TRY(e2);
    return STARK_OK;
}

// chunk: the proof bytes of a chunk (host); bases[i]: offset of canonical proof i inside it (multiple of 4);
// io: per proof 8 + 8 canonical limbs (input, slope). status_out[i] as the general path would set it.
int fast_verify(stark_ctx* ctx, const uint8_t* chunk, size_t chunk_len, const uint64_t* bases, const uint32_t* io, size_t mc,
                const CheckParams& CP, const uint32_t* d_mini, uint8_t* status_out) {
    if (mc == 0) return STARK_OK;
    hipStream_t st = (hipStream_t)stark_ctx_stream(ctx);
    auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t o_base = up(chunk_len + 16), o_io = o_base + up(8 * mc);
    DevTemplate* D;
    int rc = get_dev_template(ctx, &D);
    if (rc != STARK_OK) return rc;
    uint8_t* d = nullptr;
    if ((rc = kept_buffer(ctx, D, 1, o_io + up(64 * mc), &d)) != STARK_OK) return rc;
    F_TRY(copy_h2d(ctx, d, chunk, chunk_len, st));
    F_TRY(hipMemcpyAsync(d + o_base, bases, 8 * mc, hipMemcpyHostToDevice, st));
    F_TRY(hipMemcpyAsync(d + o_io, io, 64 * mc, hipMemcpyHostToDevice, st));
    return fast_verify_resident(ctx, d, (const uint64_t*)(d + o_base), (const uint32_t*)(d + o_io), mc, CP, d_mini, status_out);
}

// Kept buffer 3, at least `bytes` long: the per-call device scratch of the resident entry point (no allocation and no
// release, which waits for the device, once it has grown to the largest call).
int fast_resident_scratch(stark_ctx* ctx, size_t bytes, uint8_t** out) {
    DevTemplate* D;
    int rc = get_dev_template(ctx, &D);
    if (rc != STARK_OK) return rc;
    return kept_buffer(ctx, D, 3, bytes, out);
}

// Device-resident proofs at d_proofs + i * stride: canonical_out[i] (host) = 1 where the structure words match.
int fast_classify_resident(stark_ctx* ctx, const uint8_t* d_proofs, uint64_t stride, size_t n, uint8_t* canonical_out) {
    if (n == 0) return STARK_OK;
    if (stride < tmpl().proof_size || stride % 4 != 0) {
        memset(canonical_out, 0, n);
        return STARK_OK;
    }
    hipStream_t st = (hipStream_t)stark_ctx_stream(ctx);
    DevTemplate* D;
    int rc = get_dev_template(ctx, &D);
    if (rc != STARK_OK) return rc;
    uint8_t* d = nullptr;  // kept buffer 2 (the verdicts of the checks that follow reuse it: this call ends synchronised)
    if ((rc = kept_buffer(ctx, D, 2, (n + 255) & ~(size_t)255, &d)) != STARK_OK) return rc;
    hipLaunchKernelGGL(vclassify_kernel, dim3((unsigned)n), dim3(256), 0, st, d_proofs, stride, D->words, D->n_words, d);
    F_TRY(hipGetLastError());
    F_TRY(hipMemcpyAsync(canonical_out, d, n, hipMemcpyDeviceToHost, st));
    F_TRY(hipStreamSynchronize(st));
    return STARK_OK;
}

// d_io[k] = (input, slope) of proof who[k], from device-resident inputs / outputs (n x 8 limbs each)
int fast_make_io(stark_ctx* ctx, const uint32_t* d_inputs, const uint32_t* d_outputs, const uint32_t* d_who, size_t mc,
                 const uint32_t inv_last_m1[8], uint32_t* d_io) {
    if (mc == 0) return STARK_OK;
    hipStream_t st = (hipStream_t)stark_ctx_stream(ctx);
    InvArg a;
    memcpy(a.l, inv_last_m1, 32);
    hipLaunchKernelGGL(vio_kernel, dim3((unsigned)((mc + 63) / 64)), dim3(64), 0, st, d_inputs, d_outputs, d_who, (uint32_t)mc, a, d_io,
                       (uint64_t)0, (uint64_t)0, (uint64_t*)nullptr);
    F_TRY(hipGetLastError());
    return STARK_OK;
}

// Proofs first .. first + mc of a device-resident array in one go, without asking first which of them have the fixed
// structure: the structure words are compared (canonical_out[k] = 1 where they match) and the checks are run on every
// proof, then both answers come back in one copy. status_out[k] means something only where canonical_out[k] is 1; the
// caller sends the others through the general path. Running the fast path's kernels on a proof of another structure is
// safe: every offset they read at comes from the fixed instance's template, none from the proof, and all lie inside
// the first proof_size <= stride bytes.
int fast_verify_resident_run(stark_ctx* ctx, const uint8_t* d_proofs, uint64_t stride, size_t first, size_t mc,
                             const uint32_t* d_inputs, const uint32_t* d_outputs, const uint32_t inv_last_m1[8],
                             const CheckParams& CP, const uint32_t* d_mini, uint8_t* canonical_out, uint8_t* status_out) {
    if (mc == 0) return STARK_OK;
    if (stride < tmpl().proof_size || stride % 4 != 0) {
        memset(canonical_out, 0, mc);
        return STARK_OK;
    }
    auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
    hipStream_t st = (hipStream_t)stark_ctx_stream(ctx);
    DevTemplate* D;
    int rc = get_dev_template(ctx, &D);
    if (rc != STARK_OK) return rc;
    uint8_t *d = nullptr, *d_ans = nullptr;
    if ((rc = kept_buffer(ctx, D, 3, up(8 * mc) + up(64 * mc), &d)) != STARK_OK) return rc;
    if ((rc = kept_buffer(ctx, D, 2, 2 * up(mc), &d_ans)) != STARK_OK) return rc;
    uint64_t* d_base = (uint64_t*)d;
    uint32_t* d_io = (uint32_t*)(d + up(8 * mc));
    InvArg a;
    memcpy(a.l, inv_last_m1, 32);
    hipLaunchKernelGGL(vio_kernel, dim3((unsigned)((mc + 63) / 64)), dim3(64), 0, st, d_inputs, d_outputs, (const uint32_t*)nullptr,
                       (uint32_t)mc, a, d_io, (uint64_t)first, stride, d_base);
    F_TRY(hipGetLastError());
    // the comparison of the structure words needs nothing from the sampling and nothing needs it before the verdicts are
    // copied out: it goes beside the scalar checks (in front of the sampling it held the branch launch back by its 50 us)
    const std::function<void(hipStream_t)> classify = [&](hipStream_t s) {
        hipLaunchKernelGGL(vclassify_kernel, dim3((unsigned)mc), dim3(256), 0, s, d_proofs + stride * first, stride, D->words,
                           D->n_words, d_ans);
    };
    if ((rc = fast_launch(ctx, D, d_proofs, d_base, d_io, mc, CP, d_mini, st, d_ans + up(mc), &classify)) != STARK_OK) return rc;
    // both answers in one copy
    std::vector<uint8_t> host(2 * up(mc));
    F_TRY(hipMemcpyAsync(host.data(), d_ans, 2 * up(mc), hipMemcpyDeviceToHost, st));
    F_TRY(hipStreamSynchronize(st));
    memcpy(canonical_out, host.data(), mc);
    memcpy(status_out, host.data() + up(mc), mc);
    return STARK_OK;
}

}  // namespace batch
}  // namespace starkhip
