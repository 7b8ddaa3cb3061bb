// C ABI of libstarkhip (see include/starkhip.h): context, transform plans, host-buffer staging.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <memory>
#include <new>
#include <string>
#include <thread>
#include <atomic>
#include <vector>

#include "../../include/starkhip.h"
#include "host_field.hpp"
#include "internal.hpp"

using namespace starkhip;
using hostfe::U256;

// Scratch of the allocation-free transform entry point (stark_ntt_batch_dev_static): the multi-pass intermediate
// lives here instead of in a buffer the context allocates. Every device the library runs on holds its own copy from
// load to process exit. In host code the array's name is only a handle: stark_ctx_create asks the runtime for the
// device address once per context (hipGetSymbolAddress, that context's device current) and only that pointer is ever
// handed to kernels.
static __device__ __align__(256) unsigned char g_static_scratch[STARK_STATIC_SCRATCH_BYTES];

namespace {

struct PassPlan {
    int log_t = 0;
    NttPassArgs args = {};  // in/out filled per call
    unsigned tiles = 0;
    bool poly_in_grid = false;  // grid.y = polynomials (multi-pass) or folded into tiles
};

struct NttPlan {
    unsigned log_n = 0;
    U256 root = {};
    bool inverse = false;
    std::vector<PassPlan> passes;
    std::vector<void*> tables;
    bool tiny = false;  // n <= 2
    uint32_t scale[8] = {1, 0, 0, 0, 0, 0, 0, 0};
    int has_scale = 0;
    uint64_t cols = 0;        // column plans: columns of the matrix; stage A tables: world << 32 | rank
    size_t table_bytes = 0;   // device memory held by `tables`
    uint64_t last_use = 0;    // for least-recently-used eviction
    // two passes of 1024-point or of 256-point tiles: the streaming kernels (ntt29.hip) take them unless a call overrides the last pass;
    // the last pass then applies the inter-pass factors, row k1 of the table holding w^(k1 * j) for its m inputs
    const uint32_t* stream_tw = nullptr;
    uint64_t stream_m = 0;
};

}  // namespace

struct stark_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    std::vector<std::unique_ptr<NttPlan>> plans;
    std::vector<std::unique_ptr<NttPlan>> shard_tables;  // two-level power tables of stark_twiddle_mul_dev (passes unused)
    std::vector<std::unique_ptr<NttPlan>> col_plans;     // column transforms of a row-major matrix (cols in NttPlan::cols)
    std::vector<std::unique_ptr<NttPlan>> stage_a_tabs;  // twiddle tables of stark_ntt_sharded_stage_a_dev (passes unused)
    void* scratch = nullptr;  // multi-pass intermediate
    size_t scratch_bytes = 0;
    void* static_scratch = nullptr;  // device address of g_static_scratch on this context's device
    void* io = nullptr;  // staging for the host-pointer entry points
    size_t io_bytes = 0;
    size_t chunk_bytes = (size_t)2048 << 20;  // input bytes transformed per group of pass launches (scratch = 1.5x)
    size_t table_budget = (size_t)8 << 30;   // twiddle tables kept across calls (STARKHIP_TABLE_MB)
    uint64_t use_counter = 0;
    std::unique_ptr<CtxAttachment> attach[CTX_SLOTS];
    // pinned staging for copies from / to pageable host memory: a pair of buffers per direction, so that an upload and a
    // download may be staged by two host threads at once
    struct PinPair {
        char* buf[2] = {nullptr, nullptr};
        hipEvent_t ev[2] = {nullptr, nullptr};
        bool busy[2] = {false, false};  // a transfer recorded in ev[i] may still read or write buf[i]
        unsigned next = 0;              // the buffer the next staged piece takes (they alternate across calls too)
        bool tried = false;
        std::unique_ptr<WorkerPool> pool;  // the host threads that fill or drain the buffers
    } up, down;
    hipStream_t aux[starkhip::CTX_AUX_STREAMS] = {};  // ctx_aux_stream: the only other streams the library creates
    // work counters of the streaming transform kernels: one block of 8 per stream that ever launched them (launches on
    // one stream run in order and zero the block in between; launches on different streams must not share one)
    std::vector<std::pair<hipStream_t, uint32_t*>> tickets;
    uint32_t* ticket_pool = nullptr;  // TICKET_POOL blocks of 256 bytes, allocated with the context: the first streams
                                      // that launch the streaming kernels take theirs from here, so that such a call
                                      // allocates nothing (further streams get a block of their own by hipMalloc)
    static constexpr size_t TICKET_POOL = 16;
    int n_cu = 0;  // compute units of the device
    unsigned copy_threads = 8;
};

namespace starkhip {

namespace {
constexpr size_t PIN_BYTES = (size_t)64 << 20;   // each of the two staging buffers
constexpr size_t STAGE_MIN = (size_t)8 << 20;    // smaller copies go straight through hipMemcpyAsync

bool pin_ready(stark_ctx* ctx, stark_ctx::PinPair& P) {
    if (!P.tried) {
        P.tried = true;
        const char* off = getenv("STARKHIP_PINNED");
        if (off && atoi(off) == 0 && *off) return false;
        bool ok = hipHostMalloc((void**)&P.buf[0], PIN_BYTES, hipHostMallocDefault) == hipSuccess &&
                  hipHostMalloc((void**)&P.buf[1], PIN_BYTES, hipHostMallocDefault) == hipSuccess &&
                  hipEventCreateWithFlags(&P.ev[0], hipEventDisableTiming) == hipSuccess &&
                  hipEventCreateWithFlags(&P.ev[1], hipEventDisableTiming) == hipSuccess;
        if (!ok) {
            (void)hipGetLastError();
            for (int i = 0; i < 2; i++) {
                if (P.buf[i]) (void)hipHostFree(P.buf[i]);
                if (P.ev[i]) (void)hipEventDestroy(P.ev[i]);
                P.buf[i] = nullptr;
                P.ev[i] = nullptr;
            }
        }
        if (const char* t = getenv("STARKHIP_HOST_THREADS"))
            if (atoi(t) > 0) ctx->copy_threads = (unsigned)atoi(t);
        if (ctx->copy_threads > 16) ctx->copy_threads = 16;
    }
    return P.buf[0] != nullptr;
}

void par_copy(char* dst, const char* src, size_t n, unsigned threads, stark_ctx::PinPair& P) {
    if (threads <= 1 || n < ((size_t)1 << 20)) {
        memcpy(dst, src, n);
        return;
    }
    if (!P.pool) P.pool.reset(new WorkerPool(threads - 1));  // the caller is the last of the `threads`
    const size_t per = ((n + threads - 1) / threads + 4095) & ~(size_t)4095;
    P.pool->run(threads, [&](size_t i) {
        const size_t lo = per * i, hi = lo + per < n ? lo + per : n;
        if (lo < hi) memcpy(dst + lo, src + lo, hi - lo);
    });
}
}  // namespace

// Is [p, p + n) host memory that the device can reach directly (hipHostMalloc, hipHostRegister, a pinned tensor)?
// Such a buffer needs no staging: the copy engine reads or writes it at the link's rate and no host thread touches the
// bytes. Both ends of the range are asked about; ordinary (pageable) memory is unknown to the runtime, which answers
// with an error on some versions and with "unregistered" on others.
static bool host_range_is_pinned(const void* p, size_t n) {
    if (!p || n == 0) return false;
    for (const char* q : {(const char*)p, (const char*)p + (n - 1)}) {
        hipPointerAttribute_t a;
        if (hipPointerGetAttributes(&a, q) != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
        if (a.type != hipMemoryTypeHost) return false;
    }
    return true;
}

// Before buf[b] is written again: wait for the transfer that last used it. Waiting here, not at the end of the call
// that started the transfer, lets the caller prepare its next piece while the last one is still on the link.
static hipError_t pin_acquire(stark_ctx::PinPair& P, unsigned b) {
    if (!P.busy[b]) return hipSuccess;
    P.busy[b] = false;
    return hipEventSynchronize(P.ev[b]);
}

// Pageable host memory -> device through the two pinned upload buffers. On return `src` has been read completely;
// the last one or two pieces may still be in flight on `stream`. (A source that is pinned already goes as one
// asynchronous copy and is still being read on return: every entry point drains its streams before it returns, and
// the library's own temporaries are pageable.)
hipError_t copy_h2d(stark_ctx* ctx, void* d_dst, const void* src, size_t n, hipStream_t stream) {
    if (n < STAGE_MIN || !ctx || host_range_is_pinned(src, n) || !pin_ready(ctx, ctx->up))
        return hipMemcpyAsync(d_dst, src, n, hipMemcpyHostToDevice, stream);
    stark_ctx::PinPair& P = ctx->up;
    hipError_t e = hipSuccess;
    for (size_t o = 0; o < n && e == hipSuccess; o += PIN_BYTES) {
        const size_t len = n - o < PIN_BYTES ? n - o : PIN_BYTES;
        const unsigned b = P.next;
        P.next ^= 1u;
        if ((e = pin_acquire(P, b)) != hipSuccess) break;
        par_copy(P.buf[b], (const char*)src + o, len, ctx->copy_threads, P);
        e = hipMemcpyAsync((char*)d_dst + o, P.buf[b], len, hipMemcpyHostToDevice, stream);
        if (e == hipSuccess) e = hipEventRecord(P.ev[b], stream);
        if (e == hipSuccess) P.busy[b] = true;
    }
    return e;
}

// Device -> pageable host memory through the two pinned download buffers; complete on return.
hipError_t copy_d2h(stark_ctx* ctx, void* dst, const void* d_src, size_t n, hipStream_t stream) {
    if (n < STAGE_MIN || !ctx || host_range_is_pinned(dst, n) || !pin_ready(ctx, ctx->down)) {
        hipError_t e = hipMemcpyAsync(dst, d_src, n, hipMemcpyDeviceToHost, stream);
        return e == hipSuccess ? hipStreamSynchronize(stream) : e;
    }
    stark_ctx::PinPair& P = ctx->down;
    hipError_t e = hipSuccess;
    const size_t chunks = (n + PIN_BYTES - 1) / PIN_BYTES;
    for (size_t k = 0; k <= chunks && e == hipSuccess; k++) {
        if (k < chunks) {
            const size_t o = k * PIN_BYTES, len = n - o < PIN_BYTES ? n - o : PIN_BYTES;
            const unsigned b = (unsigned)k & 1u;
            e = pin_acquire(P, b);
            if (e == hipSuccess) e = hipMemcpyAsync(P.buf[b], (const char*)d_src + o, len, hipMemcpyDeviceToHost, stream);
            if (e == hipSuccess) e = hipEventRecord(P.ev[b], stream);
            if (e == hipSuccess) P.busy[b] = true;
        }
        if (k >= 1 && e == hipSuccess) {
            const size_t o = (k - 1) * PIN_BYTES, len = n - o < PIN_BYTES ? n - o : PIN_BYTES;
            const unsigned b = (unsigned)(k - 1) & 1u;
            e = pin_acquire(P, b);
            if (e == hipSuccess) par_copy((char*)dst + o, P.buf[b], len, ctx->copy_threads, P);
        }
    }
    return e;
}

CtxAttachment* ctx_get_attachment(stark_ctx* ctx, int slot) {
    return ctx && slot >= 0 && slot < CTX_SLOTS ? ctx->attach[slot].get() : nullptr;
}
void ctx_set_attachment(stark_ctx* ctx, int slot, CtxAttachment* a) {
    if (ctx && slot >= 0 && slot < CTX_SLOTS) ctx->attach[slot].reset(a);
}
hipError_t ctx_make_current(stark_ctx* ctx) { return ctx ? hipSetDevice(ctx->device) : hipErrorInvalidValue; }

// Auxiliary stream `which` of the context (made on first use, destroyed with the context). Rules every user keeps, so
// that work spread over the context's streams can always drain:
//  * plain streams, all of one priority. A wait whose waiter outranks its signaller can starve the signaller where
//    queues are served in strict priority order (a waiting queue still counts as busy), so the library creates no
//    priority stream at all;
//  * an event is recorded before any stream is told to wait for it, and every wait points from work queued later to
//    work queued earlier. The wait-for graph therefore follows host enqueue order and has no cycle, whichever hardware
//    queue a stream lands on and whether or not two of them share one (a shared queue runs its packets in the order
//    they were queued, which is the same order);
//  * no kernel of the library spins on memory another kernel writes: every dependency between launches is such an
//    event or stream order;
//  * an entry point that used an auxiliary stream waits for it before it returns, so between calls only the main
//    stream (and streams the caller passed in) can hold work.
hipError_t ctx_aux_stream(stark_ctx* ctx, int which, hipStream_t* out) {
    if (!ctx || which < 0 || which >= CTX_AUX_STREAMS) return hipErrorInvalidValue;
    if (!ctx->aux[which]) {
        hipError_t e = hipStreamCreateWithFlags(&ctx->aux[which], hipStreamNonBlocking);
        if (e != hipSuccess) {
            ctx->aux[which] = nullptr;
            return e;
        }
    }
    *out = ctx->aux[which];
    return hipSuccess;
}
void ctx_set_error(stark_ctx* ctx, const char* text) {
    if (ctx) ctx->err = text ? text : "";
}
}  // namespace starkhip

namespace {

int fail(stark_ctx* ctx, int code, const char* fmt, ...) {
    if (ctx) {
        char buf[512];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof buf, fmt, ap);
        va_end(ap);
        ctx->err = buf;
    }
    return code;
}

#define HIP_TRY(ctx, call)                                                                      \
    do {                                                                                        \
        hipError_t e_ = (call);                                                                 \
        if (e_ != hipSuccess)                                                                   \
            return fail(ctx, e_ == hipErrorOutOfMemory ? STARK_ERR_NOMEM : STARK_ERR_HIP,       \
                        "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

int ensure_buffer(stark_ctx* ctx, void** buf, size_t* have, size_t need) {
    if (*have >= need) return STARK_OK;
    if (*buf) {
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        HIP_TRY(ctx, hipFree(*buf));
        *buf = nullptr;
        *have = 0;
    }
    HIP_TRY(ctx, hipMalloc(buf, need));
    *have = need;
    return STARK_OK;
}

int env_int(const char* name, int dflt) {
    const char* s = getenv(name);
    return s && *s ? atoi(s) : dflt;
}

// How a transform of 2^log_n points is cut into passes of at most 2^12 points.
std::vector<int> choose_split(unsigned log_n) {
    const char* s = getenv("STARKHIP_SPLIT");  // e.g. "8,8,8": tuning aid, must sum to log_n
    if (s && *s) {
        std::vector<int> v;
        int sum = 0;
        for (const char* p = s; *p;) {
            int x = atoi(p);
            if (x >= 2 && x <= 12) {
                v.push_back(x);
                sum += x;
            }
            while (*p && *p != ',') p++;
            if (*p == ',') p++;
        }
        if (sum == (int)log_n && v.size() >= 1 && v.size() <= 3) return v;
    }
    if (log_n <= 12) return {(int)log_n};
    // two passes up to 2^21, the smaller tile first (its strided side is the costly one); from 2^22 on
    // three passes of 2^7..2^8-point tiles beat tiles of 2^11..2^12 points (measured 10..12 % faster at 2^22 and 2^24)
    if (log_n <= 21) return {(int)log_n / 2, (int)(log_n + 1) / 2};
    int a = ((int)log_n + 2) / 3, b = ((int)log_n - a + 1) / 2;
    return {a, b, (int)log_n - a - b};
}

// Tiles per workgroup (log2): as many as fit the thread and LDS budget and the extent.
int choose_log_c(int log_t, unsigned log_extent, int target_threads) {
    int log_q = log_t - 2;
    int log_c = 0;
    int lim = env_int("STARKHIP_LDS_LIMIT", 80 * 1024);
    auto lds = [&](int lc) { return (int)ntt29_pass_lds_bytes(log_t, lc); };
    while (log_c + 1 <= (int)log_extent && (1 << (log_q + log_c + 1)) <= target_threads && lds(log_c + 1) <= lim)
        log_c++;
    return log_c;
}

// out[i] = scale * base^((row0 + i / row_len) * (i % row_len)), or scale * base^i when row_len == 0
int make_table(stark_ctx* ctx, NttPlan* plan, const U256& base, size_t count, uint64_t row_len, const U256& scale,
               uint32_t** out, bool unpack = false, uint64_t row0 = 0) {
    uint32_t pw[64 * 8];
    U256 b = base;
    for (int k = 0; k < 64; k++) {
        hostfe::to_limbs(b, pw + 8 * k);
        b = hostfe::mul(b, b);
    }
    void* d_pw = nullptr;
    void* d_tab = nullptr;
    HIP_TRY(ctx, hipMalloc(&d_pw, sizeof pw));
    hipError_t e = hipMalloc(&d_tab, count * 32);
    if (e != hipSuccess) {
        (void)hipFree(d_pw);
        return fail(ctx, STARK_ERR_NOMEM, "twiddle table of %zu bytes: %s", count * 32, hipGetErrorString(e));
    }
    plan->tables.push_back(d_tab);
    plan->table_bytes += count * (unpack ? 48 : 32);
    uint32_t sc[8];
    hostfe::to_limbs(scale, sc);
    e = hipMemcpyAsync(d_pw, pw, sizeof pw, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = launch_fill_powers((uint32_t*)d_tab, count, row_len, row0, (const uint32_t*)d_pw, sc, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d_pw);
    if (e != hipSuccess) return fail(ctx, STARK_ERR_HIP, "twiddle table build: %s", hipGetErrorString(e));
    if (unpack) {  // 12 dwords per entry for the 29-bit-limb kernels
        void* d_un = nullptr;
        e = hipMalloc(&d_un, count * 48);
        if (e != hipSuccess) return fail(ctx, STARK_ERR_NOMEM, "twiddle table of %zu bytes: %s", count * 48, hipGetErrorString(e));
        e = launch_unpack_table((const uint32_t*)d_tab, (uint32_t*)d_un, count, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        (void)hipFree(d_tab);
        plan->tables.back() = d_un;
        if (e != hipSuccess) return fail(ctx, STARK_ERR_HIP, "twiddle table unpack: %s", hipGetErrorString(e));
        d_tab = d_un;
    }
    *out = (uint32_t*)d_tab;
    return STARK_OK;
}

void free_plan(NttPlan* p) {
    for (void* t : p->tables) (void)hipFree(t);
    p->tables.clear();
    p->table_bytes = 0;
}

// nine 29-bit limbs of a canonical value
void to_limbs29(const U256& x, uint32_t out[9]) {
    for (int i = 0; i < 9; i++) {
        const unsigned bit = 29u * (unsigned)i, w = bit / 64, sh = bit % 64;
        uint64_t v = x.w[w] >> sh;
        if (sh > 35 && w + 1 < 4) v |= x.w[w + 1] << (64 - sh);
        out[i] = (uint32_t)(v & 0x1FFFFFFFu);
    }
}

// the primitive 4th root of unity of a tile whose root of unity is w_t (order 2^log_t >= 4)
void set_quarter_root(NttPassArgs& a, const U256& w_t, int log_t) {
    const U256 q = log_t >= 2 ? hostfe::pow_u64(w_t, (uint64_t)1 << (log_t - 2)) : hostfe::from_u64(1);
    to_limbs29(q, a.iq);
    to_limbs29(hostfe::mul(q, hostfe::pow_u64(hostfe::from_u64(2), 145)), a.iq2);
}

// `root` has order exactly 2^log_n (checked by the caller).
int build_plan(stark_ctx* ctx, unsigned log_n, const U256& root, bool inverse, NttPlan** out) {
    std::unique_ptr<NttPlan> plan(new NttPlan);
    plan->log_n = log_n;
    plan->root = root;
    plan->inverse = inverse;
    const uint64_t n = (uint64_t)1 << log_n;
    // the transform's own root and the factor applied to every output
    U256 w = inverse ? hostfe::pow_u64(root, n - 1) : root;  // root^-1 = root^(n-1)
    U256 scale = inverse ? hostfe::inverse(hostfe::from_u64(n)) : hostfe::from_u64(1);
    hostfe::to_limbs(scale, plan->scale);
    plan->has_scale = inverse && log_n > 0;
    const U256 one = hostfe::from_u64(1);
    int rc;

    if (log_n <= 1) {
        plan->tiny = true;
        *out = plan.get();
        ctx->plans.push_back(std::move(plan));
        return STARK_OK;
    }

    const bool l29 = true;  // tile twiddle tables are stored unpacked (12 dwords per entry)
    std::vector<int> split = choose_split(log_n);
    const int target_threads = env_int("STARKHIP_THREADS", 512);
    const int forced_log_c = env_int("STARKHIP_LOGC", -1);
    const bool tiled_mid = env_int("STARKHIP_TILED_MID", 1) != 0;
    auto pick_c = [&](int log_t, unsigned log_extent) {
        int lc = choose_log_c(log_t, log_extent, log_t >= 12 ? 1024 : target_threads);
        if (forced_log_c >= 0 && forced_log_c <= (int)log_extent && (1 << (log_t - 2 + forced_log_c)) <= 1024 &&
            ntt29_pass_lds_bytes(log_t, forced_log_c) <= 160 * 1024)
            lc = forced_log_c;
        return lc;
    };

    if (split.size() == 1) {
        PassPlan p;
        p.log_t = (int)log_n;
        int log_c = 0;  // polynomials per workgroup: fill 256 lanes
        while ((1 << (p.log_t - 2 + log_c)) < 256) log_c++;
        NttPassArgs& a = p.args;
        a.log_c = log_c;
        a.dsh = 31;
        a.dmask = 0xFFFFFFFFu;
        a.inB = a.outB = n << log_c;
        a.inC = a.outC = n;
        a.inS = a.outS = 1;
        a.load_cfast = a.store_cfast = 0;
        a.canon_in = 1;
        a.has_scale = plan->has_scale;
        memcpy(a.scale, plan->scale, sizeof a.scale);
        uint32_t* W;
        if ((rc = make_table(ctx, plan.get(), w, n, 0, one, &W, l29)) != STARK_OK) {
            free_plan(plan.get());
            return rc;
        }
        a.W = W;
        set_quarter_root(a, w, (int)log_n);
        if (plan->has_scale && log_n >= 3 && env_int("STARKHIP_FOLD_SCALE", 1)) {
            // inverse: n^-1 rides on the last round (the kernel's last round has twiddles from 8 points on)
            uint32_t* Ws;
            if ((rc = make_table(ctx, plan.get(), w, n, 0, scale, &Ws, l29)) != STARK_OK) {
                free_plan(plan.get());
                return rc;
            }
            a.Wlast = Ws;
        }
        plan->passes.push_back(p);
    } else {
        // n = n1 * m, m = n2 (* n3)
        const int l1 = split[0];
        const unsigned lm = log_n - l1;
        const uint64_t n1 = (uint64_t)1 << l1, m = (uint64_t)1 << lm;
        uint32_t *W1, *TW1;
        U256 w1 = hostfe::pow_u64(w, m);  // order n1
        if ((rc = make_table(ctx, plan.get(), w1, n1, 0, one, &W1, l29)) != STARK_OK ||
            (rc = make_table(ctx, plan.get(), w, n, m, scale, &TW1)) != STARK_OK) {
            free_plan(plan.get());
            return rc;
        }
        {
            PassPlan p;
            p.log_t = l1;
            p.poly_in_grid = true;
            NttPassArgs& a = p.args;
            a.log_c = pick_c(l1, lm);
            const uint64_t C = (uint64_t)1 << a.log_c;
            a.dsh = 31;
            a.dmask = 0xFFFFFFFFu;
            a.poly_stride = n;
            a.inB = a.outB = a.twB = C;
            a.inC = a.outC = a.twC = 1;
            a.inS = a.outS = a.twS = m;
            a.c_limit = ~0ull;
            a.load_cfast = a.store_cfast = 1;
            a.canon_in = 1;
            a.W = W1;
            set_quarter_root(a, w1, l1);
            a.TW = TW1;
            p.tiles = (unsigned)(m >> a.log_c);
            if (split.size() == 2 && tiled_mid) {
                // two passes: the intermediate is stored tile by tile, [tile][k1][c] (n1 * C elements each), so that every
                // workgroup writes one contiguous run of whole memory lines instead of n1 pieces of C * 36 bytes
                a.outB = n1 * C;
                a.outS = C;
            }
            plan->passes.push_back(p);
        }
        if (split.size() == 2) {
            const int l2 = split[1];
            uint32_t* W2;
            U256 w2 = hostfe::pow_u64(w, n1);  // order m
            if ((rc = make_table(ctx, plan.get(), w2, m, 0, one, &W2, l29)) != STARK_OK) {
                free_plan(plan.get());
                return rc;
            }
            PassPlan p;
            p.log_t = l2;
            p.poly_in_grid = true;
            NttPassArgs& a = p.args;
            a.log_c = pick_c(l2, l1);
            const uint64_t C = (uint64_t)1 << a.log_c;
            a.dsh = 31;
            a.dmask = 0xFFFFFFFFu;
            a.poly_stride = n;
            a.inB = C * m;
            a.inC = m;
            a.inS = 1;
            a.outB = C;
            a.outC = 1;
            a.outS = n1;
            a.c_limit = ~0ull;
            a.load_cfast = 0;
            a.store_cfast = 1;
            if (tiled_mid) {
                // row k1 = lo * C + c of the tiled intermediate: element j is in tile j / C1 at [k1][j % C1]; the rows of
                // one workgroup are adjacent there, so the lanes take the row index fastest
                const uint32_t lc1 = plan->passes[0].args.log_c;
                a.inB = C << lc1;
                a.inC = (uint64_t)1 << lc1;
                a.in_split = 1;
                a.jsh = lc1;
                a.jmask = (1u << lc1) - 1u;
                a.inS2 = n1 << lc1;
                a.load_cfast = 1;  // element-fastest lanes (72-byte runs per wavefront) measured 1.5 % slower
            }
            a.W = W2;
            set_quarter_root(a, w2, l2);
            p.tiles = (unsigned)(n1 >> a.log_c);
            plan->passes.push_back(p);
            if (tiled_mid && l1 == l2 && (l1 == 10 || l1 == 8) && plan->passes[0].args.log_c == (uint32_t)(11 - l1) &&
                a.log_c == (uint32_t)(11 - l1) && p.tiles % 8 == 0 && plan->passes[0].tiles % 8 == 0 &&
                env_int("STARKHIP_STREAM", 1) != 0) {
                plan->stream_tw = TW1;
                plan->stream_m = m;
            }
        } else {
            const int l2 = split[1], l3 = split[2];
            const uint64_t n2 = (uint64_t)1 << l2, n3 = (uint64_t)1 << l3;
            uint32_t *W2, *W3, *TW2;
            U256 wm = hostfe::pow_u64(w, n1);      // order m
            U256 w2 = hostfe::pow_u64(wm, n3);     // order n2
            U256 w3 = hostfe::pow_u64(wm, n2);     // order n3
            if ((rc = make_table(ctx, plan.get(), w2, n2, 0, one, &W2, l29)) != STARK_OK ||
                (rc = make_table(ctx, plan.get(), w3, n3, 0, one, &W3, l29)) != STARK_OK ||
                (rc = make_table(ctx, plan.get(), wm, m, n3, one, &TW2)) != STARK_OK) {
                free_plan(plan.get());
                return rc;
            }
            {
                PassPlan p;
                p.log_t = l2;
                p.poly_in_grid = true;
                NttPassArgs& a = p.args;
                a.log_c = pick_c(l2, (unsigned)l3);
                const uint64_t C = (uint64_t)1 << a.log_c;
                a.dsh = (uint32_t)(l3 - (int)a.log_c);  // tiles per row = n3 / C
                a.dmask = (uint32_t)((n3 >> a.log_c) - 1);
                a.poly_stride = n;
                a.inA = a.outA = m;
                a.inB = a.outB = a.twB = C;
                a.inC = a.outC = a.twC = 1;
                a.inS = a.outS = a.twS = n3;
                a.c_limit = ~0ull;
                a.load_cfast = a.store_cfast = 1;
                a.W = W2;
                set_quarter_root(a, w2, l2);
                a.TW = TW2;
                p.tiles = (unsigned)(n1 * (n3 >> a.log_c));
                plan->passes.push_back(p);
            }
            {
                PassPlan p;
                p.log_t = l3;
                p.poly_in_grid = true;
                NttPassArgs& a = p.args;
                a.log_c = pick_c(l3, (unsigned)l1);
                const uint64_t C = (uint64_t)1 << a.log_c;
                a.dsh = (uint32_t)l2;
                a.dmask = (uint32_t)(n2 - 1);
                a.poly_stride = n;
                a.inA = C * m;
                a.inB = n3;
                a.inC = m;
                a.inS = 1;
                a.outA = C;
                a.outB = n1;
                a.outC = 1;
                a.outS = n1 * n2;
                a.c_limit = ~0ull;
                a.load_cfast = 0;
                a.store_cfast = 1;
                a.W = W3;
                set_quarter_root(a, w3, l3);
                p.tiles = (unsigned)((n1 >> a.log_c) * n2);
                plan->passes.push_back(p);
            }
        }
    }
    *out = plan.get();
    ctx->plans.push_back(std::move(plan));
    return STARK_OK;
}

// Cached plans are kept until their tables exceed the budget; then the least recently used ones go.
// Their tables may still be read by kernels in flight on any stream, hence the device-wide wait.
int evict_plans(stark_ctx* ctx, const NttPlan* keep) {
    size_t total = 0;
    for (auto& p : ctx->plans) total += p->table_bytes;
    if (total <= ctx->table_budget) return STARK_OK;
    HIP_TRY(ctx, hipDeviceSynchronize());
    while (total > ctx->table_budget) {
        size_t victim = ctx->plans.size();
        for (size_t i = 0; i < ctx->plans.size(); i++)
            if (ctx->plans[i].get() != keep && ctx->plans[i]->table_bytes &&
                (victim == ctx->plans.size() || ctx->plans[i]->last_use < ctx->plans[victim]->last_use))
                victim = i;
        if (victim == ctx->plans.size()) break;
        total -= ctx->plans[victim]->table_bytes;
        free_plan(ctx->plans[victim].get());
        ctx->plans.erase(ctx->plans.begin() + (long)victim);
    }
    return STARK_OK;
}

int get_plan(stark_ctx* ctx, unsigned log_n, const U256& root, bool inverse, NttPlan** out) {
    for (auto& p : ctx->plans)
        if (p->log_n == log_n && p->inverse == inverse && hostfe::eq(p->root, root)) {
            p->last_use = ++ctx->use_counter;
            *out = p.get();
            return STARK_OK;
        }
    int rc = build_plan(ctx, log_n, root, inverse, out);
    if (rc != STARK_OK) return rc;
    (*out)->last_use = ++ctx->use_counter;
    return evict_plans(ctx, *out);
}

// root^n == 1 and, for n > 1, root^(n/2) != 1  (n = 2^log_n): the order is exactly n
bool has_order(const U256& root, unsigned log_n) {
    U256 x = root;
    const U256 one = hostfe::from_u64(1);
    for (unsigned i = 0; i < log_n; i++) {
        if (i + 1 == log_n && hostfe::eq(x, one)) return false;
        x = hostfe::mul(x, x);
    }
    return hostfe::eq(x, one);
}

// fixed != nullptr: the multi-pass intermediate goes to that buffer of fixed_bytes and nothing is allocated
// What the last pass of a plan does differently when it is stage A of a transform sharded over several GPUs: every
// output is multiplied by a table entry and stored into the send block of the rank that owns it.
struct LastPassOverride {
    const uint32_t* TW = nullptr;  // [polynomial][output index], polynomial stride tw_poly elements
    uint64_t tw_poly = 0;
    uint64_t out_poly = 0;         // polynomial stride of the output
    uint32_t ksh = 31;             // output index k goes to (k >> ksh) * outS2 + (k & (2^ksh - 1))
    uint64_t outS2 = 0;
};

int run_plan(stark_ctx* ctx, NttPlan* plan, const uint32_t* d_in, uint32_t* d_out, size_t batch, hipStream_t stream,
             void* fixed = nullptr, size_t fixed_bytes = 0, const LastPassOverride* ov = nullptr) {
    const uint64_t n = (uint64_t)1 << plan->log_n;
    if (batch == 0) return STARK_OK;
    if (plan->tiny) {
        HIP_TRY(ctx, launch_ntt_tiny(d_in, d_out, (int)plan->log_n, batch, plan->scale, plan->has_scale, stream));
        return STARK_OK;
    }
    if (plan->passes.size() == 1) {
        PassPlan& p = plan->passes[0];
        const uint64_t C = (uint64_t)1 << p.args.log_c;
        // at most 2^31 tiles per launch
        const size_t max_polys = ((size_t)1 << 30) * C;
        for (size_t done = 0; done < batch; done += max_polys) {
            size_t cnt = batch - done < max_polys ? batch - done : max_polys;
            NttPassArgs a = p.args;
            a.in = d_in + 8 * n * done;
            a.out = d_out + 8 * n * done;
            a.c_limit = cnt;
            a.ksh = 31;
            a.kmask = 0xFFFFFFFFu;
            if (ov) {  // polynomials are folded into the tile index here: lo * C + c
                a.Wlast = nullptr;  // the override's table carries the plan's output factor
                a.out = d_out + 8 * ov->out_poly * done;
                a.outB = ov->out_poly << a.log_c;
                a.outC = ov->out_poly;
                a.TW = ov->TW + 8 * ov->tw_poly * done;
                a.twB = ov->tw_poly << a.log_c;
                a.twC = ov->tw_poly;
                a.twS = 1;
                a.ksh = ov->ksh;
                a.kmask = (1u << ov->ksh) - 1u;
                a.outS2 = ov->outS2;
            }
            HIP_TRY(ctx, launch_ntt29_pass(p.log_t, a, (unsigned)((cnt + C - 1) / C), 1, stream));
        }
        return STARK_OK;
    }
    // multi-pass: first pass into scratch, middle pass in place there, last pass to d_out
    const size_t mid_bytes = ntt29_mid_bytes();  // bytes per element of the intermediate (nine unpacked limbs)
    size_t chunk = ctx->chunk_bytes / (n * 32);
    if (chunk < 1) chunk = 1;
    if (chunk > batch) chunk = batch;
    if (chunk > 65535) chunk = 65535;
    uint32_t* tmp;
    if (fixed) {
        if (n * mid_bytes > fixed_bytes)
            return fail(ctx, STARK_ERR_UNSUPPORTED, "2^%u points need %zu bytes of intermediate, the static scratch has %zu",
                        plan->log_n, (size_t)(n * mid_bytes), fixed_bytes);
        if (chunk > fixed_bytes / (n * mid_bytes)) chunk = fixed_bytes / (n * mid_bytes);
        tmp = (uint32_t*)fixed;
    } else {
        int rc = ensure_buffer(ctx, &ctx->scratch, &ctx->scratch_bytes, chunk * n * mid_bytes);
        if (rc != STARK_OK) return rc;
        tmp = (uint32_t*)ctx->scratch;
    }
    for (size_t done = 0; done < batch; done += chunk) {
        size_t cnt = batch - done < chunk ? batch - done : chunk;
        for (size_t k = 0; k < plan->passes.size(); k++) {
            PassPlan& p = plan->passes[k];
            NttPassArgs a = p.args;
            a.in = k == 0 ? d_in + 8 * n * done : tmp;
            a.out = k + 1 == plan->passes.size() ? d_out + 8 * n * done : tmp;
            a.in_raw = k > 0;
            a.out_raw = k + 1 < plan->passes.size();
            a.outP = a.poly_stride;
            a.ksh = 31;
            a.kmask = 0xFFFFFFFFu;
            // The streaming kernels pay from 16 work items per resident workgroup on (16 polynomials of 2^20 points on 256
            // compute units; measured: -3 % at 4 polynomials, -1 % at 8, even at 16, +1.5 % at 32, +3.4 % at 64: with
            // few items little is left to fetch ahead, and zeroing the counters is a launch of its own); calls that
            // override the last pass and the allocation-free entry point keep to the plain kernels.
            if (plan->stream_tw && !ov && !fixed &&
                (uint64_t)p.tiles * cnt >= (uint64_t)env_int("STARKHIP_STREAM_MIN", 16) * ntt29_stream_resident(ctx->n_cu)) {
                uint32_t* tk = nullptr;
                for (auto& t : ctx->tickets)
                    if (t.first == stream) tk = t.second;
                if (!tk) {
                    if (ctx->tickets.size() < stark_ctx::TICKET_POOL) tk = ctx->ticket_pool + 64 * ctx->tickets.size();
                    else HIP_TRY(ctx, hipMalloc((void**)&tk, 256));
                    ctx->tickets.emplace_back(stream, tk);
                }
                a.tickets = tk;
                a.stream = 1;
                if (k == 1) {
                    a.TW = plan->stream_tw;
                    a.twP = a.twA = 0;
                    a.twB = plan->stream_m << a.log_c;
                    a.twC = plan->stream_m;
                    a.twS = 1;
                }
            }
            if (ov && k + 1 == plan->passes.size()) {
                // the last pass stores output index o = k1 + n1' * k (outS = n1'); the split point 2^ksh of o is at or
                // above n1' (checked by the caller), so it splits k alone
                unsigned log_s = 0;
                while (((uint64_t)1 << log_s) < a.outS) log_s++;
                a.out = d_out + 8 * ov->out_poly * done;
                a.outP = ov->out_poly;
                a.TW = ov->TW + 8 * ov->tw_poly * done;
                a.twP = ov->tw_poly;
                a.twA = a.outA;
                a.twB = a.outB;
                a.twC = a.outC;
                a.twS = a.outS;
                a.ksh = ov->ksh - log_s;
                a.kmask = (1u << a.ksh) - 1u;
                a.outS2 = ov->outS2;
            }
            HIP_TRY(ctx, launch_ntt29_pass(p.log_t, a, p.tiles, (unsigned)cnt, stream));
        }
    }
    return STARK_OK;
}

int ntt_common(stark_ctx* ctx, const void* d_in, void* d_out, unsigned log_n, size_t batch, const uint32_t root[8],
               int inverse, hipStream_t stream) {
    if (log_n > 30) return fail(ctx, STARK_ERR_UNSUPPORTED, "log_n %u above 30", log_n);
    U256 r = hostfe::canon(hostfe::from_limbs(root));
    if (!has_order(r, log_n))
        return fail(ctx, STARK_ERR_INVALID, "root of unity does not have order 2^%u", log_n);
    NttPlan* plan;
    int rc = get_plan(ctx, log_n, r, inverse != 0, &plan);
    if (rc != STARK_OK) return rc;
    return run_plan(ctx, plan, (const uint32_t*)d_in, (uint32_t*)d_out, batch, stream);
}

// Host buffers, more than one staging buffer of polynomials: groups of `per_group` polynomials go through three stages
// that overlap, each with its own stream and host thread: a helper thread stages group g + 1 into the device buffer
// (ctx->io) while this thread queues the transform of group g behind its upload event and a second helper thread
// drains group g - 1 behind its transform event. The link carries both directions at once, so a batch costs about
// one direction's transfer time instead of two plus the transforms.
int ntt_host_pipelined(stark_ctx* ctx, uint32_t* limbs, unsigned log_n, size_t batch, size_t per_group, NttPlan* plan) {
    const size_t poly_bytes = (size_t)32 << log_n, poly_words = (size_t)8 << log_n;
    const size_t groups = (batch + per_group - 1) / per_group;
    hipStream_t stream2, stream3;  // transforms and downloads; the uploads take the context's main stream
    HIP_TRY(ctx, ctx_aux_stream(ctx, 0, &stream2));
    HIP_TRY(ctx, ctx_aux_stream(ctx, 1, &stream3));
    // the multi-pass intermediate is sized before anything is queued: growing it later would free memory in use
    if (plan->passes.size() > 1) {
        const size_t per_call = std::min(per_group, std::max<size_t>(1, ctx->chunk_bytes / poly_bytes));
        int rc = ensure_buffer(ctx, &ctx->scratch, &ctx->scratch_bytes, per_call * ((size_t)1 << log_n) * ntt29_mid_bytes());
        if (rc != STARK_OK) return rc;
    }
    std::vector<hipEvent_t> up(groups, nullptr), done(groups, nullptr);
    auto drop_events = [&] {
        for (hipEvent_t e : up)
            if (e) (void)hipEventDestroy(e);
        for (hipEvent_t e : done)
            if (e) (void)hipEventDestroy(e);
    };
    for (size_t g = 0; g < groups; g++)
        if (hipEventCreateWithFlags(&up[g], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&done[g], hipEventDisableTiming) != hipSuccess) {
            drop_events();
            return fail(ctx, STARK_ERR_HIP, "event creation failed");
        }
    uint32_t* d = (uint32_t*)ctx->io;
    // two staging threads at once: six copy threads each measured best on 16 host cores (8: +5 %, 12: +15 % time)
    const unsigned saved_threads = ctx->copy_threads;
    (void)pin_ready(ctx, ctx->up);  // reads STARKHIP_HOST_THREADS on first use
    if (ctx->copy_threads > 6 && !getenv("STARKHIP_HOST_THREADS")) ctx->copy_threads = 6;
    std::atomic<size_t> uploaded{0}, transformed{0};
    std::atomic<int> up_err{0}, down_err{0};
    auto count_of = [&](size_t g) { return std::min(per_group, batch - g * per_group); };
    std::thread uploader([&] {
        hipError_t e = hipSetDevice(ctx->device);  // the current device is per thread
        for (size_t g = 0; g < groups; g++) {
            if (e == hipSuccess)
                e = copy_h2d(ctx, d + g * per_group * poly_words, limbs + g * per_group * poly_words, count_of(g) * poly_bytes, ctx->stream);
            if (e == hipSuccess) e = hipEventRecord(up[g], ctx->stream);
            if (e != hipSuccess) up_err.store((int)e);
            uploaded.store(g + 1, std::memory_order_release);  // also after an error: nobody may wait for ever
        }
    });
    // The download is pipelined by hand (copy_d2h completes one call before the next starts): the transfer of group g
    // into one pinned buffer is queued before group g - 1 is copied out of the other.
    const bool staged_down = per_group * poly_bytes <= PIN_BYTES && pin_ready(ctx, ctx->down);
    std::thread downloader([&] {
        hipError_t e = hipSetDevice(ctx->device);
        stark_ctx::PinPair& P = ctx->down;
        for (size_t g = 0; g <= groups; g++) {
            if (g < groups) {
                while (transformed.load(std::memory_order_acquire) <= g) std::this_thread::yield();
                if (up_err.load() || down_err.load()) break;  // the main thread stopped queueing: nothing more to fetch
                const size_t off = g * per_group * poly_words, len = count_of(g) * poly_bytes;
                if (e == hipSuccess) e = hipStreamWaitEvent(stream3, done[g], 0);
                if (!staged_down) {
                    if (e == hipSuccess) e = copy_d2h(ctx, limbs + off, d + off, len, stream3);
                } else {
                    if (e == hipSuccess) e = hipMemcpyAsync(P.buf[g & 1], d + off, len, hipMemcpyDeviceToHost, stream3);
                    if (e == hipSuccess) e = hipEventRecord(P.ev[g & 1], stream3);
                }
            }
            if (staged_down && g >= 1 && e == hipSuccess) {
                const size_t off = (g - 1) * per_group * poly_words, len = count_of(g - 1) * poly_bytes;
                e = hipEventSynchronize(P.ev[(g - 1) & 1]);
                if (e == hipSuccess) par_copy((char*)(limbs + off), P.buf[(g - 1) & 1], len, ctx->copy_threads, P);
            }
            if (e != hipSuccess) {
                down_err.store((int)e);
                break;
            }
        }
    });
    int rc = STARK_OK;
    for (size_t g = 0; g < groups; g++) {
        while (uploaded.load(std::memory_order_acquire) <= g) std::this_thread::yield();
        if (rc == STARK_OK && !up_err.load()) {
            try {  // nothing may unwind past the two running threads
                hipError_t e = hipStreamWaitEvent(stream2, up[g], 0);
                if (e != hipSuccess) rc = fail(ctx, STARK_ERR_HIP, "hipStreamWaitEvent: %s", hipGetErrorString(e));
                uint32_t* dg = d + g * per_group * poly_words;
                if (rc == STARK_OK) rc = run_plan(ctx, plan, dg, dg, count_of(g), stream2);
                if (rc == STARK_OK && (e = hipEventRecord(done[g], stream2)) != hipSuccess)
                    rc = fail(ctx, STARK_ERR_HIP, "hipEventRecord: %s", hipGetErrorString(e));
            } catch (...) {
                rc = STARK_ERR_NOMEM;
            }
            if (rc != STARK_OK) down_err.store((int)hipErrorUnknown);  // tells the downloader to skip the rest
        }
        transformed.store(g + 1, std::memory_order_release);
    }
    uploader.join();
    downloader.join();
    ctx->copy_threads = saved_threads;
    hipError_t e1 = hipStreamSynchronize(ctx->stream), e2 = hipStreamSynchronize(stream2), e3 = hipStreamSynchronize(stream3);
    drop_events();
    if (rc != STARK_OK) return rc;
    if (up_err.load()) return fail(ctx, STARK_ERR_HIP, "staged upload: %s", hipGetErrorString((hipError_t)up_err.load()));
    if (down_err.load()) return fail(ctx, STARK_ERR_HIP, "staged download: %s", hipGetErrorString((hipError_t)down_err.load()));
    HIP_TRY(ctx, e1);
    HIP_TRY(ctx, e2);
    HIP_TRY(ctx, e3);
    return STARK_OK;
}

// ---- column transforms of a row-major [2^log_n][cols] matrix (cols a power of two) ------------------------------
// Every column is one polynomial (element stride cols, polynomial stride 1): what stage C of a sharded transform
// needs, where the matrix is the [n1][n2 / world] block the all-to-all leaves on a rank. Adjacent columns are adjacent
// in memory, so a workgroup takes C adjacent columns of one tile position and every global access is a run of C
// elements. Up to 2^6 rows: one pass. Above: two passes, rows i = k1-major index split N = N1 * M as in build_plan, with
// the matrix seen as [N1][M * cols] (pass 1: columns of that matrix) and as [N1][M][cols] (pass 2: over M).
int build_col_plan(stark_ctx* ctx, unsigned log_n, uint64_t cols, const U256& root, bool inverse, NttPlan** out) {
    std::unique_ptr<NttPlan> plan(new NttPlan);
    plan->log_n = log_n;
    plan->root = root;
    plan->inverse = inverse;
    plan->cols = cols;
    const uint64_t N = (uint64_t)1 << log_n;
    const U256 w = inverse ? hostfe::pow_u64(root, N - 1) : root;
    const U256 scale = inverse ? hostfe::inverse(hostfe::from_u64(N)) : hostfe::from_u64(1);
    const U256 one = hostfe::from_u64(1);
    hostfe::to_limbs(scale, plan->scale);
    plan->has_scale = inverse;
    unsigned log_cols = 0;
    while (((uint64_t)1 << log_cols) < cols) log_cols++;
    const int target_threads = env_int("STARKHIP_THREADS", 512);
    int rc;
    auto fail_free = [&](int code) {
        free_plan(plan.get());
        return code;
    };
    if (log_n <= 6) {
        PassPlan p;
        p.log_t = (int)log_n;
        NttPassArgs& a = p.args;
        a.log_c = choose_log_c((int)log_n, log_cols, target_threads);
        const uint64_t C = (uint64_t)1 << a.log_c;
        a.dsh = 31;
        a.dmask = 0xFFFFFFFFu;
        a.inB = a.outB = C;
        a.inC = a.outC = 1;
        a.inS = a.outS = cols;
        a.c_limit = ~0ull;
        a.load_cfast = a.store_cfast = 1;
        a.canon_in = 1;
        a.has_scale = plan->has_scale;
        memcpy(a.scale, plan->scale, sizeof a.scale);
        uint32_t* W;
        if ((rc = make_table(ctx, plan.get(), w, N, 0, one, &W, true)) != STARK_OK) return fail_free(rc);
        a.W = W;
        set_quarter_root(a, w, (int)log_n);
        p.tiles = (unsigned)(cols >> a.log_c);
        plan->passes.push_back(p);
    } else {
        const int l1 = (int)log_n / 2, l2 = (int)log_n - l1;
        const uint64_t N1 = (uint64_t)1 << l1, M = (uint64_t)1 << l2;
        const U256 w1 = hostfe::pow_u64(w, M), w2 = hostfe::pow_u64(w, N1);
        uint32_t *W1, *W2, *TW;
        if ((rc = make_table(ctx, plan.get(), w1, N1, 0, one, &W1, true)) != STARK_OK ||
            (rc = make_table(ctx, plan.get(), w2, M, 0, one, &W2, true)) != STARK_OK ||
            (rc = make_table(ctx, plan.get(), w, N, M, scale, &TW)) != STARK_OK)
            return fail_free(rc);
        {
            PassPlan p;  // tile (j, column group): rows k1 at stride M * cols
            p.log_t = l1;
            NttPassArgs& a = p.args;
            a.log_c = choose_log_c(l1, log_cols, target_threads);
            const uint64_t C = (uint64_t)1 << a.log_c;
            a.dsh = log_cols - a.log_c;
            a.dmask = (uint32_t)((cols >> a.log_c) - 1);
            a.inA = a.outA = cols;
            a.inB = a.outB = C;
            a.inC = a.outC = 1;
            a.inS = a.outS = M * cols;
            a.twA = 1;
            a.twS = M;
            a.c_limit = ~0ull;
            a.load_cfast = a.store_cfast = 1;
            a.canon_in = 1;
            a.W = W1;
            set_quarter_root(a, w1, l1);
            a.TW = TW;
            p.tiles = (unsigned)(M * (cols >> a.log_c));
            plan->passes.push_back(p);
        }
        {
            PassPlan p;  // tile (k1, column group): over j at stride cols; output row k1 + N1 * k
            p.log_t = l2;
            NttPassArgs& a = p.args;
            a.log_c = choose_log_c(l2, log_cols, target_threads);
            const uint64_t C = (uint64_t)1 << a.log_c;
            a.dsh = log_cols - a.log_c;
            a.dmask = (uint32_t)((cols >> a.log_c) - 1);
            a.inA = M * cols;
            a.outA = cols;
            a.inB = a.outB = C;
            a.inC = a.outC = 1;
            a.inS = cols;
            a.outS = N1 * cols;
            a.c_limit = ~0ull;
            a.load_cfast = a.store_cfast = 1;
            a.W = W2;
            set_quarter_root(a, w2, l2);
            p.tiles = (unsigned)(N1 * (cols >> a.log_c));
            plan->passes.push_back(p);
        }
    }
    *out = plan.get();
    ctx->col_plans.push_back(std::move(plan));
    return STARK_OK;
}

// d_in, d_out: [2^log_n][cols] elements; the whole matrix is one launch per pass (polynomials are in the tile grid)
int run_col_plan(stark_ctx* ctx, NttPlan* plan, const uint32_t* d_in, uint32_t* d_out, hipStream_t stream) {
    const uint64_t total = plan->cols << plan->log_n;
    for (const PassPlan& p : plan->passes)
        if ((uint64_t)p.tiles >= (1ull << 31)) return fail(ctx, STARK_ERR_UNSUPPORTED, "matrix too large for one launch");
    uint32_t* tmp = nullptr;
    if (plan->passes.size() > 1) {
        int rc = ensure_buffer(ctx, &ctx->scratch, &ctx->scratch_bytes, total * ntt29_mid_bytes());
        if (rc != STARK_OK) return rc;
        tmp = (uint32_t*)ctx->scratch;
    }
    for (size_t k = 0; k < plan->passes.size(); k++) {
        const PassPlan& p = plan->passes[k];
        NttPassArgs a = p.args;
        a.in = k == 0 ? d_in : tmp;
        a.out = k + 1 == plan->passes.size() ? d_out : tmp;
        a.in_raw = k > 0;
        a.out_raw = k + 1 < plan->passes.size();
        a.ksh = 31;
        a.kmask = 0xFFFFFFFFu;
        HIP_TRY(ctx, launch_ntt29_pass(p.log_t, a, p.tiles, 1, stream));
    }
    return STARK_OK;
}

struct ShardGeom {
    unsigned l1, l2;      // n = n1 * n2, n1 = 2^l1 (input index j = j1 + n1 * j2), n2 = 2^l2 (output index k = k1 * n2 + k2)
    uint64_t n1, n2, r1, r2;  // r1 = n1 / world rows of stage A per rank, r2 = n2 / world columns of stage C per rank
    unsigned log_r2;
};

// 0 = ok, else an error already recorded in the context
int shard_geometry(stark_ctx* ctx, unsigned log_n, unsigned world, unsigned rank, ShardGeom& g) {
    if (log_n > 30) return fail(ctx, STARK_ERR_UNSUPPORTED, "log_n %u above 30", log_n);
    if (world == 0 || (world & (world - 1)) || rank >= world)
        return fail(ctx, STARK_ERR_INVALID, "world %u must be a power of two and rank %u below it", world, rank);
    g.l1 = log_n / 2;
    g.l2 = log_n - g.l1;
    g.n1 = (uint64_t)1 << g.l1;
    g.n2 = (uint64_t)1 << g.l2;
    if (g.n1 % world || g.n2 % world) return fail(ctx, STARK_ERR_INVALID, "world %u must divide n1 = 2^%u and n2 = 2^%u", world, g.l1, g.l2);
    g.r1 = g.n1 / world;
    g.r2 = g.n2 / world;
    g.log_r2 = 0;
    while (((uint64_t)1 << g.log_r2) < g.r2) g.log_r2++;
    if (g.l1 < 2) return fail(ctx, STARK_ERR_UNSUPPORTED, "sharded transforms start at 2^4 points");
    return STARK_OK;
}

}  // namespace

// Nothing may unwind across the C ABI: every entry point is a function-try-block.
#define STARK_CATCH                                              \
    catch (const std::bad_alloc&) { return STARK_ERR_NOMEM; }    \
    catch (...) { return STARK_ERR_HIP; }

extern "C" {

int stark_ntt_columns_dev(stark_ctx* ctx, const void* d_in, void* d_out, unsigned log_n, size_t cols,
                          const uint32_t root_of_unity[8], int inverse, void* stream) try {
    if (!ctx) return STARK_ERR_INVALID;
    if (!root_of_unity || (cols && (!d_in || !d_out))) return fail(ctx, STARK_ERR_INVALID, "null argument");
    if (log_n > 24) return fail(ctx, STARK_ERR_UNSUPPORTED, "log_n %u above 24", log_n);
    if (log_n < 2) return fail(ctx, STARK_ERR_UNSUPPORTED, "column transforms start at 4 rows");
    if (cols & (cols - 1)) return fail(ctx, STARK_ERR_UNSUPPORTED, "cols %zu is not a power of two", cols);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const U256 r = hostfe::canon(hostfe::from_limbs(root_of_unity));
    if (!has_order(r, log_n)) return fail(ctx, STARK_ERR_INVALID, "root of unity does not have order 2^%u", log_n);
    if (cols == 0) return STARK_OK;
    NttPlan* plan = nullptr;
    for (auto& p : ctx->col_plans)
        if (p->log_n == log_n && p->cols == cols && p->inverse == (inverse != 0) && hostfe::eq(p->root, r)) plan = p.get();
    if (!plan) {
        if (ctx->col_plans.size() >= 32) {  // tables of a column plan are small (the size of one column): a plain cap
            HIP_TRY(ctx, hipDeviceSynchronize());
            free_plan(ctx->col_plans.front().get());
            ctx->col_plans.erase(ctx->col_plans.begin());
        }
        int rc = build_col_plan(ctx, log_n, cols, r, inverse != 0, &plan);
        if (rc != STARK_OK) return rc;
    }
    return run_col_plan(ctx, plan, (const uint32_t*)d_in, (uint32_t*)d_out, stream ? (hipStream_t)stream : ctx->stream);
} STARK_CATCH

int stark_ntt_sharded_stage_a_dev(stark_ctx* ctx, const void* d_in, void* d_send, unsigned log_n, unsigned world,
                                  unsigned rank, const uint32_t root_of_unity[8], int inverse, void* stream) try {
    if (!ctx) return STARK_ERR_INVALID;
    if (!root_of_unity || !d_in || !d_send || d_in == d_send) return fail(ctx, STARK_ERR_INVALID, "null or aliased buffers");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    ShardGeom g;
    int rc = shard_geometry(ctx, log_n, world, rank, g);
    if (rc != STARK_OK) return rc;
    const U256 w = hostfe::canon(hostfe::from_limbs(root_of_unity));
    if (!has_order(w, log_n)) return fail(ctx, STARK_ERR_INVALID, "root of unity does not have order 2^%u", log_n);
    // the n2-point row transforms: the ordinary plan of root^n1
    const U256 wa = hostfe::pow_u64(w, g.n1);
    NttPlan* plan;
    if ((rc = get_plan(ctx, g.l2, wa, inverse != 0, &plan)) != STARK_OK) return rc;
    if (plan->tiny) return fail(ctx, STARK_ERR_UNSUPPORTED, "sharded transforms start at 2^4 points");
    // the last pass stores output index k1' + n1' * k; the send blocks split the index at r2, which must not cut k1'
    if (plan->passes.size() > 1 && g.r2 < plan->passes.back().args.outS)
        return fail(ctx, STARK_ERR_UNSUPPORTED, "world %u too large for a fused stage A at 2^%u points", world, log_n);
    // twiddles w^(+-(rank * r1 + a) * k2) for this rank's rows, times n2^-1 when the plan is a single inverse pass
    // (whose own scaling the table replaces)
    NttPlan* tab = nullptr;
    const uint64_t key = ((uint64_t)world << 32) | rank;
    for (auto& p : ctx->stage_a_tabs)
        if (p->log_n == log_n && p->cols == key && p->inverse == (inverse != 0) && hostfe::eq(p->root, w)) tab = p.get();
    if (!tab) {
        auto t = std::make_unique<NttPlan>();
        t->log_n = log_n;
        t->root = w;
        t->inverse = inverse != 0;
        t->cols = key;
        const U256 base = inverse ? hostfe::pow_u64(w, ((uint64_t)1 << log_n) - 1) : w;
        const U256 scale = (inverse && plan->passes.size() == 1) ? hostfe::inverse(hostfe::from_u64(g.n2)) : hostfe::from_u64(1);
        uint32_t* tw;
        if ((rc = make_table(ctx, t.get(), base, (size_t)(g.r1 * g.n2), g.n2, scale, &tw, false, (uint64_t)rank * g.r1)) != STARK_OK) {
            free_plan(t.get());
            return rc;
        }
        tab = t.get();
        // a table is as large as the rank's slice of the data: keep the four most recent ones
        if (ctx->stage_a_tabs.size() >= 4) {
            HIP_TRY(ctx, hipDeviceSynchronize());  // kernels in flight on any stream may still read the oldest
            free_plan(ctx->stage_a_tabs.front().get());
            ctx->stage_a_tabs.erase(ctx->stage_a_tabs.begin());
        }
        ctx->stage_a_tabs.push_back(std::move(t));
    }
    LastPassOverride ov;
    ov.TW = (const uint32_t*)tab->tables[0];
    ov.tw_poly = g.n2;
    ov.out_poly = g.r2;               // row a of block d starts at d * r1 * r2 + a * r2
    ov.ksh = g.log_r2;
    ov.outS2 = g.r1 * g.r2;
    return run_plan(ctx, plan, (const uint32_t*)d_in, (uint32_t*)d_send, (size_t)g.r1, stream ? (hipStream_t)stream : ctx->stream,
                    nullptr, 0, &ov);
} STARK_CATCH

int stark_ntt_sharded_stage_c_dev(stark_ctx* ctx, const void* d_recv, void* d_out, unsigned log_n, unsigned world,
                                  const uint32_t root_of_unity[8], int inverse, void* stream) try {
    if (!ctx) return STARK_ERR_INVALID;
    if (!root_of_unity || !d_recv || !d_out) return fail(ctx, STARK_ERR_INVALID, "null argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    ShardGeom g;
    int rc = shard_geometry(ctx, log_n, world, 0, g);
    if (rc != STARK_OK) return rc;
    const U256 w = hostfe::canon(hostfe::from_limbs(root_of_unity));
    if (!has_order(w, log_n)) return fail(ctx, STARK_ERR_INVALID, "root of unity does not have order 2^%u", log_n);
    uint32_t wc[8];
    hostfe::to_limbs(hostfe::pow_u64(w, g.n2), wc);  // order n1
    return stark_ntt_columns_dev(ctx, d_recv, d_out, g.l1, (size_t)g.r2, wc, inverse, stream);
} STARK_CATCH

int stark_ntt_sharded_dev(stark_ctx* ctx, void* d_x, void* d_y, unsigned log_n, unsigned world, unsigned rank,
                          const uint32_t root_of_unity[8], int inverse, stark_all_to_all_fn exchange, void* user,
                          void* stream) try {
    if (!ctx) return STARK_ERR_INVALID;
    if (world > 1 && !exchange) return fail(ctx, STARK_ERR_INVALID, "an exchange function is required when world > 1");
    void* st = stream ? stream : (void*)ctx->stream;
    int rc = stark_ntt_sharded_stage_a_dev(ctx, d_x, d_y, log_n, world, rank, root_of_unity, inverse, st);
    if (rc != STARK_OK) return rc;
    // An exchange function runs whenever one is given, also on a single rank (a one-rank communicator then moves the
    // one block to itself); without one the send blocks of the single rank are the received ones.
    const void* recv = d_y;
    if (exchange) {
        const size_t block = (((size_t)1 << log_n) / world / world) * 32;
        if (exchange(user, d_y, d_x, block, st) != 0) return fail(ctx, STARK_ERR_HIP, "the exchange function reported a failure");
        recv = d_x;
    }
    // stage C is out of place: from the received blocks into the other buffer
    return stark_ntt_sharded_stage_c_dev(ctx, recv, exchange ? d_y : d_x, log_n, world, root_of_unity, inverse, st);
} STARK_CATCH


int stark_version(void) { return 100; }

int stark_ctx_create(int device, stark_ctx** out) try {
    if (!out) return STARK_ERR_INVALID;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count) return STARK_ERR_HIP;
    if (hipSetDevice(device) != hipSuccess) return STARK_ERR_HIP;
    stark_ctx* ctx = new stark_ctx;
    ctx->device = device;
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
        delete ctx;
        return STARK_ERR_HIP;
    }
    // device address of the static scratch on this device (each device holds its own copy at its own address)
    if (hipGetSymbolAddress(&ctx->static_scratch, HIP_SYMBOL(g_static_scratch)) != hipSuccess || !ctx->static_scratch) {
        (void)hipStreamDestroy(ctx->stream);
        delete ctx;
        return STARK_ERR_HIP;
    }
    if (hipDeviceGetAttribute(&ctx->n_cu, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || ctx->n_cu <= 0) {
        (void)hipStreamDestroy(ctx->stream);
        delete ctx;
        return STARK_ERR_HIP;
    }
    if (hipMalloc((void**)&ctx->ticket_pool, 256 * stark_ctx::TICKET_POOL) != hipSuccess) {
        (void)hipStreamDestroy(ctx->stream);
        delete ctx;
        return STARK_ERR_NOMEM;
    }
    const char* cb = getenv("STARKHIP_CHUNK_MB");
    if (cb && atoi(cb) > 0) ctx->chunk_bytes = (size_t)atoi(cb) << 20;
    const char* tb = getenv("STARKHIP_TABLE_MB");
    if (tb && atoi(tb) > 0) ctx->table_budget = (size_t)atoi(tb) << 20;
    *out = ctx;
    return STARK_OK;
} STARK_CATCH

void stark_ctx_destroy(stark_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (hipStream_t a : ctx->aux)
        if (a) (void)hipStreamSynchronize(a);
    for (auto& a : ctx->attach) a.reset();
    for (auto& p : ctx->plans) free_plan(p.get());
    for (auto& p : ctx->shard_tables) free_plan(p.get());
    for (auto& p : ctx->col_plans) free_plan(p.get());
    for (auto& p : ctx->stage_a_tabs) free_plan(p.get());
    if (ctx->scratch) (void)hipFree(ctx->scratch);
    if (ctx->io) (void)hipFree(ctx->io);
    for (size_t i = stark_ctx::TICKET_POOL; i < ctx->tickets.size(); i++) (void)hipFree(ctx->tickets[i].second);
    if (ctx->ticket_pool) (void)hipFree(ctx->ticket_pool);
    for (stark_ctx::PinPair* P : {&ctx->up, &ctx->down})
        for (int i = 0; i < 2; i++) {
            if (P->busy[i]) (void)hipEventSynchronize(P->ev[i]);  // a transfer on a caller's stream
            if (P->buf[i]) (void)hipHostFree(P->buf[i]);
            if (P->ev[i]) (void)hipEventDestroy(P->ev[i]);
        }
    for (hipStream_t a : ctx->aux)
        if (a) (void)hipStreamDestroy(a);
    (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

const char* stark_last_error(const stark_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

void* stark_ctx_stream(stark_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

int stark_ctx_synchronize(stark_ctx* ctx) try {
    if (!ctx) return STARK_ERR_INVALID;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return STARK_OK;
} STARK_CATCH

int stark_twiddle_mul_dev(stark_ctx* ctx, void* d_data, size_t rows, size_t cols, uint64_t row0, uint64_t col0,
                          unsigned log_order, const uint32_t root_of_unity[8], int inverse, void* stream) try {
    if (!ctx) return STARK_ERR_INVALID;
    if (!root_of_unity || (rows && cols && !d_data)) return fail(ctx, STARK_ERR_INVALID, "null argument");
    if (log_order > 30) return fail(ctx, STARK_ERR_UNSUPPORTED, "log_order %u above 30", log_order);
    if (rows && cols > (~(size_t)0) / rows) return fail(ctx, STARK_ERR_INVALID, "rows * cols overflows");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    U256 r = hostfe::canon(hostfe::from_limbs(root_of_unity));
    if (!has_order(r, log_order))
        return fail(ctx, STARK_ERR_INVALID, "root of unity does not have order 2^%u", log_order);
    NttPlan* tab = nullptr;
    for (auto& p : ctx->shard_tables)
        if (p->log_n == log_order && p->inverse == (inverse != 0) && hostfe::eq(p->root, r)) tab = p.get();
    const unsigned lo_bits = (log_order + 1) / 2, hi_bits = log_order - lo_bits;
    if (!tab) {
        auto plan = std::make_unique<NttPlan>();
        plan->log_n = log_order;
        plan->root = r;
        plan->inverse = inverse != 0;
        const U256 w = inverse ? hostfe::pow_u64(r, ((uint64_t)1 << log_order) - 1) : r;
        const U256 one = hostfe::from_u64(1);
        uint32_t *lo, *hi;
        int rc;
        if ((rc = make_table(ctx, plan.get(), w, (size_t)1 << lo_bits, 0, one, &lo)) != STARK_OK ||
            (rc = make_table(ctx, plan.get(), hostfe::pow_u64(w, (uint64_t)1 << lo_bits), (size_t)1 << hi_bits, 0, one,
                             &hi)) != STARK_OK) {
            free_plan(plan.get());
            return rc;
        }
        tab = plan.get();
        ctx->shard_tables.push_back(std::move(plan));
    }
    HIP_TRY(ctx, launch_twiddle_mul((uint32_t*)d_data, rows, cols, row0, col0, log_order, lo_bits,
                                    (const uint32_t*)tab->tables[0], (const uint32_t*)tab->tables[1],
                                    stream ? (hipStream_t)stream : ctx->stream));
    return STARK_OK;
} STARK_CATCH

int stark_permute3_dev(stark_ctx* ctx, const void* d_in, void* d_out, size_t a, size_t b, size_t chunk, void* stream) try {
    if (!ctx) return STARK_ERR_INVALID;
    if (a && b && chunk && (!d_in || !d_out || d_in == d_out)) return fail(ctx, STARK_ERR_INVALID, "null or aliased buffers");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, launch_permute3(d_in, d_out, a, b, chunk, stream ? (hipStream_t)stream : ctx->stream));
    return STARK_OK;
} STARK_CATCH

int stark_ntt_batch_dev(stark_ctx* ctx, const void* d_in, void* d_out, unsigned log_n, size_t batch,
                        const uint32_t root_of_unity[8], int inverse, void* stream) try {
    if (!ctx) return STARK_ERR_INVALID;
    if (!root_of_unity || (batch && (!d_in || !d_out))) return fail(ctx, STARK_ERR_INVALID, "null argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    return ntt_common(ctx, d_in, d_out, log_n, batch, root_of_unity, inverse, stream ? (hipStream_t)stream : ctx->stream);
} STARK_CATCH

int stark_ntt_batch_dev_static(stark_ctx* ctx, const void* d_in, void* d_out, unsigned log_n, size_t batch,
                               const uint32_t root_of_unity[8], int inverse, void* stream) try {
    if (!ctx) return STARK_ERR_INVALID;
    if (!root_of_unity || (batch && (!d_in || !d_out))) return fail(ctx, STARK_ERR_INVALID, "null argument");
    if (log_n > 30) return fail(ctx, STARK_ERR_UNSUPPORTED, "log_n %u above 30", log_n);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const U256 r = hostfe::canon(hostfe::from_limbs(root_of_unity));
    if (!has_order(r, log_n)) return fail(ctx, STARK_ERR_INVALID, "root of unity does not have order 2^%u", log_n);
    // tables are never built here: only a plan an earlier stark_ntt_batch_dev call left in the context is used
    NttPlan* plan = nullptr;
    for (auto& p : ctx->plans)
        if (p->log_n == log_n && p->inverse == (inverse != 0) && hostfe::eq(p->root, r)) plan = p.get();
    if (!plan)
        return fail(ctx, STARK_ERR_INVALID,
                    "no twiddle tables for this (log_n, root, direction) yet: call stark_ntt_batch_dev once (batch 0 is enough)");
    plan->last_use = ++ctx->use_counter;
    return run_plan(ctx, plan, (const uint32_t*)d_in, (uint32_t*)d_out, batch, stream ? (hipStream_t)stream : ctx->stream,
                    ctx->static_scratch, STARK_STATIC_SCRATCH_BYTES);
} STARK_CATCH

int stark_ntt_batch(stark_ctx* ctx, uint32_t* limbs, unsigned log_n, size_t batch, const uint32_t root_of_unity[8],
                    int inverse) try {
    if (!ctx) return STARK_ERR_INVALID;
    if (!root_of_unity || (batch && !limbs)) return fail(ctx, STARK_ERR_INVALID, "null argument");
    if (log_n > 30) return fail(ctx, STARK_ERR_UNSUPPORTED, "log_n %u above 30", log_n);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    // validate before touching the device so that a rejected call writes nothing
    U256 r = hostfe::canon(hostfe::from_limbs(root_of_unity));
    if (!has_order(r, log_n)) return fail(ctx, STARK_ERR_INVALID, "root of unity does not have order 2^%u", log_n);
    if (batch == 0) return STARK_OK;
    size_t bytes = (batch << log_n) * 32;
    int rc = ensure_buffer(ctx, &ctx->io, &ctx->io_bytes, bytes);
    if (rc != STARK_OK) return rc;
    {
        // several staging buffers' worth of polynomials: upload, transform and download overlap group by group
        const bool pipelined = env_int("STARKHIP_HOST_PIPELINE", 1) != 0;
        const size_t poly_bytes = (size_t)32 << log_n;
        const size_t per_group = poly_bytes >= PIN_BYTES / 2 ? 1 : PIN_BYTES / 2 / poly_bytes;  // half a staging buffer
        if (pipelined && batch > per_group && bytes >= 2 * STAGE_MIN) {
            NttPlan* plan;
            if ((rc = get_plan(ctx, log_n, r, inverse != 0, &plan)) != STARK_OK) return rc;
            return ntt_host_pipelined(ctx, limbs, log_n, batch, per_group, plan);
        }
    }
    HIP_TRY(ctx, copy_h2d(ctx, ctx->io, limbs, bytes, ctx->stream));
    rc = ntt_common(ctx, ctx->io, ctx->io, log_n, batch, root_of_unity, inverse, ctx->stream);
    if (rc != STARK_OK) return rc;
    HIP_TRY(ctx, copy_d2h(ctx, limbs, ctx->io, bytes, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return STARK_OK;
} STARK_CATCH

static int fft_host(stark_ctx* ctx, const uint32_t* v, size_t n, const uint32_t root[8], const uint32_t modulus[8],
                    uint32_t* out, int inverse) {
    if (!ctx) return STARK_ERR_INVALID;
    if (!v || !root || !modulus || !out) return fail(ctx, STARK_ERR_INVALID, "null argument");
    if (!hostfe::eq(hostfe::from_limbs(modulus), hostfe::MODULUS))
        return fail(ctx, STARK_ERR_UNSUPPORTED, "modulus is not 2^256 - 351*2^32 + 1");
    if (n == 0 || (n & (n - 1))) return fail(ctx, STARK_ERR_INVALID, "length %zu is not a power of two", n);
    unsigned log_n = 0;
    while (((size_t)1 << log_n) < n) log_n++;
    if (out != v) memmove(out, v, n * 32);
    // stark_ntt_batch validates the root before it writes; on rejection `out` holds the input copy
    return stark_ntt_batch(ctx, out, log_n, 1, root, inverse);
}

int stark_fft_inv(stark_ctx* ctx, const uint32_t* v, size_t n, const uint32_t root_of_unity[8],
                  const uint32_t modulus[8], uint32_t* out) try {
    return fft_host(ctx, v, n, root_of_unity, modulus, out, 1);
} STARK_CATCH

int stark_fft(stark_ctx* ctx, const uint32_t* v, size_t n, const uint32_t root_of_unity[8], const uint32_t modulus[8],
              uint32_t* out) try {
    return fft_host(ctx, v, n, root_of_unity, modulus, out, 0);
} STARK_CATCH

int stark_merkle_verify_batch_dev(stark_ctx* ctx, const void* d_values, const void* d_siblings, uint32_t value_size,
                                  const void* d_witnesses, uint32_t num_witnesses, const void* d_indices,
                                  const void* d_roots, const void* d_root_ids, size_t n, void* d_accept,
                                  void* d_digests, void* stream) try {
    if (!ctx) return STARK_ERR_INVALID;
    if (num_witnesses < 1 || num_witnesses > 30)
        return fail(ctx, STARK_ERR_INVALID, "num_witnesses %u outside 1..30", num_witnesses);
    if (value_size < 32) return fail(ctx, STARK_ERR_INVALID, "value_size %u below 32", value_size);
    if (n == 0) return STARK_OK;
    if (!d_values || !d_siblings || !d_witnesses || !d_indices || !d_roots || !d_accept)
        return fail(ctx, STARK_ERR_INVALID, "null argument");
    if (n > ((size_t)1 << 31) * 256) return fail(ctx, STARK_ERR_UNSUPPORTED, "too many branches in one call");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, launch_merkle_verify((const uint8_t*)d_values, (const uint8_t*)d_siblings, value_size,
                                      (const uint8_t*)d_witnesses, num_witnesses, (const uint32_t*)d_indices,
                                      (const uint8_t*)d_roots, (const uint32_t*)d_root_ids, n, (uint8_t*)d_accept,
                                      (uint8_t*)d_digests, stream ? (hipStream_t)stream : ctx->stream));
    return STARK_OK;
} STARK_CATCH

int stark_merkle_verify_batch(stark_ctx* ctx, const uint8_t* values, const uint8_t* siblings, uint32_t value_size,
                              const uint8_t* witnesses, uint32_t num_witnesses, const uint32_t* indices,
                              const uint8_t* roots, uint32_t num_roots, const uint32_t* root_ids, size_t n,
                              uint8_t* accept, uint8_t* digests) try {
    if (!ctx) return STARK_ERR_INVALID;
    if (num_witnesses < 1 || num_witnesses > 30)
        return fail(ctx, STARK_ERR_INVALID, "num_witnesses %u outside 1..30", num_witnesses);
    if (value_size < 32) return fail(ctx, STARK_ERR_INVALID, "value_size %u below 32", value_size);
    if (n == 0) return STARK_OK;
    if (!values || !siblings || !witnesses || !indices || !roots || !accept || num_roots == 0)
        return fail(ctx, STARK_ERR_INVALID, "null argument");
    if (root_ids)
        for (size_t i = 0; i < n; i++)
            if (root_ids[i] >= num_roots) return fail(ctx, STARK_ERR_INVALID, "root id %u out of range", root_ids[i]);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    // one staging allocation, every section 256-byte aligned
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t sz_val = al(n * value_size), sz_wit = al((size_t)num_witnesses * n * 32), sz_idx = al(n * 4);
    const size_t sz_roots = al((size_t)num_roots * 32), sz_acc = al(n), sz_dig = al(n * 32);
    size_t total = 2 * sz_val + sz_wit + 2 * sz_idx + sz_roots + sz_acc + sz_dig;
    int rc = ensure_buffer(ctx, &ctx->io, &ctx->io_bytes, total);
    if (rc != STARK_OK) return rc;
    uint8_t* base = (uint8_t*)ctx->io;
    uint8_t *d_val = base, *d_sib = d_val + sz_val, *d_wit = d_sib + sz_val, *d_idx = d_wit + sz_wit;
    uint8_t *d_rid = d_idx + sz_idx, *d_roots = d_rid + sz_idx, *d_acc = d_roots + sz_roots, *d_dig = d_acc + sz_acc;
    hipStream_t s = ctx->stream;
    HIP_TRY(ctx, copy_h2d(ctx, d_val, values, n * value_size, s));
    HIP_TRY(ctx, copy_h2d(ctx, d_sib, siblings, n * value_size, s));
    HIP_TRY(ctx, copy_h2d(ctx, d_wit, witnesses, (size_t)num_witnesses * n * 32, s));
    HIP_TRY(ctx, hipMemcpyAsync(d_idx, indices, n * 4, hipMemcpyHostToDevice, s));
    if (root_ids) HIP_TRY(ctx, hipMemcpyAsync(d_rid, root_ids, n * 4, hipMemcpyHostToDevice, s));
    HIP_TRY(ctx, hipMemcpyAsync(d_roots, roots, (size_t)num_roots * 32, hipMemcpyHostToDevice, s));
    rc = stark_merkle_verify_batch_dev(ctx, d_val, d_sib, value_size, d_wit, num_witnesses, d_idx, d_roots,
                                       root_ids ? d_rid : nullptr, n, d_acc, digests ? d_dig : nullptr, s);
    if (rc != STARK_OK) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(accept, d_acc, n, hipMemcpyDeviceToHost, s));
    if (digests) HIP_TRY(ctx, hipMemcpyAsync(digests, d_dig, n * 32, hipMemcpyDeviceToHost, s));
    HIP_TRY(ctx, hipStreamSynchronize(s));
    return STARK_OK;
} STARK_CATCH

int stark_merkle_build_dev(stark_ctx* ctx, void* d_nodes, unsigned log_leaves, void* stream) try {
    if (!ctx) return STARK_ERR_INVALID;
    if (!d_nodes) return fail(ctx, STARK_ERR_INVALID, "null argument");
    if (log_leaves > 31) return fail(ctx, STARK_ERR_UNSUPPORTED, "log_leaves %u above 31", log_leaves);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, launch_merkle_build((uint8_t*)d_nodes, log_leaves, stream ? (hipStream_t)stream : ctx->stream));
    return STARK_OK;
} STARK_CATCH

int stark_merkle_build(stark_ctx* ctx, const uint8_t* leaves, unsigned log_leaves, uint8_t* nodes) try {
    if (!ctx) return STARK_ERR_INVALID;
    if (!leaves || !nodes) return fail(ctx, STARK_ERR_INVALID, "null argument");
    if (log_leaves > 31) return fail(ctx, STARK_ERR_UNSUPPORTED, "log_leaves %u above 31", log_leaves);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t L = (size_t)1 << log_leaves;
    int rc = ensure_buffer(ctx, &ctx->io, &ctx->io_bytes, 64 * L);
    if (rc != STARK_OK) return rc;
    uint8_t* d = (uint8_t*)ctx->io;
    HIP_TRY(ctx, hipMemsetAsync(d, 0, 32, ctx->stream));
    HIP_TRY(ctx, copy_h2d(ctx, d + 32 * L, leaves, 32 * L, ctx->stream));
    HIP_TRY(ctx, launch_merkle_build(d, log_leaves, ctx->stream));
    HIP_TRY(ctx, copy_d2h(ctx, nodes, d, 64 * L, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return STARK_OK;
} STARK_CATCH

int stark_pseudorandom_indices_dev(stark_ctx* ctx, const void* d_seeds, size_t n, uint32_t count, uint32_t modulus,
                                   uint32_t exclude_multiples_of, void* d_out, void* stream) try {
    if (!ctx) return STARK_ERR_INVALID;
    if (n && count && (!d_seeds || !d_out)) return fail(ctx, STARK_ERR_INVALID, "null argument");
    if (exclude_multiples_of == 1) return fail(ctx, STARK_ERR_INVALID, "exclude_multiples_of = 1 divides by zero in the reference");
    const uint32_t real = exclude_multiples_of ? (uint32_t)(modulus * (exclude_multiples_of - 1) / exclude_multiples_of) : modulus;
    if (real == 0) return fail(ctx, STARK_ERR_INVALID, "modulus reduces to zero (the reference divides by zero)");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, launch_sample_indices((const uint8_t*)d_seeds, n, count, real, exclude_multiples_of, (uint32_t*)d_out,
                                       stream ? (hipStream_t)stream : ctx->stream));
    return STARK_OK;
} STARK_CATCH

int stark_pseudorandom_indices(stark_ctx* ctx, const uint8_t* seeds, size_t n, uint32_t count, uint32_t modulus,
                               uint32_t exclude_multiples_of, uint32_t* out) try {
    if (!ctx) return STARK_ERR_INVALID;
    if (n && count && (!seeds || !out)) return fail(ctx, STARK_ERR_INVALID, "null argument");
    if (n == 0 || count == 0) return STARK_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t in_bytes = (32 * n + 255) & ~(size_t)255, out_bytes = 4 * n * (size_t)count;
    int rc = ensure_buffer(ctx, &ctx->io, &ctx->io_bytes, in_bytes + out_bytes);
    if (rc != STARK_OK) return rc;
    uint8_t* d = (uint8_t*)ctx->io;
    HIP_TRY(ctx, hipMemcpyAsync(d, seeds, 32 * n, hipMemcpyHostToDevice, ctx->stream));
    rc = stark_pseudorandom_indices_dev(ctx, d, n, count, modulus, exclude_multiples_of, d + in_bytes, ctx->stream);
    if (rc != STARK_OK) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(out, d + in_bytes, out_bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return STARK_OK;
} STARK_CATCH

// Pure host helpers: no context, no device.
void stark_be32_to_limbs(const uint8_t* be, size_t n, uint32_t* limbs) {
    for (size_t i = 0; i < n; i++)
        for (int l = 0; l < 8; l++) {
            const uint8_t* b = be + 32 * i + 4 * (7 - l);
            limbs[8 * i + l] = ((uint32_t)b[0] << 24) | ((uint32_t)b[1] << 16) | ((uint32_t)b[2] << 8) | b[3];
        }
}

void stark_limbs_to_be32(const uint32_t* limbs, size_t n, uint8_t* be) {
    for (size_t i = 0; i < n; i++)
        for (int l = 0; l < 8; l++) {
            const uint32_t v = limbs[8 * i + l];
            uint8_t* b = be + 32 * i + 4 * (7 - l);
            b[0] = (uint8_t)(v >> 24);
            b[1] = (uint8_t)(v >> 16);
            b[2] = (uint8_t)(v >> 8);
            b[3] = (uint8_t)v;
        }
}

int stark_field_op(stark_ctx* ctx, int op, const uint32_t* a, const uint32_t* b, uint32_t* r, size_t n) try {
    if (!ctx) return STARK_ERR_INVALID;
    if (op < 0 || op > 7) return fail(ctx, STARK_ERR_INVALID, "unknown op %d", op);
    if (n == 0) return STARK_OK;
    if (!a || !r || (op != 3 && !b)) return fail(ctx, STARK_ERR_INVALID, "null argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc = ensure_buffer(ctx, &ctx->io, &ctx->io_bytes, 3 * n * 32);
    if (rc != STARK_OK) return rc;
    uint32_t *da = (uint32_t*)ctx->io, *db = da + 8 * n, *dr = db + 8 * n;
    hipStream_t s = ctx->stream;
    HIP_TRY(ctx, hipMemcpyAsync(da, a, n * 32, hipMemcpyHostToDevice, s));
    if (op != 3) HIP_TRY(ctx, hipMemcpyAsync(db, b, n * 32, hipMemcpyHostToDevice, s));
    HIP_TRY(ctx, launch_field_op(op, da, db, dr, n, s));
    HIP_TRY(ctx, hipMemcpyAsync(r, dr, n * 32, hipMemcpyDeviceToHost, s));
    HIP_TRY(ctx, hipStreamSynchronize(s));
    return STARK_OK;
} STARK_CATCH

int stark_fill_random_field_dev(stark_ctx* ctx, void* d_out, size_t n, uint64_t seed, void* stream) try {
    if (!ctx) return STARK_ERR_INVALID;
    if (n && !d_out) return fail(ctx, STARK_ERR_INVALID, "null argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, launch_fill_random_field((uint32_t*)d_out, n, seed, stream ? (hipStream_t)stream : ctx->stream));
    return STARK_OK;
} STARK_CATCH

}  // extern "C"
