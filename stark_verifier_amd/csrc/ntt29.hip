// Number-theoretic transform over p = 2^256 - 351*2^32 + 1 for gfx950: the tile-transform pass kernel.
//
// Computes what the reference's _fft does (reference src/fft.rs:37-61; base case simple_ft :18-35): natural-order
// in, natural-order out, out[i] = sum_j v[j] * w^(i*j) mod p, every output the canonical residue. The reference
// recurses on heap BigInts; here a transform of 2^log_n points is one to three passes, each pass a grid of
// workgroups that transform tiles of up to 4096 points held in LDS:
//
//   * inside a tile: Stockham autosort, radix-4 per lane (two butterfly levels per LDS round trip, preceded by one
//     radix-2 level when log2(tile) is odd). A lane keeps its four elements in registers.
//   * across tiles (domains larger than LDS): the four-step split n = n1*m. Pass 1 transforms columns (stride m) and
//     multiplies output (k1, j) by w^(k1*j) from a table; the remaining passes transform the rows and store
//     transposed, out[k1 + n1*k]. Workgroups take several adjacent columns / rows so every global access is a run of
//     contiguous elements.
//
// The inverse transform (reference fft_inv, src/fft.rs:64-86) runs the same kernel with the inverse root; its n^-1
// factor is folded into the inter-pass twiddle table (or, for single-pass sizes, applied at the final store).
//
// Representation inside the kernel (29-bit limbs, "carry-free", field29.cuh):
//   * on load an element (8 x 32-bit limbs in HBM) is unpacked to nine 29-bit limbs; in LDS an
//     element is nine adjacent dwords;
//   * butterfly additions and subtractions are limb-wise v_add_u32 / v_sub_u32 with no carry
//     chain, the products are carry-free v_mad_u64_u32 columns (field29.cuh);
//   * between passes of a multi-pass transform the intermediate stays unpacked (nine dwords per
//     element, back to back), so only the first load and the last store convert; the last
//     store reduces to the canonical residue and repacks.
// Tile twiddles W are stored unpacked (12 dwords per entry); the large inter-pass table TW
// stays packed (8 dwords) and is unpacked on load.
#include "field29.cuh"
#include "internal.hpp"
#include <stdlib.h>
#include <atomic>
#include <type_traits>

namespace starkhip {

// Slot swizzle inside a sub-image: XOR bits 2..6 into bits 0..4. With it both the unit-stride
// reads (element j + q*T/4) and the strided Stockham writes (stride 4*Ns) of every round land on
// distinct slots mod 32 per 32-lane group; without it the writes are up to 4-way conflicted.
__device__ __forceinline__ uint32_t swz(uint32_t p) { return p ^ ((p >> 2) & 31u); }

// An element's nine limbs are adjacent in LDS (slot * 9 + limb): one address per element, the
// limbs at immediate offsets, two per ds_read2/ds_write2. 9 is odd, so slots that differ mod
// 64 stay on different banks.
__device__ __forceinline__ void lds_put29(uint32_t* lds, uint32_t off, const F29& v) {
    uint32_t* q = lds + off * 9;
#pragma unroll
    for (int l = 0; l < 9; l++) q[l] = v.l[l];
}

__device__ __forceinline__ F29 lds_get29(const uint32_t* lds, uint32_t off) {
    const uint32_t* q = lds + off * 9;
    F29 v;
#pragma unroll
    for (int l = 0; l < 9; l++) v.l[l] = q[l];
    return v;
}

// unpacked table entry in global memory: 12 dwords (limbs 0..8, three unused), 16-byte aligned
__device__ __forceinline__ F29 f29_load_raw(const uint32_t* p) {
    const uint4* q = reinterpret_cast<const uint4*>(p);
    uint4 a = q[0], b = q[1];
    F29 r;
    r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w;
    r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
    r.l[8] = p[8];  // one dword: the three padding dwords are never fetched into registers
    return r;
}

// unpacked element of the multi-pass intermediate: nine dwords, elements back to back (36 bytes; a padded 48-byte
// element costs 15 % more memory traffic and is no faster, measured)
constexpr uint32_t RAW_W = 9;
struct __attribute__((aligned(4))) U3 {
    uint32_t x, y, z;
};

// three 12-byte accesses: an element starts on a multiple of 4 bytes only
__device__ __forceinline__ F29 f29_load_mid(const uint32_t* p) {
    const U3* q = reinterpret_cast<const U3*>(p);
    const U3 a = q[0], b = q[1], c = q[2];
    F29 r;
    r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z;
    r.l[3] = b.x; r.l[4] = b.y; r.l[5] = b.z;
    r.l[6] = c.x; r.l[7] = c.y; r.l[8] = c.z;
    return r;
}

__device__ __forceinline__ void f29_store_mid(uint32_t* p, const F29& v) {
    U3* q = reinterpret_cast<U3*>(p);
    q[0] = U3{v.l[0], v.l[1], v.l[2]};
    q[1] = U3{v.l[3], v.l[4], v.l[5]};
    q[2] = U3{v.l[6], v.l[7], v.l[8]};
}

__device__ __forceinline__ void f29_store_raw(uint32_t* p, const F29& v) {
    uint4* q = reinterpret_cast<uint4*>(p);
    q[0] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
    q[1] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
    q[2] = make_uint4(v.l[8], 0u, 0u, 0u);
}

// ---- LDS image of a tile -------------------------------------------------------------------
// A tile of T = 2^LOGT points is worked on by T/4 lanes (radix-4 per lane) = W = T/256 wavefronts
// (one when T <= 256). Position p of the tile lives in sub-image (p mod W) at offset p / W, so
// that from the round with sub-transform length Ns >= W onwards every wavefront reads and
// writes only its own sub-image: the lane that runs butterfly j of such a round is lane j / W
// of wavefront j mod W, and the Stockham write index keeps the low log2(W) bits of j in place.
// Those rounds exchange data through LDS without any workgroup barrier; only the rounds before
// them (and a last round that switches to a store-friendly lane order) need s_barrier.
template <int LOGT>
struct TileGeom {
    static constexpr int LW = LOGT > 8 ? LOGT - 8 : 0;  // log2(wavefronts per tile)
    static constexpr uint32_t W = 1u << LW;
    static constexpr uint32_t B = (1u << LOGT) >> LW;   // sub-image length (<= 256)
    static constexpr uint32_t SB = B + (LW ? 8u : 0u);  // sub-image stride: +8 keeps cross-wave rounds off one bank
    static constexpr uint32_t ROW = W * SB;             // element slots of one tile
};

__host__ __device__ inline uint32_t tile_pad29(uint32_t log_c) {
    if (log_c == 0) return 0;
    return log_c >= 5 ? 1u : (32u >> log_c);
}

size_t ntt29_mid_bytes() { return 4 * RAW_W; }

size_t ntt29_pass_lds_bytes(int log_t, int log_c) {
    const int lw = log_t > 8 ? log_t - 8 : 0;
    size_t row = (((size_t)1 << log_t) >> lw) + (lw ? 8 : 0);
    row = (row << lw) + tile_pad29((uint32_t)log_c);
    return 36 * row << log_c;
}

struct Quad29 {
    F29 a, b, c, d;
};

#ifdef STARKHIP_STAMPS
// timing instrumentation (debug builds only): lane 0 of every wavefront records the 100 MHz wall clock at entry (0) and,
// per round r, once its inputs are in registers (1 + 3r, after waiting for them), once its outputs are computed and
// handed to LDS or memory (2 + 3r) and once the exchange is synchronised or the stores have drained (3 + 3r); slots 16
// and 17 hold the hardware's wavefront placement (HW_ID, XCC_ID)
__device__ unsigned long long g_stamps[2][8192][8][24];
#define STAMP(i)                                                                                              \
    do {                                                                                                      \
        if ((threadIdx.x & 63u) == 0 && item < 8192)                                                          \
            g_stamps[in_raw ? 1 : 0][item][(threadIdx.x >> 6) & 7u][i] = wall_clock64();                      \
    } while (0)
#define STAMPW(i)                           \
    do {                                    \
        __builtin_amdgcn_s_waitcnt(0);      \
        STAMP(i);                           \
    } while (0)
#else
#define STAMP(i) do {} while (0)
#define STAMPW(i) do {} while (0)
#endif

// KIND: instantiations for the shapes that carry the load, with their flags known at compile time (fewer scalar
// registers and branches: the general kernel sits at the scalar-register limit, and what spills costs every launch 1 %).
//   0 general: every flag read from the arguments
//   1 first pass of several from packed input: unpacked output, an inter-pass table, tile index fastest on both sides
//   2 last pass from an unpacked intermediate (tiled or not) to packed output, no output table, tile index fastest on
//     both sides
//   3 single pass, inverse, the output factor riding on the last round (Wlast); otherwise as 6
//   4 middle pass of three: unpacked on both sides, an inter-pass table, tile index fastest on both sides
//   5 last pass of three: rows of the unpacked intermediate (element index fastest) to packed output, no output table
//   6 single pass, no output factor: packed on both sides, element index fastest, no table
// Kinds 1, 2, 4, 5 (the passes of multi-pass plans) also know that every tile of the grid exists and how many tiles a
// workgroup takes; all but 0 know that the output index is not split.
//   7, 8 "streaming" first and last pass of two at 1024- or 256-point tiles: a fixed grid of resident workgroups, each walking
//     through work items blockIdx.x, blockIdx.x + gridDim.x, ...; the next item's input is fetched straight into LDS
//     (global_load_lds, no registers) while the last round of the current item computes and stores, so that no
//     wavefront waits for global memory at the start of an item (see "landing" below). That costs a workgroup barrier
//     after the last round's LDS reads (the image must be dead before the fetch is queued) and, in kind 8, two around
//     the reads of round 0. Kind 7 leaves the inter-pass factors to kind 8, which multiplies by them as it reads its
//     input: a last round then issues no load after the fetch (loads return in order, so a table entry requested
//     behind the fetch would wait for it).
constexpr int KIND_GENERAL = 0, KIND_FIRST = 1, KIND_LAST = 2, KIND_FOLD = 3, KIND_MID = 4, KIND_LAST3 = 5, KIND_SINGLE = 6,
              KIND_SFIRST = 7, KIND_SLAST = 8;

typedef __attribute__((address_space(3))) void* lds_dma_dst_t;
typedef __attribute__((address_space(1))) const void* lds_dma_src_t;

template <int LOGT, int KIND>
__global__ void __launch_bounds__(1024) ntt29_pass_kernel(const NttPassArgs a) {
    constexpr bool STREAM = KIND == KIND_SFIRST || KIND == KIND_SLAST;
    static_assert(!STREAM || LOGT == 10 || LOGT == 8, "the landing layouts of the streaming kinds are laid out for 1024- and 256-point tiles");
    constexpr bool MULTI = KIND == KIND_FIRST || KIND == KIND_LAST || KIND == KIND_MID || KIND == KIND_LAST3 || STREAM;
    constexpr bool ONE = KIND == KIND_FOLD || KIND == KIND_SINGLE;
    constexpr bool SPECIAL = MULTI || ONE;
    constexpr bool FOLD = KIND == KIND_FOLD;
    // The multiplier's wide-limb form (field29.cuh) is a tenth shorter, but in the high columns it holds about five more
    // registers. Where four elements and prefetched table entries are alive (the twiddle multiplications of a round, the
    // inter-pass table) the kinds that already fill the 128 registers of four wavefronts per SIMD (passes that
    // prefetch an inter-pass table, the folded inverse) would spill with it (the compiler's resource report) and use
    // it in the low columns only, where it costs none; which instantiations those are is read off the compiler's
    // resource report (with the scheduling option this file is built with, see the Makefile): the folded inverse, first
    // and middle passes at 64- and 256-point tiles, some sizes of the general kernel. The multiplication by the 4th
    // root of unity comes when the twiddles are dead and has the registers in every kind.
#ifdef STARKHIP_MULW
    constexpr int MULW_TW = STARKHIP_MULW & 3, MULW = MULW_TW, MULW_U = STARKHIP_MULW >> 2;
#else
    constexpr bool tight = KIND == KIND_FOLD || ((KIND == KIND_FIRST || KIND == KIND_MID) && (LOGT <= 6 || LOGT == 8)) ||
                           (KIND == KIND_GENERAL && (LOGT == 5 || LOGT == 7 || LOGT == 10 || LOGT == 11));
    constexpr int MULW_TW = tight ? 2 : 3, MULW = MULW_TW, MULW_U = 3;
#endif
    const bool in_raw = (KIND == KIND_FIRST || KIND == KIND_SFIRST || ONE) ? false : MULTI ? true : a.in_raw != 0;
    const bool out_raw = (KIND == KIND_FIRST || KIND == KIND_MID || KIND == KIND_SFIRST) ? true : SPECIAL ? false : a.out_raw != 0;
    const bool load_cfast = (KIND == KIND_LAST3 || ONE) ? false : MULTI ? true : a.load_cfast != 0;
    const bool store_cfast = ONE ? false : MULTI ? true : a.store_cfast != 0;
    const bool has_tw = (KIND == KIND_FIRST || KIND == KIND_MID) ? true : SPECIAL ? false : a.TW != nullptr;
    const bool has_scale = FOLD ? true : SPECIAL ? false : a.has_scale != 0;
    using G = TileGeom<LOGT>;
    constexpr uint32_t T = 1u << LOGT, Q = T / 4, LOGQ = LOGT - 2;
    constexpr int ROUNDS = (LOGT + 1) / 2;
    constexpr int LW = G::LW;
    extern __shared__ uint32_t lds[];

    // the special kinds also fix the tiles per workgroup at what the planner picks for this tile size (512 lanes)
    const uint32_t log_c = MULTI ? (uint32_t)(LOGT <= 11 ? 11 - LOGT : 0) : a.log_c, C = 1u << log_c;
    const uint32_t Tp = G::ROW + tile_pad29(log_c);
    uint32_t tid = threadIdx.x;  // (re-read opaquely per item by the streaming kinds, see the item loop)
    const uint32_t in_w = in_raw ? RAW_W : 8u, out_w = out_raw ? RAW_W : 8u;
    // Work item -> (tile, polynomial). Workgroups are dealt round-robin over the 8 XCDs, each with
    // its own L2. With the grouped order, the workgroups that share an L2 take 2^grp_log adjacent
    // tiles (whose strided 64..96-byte pieces complete each other's 128-byte lines) of every
    // polynomial in turn (which re-use the same rows of the inter-pass twiddle table), while the
    // chip as a whole still sweeps memory front to back.
    auto decode = [&](uint32_t item, uint32_t& tau_, uint32_t& pidx_) {
        if (a.grp_log) {
            const uint32_t g = a.grp_log - 1, xcd = item & 7u, s = item >> 3, s2 = s >> g;
            pidx_ = s2 % a.polys;
            tau_ = ((((s2 / a.polys) << 3) | xcd) << g) | (s & ((1u << g) - 1u));
        } else {
            tau_ = item % a.tiles;
            pidx_ = item / a.tiles;
        }
    };
    uint32_t item = blockIdx.x;  // the streaming kinds go on to item + gridDim.x, ... below
    const uint32_t items = STREAM ? a.tiles * a.polys : 0u;
    uint32_t tau, pidx, hi, lo;
    uint64_t poly;
    auto set_item = [&](uint32_t it) {
        decode(it, tau, pidx);
        hi = tau >> a.dsh;
        lo = tau & a.dmask;
        poly = (uint64_t)pidx * a.poly_stride;
    };
    set_item(item);

    // LDS slot of position p of tile c (an element occupies nine dwords at slot * 9)
    auto at = [&](uint32_t c, uint32_t p) -> uint32_t {
        return c * Tp + (p & (G::W - 1)) * G::SB + swz(p >> LW);
    };
    // lane orders: "edge" = as the global access wants it (tile index fastest, or element fastest),
    // "local" = wavefront w of tile c runs butterflies j = lane * W + w
    auto edge_map = [&](bool cfast, uint32_t& c, uint32_t& j) {
        if (cfast) {
            c = tid & (C - 1);
            j = tid >> log_c;
        } else {
            c = tid >> LOGQ;
            j = tid & (Q - 1);
        }
    };
    auto local_map = [&](uint32_t& c, uint32_t& j) {
        if (Q >= 64) {
            const uint32_t wave = tid >> 6, lane = tid & 63u;
            c = wave >> LW;
            j = (lane << LW) | (wave & (G::W - 1));
        } else {
            c = tid >> LOGQ;
            j = tid & (Q - 1);
        }
    };
    // the element-fastest edge order coincides with the local order when a tile is one wavefront or less
    const bool load_is_local = !load_cfast && LW == 0;

    // global -> registers for the lane's four inputs (first round; zeros for tiles past the end)
    auto load_inputs = [&](uint32_t c, uint32_t j, bool valid) -> Quad29 {
        F29 x0, x1, x2, x3;
#ifdef STARKHIP_ABL_NOLOAD
        if (valid) {  // timing-only build: inputs made up in registers, no global load
#pragma unroll
            for (int l = 0; l < 9; l++) {
                x0.l[l] = (j * 2654435761u + l * 40503u + c) & F29_MASK;
                x1.l[l] = (x0.l[l] * 3u + hi + lo) & F29_MASK;
                x2.l[l] = (x1.l[l] * 5u + pidx) & F29_MASK;
                x3.l[l] = (x2.l[l] * 7u + 1u) & F29_MASK;
            }
            return Quad29{x0, x1, x2, x3};
        }
#endif
        if (valid) {
            const uint32_t* src = a.in + in_w * (poly + hi * a.inA + lo * a.inB + c * a.inC);
            // offset of input j (in dwords): j * inS, or split at bit jsh of j
            auto jo = [&](uint32_t jj) -> uint64_t {
                if (SPECIAL && KIND != KIND_LAST) return (uint64_t)in_w * ((uint64_t)jj * a.inS);
                return (uint64_t)in_w * ((uint64_t)(jj >> a.jsh) * a.inS2 + (uint64_t)(jj & a.jmask) * a.inS);
            };
            if (in_raw) {
                x0 = f29_load_mid(src + jo(j));
                x1 = f29_load_mid(src + jo(j + Q));
                x2 = f29_load_mid(src + jo(j + 2 * Q));
                x3 = f29_load_mid(src + jo(j + 3 * Q));
            } else {
                x0 = f29_from_fe(fe_load(src + jo(j)));
                x1 = f29_from_fe(fe_load(src + jo(j + Q)));
                x2 = f29_from_fe(fe_load(src + jo(j + 2 * Q)));
                x3 = f29_from_fe(fe_load(src + jo(j + 3 * Q)));
            }
        } else {
#pragma unroll
            for (int l = 0; l < 9; l++) x0.l[l] = x1.l[l] = x2.l[l] = x3.l[l] = 0;
        }
        return Quad29{x0, x1, x2, x3};
    };
    // Is the exchange after a round confined to each wavefront's own sub-image? (the round and the
    // next one both run in the local lane order, and Ns >= W keeps the low bits of j)
    auto exchange_is_local = [&](int round, bool local, uint32_t ns_now, uint32_t ns_next) {
        const bool next_last_edge = (round + 1 == ROUNDS - 1) && !(!store_cfast && LW == 0);
        return round != ROUNDS - 1 && local && ns_now >= (uint32_t)LW && ns_next >= (uint32_t)LW && !next_last_edge;
    };

    // ---- landing (streaming kinds) ----------------------------------------------------------------------------
    // A global_load_lds instruction deposits 4 or 16 bytes per lane at consecutive addresses from a wave-uniform LDS
    // base (lane L at base + 4 L or 16 L), each lane naming its own global address, which need only be 4-byte aligned
    // (measured, tools/probe/lds_dma_probe.hip). Both kinds run round 0 tile index fastest, lane (c, j) = (tid mod C,
    // tid / C) with C = 2 tiles of 1024 points or 8 tiles of 256 points per workgroup; a wavefront covers J = 64 / C
    // consecutive j and all C tiles.
    //   kind 7 (packed 32-byte elements, C adjacent columns). Every wavefront fetches exactly what it reads itself in
    //     round 0, into the part of the image that only it writes in round 0; nothing else touches that part between the
    //     barrier after the last round's LDS reads and the barrier after round 0, so the fetch needs no barrier of its
    //     own. Lane (c, j) writes positions 4j .. 4j + 3 of tile c: at 1024 points slot swz(j) of sub-images 0..3, so
    //     wavefront w owns the 32 slots from 32w of sub-image k of tile c (chunk i = 4c + k); at 256 points, where a tile
    //     is one sub-image, the 32 slots from 32w of tile c (chunk i = c): eight chunks of 1152 bytes either way. One
    //     instruction per chunk: instruction i = 2q + h takes rows Jw + (T/4) q + (J/2) h + (0 .. J/2 - 1) of all C
    //     columns, 2C lanes per run of 32 C bytes; lane (c, jj = j mod J) finds input q in chunk 2q + jj / (J/2) at byte
    //     32 ((jj mod (J/2)) * C + c).
    //   kind 8 (36-byte elements of the tiled intermediate: runs of C * C elements holding rows c = 0 .. C-1 at C
    //     consecutive j, 144 or 2304 bytes). A wavefront's 576 pieces of 16 bytes do not fit its own chunks in whole
    //     instructions, so the workgroup's runs land back to back from LDS byte 0 (nine instructions per wavefront) and
    //     round 0 reads element (c, j) at dword 9 ((j / C) * C * C + c * C + j mod C), between two barriers: every
    //     wavefront's pieces have landed / every lane has read, before round 0 overwrites the image. (Measured at 1024
    //     points against a variant that keeps kind 8 free of those barriers by running round 0 element index fastest,
    //     each wavefront fetching its own 36-byte elements as 16 + 16 + 4 bytes: 3 % slower, the fetch then takes its
    //     runs apart.)
    constexpr uint32_t SLC = LOGT <= 11 ? 11 - LOGT : 0, SC = 1u << SLC, SJ = 64u >> SLC;  // log2 C, C, J of the streaming kinds
    uint32_t wv = STREAM ? (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6)) : 0u, ln = tid & 63u;
    auto chunk = [&](uint32_t i) -> uint32_t {  // dword index of chunk i of this wavefront
        if (LOGT == 10) return ((i >> 2) * Tp + (i & 3u) * G::SB + 32u * wv) * 9u;
        return (i * Tp + 32u * wv) * 9u;
    };
    auto issue_landing = [&](uint32_t it) {
        uint32_t tau2, pidx2;
        decode(it, tau2, pidx2);
        [[maybe_unused]] const uint64_t base = (uint64_t)pidx2 * a.poly_stride + (uint64_t)(tau2 >> a.dsh) * a.inA + (uint64_t)(tau2 & a.dmask) * a.inB;
#if defined(__HIP_DEVICE_COMPILE__)
        if (KIND == KIND_SFIRST) {
            const uint64_t rs = a.inS * 32;  // bytes between rows
            const char* g = reinterpret_cast<const char*>(a.in) + 32 * base + (uint64_t)(SJ * wv + (ln >> (SLC + 1))) * rs +
                            (ln & (2u * SC - 1u)) * 16u;
#pragma unroll
            for (uint32_t i = 0; i < 8; i++)
                __builtin_amdgcn_global_load_lds((lds_dma_src_t)(g + (uint64_t)(Q * (i >> 1) + (SJ / 2) * (i & 1u)) * rs),
                                                 (lds_dma_dst_t)(lds + chunk(i)), 16, 0, 0);
        } else if (KIND == KIND_SLAST) {
            constexpr uint32_t PPR = 9u * SC * SC / 4u;  // 16-byte pieces per run
            const uint32_t rs = (uint32_t)a.inS2 * 36u;  // bytes between runs (the last run starts below 2^32)
            const char* g = reinterpret_cast<const char*>(a.in) + 36 * base;
#pragma unroll
            for (uint32_t i = 0; i < 9; i++) {
                const uint32_t piece = (wv * 9u + i) * 64u + ln, t = piece / PPR, part = piece - PPR * t;
                __builtin_amdgcn_global_load_lds((lds_dma_src_t)(g + ((uint64_t)t * rs + part * 16u)),
                                                 (lds_dma_dst_t)(lds + (wv * 9u + i) * 256u), 16, 0, 0);
            }
        }
#endif
    };
    // the lane's four inputs from the landing (round 0 of a streaming kind), lane (c, j) = (tid mod C, tid / C)
    auto landed_inputs = [&]() -> Quad29 {
        F29 x[4];
        const uint32_t cc = ln & (SC - 1u);
#pragma unroll
        for (uint32_t q = 0; q < 4; q++) {
            if (KIND == KIND_SFIRST) {
                const uint32_t jj = ln >> SLC;  // j mod J
                const uint4* p4 = reinterpret_cast<const uint4*>(lds + chunk(2 * q + jj / (SJ / 2)) + 8u * ((jj % (SJ / 2)) * SC + cc));
                const uint4 u0 = p4[0], u1 = p4[1];
                Fe f;
                f.l[0] = u0.x; f.l[1] = u0.y; f.l[2] = u0.z; f.l[3] = u0.w;
                f.l[4] = u1.x; f.l[5] = u1.y; f.l[6] = u1.z; f.l[7] = u1.w;
                x[q] = f29_from_fe(f);
            } else {
                const uint32_t jq = (tid >> SLC) + q * Q;
                const uint32_t* p1 = lds + 9u * (((jq >> SLC) << (2 * SLC)) + (cc << SLC) + (jq & (SC - 1u)));
#pragma unroll
                for (int l = 0; l < 9; l++) x[q].l[l] = p1[l];
            }
        }
        return Quad29{x[0], x[1], x[2], x[3]};
    };
    // Which item a streaming workgroup takes next: not item + gridDim.x. The two workgroups of a compute unit do not
    // advance at the same rate (the older wavefronts win the issue arbitration, and resident workgroups never change
    // age: one was measured at 28 us per item beside one at 70), so equal shares would leave half of the chip waiting
    // for the other half. Instead every XCD has a ticket counter (a.tickets[xcd], zeroed before the launch): a
    // workgroup's first item is its own index, every further one ticket * 8 + xcd, drawn by thread 0 at the start of the
    // last round and handed to the other wavefronts through the LDS word behind the image. It is drawn a round early
    // because the fetch of the next item is queued in the last round; a workgroup that draws a ticket past the end
    // fetches its current item once more (nobody reads it), so that the number of fetch instructions is fixed.
    constexpr uint32_t NO_ITEM = 0xFFFFFFFFu;
    uint32_t nxt = NO_ITEM;
    const uint32_t lds_next = C * Tp * 9u;  // dword index of the hand-over word

    constexpr bool ODD = (LOGT & 1) != 0;
    if (STREAM) issue_landing(item);
    do {  // one pass for every kind but the streaming ones
    if (STREAM) {
        // everything a lane derives from its index is worked out again per item: kept across items (the compiler hoists
        // it out of the loop) it fills registers that the last round then spills, and a reload from scratch there would
        // wait behind the fetch
        asm volatile("" : "+v"(tid));
        ln = tid & 63u;
        wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    }
    uint32_t ns_log = 0;
    STAMP(0);
#ifdef STARKHIP_STAMPS
    if ((threadIdx.x & 63u) == 0 && item < 8192) {
        g_stamps[in_raw ? 1 : 0][item][(threadIdx.x >> 6) & 7u][16] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
        g_stamps[in_raw ? 1 : 0][item][(threadIdx.x >> 6) & 7u][17] = __builtin_amdgcn_s_getreg((31 << 11) | 20);
    }
#endif
    if (ODD) {
        // odd log2(T): one radix-2 level first, on tight inputs; outputs stay lazy (limbs < 3 * 2^29)
        uint32_t c, j;
        edge_map(load_cfast, c, j);
        const bool valid = MULTI ? true : (uint64_t)lo * C + c < a.c_limit;
        const Quad29 q = load_inputs(c, j, valid);
        const F29 &x0 = q.a, &x1 = q.b, &x2 = q.c, &x3 = q.d;
        F29 y0 = f29_add(x0, x2), y1 = f29_sub(x0, x2), y2 = f29_add(x1, x3), y3 = f29_sub(x1, x3);
        lds_put29(lds, at(c, 2 * j), y0);
        lds_put29(lds, at(c, 2 * j + 1), y1);
        lds_put29(lds, at(c, 2 * (j + Q)), y2);
        lds_put29(lds, at(c, 2 * (j + Q) + 1), y3);
        if (exchange_is_local(0, load_is_local, 0, 1)) __builtin_amdgcn_wave_barrier();
        else __syncthreads();
        ns_log = 1;
    }
    // One round. The streaming kinds instantiate it three times (first, middle, last round, the flags compile-time
    // constants): with one body for all rounds the compiler cannot tell that the fetch queued in a last round has been
    // waited for before the middle rounds and puts a wait for all outstanding loads in front of every LDS access there.
    auto do_round = [&](auto first_, auto last_, int round) __attribute__((always_inline)) {
        const bool first = first_, last = last_;
        // which lane order this round runs in
        bool local;
        uint32_t c, j;
        if (first) {
            local = load_is_local;
            edge_map(load_cfast, c, j);
        } else if (last) {
            local = !store_cfast && LW == 0;
            edge_map(store_cfast, c, j);
        } else if (ns_log >= (uint32_t)LW) {
            local = true;
            local_map(c, j);
        } else {
            local = false;
            edge_map(load_cfast, c, j);
        }
        const bool valid = MULTI ? true : (uint64_t)lo * C + c < a.c_limit;

        // the first of the round's three twiddles w_T^(r * (j mod Ns) * T/(4 Ns)) is requested before
        // the LDS reads so that both latencies overlap
        F29 tw1;
        const uint32_t e = ns_log > 0 ? (j & ((1u << ns_log) - 1u)) << (LOGQ - ns_log) : 0u;
        // the output factor of a single-pass inverse rides on the last round's twiddles when that round has any
        const bool folded = FOLD && last && ns_log > 0;
        const uint32_t* Wt = folded ? a.Wlast : a.W;
        if (ns_log > 0) tw1 = f29_load_raw(Wt + 12 * (size_t)e);
        // a streaming kind's last round requests all three twiddles here: once the next item's fetch is queued, any
        // later load would come back behind it
        F29 tw2, tw3;
        uint32_t ticket = 0;
        if (STREAM && last) {
            tw2 = f29_load_raw(Wt + 12 * (size_t)(2 * e));
            tw3 = f29_load_raw(Wt + 12 * (size_t)(3 * e));
            // The draw, by thread 0 alone. Written out so that nothing waits for it here: inside a branch the compiler
            // waits for the returned value where the branch ends; this way the value is first looked at behind the
            // wait for all loads below, a round of LDS reads later. (exec is narrowed to thread 0 of wavefront 0 and
            // empty in the other wavefronts, where the instruction then does nothing.)
            const uint64_t only0 = __ballot(tid == 0);
            uint64_t saved_exec;
            asm volatile(
                "s_mov_b64 %[sv], exec\n\t"
                "s_mov_b64 exec, %[m]\n\t"
                "global_atomic_add %[d], %[off], %[one], %[base] sc0\n\t"
                "s_mov_b64 exec, %[sv]"
                : [d] "=&v"(ticket), [sv] "=&s"(saved_exec)
                : [m] "s"(only0), [off] "v"((blockIdx.x & 7u) * 4u), [one] "v"(1u), [base] "s"(a.tickets)
                : "memory");
        }
        F29 x0, x1, x2, x3;
        if (STREAM && first) {
            // kind 8 multiplies its inputs by the inter-pass factors w^(k1 * j) (row k1 = lo * C + c, contiguous in j)
            // (requesting them for the next item behind the last round's arithmetic instead was measured: no gain)
            Fe t0, t1, t2, t3;
            if (KIND == KIND_SLAST) {
                const uint32_t* tw = a.TW + 8 * (pidx * a.twP + hi * a.twA + lo * a.twB + c * a.twC);
                t0 = fe_load(tw + 8 * ((uint64_t)j * a.twS));
                t1 = fe_load(tw + 8 * ((uint64_t)(j + Q) * a.twS));
                t2 = fe_load(tw + 8 * ((uint64_t)(j + 2 * Q) * a.twS));
                t3 = fe_load(tw + 8 * ((uint64_t)(j + 3 * Q) * a.twS));
            }
            __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): this wavefront's landing (and the item before's stores)
            if (KIND == KIND_SLAST) __syncthreads();  // ... and every other wavefront's
            const Quad29 q = landed_inputs();
            if (KIND == KIND_SLAST) __syncthreads();  // all read before round 0 writes over the landing
            x0 = q.a;
            x1 = q.b;
            x2 = q.c;
            x3 = q.d;
            if (KIND == KIND_SLAST) {
                x0 = f29_mul<MULW>(x0, f29_from_fe(t0));
                x1 = f29_mul<MULW>(x1, f29_from_fe(t1));
                x2 = f29_mul<MULW>(x2, f29_from_fe(t2));
                x3 = f29_mul<MULW>(x3, f29_from_fe(t3));
            }
        } else if (first) {
            const Quad29 q = load_inputs(c, j, valid);
            x0 = q.a;
            x1 = q.b;
            x2 = q.c;
            x3 = q.d;
        } else {
            x0 = lds_get29(lds, at(c, j));
            x1 = lds_get29(lds, at(c, j + Q));
            x2 = lds_get29(lds, at(c, j + 2 * Q));
            x3 = lds_get29(lds, at(c, j + 3 * Q));
        }
        const bool out_local = exchange_is_local(round, local, ns_log, ns_log + 2);
        STAMPW(1 + 3 * round);
        if (!first) {
            // inputs are in registers. If other wavefronts will overwrite them (cross-wave exchange)
            // every wavefront must have finished reading first; a wave-local exchange only needs the
            // compiler to keep this wavefront's own LDS order.
            if (out_local) __builtin_amdgcn_wave_barrier();
            else if (!last) __syncthreads();
            else if (STREAM) {
                // every wavefront holds its inputs: the image is dead, and the next item's input may land in it
                // The round's twiddles must be in before the fetch is queued: the compiler answers any wait for a load
                // with a wait for every outstanding load while a global_load_lds is in flight (it cannot tell which LDS
                // bytes that one writes), so nothing may be waited for between here and the next item. The scheduling
                // barriers keep the fetch where it is written: ahead of the round's arithmetic.
                __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
                if (tid == 0) {
                    const uint32_t t = ticket + (gridDim.x >> 3);
                    lds[lds_next] = t < (items >> 3) ? 8u * t + (blockIdx.x & 7u) : NO_ITEM;
                }
                __syncthreads();
                nxt = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds[lds_next]);
                __builtin_amdgcn_sched_barrier(0);
                issue_landing(nxt == NO_ITEM ? item : nxt);
                __builtin_amdgcn_sched_barrier(0);
            }
        }

        if (ns_log > 0) {
            // the three twiddled inputs come out of the multiplier tight; the untwiddled one is brought
            // back to 29-bit limbs by a carry pass. Each later twiddle is requested one multiplication
            // ahead of its use.
            if (!(STREAM && last)) tw2 = f29_load_raw(Wt + 12 * (size_t)(2 * e));
            x1 = f29_mul<MULW_TW>(x1, tw1);
            if (!(STREAM && last)) tw3 = f29_load_raw(Wt + 12 * (size_t)(3 * e));
            x2 = f29_mul<MULW_TW>(x2, tw2);
            x3 = f29_mul<MULW_TW>(x3, tw3);
            if (folded) x0 = f29_mul<MULW_TW>(x0, f29_load_raw(Wt));  // entry 0 of the scaled table is the factor itself
            else x0 = f29_carry(x0);
        }
        // radix-4 butterfly, I = w_T^(T/4); every subtrahend is tight or a sum of two tight values
        F29 y0, y1, y2, y3;
        const uint32_t d = ((j >> ns_log) << (ns_log + 2)) + (j & ((1u << ns_log) - 1u));
        {
            F29 a0 = f29_add(x0, x2), a1 = f29_sub(x0, x2), a2 = f29_add(x1, x3);
            y0 = f29_add(a0, a2);
            y2 = f29_sub2(a0, a2);
            if (!last) {  // these two do not wait for the multiplication by I
                lds_put29(lds, at(c, d), y0);
                lds_put29(lds, at(c, d + (2u << ns_log)), y2);
            }
            // the inter-pass twiddles of the first two outputs are requested before the last multiplication
            // of the butterfly, the other two one multiplication ahead of their use
            const bool tw_on = last && valid && has_tw;
            const uint32_t* tw = a.TW + 8 * (pidx * a.twP + hi * a.twA + lo * a.twB + c * a.twC);
            // How far ahead the four table entries are requested is a trade between latency and registers: two ahead
            // everywhere except at 1024-point tiles, where that costs a spilled register (one ahead there: +1.4 %;
            // one ahead at 256-point tiles: -2.4 %)
            constexpr bool TW_ONE_AHEAD = LOGT == 10;
            Fe t0, t1;
            if (tw_on) {
                t0 = fe_load(tw + 8 * ((uint64_t)j * a.twS));
                if (!TW_ONE_AHEAD) t1 = fe_load(tw + 8 * ((uint64_t)(j + Q) * a.twS));
            }
            F29 a3 = f29_mul_uniform2<MULW_U>(f29_sub(x1, x3), a.iq, a.iq2);
            y1 = f29_add(a1, a3);
            y3 = f29_sub(a1, a3);
            if (tw_on) {
                if (TW_ONE_AHEAD) {
                    t1 = fe_load(tw + 8 * ((uint64_t)(j + Q) * a.twS));
                    y0 = f29_mul<MULW>(y0, f29_from_fe(t0));
                    const Fe t2 = fe_load(tw + 8 * ((uint64_t)(j + 2 * Q) * a.twS));
                    y1 = f29_mul<MULW>(y1, f29_from_fe(t1));
                    const Fe t3 = fe_load(tw + 8 * ((uint64_t)(j + 3 * Q) * a.twS));
                    y2 = f29_mul<MULW>(y2, f29_from_fe(t2));
                    y3 = f29_mul<MULW>(y3, f29_from_fe(t3));
                } else {
                    const Fe t2 = fe_load(tw + 8 * ((uint64_t)(j + 2 * Q) * a.twS));
                    y0 = f29_mul<MULW>(y0, f29_from_fe(t0));
                    const Fe t3 = fe_load(tw + 8 * ((uint64_t)(j + 3 * Q) * a.twS));
                    y1 = f29_mul<MULW>(y1, f29_from_fe(t1));
                    y2 = f29_mul<MULW>(y2, f29_from_fe(t2));
                    y3 = f29_mul<MULW>(y3, f29_from_fe(t3));
                }
            }
        }
        if (!last) {
            lds_put29(lds, at(c, d + (1u << ns_log)), y1);
            lds_put29(lds, at(c, d + (3u << ns_log)), y3);
            STAMP(2 + 3 * round);
            if (out_local) __builtin_amdgcn_wave_barrier();
            else __syncthreads();  // the next round's inputs come from other wavefronts
            STAMP(3 + 3 * round);
        } else if (valid) {
            if (!has_tw && has_scale && !folded) {
                Fe s;
#pragma unroll
                for (int l = 0; l < 8; l++) s.l[l] = a.scale[l];
                const F29 sc = f29_from_fe(s);
                y0 = f29_mul<MULW>(y0, sc);
                y1 = f29_mul<MULW>(y1, sc);
                y2 = f29_mul<MULW>(y2, sc);
                y3 = f29_mul<MULW>(y3, sc);
            }
            uint32_t* dst = a.out + out_w * ((uint64_t)pidx * a.outP + hi * a.outA + lo * a.outB + c * a.outC);
            // offset of output k (in dwords): k * outS, or split at bit ksh of k
            auto ko = [&](uint32_t k) -> uint64_t {
                if (SPECIAL) return (uint64_t)out_w * ((uint64_t)k * a.outS);
                return (uint64_t)out_w * ((uint64_t)(k >> a.ksh) * a.outS2 + (uint64_t)(k & a.kmask) * a.outS);
            };
            if (out_raw) {
                f29_store_mid(dst + ko(j), y0);
                f29_store_mid(dst + ko(j + Q), y1);
                f29_store_mid(dst + ko(j + 2 * Q), y2);
                f29_store_mid(dst + ko(j + 3 * Q), y3);
            } else {
                fe_store(dst + ko(j), f29_to_fe(y0));
                fe_store(dst + ko(j + Q), f29_to_fe(y1));
                fe_store(dst + ko(j + 2 * Q), f29_to_fe(y2));
                fe_store(dst + ko(j + 3 * Q), f29_to_fe(y3));
            }
        }
        ns_log += 2;
    };
    if (STREAM) {
        do_round(std::true_type{}, std::false_type{}, 0);
#pragma unroll 1
        for (int round = 1; round < ROUNDS - 1; round++) do_round(std::false_type{}, std::false_type{}, round);
        do_round(std::false_type{}, std::true_type{}, ROUNDS - 1);
    } else {
#pragma unroll 1
        for (int round = ODD ? 1 : 0; round < ROUNDS; round++) do_round(!ODD && round == 0, round == ROUNDS - 1, round);
    }
    STAMP(2 + 3 * (ROUNDS - 1));
#ifdef STARKHIP_STAMPS
    STAMPW(3 + 3 * (ROUNDS - 1));
#endif
    if (!STREAM) break;
    item = nxt;
    if (item == NO_ITEM) break;
    set_item(item);
    } while (true);
}

#ifdef STARKHIP_STAMPS
extern "C" int stark_debug_read_stamps(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(g_stamps));
}
#endif

// How many workgroups of a streaming kind stay resident on a device of n_cu compute units: two per unit (LDS), a
// multiple of the 8 XCDs so that a work item and the workgroup that takes it share an L2.
unsigned ntt29_stream_resident(int n_cu) {
    static const int per_cu = getenv("STARKHIP_STREAM_WGS") ? atoi(getenv("STARKHIP_STREAM_WGS")) : 2;
    unsigned r = ((unsigned)(n_cu * (per_cu > 0 ? per_cu : 2)) / 8u) * 8u;
    return r < 8 ? 8 : r;
}

template <int LOGT, int KIND>
static hipError_t launch_one29(const NttPassArgs& a, unsigned tiles, unsigned polys, hipStream_t stream) {
    size_t lds = ntt29_pass_lds_bytes(LOGT, (int)a.log_c);
    // STARKHIP_LDS_PAD=bytes: unused LDS on top, to measure how the kernel responds to fewer resident workgroups
    static const size_t lds_pad = getenv("STARKHIP_LDS_PAD") ? (size_t)atol(getenv("STARKHIP_LDS_PAD")) : 0;
    lds += lds_pad;
    unsigned threads = (1u << (LOGT - 2)) << a.log_c;
    if (threads > 1024 || lds > 160 * 1024) return hipErrorInvalidConfiguration;
    // per device: dynamic LDS above 64 KiB must be opted into. Contexts of several host threads may get here at once:
    // the flags are atomic, and setting the attribute twice is harmless.
    static std::atomic<bool> attr_set[64];
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev >= 64 || !attr_set[dev].load(std::memory_order_acquire)) {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&ntt29_pass_kernel<LOGT, KIND>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        if (dev < 64) attr_set[dev].store(true, std::memory_order_release);
    }
    NttPassArgs b = a;
    static const int grp = getenv("STARKHIP_GROUP") ? atoi(getenv("STARKHIP_GROUP")) : 4;  // log2 of tiles per L2 group, -1 = off
    b.tiles = tiles;
    b.polys = polys;
    if (!b.in_split) {
        b.jsh = 31;
        b.jmask = 0xFFFFFFFFu;
    }
    b.grp_log = 0;
    // the largest group up to 2^grp tiles that divides the tiles evenly over the 8 XCDs
    for (int g = grp; g >= 0; g--)
        if (tiles % (8u << g) == 0) {
            b.grp_log = (uint32_t)g + 1;
            break;
        }
    if ((uint64_t)tiles * polys >= (1ull << 31)) return hipErrorInvalidConfiguration;
    unsigned grid = tiles * polys;
    if (KIND == KIND_SFIRST || KIND == KIND_SLAST) {
        static std::atomic<int> cus[64];
        int n_cu = dev < 64 ? cus[dev].load(std::memory_order_acquire) : 0;
        if (n_cu == 0) {
            e = hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev);
            if (e != hipSuccess) return e;
            if (dev < 64) cus[dev].store(n_cu, std::memory_order_release);
        }
        const unsigned resident = ntt29_stream_resident(n_cu);
        if (grid > resident) grid = resident;
        lds += 16;  // the hand-over word behind the image
        if (lds > 160 * 1024 || !b.tickets) return hipErrorInvalidConfiguration;
        e = hipMemsetAsync(b.tickets, 0, 8 * sizeof(uint32_t), stream);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((ntt29_pass_kernel<LOGT, KIND>), dim3(grid), dim3(threads), lds, stream, b);
    return hipGetLastError();
}

template <int LOGT>
static hipError_t launch_either29(const NttPassArgs& a, unsigned tiles, unsigned polys, hipStream_t stream) {
    static const bool special = !(getenv("STARKHIP_SPECIALIZE") && atoi(getenv("STARKHIP_SPECIALIZE")) == 0);
    const bool plain_index = a.ksh == 31 && !a.in_split;
    NttPassArgs g = a;
    g.Wlast = nullptr;  // only KIND_FOLD carries the output factor on its last round
    // single pass: packed on both sides, element index fastest
    if (LOGT >= 3 && !a.in_split && !a.in_raw && !a.out_raw && !a.load_cfast && !a.store_cfast) {
        constexpr bool S = LOGT >= 3;
        if (!a.TW && plain_index) {
            if (a.Wlast != nullptr && a.has_scale) return launch_one29<LOGT, (S ? KIND_FOLD : KIND_GENERAL)>(a, tiles, polys, stream);
            if (special && !a.has_scale) return launch_one29<LOGT, (S ? KIND_SINGLE : KIND_GENERAL)>(g, tiles, polys, stream);
        }
        // (with an output table and a split output index, stage A of a sharded transform: its own instantiation was 2 %
        // slower than the general kernel at 2^12 points, 128 registers against 123)
    }
    if (a.stream) {  // the planner sets this only for the shapes the streaming kinds are written for
        constexpr bool S = LOGT == 10 || LOGT == 8;
        if (!S || a.log_c != (uint32_t)(11 - LOGT) || !a.load_cfast || !a.store_cfast || a.ksh != 31 || a.c_limit != ~0ull || a.has_scale || tiles % 8u)
            return hipErrorInvalidConfiguration;
        if (!a.in_raw) return launch_one29<LOGT, (S ? KIND_SFIRST : KIND_GENERAL)>(g, tiles, polys, stream);
        return launch_one29<LOGT, (S ? KIND_SLAST : KIND_GENERAL)>(g, tiles, polys, stream);
    }
    if (special && LOGT >= 6 && !a.has_scale && a.store_cfast && a.ksh == 31 && a.c_limit == ~0ull &&
        a.log_c == (uint32_t)(LOGT <= 11 ? 11 - LOGT : 0)) {  // multi-pass plans in their usual shape
        constexpr bool S = LOGT >= 6;
        if (a.TW && a.load_cfast && !a.in_split && a.out_raw) {
            if (!a.in_raw) return launch_one29<LOGT, (S ? KIND_FIRST : KIND_GENERAL)>(g, tiles, polys, stream);
            return launch_one29<LOGT, (S ? KIND_MID : KIND_GENERAL)>(g, tiles, polys, stream);
        }
        if (!a.TW && a.in_raw && !a.out_raw) {
            if (a.load_cfast) return launch_one29<LOGT, (S ? KIND_LAST : KIND_GENERAL)>(g, tiles, polys, stream);
            if (!a.load_cfast && !a.in_split) return launch_one29<LOGT, (S ? KIND_LAST3 : KIND_GENERAL)>(g, tiles, polys, stream);
        }
    }
    return launch_one29<LOGT, KIND_GENERAL>(g, tiles, polys, stream);
}

hipError_t launch_ntt29_pass(int log_t, const NttPassArgs& a, unsigned tiles, unsigned polys, hipStream_t stream) {
    switch (log_t) {
        case 2: return launch_either29<2>(a, tiles, polys, stream);
        case 3: return launch_either29<3>(a, tiles, polys, stream);
        case 4: return launch_either29<4>(a, tiles, polys, stream);
        case 5: return launch_either29<5>(a, tiles, polys, stream);
        case 6: return launch_either29<6>(a, tiles, polys, stream);
        case 7: return launch_either29<7>(a, tiles, polys, stream);
        case 8: return launch_either29<8>(a, tiles, polys, stream);
        case 9: return launch_either29<9>(a, tiles, polys, stream);
        case 10: return launch_either29<10>(a, tiles, polys, stream);
        case 11: return launch_either29<11>(a, tiles, polys, stream);
        case 12: return launch_either29<12>(a, tiles, polys, stream);
        default: return hipErrorInvalidValue;
    }
}

// Unpack a table of packed canonical elements (8 dwords each) into 12-dword entries.
__global__ void unpack_table_kernel(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, size_t count) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    f29_store_raw(out + 12 * i, f29_from_fe(fe_load(in + 8 * i)));
}

hipError_t launch_unpack_table(const uint32_t* in, uint32_t* out, size_t count, hipStream_t stream) {
    if (count == 0) return hipSuccess;
    hipLaunchKernelGGL(unpack_table_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, stream, in, out, count);
    return hipGetLastError();
}

}  // namespace starkhip
