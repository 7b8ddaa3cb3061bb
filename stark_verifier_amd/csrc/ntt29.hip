// Tile-transform pass kernel on 29-bit-limb ("carry-free") arithmetic — the production path.
//
// Same transform and the same pass / tile geometry as described in ntt.hip (what the
// reference's _fft computes, reference src/fft.rs:37-61, natural order in and out, canonical
// outputs); what differs is the representation inside the kernel:
//   * on load an element (8 x 32-bit limbs in HBM) is unpacked to nine 29-bit limbs; LDS holds
//     nine limb planes per tile;
//   * butterfly additions and subtractions are limb-wise v_add_u32 / v_sub_u32 with no carry
//     chain, the products are carry-free v_mad_u64_u32 columns (field29.cuh);
//   * between passes of a multi-pass transform the intermediate stays unpacked (12 dwords per
//     element, 16-byte aligned), so only the first load and the last store convert; the last
//     store reduces to the canonical residue and repacks.
// Tile twiddles W are stored unpacked (12 dwords per entry); the large inter-pass table TW
// stays packed (8 dwords) and is unpacked on load.
#include "field29.cuh"
#include "internal.hpp"
#include <stdlib.h>

namespace starkhip {

__device__ __forceinline__ void lds_put29(uint32_t* lds, uint32_t plane, uint32_t off, const F29& v) {
#pragma unroll
    for (int l = 0; l < 9; l++) lds[l * plane + off] = v.l[l];
}

__device__ __forceinline__ F29 lds_get29(const uint32_t* lds, uint32_t plane, uint32_t off) {
    F29 v;
#pragma unroll
    for (int l = 0; l < 9; l++) v.l[l] = lds[l * plane + off];
    return v;
}

// unpacked element in global memory: 12 dwords (limbs 0..8, three unused)
__device__ __forceinline__ F29 f29_load_raw(const uint32_t* p) {
    const uint4* q = reinterpret_cast<const uint4*>(p);
    uint4 a = q[0], b = q[1], c = q[2];
    F29 r;
    r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w;
    r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
    r.l[8] = c.x;
    return r;
}

__device__ __forceinline__ void f29_store_raw(uint32_t* p, const F29& v) {
    uint4* q = reinterpret_cast<uint4*>(p);
    q[0] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
    q[1] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
    q[2] = make_uint4(v.l[8], 0u, 0u, 0u);
}

__host__ __device__ inline uint32_t tile_pad29(uint32_t log_c) {
    if (log_c == 0) return 0;
    return log_c >= 5 ? 1u : (32u >> log_c);
}

size_t ntt29_pass_lds_bytes(int log_t, int log_c) {
    size_t tp = ((size_t)1 << log_t) + tile_pad29((uint32_t)log_c);
    return 36 * tp << log_c;
}

// Diagnostic phase stamp (only when a.stamps is set): wave 0 of the workgroup records the shader
// clock after draining its own memory counters, so the differences are phase durations.
#define STAMP(slot)                                                                     \
    do {                                                                                \
        if (a.stamps && tid == 0 && w == blockIdx.x + gridDim.x) {                      \
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                 \
            a.stamps[(size_t)blockIdx.x * 32 + (slot)] = __builtin_amdgcn_s_memtime();  \
        }                                                                               \
    } while (0)

template <int LOGT>
__global__ void __launch_bounds__(1024) ntt29_pass_kernel(const NttPassArgs a) {
    constexpr uint32_t T = 1u << LOGT, Q = T / 4, LOGQ = LOGT - 2;
    constexpr int ROUNDS = (LOGT + 1) / 2;
    extern __shared__ uint32_t lds[];

    const uint32_t log_c = a.log_c, C = 1u << log_c;
    const uint32_t Tp = T + tile_pad29(log_c);
    const uint32_t plane = Tp << log_c;
    const uint32_t tid = threadIdx.x;
    const uint32_t in_w = a.in_raw ? 12u : 8u, out_w = a.out_raw ? 12u : 8u;

    // Spread the workgroups' start times over one tile period so that across the chip (and
    // between the workgroups sharing a CU) load, compute and store phases overlap instead of
    // every workgroup hitting HBM at the same moment.
    if (a.stagger_cycles) {
        const uint32_t delay = (uint32_t)(((uint64_t)((blockIdx.x * 2654435761u) >> 8) * a.stagger_cycles) >> 24);
        const uint64_t t0 = __builtin_amdgcn_s_memtime();
        while (__builtin_amdgcn_s_memtime() - t0 < delay) __builtin_amdgcn_s_sleep(32);
    }

    const uint64_t total = (uint64_t)a.tiles * a.polys;
#pragma unroll 1
    for (uint64_t w = blockIdx.x; w < total; w += gridDim.x) {
    uint32_t tau, pidx;
    if (a.xcd_map) {  // tiles is a multiple of 8: items w, w+8, w+16.. (one XCD) walk the polynomials of one tile
        const uint64_t q = w >> 3;
        pidx = (uint32_t)(q % a.polys);
        tau = (uint32_t)(q / a.polys) * 8u + ((uint32_t)w & 7u);
    } else {
        tau = (uint32_t)(w % a.tiles);
        pidx = (uint32_t)(w / a.tiles);
    }
    const uint32_t hi = tau >> a.dsh, lo = tau & a.dmask;
    const uint64_t poly = (uint64_t)pidx * a.poly_stride;
    if (w != blockIdx.x) __syncthreads();  // the previous item's last LDS reads are done
    STAMP(0);

    uint32_t ns_log = 0;
    F29 carry_keep;
#pragma unroll
    for (int l = 0; l < 9; l++) carry_keep.l[l] = tid + l;
#pragma unroll 1
    for (int round = 0; round < ROUNDS; round++) {
        const bool first = round == 0, last = round == ROUNDS - 1;
        const bool cfast = last ? a.store_cfast : a.load_cfast;
        uint32_t c, j;
        if (cfast) {
            c = tid & (C - 1);
            j = tid >> log_c;
        } else {
            c = tid >> LOGQ;
            j = tid & (Q - 1);
        }
        const bool valid = (uint64_t)lo * C + c < a.c_limit;
        const uint32_t row = c * Tp;

        F29 x0, x1, x2, x3;
        if (first) {
            if (valid && !(a.ablate & 4)) {
                const uint32_t* src = a.in + in_w * (poly + hi * a.inA + lo * a.inB + c * a.inC);
                const uint64_t s = (uint64_t)in_w * a.inS;
                if (a.in_raw) {
                    x0 = f29_load_raw(src + s * j);
                    x1 = f29_load_raw(src + s * (j + Q));
                    x2 = f29_load_raw(src + s * (j + 2 * Q));
                    x3 = f29_load_raw(src + s * (j + 3 * Q));
                } else {
                    x0 = f29_from_fe(fe_load(src + s * j));
                    x1 = f29_from_fe(fe_load(src + s * (j + Q)));
                    x2 = f29_from_fe(fe_load(src + s * (j + 2 * Q)));
                    x3 = f29_from_fe(fe_load(src + s * (j + 3 * Q)));
                }
            } else {
#pragma unroll
                for (int l = 0; l < 9; l++) x0.l[l] = x1.l[l] = x2.l[l] = x3.l[l] = (a.ablate & 4) ? (tid * 2654435761u + l) >> 4 : 0;
            }
        } else if (a.ablate & 2) {
            x0 = x1 = x2 = x3 = carry_keep;
        } else {
            x0 = lds_get29(lds, plane, row + j);
            x1 = lds_get29(lds, plane, row + j + Q);
            x2 = lds_get29(lds, plane, row + j + 2 * Q);
            x3 = lds_get29(lds, plane, row + j + 3 * Q);
        }

        if (first) STAMP(1);
        else STAMP(2 + 4 * round);
        if (a.ablate & 16) {  // timing only: the pass's memory traffic without its arithmetic
            if (valid) {
                uint32_t* dst = a.out + out_w * (poly + hi * a.outA + lo * a.outB + c * a.outC);
                const uint64_t s = (uint64_t)out_w * a.outS;
                if (a.out_raw) {
                    f29_store_raw(dst + s * j, x0);
                    f29_store_raw(dst + s * (j + Q), x1);
                    f29_store_raw(dst + s * (j + 2 * Q), x2);
                    f29_store_raw(dst + s * (j + 3 * Q), x3);
                } else {
                    Fe t;
#pragma unroll
                    for (int l = 0; l < 8; l++) t.l[l] = x0.l[l] ^ x1.l[l];
                    fe_store(dst + s * j, t);
                    fe_store(dst + s * (j + Q), t);
                    fe_store(dst + s * (j + 2 * Q), t);
                    fe_store(dst + s * (j + 3 * Q), t);
                }
            }
            break;
        }
        if ((LOGT & 1) && first) {
            // radix-2 level on tight inputs; outputs stay lazy (limbs < 3 * 2^29)
            F29 y0 = f29_add(x0, x2), y1 = f29_sub(x0, x2), y2 = f29_add(x1, x3), y3 = f29_sub(x1, x3);
            lds_put29(lds, plane, row + 2 * j, y0);
            lds_put29(lds, plane, row + 2 * j + 1, y1);
            lds_put29(lds, plane, row + 2 * (j + Q), y2);
            lds_put29(lds, plane, row + 2 * (j + Q) + 1, y3);
            __syncthreads();
            ns_log = 1;
            continue;
        }

        if (ns_log > 0) {
            // the three twiddled inputs come out of the multiplier tight; the untwiddled one
            // is brought back to 29-bit limbs by a carry pass
            const uint32_t e = (a.ablate & 1) ? 0u : (j & ((1u << ns_log) - 1u)) << (LOGQ - ns_log);
            x1 = f29_mul(x1, f29_load_raw(a.W + 12 * (size_t)e));
            x2 = f29_mul(x2, f29_load_raw(a.W + 12 * (size_t)(2 * e)));
            x3 = f29_mul(x3, f29_load_raw(a.W + 12 * (size_t)(3 * e)));
            x0 = f29_carry(x0);
        }
        // radix-4 butterfly, I = w_T^(T/4); every subtrahend is tight or a sum of two tight values
        STAMP(3 + 4 * round);
        F29 y0, y1, y2, y3;
        {
            F29 a0 = f29_add(x0, x2), a1 = f29_sub(x0, x2), a2 = f29_add(x1, x3);
            F29 a3 = f29_mul(f29_sub(x1, x3), f29_load_raw(a.W + 12 * (size_t)Q));
            y0 = f29_add(a0, a2);
            y1 = f29_add(a1, a3);
            y2 = f29_sub2(a0, a2);
            y3 = f29_sub(a1, a3);
        }

        STAMP(4 + 4 * round);
        if (!last && (a.ablate & 2)) {
            carry_keep = f29_carry(f29_add(f29_add(y0, y1), f29_add(y2, y3)));
        } else if (!last) {
            const uint32_t d = row + ((j >> ns_log) << (ns_log + 2)) + (j & ((1u << ns_log) - 1u));
            if (!first) __syncthreads();
            lds_put29(lds, plane, d, y0);
            lds_put29(lds, plane, d + (1u << ns_log), y1);
            lds_put29(lds, plane, d + (2u << ns_log), y2);
            lds_put29(lds, plane, d + (3u << ns_log), y3);
            __syncthreads();
        } else if (valid && (!(a.ablate & 8) || y0.l[0] == 0xFFFFFFFFu)) {
            if (a.TW || a.has_scale) {
                Fe s;
#pragma unroll
                for (int l = 0; l < 8; l++) s.l[l] = a.scale[l];
                const uint32_t* tw = a.TW ? a.TW + 8 * (hi * a.twA + lo * a.twB + c * a.twC) : nullptr;
                y0 = f29_mul(y0, f29_from_fe(tw ? fe_load(tw + 8 * ((uint64_t)j * a.twS)) : s));
                y1 = f29_mul(y1, f29_from_fe(tw ? fe_load(tw + 8 * ((uint64_t)(j + Q) * a.twS)) : s));
                y2 = f29_mul(y2, f29_from_fe(tw ? fe_load(tw + 8 * ((uint64_t)(j + 2 * Q) * a.twS)) : s));
                y3 = f29_mul(y3, f29_from_fe(tw ? fe_load(tw + 8 * ((uint64_t)(j + 3 * Q) * a.twS)) : s));
            }
            uint32_t* dst = a.out + out_w * (poly + hi * a.outA + lo * a.outB + c * a.outC);
            const uint64_t s = (uint64_t)out_w * a.outS;
            if (a.out_raw) {
                f29_store_raw(dst + s * j, y0);
                f29_store_raw(dst + s * (j + Q), y1);
                f29_store_raw(dst + s * (j + 2 * Q), y2);
                f29_store_raw(dst + s * (j + 3 * Q), y3);
            } else {
                fe_store(dst + s * j, f29_to_fe(y0));
                fe_store(dst + s * (j + Q), f29_to_fe(y1));
                fe_store(dst + s * (j + 2 * Q), f29_to_fe(y2));
                fe_store(dst + s * (j + 3 * Q), f29_to_fe(y3));
            }
        }
        STAMP(5 + 4 * round);
        ns_log += 2;
    }
    }  // work items
}

template <int LOGT>
static hipError_t launch_one29(const NttPassArgs& a, unsigned tiles, unsigned polys, hipStream_t stream) {
    size_t lds = ntt29_pass_lds_bytes(LOGT, (int)a.log_c);
    unsigned threads = (1u << (LOGT - 2)) << a.log_c;
    if (threads > 1024 || lds > 160 * 1024) return hipErrorInvalidConfiguration;
    static bool attr_set[64] = {};  // per device: dynamic LDS above 64 KiB must be opted into
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev >= 64 || !attr_set[dev]) {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&ntt29_pass_kernel<LOGT>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        if (dev < 64) attr_set[dev] = true;
    }
    static int cus[64] = {};
    int ncu = dev < 64 ? cus[dev] : 0;
    if (ncu == 0) {
        e = hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
        if (e != hipSuccess) return e;
        if (dev < 64) cus[dev] = ncu;
    }
    // resident workgroups per CU: LDS, and four waves per SIMD (the kernel's register budget)
    unsigned per_cu = (unsigned)(160 * 1024 / lds);
    if (per_cu > 1024 / threads) per_cu = 1024 / threads;
    if (per_cu < 1) per_cu = 1;
    NttPassArgs b = a;
    b.tiles = tiles;
    b.polys = polys;
    b.xcd_map = (getenv("STARKHIP_XCDMAP") && tiles % 8 == 0 && tiles >= 8) ? 1u : 0u;
    const uint64_t total = (uint64_t)tiles * polys;
    // One work item per workgroup unless a persistent grid is asked for (STARKHIP_PERSISTENT=1):
    // on this kernel the hardware dispatcher's refill of finished slots measured faster than a
    // resident loop over items.
    static const bool persistent = getenv("STARKHIP_PERSISTENT") && atoi(getenv("STARKHIP_PERSISTENT")) > 0;
    uint64_t grid = persistent ? (uint64_t)ncu * per_cu : total;
    if (grid > total) grid = total;
    if (grid > 0x7FFFFFFFull) grid = 0x7FFFFFFFull;
    if (total < 4 * grid) b.stagger_cycles = 0;  // too few items per workgroup for the stagger to pay
    hipLaunchKernelGGL(ntt29_pass_kernel<LOGT>, dim3((unsigned)grid), dim3(threads), lds, stream, b);
    return hipGetLastError();
}

hipError_t launch_ntt29_pass(int log_t, const NttPassArgs& a, unsigned tiles, unsigned polys, hipStream_t stream) {
    switch (log_t) {
        case 2: return launch_one29<2>(a, tiles, polys, stream);
        case 3: return launch_one29<3>(a, tiles, polys, stream);
        case 4: return launch_one29<4>(a, tiles, polys, stream);
        case 5: return launch_one29<5>(a, tiles, polys, stream);
        case 6: return launch_one29<6>(a, tiles, polys, stream);
        case 7: return launch_one29<7>(a, tiles, polys, stream);
        case 8: return launch_one29<8>(a, tiles, polys, stream);
        case 9: return launch_one29<9>(a, tiles, polys, stream);
        case 10: return launch_one29<10>(a, tiles, polys, stream);
        case 11: return launch_one29<11>(a, tiles, polys, stream);
        case 12: return launch_one29<12>(a, tiles, polys, stream);
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
