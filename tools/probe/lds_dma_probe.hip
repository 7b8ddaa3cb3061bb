// How the direct global -> LDS loads of gfx950 place their data: every lane of a wavefront names its own global
// address; where in LDS do 4, 12 and 16 bytes per lane land relative to the base the instruction is given?
//   hipcc -O2 --offload-arch=gfx950 -o lds_dma_probe lds_dma_probe.hip && ./lds_dma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef __attribute__((address_space(1))) const void* gbl_ptr_t;

template <int BYTES>
__global__ void probe(const uint32_t* __restrict__ src, uint32_t* __restrict__ out, int lds_base_dwords, int skip_odd) {
    extern __shared__ uint32_t lds[];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    for (uint32_t i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = 0xDEAD0000u + i;
    __syncthreads();
    // lane L fetches BYTES bytes from piece (63 - L) of the source: a permutation, so placement shows
    const uint32_t* g = src + (BYTES / 4) * (63u - lane) + 1024 * wave;
    uint32_t* l = lds + lds_base_dwords + 1024 * wave;  // wave-uniform
    if (!(skip_odd && (lane & 1)))
    {
#if defined(__HIP_DEVICE_COMPILE__)
        if constexpr (BYTES == 4) __builtin_amdgcn_global_load_lds((gbl_ptr_t)g, (lds_ptr_t)l, 4, 0, 0);
        if constexpr (BYTES == 12) __builtin_amdgcn_global_load_lds((gbl_ptr_t)g, (lds_ptr_t)l, 12, 0, 0);
        if constexpr (BYTES == 16) __builtin_amdgcn_global_load_lds((gbl_ptr_t)g, (lds_ptr_t)l, 16, 0, 0);
#endif
    }
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < 4096; i += blockDim.x) out[i] = lds[i];
}

// the same with the landing high up in a 160 KiB allocation: does the base register carry more than 16 address bits?
__global__ void probe_high(const uint32_t* __restrict__ src, uint32_t* __restrict__ out, uint32_t base_dwords) {
    extern __shared__ uint32_t lds[];
    const uint32_t lane = threadIdx.x & 63u;
    for (uint32_t i = threadIdx.x; i < 40960; i += blockDim.x) lds[i] = 0xDEAD0000u + i;
    __syncthreads();
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_global_load_lds((gbl_ptr_t)(src + 4 * lane), (lds_ptr_t)(lds + base_dwords), 16, 0, 0);
#endif
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < 40960; i += blockDim.x) out[i] = lds[i];
}

// 16 bytes per lane from global addresses that are only 4-byte aligned (elements of 36 bytes back to back)
__global__ void probe_unaligned(const uint32_t* __restrict__ src, uint32_t* __restrict__ out) {
    extern __shared__ uint32_t lds[];
    const uint32_t lane = threadIdx.x & 63u;
    for (uint32_t i = threadIdx.x; i < 1024; i += blockDim.x) lds[i] = 0xDEAD0000u + i;
    __syncthreads();
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_global_load_lds((gbl_ptr_t)(src + 9 * lane), (lds_ptr_t)(lds), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((gbl_ptr_t)(src + 9 * lane + 4), (lds_ptr_t)(lds + 256), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((gbl_ptr_t)(src + 9 * lane + 8), (lds_ptr_t)(lds + 512), 4, 0, 0);
#endif
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < 1024; i += blockDim.x) out[i] = lds[i];
}

static int run_unaligned() {
    std::vector<uint32_t> h(4096), o(1024);
    for (int i = 0; i < 4096; i++) h[i] = i;
    uint32_t *d, *dout;
    hipMalloc(&d, 4096 * 4);
    hipMalloc(&dout, 1024 * 4);
    hipMemcpy(d, h.data(), 4096 * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe_unaligned, dim3(1), dim3(64), 4096, 0, d, dout);
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) { printf("unaligned: %s\n", hipGetErrorString(e)); return 1; }
    hipMemcpy(o.data(), dout, 1024 * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int L = 0; L < 64; L++) {
        for (int k = 0; k < 4; k++) {
            if (o[4 * L + k] != (uint32_t)(9 * L + k)) bad++;
            if (o[256 + 4 * L + k] != (uint32_t)(9 * L + 4 + k)) bad++;
        }
        if (o[512 + L] != (uint32_t)(9 * L + 8)) bad++;
    }
    printf("16 bytes per lane from 4-byte-aligned global addresses (36-byte elements as 16 + 16 + 4): %s (%d mismatches)\n",
           bad ? "WRONG" : "as asked", bad);
    hipFree(d);
    hipFree(dout);
    return bad != 0;
}

static int run_high(uint32_t base_dwords) {
    std::vector<uint32_t> h(4096), o(40960);
    for (int i = 0; i < 4096; i++) h[i] = i;
    uint32_t *d, *dout;
    hipMalloc(&d, 4096 * 4);
    hipMalloc(&dout, 40960 * 4);
    hipMemcpy(d, h.data(), 4096 * 4, hipMemcpyHostToDevice);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&probe_high), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(probe_high, dim3(1), dim3(64), 160 * 1024, 0, d, dout, base_dwords);
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) { printf("high landing: %s\n", hipGetErrorString(e)); return 1; }
    hipMemcpy(o.data(), dout, 40960 * 4, hipMemcpyDeviceToHost);
    int bad = 0, first_changed = -1;
    for (int i = 0; i < 40960; i++) {
        const bool in = i >= (int)base_dwords && i < (int)base_dwords + 256;
        const uint32_t want = in ? (uint32_t)(i - base_dwords) : 0xDEAD0000u + i;
        if (o[i] != want) bad++;
        if (o[i] != 0xDEAD0000u + (uint32_t)i && first_changed < 0) first_changed = i;
    }
    printf("landing at byte %u of a 160 KiB allocation: %s (first changed dword at byte %d)\n", base_dwords * 4,
           bad ? "NOT where asked" : "as asked", first_changed * 4);
    hipFree(d);
    hipFree(dout);
    return bad != 0;
}

template <int BYTES>
static int run(const char* name, int skip_odd) {
    std::vector<uint32_t> h(4096), o(4096);
    for (int i = 0; i < 4096; i++) h[i] = i;
    uint32_t *d, *dout;
    hipMalloc(&d, 4096 * 4);
    hipMalloc(&dout, 4096 * 4);
    hipMemcpy(d, h.data(), 4096 * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe<BYTES>, dim3(1), dim3(128), 4096 * 4, 0, d, dout, 8, skip_odd);
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) { printf("%s: %s\n", name, hipGetErrorString(e)); return 1; }
    hipMemcpy(o.data(), dout, 4096 * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    const int per = BYTES / 4, stride = BYTES == 12 ? 4 : per;  // 12 bytes per lane land at a 16-byte lane stride
    for (int w = 0; w < 2; w++)
        for (int L = 0; L < 64; L++)
            for (int k = 0; k < stride; k++) {
                const uint32_t got = o[1024 * w + 8 + stride * L + k];
                const bool skipped = (skip_odd && (L & 1)) || k >= per;
                const uint32_t want = skipped ? 0xDEAD0000u + 1024 * w + 8 + stride * L + k : (uint32_t)(per * (63 - L) + k + 1024 * w);
                if (got != want) {
                    if (bad < 6) printf("  %s wave %d lane %d dword %d: got %#x want %#x\n", name, w, L, k, got, want);
                    bad++;
                }
            }
    // nothing outside the landing may have changed
    for (int w = 0; w < 2; w++)
        for (int i = 0; i < 1024; i++) {
            if (i >= 8 && i < 8 + stride * 64) continue;
            if (o[1024 * w + i] != 0xDEAD0000u + 1024 * w + i) bad++;
        }
    printf("%s (%d bytes per lane%s): %s (%d mismatches): lane L lands at base + %d * L bytes\n", name, BYTES,
           skip_odd ? ", odd lanes masked off" : "", bad ? "UNEXPECTED" : "as assumed", bad, 4 * stride);
    hipFree(d);
    hipFree(dout);
    return bad != 0;
}

int main() {
    int rc = 0;
    rc |= run<4>("global_load_lds_dword", 0);
    rc |= run<12>("global_load_lds_dwordx3", 0);
    rc |= run<16>("global_load_lds_dwordx4", 0);
    rc |= run<16>("global_load_lds_dwordx4", 1);
    rc |= run<12>("global_load_lds_dwordx3", 1);
    rc |= run_unaligned();
    rc |= run_high(60000 / 4);
    rc |= run_high(70000 / 4 * 1);
    rc |= run_high(76320 / 4);
    rc |= run_high(140000 / 4);
    return rc;
}
