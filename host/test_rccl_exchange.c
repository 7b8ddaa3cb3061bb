/* One transform through stark_ntt_sharded_dev with the exchange done by RCCL from a C host: a one-rank communicator
 * (ncclCommInitRank, world 1) and the grouped ncclSend / ncclRecv all-to-all a multi-GPU host would pass in, compared
 * byte for byte with the single-call transform stark_ntt_batch_dev (the reference's _fft, src/fft.rs:37-61, and fft_inv,
 * :64-86). This is the exchange a Rust host would issue through the same C ABI (the reference is a cdylib,
 * Cargo.toml:21-22). Plain C on purpose: it is the text of the binding example, compiled.
 *
 * Build (make -C stark_verifier_amd/csrc host):
 *   gcc -std=c11 -O2 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include host/test_rccl_exchange.c \
 *       -Lstark_verifier_amd -lstarkhip -L/opt/rocm/lib -lrccl -lamdhip64 -o host/test_rccl_exchange
 * Usage: test_rccl_exchange [log_n ...]   (default: 16 24)
 */
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/starkhip.h"

struct exchange_state {
    ncclComm_t comm;
    int world;
    int calls;
};

/* block d of `send` goes to rank d, block s of `recv` comes from rank s: one send and one receive per peer inside one
 * RCCL group, so on a full node every xGMI link carries one block at the same time */
static int rccl_all_to_all(void* user, const void* send, void* recv, size_t block_bytes, void* stream) {
    struct exchange_state* ex = (struct exchange_state*)user;
    ex->calls++;
    if (ncclGroupStart() != ncclSuccess) return 1;
    for (int peer = 0; peer < ex->world; peer++) {
        if (ncclSend((const char*)send + (size_t)peer * block_bytes, block_bytes, ncclUint8, peer, ex->comm, (hipStream_t)stream) != ncclSuccess)
            return 1;
        if (ncclRecv((char*)recv + (size_t)peer * block_bytes, block_bytes, ncclUint8, peer, ex->comm, (hipStream_t)stream) != ncclSuccess)
            return 1;
    }
    return ncclGroupEnd() == ncclSuccess ? 0 : 1;
}

#define HIPCHK(x)                                                                         \
    do {                                                                                  \
        hipError_t e_ = (x);                                                              \
        if (e_ != hipSuccess) {                                                           \
            printf("FAIL: %s: %s\n", #x, hipGetErrorString(e_));                          \
            return 1;                                                                     \
        }                                                                                 \
    } while (0)
#define STARKCHK(x)                                                                       \
    do {                                                                                  \
        int r_ = (x);                                                                     \
        if (r_ != STARK_OK) {                                                             \
            printf("FAIL: %s = %d: %s\n", #x, r_, stark_last_error(ctx));                 \
            return 1;                                                                     \
        }                                                                                 \
    } while (0)

/* 7^((p-1)/2^k) for k = 16 and 24 (reference src/lib.rs:114 derives its root of unity the same way), big-endian hex */
static const char* ROOT16 = "5caa52206ecc9d72853f7c616e06672484967c1e2e5a87d6922259c0982b76cc";
static const char* ROOT24 = "6b7f42cee2b0355e3c9637dea849609e83b5488d1eb2644e1729a9dc31f6259e";

static void root_limbs(const char* hex, uint32_t out[8]) {
    uint8_t be[32];
    for (int i = 0; i < 32; i++) {
        unsigned v;
        sscanf(hex + 2 * i, "%2x", &v);
        be[i] = (uint8_t)v;
    }
    stark_be32_to_limbs(be, 1, out);
}

static int run(stark_ctx* ctx, struct exchange_state* ex, unsigned log_n, hipStream_t st) {
    const size_t n = (size_t)1 << log_n, bytes = n * 32;
    const size_t n1 = (size_t)1 << (log_n / 2), n2 = n / n1;
    uint32_t root[8];
    root_limbs(log_n == 16 ? ROOT16 : ROOT24, root);
    void *d_nat, *d_rows, *d_other, *d_ref;
    HIPCHK(hipMalloc(&d_nat, bytes));
    HIPCHK(hipMalloc(&d_rows, bytes));
    HIPCHK(hipMalloc(&d_other, bytes));
    HIPCHK(hipMalloc(&d_ref, bytes));
    uint8_t* h_a = (uint8_t*)malloc(bytes);
    uint8_t* h_b = (uint8_t*)malloc(bytes);
    if (!h_a || !h_b) return 1;
    for (int inverse = 0; inverse <= 1; inverse++) {
        STARKCHK(stark_fill_random_field_dev(ctx, d_nat, n, 0x5354524BULL + log_n + (uint64_t)inverse, st));
        /* the reference result: one call on the natural-order vector */
        STARKCHK(stark_ntt_batch_dev(ctx, d_nat, d_ref, log_n, 1, root, inverse, st));
        /* the single rank's "rows" slice [n1][n2], element (a, j2) = x[a + n1 * j2]: the transpose of x seen as [n2][n1] */
        STARKCHK(stark_permute3_dev(ctx, d_nat, d_rows, n2, n1, 1, st));
        const int before = ex->calls;
        STARKCHK(stark_ntt_sharded_dev(ctx, d_rows, d_other, log_n, 1, 0, root, inverse, rccl_all_to_all, ex, st));
        HIPCHK(hipStreamSynchronize(st));
        if (ex->calls != before + 1) {
            printf("FAIL: the exchange ran %d times\n", ex->calls - before);
            return 1;
        }
        /* with one rank the "columns" result [n1][n2] is the natural order; it is in the second buffer because an
         * exchange ran */
        HIPCHK(hipMemcpy(h_a, d_other, bytes, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(h_b, d_ref, bytes, hipMemcpyDeviceToHost));
        if (memcmp(h_a, h_b, bytes) != 0) {
            size_t k = 0;
            while (h_a[k] == h_b[k]) k++;
            printf("FAIL: 2^%u %s differs from stark_ntt_batch_dev at byte %zu\n", log_n, inverse ? "inverse" : "forward", k);
            return 1;
        }
        /* the same through the two stage calls with the exchange issued by this host between them */
        STARKCHK(stark_permute3_dev(ctx, d_nat, d_rows, n2, n1, 1, st));
        STARKCHK(stark_ntt_sharded_stage_a_dev(ctx, d_rows, d_other, log_n, 1, 0, root, inverse, st));
        if (rccl_all_to_all(ex, d_other, d_rows, bytes, st) != 0) {
            printf("FAIL: exchange\n");
            return 1;
        }
        STARKCHK(stark_ntt_sharded_stage_c_dev(ctx, d_rows, d_other, log_n, 1, root, inverse, st));
        HIPCHK(hipStreamSynchronize(st));
        HIPCHK(hipMemcpy(h_a, d_other, bytes, hipMemcpyDeviceToHost));
        if (memcmp(h_a, h_b, bytes) != 0) {
            printf("FAIL: 2^%u %s (stage calls) differs from stark_ntt_batch_dev\n", log_n, inverse ? "inverse" : "forward");
            return 1;
        }
        printf("ok 2^%u %s: %zu bytes through ncclSend/ncclRecv, byte-identical to stark_ntt_batch_dev\n", log_n,
               inverse ? "inverse" : "forward", bytes);
    }
    free(h_a);
    free(h_b);
    HIPCHK(hipFree(d_nat));
    HIPCHK(hipFree(d_rows));
    HIPCHK(hipFree(d_other));
    HIPCHK(hipFree(d_ref));
    return 0;
}

int main(int argc, char** argv) {
    stark_ctx* ctx = NULL;
    HIPCHK(hipSetDevice(0));
    struct exchange_state ex = {0};
    ncclUniqueId id;
    if (ncclGetUniqueId(&id) != ncclSuccess || ncclCommInitRank(&ex.comm, 1, id, 0) != ncclSuccess) {
        printf("FAIL: one-rank RCCL communicator\n");
        return 1;
    }
    ex.world = 1;
    if (stark_ctx_create(0, &ctx) != STARK_OK) {
        printf("FAIL: stark_ctx_create\n");
        return 1;
    }
    hipStream_t st = (hipStream_t)stark_ctx_stream(ctx);
    int rc = 0;
    if (argc > 1) {
        for (int i = 1; i < argc && !rc; i++) rc = run(ctx, &ex, (unsigned)atoi(argv[i]), st);
    } else {
        rc = run(ctx, &ex, 16, st);
        if (!rc) rc = run(ctx, &ex, 24, st);
    }
    stark_ctx_destroy(ctx);
    ncclCommDestroy(ex.comm);
    if (!rc) printf("PASS\n");
    return rc;
}
