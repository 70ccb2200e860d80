// Register-resident MFMA peak probe ("run a register-resident MFMA-peak micro-kernel on the box
// and report utilisation against both vendor and measured peak").  No reference counterpart: measurement aid.
//
// Every wave keeps its A/B fragments and NACC independent accumulators in registers and issues back-to-back
// v_mfma_f32_16x16x32_bf16 (kind 0) or v_mfma_f32_32x32x16_bf16 (kind 1); no LDS, no memory traffic inside the
// loop.  Operands are loaded once from `seed` (caller-supplied bf16 data: random data draws more power and runs
// at lower clocks than zeros, so the caller chooses what "peak" means).
#include "hk_common.h"

template <int KIND>
__global__ __launch_bounds__(256) void hk_mfma_probe_kernel(const uint16_t* __restrict__ seed, float* __restrict__ out,
                                                            int iters) {
    const int lane = threadIdx.x;
    const bf16x8 a = *(const bf16x8*)(seed + (size_t)(lane & 255) * 8);
    const bf16x8 b = *(const bf16x8*)(seed + (size_t)((lane + 64) & 255) * 8);
    float s = 0.f;
    if constexpr (KIND == 0) {
        f32x4 acc[8];
#pragma unroll
        for (int i = 0; i < 8; i++) acc[i] = f32x4{(float)i, 0, 0, (float)a[i]};   // distinct chains (no CSE)
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int i = 0; i < 8; i++)   // asm: the builtin form gets its accumulators shuffled between AGPRs every iteration
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
        }
#pragma unroll
        for (int i = 0; i < 8; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    } else {
        f32x16 acc[4];
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 16; j++) acc[i][j] = (float)a[(i + j) & 7] * (float)(i + 1);   // distinct chains
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int i = 0; i < 4; i++)
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
        }
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 16; j++) s += acc[i][j];
    }
    out[(size_t)blockIdx.x * 256 + lane] = s;
}

// flops of one launch: grid * 4 waves * iters * (8 * 2*16*16*32 | 4 * 2*32*32*16) = grid * iters * 524288
extern "C" int hk_mfma_probe(const void* seed_bf16_2048, float* out, int32_t grid, int32_t iters, int32_t kind,
                             void* stream) {
    HK_REQUIRE(seed_bf16_2048 && out, "hk_mfma_probe: null pointer");
    HK_REQUIRE(grid > 0 && grid <= 65536 && iters > 0 && (kind == 0 || kind == 1), "hk_mfma_probe: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    if (kind == 0)
        hk_mfma_probe_kernel<0><<<grid, 256, 0, st>>>((const uint16_t*)seed_bf16_2048, out, iters);
    else
        hk_mfma_probe_kernel<1><<<grid, 256, 0, st>>>((const uint16_t*)seed_bf16_2048, out, iters);
    HK_CHECK_LAUNCH("hk_mfma_probe");
    return HK_OK;
}
