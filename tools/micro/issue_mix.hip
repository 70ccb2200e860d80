// Issue-port micro-benchmark for gfx950: how v_exp_f32, plain VALU and v_mfma_f32_32x32x16_bf16 share a SIMD.
// Every mode runs `iters` iterations of a fixed instruction block per wave; reported: ns per iteration and the
// same in cycles of a 16-MFMA block (512 matrix-pipe cycles) measured in the same process.
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/issue_mix.hip -o tools/micro/issue_mix
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define MFMA(i) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[(i) & 3]) : "v"(a), "v"(b))
#define EXP(i) asm volatile("v_exp_f32 %0, %0" : "+v"(e[(i) & 31]))
#define ADD(i) asm volatile("v_add_f32 %0, %1, %0" : "+v"(s[(i) & 3]) : "v"(e[(i) & 31]))
#define CVT(i) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(w[(i) & 15]) : "v"(e[(2 * (i)) & 31]), "v"(e[(2 * (i) + 1) & 31]))
#define DOT(i) asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(s[(i) & 1]) : "v"(one), "v"(w[(i) & 15]))

template <int MODE>
__global__ __launch_bounds__(256) void k(const uint16_t* seed, float* out, int iters) {
    const int lane = threadIdx.x;
    const bf16x8 a = *(const bf16x8*)(seed + (size_t)(lane & 255) * 8);
    const bf16x8 b = *(const bf16x8*)(seed + (size_t)((lane + 64) & 255) * 8);
    f32x16 acc[4];
    float e[32], s[4] = {0, 0, 0, 0};
    uint32_t w[16];
    const uint32_t one = 0x3f803f80u;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 16; j++) acc[i][j] = (float)a[(i + j) & 7];
    for (int i = 0; i < 32; i++) e[i] = -0.001f * (float)(lane + i);
    for (int i = 0; i < 16; i++) w[i] = 0;
    for (int it = 0; it < iters; it++) {
        if constexpr (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 16; i++) MFMA(i);
        } else if constexpr (MODE == 1) {
#pragma unroll
            for (int i = 0; i < 32; i++) EXP(i);
        } else if constexpr (MODE == 2) {
#pragma unroll
            for (int i = 0; i < 16; i++) { MFMA(i); EXP(2 * i); EXP(2 * i + 1); }
        } else if constexpr (MODE == 3) {
#pragma unroll
            for (int i = 0; i < 16; i++) MFMA(i);
#pragma unroll
            for (int i = 0; i < 32; i++) EXP(i);
        } else if constexpr (MODE == 4) {
#pragma unroll
            for (int i = 0; i < 32; i++) ADD(i);
        } else if constexpr (MODE == 5) {   // the pipelined attention body: 8 x (mfma + 4 exp + 2 cvt + 2 dot2), then 8 mfma
#pragma unroll
            for (int i = 0; i < 8; i++) { MFMA(i); EXP(4 * i); EXP(4 * i + 1); EXP(4 * i + 2); EXP(4 * i + 3); CVT(2 * i); CVT(2 * i + 1); DOT(2 * i); DOT(2 * i + 1); }
#pragma unroll
            for (int i = 8; i < 16; i++) MFMA(i);
        } else if constexpr (MODE == 6) {   // 16 x (mfma + 4 plain VALU)
#pragma unroll
            for (int i = 0; i < 16; i++) { MFMA(i); ADD(4 * i); ADD(4 * i + 1); ADD(4 * i + 2); ADD(4 * i + 3); }
        } else if constexpr (MODE == 7) {   // 16 x (mfma + 7 plain VALU): VALU fills the whole 32-cycle MFMA shadow?
#pragma unroll
            for (int i = 0; i < 16; i++) { MFMA(i); ADD(0); ADD(1); ADD(2); ADD(3); ADD(4); ADD(5); ADD(6); }
        } else if constexpr (MODE == 8) {   // 32 cvt_pk
#pragma unroll
            for (int i = 0; i < 32; i++) CVT(i);
        } else if constexpr (MODE == 9) {   // 32 dot2c
#pragma unroll
            for (int i = 0; i < 32; i++) DOT(i);
        }
    }
    float r = s[0] + s[1] + s[2] + s[3];
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 16; j++) r += acc[i][j];
    for (int i = 0; i < 32; i++) r += e[i];
    for (int i = 0; i < 16; i++) r += (float)w[i];
    out[(size_t)blockIdx.x * 256 + lane] = r;
}

template <int MODE>
float run(const uint16_t* seed, float* out, int grid, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<MODE><<<grid, 256>>>(seed, out, iters / 10);
    hipEventRecord(e0);
    k<MODE><<<grid, 256>>>(seed, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e6f / iters;   // ns per iteration
}

int main() {
    uint16_t h[2048];
    srand(1);
    for (int i = 0; i < 2048; i++) h[i] = (uint16_t)(0x3c00 + (rand() & 0x3ff) + ((rand() & 1) << 15));
    uint16_t* seed;
    float* out;
    hipMalloc(&seed, sizeof(h));
    hipMalloc(&out, 4096 * 256 * 4);
    hipMemcpy(seed, h, sizeof(h), hipMemcpyHostToDevice);
    const char* names[] = {"16 mfma", "32 exp", "16x(mfma+2exp)", "16 mfma; 32 exp", "32 add", "attn body", "16x(mfma+4add)", "16x(mfma+7add)", "32 cvt_pk", "32 dot2c"};
    for (int grid : {8, 256, 512, 768}) {
        const int iters = 20000;
        float t[10];
        t[0] = run<0>(seed, out, grid, iters);
        t[1] = run<1>(seed, out, grid, iters);
        t[2] = run<2>(seed, out, grid, iters);
        t[3] = run<3>(seed, out, grid, iters);
        t[4] = run<4>(seed, out, grid, iters);
        t[5] = run<5>(seed, out, grid, iters);
        t[6] = run<6>(seed, out, grid, iters);
        t[7] = run<7>(seed, out, grid, iters);
        t[8] = run<8>(seed, out, grid, iters);
        t[9] = run<9>(seed, out, grid, iters);
        const float waves = grid <= 256 ? 1.f : grid / 256.f;
        for (int m = 0; m < 10; m++)
            printf("grid %4d  %-18s %8.1f ns/iter  = %7.0f cycles (16 mfma alone at this grid = %d x 512)\n", grid, names[m], t[m],
                   t[m] / t[0] * 512.f * waves, (int)waves);
        fflush(stdout);
    }
    return 0;
}
