// Micro-benchmark: one 512-thread workgroup per CU stores a 256 x 256 bf16 tile (8 waves x 16 stores of 16 B per lane)
// into a row-major matrix with a 6 KB row pitch, as the GEMM epilogue does.  PAT 0: one store = 8 rows x 128 B
// (the LDS-transposed epilogue); PAT 1: one store = 16 rows x 64 B (what an in-register lane transpose would give).
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/store_pattern.hip -o tools/micro/store_pattern
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
template <int PAT>
__global__ __launch_bounds__(512) void k(char* out, int ld_bytes, int reps) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    u32x4 v = {1u, 2u, 3u, (unsigned)lane};
    for (int rep = 0; rep < reps; ++rep) {
        // tile (blockIdx.x + rep * gridDim.x): 256 rows; 12 tile columns per tile row
        const int t = blockIdx.x + rep * gridDim.x;
        const int tm = t / 12, tn = t % 12;
        char* base = out + ((size_t)tm * 256 + wm * 128) * ld_bytes + tn * 512 + wn * 128;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            char* p;
            if (PAT == 0) p = base + (size_t)(i * 8 + (lane >> 3)) * ld_bytes + (lane & 7) * 16;
            else p = base + (size_t)((i >> 1) * 16 + (lane >> 2)) * ld_bytes + (i & 1) * 64 + (lane & 3) * 16;
            __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(p));
        }
    }
}
int main() {
    const int ld = 6144, reps = 18, grid = 256;
    char* out;
    const size_t bytes = (size_t)((grid * reps + 11) / 12) * 256 * ld;
    hipMalloc(&out, bytes);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int round = 0; round < 3; ++round)
        for (int pat = 0; pat < 2; ++pat) {
            hipEventRecord(e0);
            if (pat == 0) k<0><<<grid, 512>>>(out, ld, reps); else k<1><<<grid, 512>>>(out, ld, reps);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            printf("pattern %d: %.1f us for %d tiles per CU = %.2f us per tile, %.2f TB/s\n", pat, ms * 1e3, reps, ms * 1e3 / reps,
                   (double)grid * reps * 131072 / ms / 1e9);
        }
    return 0;
}
