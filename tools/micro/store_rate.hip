// Micro-benchmark: how fast can one CU issue / retire 16-byte-per-lane global stores?
// Each wave issues NST stores of 1 KB (64 lanes x 16 B) to its own region; lane 0 records the cycles
// until the last store has ISSUED and until all have COMPLETED (vmcnt(0)).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
template <int NST>
__global__ void k(u32x4* out, unsigned long long* stamps, int stride_elems) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    u32x4* p = out + ((size_t)(blockIdx.x * nw + wave) * NST) * stride_elems + lane;
    u32x4 v = {1u, 2u, 3u, (unsigned)lane};
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int i = 0; i < NST; ++i) p[(size_t)i * stride_elems] = v;
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned long long t2 = __builtin_amdgcn_s_memtime();
    if (lane == 0) {
        stamps[(blockIdx.x * nw + wave) * 2] = t1 - t0;
        stamps[(blockIdx.x * nw + wave) * 2 + 1] = t2 - t0;
    }
}
int main() {
    const int NST = 16;
    for (int threads : {256, 512, 1024}) for (int grid : {256, 512, 1024}) {
        int nw = threads / 64;
        size_t nwaves = (size_t)grid * nw;
        u32x4* out; unsigned long long* st;
        hipMalloc(&out, nwaves * NST * 1024); hipMalloc(&st, nwaves * 16);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k<NST>, dim3(grid), dim3(threads), 0, 0, out, st, 64);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(nwaves * 2);
        hipMemcpy(h.data(), st, nwaves * 16, hipMemcpyDeviceToHost);
        std::vector<unsigned long long> a, b;
        for (size_t i = 0; i < nwaves; ++i) { a.push_back(h[2*i]); b.push_back(h[2*i+1]); }
        std::sort(a.begin(), a.end()); std::sort(b.begin(), b.end());
        double bytes = (double)nwaves * NST * 1024;
        printf("threads %4d grid %4d: kernel %.1f us (%.2f TB/s) | issue cycles med %llu p90 %llu | complete med %llu p90 %llu | B/clk/CU(issue med) %.1f\n",
               threads, grid, ms * 1e3, bytes / ms / 1e9, a[a.size()/2], a[a.size()*9/10], b[b.size()/2], b[b.size()*9/10],
               (double)nw * NST * 1024 / (double)b[b.size()/2]);
        hipFree(out); hipFree(st);
    }
    return 0;
}
