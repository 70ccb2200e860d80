// Weight gradient of a Linear layer in ONE launch (row f2, the stage-3 step; tasks/train_it.py:63-80 runs
// loss.backward() through nn.Linear's autograd):
//
//     dW[n][k] = sum_r dY[r][n] * X[r][k]          db[n] = sum_r dY[r][n]
//
// Both operands are row-major with the CONTRACTION index r as their row index, i.e. neither is contraction-contiguous
// the way an MFMA fragment wants it.  Round 2 made transposed copies (hk_transpose_bf16 of dY and of X), ran the
// inference GEMM on them and took the bias gradient with hk_colsum: four launches per Linear, ~360 of the backward's
// 504.  Here the tiles are staged row-major by LDS-DMA and read back with ds_read_b64_tr_b16, gfx950's transposing LDS
// read: per 16-lane group it gathers a block of 4 rows x 16 columns and hands lane i column i -- four consecutive
// contraction elements of one output row / column, exactly half an operand fragment of v_mfma_f32_16x16x32_bf16.
//
// Workgroup = 4 waves on a 128 (n) x 128 (k) tile of dW, a wave owns 64 x 64; contraction in steps of 32 rows.
// LDS stage = dY tile [32][128] + X tile [32][128] bf16, 256-byte rows, 16 KB; ring of 4 stages (64 KB: two
// workgroups per CU), LDS-DMA three steps ahead, counted vmcnt, one barrier per step.  Bank swizzle (applied to the
// DMA source address, the destination being lane-linear): physical 16-byte chunk pc of row r holds logical chunk
// pc ^ f(r), f(r) = 2 * ((r & 3) | ((r >> 3) & 1) << 2): the 8 rows a 32-lane half reads in one transposed read
// (4 consecutive rows of two 8-row groups) land on 8 distinct 32-byte bank groups.
// The MFMA is issued with the X fragment as its A operand, so a lane ends up with 4 consecutive k of one row n of dW
// (16-byte fp32 stores).  Rows past R in the last step: their addresses are clamped to row R - 1 (finite data) and the
// dY fragment elements are zeroed in registers.  The bias gradient rides along as one more MFMA per 16 columns of dY
// against an all-ones fragment, on the workgroups of the first k tile only; everything is summed in a fixed order
// (no atomics): the result is bit-reproducible.
#include "hk_common.h"

namespace {

typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;

struct WgradParams {
    const __bf16* dY;
    const __bf16* X;
    float* dW;
    float* db;
    int64_t ldy, ldx, x_rpb, x_bs, lddw;
    int R, N, K;
};

__device__ __forceinline__ int64_t x_off(const WgradParams& p, int r) {
    if (p.x_rpb == 0) return (int64_t)r * p.ldx;
    const int b = r / (int)p.x_rpb;
    return (int64_t)b * p.x_bs + (int64_t)(r - b * (int)p.x_rpb) * p.ldx;
}

constexpr int WG_NS = 4;                       // ring stages
constexpr int WG_TILE = 32 * 128;              // elements of one operand tile of a stage (8 KB)
constexpr int WG_STAGE = 2 * WG_TILE;          // dY tile then X tile

__global__ __launch_bounds__(256) void hk_wgrad_kernel(const WgradParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __bf16* lds = reinterpret_cast<__bf16*>(smem);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave >> 1, wk = wave & 1;
    const int nkt = p.K / 128;
    const int tn = blockIdx.x / nkt, tk = blockIdx.x - tn * nkt;
    const int n0 = tn * 128, k0 = tk * 128;
    const int steps = (p.R + 31) / 32;

    // ---- loader: a piece = 4 rows x 256 B; wave w stages pieces w and w + 4 of each operand tile ----
    const int l_r = lane >> 4, l_pc = lane & 15;
    auto f_of = [](int r) { return ((r & 3) | (((r >> 3) & 1) << 2)) << 1; };
    auto issue = [&](int s) {
        __bf16* dst = lds + (s % WG_NS) * WG_STAGE;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int piece = wave + 4 * i;
            const int r = piece * 4 + l_r;
            const int lc = l_pc ^ f_of(r);
            const int rr = min(s * 32 + r, p.R - 1);
            glds16_v(p.dY + (int64_t)rr * p.ldy + n0 + lc * 8, dst + piece * 4 * 128);
            glds16_v(p.X + x_off(p, rr) + k0 + lc * 8, dst + WG_TILE + piece * 4 * 128);
        }
    };

    // ---- transposed fragment reads: lane = 16 g + 4 q + pp reads row 8 g + 4 h + q, columns 4 pp .. 4 pp + 3 of a
    // 16-column block; the two reads h = 0, 1 give contraction elements 8 g .. 8 g + 7 of column (lane & 15) ----
    const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    const int fr = 8 * g + q;                                  // row for h = 0; h = 1 is 4 rows (1024 B) further, same f
    const int fsw = f_of(fr);
    auto frag = [&](const __bf16* tile, int col16) -> bf16x8 {  // col16: 16-column block index inside the 128-wide tile
        const int chunk = ((col16 * 2) ^ fsw) + (pp >> 1);
        const char* a = reinterpret_cast<const char*>(tile) + fr * 256 + chunk * 16 + (pp & 1) * 8;
        const bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4_t*)(a));
        const bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4_t*)(a + 1024));
        return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    };

    f32x4 acc[4][4], accb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        accb[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const bool bias_wave = p.db != nullptr && tk == 0 && wk == 0;
    const __bf16 one = (__bf16)1.0f;
    const bf16x8 ones = bf16x8{one, one, one, one, one, one, one, one};

#pragma unroll
    for (int s = 0; s < WG_NS - 1; ++s)
        if (s < steps) issue(s);

    for (int s = 0; s < steps; ++s) {
        const int younger = min(WG_NS - 2, steps - 1 - s);     // stages requested after stage s
        if (younger >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (younger == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                           // stage s landed for everyone; step s - 1 fully read
        if (s + WG_NS - 1 < steps) issue(s + WG_NS - 1);
        const __bf16* ty = lds + (s % WG_NS) * WG_STAGE;
        const __bf16* tx = ty + WG_TILE;
        bf16x8 ay[4], bx[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) ay[i] = frag(ty, wn * 4 + i);
#pragma unroll
        for (int j = 0; j < 4; ++j) bx[j] = frag(tx, wk * 4 + j);
        if (s == steps - 1 && (p.R & 31) != 0) {
            // rows >= R were read from row R - 1: zero those contraction elements of dY (element e of this lane = row 8 g + e)
            const int nvalid = min(max(p.R - (s * 32 + 8 * g), 0), 8);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                u32x4 w = __builtin_bit_cast(u32x4, ay[i]);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const uint32_t m = (2 * e < nvalid ? 0x0000FFFFu : 0u) | (2 * e + 1 < nvalid ? 0xFFFF0000u : 0u);
                    w[e] &= m;
                }
                ay[i] = __builtin_bit_cast(bf16x8, w);
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bx[j], ay[i], acc[i][j], 0, 0, 0);
        if (bias_wave) {
#pragma unroll
            for (int i = 0; i < 4; ++i) accb[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, ay[i], accb[i], 0, 0, 0);
        }
    }

    // ---- store: lane holds dW[n = block + (lane & 15)][k = block + 4 (lane >> 4) .. + 3] ----
    const int c = lane & 15, r4 = lane >> 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int n = n0 + wn * 64 + i * 16 + c;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = k0 + wk * 64 + j * 16 + r4 * 4;
            *reinterpret_cast<f32x4*>(p.dW + (int64_t)n * p.lddw + k) = acc[i][j];
        }
        if (bias_wave && r4 == 0) p.db[n] = accb[i][0];
    }
}

}  // namespace

extern "C" int hk_wgrad_bf16(const void* dy, int64_t ldy, const void* x, const hk_rows* x_rows, float* dw, int64_t lddw,
                             float* db, int64_t R, int32_t N, int32_t K, void* stream) {
    HK_REQUIRE(dy && x && dw, "hk_wgrad_bf16: null operand pointer");
    HK_REQUIRE(R > 0 && R < (1LL << 31) && N > 0 && K > 0, "hk_wgrad_bf16: empty problem R=%ld N=%d K=%d", (long)R, N, K);
    HK_REQUIRE(N % 128 == 0 && K % 128 == 0, "hk_wgrad_bf16: N=%d and K=%d must be multiples of 128", N, K);
    HK_REQUIRE(ldy >= N && ldy % 8 == 0 && lddw >= K && lddw % 4 == 0, "hk_wgrad_bf16: bad leading dimensions");
    HK_REQUIRE(((uintptr_t)dy | (uintptr_t)x | (uintptr_t)dw) % 16 == 0 && (uintptr_t)db % 4 == 0,
               "hk_wgrad_bf16: operands must be 16-byte aligned");
    WgradParams p;
    p.dY = reinterpret_cast<const __bf16*>(dy);
    p.X = reinterpret_cast<const __bf16*>(x);
    p.dW = dw;
    p.db = db;
    p.ldy = ldy;
    p.lddw = lddw;
    p.ldx = x_rows ? x_rows->ld : K;
    p.x_rpb = x_rows ? x_rows->rows_per_batch : 0;
    p.x_bs = x_rows ? x_rows->batch_stride : 0;
    HK_REQUIRE(p.ldx >= K && p.ldx % 8 == 0 && p.x_rpb >= 0 && p.x_bs % 8 == 0, "hk_wgrad_bf16: bad row mapping of x");
    p.R = (int)R;
    p.N = N;
    p.K = K;
    constexpr int LDS = WG_NS * WG_STAGE * 2;   // 64 KB
    static std::atomic<unsigned long long> attr_done{0};
    const hipError_t e = hk_lds_attr_once(attr_done, reinterpret_cast<const void*>(&hk_wgrad_kernel), LDS);
    if (e != hipSuccess) {
        hk_set_error("hk_wgrad_bf16: hipFuncSetAttribute(%d B LDS): %s", LDS, hipGetErrorString(e));
        return HK_ELAUNCH;
    }
    const long nwg = (long)(N / 128) * (K / 128);
    HK_REQUIRE(nwg < (1L << 31), "hk_wgrad_bf16: grid too large");
    hipLaunchKernelGGL(hk_wgrad_kernel, dim3((unsigned)nwg), dim3(256), LDS, reinterpret_cast<hipStream_t>(stream), p);
    HK_CHECK_LAUNCH("hk_wgrad_bf16");
    return HK_OK;
}
