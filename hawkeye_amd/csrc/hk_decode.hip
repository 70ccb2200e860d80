// Row f1 continued: the token-by-token half of `generate` that follows the prefill -- one new token per
// sequence and step through the 32 decoder layers with the K/V cache (models/blip2/modeling_llama.py:176-231 with
// past_key_value, :744-773; called from tasks/test.py:121-133 with do_sample=False, num_beams=1).
//
// With 1-16 rows every projection is a weight-streaming problem (13.2 GB of bf16 weights per token for Vicuna-7B): the
// bound is HBM, not the matrix pipe.  hk_gemv_bf16 reads each weight row exactly once in whole 128-byte lines, keeps many
// independent loads in flight per wave, and uses v_mfma_f32_16x16x32_bf16 only as a cheap 16-row dot-product engine
// (rows past M are zero operands; the wasted flops are free under the memory bound).
#include "hk_common.h"

namespace {

// weight loads: streamed once per token, marked non-temporal (-DHK_GEMV_NT=0: plain loads, A/B aid)
#ifndef HK_GEMV_NT
#define HK_GEMV_NT 1
#endif
#if HK_GEMV_NT
#define HK_WLOAD(p) __builtin_nontemporal_load(p)
#else
#define HK_WLOAD(p) (*(p))
#endif

struct GemvParams {
    const void* x;          // bf16 [M][K], or fp32 when norm_w is set
    const __bf16* W;
    void* out;
    const float* res;
    const float* norm_w;    // RMSNorm weight [K] or null
    int M, N, K;
    int64_t ldx, ldo;
    int out_f32, act;
    float eps;
    int tiled;              // W is hk_gemv_desc.w_tiled's block layout
};

__device__ __forceinline__ float silu_mul_d(const float g, const float u) {   // the GEMM epilogue's form (hk_gemm.hip)
    const float e = __builtin_amdgcn_exp2f(g * -1.4426950408889634f);
    return g * __builtin_amdgcn_rcpf(1.0f + e) * u;
}

// y[m][n] = sum_k x[m][k] W[n][k].  One workgroup = 16 rows of W (one MFMA row block) x KS waves, wave w takes the K-blocks
// [w * per, (w + 1) * per) of 64 elements.  Inside a 64-element block lane (i = lane & 15, q = lane >> 4) reads the 32
// contiguous bytes k = 16 q .. 16 q + 15 of W row n0 + i -- four lanes cover one 128-byte line -- and the same k of x row i
// (EXEC-masked to i < M, zero otherwise); the two 8-element halves are the A / B operands of two MFMAs (any k order works
// for a dot product as long as both operands use the same one).  The accumulator of lane (j = lane & 15, q) is
// y[m = j][n0 + 4 q .. 4 q + 3].  The K slices meet in LDS and are summed in wave order (deterministic).
// W loads run one group of U K-blocks ahead of the MFMAs (two groups = 16 x 16 B per lane in flight).
// NORM: x is the fp32 residual stream and the operand is RMSNorm(x) * norm_w (modeling_llama.py:57-75) rounded to bf16 --
// hk_rmsnorm's arithmetic, without its launch and its [M, K] round trip: every workgroup sums the squares of the M <= 4
// rows itself (16 KB each from L2, while its first weight loads are in flight) and normalises the fragments as it loads them.
// KEEP (NORM only): rows whose fp32 chunks stay in registers between the statistics pass and the normalising pass (1 for M = 1,
// 4 for M = 2..4 when a thread holds <= 4 chunks per row; 0 = the second pass reads x again: 176 MB of L2 reads per gate|up launch
// at M = 4, as much as the weights themselves).
template <int KS, int MODE, int KEEP = 0>   // MODE 0: x fragments from global memory; 1: bf16 x staged in LDS; 2: RMSNorm(x) staged in LDS
__global__ __launch_bounds__(KS * 64) void hk_gemv_kernel(const GemvParams p) {
    constexpr bool NORM = MODE == 2, XLDS = MODE != 0;
    __shared__ f32x4 red[KS][64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 15, q = lane >> 4;
    const int n0 = blockIdx.x * 16;
    const int nblk = p.K >> 6;
    const int per = (nblk + KS - 1) / KS;
    const int b0 = wave * per, b1 = min(nblk, b0 + per);
    const bool mvalid = i < p.M;
    const int64_t koff = q * 16 + (int64_t)b0 * 64;
    // row-major W: lane (i, q) reads 2 x 16 B at k = 16 q of row n0 + i, 64 elements further per K-block (rows past N: a copy of
    // the last, never stored).  Tiled W (w_tiled): the 16 x 64 block of (row block, K-block) is 2 KB contiguous, stored as the two
    // 1 KB wave-loads themselves ([half][lane][8 elements]) -- a wave streams one contiguous region of memory.
    const int wstep = p.tiled ? 1024 : 64, whalf = p.tiled ? 512 : 8;
    const __bf16* wp = p.tiled ? p.W + ((int64_t)blockIdx.x * nblk + b0) * 1024 + lane * 8
                               : p.W + (int64_t)min(n0 + i, p.N - 1) * p.K + koff;
    const __bf16* xb = reinterpret_cast<const __bf16*>(p.x) + (int64_t)(mvalid ? i : 0) * p.ldx + koff;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const bf16x8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
#ifndef HK_GEMV_U
#define HK_GEMV_U 4
#endif
    constexpr int U = HK_GEMV_U;   // K-blocks per group
    bf16x8 wc[U][2];
    auto load_w = [&](bf16x8 (&w)[U][2], const __bf16* src) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            w[u][0] = HK_WLOAD(reinterpret_cast<const bf16x8*>(src + u * wstep));
            w[u][1] = HK_WLOAD(reinterpret_cast<const bf16x8*>(src + u * wstep + whalf));
        }
    };
    int b = b0;
    extern __shared__ __attribute__((aligned(16))) char xs_raw[];   // NORM: the normalised rows in bf16, [M][K + 64]
    const int xld = p.K + 64;                                       // row stride (elements): rows 128 B apart in the banks
    if constexpr (NORM) {
        // Row statistics, then bf16(norm_w * (x * rinv)) of all M rows into LDS: the main loop reads its B operands from
        // there, so its only global loads are the weights.  The statistics' loads are older than the weight loads below and
        // return first; the second pass over x hits L2.
        __shared__ float ssq[KS][4];
        constexpr int NV = 16 / KS > 0 ? 16 / KS : 1;   // K <= NV * KS * 256 = 4096 elements, checked by the host
        float ss[4] = {0.f, 0.f, 0.f, 0.f};
        f32x4 kept[KEEP > 0 ? KEEP : 1][NV];
#pragma unroll
        for (int m = 0; m < 4; ++m)
            if (m < p.M) {
                f32x4 v[NV];
#pragma unroll
                for (int j = 0; j < NV; ++j) {
                    const int k = (j * KS * 64 + tid) * 4;
                    v[j] = k < p.K ? *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(p.x) + (int64_t)m * p.ldx + k)
                                   : f32x4{0.f, 0.f, 0.f, 0.f};
                    if (m < KEEP) kept[m][j] = v[j];
                }
#pragma unroll
                for (int j = 0; j < NV; ++j) ss[m] += v[j][0] * v[j][0] + v[j][1] * v[j][1] + v[j][2] * v[j][2] + v[j][3] * v[j][3];
            }
        if (b + U <= b1) load_w(wc, wp);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) ss[m] += __shfl_xor(ss[m], o, 64);
        }
        if (lane == 0) {
#pragma unroll
            for (int m = 0; m < 4; ++m) ssq[wave][m] = ss[m];
        }
        __syncthreads();
        __bf16* xs = reinterpret_cast<__bf16*>(xs_raw);
        float rinv[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            float tot = 0.f;
#pragma unroll
            for (int w = 0; w < KS; ++w) tot += ssq[w][m];
            rinv[m] = rsqrtf(tot / (float)p.K + p.eps);
        }
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int k = (j * KS * 64 + tid) * 4;
            if (k < p.K) {
                const f32x4 g = *reinterpret_cast<const f32x4*>(p.norm_w + k);      // once per chunk, not per row
#pragma unroll
                for (int m = 0; m < 4; ++m)
                    if (m < p.M) {
                        const f32x4 v = m < KEEP ? kept[m][j]
                                                 : *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(p.x) + (int64_t)m * p.ldx + k);
                        u32x2 o;
                        o[0] = pack_bf16x2(g[0] * (v[0] * rinv[m]), g[1] * (v[1] * rinv[m]));
                        o[1] = pack_bf16x2(g[2] * (v[2] * rinv[m]), g[3] * (v[3] * rinv[m]));
                        *reinterpret_cast<u32x2*>(xs + m * xld + k) = o;
                    }
            }
        }
        __syncthreads();
    } else if constexpr (XLDS) {
        // the operand rows -> LDS (their loads are older than the weight loads and return first): the main loop's only
        // global loads are then the weights, and no MFMA waits on an x fragment queued behind them
        __bf16* xs = reinterpret_cast<__bf16*>(xs_raw);
        const int cpr = p.K >> 3, total = p.M * cpr;          // 16-byte chunks per row / in all
        for (int c0 = tid; c0 < total; c0 += 4 * KS * 64) {
            u32x4 v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c = min(c0 + j * KS * 64, total - 1), m = c / cpr;
                v[j] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const __bf16*>(p.x) + (int64_t)m * p.ldx + (c - m * cpr) * 8);
            }
            if (c0 == tid && b + U <= b1) load_w(wc, wp);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c = c0 + j * KS * 64, m = c / cpr;
                if (c < total) *reinterpret_cast<u32x4*>(xs + m * xld + (c - m * cpr) * 8) = v[j];
            }
        }
        if (tid >= total && b + U <= b1) load_w(wc, wp);      // threads without a chunk still start their weight loads
        __syncthreads();
    } else {
        if (b + U <= b1) load_w(wc, wp);
    }
    const __bf16* xl = reinterpret_cast<const __bf16*>(xs_raw) + (mvalid ? i : 0) * xld + koff;
    // the B operand of one 64-element block: bf16 x from global memory, or the normalised row from LDS
    auto load_x = [&](bf16x8& x0, bf16x8& x1, int u) __attribute__((always_inline)) {
        x0 = zero;
        x1 = zero;
        if (mvalid) {
            if constexpr (XLDS) {
                x0 = *reinterpret_cast<const bf16x8*>(xl + u * 64);
                x1 = *reinterpret_cast<const bf16x8*>(xl + u * 64 + 8);
            } else {
                x0 = *reinterpret_cast<const bf16x8*>(xb + u * 64);
                x1 = *reinterpret_cast<const bf16x8*>(xb + u * 64 + 8);
            }
        }
    };
    for (; b + U <= b1; b += U) {
        bf16x8 wn[U][2], xv[U][2];
        const bool more = b + 2 * U <= b1;      // wave-uniform
        if (more) load_w(wn, wp + U * wstep);
#pragma unroll
        for (int u = 0; u < U; ++u) load_x(xv[u][0], xv[u][1], u);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wc[u][0], xv[u][0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wc[u][1], xv[u][1], acc, 0, 0, 0);
        }
        if (more) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                wc[u][0] = wn[u][0];
                wc[u][1] = wn[u][1];
            }
        }
        wp += U * wstep;
        xb += U * 64;
        xl += U * 64;
    }
    for (; b < b1; ++b) {
        const bf16x8 w0 = HK_WLOAD(reinterpret_cast<const bf16x8*>(wp));
        const bf16x8 w1 = HK_WLOAD(reinterpret_cast<const bf16x8*>(wp + whalf));
        bf16x8 x0, x1;
        load_x(x0, x1, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0, x0, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1, x1, acc, 0, 0, 0);
        wp += wstep;
        xb += 64;
        xl += 64;
    }
    if (KS > 1) {
        red[wave][lane] = acc;
        __syncthreads();
        if (wave != 0) return;
        acc = red[0][lane];
#pragma unroll
        for (int s = 1; s < KS; ++s) acc += red[s][lane];
    }
    if (!mvalid) return;          // accumulator column j = lane & 15 is row m of x
    const int m = i, n = n0 + 4 * q;
    if (p.act == HK_ACT_SWIGLU) {  // interleaved (gate, up) rows of W -> out[m][n / 2], two values per lane
        *reinterpret_cast<uint32_t*>(reinterpret_cast<__bf16*>(p.out) + (int64_t)m * p.ldo + (n >> 1)) =
            pack_bf16x2(silu_mul_d(acc[0], acc[1]), silu_mul_d(acc[2], acc[3]));
        return;
    }
    if (n + 4 > p.N) {             // last row block of an N that is not a multiple of 4 (a vocabulary of 32001): element by element
        for (int v = 0; v < 4 && n + v < p.N; ++v) {
            const float y = acc[v] + (p.res ? p.res[(int64_t)m * p.ldo + n + v] : 0.f);
            if (p.out_f32) reinterpret_cast<float*>(p.out)[(int64_t)m * p.ldo + n + v] = y;
            else reinterpret_cast<__bf16*>(p.out)[(int64_t)m * p.ldo + n + v] = (__bf16)y;
        }
        return;
    }
    if (p.res) acc += *reinterpret_cast<const f32x4*>(p.res + (int64_t)m * p.ldo + n);
    if (p.out_f32) {
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.out) + (int64_t)m * p.ldo + n) = acc;
    } else {
        u32x2 o;
        o[0] = pack_bf16x2(acc[0], acc[1]);
        o[1] = pack_bf16x2(acc[2], acc[3]);
        *reinterpret_cast<u32x2*>(reinterpret_cast<__bf16*>(p.out) + (int64_t)m * p.ldo + n) = o;
    }
}

// The new token's q / k / v (one row of the fused projection per sequence): rotary embedding on q and k at the row's
// position (the arithmetic of hk_rope_kernel, hk_llm.hip: apply_rotary_pos_emb, modeling_llama.py:116-123), q scaled by
// qscale into qout, k and v appended to the cache at `slot` (past_key_value concat, :201-206).
// Thread -> 8 pair indices j of one (sequence, head).
__global__ __launch_bounds__(256) void hk_rope_append_kernel(const __bf16* __restrict__ qkv, const float* __restrict__ cos_t,
                                                             const float* __restrict__ sin_t, const int32_t* __restrict__ pos0,
                                                             int pos_add, __bf16* __restrict__ qout, __bf16* __restrict__ kc,
                                                             __bf16* __restrict__ vc, int B, int H, int hd, int slot,
                                                             int64_t cache_rows, float qscale) {
    const int half = hd >> 1, per_head = half >> 3;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= B * H * per_head) return;
    const int j8 = idx % per_head;
    const int head = (idx / per_head) % H;
    const int b = idx / (per_head * H);
    const int D = H * hd;
    const int p = pos0[b] + pos_add;
    const __bf16* row = qkv + (int64_t)b * 3 * D + head * hd + j8 * 8;
    const float* cp = cos_t + (int64_t)p * half + j8 * 8;
    const float* sp = sin_t + (int64_t)p * half + j8 * 8;
    const int64_t crow = ((int64_t)b * cache_rows + slot) * D + head * hd + j8 * 8;
#pragma unroll
    for (int which = 0; which < 2; ++which) {
        const u32x4 a = *reinterpret_cast<const u32x4*>(row + which * D);
        const u32x4 bb = *reinterpret_cast<const u32x4*>(row + which * D + half);
        const float sc = which == 0 ? qscale : 1.0f;
        u32x4 oa, ob;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float x1[2] = {bf16_lo(a[k]), bf16_hi(a[k])}, x2[2] = {bf16_lo(bb[k]), bf16_hi(bb[k])};
            float y1[2], y2[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const float c = cp[2 * k + e], s = sp[2 * k + e];
                y1[e] = (x1[e] * c - x2[e] * s) * sc;
                y2[e] = (x2[e] * c + x1[e] * s) * sc;
            }
            oa[k] = pack_bf16x2(y1[0], y1[1]);
            ob[k] = pack_bf16x2(y2[0], y2[1]);
        }
        __bf16* dst = which == 0 ? qout + (int64_t)b * D + head * hd + j8 * 8 : kc + crow;
        *reinterpret_cast<u32x4*>(dst) = oa;
        *reinterpret_cast<u32x4*>(dst + half) = ob;
    }
    *reinterpret_cast<u32x4*>(vc + crow) = *reinterpret_cast<const u32x4*>(row + 2 * D);
    *reinterpret_cast<u32x4*>(vc + crow + half) = *reinterpret_cast<const u32x4*>(row + 2 * D + half);
}

// One query row per (sequence, head) against the cache: scores -> LDS, fp32 softmax over all keys (upcast as
// modeling_llama.py:218; masked keys contribute exactly 0), then P.V.  q arrives scaled by softmax_scale * log2 e (exp2
// domain).  A key's head row (hd bf16) is read by LPK = hd / 8 lanes, 16 bytes each; the NT / LPK key groups of the
// workgroup take keys g, g + NG, ...  Memory-bound: K and V of the head are streamed once; NT = 1024 threads keep 4x the
// loads of one (sequence, head) in flight (B * H workgroups is all the parallelism a decode step has here).
// FUSED: the workgroup first does hk_rope_append_kernel's work for ITS head -- it is the only reader of that head's new
// row -- from the token's fused q|k|v row: rotary q -> LDS (rounded to bf16 as the two-launch route stores it), rotary k
// and v -> cache row `slot`, visible to the whole workgroup after the barrier; n_keys = slot + 1.
struct DecodeAttnParams {
    const __bf16* q;        // !FUSED: [B][D] rotary, scaled;  FUSED: the fused q|k|v rows [B][3D]
    __bf16* kc;
    __bf16* vc;
    const float* key_bias;
    __bf16* out;
    const float* cos_t;
    const float* sin_t;
    const int32_t* pos0;
    int pos_add, slot;
    float qscale;
    int H, n_keys;
    int64_t cache_rows, bias_stride;
    // key split (long caches, few (sequence, head) pairs): grid.z = nsplit slices of `chunk` keys; slice s leaves its
    // un-normalised output row, its maximum and its sum in part[((b * H + head) * nsplit + s) * (HD + 2) ...] and
    // hk_decode_attn_combine_kernel merges them; nsplit = 1: the workgroup writes the normalised bf16 row itself
    int nsplit, chunk;
    float* part;
};

template <int HD, int NT, bool FUSED>
__global__ __launch_bounds__(NT) void hk_decode_attn_kernel(const DecodeAttnParams p) {
    extern __shared__ float sc[];            // [n_keys] scores, then probabilities
    constexpr int NW = NT / 64;
    __shared__ float red[2 * NW];
    __shared__ float obuf[NW][HD];
    __shared__ float qs[HD];
    constexpr int LPK = HD / 8, NG = NT / LPK;   // lanes per key, key groups per workgroup
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int sub = tid % LPK, g = tid / LPK;
    const int head = blockIdx.x, b = blockIdx.y;
    const int D = p.H * HD;
    const int k_begin = p.nsplit > 1 ? (int)blockIdx.z * p.chunk : 0;
    const int n_keys = p.nsplit > 1 ? min(p.chunk, p.n_keys - k_begin) : p.n_keys;     // this workgroup's keys: [k_begin, k_begin + n_keys)
    const __bf16* kb = p.kc + ((int64_t)b * p.cache_rows + k_begin) * D + head * HD + sub * 8;
    const __bf16* vb = p.vc + ((int64_t)b * p.cache_rows + k_begin) * D + head * HD + sub * 8;
    float qf[8];
    if constexpr (FUSED) {
        constexpr int half = HD / 2;
        const __bf16* row = p.q + (int64_t)b * 3 * D + head * HD;
        const int64_t crow = ((int64_t)b * p.cache_rows + p.slot) * D + head * HD;
        if (tid < half) {
            const int pos = p.pos0[b] + p.pos_add;
            const float c = p.cos_t[(int64_t)pos * half + tid], s = p.sin_t[(int64_t)pos * half + tid];
            const float q1 = (float)row[tid], q2 = (float)row[tid + half];
            const float k1 = (float)row[D + tid], k2 = (float)row[D + tid + half];
            qs[tid] = (float)(__bf16)((q1 * c - q2 * s) * p.qscale);
            qs[tid + half] = (float)(__bf16)((q2 * c + q1 * s) * p.qscale);
            if (p.slot >= k_begin && p.slot < k_begin + n_keys) {   // the slice that holds the new row writes it; no other slice reads it
                p.kc[crow + tid] = (__bf16)(k1 * c - k2 * s);
                p.kc[crow + tid + half] = (__bf16)(k2 * c + k1 * s);
            }
        } else if (tid < half + HD / 8 && p.slot >= k_begin && p.slot < k_begin + n_keys) {
            const int j = (tid - half) * 8;
            *reinterpret_cast<u32x4*>(p.vc + crow + j) = *reinterpret_cast<const u32x4*>(row + 2 * D + j);
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 8; ++k) qf[k] = qs[sub * 8 + k];
    } else {
        const u32x4 u = *reinterpret_cast<const u32x4*>(p.q + (int64_t)b * D + head * HD + sub * 8);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            qf[2 * k] = bf16_lo(u[k]);
            qf[2 * k + 1] = bf16_hi(u[k]);
        }
    }
    const float* bias = p.key_bias ? p.key_bias + (int64_t)b * p.bias_stride + k_begin : nullptr;
    // ---- scores ----
    constexpr int U = 4;
    for (int k0 = g; k0 < n_keys; k0 += U * NG) {
        u32x4 kv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int key = min(k0 + u * NG, n_keys - 1);
            kv[u] = *reinterpret_cast<const u32x4*>(kb + (int64_t)key * D);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) s += qf[2 * k] * bf16_lo(kv[u][k]) + qf[2 * k + 1] * bf16_hi(kv[u][k]);
#pragma unroll
            for (int o = LPK / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
            const int key = k0 + u * NG;
            if (sub == 0 && key < n_keys) sc[key] = s + (bias ? bias[key] * HK_LOG2E : 0.f);
        }
    }
    __syncthreads();
    // ---- softmax over the row ----
    float mx = -3.0e38f;
    for (int k = tid; k < n_keys; k += NT) mx = fmaxf(mx, sc[k]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = red[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) mx = fmaxf(mx, red[w]);
    float sum = 0.f;
    for (int k = tid; k < n_keys; k += NT) {
        const float pr = __builtin_amdgcn_exp2f(sc[k] - mx);
        sc[k] = pr;
        sum += pr;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    if (lane == 0) red[NW + wave] = sum;
    __syncthreads();
    sum = red[NW];
#pragma unroll
    for (int w = 1; w < NW; ++w) sum += red[NW + w];
    // ---- P . V ----
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int k0 = g; k0 < n_keys; k0 += U * NG) {
        u32x4 vv[U];
        float pr[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int key = k0 + u * NG;
            pr[u] = key < n_keys ? sc[key] : 0.f;
            vv[u] = *reinterpret_cast<const u32x4*>(vb + (int64_t)min(key, n_keys - 1) * D);
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                acc[2 * k] += pr[u] * bf16_lo(vv[u][k]);
                acc[2 * k + 1] += pr[u] * bf16_hi(vv[u][k]);
            }
    }
    // key groups of one wave (lanes sub + LPK * t), then the waves through LDS (summed in wave order)
#pragma unroll
    for (int o = LPK; o < 64; o <<= 1)
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] += __shfl_xor(acc[k], o, 64);
    if (lane < LPK) {
#pragma unroll
        for (int k = 0; k < 8; ++k) obuf[wave][lane * 8 + k] = acc[k];
    }
    __syncthreads();
    if (p.nsplit > 1) {
        float* pr = p.part + (((int64_t)b * p.H + head) * p.nsplit + blockIdx.z) * (HD + 2);
        if (tid < HD) {
            float o = obuf[0][tid];
#pragma unroll
            for (int w = 1; w < NW; ++w) o += obuf[w][tid];
            pr[tid] = o;
        } else if (tid == HD) {
            pr[HD] = mx;
            pr[HD + 1] = sum;
        }
        return;
    }
    if (tid < HD / 2) {
        const int d = tid * 2;
        const float inv = 1.0f / sum;
        float o0 = obuf[0][d], o1 = obuf[0][d + 1];
#pragma unroll
        for (int w = 1; w < NW; ++w) {
            o0 += obuf[w][d];
            o1 += obuf[w][d + 1];
        }
        *reinterpret_cast<uint32_t*>(p.out + (int64_t)b * D + head * HD + d) = pack_bf16x2(o0 * inv, o1 * inv);
    }
}

// Merge of the key slices of hk_decode_attn_kernel: out = sum_s o_s 2^(m_s - M) / sum_s l_s 2^(m_s - M), slices in order.
template <int HD>
__global__ __launch_bounds__(HD) void hk_decode_attn_combine_kernel(const float* __restrict__ part, __bf16* __restrict__ out, int H,
                                                                    int nsplit) {
    const int head = blockIdx.x, b = blockIdx.y, d = threadIdx.x;
    const float* pr = part + ((int64_t)b * H + head) * nsplit * (HD + 2);
    float M = pr[HD];
    for (int s = 1; s < nsplit; ++s) M = fmaxf(M, pr[s * (HD + 2) + HD]);
    float o = 0.f, l = 0.f;
    for (int s = 0; s < nsplit; ++s) {
        const float w = __builtin_amdgcn_exp2f(pr[s * (HD + 2) + HD] - M);
        o += pr[s * (HD + 2) + d] * w;
        l += pr[s * (HD + 2) + HD + 1] * w;
    }
    out[((int64_t)b * H + head) * HD + d] = (__bf16)(o / l);
}

// Greedy next token (do_sample=False, num_beams=1: tasks/test.py:525, test_recursive_grounding.py:298): index of the
// largest logit of each row, the lowest index among equals (torch.argmax's choice).  One workgroup of 1024 threads per row,
// 16-byte loads where the row allows.
__global__ __launch_bounds__(1024) void hk_argmax_kernel(const float* __restrict__ logits, int64_t* __restrict__ out, int V,
                                                         int64_t ld) {
    __shared__ float bv[16];
    __shared__ int bi[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* row = logits + (int64_t)blockIdx.x * ld;
    float best = -3.4e38f;
    int idx = 0x7fffffff;
    auto take = [&](float v, int k) {
        if (v > best || (v == best && k < idx)) {
            best = v;
            idx = k;
        }
    };
    const bool vec = ((uintptr_t)row & 15) == 0;
    const int V4 = vec ? (V & ~3) : 0;
    for (int k = tid * 4; k < V4; k += 4096) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(row + k);
        take(v[0], k);
        take(v[1], k + 1);
        take(v[2], k + 2);
        take(v[3], k + 3);
    }
    for (int k = V4 + tid; k < V; k += 1024) take(row[k], k);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(idx, o, 64);
        take(ov, oi);
    }
    if (lane == 0) {
        bv[wave] = best;
        bi[wave] = idx;
    }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 16; ++w) take(bv[w], bi[w]);
        out[blockIdx.x] = idx == 0x7fffffff ? 0 : idx;
    }
}

}  // namespace

namespace {
template <int KS, int MODE, int KEEP>
int launch_gemv(const GemvParams& p, dim3 grid, size_t xlds, hipStream_t s) {
    static std::atomic<unsigned long long> done{0};
    if (xlds > 40 * 1024) {
        const hipError_t e = hk_lds_attr_once(done, reinterpret_cast<const void*>(&hk_gemv_kernel<KS, MODE, KEEP>), 128 * 1024);
        HK_REQUIRE(e == hipSuccess, "hk_gemv_bf16: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
    }
    hipLaunchKernelGGL((hk_gemv_kernel<KS, MODE, KEEP>), grid, dim3(KS * 64), xlds, s, p);
    return HK_OK;
}
}  // namespace

extern "C" int hk_gemv_bf16(const hk_gemv_desc* d, void* stream) {
    HK_REQUIRE(d && d->x && d->W && d->out, "hk_gemv_bf16: null pointer");
    HK_REQUIRE(d->M >= 1 && d->M <= 16, "hk_gemv_bf16: M=%d rows (1..16: the decode step; larger batches go through hk_gemm_bf16)", d->M);
    HK_REQUIRE(d->N > 0 && d->K > 0 && d->K % 64 == 0, "hk_gemv_bf16: N=%d must be positive, K=%d a multiple of 64", d->N, d->K);
    HK_REQUIRE(d->act == HK_ACT_NONE || d->act == HK_ACT_SWIGLU, "hk_gemv_bf16: act %d not supported (none, swiglu)", d->act);
    HK_REQUIRE(d->act != HK_ACT_SWIGLU || d->N % 4 == 0, "hk_gemv_bf16: swiglu needs N=%d to be a multiple of 4", d->N);
    HK_REQUIRE(d->act != HK_ACT_SWIGLU || (d->out_f32 == 0 && d->residual == nullptr && d->ldo >= d->N / 2 && d->ldo % 2 == 0),
               "hk_gemv_bf16: swiglu writes bf16 [M][N/2], no residual");
    HK_REQUIRE(d->act == HK_ACT_SWIGLU || (d->ldo >= d->N && d->ldo % 4 == 0), "hk_gemv_bf16: ldo=%ld must be >= N and a multiple of 4",
               (long)d->ldo);
    HK_REQUIRE(d->ldx >= d->K && d->ldx % 8 == 0, "hk_gemv_bf16: ldx=%ld must be >= K and a multiple of 8", (long)d->ldx);
    HK_REQUIRE(d->residual == nullptr || d->out_f32 != 0, "hk_gemv_bf16: a residual needs the fp32 output");
    HK_REQUIRE(((uintptr_t)d->x | (uintptr_t)d->W | (uintptr_t)d->out | (uintptr_t)d->residual | (uintptr_t)d->norm_weight) % 16 == 0,
               "hk_gemv_bf16: operands must be 16-byte aligned");
    HK_REQUIRE(d->w_tiled == 0 || d->w_tiled == 1, "hk_gemv_bf16: w_tiled must be 0 or 1");
    HK_REQUIRE(d->norm_weight == nullptr || (d->M <= 4 && d->norm_eps > 0.f && d->x != d->out && d->K <= 4096),
               "hk_gemv_bf16: the fused RMSNorm takes M <= 4 fp32 rows of K <= 4096, eps > 0, and cannot run in place");
    GemvParams p;
    p.x = d->x;
    p.W = reinterpret_cast<const __bf16*>(d->W);
    p.out = d->out;
    p.res = d->residual;
    p.norm_w = d->norm_weight;
    p.M = d->M;
    p.N = d->N;
    p.K = d->K;
    p.ldx = d->ldx;
    p.ldo = d->ldo;
    p.out_f32 = d->out_f32;
    p.act = d->act;
    p.eps = d->norm_eps;
    p.tiled = d->w_tiled;
    // K slices (waves per workgroup): about 12 waves per CU, at least 4 K-blocks per slice, at most 8 slices (measured with cold
    // weights, tools/gemv_bench.py, us at 1 / 2 / 4 / 8 / 16 slices: qkv 24.4 23.6 21.1 22.2 26.1, o_proj 15.8 11.0 9.3 9.2 9.7,
    // gate|up 39.9 39.8 38.1 39.0 47.4, down_proj 39.0 26.1 21.2 20.6 21.1, lm_head 52.2 52.5 50.4 51.9 62.5)
    const int nb = (d->N + 15) / 16, nblk = d->K / 64;
    int ks = 1;
    while (ks < 8 && (long)nb * ks < 3000 && nblk / (ks * 2) >= 4) ks *= 2;
    static const int force = getenv("HK_GEMV_KS") ? atoi(getenv("HK_GEMV_KS")) : 0;   // A/B aid
    if (force == 1 || force == 2 || force == 4 || force == 8 || force == 16) ks = force;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const dim3 grid((unsigned)nb);
    // operand rows staged in LDS when they fit beside a few workgroups per CU (always for the fused norm: K <= 4096, M <= 4);
    // a launch of at most one workgroup per CU (o_proj / down_proj: 256 row blocks) may take most of the CU's LDS for them
    const size_t xbytes = (size_t)p.M * (p.K + 64) * 2;
    static const int xl_env = getenv("HK_GEMV_XLDS") ? atoi(getenv("HK_GEMV_XLDS")) : 1;   // A/B aid: 0 = fragments from global memory
    const size_t xmax = nb <= 256 ? 128 * 1024 : 40 * 1024;      // + up to 16 KB of static LDS
    const int mode = p.norm_w ? 2 : (xl_env != 0 && xbytes <= xmax) ? 1 : 0;
    const size_t xlds = mode ? xbytes : 0;
    static const int keep_env = getenv("HK_GEMV_KEEP") ? atoi(getenv("HK_GEMV_KEEP")) : 1;   // A/B aid: 0 = the second pass re-reads x
    const int keep = (mode == 2 && keep_env) ? (p.M == 1 ? 1 : (ks >= 4 ? 4 : 0)) : 0;
    int rc = HK_OK;
#define HK_GEMV(KS_)                                                                         \
    case KS_:                                                                                \
        if (mode == 2 && keep == 1) rc = launch_gemv<KS_, 2, 1>(p, grid, xlds, s);          \
        else if (mode == 2 && keep == 4) rc = launch_gemv<KS_, 2, (KS_ >= 4 ? 4 : 0)>(p, grid, xlds, s); \
        else if (mode == 2) rc = launch_gemv<KS_, 2, 0>(p, grid, xlds, s);                  \
        else if (mode == 1) rc = launch_gemv<KS_, 1, 0>(p, grid, xlds, s);                  \
        else rc = launch_gemv<KS_, 0, 0>(p, grid, 0, s);                                    \
        break;
    switch (ks) {
        HK_GEMV(1) HK_GEMV(2) HK_GEMV(4) HK_GEMV(8)
        default:
            if (mode == 2) rc = launch_gemv<16, 2, 0>(p, grid, xlds, s);
            else if (mode == 1) rc = launch_gemv<16, 1, 0>(p, grid, xlds, s);
            else rc = launch_gemv<16, 0, 0>(p, grid, 0, s);
    }
#undef HK_GEMV
    if (rc != HK_OK) return rc;
    HK_CHECK_LAUNCH("hk_gemv_bf16");
    return HK_OK;
}

extern "C" int hk_decode_rope_append(const void* qkv, const float* cos_table, const float* sin_table, const int32_t* pos0,
                                     int32_t pos_add, void* q_out, void* k_cache, void* v_cache, int32_t B, int32_t H,
                                     int32_t head_dim, int32_t slot, int64_t cache_rows, int32_t table_rows, float qscale,
                                     void* stream) {
    HK_REQUIRE(qkv && cos_table && sin_table && pos0 && q_out && k_cache && v_cache, "hk_decode_rope_append: null pointer");
    HK_REQUIRE(B > 0 && H > 0 && (head_dim == 64 || head_dim == 128), "hk_decode_rope_append: bad shape B=%d H=%d head_dim=%d", B, H,
               head_dim);
    HK_REQUIRE(slot >= 0 && slot < cache_rows, "hk_decode_rope_append: slot %d outside the cache (%ld rows)", slot, (long)cache_rows);
    HK_REQUIRE(pos_add >= 0 && pos_add < table_rows, "hk_decode_rope_append: position offset %d outside the rotary table (%d rows)",
               pos_add, table_rows);
    HK_REQUIRE(((uintptr_t)qkv | (uintptr_t)q_out | (uintptr_t)k_cache | (uintptr_t)v_cache | (uintptr_t)cos_table |
                (uintptr_t)sin_table) % 16 == 0, "hk_decode_rope_append: operands must be 16-byte aligned");
    const int n = B * H * (head_dim / 16);
    hipLaunchKernelGGL(hk_rope_append_kernel, dim3((n + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       reinterpret_cast<const __bf16*>(qkv), cos_table, sin_table, pos0, pos_add, reinterpret_cast<__bf16*>(q_out),
                       reinterpret_cast<__bf16*>(k_cache), reinterpret_cast<__bf16*>(v_cache), B, H, head_dim, slot, cache_rows,
                       qscale);
    HK_CHECK_LAUNCH("hk_decode_rope_append");
    return HK_OK;
}

namespace {
template <int HD, int NT, bool FUSED>
int launch_decode_attn(const DecodeAttnParams& p, int B, hipStream_t s) {
    static std::atomic<unsigned long long> done{0};
    const size_t lds = (size_t)(p.nsplit > 1 ? p.chunk : p.n_keys) * sizeof(float);
    if (lds > 32 * 1024) {
        const hipError_t e = hk_lds_attr_once(done, reinterpret_cast<const void*>(&hk_decode_attn_kernel<HD, NT, FUSED>), 60 * 1024);
        HK_REQUIRE(e == hipSuccess, "hk_decode_attn: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
    }
    hipLaunchKernelGGL((hk_decode_attn_kernel<HD, NT, FUSED>), dim3((unsigned)p.H, (unsigned)B, (unsigned)p.nsplit), dim3(NT), lds, s, p);
    if (p.nsplit > 1)
        hipLaunchKernelGGL((hk_decode_attn_combine_kernel<HD>), dim3((unsigned)p.H, (unsigned)B), dim3(HD), 0, s, p.part, p.out, p.H, p.nsplit);
    return HK_OK;
}
// slices of whole 64-key tiles, none empty; 1 = no split
void set_key_split(DecodeAttnParams& p, int key_splits, float* ws) {
    p.nsplit = 1;
    p.chunk = 0;
    p.part = ws;
    if (key_splits > 1 && ws != nullptr && p.n_keys >= 128) {
        const int want = key_splits < 16 ? key_splits : 16;
        const int chunk = ((p.n_keys + want - 1) / want + 63) / 64 * 64;
        p.chunk = chunk;
        p.nsplit = (p.n_keys + chunk - 1) / chunk;
    }
}
template <bool FUSED>
int dispatch_decode_attn(const DecodeAttnParams& p, int B, int head_dim, hipStream_t s) {
    // 1024 threads per (sequence, head) unless the call already has a few workgroups per CU (HK_DECODE_ATTN_NT: A/B aid)
    static const int force = getenv("HK_DECODE_ATTN_NT") ? atoi(getenv("HK_DECODE_ATTN_NT")) : 0;
    const bool big = force == 1024 || (force != 256 && (long)B * p.H < 1024 && p.n_keys >= 128);
    if (head_dim == 128) return big ? launch_decode_attn<128, 1024, FUSED>(p, B, s) : launch_decode_attn<128, 256, FUSED>(p, B, s);
    return big ? launch_decode_attn<64, 1024, FUSED>(p, B, s) : launch_decode_attn<64, 256, FUSED>(p, B, s);
}
}  // namespace

extern "C" int hk_decode_attn_bf16(const void* q, const void* k_cache, const void* v_cache, const float* key_bias, void* out,
                                   int32_t B, int32_t H, int32_t head_dim, int32_t n_keys, int64_t cache_rows,
                                   int64_t bias_stride, int32_t key_splits, float* split_ws, void* stream) {
    HK_REQUIRE(q && k_cache && v_cache && out, "hk_decode_attn_bf16: null pointer");
    HK_REQUIRE(B > 0 && B < 65536 && H > 0 && (head_dim == 64 || head_dim == 128), "hk_decode_attn_bf16: bad shape B=%d H=%d head_dim=%d",
               B, H, head_dim);
    HK_REQUIRE(n_keys >= 1 && n_keys <= cache_rows && n_keys <= 12288,
               "hk_decode_attn_bf16: n_keys=%d must be 1..min(cache rows, 12288: the score row lives in LDS)", n_keys);
    HK_REQUIRE(key_bias == nullptr || bias_stride >= n_keys, "hk_decode_attn_bf16: bias_stride < n_keys");
    HK_REQUIRE(((uintptr_t)q | (uintptr_t)k_cache | (uintptr_t)v_cache | (uintptr_t)out) % 16 == 0,
               "hk_decode_attn_bf16: operands must be 16-byte aligned");
    DecodeAttnParams p = {};
    p.q = reinterpret_cast<const __bf16*>(q);
    p.kc = reinterpret_cast<__bf16*>(const_cast<void*>(k_cache));
    p.vc = reinterpret_cast<__bf16*>(const_cast<void*>(v_cache));
    p.key_bias = key_bias;
    p.out = reinterpret_cast<__bf16*>(out);
    p.H = H;
    p.n_keys = n_keys;
    p.cache_rows = cache_rows;
    p.bias_stride = bias_stride;
    set_key_split(p, key_splits, split_ws);
    const int rc = dispatch_decode_attn<false>(p, B, head_dim, reinterpret_cast<hipStream_t>(stream));
    if (rc != HK_OK) return rc;
    HK_CHECK_LAUNCH("hk_decode_attn_bf16");
    return HK_OK;
}

extern "C" int hk_decode_attn_rope_bf16(const void* qkv, const float* cos_table, const float* sin_table, const int32_t* pos0,
                                        int32_t pos_add, void* k_cache, void* v_cache, const float* key_bias, void* out, int32_t B,
                                        int32_t H, int32_t head_dim, int32_t slot, int64_t cache_rows, int64_t bias_stride,
                                        int32_t table_rows, float q_scale, int32_t key_splits, float* split_ws, void* stream) {
    HK_REQUIRE(qkv && cos_table && sin_table && pos0 && k_cache && v_cache && out, "hk_decode_attn_rope_bf16: null pointer");
    HK_REQUIRE(B > 0 && B < 65536 && H > 0 && (head_dim == 64 || head_dim == 128),
               "hk_decode_attn_rope_bf16: bad shape B=%d H=%d head_dim=%d", B, H, head_dim);
    HK_REQUIRE(slot >= 0 && slot < cache_rows && slot < 12288,
               "hk_decode_attn_rope_bf16: slot %d outside the cache (%ld rows; at most 12288 keys: the score row lives in LDS)", slot,
               (long)cache_rows);
    HK_REQUIRE(pos_add >= 0 && pos_add < table_rows, "hk_decode_attn_rope_bf16: position offset %d outside the rotary table (%d rows)",
               pos_add, table_rows);
    HK_REQUIRE(key_bias == nullptr || bias_stride > slot, "hk_decode_attn_rope_bf16: bias_stride <= slot");
    HK_REQUIRE(((uintptr_t)qkv | (uintptr_t)k_cache | (uintptr_t)v_cache | (uintptr_t)out) % 16 == 0,
               "hk_decode_attn_rope_bf16: operands must be 16-byte aligned");
    DecodeAttnParams p = {};
    p.q = reinterpret_cast<const __bf16*>(qkv);
    p.kc = reinterpret_cast<__bf16*>(k_cache);
    p.vc = reinterpret_cast<__bf16*>(v_cache);
    p.key_bias = key_bias;
    p.out = reinterpret_cast<__bf16*>(out);
    p.cos_t = cos_table;
    p.sin_t = sin_table;
    p.pos0 = pos0;
    p.pos_add = pos_add;
    p.slot = slot;
    p.qscale = q_scale;
    p.H = H;
    p.n_keys = slot + 1;
    p.cache_rows = cache_rows;
    p.bias_stride = bias_stride;
    set_key_split(p, key_splits, split_ws);
    const int rc = dispatch_decode_attn<true>(p, B, head_dim, reinterpret_cast<hipStream_t>(stream));
    if (rc != HK_OK) return rc;
    HK_CHECK_LAUNCH("hk_decode_attn_rope_bf16");
    return HK_OK;
}

extern "C" int hk_argmax_f32(const float* logits, int64_t* out, int32_t rows, int32_t V, int64_t ld, void* stream) {
    HK_REQUIRE(logits && out && rows > 0 && V > 0 && ld >= V, "hk_argmax_f32: bad arguments rows=%d V=%d", rows, V);
    hipLaunchKernelGGL(hk_argmax_kernel, dim3((unsigned)rows), dim3(1024), 0, reinterpret_cast<hipStream_t>(stream), logits, out, V, ld);
    HK_CHECK_LAUNCH("hk_argmax_f32");
    return HK_OK;
}
