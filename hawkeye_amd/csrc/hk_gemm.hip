// bf16 MFMA GEMM for gfx950:  C = act((A . W^T + bias) * colscale) + residual
//
// A [M,K] and W [N,K] are both K-contiguous (activations row-major, nn.Linear weights as stored),
// so both operands stream into LDS with the same 16-byte LDS-DMA pattern (global_load_lds_dwordx4)
// and are read back as 16-byte MFMA fragments (ds_read_b128).
//
// Tile: BM x BN x 64 per workgroup, WM x WN waves, v_mfma_f32_16x16x32_bf16.  The MFMA is issued
// "transposed" (W fragment as the A operand, activation fragment as the B operand), so the 16x16
// result tile has the output ROW on the lane and four consecutive output COLUMNS in the lane's
// four accumulator registers: the epilogue then loads bias/residual and stores C as 8/16-byte
// vectors per lane with no cross-lane traffic.
//
// LDS image (per stage): [BM rows][64 k] then [BN rows][64 k], 128-byte rows.  The LDS-DMA
// destination is lane-linear, so the bank swizzle is applied to the per-lane SOURCE address and
// undone on the read (guide rule 21): physical 16-B chunk pc of row r holds logical chunk
// pc ^ ((r >> 1) & 7), which makes every ds_read_b128 16-lane group hit 16 distinct 16-B slots.
//
// Pipeline: two LDS stages; the loads of K-tile t+1 are issued before the MFMAs of tile t and
// retired by one vmcnt(0)+barrier per K-tile.
#include <stdlib.h>
#include <type_traits>
#include "hk_common.h"

#ifndef HK_DUO_PF
#define HK_DUO_PF 1   // residual prefetch distance of the duo kernel's in-place epilogue (rounds)
#endif
#ifndef HK_GEMM_DUO_DEFAULT
#define HK_GEMM_DUO_DEFAULT 0
#endif
// Diagnostic switches (timing-only kernels, phase stamps) exist only in the -DHK_DIAG build (make diag)
#ifdef HK_DIAG
#define HK_DBG(p) ((p).dbg)
#define HK_TRACE(p) ((p).trace)
#else
#define HK_DBG(p) 0
#define HK_TRACE(p) ((unsigned long long*)nullptr)
#endif

namespace {

struct RowMap {
    int64_t ld, rpb, bs;
    __device__ __forceinline__ int64_t off(int m) const {
        if (rpb == 0) return (int64_t)m * ld;
        const int b = m / (int)rpb;
        return (int64_t)b * bs + (int64_t)(m - b * (int)rpb) * ld;
    }
};

struct GemmParams {
    const __bf16* A;
    const __bf16* W;
    const float* bias;
    const void* res;
    void* C;
    const float* row_stats;  // folded LayerNorm of A: [ceil256(M)][2] (mean, rstd) or null
    const float* col_c;      // with row_stats: [N] column sums of W
    float* row_part;         // out: per row and 64-column group (sum, sum of squares) of C as stored (bf16), or null
    int M, N, K;
    RowMap a, c, r;
    int64_t ldw;
    int act, out_f32, res_bf16, scale_cols, dbg, gm;
    unsigned long long* trace;  // debug: per workgroup [hwid, t_start, then (t_loop_end, t_epi_end) per tile]
    float scale;
    int nbm, nbn;
    int m_off;  // multi-stage kernel only: logical row of this launch's row 0 (tail launch of a split problem)
    // multi-stage kernel only: split-K.  ksplit slices of kper K-tiles each; grid = tiles x ksplit; slice z writes its
    // fp32 partial product to C + z * split_stride, slice 0 carries bias and residual
    int ksplit, kper;
    int64_t split_stride;
    // hk_gemm_desc.w_tiled: W in 64 x 64 blocks, [N/64][K/64][64][64].  wrow(n) = element offset of row n's first K-tile, wk =
    // elements from one K-tile of a row to the next (row-major: n * ldw and 64)
    int wt, wnkt;
    __device__ __forceinline__ int64_t wrow(int n) const { return wt ? ((int64_t)(n >> 6) * wnkt * 4096 + (n & 63) * 64) : (int64_t)n * ldw; }
    __device__ __forceinline__ int wk() const { return wt ? 4096 : 64; }
};

// Exact-erf GELU (nn.GELU default, vit.py:44; ACT2FN["gelu"], Qformer.py:383) to fp32 rounding noise:
// erfc(t) = exp2(t * p(t)) on t in [0, 6], |err| <= 3.2e-7 (degree-5 fit), and
//     gelu(x) = 0.5 x + 0.5 |x| (1 - erfc(|x| / sqrt 2))
// which needs no sign select.  Written on float2 so hipcc emits v_pk_fma_f32 / v_pk_mul_f32: 8 VALU
// instructions per element instead of 14.
typedef __attribute__((ext_vector_type(2))) float f32x2;
__device__ __forceinline__ f32x2 gelu_erf2(const f32x2 x) {
    const f32x2 a = __builtin_elementwise_max(x, -x);
    f32x2 t = a * 0.70710678118654752f;
    t = __builtin_elementwise_min(t, f32x2{6.0f, 6.0f});
    f32x2 p = f32x2{0.00014203831960912794f, 0.00014203831960912794f};
    p = __builtin_elementwise_fma(p, t, f32x2{-0.0036642642226070166f, -0.0036642642226070166f});
    p = __builtin_elementwise_fma(p, t, f32x2{0.030896173790097237f, 0.030896173790097237f});
    p = __builtin_elementwise_fma(p, t, f32x2{-0.14969943463802338f, -0.14969943463802338f});
    p = __builtin_elementwise_fma(p, t, f32x2{-0.9181654453277588f, -0.9181654453277588f});
    p = __builtin_elementwise_fma(p, t, f32x2{-1.6279250383377075f, -1.6279250383377075f});
    const f32x2 q = p * t;
    f32x2 e;
    e[0] = __builtin_amdgcn_exp2f(q[0]);
    e[1] = __builtin_amdgcn_exp2f(q[1]);
    const f32x2 h = a * 0.5f;
    return __builtin_elementwise_fma(x, f32x2{0.5f, 0.5f}, __builtin_elementwise_fma(-h, e, h));
}
__device__ __forceinline__ f32x4 gelu_erf4(const f32x4 v) {
    const f32x2 lo = gelu_erf2(f32x2{v[0], v[1]}), hi = gelu_erf2(f32x2{v[2], v[3]});
    return f32x4{lo[0], lo[1], hi[0], hi[1]};
}

// The same function for a bf16 destination: a degree-3 fit (minimax on the GELU error, t in [0, 6]) keeps
// |gelu error| <= 8.6e-6 -- under half a bf16 ulp wherever |gelu| >= 4.4e-3 and absolutely negligible below -- and
//     gelu(x) = max(x, 0) - |x| * w,   w = erfc(|x| / sqrt 2) / 2 = exp2(t p(t) - 1)
// costs 8 VALU instructions per element (|x| is an operand modifier), 5 fewer than the fp32-exact form.
// (Measured, not kept: the same arithmetic on element pairs with v_pk_fma_f32, 11 instructions per pair instead of 16, same bits --
// fc1+GELU 776 / 789 -> 794 / 783 us on one box: the epilogue is not bound by the activation's instruction count.)
#ifndef HK_GELU_VARIANT
#define HK_GELU_VARIANT 1
#endif
__device__ __forceinline__ float gelu_erf_bf16dst(const float x) {
    // polynomial in a = min(|x|, 6 sqrt 2) directly (the 1/sqrt 2 lives in the coefficients); past the clamp
    // w <= 1e-17, so using the clamped a in the product changes nothing representable
    const float a = __builtin_fminf(__builtin_fabsf(x), 8.485281374238570f);
    float p = 0.00416090851649642f;
    p = __builtin_fmaf(p, a, -0.04573248326778412f);
    p = __builtin_fmaf(p, a, -0.46493393182754517f);
    p = __builtin_fmaf(p, a, -1.1495658159255981f);
    const float w = __builtin_amdgcn_exp2f(__builtin_fmaf(p, a, -1.0f));
    return __builtin_fmaf(-a, w, __builtin_fmaxf(x, 0.0f));
}
__device__ __forceinline__ f32x4 gelu_erf4_bf16dst(const f32x4 v) {
#if HK_GELU_VARIANT == 0
    return gelu_erf4(v);
#else
    return f32x4{gelu_erf_bf16dst(v[0]), gelu_erf_bf16dst(v[1]), gelu_erf_bf16dst(v[2]), gelu_erf_bf16dst(v[3])};
#endif
}

// Sum over the 8 lanes that share (lane >> 3): two quad permutes and a half-row mirror on the DPP path (every lane
// ends up with the total; EXEC must be all ones).
__device__ __forceinline__ float sum8_dpp(float x) {
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, true));
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4E, 0xF, 0xF, true));
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x141, 0xF, 0xF, true));
    return x;
}

// 4 consecutive residual elements (fp32 or bf16) at element offset `off`
__device__ __forceinline__ f32x4 load_res4(const GemmParams& p, int64_t off) {
    if (p.res_bf16) {
        const u32x2 w = *reinterpret_cast<const u32x2*>(reinterpret_cast<const __bf16*>(p.res) + off);
        return f32x4{bf16_lo(w[0]), bf16_hi(w[0]), bf16_lo(w[1]), bf16_hi(w[1])};
    }
    return *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(p.res) + off);
}

// SwiGLU on interleaved columns (HK_ACT_SWIGLU): a lane's 4 consecutive columns are (gate, up, gate, up) of two output
// columns; silu(g) * u with silu(g) = g / (1 + 2^(-g log2 e)) (g -> -inf: 2^(..) = inf, 1 / inf = 0, g * 0 = -0).
__device__ __forceinline__ float silu_mul(const float g, const float u) {
    const float e = __builtin_amdgcn_exp2f(g * -1.4426950408889634f);
    return g * __builtin_amdgcn_rcpf(1.0f + e) * u;
}

// One lane's 4 consecutive output columns of one row: bias, column scale, activation, residual, store.
__device__ __forceinline__ f32x4 epi_store(const GemmParams& p, f32x4 v, const f32x4 bias, const f32x4 cc, const f32x4 res,
                                           int n, int64_t coff, float rs, float rm) {
    if (p.row_stats) {  // LayerNorm folded: rstd * acc + (bias' - rstd * mean * colsum)
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = __builtin_fmaf(v[e], rs, __builtin_fmaf(-rm, cc[e], bias[e]));
    } else {
        v += bias;
    }
    if (p.scale_cols != 0 && n < p.scale_cols) v *= p.scale;
    if (p.act == HK_ACT_SWIGLU) {   // bf16 out, no residual (checked on the host): two output columns at n / 2
        *reinterpret_cast<uint32_t*>(reinterpret_cast<__bf16*>(p.C) + coff + (n >> 1)) = pack_bf16x2(silu_mul(v[0], v[1]), silu_mul(v[2], v[3]));
        return v;
    }
    if (p.act == HK_ACT_GELU_ERF) v = p.out_f32 ? gelu_erf4(v) : gelu_erf4_bf16dst(v);
    if (p.res) v += res;
    if (p.out_f32) {
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.C) + coff + n) = v;
    } else {
        u32x2 o;
        o[0] = pack_bf16x2(v[0], v[1]);
        o[1] = pack_bf16x2(v[2], v[3]);
        *reinterpret_cast<u32x2*>(reinterpret_cast<__bf16*>(p.C) + coff + n) = o;
        // the values as stored: row statistics taken from them (row_part) describe exactly the matrix the consumer reads
        v = f32x4{bf16_lo(o[0]), bf16_hi(o[0]), bf16_lo(o[1]), bf16_hi(o[1])};
    }
    return v;
}

// Tile raster: groups of GM m-panels, inside a group m fastest, so any run of 32 consecutive tile ids
// (what the 32 CUs of one XCD work on at a time) is a compact GM x (32 / GM) patch: each A panel and each
// W panel it touches is shared by several concurrent tiles in that XCD's L2.
__device__ __forceinline__ void tile_coords(int lid, int nbm, int nbn, int gm_, int& tm, int& tn) {
    const int per = gm_ * nbn;
    const int g = lid / per, r = lid - g * per;
    const int gm = min(gm_, nbm - g * gm_);
    tn = r / gm;
    tm = g * gm_ + (r - tn * gm);
}

__device__ __forceinline__ int xcd_remap(int orig, int nwg) {
    const int xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
}

// ---- epilogue of one 128x64 wave sub-tile (no workgroup barrier inside; LDS regions private to the wave) ----
// Shared by the persistent 256x256 kernel (8 waves) and the two-workgroups-per-CU 256x128 kernel (4 waves): both give a
// wave 128 rows x 64 columns in the same accumulator layout.  `stg`: the wave's 4 KB transpose slab; `aux`: where the
// main loop staged the epilogue operands (row statistics at +0, bias at +1024, column sums of W at +1280; the 256x256
// kernel keeps them in the slab itself, they are read into registers before the first transpose overwrites them).
// The accumulator layout (lane = 1 row x 4 columns of a 16x16 tile) would store 32-byte pieces of
// 16 rows per instruction; instead 16 (fp32) / 32 (bf16) rows at a time go through LDS, XOR-swizzled,
// and leave as whole 256 / 128-byte row segments, 16 bytes per lane.  The fp32 residual is read the
// same way.  Bias, column scale and GELU are applied on the accumulator layout, before LDS.
#ifndef HK_EPI_PF
#define HK_EPI_PF 2   // residual prefetch distance of the in-place epilogue, in rounds (2-6 measured alike; 3 spills beside the FAST path)
#endif
// FAST (round 3): the wave's 128 rows lie inside M and the C / residual rows are plain (no rows_per_batch map) -- every ViT
// launch except a last partial row panel.  Then a row address is a wave-uniform 64-bit base (tile, wave, round: SALU) plus
// ONE 32-bit lane offset, there is no bounds predicate, and the staged rows of a round are read back together.  The general
// form computed RowMap::off() -- two quarter-rate 32-bit multiplies and a 64-bit multiply-add -- per lane and store, kept
// each store in an EXEC-masked block of its own and so waited for each LDS read right where it was issued: 16 exposed LDS
// round trips and ~200 address instructions per wave and tile in an epilogue that is bound by vector issue.
// NI: 16-row blocks per M-half of the wave's sub-tile (4: 128 rows, the 256-row tile; 3: 96 rows, the 192-row tile).
template <int EPI, int PF_ROUNDS, bool FAST, int NI = 4>
__device__ __forceinline__ void pp_epilogue_impl(const GemmParams& p, f32x4 (&acc)[2][2][NI][2], char* stg, const char* aux,
                                                 const int lane, const int wm, const int wn, const int m0, const int n0) {
    if (HK_DBG(p) & 1) {  // dbg 1: timing-only build, no epilogue (accumulators kept live)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int i = 0; i < NI; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) asm volatile("" ::"v"(acc[a][b][i][j]));
        return;
    }
    // bias of the wave's 64 columns: staged into the slab (bytes 1024..1279) by one 4-byte LDS-DMA at the start of
    // the tile (main loop below) -- a global load here would be the youngest vector-memory operation of the wave, and
    // waiting for it would also wait for every operand load of the next tile already in flight.
    f32x4 bias[2][2];
#pragma unroll
    for (int nh = 0; nh < 2; ++nh)
#pragma unroll
        for (int j = 0; j < 2; ++j)
            bias[nh][j] = p.bias != nullptr
                              ? *reinterpret_cast<const f32x4*>(aux + 1024 + (nh * 32 + j * 16 + (lane >> 4) * 4) * 4)
                              : f32x4{0.f, 0.f, 0.f, 0.f};
    constexpr bool FOLD = EPI == 3;
    float rs[2][NI], rm[2][NI];
    f32x4 cc[2][2];
    if (FOLD) {
        typedef __attribute__((ext_vector_type(2))) float f32x2s;
#pragma unroll
        for (int mh = 0; mh < 2; ++mh)
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                // (mean, rstd) of row (mh*NI + i)*16 + (lane & 15) of the wave's 32 NI: staged by the main loop
                const f32x2s st = *reinterpret_cast<const f32x2s*>(aux + ((mh * NI + i) * 16 + (lane & 15)) * 8);
                rs[mh][i] = st[1];
                rm[mh][i] = st[1] * st[0];
            }
#pragma unroll
        for (int nh = 0; nh < 2; ++nh)
#pragma unroll
            for (int j = 0; j < 2; ++j)   // column sums of W: staged like the bias, slab bytes 1280..1535
                cc[nh][j] = *reinterpret_cast<const f32x4*>(aux + 1280 + (nh * 32 + j * 16 + (lane >> 4) * 4) * 4);
    }
    auto pre = [&](f32x4 v, int nh, int j, int mh, int i) {
        const int n = n0 + wn * 64 + nh * 32 + j * 16 + (lane >> 4) * 4;
        if (FOLD) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                v[e] = __builtin_fmaf(v[e], rs[mh][i], __builtin_fmaf(-rm[mh][i], cc[nh][j][e], bias[nh][j][e]));
        } else {
            v += bias[nh][j];
        }
        if (p.scale_cols != 0 && n < p.scale_cols) v *= p.scale;  // uniform test first: most GEMMs scale nothing
        if (p.act == HK_ACT_GELU_ERF) v = (EPI == 0 || EPI == 3 || !p.out_f32) ? gelu_erf4_bf16dst(v) : gelu_erf4(v);
        return v;
    };
    const int mw = m0 + wm * (32 * NI);  // first row / column of this wave's sub-tile
    const int nw = n0 + wn * 64;
    const int r16 = lane & 15, g4 = lane >> 4;
    if (EPI == 5) {
        // SwiGLU on interleaved (gate, up) columns: the wave's 64 columns become 32 output columns, 64 B per row.  32 rows per
        // round through the slab: a lane's two results are 4 bytes of 16-byte chunk nh*2 + j of its row (the four lanes
        // g4 = 0..3 fill the chunk); chunk c of row r lives at c ^ ((r >> 2) & 3), which spreads the 16 rows x 16 B of one
        // ds_write_b32 over all banks; read back as whole 64-byte row segments, 16 rows per store instruction.
        const int on = nw >> 1;   // first output column of the wave
#pragma unroll
        for (int r2 = 0; r2 < NI; ++r2) {
#pragma unroll
            for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                for (int nh = 0; nh < 2; ++nh)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int mh = (2 * r2 + ii) / NI, ib = (2 * r2 + ii) % NI;
                        const f32x4 v = acc[mh][nh][ib][j] + bias[nh][j];
                        const int row = ii * 16 + r16;
                        const int ch = nh * 2 + j;
                        *reinterpret_cast<uint32_t*>(stg + row * 64 + ((ch ^ ((row >> 2) & 3)) << 4) + g4 * 4) =
                            pack_bf16x2(silu_mul(v[0], v[1]), silu_mul(v[2], v[3]));
                    }
            u32x4 o[2];
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int row = it * 16 + (lane >> 2), c = lane & 3;
                o[it] = *reinterpret_cast<const u32x4*>(stg + row * 64 + ((c ^ ((row >> 2) & 3)) << 4));
            }
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int m = mw + r2 * 32 + it * 16 + (lane >> 2), n = nw + (lane & 3) * 16;   // n: first W row of the lane's 8 outputs
                if (m < p.M && n < p.N)
                    __builtin_nontemporal_store(o[it], reinterpret_cast<u32x4*>(reinterpret_cast<__bf16*>(p.C) + p.c.off(m) + on + (lane & 3) * 8));
            }
        }
        return;
    }
    if (EPI == 0 || EPI == 3) {
        // bf16 out, no residual: 32 rows x 128 B per round; 16-B chunk c of row r lives at chunk c ^ (r & 7)
#pragma unroll
            for (int r2 = 0; r2 < NI; ++r2) {   // round r2: 16-row blocks 2 r2, 2 r2 + 1 of the wave's 2 NI
#pragma unroll
                for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                    for (int nh = 0; nh < 2; ++nh)
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            const int mh = (2 * r2 + ii) / NI, ib = (2 * r2 + ii) % NI;
                            const f32x4 v = pre(acc[mh][nh][ib][j], nh, j, mh, ib);
                            u32x2 o;
                            o[0] = pack_bf16x2(v[0], v[1]);
                            o[1] = pack_bf16x2(v[2], v[3]);
                            const int row = ii * 16 + r16;
                            const int ch = nh * 4 + j * 2 + (g4 >> 1);
                            *reinterpret_cast<u32x2*>(stg + row * 128 + ((ch ^ (row & 7)) << 4) + (g4 & 1) * 8) = o;
                        }
                if (FAST) {
                    u32x4 o[4];
#pragma unroll
                    for (int it = 0; it < 4; ++it) {
                        const int row = it * 8 + (lane >> 3), ch = lane & 7;
                        o[it] = *reinterpret_cast<const u32x4*>(stg + row * 128 + ((ch ^ (row & 7)) << 4));
                    }
                    char* rb = reinterpret_cast<char*>(p.C) + ((int64_t)(mw + r2 * 32) * p.c.ld + nw) * 2;   // wave-uniform
                    const uint32_t lo = (uint32_t)(((lane >> 3) * (int)p.c.ld + (lane & 7) * 8) * 2);
#pragma unroll
                    for (int it = 0; it < 4; ++it)
                        __builtin_nontemporal_store(o[it], reinterpret_cast<u32x4*>(rb + (int64_t)it * 8 * p.c.ld * 2 + lo));
                } else {
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int row = it * 8 + (lane >> 3), ch = lane & 7;
                    const u32x4 o = *reinterpret_cast<const u32x4*>(stg + row * 128 + ((ch ^ (row & 7)) << 4));
                    const int m = mw + r2 * 32 + row, n = nw + ch * 8;
                    // streaming output (qkv, hid, cross K/V: hundreds of MB read once by the next kernel): non-temporal
                    // stores measured -1.5 % step time in situ; the in-place residual stream and the attention
                    // output, which are re-read at once, measured no gain and keep plain stores
                    if (m < p.M && n < p.N)
                        __builtin_nontemporal_store(o, reinterpret_cast<u32x4*>(reinterpret_cast<__bf16*>(p.C) + p.c.off(m) + n));
                }
                }
            }
    } else {
        // fp32 staging, 16 rows x 256 B per round; 16-B chunk c of row r lives at chunk c ^ r.  After
        // the transpose a lane owns 4 consecutive columns of rows 4 it + g4: the residual (fp32 or bf16,
        // possibly the output buffer itself) is read and the result stored as whole row segments, rounded
        // once.  The residual of round k+1 is requested before the stores of round k are issued, so no
        // round waits for a full memory round trip behind its predecessor's stores.
        if (EPI == 2 || EPI == 4) {
            // bf16 residual stream updated in place: a lane owns 8 consecutive columns (16 B of bf16)
            // of rows 8 it + (lane >> 3); two staged fp32 chunks + one 16-B residual load per store.
            // The residual rows of round r + PF are requested before round r is staged; the loads are unconditional
            // (row clamped, N is a multiple of 256 here) so the compiler can count its waits instead of draining
            // the queue.  HK_EPI_PF (compile time): prefetch distance in rounds.
            constexpr int PF = PF_ROUNDS;
            u32x4 rr[PF + 1][2];
            // FAST: wave-uniform bases of the wave's residual / output / row-sum rows + one 32-bit lane offset each
            const char* rbase = reinterpret_cast<const char*>(p.res) + ((int64_t)mw * p.r.ld + nw) * 2;
            char* cbase = reinterpret_cast<char*>(p.C) + ((int64_t)mw * p.c.ld + nw) * 2;
            // (FAST requires equal row strides of C and the residual -- the in-place update -- so one lane offset serves both)
            const uint32_t rlo = (uint32_t)(((lane >> 3) * (int)p.c.ld + (lane & 7) * 8) * 2);
            const uint32_t clo = rlo;
            auto res_load8 = [&](int rnd, u32x4(&dst)[2]) __attribute__((always_inline)) {
#pragma unroll
                for (int it = 0; it < 2; ++it) {
                    if (FAST) {
                        dst[it] = *reinterpret_cast<const u32x4*>(rbase + (int64_t)(rnd * 16 + it * 8) * p.r.ld * 2 + rlo);
                    } else {
                        const int m = min(mw + rnd * 16 + it * 8 + (lane >> 3), p.M - 1), n = nw + (lane & 7) * 8;
                        dst[it] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const __bf16*>(p.res) + p.r.off(m) + n);
                    }
                }
            };
#pragma unroll
            for (int q = 0; q < PF; ++q) res_load8(q, rr[q]);
#pragma unroll
            for (int rnd = 0; rnd < 2 * NI; ++rnd) {
                const int mh = rnd / NI, i = rnd % NI;
#pragma unroll
                for (int nh = 0; nh < 2; ++nh)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int ch = nh * 8 + j * 4 + g4;
                        *reinterpret_cast<f32x4*>(stg + r16 * 256 + ((ch ^ r16) << 4)) = pre(acc[mh][nh][i][j], nh, j, mh, i);
                    }
                if (rnd + PF < 2 * NI) res_load8(rnd + PF, rr[(rnd + PF) % (PF + 1)]);
#pragma unroll
                for (int it = 0; it < 2; ++it) {
                    const int row = it * 8 + (lane >> 3), c8 = lane & 7;
                    const f32x4 a = *reinterpret_cast<const f32x4*>(stg + row * 256 + (((2 * c8) ^ row) << 4));
                    const f32x4 b = *reinterpret_cast<const f32x4*>(stg + row * 256 + (((2 * c8 + 1) ^ row) << 4));
                    const int m = mw + rnd * 16 + row, n = nw + c8 * 8;
                    const u32x4 r = rr[rnd % (PF + 1)][it];
                    const float v0 = a[0] + bf16_lo(r[0]), v1 = a[1] + bf16_hi(r[0]), v2 = a[2] + bf16_lo(r[1]),
                                v3 = a[3] + bf16_hi(r[1]), v4 = b[0] + bf16_lo(r[2]), v5 = b[1] + bf16_hi(r[2]),
                                v6 = b[2] + bf16_lo(r[3]), v7 = b[3] + bf16_hi(r[3]);
                    u32x4 o;
                    o[0] = pack_bf16x2(v0, v1);
                    o[1] = pack_bf16x2(v2, v3);
                    o[2] = pack_bf16x2(v4, v5);
                    o[3] = pack_bf16x2(v6, v7);
                    if (EPI == 4) {
                        // LayerNorm statistics of the updated residual stream, for the GEMM that consumes it with
                        // the norm folded in: (sum, sum of squares) of this wave's 64 columns of the row -- of the
                        // bf16 values as STORED, the ones the consumer multiplies: with the mean taken from the
                        // unrounded sums the folded mean term would not cancel exactly against the MFMA's
                        const float w0 = bf16_lo(o[0]), w1 = bf16_hi(o[0]), w2 = bf16_lo(o[1]), w3 = bf16_hi(o[1]),
                                    w4 = bf16_lo(o[2]), w5 = bf16_hi(o[2]), w6 = bf16_lo(o[3]), w7 = bf16_hi(o[3]);
                        float s1 = ((w0 + w1) + (w2 + w3)) + ((w4 + w5) + (w6 + w7));
                        float s2 = w0 * w0;
                        s2 = __builtin_fmaf(w1, w1, s2);
                        s2 = __builtin_fmaf(w2, w2, s2);
                        s2 = __builtin_fmaf(w3, w3, s2);
                        s2 = __builtin_fmaf(w4, w4, s2);
                        s2 = __builtin_fmaf(w5, w5, s2);
                        s2 = __builtin_fmaf(w6, w6, s2);
                        s2 = __builtin_fmaf(w7, w7, s2);
                        s1 = sum8_dpp(s1);
                        s2 = sum8_dpp(s2);
                        typedef __attribute__((ext_vector_type(2))) float f32x2s;
                        if (FAST) {
                            float* rp = p.row_part + ((int64_t)(mw + rnd * 16 + it * 8) * (p.N >> 6) + (nw >> 6)) * 2;   // wave-uniform
                            if (c8 == 0) *reinterpret_cast<f32x2s*>(rp + (uint32_t)((lane >> 3) * (p.N >> 6) * 2)) = f32x2s{s1, s2};
                        } else if (c8 == 0 && m < p.M) {
                            *reinterpret_cast<f32x2s*>(p.row_part + ((int64_t)m * (p.N >> 6) + (nw >> 6)) * 2) = f32x2s{s1, s2};
                        }
                    }
                    if (FAST)
                        *reinterpret_cast<u32x4*>(cbase + (int64_t)(rnd * 16 + it * 8) * p.c.ld * 2 + clo) = o;
                    else if (m < p.M && n < p.N)
                        *reinterpret_cast<u32x4*>(reinterpret_cast<__bf16*>(p.C) + p.c.off(m) + n) = o;
                }
            }
            return;
        }
        f32x4 rcur[4], rnxt[4];
        auto res_load = [&](int rnd, f32x4(&rr)[4]) {
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int m = mw + rnd * 16 + it * 4 + g4, n = nw + r16 * 4;
                rr[it] = (m < p.M && n < p.N) ? load_res4(p, p.r.off(m) + n) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        };
        if (p.res) res_load(0, rcur);
#pragma unroll
        for (int rnd = 0; rnd < 2 * NI; ++rnd) {
            const int mh = rnd / NI, i = rnd % NI;
#pragma unroll
            for (int nh = 0; nh < 2; ++nh)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int ch = nh * 8 + j * 4 + g4;
                    *reinterpret_cast<f32x4*>(stg + r16 * 256 + ((ch ^ r16) << 4)) = pre(acc[mh][nh][i][j], nh, j, mh, i);
                }
            if (p.res && rnd < 2 * NI - 1) res_load(rnd + 1, rnxt);
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int row = it * 4 + g4, ch = r16;
                f32x4 v = *reinterpret_cast<const f32x4*>(stg + row * 256 + ((ch ^ row) << 4));
                const int m = mw + rnd * 16 + row, n = nw + ch * 4;
                if (m < p.M && n < p.N) {
                    if (p.res) v += rcur[it];
                    if (p.out_f32) {
                        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.C) + p.c.off(m) + n) = v;
                    } else {
                        u32x2 o;
                        o[0] = pack_bf16x2(v[0], v[1]);
                        o[1] = pack_bf16x2(v[2], v[3]);
                        *reinterpret_cast<u32x2*>(reinterpret_cast<__bf16*>(p.C) + p.c.off(m) + n) = o;
                    }
                }
            }
#pragma unroll
            for (int it = 0; it < 4; ++it) rcur[it] = rnxt[it];
        }
    }
}

#ifndef HK_EPI_FAST
#define HK_EPI_FAST 1   // 0: every tile takes the general addressing path (A/B builds)
#endif
template <int EPI, int PF_ROUNDS = HK_EPI_PF, int NI = 4>
__device__ __forceinline__ void pp_epilogue(const GemmParams& p, f32x4 (&acc)[2][2][NI][2], char* stg, const char* aux,
                                            const int lane, const int wm, const int wn, const int m0, const int n0) {
    // wave-uniform: whole (32 NI) x 64 sub-tile inside the problem, plain rows for C (and for the in-place residual)
    const bool fast = HK_EPI_FAST != 0 && EPI != 1 && EPI != 5 && p.c.rpb == 0 && (EPI == 0 || EPI == 3 || (p.r.rpb == 0 && p.r.ld == p.c.ld)) &&
                      m0 + wm * (32 * NI) + 32 * NI <= p.M && n0 + wn * 64 + 64 <= p.N;
    if (fast) pp_epilogue_impl<EPI, PF_ROUNDS, true, NI>(p, acc, stg, aux, lane, wm, wn, m0, n0);
    else pp_epilogue_impl<EPI, PF_ROUNDS, false, NI>(p, acc, stg, aux, lane, wm, wn, m0, n0);
}

// ---- per-quadrant epilogue of the 256x256 kernel (HK_EPI_QUAD) ----
// A wave's 64x32 output quadrant (MH, NH) is final after phase p of a tile's LAST K-tile and is not touched again before
// phase p of the next tile's first K-tile.  Its epilogue therefore runs in the read / stage section of the phase that
// follows, i.e. while the partner wave of the SIMD (the other M-group, one barrier behind or ahead) is inside its
// 16-MFMA cluster: the vector math, the LDS transpose and the store issue of the epilogue then share the SIMD with matrix
// instructions instead of leaving the matrix pipe idle for a whole epilogue per tile.
// Slab use (4 KB per wave): bytes 0..1535 keep the tile's epilogue operands (row statistics, bias, column sums of W) for
// all four quadrants; the transpose goes through bytes 2048..4095, 32 rows x 64 B at a time (two rounds per quadrant):
// 16-byte chunk c of row r at chunk c ^ (((r >> 2) & 1) << 1), which spreads the 16 rows x 32 B of one ds_write_b64 over
// all banks.  Stores are 16 rows x 64 B per instruction; the other 64-byte half of each 128-byte line follows one
// quadrant later.  FAST-form addressing only (whole sub-tile inside the problem, plain rows): the caller checks.
// MEASURED SLOWER, so off by default (HK_EPI_QUAD=1 builds it; same bits as the ordinary epilogue, tools/ab_quad.sh,
// profiles/r03_epi_quad_ab.txt): qkv 567 -> 595 us, fc1+GELU 797 -> 817, cross K/V 1557 -> 1620, step 84.6 -> 87.3 ms.  Timing probes of
// the same form: without its global stores 558 / 780 / 1461, without stores and LDS transpose (vector math only) 535 / 760 /
// 1386 -- against ~455 for qkv with no epilogue at all, i.e. even the bare vector math costs ~4 us per tile when it is
// interleaved, three times what it costs in the ordinary epilogue: a read section is already as long as the partner's MFMA
// cluster (its fragment reads share the LDS with the other group's), every barrier interval of the last K-tile grows by the
// section's vector work, and the two groups' sections -- concurrent in the ordinary epilogue -- are serialised by the barriers.
#ifndef HK_EPI_QUAD
#define HK_EPI_QUAD 0
#endif

template <int EPI, int MH, int NH>
__device__ __forceinline__ void pp_quad_epilogue(const GemmParams& p, f32x4 (&acc)[2][2][4][2], char* stg, const int lane,
                                                 const int wm, const int wn, const int m0, const int n0) {
    static_assert(EPI == 0 || EPI == 3, "quadrant epilogue: bf16-out kinds");
    constexpr bool FOLD = EPI == 3;
    typedef __attribute__((ext_vector_type(2))) float f32x2s;
    const int r16 = lane & 15, g4 = lane >> 4;
    f32x4 bias[2], cc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        bias[j] = p.bias != nullptr ? *reinterpret_cast<const f32x4*>(stg + 1024 + (NH * 32 + j * 16 + g4 * 4) * 4)
                                    : f32x4{0.f, 0.f, 0.f, 0.f};
        if (FOLD) cc[j] = *reinterpret_cast<const f32x4*>(stg + 1280 + (NH * 32 + j * 16 + g4 * 4) * 4);
    }
    char* tr = stg + 2048;
    const int nq = n0 + wn * 64 + NH * 32;
    char* rb = reinterpret_cast<char*>(p.C) + ((int64_t)(m0 + wm * 128 + MH * 64) * p.c.ld + nq) * 2;   // wave-uniform
    const uint32_t lo = (uint32_t)(((lane >> 2) * (int)p.c.ld + (lane & 3) * 8) * 2);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
        for (int ii = 0; ii < 2; ++ii) {
            const int i = half * 2 + ii;
            float rs = 1.f, rm = 0.f;
            if (FOLD) {
                const f32x2s st = *reinterpret_cast<const f32x2s*>(stg + (MH * 64 + i * 16 + r16) * 8);
                rs = st[1];
                rm = st[1] * st[0];
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                f32x4 v = acc[MH][NH][i][j];
                if (FOLD) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = __builtin_fmaf(v[e], rs, __builtin_fmaf(-rm, cc[j][e], bias[j][e]));
                } else {
                    v += bias[j];
                }
                if (p.scale_cols != 0 && nq + j * 16 + g4 * 4 < p.scale_cols) v *= p.scale;
                if (p.act == HK_ACT_GELU_ERF) v = gelu_erf4_bf16dst(v);
                u32x2 o;
                o[0] = pack_bf16x2(v[0], v[1]);
                o[1] = pack_bf16x2(v[2], v[3]);
                const int row = ii * 16 + r16;
                const int ch = j * 2 + (g4 >> 1);
                *reinterpret_cast<u32x2*>(tr + row * 64 + ((ch ^ (((row >> 2) & 1) << 1)) << 4) + (g4 & 1) * 8) = o;
                acc[MH][NH][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
        u32x4 o[2];
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int row = it * 16 + (lane >> 2), c = lane & 3;
            o[it] = *reinterpret_cast<const u32x4*>(tr + row * 64 + ((c ^ (((row >> 2) & 1) << 1)) << 4));
        }
#pragma unroll
        for (int it = 0; it < 2; ++it)
            __builtin_nontemporal_store(o[it], reinterpret_cast<u32x4*>(rb + (int64_t)(half * 32 + it * 16) * p.c.ld * 2 + lo));
    }
}

// ---------------------------------------------------------------------------------------------
// 256x256x64 "ping-pong" kernel: 8 waves as 2 (M) x 4 (N); the two M-groups of waves run one
// barrier apart, so on every SIMD one wave is inside a 16-MFMA cluster while its partner reads
// LDS fragments and issues LDS-DMA for the tiles ahead.
//
// LDS: 2 K-tile buffers x {A rows 0-127, A rows 128-255, W rows 0-127, W rows 128-255} x [128][64] bf16.
// Per K-tile a wave runs 4 phases, one 64x32 output quadrant each:
//     phase      reads (ds_read_b128)            stages (2 LDS-DMA per thread)       MFMA
//     1          A rows 0-63 (8), W cols 0-31 (4)   A-half1 of tile t+1                 q(0,0)
//     2          W cols 32-63 (4)                   W-half0 of tile t+1                 q(0,1)
//     3          A rows 64-127 (8)                  W-half1 of tile t+1                 q(1,1)
//     4          W cols 0-31 (4)                    A-half0 of tile t+2, vmcnt(2)       q(1,0)
// Each phase is  {reads, stage} s_barrier {lgkmcnt(0), 16 MFMA} s_barrier.  With B(k) the k-th
// barrier instance, group 0 runs phase p of tile t between B(8t+2p-2) and B(8t+2p), group 1 one
// instance later.  Hazards (interval i = between B(i) and B(i+1); a read issued in interval i has
// completed before its wave arrives at B(i+2)):
//   WAR  a slot of buffer t&1 is restaged for tile t+2 only after every read of tile t from it:
//        A-half0: last read group 0 phase 3 (interval 8t+4) -> staged from interval 8t+6 (phase 4) on;
//        A-half1: last read group 1 phase 3 (8t+5)          -> staged from 8t+8 (phase 1 of t+1) on;
//        W-halves: last read group 1 phase 4 (8t+7)         -> staged from 8t+10 (phases 2,3 of t+1) on.
//   RAW  every wave retires all LDS-DMA of tile t+1 with the vmcnt(2) of its phase 4 of tile t
//        (the 2 left in flight are A-half0 of t+2) and then arrives at a barrier (group 1: B(8t+8))
//        that every reader of tile t+1 (first: group 0 in interval 8t+8) has to pass first.
// Loads therefore stay in flight across 6-20 barriers and vmcnt never drains to 0 in the loop.
// ---------------------------------------------------------------------------------------------
#define HK_BAR() __builtin_amdgcn_s_barrier()

// Persistent form: one workgroup per CU walks a list of output tiles and treats (tile, K-tile) as ONE
// stream of K-tiles, so the staging pipeline above never drains at a tile boundary: while a tile's
// last K-tiles are multiplied the next tile's first K-tiles are already landing, the epilogue's
// stores are fire-and-forget, and there is no workgroup launch / prologue latency / store-ack wait
// per tile (measured 6-7 us of a 30-40 us tile at K = 1024 in the one-tile-per-workgroup form).
// The epilogue transposes each wave's 128x64 sub-tile through a private 4 KB LDS region above the
// operand buffers (128 KB + 8 x 4 KB = all 160 KB of the CU) and writes whole row segments.
// EPI selects the one epilogue this instantiation carries: 0 = bf16 out, no residual; 1 = fp32 staging
// (fp32 out and/or fp32 residual); 2 = bf16 residual stream updated in place (4 = 2 that also emits the row sums
// the next LayerNorm needs, row_part); 5 = bf16 out with SwiGLU on interleaved (gate, up) columns; 3 = 0 with the LayerNorm of A folded
// in (row_stats / col_c).  The folded form is its own instantiation because its extra live values cost the main loop
// of EPI 0 ~20 % when they share one body (measured in situ), and its row statistics reach the epilogue without
// registers or exposed latency: at the start of a tile every wave requests the (mean, rstd) pairs of
// its 128 rows -- 1 KB, one LDS-DMA instruction -- into its idle staging slab; the counted vmcnt of that K-tile's
// first phase 4 retires it with the operand loads.
// NI = 3 (round 3): the same kernel on a 192 x 256 tile -- a wave owns 96 x 64 (three 16-row blocks per M-half, 12 MFMAs
// per phase), the A half-tiles are 96 rows (12 LDS-DMA pieces: waves 4-7 stage the pieces of waves 0-3 a second time as
// their second piece, same source and destination, so every wave keeps 2 pieces per stage and the counted waits are
// unchanged; the LDS layout keeps 128-row slots).  For problems whose 256-row tiles make one partial round: 2816 x 4096
// (prefill o_proj / down_proj) is 176 tiles of 256 rows = 69 % of the CUs for a whole tile time, or 240 tiles of 192 rows
// for three quarters of it.
template <int EPI, int NI = 4>
__global__ __launch_bounds__(512) void hk_gemm_pp_kernel(const GemmParams p) {
    constexpr int BK = 64;
    constexpr int HALF = 128 * BK;  // elements in one half-tile slot (16 KB)
    constexpr int BUF = 4 * HALF;   // A-h0, A-h1, W-h0, W-h1
    constexpr int AH = 32 * NI;     // rows of an A half-tile = rows of a wave's sub-tile
    constexpr int BM = 2 * AH;      // rows of the output tile

    extern __shared__ __attribute__((aligned(16))) char smem[];
    __bf16* lds = reinterpret_cast<__bf16*>(smem);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;

    // ---- this workgroup's tile list: the workgroups that share an XCD (blockIdx % 8) split one
    // contiguous run of tiles (n fastest) round-robin, so concurrent tiles share A panels in that L2 ----
    const int nwg = p.nbm * p.nbn;
    const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
    const int nbx = ((int)gridDim.x - xcd + 7) >> 3;  // workgroups on this XCD
    const int q8 = nwg >> 3, r8 = nwg & 7;
    const int c0 = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
    const int cnt = q8 + (xcd < r8 ? 1 : 0);
    const int ntile = jb < cnt ? (cnt - jb + nbx - 1) / nbx : 0;
    const int KT = (HK_DBG(p) & 2) ? 1 : p.K / BK;  // dbg 2: timing-only build, one K-tile per tile
    if (ntile == 0) return;                      // uniform per workgroup

    // ---- loader: one LDS-DMA wave-instruction = 8 rows x 128 B; a half-tile = 2 per thread.
    // Each slot kind (A-h0, A-h1, W-h0, W-h1) has its own cursor into the K-tile stream. ----
    const int lrow = lane >> 3, lpc = lane & 7;
    uint32_t go[4][2];  // [slot kind][round]: byte offset (from A / W) of the row piece of the tile being staged
    int s_tile[4], s_k[4], s_par[4];
    auto set_ptrs = [&](int kind) {
        const int lid = c0 + jb + s_tile[kind] * nbx;
        int tm, tn;
        tile_coords(lid, p.nbm, p.nbn, p.gm, tm, tn);
        const int t0 = kind < 2 ? tm * BM : tn * 256;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            // piece (8 rows) this wave stages in round i; A pieces past the half-tile's AH rows repeat an earlier wave's
            int pc = i * 8 + wave;
            if constexpr (NI != 4) pc = (kind < 2 && pc * 8 >= AH) ? pc - 4 : pc;
            const int row = pc * 8 + lrow;
            const int sw = (lpc ^ ((row >> 1) & 7)) << 3;
            if (kind < 2) {
                const int m = min(t0 + kind * AH + row, p.M - 1);
                go[kind][i] = (uint32_t)((p.a.off(m) + sw) * 2);
            } else {
                const int n = min(t0 + (kind - 2) * 128 + row, p.N - 1);
                go[kind][i] = (uint32_t)((p.wrow(n) + sw) * 2);
            }
        }
    };
    auto stage = [&](int kind) -> bool {  // stage the cursor's half-tile, advance the cursor
        if (s_tile[kind] >= ntile) return false;
        __bf16* dst = lds + s_par[kind] * BUF + kind * HALF + wave * 8 * BK;
        // scalar base (operand + K offset) + the piece's 32-bit lane offset: no 64-bit vector address per piece
        const __bf16* kb = kind < 2 ? p.A + s_k[kind] * BK : p.W + s_k[kind] * p.wk();
        glds16_s(kb, go[kind][0], dst);
        if constexpr (NI == 4) glds16_s(kb, go[kind][1], dst + 64 * BK);
        else glds16_s(kb, go[kind][1], dst + ((kind < 2 && (8 + wave) * 8 >= AH) ? 32 : 64) * BK);
        s_par[kind] ^= 1;
        if (++s_k[kind] == KT) {
            s_k[kind] = 0;
            if (++s_tile[kind] < ntile) set_ptrs(kind);
        }
        return true;
    };
#pragma unroll
    for (int kind = 0; kind < 4; ++kind) {
        s_tile[kind] = 0;
        s_k[kind] = 0;
        s_par[kind] = 0;
        set_ptrs(kind);
    }

    // ---- fragment offsets (elements): row (lane & 15), logical chunk kk*4 + (lane >> 4) ----
    const int sw = (lane >> 1) & 7;
    const int foff0 = (lane & 15) * BK + ((((lane >> 4)) ^ sw) << 3);
    const int foff1 = (lane & 15) * BK + (((4 + (lane >> 4)) ^ sw) << 3);
    const int a_base = wm * HALF;                                        // + mh*64*BK
    const int w_base = (2 + (wn >> 1)) * HALF + (wn & 1) * 64 * BK;      // + nh*32*BK

    f32x4 acc[2][2][NI][2];
    bf16x8 xa[2][NI], wb[2][2];
    auto zero_acc = [&]() {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int i = 0; i < NI; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[a][b][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    zero_acc();
#ifndef HK_ZERO_C
#define HK_ZERO_C 1
#endif
    constexpr bool ZC_AFTER = HK_ZERO_C != 0 && (EPI == 2 || EPI == 4);   // see the K loop
    constexpr bool QUAD = HK_EPI_QUAD != 0 && (EPI == 0 || EPI == 3) && NI == 4;   // quadrant epilogues inside the last K-tile

    auto load_a = [&](int buf, int mh) {
        const __bf16* s0 = lds + buf * BUF + a_base + mh * (16 * NI) * BK;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            xa[0][i] = *reinterpret_cast<const bf16x8*>(s0 + i * 16 * BK + foff0);
            xa[1][i] = *reinterpret_cast<const bf16x8*>(s0 + i * 16 * BK + foff1);
        }
    };
    auto load_w = [&](int buf, int nh) {
        const __bf16* s0 = lds + buf * BUF + w_base + nh * 32 * BK;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            wb[0][j] = *reinterpret_cast<const bf16x8*>(s0 + j * 16 * BK + foff0);
            wb[1][j] = *reinterpret_cast<const bf16x8*>(s0 + j * 16 * BK + foff1);
        }
    };
    // FIRST: the first K-tile of an output tile multiplies into a zero C operand (an inline constant of the MFMA) instead of
    // into accumulators that 128 v_mov per wave and tile would have had to clear first (HK_ZERO_C=0: A/B build with the clear)
#ifndef HK_ZERO_C
#define HK_ZERO_C 1
#endif
#define HK_MMA(MH, NH)                                                                                      \
    do {                                                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                                  \
        HK_BAR();                                                                                           \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                                  \
        __builtin_amdgcn_s_setprio(1);                                                                      \
        _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) _Pragma("unroll") for (int i = 0; i < NI; ++i)     \
            _Pragma("unroll") for (int j = 0; j < 2; ++j) acc[MH][NH][i][j] =                               \
                __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[kk][j], xa[kk][i],                               \
                    (FIRST && kk == 0) ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[MH][NH][i][j], 0, 0, 0);           \
        __builtin_amdgcn_s_setprio(0);                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                  \
        HK_BAR();                                                                                           \
    } while (0)

    char* stg = smem + 2 * BUF * 2 + wave * 4096;
    auto epilogue = [&](int m0, int n0) __attribute__((always_inline)) { pp_epilogue<EPI, HK_EPI_PF, NI>(p, acc, stg, stg, lane, wm, wn, m0, n0); };

    // (Measured and not kept: starting eight classes of workgroups a fraction of a tile time apart (s_sleep), so that
    // a CU's epilogue stores would meet the others' main loops instead of 255 other store bursts: every launch got slower
    // by exactly the largest delay, fc1 850 -> 885 / 921 / 960 us at 1 / 2 / 4 sleep units -- the epilogue is not bound by
    // the synchronised write burst.  What the bf16-out epilogue of a 256x256 tile costs, from diagnostic builds that drop
    // parts of it (fc1 shape, one box, us per launch, 24.5 tiles per CU): complete 760 (no activation) / 960 (GELU); without
    // the global stores 706 / 878; without the bias / GELU math 726 / 834; without both 660 / 750; without any epilogue
    // 616 / 694 -- stores 7-9 %, GELU math 10 %, LDS transpose + pack + the idle matrix pipe around them the rest: no
    // single part dominates, and none overlaps the main loop.  Without the LDS transpose -- a lane storing 2 x 16 B of one
    // row straight from its accumulators, 16-byte pieces 32 bytes apart per instruction, a timing probe with the columns
    // left unpermuted -- qkv 690 -> 762 us and fc1+GELU 852 -> 977 us: whole 128-byte row segments are worth the transpose.)
    // ---- prologue: K-tile 0 of the stream complete, A-half0 of K-tile 1 in flight ----
    stage(0);
    stage(1);
    stage(2);
    stage(3);
    if (stage(0)) {
        asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    HK_BAR();
    if (wm == 1) HK_BAR();  // stagger group 1 by one barrier

    unsigned long long* trc = HK_TRACE(p) ? HK_TRACE(p) + (size_t)blockIdx.x * 64 : nullptr;
    if (trc && tid == 0) {
        trc[0] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));  // HW_ID
        trc[1] = __builtin_amdgcn_s_memrealtime();
    }
    int buf = 0;
    for (int ti = 0; ti < ntile; ++ti) {
        if (p.bias != nullptr) {
            // this tile's bias slice -> the wave's staging slab (idle until the epilogue); columns past N are never stored
            int tm_, tn_;
            tile_coords(c0 + jb + ti * nbx, p.nbm, p.nbn, p.gm, tm_, tn_);
            glds4_s(p.bias, (uint32_t)min(tn_ * 256 + wn * 64 + lane, p.N - 1) * 4u, stg + 1024);
            if (EPI == 3) glds4_s(p.col_c, (uint32_t)min(tn_ * 256 + wn * 64 + lane, p.N - 1) * 4u, stg + 1280);
        }
        if (EPI == 3) {
            // folded LayerNorm: (mean, rstd) of the wave's 128 rows -> its staging slab (idle until the epilogue),
            // 2 rows per lane.  Issued outside the K loop, whose body stays that of EPI 0; older than every operand
            // load of this tile, so the first counted vmcnt retires it.
            int tm_, tn_;
            tile_coords(c0 + jb + ti * nbx, p.nbm, p.nbn, p.gm, tm_, tn_);
            // (128 rows from the wave's first; the statistics array is padded to a multiple of 256 rows: clamp for BM = 192)
            if constexpr (NI == 4) glds16_v(p.row_stats + 2 * ((int64_t)tm_ * 256 + wm * 128 + 2 * lane), stg);
            else glds16_v(p.row_stats + 2 * (int64_t)min(tm_ * BM + wm * AH + 2 * lane, (p.M + 255) / 256 * 256 - 2), stg);
        }
        auto ktile = [&](auto first_) __attribute__((always_inline)) {
            constexpr bool FIRST = decltype(first_)::value;
            // phase 1
            load_a(buf, 0);
            load_w(buf, 0);
            stage(1);
            HK_MMA(0, 0);
            // phase 2
            load_w(buf, 1);
            stage(2);
            HK_MMA(0, 1);
            // phase 3
            load_a(buf, 1);
            stage(3);
            HK_MMA(1, 1);
            // phase 4
            load_w(buf, 0);
            if (stage(0)) {
                asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            HK_MMA(1, 0);
            buf ^= 1;
        };
        // measured per epilogue kind on one box (zc0 / zc1 builds interleaved): proj 245.5 -> 241.0 us, fc2 681.7 -> 672.4; qkv
        // 562.5 -> 560.8 (noise), fc1+GELU 766.0 -> 773.4 (slower: that instantiation sits at 256 registers with the peeled copy
        // of the K-tile body) -- so only the in-place residual epilogues take the zero-C form
        constexpr bool ZC = HK_ZERO_C != 0 && (EPI == 2 || EPI == 4);
        const int lid = c0 + jb + ti * nbx;
        int tm, tn;
        tile_coords(lid, p.nbm, p.nbn, p.gm, tm, tn);
        if constexpr (QUAD) {
            // wave-uniform: the wave's whole sub-tile inside the problem and plain C rows -> its quadrants leave during the
            // last K-tile; otherwise the ordinary epilogue after it.  Both forms pass the same barriers.
            // (two K-tiles at least: the tile's epilogue operands are retired by the first K-tile's counted wait)
            const bool fastq = KT >= 2 && p.c.rpb == 0 && tm * BM + wm * AH + AH <= p.M && tn * 256 + wn * 64 + 64 <= p.N && HK_DBG(p) == 0;
            const int kt_main = fastq ? KT - 1 : KT;
            for (int t = 0; t < kt_main; ++t) ktile(std::false_type{});
            if (fastq) {
                const int m0 = tm * 256, n0 = tn * 256;
                constexpr bool FIRST = false;
                // phase 1
                load_a(buf, 0);
                load_w(buf, 0);
                stage(1);
                HK_MMA(0, 0);
                // phase 2: + epilogue of q(0,0)
                load_w(buf, 1);
                stage(2);
                pp_quad_epilogue<EPI, 0, 0>(p, acc, stg, lane, wm, wn, m0, n0);
                HK_MMA(0, 1);
                // phase 3: + epilogue of q(0,1)
                load_a(buf, 1);
                stage(3);
                pp_quad_epilogue<EPI, 0, 1>(p, acc, stg, lane, wm, wn, m0, n0);
                HK_MMA(1, 1);
                // phase 4: the counted wait leaves this phase's 2 LDS-DMA and the 4 stores of q(0,1) in flight; + epilogue of q(1,1)
                load_w(buf, 0);
                if (stage(0)) {
                    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                } else {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                pp_quad_epilogue<EPI, 1, 1>(p, acc, stg, lane, wm, wn, m0, n0);
                HK_MMA(1, 0);
                buf ^= 1;
                // q(1,0): in front of the next tile's first read section (the partner group is one barrier behind / ahead)
                pp_quad_epilogue<EPI, 1, 0>(p, acc, stg, lane, wm, wn, m0, n0);
            } else {
                pp_epilogue_impl<EPI, HK_EPI_PF, false, NI>(p, acc, stg, stg, lane, wm, wn, tm * BM, tn * 256);
                zero_acc();
            }
        } else {
        if constexpr (ZC) {
            ktile(std::true_type{});
            for (int t = 1; t < KT; ++t) ktile(std::false_type{});
        } else {
            for (int t = 0; t < KT; ++t) ktile(std::false_type{});
        }
        if (trc && tid == 0 && ti < 31) trc[2 + 2 * ti] = __builtin_amdgcn_s_memrealtime();
        epilogue(tm * BM, tn * 256);
        if (trc && tid == 0 && ti < 31) trc[3 + 2 * ti] = __builtin_amdgcn_s_memrealtime();
        if constexpr (!ZC_AFTER) zero_acc();
        }
    }
    if (wm == 0) HK_BAR();  // equalise the barrier count of the two groups
#undef HK_MMA
}

// ---------------------------------------------------------------------------------------------
// "Duo" kernel: TWO persistent workgroups per CU, each 4 waves (one per SIMD) on a 256x128 output tile, so that one
// workgroup's epilogue (store-bound, matrix pipe idle: 20-30 % of the K = 1024 launches of the 256x256 kernel) runs
// under the other workgroup's main loop.  A wave owns 128 rows x 64 columns exactly as in the 256x256 kernel (same
// accumulator layout, same LDS fragment traffic per MFMA, the same pp_epilogue<EPI>); what changes is the operand
// supply: a workgroup stages (256 + 128) rows per K-step instead of (256 + 256) for twice the MFMAs, i.e. 1.5x the
// L2 -> LDS bytes per flop, issued by 4 waves instead of 8.
//
// K-step = 32 (one v_mfma_f32_16x16x32_bf16 deep), ring of 3 stages x {A [256][32], W [128][32]} = 72 KB, + 6 KB of
// epilogue operands (per wave: row statistics 1 KB, bias 256 B, column sums 256 B) = 78 KB per workgroup, 156 KB per CU.
// LDS rows are 64 B; physical 16-B chunk pc of row r holds logical chunk pc ^ g((r >> 2) & 3), g = (0, 2, 3, 1), which
// makes every ds_read_b128 16-lane group ({0-3,12-15,20-27}, ...) hit 16 distinct 16-B slots.
//
// Stream of K-steps s = 0, 1, ... over the workgroup's tile list (continuous across tiles), stage s in slot s % 3.
// Per step a wave runs
//     H1   16 MFMA: quadrants (0,0), (0,1)   operands A-half0(s), W(s)
//     mid  vmcnt -> own pieces of stage s+1 landed; s_barrier -> everyone's landed, and every fragment of stage s has
//          been read (during step s-1) => slot s % 3 is free: issue the LDS-DMA of stage s+3 into it; read A-half0(s+1)
//     H2   16 MFMA: quadrants (1,0), (1,1)   operands A-half1(s), W(s);  W-half0(s+1) is read after (1,0), W-half1(s+1)
//          and A-half1(s+1) after (1,1)
// so a stage is requested two full steps before its first read and every fragment is read >= 8 MFMAs ahead of its use.
// At the last step of a tile the issue of stage s+3 and the reads of stage s+1 wait until after the epilogue: the
// free slot's A region -- 4 KB per wave, exactly the bytes that wave's own A pieces fill -- is the wave's transpose
// slab, so no barrier separates the epilogue from anything and its registers are free of next-step fragments.
// ---------------------------------------------------------------------------------------------
template <int EPI>
__global__ __launch_bounds__(256, 2) void hk_gemm_duo_kernel(const GemmParams p) {
    constexpr int BK = 32;
    constexpr int A_EL = 256 * BK, W_EL = 128 * BK;   // elements per stage
    constexpr int STAGE = A_EL + W_EL;                // 24 KB
    constexpr int NS = 3;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    __bf16* lds = reinterpret_cast<__bf16*>(smem);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    char* aux = smem + NS * STAGE * 2 + wave * 1536;

    // ---- tile list: as in the 256x256 kernel, with 128-column tiles ----
    const int nwg = p.nbm * p.nbn;
    const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
    const int nbx = ((int)gridDim.x - xcd + 7) >> 3;
    const int q8 = nwg >> 3, r8 = nwg & 7;
    const int c0 = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
    const int cnt = q8 + (xcd < r8 ? 1 : 0);
    const int ntile = jb < cnt ? (cnt - jb + nbx - 1) / nbx : 0;
    const int KT = (HK_DBG(p) & 2) ? 2 : p.K / BK;
    if (ntile == 0) return;
    const int S = ntile * KT;

    // ---- loader: one LDS-DMA wave-instruction = 16 rows x 64 B.  Wave w fills A rows [64 w, 64 w + 64) (4 pieces: bytes
    // [4096 w, 4096 w + 4096) of the stage) and W rows [32 w, 32 w + 32) (2 pieces).  The host guarantees plain row-major
    // operands and whole tiles (M % 256 == 0, N % 128 == 0; leftover rows go to the multi-stage kernel), so everything that
    // depends on the tile, the wave, the piece or k lives in the SCALAR base and a lane keeps two constant byte offsets. ----
    // (the two offsets are recomputed from the lane id at every use -- a handful of VALU instructions per K-step -- instead
    // of being kept live across the epilogue, where they were spilled and their reload drained the whole vmcnt queue)
    auto lane_offsets = [&](uint32_t& voa, uint32_t& vow) __attribute__((always_inline)) {
        int l = lane;
        asm volatile("" : "+v"(l));   // opaque per use: not a loop invariant the compiler may hoist and spill
        const int lrow = l >> 2;
        const int lsw = (l & 3) ^ ((0x78 >> (((l >> 4) & 3) * 2)) & 3);   // logical chunk this lane fetches
        voa = (uint32_t)((lrow * (int)p.a.ld + lsw * 8) * 2);
        vow = (uint32_t)((lrow * (int)p.ldw + lsw * 8) * 2);
    };
    const __bf16 *sa = nullptr, *sw = nullptr;   // first element of this wave's rows of the tile being staged
    int s_tile = 0, s_k = 0, s_slot = 0;
    auto set_ptrs = [&]() {
        int tm, tn;
        tile_coords(c0 + jb + s_tile * nbx, p.nbm, p.nbn, p.gm, tm, tn);
        sa = p.A + (int64_t)(tm * 256 + wave * 64) * p.a.ld;
        sw = p.W + (int64_t)(tn * 128 + wave * 32) * p.ldw;
    };
    auto stage = [&]() -> bool {
        if (s_tile >= ntile) return false;
        __bf16* dst = lds + s_slot * STAGE;
        const __bf16* ka = sa + s_k * BK;
        const __bf16* kw = sw + s_k * BK;
        uint32_t voa, vow;
        lane_offsets(voa, vow);
#pragma unroll
        for (int i = 0; i < 4; ++i) glds16_s(ka + (int64_t)i * 16 * p.a.ld, voa, dst + (wave * 64 + i * 16) * BK);
#pragma unroll
        for (int i = 0; i < 2; ++i) glds16_s(kw + (int64_t)i * 16 * p.ldw, vow, dst + A_EL + (wave * 32 + i * 16) * BK);
        s_slot = s_slot == NS - 1 ? 0 : s_slot + 1;
        if (++s_k == KT) {
            s_k = 0;
            if (++s_tile < ntile) set_ptrs();
        }
        return true;
    };
    // epilogue operands of tile ti -> the wave's aux region (older than every operand load that follows them)
    auto stage_aux = [&](int ti) {
        if (p.bias == nullptr && EPI != 3) return;
        int tm_, tn_;
        tile_coords(c0 + jb + ti * nbx, p.nbm, p.nbn, p.gm, tm_, tn_);
        if (p.bias != nullptr) glds4_s(p.bias, (uint32_t)min(tn_ * 128 + wn * 64 + lane, p.N - 1) * 4u, aux + 1024);
        if (EPI == 3) {
            glds4_s(p.col_c, (uint32_t)min(tn_ * 128 + wn * 64 + lane, p.N - 1) * 4u, aux + 1280);
            glds16_v(p.row_stats + 2 * ((int64_t)tm_ * 256 + wm * 128 + 2 * lane), aux);
        }
    };

    // ---- fragment offsets (elements): row (lane & 15), logical chunk (lane >> 4) ----
    const int foff = (lane & 15) * BK + ((((lane >> 4)) ^ ((0x78 >> ((((lane & 15) >> 2)) * 2)) & 3)) << 3);
    const int a_base = wm * 128 * BK;
    const int w_base = A_EL + wn * 64 * BK;

    f32x4 acc[2][2][4][2];
    bf16x8 xa0[4], xa1[4], wf[2][2];
    auto zero_acc = [&]() {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[a][b][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    zero_acc();
    auto read_a = [&](int slot, int mh, bf16x8(&x)[4]) __attribute__((always_inline)) {
        const __bf16* s0 = lds + slot * STAGE + a_base + mh * 64 * BK + foff;
#pragma unroll
        for (int i = 0; i < 4; ++i) x[i] = *reinterpret_cast<const bf16x8*>(s0 + i * 16 * BK);
    };
    auto read_w = [&](int slot, int nh) __attribute__((always_inline)) {
        const __bf16* s0 = lds + slot * STAGE + w_base + foff + nh * 32 * BK;
#pragma unroll
        for (int j = 0; j < 2; ++j) wf[nh][j] = *reinterpret_cast<const bf16x8*>(s0 + j * 16 * BK);
    };
#define HK_DUO_MMA(MH, NH, X)                                                                                           \
    do {                                                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                                              \
        __builtin_amdgcn_s_setprio(1);                                                                                  \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j) acc[MH][NH][i][j] =  \
            __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[NH][j], X[i], acc[MH][NH][i][j], 0, 0, 0);                       \
        __builtin_amdgcn_s_setprio(0);                                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                                              \
    } while (0)

    // ---- prologue: stages 0..2 requested, stage 0 landed, its fragments read ----
    set_ptrs();
    stage_aux(0);
    stage();
    const bool has1 = stage();
    const bool has2 = stage();
    if (has2) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if (has1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    HK_BAR();
    read_a(0, 0, xa0);
    read_w(0, 0);
    read_w(0, 1);
    read_a(0, 1, xa1);

    int rslot = 0;   // slot of the step being multiplied
    int kleft = KT;  // steps left in the current tile, this one included
    int ti = 0;
    // One step.  Fragment registers: A-half0, A-half1 and ONE set of W fragments, each refilled as soon as its last MFMA
    // of the step has been issued: A-half0 at mid-step, W-half0 after quadrant (1,0), W-half1 and A-half1 after (1,1) --
    // the next step uses them in that order (quadrants (0,0), (0,1), (1,0), (1,1)), so every fragment is read at least 8
    // MFMAs (128 matrix cycles) before its first use.
    for (int s = 0; s < S; ++s) {
        const bool more = s + 1 < S;
        const bool tile_end = kleft == 1;
        const bool refill = more && !tile_end;
        const int nslot = rslot == NS - 1 ? 0 : rslot + 1;
        HK_DUO_MMA(0, 0, xa0);
        HK_DUO_MMA(0, 1, xa0);
        if (more) {
            // every LDS read of this wave so far (all of stage s) has returned: stage s's slot may be refilled after the barrier
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (s + 2 < S) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            HK_BAR();
            if (!tile_end) {
                stage();
                read_a(nslot, 0, xa0);
            }
        }
        HK_DUO_MMA(1, 0, xa1);
        if (refill) read_w(nslot, 0);
        HK_DUO_MMA(1, 1, xa1);
        if (refill) {
            read_w(nslot, 1);
            read_a(nslot, 1, xa1);
        }
        if (tile_end) {
            int tm, tn;
            tile_coords(c0 + jb + ti * nbx, p.nbm, p.nbn, p.gm, tm, tn);
            // the slot just multiplied is free (no stage s+3 was issued into it): its A region, 4 KB per wave, is the
            // wave's transpose slab -- the very bytes this wave's own A pieces will fill next
            pp_epilogue<EPI, HK_DUO_PF>(p, acc, smem + rslot * STAGE * 2 + wave * 4096, aux, lane, wm, wn, tm * 256, tn * 128);
            zero_acc();
            ++ti;
            kleft = KT + 1;
            if (more) {
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // slab and aux reads done before they are refilled
                stage_aux(ti);
                stage();
                read_a(nslot, 0, xa0);
                read_w(nslot, 0);
                read_w(nslot, 1);
                read_a(nslot, 1, xa1);
            }
        }
        --kleft;
        rslot = nslot;
    }
#undef HK_DUO_MMA
}

// Multi-stage kernel for the problems too small for the persistent 256x256 kernel (Q-Former GEMMs, small
// batches): one BM x BN x 64 tile per workgroup, STAGES LDS buffers, LDS-DMA issued STAGES-1 K-tiles ahead
// and retired by a counted vmcnt (loads in flight across the one barrier per K-tile).  These shapes are
// latency-bound -- a K-tile's MFMAs take less time than a load round trip -- so prefetch depth, not tile
// size, is what matters.
template <int BM, int BN, int WM, int WN, int STAGES>
__global__ __launch_bounds__(WM* WN * 64) void hk_gemm_kernel(const GemmParams p) {
    constexpr int NW = WM * WN;
    constexpr int BK = 64;
    constexpr int TM = BM / WM, TN = BN / WN;
    constexpr int MI = TM / 16, NJ = TN / 16;
    constexpr int A_IT = BM / 8 / NW, B_IT = BN / 8 / NW;
    constexpr int LPT = A_IT + B_IT;       // LDS-DMA instructions per thread per K-tile
    constexpr int STAGE = (BM + BN) * BK;  // elements per stage
    static_assert(BM % (8 * NW) == 0 && BN % (8 * NW) == 0, "loader split");
    static_assert(STAGES >= 2 && STAGES <= 4 && (STAGES - 2) * LPT < 64, "vmcnt range");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    __bf16* lds = reinterpret_cast<__bf16*>(smem);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;

    // split-K: consecutive workgroups are the slices of ONE tile (they share its A / W panels in time and in L2)
    const int zs = p.ksplit > 1 ? (int)(blockIdx.x % (unsigned)p.ksplit) : 0;
    const int lid = xcd_remap(p.ksplit > 1 ? (int)(blockIdx.x / (unsigned)p.ksplit) : (int)blockIdx.x, p.nbm * p.nbn);
    const int m0 = (lid / p.nbn) * BM;
    const int n0 = (lid % p.nbn) * BN;
    const int kt0 = zs * p.kper;   // first K-tile of this slice

    // ---- loader state: each wave-instruction fills 8 rows x 128 B ----
    const int lrow = lane >> 3, lpc = lane & 7;
    const __bf16* aptr[A_IT];
    const __bf16* bptr[B_IT];
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        const int row = (i * NW + wave) * 8 + lrow;
        const int m = min(m0 + row, p.M - 1);
        aptr[i] = p.A + p.a.off(m + p.m_off) + ((lpc ^ ((row >> 1) & 7)) << 3) + kt0 * BK;
    }
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
        const int row = (i * NW + wave) * 8 + lrow;
        const int n = min(n0 + row, p.N - 1);
        bptr[i] = p.W + p.wrow(n) + ((lpc ^ ((row >> 1) & 7)) << 3) + (int64_t)kt0 * p.wk();
    }
    auto stage_load = [&](int stage, int kt) {
        __bf16* As = lds + stage * STAGE;
        __bf16* Bs = As + BM * BK;
        const int k0 = kt * BK, k0w = kt * p.wk();
#pragma unroll
        for (int i = 0; i < A_IT; ++i) glds16(aptr[i] + k0, As + (i * NW + wave) * 8 * BK);
#pragma unroll
        for (int i = 0; i < B_IT; ++i) glds16(bptr[i] + k0w, Bs + (i * NW + wave) * 8 * BK);
    };

    // ---- fragment read offsets (elements): row (lane & 15), logical chunk kk*4 + (lane >> 4) ----
    const int sw = (lane >> 1) & 7;
    const int foff0 = (lane & 15) * BK + ((((lane >> 4)) ^ sw) << 3);
    const int foff1 = (lane & 15) * BK + (((4 + (lane >> 4)) ^ sw) << 3);

    f32x4 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- epilogue operands (bias, column sums, row statistics, residual) requested BEFORE the K loop, raw, with clamped
    // indices: they are the oldest entries of the vmcnt queue, land with K-tile 0, and the epilogue of these
    // latency-bound launches no longer walks through one memory round trip per 16-row group.  Lane owns row
    // (lane & 15) of each 16-row tile and 4 consecutive columns.
    const int ncol = n0 + wn * TN + (lane >> 4) * 4;
    f32x4 bias[NJ], ccv[NJ];
    float rsv[MI], meanv[MI];
    u32x4 resraw[MI][NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int nc = min(ncol + j * 16, p.N - 4);
        bias[j] = (p.bias != nullptr && zs == 0) ? *reinterpret_cast<const f32x4*>(p.bias + nc) : f32x4{0.f, 0.f, 0.f, 0.f};
        ccv[j] = p.row_stats != nullptr ? *reinterpret_cast<const f32x4*>(p.col_c + nc) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int mc = min(m0 + wm * TM + i * 16 + (lane & 15), p.M - 1) + p.m_off;
        rsv[i] = 1.f;
        meanv[i] = 0.f;
        if (p.row_stats) {
            meanv[i] = p.row_stats[2 * (int64_t)mc];
            rsv[i] = p.row_stats[2 * (int64_t)mc + 1];
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) resraw[i][j] = u32x4{0u, 0u, 0u, 0u};
        if (p.res && zs == 0) {
            const int64_t roff = p.r.off(mc);
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int nc = min(ncol + j * 16, p.N - 4);
                if (p.res_bf16) {
                    const u32x2 w = *reinterpret_cast<const u32x2*>(reinterpret_cast<const __bf16*>(p.res) + roff + nc);
                    resraw[i][j] = u32x4{w[0], w[1], 0u, 0u};
                } else {
                    resraw[i][j] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const float*>(p.res) + roff + nc);
                }
            }
        }
    }

    const int KT = p.ksplit > 1 ? min(p.kper, p.K / BK - kt0) : p.K / BK;
#pragma unroll
    for (int s = 0; s < STAGES - 1; ++s)
        if (s < KT) stage_load(s, s);

    int slot = 0;
    for (int kt = 0; kt < KT; ++kt) {
        // K-tile kt has landed once at most min(STAGES-2, KT-1-kt) younger tiles are still in flight
        const int younger = min(STAGES - 2, KT - 1 - kt);
        if (younger >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LPT) : "memory");
        else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPT) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // everyone's part of tile kt landed; everyone is done reading tile kt-1
        if (kt + STAGES - 1 < KT) stage_load(slot == 0 ? STAGES - 1 : slot - 1, kt + STAGES - 1);
        const __bf16* As = lds + slot * STAGE + (wm * TM) * BK;
        const __bf16* Bs = lds + slot * STAGE + BM * BK + (wn * TN) * BK;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int fo = kk ? foff1 : foff0;
            bf16x8 xa[MI], wb[NJ];
#pragma unroll
            for (int j = 0; j < NJ; ++j) wb[j] = *reinterpret_cast<const bf16x8*>(Bs + j * 16 * BK + fo);
#pragma unroll
            for (int i = 0; i < MI; ++i) xa[i] = *reinterpret_cast<const bf16x8*>(As + i * 16 * BK + fo);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[j], xa[i], acc[i][j], 0, 0, 0);
        }
        slot = slot + 1 == STAGES ? 0 : slot + 1;
    }

    // ---- epilogue (operands requested before the K loop) ----
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int m = m0 + wm * TM + i * 16 + (lane & 15);
        if (m >= p.M) continue;
        const int64_t coff = p.c.off(m + p.m_off) + zs * p.split_stride;
        const float rs = rsv[i], rm = rsv[i] * meanv[i];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int n = ncol + j * 16;
            if (n < p.N) {
                f32x4 res = f32x4{0.f, 0.f, 0.f, 0.f};
                if (p.res) {
                    const u32x4 w = resraw[i][j];
                    res = p.res_bf16 ? f32x4{bf16_lo(w[0]), bf16_hi(w[0]), bf16_lo(w[1]), bf16_hi(w[1])} : __builtin_bit_cast(f32x4, w);
                }
                const f32x4 v = epi_store(p, acc[i][j], bias[j], ccv[j], res, n, coff, rs, rm);
                s1 += (v[0] + v[1]) + (v[2] + v[3]);
                s2 = __builtin_fmaf(v[0], v[0], __builtin_fmaf(v[1], v[1], __builtin_fmaf(v[2], v[2], __builtin_fmaf(v[3], v[3], s2))));
            }
        }
        if (TN == 64 && p.row_part) {
            // row sums of this wave's 64 columns: the row's 4 lane groups (lane >> 4) hold 16 columns each; rows past
            // M skipped above are skipped by all four of their lanes, so every lane read here is active
            s1 += __shfl_xor(s1, 16);
            s2 += __shfl_xor(s2, 16);
            s1 += __shfl_xor(s1, 32);
            s2 += __shfl_xor(s2, 32);
            if (lane < 16) {
                float* dst = p.row_part + ((int64_t)(m + p.m_off) * (p.N >> 6) + ((n0 + wn * TN) >> 6)) * 2;
                dst[0] = s1;
                dst[1] = s2;
            }
        }
    }
}

template <int BM, int BN, int WM, int WN, int STAGES>
int launch(const GemmParams& p, hipStream_t stream) {
    constexpr int LDS = STAGES * (BM + BN) * 64 * 2;
    static std::atomic<unsigned long long> attr_done{0};   // per device
    const hipError_t e = hk_lds_attr_once(attr_done, reinterpret_cast<const void*>(&hk_gemm_kernel<BM, BN, WM, WN, STAGES>), LDS);
    if (e != hipSuccess) {
        hk_set_error("hk_gemm_bf16: hipFuncSetAttribute(%d B LDS): %s", LDS, hipGetErrorString(e));
        return HK_ELAUNCH;
    }
    GemmParams q = p;
    q.nbm = (p.M + BM - 1) / BM;
    q.nbn = (p.N + BN - 1) / BN;
    hipLaunchKernelGGL((hk_gemm_kernel<BM, BN, WM, WN, STAGES>), dim3(q.nbm * q.nbn * (q.ksplit > 1 ? q.ksplit : 1)),
                       dim3(WM * WN * 64), LDS, stream, q);
    HK_CHECK_LAUNCH("hk_gemm_bf16");
    return HK_OK;
}

template <int EPI, int NI = 4>
int launch_pp(const GemmParams& p, hipStream_t stream) {
    constexpr int LDS = 2 * 4 * 128 * 64 * 2 + 8 * 4096;  // operand buffers + epilogue staging = 160 KB
    static std::atomic<unsigned long long> attr_done{0};   // per device
    {
        const hipError_t e = hk_lds_attr_once(attr_done, reinterpret_cast<const void*>(&hk_gemm_pp_kernel<EPI, NI>), LDS);
        if (e != hipSuccess) {
            hk_set_error("hk_gemm_bf16: hipFuncSetAttribute(%d B LDS): %s", LDS, hipGetErrorString(e));
            return HK_ELAUNCH;
        }
    }
    static int ncu = 0;  // CUs per device (one device model per process); racing first calls compute the same value
    if (ncu == 0) {
        hipError_t e = hipSuccess;
        int dev = 0, n = 0;
        if (e == hipSuccess) e = hipGetDevice(&dev);
        if (e == hipSuccess) e = hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        if (e != hipSuccess || n <= 0) {
            hk_set_error("hk_gemm_bf16: device setup (%d B LDS): %s", LDS, hipGetErrorString(e));
            return HK_ELAUNCH;
        }
        ncu = n;
    }
    GemmParams q = p;
    q.nbm = (p.M + 64 * NI - 1) / (64 * NI);
    q.nbn = (p.N + 255) / 256;
    const int ntiles = q.nbm * q.nbn;
    // HK_GEMM_VARIANT=2: one tile per workgroup (grid = tiles) instead of the persistent walk
    static const int variant = getenv("HK_GEMM_VARIANT") ? atoi(getenv("HK_GEMM_VARIANT")) : 1;
    const int grid = (variant == 2 || ntiles < ncu) ? ntiles : ncu;
#ifdef HK_DIAG
    // diagnostic build only (make diag -> libhawkeye_hip_diag.so): HK_GEMM_TRACE=<file> dumps per-workgroup timestamps of
    // the LAST launch.  This path allocates and synchronises, so it is compiled out of the product library.
    static const char* trace_path = getenv("HK_GEMM_TRACE");
    static unsigned long long* trace_buf = nullptr;
    if (trace_path) {
        if (!trace_buf && hipMalloc(&trace_buf, (size_t)16384 * 64 * 8) != hipSuccess) trace_buf = nullptr;
        if (trace_buf) (void)hipMemsetAsync(trace_buf, 0, (size_t)16384 * 64 * 8, stream);
        q.trace = grid <= 16384 ? trace_buf : nullptr;
    }
#endif
    hipLaunchKernelGGL((hk_gemm_pp_kernel<EPI, NI>), dim3(grid), dim3(512), LDS, stream, q);
    HK_CHECK_LAUNCH("hk_gemm_bf16");
#ifdef HK_DIAG
    if (trace_path && q.trace) {
        (void)hipStreamSynchronize(stream);
        unsigned long long* h = (unsigned long long*)malloc((size_t)grid * 64 * 8);
        (void)hipMemcpy(h, trace_buf, (size_t)grid * 64 * 8, hipMemcpyDeviceToHost);
        FILE* f = fopen(trace_path, "wb");
        if (f) {
            fwrite(h, 8, (size_t)grid * 64, f);
            fclose(f);
        }
        free(h);
    }
#endif
    return HK_OK;
}

std::atomic<int>& duo_mask() {
    static std::atomic<int> m{getenv("HK_GEMM_DUO") ? (int)strtol(getenv("HK_GEMM_DUO"), nullptr, 0) : HK_GEMM_DUO_DEFAULT};
    return m;
}

template <int EPI>
int launch_duo(const GemmParams& p, hipStream_t stream) {
    constexpr int LDS = 3 * (256 + 128) * 32 * 2 + 4 * 1536;  // 3 stages + epilogue operands = 78 KB: two workgroups per CU
    static std::atomic<unsigned long long> attr_done{0};      // per device
    {
        const hipError_t e = hk_lds_attr_once(attr_done, reinterpret_cast<const void*>(&hk_gemm_duo_kernel<EPI>), LDS);
        if (e != hipSuccess) {
            hk_set_error("hk_gemm_bf16: hipFuncSetAttribute(%d B LDS): %s", LDS, hipGetErrorString(e));
            return HK_ELAUNCH;
        }
    }
    static int ncu = 0;
    if (ncu == 0) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) {
            hk_set_error("hk_gemm_bf16: device setup (duo)");
            return HK_ELAUNCH;
        }
        ncu = n;
    }
    GemmParams q = p;
    q.nbm = (p.M + 255) / 256;
    q.nbn = (p.N + 127) / 128;
    const int ntiles = q.nbm * q.nbn;
    const int grid = ntiles < 2 * ncu ? ntiles : 2 * ncu;
    hipLaunchKernelGGL(hk_gemm_duo_kernel<EPI>, dim3(grid), dim3(256), LDS, stream, q);
    HK_CHECK_LAUNCH("hk_gemm_bf16");
    return HK_OK;
}

}  // namespace

extern "C" int hk_gemm_duo_mask(int mask) {
    return mask < 0 ? duo_mask().load(std::memory_order_relaxed) : duo_mask().exchange(mask & 31, std::memory_order_relaxed);
}

extern "C" int hk_gemm_bf16(const hk_gemm_desc* d, void* stream) {
    HK_REQUIRE(d != nullptr, "hk_gemm_bf16: null descriptor");
    HK_REQUIRE(d->A && d->W && d->C, "hk_gemm_bf16: null operand pointer");
    HK_REQUIRE(d->M > 0 && d->N > 0 && d->K > 0, "hk_gemm_bf16: empty problem M=%d N=%d K=%d", d->M, d->N, d->K);
    HK_REQUIRE(d->K % 64 == 0, "hk_gemm_bf16: K=%d must be a multiple of 64", d->K);
    HK_REQUIRE(d->N % 16 == 0, "hk_gemm_bf16: N=%d must be a multiple of 16", d->N);
    HK_REQUIRE(((uintptr_t)d->A | (uintptr_t)d->W | (uintptr_t)d->C | (uintptr_t)d->bias | (uintptr_t)d->residual) % 16 == 0,
               "hk_gemm_bf16: operands must be 16-byte aligned");
    HK_REQUIRE(d->a_rows.ld % 8 == 0 && d->ldw % 8 == 0 && d->a_rows.batch_stride % 8 == 0,
               "hk_gemm_bf16: A/W leading dimensions must be multiples of 8 elements");
    HK_REQUIRE(d->c_rows.ld % 4 == 0 && d->c_rows.batch_stride % 4 == 0 && d->r_rows.ld % 4 == 0 &&
                   d->r_rows.batch_stride % 4 == 0,
               "hk_gemm_bf16: C/residual leading dimensions must be multiples of 4 elements");
    HK_REQUIRE(d->a_rows.ld >= d->K && d->ldw >= d->K, "hk_gemm_bf16: lda/ldw smaller than K");
    HK_REQUIRE(d->act == HK_ACT_NONE || d->act == HK_ACT_GELU_ERF || d->act == HK_ACT_SWIGLU, "hk_gemm_bf16: unknown act %d", d->act);
    HK_REQUIRE(d->act != HK_ACT_SWIGLU || (d->out_dtype == HK_DT_BF16 && d->residual == nullptr && d->row_stats == nullptr &&
                                           d->row_part == nullptr && d->scale_cols == 0 && d->split_k <= 1 && d->N % 32 == 0 &&
                                           (uintptr_t)d->C % 16 == 0),
               "hk_gemm_bf16: HK_ACT_SWIGLU needs bf16 output [M, N / 2], N %% 32 == 0, no residual / folded LayerNorm / row_part / "
               "column scale / split_k");
    HK_REQUIRE(d->out_dtype == HK_DT_BF16 || d->out_dtype == HK_DT_F32, "hk_gemm_bf16: unknown out_dtype %d",
               d->out_dtype);
    HK_REQUIRE(d->scale_cols % 4 == 0, "hk_gemm_bf16: scale_cols must be a multiple of 4");
    HK_REQUIRE(d->res_dtype == HK_DT_BF16 || d->res_dtype == HK_DT_F32, "hk_gemm_bf16: unknown res_dtype %d",
               d->res_dtype);
    HK_REQUIRE(d->row_stats == nullptr || (d->col_c != nullptr && d->bias != nullptr && d->residual == nullptr &&
                                           d->out_dtype == HK_DT_BF16),
               "hk_gemm_bf16: row_stats (folded LayerNorm) needs col_c and bias, bf16 output and no residual");
    HK_REQUIRE(((uintptr_t)d->row_stats % 16 == 0) && ((uintptr_t)d->col_c % 16 == 0),
               "hk_gemm_bf16: row_stats / col_c misaligned");
    HK_REQUIRE(d->row_part == nullptr || (d->N % 128 == 0 && d->residual != nullptr && d->res_dtype == HK_DT_BF16 &&
                                          d->out_dtype == HK_DT_BF16 && (uintptr_t)d->row_part % 8 == 0),
               "hk_gemm_bf16: row_part needs N %% 128 == 0 and the bf16 residual-stream epilogue");

    HK_REQUIRE(d->split_k <= 1 || (d->out_dtype == HK_DT_F32 && d->act == HK_ACT_NONE && d->row_stats == nullptr &&
                                   d->row_part == nullptr && d->split_k <= d->K / 64 && d->split_k <= 64 &&
                                   d->split_stride % 4 == 0 &&
                                   (d->residual == nullptr || d->residual != d->C)),
               "hk_gemm_bf16: split_k needs fp32 output, no activation / folded LayerNorm / row_part, split_k <= K / 64, "
               "and a residual that is not the output buffer");
    if (d->split_k > 1) {
        // slice z lives at C + z * split_stride: a stride shorter than one slice's extent makes slices overwrite each other
        const int64_t c_last = d->c_rows.rows_per_batch == 0
                                   ? (int64_t)(d->M - 1) * d->c_rows.ld
                                   : (int64_t)((d->M - 1) / d->c_rows.rows_per_batch) * d->c_rows.batch_stride +
                                         (int64_t)((d->M - 1) % d->c_rows.rows_per_batch) * d->c_rows.ld;
        HK_REQUIRE(d->split_stride >= c_last + d->N,
                   "hk_gemm_bf16: split_stride=%lld is smaller than one slice's extent (%lld elements)",
                   (long long)d->split_stride, (long long)(c_last + d->N));
    }

    GemmParams p;
    p.A = reinterpret_cast<const __bf16*>(d->A);
    HK_REQUIRE(d->w_tiled == 0 || (d->w_tiled == 1 && d->N % 64 == 0 && d->K % 64 == 0),
               "hk_gemm_bf16: w_tiled needs N=%d and K=%d to be multiples of 64", d->N, d->K);
    p.W = reinterpret_cast<const __bf16*>(d->W);
    p.bias = d->bias;
    p.res = d->residual;
    p.C = d->C;
    p.row_stats = d->row_stats;
    p.col_c = d->col_c;
    p.row_part = d->row_part;
    p.M = d->M;
    p.N = d->N;
    p.K = d->K;
    p.a = RowMap{d->a_rows.ld, d->a_rows.rows_per_batch, d->a_rows.batch_stride};
    p.c = RowMap{d->c_rows.ld, d->c_rows.rows_per_batch, d->c_rows.batch_stride};
    p.r = RowMap{d->r_rows.ld, d->r_rows.rows_per_batch, d->r_rows.batch_stride};
    p.ldw = d->ldw;
    p.wt = d->w_tiled;
    p.wnkt = d->K / 64;
    p.act = d->act;
    p.out_f32 = d->out_dtype == HK_DT_F32;
    p.res_bf16 = d->res_dtype == HK_DT_BF16;
    p.scale_cols = d->scale_cols;
    p.scale = d->scale;
    p.nbm = p.nbn = 0;
    p.m_off = 0;
    p.ksplit = 1;
    p.kper = 0;
    p.split_stride = 0;
    p.trace = nullptr;
#ifdef HK_DIAG
    // diagnostic build only: HK_GEMM_DEBUG bit 1 = skip the epilogue, bit 2 = one K-tile per tile (wrong results by design)
    static const int dbg = getenv("HK_GEMM_DEBUG") ? atoi(getenv("HK_GEMM_DEBUG")) : 0;
    p.dbg = dbg;
#else
    p.dbg = 0;
#endif
    static const int gm = getenv("HK_GEMM_GM") ? atoi(getenv("HK_GEMM_GM")) : 4;
    p.gm = gm < 1 ? 1 : gm;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);

    // The persistent 256x256 kernel once its tiles fill at least half the chip; otherwise the multi-stage kernel, with the
    // narrower 128x64 tile when 128x128 tiles would leave most CUs idle.  HK_GEMM_VARIANT=0 forces the
    // multi-stage kernel everywhere (A/B aid).
    const long tiles256 = (long)((d->M + 255) / 256) * ((d->N + 255) / 256);
    static const int variant = getenv("HK_GEMM_VARIANT") ? atoi(getenv("HK_GEMM_VARIANT")) : 1;
    // HK_GEMM_MID=0 turns the 256x128-tile rule (below) off (A/B aid)
    static const int mid = getenv("HK_GEMM_MID") ? atoi(getenv("HK_GEMM_MID")) : 1;
    auto mid_window = [&](int M_) {
        const long tiles_mid = (long)((M_ + 255) / 256) * ((d->N + 127) / 128);
        static const int mid_lo = getenv("HK_GEMM_MID_LO") ? atoi(getenv("HK_GEMM_MID_LO")) : 144;   // see the rule below
        return mid != 0 && variant != 0 && tiles_mid <= 256 && tiles_mid >= mid_lo;
    };
    // HK_GEMM_192=0 turns the 192-row tiles off (A/B aid); 2 / 3: experiments (below)
    static const int r192 = getenv("HK_GEMM_192") ? atoi(getenv("HK_GEMM_192")) : 1;
    // rows a tile height pads M to: 192-row tiles waste less on skinny problems (M = 352, a B = 1 prompt: 384 rows instead of 512)
    auto padded = [](int M_, int bm) { return (M_ + bm - 1) / bm * bm; };
    static const int mid_stages = getenv("HK_GEMM_MID_STAGES") ? atoi(getenv("HK_GEMM_MID_STAGES")) : 3;
    auto small = [&](const GemmParams& q) {
        if (mid_window(q.M)) {   // also the tail rows of a split persistent launch
            const long t192 = (long)((q.M + 191) / 192) * ((d->N + 127) / 128);
            // (three stages, 120 KB: with weights that arrive cold from HBM -- the LLM's, read once per pass -- a two-deep ring
            // stalls on every K-tile: M = 352 qkv, cold weights, 70.4 -> 56.3 us; tools/cold_weights.py)
            if (r192 != 0 && padded(q.M, 192) < padded(q.M, 256) && t192 <= 256) return launch<192, 128, 4, 2, 3>(q, s);
            // three stages (144 KB) here too: whole calls, HK_GEMM_MID_STAGES=2 / 3 interleaved on one box -- B = 2 T = 16 9.10 -> 8.82 ms,
            // B = 3 T = 12 8.69 -> 8.47, B = 4 T = 8 8.11 -> 7.83; B = 1 calls and the rest within 0.3 %
            if (mid_stages == 3) return launch<256, 128, 4, 2, 3>(q, s);
            return launch<256, 128, 4, 2, 2>(q, s);
        }
        const long t128 = (long)((q.M + 127) / 128) * ((q.N + 127) / 128);
        // half a chip of 128x128 tiles or less that 128x64 tiles spread over (nearly) all CUs in one round -- the tail rows of a
        // split persistent launch (2048 x 1024 at the headline shape: 128 -> 256 workgroups): four waves of 32 x 64, which
        // keeps the 64-column row_part granule.  HK_GEMM_TAIL64=0: off (A/B aid)
        static const int tail64 = getenv("HK_GEMM_TAIL64") ? atoi(getenv("HK_GEMM_TAIL64")) : 1;
        const long t64 = (long)((q.M + 127) / 128) * ((q.N + 63) / 64);
        if (tail64 != 0 && q.row_part && t128 <= 128 && t64 > 160 && t64 <= 256) return launch<128, 64, 4, 1, 4>(q, s);
        if (q.row_part && t128 < 384) return launch<128, 128, 2, 2, 4>(q, s);  // 64 columns per wave: the row_part granule
        if (t128 >= 384) return launch<128, 128, 2, 2, 2>(q, s);  // >= 1.5 workgroups per CU: 2 stages, 2 workgroups per CU
        if (t128 >= 128) return launch<128, 128, 2, 2, 4>(q, s);
        return launch<128, 64, 2, 2, 4>(q, s);
    };
    // from well under half a chip's worth of 256x256 tiles up (one tile per CU, the rest idle) the ping-pong kernel still
    // beats the multi-stage one: prefill o_proj / down_proj at M = 2816, 176 tiles: 42.1 -> 39.3 ms; B = 1 qkv at
    // M = 2352, 120 tiles: 36.3 -> 30.0 us (a B = 1, T = 12 call 5.88 -> 5.69 ms, same box); at 40 tiles (fc2, M = 2352)
    // it loses badly (46 -> 81 us).  tools/small_gemm.py, HK_GEMM_PP_MIN=<tiles> to move the crossover.
    static const int pp_min = getenv("HK_GEMM_PP_MIN") ? atoi(getenv("HK_GEMM_PP_MIN")) : 96;
    // the persistent kernel addresses its operand pieces as scalar base + 32-bit byte offset
    const int64_t a_last = d->a_rows.rows_per_batch == 0
                               ? (int64_t)(d->M - 1) * d->a_rows.ld
                               : (int64_t)((d->M - 1) / d->a_rows.rows_per_batch) * d->a_rows.batch_stride +
                                     (int64_t)((d->M - 1) % d->a_rows.rows_per_batch) * d->a_rows.ld;
    const bool off32 = (a_last + d->K) * 2 < (1LL << 32) && (int64_t)d->N * d->ldw * 2 < (1LL << 32);
    if (d->split_k > 1) {
        // few tiles, long K: slices of the K range on separate workgroups, partial products summed by the consumer
        const int KT = d->K / 64;
        p.kper = (KT + d->split_k - 1) / d->split_k;
        p.ksplit = (KT + p.kper - 1) / p.kper;          // no empty slice
        p.split_stride = d->split_stride;
        if (p.ksplit < d->split_k) {
            hk_set_error("hk_gemm_bf16: split_k=%d leaves empty slices for K=%d (use a divisor of K / 64 = %d)", d->split_k, d->K, KT);
            return HK_EINVAL;
        }
        // kernel by the number of workgroups the slices make, like the unsplit choice below
        const long t128 = (long)((d->M + 127) / 128) * ((d->N + 127) / 128) * p.ksplit;
        // one round of 256x128 tiles (160-256 workgroups) first, as for the unsplit launches below: fc2 at M = 2352 in 3
        // slices 50.4 -> 37.4 us with the consumer's sum (unsplit 45.6 + 4.7), M = 3136 in 2 slices 60.1 -> 45.0 (46.9 + 4.7),
        // M = 1568 in 4 slices 40.3 -> 30.1 (2 slices of 128x128 tiles: 33.0); HK_GEMM_MID=0 turns it off
        const long tmid = (long)((d->M + 255) / 256) * ((d->N + 127) / 128) * p.ksplit;
        if (mid != 0 && tmid >= 160 && tmid <= 256) {
            const long t192 = (long)((d->M + 191) / 192) * ((d->N + 127) / 128) * p.ksplit;
            if (r192 != 0 && padded(d->M, 192) < padded(d->M, 256) && t192 <= 256) return launch<192, 128, 4, 2, 3>(p, s);
            if (mid_stages == 3) return launch<256, 128, 4, 2, 3>(p, s);
            return launch<256, 128, 4, 2, 2>(p, s);
        }
        if (t128 >= 384) return launch<128, 128, 2, 2, 2>(p, s);
        if (t128 >= 128) return launch<128, 128, 2, 2, 4>(p, s);
        return launch<128, 64, 2, 2, 4>(p, s);
    }
    // Between the two: when 256x128 tiles make ONE round that fills 9/16 of the chip or more (144-256 tiles; B = 1 calls at
    // T = 7-12, B = 2-5 at T = 8-16 for the N = 1024 layers), the multi-stage kernel at that tile beats both the half-empty
    // round of 256x256 tiles and two rounds of 128x128 ones -- tools/small_gemm.py, us per launch: M = 2352 qkv (240 tiles)
    // 29.5 -> 25.2; M = 1568 qkv (168) 33.7 -> 23.6, fc1 (224) 28.6 -> 23.9; it loses below (M = 784 qkv, 96 tiles: 18.6 -> 21.8;
    // proj / fc2, 56-80 tiles: 17.7 -> 22.6, 45 -> 60; whole calls with the bound at 128: B = 1 T = 4 / 5 +5 %) and above one
    // round (M = 2352 fc1, 320 tiles: 29 -> 42).  3 stages or 128x256 measured the same within 2 %.  Whole calls, rule off /
    // on: B = 2 T = 16 10.58 -> 9.48 ms, B = 4 T = 8 9.43 -> 8.35, B = 3 T = 12 10.05 -> 9.01; lower bound 160 -> 144: B = 2 T = 12
    // 8.14 -> 7.55, B = 1 T = 7 4.53 -> 4.30.  Two or three rounds of these tiles for the N = 1024 layers (257-768 tiles, in place
    // of a persistent round plus its tail) lose: B = 4 T = 12 11.47 -> 12.2 ms, B = 6 T = 12 14.86 -> 15.9, B = 8 T = 12 20.3 -> 20.95.
    const long tiles192 = (long)((d->M + 191) / 192) * ((d->N + 255) / 256);
    const int epi = p.act == HK_ACT_SWIGLU ? 5 : p.row_stats ? 3 : (!p.out_f32 && !p.res) ? 0 : (!p.out_f32 && p.res_bf16) ? (p.row_part ? 4 : 2) : 1;
    // HK_GEMM_192=2 (experiments): the 192-row tile for every launch it covers in one round, before the other small-shape rules
    if (r192 == 2 && variant != 0 && d->N % 256 == 0 && off32 && tiles192 <= 256 && (epi == 0 || epi == 1 || epi == 5))
        return epi == 0 ? launch_pp<0, 3>(p, s) : epi == 1 ? launch_pp<1, 3>(p, s) : launch_pp<5, 3>(p, s);
    // HK_GEMM_192=4 / 6 / 7 (experiments): one three-stage multi-stage tile for every launch -- 192x128, 192x192, 256x128
    if (r192 == 4 && d->split_k <= 1) return launch<192, 128, 4, 2, 3>(p, s);
    if (r192 == 6 && d->split_k <= 1) return launch<192, 192, 4, 2, 3>(p, s);
    if (r192 == 7 && d->split_k <= 1) return launch<256, 128, 4, 2, 3>(p, s);
    if (mid_window(d->M)) return small(p);
    if (variant != 0 && tiles256 >= pp_min && d->N % 256 == 0 && off32) {
        // Tail balance: the persistent kernel gives each of the 256 CUs floor or ceil(tiles / 256) tiles, so a
        // last round that is less than half full (e.g. 1568 tiles = 6.125 rounds) idles most of the chip for a
        // whole tile time.  Then the rows of that partial round go to the multi-stage kernel as a second
        // launch (its own tile raster spreads them over all CUs) and the persistent kernel runs whole rounds.
        static const int split = getenv("HK_GEMM_SPLIT") ? atoi(getenv("HK_GEMM_SPLIT")) : 1;
        static int ncu = 0;  // idempotent
        if (ncu == 0) {
            int dev = 0, n = 0;
            if (hipGetDevice(&dev) != hipSuccess ||
                hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
                n = 256;
            ncu = n;
        }
        const int nbn = d->N / 256, nbm = (d->M + 255) / 256;
        const long rem = tiles256 % ncu;
        GemmParams tail = p;
        tail.M = 0;
        // HK_GEMM_SPLIT_MAX: largest partial round (in tiles) that still goes to the second launch (default: under half a chip)
        static const int split_max = getenv("HK_GEMM_SPLIT_MAX") ? atoi(getenv("HK_GEMM_SPLIT_MAX")) : ncu / 2 - 1;
        // HK_GEMM_SPLIT_MIN=<rounds>: fewest whole rounds a launch needs before its partial round may go to the second launch.
        // One since round 3 (was two): proj / fc2 at B = 6, T = 16 are 296 tiles = one round plus 40 tiles, i.e. two rounds at 58 % --
        // whole calls, HK_GEMM_SPLIT_MIN=2 / 1 interleaved on one box: B = 6 T = 16 19.97 -> 19.30 ms, B = 7 T = 16 21.96 -> 21.44,
        // B = 7 T = 12 17.46 -> 16.78, B = 8 T = 12 18.85 -> 18.30; B = 5 / 6 T = 12, B = 10 T = 8 within +-0.5 %
        static const int split_min = getenv("HK_GEMM_SPLIT_MIN") ? atoi(getenv("HK_GEMM_SPLIT_MIN")) : 1;
        if (split && rem != 0 && rem <= split_max && tiles256 >= (long)split_min * ncu) {
            const int rows_main = (int)((tiles256 - rem) / nbn);   // whole tile rows inside the full rounds
            if (rows_main > 0 && rows_main < nbm) {
                tail.m_off = rows_main * 256;
                tail.M = d->M - tail.m_off;
                p.M = tail.m_off;
            }
        }
        // HK_GEMM_DUO: bit e set = epilogue kind e runs on the two-workgroups-per-CU 256x128 kernel (hk_gemm_duo_kernel)
        // instead of the 256x256 one (A/B switch; the default is what measured faster per epilogue kind)
        const int duo = duo_mask().load(std::memory_order_relaxed);
        const bool use_duo = epi != 5 && p.wt == 0 && ((duo >> epi) & 1) != 0 && p.K >= 128 && d->N % 128 == 0 && d->a_rows.rows_per_batch == 0 &&
                             p.M >= 256;
        if (use_duo && p.M % 256 != 0) {
            // the duo kernel addresses whole 256-row tiles through scalar bases: leftover rows join (or become) the tail launch
            const int whole = p.M / 256 * 256;
            tail.m_off = whole;
            tail.M = d->M - whole;
            p.M = whole;
        }
        // One partial round of 256-row tiles (tiles256 < CUs) that 192-row tiles still cover in one round: the round then
        // takes 3/4 of the time on more of the chip (prefill o_proj / down_proj at M = 2816: 176 -> 240 workgroups).
        // HK_GEMM_192=0 turns the rule off (A/B aid).  Instantiated for the epilogue kinds that meet such shapes.
        // Weight matrices of 64 MB and more (the LLM's qkv / gate | up / down projections) cannot stay in the 32 MB of L2 and are read
        // once per pass: they arrive COLD from HBM, and the persistent kernel's two-K-tile ring (all 160 KB of LDS) stalls on them --
        // +7...+22 % against a re-used buffer (tools/cold_weights.py).  While the problem is a few rounds, 192 x 192 tiles of the
        // multi-stage kernel with THREE stages win whenever they need fewer (shorter) rounds: a round of them measured 72-81 us
        // cold against 90-101 us for a round of the persistent kernel, hence the factor.  M = 352 gate | up 95.9 -> 81.2 us,
        // M = 704 qkv 95.4 -> 83.7, gate | up 181.6 -> 153.9, M = 1408 qkv 181.0 -> 153.2; M = 1056 / 2112 / 2816 keep the persistent
        // kernel (fewer rounds there).  With weights that stay in cache the persistent kernel wins (74 vs 80 us): ViT / Q-Former
        // weights are at most 8 MB.
        const long t192sq = (long)((d->M + 191) / 192) * ((d->N + 191) / 192);
        const bool cold_w = (int64_t)d->N * d->K * 2 >= (64LL << 20);
        if (r192 != 0 && tail.M == 0 && !use_duo && cold_w && (epi == 0 || epi == 5) &&
            0.76 * (double)((t192sq + ncu - 1) / ncu) < (double)((tiles256 + ncu - 1) / ncu) - 0.05)
            return launch<192, 192, 4, 2, 3>(p, s);
        // ... or cover it with as many workgroups and fewer padded rows (M = 352, gate | up: 172 workgroups either way, 384 rows)
        if (r192 != 0 && tail.M == 0 && !use_duo && tiles256 < ncu && tiles192 <= ncu && (epi == 0 || epi == 1 || epi == 5 || epi == 3) &&
            (tiles192 > tiles256 || (tiles192 == tiles256 && padded(d->M, 192) < padded(d->M, 256)))) {
            return epi == 0 ? launch_pp<0, 3>(p, s) : epi == 1 ? launch_pp<1, 3>(p, s) : epi == 3 ? launch_pp<3, 3>(p, s) : launch_pp<5, 3>(p, s);
        }
        // ... or a tile count just past a whole number of rounds (M = 704, gate | up: 258 tiles of 256 rows = two rounds for 2 tiles;
        // 344 tiles of 192 rows are two SHORTER rounds): rounds x relative tile time, a 192-row round measured at ~0.8 of a 256-row one
        if (r192 != 0 && tail.M == 0 && !use_duo && (epi == 0 || epi == 1 || epi == 5) && tiles256 < 2 * ncu &&
            0.8 * (double)((tiles192 + ncu - 1) / ncu) < (double)((tiles256 + ncu - 1) / ncu) - 0.05) {
            return epi == 0 ? launch_pp<0, 3>(p, s) : epi == 1 ? launch_pp<1, 3>(p, s) : launch_pp<5, 3>(p, s);
        }
        int rc;
        if (epi == 3) rc = use_duo ? launch_duo<3>(p, s) : launch_pp<3>(p, s);
        else if (epi == 0) rc = use_duo ? launch_duo<0>(p, s) : launch_pp<0>(p, s);
        else if (epi == 4) rc = use_duo ? launch_duo<4>(p, s) : launch_pp<4>(p, s);
        else if (epi == 2) rc = use_duo ? launch_duo<2>(p, s) : launch_pp<2>(p, s);
        else if (epi == 5) rc = launch_pp<5>(p, s);
        else rc = use_duo ? launch_duo<1>(p, s) : launch_pp<1>(p, s);
        if (rc != HK_OK || tail.M == 0) return rc;
        return small(tail);
    }
    return small(p);
}
