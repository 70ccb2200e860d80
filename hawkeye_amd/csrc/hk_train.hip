// Row f2: the training side of the Q-Former / llama_proj (stage-3 instruction tuning, tasks/train_it.py:63-80).
// The ViT is frozen (hawkeye_it.py:82-91) and keeps the inference kernels; what trains is the Q-Former, the query
// tokens and llama_proj.  This file holds what their forward needs beyond inference (dropout, drop-path, saved
// statistics) and the backward kernels.  Dense backward GEMMs (dgrad, wgrad) reuse hk_gemm_bf16 on transposed
// operands made by hk_transpose_bf16.
//
// Dropout / drop-path randomness is a counter-based hash, not a stateful generator: element `idx` of dropout
// site `site` is kept iff fmix32(idx ^ key(seed, site)) >= p * 2^32.  The backward regenerates the mask from the
// same (seed, site, idx), so no mask is ever stored.
#include "hk_common.h"
#include <mutex>

namespace {

__host__ __device__ __forceinline__ uint32_t fmix32(uint32_t h) {
    h ^= h >> 16;
    h *= 0x85EBCA6Bu;
    h ^= h >> 13;
    h *= 0xC2B2AE35u;
    h ^= h >> 16;
    return h;
}
inline uint32_t site_key(uint32_t seed, uint32_t site) { return fmix32(seed ^ fmix32(site * 0x9E3779B9u + 0x7F4A7C15u)); }
inline uint32_t drop_thresh(float p) {
    const double t = (double)p * 4294967296.0;
    return t <= 0.0 ? 0u : (t >= 4294967295.0 ? 4294967295u : (uint32_t)t);
}
__device__ __forceinline__ bool keep_at(uint32_t idx, uint32_t key, uint32_t thresh) { return fmix32(idx ^ key) >= thresh; }

__device__ __forceinline__ float wave_sum(float x) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
    return x;
}

__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// ------------------------------------------------------------------------------------------------------------
// Attention with dropout, head size 64.  One 16x16x32 MFMA tile layout is used everywhere:
//   A operand: lane l holds row (l & 15), contraction elements 8*(l >> 4) .. +8
//   B operand: lane l holds column (l & 15), same contraction elements
//   C/D:       lane l holds column (l & 15), rows 4*(l >> 4) .. +4
// so X[m][k] . Y[n][k]^T needs both operands contraction-contiguous; tiles that are contracted over their row
// index are staged transposed in LDS.
// ------------------------------------------------------------------------------------------------------------
constexpr int TS = 72;   // LDS row stride of a 64 x 64 bf16 tile (144 B: 16-byte aligned, breaks the 128 B bank period)

struct AttnT {
    const __bf16 *Q, *K, *V, *dO;
    __bf16 *O, *dQ, *dK, *dV;
    float *lse, *dsum, *O32;
    const float* kbias;
    int B, H, Nq, Nk;
    int64_t ldq, ldk, ldv, ldo, bsq, bsk, bsv, bso;
    int64_t lddq, lddk, lddv, bsdq, bsdk, bsdv;
    float sm_scale, inv_keep;
    uint32_t key, thresh;
};

// 64 x 64 bf16 tile, rows r0.. of a [rows, ld] matrix (rows >= limit read as zero) -> dst[r][c] and/or dstT[c][r]
__device__ __forceinline__ void load_tile(const __bf16* src, int64_t ld, int r0, int limit, __bf16 (*dst)[TS],
                                          __bf16 (*dstT)[TS], int tid) {
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int ch = tid + it * 256;
        const int r = ch >> 3, c = (ch & 7) * 8;
        bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
        if (r0 + r < limit) v = *reinterpret_cast<const bf16x8*>(src + (int64_t)(r0 + r) * ld + c);
        if (dst) *reinterpret_cast<bf16x8*>(&dst[r][c]) = v;
        if (dstT) {
#pragma unroll
            for (int e = 0; e < 8; ++e) dstT[c + e][r] = v[e];
        }
    }
}

__device__ __forceinline__ bf16x8 frag(const __bf16 (*t)[TS], int row, int k0) {
    return *reinterpret_cast<const bf16x8*>(&t[row][k0]);
}

// forward: O = dropout(softmax(Q K^T + bias)) V, also the log2-domain logsumexp of every query row
__global__ __launch_bounds__(256) void attn_train_fwd_kernel(AttnT p) {
    __shared__ __attribute__((aligned(16))) __bf16 sQ[64][TS];
    __shared__ __attribute__((aligned(16))) __bf16 sK[64][TS];
    __shared__ __attribute__((aligned(16))) __bf16 sVt[64][TS];
    __shared__ __attribute__((aligned(16))) __bf16 sP[4][16][TS];
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, lr = lane & 15, lg = lane >> 4;
    const int q0 = blockIdx.x * 64, h = blockIdx.y, b = blockIdx.z;
    const __bf16* Qb = p.Q + (int64_t)b * p.bsq + h * 64;
    const __bf16* Kb = p.K + (int64_t)b * p.bsk + h * 64;
    const __bf16* Vb = p.V + (int64_t)b * p.bsv + h * 64;
    load_tile(Qb, p.ldq, q0, p.Nq, sQ, nullptr, tid);
    __syncthreads();
    bf16x8 aq[2];
    aq[0] = frag(sQ, 16 * w + lr, 8 * lg);
    aq[1] = frag(sQ, 16 * w + lr, 8 * lg + 32);
    const int nkt = (p.Nk + 63) / 64;
    float m[4], s[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) m[r] = -INFINITY, s[r] = 0.f;
    for (int kt = 0; kt < nkt; ++kt) {
        __syncthreads();
        load_tile(Kb, p.ldk, kt * 64, p.Nk, sK, nullptr, tid);
        __syncthreads();
        f32x4 acc[4];
        float tmax[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {
            acc[cb] = f32x4{0, 0, 0, 0};
            acc[cb] = mfma16(aq[0], frag(sK, 16 * cb + lr, 8 * lg), acc[cb]);
            acc[cb] = mfma16(aq[1], frag(sK, 16 * cb + lr, 8 * lg + 32), acc[cb]);
            const int key = kt * 64 + 16 * cb + lr;
            const float kb = key < p.Nk ? (p.kbias ? p.kbias[(int64_t)b * p.Nk + key] * HK_LOG2E : 0.f) : -INFINITY;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                acc[cb][r] += kb;
                tmax[r] = fmaxf(tmax[r], acc[cb][r]);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) tmax[r] = fmaxf(tmax[r], __shfl_xor(tmax[r], o));
            const float mn = fmaxf(m[r], tmax[r]);
            float ts = 0.f;
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) ts += exp2f(acc[cb][r] - mn);
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) ts += __shfl_xor(ts, o);
            s[r] = s[r] * exp2f(m[r] - mn) + ts;
            m[r] = mn;
        }
    }
    float lse2[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        lse2[r] = m[r] + log2f(s[r]);
        const int q = q0 + 16 * w + 4 * lg + r;
        if (lr == 0 && q < p.Nq) p.lse[((int64_t)b * p.H + h) * p.Nq + q] = lse2[r];
    }
    f32x4 o[4];
#pragma unroll
    for (int db = 0; db < 4; ++db) o[db] = f32x4{0, 0, 0, 0};
    for (int kt = 0; kt < nkt; ++kt) {
        __syncthreads();
        load_tile(Kb, p.ldk, kt * 64, p.Nk, sK, nullptr, tid);
        load_tile(Vb, p.ldv, kt * 64, p.Nk, nullptr, sVt, tid);
        __syncthreads();
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {
            f32x4 acc = f32x4{0, 0, 0, 0};
            acc = mfma16(aq[0], frag(sK, 16 * cb + lr, 8 * lg), acc);
            acc = mfma16(aq[1], frag(sK, 16 * cb + lr, 8 * lg + 32), acc);
            const int key = kt * 64 + 16 * cb + lr;
            const float kb = key < p.Nk ? (p.kbias ? p.kbias[(int64_t)b * p.Nk + key] * HK_LOG2E : 0.f) : -INFINITY;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int q = q0 + 16 * w + 4 * lg + r;
                float pr = exp2f(acc[r] + kb - lse2[r]);
                if (p.thresh) {
                    const uint32_t idx = (uint32_t)(((int64_t)b * p.H + h) * p.Nq + q) * (uint32_t)p.Nk + (uint32_t)key;
                    pr = keep_at(idx, p.key, p.thresh) ? pr * p.inv_keep : 0.f;
                }
                sP[w][4 * lg + r][16 * cb + lr] = (__bf16)pr;
            }
        }
        __syncthreads();
        const bf16x8 ap0 = *reinterpret_cast<const bf16x8*>(&sP[w][lr][8 * lg]);
        const bf16x8 ap1 = *reinterpret_cast<const bf16x8*>(&sP[w][lr][8 * lg + 32]);
#pragma unroll
        for (int db = 0; db < 4; ++db) {
            o[db] = mfma16(ap0, frag(sVt, 16 * db + lr, 8 * lg), o[db]);
            o[db] = mfma16(ap1, frag(sVt, 16 * db + lr, 8 * lg + 32), o[db]);
        }
    }
    __bf16* Ob = p.O + (int64_t)b * p.bso + h * 64;
    // fp32 copy of O ([B, Nq, H*64] dense): the backward's dsum = sum_d dO.O must use the unrounded row, otherwise the
    // rounding of O shifts every dS of the row by P * delta -- a systematic error where attention is diffuse
    float* O32b = p.O32 ? p.O32 + ((int64_t)b * p.Nq * p.H + h) * 64 : nullptr;
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int q = q0 + 16 * w + 4 * lg + r;
            if (q < p.Nq) {
                Ob[(int64_t)q * p.ldo + 16 * db + lr] = (__bf16)o[db][r];
                if (O32b) O32b[(int64_t)q * p.H * 64 + 16 * db + lr] = o[db][r];
            }
        }
}

// backward, query side: dsum[q] = sum_d dO.O, dQ = softmax_scale * dS K
__global__ __launch_bounds__(256) void attn_train_bwd_q_kernel(AttnT p) {
    __shared__ __attribute__((aligned(16))) __bf16 sA[64][TS];     // Q tile, then reused for dO and O
    __shared__ __attribute__((aligned(16))) __bf16 sK[64][TS];
    __shared__ __attribute__((aligned(16))) __bf16 sKt[64][TS];
    __shared__ __attribute__((aligned(16))) __bf16 sV[64][TS];
    __shared__ __attribute__((aligned(16))) __bf16 sP[4][16][TS];
    __shared__ float sD[4][16];
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, lr = lane & 15, lg = lane >> 4;
    const int q0 = blockIdx.x * 64, h = blockIdx.y, b = blockIdx.z;
    const __bf16* Kb = p.K + (int64_t)b * p.bsk + h * 64;
    const __bf16* Vb = p.V + (int64_t)b * p.bsv + h * 64;
    bf16x8 aq[2], ado[2], ao[2];
    load_tile(p.Q + (int64_t)b * p.bsq + h * 64, p.ldq, q0, p.Nq, sA, nullptr, tid);
    __syncthreads();
    aq[0] = frag(sA, 16 * w + lr, 8 * lg);
    aq[1] = frag(sA, 16 * w + lr, 8 * lg + 32);
    __syncthreads();
    load_tile(p.dO + (int64_t)b * p.bso + h * 64, p.ldo, q0, p.Nq, sA, nullptr, tid);
    __syncthreads();
    ado[0] = frag(sA, 16 * w + lr, 8 * lg);
    ado[1] = frag(sA, 16 * w + lr, 8 * lg + 32);
    __syncthreads();
    load_tile(p.O + (int64_t)b * p.bso + h * 64, p.ldo, q0, p.Nq, sA, nullptr, tid);
    __syncthreads();
    ao[0] = frag(sA, 16 * w + lr, 8 * lg);
    ao[1] = frag(sA, 16 * w + lr, 8 * lg + 32);
    float dpart = 0.f;
    if (p.O32) {
        const int q = q0 + 16 * w + lr;
        if (q < p.Nq) {
            const float* orow = p.O32 + (((int64_t)b * p.Nq + q) * p.H + h) * 64;
#pragma unroll
            for (int e = 0; e < 8; ++e)
                dpart += (float)ado[0][e] * orow[8 * lg + e] + (float)ado[1][e] * orow[8 * lg + 32 + e];
        }
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) dpart += (float)ado[0][e] * (float)ao[0][e] + (float)ado[1][e] * (float)ao[1][e];
    }
    dpart += __shfl_xor(dpart, 16);
    dpart += __shfl_xor(dpart, 32);
    if (lg == 0) {
        sD[w][lr] = dpart;
        const int q = q0 + 16 * w + lr;
        if (q < p.Nq) p.dsum[((int64_t)b * p.H + h) * p.Nq + q] = dpart;
    }
    __syncthreads();
    float lse2[4], dsm[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int q = q0 + 16 * w + 4 * lg + r;
        lse2[r] = q < p.Nq ? p.lse[((int64_t)b * p.H + h) * p.Nq + q] : 0.f;
        dsm[r] = sD[w][4 * lg + r];
    }
    f32x4 dq[4];
#pragma unroll
    for (int db = 0; db < 4; ++db) dq[db] = f32x4{0, 0, 0, 0};
    const int nkt = (p.Nk + 63) / 64;
    for (int kt = 0; kt < nkt; ++kt) {
        __syncthreads();
        load_tile(Kb, p.ldk, kt * 64, p.Nk, sK, sKt, tid);
        load_tile(Vb, p.ldv, kt * 64, p.Nk, sV, nullptr, tid);
        __syncthreads();
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {
            f32x4 acc = f32x4{0, 0, 0, 0}, dp = f32x4{0, 0, 0, 0};
            acc = mfma16(aq[0], frag(sK, 16 * cb + lr, 8 * lg), acc);
            acc = mfma16(aq[1], frag(sK, 16 * cb + lr, 8 * lg + 32), acc);
            dp = mfma16(ado[0], frag(sV, 16 * cb + lr, 8 * lg), dp);
            dp = mfma16(ado[1], frag(sV, 16 * cb + lr, 8 * lg + 32), dp);
            const int key = kt * 64 + 16 * cb + lr;
            const float kb = key < p.Nk ? (p.kbias ? p.kbias[(int64_t)b * p.Nk + key] * HK_LOG2E : 0.f) : -INFINITY;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int q = q0 + 16 * w + 4 * lg + r;
                const float pr = exp2f(acc[r] + kb - lse2[r]);
                float g = dp[r];
                if (p.thresh) {
                    const uint32_t idx = (uint32_t)(((int64_t)b * p.H + h) * p.Nq + q) * (uint32_t)p.Nk + (uint32_t)key;
                    g = keep_at(idx, p.key, p.thresh) ? g * p.inv_keep : 0.f;
                }
                sP[w][4 * lg + r][16 * cb + lr] = (__bf16)(pr * (g - dsm[r]));
            }
        }
        __syncthreads();
        const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(&sP[w][lr][8 * lg]);
        const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(&sP[w][lr][8 * lg + 32]);
#pragma unroll
        for (int db = 0; db < 4; ++db) {
            dq[db] = mfma16(a0, frag(sKt, 16 * db + lr, 8 * lg), dq[db]);
            dq[db] = mfma16(a1, frag(sKt, 16 * db + lr, 8 * lg + 32), dq[db]);
        }
    }
    __bf16* dQb = p.dQ + (int64_t)b * p.bsdq + h * 64;
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int q = q0 + 16 * w + 4 * lg + r;
            if (q < p.Nq) dQb[(int64_t)q * p.lddq + 16 * db + lr] = (__bf16)(dq[db][r] * p.sm_scale);
        }
}

// ------------------------------------------------------------------------------------------------------------
// Round 3: key-split forms of the forward and of the query-side backward.  The kernels above give a workgroup 64 queries
// and walk ALL keys in 64-key tiles with two workgroup barriers and a synchronous global load per tile: at the stage-3
// shapes (B = 2: 48 workgroups; cross-attention 37 tiles, twice) they are one long latency chain on a fifth of the chip
// -- 3.4 ms of a 14 ms step.  Here a workgroup owns 16 queries and its four waves take the key tiles round-robin
// (tile kt on wave kt % 4), each wave on LDS of its own, so the loop has no workgroup barrier and the next tile's loads are
// in flight while a tile is computed; the waves' results meet once per pass through LDS, in wave order (bit-reproducible).
// Everything is computed TRANSPOSED, as in the inference kernel:
//     S^T = K Q^T : A = K rows (key 16 cb + lr), B = Q row of the lane's query -- both contraction-contiguous in global
//                   memory, loaded straight into fragments (no LDS);  C: lane = query lr, keys 16 cb + 4 lg + r
//     O^T = V^T P^T, dQ^T = K^T dS^T : B = the S^T-shaped tile as it lies in the accumulators (contraction element e of
//                   lane group lg = key 32 step + 16 (e >> 2) + 4 lg + (e & 3)); A = V / K tile staged ROW-MAJOR in the
//                   wave's LDS and read with ds_read_b64_tr_b16 at exactly those rows -- no transposed copy, no P tile
//     C: lane = query lr, d = 16 db + 4 lg + r: 8-byte bf16 / 16-byte fp32 stores of one row.
// Dropout indices and the saved statistic are those of the kernels above (the key-side backward regenerates the same mask).
// ------------------------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
struct alignas(16) WaveTile { __bf16 t[64][TS]; };

// the wave's 8 key fragments of tile kt: rows past Nk read row Nk - 1 (finite; their scores get bias -inf)
__device__ __forceinline__ void load_rows_frag(const __bf16* base, int64_t ld, int kt, int Nk, int lr, int lg, bf16x8 (&f)[4][2]) {
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
        const __bf16* r = base + (int64_t)min(kt * 64 + 16 * cb + lr, Nk - 1) * ld + 8 * lg;
        f[cb][0] = *reinterpret_cast<const bf16x8*>(r);
        f[cb][1] = *reinterpret_cast<const bf16x8*>(r + 32);
    }
}
// A operand of a "tile^T . X" product for contraction step `step` (32 keys) and 16-column block db, from a row-major tile
__device__ __forceinline__ bf16x8 frag_tr(const WaveTile& t, int step, int db, int lane) {
    const int lg = lane >> 4, qq = (lane >> 2) & 3, pp = lane & 3;
    const __bf16* a = &t.t[32 * step + 4 * lg + qq][16 * db + 4 * pp];
    const bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4_t*)(a));
    const bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4_t*)(a + 16 * TS));
    return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}
__device__ __forceinline__ void key_bias4(const AttnT& p, int b, int kt, int lg, float (&kb)[4][4]) {
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int key = kt * 64 + 16 * cb + 4 * lg + r;
            kb[cb][r] = key < p.Nk ? (p.kbias ? p.kbias[(int64_t)b * p.Nk + key] * HK_LOG2E : 0.f) : -INFINITY;
        }
}
__device__ __forceinline__ bf16x8 pack8(const f32x4 a, const f32x4 b) {
    return bf16x8{(__bf16)a[0], (__bf16)a[1], (__bf16)a[2], (__bf16)a[3], (__bf16)b[0], (__bf16)b[1], (__bf16)b[2], (__bf16)b[3]};
}

__global__ __launch_bounds__(256) void attn_train_fwd_ks_kernel(AttnT p) {
    __shared__ WaveTile sV[4];
    __shared__ float sStat[4][16][2];
    __shared__ __attribute__((aligned(16))) float sO[4][16][64];
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, lr = lane & 15, lg = lane >> 4;
    const int q0 = blockIdx.x * 16, h = blockIdx.y, b = blockIdx.z;
    const int q = q0 + lr;
    const __bf16* Kb = p.K + (int64_t)b * p.bsk + h * 64;
    const __bf16* Vb = p.V + (int64_t)b * p.bsv + h * 64;
    const __bf16* qp = p.Q + (int64_t)b * p.bsq + h * 64 + (int64_t)min(q, p.Nq - 1) * p.ldq + 8 * lg;
    const bf16x8 bq0 = *reinterpret_cast<const bf16x8*>(qp), bq1 = *reinterpret_cast<const bf16x8*>(qp + 32);
    const int nkt = (p.Nk + 63) / 64;
    // ---- pass 1: row maximum and sum of this wave's tiles ----
    float m = -INFINITY, s = 0.f;
    bf16x8 kf[4][2], kn[4][2];
    if (w < nkt) load_rows_frag(Kb, p.ldk, w, p.Nk, lr, lg, kf);
    for (int kt = w; kt < nkt; kt += 4) {
        if (kt + 4 < nkt) load_rows_frag(Kb, p.ldk, kt + 4, p.Nk, lr, lg, kn);
        float kb[4][4];
        key_bias4(p, b, kt, lg, kb);
        f32x4 acc[4];
        float tmax = -INFINITY;
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {
            acc[cb] = mfma16(kf[cb][0], bq0, f32x4{0, 0, 0, 0});
            acc[cb] = mfma16(kf[cb][1], bq1, acc[cb]);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                acc[cb][r] += kb[cb][r];
                tmax = fmaxf(tmax, acc[cb][r]);
            }
        }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 16));
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
        const float mn = fmaxf(m, tmax);
        float ts = 0.f;
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
#pragma unroll
            for (int r = 0; r < 4; ++r) ts += exp2f(acc[cb][r] - mn);
        ts += __shfl_xor(ts, 16);
        ts += __shfl_xor(ts, 32);
        s = s * exp2f(m - mn) + ts;
        m = mn;
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) kf[cb][0] = kn[cb][0], kf[cb][1] = kn[cb][1];
    }
    if (lg == 0) {
        sStat[w][lr][0] = m;
        sStat[w][lr][1] = s;
    }
    __syncthreads();
    float mg = -INFINITY, sg = 0.f;
#pragma unroll
    for (int ww = 0; ww < 4; ++ww) mg = fmaxf(mg, sStat[ww][lr][0]);
#pragma unroll
    for (int ww = 0; ww < 4; ++ww) sg += sStat[ww][lr][1] * exp2f(sStat[ww][lr][0] - mg);    // a wave without tiles: 0 * 2^-inf
    const float lse2 = mg + log2f(sg);
    if (w == 0 && lg == 0 && q < p.Nq) p.lse[((int64_t)b * p.H + h) * p.Nq + q] = lse2;
    // ---- pass 2: P^T = dropout(2^(S^T - lse)), O^T += V^T P^T ----
    f32x4 o[4];
#pragma unroll
    for (int db = 0; db < 4; ++db) o[db] = f32x4{0, 0, 0, 0};
    bf16x8 vr[8], vn[8];
    auto load_v = [&](int kt, bf16x8(&v)[8]) {
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int ch = lane + 64 * it;
            v[it] = *reinterpret_cast<const bf16x8*>(Vb + (int64_t)min(kt * 64 + (ch >> 3), p.Nk - 1) * p.ldv + (ch & 7) * 8);
        }
    };
    if (w < nkt) {
        load_rows_frag(Kb, p.ldk, w, p.Nk, lr, lg, kf);
        load_v(w, vr);
    }
    for (int kt = w; kt < nkt; kt += 4) {
        if (kt + 4 < nkt) {
            load_rows_frag(Kb, p.ldk, kt + 4, p.Nk, lr, lg, kn);
            load_v(kt + 4, vn);
        }
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int ch = lane + 64 * it;
            *reinterpret_cast<bf16x8*>(&sV[w].t[ch >> 3][(ch & 7) * 8]) = vr[it];
        }
        float kb[4][4];
        key_bias4(p, b, kt, lg, kb);
        f32x4 pr[4];
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {
            f32x4 acc = mfma16(kf[cb][0], bq0, f32x4{0, 0, 0, 0});
            acc = mfma16(kf[cb][1], bq1, acc);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = exp2f(acc[r] + kb[cb][r] - lse2);
                if (p.thresh) {
                    const int key = kt * 64 + 16 * cb + 4 * lg + r;
                    const uint32_t idx = (uint32_t)(((int64_t)b * p.H + h) * p.Nq + q) * (uint32_t)p.Nk + (uint32_t)key;
                    v = keep_at(idx, p.key, p.thresh) ? v * p.inv_keep : 0.f;
                }
                pr[cb][r] = v;
            }
        }
        const bf16x8 bp0 = pack8(pr[0], pr[1]), bp1 = pack8(pr[2], pr[3]);
        asm volatile("" ::: "memory");   // the tile's LDS writes stay ahead of the transposed reads (one wave: LDS runs in order)
#pragma unroll
        for (int db = 0; db < 4; ++db) {
            o[db] = mfma16(frag_tr(sV[w], 0, db, lane), bp0, o[db]);
            o[db] = mfma16(frag_tr(sV[w], 1, db, lane), bp1, o[db]);
        }
        asm volatile("" ::: "memory");
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) kf[cb][0] = kn[cb][0], kf[cb][1] = kn[cb][1];
#pragma unroll
        for (int it = 0; it < 8; ++it) vr[it] = vn[it];
    }
#pragma unroll
    for (int db = 0; db < 4; ++db) *reinterpret_cast<f32x4*>(&sO[w][lr][16 * db + 4 * lg]) = o[db];
    __syncthreads();
    {
        const int qi = tid >> 4, d4 = (tid & 15) * 4, qq = q0 + qi;
        f32x4 v = *reinterpret_cast<const f32x4*>(&sO[0][qi][d4]);
#pragma unroll
        for (int ww = 1; ww < 4; ++ww) v += *reinterpret_cast<const f32x4*>(&sO[ww][qi][d4]);
        if (qq < p.Nq) {
            *reinterpret_cast<bf16x4_t*>(p.O + (int64_t)b * p.bso + h * 64 + (int64_t)qq * p.ldo + d4) =
                bf16x4_t{(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
            if (p.O32) *reinterpret_cast<f32x4*>(p.O32 + (((int64_t)b * p.Nq + qq) * p.H + h) * 64 + d4) = v;
        }
    }
}

// query-side backward, key-split: dsum[q] = sum_d dO.O (fp32 O), dQ = softmax_scale * dS K
__global__ __launch_bounds__(256) void attn_train_bwd_q_ks_kernel(AttnT p) {
    __shared__ WaveTile sK[4];
    __shared__ __attribute__((aligned(16))) float sO[4][16][64];
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, lr = lane & 15, lg = lane >> 4;
    const int q0 = blockIdx.x * 16, h = blockIdx.y, b = blockIdx.z;
    const int q = q0 + lr, qc = min(q, p.Nq - 1);
    const __bf16* Kb = p.K + (int64_t)b * p.bsk + h * 64;
    const __bf16* Vb = p.V + (int64_t)b * p.bsv + h * 64;
    const __bf16* qp = p.Q + (int64_t)b * p.bsq + h * 64 + (int64_t)qc * p.ldq + 8 * lg;
    const __bf16* gp = p.dO + (int64_t)b * p.bso + h * 64 + (int64_t)qc * p.ldo + 8 * lg;
    const bf16x8 bq0 = *reinterpret_cast<const bf16x8*>(qp), bq1 = *reinterpret_cast<const bf16x8*>(qp + 32);
    const bf16x8 bg0 = *reinterpret_cast<const bf16x8*>(gp), bg1 = *reinterpret_cast<const bf16x8*>(gp + 32);
    float dsm = 0.f;
    if (p.O32) {
        const float* orow = p.O32 + (((int64_t)b * p.Nq + qc) * p.H + h) * 64 + 8 * lg;
#pragma unroll
        for (int e = 0; e < 8; ++e) dsm += (float)bg0[e] * orow[e] + (float)bg1[e] * orow[32 + e];
    } else {
        const __bf16* orow = p.O + (int64_t)b * p.bso + h * 64 + (int64_t)qc * p.ldo + 8 * lg;
        const bf16x8 o0 = *reinterpret_cast<const bf16x8*>(orow), o1 = *reinterpret_cast<const bf16x8*>(orow + 32);
#pragma unroll
        for (int e = 0; e < 8; ++e) dsm += (float)bg0[e] * (float)o0[e] + (float)bg1[e] * (float)o1[e];
    }
    dsm += __shfl_xor(dsm, 16);
    dsm += __shfl_xor(dsm, 32);
    if (w == 0 && lg == 0 && q < p.Nq) p.dsum[((int64_t)b * p.H + h) * p.Nq + q] = dsm;
    const float lse2 = p.lse[((int64_t)b * p.H + h) * p.Nq + qc];
    f32x4 dq[4];
#pragma unroll
    for (int db = 0; db < 4; ++db) dq[db] = f32x4{0, 0, 0, 0};
    const int nkt = (p.Nk + 63) / 64;
    bf16x8 kf[4][2], kn[4][2], vf[4][2], vn[4][2];
    if (w < nkt) {
        load_rows_frag(Kb, p.ldk, w, p.Nk, lr, lg, kf);
        load_rows_frag(Vb, p.ldv, w, p.Nk, lr, lg, vf);
    }
    for (int kt = w; kt < nkt; kt += 4) {
        if (kt + 4 < nkt) {
            load_rows_frag(Kb, p.ldk, kt + 4, p.Nk, lr, lg, kn);
            load_rows_frag(Vb, p.ldv, kt + 4, p.Nk, lr, lg, vn);
        }
        // the K fragments are rows of the tile: the same registers fill the row-major LDS tile for K^T
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {
            *reinterpret_cast<bf16x8*>(&sK[w].t[16 * cb + lr][8 * lg]) = kf[cb][0];
            *reinterpret_cast<bf16x8*>(&sK[w].t[16 * cb + lr][8 * lg + 32]) = kf[cb][1];
        }
        float kb[4][4];
        key_bias4(p, b, kt, lg, kb);
        f32x4 ds[4];
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {
            f32x4 acc = mfma16(kf[cb][0], bq0, f32x4{0, 0, 0, 0});
            acc = mfma16(kf[cb][1], bq1, acc);
            f32x4 dp = mfma16(vf[cb][0], bg0, f32x4{0, 0, 0, 0});
            dp = mfma16(vf[cb][1], bg1, dp);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float pr = exp2f(acc[r] + kb[cb][r] - lse2);
                float g = dp[r];
                if (p.thresh) {
                    const int key = kt * 64 + 16 * cb + 4 * lg + r;
                    const uint32_t idx = (uint32_t)(((int64_t)b * p.H + h) * p.Nq + q) * (uint32_t)p.Nk + (uint32_t)key;
                    g = keep_at(idx, p.key, p.thresh) ? g * p.inv_keep : 0.f;
                }
                ds[cb][r] = pr * (g - dsm);
            }
        }
        const bf16x8 bs0 = pack8(ds[0], ds[1]), bs1 = pack8(ds[2], ds[3]);
        asm volatile("" ::: "memory");
#pragma unroll
        for (int db = 0; db < 4; ++db) {
            dq[db] = mfma16(frag_tr(sK[w], 0, db, lane), bs0, dq[db]);
            dq[db] = mfma16(frag_tr(sK[w], 1, db, lane), bs1, dq[db]);
        }
        asm volatile("" ::: "memory");
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {
            kf[cb][0] = kn[cb][0], kf[cb][1] = kn[cb][1];
            vf[cb][0] = vn[cb][0], vf[cb][1] = vn[cb][1];
        }
    }
#pragma unroll
    for (int db = 0; db < 4; ++db) *reinterpret_cast<f32x4*>(&sO[w][lr][16 * db + 4 * lg]) = dq[db];
    __syncthreads();
    {
        const int qi = tid >> 4, d4 = (tid & 15) * 4, qq = q0 + qi;
        f32x4 v = *reinterpret_cast<const f32x4*>(&sO[0][qi][d4]);
#pragma unroll
        for (int ww = 1; ww < 4; ++ww) v += *reinterpret_cast<const f32x4*>(&sO[ww][qi][d4]);
        v *= p.sm_scale;
        if (qq < p.Nq)
            *reinterpret_cast<bf16x4_t*>(p.dQ + (int64_t)b * p.bsdq + h * 64 + (int64_t)qq * p.lddq + d4) =
                bf16x4_t{(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
    }
}

// backward, key side: dV = P~^T dO, dK = ln2 * dS^T Q' (Q' carries softmax_scale * log2 e); one workgroup per 64 keys
__global__ __launch_bounds__(256) void attn_train_bwd_kv_kernel(AttnT p) {
    __shared__ __attribute__((aligned(16))) __bf16 sQ[64][TS];
    __shared__ __attribute__((aligned(16))) __bf16 sQt[64][TS];
    __shared__ __attribute__((aligned(16))) __bf16 sG[64][TS];      // dO
    __shared__ __attribute__((aligned(16))) __bf16 sGt[64][TS];
    __shared__ __attribute__((aligned(16))) __bf16 sP[4][16][TS];
    __shared__ __attribute__((aligned(16))) __bf16 sS[4][16][TS];
    __shared__ float sLse[64], sDs[64];
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, lr = lane & 15, lg = lane >> 4;
    const int k0 = blockIdx.x * 64, h = blockIdx.y, b = blockIdx.z;
    const __bf16* Qb = p.Q + (int64_t)b * p.bsq + h * 64;
    const __bf16* Gb = p.dO + (int64_t)b * p.bso + h * 64;
    bf16x8 ak[2], av[2];
    {
        const int key = k0 + 16 * w + lr;
        const bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
        ak[0] = ak[1] = av[0] = av[1] = z;
        if (key < p.Nk) {
            const __bf16* kr = p.K + (int64_t)b * p.bsk + h * 64 + (int64_t)key * p.ldk;
            const __bf16* vr = p.V + (int64_t)b * p.bsv + h * 64 + (int64_t)key * p.ldv;
            ak[0] = *reinterpret_cast<const bf16x8*>(kr + 8 * lg);
            ak[1] = *reinterpret_cast<const bf16x8*>(kr + 8 * lg + 32);
            av[0] = *reinterpret_cast<const bf16x8*>(vr + 8 * lg);
            av[1] = *reinterpret_cast<const bf16x8*>(vr + 8 * lg + 32);
        }
    }
    float kb[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int key = k0 + 16 * w + 4 * lg + r;
        kb[r] = key < p.Nk ? (p.kbias ? p.kbias[(int64_t)b * p.Nk + key] * HK_LOG2E : 0.f) : -INFINITY;
    }
    f32x4 dk[4], dv[4];
#pragma unroll
    for (int db = 0; db < 4; ++db) dk[db] = dv[db] = f32x4{0, 0, 0, 0};
    const int nqt = (p.Nq + 63) / 64;
    for (int qt = 0; qt < nqt; ++qt) {
        __syncthreads();
        load_tile(Qb, p.ldq, qt * 64, p.Nq, sQ, sQt, tid);
        load_tile(Gb, p.ldo, qt * 64, p.Nq, sG, sGt, tid);
        if (tid < 64) {
            const int q = qt * 64 + tid;
            const int64_t o = ((int64_t)b * p.H + h) * p.Nq + q;
            sLse[tid] = q < p.Nq ? p.lse[o] : 0.f;
            sDs[tid] = q < p.Nq ? p.dsum[o] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int qb = 0; qb < 4; ++qb) {
            f32x4 acc = f32x4{0, 0, 0, 0}, dp = f32x4{0, 0, 0, 0};
            acc = mfma16(ak[0], frag(sQ, 16 * qb + lr, 8 * lg), acc);
            acc = mfma16(ak[1], frag(sQ, 16 * qb + lr, 8 * lg + 32), acc);
            dp = mfma16(av[0], frag(sG, 16 * qb + lr, 8 * lg), dp);
            dp = mfma16(av[1], frag(sG, 16 * qb + lr, 8 * lg + 32), dp);
            const int ql = 16 * qb + lr, q = qt * 64 + ql;
            const float l2 = sLse[ql], ds = sDs[ql];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = k0 + 16 * w + 4 * lg + r;
                const float pr = exp2f(acc[r] + kb[r] - l2);
                float pd = pr, g = dp[r];
                if (p.thresh) {
                    const uint32_t idx = (uint32_t)(((int64_t)b * p.H + h) * p.Nq + q) * (uint32_t)p.Nk + (uint32_t)key;
                    const bool kp = keep_at(idx, p.key, p.thresh);
                    pd = kp ? pr * p.inv_keep : 0.f;
                    g = kp ? g * p.inv_keep : 0.f;
                }
                sP[w][4 * lg + r][ql] = (__bf16)pd;
                sS[w][4 * lg + r][ql] = (__bf16)(pr * (g - ds));
            }
        }
        __syncthreads();
        const bf16x8 p0 = *reinterpret_cast<const bf16x8*>(&sP[w][lr][8 * lg]);
        const bf16x8 p1 = *reinterpret_cast<const bf16x8*>(&sP[w][lr][8 * lg + 32]);
        const bf16x8 s0 = *reinterpret_cast<const bf16x8*>(&sS[w][lr][8 * lg]);
        const bf16x8 s1 = *reinterpret_cast<const bf16x8*>(&sS[w][lr][8 * lg + 32]);
#pragma unroll
        for (int db = 0; db < 4; ++db) {
            dv[db] = mfma16(p0, frag(sGt, 16 * db + lr, 8 * lg), dv[db]);
            dv[db] = mfma16(p1, frag(sGt, 16 * db + lr, 8 * lg + 32), dv[db]);
            dk[db] = mfma16(s0, frag(sQt, 16 * db + lr, 8 * lg), dk[db]);
            dk[db] = mfma16(s1, frag(sQt, 16 * db + lr, 8 * lg + 32), dk[db]);
        }
    }
    __bf16* dKb = p.dK + (int64_t)b * p.bsdk + h * 64;
    __bf16* dVb = p.dV + (int64_t)b * p.bsdv + h * 64;
    const float ln2 = 0.6931471805599453f;
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int key = k0 + 16 * w + 4 * lg + r;
            if (key < p.Nk) {
                dKb[(int64_t)key * p.lddk + 16 * db + lr] = (__bf16)(dk[db][r] * ln2);
                dVb[(int64_t)key * p.lddv + 16 * db + lr] = (__bf16)dv[db][r];
            }
        }
}

// ------------------------------------------------------------------------------------------------------------
// Post-LN residual block with dropout / drop-path (BertSelfOutput, BertOutput: Qformer.py:313-318, 401-406):
//   z = drop_path_b(dropout(t)) + res ;  y = LayerNorm(z) [; y = dropout_post(y) for the embeddings, Qformer.py:119-120]
// wave per row, row in registers (D <= 1024), fp32 statistics.
// ------------------------------------------------------------------------------------------------------------
struct RowMapD {
    int64_t ld, rpb, bs;
    __device__ __forceinline__ int64_t off(int64_t m) const {
        if (rpb == 0) return m * ld;
        const int64_t bb = m / rpb;
        return bb * bs + (m - bb * rpb) * ld;
    }
};

struct DropP {
    uint32_t key, thresh;      // element dropout (thresh 0 = off)
    float inv_keep;
    uint32_t pkey, pthresh;    // drop-path, one draw per sample (pthresh 0 = off)
    float pinv_keep;
    int rows_per_sample;
};

__device__ __forceinline__ float path_scale(const DropP& d, int64_t row) {
    if (!d.pthresh) return 1.f;
    const uint32_t smp = (uint32_t)(row / d.rows_per_sample);
    return keep_at(smp, d.pkey, d.pthresh) ? d.pinv_keep : 0.f;
}

__global__ __launch_bounds__(256) void drop_add_ln_kernel(const float* t, const float* res, RowMapD rmap, DropP dr,
                                                          DropP post, const float* gamma, const float* beta,
                                                          float* z_out, float* y_f, __bf16* y_b, RowMapD omap,
                                                          int64_t rows, int D, float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    f32x4 x[4];
    const float ps = path_scale(dr, row) * (dr.thresh ? dr.inv_keep : 1.f);
    const int64_t ro = res ? rmap.off(row) : 0;
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int i = (c * 64 + lane) * 4;
        if (i < D) {
            f32x4 v = f32x4{0, 0, 0, 0};
            if (t) {
                v = *reinterpret_cast<const f32x4*>(t + row * D + i);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const bool kp = !dr.thresh || keep_at((uint32_t)(row * D + i + e), dr.key, dr.thresh);
                    v[e] = kp ? v[e] * ps : 0.f;
                }
            }
            if (res) v += *reinterpret_cast<const f32x4*>(res + ro + i);
            if (z_out) *reinterpret_cast<f32x4*>(z_out + row * D + i) = v;
            x[c] = v;
            sum += v[0] + v[1] + v[2] + v[3];
        }
    }
    const float mean = wave_sum(sum) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c)
        if ((c * 64 + lane) * 4 < D) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float d = x[c][e] - mean;
                q += d * d;
            }
        }
    const float rstd = rsqrtf(wave_sum(q) / (float)D + eps);
    const int64_t oo = omap.off(row);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int i = (c * 64 + lane) * 4;
        if (i < D) {
            const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + i);
            const f32x4 bt = *reinterpret_cast<const f32x4*>(beta + i);
            f32x4 y;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                y[e] = (x[c][e] - mean) * rstd * g[e] + bt[e];
                if (post.thresh) y[e] = keep_at((uint32_t)(row * D + i + e), post.key, post.thresh) ? y[e] * post.inv_keep : 0.f;
            }
            if (y_f) *reinterpret_cast<f32x4*>(y_f + oo + i) = y;
            if (y_b) {
                u32x2 o;
                o[0] = pack_bf16x2(y[0], y[1]);
                o[1] = pack_bf16x2(y[2], y[3]);
                *reinterpret_cast<u32x2*>(y_b + oo + i) = o;
            }
        }
    }
}

// LayerNorm backward of the block above.  dy (row-mapped) [, * post mask] -> dz fp32 [rows, D] (the gradient of the
// residual input), dt bf16 [rows, D] = dz * pre-mask (the gradient of the dense output, a GEMM operand), and per
// workgroup partial sums of dgamma / dbeta in part[gridDim.x][2][D] (reduced by hk_colsum: deterministic).
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* dy, RowMapD dmap, const float* z, const float* gamma,
                                                     DropP dr, DropP post, float* dz, __bf16* dt, float* part,
                                                     int64_t rows, int D, float eps) {
    __shared__ float red[4][2][1024];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    f32x4 ag[4], ab[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) ag[c] = ab[c] = f32x4{0, 0, 0, 0};
    for (int64_t row = (int64_t)blockIdx.x * 4 + w; row < rows; row += (int64_t)gridDim.x * 4) {
        f32x4 x[4], g[4];
        const int64_t go = dmap.off(row);
        float sum = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int i = (c * 64 + lane) * 4;
            if (i < D) {
                x[c] = *reinterpret_cast<const f32x4*>(z + row * D + i);
                g[c] = *reinterpret_cast<const f32x4*>(dy + go + i);
                if (post.thresh) {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        g[c][e] = keep_at((uint32_t)(row * D + i + e), post.key, post.thresh) ? g[c][e] * post.inv_keep : 0.f;
                }
                sum += x[c][0] + x[c][1] + x[c][2] + x[c][3];
            }
        }
        const float mean = wave_sum(sum) / (float)D;
        float q = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if ((c * 64 + lane) * 4 < D) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float d = x[c][e] - mean;
                    q += d * d;
                }
            }
        const float rstd = rsqrtf(wave_sum(q) / (float)D + eps);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int i = (c * 64 + lane) * 4;
            if (i < D) {
                const f32x4 gm = *reinterpret_cast<const f32x4*>(gamma + i);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float xh = (x[c][e] - mean) * rstd;
                    ag[c][e] += g[c][e] * xh;
                    ab[c][e] += g[c][e];
                    const float gg = g[c][e] * gm[e];
                    x[c][e] = xh;
                    g[c][e] = gg;
                    s1 += gg;
                    s2 += gg * xh;
                }
            }
        }
        s1 = wave_sum(s1) / (float)D;
        s2 = wave_sum(s2) / (float)D;
        const float ps = path_scale(dr, row) * (dr.thresh ? dr.inv_keep : 1.f);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int i = (c * 64 + lane) * 4;
            if (i < D) {
                f32x4 d, m;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    d[e] = rstd * (g[c][e] - s1 - x[c][e] * s2);
                    const bool kp = !dr.thresh || keep_at((uint32_t)(row * D + i + e), dr.key, dr.thresh);
                    m[e] = kp ? d[e] * ps : 0.f;
                }
                if (dz) *reinterpret_cast<f32x4*>(dz + row * D + i) = d;
                if (dt) {
                    u32x2 o;
                    o[0] = pack_bf16x2(m[0], m[1]);
                    o[1] = pack_bf16x2(m[2], m[3]);
                    *reinterpret_cast<u32x2*>(dt + row * D + i) = o;
                }
            }
        }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int i = (c * 64 + lane) * 4;
        if (i < D) {
            *reinterpret_cast<f32x4*>(&red[w][0][i]) = ag[c];
            *reinterpret_cast<f32x4*>(&red[w][1][i]) = ab[c];
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * D; i += 256) {
        const int k = i / D, c = i - k * D;
        part[((int64_t)blockIdx.x * 2 + k) * D + c] = red[0][k][c] + red[1][k][c] + red[2][k][c] + red[3][k][c];
    }
}

// ------------------------------------------------------------------------------------------------------------
// column sums (bias gradients, reduction of the LayerNorm partials): out[c] = sum_r x[r][c]
// grid (ceil(C/64), nsplit); nsplit > 1 writes partials [nsplit, C] that a second launch reduces.
// ------------------------------------------------------------------------------------------------------------
template <bool BF16>
__global__ __launch_bounds__(256) void colsum_kernel(const void* x, int64_t ld, int64_t R, int C, float* out) {
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    const int64_t per = (R + gridDim.y - 1) / gridDim.y;
    const int64_t r0 = (int64_t)blockIdx.y * per, r1 = r0 + per < R ? r0 + per : R;
    float acc = 0.f;
    if (c < C) {
        for (int64_t r = r0 + w; r < r1; r += 4) {
            if (BF16) acc += (float)reinterpret_cast<const __bf16*>(x)[r * ld + c];
            else acc += reinterpret_cast<const float*>(x)[r * ld + c];
        }
    }
    red[w][lane] = acc;
    __syncthreads();
    if (w == 0 && c < C) out[(int64_t)blockIdx.y * C + c] = red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane];
}

// ------------------------------------------------------------------------------------------------------------
// transpose (+ cast): x [R, C] (row-mapped rows; bf16 or fp32) -> out bf16 [C, ldo], out[c][r] = x[r][c] for r < R,
// zero for R <= r < Rpad (the wgrad GEMM contracts over r and needs a multiple of 64).
// ------------------------------------------------------------------------------------------------------------
template <bool BF16>
__global__ __launch_bounds__(256) void transpose_kernel(const void* x, RowMapD xmap, int64_t R, int C, __bf16* out,
                                                        int64_t ldo, int64_t Rpad) {
    __shared__ float tile[64][65];
    const int64_t r0 = (int64_t)blockIdx.x * 64;
    const int c0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) {
        const int64_t r = r0 + i;
        const int c = c0 + tx;
        float v = 0.f;
        if (r < R && c < C) {
            const int64_t o = xmap.off(r) + c;
            v = BF16 ? (float)reinterpret_cast<const __bf16*>(x)[o] : reinterpret_cast<const float*>(x)[o];
        }
        tile[i][tx] = v;
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
        const int c = c0 + i;
        const int64_t r = r0 + tx;
        if (c < C && r < Rpad) out[(int64_t)c * ldo + r] = (__bf16)tile[tx][i];
    }
}

// Many fp32 -> bf16 transposes in ONE launch (the W^T operands of the dgrad GEMMs, refreshed after every optimizer
// step: ~110 Linear weights; as separate launches they were 0.9 ms of launch latency for 0.25 ms of traffic).  jobs[j]
// covers workgroups [tile0, next job's tile0): a 64 x 64 tile each, row tiles fastest.
__global__ __launch_bounds__(256) void transpose_batch_kernel(const hk_transpose_job* jobs, int n_jobs) {
    __shared__ float tile[64][65];
    int lo = 0, hi = n_jobs - 1;      // last job whose tile0 <= blockIdx.x
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].tile0 <= (int)blockIdx.x) lo = mid;
        else hi = mid - 1;
    }
    const hk_transpose_job jb = jobs[lo];
    const int t = (int)blockIdx.x - jb.tile0;
    const int rt = (jb.R + 63) / 64;
    const int r0 = (t % rt) * 64, c0 = (t / rt) * 64;
    const float* x = reinterpret_cast<const float*>(jb.x);
    __bf16* out = reinterpret_cast<__bf16*>(jb.out);
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) {
        const int r = r0 + i, c = c0 + tx;
        tile[i][tx] = (r < jb.R && c < jb.C) ? x[(int64_t)r * jb.ld + c] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
        const int c = c0 + i, r = r0 + tx;
        if (c < jb.C && r < jb.R) out[(int64_t)c * jb.ldo + r] = (__bf16)tile[tx][i];
    }
}

// ------------------------------------------------------------------------------------------------------------
// exact-erf GELU forward / backward on bf16 (BertIntermediate, Qformer.py:387-390; ACT2FN["gelu"])
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gelu_fwd_kernel(const __bf16* u, __bf16* g, int64_t n) {
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 8;
    if (i >= n) return;
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(u + i);
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float x = (float)v[e];
        o[e] = (__bf16)(0.5f * x * (1.f + erff(x * 0.70710678118654752f)));
    }
    *reinterpret_cast<bf16x8*>(g + i) = o;
}

__global__ __launch_bounds__(256) void gelu_bwd_kernel(const __bf16* dg, const __bf16* u, __bf16* du, int64_t n) {
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 8;
    if (i >= n) return;
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(u + i);
    const bf16x8 d = *reinterpret_cast<const bf16x8*>(dg + i);
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float x = (float)v[e];
        const float cdf = 0.5f * (1.f + erff(x * 0.70710678118654752f));
        const float pdf = 0.3989422804014327f * __expf(-0.5f * x * x);
        o[e] = (__bf16)((float)d[e] * (cdf + x * pdf));
    }
    *reinterpret_cast<bf16x8*>(du + i) = o;
}

// ------------------------------------------------------------------------------------------------------------
// Q-Former embeddings (Qformer.py:97-118): the pre-LayerNorm rows z = cat(query_tokens, word[ids] + pos) and
// their backward (gradients of query_tokens, position rows; word rows by atomic add -- tokens repeat).
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void embed_gather_kernel(const float* query, const float* wemb, const float* pemb,
                                                           const int64_t* ids, float* z, int B, int nq, int Lt, int d,
                                                           int vocab) {
    const int S = nq + Lt;
    const int64_t n4 = (int64_t)B * S * (d / 4);
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const int c = (int)(i % (d / 4)) * 4;
    const int64_t row = i / (d / 4);
    const int b = (int)(row / S), s = (int)(row - (int64_t)b * S);
    f32x4 v;
    if (s < nq) {
        v = *reinterpret_cast<const f32x4*>(query + (int64_t)s * d + c);
    } else {
        int64_t id = ids[(int64_t)b * Lt + (s - nq)];
        id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
        v = *reinterpret_cast<const f32x4*>(wemb + id * d + c);
        v += *reinterpret_cast<const f32x4*>(pemb + (int64_t)(s - nq) * d + c);
    }
    *reinterpret_cast<f32x4*>(z + row * d + c) = v;
}

__global__ __launch_bounds__(256) void embed_bwd_kernel(const float* dz, const int64_t* ids, float* dquery, float* dword,
                                                        float* dpos, int B, int nq, int Lt, int d, int vocab) {
    const int S = nq + Lt;
    const int64_t n = (int64_t)S * d;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int s = (int)(i / d), c = (int)(i - (int64_t)s * d);
    float acc = 0.f;
    for (int b = 0; b < B; ++b) {
        const float g = dz[((int64_t)b * S + s) * d + c];
        acc += g;
        if (s >= nq) {
            int64_t id = ids[(int64_t)b * Lt + (s - nq)];
            id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
            atomicAdd(dword + id * d + c, g);
        }
    }
    if (s < nq) dquery[(int64_t)s * d + c] = acc;
    else dpos[(int64_t)(s - nq) * d + c] = acc;
}

DropP make_drop(const hk_dropout* d) {
    DropP r;
    r.key = r.thresh = r.pkey = r.pthresh = 0;
    r.inv_keep = r.pinv_keep = 1.f;
    r.rows_per_sample = 1;
    if (d) {
        r.thresh = drop_thresh(d->p);
        r.key = site_key(d->seed, d->site);
        r.inv_keep = d->p > 0.f ? 1.f / (1.f - d->p) : 1.f;
        r.pthresh = drop_thresh(d->p_path);
        r.pkey = site_key(d->seed, d->path_site);
        r.pinv_keep = d->p_path > 0.f ? 1.f / (1.f - d->p_path) : 1.f;
        r.rows_per_sample = d->rows_per_sample > 0 ? d->rows_per_sample : 1;
    }
    return r;
}

RowMapD make_map(const hk_rows* r, int D) {
    RowMapD m;
    m.ld = r ? r->ld : D;
    m.rpb = r ? r->rows_per_batch : 0;
    m.bs = r ? r->batch_stride : 0;
    return m;
}

bool drop_ok(const hk_dropout* d) {
    return !d || (d->p >= 0.f && d->p < 1.f && d->p_path >= 0.f && d->p_path < 1.f);
}

int fill_attn(const hk_attn_train_desc* d, AttnT& p, bool bwd, const char* who) {
    HK_REQUIRE(d && d->Q && d->K && d->V && d->O && d->lse, "%s: null pointer", who);
    HK_REQUIRE(!bwd || (d->dO && d->dQ && d->dK && d->dV && d->dsum), "%s: null gradient pointer", who);
    HK_REQUIRE(d->B > 0 && d->H > 0 && d->Nq > 0 && d->Nk > 0, "%s: empty problem", who);
    HK_REQUIRE((double)d->B * d->H * d->Nq * d->Nk < 4294967296.0, "%s: B*H*Nq*Nk must stay below 2^32 (dropout counter)", who);
    HK_REQUIRE(d->B < 65536 && d->H < 65536, "%s: B and H must be below 65536", who);
    HK_REQUIRE(d->p_drop >= 0.f && d->p_drop < 1.f, "%s: p_drop must be in [0,1)", who);
    const int64_t strides[] = {d->ldq, d->ldk, d->ldv, d->ldo, d->bsq, d->bsk, d->bsv, d->bso};
    for (int64_t s : strides) HK_REQUIRE(s % 8 == 0, "%s: strides must be multiples of 8 elements", who);
    HK_REQUIRE(((uintptr_t)d->Q | (uintptr_t)d->K | (uintptr_t)d->V | (uintptr_t)d->O | (uintptr_t)d->dO) % 16 == 0,
               "%s: operands must be 16-byte aligned", who);
    p.Q = (const __bf16*)d->Q; p.K = (const __bf16*)d->K; p.V = (const __bf16*)d->V; p.dO = (const __bf16*)d->dO;
    p.O = (__bf16*)d->O; p.dQ = (__bf16*)d->dQ; p.dK = (__bf16*)d->dK; p.dV = (__bf16*)d->dV;
    p.lse = d->lse; p.dsum = d->dsum; p.kbias = d->key_bias; p.O32 = d->o_f32;
    p.B = d->B; p.H = d->H; p.Nq = d->Nq; p.Nk = d->Nk;
    p.ldq = d->ldq; p.ldk = d->ldk; p.ldv = d->ldv; p.ldo = d->ldo;
    p.bsq = d->bsq; p.bsk = d->bsk; p.bsv = d->bsv; p.bso = d->bso;
    p.lddq = d->lddq; p.lddk = d->lddk; p.lddv = d->lddv; p.bsdq = d->bsdq; p.bsdk = d->bsdk; p.bsdv = d->bsdv;
    p.sm_scale = d->softmax_scale;
    p.thresh = drop_thresh(d->p_drop);
    p.key = site_key(d->seed, d->site);
    p.inv_keep = d->p_drop > 0.f ? 1.f / (1.f - d->p_drop) : 1.f;
    return HK_OK;
}

bool attn_ks() {
    static const bool on = !(getenv("HK_ATTN_TRAIN_KS") && atoi(getenv("HK_ATTN_TRAIN_KS")) == 0);
    return on;
}

}  // namespace

extern "C" int hk_attn_train_fwd(const hk_attn_train_desc* d, void* stream) {
    AttnT p;
    const int rc = fill_attn(d, p, false, "hk_attn_train_fwd");
    if (rc != HK_OK) return rc;
    // key-split form (16 queries per workgroup, waves over the key tiles) unless switched off or the fp32 copy of O is
    // not 16-byte aligned; HK_ATTN_TRAIN_KS=0: the round-2 kernels (A/B aid)
    if (attn_ks() && (uintptr_t)d->o_f32 % 16 == 0) {
        hipLaunchKernelGGL(attn_train_fwd_ks_kernel, dim3((d->Nq + 15) / 16, d->H, d->B), dim3(256), 0,
                           reinterpret_cast<hipStream_t>(stream), p);
        HK_CHECK_LAUNCH("hk_attn_train_fwd");
        return HK_OK;
    }
    hipLaunchKernelGGL(attn_train_fwd_kernel, dim3((d->Nq + 63) / 64, d->H, d->B), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), p);
    HK_CHECK_LAUNCH("hk_attn_train_fwd");
    return HK_OK;
}

extern "C" int hk_attn_train_bwd(const hk_attn_train_desc* d, void* stream) {
    AttnT p;
    const int rc = fill_attn(d, p, true, "hk_attn_train_bwd");
    if (rc != HK_OK) return rc;
    const int64_t gs[] = {d->lddq, d->lddk, d->lddv, d->bsdq, d->bsdk, d->bsdv};
    for (int64_t s : gs) HK_REQUIRE(s >= 0, "hk_attn_train_bwd: negative gradient stride");
    HK_REQUIRE(d->lddq >= (int64_t)d->H * 64 && d->lddk >= (int64_t)d->H * 64 && d->lddv >= (int64_t)d->H * 64,
               "hk_attn_train_bwd: gradient row strides smaller than H*64");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (attn_ks() && d->lddq % 4 == 0 && d->bsdq % 4 == 0 && (uintptr_t)d->dQ % 8 == 0)
        hipLaunchKernelGGL(attn_train_bwd_q_ks_kernel, dim3((d->Nq + 15) / 16, d->H, d->B), dim3(256), 0, s, p);
    else
        hipLaunchKernelGGL(attn_train_bwd_q_kernel, dim3((d->Nq + 63) / 64, d->H, d->B), dim3(256), 0, s, p);
    HK_CHECK_LAUNCH("hk_attn_train_bwd(q)");
    hipLaunchKernelGGL(attn_train_bwd_kv_kernel, dim3((d->Nk + 63) / 64, d->H, d->B), dim3(256), 0, s, p);
    HK_CHECK_LAUNCH("hk_attn_train_bwd(kv)");
    return HK_OK;
}

extern "C" int hk_drop_add_ln(const float* t, const float* res, const hk_rows* res_rows, const hk_dropout* drop,
                              const hk_dropout* post_drop, const float* gamma, const float* beta, float* z_out,
                              float* y_f32, void* y_bf16, const hk_rows* out_rows, int64_t rows, int32_t D, float eps,
                              void* stream) {
    HK_REQUIRE((t || res) && gamma && beta && (y_f32 || y_bf16), "hk_drop_add_ln: null pointer");
    HK_REQUIRE(rows > 0 && D > 0 && D % 4 == 0 && D <= 1024, "hk_drop_add_ln: bad shape rows=%ld D=%d", (long)rows, D);
    HK_REQUIRE((double)rows * D < 4294967296.0, "hk_drop_add_ln: rows*D must stay below 2^32 (dropout counter)");
    HK_REQUIRE(drop_ok(drop) && drop_ok(post_drop), "hk_drop_add_ln: dropout probabilities must be in [0,1)");
    HK_REQUIRE(((uintptr_t)t | (uintptr_t)res | (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)z_out | (uintptr_t)y_f32 |
                (uintptr_t)y_bf16) % 16 == 0, "hk_drop_add_ln: operands must be 16-byte aligned");
    const RowMapD rm = make_map(res_rows, D), om = make_map(out_rows, D);
    HK_REQUIRE(rm.ld % 4 == 0 && rm.bs % 4 == 0 && om.ld % 4 == 0 && om.bs % 4 == 0 && rm.ld >= D && om.ld >= D,
               "hk_drop_add_ln: bad row mapping");
    hipLaunchKernelGGL(drop_add_ln_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), t, res, rm, make_drop(drop), make_drop(post_drop), gamma,
                       beta, z_out, y_f32, reinterpret_cast<__bf16*>(y_bf16), om, rows, D, eps);
    HK_CHECK_LAUNCH("hk_drop_add_ln");
    return HK_OK;
}

// the two launches of hk_colsum; ws == nullptr: one launch, no partials
static int colsum_launch(const void* x, int32_t dtype, int64_t ld, int64_t R, int32_t C, float* out, float* ws,
                         hipStream_t s, const char* who) {
    int nsplit = (int)(R / 256);
    nsplit = nsplit < 1 ? 1 : (nsplit > HK_COLSUM_MAX_SPLIT ? HK_COLSUM_MAX_SPLIT : nsplit);
    if (!ws) nsplit = 1;
    float* first = nsplit > 1 ? ws : out;
    const dim3 grid((C + 63) / 64, nsplit);
    if (dtype == HK_DT_BF16) hipLaunchKernelGGL((colsum_kernel<true>), grid, dim3(256), 0, s, x, ld, R, C, first);
    else hipLaunchKernelGGL((colsum_kernel<false>), grid, dim3(256), 0, s, x, ld, R, C, first);
    HK_CHECK_LAUNCH(who);
    if (nsplit > 1) {
        hipLaunchKernelGGL((colsum_kernel<false>), dim3((C + 63) / 64, 1), dim3(256), 0, s, (const void*)ws, (int64_t)C,
                           (int64_t)nsplit, C, out);
        HK_CHECK_LAUNCH(who);
    }
    return HK_OK;
}

extern "C" int hk_colsum(const void* x, int32_t dtype, int64_t ld, int64_t R, int32_t C, float* out, float* ws,
                         void* stream) {
    HK_REQUIRE(x && out && R > 0 && C > 0 && ld >= C, "hk_colsum: bad arguments");
    HK_REQUIRE(dtype == HK_DT_BF16 || dtype == HK_DT_F32, "hk_colsum: unknown dtype %d", dtype);
    return colsum_launch(x, dtype, ld, R, C, out, ws, reinterpret_cast<hipStream_t>(stream), "hk_colsum");
}

// ------------------------------------------------------------------------------------------------------------
// hk_colsum_static: hk_colsum for callers that cannot allocate.  The [HK_COLSUM_MAX_SPLIT, C] partial sums of the
// two-stage reduction live in a file-scope device array; every GPU the library runs on holds its own copy
// (2.25 MiB) until the process ends.  In host code the array's NAME is only a stand-in for the symbol: the device
// address is looked up once per device (hipGetSymbolAddress with that device current) and only that pointer is
// ever handed to a kernel.  The first launch writes every partial the second one reads, so nothing has to be
// cleared between calls; two calls in flight on one GPU at the same time share the array and race.
// ------------------------------------------------------------------------------------------------------------
static __device__ __align__(256) float scratch[(size_t)HK_COLSUM_MAX_SPLIT * HK_COLSUM_STATIC_MAX_C];

#define HK_MAX_DEVICES 64
static std::mutex g_scratch_mu;
static float* g_scratch_ptr[HK_MAX_DEVICES];   // device address of `scratch` on device i, nullptr = not looked up yet

extern "C" int hk_colsum_static(const void* x, int32_t dtype, int64_t ld, int64_t R, int32_t C, float* out,
                                void* stream) {
    HK_REQUIRE(x && out && R > 0 && C > 0 && ld >= C, "hk_colsum_static: bad arguments");
    HK_REQUIRE(dtype == HK_DT_BF16 || dtype == HK_DT_F32, "hk_colsum_static: unknown dtype %d", dtype);
    HK_REQUIRE(C <= HK_COLSUM_STATIC_MAX_C, "hk_colsum_static: C=%d exceeds the static scratch (at most %d columns); "
               "use hk_colsum with a caller-provided workspace", C, HK_COLSUM_STATIC_MAX_C);
    int dev = -1;
    HK_REQUIRE(hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < HK_MAX_DEVICES,
               "hk_colsum_static: no current HIP device (or device index >= %d)", HK_MAX_DEVICES);
    float* ws;
    {
        std::lock_guard<std::mutex> lock(g_scratch_mu);
        ws = g_scratch_ptr[dev];
        if (!ws) {
            void* p = nullptr;
            const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
            HK_REQUIRE(e == hipSuccess && p, "hk_colsum_static: hipGetSymbolAddress failed on device %d: %s", dev,
                       hipGetErrorString(e));
            ws = g_scratch_ptr[dev] = static_cast<float*>(p);
        }
    }
    return colsum_launch(x, dtype, ld, R, C, out, ws, reinterpret_cast<hipStream_t>(stream), "hk_colsum_static");
}

extern "C" int hk_layernorm_bwd(const float* dy, const hk_rows* dy_rows, const float* z, const float* gamma,
                                const hk_dropout* drop, const hk_dropout* post_drop, float* dz, void* dt_bf16,
                                float* dgamma, float* dbeta, float* ws, int64_t rows, int32_t D, float eps,
                                void* stream) {
    HK_REQUIRE(dy && z && gamma && dgamma && dbeta && ws && (dz || dt_bf16), "hk_layernorm_bwd: null pointer");
    HK_REQUIRE(rows > 0 && D > 0 && D % 4 == 0 && D <= 1024, "hk_layernorm_bwd: bad shape rows=%ld D=%d", (long)rows, D);
    HK_REQUIRE((double)rows * D < 4294967296.0, "hk_layernorm_bwd: rows*D must stay below 2^32 (dropout counter)");
    HK_REQUIRE(dbeta == dgamma + D, "hk_layernorm_bwd: dbeta must directly follow dgamma (one [2,D] block)");
    HK_REQUIRE(drop_ok(drop) && drop_ok(post_drop), "hk_layernorm_bwd: dropout probabilities must be in [0,1)");
    HK_REQUIRE(((uintptr_t)dy | (uintptr_t)z | (uintptr_t)gamma | (uintptr_t)dz | (uintptr_t)dt_bf16 | (uintptr_t)ws) % 16 == 0,
               "hk_layernorm_bwd: operands must be 16-byte aligned");
    const RowMapD dm = make_map(dy_rows, D);
    HK_REQUIRE(dm.ld % 4 == 0 && dm.bs % 4 == 0 && dm.ld >= D, "hk_layernorm_bwd: bad row mapping");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    int64_t nb = (rows + 3) / 4;
    if (nb > HK_LNBWD_MAX_BLOCKS) nb = HK_LNBWD_MAX_BLOCKS;
    hipLaunchKernelGGL(ln_bwd_kernel, dim3((unsigned)nb), dim3(256), 0, s, dy, dm, z, gamma, make_drop(drop),
                       make_drop(post_drop), dz, reinterpret_cast<__bf16*>(dt_bf16), ws, rows, D, eps);
    HK_CHECK_LAUNCH("hk_layernorm_bwd");
    // ws = [nb][2*D] partials -> dgamma | dbeta (contiguous [2*D])
    hipLaunchKernelGGL((colsum_kernel<false>), dim3((2 * D + 63) / 64, 1), dim3(256), 0, s, (const void*)ws,
                       (int64_t)2 * D, nb, 2 * D, dgamma);
    HK_CHECK_LAUNCH("hk_layernorm_bwd(reduce)");
    return HK_OK;
}

extern "C" int hk_transpose_bf16(const void* x, int32_t in_dtype, const hk_rows* x_rows, int64_t R, int32_t C,
                                 void* out, int64_t ldo, void* stream) {
    HK_REQUIRE(x && out && R > 0 && C > 0, "hk_transpose_bf16: bad arguments");
    HK_REQUIRE(in_dtype == HK_DT_BF16 || in_dtype == HK_DT_F32, "hk_transpose_bf16: unknown dtype %d", in_dtype);
    const int64_t Rpad = (R + 63) / 64 * 64;
    HK_REQUIRE(ldo >= Rpad, "hk_transpose_bf16: ldo=%ld must cover R rounded up to 64 (%ld)", (long)ldo, (long)Rpad);
    const RowMapD xm = make_map(x_rows, C);
    HK_REQUIRE(xm.ld >= C, "hk_transpose_bf16: bad row mapping");
    const dim3 grid((unsigned)(Rpad / 64), (C + 63) / 64);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (in_dtype == HK_DT_BF16)
        hipLaunchKernelGGL((transpose_kernel<true>), grid, dim3(256), 0, s, x, xm, R, C, reinterpret_cast<__bf16*>(out), ldo, Rpad);
    else
        hipLaunchKernelGGL((transpose_kernel<false>), grid, dim3(256), 0, s, x, xm, R, C, reinterpret_cast<__bf16*>(out), ldo, Rpad);
    HK_CHECK_LAUNCH("hk_transpose_bf16");
    return HK_OK;
}

extern "C" int hk_transpose_f32_bf16_batch(const hk_transpose_job* jobs_dev, int32_t n_jobs, int32_t total_tiles, void* stream) {
    HK_REQUIRE(jobs_dev && n_jobs > 0 && total_tiles > 0, "hk_transpose_f32_bf16_batch: bad arguments");
    HK_REQUIRE((uintptr_t)jobs_dev % 8 == 0, "hk_transpose_f32_bf16_batch: job table misaligned");
    hipLaunchKernelGGL(transpose_batch_kernel, dim3((unsigned)total_tiles), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       jobs_dev, n_jobs);
    HK_CHECK_LAUNCH("hk_transpose_f32_bf16_batch");
    return HK_OK;
}

extern "C" int hk_gelu_bf16(const void* u, void* g, int64_t n, void* stream) {
    HK_REQUIRE(u && g && n > 0 && n % 8 == 0, "hk_gelu_bf16: n must be a positive multiple of 8");
    HK_REQUIRE(((uintptr_t)u | (uintptr_t)g) % 16 == 0, "hk_gelu_bf16: operands must be 16-byte aligned");
    hipLaunchKernelGGL(gelu_fwd_kernel, dim3((unsigned)((n / 8 + 255) / 256)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), reinterpret_cast<const __bf16*>(u),
                       reinterpret_cast<__bf16*>(g), n);
    HK_CHECK_LAUNCH("hk_gelu_bf16");
    return HK_OK;
}

extern "C" int hk_gelu_bwd_bf16(const void* dg, const void* u, void* du, int64_t n, void* stream) {
    HK_REQUIRE(dg && u && du && n > 0 && n % 8 == 0, "hk_gelu_bwd_bf16: n must be a positive multiple of 8");
    HK_REQUIRE(((uintptr_t)u | (uintptr_t)dg | (uintptr_t)du) % 16 == 0, "hk_gelu_bwd_bf16: operands must be 16-byte aligned");
    hipLaunchKernelGGL(gelu_bwd_kernel, dim3((unsigned)((n / 8 + 255) / 256)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), reinterpret_cast<const __bf16*>(dg),
                       reinterpret_cast<const __bf16*>(u), reinterpret_cast<__bf16*>(du), n);
    HK_CHECK_LAUNCH("hk_gelu_bwd_bf16");
    return HK_OK;
}

extern "C" int hk_qformer_embed_gather(const float* query_tokens, const float* word_emb, const float* pos_emb,
                                       const int64_t* input_ids, float* z, int32_t B, int32_t nq, int32_t Lt,
                                       int32_t d, int32_t vocab, void* stream) {
    HK_REQUIRE(query_tokens && z, "hk_qformer_embed_gather: null pointer");
    HK_REQUIRE(Lt == 0 || (word_emb && pos_emb && input_ids), "hk_qformer_embed_gather: text tables missing");
    HK_REQUIRE(B > 0 && nq >= 0 && Lt >= 0 && nq + Lt > 0 && d > 0 && d % 4 == 0 && vocab > 0,
               "hk_qformer_embed_gather: bad shape");
    const int64_t n4 = (int64_t)B * (nq + Lt) * (d / 4);
    hipLaunchKernelGGL(embed_gather_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), query_tokens, word_emb, pos_emb, input_ids, z, B, nq, Lt, d,
                       vocab);
    HK_CHECK_LAUNCH("hk_qformer_embed_gather");
    return HK_OK;
}

extern "C" int hk_qformer_embed_bwd(const float* dz, const int64_t* input_ids, float* dquery, float* dword, float* dpos,
                                    int32_t B, int32_t nq, int32_t Lt, int32_t d, int32_t vocab, void* stream) {
    HK_REQUIRE(dz && dquery, "hk_qformer_embed_bwd: null pointer");
    HK_REQUIRE(Lt == 0 || (input_ids && dword && dpos), "hk_qformer_embed_bwd: text gradients missing");
    HK_REQUIRE(B > 0 && nq >= 0 && Lt >= 0 && nq + Lt > 0 && d > 0 && vocab > 0, "hk_qformer_embed_bwd: bad shape");
    const int64_t n = (int64_t)(nq + Lt) * d;
    hipLaunchKernelGGL(embed_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), dz, input_ids, dquery, dword, dpos, B, nq, Lt, d, vocab);
    HK_CHECK_LAUNCH("hk_qformer_embed_bwd");
    return HK_OK;
}

extern "C" uint32_t hk_dropout_keep_u32(uint32_t seed, uint32_t site, uint32_t idx) { return fmix32(idx ^ site_key(seed, site)); }
