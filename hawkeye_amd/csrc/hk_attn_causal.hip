// Causal flash attention forward for the LLaMA prefill (row f1), head size HD = 128 (Vicuna-7B) or 64, bf16 in/out,
// fp32 online softmax.  Same transposed formulation as hk_attn.hip (read its header first): one workgroup = 4 waves
// = 128 query rows of one (batch, head), 64-key tiles through LDS, S^T = K . Q^T and O^T += V^T . P^T with
// v_mfma_f32_32x32x16_bf16, the S^T accumulator re-used as the B operand of the second product, V^T gathered with
// ds_read_b64_tr_b16.  Differences:
//   * HD is a template parameter (LDS rows of 2*HD bytes, HD/16 k-steps for S^T, HD/32 accumulator blocks for O^T);
//   * CAUSAL: query q sees keys <= q + (Nk - Nq) (LlamaModel._prepare_decoder_attention_mask / _make_causal_mask,
//     models/blip2/modeling_llama.py:25-47,457-480); key tiles entirely above the workgroup's diagonal are never
//     loaded, the mask is applied only on tiles that cross a wave's diagonal;
//   * when the running maximum moves, the tile's scores are RECOMPUTED against the new maximum instead of shifted,
//     so rows whose first tiles are fully masked (left padding, bias ~ -1e4) lose no precision;
//   * register-staged loads, one workgroup per SIMD set: attention is ~1.3 % of the prefill FLOPs at S = 352, so this
//     kernel is written for correctness and reasonable speed, not tuned like the ViT one.
// LDS swizzle (16-byte chunk c of key row r): K at c ^ (r & (CH-1)) with CH = HD/8 chunks per row -> conflict-free
// ds_read_b128 of 16 rows x one chunk; V unswizzled.
#include <type_traits>
#include "hk_common.h"

namespace {

struct CausalParams {
    const __bf16* Q;
    const __bf16* K;
    const __bf16* V;
    __bf16* O;
    const float* key_bias;
    int B, H, Nq, Nk;
    int64_t ldq, ldk, ldv, ldo, bsq, bsk, bsv, bso;
    int nqb, causal;
};

#ifndef HK_CAUSAL_OCC
#define HK_CAUSAL_OCC 2   // waves per SIMD the register allocation targets (2: two workgroups per CU)
#endif
#ifndef HK_CAUSAL_LPT
#define HK_CAUSAL_LPT 1
#endif
constexpr int QB = 128;
constexpr int KT = 64;
constexpr float NEG_BIG = -1.0e30f;

__device__ __forceinline__ void half_swap(float x, float& a, float& b) {
    a = x;
    b = x;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}

template <int HD, bool HAS_BIAS>
__global__ __launch_bounds__(256, HK_CAUSAL_OCC) void hk_attn_causal_kernel(const CausalParams p) {
    constexpr int ROWB = HD * 2;       // bytes per key row in LDS
    constexpr int CH = HD / 8;         // 16-byte chunks per row
    constexpr int NKS = HD / 16;       // k-steps of S^T
    constexpr int NDT = HD / 32;       // 32-dim blocks of O^T
    constexpr int PASSES = KT * CH / 256;  // 16-byte loads per thread per operand per tile
    constexpr int STAGE = 2 * KT * ROWB;
    extern __shared__ __attribute__((aligned(16))) char smem[];  // [2 stages][K | V][64 keys][ROWB], then [2][64] bias

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int r = lane & 31;
    const int h = lane >> 5;

#if HK_CAUSAL_LPT
    // longest-first over the whole grid: every (batch, head)'s last query block is dispatched before any shorter
    // one, so the short blocks fill the tail of the launch. A (batch, head)'s blocks sit nbh apart in blockIdx,
    // a multiple of 8 for the shapes served, so they share an XCD's L2 for K and V.
    const int nbh = p.B * p.H;
    const int rank = blockIdx.x / nbh;
    const int bh = blockIdx.x - rank * nbh;
    const int qb = p.nqb - 1 - rank;
#else
    const int bh = blockIdx.x / p.nqb;
    const int qb = p.nqb - 1 - (blockIdx.x - bh * p.nqb);   // longest (last) query blocks first
#endif
    const int b = bh / p.H;
    const int head = bh - b * p.H;
    const int q0 = qb * QB + wave * 32;
    const int off = p.Nk - p.Nq;       // causal: key <= q + off

    bf16x8 qf[NKS];
    {
        const int qrow = min(q0 + r, p.Nq - 1);
        const __bf16* qp = p.Q + (int64_t)b * p.bsq + (int64_t)qrow * p.ldq + head * HD + 8 * h;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(qp + 16 * ks);
    }

    // ---- tile loader: thread -> chunk (tid % CH) of keys (tid / CH) + i * (256 / CH) ----
    constexpr int KPP = 256 / CH;      // keys per pass
    const int lc = tid % CH, lkey = tid / CH;
    const __bf16* kbase = p.K + (int64_t)b * p.bsk + head * HD + lc * 8;
    const __bf16* vbase = p.V + (int64_t)b * p.bsv + head * HD + lc * 8;
    bf16x8 kr[PASSES], vr[PASSES];
    float br = 0.f;   // HAS_BIAS: threads 0-63 stage the tile's 64 key-bias values (x log2 e) through LDS
    float* bias_lds = reinterpret_cast<float*>(smem + 2 * STAGE);
    auto gload = [&](int tile) {
        if (HAS_BIAS && tid < KT)
            br = p.key_bias[(int64_t)b * p.Nk + min(tile * KT + tid, p.Nk - 1)] * HK_LOG2E;
#pragma unroll
        for (int i = 0; i < PASSES; ++i) {
            const int key = min(tile * KT + i * KPP + lkey, p.Nk - 1);
            kr[i] = *reinterpret_cast<const bf16x8*>(kbase + (int64_t)key * p.ldk);
            vr[i] = *reinterpret_cast<const bf16x8*>(vbase + (int64_t)key * p.ldv);
        }
    };
    auto lstore = [&](int stage) {
        char* ks = smem + stage * STAGE;
        char* vs = ks + KT * ROWB;
#pragma unroll
        for (int i = 0; i < PASSES; ++i) {
            const int key = i * KPP + lkey;
            *reinterpret_cast<bf16x8*>(ks + key * ROWB + ((lc ^ (key & (CH - 1))) << 4)) = kr[i];
            // V: 16-byte chunk c of key row k lives at c ^ ((k % (CH / 2)) << 1) -- the 8 key rows one transposed read touches
            // (4 h + vq, 256-byte rows: all on the same banks unswizzled, 48 % conflict cycles in the r03 counter pass) spread
            // over distinct 32-byte bank groups
            *reinterpret_cast<bf16x8*>(vs + key * ROWB + ((lc ^ ((key & (CH / 2 - 1)) << 1)) << 4)) = vr[i];
        }
        if (HAS_BIAS && tid < KT) bias_lds[stage * KT + tid] = br;
    };

    int koff[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) koff[ks] = r * ROWB + (((2 * ks + h) ^ (r & (CH - 1))) << 4);
    const int vq = (lane & 15) >> 2;
    int voff[NDT];
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) {
        const int chunk = 4 * dt + 2 * ((lane >> 4) & 1) + ((lane & 3) >> 1);     // (rows read are 4 h + vq + 8 n: same swizzle)
        voff[dt] = (4 * h + vq) * ROWB + ((chunk ^ (((4 * h + vq) & (CH / 2 - 1)) << 1)) << 4) + 8 * (lane & 1);
    }

    f32x16 ot[NDT];
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
        for (int i = 0; i < 16; ++i) ot[dt][i] = 0.f;
    float m_run = 0.f, l_run = 0.f;

    // key tiles this workgroup needs: up to the diagonal of its last query row
    const int qmax = min(qb * QB + QB - 1, p.Nq - 1);
    const int kend = p.causal ? min(p.Nk, qmax + off + 1) : p.Nk;
    const int NT = (kend + KT - 1) / KT;

    gload(0);
    lstore(0);
    __syncthreads();

    for (int t = 0; t < NT; ++t) {
        const int cur = t & 1;
        if (t + 1 < NT) gload(t + 1);
        const char* ks = smem + cur * STAGE;
        const char* vs = ks + KT * ROWB;
        f32x16 st[2];
        // does this tile need an element mask for this wave?  (ragged end, or crossing the wave's diagonal)
        const bool need_mask = (t + 1) * KT > p.Nk || (p.causal && (t + 1) * KT - 1 > q0 + off);

        auto scores = [&]() {
            const float negm = -m_run;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
                for (int i = 0; i < 16; ++i) st[kt][i] = negm;
#pragma unroll
                for (int s = 0; s < NKS; ++s) {
                    const bf16x8 kf = *reinterpret_cast<const bf16x8*>(ks + kt * 32 * ROWB + koff[s]);
                    st[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[s], st[kt], 0, 0, 0);
                }
            }
            if (HAS_BIAS) {   // register 4g + j of st[kt] <-> key 32 kt + 8 g + 4 h + j: four consecutive floats
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const f32x4 bv = *reinterpret_cast<const f32x4*>(bias_lds + cur * KT + 32 * kt + 8 * g + 4 * h);
#pragma unroll
                        for (int j = 0; j < 4; ++j) st[kt][4 * g + j] += bv[j];
                    }
            }
            if (need_mask) {
                const int lim = p.causal ? min(p.Nk - 1, q0 + r + off) : p.Nk - 1;   // last visible key of this row
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int key = t * KT + 32 * kt + (i & 3) + 8 * (i >> 2) + 4 * h;
                        if (key > lim) st[kt][i] = NEG_BIG;
                    }
            }
        };

        scores();
        {
            float mx = fmaxf(st[0][0], st[0][1]);
#pragma unroll
            for (int i = 2; i < 16; i += 2) mx = fmaxf(mx, fmaxf(st[0][i], st[0][i + 1]));
#pragma unroll
            for (int i = 0; i < 16; i += 2) mx = fmaxf(mx, fmaxf(st[1][i], st[1][i + 1]));
            float a_, b_;
            half_swap(mx, a_, b_);
            mx = fmaxf(a_, b_);
            // the first tile fixes the reference point; later it moves when some row outgrows it by 8 bits
            const bool move = t == 0 || __builtin_amdgcn_ballot_w64(mx > 8.0f) != 0;
            if (move) {
                const float delta = t == 0 ? fmaxf(mx, -1.0e20f) : fmaxf(mx, 0.f);
                const float alpha = t == 0 ? 0.f : __builtin_amdgcn_exp2f(-delta);
                m_run += delta;
                l_run *= alpha;
#pragma unroll
                for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
                    for (int i = 0; i < 16; ++i) ot[dt][i] *= alpha;
                scores();   // exact scores relative to the new reference (no large-offset subtraction)
            }
        }
        float rs = 0.f;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float e = __builtin_amdgcn_exp2f(st[kt][i]);
                st[kt][i] = e;
                rs += e;
            }
        l_run += rs;

        bf16x8 pf[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            u32x4 w;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                w[j] = pack_bf16x2(st[s >> 1][8 * (s & 1) + 2 * j], st[s >> 1][8 * (s & 1) + 2 * j + 1]);
            pf[s] = __builtin_bit_cast(bf16x8, w);
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) {
                const char* vp = vs + s * 16 * ROWB + voff[dt];
                const bf16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(vp));
                const bf16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(vp + 8 * ROWB));
                bf16x8 vf;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    vf[j] = v0[j];
                    vf[j + 4] = v1[j];
                }
                ot[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[s], ot[dt], 0, 0, 0);
            }
        }
        if (t + 1 < NT) lstore(cur ^ 1);
        __syncthreads();
    }

    float la, lb;
    half_swap(l_run, la, lb);
    const float inv = 1.0f / (la + lb);
    const int qrow = q0 + r;
    if (qrow < p.Nq) {
        __bf16* op = p.O + (int64_t)b * p.bso + (int64_t)qrow * p.ldo + head * HD + 4 * h;
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                u32x2 o;
                o[0] = pack_bf16x2(ot[dt][4 * g] * inv, ot[dt][4 * g + 1] * inv);
                o[1] = pack_bf16x2(ot[dt][4 * g + 2] * inv, ot[dt][4 * g + 3] * inv);
                *reinterpret_cast<u32x2*>(op + 32 * dt + 8 * g) = o;
            }
    }
}

template <int HD, bool BIAS>
int launch_causal(const CausalParams& p, hipStream_t s) {
    constexpr int LDS = 2 * 2 * KT * HD * 2 + (BIAS ? 2 * KT * 4 : 0);
    static std::atomic<unsigned long long> attr_done{0};   // per device
    const hipError_t e = hk_lds_attr_once(attr_done, reinterpret_cast<const void*>(&hk_attn_causal_kernel<HD, BIAS>), LDS);
    if (e != hipSuccess) {
        hk_set_error("hk_flash_attn_bf16: hipFuncSetAttribute(%d B LDS): %s", LDS, hipGetErrorString(e));
        return HK_ELAUNCH;
    }
    hipLaunchKernelGGL((hk_attn_causal_kernel<HD, BIAS>), dim3((unsigned)((long)p.nqb * p.H * p.B)), dim3(256), LDS, s, p);
    HK_CHECK_LAUNCH("hk_flash_attn_bf16");
    return HK_OK;
}

}  // namespace

// called by hk_flash_attn_bf16 (hk_attn.hip) for causal and/or head_dim == 128 problems, after its shape checks
int hk_attn_causal_dispatch(const hk_attn_desc* d, void* stream) {
    const int hd = d->head_dim == 0 ? 64 : d->head_dim;
    CausalParams p;
    p.Q = reinterpret_cast<const __bf16*>(d->Q);
    p.K = reinterpret_cast<const __bf16*>(d->K);
    p.V = reinterpret_cast<const __bf16*>(d->V);
    p.O = reinterpret_cast<__bf16*>(d->O);
    p.key_bias = d->key_bias;
    p.B = d->B;
    p.H = d->H;
    p.Nq = d->Nq;
    p.Nk = d->Nk;
    p.ldq = d->ldq;
    p.ldk = d->ldk;
    p.ldv = d->ldv;
    p.ldo = d->ldo;
    p.bsq = d->bsq;
    p.bsk = d->bsk;
    p.bsv = d->bsv;
    p.bso = d->bso;
    p.nqb = (d->Nq + QB - 1) / QB;
    p.causal = d->causal != 0;
    HK_REQUIRE(!p.causal || d->Nk >= d->Nq, "hk_flash_attn_bf16: causal needs Nk >= Nq (got %d < %d)", d->Nk, d->Nq);
    HK_REQUIRE((long)p.nqb * d->H * d->B < (1L << 31), "hk_flash_attn_bf16: grid too large");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (hd == 128) return d->key_bias ? launch_causal<128, true>(p, s) : launch_causal<128, false>(p, s);
    return d->key_bias ? launch_causal<64, true>(p, s) : launch_causal<64, false>(p, s);
}
