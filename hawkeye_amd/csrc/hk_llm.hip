// Row f1: the memory-bound pieces of the Vicuna/LLaMA prefill that consumes encode_img's output --
// RMSNorm, rotary position embedding, SwiGLU gate, and the token-embedding gather fused with the splice of the
// video tokens.  The GEMMs go through hk_gemm_bf16, the causal head-128 attention is in hk_attn_causal.hip.
#include "hk_common.h"

namespace {

// LlamaRMSNorm (models/blip2/modeling_llama.py:57-75): y = w * (x * rsqrt(mean(x^2) + eps)), fp32 statistics.
// One wave per row, the row in registers (D = 256 * NCH), 16-byte accesses.
template <int NCH, bool IN_BF16>
__global__ __launch_bounds__(256) void hk_rmsnorm_kernel(const void* __restrict__ x, const float* __restrict__ w,
                                                         __bf16* __restrict__ out, int64_t rows, int D, float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    f32x4 v[NCH];
    float ss = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int i = c * 256 + lane * 4;
        if (IN_BF16) {
            const u32x2 u = *reinterpret_cast<const u32x2*>(reinterpret_cast<const __bf16*>(x) + row * D + i);
            v[c] = f32x4{bf16_lo(u[0]), bf16_hi(u[0]), bf16_lo(u[1]), bf16_hi(u[1])};
        } else {
            v[c] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(x) + row * D + i);
        }
        ss += v[c][0] * v[c][0] + v[c][1] * v[c][1] + v[c][2] * v[c][2] + v[c][3] * v[c][3];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
    const float rinv = rsqrtf(ss / (float)D + eps);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int i = c * 256 + lane * 4;
        const f32x4 g = *reinterpret_cast<const f32x4*>(w + i);
        u32x2 o;
        o[0] = pack_bf16x2(g[0] * (v[c][0] * rinv), g[1] * (v[c][1] * rinv));
        o[1] = pack_bf16x2(g[2] * (v[c][2] * rinv), g[3] * (v[c][3] * rinv));
        *reinterpret_cast<u32x2*>(out + row * D + i) = o;
    }
}

// The same norm on the sum of split-K slices (round 3, skinny prompts): x = sum_p parts[p] (slice 0 already carries the
// residual stream, hk_gemm_desc.split_k), written back as the new fp32 residual stream, and y = RMSNorm(x) * w in bf16 --
// the residual add of o_proj / down_proj (modeling_llama.py:297-307) and the norm that follows it in one pass.
// One WORKGROUP per row (these launches have a few hundred rows: a wave per row left three quarters of the chip idle and each wave
// a chain of n_parts x D / 256 dependent loads): thread t owns elements 1024 c + 4 t .. + 3, the slices' loads of a chunk are issued
// together, the sum of squares meets through LDS.
template <int NCH>   // chunks of 1024 elements per row (D <= 1024 NCH)
__global__ __launch_bounds__(256) void hk_rmsnorm_parts_kernel(const float* __restrict__ parts, int n_parts, int64_t stride,
                                                               float* __restrict__ x, const float* __restrict__ w,
                                                               __bf16* __restrict__ out, int64_t rows, int D, float eps) {
    __shared__ float red[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t row = blockIdx.x;
    f32x4 v[NCH];
    float ss = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int i = c * 1024 + tid * 4;
        v[c] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (i < D) {
            const float* src = parts + row * D + i;
            f32x4 a = *reinterpret_cast<const f32x4*>(src);
            int q = 1;
            for (; q + 2 < n_parts; q += 3) {   // slice order kept: ((s0 + s1) + s2) + s3 ...
                const f32x4 b0 = *reinterpret_cast<const f32x4*>(src + q * stride);
                const f32x4 b1 = *reinterpret_cast<const f32x4*>(src + (q + 1) * stride);
                const f32x4 b2 = *reinterpret_cast<const f32x4*>(src + (q + 2) * stride);
                a = ((a + b0) + b1) + b2;
            }
            for (; q < n_parts; ++q) a += *reinterpret_cast<const f32x4*>(src + q * stride);
            v[c] = a;
            *reinterpret_cast<f32x4*>(x + row * D + i) = a;
            ss += a[0] * a[0] + a[1] * a[1] + a[2] * a[2] + a[3] * a[3];
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
    if (lane == 0) red[wave] = ss;
    __syncthreads();
    const float rinv = rsqrtf(((red[0] + red[1]) + (red[2] + red[3])) / (float)D + eps);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int i = c * 1024 + tid * 4;
        if (i < D) {
            const f32x4 g = *reinterpret_cast<const f32x4*>(w + i);
            u32x2 o;
            o[0] = pack_bf16x2(g[0] * (v[c][0] * rinv), g[1] * (v[c][1] * rinv));
            o[1] = pack_bf16x2(g[2] * (v[c][2] * rinv), g[3] * (v[c][3] * rinv));
            *reinterpret_cast<u32x2*>(out + row * D + i) = o;
        }
    }
}

// apply_rotary_pos_emb (modeling_llama.py:116-123) on the q and k thirds of a fused [rows, 3*H*hd] bf16 buffer, in
// place:  x' = x * cos + rotate_half(x) * sin  with rotate_half(x) = cat(-x[hd/2:], x[:hd/2]) (:108-113), cos/sin
// rows of the cached fp32 table (:82-89) selected by position_ids.  q is also multiplied by qscale (the softmax
// scale * log2 e the attention kernel expects), which the reference applies after QK^T (:195).
// Thread -> 8 consecutive pair indices j of one (row, head, q|k): elements j and j + hd/2.
__global__ __launch_bounds__(256) void hk_rope_kernel(__bf16* __restrict__ qkv, const float* __restrict__ cos_t,
                                                      const float* __restrict__ sin_t, const int32_t* __restrict__ pos,
                                                      int64_t rows, int H, int hd, int S, float qscale) {
    const int half = hd >> 1, per_head = half >> 3;  // threads per (row, head, q|k)
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * 2 * H * per_head) return;
    const int j8 = (int)(i % per_head);
    int64_t rest = i / per_head;
    const int head = (int)(rest % H);
    rest /= H;
    const int which = (int)(rest & 1);  // 0 = q, 1 = k
    const int64_t row = rest >> 1;
    const int p = pos ? pos[row] : (int)(row % S);
    __bf16* base = qkv + row * (int64_t)(3 * H * hd) + (int64_t)which * H * hd + head * hd + j8 * 8;
    const u32x4 a = *reinterpret_cast<const u32x4*>(base);
    const u32x4 b = *reinterpret_cast<const u32x4*>(base + half);
    const float* cp = cos_t + (int64_t)p * half + j8 * 8;
    const float* sp = sin_t + (int64_t)p * half + j8 * 8;
    const float sc = which == 0 ? qscale : 1.0f;
    u32x4 oa, ob;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float x1[2] = {bf16_lo(a[k]), bf16_hi(a[k])}, x2[2] = {bf16_lo(b[k]), bf16_hi(b[k])};
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
    *reinterpret_cast<u32x4*>(base) = oa;
    *reinterpret_cast<u32x4*>(base + half) = ob;
}

// LlamaMLP gate (modeling_llama.py:139-140): out = silu(gate) * up on a fused [rows, 2*I] buffer (gate | up).
__global__ __launch_bounds__(256) void hk_silu_mul_kernel(const __bf16* __restrict__ gu, __bf16* __restrict__ out,
                                                          int64_t rows, int I) {
    const int per = I >> 3;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * per) return;
    const int64_t row = i / per;
    const int c = (int)(i - row * per) * 8;
    const u32x4 g = *reinterpret_cast<const u32x4*>(gu + row * 2 * I + c);
    const u32x4 u = *reinterpret_cast<const u32x4*>(gu + row * 2 * I + I + c);
    u32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float g0 = bf16_lo(g[k]), g1 = bf16_hi(g[k]);
        const float s0 = g0 / (1.0f + __expf(-g0)), s1 = g1 / (1.0f + __expf(-g1));
        o[k] = pack_bf16x2(s0 * bf16_lo(u[k]), s1 * bf16_hi(u[k]));
    }
    *reinterpret_cast<u32x4*>(out + row * I + c) = o;
}

// embed_tokens (modeling_llama.py:444-455 with num_frame_tokens = 0) fused with the splice of the video tokens
// (tasks/test.py:79-97 get_context_emb / hawkeye_it.py:287-375): row r of the sequence buffer is table[src[r]] when
// src[r] >= 0, else video row (-src[r] - 1).  fp32 out (the residual stream).  Thread -> 8 elements.
template <bool VID_BF16>
__global__ __launch_bounds__(256) void hk_embed_splice_kernel(const int64_t* __restrict__ src,
                                                              const __bf16* __restrict__ table,
                                                              const void* __restrict__ video, float* __restrict__ out,
                                                              int64_t rows, int D) {
    const int per = D >> 3;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * per) return;
    const int64_t row = i / per;
    const int c = (int)(i - row * per) * 8;
    const int64_t s = src[row];
    f32x4 lo, hi;
    if (s >= 0 || VID_BF16) {
        const __bf16* p = s >= 0 ? table + s * D + c : reinterpret_cast<const __bf16*>(video) + (-s - 1) * D + c;
        const u32x4 u = *reinterpret_cast<const u32x4*>(p);
        lo = f32x4{bf16_lo(u[0]), bf16_hi(u[0]), bf16_lo(u[1]), bf16_hi(u[1])};
        hi = f32x4{bf16_lo(u[2]), bf16_hi(u[2]), bf16_lo(u[3]), bf16_hi(u[3])};
    } else {
        const float* p = reinterpret_cast<const float*>(video) + (-s - 1) * D + c;
        lo = *reinterpret_cast<const f32x4*>(p);
        hi = *reinterpret_cast<const f32x4*>(p + 4);
    }
    *reinterpret_cast<f32x4*>(out + row * D + c) = lo;
    *reinterpret_cast<f32x4*>(out + row * D + c + 4) = hi;
}

}  // namespace

extern "C" int hk_rmsnorm(const void* x, int32_t in_dtype, const float* weight, void* out_bf16, int64_t rows,
                          int32_t D, float eps, void* stream) {
    HK_REQUIRE(x && weight && out_bf16, "hk_rmsnorm: null pointer");
    HK_REQUIRE(rows > 0 && D > 0 && D % 256 == 0 && D <= 8192, "hk_rmsnorm: D=%d must be a multiple of 256, <= 8192", D);
    HK_REQUIRE(in_dtype == HK_DT_BF16 || in_dtype == HK_DT_F32, "hk_rmsnorm: unknown in_dtype %d", in_dtype);
    HK_REQUIRE(((uintptr_t)x | (uintptr_t)weight | (uintptr_t)out_bf16) % 16 == 0, "hk_rmsnorm: operands must be 16-byte aligned");
    HK_REQUIRE((rows + 3) / 4 < (1L << 31), "hk_rmsnorm: too many rows");
    const dim3 grid((unsigned)((rows + 3) / 4));
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    __bf16* o = reinterpret_cast<__bf16*>(out_bf16);
    const bool b = in_dtype == HK_DT_BF16;
#define HK_RMS(N)                                                                                        \
    case N:                                                                                              \
        if (b) hipLaunchKernelGGL((hk_rmsnorm_kernel<N, true>), grid, dim3(256), 0, s, x, weight, o, rows, D, eps);  \
        else hipLaunchKernelGGL((hk_rmsnorm_kernel<N, false>), grid, dim3(256), 0, s, x, weight, o, rows, D, eps); \
        break;
    switch (D / 256) {
        HK_RMS(1) HK_RMS(2) HK_RMS(3) HK_RMS(4) HK_RMS(8) HK_RMS(12) HK_RMS(16) HK_RMS(20) HK_RMS(32)
        default:
            hk_set_error("hk_rmsnorm: D=%d is not an instantiated width (256,512,768,1024,2048,3072,4096,5120,8192)", D);
            return HK_EINVAL;
    }
#undef HK_RMS
    HK_CHECK_LAUNCH("hk_rmsnorm");
    return HK_OK;
}

extern "C" int hk_rmsnorm_add_parts(const float* parts, int32_t n_parts, int64_t part_stride, float* x, const float* weight,
                                    void* out_bf16, int64_t rows, int32_t D, float eps, void* stream) {
    HK_REQUIRE(parts && x && weight && out_bf16, "hk_rmsnorm_add_parts: null pointer");
    HK_REQUIRE(n_parts >= 1 && n_parts <= 64 && part_stride >= rows * D && part_stride % 4 == 0,
               "hk_rmsnorm_add_parts: 1..64 slices, stride >= rows * D and a multiple of 4");
    HK_REQUIRE(rows > 0 && D > 0 && D % 256 == 0 && D <= 8192, "hk_rmsnorm_add_parts: D=%d must be a multiple of 256, <= 8192", D);
    HK_REQUIRE(((uintptr_t)parts | (uintptr_t)x | (uintptr_t)weight | (uintptr_t)out_bf16) % 16 == 0,
               "hk_rmsnorm_add_parts: operands must be 16-byte aligned");
    HK_REQUIRE(rows < (1L << 31), "hk_rmsnorm_add_parts: too many rows");
    const dim3 grid((unsigned)rows);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    __bf16* o = reinterpret_cast<__bf16*>(out_bf16);
    switch ((D + 1023) / 1024) {
        case 1: hipLaunchKernelGGL((hk_rmsnorm_parts_kernel<1>), grid, dim3(256), 0, s, parts, n_parts, part_stride, x, weight, o, rows, D, eps); break;
        case 2: hipLaunchKernelGGL((hk_rmsnorm_parts_kernel<2>), grid, dim3(256), 0, s, parts, n_parts, part_stride, x, weight, o, rows, D, eps); break;
        case 3: case 4: hipLaunchKernelGGL((hk_rmsnorm_parts_kernel<4>), grid, dim3(256), 0, s, parts, n_parts, part_stride, x, weight, o, rows, D, eps); break;
        default: hipLaunchKernelGGL((hk_rmsnorm_parts_kernel<8>), grid, dim3(256), 0, s, parts, n_parts, part_stride, x, weight, o, rows, D, eps); break;
    }
    HK_CHECK_LAUNCH("hk_rmsnorm_add_parts");
    return HK_OK;
}

extern "C" int hk_rope_bf16(void* qkv, const float* cos_table, const float* sin_table, const int32_t* position_ids,
                            int64_t rows, int32_t H, int32_t head_dim, int32_t seq_len, int32_t table_rows,
                            float q_scale, void* stream) {
    HK_REQUIRE(qkv && cos_table && sin_table, "hk_rope_bf16: null pointer");
    HK_REQUIRE(rows > 0 && H > 0 && head_dim >= 16 && head_dim % 16 == 0, "hk_rope_bf16: bad shape rows=%ld H=%d hd=%d",
               (long)rows, H, head_dim);
    HK_REQUIRE(position_ids != nullptr || (seq_len > 0 && seq_len <= table_rows),
               "hk_rope_bf16: seq_len=%d exceeds the cos/sin table (%d rows)", seq_len, table_rows);
    HK_REQUIRE(((uintptr_t)qkv | (uintptr_t)cos_table | (uintptr_t)sin_table) % 16 == 0, "hk_rope_bf16: operands must be 16-byte aligned");
    const int64_t n = rows * 2 * H * (head_dim / 16);
    HK_REQUIRE((n + 255) / 256 < (1L << 31), "hk_rope_bf16: problem too large");
    hipLaunchKernelGGL(hk_rope_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       reinterpret_cast<__bf16*>(qkv), cos_table, sin_table, position_ids, rows, H, head_dim, seq_len, q_scale);
    HK_CHECK_LAUNCH("hk_rope_bf16");
    return HK_OK;
}

extern "C" int hk_silu_mul_bf16(const void* gate_up, void* out, int64_t rows, int32_t I, void* stream) {
    HK_REQUIRE(gate_up && out, "hk_silu_mul_bf16: null pointer");
    HK_REQUIRE(rows > 0 && I > 0 && I % 8 == 0, "hk_silu_mul_bf16: I=%d must be a multiple of 8", I);
    HK_REQUIRE(((uintptr_t)gate_up | (uintptr_t)out) % 16 == 0, "hk_silu_mul_bf16: operands must be 16-byte aligned");
    const int64_t n = rows * (I / 8);
    HK_REQUIRE((n + 255) / 256 < (1L << 31), "hk_silu_mul_bf16: problem too large");
    hipLaunchKernelGGL(hk_silu_mul_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       reinterpret_cast<const __bf16*>(gate_up), reinterpret_cast<__bf16*>(out), rows, I);
    HK_CHECK_LAUNCH("hk_silu_mul_bf16");
    return HK_OK;
}

extern "C" int hk_embed_splice(const int64_t* src, const void* table_bf16, const void* video, int32_t video_dtype,
                               float* out, int64_t rows, int32_t D, void* stream) {
    HK_REQUIRE(src && table_bf16 && out, "hk_embed_splice: null pointer");
    HK_REQUIRE(rows > 0 && D > 0 && D % 8 == 0, "hk_embed_splice: D=%d must be a multiple of 8", D);
    HK_REQUIRE(video_dtype == HK_DT_BF16 || video_dtype == HK_DT_F32, "hk_embed_splice: unknown video_dtype %d", video_dtype);
    HK_REQUIRE(((uintptr_t)table_bf16 | (uintptr_t)video | (uintptr_t)out) % 16 == 0, "hk_embed_splice: operands must be 16-byte aligned");
    const int64_t n = rows * (D / 8);
    HK_REQUIRE((n + 255) / 256 < (1L << 31), "hk_embed_splice: problem too large");
    const dim3 grid((unsigned)((n + 255) / 256));
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (video_dtype == HK_DT_BF16)
        hipLaunchKernelGGL((hk_embed_splice_kernel<true>), grid, dim3(256), 0, s, src, reinterpret_cast<const __bf16*>(table_bf16), video, out, rows, D);
    else
        hipLaunchKernelGGL((hk_embed_splice_kernel<false>), grid, dim3(256), 0, s, src, reinterpret_cast<const __bf16*>(table_bf16), video, out, rows, D);
    HK_CHECK_LAUNCH("hk_embed_splice");
    return HK_OK;
}
