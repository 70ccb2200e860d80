// Input side of the path: the frame transform fused into the patch gather, and the
// sinusoid position table generated on the device.
#include "hk_common.h"

// the fp32 steps below restate the reference's separately rounded operations: no a*b+c contraction in this file
#pragma clang fp contract(off)

namespace {

// dataset/video_transforms.py:367-389 (ToTorchFormatTensor: uint8 HWC -> float CHW, .div(255)) and :94-107
// (GroupNormalize: t.sub_(mean).div_(std), fp32) fused with the patch gather of hk_patchify_bf16.
// Thread -> 8 consecutive pixels of one image row, all 3 channels (24 B in, 3 x 16 B out).  Each fp32 operation
// is the correctly rounded one the reference performs (true division, no reciprocal, no fma contraction).
__global__ __launch_bounds__(256) void hk_patchify_u8_kernel(const uint8_t* __restrict__ frames, __bf16* __restrict__ out,
                                                             int64_t n8, int H, int W, int P, float m0, float m1,
                                                             float m2, float s0, float s1, float s2) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n8) return;
    const int w8 = W / 8;
    const int x8 = (int)(i % w8);
    int64_t rest = i / w8;
    const int y = (int)(rest % H);
    const int64_t bt = rest / H;
    const uint8_t* src = frames + i * 24;   // (bt, y, x8*8, 0) in [B*T, H, W, 3]
    uint32_t w[6];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const u32x2 v = *reinterpret_cast<const u32x2*>(src + j * 8);
        w[2 * j] = v[0];
        w[2 * j + 1] = v[1];
    }
    const int gx = (x8 * 8) / P, kx = (x8 * 8) % P;
    const int gy = y / P, ky = y % P;
    const int64_t orow = (bt * (H / P) + gy) * (W / P) + gx;
    const float mean[3] = {m0, m1, m2}, sd[3] = {s0, s1, s2};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float v[8];
#pragma unroll
        for (int px = 0; px < 8; ++px) {
            const int byte = px * 3 + c;
            const float u = (float)((w[byte >> 2] >> ((byte & 3) * 8)) & 0xffu);
            v[px] = __fdiv_rn(__fsub_rn(__fdiv_rn(u, 255.0f), mean[c]), sd[c]);
        }
        u32x4 o;
        o[0] = pack_bf16x2(v[0], v[1]);
        o[1] = pack_bf16x2(v[2], v[3]);
        o[2] = pack_bf16x2(v[4], v[5]);
        o[3] = pack_bf16x2(v[6], v[7]);
        *reinterpret_cast<u32x4*>(out + orow * (3 * P * P) + (c * P + ky) * P + kx) = o;
    }
}

// models/blip2/vit.py:163-194 get_sinusoid_encoding_table: angle(p, j) = p / 10000^(2 (j/2) / d) in float64,
// sin on even j, cos on odd j, cast to fp32; when T != ckpt_frames the [ckpt, L, d] table is interpolated along
// the frame axis exactly as F.interpolate(mode='linear', align_corners=False) does in fp32 (vit.py:186).
__device__ __forceinline__ float sinusoid(int pos, int j, int d) {
    const double a = (double)pos / pow(10000.0, (double)(2 * (j >> 1)) / (double)d);
    return (float)((j & 1) ? cos(a) : sin(a));
}

__global__ __launch_bounds__(256) void hk_pos_embed_kernel(float* __restrict__ out, int T, int L, int D, int ckpt) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)T * L * D) return;
    const int j = (int)(i % D);
    const int l = (int)((i / D) % L);
    const int t = (int)(i / ((int64_t)D * L));
    if (ckpt <= 0 || ckpt == T) {
        out[i] = sinusoid(t * L + l, j, D);
        return;
    }
    const float scale = __fdiv_rn((float)ckpt, (float)T);
    // torch's CPU kernel evaluates these two steps fused (see hawkeye_amd/posembed.py): fma(scale, t + .5, -.5)
    float src = __fmaf_rn(scale, (float)t + 0.5f, -0.5f);
    if (src < 0.f) src = 0.f;
    const int i0 = (int)src;
    const int i1 = i0 < ckpt - 1 ? i0 + 1 : i0;
    const float l1 = __fsub_rn(src, (float)i0);
    const float l0 = __fsub_rn(1.0f, l1);
    const float v0 = sinusoid(i0 * L + l, j, D), v1 = sinusoid(i1 * L + l, j, D);
    out[i] = __fmaf_rn(l0, v0, __fmul_rn(l1, v1));
}

// Qformer.py:746-835 get_extended_attention_mask on cat(ones[B,nq], text_mask[B,Lt]) (hawkeye_it.py:250):
// additive key bias (1 - m) * -10000, fp32 [B, nq + Lt]; query columns are never masked.
__global__ __launch_bounds__(256) void hk_key_bias_kernel(const int64_t* __restrict__ mask, float* __restrict__ out,
                                                          int B, int nq, int Lt) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int S = nq + Lt;
    if (i >= B * S) return;
    const int b = i / S, s = i - b * S;
    out[i] = s < nq ? 0.f : (1.0f - (float)mask[(int64_t)b * Lt + (s - nq)]) * -10000.0f;
}

}  // namespace

extern "C" int hk_qformer_key_bias(const int64_t* text_mask, float* out, int32_t B, int32_t nq, int32_t Lt,
                                   void* stream) {
    HK_REQUIRE(out != nullptr && (Lt == 0 || text_mask != nullptr), "hk_qformer_key_bias: null pointer");
    HK_REQUIRE(B > 0 && nq >= 0 && Lt >= 0 && nq + Lt > 0 && (int64_t)B * (nq + Lt) < (1L << 31),
               "hk_qformer_key_bias: bad shape B=%d nq=%d Lt=%d", B, nq, Lt);
    const int n = B * (nq + Lt);
    hipLaunchKernelGGL(hk_key_bias_kernel, dim3((n + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       text_mask, out, B, nq, Lt);
    HK_CHECK_LAUNCH("hk_qformer_key_bias");
    return HK_OK;
}

extern "C" int hk_patchify_u8(const uint8_t* frames, void* out, int32_t B, int32_t T, int32_t H, int32_t W,
                              int32_t patch, const float* mean3, const float* std3, void* stream) {
    HK_REQUIRE(frames && out && mean3 && std3, "hk_patchify_u8: null pointer");
    HK_REQUIRE(B > 0 && T > 0 && patch >= 8 && patch % 8 == 0 && H % patch == 0 && W % patch == 0,
               "hk_patchify_u8: bad shape B=%d T=%d H=%d W=%d patch=%d", B, T, H, W, patch);
    HK_REQUIRE((uintptr_t)frames % 8 == 0 && (uintptr_t)out % 16 == 0, "hk_patchify_u8: frames must be 8-byte, out 16-byte aligned");
    for (int c = 0; c < 3; ++c) HK_REQUIRE(std3[c] != 0.f, "hk_patchify_u8: std[%d] is zero", c);
    const int64_t n8 = (int64_t)B * T * H * (W / 8);
    HK_REQUIRE((n8 + 255) / 256 < (1L << 31), "hk_patchify_u8: clip too large");
    hipLaunchKernelGGL(hk_patchify_u8_kernel, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), frames, reinterpret_cast<__bf16*>(out), n8, H, W, patch,
                       mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2]);
    HK_CHECK_LAUNCH("hk_patchify_u8");
    return HK_OK;
}

extern "C" int hk_pos_embed(float* out, int32_t T, int32_t L, int32_t D, int32_t ckpt_frames, void* stream) {
    HK_REQUIRE(out != nullptr, "hk_pos_embed: null pointer");
    HK_REQUIRE(T > 0 && L > 0 && D > 0 && D % 2 == 0, "hk_pos_embed: bad shape T=%d L=%d D=%d", T, L, D);
    const int64_t n = (int64_t)T * L * D;
    HK_REQUIRE((n + 255) / 256 < (1L << 31), "hk_pos_embed: table too large");
    hipLaunchKernelGGL(hk_pos_embed_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), out, T, L, D, ckpt_frames);
    HK_CHECK_LAUNCH("hk_pos_embed");
    return HK_OK;
}
