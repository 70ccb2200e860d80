// Row-wise, HBM-bound kernels: LayerNorm (wave per row, fp32 statistics), Q-Former embedding
// gather + LayerNorm, patch gather (fp32 clip -> bf16 GEMM operand), fp32 -> bf16 cast.
// All global accesses are 16 bytes per lane and coalesced along the row.
#include "hk_common.h"

namespace {

// A row of D = 4 * 64 * NCH (or fewer) floats spread over one wave: chunk c of lane l covers
// elements [(c * 64 + l) * 4, +4).
template <int NCH>
struct RowRegs {
    f32x4 v[NCH];
};

__device__ __forceinline__ float wave_sum(float x) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
    return x;
}

template <int NCH>
__device__ __forceinline__ void ln_finish(RowRegs<NCH>& x, int D, int lane, const float* gamma, const float* beta,
                                          float eps, __bf16* ob, float* of) {
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
        if ((c * 64 + lane) * 4 < D) s += x.v[c][0] + x.v[c][1] + x.v[c][2] + x.v[c][3];
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
        if ((c * 64 + lane) * 4 < D) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float d = x.v[c][e] - mean;
                q += d * d;
            }
        }
    const float rstd = rsqrtf(wave_sum(q) / (float)D + eps);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int i = (c * 64 + lane) * 4;
        if (i < D) {
            const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + i);
            const f32x4 bt = *reinterpret_cast<const f32x4*>(beta + i);
            f32x4 y;
#pragma unroll
            for (int e = 0; e < 4; ++e) y[e] = (x.v[c][e] - mean) * rstd * g[e] + bt[e];
            if (of) *reinterpret_cast<f32x4*>(of + i) = y;
            if (ob) {
                u32x2 o;
                o[0] = pack_bf16x2(y[0], y[1]);
                o[1] = pack_bf16x2(y[2], y[3]);
                *reinterpret_cast<u32x2*>(ob + i) = o;
            }
        }
    }
}

template <int NCH, bool IN_BF16>
__global__ __launch_bounds__(256) void hk_layernorm_kernel(const void* xin, const float* add, int n_add, int64_t add_stride,
                                                           const float* gamma, const float* beta, __bf16* out_b, float* out_f,
                                                           int64_t rows, int D, float eps, int64_t o_ld,
                                                           int64_t o_rpb, int64_t o_bs) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    RowRegs<NCH> x;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int i = (c * 64 + lane) * 4;
        if (i < D) {
            if (IN_BF16) {
                const u32x2 w = *reinterpret_cast<const u32x2*>(reinterpret_cast<const __bf16*>(xin) + row * D + i);
                x.v[c] = f32x4{bf16_lo(w[0]), bf16_hi(w[0]), bf16_lo(w[1]), bf16_hi(w[1])};
            } else {
                x.v[c] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(xin) + row * D + i);
            }
            // addends in index order: the sum of split-K partial products is the same bits on every run
            // (four loads in flight at a time: a row of 96-row problems is one latency-bound chain otherwise)
            for (int a = 0; a < n_add; a += 4) {
                f32x4 t[4];
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    t[u] = a + u < n_add ? *reinterpret_cast<const f32x4*>(add + (a + u) * add_stride + row * D + i)
                                         : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int u = 0; u < 4; ++u) x.v[c] += t[u];
            }
        }
    }
    int64_t oo = row * o_ld;
    if (o_rpb) {
        const int64_t bb = row / o_rpb;
        oo = bb * o_bs + (row - bb * o_rpb) * o_ld;
    }
    ln_finish<NCH>(x, D, lane, gamma, beta, eps, out_b ? out_b + oo : nullptr, out_f ? out_f + oo : nullptr);
}

// bf16 -> bf16 fast path (the ViT's residual stream): 16 bytes per lane per access, 8 elements per chunk.
template <int NCH8>
__global__ __launch_bounds__(256) void hk_ln_bf16_kernel(const __bf16* xin, const float* gamma, const float* beta,
                                                         __bf16* out, int64_t rows, int D, float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float v[NCH8][8];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NCH8; ++c) {
        const int i = (c * 64 + lane) * 8;
        if (i < D) {
            const u32x4 w = *reinterpret_cast<const u32x4*>(xin + row * D + i);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[c][2 * e] = bf16_lo(w[e]);
                v[c][2 * e + 1] = bf16_hi(w[e]);
                s += v[c][2 * e] + v[c][2 * e + 1];
            }
        }
    }
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < NCH8; ++c)
        if ((c * 64 + lane) * 8 < D) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float d = v[c][e] - mean;
                q += d * d;
            }
        }
    const float rstd = rsqrtf(wave_sum(q) / (float)D + eps);
#pragma unroll
    for (int c = 0; c < NCH8; ++c) {
        const int i = (c * 64 + lane) * 8;
        if (i < D) {
            const f32x4 g0 = *reinterpret_cast<const f32x4*>(gamma + i), g1 = *reinterpret_cast<const f32x4*>(gamma + i + 4);
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(beta + i), b1 = *reinterpret_cast<const f32x4*>(beta + i + 4);
            float y[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                y[e] = (v[c][e] - mean) * rstd * g0[e] + b0[e];
                y[4 + e] = (v[c][4 + e] - mean) * rstd * g1[e] + b1[e];
            }
            u32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = pack_bf16x2(y[2 * e], y[2 * e + 1]);
            *reinterpret_cast<u32x4*>(out + row * D + i) = o;
        }
    }
}

// (mean, rstd) of each bf16 row: the statistics half of hk_ln_bf16_kernel (same two-pass arithmetic, so a GEMM with
// the LayerNorm folded in sees the statistics the standalone LayerNorm would have used).
template <int NCH8>
__global__ __launch_bounds__(256) void hk_row_stats_kernel(const __bf16* xin, float* stats, int64_t rows, int D, float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float v[NCH8][8];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NCH8; ++c) {
        const int i = (c * 64 + lane) * 8;
        if (i < D) {
            const u32x4 w = *reinterpret_cast<const u32x4*>(xin + row * D + i);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[c][2 * e] = bf16_lo(w[e]);
                v[c][2 * e + 1] = bf16_hi(w[e]);
                s += v[c][2 * e] + v[c][2 * e + 1];
            }
        }
    }
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < NCH8; ++c)
        if ((c * 64 + lane) * 8 < D) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float d = v[c][e] - mean;
                q += d * d;
            }
        }
    const float rstd = rsqrtf(wave_sum(q) / (float)D + eps);
    if (lane == 0) {
        stats[2 * row] = mean;
        stats[2 * row + 1] = rstd;
    }
}

// Split-K tail of the proj / fc2 update of a SMALL call: x (bf16, in place) += sum of the fp32 slices (slice 0 carries
// the bias), rounded once; (mean, rstd) of the stored row go to `stats` for the GEMM that consumes x with its
// LayerNorm folded in.  Takes the place of hk_row_stats_finalize in the launch sequence.  Wave per row.
template <int NCH8>
__global__ __launch_bounds__(256) void hk_residual_add_parts_kernel(__bf16* x, const float* parts, int n_parts, int64_t stride,
                                                                   float* stats, int64_t rows, int D, float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float v[NCH8][8];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NCH8; ++c) {
        const int i = (c * 64 + lane) * 8;
        if (i < D) {
            const u32x4 w = *reinterpret_cast<const u32x4*>(x + row * D + i);
            f32x4 a0 = f32x4{bf16_lo(w[0]), bf16_hi(w[0]), bf16_lo(w[1]), bf16_hi(w[1])};
            f32x4 a1 = f32x4{bf16_lo(w[2]), bf16_hi(w[2]), bf16_lo(w[3]), bf16_hi(w[3])};
            for (int a = 0; a < n_parts; ++a) {      // slices in index order: the same bits on every run
                const float* pp = parts + a * stride + row * D + i;
                a0 += *reinterpret_cast<const f32x4*>(pp);
                a1 += *reinterpret_cast<const f32x4*>(pp + 4);
            }
            u32x4 o;
            o[0] = pack_bf16x2(a0[0], a0[1]);
            o[1] = pack_bf16x2(a0[2], a0[3]);
            o[2] = pack_bf16x2(a1[0], a1[1]);
            o[3] = pack_bf16x2(a1[2], a1[3]);
            *reinterpret_cast<u32x4*>(x + row * D + i) = o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {            // statistics of the values as stored
                v[c][2 * e] = bf16_lo(o[e]);
                v[c][2 * e + 1] = bf16_hi(o[e]);
                s += v[c][2 * e] + v[c][2 * e + 1];
            }
        }
    }
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < NCH8; ++c)
        if ((c * 64 + lane) * 8 < D) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float d = v[c][e] - mean;
                q += d * d;
            }
        }
    const float rstd = rsqrtf(wave_sum(q) / (float)D + eps);
    if (lane == 0 && stats) {
        stats[2 * row] = mean;
        stats[2 * row + 1] = rstd;
    }
}

// (mean, rstd) from the (sum, sum of squares) partials of a GEMM epilogue (hk_gemm_desc.row_part): 8 lanes per row,
// 16 bytes (two column groups) per lane and step, sums combined in float64.
__global__ __launch_bounds__(256) void hk_row_stats_finalize_kernel(const float* part, float* stats, int64_t rows,
                                                                   int groups, float eps) {
    const int sub = threadIdx.x & 7;
    int64_t row = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 3;
    const bool live = row < rows;
    if (!live) row = rows - 1;          // keep every lane in the shuffles
    const f32x4* pp = reinterpret_cast<const f32x4*>(part + row * groups * 2);
    double s1 = 0.0, s2 = 0.0;
    for (int g = sub; g < groups / 2; g += 8) {
        const f32x4 v = pp[g];
        s1 += (double)v[0] + (double)v[2];
        s2 += (double)v[1] + (double)v[3];
    }
#pragma unroll
    for (int d = 1; d < 8; d <<= 1) {
        s1 += __shfl_xor(s1, d);
        s2 += __shfl_xor(s2, d);
    }
    if (live && sub == 0) {
        const double D = 64.0 * groups;
        const double mean = s1 / D;
        const double var = fmax(s2 / D - mean * mean, 0.0);
        stats[2 * row] = (float)mean;
        stats[2 * row + 1] = rsqrtf((float)var + eps);
    }
}

template <int NCH>
__global__ __launch_bounds__(256) void hk_qf_embed_kernel(const float* query, const float* wemb, const float* pemb,
                                                          const int64_t* ids, const float* gamma, const float* beta,
                                                          float* out_f, __bf16* out_b, int B, int nq, int Lt, int d,
                                                          int vocab, float eps) {
    const int lane = threadIdx.x & 63;
    const int S = nq + Lt;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= (int64_t)B * S) return;
    const int b = (int)(row / S), s = (int)(row - (int64_t)b * S);
    RowRegs<NCH> x;
    const float* src;
    const float* pos = nullptr;
    if (s < nq) {
        src = query + (int64_t)s * d;
    } else {
        int64_t id = ids[(int64_t)b * Lt + (s - nq)];
        id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
        src = wemb + id * d;
        pos = pemb + (int64_t)(s - nq) * d;
    }
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int i = (c * 64 + lane) * 4;
        if (i < d) {
            x.v[c] = *reinterpret_cast<const f32x4*>(src + i);
            if (pos) x.v[c] += *reinterpret_cast<const f32x4*>(pos + i);
        }
    }
    ln_finish<NCH>(x, d, lane, gamma, beta, eps, out_b ? out_b + row * d : nullptr, out_f ? out_f + row * d : nullptr);
}

// thread -> 8 consecutive pixels of one image row (32 B in, 16 B out)
__global__ __launch_bounds__(256) void hk_patchify_kernel(const float* clip, __bf16* out, int64_t n8, int H, int W,
                                                          int P) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n8) return;
    const int w8 = W / 8;
    const int x8 = (int)(i % w8);
    int64_t rest = i / w8;
    const int y = (int)(rest % H);
    rest /= H;
    const int c = (int)(rest % 3);
    const int64_t bt = rest / 3;
    const f32x4 a = *reinterpret_cast<const f32x4*>(clip + i * 8);
    const f32x4 b = *reinterpret_cast<const f32x4*>(clip + i * 8 + 4);
    const int gx = (x8 * 8) / P, kx = (x8 * 8) % P;
    const int gy = y / P, ky = y % P;
    const int64_t orow = (bt * (H / P) + gy) * (W / P) + gx;
    const int k = (c * P + ky) * P + kx;
    u32x4 o;
    o[0] = pack_bf16x2(a[0], a[1]);
    o[1] = pack_bf16x2(a[2], a[3]);
    o[2] = pack_bf16x2(b[0], b[1]);
    o[3] = pack_bf16x2(b[2], b[3]);
    *reinterpret_cast<u32x4*>(out + orow * (3 * P * P) + k) = o;
}

__global__ __launch_bounds__(256) void hk_cast_kernel(const float* x, __bf16* out, int64_t n) {
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i + 3 < n) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(x + i);
        u32x2 o;
        o[0] = pack_bf16x2(a[0], a[1]);
        o[1] = pack_bf16x2(a[2], a[3]);
        *reinterpret_cast<u32x2*>(out + i) = o;
    } else {
        for (int64_t j = i; j < n; ++j) out[j] = (__bf16)x[j];
    }
}

template <int NCH>
void launch_ln(bool in_bf16, const void* x, const float* add, int n_add, int64_t add_stride, const float* gamma, const float* beta, __bf16* ob,
               float* of, int64_t rows, int D, float eps, hipStream_t s, int64_t o_ld, int64_t o_rpb, int64_t o_bs) {
    const dim3 grid((unsigned)((rows + 3) / 4));
    if (in_bf16)
        hipLaunchKernelGGL((hk_layernorm_kernel<NCH, true>), grid, dim3(256), 0, s, x, add, n_add, add_stride, gamma, beta, ob, of, rows, D, eps, o_ld, o_rpb, o_bs);
    else
        hipLaunchKernelGGL((hk_layernorm_kernel<NCH, false>), grid, dim3(256), 0, s, x, add, n_add, add_stride, gamma, beta, ob, of, rows, D, eps, o_ld, o_rpb, o_bs);
}

}  // namespace

extern "C" int hk_layernorm_parts(const void* x, int32_t in_dtype, const float* add, int32_t n_add, int64_t add_stride,
                                  const float* gamma, const float* beta, void* out_bf16, float* out_f32,
                                  const hk_rows* out_rows, int64_t rows, int32_t D, float eps, void* stream) {
    HK_REQUIRE(x && gamma && beta && (out_bf16 || out_f32), "hk_layernorm: null pointer");
    HK_REQUIRE(n_add >= 0 && n_add <= 64 && (n_add == 0 || add != nullptr) && add_stride % 4 == 0 &&
                   (n_add <= 1 || add_stride >= rows * (int64_t)D),
               "hk_layernorm: bad addends (n_add=%d, stride=%ld)", n_add, (long)add_stride);
    if (n_add == 0) add = nullptr;
    HK_REQUIRE(rows > 0 && D > 0 && D % 4 == 0 && D <= 4096, "hk_layernorm: bad shape rows=%ld D=%d", (long)rows, D);
    HK_REQUIRE(in_dtype == HK_DT_BF16 || in_dtype == HK_DT_F32, "hk_layernorm: unknown in_dtype %d", in_dtype);
    HK_REQUIRE(((uintptr_t)x | (uintptr_t)add | (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)out_bf16 |
                (uintptr_t)out_f32) % 16 == 0 && (D % 8 == 0 || in_dtype == HK_DT_F32),
               "hk_layernorm: operands must be 16-byte aligned");
    HK_REQUIRE((rows + 3) / 4 < (1L << 31), "hk_layernorm: too many rows");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const bool ib = in_dtype == HK_DT_BF16;
    __bf16* ob = reinterpret_cast<__bf16*>(out_bf16);
    const int64_t o_ld = out_rows ? out_rows->ld : D;
    const int64_t o_rpb = out_rows ? out_rows->rows_per_batch : 0;
    const int64_t o_bs = out_rows ? out_rows->batch_stride : 0;
    HK_REQUIRE(o_ld >= D && o_ld % 4 == 0 && o_bs % 4 == 0 && o_rpb >= 0, "hk_layernorm: bad out_rows mapping");
    if (ib && !add && out_bf16 && !out_f32 && !out_rows && D % 8 == 0 && D <= 2048) {
        const dim3 grid((unsigned)((rows + 3) / 4));
        const __bf16* xb = reinterpret_cast<const __bf16*>(x);
        const int n8 = (D + 511) / 512;
        if (n8 <= 1) hipLaunchKernelGGL((hk_ln_bf16_kernel<1>), grid, dim3(256), 0, s, xb, gamma, beta, ob, rows, D, eps);
        else if (n8 <= 2) hipLaunchKernelGGL((hk_ln_bf16_kernel<2>), grid, dim3(256), 0, s, xb, gamma, beta, ob, rows, D, eps);
        else hipLaunchKernelGGL((hk_ln_bf16_kernel<4>), grid, dim3(256), 0, s, xb, gamma, beta, ob, rows, D, eps);
        HK_CHECK_LAUNCH("hk_layernorm");
        return HK_OK;
    }
    const int nch = (D + 255) / 256;
    if (nch <= 1) launch_ln<1>(ib, x, add, n_add, add_stride, gamma, beta, ob, out_f32, rows, D, eps, s, o_ld, o_rpb, o_bs);
    else if (nch <= 2) launch_ln<2>(ib, x, add, n_add, add_stride, gamma, beta, ob, out_f32, rows, D, eps, s, o_ld, o_rpb, o_bs);
    else if (nch <= 3) launch_ln<3>(ib, x, add, n_add, add_stride, gamma, beta, ob, out_f32, rows, D, eps, s, o_ld, o_rpb, o_bs);
    else if (nch <= 4) launch_ln<4>(ib, x, add, n_add, add_stride, gamma, beta, ob, out_f32, rows, D, eps, s, o_ld, o_rpb, o_bs);
    else if (nch <= 8) launch_ln<8>(ib, x, add, n_add, add_stride, gamma, beta, ob, out_f32, rows, D, eps, s, o_ld, o_rpb, o_bs);
    else launch_ln<16>(ib, x, add, n_add, add_stride, gamma, beta, ob, out_f32, rows, D, eps, s, o_ld, o_rpb, o_bs);
    HK_CHECK_LAUNCH("hk_layernorm");
    return HK_OK;
}

extern "C" int hk_layernorm(const void* x, int32_t in_dtype, const float* add, const float* gamma, const float* beta,
                            void* out_bf16, float* out_f32, const hk_rows* out_rows, int64_t rows, int32_t D,
                            float eps, void* stream) {
    return hk_layernorm_parts(x, in_dtype, add, add ? 1 : 0, 0, gamma, beta, out_bf16, out_f32, out_rows, rows, D, eps, stream);
}

extern "C" int hk_row_stats(const void* x_bf16, float* stats, int64_t rows, int32_t D, float eps, void* stream) {
    HK_REQUIRE(x_bf16 && stats, "hk_row_stats: null pointer");
    HK_REQUIRE(rows > 0 && D > 0 && D % 8 == 0 && D <= 2048, "hk_row_stats: bad shape rows=%ld D=%d", (long)rows, D);
    HK_REQUIRE(((uintptr_t)x_bf16 % 16 == 0) && ((uintptr_t)stats % 8 == 0), "hk_row_stats: misaligned operands");
    HK_REQUIRE((rows + 3) / 4 < (1L << 31), "hk_row_stats: too many rows");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const dim3 grid((unsigned)((rows + 3) / 4));
    const __bf16* xb = reinterpret_cast<const __bf16*>(x_bf16);
    const int n8 = (D + 511) / 512;
    if (n8 <= 1) hipLaunchKernelGGL((hk_row_stats_kernel<1>), grid, dim3(256), 0, s, xb, stats, rows, D, eps);
    else if (n8 <= 2) hipLaunchKernelGGL((hk_row_stats_kernel<2>), grid, dim3(256), 0, s, xb, stats, rows, D, eps);
    else hipLaunchKernelGGL((hk_row_stats_kernel<4>), grid, dim3(256), 0, s, xb, stats, rows, D, eps);
    HK_CHECK_LAUNCH("hk_row_stats");
    return HK_OK;
}

extern "C" int hk_residual_add_parts(void* x_bf16, const float* parts, int32_t n_parts, int64_t part_stride, float* stats,
                                     int64_t rows, int32_t D, float eps, void* stream) {
    HK_REQUIRE(x_bf16 && parts, "hk_residual_add_parts: null pointer");
    HK_REQUIRE(rows > 0 && D > 0 && D % 8 == 0 && D <= 2048 && (rows + 3) / 4 < (1L << 31),
               "hk_residual_add_parts: bad shape rows=%ld D=%d", (long)rows, D);
    HK_REQUIRE(n_parts >= 1 && n_parts <= 64 && part_stride % 4 == 0 && (n_parts == 1 || part_stride >= rows * (int64_t)D),
               "hk_residual_add_parts: bad slices (n_parts=%d, stride=%ld)", n_parts, (long)part_stride);
    HK_REQUIRE(((uintptr_t)x_bf16 % 16 == 0) && ((uintptr_t)parts % 16 == 0) && ((uintptr_t)stats % 8 == 0),
               "hk_residual_add_parts: misaligned operands");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const dim3 grid((unsigned)((rows + 3) / 4));
    __bf16* xb = reinterpret_cast<__bf16*>(x_bf16);
    const int n8 = (D + 511) / 512;
    if (n8 <= 1) hipLaunchKernelGGL((hk_residual_add_parts_kernel<1>), grid, dim3(256), 0, s, xb, parts, n_parts, part_stride, stats, rows, D, eps);
    else if (n8 <= 2) hipLaunchKernelGGL((hk_residual_add_parts_kernel<2>), grid, dim3(256), 0, s, xb, parts, n_parts, part_stride, stats, rows, D, eps);
    else hipLaunchKernelGGL((hk_residual_add_parts_kernel<4>), grid, dim3(256), 0, s, xb, parts, n_parts, part_stride, stats, rows, D, eps);
    HK_CHECK_LAUNCH("hk_residual_add_parts");
    return HK_OK;
}

extern "C" int hk_row_stats_finalize(const float* part, float* stats, int64_t rows, int32_t groups, float eps,
                                     void* stream) {
    HK_REQUIRE(part && stats, "hk_row_stats_finalize: null pointer");
    HK_REQUIRE(rows > 0 && groups > 0 && groups % 2 == 0, "hk_row_stats_finalize: bad shape rows=%ld groups=%d", (long)rows, groups);
    HK_REQUIRE(((uintptr_t)part % 16 == 0) && ((uintptr_t)stats % 8 == 0), "hk_row_stats_finalize: misaligned operands");
    hipLaunchKernelGGL(hk_row_stats_finalize_kernel, dim3((unsigned)((rows * 8 + 255) / 256)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), part, stats, rows, groups, eps);
    HK_CHECK_LAUNCH("hk_row_stats_finalize");
    return HK_OK;
}

extern "C" int hk_qformer_embed_ln(const float* query_tokens, const float* word_emb, const float* pos_emb,
                                   const int64_t* input_ids, const float* gamma, const float* beta, float* out_f32,
                                   void* out_bf16, int32_t B, int32_t nq, int32_t Lt, int32_t d, int32_t vocab,
                                   float eps, void* stream) {
    HK_REQUIRE(query_tokens && gamma && beta && (out_f32 || out_bf16), "hk_qformer_embed_ln: null pointer");
    HK_REQUIRE(Lt == 0 || (word_emb && pos_emb && input_ids), "hk_qformer_embed_ln: text tables missing");
    HK_REQUIRE(B > 0 && nq >= 0 && Lt >= 0 && nq + Lt > 0 && d > 0 && d % 8 == 0 && d <= 1024 && vocab > 0,
               "hk_qformer_embed_ln: bad shape B=%d nq=%d Lt=%d d=%d", B, nq, Lt, d);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int64_t rows = (int64_t)B * (nq + Lt);
    const dim3 grid((unsigned)((rows + 3) / 4));
    __bf16* ob = reinterpret_cast<__bf16*>(out_bf16);
    const int nch = (d + 255) / 256;
#define HK_EMB(N) hipLaunchKernelGGL((hk_qf_embed_kernel<N>), grid, dim3(256), 0, s, query_tokens, word_emb, pos_emb, \
                                     input_ids, gamma, beta, out_f32, ob, B, nq, Lt, d, vocab, eps)
    if (nch <= 1) HK_EMB(1);
    else if (nch <= 2) HK_EMB(2);
    else if (nch <= 3) HK_EMB(3);
    else HK_EMB(4);
#undef HK_EMB
    HK_CHECK_LAUNCH("hk_qformer_embed_ln");
    return HK_OK;
}

extern "C" int hk_patchify_bf16(const float* clip, void* out, int32_t B, int32_t T, int32_t H, int32_t W,
                                int32_t patch, void* stream) {
    HK_REQUIRE(clip && out, "hk_patchify_bf16: null pointer");
    HK_REQUIRE(B > 0 && T > 0 && H > 0 && W > 0 && patch > 0 && patch % 8 == 0 && H % patch == 0 && W % patch == 0,
               "hk_patchify_bf16: bad shape B=%d T=%d H=%d W=%d patch=%d", B, T, H, W, patch);
    HK_REQUIRE(((uintptr_t)clip | (uintptr_t)out) % 16 == 0, "hk_patchify_bf16: operands must be 16-byte aligned");
    const int64_t n8 = (int64_t)B * T * 3 * H * W / 8;
    HK_REQUIRE((n8 + 255) / 256 < (1L << 31), "hk_patchify_bf16: clip too large");
    hipLaunchKernelGGL(hk_patchify_kernel, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), clip, reinterpret_cast<__bf16*>(out), n8, H, W, patch);
    HK_CHECK_LAUNCH("hk_patchify_bf16");
    return HK_OK;
}

extern "C" int hk_cast_f32_bf16(const float* x, void* out, int64_t n, void* stream) {
    HK_REQUIRE(x && out && n > 0, "hk_cast_f32_bf16: bad arguments");
    HK_REQUIRE(((uintptr_t)x % 16 == 0) && ((uintptr_t)out % 8 == 0), "hk_cast_f32_bf16: misaligned operands");
    const int64_t nb = (n + 1023) / 1024;
    HK_REQUIRE(nb < (1L << 31), "hk_cast_f32_bf16: too large");
    hipLaunchKernelGGL(hk_cast_kernel, dim3((unsigned)nb), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x,
                       reinterpret_cast<__bf16*>(out), n);
    HK_CHECK_LAUNCH("hk_cast_f32_bf16");
    return HK_OK;
}
