// Fused flash-style attention forward for gfx950, head size 64, bf16 in/out, fp32 online softmax.
//
// One workgroup = 4 waves = 128 query rows of one (batch, head); each wave owns 32 query rows and
// walks the keys in 64-key tiles staged through LDS (double-buffered, register-staged loads issued
// one tile ahead).  Everything is computed TRANSPOSED with v_mfma_f32_32x32x16_bf16 so that the
// query row lives on the lane:
//     S^T[key][q] = K . Q^T          (A operand = K tile rows from LDS, B operand = Q rows in registers)
//     O^T[d][q]  += V^T . P^T        (A operand = V^T via ds_read_b64_tr_b16, B operand = exp2(S^T - m))
// The S^T accumulator of a lane is already the B-operand fragment of the second product (the
// accumulator-as-operand property of the 32x32 shape), so P never crosses lanes or LDS; the only
// cross-lane step per tile is one v_permlane32_swap for the row maximum.  The k order inside one
// 16-key MFMA step is the accumulator's (key = 16s + 8(j>>2) + 4h + (j&3) for element j of lane
// half h), and the V^T fragments are gathered in that same order.
//
// Q must arrive pre-multiplied by softmax_scale * log2(e) (folded into the producing GEMM), so the
// kernel works in the exp2 domain with no per-score multiply.
//
// LDS images (128-byte rows = one key's 64 head dims), 16-byte chunk c of key row r stored at
//     K: chunk c ^ ((r >> 1) & 7)        -> conflict-free ds_read_b128 of 32 rows x one chunk
//     V: chunk c ^ (((r >> 1) & 1) << 2) -> conflict-free ds_read_b64_tr_b16 of 4 keys x 32 dims
#include <stdlib.h>
#include <type_traits>
#include "hk_common.h"

namespace {

struct AttnParams {
    const __bf16* Q;
    const __bf16* K;
    const __bf16* V;
    __bf16* O;
    const float* key_bias;
    int B, H, Nq, Nk;
    int64_t ldq, ldk, ldv, ldo, bsq, bsk, bsv, bso;
    int nqb;
    int prio;     // staged wave priority inside a tile (see the kernel); 0 = off (A/B aid)
    int o_vec16;  // O rows 16-byte aligned: the epilogue may store whole 128-byte row segments
    unsigned long long* stamp;  // diagnostic build (-DHK_ATTN_STAMP) only: per-wave phase cycle totals
    // key split (small calls): the grid's batch index is b * splits + s; slice s covers keys [s * chunk, ...), writes its
    // normalised partial output to O (= the partial buffer, batch index b * splits + s) and the row's log2-sum-exp to lse
    int splits, chunk;
    float* lse;
};

constexpr int QB = 128;   // query rows per workgroup
constexpr int KT = 64;    // keys per tile
constexpr float NEG_BIG = -1.0e30f;
constexpr float OVERFLOW_SUM = 1048576.0f;  // 2^20: a lane's 32-score partial row sum beyond this redoes the tile
constexpr float NOMAX_SUM = 1.2676506e30f;        // 2^100: MODE 2's no-maximum pass gives up beyond this partial row sum ...
constexpr float NOMAX_MIN_TOTAL = 7.8886091e-31f;  // ... or below this (2^-100) row total

// v_permlane32_swap exchanges lanes 32-63 of its first operand with lanes 0-31 of its second:
// called on two copies of x, one result holds (own, partner) and the other (partner, own).
// Inline asm, not __builtin_amdgcn_permlane32_swap: ROCm 7.2's clang returns the first result in
// both elements of the builtin's vector (fmaxf(r[0], r[1]) folds to r[0]).  The s_nop covers the
// VALU-write -> v_permlane read hazard (2 wait states), which hipcc does not pad inside asm.
__device__ __forceinline__ void half_swap(float x, float& a, float& b) {
    a = x;
    b = x;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ float half_swap_max(float x) {
    float a, b;
    half_swap(x, a, b);
    return fmaxf(a, b);
}
__device__ __forceinline__ float half_swap_sum(float x) {
    float a, b;
    half_swap(x, a, b);
    return a + b;
}

// Measured and not kept (round 2): 64 query rows per wave -- this tile body over two 32-row query blocks, so one K / V^T
// fragment read and one LDS-DMA piece feed twice the MFMAs; 256 rows per workgroup, two workgroups per CU, 232 registers
// and no scratch in the no-maximum form (with the running maximum and its redo path: 2.4 KB of scratch) -- 1355 -> 1396 us
// per launch at B=32 N=3136 (13 blocks of 256 rows for 12.25), i.e. no gain per row: the loop is not waiting on LDS reads
// or DMA issue.
//
// SPLIT: the key-split form (small calls) is an instantiation of its own -- its extra live scalars (slice base
// pointers, slice length, lse pointer) cost the production instantiation SGPR spills and 20 bytes of scratch when
// they shared one body.
template <bool HAS_BIAS, int MODE, int OCC, int LOADER, bool SPLIT = false>
__global__ __launch_bounds__(256, OCC) void hk_attn_kernel(const AttnParams p) {
    __shared__ __attribute__((aligned(16))) char smem[2 * 2 * KT * 128];  // [stage][K|V][64 keys][128 B]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: LDS-DMA bases stay on the scalar unit
    const int r = lane & 31;   // query row within the wave's block (MFMA column)
    const int h = lane >> 5;   // lane half

    // XCD-aware remap: the query blocks of one (batch, head) run on one XCD so K/V stay in its L2.
    const int nwg = gridDim.x;
    int lid;
    {
        const int orig = blockIdx.x, xcd = orig & 7, q = nwg >> 3, rem = nwg & 7;
        lid = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + (orig >> 3);
    }
    const int bh = lid / p.nqb;
    const int qb = lid - bh * p.nqb;
    int b = bh / p.H;
    const int head = bh - b * p.H;
    const int q0 = qb * QB + wave * 32;
    const int bo = b;          // batch index of the output (the partial buffer's when the keys are split)
    // this workgroup's keys: all of them, or one slice (scalars, not a modified copy of the argument block: that
    // copy lands in scratch)
    const __bf16* Kp = p.K;
    const __bf16* Vp = p.V;
    int Nk = p.Nk;
    if (SPLIT && p.splits > 1) {
        const int sl = b % p.splits;
        b /= p.splits;
        const int k0 = sl * p.chunk;
        Kp += (int64_t)k0 * p.ldk;
        Vp += (int64_t)k0 * p.ldv;
        Nk = min(p.chunk, p.Nk - k0);
    }

    // ---- Q fragments: B operand, lane (r, h) holds Q[q0 + r][16 ks + 8 h .. +8] ----
    bf16x8 qf[4];
    {
        const int qrow = min(q0 + r, p.Nq - 1);
        const __bf16* qp = p.Q + (int64_t)b * p.bsq + (int64_t)qrow * p.ldq + head * 64 + 8 * h;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(qp + 16 * ks);
    }

    // ---- K/V tile loader: 256 threads x 16 B x 2 passes per operand ----
    const __bf16* kbase = Kp + (int64_t)b * p.bsk + head * 64 + (tid & 7) * 8;
    const __bf16* vbase = Vp + (int64_t)b * p.bsv + head * 64 + (tid & 7) * 8;
    const int lkey = tid >> 3;  // 0..31 (+32 on the second pass)
    int kst[2], vst[2];         // LDS byte offsets of this thread's two chunks
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int key = i * 32 + lkey, c = tid & 7;
        kst[i] = key * 128 + ((c ^ ((key >> 1) & 7)) << 4);
        vst[i] = key * 128 + ((c ^ (((key >> 1) & 1) << 2)) << 4);
    }
    bf16x8 kr[2], vr[2];
    const __bf16* kp = kbase + (int64_t)lkey * p.ldk;   // this thread's rows of the NEXT tile to load
    const __bf16* vp_ = vbase + (int64_t)lkey * p.ldv;
    const int64_t k32 = 32 * p.ldk, v32 = 32 * p.ldv;
    auto gload = [&](int tile) __attribute__((always_inline)) {
        if ((tile + 1) * KT <= Nk) {
            kr[0] = *reinterpret_cast<const bf16x8*>(kp);
            kr[1] = *reinterpret_cast<const bf16x8*>(kp + k32);
            vr[0] = *reinterpret_cast<const bf16x8*>(vp_);
            vr[1] = *reinterpret_cast<const bf16x8*>(vp_ + v32);
        } else {  // ragged last tile: clamp to the last key (masked to -inf below)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int key = min(tile * KT + i * 32 + lkey, Nk - 1);
                kr[i] = *reinterpret_cast<const bf16x8*>(kbase + (int64_t)key * p.ldk);
                vr[i] = *reinterpret_cast<const bf16x8*>(vbase + (int64_t)key * p.ldv);
            }
        }
        kp += 2 * k32;
        vp_ += 2 * v32;
    };
    auto lstore = [&](int stage) __attribute__((always_inline)) {
        char* ks = smem + stage * (2 * KT * 128);
        char* vs = ks + KT * 128;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            *reinterpret_cast<bf16x8*>(ks + kst[i]) = kr[i];
            *reinterpret_cast<bf16x8*>(vs + vst[i]) = vr[i];
        }
    };
    // LOADER 1: LDS-DMA (global_load_lds_dwordx4) instead of register staging.  The LDS image is
    // lane-linear per wave-instruction (8 keys x 128 B), so the swizzles go on the SOURCE chunk.
    const int ck = (tid & 7) ^ ((lkey >> 1) & 7);
    const int cv = (tid & 7) ^ (((lkey >> 1) & 1) << 2);
    // address = wave-uniform base (advanced per tile on the scalar unit) + a constant 32-bit per-lane byte offset
    const char* kU = reinterpret_cast<const char*>(Kp + (int64_t)b * p.bsk + head * 64);
    const char* vU = reinterpret_cast<const char*>(Vp + (int64_t)b * p.bsv + head * 64);
    const uint32_t klane = (uint32_t)(((int64_t)lkey * p.ldk + ck * 8) * 2);
    const uint32_t vlane = (uint32_t)(((int64_t)lkey * p.ldv + cv * 8) * 2);
    const int64_t k32b = 64 * p.ldk, v32b = 64 * p.ldv;  // 32 key rows, in bytes
    auto gdma = [&](int tile, int stage) __attribute__((always_inline)) {
        char* ks = smem + stage * (2 * KT * 128) + wave * 8 * 128;
        char* vs = ks + KT * 128;
        if ((tile + 1) * KT <= Nk) {
            const char* kt_ = kU + (int64_t)tile * 2 * k32b;
            const char* vt_ = vU + (int64_t)tile * 2 * v32b;
            if constexpr (HAS_BIAS) {   // that instantiation is short of SGPRs: the builtin (vector address) form throughout
                glds16(kt_ + klane, ks);
                glds16(kt_ + k32b + klane, ks + 32 * 128);
                glds16(vt_ + vlane, vs);
                glds16(vt_ + v32b + vlane, vs + 32 * 128);
            } else {
                glds16_s(kt_, klane, ks);
                glds16_s(kt_ + k32b, klane, ks + 32 * 128);
                glds16_s(vt_, vlane, vs);
                glds16_s(vt_ + v32b, vlane, vs + 32 * 128);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int key = min(tile * KT + i * 32 + lkey, Nk - 1);
                const __bf16* ka = Kp + (int64_t)b * p.bsk + head * 64 + ck * 8 + (int64_t)key * p.ldk;
                const __bf16* va = Vp + (int64_t)b * p.bsv + head * 64 + cv * 8 + (int64_t)key * p.ldv;
                if constexpr (HAS_BIAS) {
                    glds16(ka, ks + i * 32 * 128);
                    glds16(va, vs + i * 32 * 128);
                } else {
                    glds16_v(ka, ks + i * 32 * 128);
                    glds16_v(va, vs + i * 32 * 128);
                }
            }
        }
    };

    // ---- per-lane LDS read offsets ----
    // K fragment (kt, ks): row 32 kt + r, logical chunk 2 ks + h
    const int ksw = (r >> 1) & 7;
    int koff[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) koff[ks] = r * 128 + (((2 * ks + h) ^ ksw) << 4);
    // V^T fragment (s, dt, half-read u): keys 16 s + 8 u + 4 h + ((lane & 15) >> 2),
    // dims 32 dt + 16 ((lane >> 4) & 1) + 4 (lane & 3)
    const int vq = (lane & 15) >> 2;
    const int vsb = (vq >> 1) & 1;  // swizzle bit of that key row
    int voff[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
        const int c = 4 * dt + 2 * ((lane >> 4) & 1) + ((lane & 3) >> 1);
        voff[dt] = (4 * h + vq) * 128 + ((c ^ (vsb << 2)) << 4) + 8 * (lane & 1);
    }

    f32x16 ot[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int i = 0; i < 16; ++i) ot[dt][i] = 0.f;
    float m_run = 0.f, l_run = 0.f;  // every stored quantity is relative to m_run
    // LOADER 2: -m_run kept in 16 registers as the srcC of the first MFMA of each chain, rebuilt only when m_run moves
    f32x16 cneg;
    auto set_cneg = [&]() __attribute__((always_inline)) {
        typedef __attribute__((ext_vector_type(2))) float f32x2;
        const f32x2 negm2 = {-m_run, -m_run};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            f32x2 c2;
            asm volatile("v_mov_b64 %0, %1" : "=v"(c2) : "v"(negm2));
            cneg[2 * j] = c2[0];
            cneg[2 * j + 1] = c2[1];
        }
    };
    if (LOADER == 2) set_cneg();

    const int NT = (Nk + KT - 1) / KT;
    bool bad = false;   // MODE 2, first pass: some score left the range the no-maximum form covers

    // One K/V tile.  FIRST (tile 0) and RAGGED (last tile when Nk % 64 != 0) are compile-time so the
    // steady-state body carries neither the key mask nor the first-tile special case.
    //
    // Scores are kept relative to m_run, which enters as the initial MFMA accumulator.
    // MODE 0: the row maximum of every tile is taken and m_run moves when some row of the wave
    //   outgrows it by more than 8 bits (deferred rescale).
    // MODE 1: the common path is exp2 + row sum only.  m_run is set from the true row maximum on the
    //   first tile and afterwards only when a tile "overflows" -- some lane's partial row sum leaves
    //   [0, OVERFLOW_SUM], i.e. a score exceeded m_run by ~20 bits or exp2 returned inf; that tile is
    //   redone with the maximum taken.
    // MODE 2: first a pass with NO running maximum at all -- P = exp2(s) as it comes out of the MFMA (initial
    //   accumulator = the inline constant 0: no -m block to build per tile, no first-tile maximum).  bf16 has fp32's
    //   exponent range, so P, l and O keep full relative precision for any |s| < ~100 (log2 units; a trained ViT's
    //   scaled logits stay within a few tens); the pass only records whether a partial row sum left [0, 2^100] or a
    //   row's total fell below 2^-100.  If any wave of the workgroup recorded that, the whole workgroup runs the
    //   MODE 1 pass from the start -- the result is then that pass's, bit for bit.
    // Either way the decision is wave-uniform, is taken after the previous tile's P.V is complete and
    // before any P of this tile is used, so O, l and this tile's scores are the only things at the
    // old scale when a rescale happens.
    // The LDS stage (t & 1) is a compile-time argument too -- the steady-state loop is unrolled by two -- so every
    // fragment address is base register + immediate offset.
    auto tile = [&](const int t, auto first_c, auto ragged_c, auto cur_c, auto mode_c) __attribute__((always_inline)) {
        constexpr int M = decltype(mode_c)::value;   // 0, 1 as above; 3 = the no-maximum pass of MODE 2
        constexpr bool FIRST = decltype(first_c)::value && M != 3;
        constexpr bool RAGGED = decltype(ragged_c)::value;
        constexpr int cur = decltype(cur_c)::value;
        if (t + 1 < NT) {
            if (LOADER >= 1) gdma(t + 1, cur ^ 1); else gload(t + 1);
        }
        const char* ks = smem + cur * (2 * KT * 128);
        const char* vs = ks + KT * 128;
        f32x16 st[2];
        float rs;

        auto scores = [&]() __attribute__((always_inline)) {
            if (p.prio) __builtin_amdgcn_s_setprio(1);
            // initial accumulator = -m_run in all 16 registers, built with 8 v_mov_b64 (one block serves both key
            // halves).  Keeping the block alive across tiles instead measured 1.5 % slower (register pressure).
            typedef __attribute__((ext_vector_type(2))) float f32x2;
            const f32x2 negm2 = {-m_run, -m_run};
            f32x16 cinit;
            if (M == 3) {
#pragma unroll
                for (int j = 0; j < 16; ++j) cinit[j] = 0.f;   // folds into the MFMA's inline-constant srcC
            } else if (LOADER == 2) {
                cinit = cneg;
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    f32x2 c2;
                    asm volatile("v_mov_b64 %0, %1" : "=v"(c2) : "v"(negm2));
                    cinit[2 * j] = c2[0];
                    cinit[2 * j + 1] = c2[1];
                }
            }
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const bf16x8 kf = *reinterpret_cast<const bf16x8*>(ks + kt * 32 * 128 + koff[s]);
                    st[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[s], s == 0 ? cinit : st[kt], 0, 0, 0);
                }
            }
            if (p.prio) __builtin_amdgcn_s_setprio(0);
            // register i of st[kt] <-> key t*64 + 32 kt + (i & 3) + 8 (i >> 2) + 4 h
            if (HAS_BIAS) {
                const float* kb = p.key_bias + (int64_t)b * Nk;
                typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int key0 = t * KT + 32 * kt + 8 * g + 4 * h;  // registers 4g..4g+3 <-> keys key0..key0+3
                        if (!RAGGED) {   // whole tile exists: one 16-byte load per four keys
                            const f32x4u v = *reinterpret_cast<const f32x4u*>(kb + key0);
#pragma unroll
                            for (int e = 0; e < 4; ++e) st[kt][4 * g + e] = __builtin_fmaf(v[e], HK_LOG2E, st[kt][4 * g + e]);
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                if (key0 + e < Nk) st[kt][4 * g + e] = __builtin_fmaf(kb[key0 + e], HK_LOG2E, st[kt][4 * g + e]);
                        }
                    }
            }
            if (RAGGED) {  // keys >= Nk do not exist
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int key = t * KT + 32 * kt + (i & 3) + 8 * (i >> 2) + 4 * h;
                        if (key >= Nk) st[kt][i] = NEG_BIG;
                    }
            }
        };
        auto rowmax = [&]() __attribute__((always_inline)) {
            float mx = fmaxf(st[0][0], st[0][1]);
#pragma unroll
            for (int i = 2; i < 16; i += 2) mx = fmaxf(mx, fmaxf(st[0][i], st[0][i + 1]));
#pragma unroll
            for (int i = 0; i < 16; i += 2) mx = fmaxf(mx, fmaxf(st[1][i], st[1][i + 1]));
            return half_swap_max(mx);
        };
        auto move_max = [&](const float mx) __attribute__((always_inline)) {  // m_run += delta; everything at the old scale follows
            const float delta = FIRST ? mx : fmaxf(mx, 0.f);
            const float alpha = FIRST ? 0.f : __builtin_amdgcn_exp2f(-delta);
            m_run += delta;
            if (LOADER == 2) set_cneg();
            l_run *= alpha;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int i = 0; i < 16; ++i) ot[dt][i] *= alpha;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int i = 0; i < 16; ++i) st[kt][i] -= delta;
        };
        auto exp_sum = [&]() __attribute__((always_inline)) {
            // (row sums over register pairs with v_pk_add_f32 -- 16 packed adds instead of 32 scalar ones -- measured 7 %
            // SLOWER per launch, 1357 -> 1455 us same box: packed fp32 issues dearly beside MFMAs, like v_dot2c before it;
            // row sums on the matrix pipe instead -- ones(32 x 16) . P^T as a 5th MFMA per 16-key step of the no-maximum
            // pass, no v_add_f32 at all, 168 registers, no scratch -- measured 1420 -> 1470 us with two of the four steps
            // summed that way and 1507 us with all four: the matrix pipe has no room either)
            rs = 0.f;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float e = __builtin_amdgcn_exp2f(st[kt][i]);
                    st[kt][i] = e;
                    rs += e;
                }
        };

        scores();
        if (M == 3) {
            exp_sum();
            bad |= __builtin_amdgcn_ballot_w64(!(rs <= NOMAX_SUM)) != 0;
        } else if (FIRST) {
            move_max(rowmax());
            exp_sum();
        } else if (M == 0) {
            const float mx = rowmax();
            if (__builtin_amdgcn_ballot_w64(mx > 8.0f) != 0) move_max(mx);
            exp_sum();
        } else {
            exp_sum();
            if (__builtin_amdgcn_ballot_w64(!(rs <= OVERFLOW_SUM)) != 0) {
                scores();
                move_max(rowmax());
                exp_sum();
            }
        }
        l_run += rs;
        // Staged priority: the three waves of a SIMD are in different phases of their tiles; the one feeding the matrix
        // pipe goes first (P.V chain at 2, Q.K^T chain at 1) and the exp / row-sum VALU of the others fills the gaps
        // (+4 % at the ViT shape; softmax at 1 instead measured -4 %).
        if (p.prio) __builtin_amdgcn_s_setprio(2);

        // ---- P^T fragments straight from the accumulator registers ----
        bf16x8 pf[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            u32x4 w;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                w[j] = pack_bf16x2(st[s >> 1][8 * (s & 1) + 2 * j], st[s >> 1][8 * (s & 1) + 2 * j + 1]);
            pf[s] = __builtin_bit_cast(bf16x8, w);
        }

        // ---- O^T += V^T . P^T ----
#pragma unroll
        for (int s = 0; s < 4; ++s) {
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                const char* vp = vs + s * 16 * 128 + voff[dt];
                const bf16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                    (__attribute__((address_space(3))) bf16x4*)(vp));
                const bf16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                    (__attribute__((address_space(3))) bf16x4*)(vp + 8 * 128));
                bf16x8 vf;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    vf[j] = v0[j];
                    vf[j + 4] = v1[j];
                }
                ot[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[s], ot[dt], 0, 0, 0);
            }
        }

        if (p.prio) __builtin_amdgcn_s_setprio(0);
        if (LOADER >= 1) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else if (t + 1 < NT) {
            lstore(cur ^ 1);
        }
        __syncthreads();
    };

    using T_ = std::true_type;
    using F_ = std::false_type;
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    const bool ragged = (Nk % KT) != 0;
    auto run_pass = [&](auto mode_c) __attribute__((always_inline)) {
        if (LOADER >= 1) {
            gdma(0, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            gload(0);
            lstore(0);
        }
        __syncthreads();
        auto last = [&](const int t, auto first_c) __attribute__((always_inline)) {
            if (t & 1) {
                if (ragged) tile(t, first_c, T_{}, S1{}, mode_c); else tile(t, first_c, F_{}, S1{}, mode_c);
            } else {
                if (ragged) tile(t, first_c, T_{}, S0{}, mode_c); else tile(t, first_c, F_{}, S0{}, mode_c);
            }
        };
        if (NT == 1) {
            last(0, T_{});
        } else {
            tile(0, T_{}, F_{}, S0{}, mode_c);
            int t = 1;
            for (; t + 1 < NT - 1; t += 2) {
                tile(t, F_{}, F_{}, S1{}, mode_c);
                tile(t + 1, F_{}, F_{}, S0{}, mode_c);
            }
            if (t < NT - 1) tile(t, F_{}, F_{}, S1{}, mode_c);
            last(NT - 1, F_{});
        }
    };
    float l_tot;
    if (MODE == 2) {
        run_pass(std::integral_constant<int, 3>{});
        l_tot = half_swap_sum(l_run);
        bad |= __builtin_amdgcn_ballot_w64(!(l_tot >= NOMAX_MIN_TOTAL)) != 0;
        if (__syncthreads_or(bad ? 1 : 0)) {   // uniform over the workgroup (its waves share every barrier of a pass)
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int i = 0; i < 16; ++i) ot[dt][i] = 0.f;
            m_run = 0.f;
            l_run = 0.f;
            if (LOADER == 0) {
                kp = kbase + (int64_t)lkey * p.ldk;
                vp_ = vbase + (int64_t)lkey * p.ldv;
            }
            run_pass(std::integral_constant<int, 1>{});
            l_tot = half_swap_sum(l_run);
        }
    } else {
        run_pass(std::integral_constant<int, MODE>{});
        l_tot = half_swap_sum(l_run);
    }

    // ---- normalise and store: lane (r, h) holds O[q0 + r][32 dt + 8 g + 4 h .. +4] ----
    const float inv = 1.0f / l_tot;
    if (SPLIT && p.lse != nullptr && h == 0 && q0 + r < p.Nq)   // key split: the slice's weight in the combine
        p.lse[((int64_t)bo * p.H + head) * p.Nq + q0 + r] = m_run + __builtin_amdgcn_logf(l_tot);
    if (p.o_vec16) {
        // Whole 128-byte row segments: the wave's 32 x 64 bf16 block is transposed through its own 4 KB of the
        // (now idle -- every tile ends in a workgroup barrier) K/V staging area, 16-B chunk c of row r at chunk
        // c ^ ((r >> 1) & 7), and leaves as 4 stores of 8 rows x 128 B instead of 8 stores of 32 rows x 16 B.
        char* stg = smem + wave * 4096;
        const int sw = (r >> 1) & 7;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                u32x2 o;
                o[0] = pack_bf16x2(ot[dt][4 * g] * inv, ot[dt][4 * g + 1] * inv);
                o[1] = pack_bf16x2(ot[dt][4 * g + 2] * inv, ot[dt][4 * g + 3] * inv);
                *reinterpret_cast<u32x2*>(stg + r * 128 + (((4 * dt + g) ^ sw) << 4) + 8 * h) = o;
            }
        __bf16* ob = p.O + (int64_t)bo * p.bso + head * 64 + (lane & 7) * 8;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int row = it * 8 + (lane >> 3);
            const u32x4 o = *reinterpret_cast<const u32x4*>(stg + row * 128 + (((lane & 7) ^ ((row >> 1) & 7)) << 4));
            if (q0 + row < p.Nq) *reinterpret_cast<u32x4*>(ob + (int64_t)(q0 + row) * p.ldo) = o;
        }
        return;
    }
    const int qrow = q0 + r;
    if (qrow < p.Nq) {
        __bf16* op = p.O + (int64_t)bo * p.bso + (int64_t)qrow * p.ldo + head * 64 + 4 * h;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                u32x2 o;
                o[0] = pack_bf16x2(ot[dt][4 * g] * inv, ot[dt][4 * g + 1] * inv);
                o[1] = pack_bf16x2(ot[dt][4 * g + 2] * inv, ot[dt][4 * g + 3] * inv);
                *reinterpret_cast<u32x2*>(op + 32 * dt + 8 * g) = o;
            }
    }
}


// Combine of a key-split launch: O[b][q][h][:] = sum_s w_s * part[b * S + s][q][h][:] / sum_s w_s, w_s = 2^(lse_s - max lse).
// One thread per 8 head dims (16 bytes); slices in index order (bit-reproducible).
// (Measured and not kept: the same combine INSIDE the split launch -- __threadfence(), an agent-scope ticket, the last
// slice to arrive re-reads the partials: bit-identical, but ViT self-attention at B = 1, T = 12 went 36.7 -> 159.8 us at two
// slices and cross-attention 16.7 -> 24.5 us at four: a device-scope release / acquire writes back and invalidates the
// XCD's L2 per workgroup, a kernel boundary does it once.)
__global__ __launch_bounds__(256) void hk_attn_combine_kernel(const __bf16* part, const float* lse, __bf16* O, int S, int B,
                                                             int H, int Nq, int64_t ldo, int64_t bso) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int c = (int)(idx & 7);
    int64_t rest = idx >> 3;
    const int h = (int)(rest % H);
    rest /= H;
    const int q = (int)(rest % Nq);
    const int b = (int)(rest / Nq);
    if (b >= B) return;
    float m = -3.0e38f;
    for (int s = 0; s < S; ++s) m = fmaxf(m, lse[(((int64_t)b * S + s) * H + h) * Nq + q]);
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float den = 0.f;
    for (int s = 0; s < S; ++s) {
        const int64_t bp = (int64_t)b * S + s;
        const float w = __builtin_amdgcn_exp2f(lse[(bp * H + h) * Nq + q] - m);
        const u32x4 v = *reinterpret_cast<const u32x4*>(part + ((bp * Nq + q) * H + h) * 64 + c * 8);
        den += w;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            acc[2 * e] = __builtin_fmaf(w, bf16_lo(v[e]), acc[2 * e]);
            acc[2 * e + 1] = __builtin_fmaf(w, bf16_hi(v[e]), acc[2 * e + 1]);
        }
    }
    const float inv = 1.0f / den;
    u32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = pack_bf16x2(acc[2 * e] * inv, acc[2 * e + 1] * inv);
    *reinterpret_cast<u32x4*>(O + (int64_t)b * bso + (int64_t)q * ldo + h * 64 + c * 8) = o;
}

// ---------------------------------------------------------------------------------------------------------------
// Software-pipelined variant for the ViT shape (no key bias): one loop iteration runs the softmax and P.V of tile t
// and, interleaved instruction by instruction with that softmax, the Q.K^T of tile t + 1 -- so a wave's own exp /
// convert VALU work issues in the shadow of its own MFMAs instead of waiting for another wave to cover it.
//   iteration t:  DMA K(t+2), V(t+1)  |  8 x { MFMA S(t+1) ; 4 exp2 + 2 cvt_pk + 2 dot2 of S(t) }  |  overflow check
//                 |  8 MFMA O += V(t) P(t)  |  barrier
// The K and V double buffers are therefore one tile out of phase.  exp2 does not overwrite the scores: P goes
// straight to its packed bf16 form (the MFMA operand) and the row sum is taken over those rounded values with
// v_dot2c_f32_bf16 (so the normaliser is the sum of exactly the weights P.V uses); the raw scores stay live until the
// overflow check, whose rare redo (MODE 1 of the kernel above: true row maximum, rescale O / l, shift S(t) and the
// already computed S(t+1)) therefore needs no second Q.K^T.
template <int NW, int PRIO>
__global__ __launch_bounds__(NW * 64, 2) void hk_attn_pipe_kernel(const AttnParams p) {
    __shared__ __attribute__((aligned(16))) char smem[4 * KT * 128];  // K stage 0, 1 | V stage 0, 1
    constexpr int STG = KT * 128;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31;
    const int h = lane >> 5;

    const int nwg = gridDim.x;
    int lid;
    {
        const int orig = blockIdx.x, xcd = orig & 7, q = nwg >> 3, rem = nwg & 7;
        lid = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + (orig >> 3);
    }
    const int bh = lid / p.nqb;
    const int qb = lid - bh * p.nqb;
    const int b = bh / p.H;
    const int head = bh - b * p.H;
    const int q0 = qb * (NW * 32) + wave * 32;
    const bool active = q0 < p.Nq;   // wave-uniform; an idle wave of the last query block still loads its K/V pieces

    bf16x8 qf[4];
    {
        const int qrow = min(q0 + r, p.Nq - 1);
        const __bf16* qp = p.Q + (int64_t)b * p.bsq + (int64_t)qrow * p.ldq + head * 64 + 8 * h;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(qp + 16 * ks);
    }

    // LDS-DMA addressing as in the kernel above (swizzle on the source chunk)
    const int lkey = tid >> 3;
    const int ck = (tid & 7) ^ ((lkey >> 1) & 7);
    const int cv = (tid & 7) ^ (((lkey >> 1) & 1) << 2);
    const char* kU = reinterpret_cast<const char*>(p.K + (int64_t)b * p.bsk + head * 64);
    const char* vU = reinterpret_cast<const char*>(p.V + (int64_t)b * p.bsv + head * 64);
    const uint32_t klane = (uint32_t)(((int64_t)lkey * p.ldk + ck * 8) * 2);
    const uint32_t vlane = (uint32_t)(((int64_t)lkey * p.ldv + cv * 8) * 2);
    const int64_t k32b = 64 * p.ldk, v32b = 64 * p.ldv;
    auto kdma = [&](int tile, int stage) __attribute__((always_inline)) {
        char* ks = smem + stage * STG + wave * 8 * 128;
        if ((tile + 1) * KT <= p.Nk) {
            const char* kt_ = kU + (int64_t)tile * 2 * k32b;
            glds16_s(kt_, klane, ks);
            if (NW == 4) glds16_s(kt_ + k32b, klane, ks + 32 * 128);
        } else {
#pragma unroll
            for (int i = 0; i < (NW == 4 ? 2 : 1); ++i) {
                const int key = min(tile * KT + i * 32 + lkey, p.Nk - 1);
                glds16_v(p.K + (int64_t)b * p.bsk + head * 64 + ck * 8 + (int64_t)key * p.ldk, ks + i * 32 * 128);
            }
        }
    };
    auto vdma = [&](int tile, int stage) __attribute__((always_inline)) {
        char* vs = smem + (2 + stage) * STG + wave * 8 * 128;
        if ((tile + 1) * KT <= p.Nk) {
            const char* vt_ = vU + (int64_t)tile * 2 * v32b;
            glds16_s(vt_, vlane, vs);
            if (NW == 4) glds16_s(vt_ + v32b, vlane, vs + 32 * 128);
        } else {
#pragma unroll
            for (int i = 0; i < (NW == 4 ? 2 : 1); ++i) {
                const int key = min(tile * KT + i * 32 + lkey, p.Nk - 1);
                glds16_v(p.V + (int64_t)b * p.bsv + head * 64 + cv * 8 + (int64_t)key * p.ldv, vs + i * 32 * 128);
            }
        }
    };

    const int ksw = (r >> 1) & 7;
    int koff[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) koff[ks] = r * 128 + (((2 * ks + h) ^ ksw) << 4);
    const int vq = (lane & 15) >> 2;
    const int vsb = (vq >> 1) & 1;
    int voff[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
        const int c = 4 * dt + 2 * ((lane >> 4) & 1) + ((lane & 3) >> 1);
        voff[dt] = (4 * h + vq) * 128 + ((c ^ (vsb << 2)) << 4) + 8 * (lane & 1);
    }

    f32x16 ot[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int i = 0; i < 16; ++i) ot[dt][i] = 0.f;
    float m_run = 0.f, l_run = 0.f;
    // -m_run in 16 registers: the initial accumulator (srcC) of every Q.K^T chain; rebuilt only when m_run moves.
    // Built through asm so the compiler keeps the block instead of re-materialising it with 8 moves per tile.
    f32x16 cneg;
    auto set_cneg = [&]() __attribute__((always_inline)) {
        typedef __attribute__((ext_vector_type(2))) float f32x2;
        const f32x2 negm2 = {-m_run, -m_run};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            f32x2 c2;
            asm volatile("v_mov_b64 %0, %1" : "=v"(c2) : "v"(negm2));
            cneg[2 * j] = c2[0];
            cneg[2 * j + 1] = c2[1];
        }
    };

    const int NT = (p.Nk + KT - 1) / KT;
    const bool ragged = (p.Nk % KT) != 0;
#ifdef HK_ATTN_STAMP
    unsigned long long tacc[6] = {0, 0, 0, 0, 0, 0}, tprev = 0;
#define HK_STAMP_BEGIN() tprev = __builtin_amdgcn_s_memtime()
#define HK_STAMP(i) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); tacc[i] += now_ - tprev; tprev = now_; } while (0)
#else
#define HK_STAMP_BEGIN() do {} while (0)
#define HK_STAMP(i) do {} while (0)
#endif

    auto rowmax = [&](const f32x16 (&st)[2]) __attribute__((always_inline)) {
        float mx = fmaxf(st[0][0], st[0][1]);
#pragma unroll
        for (int i = 2; i < 16; i += 2) mx = fmaxf(mx, fmaxf(st[0][i], st[0][i + 1]));
#pragma unroll
        for (int i = 0; i < 16; i += 2) mx = fmaxf(mx, fmaxf(st[1][i], st[1][i + 1]));
        return half_swap_max(mx);
    };
    auto mask_ragged = [&](f32x16 (&st)[2], const int t) __attribute__((always_inline)) {
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int key = t * KT + 32 * kt + (i & 3) + 8 * (i >> 2) + 4 * h;
                if (key >= p.Nk) st[kt][i] = NEG_BIG;
            }
    };

    if (!active) {
        // A wave of the last query block with no rows of its own: it still owns one eighth / quarter of every K/V tile.
        // Same loads, waits and barriers as the computing waves below, nothing else.
        kdma(0, 0);
        vdma(0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (NT > 1) kdma(1, 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int t = 0; t < NT; ++t) {
            if (t + 1 < NT) {
                if (t + 2 < NT) kdma(t + 2, t & 1);
                vdma(t + 1, (t & 1) ^ 1);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        return;
    }

    // ---- prologue: K(0), V(0) -> stage 0; S(0) with the true row maximum as m_run; K(1) on its way meanwhile ----
    f32x16 sa[2], sb[2];
    kdma(0, 0);
    vdma(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (NT > 1) kdma(1, 1);
    {
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
            for (int i = 0; i < 16; ++i) sa[kt][i] = 0.f;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const bf16x8 kf = *reinterpret_cast<const bf16x8*>(smem + kt * 32 * 128 + koff[s]);
                sa[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[s], sa[kt], 0, 0, 0);
            }
        }
        if (ragged && NT == 1) mask_ragged(sa, 0);
        const float mx = rowmax(sa);
        m_run = mx;
        set_cneg();
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int i = 0; i < 16; ++i) sa[kt][i] -= mx;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
    const bf16x2 ones2 = __builtin_bit_cast(bf16x2, 0x3f803f80u);

    // sc: S(t) (raw, relative to m_run); sn: receives S(t + 1).  PAR = t & 1.
    auto iter = [&](const int t, f32x16 (&sc)[2], f32x16 (&sn)[2], auto has_next_c, auto next_ragged_c, auto par_c) __attribute__((always_inline)) {
        constexpr bool HAS_NEXT = decltype(has_next_c)::value;
        constexpr bool NEXT_RAGGED = decltype(next_ragged_c)::value;
        constexpr int PAR = decltype(par_c)::value;
        const char* ks = smem + (PAR ^ 1) * STG;        // K(t + 1)
        bf16x8 kf[8];   // K fragment g: key half g & 1, k-step g >> 1; loaded two MFMAs ahead
        HK_STAMP_BEGIN();
        if (HAS_NEXT) {
            kf[0] = *reinterpret_cast<const bf16x8*>(ks + koff[0]);
            kf[1] = *reinterpret_cast<const bf16x8*>(ks + 32 * 128 + koff[0]);
            if (t + 2 < NT) kdma(t + 2, PAR);
            vdma(t + 1, PAR ^ 1);
        }
        HK_STAMP(0);
        uint32_t pw[16];
        float rs0, rs1;
        auto exp_chunk = [&](const int g) __attribute__((always_inline)) {  // score registers 4g .. 4g+3 of the 32
            const int kt = g >> 2, i0 = 4 * (g & 3);
            const float e0 = __builtin_amdgcn_exp2f(sc[kt][i0]);
            const float e1 = __builtin_amdgcn_exp2f(sc[kt][i0 + 1]);
            const float e2 = __builtin_amdgcn_exp2f(sc[kt][i0 + 2]);
            const float e3 = __builtin_amdgcn_exp2f(sc[kt][i0 + 3]);
            pw[2 * g] = pack_bf16x2(e0, e1);
            pw[2 * g + 1] = pack_bf16x2(e2, e3);
            rs0 = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, pw[2 * g]), ones2, rs0, false);
            rs1 = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, pw[2 * g + 1]), ones2, rs1, false);
        };
        rs0 = 0.f;
        rs1 = 0.f;
        if (HAS_NEXT) {
            if (PRIO & 2) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                if (g + 2 < 8) kf[g + 2] = *reinterpret_cast<const bf16x8*>(ks + ((g + 2) & 1) * 32 * 128 + koff[(g + 2) >> 1]);
                sn[g & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[g], qf[g >> 1], g < 2 ? cneg : sn[g & 1], 0, 0, 0);
                exp_chunk(g);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (PRIO & 2) __builtin_amdgcn_s_setprio(0);
            if (NEXT_RAGGED) mask_ragged(sn, t + 1);
        } else {
#pragma unroll
            for (int g = 0; g < 8; ++g) exp_chunk(g);
        }
        // V^T fragments by inline asm: the compiler orders a builtin LDS read after every LDS-DMA in flight (s_waitcnt
        // vmcnt(0) in front of the first read, i.e. the K(t+2) / V(t+1) loads just issued would have to land before P.V
        // starts).  These reads touch stage PAR only, the loads in flight write the other stage.  Fragment (s, dt) =
        // two 8-byte transposed reads; the waits below are counted by hand (LDS returns in order) and carry the
        // fragment registers as operands so no MFMA can move above its wait.
        u32x2 vr[4][2][2];
#define HK_VREAD(S, DT, U)                                                                                   \
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2"                                                   \
                     : "=v"(vr[S][DT][U])                                                                    \
                     : "v"(voff[DT]), "n"((2 + PAR) * STG + (S) * 16 * 128 + (U) * 8 * 128))
#define HK_VREAD4(S) HK_VREAD(S, 0, 0); HK_VREAD(S, 0, 1); HK_VREAD(S, 1, 0); HK_VREAD(S, 1, 1)
#define HK_VWAIT(N, S)                                                                                       \
        asm volatile("s_waitcnt lgkmcnt(" #N ")"                                                             \
                     : "+v"(vr[S][0][0]), "+v"(vr[S][0][1]), "+v"(vr[S][1][0]), "+v"(vr[S][1][1]))
        HK_STAMP(1);
        HK_VREAD4(0);
        HK_VREAD4(1);
        float rs = rs0 + rs1;
        if (__builtin_amdgcn_ballot_w64(!(rs <= OVERFLOW_SUM)) != 0) {
            const float delta = fmaxf(rowmax(sc), 0.f);
            const float alpha = __builtin_amdgcn_exp2f(-delta);
            m_run += delta;
            set_cneg();
            l_run *= alpha;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int i = 0; i < 16; ++i) ot[dt][i] *= alpha;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    sc[kt][i] -= delta;
                    if (HAS_NEXT) sn[kt][i] -= delta;
                }
            rs0 = 0.f;
            rs1 = 0.f;
#pragma unroll
            for (int g = 0; g < 8; ++g) exp_chunk(g);
            rs = rs0 + rs1;
        }
        l_run += rs;

        if (PRIO & 1) __builtin_amdgcn_s_setprio(2);
        auto pv = [&](const int s_) __attribute__((always_inline)) {
            u32x4 w;
#pragma unroll
            for (int j = 0; j < 4; ++j) w[j] = pw[4 * s_ + j];
            const bf16x8 pf = __builtin_bit_cast(bf16x8, w);
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                u32x4 v4;
                v4[0] = vr[s_][dt][0][0];
                v4[1] = vr[s_][dt][0][1];
                v4[2] = vr[s_][dt][1][0];
                v4[3] = vr[s_][dt][1][1];
                ot[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, v4), pf, ot[dt], 0, 0, 0);
            }
        };
        HK_VWAIT(4, 0);
        pv(0);
        __builtin_amdgcn_sched_barrier(0);
        HK_VREAD4(2);
        HK_VWAIT(4, 1);
        pv(1);
        __builtin_amdgcn_sched_barrier(0);
        HK_VREAD4(3);
        HK_VWAIT(4, 2);
        pv(2);
        __builtin_amdgcn_sched_barrier(0);
        HK_VWAIT(0, 3);
        pv(3);
        __builtin_amdgcn_sched_barrier(0);
#undef HK_VREAD
#undef HK_VREAD4
#undef HK_VWAIT
        if (PRIO & 1) __builtin_amdgcn_s_setprio(0);
        HK_STAMP(2);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        HK_STAMP(3);
        __syncthreads();
        HK_STAMP(4);
    };

    using T_ = std::true_type;
    using F_ = std::false_type;
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    auto run = [&](const int t, auto has_next_c, auto next_ragged_c) __attribute__((always_inline)) {
        if (t & 1) iter(t, sb, sa, has_next_c, next_ragged_c, S1{});
        else iter(t, sa, sb, has_next_c, next_ragged_c, S0{});
    };
    const int NS = ragged ? NT - 2 : NT - 1;   // iterations [0, NS) produce a whole next tile
    int t = 0;
    for (; t + 1 < NS; t += 2) {
        iter(t, sa, sb, T_{}, F_{}, S0{});
        iter(t + 1, sb, sa, T_{}, F_{}, S1{});
    }
    if (t < NS) {
        run(t, T_{}, F_{});
        ++t;
    }
    if (ragged && NT >= 2) run(NT - 2, T_{}, T_{});
    run(NT - 1, F_{}, F_{});

#ifdef HK_ATTN_STAMP
    if (p.stamp && lane == 0) {
        unsigned long long* o = p.stamp + ((size_t)blockIdx.x * NW + wave) * 8;
        for (int i = 0; i < 5; ++i) o[i] = tacc[i];
        o[5] = (unsigned long long)NT;
    }
#endif
#undef HK_STAMP
#undef HK_STAMP_BEGIN
    // ---- normalise and store (as above) ----
    const float l_tot = half_swap_sum(l_run);
    const float inv = 1.0f / l_tot;
    if (p.o_vec16) {
        char* stg = smem + wave * 4096;
        const int sw = (r >> 1) & 7;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                u32x2 o;
                o[0] = pack_bf16x2(ot[dt][4 * g] * inv, ot[dt][4 * g + 1] * inv);
                o[1] = pack_bf16x2(ot[dt][4 * g + 2] * inv, ot[dt][4 * g + 3] * inv);
                *reinterpret_cast<u32x2*>(stg + r * 128 + (((4 * dt + g) ^ sw) << 4) + 8 * h) = o;
            }
        __bf16* ob = p.O + (int64_t)b * p.bso + head * 64 + (lane & 7) * 8;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int row = it * 8 + (lane >> 3);
            const u32x4 o = *reinterpret_cast<const u32x4*>(stg + row * 128 + (((lane & 7) ^ ((row >> 1) & 7)) << 4));
            if (q0 + row < p.Nq) *reinterpret_cast<u32x4*>(ob + (int64_t)(q0 + row) * p.ldo) = o;
        }
        return;
    }
    const int qrow = q0 + r;
    if (qrow < p.Nq) {
        __bf16* op = p.O + (int64_t)b * p.bso + (int64_t)qrow * p.ldo + head * 64 + 4 * h;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                u32x2 o;
                o[0] = pack_bf16x2(ot[dt][4 * g] * inv, ot[dt][4 * g + 1] * inv);
                o[1] = pack_bf16x2(ot[dt][4 * g + 2] * inv, ot[dt][4 * g + 3] * inv);
                *reinterpret_cast<u32x2*>(op + 32 * dt + 8 * g) = o;
            }
    }
}

}  // namespace

int hk_attn_causal_dispatch(const hk_attn_desc* d, void* stream);  // hk_attn_causal.hip

extern "C" int hk_flash_attn_bf16(const hk_attn_desc* d, void* stream) {
    HK_REQUIRE(d != nullptr, "hk_flash_attn_bf16: null descriptor");
    HK_REQUIRE(d->Q && d->K && d->V && d->O, "hk_flash_attn_bf16: null operand pointer");
    HK_REQUIRE(d->B > 0 && d->H > 0 && d->Nq > 0 && d->Nk > 0, "hk_flash_attn_bf16: empty problem B=%d H=%d Nq=%d Nk=%d",
               d->B, d->H, d->Nq, d->Nk);
    HK_REQUIRE(((uintptr_t)d->Q | (uintptr_t)d->K | (uintptr_t)d->V | (uintptr_t)d->O) % 16 == 0,
               "hk_flash_attn_bf16: operands must be 16-byte aligned");
    HK_REQUIRE((d->ldq | d->ldk | d->ldv | d->bsq | d->bsk | d->bsv) % 8 == 0 && (d->ldo | d->bso) % 4 == 0,
               "hk_flash_attn_bf16: strides must keep 16-byte (Q,K,V) / 8-byte (O) alignment");
    HK_REQUIRE(d->head_dim == 0 || d->head_dim == 64 || d->head_dim == 128,
               "hk_flash_attn_bf16: head_dim %d not supported (64 or 128)", d->head_dim);
    HK_REQUIRE(d->key_splits <= 1 || (d->causal == 0 && d->head_dim != 128),
               "hk_flash_attn_bf16: key_splits needs head size 64 and a non-causal problem");
    if (d->causal != 0 || d->head_dim == 128) return hk_attn_causal_dispatch(d, stream);
    AttnParams p;
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
    p.splits = 1;
    p.chunk = 0;
    p.lse = nullptr;
    int nsplit = 1;
    if (d->key_splits > 1) {
        HK_REQUIRE(d->key_bias == nullptr && d->o_part != nullptr && d->lse_part != nullptr && d->key_splits <= 64 &&
                       (uintptr_t)d->o_part % 16 == 0 && (uintptr_t)d->lse_part % 4 == 0 && d->ldo % 8 == 0 && d->bso % 8 == 0,
                   "hk_flash_attn_bf16: key_splits needs no key_bias, o_part / lse_part buffers and 16-byte aligned O rows");
        const int chunk = ((d->Nk + d->key_splits - 1) / d->key_splits + KT - 1) / KT * KT;   // whole tiles per slice
        nsplit = (d->Nk + chunk - 1) / chunk;                                                   // no empty slice
        if (nsplit > 1) {
            p.splits = nsplit;
            p.chunk = chunk;
            p.lse = d->lse_part;
            p.O = reinterpret_cast<__bf16*>(d->o_part);     // [B * nsplit][Nq][H * 64]
            p.ldo = (int64_t)d->H * 64;
            p.bso = (int64_t)d->Nq * d->H * 64;
        }
    }
    static const int prio = getenv("HK_ATTN_PRIO") ? atoi(getenv("HK_ATTN_PRIO")) : 1;
    p.prio = prio;
    p.stamp = nullptr;
#ifdef HK_ATTN_STAMP
    if (getenv("HK_ATTN_STAMP_PTR")) p.stamp = reinterpret_cast<unsigned long long*>(strtoull(getenv("HK_ATTN_STAMP_PTR"), nullptr, 0));
#endif
    // HK_ATTN_EPI=0: A/B aid, per-lane 8-byte stores everywhere
    static const int epi = getenv("HK_ATTN_EPI") ? atoi(getenv("HK_ATTN_EPI")) : 1;
    p.o_vec16 = (epi != 0 && p.ldo % 8 == 0 && p.bso % 8 == 0) ? 1 : 0;
    const long nwg = (long)p.nqb * d->H * d->B * nsplit;
    HK_REQUIRE(nwg < (1L << 31), "hk_flash_attn_bf16: grid too large");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    // HK_ATTN_MODE: softmax variant; HK_ATTN_OCC: waves per SIMD the register allocation targets;
    // HK_ATTN_LOADER: 0 register-staged K/V tiles, 1 LDS-DMA (see the kernel).
    static const int mode = getenv("HK_ATTN_MODE") ? atoi(getenv("HK_ATTN_MODE")) : 1;
    static const int occ = getenv("HK_ATTN_OCC") ? atoi(getenv("HK_ATTN_OCC")) : 3;
    static const int ldr = getenv("HK_ATTN_LOADER") ? atoi(getenv("HK_ATTN_LOADER")) : 1;
    const dim3 grid((unsigned)nwg), block(256);
#define HK_ATTN_LAUNCH(BIAS, M, O, L) hipLaunchKernelGGL((hk_attn_kernel<BIAS, M, O, L>), grid, block, 0, s, p)
    // key-bias (Q-Former text mask) launches: the tuned loader / softmax form too; HK_ATTN_BIAS=0: the first-round variant
    static const int bvar = getenv("HK_ATTN_BIAS") ? atoi(getenv("HK_ATTN_BIAS")) : 1;
    // HK_ATTN_PIPE: 2 / 8 = the software-pipelined kernel with 4 / 8 waves per workgroup (no key bias); 0 (default) = the
    // per-tile kernel.  Measured on one device, B=32 N=3136: per-tile 1356 us, pipelined 1457 (4 waves) / 1505 (8 waves):
    // the loop is bound by vector issue slots per tile, not by latency, so overlapping a wave's own phases buys nothing.
    static const int pipe = getenv("HK_ATTN_PIPE") ? atoi(getenv("HK_ATTN_PIPE")) : 0;
    if (!d->key_bias && nsplit == 1 && (pipe == 2 || pipe == 8)) {
        const int nw = pipe == 8 ? 8 : 4;
        p.nqb = (d->Nq + nw * 32 - 1) / (nw * 32);
        const dim3 g2((unsigned)((long)p.nqb * d->H * d->B)), b2(nw * 64);
        if (nw == 4 && prio == 0) hipLaunchKernelGGL((hk_attn_pipe_kernel<4, 0>), g2, b2, 0, s, p);
        else if (nw == 4) hipLaunchKernelGGL((hk_attn_pipe_kernel<4, 1>), g2, b2, 0, s, p);
        else if (prio == 0) hipLaunchKernelGGL((hk_attn_pipe_kernel<8, 0>), g2, b2, 0, s, p);
        else hipLaunchKernelGGL((hk_attn_pipe_kernel<8, 1>), g2, b2, 0, s, p);
    }
    else if (nsplit > 1) hipLaunchKernelGGL((hk_attn_kernel<false, 1, 3, 1, true>), grid, block, 0, s, p);
    else if (d->key_bias && bvar == 0) HK_ATTN_LAUNCH(true, 0, 1, 0);
    else if (d->key_bias) HK_ATTN_LAUNCH(true, 1, 2, 1);
    else if (mode == 2 && occ == 3) HK_ATTN_LAUNCH(false, 2, 3, 1);
    else if (mode == 2) HK_ATTN_LAUNCH(false, 2, 2, 1);
    else if (ldr == 2 && mode == 1 && occ == 3) HK_ATTN_LAUNCH(false, 1, 3, 2);
    else if (ldr == 2 && mode == 1) HK_ATTN_LAUNCH(false, 1, 2, 2);
    else if (ldr == 1 && mode == 1 && occ == 3) HK_ATTN_LAUNCH(false, 1, 3, 1);
    else if (ldr == 1 && mode == 1) HK_ATTN_LAUNCH(false, 1, 2, 1);
    else if (ldr == 1 && occ == 3) HK_ATTN_LAUNCH(false, 0, 3, 1);
    else if (mode == 1 && occ == 3) HK_ATTN_LAUNCH(false, 1, 3, 0);
    else if (mode == 1) HK_ATTN_LAUNCH(false, 1, 2, 0);
    else if (occ == 3) HK_ATTN_LAUNCH(false, 0, 3, 0);
    else HK_ATTN_LAUNCH(false, 0, 2, 0);
#undef HK_ATTN_LAUNCH
    HK_CHECK_LAUNCH("hk_flash_attn_bf16");
    if (nsplit > 1) {
        const long nthr = (long)d->B * d->Nq * d->H * 8;
        hipLaunchKernelGGL(hk_attn_combine_kernel, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, s,
                           reinterpret_cast<const __bf16*>(d->o_part), d->lse_part, reinterpret_cast<__bf16*>(d->O), nsplit,
                           d->B, d->H, d->Nq, d->ldo, d->bso);
        HK_CHECK_LAUNCH("hk_flash_attn_bf16(combine)");
    }
    return HK_OK;
}
