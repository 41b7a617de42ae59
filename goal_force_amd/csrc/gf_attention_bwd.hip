// gf_attention_bwd.hip — flash-attention backward (dQ, dK, dV), head_dim 128, non-causal, bf16 in/out, fp32 accumulate.
// Used by the ControlNet training step (reference: training_loss, src/goal_force/wan_video_new.py:180-193, whose
// loss.backward() reaches F.scaled_dot_product_attention's backward through every DiT / ControlNet block, DIT:28-61).
//
// With P = softmax(scale Q K^T) rebuilt from the forward's log-sum-exp (P = exp2(c S - lse), c = scale log2 e) and
// delta = rowsum(dO . O):   dV = P^T dO,   dP = dO V^T,   dS = P . (dP - delta),   dQ = scale dS K,   dK = scale dS^T Q.
// Two kernels, no atomics (S and dP are recomputed in both; 7 tile products instead of the minimal 5):
//   * attn_bwd_dq_kernel : one wave owns 32 QUERIES (on the lane, as in the forward) and walks the KV tiles;
//   * attn_bwd_dkv_kernel: one wave owns 32 KEYS (on the lane) and walks the query tiles.
// Same operand algebra as the forward (gf_attention.hip): swapped products with v_mfma_f32_32x32x16_bf16 so the owned
// row is on the lane and the per-row scalars (lse, delta) are per-lane (dQ) or per-register broadcast reads (dK/dV); the
// fp32 accumulator converted pairwise to bf16 is the B operand of the second product; row fragments by ds_read_b128
// and transposed fragments by ds_read_b64_tr_b16 from ONE XOR-swizzled 64 x 128 tile image staged by LDS-DMA.
// Round-1 shape: tiles double buffered (the DMA of tile t+1 flies during the products of tile t, one barrier per tile),
// plain per-tile product order — the forward's slot pipeline is the template for the next step.
#include "gf_common.h"
#include <type_traits>

// Variant switches of the tile bodies (tools/attnbwd_ab.py builds the combinations side by side; round 3, S = 32760 x 40 heads,
// one process, profiles/r03/attnbwd_ab1.log):
//   GF_BWD_SEED   1: the score / dP accumulator chains START from -lse / -delta and the register-resident operand is pre-scaled by
//                    scale log2 e (p = exp2(acc), dS = p * dP': one exp2, one multiply and half a pack per score instead of fma /
//                    exp2 / subtract / multiply / convert); 0: zero start, p = exp2(fma(S, c, -lse)), dS = p * (dP - delta).
//                    MEASURED SLOWER (100 ms against 86): the 32 seeded accumulators are live before their chains start, which
//                    pushes the dQ kernel over the 256 registers of two waves per SIMD (26 spilled dwords reloaded every tile).
//   GF_BWD_HALVES 1: the tile's two 32-row halves one after the other (32 score / dP accumulators live instead of 64): 88 ms
//                    against 86 — the compiler re-interleaves the halves anyway and the pressure stays.
// The ragged last tile is the only one that pays for the bounds compare / select (no effect on the time either: 86.1 against
// 85.4 ms for round 2's loop) — the kernels are not VALU-bound; what they lack is the forward's pinned tile pipeline.
#ifndef GF_BWD_SEED
#define GF_BWD_SEED 0
#endif
#ifndef GF_BWD_HALVES
#define GF_BWD_HALVES 0
#endif

namespace {

constexpr int KVB = 64, HD = 128;
constexpr int TILE_BYTES = KVB * HD * 2;   // 16 KiB
constexpr int DQ_THREADS = 512, DQ_ROWS = 256;     // dQ: 8 waves x 32 queries, two waves per SIMD (<= 256 registers)
constexpr int DKV_THREADS = 256, DKV_ROWS = 128;   // dK/dV: 4 waves x 32 keys, one wave per SIMD (~300 registers)
constexpr int BWD_LDS = 4 * TILE_BYTES + 4 * KVB * (int)sizeof(float);   // two stages of (two tiles + lse + delta)

struct BwdArgs {
    const u16 *q, *k, *v, *o, *dout;
    const float* lse;      // [q_len, heads]
    float* delta;          // [q_len, heads] workspace
    u16 *dq, *dk, *dv;
    int q_len, kv_len, heads;
    long q_stride, k_stride, v_stride, o_stride, do_stride, dq_stride, dk_stride, dv_stride;
    float scale, scale_log2e;
};

// delta[s, h] = sum_d dO[s, h, d] * O[s, h, d]
__global__ __launch_bounds__(256) void attn_bwd_delta_kernel(const BwdArgs p) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)p.q_len * p.heads) return;
    const int s = (int)(idx / p.heads), h = (int)(idx % p.heads);
    const u16* op = p.o + (long)s * p.o_stride + h * HD;
    const u16* dp = p.dout + (long)s * p.do_stride + h * HD;
    float acc = 0.f;
#pragma unroll 4
    for (int i = 0; i < HD / 8; ++i) {
        const u16x8 a = *reinterpret_cast<const u16x8*>(op + 8 * i);
        const u16x8 b = *reinterpret_cast<const u16x8*>(dp + 8 * i);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc += bf2f(a[e]) * bf2f(b[e]);
    }
    p.delta[idx] = acc;
}

__device__ __forceinline__ void dma16(const u16* g, GF_LDS char* l) {
    unsigned keep;
    const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)l);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(g), "s"(dst)
                 : "memory");
}

// 64 rows x 256 B of `base` (rows row0.., clamped to len-1; head column offset included in base) -> swizzled LDS image.
// NW waves: wave w fills row-groups j = (16/NW) w .. (4 rows x 256 B each); the DMA writes lane-linear, so the image's chunk
// swizzle off(row,ch) = 256 row + 16 (ch ^ (((row&3)<<2) | ((row>>2)&3))) is applied to the per-lane SOURCE chunk.
template <int NW>
__device__ __forceinline__ void stage_tile(const u16* base, long stride, int row0, int len, GF_LDS char* dst, int wave, int lane) {
    constexpr int PER = 16 / NW;
    const int dma_r = lane >> 4;
#pragma unroll
    for (int jj = 0; jj < PER; ++jj) {
        const int j = PER * wave + jj;
        const long rr = min(row0 + 4 * j + dma_r, len - 1);
        const int lch = (lane & 15) ^ ((dma_r << 2) | (j & 3));
        dma16(base + rr * stride + lch * 8, dst + j * 1024);
    }
}

struct FragOffsets {
    int row[8];      // row-read offsets of fragment kd (rows 0..31; +32*256 for rows 32..63)
    int tr[2][4];    // transposed-read offsets [hf][d]
};

__device__ __forceinline__ FragOffsets frag_offsets(int lane) {
    FragOffsets f;
    const int r = lane & 31, h = lane >> 5;
    const int sK = ((r & 3) << 2) | ((r >> 2) & 3);
#pragma unroll
    for (int kd = 0; kd < 8; ++kd) f.row[kd] = 256 * r + 16 * ((2 * kd + h) ^ sK);
    const int qd = (lane & 15) >> 2, pp = lane & 3;
    const int vcl = 2 * ((lane >> 4) & 1) + (pp >> 1);
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
        const int sV = (qd << 2) | ((2 * hf + h) & 3);
#pragma unroll
        for (int d = 0; d < 4; ++d) f.tr[hf][d] = 256 * (8 * hf + 4 * h + qd) + 16 * ((4 * d + vcl) ^ sV) + 8 * (pp & 1);
    }
    return f;
}

__device__ __forceinline__ bf16x8 tr_frag(GF_LDS char* buf, const FragOffsets& f, int d, int kt, int s) {
    const int imm = 256 * (32 * kt + 16 * s);
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((GF_LDS s16x4*)(buf + f.tr[0][d] + imm));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((GF_LDS s16x4*)(buf + f.tr[1][d] + imm));
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    const s16x8 vv = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, vv);
}

__device__ __forceinline__ void mfma32(f32x16& acc, const bf16x8& a, const bf16x8& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
}

__device__ __forceinline__ void zero16(f32x16& a) {
#pragma unroll
    for (int e = 0; e < 16; ++e) a[e] = 0.f;
}
__device__ __forceinline__ void splat16(f32x16& a, float v) {
#pragma unroll
    for (int e = 0; e < 16; ++e) a[e] = v;
}
// 8 fp32 -> one bf16x8 fragment, two values per v_cvt_pk_bf16_f32
__device__ __forceinline__ bf16x8 pack8(const float (&x)[8]) {
    const u32x4 w = {pack2bf(x[0], x[1]), pack2bf(x[2], x[3]), pack2bf(x[4], x[5]), pack2bf(x[6], x[7])};
    return __builtin_bit_cast(bf16x8, w);
}
// operand pre-scaled by c = scale * log2(e) (one rounding to bf16, as the forward's kernel 3 does with Q): the score accumulator
// is then directly the base-2 exponent
__device__ __forceinline__ bf16x8 scale8(const bf16x8& v, float c) {
    float x[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = (float)v[j] * c;
    return pack8(x);
}

// transposed accumulator acc[d][e] (lane = owned row, register e of block d = column 32d + (e&3) + 8(e>>2) + 4h) -> row of out
__device__ __forceinline__ void store_rows(u16* rowp, const f32x16 (&acc)[4], float mul, int h) {
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            u32x2 pk;
            pk[0] = pack2bf(acc[d][4 * rg + 0] * mul, acc[d][4 * rg + 1] * mul);
            pk[1] = pack2bf(acc[d][4 * rg + 2] * mul, acc[d][4 * rg + 3] * mul);
            *reinterpret_cast<u32x2*>(rowp + 4 * h + 32 * d + 8 * rg) = pk;
        }
}

// ---- dQ: wave owns queries q0 + r ------------------------------------------------------------------------------------
__global__ __launch_bounds__(DQ_THREADS, 2) void attn_bwd_dq_kernel(const BwdArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    GF_LDS char* lds = (GF_LDS char*)smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int nqb = (p.q_len + DQ_ROWS - 1) / DQ_ROWS;
    const int head = blockIdx.x / nqb, qb = blockIdx.x % nqb;
    const int q0 = qb * DQ_ROWS + wave * 32;
    const int qr = min(q0 + r, p.q_len - 1);

    bf16x8 qf[8], dof[8];
    {
        const u16* qp = p.q + (long)qr * p.q_stride + head * HD + 8 * h;
        const u16* dp = p.dout + (long)qr * p.do_stride + head * HD + 8 * h;
#pragma unroll
        for (int kd = 0; kd < 8; ++kd) {
            qf[kd] = *reinterpret_cast<const bf16x8*>(qp + 16 * kd);
            dof[kd] = *reinterpret_cast<const bf16x8*>(dp + 16 * kd);
        }
    }
    const float lse = p.lse[(long)qr * p.heads + head];
    const float dl = p.delta[(long)qr * p.heads + head];
    // Q pre-scaled by c = scale log2 e: S' = (c Q) K^T accumulated ON TOP OF -lse (the chain's initial accumulator) is the exponent
    // itself, and dP accumulated on top of -delta is (dP - delta): p = exp2(S'), dS = p * dP' — per score one v_exp_f32, one multiply
    // and half a v_cvt_pk instead of fma / exp2 / subtract / multiply / compare / select / convert
#if GF_BWD_SEED
#pragma unroll
    for (int kd = 0; kd < 8; ++kd) qf[kd] = scale8(qf[kd], p.scale_log2e);
#endif
    const float c = p.scale_log2e;
    (void)c;
    const FragOffsets fo = frag_offsets(lane);
    f32x16 dq[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) zero16(dq[d]);

    const int nt = (p.kv_len + KVB - 1) / KVB;
    auto stage = [&](int t) {   // stage s = t & 1: K at s*32K, V at s*32K + 16K
        GF_LDS char* b = lds + (t & 1) * 2 * TILE_BYTES;
        stage_tile<8>(p.k + head * HD, p.k_stride, t * KVB, p.kv_len, b, wave, lane);
        stage_tile<8>(p.v + head * HD, p.v_stride, t * KVB, p.kv_len, b + TILE_BYTES, wave, lane);
    };
    // MASKED: the last tile when kv_len is not a multiple of 64 (its staged rows past the end are clamped copies: their p is 0)
    auto tile = [&](int t, auto masked) {
        constexpr bool MASKED = decltype(masked)::value;
        GF_LDS char* kbuf = lds + (t & 1) * 2 * TILE_BYTES;
        GF_LDS char* vbuf = kbuf + TILE_BYTES;
        bf16x8 dsf[2][2];
        auto softmax_half = [&](int half, const f32x16& sc, const f32x16& dp) {
#pragma unroll
            for (int s8 = 0; s8 < 2; ++s8) {
                float x[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int e = 8 * s8 + j;
#if GF_BWD_SEED
                    float pr = __builtin_amdgcn_exp2f(sc[e]);
#else
                    float pr = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[e], c, -lse));
#endif
                    if constexpr (MASKED) pr = (t * KVB + 32 * half + (e & 3) + 8 * (e >> 2) + 4 * h < p.kv_len) ? pr : 0.f;
#if GF_BWD_SEED
                    x[j] = pr * dp[e];
#else
                    x[j] = pr * (dp[e] - dl);
#endif
                }
                dsf[half][s8] = pack8(x);
            }
        };
        auto init = [&](f32x16& sc, f32x16& dp) {
#if GF_BWD_SEED
            splat16(sc, -lse);
            splat16(dp, -dl);
#else
            zero16(sc);
            zero16(dp);
#endif
        };
#if GF_BWD_HALVES
        // one 32-key half at a time: only 32 score / dP accumulators are live beside dQ^T, Q and dO
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            f32x16 sc, dp;
            init(sc, dp);
#pragma unroll
            for (int kd = 0; kd < 8; ++kd) {
                const bf16x8 kk = *(GF_LDS bf16x8*)(kbuf + fo.row[kd] + half * 32 * 256);
                const bf16x8 vv = *(GF_LDS bf16x8*)(vbuf + fo.row[kd] + half * 32 * 256);
                mfma32(sc, kk, qf[kd]);      // S'^T[key, query] = c K Q^T - lse
                mfma32(dp, vv, dof[kd]);     // dP'^T[key, query] = V dO^T - delta
            }
            softmax_half(half, sc, dp);
        }
#else
        {
            f32x16 sc[2], dp[2];
            init(sc[0], dp[0]);
            init(sc[1], dp[1]);
#pragma unroll
            for (int kd = 0; kd < 8; ++kd) {
                const bf16x8 k0 = *(GF_LDS bf16x8*)(kbuf + fo.row[kd]);
                const bf16x8 k1 = *(GF_LDS bf16x8*)(kbuf + fo.row[kd] + 32 * 256);
                const bf16x8 v0 = *(GF_LDS bf16x8*)(vbuf + fo.row[kd]);
                const bf16x8 v1 = *(GF_LDS bf16x8*)(vbuf + fo.row[kd] + 32 * 256);
                mfma32(sc[0], k0, qf[kd]);
                mfma32(sc[1], k1, qf[kd]);
                mfma32(dp[0], v0, dof[kd]);
                mfma32(dp[1], v1, dof[kd]);
            }
            softmax_half(0, sc[0], dp[0]);
            softmax_half(1, sc[1], dp[1]);
        }
#endif
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int d = 0; d < 4; ++d) mfma32(dq[d], tr_frag(kbuf, fo, d, kt, s), dsf[kt][s]);   // dQ^T += K^T dS^T
    };
    stage(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const bool ragged = (p.kv_len % KVB) != 0;
    for (int t = 0; t < nt; ++t) {
        if (t + 1 < nt) stage(t + 1);          // lands while this tile is multiplied; its buffer was released by the last barrier
        if (ragged && t == nt - 1) tile(t, std::true_type{});
        else tile(t, std::false_type{});
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    if (q0 + r < p.q_len) store_rows(p.dq + (long)(q0 + r) * p.dq_stride + head * HD, dq, p.scale, h);
}

// ---- dK, dV: wave owns keys k0 + r -------------------------------------------------------------------------------------
// PART 0: dK and dV in one pass (≈320 registers: ONE wave per SIMD).  PART 1: dV only (S, dV: 2 products), PART 2: dK only (S, dP,
// dK: 3 products) — each fits 256 registers, so two workgroups share a CU (two waves per SIMD) like the dQ kernel; the pair
// recomputes S once more (5 products instead of 4) and still takes less time than PART 0: a lone wave per SIMD leaves the matrix
// pipe idle through every one of its own waits.
template <int PART>
__global__ __launch_bounds__(DKV_THREADS, PART == 0 ? 1 : 2) void attn_bwd_dkv_kernel(const BwdArgs p) {
    constexpr bool DO_V = PART != 2, DO_K = PART != 1;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    GF_LDS char* lds = (GF_LDS char*)smem;
    GF_LDS float* scal = (GF_LDS float*)(lds + 4 * TILE_BYTES);    // [stage][lse 64 | delta 64]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int nkb = (p.kv_len + DKV_ROWS - 1) / DKV_ROWS;
    const int head = blockIdx.x / nkb, kb = blockIdx.x % nkb;
    const int k0 = kb * DKV_ROWS + wave * 32;
    const int kr = min(k0 + r, p.kv_len - 1);

    bf16x8 kf[8], vf[8];
    {
        const u16* kp = p.k + (long)kr * p.k_stride + head * HD + 8 * h;
        const u16* vp = p.v + (long)kr * p.v_stride + head * HD + 8 * h;
#pragma unroll
        for (int kd = 0; kd < 8; ++kd) {
            kf[kd] = *reinterpret_cast<const bf16x8*>(kp + 16 * kd);
            if constexpr (DO_K) vf[kd] = *reinterpret_cast<const bf16x8*>(vp + 16 * kd);
        }
    }
    // K pre-scaled by c = scale log2 e (this wave's own 32 keys, once): with -lse of the tile's queries as the initial accumulator the
    // score chain ends on the exponent itself; dP starts from -delta.  (The UNSCALED keys are not needed: dK = scale dS^T Q.)
#if GF_BWD_SEED
#pragma unroll
    for (int kd = 0; kd < 8; ++kd) kf[kd] = scale8(kf[kd], p.scale_log2e);
#endif
    const float c = p.scale_log2e;
    (void)c;
    const FragOffsets fo = frag_offsets(lane);
    f32x16 dk[4], dv[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        zero16(dk[d]);
        zero16(dv[d]);
    }

    const int nt = (p.q_len + KVB - 1) / KVB;
    auto stage = [&](int t) {   // stage s = t & 1: Q at s*32K, dO at s*32K + 16K, lse/delta at scal + s*128
        GF_LDS char* b = lds + (t & 1) * 2 * TILE_BYTES;
        if (tid < KVB) {   // before the DMA pieces: the wait for these two loads must not cover the tiles in flight
            const long qi = min(t * KVB + tid, p.q_len - 1);
            scal[(t & 1) * 2 * KVB + tid] = p.lse[qi * p.heads + head];
            if constexpr (DO_K) scal[(t & 1) * 2 * KVB + KVB + tid] = p.delta[qi * p.heads + head];
        }
        stage_tile<4>(p.q + head * HD, p.q_stride, t * KVB, p.q_len, b, wave, lane);
        stage_tile<4>(p.dout + head * HD, p.do_stride, t * KVB, p.q_len, b + TILE_BYTES, wave, lane);
    };
    auto tile = [&](int t, auto masked) {
        constexpr bool MASKED = decltype(masked)::value;   // the last query tile when q_len % 64 != 0: clamped copies get p = 0
        GF_LDS char* qbuf = lds + (t & 1) * 2 * TILE_BYTES;
        GF_LDS char* dobuf = qbuf + TILE_BYTES;
        GF_LDS float* lse_s = scal + (t & 1) * 2 * KVB;
        GF_LDS float* dl_s = lse_s + KVB;
        bf16x8 pf[2][2], dsf[2][2];
        auto init = [&](int half, f32x16& sc, f32x16& dp, f32x16& lsev, f32x16& dlv) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int qi = 32 * half + 8 * g + 4 * h;      // 4 consecutive queries of registers 4g..4g+3
                const f32x4 l4 = *(GF_LDS f32x4*)(lse_s + qi);
                f32x4 d4 = {0.f, 0.f, 0.f, 0.f};
                if constexpr (DO_K) d4 = *(GF_LDS f32x4*)(dl_s + qi);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
#if GF_BWD_SEED
                    sc[4 * g + i] = -l4[i];
                    dp[4 * g + i] = -d4[i];
#else
                    sc[4 * g + i] = 0.f;
                    dp[4 * g + i] = 0.f;
                    (void)l4;
                    (void)d4;
                    (void)lsev;
                    (void)dlv;
#endif
                }
            }
        };
        auto softmax_half = [&](int half, const f32x16& sc, const f32x16& dp, const f32x16& lsev, const f32x16& dlv) {
#pragma unroll
            for (int s8 = 0; s8 < 2; ++s8) {
                float xp[8], xs[8];
#if !GF_BWD_SEED     // the rows' lse / delta are read where they are used (4 queries per ds_read_b128), not held across the chains
                f32x4 l4[2], d4[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
                for (int g2 = 0; g2 < 2; ++g2) {
                    const int qi = 32 * half + 8 * (2 * s8 + g2) + 4 * h;
                    l4[g2] = *(GF_LDS f32x4*)(lse_s + qi);
                    if constexpr (DO_K) d4[g2] = *(GF_LDS f32x4*)(dl_s + qi);
                }
#endif
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int e = 8 * s8 + j;
#if GF_BWD_SEED
                    float pr = __builtin_amdgcn_exp2f(sc[e]);
#else
                    float pr = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[e], c, -l4[j >> 2][j & 3]));
#endif
                    if constexpr (MASKED) pr = (t * KVB + 32 * half + 8 * (e >> 2) + 4 * h + (e & 3) < p.q_len) ? pr : 0.f;
                    xp[j] = pr;
#if GF_BWD_SEED
                    if constexpr (DO_K) xs[j] = pr * dp[e];
#else
                    if constexpr (DO_K) xs[j] = pr * (dp[e] - d4[j >> 2][j & 3]);
#endif
                }
                if constexpr (DO_V) pf[half][s8] = pack8(xp);
                if constexpr (DO_K) dsf[half][s8] = pack8(xs);
            }
        };
#if GF_BWD_HALVES
#pragma unroll
        for (int half = 0; half < 2; ++half) {      // one 32-query half at a time (register pressure, as in the dQ kernel)
            f32x16 sc, dp, lsev, dlv;
            init(half, sc, dp, lsev, dlv);
#pragma unroll
            for (int kd = 0; kd < 8; ++kd) {
                const bf16x8 qq = *(GF_LDS bf16x8*)(qbuf + fo.row[kd] + half * 32 * 256);
                mfma32(sc, qq, kf[kd]);     // S'[query, key] = c Q K^T - lse: lane = key, registers = queries
                if constexpr (DO_K) {
                    const bf16x8 dd = *(GF_LDS bf16x8*)(dobuf + fo.row[kd] + half * 32 * 256);
                    mfma32(dp, dd, vf[kd]);     // dP'[query, key] = dO V^T - delta
                }
            }
            softmax_half(half, sc, dp, lsev, dlv);
        }
#else
        {
            f32x16 sc[2], dp[2], lsev[2], dlv[2];
            init(0, sc[0], dp[0], lsev[0], dlv[0]);
            init(1, sc[1], dp[1], lsev[1], dlv[1]);
#pragma unroll
            for (int kd = 0; kd < 8; ++kd) {
                const bf16x8 q0f = *(GF_LDS bf16x8*)(qbuf + fo.row[kd]);
                const bf16x8 q1f = *(GF_LDS bf16x8*)(qbuf + fo.row[kd] + 32 * 256);
                mfma32(sc[0], q0f, kf[kd]);
                mfma32(sc[1], q1f, kf[kd]);
                if constexpr (DO_K) {
                    const bf16x8 d0f = *(GF_LDS bf16x8*)(dobuf + fo.row[kd]);
                    const bf16x8 d1f = *(GF_LDS bf16x8*)(dobuf + fo.row[kd] + 32 * 256);
                    mfma32(dp[0], d0f, vf[kd]);
                    mfma32(dp[1], d1f, vf[kd]);
                }
            }
            softmax_half(0, sc[0], dp[0], lsev[0], dlv[0]);
            softmax_half(1, sc[1], dp[1], lsev[1], dlv[1]);
        }
#endif
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    if constexpr (DO_V) mfma32(dv[d], tr_frag(dobuf, fo, d, kt, s), pf[kt][s]);    // dV^T += dO^T P
                    if constexpr (DO_K) mfma32(dk[d], tr_frag(qbuf, fo, d, kt, s), dsf[kt][s]);    // dK^T += Q^T dS
                }
    };
    stage(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const bool ragged = (p.q_len % KVB) != 0;
    for (int t = 0; t < nt; ++t) {
        if (t + 1 < nt) stage(t + 1);
        if (ragged && t == nt - 1) tile(t, std::true_type{});
        else tile(t, std::false_type{});
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    if (k0 + r < p.kv_len) {
        if constexpr (DO_K) store_rows(p.dk + (long)(k0 + r) * p.dk_stride + head * HD, dk, p.scale, h);
        if constexpr (DO_V) store_rows(p.dv + (long)(k0 + r) * p.dv_stride + head * HD, dv, 1.0f, h);
    }
}

}  // namespace

extern "C" GF_API int gf_flash_attn_bwd(const void* q, const void* k, const void* v, const void* o, const void* dout,
                                        const float* lse, float* delta_ws, void* dq, void* dk, void* dv, int64_t q_len,
                                        int64_t kv_len, int64_t heads, int64_t head_dim, int64_t q_stride, int64_t k_stride,
                                        int64_t v_stride, int64_t o_stride, int64_t do_stride, int64_t dq_stride,
                                        int64_t dk_stride, int64_t dv_stride, float scale, void* stream) {
    GF_CHECK_ARG(q && k && v && o && dout && lse && delta_ws && dq && dk && dv, "gf_flash_attn_bwd: null pointer");
    if (head_dim != HD) {
        gf_set_error("gf_flash_attn_bwd: head_dim=%ld unsupported (kernels are built for 128)", (long)head_dim);
        return GF_ERR_UNSUPPORTED;
    }
    GF_CHECK_ARG(q_len > 0 && kv_len > 0 && heads > 0 && q_len < (1 << 30) && kv_len < (1 << 30),
                 "gf_flash_attn_bwd: bad lengths q=%ld kv=%ld heads=%ld", (long)q_len, (long)kv_len, (long)heads);
    const int64_t strides[8] = {q_stride, k_stride, v_stride, o_stride, do_stride, dq_stride, dk_stride, dv_stride};
    for (int i = 0; i < 8; ++i)
        GF_CHECK_ARG(strides[i] % 8 == 0 && strides[i] >= heads * HD, "gf_flash_attn_bwd: strides must cover heads*128 and be multiples of 8");
    GF_CHECK_ARG(gf_aligned16(q) && gf_aligned16(k) && gf_aligned16(v) && gf_aligned16(o) && gf_aligned16(dout) &&
                     gf_aligned16(dq) && gf_aligned16(dk) && gf_aligned16(dv),
                 "gf_flash_attn_bwd: 16-byte alignment required");
    const int lds_bytes = BWD_LDS;
    static GfDeviceOnce once;
    hipError_t e = gf_once_per_device(once, [] {
        const void* fns[4] = {reinterpret_cast<const void*>(attn_bwd_dq_kernel), reinterpret_cast<const void*>(attn_bwd_dkv_kernel<0>),
                              reinterpret_cast<const void*>(attn_bwd_dkv_kernel<1>), reinterpret_cast<const void*>(attn_bwd_dkv_kernel<2>)};
        for (const void* fn : fns) {
            hipError_t r = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, BWD_LDS);
            if (r != hipSuccess) return r;
        }
        return hipSuccess;
    });
    if (e != hipSuccess) {
        gf_set_error("gf_flash_attn_bwd: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
        return GF_ERR_LAUNCH;
    }
    BwdArgs a;
    a.q = (const u16*)q; a.k = (const u16*)k; a.v = (const u16*)v; a.o = (const u16*)o; a.dout = (const u16*)dout;
    a.lse = lse; a.delta = delta_ws;
    a.dq = (u16*)dq; a.dk = (u16*)dk; a.dv = (u16*)dv;
    a.q_len = (int)q_len; a.kv_len = (int)kv_len; a.heads = (int)heads;
    a.q_stride = q_stride; a.k_stride = k_stride; a.v_stride = v_stride; a.o_stride = o_stride; a.do_stride = do_stride;
    a.dq_stride = dq_stride; a.dk_stride = dk_stride; a.dv_stride = dv_stride;
    a.scale = scale; a.scale_log2e = scale * 1.4426950408889634f;
    hipStream_t s = (hipStream_t)stream;
    const long nd = q_len * heads;
    hipLaunchKernelGGL(attn_bwd_delta_kernel, dim3((unsigned)((nd + 255) / 256)), dim3(256), 0, s, a);
    const unsigned nqb = (unsigned)((q_len + DQ_ROWS - 1) / DQ_ROWS), nkb = (unsigned)((kv_len + DKV_ROWS - 1) / DKV_ROWS);
    hipLaunchKernelGGL(attn_bwd_dq_kernel, dim3(nqb * (unsigned)heads), dim3(DQ_THREADS), lds_bytes, s, a);
    // GF_ATTN_BWD_FUSED_DKV=1: dK and dV in one pass at one wave per SIMD (the first version, A/B)
    const int fused = gf_options().bwd_fused_dkv.load(std::memory_order_relaxed);
    if (fused) {
        hipLaunchKernelGGL(attn_bwd_dkv_kernel<0>, dim3(nkb * (unsigned)heads), dim3(DKV_THREADS), lds_bytes, s, a);
    } else {
        hipLaunchKernelGGL(attn_bwd_dkv_kernel<1>, dim3(nkb * (unsigned)heads), dim3(DKV_THREADS), lds_bytes, s, a);
        hipLaunchKernelGGL(attn_bwd_dkv_kernel<2>, dim3(nkb * (unsigned)heads), dim3(DKV_THREADS), lds_bytes, s, a);
    }
    GF_CHECK_LAUNCH("gf_flash_attn_bwd");
    return GF_OK;
}
