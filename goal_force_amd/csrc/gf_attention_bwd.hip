// gf_attention_bwd.hip — flash-attention backward (dQ, dK, dV), head_dim 128, non-causal, bf16 in/out, fp32 accumulate.
// Used by the ControlNet training step (reference: training_loss, src/goal_force/wan_video_new.py:180-193, whose
// loss.backward() reaches F.scaled_dot_product_attention's backward through every DiT / ControlNet block, DIT:28-61).
//
// With P = softmax(scale Q K^T) rebuilt from the forward's log-sum-exp (P = exp2(c S - lse), c = scale log2 e) and
// delta = rowsum(dO . O):   dV = P^T dO,   dP = dO V^T,   dS = P . (dP - delta),   dQ = scale dS K,   dK = scale dS^T Q.
// Two kernels, no atomics (S and dP are recomputed in both; 7 tile products instead of the minimal 5): attn_bwd_dq16_kernel (a wave
// owns 32 QUERIES and walks the key tiles) and attn_bwd_dkv48_kernel (wave pairs own 48 KEYS and walk 32-query granules), both on
// v_mfma_f32_16x16x32_bf16; the design notes stand in front of them below.  The first kernels (32x32x16 MFMAs, separate dQ / dV / dK
// passes, 8 products), the 32-key dK/dV kernel and the pre-scaled-Q variant are NOT in this file:
// tools/patches/attention_bwd_experiments.patch re-creates them.
#include "gf_common.h"
#include <type_traits>

namespace {

constexpr int KVB = 64, HD = 128;
constexpr int TILE_BYTES = KVB * HD * 2;   // 16 KiB
constexpr int DQ_THREADS = 512, DQ_ROWS = 256;     // dQ: 8 waves x 32 queries, two waves per SIMD (<= 256 registers)

struct BwdArgs {
    const u16 *q, *k, *v, *o, *dout;
    const float* lse;      // [q_len, heads]
    float* delta;          // [q_len, heads] workspace
    u16 *dq, *dk, *dv;
    int q_len, kv_len, heads;
    long q_stride, k_stride, v_stride, o_stride, do_stride, dq_stride, dk_stride, dv_stride;
    float scale, scale_log2e;
};

// =====================================================================================================================
// The backward on v_mfma_f32_16x16x32_bf16 (the forward's kernel 3 algebra: 12 % less power per FLOP than 32x32x16 on
// random data) with 7 tile products instead of 8:
//   * ONE LDS image per operand tile serves row fragments (ds_read_b128: A = X[16 rows x 32 d]) and transposed fragments (two
//     ds_read_b64_tr_b16: A = X^T[16 d x 32 rows], rows in the order of a B operand built from two accumulator tiles): rows of 256
//     bytes, 16-byte chunk c of row r at c ^ 2 (r & 7) — both read shapes touch every bank once.  What limits these kernels is the
//     LDS-DMA fill rate (a CU takes ~25 GB/s while its MFMAs run): staging each operand once, in one layout, is what counts;
//   * every LDS-DMA is the buffer form with a per-lane offset computed once and the tile position in the descriptor (no VALU per
//     request; the first version spent 30 ms of 80 on address arithmetic); rows past the end of a ragged sequence arrive as zeros
//     (num_records), which also makes masking unnecessary: a padded key / query meets a zero row in the last product;
//   * dQ: a wave owns 32 queries (two 16-query blocks on the lane) and walks 64-key tiles — S^T = K Q^T, dP^T = V dO^T,
//     dQ^T += K^T dS^T: 96 MFMAs per tile;
//   * dK and dV in ONE kernel by wave PAIRS that own the same 48 keys (attn_bwd_dkv48_kernel) and walk 32-query granules:
//       wave A: S = Q K^T, P = exp2(c S - lse), P (fp32, accumulator layout) -> LDS hand-off, dV^T += dO^T P;
//       wave B: dP = dO V^T, dS = P (dP - delta) with the P wave A left one granule earlier, dK^T += Q^T dS.
//     A SIMD holds one A and one B wave, so the exp2 of one role overlaps the MFMAs of the other; S is computed once for dK and dV
//     (the split kernels computed it twice).  ONE barrier per granule publishes the landed DMA pieces and the hand-off and frees the
//     ring slots.  Q rows and dO rows in rings of 4 granules requested two ahead (counted vmcnt), the hand-off double buffered;
//   * the vector issue port is the scarce resource once the MFMAs flow (an MFMA holds it for 8 of its 16 cycles): per-lane LDS
//     pointers are pinned in registers and the loops unrolled over their ring slots (address = register + immediate; `lds + offset`
//     costs a v_add_u32 per read because the dynamic LDS base is a link-time symbol), accumulation chains start from 0 / -delta through
//     the first MFMA's C operand, and the file is built with -fno-slp-vectorize (v_pk_mul_f32 costs more beside MFMAs than two v_mul).
// Arithmetic per element as in the first kernels (p = exp2(fma(S, c, -lse)), dS = p (dP - delta), one rounding to bf16 per MFMA
// operand); the summation ORDER over keys / queries differs (16x16x32 adds 32 products per step), so results differ in the last bits.
constexpr int GR = 32;                         // queries per granule of the dK/dV kernel
constexpr int GR_BYTES = GR * HD * 2;          // 8 KiB: one operand array of one granule
constexpr int KV16_RING = 4;
constexpr int DQ16_STAGE = 2 * TILE_BYTES;     // K rows | V rows
constexpr int DQ16_LDS = 2 * DQ16_STAGE;

struct Bwd16Args {
    BwdArgs b;
    const float* sd;     // [heads][ngp][32 lse | 32 delta]
    int ngp;             // granule records per head
    const u16* qs;       // unused (null): the slot of the measured-and-dropped pre-scaled Q copy (tools/patches/attention_bwd_experiments.patch)
};

// delta[s, h] = sum_d dO[s, h, d] O[s, h, d], stored twice: [q_len, heads] (per-lane reads of the dQ kernel) and with lse as the
// dK/dV kernel's granule records sd[h][s / 32][32 x -lse | 32 x -delta] (zeros past q_len)
__global__ __launch_bounds__(256) void attn_bwd_delta16_kernel(const Bwd16Args a, float* sd) {
    const BwdArgs& p = a.b;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)a.ngp * GR * p.heads) return;
    const int s = (int)(idx / p.heads), h = (int)(idx % p.heads);
    float acc = 0.f, l = 0.f;
    if (s < p.q_len) {
        const u16* op = p.o + (long)s * p.o_stride + h * HD;
        const u16* dp = p.dout + (long)s * p.do_stride + h * HD;
#pragma unroll 4
        for (int i = 0; i < HD / 8; ++i) {
            const u16x8 x = *reinterpret_cast<const u16x8*>(op + 8 * i);
            const u16x8 y = *reinterpret_cast<const u16x8*>(dp + 8 * i);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc += bf2f(x[e]) * bf2f(y[e]);
        }
        p.delta[idx] = acc;
        l = p.lse[idx];
    }
    float* rec = sd + ((long)h * a.ngp + (s >> 5)) * (2 * GR) + (s & 31);
    rec[0] = -l;            // negated: the records are the INITIAL accumulators of the dK/dV kernel's S and dP chains
    rec[GR] = -acc;
}

typedef __attribute__((ext_vector_type(4))) unsigned u32x4s;
__device__ __forceinline__ void mfma16(f32x4& acc, const bf16x8& a, const bf16x8& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
}
// buffer descriptor over [base, base + bytes): a request (lane offset + 16 bytes) beyond it returns zeros
__device__ __forceinline__ u32x4s make_srd(const void* base, unsigned bytes) {
    const unsigned long b = (unsigned long)base;
    u32x4s s;
    s[0] = (unsigned)b;
    s[1] = (unsigned)(b >> 32) & 0xffffu;
    s[2] = bytes;
    s[3] = 0x00020000u;
    return s;
}
// LDS-DMA, buffer form: lane L's 16 (4) bytes at srd base + voff + soff land at l + 16 L (4 L).  M0 (the LDS destination) is a
// reserved register hipcc sets itself before the few instructions that read it (none in these kernels): set here, not restored.
__device__ __forceinline__ void dma16b(const u32x4s& srd, unsigned voff, unsigned soff, GF_LDS char* l) {
    const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)l);
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %3 offen lds" : : "v"(voff), "s"(srd), "s"(dst), "s"(soff) : "memory");
}
__device__ __forceinline__ void dma4b(const u32x4s& srd, unsigned voff, unsigned soff, GF_LDS char* l) {
    const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)l);
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dword %0, %1, %3 offen lds" : : "v"(voff), "s"(srd), "s"(dst), "s"(soff) : "memory");
}
__device__ __forceinline__ bf16x8 pack44(const f32x4& a, const f32x4& b) {
    const u32x4 w = {pack2bf(a[0], a[1]), pack2bf(a[2], a[3]), pack2bf(b[0], b[1]), pack2bf(b[2], b[3])};
    return __builtin_bit_cast(bf16x8, w);
}
// XCD-aware block order (as the forward): the blocks of one head run on one XCD, whose L2 then holds that head's streamed operands
__device__ __forceinline__ void head_block(int pid, int heads, int nblk, int& head, int& blk) {
    if ((heads & 7) == 0) {
        const int xcd = pid & 7, idx = pid >> 3;
        head = xcd + 8 * (idx / nblk);
        blk = idx % nblk;
    } else {
        head = pid / nblk;
        blk = pid % nblk;
    }
}
// descriptor over rows row0 .. len - 1 of a [len, stride] bf16 tensor, head column offset included: rows >= len read as zeros
__device__ __forceinline__ u32x4s rows_srd(const u16* base, long stride, int head, int row0, int len) {
    const int rem = len - row0;
    return make_srd(base + (long)row0 * stride + head * HD, rem > 0 ? (unsigned)rem * (unsigned)stride * 2u : 0u);
}
// The tile image: 32-row blocks of 256-byte rows, chunk c of row r at c ^ 2 (r & 7).
//   rows fragment (block, ks), lane (r = lane & 15, g = lane >> 4): X[16 blk + r][32 ks + 8 g ..]  = one b128 at row_off[ks] + 4096 blk
//   transposed fragment (32-row group, db): X^T[16 db + r][k], k position 8 g + i <-> row 4 g + i (i < 4) / 16 + 4 g + (i - 4): two
//   ds_read_b64_tr_b16 (a 16-lane group hands in 4 rows x 16 columns as 4-element pieces, lane i gets column i): lane i's piece is row
//   4 g + (i >> 2) (+ 16), columns 16 db + 4 (i & 3) ..: tr_off + 32 db (the swizzle moves whole 32-byte pairs: (2 db + c) ^ 2 x)
struct ImgOffsets {
    int row[4];
    int tr[8];       // first read (rows 4 g + ..); the second (rows 16 + 4 g + ..) is 4096 bytes on: same row & 7, same swizzle
};
__device__ __forceinline__ ImgOffsets img_offsets(int lane) {
    ImgOffsets f;
    const int r = lane & 15, g = lane >> 4;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) f.row[ks] = 256 * r + 16 * ((4 * ks + g) ^ (2 * (r & 7)));
    const int row = 4 * g + (r >> 2);
#pragma unroll
    for (int db = 0; db < 8; ++db) f.tr[db] = 256 * row + 16 * ((2 * db + ((r & 3) >> 1)) ^ (2 * (row & 7))) + 8 * (r & 1);
    return f;
}
__device__ __forceinline__ bf16x8 tr16_frag(GF_LDS char* blk32, const ImgOffsets& f, int db) {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((GF_LDS s16x4*)(blk32 + f.tr[db]));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((GF_LDS s16x4*)(blk32 + f.tr[db] + 4096));
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    const s16x8 vv = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, vv);
}
// per-lane source offset (bytes) of LDS-DMA piece `piece` (rows 4 piece .. 4 piece + 3) of a rows image: lane L fetches the chunk that
// belongs at its physical place (row 4 piece + (L >> 4), physical chunk L & 15)
__device__ __forceinline__ unsigned img_src_off(int piece, int lane, long stride) {
    const int row = 4 * piece + (lane >> 4);
    const int lch = (lane & 15) ^ (2 * (row & 7));
    return ((unsigned)row * (unsigned)stride + lch * 8) * 2u;
}

// ---- dQ: wave owns queries q0 .. q0 + 31 (block qb: query q0 + 16 qb + (lane & 15)) ---------------------------------------
__global__ __launch_bounds__(DQ_THREADS, 2) void attn_bwd_dq16_kernel(const Bwd16Args a) {
    const BwdArgs& p = a.b;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    GF_LDS char* lds = (GF_LDS char*)smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, g = lane >> 4;
    const int nqb = (p.q_len + DQ_ROWS - 1) / DQ_ROWS;
    int head, qblk;
    head_block(blockIdx.x, p.heads, nqb, head, qblk);
    const int q0 = qblk * DQ_ROWS + wave * 32;

    bf16x8 qf[2][4], dof[2][4];      // B operands: Q^T / dO^T [32 d x 16 queries]: lane = query column, 8 d at 32 ks + 8 g
    float lse[2], dl[2];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const int qr = min(q0 + 16 * qb + r, p.q_len - 1);
        const u16* qp = p.q + (long)qr * p.q_stride + head * HD + 8 * g;
        const u16* dp = p.dout + (long)qr * p.do_stride + head * HD + 8 * g;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            qf[qb][ks] = *reinterpret_cast<const bf16x8*>(qp + 32 * ks);
            dof[qb][ks] = *reinterpret_cast<const bf16x8*>(dp + 32 * ks);
        }
        lse[qb] = p.lse[(long)qr * p.heads + head];
        dl[qb] = p.delta[(long)qr * p.heads + head];
    }
    // the compiler must see these loads completed HERE (it cannot count the asm LDS-DMA requests: a vmcnt wait it placed at the first use
    // inside the loop would wait for the newest DMA batch as well)
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            asm volatile("" : "+v"(qf[qb][ks]));
            asm volatile("" : "+v"(dof[qb][ks]));
        }
        asm volatile("" : "+v"(lse[qb]), "+v"(dl[qb]));
    }
    f32x4 ndl[2];                    // -delta of this lane's query: the dP chains start from it (dS = P * chain result, one multiply per score)
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) ndl[qb] = f32x4{-dl[qb], -dl[qb], -dl[qb], -dl[qb]};
    const float c = p.scale_log2e;
    const ImgOffsets fo = img_offsets(lane);
    f32x4 dq[8][2];
#pragma unroll
    for (int db = 0; db < 8; ++db)
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) dq[db][qb] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nt = (p.kv_len + KVB - 1) / KVB;
    // a tile's two images are 32 pieces of 1 KiB: wave w stages pieces w and w + 8 of each (the swizzle of a row depends on row & 7:
    // the same per-lane offset serves both, 32 rows apart).  Two stages (a tile is ~2.4 us of MFMAs: one tile ahead is enough), 64 KiB:
    // with the loop unrolled by two every LDS address is a pinned per-lane pointer + an immediate (the dynamic LDS base is a link-time
    // symbol: `lds + offset` inside the loop cost a v_add per read, 72 per tile).
    const unsigned k_voff = img_src_off(wave, lane, p.k_stride), v_voff = img_src_off(wave, lane, p.v_stride);
    auto stage = [&](int t, int st) {
        GF_LDS char* b = lds + st * DQ16_STAGE + wave * 1024;
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {       // a descriptor per half tile: the SGPR offset of a request is not bounds-checked
            dma16b(rows_srd(p.k, p.k_stride, head, t * KVB + 32 * jj, p.kv_len), k_voff, 0u, b + jj * 8192);
            dma16b(rows_srd(p.v, p.v_stride, head, t * KVB + 32 * jj, p.kv_len), v_voff, 0u, b + TILE_BYTES + jj * 8192);
        }
    };
    GF_LDS char* prow[4];            // + stage * DQ16_STAGE (+ TILE_BYTES: V) + 4096 * key block
    GF_LDS char* ptr[8];             // + stage * DQ16_STAGE + 8192 * half (+ 4096: second read)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        prow[ks] = lds + fo.row[ks];
        asm volatile("" : "+v"(prow[ks]));
    }
#pragma unroll
    for (int db = 0; db < 8; ++db) {
        ptr[db] = lds + fo.tr[db];
        asm volatile("" : "+v"(ptr[db]));
    }
    // No masking of a ragged last tile: rows past kv_len arrive as zeros, so those keys' dS (finite: p = exp2(-lse), dP = 0) meets a
    // zero row of K in the last product.
    // One 32-key half of a tile; the order is pinned by scheduling barriers (left alone, hipcc reads a fragment, waits for it, issues its
    // two MFMAs, and so on — the matrix pipe idles through every LDS round trip).  F = 8 fragment registers, refilled in halves as soon as
    // the MFMAs that read them have been issued; the exp2 / dS arithmetic of key block 2 h is INTERLEAVED with the MFMAs of block 2 h + 1
    // (both waves of a SIMD leave the tile barrier together: phases of pure arithmetic would coincide and leave the matrix pipe idle):
    //   F = K, V row fragments of block 2 h (F[2 ks] = K, F[2 ks + 1] = V; requested inside the previous half's last product)
    //   8 MFMAs (ks 0, 1); F[0..3] <- block 2 h + 1;  8 MFMAs (ks 2, 3); F[4..7] <- block 2 h + 1
    //   8 MFMAs of block 2 h + 1 with the arithmetic of block 2 h between them; F[0..3] <- K^T fragments d blocks 0..3 (transposed reads)
    //   8 MFMAs; F[4..7] <- d blocks 4..7;  arithmetic of block 2 h + 1 (covers the reads);
    //   dQ^T: 8 MFMAs, F[0..3] <- next half's first fragments, 8 MFMAs, F[4..7] <- the rest.
    bf16x8 F[8];
#define DQ16_SB() __builtin_amdgcn_sched_barrier(0)
    auto load_kb = [&](int sto, int kb, int lo, int hi) __attribute__((always_inline)) {      // F[2 ks] = K, F[2 ks + 1] = V fragment (kb, ks); sto = stage offset
#pragma unroll
        for (int ks = lo; ks < hi; ++ks) {
            F[2 * ks] = *(GF_LDS bf16x8*)(prow[ks] + (sto + 4096 * kb));
            F[2 * ks + 1] = *(GF_LDS bf16x8*)(prow[ks] + (sto + TILE_BYTES + 4096 * kb));
        }
    };
    auto load_tr = [&](int sto, int h, int db) __attribute__((always_inline)) {
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((GF_LDS s16x4*)(ptr[db] + (sto + 8192 * h)));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((GF_LDS s16x4*)(ptr[db] + (sto + 8192 * h + 4096)));
        return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    };
    auto half = [&](int sto, int h, bool has_next) __attribute__((always_inline)) {
        u32x4 dsw[2];            // dS^T as B operand [32 keys x 16 queries] per query block: words 0, 1 = key block 2 h, words 2, 3 = 2 h + 1
        f32x4 sc[2][2], dp[2][2];
        const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
        (void)zero4;
        auto first = [&](int kbb, int lo, int hi) __attribute__((always_inline)) {     // (the chains start from 0 / -delta through the first MFMA's C operand)
#pragma unroll
            for (int ks = lo; ks < hi; ++ks)
#pragma unroll
                for (int qb = 0; qb < 2; ++qb) {
                    sc[kbb][qb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(F[2 * ks], qf[qb][ks], ks == 0 ? zero4 : sc[kbb][qb], 0, 0, 0);      // S^T[key 32 h + 16 kbb + 4 g + j, query 16 qb + r]
                    dp[kbb][qb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(F[2 * ks + 1], dof[qb][ks], ks == 0 ? ndl[qb] : dp[kbb][qb], 0, 0, 0);   // dP^T - delta
                }
        };
        auto softmax = [&](int kbb) __attribute__((always_inline)) {
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) {
                float x[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float pr = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[kbb][qb][j], c, -lse[qb]));
                    x[j] = pr * dp[kbb][qb][j];
                }
                dsw[qb][2 * kbb] = pack2bf(x[0], x[1]);
                dsw[qb][2 * kbb + 1] = pack2bf(x[2], x[3]);
            }
        };
        DQ16_SB();
        first(0, 0, 2);
        DQ16_SB();
        load_kb(sto, 2 * h + 1, 0, 2);
        DQ16_SB();
        first(0, 2, 4);
        DQ16_SB();
        load_kb(sto, 2 * h + 1, 2, 4);
        DQ16_SB();
        softmax(0);
        first(1, 0, 2);
#pragma unroll
        for (int i = 0; i < 8; ++i) {        // one MFMA, then its share of the 8 exp2 and ~30 other arithmetic instructions
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x400, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
        }
        DQ16_SB();
#pragma unroll
        for (int db = 0; db < 4; ++db) F[db] = load_tr(sto, h, db);
        DQ16_SB();
        first(1, 2, 4);
        DQ16_SB();
#pragma unroll
        for (int db = 4; db < 8; ++db) F[db] = load_tr(sto, h, db);
        DQ16_SB();
        softmax(1);
        DQ16_SB();
#pragma unroll
        for (int db = 0; db < 4; ++db)
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) mfma16(dq[db][qb], F[db], __builtin_bit_cast(bf16x8, dsw[qb]));     // dQ^T[d 16 db + 4 g + j, query] += K^T dS^T
        DQ16_SB();
        if (has_next) load_kb(sto, 2 * h + 2, 0, 2);
        DQ16_SB();
#pragma unroll
        for (int db = 4; db < 8; ++db)
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) mfma16(dq[db][qb], F[db], __builtin_bit_cast(bf16x8, dsw[qb]));
        DQ16_SB();
        if (has_next) load_kb(sto, 2 * h + 2, 2, 4);
        DQ16_SB();
    };
#define DQ16_WAIT_BARRIER(N) asm volatile("s_waitcnt vmcnt(" #N ")\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
    stage(0, 0);
#define DQ16_TILE(t, ST)                                                                                              \
    {   /* tile t landed (this wave's pieces: vmcnt(0); everybody's behind the barrier); the other stage is free */   \
        DQ16_WAIT_BARRIER(0);                                                                                         \
        if ((t) + 1 < nt) stage((t) + 1, 1 - (ST));                                                                   \
        load_kb((ST) * DQ16_STAGE, 0, 0, 4);                                                                          \
        half((ST) * DQ16_STAGE, 0, true);                                                                             \
        half((ST) * DQ16_STAGE, 1, false);                                                                            \
    }
#pragma unroll 1
    for (int t = 0; t < nt; t += 2) {
        DQ16_TILE(t, 0)
        if (t + 1 < nt) DQ16_TILE(t + 1, 1)
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the last requests must not outlive the workgroup's LDS
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const int qrow = q0 + 16 * qb + r;
        if (qrow < p.q_len) {
            u16* rowp = p.dq + (long)qrow * p.dq_stride + head * HD + 4 * g;
#pragma unroll
            for (int db = 0; db < 8; ++db) {
                u32x2 pk;
                pk[0] = pack2bf(dq[db][qb][0] * p.scale, dq[db][qb][1] * p.scale);
                pk[1] = pack2bf(dq[db][qb][2] * p.scale, dq[db][qb][3] * p.scale);
                *reinterpret_cast<u32x2*>(rowp + 16 * db) = pk;
            }
        }
    }
}

// ---- dK, dV: wave pair (w, w + NP) owns keys k0 .. k0 + 31 (block kb: key k0 + 16 kb + (lane & 15)) ---------------------------
#define KV16_WAIT_BARRIER(N) asm volatile("s_waitcnt vmcnt(" #N ")\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define KV16_SB() __builtin_amdgcn_sched_barrier(0)
// ---- dK, dV with THREE key blocks per wave (48 keys; 192 per workgroup of four pairs) -------------------------------------------
// With 48 keys per wave (the 32-key predecessor: tools/patches/attention_bwd_experiments.patch) every row / transposed fragment
// feeds three MFMAs instead of two and a granule's staged bytes serve 192 keys: two thirds of the LDS traffic, DMA bytes and barriers
// per MFMA.  96 accumulators + a 48-register operand leave no room to carry the second product over the barrier: an iteration is
// S / dP (24 MFMAs), the transposed reads, the arithmetic, dV / dK (24 MFMAs); wave B works one granule behind wave A as before.
// Measured: 33.5 ms against 34.3 (54.9 against 56.6 ms for the whole backward, one process) — both kernels sit at ~55 % MFMA-busy.
struct Kv48 {
    static constexpr int NP = 4, KB = 3, THREADS = 512, ROWS = 16 * KB * NP;       // 192 keys per workgroup
    static constexpr int Q = 0, DO = KV16_RING * GR_BYTES, H = 2 * KV16_RING * GR_BYTES;
    static constexpr int HBUF = NP * 2 * KB * 1024;        // one hand-off buffer: 4 pairs x 6 accumulator tiles x 1 KiB
    static constexpr int S = H + 2 * HBUF;
    static constexpr int LDS = S + KV16_RING * 2 * GR * (int)sizeof(float);
};
__global__ __launch_bounds__(Kv48::THREADS) void attn_bwd_dkv48_kernel(const Bwd16Args a) {
    using L = Kv48;
    constexpr int KB = L::KB;
    const BwdArgs& p = a.b;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    GF_LDS char* lds = (GF_LDS char*)smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, g = lane >> 4;
    const bool roleB = wave >= L::NP;
    const int pair = roleB ? wave - L::NP : wave;
    const int nkb = (p.kv_len + L::ROWS - 1) / L::ROWS;
    int head, kblk;
    head_block(blockIdx.x, p.heads, nkb, head, kblk);
    const int k0 = kblk * L::ROWS + pair * (16 * KB);

    bf16x8 own[KB][4];               // wave A: K^T, wave B: V^T [32 d x 16 keys] per key block
    {
        const u16* base = roleB ? p.v : p.k;
        const long stride = roleB ? p.v_stride : p.k_stride;
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
            const int kr = min(k0 + 16 * kb + r, p.kv_len - 1);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                own[kb][ks] = *reinterpret_cast<const bf16x8*>(base + (long)kr * stride + head * HD + 8 * g + 32 * ks);
        }
    }
#pragma unroll
    for (int kb = 0; kb < KB; ++kb)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) asm volatile("" : "+v"(own[kb][ks]));
    const float c = p.scale_log2e;
    const ImgOffsets fo = img_offsets(lane);
    f32x4 acc[8][KB];
#pragma unroll
    for (int db = 0; db < 8; ++db)
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) acc[db][kb] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int ng = (p.q_len + GR - 1) / GR;
    const u16* q_src = p.q;
    const long q_src_stride = p.q_stride;
    const unsigned q_voff = img_src_off(wave & 7, lane, q_src_stride), do_voff = img_src_off(wave & 7, lane, p.do_stride);
    const u32x4s srd_sd = make_srd(a.sd + (long)head * a.ngp * (2 * GR), 0xffffffffu);
    auto issue = [&](int i, int slot2) __attribute__((always_inline)) {        // granule i + 2 -> slot (i + 2) & 3
        const int ga = i + 2;
        GF_LDS char* dst = lds + slot2 * GR_BYTES + wave * 1024;
        dma16b(rows_srd(q_src, q_src_stride, head, ga * GR, p.q_len), q_voff, 0u, dst + L::Q);
        dma16b(rows_srd(p.dout, p.do_stride, head, ga * GR, p.q_len), do_voff, 0u, dst + L::DO);
        if (wave == 0) dma4b(srd_sd, (unsigned)lane * 4u, (unsigned)min(ga, a.ngp - 1) * (2u * GR * 4u), lds + L::S + slot2 * (2 * GR * 4));
    };
    GF_LDS char* prow[4];
    GF_LDS char* ptr[8];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        prow[ks] = lds + fo.row[ks];
        asm volatile("" : "+v"(prow[ks]));
    }
#pragma unroll
    for (int db = 0; db < 8; ++db) {
        ptr[db] = lds + fo.tr[db];
        asm volatile("" : "+v"(ptr[db]));
    }
    GF_LDS char* hbase = lds + L::H + pair * (2 * KB * 1024) + lane * 16;
    GF_LDS char* sbase = lds + L::S + 16 * g;
    asm volatile("" : "+v"(hbase), "+v"(sbase));
    auto tr_frag = [&](int off, int db) __attribute__((always_inline)) {
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((GF_LDS s16x4*)(ptr[db] + off));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((GF_LDS s16x4*)(ptr[db] + (off + 4096)));
        return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    };
    // wave A, iteration i = granule i (slot i & 3)
    auto stepA = [&](int i, auto slot_c) __attribute__((always_inline)) {
        constexpr int SLOT = decltype(slot_c)::value;
        if (i >= ng) return;
        bf16x8 F[8];
        f32x4 l4[2], sc[2][KB];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) F[2 * ks + qb] = *(GF_LDS bf16x8*)(prow[ks] + (L::Q + SLOT * GR_BYTES + 4096 * qb));
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) l4[qb] = *(GF_LDS f32x4*)(sbase + (SLOT * 256 + 64 * qb));
        KV16_SB();
        const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int qb = 0; qb < 2; ++qb)
#pragma unroll
                for (int kb = 0; kb < KB; ++kb)
                    sc[qb][kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(F[2 * ks + qb], own[kb][ks], ks == 0 ? zero4 : sc[qb][kb], 0, 0, 0);
        KV16_SB();
#pragma unroll
        for (int db = 0; db < 8; ++db) F[db] = tr_frag(L::DO + SLOT * GR_BYTES, db);
        KV16_SB();
        bf16x8 pf[KB];
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    sc[qb][kb][j] = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[qb][kb][j], c, l4[qb][j]));   // l4 = -lse
                *(GF_LDS f32x4*)(hbase + ((SLOT & 1) * L::HBUF + (2 * kb + qb) * 1024)) = sc[qb][kb];
            }
            pf[kb] = pack44(sc[0][kb], sc[1][kb]);
        }
        KV16_SB();
#pragma unroll
        for (int db = 0; db < 8; ++db)
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) mfma16(acc[db][kb], F[db], pf[kb]);                     // dV^T[d, key] += dO^T P
        KV16_SB();
    };
    // wave B, iteration i = granule i - 1 (slot (i - 1) & 3)
    auto stepB = [&](int i, auto slot_c) __attribute__((always_inline)) {
        constexpr int PREV = (decltype(slot_c)::value + 3) & 3;
        if (i < 1) return;
        bf16x8 F[8];
        f32x4 d4[2], dp[2][KB], pp[2][KB];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) F[2 * ks + qb] = *(GF_LDS bf16x8*)(prow[ks] + (L::DO + PREV * GR_BYTES + 4096 * qb));
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) d4[qb] = *(GF_LDS f32x4*)(sbase + (PREV * 256 + 128 + 64 * qb));
        KV16_SB();
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int qb = 0; qb < 2; ++qb)
#pragma unroll
                for (int kb = 0; kb < KB; ++kb)
                    dp[qb][kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(F[2 * ks + qb], own[kb][ks], ks == 0 ? d4[qb] : dp[qb][kb], 0, 0, 0);   // dP - delta
        KV16_SB();
#pragma unroll
        for (int db = 0; db < 8; ++db) F[db] = tr_frag(L::Q + PREV * GR_BYTES, db);
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) pp[qb][kb] = *(GF_LDS f32x4*)(hbase + ((PREV & 1) * L::HBUF + (2 * kb + qb) * 1024));
        KV16_SB();
        bf16x8 dsf[KB];
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
#pragma unroll
            for (int qb = 0; qb < 2; ++qb)
#pragma unroll
                for (int j = 0; j < 4; ++j) dp[qb][kb][j] *= pp[qb][kb][j];
            dsf[kb] = pack44(dp[0][kb], dp[1][kb]);
        }
        KV16_SB();
#pragma unroll
        for (int db = 0; db < 8; ++db)
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) mfma16(acc[db][kb], F[db], dsf[kb]);                    // dK^T[d, key] += Q^T dS
        KV16_SB();
    };
    issue(-2, 0);
    issue(-1, 1);
    auto wait_barrier = [&]() __attribute__((always_inline)) {
        if (wave == 0) KV16_WAIT_BARRIER(3);
        else KV16_WAIT_BARRIER(2);
    };
    typedef std::integral_constant<int, 0> S0;
    typedef std::integral_constant<int, 1> S1;
    typedef std::integral_constant<int, 2> S2;
    typedef std::integral_constant<int, 3> S3;
#define KV48_ITER(STEP, k, SC)          \
    if (i + k <= ng) {                  \
        wait_barrier();                 \
        issue(i + k, (k + 2) & 3);      \
        STEP(i + k, SC{});              \
    }
    if (!roleB) {
#pragma unroll 1
        for (int i = 0; i <= ng; i += 4) {
            KV48_ITER(stepA, 0, S0)
            KV48_ITER(stepA, 1, S1)
            KV48_ITER(stepA, 2, S2)
            KV48_ITER(stepA, 3, S3)
        }
    } else {
#pragma unroll 1
        for (int i = 0; i <= ng; i += 4) {
            KV48_ITER(stepB, 0, S0)
            KV48_ITER(stepB, 1, S1)
            KV48_ITER(stepB, 2, S2)
            KV48_ITER(stepB, 3, S3)
        }
    }
    const float mul = roleB ? p.scale : 1.0f;
    u16* outp = roleB ? p.dk : p.dv;
    const long ostride = roleB ? p.dk_stride : p.dv_stride;
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
        const int krow = k0 + 16 * kb + r;
        if (krow < p.kv_len) {
            u16* rowp = outp + (long)krow * ostride + head * HD + 4 * g;
#pragma unroll
            for (int db = 0; db < 8; ++db) {
                u32x2 pk;
                pk[0] = pack2bf(acc[db][kb][0] * mul, acc[db][kb][1] * mul);
                pk[1] = pack2bf(acc[db][kb][2] * mul, acc[db][kb][3] * mul);
                *reinterpret_cast<u32x2*>(rowp + 16 * db) = pk;
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

}  // namespace

static inline int64_t pad64(int64_t n) { return (n + 63) / 64 * 64; }
static inline int64_t align256(int64_t n) { return (n + 255) / 256 * 256; }

// bytes of the caller-owned workspace of gf_flash_attn_bwd: delta [q_len, heads] fp32 | granule records [heads][pad64(q_len)/32][64] fp32
extern "C" GF_API int64_t gf_flash_attn_bwd_workspace_bytes(int64_t q_len, int64_t kv_len, int64_t heads) {
    if (q_len <= 0 || kv_len <= 0 || heads <= 0) return 0;
    return align256(q_len * heads * 4) + align256(heads * pad64(q_len) * 2 * 4);
}

extern "C" GF_API int gf_flash_attn_bwd(const void* q, const void* k, const void* v, const void* o, const void* dout,
                                        const float* lse, void* workspace, void* dq, void* dk, void* dv, int64_t q_len,
                                        int64_t kv_len, int64_t heads, int64_t head_dim, int64_t q_stride, int64_t k_stride,
                                        int64_t v_stride, int64_t o_stride, int64_t do_stride, int64_t dq_stride,
                                        int64_t dk_stride, int64_t dv_stride, float scale, void* stream) {
    GF_CHECK_ARG(q && k && v && o && dout && lse && workspace && dq && ((dk && dv) || (!dk && !dv)),
                 "gf_flash_attn_bwd: null pointer (dk and dv may be NULL together: only dq is computed)");
    const bool want_dkv = dk != nullptr;
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
                     gf_aligned16(dq) && (!want_dkv || (gf_aligned16(dk) && gf_aligned16(dv))) && gf_aligned16(workspace),
                 "gf_flash_attn_bwd: 16-byte alignment required");
    GF_CHECK_ARG((q_len + 64) * q_stride * 2 < (1LL << 32) &&
                     (q_len + 64) * do_stride * 2 < (1LL << 32) && (kv_len + 64) * k_stride * 2 < (1LL << 32) && (kv_len + 64) * v_stride * 2 < (1LL << 32),
                 "gf_flash_attn_bwd: a sequence (len x stride) must stay below 4 GiB");
    static GfDeviceOnce once;
    hipError_t e = gf_once_per_device(once, [] {
        hipError_t r = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dq16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, DQ16_LDS);
        if (r != hipSuccess) return r;
        return hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dkv48_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, Kv48::LDS);
    });
    if (e != hipSuccess) {
        gf_set_error("gf_flash_attn_bwd: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
        return GF_ERR_LAUNCH;
    }
    Bwd16Args b;
    BwdArgs& a = b.b;
    a.q = (const u16*)q; a.k = (const u16*)k; a.v = (const u16*)v; a.o = (const u16*)o; a.dout = (const u16*)dout;
    a.lse = lse; a.delta = (float*)workspace;
    a.dq = (u16*)dq; a.dk = (u16*)dk; a.dv = (u16*)dv;
    a.q_len = (int)q_len; a.kv_len = (int)kv_len; a.heads = (int)heads;
    a.q_stride = q_stride; a.k_stride = k_stride; a.v_stride = v_stride; a.o_stride = o_stride; a.do_stride = do_stride;
    a.dq_stride = dq_stride; a.dk_stride = dk_stride; a.dv_stride = dv_stride;
    a.scale = scale; a.scale_log2e = scale * 1.4426950408889634f;
    hipStream_t s = (hipStream_t)stream;
    const unsigned nqb = (unsigned)((q_len + DQ_ROWS - 1) / DQ_ROWS);
    const int64_t q_pad = pad64(q_len);
    b.ngp = (int)(q_pad / GR);
    float* sd = (float*)((char*)workspace + align256(q_len * heads * 4));
    b.sd = sd;
    b.qs = nullptr;
    hipLaunchKernelGGL(attn_bwd_delta16_kernel, dim3((unsigned)((q_pad * heads + 255) / 256)), dim3(256), 0, s, b, sd);
    hipLaunchKernelGGL(attn_bwd_dq16_kernel, dim3(nqb * (unsigned)heads), dim3(DQ_THREADS), DQ16_LDS, s, b);
    if (!want_dkv) {
        GF_CHECK_LAUNCH("gf_flash_attn_bwd");
        return GF_OK;
    }
    const unsigned nkb48 = (unsigned)((kv_len + Kv48::ROWS - 1) / Kv48::ROWS);
    hipLaunchKernelGGL(attn_bwd_dkv48_kernel, dim3(nkb48 * (unsigned)heads), dim3(Kv48::THREADS), Kv48::LDS, s, b);
    GF_CHECK_LAUNCH("gf_flash_attn_bwd");
    return GF_OK;
}
