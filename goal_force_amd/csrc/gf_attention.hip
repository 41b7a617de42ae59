// gf_attention.hip — flash-attention forward, head_dim 128, non-causal, no mask, bf16 in/out.
// Replaces flash_attention() (reference diffsynth/models/wan_video_dit.py:28-61): self-attention over
// S = f*h*w video tokens (32760 at 832x480x81f) and cross-attention over 512 text tokens.
//
// CDNA4 design (gfx950), one workgroup = 8 waves = 256 query rows of one head, 32 rows per wave:
//   * swapped QK^T: S^T = K_tile · Q^T with v_mfma_f32_32x32x16_bf16, so the query is on the LANE and
//     the 64 keys of a KV tile sit in 2x16 accumulator registers of lanes l and l^32.  Row max / row
//     sum are in-register plus ONE cross-lane exchange; no LDS round trip for P.
//   * O^T = V^T · P^T: the S^T accumulator, converted pairwise to bf16, IS the B operand of the PV
//     MFMA (k order of the accumulator layout), and V^T fragments come from the row-major V tile by
//     ds_read_b64_tr_b16 (hardware transpose).  O^T keeps the query on the lane, so the online
//     softmax rescale is a per-lane scalar multiply.
//   * K and V tiles (64 keys x 128 d, 16 KiB each) are triple buffered in LDS in ONE image,
//     off(row,ch) = 256*row + 16*(ch ^ (((row&3)<<2) | ((row>>2)&3))), which is conflict-free for both
//     the ds_read_b128 row reads (K) and the transposed reads (V).  Global->register loads of tile
//     t+1 are issued before the MFMAs of tile t and written to LDS after them (latency hidden
//     under the matrix work); one barrier per tile.
//   * the two waves of a SIMD (w, w+4) run rotated by one phase: waves 4-7 defer each tile's PV product to the next
//     iteration, so one partner's MFMA phase runs beside the other's softmax (VALU/transcendental) phase instead
//     of both contending for the matrix pipe and then for the VALU in lockstep.
//   * blockIdx -> (head, q-block) is XCD-aware: the 32 CUs of an XCD work on the same head at the
//     same time so its K/V stream (16.8 MB at S=32760) is shared through that XCD's L2.
//   * softmax in fp32 with exp2 and the scale folded into one FMA; lazy rescale (skip the O
//     rescale while the running max grows by less than 2^THR_LOG2; exact in exact arithmetic).
#include "gf_common.h"

namespace {

constexpr int AT_THREADS = 512;
constexpr int QB = 256;          // query rows per workgroup
constexpr int KVB = 64;          // keys per tile
constexpr int HD = 128;          // head dim
constexpr int KV_TILE_BYTES = KVB * HD * 2;        // 16 KiB
constexpr int AT_STAGE_BYTES = 2 * KV_TILE_BYTES;  // K + V
constexpr int AT_LDS = 3 * AT_STAGE_BYTES;         // 96 KiB (three stages)

__device__ __forceinline__ int kv_off(int row, int ch) {
    return 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3)));
}

struct AttnArgs {
    const u16* q;
    const u16* k;
    const u16* v;
    u16* o;
    int q_len, kv_len, heads, n_qblocks;
    long q_stride, k_stride, v_stride, o_stride;
    float scale_log2e;  // softmax scale * log2(e)
};

__global__ __launch_bounds__(AT_THREADS, 2) void flash_attn_fwd_kernel(const AttnArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    GF_LDS char* lds = (GF_LDS char*)smem;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31;   // query column of this lane / row index inside a 32-row operand
    const int h = lane >> 5;   // lane half

    // ---- block -> (head, q-block) ---------------------------------------------------------------
    int head, qb;
    {
        const int pid = blockIdx.x;
        if ((p.heads & 7) == 0) {
            const int xcd = pid & 7, idx = pid >> 3;
            head = xcd + 8 * (idx / p.n_qblocks);
            qb = idx % p.n_qblocks;
        } else {
            head = pid / p.n_qblocks;
            qb = pid % p.n_qblocks;
        }
    }
    const int q0 = qb * QB + wave * 32;

    // ---- Q fragments (B operand of S^T = K Q^T): lane holds Q[q0+r][16*kd + 8h .. +8) ------------
    bf16x8 qf[8];
    {
        const int qr = min(q0 + r, p.q_len - 1);
        const u16* qp = p.q + (long)qr * p.q_stride + head * HD + 8 * h;
#pragma unroll
        for (int kd = 0; kd < 8; ++kd) qf[kd] = *reinterpret_cast<const bf16x8*>(qp + 16 * kd);
    }

    // ---- K/V staging: thread handles 16-byte chunk (row = tid>>4 [+32], ch = tid&15) -------------
    const int st_row = tid >> 4, st_ch = tid & 15;
    const u16* kbase = p.k + head * HD + st_ch * 8;
    const u16* vbase = p.v + head * HD + st_ch * 8;
    const int st_off0 = kv_off(st_row, st_ch);
    const int st_off1 = kv_off(st_row + 32, st_ch);
    u32x4 kreg0, kreg1, vreg0, vreg1;
    // 32-bit element offsets of this thread's two rows inside tile 0 (host guarantees kv_len*stride < 2^31); a full
    // tile t adds the wave-uniform t*64*stride, only the ragged last tile needs the per-lane clamp.
    const unsigned ko0 = (unsigned)st_row * (unsigned)p.k_stride, ko1 = (unsigned)(st_row + 32) * (unsigned)p.k_stride;
    const unsigned vo0 = (unsigned)st_row * (unsigned)p.v_stride, vo1 = (unsigned)(st_row + 32) * (unsigned)p.v_stride;
    const unsigned kstep = KVB * (unsigned)p.k_stride, vstep = KVB * (unsigned)p.v_stride;
    auto load_tile = [&](int t) {
        if ((t + 1) * KVB <= p.kv_len) {
            const unsigned tk = (unsigned)t * kstep, tv = (unsigned)t * vstep;   // scalar
            kreg0 = *reinterpret_cast<const u32x4*>(kbase + (ko0 + tk));
            kreg1 = *reinterpret_cast<const u32x4*>(kbase + (ko1 + tk));
            vreg0 = *reinterpret_cast<const u32x4*>(vbase + (vo0 + tv));
            vreg1 = *reinterpret_cast<const u32x4*>(vbase + (vo1 + tv));
        } else {
            const int k0 = t * KVB;
            const long r0 = min(k0 + st_row, p.kv_len - 1);
            const long r1 = min(k0 + st_row + 32, p.kv_len - 1);
            kreg0 = *reinterpret_cast<const u32x4*>(kbase + r0 * p.k_stride);
            kreg1 = *reinterpret_cast<const u32x4*>(kbase + r1 * p.k_stride);
            vreg0 = *reinterpret_cast<const u32x4*>(vbase + r0 * p.v_stride);
            vreg1 = *reinterpret_cast<const u32x4*>(vbase + r1 * p.v_stride);
        }
    };
    auto write_tile = [&](int buf) {
        GF_LDS char* kb = lds + buf * AT_STAGE_BYTES;
        GF_LDS char* vb = kb + KV_TILE_BYTES;
        *(GF_LDS u32x4*)(kb + st_off0) = kreg0;
        *(GF_LDS u32x4*)(kb + st_off1) = kreg1;
        *(GF_LDS u32x4*)(vb + st_off0) = vreg0;
        *(GF_LDS u32x4*)(vb + st_off1) = vreg1;
    };

    // ---- per-lane LDS read offsets ----------------------------------------------------------------
    // K row read for subtile kt, d-step kd: row = 32*kt + r, chunk = 2*kd + h
    //   off = 256*row + 16*((2kd+h) ^ sK), sK = ((r&3)<<2)|((r>>2)&3)   (32*kt does not change row&3, (row>>2)&3)
    const int sK = ((r & 3) << 2) | ((r >> 2) & 3);
    const int k_row_off = 256 * r;
    // V transposed read (ds_read_b64_tr_b16): 16-lane group g = lane>>4, i = lane&15, qd = i>>2, pp = i&3.
    //   rows key0 + qd with key0 = 32kt + 16s + 4h (+8 for the second half of the fragment),
    //   chunk = 4*dblk + 2*(g&1) + (pp>>1), byte +8*(pp&1)
    const int g1 = (lane >> 4) & 1;
    const int qd = (lane & 15) >> 2, pp = lane & 3;
    // row = 32kt + 16s + 8*half + 4h + qd : row&3 = qd, (row>>2)&3 = (2*half + h) & 3 -> depends on half
    const int v_row_base = 4 * h + qd;  // + 32kt + 16s + 8*half
    const int v_ch_lo = 2 * g1 + (pp >> 1);  // + 4*dblk
    const int v_byte = 8 * (pp & 1);
    const int sV0 = (qd << 2) | ((0 * 2 + h) & 3);  // half 0
    const int sV1 = (qd << 2) | ((1 * 2 + h) & 3);  // half 1

    f32x16 oacc[4];
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int e = 0; e < 16; ++e) oacc[d][e] = 0.f;
    float m_run = -1.0e30f;  // running max (raw score units)
    float l_run = 0.f;       // partial row sum of this lane's 32 keys per tile
    const float c = p.scale_log2e;
    const int nt = (p.kv_len + KVB - 1) / KVB;
    bf16x8 pf[2][2];  // P^T fragments of the tile whose PV product is pending

    // ---- S^T = K · Q^T for the tile in stage `kb`, online softmax, P^T fragments into pf -------------
    auto scores_softmax = [&](GF_LDS char* kb, int t) {
        f32x16 s0, s1;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            s0[e] = 0.f;
            s1[e] = 0.f;
        }
#pragma unroll
        for (int kd = 0; kd < 8; ++kd) {
            const int off = k_row_off + 16 * ((2 * kd + h) ^ sK);
            const bf16x8 k0f = *(GF_LDS bf16x8*)(kb + off);
            const bf16x8 k1f = *(GF_LDS bf16x8*)(kb + off + 32 * 256);
            s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k0f, qf[kd], s0, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k1f, qf[kd], s1, 0, 0, 0);
        }
        // s{kt}[e] = score(key = 64t + 32kt + (e&3) + 8*(e>>2) + 4h, query q0 + r)
        if (t == nt - 1 && (p.kv_len & (KVB - 1)) != 0) {  // mask the ragged tail (wave-uniform branch)
            const int kbase_i = t * KVB + 4 * h;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int key = kbase_i + (e & 3) + 8 * (e >> 2);
                if (key >= p.kv_len) s0[e] = -INFINITY;
                if (key + 32 >= p.kv_len) s1[e] = -INFINITY;
            }
        }
        float mx = s0[0];
#pragma unroll
        for (int e = 1; e < 16; ++e) mx = fmaxf(mx, s0[e]);
#pragma unroll
        for (int e = 0; e < 16; ++e) mx = fmaxf(mx, s1[e]);
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        // lazy rescale: keep the old max while no row of the wave grew by more than 2^6 (p <= 64: harmless for
        // bf16 P and fp32 sums; exact in exact arithmetic).  Everything still at the old scale (O, l) is rescaled
        // exactly once; P of this tile is exponentiated after the decision.
        if (!__all((mx - m_run) * c <= 6.0f)) {
            const float m_new = fmaxf(m_run, mx);
            const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);
            m_run = m_new;
            l_run *= alpha;
#pragma unroll
            for (int d = 0; d < 4; ++d)
#pragma unroll
                for (int e = 0; e < 16; ++e) oacc[d][e] *= alpha;
        }
        const float mc = m_run * c;
        float rs = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            s0[e] = __builtin_amdgcn_exp2f(__builtin_fmaf(s0[e], c, -mc));
            s1[e] = __builtin_amdgcn_exp2f(__builtin_fmaf(s1[e], c, -mc));
            rs += s0[e] + s1[e];
        }
        l_run += rs;
        // P^T fragments (B operand of O^T = V^T P^T): k-step s of subtile kt = registers 8s..8s+7
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                pf[0][s][e] = (__bf16)s0[8 * s + e];
                pf[1][s][e] = (__bf16)s1[8 * s + e];
            }
    };

    // ---- O^T += V^T · P^T for the V tile in stage `vb` and the pending pf ------------------------------
    auto pv = [&](GF_LDS char* vb) {
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int rowa = 32 * kt + 16 * s + v_row_base;  // half 0; half 1 = +8
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    const int ch = 4 * d + v_ch_lo;
                    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (GF_LDS s16x4*)(vb + 256 * rowa + 16 * (ch ^ sV0) + v_byte));
                    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (GF_LDS s16x4*)(vb + 256 * (rowa + 8) + 16 * (ch ^ sV1) + v_byte));
                    typedef __attribute__((ext_vector_type(8))) short s16x8;
                    const s16x8 vv = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                    oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vv), pf[kt][s],
                                                                      oacc[d], 0, 0, 0);
                }
            }
    };

    load_tile(0);
    write_tile(0);
    __syncthreads();

    // Three LDS stages; tile t lives in stage t % 3.  The two waves that share a SIMD (w and w+4) run the same
    // work rotated by one phase so that one's MFMA phase meets the other's softmax (VALU) phase:
    //   waves 0-3:  [QK^T(t), softmax(t)]  [PV(t)]                 | barrier
    //   waves 4-7:  [PV(t-1)]              [QK^T(t), softmax(t)]   | barrier      (+ PV(nt-1) after the loop)
    // Tile t+1 is written into stage (t+1)%3 at the end of iteration t; its previous tenant (tile t-2) was last read
    // by waves 4-7 in iteration t-1, which the barrier of t-1 fences.
    int cur = 0, prev = 2, nxt = 1;
    if (wave < 4) {
        for (int t = 0; t < nt; ++t) {
            GF_LDS char* kb = lds + cur * AT_STAGE_BYTES;
            if (t + 1 < nt) load_tile(t + 1);  // global -> registers, consumed after the MFMAs below
            scores_softmax(kb, t);
            pv(kb + KV_TILE_BYTES);
            if (t + 1 < nt) write_tile(nxt);
            __syncthreads();
            prev = cur;
            cur = nxt;
            nxt = (nxt == 2) ? 0 : nxt + 1;
        }
    } else {
        for (int t = 0; t < nt; ++t) {
            GF_LDS char* kb = lds + cur * AT_STAGE_BYTES;
            if (t + 1 < nt) load_tile(t + 1);
            if (t > 0) pv(lds + prev * AT_STAGE_BYTES + KV_TILE_BYTES);
            scores_softmax(kb, t);
            if (t + 1 < nt) write_tile(nxt);
            __syncthreads();
            prev = cur;
            cur = nxt;
            nxt = (nxt == 2) ? 0 : nxt + 1;
        }
        pv(lds + prev * AT_STAGE_BYTES + KV_TILE_BYTES);  // stage of tile nt-1: nobody writes after the loop
    }

    // ---- epilogue: O = O^T / l ----------------------------------------------------------------------
    const float l_tot = l_run + __shfl_xor(l_run, 32);
    const float inv = 1.0f / l_tot;
    const int qrow = q0 + r;
    if (qrow < p.q_len) {
        u16* op = p.o + (long)qrow * p.o_stride + head * HD + 4 * h;
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                // registers 4rg..4rg+3 = d index 32d + 8rg + 4h + {0..3}
                u32x2 pk;
                pk[0] = pack2bf(oacc[d][4 * rg + 0] * inv, oacc[d][4 * rg + 1] * inv);
                pk[1] = pack2bf(oacc[d][4 * rg + 2] * inv, oacc[d][4 * rg + 3] * inv);
                *reinterpret_cast<u32x2*>(op + 32 * d + 8 * rg) = pk;
            }
    }
}

}  // namespace

extern "C" GF_API int gf_flash_attn_fwd(const void* q, const void* k, const void* v, void* o, int64_t q_len, int64_t kv_len,
                                 int64_t heads, int64_t head_dim, int64_t q_stride, int64_t k_stride,
                                 int64_t v_stride, int64_t o_stride, float scale, void* stream) {
    GF_CHECK_ARG(q && k && v && o, "gf_flash_attn_fwd: null pointer");
    if (head_dim != HD) {
        gf_set_error("gf_flash_attn_fwd: head_dim=%ld unsupported (kernel is built for 128)", (long)head_dim);
        return GF_ERR_UNSUPPORTED;
    }
    GF_CHECK_ARG(q_len >= 0 && kv_len > 0 && heads > 0, "gf_flash_attn_fwd: bad lengths q=%ld kv=%ld heads=%ld",
                 (long)q_len, (long)kv_len, (long)heads);
    GF_CHECK_ARG(q_stride % 8 == 0 && k_stride % 8 == 0 && v_stride % 8 == 0 && o_stride % 4 == 0 &&
                     q_stride >= heads * HD && k_stride >= heads * HD && v_stride >= heads * HD &&
                     o_stride >= heads * HD,
                 "gf_flash_attn_fwd: strides must cover heads*128 and be multiples of 8");
    GF_CHECK_ARG(gf_aligned16(q) && gf_aligned16(k) && gf_aligned16(v) && gf_aligned16(o),
                 "gf_flash_attn_fwd: 16-byte alignment required");
    GF_CHECK_ARG(q_len < (1 << 30) && kv_len < (1 << 30), "gf_flash_attn_fwd: sequence too long");
    GF_CHECK_ARG((kv_len + 64) * k_stride < (1LL << 31) && (kv_len + 64) * v_stride < (1LL << 31),
                 "gf_flash_attn_fwd: kv_len*stride must stay below 2^31 elements");
    if (q_len == 0) return GF_OK;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(flash_attn_fwd_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, AT_LDS);
        if (e != hipSuccess) {
            gf_set_error("gf_flash_attn_fwd: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
            return GF_ERR_LAUNCH;
        }
        attr_set = true;
    }
    AttnArgs a;
    a.q = (const u16*)q;
    a.k = (const u16*)k;
    a.v = (const u16*)v;
    a.o = (u16*)o;
    a.q_len = (int)q_len;
    a.kv_len = (int)kv_len;
    a.heads = (int)heads;
    a.n_qblocks = (int)((q_len + QB - 1) / QB);
    a.q_stride = q_stride;
    a.k_stride = k_stride;
    a.v_stride = v_stride;
    a.o_stride = o_stride;
    a.scale_log2e = scale * 1.4426950408889634f;
    hipLaunchKernelGGL(flash_attn_fwd_kernel, dim3((unsigned)(a.n_qblocks * a.heads)), dim3(AT_THREADS), AT_LDS,
                       (hipStream_t)stream, a);
    GF_CHECK_LAUNCH("gf_flash_attn_fwd");
    return GF_OK;
}
