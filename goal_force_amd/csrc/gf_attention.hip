// gf_attention.hip — flash-attention forward, head_dim 128, non-causal, no mask, bf16 in/out.
// Replaces flash_attention() (reference diffsynth/models/wan_video_dit.py:28-61): self-attention over
// S = f*h*w video tokens (32760 at 832x480x81f) and cross-attention over 512 text tokens.
//
// Two kernels ship, both with one workgroup = 8 waves = 256 query rows of one head, 32 rows per wave, two waves per SIMD:
//   * kernel 2 (`flash_attn_fwd_kernel2`, v_mfma_f32_32x32x16_bf16): key sequences below 2048 — the cross-attention over the
//     text tokens, with the optional multiplicity of the last key (gf_flash_attn_fwd_lastmult) — and the plain-V entry points;
//   * kernel 3 (`flash_attn_fwd_kernel3`, v_mfma_f32_16x16x32_bf16): the self-attention of the DiT / ControlNet blocks, V^T
//     written by the V projection (gf_linear_vt32) or by gf_transpose_v32.
// Common to both:
//   * swapped QK^T: S^T = K_tile · Q^T, so a query stays on ONE lane; row maxima and sums are in-register, no LDS round trip for P;
//   * O^T = V^T · P^T: the S^T accumulator, converted pairwise to bf16, IS the B operand of the PV MFMA, and O^T keeps the query
//     on the lane, so the online-softmax rescale is a per-lane scalar multiply;
//   * K / V tiles (64 keys) are staged by LDS-DMA (`buffer_load ... lds`, tile offset as an SGPR); one barrier per tile;
//   * a tile-level software pipeline inside ONE instruction stream per wave (PV of tile p-1 and QK^T of tile p+1 beside the
//     softmax of tile p), pinned with sched_group_barriers;
//   * blockIdx -> (head, q-block) is XCD-aware: the 32 CUs of an XCD work on the same head at the same time so its K/V stream
//     (16.8 MB at S = 32760) is shared through that XCD's L2;
//   * softmax in fp32 in the exp2 domain; lazy rescale (O is only rescaled when the running maximum has moved far enough).
// The phase-serial first kernel and the measured-and-dropped variants of kernel 3 (ORMAX watch, head-major map, what-if and clock
// builds) are NOT in this file: tools/patches/attention_experiments.patch re-creates them on top of it (tools/build_variants.sh).
#include "gf_common.h"
#include <cstdlib>
#include <type_traits>

namespace {

constexpr int QB = 256;          // query rows per workgroup
constexpr int KVB = 64;          // keys per tile
constexpr int HD = 128;          // head dim
constexpr int KV_TILE_BYTES = KVB * HD * 2;        // 16 KiB

// kernel 3's schedule parameters (A/B history: EXPERIMENTS.md appendix B)
constexpr int K3_RING = 2;       // operand fragments in flight (registers are the scarce resource at two waves per SIMD)
constexpr int K3_DMA0 = 10;      // slot of the phase's first staging piece ...
constexpr int K3_DMAS = 8;       // ... and the slot distance between pieces (4 per wave and phase)

struct AttnArgs {
    const u16* q;
    const u16* k;
    const u16* v;
    u16* o;
    int q_len, kv_len, heads, n_qblocks;
    long q_stride, k_stride, v_stride, o_stride;
    float scale_log2e;  // softmax scale * log2(e)
    unsigned long long* dbg;   // unused (kept so that diagnostic builds from tools/patches need no ABI change inside the library)
    float* lse;         // optional [q_len, heads] fp32: log2-domain log-sum-exp of the scaled scores (training backward)
    const u16* vt;      // kernel 2, VT form: V^T [heads][128][kv_pad] from gf_transpose_v (keys permuted inside 16-groups)
    long kv_pad;
    float last_key_bias;   // kernel 2 (gf_flash_attn_fwd_lastmult): added to the RAW scores of key kv_len - 1 = log2(multiplicity) / (scale log2 e):
                           // that key then counts `multiplicity` times in the softmax — a run of identical trailing keys folded into one
};

// ================================================================================================================
// Kernel 2: 8 waves x 32 query rows, two waves per SIMD; every wave runs a tile-level software pipeline inside ONE
// instruction stream.  Phase p of a wave:
//     matrix stream :  O^T += V(p-1)^T P(p-1)^T   (16 MFMAs)   then   S(p+1)^T = K(p+1) Q^T   (16 MFMAs)
//     VALU stream   :  softmax of S(p)  ->  P(p)  (row max, 32 exp2, row sum, bf16 pack)
// tools/issue_probe.py (profiles/r01/issue_probe.txt) measured what bounds this loop on gfx950: a wave issues serially
// (MFMA 4, VALU ~5, v_exp_f32 ~12.5, ds_read ~6.7 cycles each) against 32 cycles per 32x32x16 MFMA, so a wave that runs
// its matrix phase and its softmax phase one after the other leaves the matrix pipe idle half the time and two such
// waves only help when they happen to be out of phase.  Here each 64-cycle slot (two waves share the pipe) carries
// one MFMA, its operand reads for three slots ahead and ONE score's softmax work, pinned in that order: both waves
// of a SIMD present the same steady mix of matrix / VALU / LDS work all the time.
//   * S is double buffered in registers (tile parity), P needs one buffer: PV(p-1) reads fragment (kt, s) in slots
//     4(2kt+s)..+3, the softmax of tile p rewrites it from slot 5 + 8(2kt+s) on.
//   * lazy rescale is deferred by one phase: a new running max found in phase p scales l at once and P(p) is taken
//     against it, but O (still receiving PV(p-1), which is in the OLD scale) is multiplied at the top of phase p+1.
//   * K and V tiles are staged separately by LDS-DMA, two buffers each (64 KiB): K(p+2) and V(p) are issued at the top
//     of phase p and waited for at its closing barrier (K(j) is consumed in phase j-1, V(j) in phase j+1).
constexpr int AT2_THREADS = 512;
constexpr int AT2_V_BASE = 2 * KV_TILE_BYTES;
constexpr int AT2_LDS = 4 * KV_TILE_BYTES;

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}
__device__ __forceinline__ void mfma32(f32x16& acc, const bf16x8& a, const bf16x8& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
}

// VT = true: V arrives pre-transposed ([head][d][key], gf_transpose_v) so that a PV MFMA's A fragment (32 d-rows x 16
// keys) is ONE ds_read_b128 instead of two ds_read_b64_tr_b16: one LDS instruction less per PV slot of an issue-bound loop
// (+2.5 % in a what-if build); inside every group of 16 keys the copy stores keys 0-3, 8-11, 4-7, 12-15, the order in which a
// lane half holds its eight scores of the 16-key step.
template <bool VT>
__global__ __launch_bounds__(AT2_THREADS, 2) void flash_attn_fwd_kernel2(const AttnArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    GF_LDS char* lds = (GF_LDS char*)smem;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
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

    bf16x8 qf[8];
    {
        const int qr = min(q0 + r, p.q_len - 1);
        const u16* qp = p.q + (long)qr * p.q_stride + head * HD + 8 * h;
#pragma unroll
        for (int kd = 0; kd < 8; ++kd) qf[kd] = *reinterpret_cast<const bf16x8*>(qp + 16 * kd);
    }

    // ---- DMA staging: wave w fills row-groups j = 2w, 2w+1 (4 rows x 256 B each) of a K or V tile
    const int dma_r = lane >> 4;
    unsigned dma_off[2][2];
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
        const int j = 2 * wave + jj;
        const int row = 4 * j + dma_r;
        const int lch = (lane & 15) ^ ((dma_r << 2) | (j & 3));
        dma_off[jj][0] = (unsigned)row * (unsigned)p.k_stride + head * HD + lch * 8;
        dma_off[jj][1] = (unsigned)row * (unsigned)p.v_stride + head * HD + lch * 8;
    }
    const unsigned kstep = KVB * (unsigned)p.k_stride, vstep = KVB * (unsigned)p.v_stride;
    auto dma16 = [&](const u16* g, GF_LDS char* l) {
        unsigned keep;
        const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)l);
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(g), "s"(dst)
                     : "memory");
    };
    // The steady-state pieces go through the BUFFER form of the same LDS-DMA: descriptor (wave-uniform base of K, V or V^T) +
    // this lane's constant 32-bit byte offset + the tile's byte offset as an SGPR soffset.  No vector address arithmetic per
    // piece (the global form adds a 64-bit per-lane pointer: two VALU instructions in an issue-bound loop) and one SALU
    // instruction less.  kv_len * stride * 2 < 2^32 is checked on the host, so voffset + soffset never wraps.
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4s;
    auto make_srd = [](const void* base) {
        const unsigned long b = (unsigned long)base;
        u32x4s r;
        r[0] = (unsigned)b;
        r[1] = (unsigned)(b >> 32) & 0xffffu;
        r[2] = 0xffffffffu;     // num_records: the host checks the extents; ragged tiles take the clamped global form
        r[3] = 0x00020000u;
        return r;
    };
    const u32x4s srd_k = make_srd(p.k), srd_v = make_srd(VT ? (const void*)p.vt : (const void*)p.v);
    auto dma16b = [&](const u32x4s& srd, unsigned voff_bytes, unsigned soff_bytes, GF_LDS char* l) {
        unsigned keep;
        const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)l);
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(voff_bytes), "s"(srd), "s"(dst), "s"(soff_bytes)
                     : "memory");
    };
    // which = 0: K tile t -> K buffer buf;  which = 1: V tile t -> V buffer buf
    // VT: the V tile image is 128 d-rows x 128 B (64 keys), chunk ^ ((row >> 1) & 7); wave w stages rows 16w..16w+15 as two
    // pieces of 8 rows (lane L: row L>>3, physical chunk L&7)
    unsigned vt_off[2];
    if constexpr (VT) {
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int row = 16 * wave + 8 * jj + (lane >> 3);
            const int lch = (lane & 7) ^ ((row >> 1) & 7);
            vt_off[jj] = (unsigned)(((long)head * HD + row) * p.kv_pad + lch * 8);
        }
    }
    auto stage = [&](int which, int t, int buf) {
        GF_LDS char* base = lds + which * AT2_V_BASE + buf * KV_TILE_BYTES + wave * 2048;
        if constexpr (VT) {
            if (which) {
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    dma16b(srd_v, vt_off[jj] * 2u, (unsigned)t * (KVB * 2u), base + jj * 1024);
                }
                return;
            }
        }
        const u16* g = which ? p.v : p.k;
        if ((t + 1) * KVB <= p.kv_len) {
            const unsigned tt = (unsigned)t * (which ? vstep : kstep);
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                dma16b(which ? srd_v : srd_k, dma_off[jj][which] * 2u, tt * 2u, base + jj * 1024);
            }
        } else {   // ragged last tile: clamp the row (masked later), 64-bit addressing
            const long stride = which ? p.v_stride : p.k_stride;
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const int j = 2 * wave + jj;
                const long rr = min(t * KVB + 4 * j + dma_r, p.kv_len - 1);
                const int lch = (lane & 15) ^ ((dma_r << 2) | (j & 3));
                dma16(g + rr * stride + head * HD + lch * 8, base + jj * 1024);
            }
        }
    };

    // one of the two DMA pieces of stage(which, t, buf)
    auto stage_piece = [&](int which, int t, int buf, int jj) {
        GF_LDS char* base = lds + which * AT2_V_BASE + buf * KV_TILE_BYTES + wave * 2048;
        if constexpr (VT) {
            if (which) {
                dma16b(srd_v, vt_off[jj] * 2u, (unsigned)t * (KVB * 2u), base + jj * 1024);
                return;
            }
        }
        const u16* g = which ? p.v : p.k;
        if ((t + 1) * KVB <= p.kv_len) {
            const unsigned tt = (unsigned)t * (which ? vstep : kstep);
            dma16b(which ? srd_v : srd_k, dma_off[jj][which] * 2u, tt * 2u, base + jj * 1024);
        } else {
            const long stride = which ? p.v_stride : p.k_stride;
            const int j = 2 * wave + jj;
            const long rr = min(t * KVB + 4 * j + dma_r, p.kv_len - 1);
            const int lch = (lane & 15) ^ ((dma_r << 2) | (j & 3));
            dma16(g + rr * stride + head * HD + lch * 8, base + jj * 1024);
        }
    };

    int vt_rd[4];   // VT: byte offset of this lane's 16-byte chunk of step (kt, s) inside its d-row: row r, chunk (4kt+2s+h)^key
    if constexpr (VT) {
#pragma unroll
        for (int c4 = 0; c4 < 4; ++c4) vt_rd[c4] = AT2_V_BASE + r * 128 + (((2 * c4 + h) ^ ((r >> 1) & 7)) << 4);
    }
    int koff[8], voff[2][4];
    {
        const int sK = ((r & 3) << 2) | ((r >> 2) & 3);
#pragma unroll
        for (int kd = 0; kd < 8; ++kd) koff[kd] = 256 * r + 16 * ((2 * kd + h) ^ sK);
        const int qd = (lane & 15) >> 2, pp = lane & 3;
        const int vcl = 2 * ((lane >> 4) & 1) + (pp >> 1);
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            const int sV = (qd << 2) | ((2 * hf + h) & 3);
#pragma unroll
            for (int d = 0; d < 4; ++d)
                voff[hf][d] = AT2_V_BASE + 256 * (8 * hf + 4 * h + qd) + 16 * ((4 * d + vcl) ^ sV) + 8 * (pp & 1);
        }
    }

    f32x16 oacc[4], sc[2][2];   // sc[tile parity][key half]
    bf16x8 pf[2][2];            // P fragments [kt][s]
    float m_run = -1.0e30f, l_run = 0.f, alpha_pend = 1.f;
    bool pend = false;          // wave-uniform: O still has to be multiplied by alpha_pend
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int e = 0; e < 16; ++e) oacc[d][e] = 0.f;
    const float c = p.scale_log2e;
    const int nt = (p.kv_len + KVB - 1) / KVB;
    const bool ragged = (p.kv_len & (KVB - 1)) != 0;

    typedef std::integral_constant<int, 0> C0;
    typedef std::integral_constant<int, 1> C1;

    // ---- unpipelined building blocks (prologue, first and last phase)
    auto qk_plain = [&](auto par_c) {   // S(par) = K(buffer par) Q^T
        constexpr int PAR = decltype(par_c)::value;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            sc[PAR][0][e] = 0.f;
            sc[PAR][1][e] = 0.f;
        }
#pragma unroll
        for (int kd = 0; kd < 8; ++kd) {
            const bf16x8 k0f = *(GF_LDS bf16x8*)(lds + koff[kd] + PAR * KV_TILE_BYTES);
            const bf16x8 k1f = *(GF_LDS bf16x8*)(lds + koff[kd] + PAR * KV_TILE_BYTES + 32 * 256);
            mfma32(sc[PAR][0], k0f, qf[kd]);
            mfma32(sc[PAR][1], k1f, qf[kd]);
        }
    };
    auto pv_plain = [&](auto buf_c) {   // O^T += V(buffer)^T P^T
        constexpr int BUF = decltype(buf_c)::value;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int imm = BUF * KV_TILE_BYTES + 256 * (32 * kt + 16 * s);
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    if constexpr (VT) {
                        const bf16x8 vv = *(GF_LDS bf16x8*)(lds + vt_rd[2 * kt + s] + BUF * KV_TILE_BYTES + d * 4096);
                        mfma32(oacc[d], vv, pf[kt][s]);
                    } else {
                        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((GF_LDS s16x4*)(lds + voff[0][d] + imm));
                        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((GF_LDS s16x4*)(lds + voff[1][d] + imm));
                        typedef __attribute__((ext_vector_type(8))) short s16x8;
                        const s16x8 vv = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                        mfma32(oacc[d], __builtin_bit_cast(bf16x8, vv), pf[kt][s]);
                    }
                }
            }
    };
    auto apply_pending = [&]() {
        if (pend) {
#pragma unroll
            for (int d = 0; d < 4; ++d)
#pragma unroll
                for (int e = 0; e < 16; ++e) oacc[d][e] *= alpha_pend;
            pend = false;
        }
    };
    // new running max of tile scores `mx` (already exchanged between the lane halves): rare
    auto new_max = [&](float mx) {
        if (!__all((mx - m_run) * c <= 6.0f)) {
            const float m_new = fmaxf(m_run, mx);
            const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);
            m_run = m_new;
            l_run *= alpha;
            alpha_pend = alpha;   // at most one outstanding: applied at the top of the next phase
            pend = true;
        }
    };
    auto softmax_plain = [&](auto par_c, int t) {
        constexpr int PAR = decltype(par_c)::value;
        if ((ragged || p.last_key_bias != 0.f) && t == nt - 1) {
            const int kbase_i = t * KVB + 4 * h;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int key = kbase_i + (e & 3) + 8 * (e >> 2);
                if (key >= p.kv_len) sc[PAR][0][e] = -INFINITY;
                if (key + 32 >= p.kv_len) sc[PAR][1][e] = -INFINITY;
                // the last key stands for `multiplicity` identical keys: exp2(c s + log2 m) = m exp2(c s)
                if (key == p.kv_len - 1) sc[PAR][0][e] += p.last_key_bias;
                if (key + 32 == p.kv_len - 1) sc[PAR][1][e] += p.last_key_bias;
            }
        }
        float mx = sc[PAR][0][0];
#pragma unroll
        for (int e = 1; e < 16; ++e) mx = fmaxf(mx, sc[PAR][0][e]);
#pragma unroll
        for (int e = 0; e < 16; ++e) mx = fmaxf(mx, sc[PAR][1][e]);
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        new_max(mx);
        const float mc = m_run * c;
        float rs = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            sc[PAR][0][e] = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[PAR][0][e], c, -mc));
            sc[PAR][1][e] = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[PAR][1][e], c, -mc));
            rs += sc[PAR][0][e] + sc[PAR][1][e];
        }
        l_run += rs;
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                pf[0][s][e] = (__bf16)sc[PAR][0][8 * s + e];
                pf[1][s][e] = (__bf16)sc[PAR][1][8 * s + e];
            }
    };

    // ---- the steady phase p (1 <= p <= nt-2), PAR = p & 1:  matrix ops g = 0..31
    //   g <  16: PV(p-1)   kt = g>>3, s = (g>>2)&1, d = g&3;  V^T fragment from V buffer 1-PAR (two transposed reads)
    //   g >= 16: QK(p+1)   kd = (g-16)>>1, half = (g-16)&1;   K fragment from K buffer 1-PAR (one row read) -> sc[1-PAR]
    // The A operand of op g is read in slot g-3 into ring entry g&3 (ops 0..2 at the top of the phase: their tiles
    // only land at the closing barrier of the phase before); hipcc counts lgkmcnt itself.  sched_barrier(0) after every
    // slot pins the written order; the empty volatile asms pin each score's VALU work INTO its slot (instruction
    // selection would otherwise hoist all 32 exps to the top of the phase).
    bf16x8 fr[4];
    auto frag_load = [&](auto g_c, auto par_c) {
        constexpr int G = decltype(g_c)::value, PAR = decltype(par_c)::value;
        if constexpr (G < 16) {
            constexpr int kt = G >> 3, sq = (G >> 2) & 1, d = G & 3;
            if constexpr (VT) {
                fr[G & 3] = *(GF_LDS bf16x8*)(lds + vt_rd[2 * kt + sq] + (1 - PAR) * KV_TILE_BYTES + d * 4096);
            } else {
                constexpr int imm = (1 - PAR) * KV_TILE_BYTES + 256 * (32 * kt + 16 * sq);
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((GF_LDS s16x4*)(lds + voff[0][d] + imm));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((GF_LDS s16x4*)(lds + voff[1][d] + imm));
                typedef __attribute__((ext_vector_type(8))) short s16x8;
                const s16x8 vv = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                fr[G & 3] = __builtin_bit_cast(bf16x8, vv);
            }
        } else if constexpr (G < 32) {
            constexpr int kd = (G - 16) >> 1, half = (G - 16) & 1;
            fr[G & 3] = *(GF_LDS bf16x8*)(lds + koff[kd] + (1 - PAR) * KV_TILE_BYTES + half * 32 * 256);
        }
    };
    auto mfma_op = [&](auto g_c, auto par_c) {
        constexpr int G = decltype(g_c)::value, PAR = decltype(par_c)::value;
        if constexpr (G < 16) {
            constexpr int kt = G >> 3, sq = (G >> 2) & 1, d = G & 3;
            mfma32(oacc[d], fr[G & 3], pf[kt][sq]);
        } else {
            constexpr int kd = (G - 16) >> 1, half = (G - 16) & 1;
            if constexpr (kd == 0) {
                const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                sc[1 - PAR][half] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[G & 3], qf[kd], zero, 0, 0, 0);
            } else {
                mfma32(sc[1 - PAR][half], fr[G & 3], qf[kd]);
            }
        }
    };
    // The four DMA pieces of a phase are issued inside its slots (V(p) in slots 5 and 9, K(p+2) in 13 and 17) instead of back to
    // back at its top: +1 % (18 of the phase's 32 slots still lie between the last piece and the closing wait).  Staggering them
    // over the waves as well (two waves per slot) measured 1.7 % SLOWER.
    auto dma_slot = [&](auto g_c, auto par_c, int pidx) {
        constexpr int G = decltype(g_c)::value, PAR = decltype(par_c)::value;
        if constexpr (G == 5) stage_piece(1, pidx, PAR, 0);
        if constexpr (G == 9) stage_piece(1, pidx, PAR, 1);
        if constexpr (G == 13) { if (pidx + 2 < nt) stage_piece(0, pidx + 2, PAR, 0); }
        if constexpr (G == 17) { if (pidx + 2 < nt) stage_piece(0, pidx + 2, PAR, 1); }
    };
    auto phase = [&](auto par_c, int pidx) {
        constexpr int PAR = decltype(par_c)::value;
        apply_pending();
        frag_load(std::integral_constant<int, 0>{}, par_c);
        frag_load(std::integral_constant<int, 1>{}, par_c);
        frag_load(std::integral_constant<int, 2>{}, par_c);
        float mx = -INFINITY, mo = 0.f;
        static_for<0, 4>([&](auto i_c) {
            constexpr int G = decltype(i_c)::value;
            mfma_op(i_c, par_c);
            frag_load(std::integral_constant<int, G + 3>{}, par_c);
            dma_slot(i_c, par_c, pidx);
            if constexpr (G == 1 || G == 2) {   // two independent v_max3 chains per slot (a wave issues in order)
                float m0 = mx, m1 = -INFINITY;
#pragma unroll
                for (int e = 0; e < 16; e += 4) {
                    m0 = fmaxf(fmaxf(m0, sc[PAR][G - 1][e]), sc[PAR][G - 1][e + 1]);
                    m1 = fmaxf(fmaxf(m1, sc[PAR][G - 1][e + 2]), sc[PAR][G - 1][e + 3]);
                }
                mx = fmaxf(m0, m1);
                if constexpr (G == 2) mo = __shfl_xor(mx, 32);
            } else if constexpr (G == 3) {
                mx = fmaxf(mx, mo);
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        new_max(mx);
        const float mc = m_run * c;
        float rs = 0.f, pe[2];
        static_for<4, 32>([&](auto i_c) {
            constexpr int G = decltype(i_c)::value;
            mfma_op(i_c, par_c);
            frag_load(std::integral_constant<int, G + 3>{}, par_c);
            dma_slot(i_c, par_c, pidx);
            // scores 0..7 two per slot in slots 4-7, scores 8..31 one per slot in slots 8-31
            constexpr int n_el = (G < 8) ? 2 : 1;
            constexpr int el0 = (G < 8) ? 2 * (G - 4) : G;
            static_for<0, n_el>([&](auto k_c) {
                constexpr int el = el0 + decltype(k_c)::value, half = el >> 4, e = el & 15;
                float sv = sc[PAR][half][e];
                asm volatile("" : "+v"(sv));
                pe[el & 1] = __builtin_amdgcn_exp2f(__builtin_fmaf(sv, c, -mc));
                rs += pe[el & 1];
                if constexpr (el & 1) {
                    asm volatile("" : "+v"(pe[0]), "+v"(pe[1]), "+v"(rs));
                    pf[half][e >> 3][(e & 7) - 1] = (__bf16)pe[0];
                    pf[half][e >> 3][e & 7] = (__bf16)pe[1];
                }
            });
            __builtin_amdgcn_sched_barrier(0);
        });
        l_run += rs;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    };

    // ---- prologue: K(0), K(1) staged; S(0)
    stage(0, 0, 0);
    if (nt > 1) stage(0, 1, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    qk_plain(C0{});
    __syncthreads();   // every wave has read K(0) before K(2) overwrites its buffer
    // ---- phase 0 (no PV yet): K(2), V(0) in flight; S(1); softmax(0)
    if (nt > 2) stage(0, 2, 0);
    stage(1, 0, 0);
    if (nt > 1) qk_plain(C1{});
    softmax_plain(C0{}, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // ---- steady phases 1 .. nt-2
    int pi = 1;
    for (; pi + 1 <= nt - 2; pi += 2) {
        phase(C1{}, pi);
        phase(C0{}, pi + 1);
    }
    if (pi <= nt - 2) {
        phase(C1{}, pi);
        ++pi;
    }
    // ---- last phase p = nt-1 (nt >= 2): V(nt-1) in flight; PV(nt-2); softmax(nt-1)
    if (nt >= 2) {
        const int par = (nt - 1) & 1;
        stage(1, nt - 1, par);
        apply_pending();
        if (par) {
            pv_plain(C0{});
            softmax_plain(C1{}, nt - 1);
        } else {
            pv_plain(C1{});
            softmax_plain(C0{}, nt - 1);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    // ---- PV(nt-1)
    apply_pending();
    if ((nt - 1) & 1)
        pv_plain(C1{});
    else
        pv_plain(C0{});

    // ---- epilogue
    {
        const float l_tot = l_run + __shfl_xor(l_run, 32);
        const float inv = 1.0f / l_tot;
        const int qrow = q0 + r;
        // P = exp2(c S - lse) is the softmax row: what the backward kernels rebuild the probabilities from
        if (p.lse && h == 0 && qrow < p.q_len) p.lse[(long)qrow * p.heads + head] = m_run * c + __builtin_amdgcn_logf(l_tot);
        // O leaves through LDS: a lane holds 8-byte pieces of ITS row (16 stores at a 10 KB row pitch touch 32 lines each and are
        // issue-bound); the wave's 32 x 256 B image, 16-byte chunk c of row r at chunk c ^ (r & 15), goes out as 8 stores of
        // four whole 256-byte rows.  The K/V tiles are dead: every wave is past its last V read at this barrier.
        __syncthreads();
        GF_LDS char* ob = lds + wave * 8192;
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                u32x2 pk;
                pk[0] = pack2bf(oacc[d][4 * rg + 0] * inv, oacc[d][4 * rg + 1] * inv);
                pk[1] = pack2bf(oacc[d][4 * rg + 2] * inv, oacc[d][4 * rg + 3] * inv);
                *(GF_LDS u32x2*)(ob + r * 256 + (((4 * d + rg) ^ (r & 15)) << 4) + 8 * h) = pk;
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // wave-local image: LDS is in order, no barrier needed
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int row = 4 * it + (lane >> 4), ch = lane & 15;
            const u16x8 v8 = *(GF_LDS u16x8*)(ob + row * 256 + ((ch ^ (row & 15)) << 4));
            if (q0 + row < p.q_len)
                *reinterpret_cast<u16x8*>(p.o + (long)(q0 + row) * p.o_stride + head * HD + 8 * ch) = v8;
        }
    }
}

// ================================================================================================================
// Kernel 3 (round 2): the same tile pipeline on v_mfma_f32_16x16x32_bf16.
// Why: on RANDOM operands the chip lowers its clock under an MFMA-dense load, and the 32x32x16 form draws more per FLOP:
// a bare MFMA loop with every CU busy delivers 1.77 PFLOP/s with 32x32x16 and 2.00 with 16x16x32 (zeros: 2.45 both;
// tools/probes/mfma_shape_power.hip).  Kernel 2 runs at 1.62 PFLOP/s on all-zero q/k/v and 1.17-1.25 on random data with the
// SAME instruction stream (tools/microbench.py attn --zeros): it is bound by that clock, not by its issue slots.
//
// Layouts (r = lane & 15, g = lane >> 4; a wave still owns 32 query rows = two 16-query blocks qb):
//   S^T = K Q^T : A = K[16 keys x 32 d] (one ds_read_b128: row 16 kb + r, 16-byte chunk 4 ks + g), B = Q^T (registers),
//                 D = sc[kb][qb] (f32x4): query 16 qb + r, keys 16 kb + 4 g + j.
//   The lane's 32 scores of a tile belong to ONE query per qb, spread over the four g-lanes of that query: the running max is
//   kept identical in those four lanes, the lazy-rescale test `all scores <= max + 6` needs no cross-lane step (it is the same
//   test on the partial maxima), row sums are taken on the matrix pipe (a ninth "d block" against a ones fragment);
//   only the rare rescale reduces over g.
//   O^T += V^T P^T : B = P^T[32 keys x 16 queries] = {sc[2kk][qb][0..3], sc[2kk+1][qb][0..3]} converted in place (lane-local),
//                 A = V^T[16 d x 32 keys] (one ds_read_b128: row 16 db + r, chunk 4 kk + g of the pre-transposed copy, whose
//                 keys are stored inside every group of 32 in the B operand's order: position 8 g + i <-> key 4 g + i (i < 4),
//                 16 + 4 g + i - 4 (i >= 4): gf_transpose_v32, or gf_linear_vt32 straight from the V projection), D = oacc[db][qb]
//                 (f32x4): query 16 qb + r, d = 16 db + 4 g + j.
//   Per 64-key tile and wave: 32 + 32 MFMAs of 16 cycles (kernel 2: 16 + 16 of 32), the same 32 fragment reads (each feeds the
//   two query blocks), the same 32 scores per lane.  K image: 256-byte rows, chunk ^ (row & 15) (conflict-free for the
//   16-row x 4-chunk fragment read; the 32x32x16 image is 2-way here); V^T image as in kernel 2.
// NQ = 16-query blocks per wave: 2 -> 8 waves x 32 rows, two waves per SIMD (256 registers each).
constexpr int K3_NQ = 2;
template <int NQ> struct At3 {
    static constexpr int WAVES = 16 / NQ, THREADS = 64 * WAVES, ROWS = 16 * NQ, PIECES = NQ;   // PIECES: 1-KiB DMA pieces per wave and tile
    static constexpr int SLOTS = 32 * NQ;
};
constexpr int AT3_V_BASE = 2 * KV_TILE_BYTES;
constexpr int AT3_LDS = 4 * KV_TILE_BYTES;

__device__ __forceinline__ void mfma16(f32x4& acc, const bf16x8& a, const bf16x8& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
}

template <int NQ>
__global__ __launch_bounds__(At3<NQ>::THREADS, (NQ == 2 ? 2 : 1)) void flash_attn_fwd_kernel3(const AttnArgs p) {
    constexpr int NP = At3<NQ>::PIECES, NW = At3<NQ>::WAVES, RW = At3<NQ>::ROWS, NS = At3<NQ>::SLOTS;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    GF_LDS char* lds = (GF_LDS char*)smem;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, g = lane >> 4;
    int head, qb0;
    {
        const int pid = blockIdx.x;
        // XCD-aware: the 32 CUs of an XCD walk the query blocks of ONE head together, K / V of that head come from the XCD's L2
        if ((p.heads & 7) == 0) {
            const int xcd = pid & 7, idx = pid >> 3;
            head = xcd + 8 * (idx / p.n_qblocks);
            qb0 = idx % p.n_qblocks;
        } else {
            head = pid / p.n_qblocks;
            qb0 = pid % p.n_qblocks;
        }
    }
    const int q0 = qb0 * QB + wave * RW;

    bf16x8 qf[NQ][4];   // [qb][ks]: Q[q0 + 16 qb + r][32 ks + 8 g .. +8)
#pragma unroll
    for (int qb = 0; qb < NQ; ++qb) {
        const int qr = min(q0 + 16 * qb + r, p.q_len - 1);
        const u16* qp = p.q + (long)qr * p.q_stride + head * HD + 8 * g;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            // Q carries the softmax scale and log2(e): S' = (c q) k is the score in the exp2 domain, and with the running maximum
            // as the MFMA's initial accumulator the exponent argument leaves the matrix pipe ready — no scale-and-subtract per
            // score (32 VALU instructions per tile and wave in a loop that is bound by its issue slots, not by the matrix pipe).
            // (the loads are issued here; the scaling waits for them below, after the first K tiles have been requested as well)
            qf[qb][ks] = *reinterpret_cast<const bf16x8*>(qp + 32 * ks);
        }
    }

    // ---- staging (LDS-DMA, buffer form): a tile is 16 pieces of 1 KiB; piece P = NW jj + wave (jj < NP) is staged by `wave`:
    // K rows 4 P .. 4 P + 3 (lane L: row 4 P + (L >> 4), physical chunk L & 15 = logical chunk ^ (row & 15)), V^T rows 8 P .. 8 P + 7
    // (lane L: row 8 P + (L >> 3), physical chunk L & 7 = logical chunk ^ ((row >> 1) & 7)).  With this piece order the lane's
    // swizzle does not depend on jj (row & 15 = 4 wave + (L >> 4)): ONE per-lane offset each for K and V^T, the piece and the tile
    // position go into the instruction's SGPR offset.
    const int dma_r = lane >> 4;
    unsigned k_off0, vt_off0;
    {
        const int row = 4 * wave + dma_r;
        const int lch = (lane & 15) ^ (row & 15);
        k_off0 = ((unsigned)row * (unsigned)p.k_stride + head * HD + lch * 8) * 2u;                 // bytes
        const int vrow = 8 * wave + (lane >> 3);
        const int vch = (lane & 7) ^ ((vrow >> 1) & 7);
        vt_off0 = (unsigned)(((long)head * HD + vrow) * p.kv_pad + vch * 8) * 2u;                   // bytes
    }
    const unsigned kstep = KVB * (unsigned)p.k_stride * 2u;
    const unsigned k_piece = 4u * NW * (unsigned)p.k_stride * 2u, vt_piece = 8u * NW * (unsigned)p.kv_pad * 2u;   // bytes between a wave's pieces
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4s;
    auto make_srd = [](const void* base) {
        const unsigned long b = (unsigned long)base;
        u32x4s s;
        s[0] = (unsigned)b;
        s[1] = (unsigned)(b >> 32) & 0xffffu;
        s[2] = 0xffffffffu;
        s[3] = 0x00020000u;
        return s;
    };
    const u32x4s srd_v = make_srd(p.vt);
    auto dma16b = [&](const u32x4s& srd, unsigned voff_bytes, unsigned soff_bytes, GF_LDS char* l) __attribute__((always_inline)) {
        // M0 (the LDS destination) is ours: it is a reserved register the compiler never allocates and, on gfx9+, sets itself right
        // before the few instructions that read it (none in this kernel) — so it is set here and not saved / restored
        const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)l);
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %3 offen lds"
                     :
                     : "v"(voff_bytes), "s"(srd), "s"(dst), "s"(soff_bytes)
                     : "memory");
    };
    // The K descriptor of tile t starts at the tile and ends with the key sequence (num_records = bytes of the rows that exist,
    // 0 for a tile past the end): rows of a ragged last tile beyond kv_len, and whole tiles that do not exist, arrive as zeros
    // with no test in the instruction stream (their scores are masked; a tile past the end is never read).
    auto srd_k_tile = [&](int t) __attribute__((always_inline)) {
        const unsigned long bb = (unsigned long)p.k + (unsigned long)(unsigned)t * kstep;
        const int rem = p.kv_len - t * KVB;
        u32x4s sd;
        sd[0] = (unsigned)bb;
        sd[1] = (unsigned)(bb >> 32) & 0xffffu;
        sd[2] = rem > 0 ? (unsigned)rem * (unsigned)p.k_stride * 2u : 0u;
        sd[3] = 0x00020000u;
        return sd;
    };
    // which = 0: K tile t -> K buffer buf;  which = 1: V^T tile t -> V buffer buf;  jj = which of the wave's NP pieces
    auto stage_piece = [&](int which, int t, int buf, int jj) __attribute__((always_inline)) {
        GF_LDS char* base = lds + which * AT3_V_BASE + buf * KV_TILE_BYTES + (NW * jj + wave) * 1024;
        if (which)
            dma16b(srd_v, vt_off0, (unsigned)t * (KVB * 2u) + (unsigned)jj * vt_piece, base);
        else
            dma16b(srd_k_tile(t), k_off0, (unsigned)jj * k_piece, base);
    };
    auto stage = [&](int which, int t, int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int jj = 0; jj < NP; ++jj) stage_piece(which, t, buf, jj);
    };

    // ---- fragment read offsets (bytes inside a buffer): K (kb, ks): koff[ks] + kb * 4096;  V^T (db, kk): voff[kk] + db * 2048
    int koff[4], voff[2];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) koff[ks] = 256 * r + 16 * ((4 * ks + g) ^ r);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) voff[kk] = AT3_V_BASE + 128 * r + 16 * ((4 * kk + g) ^ ((r >> 1) & 7));

    // oacc[db][qb], db < 8: O^T;  oacc[8][qb]: the row sums — a ninth "d block" whose V^T fragment is the constant `ones` (row 0
    // all ones): the matrix pipe adds up the bf16 P it multiplies with anyway (4 MFMAs per tile instead of 32 v_add_f32), and the
    // sums are rescaled together with O.  sc[tile parity][kb][qb] = S' - m_run (log2 domain).
    constexpr int NDB = 9;
    f32x4 oacc[NDB][NQ], sc[2][4][NQ];
    u32x4 pfw[2][NQ];                // P fragments [kk][qb] as four packed bf16 pairs (word w = keys 2w, 2w+1 of the B operand)
    auto pf = [&](int kk, int qb) __attribute__((always_inline)) { return __builtin_bit_cast(bf16x8, pfw[kk][qb]); };
    // The running maximum lives in the exp2 domain as its negative splat only (negm), and the scores are kept relative to it.
    float alpha_pend[NQ];
    f32x4 negm[NQ];                  // -(running maximum) splat: the QK^T chains start from it
#pragma unroll
    for (int qb = 0; qb < NQ; ++qb) {
        alpha_pend[qb] = 1.f;
        negm[qb] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    bool pend = false;               // wave-uniform: O still has to be multiplied by alpha_pend
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
        for (int qb = 0; qb < NQ; ++qb) oacc[db][qb] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 ones;
    {
        const __bf16 o1 = (__bf16)(r == 0 ? 1.0f : 0.0f);
#pragma unroll
        for (int e = 0; e < 8; ++e) ones[e] = o1;
    }
    const int nt = (p.kv_len + KVB - 1) / KVB;
    const bool ragged = (p.kv_len & (KVB - 1)) != 0;
    typedef std::integral_constant<int, 0> C0;
    typedef std::integral_constant<int, 1> C1;

    // One wave per SIMD with 64 query rows (NQ = 4: every fragment read feeds four MFMAs, O^T and Q in AGPRs through register-class
    // constraints on asm MFMAs) was built and measured: 18.7 ms on all-zero inputs against 13.1 ms — a lone wave cannot issue the
    // phase's ~8 instructions per MFMA pair fast enough; this file keeps the NQ arithmetic but ships and tests NQ = 2 only.
    static_assert(NQ == 2, "kernel 3 is validated for two query blocks per wave");
    auto mfma_pv = [&](f32x4& acc, const bf16x8& va, const bf16x8& pb) __attribute__((always_inline)) { mfma16(acc, va, pb); };
    auto mfma_qk = [&](f32x4& acc, const bf16x8& ka, const bf16x8& qb_) __attribute__((always_inline)) { mfma16(acc, ka, qb_); };
    auto mfma_qk_first = [&](f32x4& dst, const bf16x8& ka, const bf16x8& qb_, const f32x4& c0) __attribute__((always_inline)) {
        dst = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ka, qb_, c0, 0, 0, 0);
    };

    // ---- unpipelined building blocks (prologue, first and last phase)
    auto qk_plain = [&](auto par_c) __attribute__((always_inline)) {   // S(par) = K(buffer par) Q^T
        constexpr int PAR = decltype(par_c)::value;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int qb = 0; qb < NQ; ++qb) sc[PAR][kb][qb] = negm[qb];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                const bf16x8 kf = *(GF_LDS bf16x8*)(lds + koff[ks] + kb * 4096 + PAR * KV_TILE_BYTES);
#pragma unroll
                for (int qb = 0; qb < NQ; ++qb) mfma_qk(sc[PAR][kb][qb], kf, qf[qb][ks]);
            }
    };
    auto pv_plain = [&](auto buf_c) __attribute__((always_inline)) {   // O^T += V(buffer)^T P^T
        constexpr int BUF = decltype(buf_c)::value;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int db = 0; db < 8; ++db) {
                const bf16x8 vf = *(GF_LDS bf16x8*)(lds + voff[kk] + db * 2048 + BUF * KV_TILE_BYTES);
#pragma unroll
                for (int qb = 0; qb < NQ; ++qb) mfma_pv(oacc[db][qb], vf, pf(kk, qb));
            }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int qb = 0; qb < NQ; ++qb) mfma_pv(oacc[NDB - 1][qb], ones, pf(kk, qb));
    };
    auto apply_pending = [&]() __attribute__((always_inline)) {
        if (pend) {
    #pragma unroll
            for (int db = 0; db < NDB; ++db)
#pragma unroll
                for (int qb = 0; qb < NQ; ++qb)
#pragma unroll
                    for (int e = 0; e < 4; ++e) oacc[db][qb][e] *= alpha_pend[qb];
            pend = false;
        }
    };
    // mx[qb] = this lane's partial maximum of its query's scores in tile PAR, RELATIVE to the running maximum (the scores are
    // S' - m_run).  Rare path (`first`: always): the running maximum moves to the tile's row maximum; the tile's scores, which the
    // matrix pipe produced against the old maximum, are corrected here, later tiles start from the new one (negm).
    auto new_max = [&](auto par_c, const float (&mx)[NQ], bool first) {
        constexpr int PAR = decltype(par_c)::value;
        float over = mx[0];        // how far any score of the tile is above its running maximum
#pragma unroll
        for (int qb = 1; qb < NQ; ++qb) over = fmaxf(over, mx[qb]);
        constexpr float QUIET = 6.0f;      // lazy rescale: scores may rise 6 (a factor 64 in P) above the last maximum
        const bool quiet = __all(over <= QUIET);
        if (first || !quiet) {
#pragma unroll
            for (int qb = 0; qb < NQ; ++qb) {
                float d = mx[qb];                        // reduce over the four g-lanes of the query
                d = fmaxf(d, __shfl_xor(d, 16));
                d = fmaxf(d, __shfl_xor(d, 32));
                if (!first) d = fmaxf(d, 0.f);       // the maximum never moves down (the first tile sets it, whatever its sign)
                const float m_new = d - negm[qb][0];
                const float alpha = __builtin_amdgcn_exp2f(-d);
#pragma unroll
                for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                    for (int e = 0; e < 4; ++e) sc[PAR][kb][qb][e] -= d;
#pragma unroll
                for (int e = 0; e < 4; ++e) negm[qb][e] = -m_new;
                alpha_pend[qb] = alpha;   // at most one outstanding: applied at the top of the next phase
            }
            pend = !first;                // (nothing has been accumulated yet when the first tile sets the maximum)
        }
    };
    auto mask_ragged = [&](auto par_c, int t) __attribute__((always_inline)) {
        constexpr int PAR = decltype(par_c)::value;
        if (ragged && t == nt - 1) {
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (t * KVB + 16 * kb + 4 * g + j >= p.kv_len) {
#pragma unroll
                        for (int qb = 0; qb < NQ; ++qb) sc[PAR][kb][qb][j] = -INFINITY;
                    }
        }
    };
    auto softmax_plain = [&](auto par_c, int t) __attribute__((always_inline)) {
        constexpr int PAR = decltype(par_c)::value;
        mask_ragged(par_c, t);
        float mx[NQ];
#pragma unroll
        for (int qb = 0; qb < NQ; ++qb) {
            mx[qb] = sc[PAR][0][qb][0];
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                for (int j = 0; j < 4; ++j) mx[qb] = fmaxf(mx[qb], sc[PAR][kb][qb][j]);
        }
        new_max(par_c, mx, t == 0);
#pragma unroll
        for (int qb = 0; qb < NQ; ++qb) {
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                for (int j = 0; j < 4; j += 2) {
                    const float p0 = __builtin_amdgcn_exp2f(sc[PAR][kb][qb][j]);
                    const float p1 = __builtin_amdgcn_exp2f(sc[PAR][kb][qb][j + 1]);
                    pfw[kb >> 1][qb][(kb & 1) * 2 + (j >> 1)] = pack2bf(p0, p1);
                }
        }
    };

    // ---- the steady phase p (1 <= p <= nt-2), PAR = p & 1: NS = 32 NQ matrix slots; slot s = (fragment f = s / NQ, query block
    // qb = s % NQ): a fragment read from LDS serves the NQ query blocks of the wave.
    //   f <  16: PV(p-1)  kk = f >> 3, db = f & 7          A = V^T fragment (db, kk) of V buffer 1-PAR
    //   f >= 16: QK(p+1)  ks = (f-16) >> 2, kb = (f-16) & 3  A = K fragment (kb, ks) of K buffer 1-PAR
    // Fragment f is read RING - 1 fragments ahead into ring entry f % RING.  The partial maxima of S(p) are taken in slots
    // 0 .. 4 NQ - 1, the rescale test follows, its 16 NQ scores per lane are exponentiated in slots 8 NQ .. NS - 1.
    constexpr int RING = K3_RING;
    bf16x8 fr[RING];
    auto frag_load = [&](auto f_c, auto par_c) __attribute__((always_inline)) {
        constexpr int F = decltype(f_c)::value, PAR = decltype(par_c)::value;
        if constexpr (F < 16) {
            constexpr int kk = F >> 3, db = F & 7;
            fr[F % RING] = *(GF_LDS bf16x8*)(lds + voff[kk] + db * 2048 + (1 - PAR) * KV_TILE_BYTES);
        } else if constexpr (F < 32) {
            constexpr int ks = (F - 16) >> 2, kb = (F - 16) & 3;
            fr[F % RING] = *(GF_LDS bf16x8*)(lds + koff[ks] + kb * 4096 + (1 - PAR) * KV_TILE_BYTES);
        }
    };
    auto mfma_op = [&](auto s_c, auto par_c) __attribute__((always_inline)) {
        constexpr int S = decltype(s_c)::value, PAR = decltype(par_c)::value;
        // query-block order reversed on odd fragments: one MFMA operand changes per slot instead of both
        constexpr int F = S / NQ, qb = (F & 1) ? NQ - 1 - S % NQ : S % NQ;
        if constexpr (F < 16) {
            constexpr int kk = F >> 3, db = F & 7;
            mfma_pv(oacc[db][qb], fr[F % RING], pf(kk, qb));
        } else {
            constexpr int ks = (F - 16) >> 2, kb = (F - 16) & 3;
            if constexpr (ks == 0) {
                mfma_qk_first(sc[1 - PAR][kb][qb], fr[F % RING], qf[qb][ks], negm[qb]);
            } else {
                mfma_qk(sc[1 - PAR][kb][qb], fr[F % RING], qf[qb][ks]);
            }
        }
    };
    // staging: the wave's NP V^T pieces of tile p, then its NP K pieces of tile p+2, one every K3_DMAS slots from slot K3_DMA0
    auto dma_slot = [&](auto s_c, auto par_c, int pidx) __attribute__((always_inline)) {
        constexpr int S = decltype(s_c)::value, PAR = decltype(par_c)::value;
        if constexpr (S >= K3_DMA0 && (S - K3_DMA0) % K3_DMAS == 0 && (S - K3_DMA0) / K3_DMAS < 2 * NP) {
            constexpr int i = (S - K3_DMA0) / K3_DMAS;
            if constexpr (i < NP) {
                stage_piece(1, pidx, PAR, i);
            } else {
                stage_piece(0, pidx + 2, PAR, i - NP);
            }
        }
    };
    auto phase = [&](auto par_c, int pidx) __attribute__((always_inline)) {
        constexpr int PAR = decltype(par_c)::value;
        apply_pending();
        static_for<0, RING - 1>([&](auto f_c) { frag_load(f_c, par_c); });
        float mx[NQ];
#pragma unroll
        for (int qb = 0; qb < NQ; ++qb) mx[qb] = -INFINITY;
        static_for<0, 4 * NQ>([&](auto s_c) {
            constexpr int S = decltype(s_c)::value;
            mfma_op(s_c, par_c);
            if constexpr (S % NQ == 0) frag_load(std::integral_constant<int, S / NQ + RING - 1>{}, par_c);
            // partial maxima: slot S takes key block S / NQ of query block S % NQ (4 scores: two v_max3)
            {
                constexpr int kb = S / NQ, qb = S % NQ;
                const f32x4& v = sc[PAR][kb][qb];
                mx[qb] = fmaxf(fmaxf(mx[qb], v[0]), v[1]);
                mx[qb] = fmaxf(fmaxf(mx[qb], v[2]), v[3]);
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        new_max(par_c, mx, false);
        float pe[2];
        static_for<4 * NQ, NS>([&](auto s_c) {
            constexpr int S = decltype(s_c)::value;
            mfma_op(s_c, par_c);
            // the row-sum MFMAs of PV(p-1) ride in the PV half: after the last slot that reads pf[kk]
            if constexpr (S == 8 * NQ - 1 || S == 16 * NQ - 1) {
#pragma unroll
                for (int qb = 0; qb < NQ; ++qb) mfma_pv(oacc[NDB - 1][qb], ones, pf(S / (8 * NQ), qb));
            }
            if constexpr (S % NQ == 0) frag_load(std::integral_constant<int, S / NQ + RING - 1>{}, par_c);
            dma_slot(s_c, par_c, pidx);
            // scores of S(p): two per three slots from slot 8 NQ on.  P(p) overwrites pf, which PV(p-1) still reads: pf[0][*] until
            // slot 8 NQ - 1, pf[1][*] until slot 16 NQ - 1 — so the scores that land in pf[0] (key blocks 0, 1) come first (slots
            // 8 NQ .. 20 NQ - 1) and those for pf[1] (key blocks 2, 3) from slot 20 NQ on.
            // e -> kk = e / (8 NQ), qb = (e >> 3) % NQ, kb = 2 kk + ((e >> 2) & 1), j = e & 3.
            // The exp2 and the pack are inline asm (a compiler-placed exp2 brought a register copy, a wait state and a late pack
            // per score: 5.8 issue slots per score instead of 2).  v_exp_f32 is a transcendental op: its result may not be read
            // by the next instruction — the row-sum add of a score sits one slot after its exp2, the pack of a pair in the pair's
            // third slot, each behind that slot's MFMA.
            if constexpr (S >= 8 * NQ) {
                constexpr int T = (S - 8 * NQ) % 3, e = ((S - 8 * NQ) / 3) * 2 + (T == 2 ? 1 : T);
                constexpr int kk = e / (8 * NQ), qb = (e >> 3) % NQ, kb = 2 * kk + ((e >> 2) & 1), j = e & 3;
                if constexpr (T != 2) {
                    asm volatile("v_exp_f32 %0, %1" : "=v"(pe[T]) : "v"(sc[PAR][kb][qb][j]));
                } else {
                    unsigned w;
                    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(w) : "v"(pe[0]), "v"(pe[1]));
                    pfw[kk][qb][(kb & 1) * 2 + (j >> 1)] = w;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    };

    // ---- prologue: K(0), K(1) staged; S(0)
    stage(0, 0, 0);
    if (nt > 1) stage(0, 1, 1);
    // Q <- bf16(Q * scale * log2 e), under the latency of the staging just issued
#pragma unroll
    for (int qb = 0; qb < NQ; ++qb)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int e = 0; e < 8; ++e) qf[qb][ks][e] = (__bf16)((float)qf[qb][ks][e] * p.scale_log2e);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    qk_plain(C0{});
    __syncthreads();   // every wave has read K(0) before K(2) overwrites its buffer
    // ---- phase 0 (no PV yet): K(2), V(0) in flight; S(1); softmax(0)
    if (nt > 2) stage(0, 2, 0);
    stage(1, 0, 0);
    softmax_plain(C0{}, 0);          // sets the running maximum (tile 0's row maximum) ...
    if (nt > 1) qk_plain(C1{});      // ... which S(1) already starts from
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // ---- steady phases 1 .. nt-2
    int pi = 1;
    for (; pi + 1 <= nt - 2; pi += 2) {
        phase(C1{}, pi);
        phase(C0{}, pi + 1);
    }
    if (pi <= nt - 2) {
        phase(C1{}, pi);
        ++pi;
    }
    // ---- last phase p = nt-1 (nt >= 2): V(nt-1) in flight; PV(nt-2); softmax(nt-1)
    if (nt >= 2) {
        const int par = (nt - 1) & 1;
        stage(1, nt - 1, par);
        apply_pending();
        if (par) {
            pv_plain(C0{});
            softmax_plain(C1{}, nt - 1);
        } else {
            pv_plain(C1{});
            softmax_plain(C0{}, nt - 1);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    // ---- PV(nt-1)
    apply_pending();
    if ((nt - 1) & 1)
        pv_plain(C1{});
    else
        pv_plain(C0{});

    // ---- epilogue
    {
        float inv[NQ];
#pragma unroll
        for (int qb = 0; qb < NQ; ++qb) {
            const float l = __shfl(oacc[NDB - 1][qb][0], r);        // row 0 of the ninth block: lane g == 0 of the query holds it
            inv[qb] = 1.0f / l;
            const int qrow = q0 + 16 * qb + r;
            if (p.lse && g == 0 && qrow < p.q_len) p.lse[(long)qrow * p.heads + head] = -negm[qb][0] + __builtin_amdgcn_logf(l);
        }
        // O leaves through LDS as whole 256-byte rows: the wave's RW x 256 B image, 16-byte chunk ch of row q at chunk ch ^ (q & 15);
        // a lane holds the 8-byte pieces d = 16 db + 4 g .. +3 of its NQ rows.  The K/V tiles are dead at this barrier.
        __syncthreads();
        GF_LDS char* ob = lds + wave * (RW * 256);
#pragma unroll
        for (int qb = 0; qb < NQ; ++qb)
#pragma unroll
            for (int db = 0; db < 8; ++db) {
                u32x2 pk;
                pk[0] = pack2bf(oacc[db][qb][0] * inv[qb], oacc[db][qb][1] * inv[qb]);
                pk[1] = pack2bf(oacc[db][qb][2] * inv[qb], oacc[db][qb][3] * inv[qb]);
                const int row = 16 * qb + r;
                *(GF_LDS u32x2*)(ob + row * 256 + (((2 * db + (g >> 1)) ^ (row & 15)) << 4) + 8 * (g & 1)) = pk;
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // wave-local image: LDS is in order, no barrier needed
#pragma unroll
        for (int it = 0; it < RW / 4; ++it) {
            const int row = 4 * it + (lane >> 4), ch = lane & 15;
            const u16x8 v8 = *(GF_LDS u16x8*)(ob + row * 256 + ((ch ^ (row & 15)) << 4));
            if (q0 + row < p.q_len)
                *reinterpret_cast<u16x8*>(p.o + (long)(q0 + row) * p.o_stride + head * HD + 8 * ch) = v8;
        }
    }
}

// V [kv_len, heads*128] -> V^T [heads][128][kv_pad] bf16 for kernel 3: keys >= kv_len zero; inside every group of 32 keys position
// 8 g + i holds key 4 g + i (i < 4) or 16 + 4 g + (i - 4) (i >= 4) — the k order of the 16x16x32 B operand built from two score tiles.
__global__ __launch_bounds__(256) void transpose_v32_kernel(const u16* __restrict__ v, u16* __restrict__ vt, int kv_len, long kv_pad,
                                                            long v_stride) {
    __shared__ u16 tile[KVB][HD + 8];
    const int t0 = blockIdx.x * KVB, head = blockIdx.y, tid = threadIdx.x;
#pragma unroll
    for (int it = 0; it < 4; ++it) {                 // 64 rows x 16 chunks of 16 B
        const int idx = it * 256 + tid, row = idx >> 4, ch = idx & 15;
        u16x8 val = {0, 0, 0, 0, 0, 0, 0, 0};
        if (t0 + row < kv_len) val = *reinterpret_cast<const u16x8*>(v + (long)(t0 + row) * v_stride + head * HD + ch * 8);
        *reinterpret_cast<u16x8*>(&tile[row][ch * 8]) = val;
    }
    __syncthreads();
    const int d = tid >> 1, half = tid & 1;
    u16* dst = vt + ((long)head * HD + d) * kv_pad + t0 + 32 * half;
#pragma unroll
    for (int cg = 0; cg < 4; ++cg) {                 // chunk cg = the 8 positions of lane group g = cg
        u16x8 o;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int key = 32 * half + (i < 4 ? 4 * cg + i : 16 + 4 * cg + (i - 4));
            o[i] = tile[key][d];
        }
        *reinterpret_cast<u16x8*>(dst + 8 * cg) = o;
    }
}

// V [kv_len, heads*128] -> V^T [heads][128][kv_pad] bf16, keys >= kv_len zero; inside every group of 16 keys the order is
// 0-3, 8-11, 4-7, 12-15 (what a lane half of the 32x32x16 B operand holds).  One workgroup per (64-key tile, head).
__global__ __launch_bounds__(256) void transpose_v_kernel(const u16* __restrict__ v, u16* __restrict__ vt, int kv_len, long kv_pad,
                                                          long v_stride) {
    __shared__ u16 tile[KVB][HD + 8];
    const int t0 = blockIdx.x * KVB, head = blockIdx.y, tid = threadIdx.x;
#pragma unroll
    for (int it = 0; it < 4; ++it) {                 // 64 rows x 16 chunks of 16 B
        const int idx = it * 256 + tid, row = idx >> 4, ch = idx & 15;
        u16x8 val = {0, 0, 0, 0, 0, 0, 0, 0};
        if (t0 + row < kv_len) val = *reinterpret_cast<const u16x8*>(v + (long)(t0 + row) * v_stride + head * HD + ch * 8);
        *reinterpret_cast<u16x8*>(&tile[row][ch * 8]) = val;
    }
    __syncthreads();
    const int d = tid >> 1, half = tid & 1;
    u16* dst = vt + ((long)head * HD + d) * kv_pad + t0 + 32 * half;
#pragma unroll
    for (int c = 0; c < 4; ++c) {                    // 4 chunks of 8 keys
        u16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int pos = 32 * half + 8 * c + j;   // physical position in the tile row
            const int q16 = pos & 15;
            const int key = (pos & ~15) | ((q16 & 3) | ((q16 & 4) << 1) | ((q16 & 8) >> 1));   // swap bits 2 and 3
            o[j] = tile[key][d];
        }
        *reinterpret_cast<u16x8*>(dst + 8 * c) = o;
    }
}

}  // namespace

static int flash_attn_fwd_impl(const void* q, const void* k, const void* v, void* o, float* lse, int64_t q_len, int64_t kv_len,
                               int64_t heads, int64_t head_dim, int64_t q_stride, int64_t k_stride,
                               int64_t v_stride, int64_t o_stride, float scale, void* stream, const void* vt = nullptr,
                               int64_t kv_pad = 0, float last_key_multiplicity = 1.0f) {
    if (vt) {   // pre-transposed V: v itself is not read
        v = k;
        v_stride = k_stride;
    }
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
    static GfDeviceOnce once;
    hipError_t e = gf_once_per_device(once, [] {
        hipError_t r = hipFuncSetAttribute(reinterpret_cast<const void*>(flash_attn_fwd_kernel2<false>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, AT2_LDS);
        if (r == hipSuccess)
            r = hipFuncSetAttribute(reinterpret_cast<const void*>(flash_attn_fwd_kernel2<true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, AT2_LDS);
        return r;
    });
    if (e != hipSuccess) {
        gf_set_error("gf_flash_attn_fwd: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
        return GF_ERR_LAUNCH;
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
    a.lse = lse;
    a.vt = (const u16*)vt;
    a.kv_pad = kv_pad;
    a.last_key_bias = 0.f;
    if (last_key_multiplicity != 1.0f) {
        GF_CHECK_ARG(last_key_multiplicity >= 1.0f && !vt && scale > 0.f,
                     "gf_flash_attn_fwd_lastmult: multiplicity >= 1, plain-V form only (not the V^T form)");
        a.last_key_bias = log2f(last_key_multiplicity) / a.scale_log2e;
    }
    a.dbg = nullptr;
    if (vt)
        hipLaunchKernelGGL(flash_attn_fwd_kernel2<true>, dim3((unsigned)(a.n_qblocks * a.heads)), dim3(AT2_THREADS), AT2_LDS,
                           (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL(flash_attn_fwd_kernel2<false>, dim3((unsigned)(a.n_qblocks * a.heads)), dim3(AT2_THREADS), AT2_LDS,
                           (hipStream_t)stream, a);
    GF_CHECK_LAUNCH("gf_flash_attn_fwd");
    return GF_OK;
}

extern "C" GF_API int gf_flash_attn_fwd(const void* q, const void* k, const void* v, void* o, int64_t q_len, int64_t kv_len,
                                 int64_t heads, int64_t head_dim, int64_t q_stride, int64_t k_stride,
                                 int64_t v_stride, int64_t o_stride, float scale, void* stream) {
    return flash_attn_fwd_impl(q, k, v, o, nullptr, q_len, kv_len, heads, head_dim, q_stride, k_stride, v_stride, o_stride, scale,
                               stream);
}

extern "C" GF_API int gf_flash_attn_fwd_lastmult(const void* q, const void* k, const void* v, void* o, int64_t q_len, int64_t kv_len,
                                                  int64_t heads, int64_t head_dim, int64_t q_stride, int64_t k_stride, int64_t v_stride,
                                                  int64_t o_stride, float scale, float last_key_multiplicity, void* stream) {
    return flash_attn_fwd_impl(q, k, v, o, nullptr, q_len, kv_len, heads, head_dim, q_stride, k_stride, v_stride, o_stride, scale,
                               stream, nullptr, 0, last_key_multiplicity);
}

extern "C" GF_API int gf_flash_attn_fwd_lse(const void* q, const void* k, const void* v, void* o, float* lse, int64_t q_len,
                                     int64_t kv_len, int64_t heads, int64_t head_dim, int64_t q_stride, int64_t k_stride,
                                     int64_t v_stride, int64_t o_stride, float scale, void* stream) {
    GF_CHECK_ARG(lse, "gf_flash_attn_fwd_lse: null lse");
    return flash_attn_fwd_impl(q, k, v, o, lse, q_len, kv_len, heads, head_dim, q_stride, k_stride, v_stride, o_stride, scale,
                               stream);
}

extern "C" GF_API int gf_transpose_v(const void* v, int64_t v_stride, void* vt, int64_t kv_len, int64_t kv_pad, int64_t heads,
                                     void* stream) {
    GF_CHECK_ARG(v && vt, "gf_transpose_v: null pointer");
    GF_CHECK_ARG(kv_len > 0 && heads > 0 && kv_pad >= kv_len && kv_pad % KVB == 0 && v_stride % 8 == 0 && v_stride >= heads * HD,
                 "gf_transpose_v: kv_pad must be a multiple of 64 covering kv_len; v_stride must cover heads*128");
    GF_CHECK_ARG(gf_aligned16(v) && gf_aligned16(vt), "gf_transpose_v: 16-byte alignment required");
    GF_CHECK_ARG(heads * HD * kv_pad < (1LL << 31), "gf_transpose_v: heads*128*kv_pad must stay below 2^31 elements");
    hipLaunchKernelGGL(transpose_v_kernel, dim3((unsigned)(kv_pad / KVB), (unsigned)heads), dim3(256), 0, (hipStream_t)stream,
                       (const u16*)v, (u16*)vt, (int)kv_len, (long)kv_pad, (long)v_stride);
    GF_CHECK_LAUNCH("gf_transpose_v");
    return GF_OK;
}

extern "C" GF_API int gf_flash_attn_fwd_vt(const void* q, const void* k, const void* vt, void* o, float* lse, int64_t q_len,
                                           int64_t kv_len, int64_t kv_pad, int64_t heads, int64_t head_dim, int64_t q_stride,
                                           int64_t k_stride, int64_t o_stride, float scale, void* stream) {
    GF_CHECK_ARG(vt && kv_pad >= kv_len && kv_pad % KVB == 0 && gf_aligned16(vt), "gf_flash_attn_fwd_vt: bad V^T buffer");
    GF_CHECK_ARG(heads * HD * kv_pad < (1LL << 31), "gf_flash_attn_fwd_vt: heads*128*kv_pad must stay below 2^31 elements");
    return flash_attn_fwd_impl(q, k, k, o, lse, q_len, kv_len, heads, head_dim, q_stride, k_stride, k_stride, o_stride, scale,
                               stream, vt, kv_pad);
}

extern "C" GF_API int gf_transpose_v32(const void* v, int64_t v_stride, void* vt, int64_t kv_len, int64_t kv_pad, int64_t heads,
                                       void* stream) {
    GF_CHECK_ARG(v && vt, "gf_transpose_v32: null pointer");
    GF_CHECK_ARG(kv_len > 0 && heads > 0 && kv_pad >= kv_len && kv_pad % KVB == 0 && v_stride % 8 == 0 && v_stride >= heads * HD,
                 "gf_transpose_v32: kv_pad must be a multiple of 64 covering kv_len; v_stride must cover heads*128");
    GF_CHECK_ARG(gf_aligned16(v) && gf_aligned16(vt), "gf_transpose_v32: 16-byte alignment required");
    GF_CHECK_ARG(heads * HD * kv_pad < (1LL << 31), "gf_transpose_v32: heads*128*kv_pad must stay below 2^31 elements");
    hipLaunchKernelGGL(transpose_v32_kernel, dim3((unsigned)(kv_pad / KVB), (unsigned)heads), dim3(256), 0, (hipStream_t)stream,
                       (const u16*)v, (u16*)vt, (int)kv_len, (long)kv_pad, (long)v_stride);
    GF_CHECK_LAUNCH("gf_transpose_v32");
    return GF_OK;
}

extern "C" GF_API int gf_flash_attn_fwd_vt32(const void* q, const void* k, const void* vt, void* o, float* lse, int64_t q_len,
                                             int64_t kv_len, int64_t kv_pad, int64_t heads, int64_t head_dim, int64_t q_stride,
                                             int64_t k_stride, int64_t o_stride, float scale, void* stream) {
    GF_CHECK_ARG(q && k && vt && o, "gf_flash_attn_fwd_vt32: null pointer");
    if (head_dim != HD) {
        gf_set_error("gf_flash_attn_fwd_vt32: head_dim=%ld unsupported (kernel is built for 128)", (long)head_dim);
        return GF_ERR_UNSUPPORTED;
    }
    GF_CHECK_ARG(q_len >= 0 && kv_len >= 2 * KVB && heads > 0 && q_len < (1 << 30) && kv_len < (1 << 30),
                 "gf_flash_attn_fwd_vt32: bad lengths q=%ld kv=%ld heads=%ld (kv_len >= 128)", (long)q_len, (long)kv_len, (long)heads);
    GF_CHECK_ARG(kv_pad >= kv_len && kv_pad % KVB == 0 && heads * HD * kv_pad < (1LL << 31), "gf_flash_attn_fwd_vt32: bad V^T buffer");
    GF_CHECK_ARG(q_stride % 8 == 0 && k_stride % 8 == 0 && o_stride % 4 == 0 && q_stride >= heads * HD && k_stride >= heads * HD &&
                     o_stride >= heads * HD,
                 "gf_flash_attn_fwd_vt32: strides must cover heads*128 and be multiples of 8");
    GF_CHECK_ARG(gf_aligned16(q) && gf_aligned16(k) && gf_aligned16(vt) && gf_aligned16(o),
                 "gf_flash_attn_fwd_vt32: 16-byte alignment required");
    GF_CHECK_ARG((kv_len + 64) * k_stride < (1LL << 31), "gf_flash_attn_fwd_vt32: kv_len*stride must stay below 2^31 elements");
    if (q_len == 0) return GF_OK;
    static GfDeviceOnce once;
    hipError_t e = gf_once_per_device(once, [] {
        return hipFuncSetAttribute(reinterpret_cast<const void*>(flash_attn_fwd_kernel3<K3_NQ>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, AT3_LDS);
    });
    if (e != hipSuccess) {
        gf_set_error("gf_flash_attn_fwd_vt32: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
        return GF_ERR_LAUNCH;
    }
    AttnArgs a;
    a.q = (const u16*)q;
    a.k = (const u16*)k;
    a.v = (const u16*)k;
    a.o = (u16*)o;
    a.q_len = (int)q_len;
    a.kv_len = (int)kv_len;
    a.heads = (int)heads;
    a.n_qblocks = (int)((q_len + QB - 1) / QB);
    a.q_stride = q_stride;
    a.k_stride = k_stride;
    a.v_stride = k_stride;
    a.o_stride = o_stride;
    a.scale_log2e = scale * 1.4426950408889634f;
    a.lse = lse;
    a.vt = (const u16*)vt;
    a.kv_pad = kv_pad;
    a.dbg = nullptr;
    a.last_key_bias = 0.f;
    hipLaunchKernelGGL(flash_attn_fwd_kernel3<K3_NQ>, dim3((unsigned)(a.n_qblocks * a.heads)), dim3(At3<K3_NQ>::THREADS), AT3_LDS,
                       (hipStream_t)stream, a);
    GF_CHECK_LAUNCH("gf_flash_attn_fwd_vt32");
    return GF_OK;
}
