// gf_gemm.hip — bf16 GEMM  C[M,N] = epi(A[M,K] · W[N,K]^T + bias)  for every Linear / 1x1 conv of
// the Wan DiT forward (reference: F.linear call sites diffsynth/models/wan_video_dit.py:131-146,
// 209-210, 263, 309-320; zero-conv src/goal_force/wan_video_new.py:1565-1570).
//
// CDNA4 design (gfx950):
//   * 256x256 output tile per 512-thread workgroup (8 waves as 2(M) x 4(N), 128x64 per wave),
//     K-step 64, v_mfma_f32_16x16x32_bf16, fp32 accumulators (128 VGPRs per lane).
//   * Both operands have K contiguous (activations [M,K], nn.Linear weight [N,K]), which is the
//     native fragment order of the 16x16x32 MFMA: each lane reads 8 consecutive k of one row.
//   * Tiles are staged HBM -> LDS with global_load_lds_dwordx4 (LDS-DMA, 1 KiB per wave
//     instruction, no VGPR round trip), double buffered (2 x (32 KiB A + 32 KiB B) = 128 KiB),
//     one barrier per K-step: the DMA of tile t+1 is in flight while tile t is multiplied.
//   * LDS image: 128-byte rows, 16-byte chunk index XOR (row & 7).  The DMA writes lane-linear, so
//     the permutation is applied to the per-lane SOURCE address and again on the ds_read_b128
//     (conflict-free for the 16x16x32 fragment read).
//   * MFMA operands are swapped (D = W_frag x A_frag) so each lane ends up with 4 consecutive n of
//     one m; the epilogue transposes through LDS and touches HBM only with full 128-byte rows
//     (16 B per lane) for C, the residual and the bias/gate vectors.
//   * blockIdx -> tile map is XCD-aware: each XCD (blockIdx % 8) owns a contiguous range of tiles,
//     walked in groups of 8 M-tiles x all N-tiles so the 32 co-resident blocks of an XCD share
//     A/W panels through that XCD's L2.
//   * M and N tails: loads clamp the row index, stores are masked.
#include "gf_common.h"
#include <cstdlib>
#include <type_traits>

// The first bf16 kernels that no longer ship (the one-barrier 8-wave kernel in bf16, the slot-scheduled `sl` / `sl8` kernels) and the
// stamp / what-if / alternative-loop diagnostic builds are NOT in this file: tools/patches/gemm_experiments.patch re-creates them.

namespace {

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int GEMM_THREADS = 512;
constexpr int TILE_BYTES = BM * BK * 2;       // 32 KiB per operand tile
constexpr int STAGE_BYTES = 2 * TILE_BYTES;   // A + B
constexpr int GEMM_LDS = 2 * STAGE_BYTES;     // 128 KiB
constexpr int GROUP_M = 8;           // 16: -14 % (18 instead of 12 operand slices per K step and XCD), 4: +-2 % (EXPERIMENTS.md)

// Implicit-GEMM convolution (CONV = true instantiations of the phased kernel): the A operand is never materialised — row m
// = output pixel (j, Y, X), column k = (tap, channel) with tap = (dt*ks + dy)*ks + dx, exactly the patch matrix of
// gf_vae.hip's im2col_kernel (same modes), and every 16-byte LDS-DMA piece (8 channels of one tap of one pixel) is fetched
// straight from the channels-last activation [T, H, W, C] (or the 2-frame causal cache, or a zero page for padding).
struct ConvGeom {
    const u16* src;     // [T_in, H, W, C]
    const u16* cache;   // [2, H, W, C]: frames -2, -1 (kt == 3)
    const u16* zero;    // >= 16 zero bytes
    int H, W, C, Ho, Wo, kt, ks, mode, t_stride, t_off, taps;
    int hist_front;     // kt == 3 without `cache`: frames -2, -1 lie directly in front of src (one [2 + T, H, W, C] buffer)
    int c64;            // C % 64 == 0 and no K padding: a 64-channel K tile lies inside ONE tap (the tap arithmetic is then scalar)
    long frame;         // H*W*C
    float inv_c, inv_ks2;
};

struct GemmArgs {
    const u16* A;
    const u16* W;
    const u16* bias;
    u16* C;
    const u16* R;
    const u16* gate;
    const float* row_scale;  // FP8 only: per-row dequant scale of A (gf_quant_fp8_rowscale)
    int M, N, K;
    long lda, ldw, ldc, ldr;
    long sA = 0, sW = 0, sC = 0;   // gf_gemm_bf16_batched (gemm_ph_kernel, blockIdx.y = batch index): element strides between the problems
    int tiles_m, tiles_n;
    unsigned long long* rsv0 = nullptr;   // reserved: the kernarg layout the shipped kernels were scheduled and measured with is kept
    ConvGeom cv;
    int rsv1 = 0;
    int stagger;  // gemm_a4_kernel: column tile j starts its K loop at K tile (stagger * j) mod nk (option "a4_stagger", default 2; 0 = off)
    int wrows;    // gemm_a4_kernel: rows of W that exist (= N except for GF_EPI_VT32, where N = kv_pad covers zero columns past kv_len)
    int stagger_rows;   // gemm_a4_kernel: the K-loop rotation follows the ROW tile (GF_EPI_VT32: the operands are swapped, see gf_linear_vt32)
    int group_m;        // gemm_a4_kernel: row tiles per workgroup-order group (set by launch_gemm_a4)
    int rsv2 = 0;
};

// internal epilogue of gf_linear_vt32 (not in goalforce.h's enum): C rows = output features, C columns = keys in kernel 3's
// order, bias per ROW, columns >= wrows zero
constexpr int GF_EPI_VT32 = 6;

__device__ __forceinline__ void glds16(const void* g, GF_LDS char* l) {
    __builtin_amdgcn_global_load_lds((const GF_GLOBAL void*)g, (GF_LDS void*)l, 16, 0, 0);
}

typedef __attribute__((ext_vector_type(8))) int i32x8;

// The Linear's own output is rounded to bf16 before anything else happens to it.  Where the epilogue applies a function to it
// (GELU, SiLU) that rounding is explicit (rbf); otherwise the pack below IS that rounding — rounding twice to the same grid
// changes nothing, and leaving the first one out saves ~10 VALU instructions per 4 outputs (half of the epilogue's VALU work).
template <int EPI>
__device__ __forceinline__ float gf_epi_act(float lin) {
    if constexpr (EPI == GF_EPI_BIAS_GELU_TANH) return gelu_tanh_f(rbf(lin));
    else if constexpr (EPI == GF_EPI_BIAS_SILU) {
        const float y = rbf(lin);
        return y / (1.0f + expf(-y));
    } else return lin;
}

// FP8 = false: A/W are bf16, K-step 64 elements.  FP8 = true: A/W are OCP e4m3 bytes, K-step 128 elements — the
// LDS image is the same 128-byte rows, each lane's fragment is 32 consecutive k (two 16-byte chunks) and the
// product runs on v_mfma_scale_f32_16x16x128_f8f6f4 with unit (E8M0 = 127) block scales: 2x the bf16 MFMA rate.
// p.lda / p.ldw / p.K are in ELEMENTS of the operand type; the staging code works in 16-byte chunks.
template <int EPI, bool FP8>
__global__ __launch_bounds__(GEMM_THREADS, 2) void gemm_kernel(const GemmArgs p) {
    static_assert(FP8, "only the fp8 form of this kernel ships (launch_gemm); bf16 with M < 512 runs gemm_ph_kernel");
    constexpr int ESZ = FP8 ? 1 : 2;          // bytes per operand element
    constexpr int BKE = 128 / ESZ;            // K elements per 128-byte tile row
    extern __shared__ __attribute__((aligned(16))) char smem[];
    GF_LDS char* lds = (GF_LDS char*)smem;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // ---- XCD-aware tile mapping (bijective for any grid size) ------------------------------
    const int nwg = p.tiles_m * p.tiles_n;
    int v;
    {
        const int pid = blockIdx.x;
        const int xcd = pid & 7, local = pid >> 3;
        const int q = nwg >> 3, r = nwg & 7;
        v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
    }
    const int per_group = GROUP_M * p.tiles_n;
    const int group = v / per_group;
    const int first_m = group * GROUP_M;
    const int gsz = min(p.tiles_m - first_m, GROUP_M);
    const int in_group = v - group * per_group;
    const int tile_m = first_m + in_group % gsz;
    const int tile_n = in_group / gsz;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    // ---- staging addresses: wave w issues row-groups g = 4w..4w+3 (8 rows x 128 B each) -------
    // lane l of a DMA instruction fills LDS chunk (l&7) of row (l>>3); it must hold the logical
    // chunk (l&7) ^ (row&7) of that row.
    const int srow = lane >> 3;
    const int schunk = (lane & 7) ^ srow;
    const char* a_src[4];
    const char* b_src[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (wave * 4 + i) * 8 + srow;
        const int am = min(m0 + row, p.M - 1);
        const int bn = min(n0 + row, p.N - 1);
        a_src[i] = (const char*)p.A + ((long)am * p.lda) * ESZ + schunk * 16;
        b_src[i] = (const char*)p.W + ((long)bn * p.ldw) * ESZ + schunk * 16;
    }
    auto stage = [&](int buf, int kt) {
        GF_LDS char* sa = lds + buf * STAGE_BYTES + wave * 4096;
        GF_LDS char* sb = sa + TILE_BYTES;
        const int koff = kt * 128;  // bytes along K
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            glds16(a_src[i] + koff, sa + i * 1024);
            glds16(b_src[i] + koff, sb + i * 1024);
        }
    };

    // ---- fragment read addresses ---------------------------------------------------------------
    const int wm = wave >> 2, wn = wave & 3;
    const int frow = lane & 15;   // row inside a 16-row fragment
    const int fq = lane >> 4;     // k-chunk (8 elements) inside a 32-deep MFMA step
    // byte offset of (row, chunk) in a tile: row*128 + ((chunk ^ (row&7)) << 4); row&7 == frow&7
    // because all fragment bases are multiples of 16.
    const int sw = frow & 7;
    const int a_base = (wm * 128 + frow) * 128;
    const int b_base = TILE_BYTES + (wn * 64 + frow) * 128;
    const int ch0 = ((0 * 4 + fq) ^ sw) << 4;   // k-substep 0
    const int ch1 = ((1 * 4 + fq) ^ sw) << 4;   // k-substep 1

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = p.K / BKE;
    stage(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();  // tile kt landed for every wave; everyone is done reading the other buffer
        if (kt + 1 < nk) stage((kt + 1) & 1, kt + 1);
        GF_LDS char* sbuf = lds + (kt & 1) * STAGE_BYTES;
        if constexpr (!FP8) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int ch = ks ? ch1 : ch0;
                bf16x8 af[8], bfr[4];
#pragma unroll
                for (int i = 0; i < 8; ++i) af[i] = *(GF_LDS bf16x8*)(sbuf + a_base + i * 2048 + ch);
#pragma unroll
                for (int j = 0; j < 4; ++j) bfr[j] = *(GF_LDS bf16x8*)(sbuf + b_base + j * 2048 + ch);
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        // swapped operands: D[n-local][m-local]; lane holds 4 consecutive n of one m
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
            }
        } else {
            // lane (row = lane&15, kb = lane>>4) holds bytes [32kb, 32kb+32) of its row = chunks 2kb, 2kb+1.
            // A and W fragments are loaded by the same rule, so slot e of lane-group kb pairs the same k on both.
            const int c0 = ((2 * fq) ^ sw) << 4, c1 = ((2 * fq + 1) ^ sw) << 4;
            i32x8 bfr[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const u32x4 lo = *(GF_LDS u32x4*)(sbuf + b_base + j * 2048 + c0);
                const u32x4 hi = *(GF_LDS u32x4*)(sbuf + b_base + j * 2048 + c1);
                bfr[j] = i32x8{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const u32x4 lo = *(GF_LDS u32x4*)(sbuf + a_base + i * 2048 + c0);
                const u32x4 hi = *(GF_LDS u32x4*)(sbuf + a_base + i * 2048 + c1);
                const i32x8 af = i32x8{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(bfr[j], af, acc[i][j], 0, 0, 0,
                                                                                  0x7F7F7F7F, 0, 0x7F7F7F7F);
            }
        }
    }

    // ---- epilogue -----------------------------------------------------------------------------
    // acc[i][j][r] = C[m = m0 + wm*128 + i*16 + (lane&15)][n = n0 + wn*64 + j*16 + (lane>>4)*4 + r]
    __syncthreads();  // all waves finished reading the stage buffers
    GF_LDS char* ep = lds + wave * 16384;  // private 128 rows x 128 B (64 bf16) per wave
    {
        const int nb = n0 + wn * 64 + fq * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float bv[4] = {0.f, 0.f, 0.f, 0.f};
            const int n = nb + j * 16;
            if (p.bias && n < p.N) {  // N % 8 == 0 so n..n+3 are all valid
                const u16x4 b4 = *reinterpret_cast<const u16x4*>(p.bias + n);
#pragma unroll
                for (int r = 0; r < 4; ++r) bv[r] = bf2f(b4[r]);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float y[4];
                float rs = 1.0f;
                if constexpr (FP8) rs = p.row_scale[min(m0 + wm * 128 + i * 16 + frow, p.M - 1)];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    y[r] = gf_epi_act<EPI>(acc[i][j][r] * rs + bv[r]);  // (x scale_a for fp8)
                }
                u32x2 pk;
                pk[0] = pack2bf(y[0], y[1]);
                pk[1] = pack2bf(y[2], y[3]);
                const int row = i * 16 + frow;
                const int slot = (j * 4 + fq) ^ ((row & 7) << 1);  // 8-byte slot, pairs stay together
                *(GF_LDS u32x2*)(ep + row * 128 + slot * 8) = pk;
            }
        }
    }
    // each wave reads back only its own region: no workgroup barrier needed, only the LDS wait
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    {
        const int rr = lane >> 3, cc = lane & 7;
        const int n = n0 + wn * 64 + cc * 8;
        const bool n_ok = n < p.N;
        u16x8 g8;
        if (EPI == GF_EPI_BIAS_GATE_RESID && n_ok) g8 = *reinterpret_cast<const u16x8*>(p.gate + n);
#pragma unroll 4
        for (int it = 0; it < 16; ++it) {
            const int row = it * 8 + rr;
            const int m = m0 + wm * 128 + row;
            const u16x8 yv = *(GF_LDS u16x8*)(ep + row * 128 + ((cc ^ (row & 7)) << 4));
            if (m < p.M && n_ok) {
                u16x8 o = yv;
                if (EPI == GF_EPI_BIAS_GATE_RESID || EPI == GF_EPI_BIAS_RESID || EPI == GF_EPI_BIAS_MUL) {
                    const u16x8 r8 = *reinterpret_cast<const u16x8*>(p.R + (long)m * p.ldr + n);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        float t = bf2f(yv[e]);
                        if (EPI == GF_EPI_BIAS_GATE_RESID) t = rbf(bf2f(g8[e]) * t);  // gate * residual
                        o[e] = (EPI == GF_EPI_BIAS_MUL) ? f2bf(t * bf2f(r8[e]))       // fc1(x) * gelu(gate(x))
                                                        : f2bf(bf2f(r8[e]) + t);      // x + ...
                    }
                }
                *reinterpret_cast<u16x8*>(p.C + (long)m * p.ldc + n) = o;
            }
        }
    }
}

// ================================================================================================
// Phased kernel (the one that ships): same 256x256 tile / 8 waves / 128-byte-row LDS image, but the K loop is cut
// into 4 phases per K-tile and the operand tiles into four 16 KiB HALF-TILES per K-tile
//     A0 = rows   0..127, A1 = rows 128..255 of the A tile;   B0 / B1 likewise for the W tile,
// laid out per double-buffer as [A0 | A1 | B0 | B1] (64 KiB) x 2 = 128 KiB.
//
//   * wave (wr, wc) owns four C quadrants (a, b): 64 rows [wr*64, +64) of A-half a  x  32 columns [wc*32, +32) of
//     B-half b.  Phase p of a K-tile multiplies ONE quadrant over the whole K-step (16 MFMAs bf16 / 8 MFMAs fp8):
//        p0: read A-sub0 + B-sub0, quadrant (0,0)      p1: read B-sub1, quadrant (0,1)
//        p2: read A-sub1,          quadrant (1,1)      p3: no LDS read,  quadrant (1,0)   (B-sub0 kept in registers)
//   * half-tiles stream through LDS-DMA in the fixed order  A0,B0,B1,A1 | A0,B0,...  exactly one per phase, SIX phases
//     ahead of the phase that first reads it; `s_waitcnt vmcnt(8)` (4 half-tiles x 2 DMA instructions stay in flight)
//     retires what the NEXT phase reads, so HBM/L2 latency is covered by >= 5 phases of matrix work and the DMA
//     stays in flight across barriers (raw s_barrier, never __syncthreads).  A region is re-staged >= 2 phases after
//     its last ds_read.
//   * the two waves that share a SIMD (w and w+4 = wave rows 0 and 1) run offset by half a phase: each phase is
//     [L: issue DMA + ds_reads] barrier [M: MFMA cluster at raised priority] barrier, and wave row 1 starts one
//     barrier late — one partner feeds the matrix pipe while the other feeds LDS.
//   * epilogue: all waves park their bf16 quadrants in one swizzled 256x256 LDS image, then every wave writes full
//     512-byte rows (16 B per lane), reading residual / gate / bias in the same coalesced pattern.
// NB = number of 128-column halves of the tile: 2 = the 256 x 256 tile described above; 1 = a 256 x 128 tile (convolutions with
// Cout <= 128: the decoder's 96-channel level and its RGB head would otherwise multiply 160 / 248 padding columns) — two phases
// per K-tile, (A0, B0) and (A1, B0), half-tile stream A0 B0 A1 with three stage calls (6 DMA instructions) in flight.
// NB = 3: a 256 x 192 tile (Cout = 192, 384: the 256-wide tile multiplied 64 / 128 padding columns, a quarter of the MFMAs): the B1
// half-tile holds 64 weight rows (one DMA instruction per wave, so 7 instead of 8 stay in flight), wave (wr, wc) owns ONE 16-column
// block of it (columns 128 + 16 wc): quadrants (a, 1) are 64 x 16 — 48 MFMAs per wave and K tile instead of 64.
// CONV: 0 = dense A operand; 1 = the general gather (all modes, history from `cache`); 2 = stride-1 gather whose source frames
// are one contiguous run (mode 0 with the causal history in front of src, or no temporal taps): each A row keeps a 64-bit
// pointer to its pixel in its first source frame plus 4 border flags, and a piece's address is that pointer + one per-tap offset.
template <int EPI, bool FP8, int CONV = 0, int NB = 2>
__global__ __launch_bounds__(GEMM_THREADS, 2) void gemm_ph_kernel(const GemmArgs p) {
    static_assert(!(CONV && FP8), "the implicit-GEMM convolution is bf16 only");
    static_assert(NB == 2 || ((NB == 1 || NB == 3) && !FP8), "NB = 1 and 3 are built for bf16");
    constexpr bool WIDE = NB >= 2;                      // two B half-tiles per K tile
    constexpr int NBH = WIDE ? 2 : 1;                   // B half-tiles
    constexpr int J1 = NB == 3 ? 1 : 2;                 // 16-column blocks a wave owns in B1
    constexpr int BNT = NB == 3 ? 192 : NB * 128;       // tile width
    constexpr int ESZ = FP8 ? 1 : 2;
    constexpr int BKE = 128 / ESZ;
    constexpr int HALF_BYTES = 128 * 128;  // 16 KiB
    extern __shared__ __attribute__((aligned(16))) char smem[];
    GF_LDS char* lds = (GF_LDS char*)smem;

    // batched launches (gf_gemm_bf16_batched): problem blockIdx.y; strides 0 and gridDim.y = 1 otherwise
    const u16* const pA = p.A + (long)blockIdx.y * p.sA;
    const u16* const pW = p.W + (long)blockIdx.y * p.sW;
    u16* const pC = p.C + (long)blockIdx.y * p.sC;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;

    const int nwg = p.tiles_m * p.tiles_n;
    int v;
    {
        const int pid = blockIdx.x;
        const int xcd = pid & 7, local = pid >> 3;
        const int q = nwg >> 3, r = nwg & 7;
        v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
    }
    const int per_group = GROUP_M * p.tiles_n;
    const int group = v / per_group;
    const int first_m = group * GROUP_M;
    const int gsz = min(p.tiles_m - first_m, GROUP_M);
    const int in_group = v - group * per_group;
    const int m0 = (first_m + in_group % gsz) * BM, n0 = (in_group / gsz) * BNT;

    // ---- DMA sources: wave w fills row-groups 2w, 2w+1 (8 rows x 128 B each) of every half-tile -------------------
    const int srow = lane >> 3;
    const int schunk = (lane & 7) ^ srow;
    const char* srcp[4][2];  // [kind: A0, B0, B1, A1][i]
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (wave * 2 + i) * 8 + srow;  // 0..127 inside the half-tile
        const long a0 = min(m0 + row, p.M - 1), a1 = min(m0 + 128 + row, p.M - 1);
        const long b0 = min(n0 + row, p.N - 1), b1 = min(n0 + 128 + row, p.N - 1);
        srcp[0][i] = (const char*)pA + (a0 * p.lda) * ESZ + schunk * 16;
        srcp[3][i] = (const char*)pA + (a1 * p.lda) * ESZ + schunk * 16;
        srcp[1][i] = (const char*)pW + (b0 * p.ldw) * ESZ + schunk * 16;
        srcp[2][i] = (const char*)pW + (b1 * p.ldw) * ESZ + schunk * 16;
    }
    const char* srcp_b1h = (const char*)pW + (min((long)(n0 + 128 + wave * 8 + srow), (long)p.N - 1) * p.ldw) * ESZ + schunk * 16;   // NB = 3
    // CONV: this lane's four A rows as output pixels: (Y << 16 | X) and the first source frame of the causal window
    int cyx[2][2], cf[2][2];
    const u16* rowp[2][2];   // CONV == 2: the row's pixel in its first source frame
    int rmask[2][2];         // CONV == 2: 1 no row above, 2 none below, 4 no column to the left, 8 none to the right, 16 always
    if constexpr (CONV != 0) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int m = min(m0 + a * 128 + (wave * 2 + i) * 8 + srow, p.M - 1);
                const int X = m % p.cv.Wo, t = m / p.cv.Wo;
                const int Y = t % p.cv.Ho, j = t / p.cv.Ho;
                const int f0 = p.cv.t_off + j * p.cv.t_stride - (p.cv.kt - 1);
                if constexpr (CONV == 2) {
                    rowp[a][i] = p.cv.src + ((long)f0 * p.cv.frame + ((long)Y * p.cv.W + X) * p.cv.C);
                    rmask[a][i] = 16 | (Y == 0 ? 1 : 0) | (Y == p.cv.H - 1 ? 2 : 0) | (X == 0 ? 4 : 0) | (X == p.cv.W - 1 ? 8 : 0);
                } else {
                    cyx[a][i] = (Y << 16) | X;
                    cf[a][i] = f0;
                }
            }
    }
    // CONV == 2 with C % 64 == 0 (the 192- and 384-channel VAE levels): a K tile = 64 channels of ONE tap, so the tap of the tile an A
    // kind stages next is wave-uniform state advanced by scalar instructions (tiles are staged in order, once per kind); per lane and
    // piece only the border test and one 64-bit add are left.  The general form below derives the tap per lane from the K index
    // (two float multiplies, a 64-bit multiply-add, ...): ~45 vector instructions per staged half-tile, which made these convolutions
    // vector-issue-bound (48 MFMAs x 8 issue cycles + ~90 x 4 per K tile against 768 MFMA cycles).
    int s_c0[2] = {0, 0}, s_dt[2] = {0, 0}, s_dy[2] = {0, 0}, s_dx[2] = {0, 0};
    if constexpr (CONV == 2) {
        if (p.cv.c64) {
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int i = 0; i < 2; ++i) rowp[a][i] += schunk * 8;        // the lane's 8 channels inside the tile
        }
    }
    const int nk = p.K / BKE;
    // region byte offsets inside one double-buffer, indexed by kind
    auto region = [](int kind) { return kind == 0 ? 0 : (kind == 3 ? HALF_BYTES : (kind == 1 ? 2 * HALF_BYTES : 3 * HALF_BYTES)); };
    auto stage = [&](int tile, int kind) {  // issue the 2 DMA instructions of half-tile (tile, kind)
        GF_LDS char* dst = lds + (tile & 1) * STAGE_BYTES + region(kind) + wave * 2048;
        if constexpr (CONV != 0) {
            if (kind == 0 || kind == 3) {
                const ConvGeom& g = p.cv;
                const int a = kind == 3;
                if constexpr (CONV == 2) {
                    if (g.c64) {
                        const int hf = g.ks >> 1;
                        const long toff = (long)s_dt[a] * g.frame + (long)(((s_dy[a] - hf) * g.W + (s_dx[a] - hf)) * g.C + s_c0[a]);
                        int need = 0;
                        if (g.ks == 3) need = (s_dy[a] == 0 ? 1 : 0) | (s_dy[a] == 2 ? 2 : 0) | (s_dx[a] == 0 ? 4 : 0) | (s_dx[a] == 2 ? 8 : 0);
#pragma unroll
                        for (int i = 0; i < 2; ++i) glds16((need & rmask[a][i]) ? g.zero : rowp[a][i] + toff, dst + i * 1024);
                        s_c0[a] += 64;                                       // the next tile of this kind
                        if (s_c0[a] == g.C) {
                            s_c0[a] = 0;
                            if (++s_dx[a] == g.ks) {
                                s_dx[a] = 0;
                                if (++s_dy[a] == g.ks) {
                                    s_dy[a] = 0;
                                    ++s_dt[a];
                                }
                            }
                        }
                        return;
                    }
                }
                const int ke = tile * 64 + schunk * 8;                       // first of this lane's 8 channels along K
                const int tap = (int)(((float)ke + 0.5f) * g.inv_c);         // exact: ke < 2^16, C <= 512
                const int c = ke - tap * g.C;
                const int dt = (int)(((float)tap + 0.5f) * g.inv_ks2);
                const int rem = tap - dt * g.ks * g.ks;
                int dy = 0, dx = 0;
                if (g.ks == 3) {
                    dy = (rem * 11) >> 5;                                    // rem / 3 for rem < 9
                    dx = rem - 3 * dy;
                }
                const int half = g.ks >> 1;
                if constexpr (CONV == 2) {
                    // one offset per tap: dt frames, dy - half rows, dx - half columns, c channels; `need` = the border flags
                    // that make this tap a padding tap (bit 4: a K-padding column past the last tap)
                    const long toff = (long)dt * g.frame + (long)(((dy - half) * g.W + (dx - half)) * g.C + c);
                    int need = tap < g.taps ? 0 : 16;
                    if (g.ks == 3) need |= (dy == 0 ? 1 : 0) | (dy == 2 ? 2 : 0) | (dx == 0 ? 4 : 0) | (dx == 2 ? 8 : 0);
#pragma unroll
                    for (int i = 0; i < 2; ++i) glds16((need & rmask[a][i]) ? g.zero : rowp[a][i] + toff, dst + i * 1024);
                    return;
                }
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int Y = cyx[a][i] >> 16, X = cyx[a][i] & 0xffff;
                    int sy, sx;
                    bool ok;
                    if (g.mode == 2) {                                       // stride 2, zero pad on the bottom / right only
                        sy = 2 * Y + dy;
                        sx = 2 * X + dx;
                        ok = sy < g.H && sx < g.W;
                    } else {
                        const int yy = Y + dy - half, xx = X + dx - half;
                        ok = yy >= 0 && yy < g.Ho && xx >= 0 && xx < g.Wo;
                        sy = g.mode == 1 ? (yy >> 1) : yy;                   // conv on the nearest-2x upsampled image
                        sx = g.mode == 1 ? (xx >> 1) : xx;
                    }
                    const int f = cf[a][i] + dt;
                    const u16* base = (f >= 0 || g.hist_front) ? g.src + (long)f * g.frame : g.cache + (long)(2 + f) * g.frame;
                    const u16* src = (ok && tap < g.taps) ? base + ((long)sy * g.W + sx) * g.C + c : g.zero;
                    glds16(src, dst + i * 1024);
                }
                return;
            }
        }
        const long koff = (long)tile * 128;
        if constexpr (NB == 3) {
            if (kind == 2) {                               // 64 rows: one instruction per wave (rows 8 wave .. 8 wave + 7)
                glds16(srcp_b1h + koff, lds + (tile & 1) * STAGE_BYTES + region(2) + wave * 1024);
                return;
            }
        }
        glds16(srcp[kind][0] + koff, dst);
        glds16(srcp[kind][1] + koff, dst + 1024);
    };

    // ---- fragment read offsets -----------------------------------------------------------------------------------
    const int frow = lane & 15, fq = lane >> 4, sw = frow & 7;
    const int a_off = (wr * 64 + frow) * 128;   // + i*2048, i < 4
    const int b_off = (wc * 32 + frow) * 128;   // + j*2048, j < 2
    const int b_off1 = NB == 3 ? (wc * 16 + frow) * 128 : b_off;   // B1 of the 192-wide tile: one block per wave
    int chb[4];                                 // swizzled 16-byte chunk offsets of this lane
    if constexpr (!FP8) {
        chb[0] = ((0 + fq) ^ sw) << 4;          // k-substep 0
        chb[1] = ((4 + fq) ^ sw) << 4;          // k-substep 1
        chb[2] = chb[3] = 0;
    } else {
        chb[0] = ((2 * fq) ^ sw) << 4;          // bytes [32fq, 32fq+16)
        chb[1] = ((2 * fq + 1) ^ sw) << 4;      // bytes [32fq+16, 32fq+32)
        chb[2] = chb[3] = 0;
    }
    typedef u32x4 frag_t[2];                    // bf16: [ksub] 8 bf16 each; fp8: 32 bytes = one K=128 operand
    frag_t afr[4], bfr[2][2];                   // afr[i], bfr[b][j]
    auto read_a = [&](GF_LDS char* buf, int a) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            afr[i][0] = *(GF_LDS u32x4*)(buf + region(a ? 3 : 0) + a_off + i * 2048 + chb[0]);
            afr[i][1] = *(GF_LDS u32x4*)(buf + region(a ? 3 : 0) + a_off + i * 2048 + chb[1]);
        }
    };
    auto read_b = [&](GF_LDS char* buf, int b) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if (b && j >= J1) continue;
            bfr[b][j][0] = *(GF_LDS u32x4*)(buf + region(b ? 2 : 1) + (b ? b_off1 : b_off) + j * 2048 + chb[0]);
            bfr[b][j][1] = *(GF_LDS u32x4*)(buf + region(b ? 2 : 1) + (b ? b_off1 : b_off) + j * 2048 + chb[1]);
        }
    };

    f32x4 acc[2][NBH][4][2];  // [a][b][i][j]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < NBH; ++b)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[a][b][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // (Tried and measured: issuing the phase's two DMA pieces inside this burst instead of in the load segment — the
    // stamps of tools/gemm_stamps.py put the load segment at 527 cycles against a 340-cycle burst — ran 4 % SLOWER: the two
    // waves of a SIMD share one issue budget, moving work between them does not net.)
    auto mma = [&](int a, int b) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if (b && j >= J1) continue;
                if constexpr (!FP8) {
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks)
                        acc[a][b][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                            __builtin_bit_cast(bf16x8, bfr[b][j][ks]), __builtin_bit_cast(bf16x8, afr[i][ks]),
                            acc[a][b][i][j], 0, 0, 0);
                } else {
                    const i32x8 bw = i32x8{(int)bfr[b][j][0][0], (int)bfr[b][j][0][1], (int)bfr[b][j][0][2], (int)bfr[b][j][0][3],
                                           (int)bfr[b][j][1][0], (int)bfr[b][j][1][1], (int)bfr[b][j][1][2], (int)bfr[b][j][1][3]};
                    const i32x8 aw = i32x8{(int)afr[i][0][0], (int)afr[i][0][1], (int)afr[i][0][2], (int)afr[i][0][3],
                                           (int)afr[i][1][0], (int)afr[i][1][1], (int)afr[i][1][2], (int)afr[i][1][3]};
                    acc[a][b][i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(bw, aw, acc[a][b][i][j], 0, 0, 0,
                                                                                         0x7F7F7F7F, 0, 0x7F7F7F7F);
                }
            }
        __builtin_amdgcn_s_setprio(0);
    };

    // ---- prologue: half-tiles seq 0..5 = (t0: A0 B0 B1 A1) (t1: A0 B0) ------------------------------------------------
    const int total = (WIDE ? 4 : 3) * nk;  // half-tiles in the stream
    if constexpr (WIDE) {
        stage(0, 0);
        stage(0, 1);
        stage(0, 2);
        stage(0, 3);
        if (nk > 1) {
            stage(1, 0);
            stage(1, 1);
            if constexpr (NB == 3) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // seq 0,1 landed
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
    } else {   // stream A0 B0 A1 per tile: seq 3t, 3t+1, 3t+2
        stage(0, 0);
        stage(0, 1);
        stage(0, 3);
        if (nk > 1) {
            stage(1, 0);
            stage(1, 1);
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");  // seq 0,1 landed
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
    }
    if (wr == 1) __builtin_amdgcn_s_barrier();  // wave row 1 runs half a phase behind wave row 0

#define GF_PHASE_END(SEQ_ISSUED)                                                  \
    if ((SEQ_ISSUED) >= total) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  \
    else if constexpr (NB == 3) asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); \
    else if constexpr (NB == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); \
    else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");                        \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                           \
    __builtin_amdgcn_sched_barrier(0);                                           \
    __builtin_amdgcn_s_barrier();

#define GF_AFTER_MMA                       \
    __builtin_amdgcn_sched_barrier(0);     \
    __builtin_amdgcn_s_barrier();
    if constexpr (NB == 1) {
        for (int c = 0; c < nk; ++c) {
            GF_LDS char* buf = lds + (c & 1) * STAGE_BYTES;
            const int g = 3 * c;
            // ---- phase 0: stream A1(c+1); read A-sub0, B-sub0; half (0,0)
            if (g + 5 < total) stage(c + 1, 3);
            read_b(buf, 0);
            read_a(buf, 0);
            GF_PHASE_END(g + 5)
            mma(0, 0);
            GF_AFTER_MMA
            // ---- phase 1: stream A0(c+2), B0(c+2); read A-sub1 (B-sub0 still in registers); half (1,0)
            if (g + 6 < total) {
                stage(c + 2, 0);
                stage(c + 2, 1);
            }
            read_a(buf, 1);
            GF_PHASE_END(g + 7)
            mma(1, 0);
            GF_AFTER_MMA
        }
    } else
    for (int c = 0; c < nk; ++c) {
        GF_LDS char* buf = lds + (c & 1) * STAGE_BYTES;
        const int g = 4 * c;
        // ---- phase 0: stream B1(c+1); read A-sub0, B-sub0; quadrant (0,0)
        if (g + 6 < total) stage(c + 1, 2);
        read_b(buf, 0);
        read_a(buf, 0);
        GF_PHASE_END(g + 6)
        mma(0, 0);
        GF_AFTER_MMA
        // ---- phase 1: stream A1(c+1); read B-sub1; quadrant (0,1)
        if (g + 7 < total) stage(c + 1, 3);
        read_b(buf, 1);
        GF_PHASE_END(g + 7)
        mma(0, 1);
        GF_AFTER_MMA
        // ---- phase 2: stream A0(c+2); read A-sub1; quadrant (1,1)
        if (g + 8 < total) stage(c + 2, 0);
        read_a(buf, 1);
        GF_PHASE_END(g + 8)
        mma(1, 1);
        GF_AFTER_MMA
        // ---- phase 3: stream B0(c+2); no LDS read (B-sub0 still in registers); quadrant (1,0)
        if (g + 9 < total) stage(c + 2, 1);
        GF_PHASE_END(g + 9)
        mma(1, 0);
        GF_AFTER_MMA
    }
#undef GF_AFTER_MMA
#undef GF_PHASE_END
    if (wr == 0) __builtin_amdgcn_s_barrier();  // pairs with wave row 1's last barrier: everyone is past its LDS reads

    // The residual / multiplier rows of this lane's 16 store pieces are requested FIRST (64 registers, free once the accumulators are on
    // their way to the image): their HBM latency passes under the image writes and the barrier instead of once per group of four stores —
    // with one workgroup per CU nothing else covers an epilogue (VAE convolutions with the ResidualBlock's shortcut: 4.63 -> 4.4x ms).
    constexpr bool HAS_R = EPI == GF_EPI_BIAS_GATE_RESID || EPI == GF_EPI_BIAS_RESID || EPI == GF_EPI_BIAS_MUL;
    u16x8 rpre[HAS_R ? 16 : 1];
    if constexpr (HAS_R) {
        const int cc = lane & 31, n = n0 + cc * 8;
        const bool n_ok = n < p.N && cc * 8 < BNT;
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int m = m0 + wave * 32 + it * 2 + (lane >> 5);
            rpre[it] = u16x8{0, 0, 0, 0, 0, 0, 0, 0};
            if (m < p.M && n_ok) rpre[it] = *reinterpret_cast<const u16x8*>(p.R + (long)m * p.ldr + n);
        }
    }

    // ---- epilogue: bf16 quadrants -> swizzled 256 x 256 LDS image (512-byte rows) -> full-row stores ---------------------
    // acc[a][b][i][j][r] = C[m0 + a*128 + wr*64 + i*16 + frow][n0 + b*128 + wc*32 + j*16 + fq*4 + r]
#pragma unroll
    for (int b = 0; b < NBH; ++b)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if (b && j >= J1) continue;
            const int ncol = (NB == 3 && b) ? 128 + wc * 16 + fq * 4 : b * 128 + wc * 32 + j * 16 + fq * 4;  // column inside the tile
            float bv[4] = {0.f, 0.f, 0.f, 0.f};
            if (p.bias && n0 + ncol < p.N) {
                const u16x4 b4 = *reinterpret_cast<const u16x4*>(p.bias + n0 + ncol);
#pragma unroll
                for (int r = 0; r < 4; ++r) bv[r] = bf2f(b4[r]);
            }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int row = a * 128 + wr * 64 + i * 16 + frow;
                    float rs = 1.0f;
                    if constexpr (FP8) rs = p.row_scale[min(m0 + row, p.M - 1)];
                    float y[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        y[r] = gf_epi_act<EPI>(acc[a][b][i][j][r] * rs + bv[r]);
                    }
                    u32x2 pk;
                    pk[0] = pack2bf(y[0], y[1]);
                    pk[1] = pack2bf(y[2], y[3]);
                    const int slot = (ncol >> 2) ^ ((row & 15) << 1);  // 8-byte slot (64 per row); 16-byte pairs stay together
                    *(GF_LDS u32x2*)(lds + row * 512 + slot * 8) = pk;
                }
        }
    __syncthreads();
    {
        const int cc = lane & 31;                 // 16-byte chunk of the row
        const int n = n0 + cc * 8;
        const bool n_ok = n < p.N && cc * 8 < BNT;
        u16x8 g8;
        if (EPI == GF_EPI_BIAS_GATE_RESID && n_ok) g8 = *reinterpret_cast<const u16x8*>(p.gate + n);
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int row = wave * 32 + it * 2 + (lane >> 5);
            const int m = m0 + row;
            const u16x8 yv = *(GF_LDS u16x8*)(lds + row * 512 + ((cc ^ (row & 15)) << 4));
            if (m < p.M && n_ok) {
                u16x8 o = yv;
                if (EPI == GF_EPI_BIAS_GATE_RESID || EPI == GF_EPI_BIAS_RESID || EPI == GF_EPI_BIAS_MUL) {
                    const u16x8 r8 = rpre[HAS_R ? it : 0];
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        float t = bf2f(yv[e]);
                        if (EPI == GF_EPI_BIAS_GATE_RESID) t = rbf(bf2f(g8[e]) * t);  // gate * residual
                        o[e] = (EPI == GF_EPI_BIAS_MUL) ? f2bf(t * bf2f(r8[e]))       // fc1(x) * gelu(gate(x))
                                                        : f2bf(bf2f(r8[e]) + t);      // x + ...
                    }
                }
                *reinterpret_cast<u16x8*>(pC + (long)m * p.ldc + n) = o;
            }
        }
    }
}

// ================================================================================================================
// gemm_a4_kernel — the bf16 GEMM of the large DiT shapes (M >= 512): ONE wave per SIMD.
//   * 256 threads = 4 waves (wm, wn) in 2 x 2 on a 256 x 256 x 64 tile, 128 x 128 of C per wave = 8 x 8 tiles of
//     v_mfma_f32_16x16x32_bf16; the 256 fp32 accumulators live in AGPRs a[0:255], 128 fragment registers in v[128:255].
//     Per K tile a wave issues 128 MFMAs, 32 ds_read_b128 and 16 LDS-DMA pieces (the 8-wave phased kernel: 2 x 64 MFMAs,
//     2 x 24 reads, 2 x 8 pieces per SIMD) and crosses 3 barriers instead of 8.
//   * staging by `buffer_load_dwordx4 ... offen lds`: descriptor + SGPR row-group offset + one per-lane voffset that
//     advances by 128 B per K tile — no vector address arithmetic and no M0 save/restore per piece (the phased kernel
//     spends ~100 issue cycles per piece on global_load_lds with 64-bit per-lane addresses, profiles/r01/gemm_stamps.txt);
//     rows past M / N and tiles past K are cut off by the descriptor's num_records (reads return 0): no clamping.
//   * the K loop is ONE asm statement with hand-allocated registers and a fixed slot for every read, piece, counted wait
//     and barrier (tools/gen_gemm_a4.py -> gf_gemm_a4_loop.inc; hipcc spills at this pressure and re-orders the slots).
//     Register double buffering by k-sub-step: while the 64 MFMAs of sub-step 0 run, the fragments of sub-step 1 are
//     read; while those run, the sub-step-0 fragments of the NEXT tile (other LDS stage) are read; tile t+2 streams into
//     the stage tile t has just been read out of.
//   * same XOR-swizzled 128-byte-row LDS image, swapped operands and fused epilogues (bias / GELU-tanh / SiLU /
//     gate*+residual / +residual / *multiply, the reference's bf16 rounding sequence) as the other kernels; the epilogue
//     transposes each wave's 128 x 128 through a private 32 KiB LDS image and stores whole 256-byte row segments.
#include "gf_gemm_a4_loop.inc"
#include "gf_gemm_a4f8_loop.inc"   // the fp8 K loop (tools/gen_gemm_a4f8.py)
constexpr int A4_THREADS = 256;

template <int I>
__device__ __forceinline__ float a4_acc() {
    float x;
    asm volatile("v_accvgpr_read_b32 %0, a[%1]" : "=v"(x) : "n"(I));
    return x;
}
template <int I, int N, class F>
__device__ __forceinline__ void a4_static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        a4_static_for<I + 1, N>(f);
    }
}

// FP8 = true (gf_gemm_fp8, the reference's fp8_linear contract VRAM:115-151): the operands are OCP e4m3 bytes, a K tile is still
// 128 bytes per row (= 128 elements, one k step of v_mfma_f32_16x16x128_f8f6f4), the loop is gf_gemm_a4f8_loop.inc (two barriers
// per tile, staging by half tiles: tools/gen_gemm_a4f8.py) and the epilogue multiplies each row by its activation scale first.
// p.lda / p.ldw / p.K are in ELEMENTS of the operand type.
template <int EPI, bool FP8 = false>
__global__ __launch_bounds__(A4_THREADS, 1) void gemm_a4_kernel(const GemmArgs p) {
    constexpr unsigned ESZ = FP8 ? 1u : 2u;    // bytes per operand element
    extern __shared__ __attribute__((aligned(16))) char smem[];
    GF_LDS char* lds = (GF_LDS char*)smem;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    const int nwg = p.tiles_m * p.tiles_n;
    int v;
    {
        const int pid = blockIdx.x;
        const int xcd = pid & 7, local = pid >> 3;
        const int q = nwg >> 3, r = nwg & 7;
        v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
    }
    const int gm = p.group_m;                        // row tiles per group (launch_gemm_a4: 8, or 4 for long K loops)
    const int per_group = gm * p.tiles_n;
    const int group = v / per_group;
    const int first_m = group * gm;
    const int gsz = min(p.tiles_m - first_m, gm);
    const int in_group = v - group * per_group;
    const int m0 = (first_m + in_group % gsz) * BM, n0 = (in_group / gsz) * BN;

    // ---- staging: wave w fills 64 rows of both operand tiles as 8 pieces of 8 rows x 128 B (GF_A4_ROWMAP: piece p = rows
    // 32 p + 8 w .., the four waves interleaved; else rows 64 w + 8 p ..); lane l of a piece
    // fills LDS chunk (l & 7) of row (l >> 3) and must fetch logical chunk (l & 7) ^ (row & 7) of that row
    const int srow = lane >> 3;
    unsigned voffA = (unsigned)srow * (unsigned)p.lda * ESZ + (unsigned)(((lane & 7) ^ srow) << 4);
    unsigned voffB = (unsigned)srow * (unsigned)p.ldw * ESZ + (unsigned)(((lane & 7) ^ srow) << 4);
    // L2 warm-up loads (one line per lane): lane l = row l of this wave's 64 staging rows
    unsigned pfA = (unsigned)lane * (unsigned)p.lda * ESZ, pfB = (unsigned)lane * (unsigned)p.ldw * ESZ;
    const unsigned long baseA = (unsigned long)((const char*)p.A + (long)m0 * p.lda * ESZ);
    const unsigned long baseB = (unsigned long)((const char*)p.W + (long)n0 * p.ldw * ESZ);
    const unsigned aLo = (unsigned)baseA, aHi = (unsigned)(baseA >> 32) & 0xffffu;
    const unsigned bLo = (unsigned)baseB, bHi = (unsigned)(baseB >> 32) & 0xffffu;
    const unsigned nrA = (unsigned)(((long)(min(p.M - m0, BM) - 1) * p.lda + p.K) * ESZ);   // valid bytes from the tile's first row
    const int wvalid = min(p.wrows - n0, BN);                                                // rows of this W tile that exist
    const unsigned nrB = wvalid > 0 ? (unsigned)(((long)(wvalid - 1) * p.ldw + p.K) * ESZ) : 0u;
    const unsigned stA = 32u * (unsigned)p.lda * ESZ, stB = 32u * (unsigned)p.ldw * ESZ;
    const unsigned soA = (unsigned)wave * 8u * (unsigned)p.lda * ESZ, soB = (unsigned)wave * 8u * (unsigned)p.ldw * ESZ;
    const unsigned ldsW = (unsigned)(unsigned long)lds + (unsigned)wave * 1024u;
    const unsigned nk = (unsigned)(p.K * (int)ESZ / 128);       // K tiles of 128 bytes per row
    // Staggered K start: the workgroups of column tile j begin their K loop at K tile (2 j) mod nk and wrap around, so
    // that the ~32 column tiles in flight at any time fetch from different 256-byte blocks of their rows.  All rows of the
    // operands have the same pitch, so without this every workgroup of the chip walks the SAME few memory channels at the
    // same time (pitch 10 KiB: 16 of 128 channel slots).  The sum over k is only rotated; it depends on the column tile
    // alone, so an output element's bits do not depend on how the rows are cut into tiles or sharded over GPUs.
    const unsigned k0 = p.stagger ? (unsigned)((p.stagger * (p.stagger_rows ? m0 / BM : n0 / BN)) % (int)nk) : 0u;
    const unsigned kb = (unsigned)p.K * ESZ;

    // ---- fragment read addresses: (row, chunk) at row * 128 + ((chunk ^ (row & 7)) << 4); sub-step ks reads chunk 4 ks + fq
    const int frow = lane & 15, fq = lane >> 4, sw = frow & 7;
    const unsigned lbase = (unsigned)(unsigned long)lds;
    unsigned rdA0 = lbase + (unsigned)((wm * 128 + frow) * 128 + (((0 + fq) ^ sw) << 4));
    unsigned rdA1 = lbase + (unsigned)((wm * 128 + frow) * 128 + (((4 + fq) ^ sw) << 4));
    unsigned rdB0 = lbase + (unsigned)(TILE_BYTES + (wn * 128 + frow) * 128 + (((0 + fq) ^ sw) << 4));
    unsigned rdB1 = lbase + (unsigned)(TILE_BYTES + (wn * 128 + frow) * 128 + (((4 + fq) ^ sw) << 4));

    // bias / gate of this lane's columns: requested BEFORE the K loop (16 + 4 VGPRs below the loop's registers), so that the
    // epilogue does not open with an exposed L2 round trip
    u16x4 bpre[8];
    u16x8 gpre = {0, 0, 0, 0, 0, 0, 0, 0};
    {
        const int fq_ = lane >> 4;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int n = n0 + wn * 128 + j * 16 + fq_ * 4;
            bpre[j] = (EPI != GF_EPI_VT32 && p.bias && n < p.N) ? *reinterpret_cast<const u16x4*>(p.bias + n) : u16x4{0, 0, 0, 0};
        }
        const int ng = n0 + wn * 128 + (lane & 15) * 8;
        if (EPI == GF_EPI_BIAS_GATE_RESID && ng < p.N) gpre = *reinterpret_cast<const u16x8*>(p.gate + ng);
        if (EPI == GF_EPI_VT32 && p.bias) {   // the row biases of this lane's eight 16-row blocks travel in bpre[i][0]
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int mrow = m0 + wm * 128 + i * 16 + (lane & 15);
                if (mrow < p.M) bpre[i][0] = p.bias[mrow];
            }
        }
    }
    // fp8: the activation scales of this lane's eight rows (one per 16-row block), requested before the K loop as well
    float rsc[FP8 ? 8 : 1];
    if constexpr (FP8 && EPI != GF_EPI_VT32) {
#pragma unroll
        for (int i = 0; i < 8; ++i) rsc[i] = p.row_scale[min(m0 + wm * 128 + i * 16 + (lane & 15), p.M - 1)];
    }
    if constexpr (FP8) {
        GF_A4F8_LOOP_ASM(voffA, voffB, rdA0, rdA1, rdB0, rdB1, aLo, aHi, nrA, bLo, bHi, nrB, soA, stA, soB, stB, ldsW, nk, k0, kb);
    } else
    {
        GF_A4_LOOP_ASM(voffA, voffB, pfA, pfB, rdA0, rdA1, rdB0, rdB1, aLo, aHi, nrA, bLo, bHi, nrB, soA, stA, soB, stB, ldsW, nk, k0, kb);
    }

    // ---- epilogue: a[(i*8+j)*4 + r] = C[m0 + wm*128 + 16 i + frow][n0 + wn*128 + 16 j + 4 fq + r]  (the asm ended on a barrier:
    // every wave is past its LDS reads and every LDS-DMA has landed)
    GF_LDS char* ep = lds + wave * 32768;   // private 128 rows x 256 B; 8-byte slot s of row r at slot s ^ ((r & 15) << 1)
    // The residual / multiplier operand of the epilogue is requested FIRST, all 32 row pieces of this lane (128 VGPRs: the
    // fragment registers are free now): its HBM latency passes under the accumulator conversion below.  Loading it inside the
    // store loop, four pieces in flight, cost the three residual GEMMs of a block 0.12-0.17 ms each (D->D: 1.36 -> 1.2x ms).
    constexpr bool HAS_R = EPI == GF_EPI_BIAS_GATE_RESID || EPI == GF_EPI_BIAS_RESID || EPI == GF_EPI_BIAS_MUL;
    u16x8 rres[HAS_R ? 32 : 1];
    if constexpr (HAS_R) {
        const int nr = n0 + wn * 128 + (lane & 15) * 8;
#pragma unroll
        for (int it = 0; it < 32; ++it) {
            const int mr = m0 + wm * 128 + it * 4 + (lane >> 4);
            if (mr < p.M && nr < p.N) {
                rres[it] = __builtin_nontemporal_load(reinterpret_cast<const u16x8*>(p.R + (long)mr * p.ldr + nr));
            } else {
                rres[it] = u16x8{0, 0, 0, 0, 0, 0, 0, 0};
            }
        }
    }
    // fp8 V^T projection: the tokens' activation scales of this lane's 8 x 4 columns (zeros past the last token)
    f32x4 csc[(FP8 && EPI == GF_EPI_VT32) ? 8 : 1];
    if constexpr (FP8 && EPI == GF_EPI_VT32) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int n = n0 + wn * 128 + j * 16 + fq * 4;
#pragma unroll
            for (int r = 0; r < 4; ++r) csc[j][r] = (n + r < p.wrows) ? p.row_scale[n + r] : 0.f;
        }
    }
    a4_static_for<0, 8>([&](auto j_c) {
        constexpr int j = decltype(j_c)::value;
        const int n = n0 + wn * 128 + j * 16 + fq * 4;
        const float bv[4] = {bf2f(bpre[j][0]), bf2f(bpre[j][1]), bf2f(bpre[j][2]), bf2f(bpre[j][3])};   // zeros where there is no bias
        a4_static_for<0, 8>([&](auto i_c) {
            constexpr int i = decltype(i_c)::value;
            constexpr int A0 = (i * 8 + j) * 4;
            float y[4] = {a4_acc<A0>(), a4_acc<A0 + 1>(), a4_acc<A0 + 2>(), a4_acc<A0 + 3>()};
            if constexpr (FP8 && EPI != GF_EPI_VT32) {   // x scale_a of the row, + bias: the same expression as the 8-wave fp8 kernel (gemm_kernel<EPI, true>)
#pragma unroll
                for (int r = 0; r < 4; ++r) y[r] = y[r] * rsc[i] + bv[r];
            }
            if constexpr (EPI == GF_EPI_VT32) {   // bias of the ROW (output feature); key columns that do not exist are zero
                const float bm = bf2f(bpre[i][0]);       // requested before the K loop (0 where there is no bias / no row)
                if constexpr (FP8) {   // operands swapped: the activation's scale_a belongs to the COLUMN (token); same x scale + bias expression
#pragma unroll
                    for (int r = 0; r < 4; ++r) y[r] = (n + r < p.wrows) ? y[r] * csc[j][r] + bm : 0.f;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) y[r] = (n + r < p.wrows) ? y[r] + bm : 0.f;
                }
            } else if constexpr (EPI == GF_EPI_BIAS_GELU_TANH) {
                // the Linear's output rounded to bf16 (one v_cvt_pk per pair, unpacked by a shift and a mask), then GELU on the lane's
                // four values together — the same operations as gf_epi_act<GELU>, about half the issue slots (FFN1's epilogue cost it 6 %)
                const unsigned l01 = FP8 ? pack2bf(y[0], y[1]) : pack2bf(y[0] + bv[0], y[1] + bv[1]);
                const unsigned l23 = FP8 ? pack2bf(y[2], y[3]) : pack2bf(y[2] + bv[2], y[3] + bv[3]);
                gf_f32x2 g01 = {__uint_as_float(l01 << 16), __uint_as_float(l01 & 0xffff0000u)};
                gf_f32x2 g23 = {__uint_as_float(l23 << 16), __uint_as_float(l23 & 0xffff0000u)};
                gelu_tanh_f4(g01, g23);
                y[0] = g01[0];
                y[1] = g01[1];
                y[2] = g23[0];
                y[3] = g23[1];
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    y[r] = gf_epi_act<EPI>(FP8 ? y[r] : y[r] + bv[r]);
                }
            }
            u32x2 pk;
            pk[0] = pack2bf(y[0], y[1]);
            pk[1] = pack2bf(y[2], y[3]);
            const int row = i * 16 + frow;
            const int slot = (j * 4 + fq) ^ ((row & 15) << 1);
            *(GF_LDS u32x2*)(ep + row * 256 + slot * 8) = pk;
        });
    });
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // wave-private image: LDS is in order, no barrier needed
    {
        const int cc = lane & 15;                 // 16-byte chunk of the 256-byte row segment
        const int n = n0 + wn * 128 + cc * 8;
        const bool n_ok = n < p.N;
        const u16x8 g8 = gpre;
#pragma unroll
        for (int it = 0; it < 32; ++it) {
            const int row = it * 4 + (lane >> 4);
            const int m = m0 + wm * 128 + row;
            u16x8 yv;
            if constexpr (EPI == GF_EPI_VT32) {
                // positions 8 cc .. 8 cc + 7 of the row = keys 4 g .. 4 g + 3 and 16 + 4 g .. of 32-key group G (cc = 4 G + g): the k
                // order of the 16x16x32 B operand that kernel 3 builds from two score tiles (gf_transpose_v32)
                const int G = cc >> 2, g = cc & 3, sw = (row & 15) << 1;
                const u16x4 lo = *(GF_LDS u16x4*)(ep + row * 256 + (((8 * G + g) ^ sw) << 3));
                const u16x4 hi = *(GF_LDS u16x4*)(ep + row * 256 + (((8 * G + 4 + g) ^ sw) << 3));
                yv = u16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            } else {
                yv = *(GF_LDS u16x8*)(ep + row * 256 + ((cc ^ (row & 15)) << 4));
            }
            if (m < p.M && n_ok) {
                u16x8 o = yv;
                if constexpr (HAS_R) {
                    const u16x8 r8 = rres[it];
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        float t = bf2f(yv[e]);
                        if (EPI == GF_EPI_BIAS_GATE_RESID) t = rbf(bf2f(g8[e]) * t);  // gate * residual
                        o[e] = (EPI == GF_EPI_BIAS_MUL) ? f2bf(t * bf2f(r8[e]))       // fc1(x) * gelu(gate(x))
                                                        : f2bf(bf2f(r8[e]) + t);      // x + ...
                    }
                }
                __builtin_nontemporal_store(o, reinterpret_cast<u16x8*>(p.C + (long)m * p.ldc + n));
            }
        }
    }
}

template <int EPI, bool FP8 = false>
int launch_gemm_a4(const GemmArgs& a0, hipStream_t stream) {
    GemmArgs a = a0;
    {
        // an XCD runs 32 consecutive tiles of the order = group_m row tiles x 32 / group_m column tiles: 8 x 4 and 4 x 8 both stage
        // 12 operand slices per K step; with the long K loop of F->D (K = 13824) 4 x 8 measured +1.6 %, at K = 5120 8 x 4 +1-2 %
        const int g = gf_options().a4_group_m.load(std::memory_order_relaxed);
        a.group_m = g > 0 ? g : (a.K * (FP8 ? 1 : 2) >= 16384 ? 4 : GROUP_M);   // by the K loop's length in bytes per row
    }
    static GfDeviceOnce once;
    hipError_t e = gf_once_per_device(once, [] {
        return hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_a4_kernel<EPI, FP8>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS);
    });
    if (e != hipSuccess) {
        gf_set_error("gf_gemm: hipFuncSetAttribute(%d B LDS) failed: %s", GEMM_LDS, hipGetErrorString(e));
        return GF_ERR_LAUNCH;
    }
    hipLaunchKernelGGL((gemm_a4_kernel<EPI, FP8>), dim3((unsigned)(a.tiles_m * a.tiles_n)), dim3(A4_THREADS), GEMM_LDS, stream, a);
    GF_CHECK_LAUNCH(FP8 ? "gf_gemm_fp8" : "gf_gemm_bf16");
    return GF_OK;
}

template <int EPI, bool FP8>
int launch_gemm(const GemmArgs& a, hipStream_t stream) {
    // the 4-wave kernel for the large shapes (every Linear of a DiT block at S >= 512); `prefer_8wave` (gf_set_option, tests: the
    // cross-kernel bit-identity checks) sends them to the 8-wave kernel below.  Its 32-bit staging offsets cover 256 rows of either operand.
    const bool use_a4 = gf_options().prefer_8wave.load(std::memory_order_relaxed) == 0;
    if constexpr (!FP8) {
        if (use_a4 && a.M >= 512 && a.K % 64 == 0 && 256L * a.lda * 2 + a.K * 2L < (1L << 31) &&
            256L * a.ldw * 2 + a.K * 2L < (1L << 31))
            return launch_gemm_a4<EPI>(a, stream);
    } else {
        if (use_a4 && a.M >= 512 && a.K % 128 == 0 && 256L * a.lda + a.K < (1L << 31) && 256L * a.ldw + a.K < (1L << 31))
            return launch_gemm_a4<EPI, true>(a, stream);
    }
    // small M: bf16 runs the phased 8-wave kernel, fp8 the one-barrier-per-K-tile 8-wave kernel (at 254 VGPRs the phased fp8
    // variant measured 10-15 % slower)
    static GfDeviceOnce once;   // per instantiation
    hipError_t e = gf_once_per_device(once, [] {
        const void* fn;
        if constexpr (FP8) fn = reinterpret_cast<const void*>(gemm_kernel<EPI, true>);
        else fn = reinterpret_cast<const void*>(gemm_ph_kernel<EPI, false>);
        return hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS);
    });
    if (e != hipSuccess) {
        gf_set_error("gf_gemm: hipFuncSetAttribute(%d B LDS) failed: %s", GEMM_LDS, hipGetErrorString(e));
        return GF_ERR_LAUNCH;
    }
    if constexpr (FP8)
        hipLaunchKernelGGL((gemm_kernel<EPI, true>), dim3((unsigned)(a.tiles_m * a.tiles_n)), dim3(GEMM_THREADS), GEMM_LDS, stream, a);
    else
        hipLaunchKernelGGL((gemm_ph_kernel<EPI, false>), dim3((unsigned)(a.tiles_m * a.tiles_n)), dim3(GEMM_THREADS), GEMM_LDS,
                           stream, a);
    GF_CHECK_LAUNCH(FP8 ? "gf_gemm_fp8" : "gf_gemm_bf16");
    return GF_OK;
}

template <int EPI, int NB, int CONV>
int launch_conv(const GemmArgs& a, hipStream_t stream, int batch = 1) {
    static GfDeviceOnce once;
    hipError_t e = gf_once_per_device(once, [] {
        return hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_ph_kernel<EPI, false, CONV, NB>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS);
    });
    if (e != hipSuccess) {
        gf_set_error("gf_conv3d: hipFuncSetAttribute(%d B LDS) failed: %s", GEMM_LDS, hipGetErrorString(e));
        return GF_ERR_LAUNCH;
    }
    hipLaunchKernelGGL((gemm_ph_kernel<EPI, false, CONV, NB>), dim3((unsigned)(a.tiles_m * a.tiles_n), (unsigned)batch), dim3(GEMM_THREADS),
                       GEMM_LDS, stream, a);
    GF_CHECK_LAUNCH("gf_conv3d_bf16");
    return GF_OK;
}

// 256 zero bytes per device: where the convolution's padding taps and K-padding columns are fetched from.  A __device__
// array of the code object (zero-initialised when the module is loaded on a device): looking up its address neither
// allocates nor synchronises, so gf_conv3d_bf16 keeps the header's contract and works under stream capture.
}  // namespace
// external linkage + default visibility: hipGetSymbolAddress looks the variable up by name in the code object (a variable of
// the anonymous namespace is a local symbol there: "Cannot create GlobalVar Obj")
__device__ __attribute__((used, visibility("default"))) u16 gf_conv_zero_page[128] = {};   // not `const`: that would be internal linkage
namespace {

const u16* conv_zero_page() {
    static GfDeviceOnce once;
    static const u16* page[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    hipError_t e = gf_once_per_device(once, [dev] {
        void* z = nullptr;
        hipError_t r = hipGetSymbolAddress(&z, HIP_SYMBOL(gf_conv_zero_page));
        if (r == hipSuccess) page[dev] = (const u16*)z;
        return r;
    });
    return e == hipSuccess ? page[dev] : nullptr;
}

}  // namespace

static int gemm_dispatch(bool fp8, const void* A, int64_t lda, const void* W, int64_t ldw, const float* row_scale,
                         const void* bias, void* C, int64_t ldc, int64_t M, int64_t N, int64_t K, int epilogue,
                         const void* resid, int64_t ldr, const void* gate, void* stream) {
    const char* fn = fp8 ? "gf_gemm_fp8" : "gf_gemm_bf16";
    const int bke = fp8 ? 128 : 64, lal = fp8 ? 16 : 8;
    GF_CHECK_ARG(A && W && C, "%s: null A/W/C", fn);
    GF_CHECK_ARG(M >= 0 && N > 0 && K > 0, "%s: bad sizes M=%ld N=%ld K=%ld", fn, (long)M, (long)N, (long)K);
    GF_CHECK_ARG(K % bke == 0, "%s: K=%ld must be a multiple of %d", fn, (long)K, bke);
    GF_CHECK_ARG(N % 8 == 0, "%s: N=%ld must be a multiple of 8", fn, (long)N);
    GF_CHECK_ARG(lda % lal == 0 && ldw % lal == 0 && ldc % 8 == 0 && lda >= K && ldw >= K && ldc >= N,
                 "%s: leading dimensions must be multiples of %d (A/W) / 8 (C) and cover the row", fn, lal);
    GF_CHECK_ARG(gf_aligned16(A) && gf_aligned16(W) && gf_aligned16(C) && (!bias || gf_aligned16(bias)),
                 "%s: 16-byte alignment required", fn);
    GF_CHECK_ARG(M < (1 << 30) && N < (1 << 30), "%s: M/N too large", fn);
    GF_CHECK_ARG(!fp8 || row_scale, "%s: row_scale is required", fn);
    const bool need_r = epilogue == GF_EPI_BIAS_GATE_RESID || epilogue == GF_EPI_BIAS_RESID || epilogue == GF_EPI_BIAS_MUL;
    GF_CHECK_ARG(!need_r || (resid && ldr % 8 == 0 && ldr >= N && gf_aligned16(resid)),
                 "%s: residual epilogue needs an aligned resid with ldr >= N", fn);
    GF_CHECK_ARG(epilogue != GF_EPI_BIAS_GATE_RESID || (gate && gf_aligned16(gate)),
                 "%s: gate epilogue needs an aligned gate vector", fn);
    if (M == 0) return GF_OK;
    GemmArgs a;
    a.A = (const u16*)A;
    a.W = (const u16*)W;
    a.bias = (const u16*)bias;
    a.C = (u16*)C;
    a.R = (const u16*)resid;
    a.gate = (const u16*)gate;
    a.row_scale = row_scale;
    a.M = (int)M;
    a.N = (int)N;
    a.K = (int)K;
    a.lda = lda;
    a.ldw = ldw;
    a.ldc = ldc;
    a.ldr = ldr;
    a.tiles_m = (int)((M + BM - 1) / BM);
    a.tiles_n = (int)((N + BN - 1) / BN);
    // K tiles between the K-loop starts of neighbouring column tiles (0 = off).  The rotation changes the ORDER of the sum over k
    // (not the sum): a4 outputs match the unrotated kernels (M < 512: the 8-wave path) to fp32 rounding, not bit for bit.
    a.stagger = gf_options().a4_stagger.load(std::memory_order_relaxed);
    a.wrows = (int)N;
    a.stagger_rows = 0;
    hipStream_t s = (hipStream_t)stream;
#define GF_GEMM_CASE(E) case E: return fp8 ? launch_gemm<E, true>(a, s) : launch_gemm<E, false>(a, s);
    switch (epilogue) {
        GF_GEMM_CASE(GF_EPI_BIAS)
        GF_GEMM_CASE(GF_EPI_BIAS_GELU_TANH)
        GF_GEMM_CASE(GF_EPI_BIAS_GATE_RESID)
        GF_GEMM_CASE(GF_EPI_BIAS_RESID)
        GF_GEMM_CASE(GF_EPI_BIAS_SILU)
        GF_GEMM_CASE(GF_EPI_BIAS_MUL)
        default:
            gf_set_error("%s: unknown epilogue %d", fn, epilogue);
            return GF_ERR_INVALID_ARG;
    }
#undef GF_GEMM_CASE
}

extern "C" GF_API int gf_gemm_bf16(const void* A, int64_t lda, const void* W, int64_t ldw, const void* bias, void* C,
                                   int64_t ldc, int64_t M, int64_t N, int64_t K, int epilogue, const void* resid,
                                   int64_t ldr, const void* gate, void* stream) {
    return gemm_dispatch(false, A, lda, W, ldw, nullptr, bias, C, ldc, M, N, K, epilogue, resid, ldr, gate, stream);
}

extern "C" GF_API int gf_gemm_fp8(const void* A8, int64_t lda, const void* W8, int64_t ldw, const float* row_scale,
                                  const void* bias, void* C, int64_t ldc, int64_t M, int64_t N, int64_t K, int epilogue,
                                  const void* resid, int64_t ldr, const void* gate, void* stream) {
    return gemm_dispatch(true, A8, lda, W8, ldw, row_scale, bias, C, ldc, M, N, K, epilogue, resid, ldr, gate, stream);
}

// The V projection of a self-attention written directly as kernel 3's operand: vt[n][pos(s)] = bf16(sum_k x[s,k] w[n,k] + bias[n]),
// [N rows][kv_pad] with the keys of every 32-group in the order gf_transpose_v32 produces and zeros from kv_len to kv_pad.
// It is the SAME 4-wave GEMM with the operands swapped (A := w, W := x: C = w x^T is V^T row-major), the bias taken per row, the
// K-loop rotation taken from the row tile (so every element sums its products in the order gf_gemm_bf16(x, w) uses) and the
// key permutation applied where the epilogue reads its LDS image: bit-identical to gf_gemm_bf16 + gf_transpose_v32, one
// kernel and one 2 x 335 MB pass over V less per attention.
extern "C" GF_API int gf_linear_vt32(const void* x, int64_t ldx, const void* w, int64_t ldw, const void* bias, void* vt,
                                     int64_t kv_len, int64_t kv_pad, int64_t N, int64_t K, void* stream) {
    GF_CHECK_ARG(x && w && vt, "gf_linear_vt32: null x/w/vt");
    GF_CHECK_ARG(kv_len > 0 && kv_pad >= kv_len && kv_pad % 64 == 0 && kv_pad - kv_len < 64, "gf_linear_vt32: kv_pad = kv_len rounded up to 64");
    GF_CHECK_ARG(N >= 512 && N % 128 == 0 && K > 0 && K % 64 == 0, "gf_linear_vt32: N=%ld (>= 512, multiple of 128), K=%ld (multiple of 64)", (long)N, (long)K);
    GF_CHECK_ARG(ldx % 8 == 0 && ldw % 8 == 0 && ldx >= K && ldw >= K, "gf_linear_vt32: leading dimensions must be multiples of 8 covering K");
    GF_CHECK_ARG(gf_aligned16(x) && gf_aligned16(w) && gf_aligned16(vt) && (!bias || ((uintptr_t)bias & 1u) == 0), "gf_linear_vt32: alignment");
    GF_CHECK_ARG(N * kv_pad < (1LL << 31) && 256L * ldx * 2 + K * 2L < (1L << 31) && 256L * ldw * 2 + K * 2L < (1L << 31),
                 "gf_linear_vt32: operand too large for 32-bit staging offsets");
    GemmArgs a;
    a.A = (const u16*)w;
    a.W = (const u16*)x;
    a.bias = (const u16*)bias;
    a.C = (u16*)vt;
    a.R = nullptr;
    a.gate = nullptr;
    a.row_scale = nullptr;
    a.M = (int)N;
    a.N = (int)kv_pad;
    a.K = (int)K;
    a.lda = ldw;
    a.ldw = ldx;
    a.ldc = kv_pad;
    a.ldr = 0;
    a.tiles_m = (int)((N + BM - 1) / BM);
    a.tiles_n = (int)((kv_pad + BN - 1) / BN);
    a.stagger = gf_options().a4_stagger.load(std::memory_order_relaxed);
    a.wrows = (int)kv_len;
    a.stagger_rows = 1;
    return launch_gemm_a4<GF_EPI_VT32>(a, (hipStream_t)stream);
}

// gf_linear_vt32 on the fp8_linear contract (BASELINE config 5): x8 / w8 are e4m3 bytes, x_scale the tokens' activation scales
// (gf_quant_fp8_rowscale / gf_layernorm_modulate_fp8); vt[n][pos(s)] = bf16(sum_k x8[s,k] w8[n,k] * x_scale[s] + bias[n]) —
// bit-identical to gf_gemm_fp8(x8, w8, x_scale, bias) followed by gf_transpose_v32 (same kernel, operands swapped, K rotation by
// row tile).
extern "C" GF_API int gf_linear_vt32_fp8(const void* x8, int64_t ldx, const float* x_scale, const void* w8, int64_t ldw, const void* bias,
                                         void* vt, int64_t kv_len, int64_t kv_pad, int64_t N, int64_t K, void* stream) {
    GF_CHECK_ARG(x8 && w8 && vt && x_scale, "gf_linear_vt32_fp8: null x8/w8/vt/x_scale");
    GF_CHECK_ARG(kv_len > 0 && kv_pad >= kv_len && kv_pad % 64 == 0 && kv_pad - kv_len < 64, "gf_linear_vt32_fp8: kv_pad = kv_len rounded up to 64");
    GF_CHECK_ARG(N >= 512 && N % 128 == 0 && K > 0 && K % 128 == 0, "gf_linear_vt32_fp8: N=%ld (>= 512, multiple of 128), K=%ld (multiple of 128)", (long)N, (long)K);
    GF_CHECK_ARG(ldx % 16 == 0 && ldw % 16 == 0 && ldx >= K && ldw >= K, "gf_linear_vt32_fp8: leading dimensions must be multiples of 16 covering K");
    GF_CHECK_ARG(gf_aligned16(x8) && gf_aligned16(w8) && gf_aligned16(vt) && (!bias || ((uintptr_t)bias & 1u) == 0), "gf_linear_vt32_fp8: alignment");
    GF_CHECK_ARG(N * kv_pad < (1LL << 31) && 256L * ldx + K < (1L << 31) && 256L * ldw + K < (1L << 31),
                 "gf_linear_vt32_fp8: operand too large for 32-bit staging offsets");
    GemmArgs a;
    a.A = (const u16*)w8;
    a.W = (const u16*)x8;
    a.bias = (const u16*)bias;
    a.C = (u16*)vt;
    a.R = nullptr;
    a.gate = nullptr;
    a.row_scale = x_scale;
    a.M = (int)N;
    a.N = (int)kv_pad;
    a.K = (int)K;
    a.lda = ldw;
    a.ldw = ldx;
    a.ldc = kv_pad;
    a.ldr = 0;
    a.tiles_m = (int)((N + BM - 1) / BM);
    a.tiles_n = (int)((kv_pad + BN - 1) / BN);
    a.stagger = gf_options().a4_stagger.load(std::memory_order_relaxed);
    a.wrows = (int)kv_len;
    a.stagger_rows = 1;
    return launch_gemm_a4<GF_EPI_VT32, true>(a, (hipStream_t)stream);
}

// `batch` independent products C_b = A_b W_b^T (no bias, no epilogue) in ONE launch: problem b reads A + b strideA, W + b strideW and
// writes C + b strideC (element strides; a stride may be smaller than a matrix: heads side by side in one [L, H d] tensor have
// strideA = d, lda = H d).  For the small per-head / per-frame products of the umT5 and VAE attention (q k^T, p v: M, N of a few
// hundred rows), which as separate launches fill a sixtieth of the chip each.  The 8-wave kernel (256 x 256 tile, 256 x 128 for
// N <= 128); every element sums its K products in K order, like gf_gemm_bf16 on this kernel: bit-identical to `batch` separate calls
// there (the 4-wave kernel that gf_gemm_bf16 takes for M >= 512 starts its K loop at a rotated tile per column tile: another order).
extern "C" GF_API int gf_gemm_bf16_batched(const void* A, int64_t lda, int64_t strideA, const void* W, int64_t ldw, int64_t strideW,
                                           void* C, int64_t ldc, int64_t strideC, int64_t M, int64_t N, int64_t K, int64_t batch,
                                           void* stream) {
    GF_CHECK_ARG(A && W && C, "gf_gemm_bf16_batched: null A/W/C");
    GF_CHECK_ARG(M > 0 && N > 0 && K > 0 && batch > 0 && batch < 65536, "gf_gemm_bf16_batched: bad sizes");
    GF_CHECK_ARG(K % 64 == 0 && N % 8 == 0, "gf_gemm_bf16_batched: K=%ld must be a multiple of 64 and N=%ld of 8", (long)K, (long)N);
    GF_CHECK_ARG(lda >= K && ldw >= K && ldc >= N && lda % 8 == 0 && ldw % 8 == 0 && ldc % 8 == 0 && strideA % 8 == 0 && strideW % 8 == 0 &&
                     strideC % 8 == 0 && strideA >= 0 && strideW >= 0 && strideC >= 0,
                 "gf_gemm_bf16_batched: leading dimensions / strides must be multiples of 8 elements covering K / N");
    GF_CHECK_ARG(gf_aligned16(A) && gf_aligned16(W) && gf_aligned16(C), "gf_gemm_bf16_batched: 16-byte alignment required");
    GF_CHECK_ARG(M < (1L << 30) && N < (1L << 30), "gf_gemm_bf16_batched: M / N too large");
    GemmArgs a;
    a.A = (const u16*)A;
    a.W = (const u16*)W;
    a.bias = nullptr;
    a.C = (u16*)C;
    a.R = nullptr;
    a.gate = nullptr;
    a.row_scale = nullptr;
    a.M = (int)M;
    a.N = (int)N;
    a.K = (int)K;
    a.lda = lda;
    a.ldw = ldw;
    a.ldc = ldc;
    a.ldr = 0;
    a.sA = strideA;
    a.sW = strideW;
    a.sC = strideC;
    const bool narrow = N <= 128;
    a.tiles_m = (int)((M + BM - 1) / BM);
    a.tiles_n = narrow ? (int)((N + 127) / 128) : (int)((N + BN - 1) / BN);
    a.stagger = 0;
    a.wrows = (int)N;
    a.stagger_rows = 0;
    a.group_m = 0;
    a.cv = ConvGeom{};
    return narrow ? launch_conv<GF_EPI_BIAS, 1, 0>(a, (hipStream_t)stream, (int)batch) : launch_conv<GF_EPI_BIAS, 2, 0>(a, (hipStream_t)stream, (int)batch);
}

// Causal 3-D / 2-D convolution of the Wan VAE as ONE implicit GEMM (no patch matrix in HBM): out[(j, Y, X), n] =
// bias[n] + sum over (dt, dy, dx, c) of in[t_off + j*t_stride - (kt-1) + dt, Y', X', c] * Wm[n, ((dt*ks + dy)*ks + dx)*C + c]
// with the gather rules of gf_vae_im2col (mode 0 / 1 / 2), frames < 0 taken from `cache` (its 2 frames are frames -2, -1).
// Wm is [N, ldw] bf16 with zero columns from taps*C up to K (K a multiple of 64).  Results are bit-identical to
// gf_vae_im2col + gf_gemm_bf16 (same K order, same kernel).
extern "C" GF_API int gf_conv3d_bf16(const void* src, const void* cache, const void* Wm, int64_t ldw, const void* bias,
                                     void* out, int64_t ldc, int64_t T_in, int64_t T_out, int64_t H, int64_t W, int64_t C,
                                     int kt, int ks, int mode, int t_stride, int t_off, int64_t N, int64_t K, int epilogue,
                                     const void* resid, int64_t ldr, void* stream) {
    GF_CHECK_ARG(src && Wm && out, "gf_conv3d_bf16: null src/W/out");
    GF_CHECK_ARG(T_in > 0 && T_out >= 0 && H > 0 && W > 0 && H < 32768 && W < 32768, "gf_conv3d_bf16: bad frame size");
    GF_CHECK_ARG(C > 0 && C % 8 == 0 && C <= 512, "gf_conv3d_bf16: C=%ld must be a multiple of 8 and <= 512", (long)C);
    GF_CHECK_ARG((kt == 1 || kt == 3) && (ks == 1 || ks == 3), "gf_conv3d_bf16: kt/ks must be 1 or 3");
    GF_CHECK_ARG(mode >= 0 && mode <= 2 && (mode == 0 || ks == 3), "gf_conv3d_bf16: bad mode");
    GF_CHECK_ARG(mode != 2 || (H % 2 == 0 && W % 2 == 0), "gf_conv3d_bf16: stride-2 mode needs even H, W");
    // kt == 3 with cache == NULL: the two history frames lie directly in front of src (src points 2 frames into one buffer)
    GF_CHECK_ARG(t_stride >= 1 && t_off >= 0 && (T_out == 0 || t_off + (T_out - 1) * t_stride < T_in),
                 "gf_conv3d_bf16: output frames reach past the input");
    const long taps = (long)kt * ks * ks;
    GF_CHECK_ARG(K % 64 == 0 && K >= taps * C && K < 65536 && ldw >= K && ldw % 8 == 0,
                 "gf_conv3d_bf16: K=%ld must be a multiple of 64 covering %ld taps x %ld channels", (long)K, taps, (long)C);
    GF_CHECK_ARG(N > 0 && N % 8 == 0 && ldc >= N && ldc % 8 == 0, "gf_conv3d_bf16: N / ldc must be multiples of 8");
    GF_CHECK_ARG(gf_aligned16(src) && gf_aligned16(Wm) && gf_aligned16(out) && (!cache || gf_aligned16(cache)) &&
                     (!bias || gf_aligned16(bias)), "gf_conv3d_bf16: 16-byte alignment required");
    GF_CHECK_ARG(epilogue == GF_EPI_BIAS || epilogue == GF_EPI_BIAS_RESID, "gf_conv3d_bf16: epilogue must be BIAS or BIAS_RESID");
    GF_CHECK_ARG(epilogue != GF_EPI_BIAS_RESID || (resid && ldr % 8 == 0 && ldr >= N && gf_aligned16(resid)),
                 "gf_conv3d_bf16: residual epilogue needs an aligned resid with ldr >= N");
    const long Ho = mode == 1 ? 2 * H : (mode == 2 ? H / 2 : H), Wo = mode == 1 ? 2 * W : (mode == 2 ? W / 2 : W);
    const long M = T_out * Ho * Wo;
    GF_CHECK_ARG(M < (1L << 30), "gf_conv3d_bf16: too many output pixels");
    if (M == 0) return GF_OK;
    GemmArgs a;
    a.A = nullptr;
    a.W = (const u16*)Wm;
    a.bias = (const u16*)bias;
    a.C = (u16*)out;
    a.R = (const u16*)resid;
    a.gate = nullptr;
    a.row_scale = nullptr;
    a.M = (int)M;
    a.N = (int)N;
    a.K = (int)K;
    a.lda = 0;
    a.ldw = ldw;
    a.ldc = ldc;
    a.ldr = ldr;
    // Cout <= 128 (the 96-channel level, the RGB head): the 256 x 128 tile; GF_CONV_NB=2 forces the 256 x 256 tile (A/B timing)
    const int force_nb = gf_options().conv_nb.load(std::memory_order_relaxed);
    const bool narrow = force_nb ? force_nb == 1 : N <= 128;
    // 192-wide tiles where they cover N with fewer padded columns than 256-wide ones (Cout = 192, 384: none instead of a quarter)
    const bool w192 = !narrow && !force_nb && ((N + 191) / 192) * 192 < ((N + BN - 1) / BN) * BN;
    a.tiles_m = (int)((M + BM - 1) / BM);
    a.tiles_n = narrow ? (int)((N + 127) / 128) : (w192 ? (int)((N + 191) / 192) : (int)((N + BN - 1) / BN));
    a.stagger = 0;
    a.wrows = (int)N;
    a.stagger_rows = 0;
    a.cv.src = (const u16*)src;
    a.cv.cache = (const u16*)cache;
    a.cv.zero = conv_zero_page();
    if (!a.cv.zero) {
        gf_set_error("gf_conv3d_bf16: could not allocate the zero page");
        return GF_ERR_LAUNCH;
    }
    a.cv.H = (int)H;
    a.cv.W = (int)W;
    a.cv.C = (int)C;
    a.cv.Ho = (int)Ho;
    a.cv.Wo = (int)Wo;
    a.cv.kt = kt;
    a.cv.ks = ks;
    a.cv.mode = mode;
    a.cv.t_stride = t_stride;
    a.cv.t_off = t_off;
    a.cv.taps = (int)taps;
    a.cv.frame = (long)H * W * C;
    a.cv.inv_c = 1.0f / (float)C;
    a.cv.inv_ks2 = 1.0f / (float)(ks * ks);
    hipStream_t s = (hipStream_t)stream;
    a.cv.hist_front = (kt == 3 && !cache) ? 1 : 0;
    a.cv.c64 = (C % 64 == 0 && K == taps * C && (ks == 1 || ks == 3)) ? 1 : 0;
    // the 96-channel full-resolution level (ResidualBlock convolutions with their history in front of src): direct convolution,
    // bit-identical to the implicit GEMM below (gf_conv_direct.hip; GF_CONV_DIRECT=0 switches it off)
    if (kt == 3 && ks == 3 && mode == 0 && !cache && t_stride == 1 && C == 96 && (N == 96 || N <= 16) && ldc == N &&
        (epilogue == GF_EPI_BIAS || ldr == N) && gf_options().conv_direct.load(std::memory_order_relaxed)) {
        const u16* walk = (const u16*)src + (long)(t_off - 2) * H * W * C;
        const int rc = gf_conv3d_direct_c96(walk, Wm, ldw, bias, out, T_out, H, W, N, epilogue, resid, a.cv.zero, stream);
        if (rc != GF_ERR_UNSUPPORTED) return rc;
    }
    // the decoder's full-resolution upsample convolution (nearest 2x + 3x3, 192 -> 96): direct form, bit-identical as well
    if (kt == 1 && ks == 3 && mode == 1 && t_stride == 1 && C == 192 && N == 96 && ldc == N && epilogue == GF_EPI_BIAS && (H % 4) == 0 &&
        (W % 16) == 0 && gf_options().conv_direct.load(std::memory_order_relaxed)) {
        const u16* src0 = (const u16*)src + (long)t_off * H * W * C;
        const int rc = gf_conv2d_up_direct_c192(src0, Wm, ldw, bias, out, T_out, H, W, a.cv.zero, stream);
        if (rc != GF_ERR_UNSUPPORTED) return rc;
    }
    // contiguous source frames and stride-1 taps: the pointer-per-row gather (GF_CONV_GATHER=1 forces the general one, A/B)
    const int force_g = gf_options().conv_gather.load(std::memory_order_relaxed);
    const bool fast = force_g != 1 && mode == 0 && (kt == 1 || !cache);
#define GF_CONV_CASE(E, NBV, CV) return launch_conv<E, NBV, CV>(a, s)
    if (fast) {
        if (narrow) { if (epilogue == GF_EPI_BIAS) GF_CONV_CASE(GF_EPI_BIAS, 1, 2); else GF_CONV_CASE(GF_EPI_BIAS_RESID, 1, 2); }
        if (w192) { if (epilogue == GF_EPI_BIAS) GF_CONV_CASE(GF_EPI_BIAS, 3, 2); else GF_CONV_CASE(GF_EPI_BIAS_RESID, 3, 2); }
        if (epilogue == GF_EPI_BIAS) GF_CONV_CASE(GF_EPI_BIAS, 2, 2); else GF_CONV_CASE(GF_EPI_BIAS_RESID, 2, 2);
    }
    if (narrow) { if (epilogue == GF_EPI_BIAS) GF_CONV_CASE(GF_EPI_BIAS, 1, 1); else GF_CONV_CASE(GF_EPI_BIAS_RESID, 1, 1); }
    if (w192) { if (epilogue == GF_EPI_BIAS) GF_CONV_CASE(GF_EPI_BIAS, 3, 1); else GF_CONV_CASE(GF_EPI_BIAS_RESID, 3, 1); }
    if (epilogue == GF_EPI_BIAS) GF_CONV_CASE(GF_EPI_BIAS, 2, 1); else GF_CONV_CASE(GF_EPI_BIAS_RESID, 2, 1);
#undef GF_CONV_CASE
}
