// gf_conv_a4.hip — the 3x3x3 causal convolutions of the Wan VAE's 192- and 384-channel levels (decoder: the middle blocks and the
// first three groups of residual blocks, VAE:736-838; encoder: the last three levels, VAE:517-617; CausalConv3d VAE:33-52) as a
// DIRECT convolution on the 4-wave GEMM's K loop.
//
// Before: gf_conv3d_bf16 ran these as an implicit GEMM on the 8-wave kernel (gemm_ph_kernel<CONV>): every 16-byte piece of the A tile
// is gathered with per-lane address arithmetic, 256 x 192 tile, 0.90 PFLOP/s at the 192-channel level (4.5 ms per 4.02 TFLOP).
// Here the activation is kept in a ZERO-BORDERED buffer  xp[2 + T][H + 2][W + 2][C]  (two history frames in front = the causal padding,
// one pixel of zeros around every frame = the spatial padding; written in that layout by gf_vae_rmsnorm_silu_padded) and the GEMM's row
// index is the PADDED position m = (t (H + 2) + y) (W + 2) + x.  Tap (dt, dy, dx) of row m is buffer row m + (dt (H + 2) + dy) (W + 2) + dx:
// the A tile of a K tile is 256 consecutive buffer rows x 64 channels — a plain strided matrix.  No gather, no border test, no patch
// matrix; one wave per SIMD with 128 x 96 of the 256 x 192 C tile in AGPRs; the K loop is one generated asm statement
// (tools/gen_conv_a4.py -> gf_conv_a4_loop.inc).  Rows of the padded border (2.6 % at 120 x 208) are computed and dropped.
//
// Arithmetic: every output element sums its products in the order of the implicit GEMM (K = (dt, dy, dx, cin), 32 channels per
// v_mfma_f32_16x16x32_bf16, one accumulation chain from tile 0), bias added in fp32, ONE rounding to bf16, the residual added to the
// rounded value and rounded again — bit-identical to gf_conv3d_bf16 (tests/test_vae.py).
#include "gf_common.h"
#include "gf_conv_a4_loop.inc"
#include <type_traits>

namespace {

constexpr int CA_BM = 256, CA_BN = 192, CA_THREADS = 256;
constexpr int CA_TILE_BYTES = CA_BM * 128;        // A tile: 256 rows x 128 B; the W tile (192 rows) sits behind it
constexpr int CA_LDS = 131072;                    // 2 stages x 64 KiB

struct ConvA4Args {
    const u16* xp;      // padded activation, row 0 = padded position (frame 0 of the history, y = 0, x = 0): [(2 + T)(H + 2)(W + 2), C]
    const u16* w;       // [N, ldw], K order (dt, dy, dx, cin)
    const u16* bias;    // [N] or null
    u16* out;           // [T H W, ldo]
    const u16* resid;   // [T H W, ldr] (GF_EPI_BIAS_RESID)
    int T, H, W, C, N;
    int kt;             // temporal taps: 3 (causal 3x3x3, two history frames in front of xp) or 1 (3x3 per frame, no history frames)
    long ldw, ldo, ldr;
    long xp_rows;       // rows of xp = (kt - 1 + T)(H + 2)(W + 2)
    int tiles_m, tiles_n;
};

template <int I>
__device__ __forceinline__ float ca_acc() {
    float x;
    asm volatile("v_accvgpr_read_b32 %0, a[%1]" : "=v"(x) : "n"(I));
    return x;
}
template <int I, int N, class F>
__device__ __forceinline__ void ca_static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        ca_static_for<I + 1, N>(f);
    }
}

template <int EPI>
__global__ __launch_bounds__(CA_THREADS, 1) void conv_a4_kernel(const ConvA4Args p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    GF_LDS char* lds = (GF_LDS char*)smem;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int Hp = p.H + 2, Wp = p.W + 2;

    // workgroup -> tile: an XCD (blockIdx % 8) walks a CONTIGUOUS range of row tiles (neighbouring row tiles read overlapping buffer
    // rows — a tap shifts the window by at most two pixel rows — so an XCD's L2 serves most of the nine spatial taps' re-reads), the
    // column tiles of a row tile back to back
    const int nwg = p.tiles_m * p.tiles_n;
    int v;
    {
        const int pid = blockIdx.x;
        const int xcd = pid & 7, local = pid >> 3;
        const int q = nwg >> 3, r = nwg & 7;
        v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
    }
    const int m0 = (v / p.tiles_n) * CA_BM, n0 = (v % p.tiles_n) * CA_BN;

    // ---- staging: piece q of wave w = rows 32 q + 8 w .. + 7 of the tile; lane l fills LDS chunk (l & 7) of row (l >> 3) and
    // fetches logical chunk (l & 7) ^ (row & 7) of that row (the GEMM's XOR-swizzled 128-byte-row image)
    const int srow = lane >> 3;
    const unsigned rowA = (unsigned)p.C * 2u, rowB = (unsigned)p.ldw * 2u;        // bytes per buffer row / weight row
    unsigned voffA = (unsigned)srow * rowA + (unsigned)(((lane & 7) ^ srow) << 4);
    unsigned voffB = (unsigned)srow * rowB + (unsigned)(((lane & 7) ^ srow) << 4);
    const unsigned long baseA = (unsigned long)((const char*)p.xp + (long)m0 * rowA);
    const unsigned long baseB = (unsigned long)((const char*)p.w + (long)n0 * rowB);
    const unsigned aLo = (unsigned)baseA, aHi = (unsigned)(baseA >> 32) & 0xffffu;
    const unsigned bLo = (unsigned)baseB, bHi = (unsigned)(baseB >> 32) & 0xffffu;
    // rows past the end of the buffer (the last row tile's taps) read as zeros: num_records = the bytes that exist behind the tile
    const long left = (p.xp_rows - m0) * (long)rowA;
    const unsigned nrA = left > 0 ? (unsigned)(left < 0xffffffffL ? left : 0xffffffffL) : 0u;
    const int wvalid = min(p.N - n0, CA_BN);
    const unsigned nrB = wvalid > 0 ? (unsigned)(((long)(wvalid - 1) * p.ldw + 9L * p.kt * p.C) * 2) : 0u;
    const unsigned stA = 32u * rowA, stB = 32u * rowB;
    const unsigned soA = (unsigned)wave * 8u * rowA, soB = (unsigned)wave * 8u * rowB;
    const unsigned ldsW = (unsigned)(unsigned long)lds + (unsigned)wave * 1024u;
    const unsigned nk = 9u * (unsigned)p.kt * (unsigned)p.C / 64u;
    const unsigned run = 3u * (unsigned)p.C / 64u;                                  // K tiles of a pixel row's three taps
    const unsigned jr = (unsigned)Wp * rowA - 3u * rowA + 128u;                    // last tile of a run -> first tile of the next pixel row
    const unsigned jf = (unsigned)Hp * (unsigned)Wp * rowA - 2u * (unsigned)Wp * rowA - 3u * rowA + 128u;   // ... -> first row of the next frame

    // ---- fragment read addresses: (row, chunk) at row * 128 + ((chunk ^ (row & 7)) << 4); sub-step ks reads chunk 4 ks + fq
    const int frow = lane & 15, fq = lane >> 4, sw = frow & 7;
    const unsigned lbase = (unsigned)(unsigned long)lds;
    unsigned rdA0 = lbase + (unsigned)((wm * 128 + frow) * 128 + (((0 + fq) ^ sw) << 4));
    unsigned rdA1 = lbase + (unsigned)((wm * 128 + frow) * 128 + (((4 + fq) ^ sw) << 4));
    unsigned rdB0 = lbase + (unsigned)(CA_TILE_BYTES + (wn * 96 + frow) * 128 + (((0 + fq) ^ sw) << 4));
    unsigned rdB1 = lbase + (unsigned)(CA_TILE_BYTES + (wn * 96 + frow) * 128 + (((4 + fq) ^ sw) << 4));

    // bias of this lane's columns, requested before the K loop
    u16x4 bpre[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const int n = n0 + wn * 96 + j * 16 + fq * 4;
        bpre[j] = (p.bias && n < p.N) ? *reinterpret_cast<const u16x4*>(p.bias + n) : u16x4{0, 0, 0, 0};
    }

    GF_CONV_A4_LOOP_ASM(voffA, voffB, rdA0, rdA1, rdB0, rdB1, aLo, aHi, nrA, bLo, bHi, nrB, soA, stA, soB, stB, ldsW, nk, run, jr, jf);

    // ---- epilogue: a[(i*8+j)*4 + r] = C[m0 + wm*128 + 16 i + frow][n0 + wn*96 + 16 j + 4 fq + r]  (the asm ended on a barrier)
    GF_LDS char* ep = lds + wave * 32768;   // private 128 rows x 256 B (192 used); 8-byte slot s of row r at slot s ^ ((r & 15) << 1)
    // output rows of this lane's 8 store passes (a lane stores three 16-byte chunks = 48 B of one row: 4 lanes per row, 16 rows per
    // pass) and, with a residual, its 24 pieces — requested FIRST (96 VGPRs: the fragment registers are free now), so that their HBM
    // latency passes under the accumulator conversion below (inside the store loop they cost 0.39 of 3.3 ms at the 192-channel level)
    const int rl = lane >> 2, c3 = (lane & 3) * 3;
    const int ncol = n0 + wn * 96 + c3 * 8;
    int orow[8];            // T H W < 2^31 (checked by the launcher)
    {
        const int frame = Hp * Wp;
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int m = m0 + wm * 128 + it * 16 + rl;        // padded position
            const int t = m / frame, rem = m - t * frame;
            const int y = rem / Wp, x = rem - y * Wp;
            orow[it] = (t < p.T && y < p.H && x < p.W) ? (t * p.H + y) * p.W + x : -1;
        }
    }
    constexpr bool HAS_R = EPI == GF_EPI_BIAS_RESID;
    u16x8 rres[HAS_R ? 24 : 1];
    if constexpr (HAS_R) {
#pragma unroll
        for (int it = 0; it < 8; ++it)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const int nn = ncol + c * 8;
                rres[it * 3 + c] = (orow[it] >= 0 && nn < p.N)
                                       ? __builtin_nontemporal_load(reinterpret_cast<const u16x8*>(p.resid + (long)orow[it] * p.ldr + nn))
                                       : u16x8{0, 0, 0, 0, 0, 0, 0, 0};
            }
    }
    ca_static_for<0, 6>([&](auto j_c) {
        constexpr int j = decltype(j_c)::value;
        const float bv[4] = {bf2f(bpre[j][0]), bf2f(bpre[j][1]), bf2f(bpre[j][2]), bf2f(bpre[j][3])};
        ca_static_for<0, 8>([&](auto i_c) {
            constexpr int i = decltype(i_c)::value;
            constexpr int A0 = (i * 8 + j) * 4;
            u32x2 pk;
            pk[0] = pack2bf(ca_acc<A0>() + bv[0], ca_acc<A0 + 1>() + bv[1]);
            pk[1] = pack2bf(ca_acc<A0 + 2>() + bv[2], ca_acc<A0 + 3>() + bv[3]);
            const int row = i * 16 + frow;
            const int slot = (j * 4 + fq) ^ ((row & 15) << 1);
            *(GF_LDS u32x2*)(ep + row * 256 + slot * 8) = pk;
        });
    });
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // wave-private image: LDS is in order, no barrier needed
    {
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int row = it * 16 + rl;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const int cc = c3 + c;
                const u16x8 yv = *(GF_LDS u16x8*)(ep + row * 256 + ((cc ^ (row & 15)) << 4));
                const int nn = ncol + c * 8;
                if (orow[it] >= 0 && nn < p.N) {
                    u16x8 o = yv;
                    if constexpr (HAS_R) {
                        const u16x8 r8 = rres[it * 3 + c];
#pragma unroll
                        for (int e = 0; e < 8; ++e) o[e] = f2bf(bf2f(r8[e]) + bf2f(yv[e]));
                    }
                    // plain stores: a row's two 192-byte halves come from two waves, and the 128-byte line they share is merged in L2
                    // (non-temporal stores sent it to the fabric twice: 2.2 GB written per 0.78 GB of output)
                    *reinterpret_cast<u16x8*>(p.out + (long)orow[it] * p.ldo + nn) = o;
                }
            }
        }
    }
}

template <int EPI>
int launch_conv_a4(const ConvA4Args& a, hipStream_t stream) {
    static GfDeviceOnce once;
    hipError_t e = gf_once_per_device(once, [] {
        return hipFuncSetAttribute(reinterpret_cast<const void*>(conv_a4_kernel<EPI>), hipFuncAttributeMaxDynamicSharedMemorySize, CA_LDS);
    });
    if (e != hipSuccess) {
        gf_set_error("gf_conv3d_padded_bf16: hipFuncSetAttribute(%d B LDS) failed: %s", CA_LDS, hipGetErrorString(e));
        return GF_ERR_LAUNCH;
    }
    hipLaunchKernelGGL((conv_a4_kernel<EPI>), dim3((unsigned)(a.tiles_m * a.tiles_n)), dim3(CA_THREADS), CA_LDS, stream, a);
    GF_CHECK_LAUNCH("gf_conv3d_padded_bf16");
    return GF_OK;
}

}  // namespace

// See include/goalforce.h.  xp = the zero-bordered activation [kt - 1 + T, H + 2, W + 2, C] (kt = 3: frames 0, 1 = the causal history).
extern "C" GF_API int gf_conv3d_padded_bf16(const void* xp, const void* Wm, int64_t ldw, const void* bias, void* out, int64_t ldo,
                                            int64_t T, int64_t H, int64_t W, int64_t C, int64_t N, int64_t kt, int epilogue,
                                            const void* resid, int64_t ldr, void* stream) {
    GF_CHECK_ARG(xp && Wm && out && T > 0 && H > 0 && W > 0 && (kt == 1 || kt == 3), "gf_conv3d_padded_bf16: bad arguments (kt = 1 or 3)");
    if (!(C == 192 || C == 384) || N % 8 != 0 || N <= 0) {
        gf_set_error("gf_conv3d_padded_bf16: C=%ld / N=%ld outside the kernel's shapes (C = 192 or 384, N a multiple of 8)", (long)C, (long)N);
        return GF_ERR_UNSUPPORTED;
    }
    GF_CHECK_ARG(epilogue == GF_EPI_BIAS || (epilogue == GF_EPI_BIAS_RESID && resid && ldr >= N && ldr % 8 == 0),
                 "gf_conv3d_padded_bf16: epilogue must be BIAS or BIAS_RESID (with a residual of row stride >= N)");
    GF_CHECK_ARG(ldw >= 9 * kt * C && ldw % 8 == 0 && ldo >= N && ldo % 8 == 0, "gf_conv3d_padded_bf16: bad leading dimensions");
    GF_CHECK_ARG(gf_aligned16(xp) && gf_aligned16(Wm) && gf_aligned16(out) && (!resid || gf_aligned16(resid)) && (!bias || (((uintptr_t)bias) & 7u) == 0),
                 "gf_conv3d_padded_bf16: 16-byte alignment required (bias: 8)");
    const int64_t Hp = H + 2, Wp = W + 2;
    const int64_t rows = (T + kt - 1) * Hp * Wp;
    GF_CHECK_ARG(rows * C * 2 < (1LL << 32) && N * ldw * 2 < (1LL << 32) && T * Hp * Wp < (1LL << 31) && T * H * W < (1LL << 31),
                 "gf_conv3d_padded_bf16: the padded activation and the weights must each stay below 4 GiB");
    ConvA4Args a;
    a.xp = (const u16*)xp;
    a.w = (const u16*)Wm;
    a.bias = (const u16*)bias;
    a.out = (u16*)out;
    a.resid = (const u16*)resid;
    a.T = (int)T;
    a.H = (int)H;
    a.W = (int)W;
    a.C = (int)C;
    a.N = (int)N;
    a.kt = (int)kt;
    a.ldw = ldw;
    a.ldo = ldo;
    a.ldr = ldr;
    a.xp_rows = rows;
    a.tiles_m = (int)((T * Hp * Wp + CA_BM - 1) / CA_BM);
    a.tiles_n = (int)((N + CA_BN - 1) / CA_BN);
    hipStream_t s = (hipStream_t)stream;
    return epilogue == GF_EPI_BIAS_RESID ? launch_conv_a4<GF_EPI_BIAS_RESID>(a, s) : launch_conv_a4<GF_EPI_BIAS>(a, s);
}
