// gf_conv_direct.hip — direct 3x3x3 causal convolution for the 96-channel full-resolution level of the Wan VAE (decoder
// ResidualBlocks and the encoder's first level: CausalConv3d VAE:33-52 inside ResidualBlock VAE:267-301), the layers that sat
// furthest below their roofline: as an implicit GEMM (gf_conv3d_bf16 -> gemm_ph_kernel<CONV = 2>) every input pixel is fetched 27
// times through L2 (once per tap), a 128-wide N tile is three quarters full at Cout = 96, and the 192-byte pixel pitch straddles
// 128-byte lines: 7.6 ms for 3.97 TFLOP (0.52 PFLOP/s) per convolution of a 240 x 416 x 80 tile.
//
// Here a workgroup (8 waves, two per SIMD) owns an 8 x 32 patch of output pixels x all 96 output channels and WALKS THE FRAMES:
//   * the (8+2) x (32+2) halo of ONE input frame is staged into LDS once (LDS-DMA, 224-byte pixel pitch = 192 + 32: the
//     16-pixel x 32-channel MFMA fragments are conflict-free ds_read_b128s, and a tap (dy, dx) is a constant address offset);
//   * an input frame g contributes to THREE output frames (g - dt, temporal tap dt = 0, 1, 2), so three accumulator sets are live
//     and the frame's 27 taps all run against the one staged halo: the input is fetched once, not 27 (nor 3) times.  The sets are
//     named by temporal tap and move up one place per frame by register moves, so the frame loop is ONE code path (three
//     instantiations selected by g mod 3 made hipcc spill accumulators);
//   * wave w = (pixel group w >> 1: two rows = four 16-pixel blocks) x (channel half w & 1: three 16-channel blocks): 3 sets x 12
//     tiles x 4 = 144 accumulator registers, so two waves fit a SIMD and cover each other's LDS latencies and barrier waits (the
//     first version, 4 waves x 3 blocks x 6 channel blocks at one wave per SIMD, ran at 0.71 PFLOP/s; hipcc spills at the 288
//     accumulators a 4-wave 8 x 32 patch would need);
//   * the weights of one tap (96 x 96 bf16, rows at the same 224-byte pitch) stream through a 3-stage LDS ring, two taps ahead
//     (an LDS-DMA piece needs ~1.1 us to land under load: one tap ahead left every tap waiting), one barrier per tap; per tap and wave 36 v_mfma_f32_16x16x32_bf16 (operands swapped as in the GEMM: a lane ends with 4
//     consecutive output channels of one pixel) on 12 A + 9 B fragment reads;
//   * when an output frame is complete (after its third input frame) its accumulators go through a wave-private LDS image and
//     leave as 96-byte half pixels, 16 bytes per lane (bias, optional residual with the GEMM epilogue's rounding sequence); that
//     epilogue runs while the next input frame's halo is in flight.
// Every output element sums its products in the order of the implicit GEMM (taps (dt, dy, dx) in sequence, 32 channels per MFMA):
// the results are BIT-IDENTICAL to gemm_ph_kernel<CONV> (tests/test_vae.py).  Frames are cut into segments so that a launch has a
// few workgroups per CU; a segment's first two input frames only feed the later temporal taps.
#include "gf_common.h"
#include <type_traits>

namespace {

constexpr int CD_C = 96;
constexpr int CD_TH = 8, CD_TW = 32, CD_HH = CD_TH + 2, CD_HW = CD_TW + 2;
constexpr int CD_WAVES = 8, CD_THREADS = 64 * CD_WAVES;
constexpr int CD_PITCH = 224, CD_SLOTS = 14;                                          // 16-byte slots per pixel / weight row: 12 data + 2 pad
constexpr int CD_HALO_INSTR = (CD_HH * CD_HW * CD_SLOTS + 63) / 64;                  // 75 wave-instructions of 1 KiB
constexpr int CD_HALO_BYTES = CD_HALO_INSTR * 1024;                                  // 76800
constexpr int CD_STAGES = 3;                                                         // weight ring: the tap running, the next one, the one after
constexpr int CD_EP_BYTES = 16 * 96;                                                 // per wave and pass: 16 pixels x its 48 channels (in the free ring stage)
constexpr int CD_HK = (CD_HALO_INSTR + CD_WAVES - 1) / CD_WAVES;                     // halo DMA instructions per wave: 10
// NCB = 16-channel blocks of output: 6 (Cout = 96: wave = pixel group w >> 1 x channel half w & 1) or 1 (Cout <= 16, the decoder's RGB
// head: wave = pixel row w, two 16-pixel blocks; weight rows past Cout come from the zero page)
template <int NCB> struct CdShape {
    static constexpr int ROWS = NCB * 16;                                            // weight rows staged per tap
    static constexpr int W_INSTR = (ROWS * CD_SLOTS + 63) / 64;                      // 21 / 4 wave-instructions per tap
    static constexpr int W_BYTES = W_INSTR * 1024;                                   // 21504 / 4096
    static constexpr int WK = (W_INSTR + CD_WAVES - 1) / CD_WAVES;                   // per wave: 3 / 1
    static constexpr int PB = NCB == 6 ? 4 : 2, CB = NCB == 6 ? 3 : 1;               // per wave: pixel blocks x channel blocks
    static constexpr int LDS = CD_HALO_BYTES + CD_STAGES * W_BYTES;                  // 141312 / 89088
};

struct CdArgs {
    const u16* src;     // input frame 0 of the walk = the frame output 0 reads with dt = 0 (history included): [*, H, W, 96]
    const u16* w;       // [96, ldw], K order (dt, dy, dx, cin)
    const u16* bias;    // [96] or null
    u16* out;           // [T_out, H, W, 96]
    const u16* resid;   // [T_out, H, W, 96] (EPI_BIAS_RESID)
    const u16* zero;    // >= 16 zero bytes
    int H, W, T_out, tseg, tiles_x, tiles_y;
    int N;              // output channels (96, or <= 16 for the NCB = 1 kernel; out / resid rows are N wide)
    long frame;         // H * W * 96
    long ldw;
};

// One asm statement: counted wait for this wave's own LDS-DMA requests + its LDS reads, then the workgroup barrier.  NOT
// __syncthreads(): its fence makes hipcc wait for vmcnt(0) — every request in flight, the newest weight prefetch included —
// which turns a ring that is two taps deep into one that is one tap deep.
#define CD_WAIT_BARRIER(N) asm volatile("s_waitcnt vmcnt(" #N ")\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

__device__ __forceinline__ void cd_glds16(const void* g, GF_LDS char* l) {
    __builtin_amdgcn_global_load_lds((const GF_GLOBAL void*)g, (GF_LDS void*)l, 16, 0, 0);
}

template <int EPI, int NCB>
__global__ __launch_bounds__(CD_THREADS, 2) void conv3d_c96_kernel(const CdArgs p) {
    using SH = CdShape<NCB>;
    constexpr int PB = SH::PB, CB = SH::CB, CD_W_INSTR = SH::W_INSTR, CD_W_BYTES = SH::W_BYTES, CD_WK = SH::WK;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    GF_LDS char* lds = (GF_LDS char*)smem;
    GF_LDS char* halo = lds;
    GF_LDS char* wbuf = lds + CD_HALO_BYTES;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // NCB = 6: pixel group wp (rows 2 wp, 2 wp + 1), channel half wc (blocks 3 wc ..); NCB = 1: pixel row wp = wave, one channel block
    const int wp = NCB == 6 ? wave >> 1 : wave, wc = NCB == 6 ? wave & 1 : 0;
    const int row0 = NCB == 6 ? 2 * wp : wp;
    const int l15 = lane & 15, kc = lane >> 4;

    const int spatial = p.tiles_x * p.tiles_y;
    const int seg = blockIdx.x / spatial, sp = blockIdx.x - seg * spatial;
    const int y0 = (sp / p.tiles_x) * CD_TH, x0 = (sp % p.tiles_x) * CD_TW;
    const int j0 = seg * p.tseg, j1 = min(p.T_out, j0 + p.tseg);

    // ---- DMA sources.  The weight rows' offsets (3 per lane) are kept; a halo slot's place (pixel, 16-byte piece) does not depend
    // on the frame either, but keeping 10 per-lane offsets and the 64-bit addresses made from them alive across the frame loop
    // costs more registers than the ~25 integer instructions per DMA instruction that recompute them (one frame = 27 taps x 36
    // MFMAs per wave): `lane_v` is made opaque once per frame so that the compiler does not hoist the arithmetic out of the loop.
    int woff[CD_WK];
#pragma unroll
    for (int k = 0; k < CD_WK; ++k) {
        const int slot = (wave + CD_WAVES * k) * 64 + lane;
        const int n = slot / CD_SLOTS, sl = min(slot % CD_SLOTS, 11);                    // pad slots fetch valid bytes nobody reads
        woff[k] = n < p.N ? (int)(n * p.ldw * 2) + sl * 16 : -1;                         // rows past Cout: zeros
    }
    auto issue_w = [&](int tap, int stage_off) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < CD_WK; ++k) {
            const int i = wave + CD_WAVES * k;
            if (i < CD_W_INSTR) {
                int o = woff[k];
                asm volatile("" : "+v"(o));                // opaque: nine hoisted 64-bit row addresses (one per slot and piece) cost more registers than three adds per piece
                cd_glds16(o < 0 ? (const char*)p.zero : (const char*)p.w + o + tap * (CD_C * 2), wbuf + stage_off + i * 1024);
            }
        }
    };
    // DMA instructions this wave issues per tap: 3 or 2 (NCB = 6), 1 or 0 (NCB = 1) — what the counted wait leaves in flight
    const bool more = wave + CD_WAVES * (CD_WK - 1) < CD_W_INSTR;
    auto issue_halo = [&](int g, int lane_v) __attribute__((always_inline)) {
        const char* base = (const char*)(p.src + (long)g * p.frame);
#pragma unroll
        for (int k = 0; k < CD_HK; ++k) {
            const int i = wave + CD_WAVES * k;
            if (i < CD_HALO_INSTR) {
                const int slot = i * 64 + lane_v;
                const int px = slot / CD_SLOTS, sl = slot - px * CD_SLOTS;
                const int hy = px / CD_HW, hx = px - hy * CD_HW;
                const int y = y0 - 1 + hy, x = x0 - 1 + hx;
                const bool ok = px < CD_HH * CD_HW && sl < 12 && y >= 0 && y < p.H && x >= 0 && x < p.W;
                cd_glds16(ok ? base + (long)((y * p.W + x) * (CD_C * 2) + sl * 16) : (const char*)p.zero, halo + i * 1024);
            }
        }
    };

    // fragment read bases: lane (row l15, k chunk kc) of a 16-row x 32-channel fragment
    const int lane_off = l15 * CD_PITCH + kc * 16;
    GF_LDS char* const a_base = halo + lane_off + row0 * CD_HW * CD_PITCH;
    GF_LDS char* const b_base = wbuf + lane_off + CB * wc * 16 * CD_PITCH;
    // pixel block b of this wave inside the halo image: NCB = 6: row b >> 1, x (b & 1) * 16; NCB = 1: x 16 b
    auto blk_off = [](int b) { return NCB == 6 ? ((b >> 1) * CD_HW + (b & 1) * 16) * CD_PITCH : b * 16 * CD_PITCH; };

    f32x4 acc[3][PB][CB];
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int b = 0; b < PB; ++b)
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) acc[s][b][cb] = f32x4{0.f, 0.f, 0.f, 0.f};

    // The weight ring: three stages.  An LDS-DMA piece takes ~1.1 us from issue to landing under load — two taps of this
    // workgroup's MFMA time — so a tap's weights are requested TWO taps ahead and the counted wait at a tap's start leaves the
    // newest request in flight (with a 2-stage ring and vmcnt(0) every tap waited ~1 us: 5.6 ms per convolution, measured).
    int st_cur = 0, st_nxt = CD_W_BYTES, st_free = 2 * CD_W_BYTES;     // byte offsets of the stages holding tap t, t + 1 and the free one
    // The same 27 stages in the order (dy, dx) x dt: the three temporal taps of one spatial tap read the SAME pixels of the halo, so the A
    // fragments of a position (3 k steps x PB blocks) are read once and serve three stages — 12 + 27 fragment reads per position instead
    // of 36 + 27 (3.72 against 3.84 ms interleaved in one process, profiles/r03/conv_bench_pipe2.log; a further version with the barrier one
    // k step early, hand-counted lgkmcnt waits and the next position's A fragments prefetched was no faster: EXPERIMENTS.md).  Every
    // accumulator set still sums its taps in the order dy, dx: bit-identical.
    int s2_next2 = 0, dt_next2 = 2;                                     // the stage two ahead: weights of tap dt * 9 + s
    auto frame_taps = [&](bool v0, bool v1, bool v2) __attribute__((always_inline)) {
#pragma unroll 1
        for (int s = 0; s < 9; ++s) {
            const int dy = (s * 11) >> 5, dx = s - 3 * dy;
            GF_LDS char* const aaddr = a_base + (dy * CD_HW + dx) * CD_PITCH;
            bf16x8 af[3][PB], bfr[2][CB];
            auto stage = [&](auto set_c, bool valid) __attribute__((always_inline)) {
                constexpr int SET = decltype(set_c)::value;
                if (!(SET == 0 && s == 0)) {               // (the frame's first stage: the halo wait did this already)
                    if constexpr (NCB == 6) {
                        if (more) CD_WAIT_BARRIER(3);
                        else CD_WAIT_BARRIER(2);
                    } else {
                        if (more) CD_WAIT_BARRIER(1);
                        else CD_WAIT_BARRIER(0);
                    }
                }
                issue_w(dt_next2 * 9 + s2_next2, st_free);
                if (++dt_next2 == 3) {
                    dt_next2 = 0;
                    s2_next2 = s2_next2 == 8 ? 0 : s2_next2 + 1;
                }
                GF_LDS char* const baddr = b_base + st_cur;
                auto load = [&](int ks, int set) __attribute__((always_inline)) {
                    if constexpr (SET == 0) {
#pragma unroll
                        for (int b = 0; b < PB; ++b) af[ks][b] = *(GF_LDS bf16x8*)(aaddr + blk_off(b) + ks * 64);
                    }
#pragma unroll
                    for (int cb = 0; cb < CB; ++cb) bfr[set][cb] = *(GF_LDS bf16x8*)(baddr + cb * 16 * CD_PITCH + ks * 64);
                };
                load(0, 0);
#pragma unroll
                for (int ks = 0; ks < 3; ++ks) {
                    if (ks < 2) load(ks + 1, (ks + 1) & 1);
                    __builtin_amdgcn_sched_barrier(0);
                    if (valid) {
#pragma unroll
                        for (int b = 0; b < PB; ++b)
#pragma unroll
                            for (int cb = 0; cb < CB; ++cb)
                                acc[SET][b][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[ks & 1][cb], af[ks][b], acc[SET][b][cb], 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                const int t = st_cur;
                st_cur = st_nxt;
                st_nxt = st_free;
                st_free = t;
            };
            stage(std::integral_constant<int, 0>{}, v0);
            stage(std::integral_constant<int, 1>{}, v1);
            stage(std::integral_constant<int, 2>{}, v2);
        }
    };
    // this lane's bias values: channels (CB wc + cb) * 16 + 4 kc .. + 3
    u16x4 bias4[CB];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) {
        const int n = (CB * wc + cb) * 16 + kc * 4;
        bias4[cb] = (p.bias && n < p.N) ? *reinterpret_cast<const u16x4*>(p.bias + n) : u16x4{0, 0, 0, 0};
    }
    // Output frame j is complete in set 2: bias (+ residual), out.
    //   NCB = 6: four passes of one 16-pixel block x this wave's 48 channels through the FREE stage of the weight ring (nothing is
    //   requested into it before the frame's first tap), out as 16-byte pieces.  The residual pieces are requested FIRST and the
    //   next halo's DMA behind them (`after_loads`): vmcnt retires in order, so a load issued behind the DMA would make the epilogue
    //   wait for the whole halo.
    //   NCB = 1: a lane holds channels 4 kc .. 4 kc + 3 of its pixel — 8 bytes of the pixel's N-channel row, stored as they are.
    auto epilogue = [&](int j, auto&& after_loads) __attribute__((always_inline)) {
        if constexpr (NCB == 1) {
            u32x2 rr[PB];
            long offs[PB];
            bool okk[PB];
#pragma unroll
            for (int b = 0; b < PB; ++b) {
                const int y = y0 + wp, x = x0 + b * 16 + l15;
                okk[b] = y < p.H && x < p.W && kc * 4 < p.N;
                offs[b] = (((long)j * p.H + y) * p.W + x) * p.N + kc * 4;
                if constexpr (EPI == GF_EPI_BIAS_RESID) {
                    if (okk[b]) rr[b] = *reinterpret_cast<const u32x2*>(p.resid + offs[b]);
                }
            }
            after_loads();
#pragma unroll
            for (int b = 0; b < PB; ++b) {
                const f32x4 a = acc[2][b][0];
                u32x2 pk;
                pk[0] = pack2bf(a[0] + bf2f(bias4[0][0]), a[1] + bf2f(bias4[0][1]));
                pk[1] = pack2bf(a[2] + bf2f(bias4[0][2]), a[3] + bf2f(bias4[0][3]));
                if constexpr (EPI == GF_EPI_BIAS_RESID) {
#pragma unroll
                    for (int h2 = 0; h2 < 2; ++h2) {
                        const float lo = bf2f((u16)(rr[b][h2] & 0xffffu)) + bf2f((u16)(pk[h2] & 0xffffu));
                        const float hi = bf2f((u16)(rr[b][h2] >> 16)) + bf2f((u16)(pk[h2] >> 16));
                        pk[h2] = pack2bf(lo, hi);
                    }
                }
                if (okk[b]) *reinterpret_cast<u32x2*>(p.out + offs[b]) = pk;
            }
        } else {
        GF_LDS char* ep = wbuf + st_free + wave * CD_EP_BYTES;
        u16x8 rres[4][2];
        long offs[4][2];
        bool okk[4][2];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int y = y0 + 2 * wp + (b >> 1), xb = x0 + (b & 1) * 16;
            const long blkbase = (((long)j * p.H + y) * p.W + xb) * CD_C + wc * 48;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int s = i * 64 + lane;                          // 16-byte piece s of the block image: pixel s / 6, piece s % 6 of its 96 bytes
                const int px = s / 6, sl = s - px * 6;
                okk[b][i] = s < 96 && y < p.H && xb + px < p.W;
                offs[b][i] = blkbase + (long)px * CD_C + sl * 8;
                if constexpr (EPI == GF_EPI_BIAS_RESID) {
                    if (okk[b][i]) rres[b][i] = *reinterpret_cast<const u16x8*>(p.resid + offs[b][i]);
                }
            }
        }
        after_loads();
#pragma unroll
        for (int b = 0; b < 4; ++b) {
#pragma unroll
            for (int cb = 0; cb < 3; ++cb) {
                const f32x4 a = acc[2][b][cb];
                u32x2 pk;
                pk[0] = pack2bf(a[0] + bf2f(bias4[cb][0]), a[1] + bf2f(bias4[cb][1]));
                pk[1] = pack2bf(a[2] + bf2f(bias4[cb][2]), a[3] + bf2f(bias4[cb][3]));
                *(GF_LDS u32x2*)(ep + l15 * 96 + cb * 32 + kc * 8) = pk;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // wave-private image
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                if (okk[b][i]) {
                    const u16x8 v = *(GF_LDS u16x8*)(ep + (i * 64 + lane) * 16);
                    u16x8 o = v;
                    if constexpr (EPI == GF_EPI_BIAS_RESID) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) o[e] = f2bf(bf2f(rres[b][i][e]) + bf2f(v[e]));
                    }
                    *reinterpret_cast<u16x8*>(p.out + offs[b][i]) = o;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the image is rewritten by the next block
        }
        }
    };
    // The sets are named by temporal tap: set dt collects output frame g - dt while input frame g is processed.  At the start of a
    // frame step set 2 holds the output completed by the previous frame: it is stored (under the new halo's DMA), then the sets
    // move up one place (register moves: 96 per frame and lane against 972 MFMAs) and set 0 starts the new output frame from 0.
    auto rotate = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int b = 0; b < PB; ++b)
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) {
                acc[2][b][cb] = acc[1][b][cb];
                acc[1][b][cb] = acc[0][b][cb];
                acc[0][b][cb] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
    };
    issue_w(0, 0);
    issue_w(9, CD_W_BYTES);      // (the stage order is (dy, dx) x dt: the second stage is tap dt = 1 of position 0)
#pragma unroll 1
    for (int g = j0; g < j1 + 2; ++g) {                    // input frame g of the walk: output j reads g = j, j + 1, j + 2
        CD_WAIT_BARRIER(0);                                // every wave is done with the previous frame's halo (and its last tap)
        int lane_v = lane;
        asm volatile("" : "+v"(lane_v));                   // opaque: see issue_halo
        if (g - 3 >= j0) epilogue(g - 3, [&]() __attribute__((always_inline)) { issue_halo(g, lane_v); });   // completed a frame ago, stored under the DMA
        else issue_halo(g, lane_v);
        rotate();
        CD_WAIT_BARRIER(0);                                // halo g (and the weights of its first two taps) landed
        frame_taps(g >= j0 && g < j1, g - 1 >= j0 && g - 1 < j1, g - 2 >= j0 && g - 2 < j1);
    }
    CD_WAIT_BARRIER(0);                                    // the ring's last prefetches (never read) must not outlive the workgroup's LDS
    epilogue(j1 - 1, []() {});                             // the segment's last output frame finished with input j1 + 1 (set 2)
}

// ---------------------------------------------------------------------------------------------------------------------------
// The decoder's last spatial upsample (Resample "upsample2d", VAE:91-99: nearest-exact 2x + Conv2d 3x3, 192 -> 96 channels at full
// resolution): as an implicit GEMM it ran at 0.37 PFLOP/s (7.2 ms for 2.65 TFLOP per launch: every source pixel gathered 9 x 4 times
// through L2, the N tile three quarters full).  Direct form on the same skeleton as above:
//   * a workgroup owns an 8 x 32 patch of OUTPUT pixels x 96 channels of one frame after the other; the patch's source is a
//     (4 + 2) x (16 + 2) halo of the HALF-resolution frame (108 pixels x 192 channels, 416-byte pitch = 384 + 32: conflict-free
//     fragment reads; two output pixels share a source pixel: the same address, a broadcast), double buffered — the next frame's
//     halo is requested late in the current frame so that the counted weight waits stay simple;
//   * the K loop is 9 taps x 2 channel halves = 18 stages of a [96 x 96] weight block — the weight ring, the B fragments, the
//     barrier per stage and the epilogue are the 96-channel kernel's; the tap (dy, dx) of output pixel (y, x) reads source pixel
//     ((y + dy - 1) >> 1, (x + dx - 1) >> 1): the row part is wave-uniform, the column part three per-lane offsets per 16-pixel block;
//   * one accumulator set (48 registers); a frame's result leaves at the start of the next frame, under its first weight waits.
// Same summation order as the implicit GEMM (taps in sequence, 32 channels per MFMA): bit-identical (tests/test_vae.py).
constexpr int UP_C = 192, UP_PITCH = 416, UP_SLOTS = 26;                             // 24 data + 2 pad slots of 16 bytes per source pixel
constexpr int UP_HH = CD_TH / 2 + 2, UP_HW = CD_TW / 2 + 2;                         // 6 x 18 source pixels
constexpr int UP_HALO_INSTR = (UP_HH * UP_HW * UP_SLOTS + 63) / 64;                  // 44
constexpr int UP_HALO_BYTES = UP_HALO_INSTR * 1024;                                  // 45056
constexpr int UP_HK = (UP_HALO_INSTR + CD_WAVES - 1) / CD_WAVES;                     // 6
constexpr int UP_STAGES = 18;
constexpr int UP_LDS = 2 * UP_HALO_BYTES + CD_STAGES * CdShape<6>::W_BYTES;          // 154624

struct UpArgs {
    const u16* src;     // [T, H/2, W/2, 192]: frame 0 = the frame output 0 reads
    const u16* w;       // [96, ldw], K order (dy, dx, cin)
    const u16* bias;
    u16* out;           // [T_out, H, W, 96]
    const u16* zero;
    int H, W, T_out, tseg, tiles_x, tiles_y;     // H, W: OUTPUT resolution
    long sframe;        // (H/2) * (W/2) * 192
    long ldw;
};

__global__ __launch_bounds__(CD_THREADS, 1) void conv2d_up_c192_kernel(const UpArgs p) {
    using SH = CdShape<6>;
    constexpr int PB = 4, CB = 3, W_INSTR = SH::W_INSTR, W_BYTES = SH::W_BYTES, WK = SH::WK;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    GF_LDS char* lds = (GF_LDS char*)smem;
    GF_LDS char* wbuf = lds + 2 * UP_HALO_BYTES;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wp = wave >> 1, wc = wave & 1;
    const int l15 = lane & 15, kc = lane >> 4;
    const int spatial = p.tiles_x * p.tiles_y;
    const int seg = blockIdx.x / spatial, sp = blockIdx.x - seg * spatial;
    const int y0 = (sp / p.tiles_x) * CD_TH, x0 = (sp % p.tiles_x) * CD_TW;
    const int j0 = seg * p.tseg, j1 = min(p.T_out, j0 + p.tseg);
    const int Hs = p.H >> 1, Ws = p.W >> 1, sy0 = (y0 >> 1) - 1, sx0 = (x0 >> 1) - 1;

    int woff[WK];
#pragma unroll
    for (int k = 0; k < WK; ++k) {
        const int slot = (wave + CD_WAVES * k) * 64 + lane;
        const int n = slot / CD_SLOTS, sl = min(slot % CD_SLOTS, 11);
        woff[k] = (int)(n * p.ldw * 2) + sl * 16;
    }
    auto issue_w = [&](int stage, int stage_off) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < WK; ++k) {
            const int i = wave + CD_WAVES * k;
            if (i < W_INSTR) cd_glds16((const char*)p.w + woff[k] + stage * (CD_C * 2), wbuf + stage_off + i * 1024);
        }
    };
    const bool more = wave + CD_WAVES * (WK - 1) < W_INSTR;          // 3 weight requests per stage (else 2)
    const bool hmore = wave + CD_WAVES * (UP_HK - 1) < UP_HALO_INSTR; // 6 halo requests per frame (else 5)
    auto issue_halo = [&](int g, int lane_v, GF_LDS char* hbuf) __attribute__((always_inline)) {
        const char* base = (const char*)(p.src + (long)g * p.sframe);
#pragma unroll
        for (int k = 0; k < UP_HK; ++k) {
            const int i = wave + CD_WAVES * k;
            if (i < UP_HALO_INSTR) {
                const int slot = i * 64 + lane_v;
                const int px = slot / UP_SLOTS, sl = slot - px * UP_SLOTS;
                const int hy = px / UP_HW, hx = px - hy * UP_HW;
                const int y = sy0 + hy, x = sx0 + hx;
                const bool ok = px < UP_HH * UP_HW && sl < 24 && y >= 0 && y < Hs && x >= 0 && x < Ws;
                cd_glds16(ok ? base + (long)((y * Ws + x) * (UP_C * 2) + sl * 16) : (const char*)p.zero, hbuf + i * 1024);
            }
        }
    };
    // A fragment of pixel block b (row 2 wp + (b >> 1), columns 16 (b & 1) + l15) at tap (dy, dx): halo pixel
    // (((row + dy - 1) >> 1) + 1, ((col + dx - 1) >> 1) + 1); per lane: the column part for (dx, b & 1)
    int colo[3][2];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx)
#pragma unroll
        for (int xb = 0; xb < 2; ++xb) colo[dx][xb] = (((16 * xb + l15 + dx - 1) >> 1) + 1) * UP_PITCH + kc * 16;
    GF_LDS char* const b_base = wbuf + l15 * CD_PITCH + kc * 16 + CB * wc * 16 * CD_PITCH;

    f32x4 acc[PB][CB];
    u16x4 bias4[CB];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) {
        const int n = (CB * wc + cb) * 16 + kc * 4;
        bias4[cb] = p.bias ? *reinterpret_cast<const u16x4*>(p.bias + n) : u16x4{0, 0, 0, 0};
    }
    int st_cur = 0, st_nxt = W_BYTES, st_free = 2 * W_BYTES;
    int stage_next2 = 2;
    // frame j's accumulators -> out (bias), staged as in the kernel above — in the halo buffer that is idle at a frame's start (the
    // weight ring is requested into again right behind the epilogue, with no barrier in between)
    auto epilogue = [&](int j, GF_LDS char* idle) __attribute__((always_inline)) {
        GF_LDS char* ep = idle + wave * CD_EP_BYTES;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int y = y0 + 2 * wp + (b >> 1), xb = x0 + (b & 1) * 16;
            const long blkbase = (((long)j * p.H + y) * p.W + xb) * CD_C + wc * 48;
#pragma unroll
            for (int cb = 0; cb < 3; ++cb) {
                const f32x4 a = acc[b][cb];
                u32x2 pk;
                pk[0] = pack2bf(a[0] + bf2f(bias4[cb][0]), a[1] + bf2f(bias4[cb][1]));
                pk[1] = pack2bf(a[2] + bf2f(bias4[cb][2]), a[3] + bf2f(bias4[cb][3]));
                *(GF_LDS u32x2*)(ep + l15 * 96 + cb * 32 + kc * 8) = pk;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int s = i * 64 + lane;
                const int px = s / 6, sl = s - px * 6;
                if (s < 96 && y < p.H && xb + px < p.W)
                    *reinterpret_cast<u16x8*>(p.out + blkbase + (long)px * CD_C + sl * 8) = *(GF_LDS u16x8*)(ep + s * 16);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    };
    issue_w(0, 0);
    issue_w(1, W_BYTES);
    {
        int lane_v = lane;
        asm volatile("" : "+v"(lane_v));
        issue_halo(j0, lane_v, lds);
    }
#pragma unroll 1
    for (int g = j0; g < j1; ++g) {
        GF_LDS char* const hbuf = lds + ((g - j0) & 1) * UP_HALO_BYTES;
        GF_LDS char* const hnext = lds + (((g - j0) & 1) ^ 1) * UP_HALO_BYTES;
        CD_WAIT_BARRIER(0);                                // halo g and the first two stages' weights landed; the other halo buffer is free
        if (g > j0) epilogue(g - 1, hnext);
#pragma unroll
        for (int b = 0; b < PB; ++b)
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) acc[b][cb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
        for (int s = 0; s < UP_STAGES; ++s) {
            if (s > 0) {
                // this stage's weights landed; in flight stay the newest weight request (3 or 2) and, from stage 15 on, the next halo
                if (s <= 15) {
                    if (more) CD_WAIT_BARRIER(3);
                    else CD_WAIT_BARRIER(2);
                } else if (hmore) {
                    if (more) CD_WAIT_BARRIER(9);
                    else CD_WAIT_BARRIER(8);
                } else {
                    if (more) CD_WAIT_BARRIER(8);
                    else CD_WAIT_BARRIER(7);
                }
            }
            issue_w(stage_next2, st_free);
            stage_next2 = stage_next2 == UP_STAGES - 1 ? 0 : stage_next2 + 1;
            if (s == 15) {
                int lane_v = lane;
                asm volatile("" : "+v"(lane_v));
                issue_halo(g + 1 < j1 ? g + 1 : g, lane_v, hnext);     // (last frame: a re-fetch nobody reads keeps the counts uniform)
            }
            const int tap = s >> 1, hf = s & 1;
            const int dy = (tap * 11) >> 5, dx = tap - 3 * dy;
            GF_LDS char* const baddr = b_base + st_cur;
            GF_LDS char* arow[2];
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) arow[rr] = hbuf + (((2 * wp + rr + dy - 1) >> 1) + 1) * (UP_HW * UP_PITCH) + hf * (CD_C * 2);
            const int c0 = dx == 0 ? colo[0][0] : (dx == 1 ? colo[1][0] : colo[2][0]);
            const int c1 = dx == 0 ? colo[0][1] : (dx == 1 ? colo[1][1] : colo[2][1]);
            bf16x8 af[2][PB], bfr[2][CB];
            auto load = [&](int ks, int set) __attribute__((always_inline)) {
#pragma unroll
                for (int b = 0; b < PB; ++b) af[set][b] = *(GF_LDS bf16x8*)(arow[b >> 1] + ((b & 1) ? c1 : c0) + ks * 64);
#pragma unroll
                for (int cb = 0; cb < CB; ++cb) bfr[set][cb] = *(GF_LDS bf16x8*)(baddr + cb * 16 * CD_PITCH + ks * 64);
            };
            load(0, 0);
#pragma unroll
            for (int ks = 0; ks < 3; ++ks) {
                if (ks < 2) load(ks + 1, (ks + 1) & 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int b = 0; b < PB; ++b)
#pragma unroll
                    for (int cb = 0; cb < CB; ++cb)
                        acc[b][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[ks & 1][cb], af[ks & 1][b], acc[b][cb], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            const int t = st_cur;
            st_cur = st_nxt;
            st_nxt = st_free;
            st_free = t;
        }
    }
    CD_WAIT_BARRIER(0);                                    // the ring's last prefetches and the spare halo must not outlive the workgroup's LDS
    epilogue(j1 - 1, lds + (((j1 - j0) & 1)) * UP_HALO_BYTES);       // either buffer is idle now
}

template <int EPI, int NCB>
int launch_cd(const CdArgs& a, unsigned grid, hipStream_t stream) {
    static GfDeviceOnce once;
    hipError_t e = gf_once_per_device(once, [] {
        return hipFuncSetAttribute(reinterpret_cast<const void*>(conv3d_c96_kernel<EPI, NCB>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   CdShape<NCB>::LDS);
    });
    if (e != hipSuccess) {
        gf_set_error("gf_conv3d_bf16 (direct): hipFuncSetAttribute(%d B LDS) failed: %s", CdShape<NCB>::LDS, hipGetErrorString(e));
        return GF_ERR_LAUNCH;
    }
    hipLaunchKernelGGL((conv3d_c96_kernel<EPI, NCB>), dim3(grid), dim3(CD_THREADS), CdShape<NCB>::LDS, stream, a);
    GF_CHECK_LAUNCH("gf_conv3d_bf16 (direct)");
    return GF_OK;
}

}  // namespace

// Called by gf_conv3d_bf16 (gf_gemm.hip) for: kt = ks = 3, stride 1, no resampling, history in front of src, C = 96, N = 96 or
// N <= 16 (the decoder's RGB head), contiguous out / resid rows of N channels.  `src_walk` = the frame output 0 reads with its
// FIRST temporal tap.  Returns GF_ERR_UNSUPPORTED when the shape is outside what the kernels cover (the caller then takes the
// implicit GEMM).
int gf_conv3d_direct_c96(const void* src_walk, const void* Wm, int64_t ldw, const void* bias, void* out, int64_t T_out, int64_t H,
                         int64_t W, int64_t N, int epilogue, const void* resid, const void* zero_page, void* stream) {
    if (H * W * CD_C * 2 >= (1LL << 31) || ldw * 2 * CD_C >= (1LL << 31) || T_out <= 0) return GF_ERR_UNSUPPORTED;
    if (!(N == CD_C || (N <= 16 && N % 4 == 0))) return GF_ERR_UNSUPPORTED;
    CdArgs a;
    a.src = (const u16*)src_walk;
    a.w = (const u16*)Wm;
    a.bias = (const u16*)bias;
    a.out = (u16*)out;
    a.resid = (const u16*)resid;
    a.zero = (const u16*)zero_page;
    a.H = (int)H;
    a.W = (int)W;
    a.T_out = (int)T_out;
    a.N = (int)N;
    a.tiles_x = (int)((W + CD_TW - 1) / CD_TW);
    a.tiles_y = (int)((H + CD_TH - 1) / CD_TH);
    a.frame = H * W * CD_C;
    a.ldw = ldw;
    // segments of frames: enough workgroups for a few rounds over the 256 CUs, segments not shorter than 4 frames (each pays two
    // input frames of lead-in)
    const long spatial = (long)a.tiles_x * a.tiles_y;
    long nseg = (4 * 256 + spatial - 1) / spatial;
    if (nseg > (T_out + 3) / 4) nseg = (T_out + 3) / 4;
    if (nseg < 1) nseg = 1;
    a.tseg = (int)((T_out + nseg - 1) / nseg);
    nseg = (T_out + a.tseg - 1) / a.tseg;
    const unsigned grid = (unsigned)(spatial * nseg);
    hipStream_t s = (hipStream_t)stream;
    if (N == CD_C) return epilogue == GF_EPI_BIAS_RESID ? launch_cd<GF_EPI_BIAS_RESID, 6>(a, grid, s) : launch_cd<GF_EPI_BIAS, 6>(a, grid, s);
    return epilogue == GF_EPI_BIAS_RESID ? launch_cd<GF_EPI_BIAS_RESID, 1>(a, grid, s) : launch_cd<GF_EPI_BIAS, 1>(a, grid, s);
}

// Called by gf_conv3d_bf16 for: kt = 1, ks = 3, mode 1 (nearest-exact 2x upsample folded in), C = 192, N = 96, bias epilogue, contiguous
// out rows.  `src0` = the source frame output 0 reads; Hs, Ws = SOURCE resolution.
int gf_conv2d_up_direct_c192(const void* src0, const void* Wm, int64_t ldw, const void* bias, void* out, int64_t T_out, int64_t Hs,
                             int64_t Ws, const void* zero_page, void* stream) {
    const int64_t H = 2 * Hs, W = 2 * Ws;
    if (Hs * Ws * UP_C * 2 >= (1LL << 31) || ldw * 2 * CD_C >= (1LL << 31) || T_out <= 0) return GF_ERR_UNSUPPORTED;
    UpArgs a;
    a.src = (const u16*)src0;
    a.w = (const u16*)Wm;
    a.bias = (const u16*)bias;
    a.out = (u16*)out;
    a.zero = (const u16*)zero_page;
    a.H = (int)H;
    a.W = (int)W;
    a.T_out = (int)T_out;
    a.tiles_x = (int)((W + CD_TW - 1) / CD_TW);
    a.tiles_y = (int)((H + CD_TH - 1) / CD_TH);
    a.sframe = Hs * Ws * UP_C;
    a.ldw = ldw;
    const long spatial = (long)a.tiles_x * a.tiles_y;
    long nseg = (3 * 256 + spatial - 1) / spatial;       // one workgroup per CU at a time: a few rounds over the chip
    if (nseg > T_out) nseg = T_out;
    if (nseg < 1) nseg = 1;
    a.tseg = (int)((T_out + nseg - 1) / nseg);
    nseg = (T_out + a.tseg - 1) / a.tseg;
    static GfDeviceOnce once;
    hipError_t e = gf_once_per_device(once, [] {
        return hipFuncSetAttribute(reinterpret_cast<const void*>(conv2d_up_c192_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, UP_LDS);
    });
    if (e != hipSuccess) {
        gf_set_error("gf_conv3d_bf16 (direct upsample): hipFuncSetAttribute(%d B LDS) failed: %s", UP_LDS, hipGetErrorString(e));
        return GF_ERR_LAUNCH;
    }
    hipLaunchKernelGGL(conv2d_up_c192_kernel, dim3((unsigned)(spatial * nseg)), dim3(CD_THREADS), UP_LDS, (hipStream_t)stream, a);
    GF_CHECK_LAUNCH("gf_conv3d_bf16 (direct upsample)");
    return GF_OK;
}
