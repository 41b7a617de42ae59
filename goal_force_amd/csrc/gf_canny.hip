// gf_canny.hip — the Canny-edge control signal of ControlSignalDataset_CannyEdge (src/goal_force/unified_dataset.py:406-613,
// `_generate_control_video` :559-578) on the GPU.  The reference calls, per frame, controlnet_aux.CannyDetector (resize so that the
// short side is 512 and both sides are multiples of 64: cv2.resize INTER_LANCZOS4 when enlarging, INTER_AREA otherwise;
// cv2.Canny(img, 100, 200); grey -> 3 channels), then cv2.resize(..., INTER_AREA) back to the frame size and `x / 127.5 - 1`
// -> bf16.  Neither OpenCV nor controlnet_aux is in this image, so what is restated here is OpenCV's PUBLISHED algorithm
// (imgproc/src/resize.cpp, canny.cpp), integer for integer where it is integer work — "parity unpinned" (oracle/canny_oracle.py).
// All frames of a clip go through each kernel in one launch; every kernel is byte/integer work bound by HBM (a 49-frame clip is
// 59 MB in, 118 MB out), one thread per pixel, rows contiguous across the lanes of a wave.
#include "gf_common.h"

namespace {

constexpr int CN_THREADS = 256;

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// cv2.resize(INTER_LANCZOS4) on 8-bit pixels (resize.cpp: HResizeLanczos4<uchar,int,short> + VResizeLanczos4 with
// FixedPtCast<int, uchar, 22>): 8 x 8 taps at sx - 3 .. sx + 4 (replicated border), coefficients as 11-bit fixed point from the
// host tables, the two passes' products summed exactly, one rounding shift by 22 at the end.
__global__ __launch_bounds__(CN_THREADS) void lanczos4_u8_kernel(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst,
                                                                 const int* __restrict__ xofs, const short* __restrict__ xco,
                                                                 const int* __restrict__ yofs, const short* __restrict__ yco, int T,
                                                                 int H, int W, int Hd, int Wd) {
    const long total = (long)T * Hd * Wd;
    for (long i = (long)blockIdx.x * CN_THREADS + threadIdx.x; i < total; i += (long)gridDim.x * CN_THREADS) {
        const int dx = (int)(i % Wd), dy = (int)((i / Wd) % Hd), t = (int)(i / ((long)Wd * Hd));
        const int sx = xofs[dx], sy = yofs[dy];
        const unsigned char* f = src + (long)t * H * W * 3;
        long acc[3] = {0, 0, 0};
#pragma unroll
        for (int ky = 0; ky < 8; ++ky) {
            const unsigned char* row = f + (long)clampi(sy + ky - 3, 0, H - 1) * W * 3;
            int h[3] = {0, 0, 0};
#pragma unroll
            for (int kx = 0; kx < 8; ++kx) {
                const unsigned char* px = row + clampi(sx + kx - 3, 0, W - 1) * 3;
                const int a = xco[dx * 8 + kx];
                h[0] += px[0] * a;
                h[1] += px[1] * a;
                h[2] += px[2] * a;
            }
            const long b = yco[dy * 8 + ky];
            acc[0] += b * h[0];
            acc[1] += b * h[1];
            acc[2] += b * h[2];
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) dst[i * 3 + c] = (unsigned char)clampi((int)((acc[c] + (1L << 21)) >> 22), 0, 255);
    }
}

// cv2.resize(INTER_AREA), shrinking, 8-bit (resize.cpp: computeResizeAreaTab + ResizeArea_): every destination pixel is the
// overlap-weighted mean of the source cells it covers; fp32 weights from the host tables (CSR: start / source index / alpha),
// horizontal sums first, then the vertical combination, one float -> u8 rounding (round half to even) at the end.  NC = 3: colour
// frames.  NC = 1 with `edges`: the source is the Canny state map (2 = edge -> 255, else 0) and the result goes straight to the
// dataset's output format: 3 equal channels of bf16(x / 127.5 - 1).
template <int NC, bool EDGES>
__global__ __launch_bounds__(CN_THREADS) void area_u8_kernel(const unsigned char* __restrict__ src, void* __restrict__ dst,
                                                             const int* __restrict__ xstart, const int* __restrict__ xsrc,
                                                             const float* __restrict__ xalpha, const int* __restrict__ ystart,
                                                             const int* __restrict__ ysrc, const float* __restrict__ yalpha, int T, int H,
                                                             int W, int Hd, int Wd) {
    const long total = (long)T * Hd * Wd;
    for (long i = (long)blockIdx.x * CN_THREADS + threadIdx.x; i < total; i += (long)gridDim.x * CN_THREADS) {
        const int dx = (int)(i % Wd), dy = (int)((i / Wd) % Hd), t = (int)(i / ((long)Wd * Hd));
        const unsigned char* f = src + (long)t * H * W * NC;
        float sum[NC];
        bool first = true;
        for (int jy = ystart[dy]; jy < ystart[dy + 1]; ++jy) {
            const unsigned char* row = f + (long)ysrc[jy] * W * NC;
            float buf[NC];
#pragma unroll
            for (int c = 0; c < NC; ++c) buf[c] = 0.f;
            for (int jx = xstart[dx]; jx < xstart[dx + 1]; ++jx) {
                const unsigned char* px = row + xsrc[jx] * NC;
                const float a = xalpha[jx];
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    const float s = EDGES ? (px[c] == 2 ? 255.f : 0.f) : (float)px[c];
                    float m = s * a;                                      // OpenCV: buf[dx] += S[sx] * alpha — product and sum round
                    asm volatile("" : "+v"(m));                           // separately: the empty asm keeps hipcc (-ffp-contract=fast,
                    buf[c] = buf[c] + m;                                  // which also fuses __fmul_rn / __fadd_rn) from forming an fma
                }
            }
            const float b = yalpha[jy];
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                float m = b * buf[c];
                asm volatile("" : "+v"(m));
                sum[c] = first ? m : sum[c] + m;
            }
            first = false;
        }
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int v = clampi((int)__builtin_rintf(sum[c]), 0, 255);                  // saturate_cast<uchar>: cvRound
            if constexpr (EDGES) {
                float qv = __fdiv_rn((float)v, 127.5f);                                  // uint8 -> float32 / 127.5 - 1.0 -> bf16
                asm volatile("" : "+v"(qv));
                const u16 o = f2bf(qv - 1.0f);
                u16* d = (u16*)dst + i * 3;
                d[0] = o;
                d[1] = o;
                d[2] = o;
            } else {
                ((unsigned char*)dst)[i * NC + c] = (unsigned char)v;
            }
        }
    }
}

// cv2.Canny, aperture 3, L1 gradient (canny.cpp): Sobel dx / dy per channel with a replicated border; of the 3 channels the first
// one with the largest |dx| + |dy| supplies the pixel's gradient.  out: mag int32, dxy = (dy << 16) | (dx & 0xffff).
__global__ __launch_bounds__(CN_THREADS) void canny_grad_kernel(const unsigned char* __restrict__ img, int* __restrict__ mag,
                                                                int* __restrict__ dxy, int T, int H, int W) {
    const long total = (long)T * H * W;
    for (long i = (long)blockIdx.x * CN_THREADS + threadIdx.x; i < total; i += (long)gridDim.x * CN_THREADS) {
        const int x = (int)(i % W), y = (int)((i / W) % H), t = (int)(i / ((long)W * H));
        const unsigned char* f = img + (long)t * H * W * 3;
        const int xm = x > 0 ? x - 1 : 0, xp = x < W - 1 ? x + 1 : W - 1;
        const unsigned char* r0 = f + (long)(y > 0 ? y - 1 : 0) * W * 3;
        const unsigned char* r1 = f + (long)y * W * 3;
        const unsigned char* r2 = f + (long)(y < H - 1 ? y + 1 : H - 1) * W * 3;
        int bm = -1, bdx = 0, bdy = 0;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int a00 = r0[xm * 3 + c], a01 = r0[x * 3 + c], a02 = r0[xp * 3 + c];
            const int a10 = r1[xm * 3 + c], a12 = r1[xp * 3 + c];
            const int a20 = r2[xm * 3 + c], a21 = r2[x * 3 + c], a22 = r2[xp * 3 + c];
            const int gx = (a02 + 2 * a12 + a22) - (a00 + 2 * a10 + a20);
            const int gy = (a20 + 2 * a21 + a22) - (a00 + 2 * a01 + a02);
            const int m = (gx < 0 ? -gx : gx) + (gy < 0 ? -gy : gy);
            if (m > bm) {
                bm = m;
                bdx = gx;
                bdy = gy;
            }
        }
        mag[i] = bm;
        dxy[i] = (int)(((unsigned)bdy << 16) | ((unsigned)bdx & 0xffffu));
    }
}

// Non-maximum suppression + double threshold (canny.cpp): state 1 = not an edge, 0 = might belong to an edge, 2 = edge.
// The magnitude map is zero outside the image.  Direction by OpenCV's fixed-point tangents (TG22 = tan 22.5 deg * 2^15).
__global__ __launch_bounds__(CN_THREADS) void canny_nms_kernel(const int* __restrict__ mag, const int* __restrict__ dxy,
                                                               unsigned char* __restrict__ state, int T, int H, int W, int low, int high) {
    const long total = (long)T * H * W;
    for (long i = (long)blockIdx.x * CN_THREADS + threadIdx.x; i < total; i += (long)gridDim.x * CN_THREADS) {
        const int x = (int)(i % W), y = (int)((i / W) % H);
        const int* mf = mag + (i - (long)y * W - x);
        auto M = [&](int yy, int xx) { return (yy < 0 || yy >= H || xx < 0 || xx >= W) ? 0 : mf[(long)yy * W + xx]; };
        const int m = mf[(long)y * W + x];
        unsigned char st = 1;
        if (m > low) {
            const int v = dxy[i];
            const int gx = (int)(short)(v & 0xffff), gy = v >> 16;
            const int ax = gx < 0 ? -gx : gx;
            const int ay = (gy < 0 ? -gy : gy) << 15;
            const int tg22x = ax * 13573;                       // (int)(0.4142135623730950488016887242097 * (1 << 15) + 0.5)
            bool keep;
            if (ay < tg22x) {
                keep = m > M(y, x - 1) && m >= M(y, x + 1);
            } else {
                const int tg67x = tg22x + (ax << 16);
                if (ay > tg67x) {
                    keep = m > M(y - 1, x) && m >= M(y + 1, x);
                } else {
                    const int s = ((gx ^ gy) < 0) ? -1 : 1;
                    keep = m > M(y - 1, x - s) && m > M(y + 1, x + s);
                }
            }
            if (keep) st = m > high ? 2 : 0;
        }
        state[i] = st;
    }
}

// Hysteresis: a "might be" pixel (0) with an edge pixel (2) among its 8 neighbours becomes an edge.  One workgroup owns a
// 32 x 32 tile (+ 1 halo) in LDS and iterates until its tile is stable; *changed is raised when a tile changed, the host repeats
// the launch until none does.  Updates are monotone (0 -> 2), so the order in which tiles run does not matter for the fixpoint.
__global__ __launch_bounds__(CN_THREADS) void canny_hyst_kernel(unsigned char* __restrict__ state, int* __restrict__ changed, int H, int W,
                                                                int tiles_x, int tiles_y) {
    __shared__ unsigned char tile[34][36];
    __shared__ int again, any;
    const int tpf = tiles_x * tiles_y;
    const int t = blockIdx.x / tpf, tin = blockIdx.x % tpf;
    const int y0 = (tin / tiles_x) * 32, x0 = (tin % tiles_x) * 32;
    unsigned char* f = state + (long)t * H * W;
    for (int k = threadIdx.x; k < 34 * 34; k += CN_THREADS) {
        const int ly = k / 34, lx = k % 34, gy = y0 + ly - 1, gx = x0 + lx - 1;
        tile[ly][lx] = (gy < 0 || gy >= H || gx < 0 || gx >= W) ? 1 : f[(long)gy * W + gx];
    }
    if (threadIdx.x == 0) any = 0;
    __syncthreads();
    const int lx = (threadIdx.x & 31) + 1, lyb = (threadIdx.x >> 5) * 4 + 1;      // 4 rows per thread
    for (;;) {
        if (threadIdx.x == 0) again = 0;
        __syncthreads();
        bool ch = false;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int ly = lyb + j;
            if (tile[ly][lx] == 0) {
                const bool e = tile[ly - 1][lx - 1] == 2 || tile[ly - 1][lx] == 2 || tile[ly - 1][lx + 1] == 2 || tile[ly][lx - 1] == 2 ||
                               tile[ly][lx + 1] == 2 || tile[ly + 1][lx - 1] == 2 || tile[ly + 1][lx] == 2 || tile[ly + 1][lx + 1] == 2;
                if (e) {
                    tile[ly][lx] = 2;
                    ch = true;
                }
            }
        }
        if (ch) again = 1;
        __syncthreads();
        if (!again) break;
        if (threadIdx.x == 0) any = 1;
        __syncthreads();
    }
    if (any) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int gy = y0 + lyb + j - 1, gx = x0 + lx - 1;
            if (gy < H && gx < W && tile[lyb + j][lx] == 2) f[(long)gy * W + gx] = 2;
        }
        if (threadIdx.x == 0) atomicOr(changed, 1);
    }
}

inline unsigned cn_grid(long n) {
    const long b = (n + CN_THREADS - 1) / CN_THREADS;
    return (unsigned)(b < 1 ? 1 : (b > 65535L * 16 ? 65535L * 16 : b));
}

}  // namespace

extern "C" GF_API int gf_resize_lanczos4_u8(const void* src, void* dst, const int* xofs, const short* xcoef, const int* yofs,
                                            const short* ycoef, int64_t frames, int64_t H, int64_t W, int64_t Hd, int64_t Wd, void* stream) {
    GF_CHECK_ARG(src && dst && xofs && xcoef && yofs && ycoef, "gf_resize_lanczos4_u8: null pointer");
    GF_CHECK_ARG(frames >= 0 && H > 0 && W > 0 && Hd > 0 && Wd > 0, "gf_resize_lanczos4_u8: bad sizes");
    if (frames == 0) return GF_OK;
    hipLaunchKernelGGL(lanczos4_u8_kernel, dim3(cn_grid(frames * Hd * Wd)), dim3(CN_THREADS), 0, (hipStream_t)stream,
                       (const unsigned char*)src, (unsigned char*)dst, xofs, xcoef, yofs, ycoef, (int)frames, (int)H, (int)W, (int)Hd, (int)Wd);
    GF_CHECK_LAUNCH("gf_resize_lanczos4_u8");
    return GF_OK;
}

extern "C" GF_API int gf_resize_area_u8(const void* src, void* dst, const int* xstart, const int* xsrc, const float* xalpha, const int* ystart,
                                        const int* ysrc, const float* yalpha, int64_t frames, int64_t H, int64_t W, int64_t Hd, int64_t Wd,
                                        int mode, void* stream) {
    GF_CHECK_ARG(src && dst && xstart && xsrc && xalpha && ystart && ysrc && yalpha, "gf_resize_area_u8: null pointer");
    GF_CHECK_ARG(frames >= 0 && H > 0 && W > 0 && Hd > 0 && Wd > 0 && (mode == 0 || mode == 1), "gf_resize_area_u8: bad sizes / mode");
    if (frames == 0) return GF_OK;
    if (mode == 0)
        hipLaunchKernelGGL((area_u8_kernel<3, false>), dim3(cn_grid(frames * Hd * Wd)), dim3(CN_THREADS), 0, (hipStream_t)stream,
                           (const unsigned char*)src, dst, xstart, xsrc, xalpha, ystart, ysrc, yalpha, (int)frames, (int)H, (int)W, (int)Hd, (int)Wd);
    else
        hipLaunchKernelGGL((area_u8_kernel<1, true>), dim3(cn_grid(frames * Hd * Wd)), dim3(CN_THREADS), 0, (hipStream_t)stream,
                           (const unsigned char*)src, dst, xstart, xsrc, xalpha, ystart, ysrc, yalpha, (int)frames, (int)H, (int)W, (int)Hd, (int)Wd);
    GF_CHECK_LAUNCH("gf_resize_area_u8");
    return GF_OK;
}

extern "C" GF_API int gf_canny_u8(const void* img, void* state, void* mag_ws, void* dxy_ws, int* changed, int64_t frames, int64_t H, int64_t W,
                                  int low, int high, int max_passes, void* stream) {
    GF_CHECK_ARG(img && state && mag_ws && dxy_ws && changed, "gf_canny_u8: null pointer");
    GF_CHECK_ARG(frames >= 0 && H > 0 && W > 0 && max_passes > 0, "gf_canny_u8: bad sizes");
    if (frames == 0) return GF_OK;
    hipStream_t st = (hipStream_t)stream;
    const long n = frames * H * W;
    hipLaunchKernelGGL(canny_grad_kernel, dim3(cn_grid(n)), dim3(CN_THREADS), 0, st, (const unsigned char*)img, (int*)mag_ws, (int*)dxy_ws,
                       (int)frames, (int)H, (int)W);
    hipLaunchKernelGGL(canny_nms_kernel, dim3(cn_grid(n)), dim3(CN_THREADS), 0, st, (const int*)mag_ws, (const int*)dxy_ws,
                       (unsigned char*)state, (int)frames, (int)H, (int)W, low, high);
    GF_CHECK_LAUNCH("gf_canny_u8 (gradient / nms)");
    // hysteresis to the fixpoint: the flag is read back after every pass (a dataset loader, not the sampling loop: the sync is fine)
    const int tx = (int)((W + 31) / 32), ty = (int)((H + 31) / 32);
    for (int pass = 0; pass < max_passes; ++pass) {
        if (hipMemsetAsync(changed, 0, sizeof(int), st) != hipSuccess) {
            gf_set_error("gf_canny_u8: hipMemsetAsync failed");
            return GF_ERR_LAUNCH;
        }
        hipLaunchKernelGGL(canny_hyst_kernel, dim3((unsigned)(frames * tx * ty)), dim3(CN_THREADS), 0, st, (unsigned char*)state, changed,
                           (int)H, (int)W, tx, ty);
        GF_CHECK_LAUNCH("gf_canny_u8 (hysteresis)");
        int h = 0;
        if (hipMemcpyAsync(&h, changed, sizeof(int), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) {
            gf_set_error("gf_canny_u8: reading the hysteresis flag failed");
            return GF_ERR_LAUNCH;
        }
        if (!h) return GF_OK;
    }
    gf_set_error("gf_canny_u8: hysteresis did not reach its fixpoint in %d passes", max_passes);
    return GF_ERR_LAUNCH;
}
