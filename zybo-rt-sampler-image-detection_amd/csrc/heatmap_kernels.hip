// heatmap_kernels.hip -- power map -> colour heat-map -> upscaled, temporally blended overlay, all on the device.
//
// Reference (display side of the hot path): PC/src/visual.py
//   calculate_heatmap            :143-188  clip 1e-12, log10, subtract log10(min), divide by max, keep levels >= amount,
//                                          ((l - amount) / amount) ** exponent, int(255 * .) -> reversed-jet LUT,
//                                          written at [MAX_RES_Y-1-y, MAX_RES_X-1-x] (the flip), should_overlay = max > threshold
//   cv2.resize(..., INTER_LINEAR)  :186    uint8 bilinear upscale to the display size
//   cv2.addWeighted(prev,.5,new,.5):450    temporal blend, then addWeighted(frame, .9, res, .9) onto the camera frame :452
//   find_power_center            :295-322  5x5 Gaussian (sigma 1), >= 95 % mask, cube-weighted centroid
// cv2 is not available where this was built (and its version is not pinned by the reference), so resize / blend /
// blur follow OpenCV's documented uint8 algorithms (11-bit fixed-point bilinear weights, half-pixel centres,
// saturate_cast<uchar>(lrint(.)), BORDER_REFLECT_101): parity for those three is "unpinned" -- see DESIGN.md.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <stdint.h>

namespace bf {

namespace {

struct Rgb { unsigned char r, g, b; };
__constant__ Rgb kJet[256] = {
#include "jet_lut.inc"
};

__device__ __forceinline__ float block_reduce(float v, float* red, bool is_max)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    for (int off = 32; off > 0; off >>= 1) {
        const float o = __shfl_xor(v, off, 64);
        v = is_max ? fmaxf(v, o) : fminf(v, o);
    }
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float r = red[0];
    for (int w = 1; w < nw; ++w) r = is_max ? fmaxf(r, red[w]) : fminf(r, red[w]);
    return r;
}

// One workgroup per frame: statistics of the map, then the small colour image [res_y][res_x][3] (flipped, as the
// reference indexes it) and the should_overlay flag.
__global__ void __launch_bounds__(1024) colorize_kernel(const float* __restrict__ power, int res_x, int res_y, float threshold,
                                                       float amount, float exponent, unsigned char* __restrict__ small,
                                                       int* __restrict__ should_overlay)
{
    __shared__ float red[16];                 // (one slot per wave: up to 1024 threads)
    const int D = res_x * res_y;
    const float* img = power + (size_t)blockIdx.x * D;
    unsigned char* out = small + (size_t)blockIdx.x * D * 3;
    float vmax = -INFINITY, smin = INFINITY;
    for (int i = threadIdx.x; i < D; i += blockDim.x) {
        const float v = img[i];
        vmax = fmaxf(vmax, v);
        smin = fminf(smin, fmaxf(v, 1e-12f));
    }
    vmax = block_reduce(vmax, red, true);
    smin = block_reduce(smin, red, false);
    const bool overlay = vmax > threshold;
    const float lmin = log10f(smin);
    // max over the image of log10(clip(v)) - log10(min): log10 is monotonic, so it is log10(clip(max)) - lmin
    const float lmax = log10f(fmaxf(vmax, 1e-12f)) - lmin;
    if (threadIdx.x == 0) should_overlay[blockIdx.x] = overlay ? 1 : 0;
    for (int i = threadIdx.x; i < D; i += blockDim.x) {
        const int x = i / res_y, y = i - x * res_y;          // image[x, y], flat x*res_y + y
        Rgb c{0, 0, 0};
        if (overlay) {
            float l = (log10f(fmaxf(img[i], 1e-12f)) - lmin) / lmax;
            if (l >= amount) {
                l = (l - amount) / amount;
                int cv = (int)(255.0f * powf(l, exponent));
                cv = cv < 0 ? 0 : cv > 255 ? 255 : cv;
                c = kJet[cv];
            }
        }
        unsigned char* p = out + ((size_t)(res_y - 1 - y) * res_x + (res_x - 1 - x)) * 3;
        p[0] = c.r; p[1] = c.g; p[2] = c.b;
    }
}

__device__ __forceinline__ unsigned char sat_u8(float v)
{
    const float r = rintf(v);   // cv::saturate_cast<uchar>(double) rounds to nearest even
    return (unsigned char)(r < 0.f ? 0.f : r > 255.f ? 255.f : r);
}

// One thread per output pixel (all 3 channels); the frames of the batch are walked in order because the temporal
// blend is a recurrence: res_f = sat(0.5 prev + 0.5 new_f), prev = res_f.
__global__ void __launch_bounds__(256) overlay_kernel(const unsigned char* __restrict__ small, int frames, int sw, int sh, int ow, int oh,
                                                      unsigned char* __restrict__ prev, const unsigned char* __restrict__ camera,
                                                      unsigned char* __restrict__ out, float w_prev, float w_new, float w_cam, float w_heat)
{
    const int px = blockIdx.x * blockDim.x + threadIdx.x;
    if (px >= ow * oh) return;
    const int oy = px / ow, ox = px - oy * ow;
    // cv2.resize INTER_LINEAR, 8-bit: fx = (ox + 0.5) * sw/ow - 0.5, clamp, weights in 11-bit fixed point
    auto coord = [](int o, int ssz, int dsz, int& i0, int& i1, int& w0, int& w1) {
        float f = (float)((o + 0.5) * ((double)ssz / dsz) - 0.5);
        int i = (int)floorf(f);
        f -= i;
        if (i < 0) { i = 0; f = 0.f; }
        if (i >= ssz - 1) { i = ssz - 1; f = 0.f; i1 = i; } else i1 = i + 1;
        i0 = i;
        w0 = (int)rintf((1.0f - f) * 2048.0f);   // each weight is rounded on its own (saturate_cast<short>)
        w1 = (int)rintf(f * 2048.0f);
    };
    int x0, x1, wx0, wx1, y0, y1, wy0, wy1;
    coord(ox, sw, ow, x0, x1, wx0, wx1);
    coord(oy, sh, oh, y0, y1, wy0, wy1);
    float pv[3];
    const size_t o3 = (size_t)px * 3;
    for (int ch = 0; ch < 3; ++ch) pv[ch] = prev[o3 + ch];
    for (int f = 0; f < frames; ++f) {
        const unsigned char* s = small + (size_t)f * sw * sh * 3;
        for (int ch = 0; ch < 3; ++ch) {
            const int a = s[((size_t)y0 * sw + x0) * 3 + ch], b = s[((size_t)y0 * sw + x1) * 3 + ch];
            const int c = s[((size_t)y1 * sw + x0) * 3 + ch], d = s[((size_t)y1 * sw + x1) * 3 + ch];
            const int up = (((wy0 * ((a * wx0 + b * wx1) >> 4)) >> 16) + ((wy1 * ((c * wx0 + d * wx1) >> 4)) >> 16) + 2) >> 2;   // OpenCV's VResizeLinear<uchar>
            const unsigned char res = sat_u8(w_prev * pv[ch] + w_new * (float)up);
            pv[ch] = res;
            const size_t oi = ((size_t)f * ow * oh + px) * 3 + ch;
            out[oi] = camera ? sat_u8(w_cam * (float)camera[oi] + w_heat * (float)res) : res;
        }
    }
    for (int ch = 0; ch < 3; ++ch) prev[o3 + ch] = (unsigned char)pv[ch];
}

// The same recurrence as a tiled kernel: a workgroup owns a 128 x 8 tile of the output (256 threads x 4 pixels of a row) and first copies the few source
// pixels the tile's bilinear taps touch -- for EVERY frame of the chunk -- into LDS as one dword per pixel (101 -> 640 upscaling: 23 x 4 source pixels per
// frame, 23 KB for 64 frames).  The frame loop then reads its taps from LDS (16 ds_read_b32 instead of 48 global byte loads per thread and frame), moves
// camera / output / prev as three dwords per thread instead of nine single bytes, and has the camera words of the next kOvAhead frames in flight while it
// blends (the loop is a recurrence per pixel, so without that every frame waits for an HBM round trip).  The one-pixel kernel above took 0.19 ms for 64
// frames of 640 x 640; the bytes alone are 0.03.  Needs out_w % 4 == 0 and 4-byte aligned buffers (launch_overlay checks and falls back to the kernel above);
// identical arithmetic, identical results.
constexpr int kOvTW = 128, kOvTH = 8, kOvAhead = 8, kOvLdsBytes = 48 * 1024;

__global__ void __launch_bounds__(256) overlay_tile_kernel(const unsigned char* __restrict__ small, int frames, int sw, int sh, int ow, int oh,
                                                           unsigned char* __restrict__ prev, const unsigned char* __restrict__ camera,
                                                           unsigned char* __restrict__ out, float w_prev, float w_new, float w_cam, float w_heat,
                                                           int bx, int by, int chunk)
{
    extern __shared__ unsigned ov_lds[];                          // [chunk][by][bx] source pixels, r | g << 8 | b << 16
    auto coord = [](int o, int ssz, int dsz, int& i0, int& i1, int& w0, int& w1) {
        float f = (float)((o + 0.5) * ((double)ssz / dsz) - 0.5);
        int i = (int)floorf(f);
        f -= i;
        if (i < 0) { i = 0; f = 0.f; }
        if (i >= ssz - 1) { i = ssz - 1; f = 0.f; i1 = i; } else i1 = i + 1;
        i0 = i;
        w0 = (int)rintf((1.0f - f) * 2048.0f);
        w1 = (int)rintf(f * 2048.0f);
    };
    const int tx0 = blockIdx.x * kOvTW, ty0 = blockIdx.y * kOvTH;
    int sx_lo, sy_lo, t0, t1, t2;
    coord(tx0, sw, ow, sx_lo, t0, t1, t2);                        // the tile's first tap column / row (coord is monotonic in o)
    coord(ty0, sh, oh, sy_lo, t0, t1, t2);
    const int ox = tx0 + (threadIdx.x & 31) * 4, oy = ty0 + (threadIdx.x >> 5);
    const bool live = ox < ow && oy < oh;                         // ow % 4 == 0: a group of four is inside or outside as a whole
    int x0[4], x1[4], wx0[4], wx1[4], y0 = 0, y1 = 0, wy0 = 0, wy1 = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) coord(min(ox + i, ow - 1), sw, ow, x0[i], x1[i], wx0[i], wx1[i]);
    coord(min(oy, oh - 1), sh, oh, y0, y1, wy0, wy1);
    const int P = bx * by;
    int ta[4], tb[4];                                             // LDS word of the (y0, x0) / (y0, x1) tap; the y1 row is dy words further
#pragma unroll
    for (int i = 0; i < 4; ++i) { ta[i] = (y0 - sy_lo) * bx + (x0[i] - sx_lo); tb[i] = (y0 - sy_lo) * bx + (x1[i] - sx_lo); }
    const int dy = (y1 - y0) * bx;
    const size_t o3 = ((size_t)oy * ow + ox) * 3, fstride = (size_t)ow * oh * 3;
    union B12 { unsigned u[3]; unsigned char b[12]; };
    float pv[12];
    if (live) {
        B12 pw;
#pragma unroll
        for (int k = 0; k < 3; ++k) pw.u[k] = reinterpret_cast<const unsigned*>(prev + o3)[k];
#pragma unroll
        for (int k = 0; k < 12; ++k) pv[k] = pw.b[k];
    }
    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
    u16x2 wxp[4];                                                 // (wx0, wx1) as the second operand of v_dot2_u32_u16
#pragma unroll
    for (int i = 0; i < 4; ++i) wxp[i] = u16x2{(unsigned short)wx0[i], (unsigned short)wx1[i]};
    // sat_u8 without leaving float: rint, clamp (the operands are finite), and the byte is packed by v_cvt_pk_u8_f32 from an exact integer
    auto blend = [](float wa, float a, float wb, float b) { return __builtin_amdgcn_fmed3f(rintf(wa * a + wb * b), 0.f, 255.f); };
    for (int f0 = 0; f0 < frames; f0 += chunk) {
        const int nf = min(chunk, frames - f0);
        __syncthreads();                                          // the previous chunk's taps have been read
        for (int r = threadIdx.x; r < P; r += 256) {
            const int sy = r / bx, sx = r - sy * bx;
            const unsigned char* g = small + ((size_t)f0 * sh * sw + (size_t)min(sy_lo + sy, sh - 1) * sw + min(sx_lo + sx, sw - 1)) * 3;
#pragma unroll 8
            for (int f = 0; f < nf; ++f, g += (size_t)sh * sw * 3) ov_lds[f * P + r] = (unsigned)g[0] | ((unsigned)g[1] << 8) | ((unsigned)g[2] << 16);
        }
        __syncthreads();
        if (!live) continue;
        const unsigned char* cam_p = camera ? camera + (size_t)f0 * fstride + o3 : nullptr;
        unsigned char* out_p = out + (size_t)f0 * fstride + o3;
        for (int g0 = 0; g0 < nf; g0 += kOvAhead) {
            unsigned cam[kOvAhead][3];
            if (camera) {
#pragma unroll
                for (int u = 0; u < kOvAhead; ++u, cam_p += fstride)
                    if (g0 + u < nf) {
#pragma unroll
                        for (int k = 0; k < 3; ++k) cam[u][k] = reinterpret_cast<const unsigned*>(cam_p)[k];
                    }
            }
#pragma unroll
            for (int u = 0; u < kOvAhead; ++u, out_p += fstride) {
                if (g0 + u >= nf) break;
                const unsigned* s = ov_lds + (g0 + u) * P;
                unsigned res[3] = {0u, 0u, 0u};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const unsigned a4 = s[ta[i]], b4 = s[tb[i]], c4 = s[ta[i] + dy], d4 = s[tb[i] + dy];
#pragma unroll
                    for (int ch = 0; ch < 3; ++ch) {
                        const int e = 3 * i + ch;
                        const unsigned sel = 0x0c000c00u | ((4u + ch) << 16) | (unsigned)ch;          // (a | b << 16) of channel ch
                        const unsigned h0 = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, __builtin_amdgcn_perm(b4, a4, sel)), wxp[i], 0u, false);
                        const unsigned h1 = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, __builtin_amdgcn_perm(d4, c4, sel)), wxp[i], 0u, false);
                        const unsigned up = ((__umul24(wy0, h0 >> 4) >> 16) + (__umul24(wy1, h1 >> 4) >> 16) + 2u) >> 2;
                        const float r = blend(w_prev, pv[e], w_new, (float)up);
                        pv[e] = r;
                        const float o = camera ? blend(w_cam, (float)((cam[u][e >> 2] >> (8 * (e & 3))) & 255u), w_heat, r) : r;
                        res[e >> 2] = __builtin_amdgcn_cvt_pk_u8_f32(o, e & 3, res[e >> 2]);
                    }
                }
#pragma unroll
                for (int k = 0; k < 3; ++k) reinterpret_cast<unsigned*>(out_p)[k] = res[k];
            }
        }
    }
    if (live) {
        B12 pw;
#pragma unroll
        for (int k = 0; k < 12; ++k) pw.b[k] = (unsigned char)pv[k];
#pragma unroll
        for (int k = 0; k < 3; ++k) reinterpret_cast<unsigned*>(prev + o3)[k] = pw.u[k];
    }
}

// The detector's letterbox (what ultralytics' predict does to a frame before the network, yolo_smooth_tracking.py:13-23): the frame resized with
// cv2.resize(INTER_LINEAR) to new_w x new_h -- the same 8-bit fixed-point bilinear as above -- centred in an out_w x out_h canvas of the border value.
// One thread per output pixel, 3 channels.
__global__ void __launch_bounds__(256) letterbox_kernel(const unsigned char* __restrict__ src, int sh, int sw, unsigned char* __restrict__ out, int oh, int ow, int new_h,
                                                        int new_w, int top, int left, int value)
{
    const int px = blockIdx.x * blockDim.x + threadIdx.x;
    if (px >= ow * oh) return;
    const int oy = px / ow, ox = px - oy * ow;
    unsigned char* o = out + (size_t)px * 3;
    const int y = oy - top, x = ox - left;
    if (y < 0 || y >= new_h || x < 0 || x >= new_w) { o[0] = o[1] = o[2] = (unsigned char)value; return; }
    if (new_h == sh && new_w == sw) {                          // (no resampling: cv2.resize is skipped for equal shapes)
        const unsigned char* s = src + ((size_t)y * sw + x) * 3;
        o[0] = s[0]; o[1] = s[1]; o[2] = s[2];
        return;
    }
    auto coord = [](int d, int ssz, int dsz, int& i0, int& i1, int& w0, int& w1) {
        float f = (float)((d + 0.5) * ((double)ssz / dsz) - 0.5);
        int i = (int)floorf(f);
        f -= i;
        if (i < 0) { i = 0; f = 0.f; }
        if (i >= ssz - 1) { i = ssz - 1; f = 0.f; i1 = i; } else i1 = i + 1;
        i0 = i;
        w0 = (int)rintf((1.0f - f) * 2048.0f);
        w1 = (int)rintf(f * 2048.0f);
    };
    int x0, x1, wx0, wx1, y0, y1, wy0, wy1;
    coord(x, sw, new_w, x0, x1, wx0, wx1);
    coord(y, sh, new_h, y0, y1, wy0, wy1);
    for (int ch = 0; ch < 3; ++ch) {
        const int a = src[((size_t)y0 * sw + x0) * 3 + ch], b = src[((size_t)y0 * sw + x1) * 3 + ch];
        const int c = src[((size_t)y1 * sw + x0) * 3 + ch], d = src[((size_t)y1 * sw + x1) * 3 + ch];
        o[ch] = (unsigned char)((((wy0 * ((a * wx0 + b * wx1) >> 4)) >> 16) + ((wy1 * ((c * wx0 + d * wx1) >> 4)) >> 16) + 2) >> 2);
    }
}

// find_power_center: one workgroup per frame.  image[x][y] float32 (rows = x).  Returns (center_x, center_y) in the
// reference's naming: centroid over columns then rows of the smoothed map.
__global__ void __launch_bounds__(256) power_center_kernel(const float* __restrict__ power, int rows, int cols, float* __restrict__ centers,
                                                           float* __restrict__ smooth_ws)
{
    __shared__ float red[8];
    const int D = rows * cols;
    const float* img = power + (size_t)blockIdx.x * D;
    float* sm = smooth_ws + (size_t)blockIdx.x * D;
    // cv2.getGaussianKernel(5, 1.0): exp(-(i-2)^2/2) normalised
    const float g0 = 0.05448868f, g1 = 0.24420134f, g2 = 0.40261996f;
    const float gk[5] = {g0, g1, g2, g1, g0};
    auto refl = [](int i, int n) { if (n == 1) return 0; while (i < 0 || i >= n) { i = i < 0 ? -i : 2 * (n - 1) - i; } return i; };   // BORDER_REFLECT_101
    float vmax = -INFINITY;
    for (int i = threadIdx.x; i < D; i += blockDim.x) {
        const int r = i / cols, c = i - r * cols;
        float acc = 0.f;
        for (int dr = -2; dr <= 2; ++dr) {
            float rowacc = 0.f;
            const int rr = refl(r + dr, rows);
            for (int dc = -2; dc <= 2; ++dc) rowacc += gk[dc + 2] * fmaxf(img[rr * cols + refl(c + dc, cols)], 1e-12f);
            acc += gk[dr + 2] * rowacc;
        }
        sm[i] = acc;
        vmax = fmaxf(vmax, acc);
    }
    vmax = block_reduce(vmax, red, true);
    const float thr = vmax * 0.95f;
    // weighted centroid in float64 like NumPy's float32*bool -> float32 weights, int64 indices * float32 -> float64 sums
    double sw = 0.0, sx = 0.0, sy = 0.0;
    for (int i = threadIdx.x; i < D; i += blockDim.x) {
        const float v = sm[i];
        if (v >= thr) {
            const float w = v * v * v;
            const int r = i / cols, c = i - r * cols;
            sw += (double)w; sx += (double)c * (double)w; sy += (double)r * (double)w;
        }
    }
    __shared__ double dred[3][4];
    for (int off = 32; off > 0; off >>= 1) { sw += __shfl_xor(sw, off, 64); sx += __shfl_xor(sx, off, 64); sy += __shfl_xor(sy, off, 64); }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { dred[0][threadIdx.x >> 6] = sw; dred[1][threadIdx.x >> 6] = sx; dred[2][threadIdx.x >> 6] = sy; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = 0, b = 0, c = 0;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) { a += dred[0][w]; b += dred[1][w]; c += dred[2][w]; }
        centers[blockIdx.x * 2 + 0] = (float)(b / a);
        centers[blockIdx.x * 2 + 1] = (float)(c / a);
    }
}

}  // namespace

hipError_t launch_colorize(const float* d_power, int frames, int res_x, int res_y, float threshold, float amount, float exponent,
                           unsigned char* d_small, int* d_overlay, hipStream_t stream)
{
    // one workgroup per frame (two reductions over the map, then the mapping): 1024 threads -- a batch of 64 frames occupies 64 CUs either way, and the
    // log10 / pow per pixel is what a frame's workgroup spends its time on (51 -> 26 us per 64 frames of 101 x 101)
    const int threads = (long long)res_x * res_y >= 4096 ? 1024 : 256;
    hipLaunchKernelGGL(colorize_kernel, dim3(frames), dim3(threads), 0, stream, d_power, res_x, res_y, threshold, amount, exponent, d_small, d_overlay);
    return hipGetLastError();
}

hipError_t launch_overlay(const unsigned char* d_small, int frames, int small_w, int small_h, int out_w, int out_h, unsigned char* d_prev,
                          const unsigned char* d_camera, unsigned char* d_out, float w_prev, float w_new, float w_cam, float w_heat,
                          hipStream_t stream)
{
    const int px = out_w * out_h;
    // source pixels under one tile: a tile spans at most ceil(tile * scale) source steps, plus the tap to the right / below and one for the rounding of the first
    const int bx = (int)std::ceil((double)kOvTW * small_w / out_w) + 2, by = (int)std::ceil((double)kOvTH * small_h / out_h) + 2;
    const int chunk = std::min(frames, kOvLdsBytes / (bx * by * 4));
    const bool tiled = (out_w & 3) == 0 && chunk >= std::min(frames, 8) &&
                       ((reinterpret_cast<uintptr_t>(d_prev) | reinterpret_cast<uintptr_t>(d_out) | reinterpret_cast<uintptr_t>(d_camera)) & 3) == 0;
    if (tiled)
        hipLaunchKernelGGL(overlay_tile_kernel, dim3((out_w + kOvTW - 1) / kOvTW, (out_h + kOvTH - 1) / kOvTH), dim3(256), (size_t)chunk * bx * by * 4, stream, d_small,
                           frames, small_w, small_h, out_w, out_h, d_prev, d_camera, d_out, w_prev, w_new, w_cam, w_heat, bx, by, chunk);
    else
        hipLaunchKernelGGL(overlay_kernel, dim3((px + 255) / 256), dim3(256), 0, stream, d_small, frames, small_w, small_h, out_w, out_h, d_prev,
                           d_camera, d_out, w_prev, w_new, w_cam, w_heat);
    return hipGetLastError();
}

hipError_t launch_letterbox(const unsigned char* d_src, int sh, int sw, unsigned char* d_out, int oh, int ow, int new_h, int new_w, int top, int left, int value,
                            hipStream_t stream)
{
    if (sh < 1 || sw < 1 || oh < 1 || ow < 1 || new_h < 1 || new_w < 1 || top < 0 || left < 0 || top + new_h > oh || left + new_w > ow) return hipErrorInvalidValue;
    hipLaunchKernelGGL(letterbox_kernel, dim3((unsigned)((oh * ow + 255) / 256)), dim3(256), 0, stream, d_src, sh, sw, d_out, oh, ow, new_h, new_w, top, left, value);
    return hipGetLastError();
}

hipError_t launch_power_center(const float* d_power, int frames, int rows, int cols, float* d_centers, float* d_workspace, hipStream_t stream)
{
    hipLaunchKernelGGL(power_center_kernel, dim3(frames), dim3(256), 0, stream, d_power, rows, cols, d_centers, d_workspace);
    return hipGetLastError();
}

}  // namespace bf
