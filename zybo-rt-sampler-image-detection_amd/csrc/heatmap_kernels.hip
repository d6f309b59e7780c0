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
__global__ void __launch_bounds__(256) colorize_kernel(const float* __restrict__ power, int res_x, int res_y, float threshold,
                                                       float amount, float exponent, unsigned char* __restrict__ small,
                                                       int* __restrict__ should_overlay)
{
    __shared__ float red[8];
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
    hipLaunchKernelGGL(colorize_kernel, dim3(frames), dim3(256), 0, stream, d_power, res_x, res_y, threshold, amount, exponent, d_small, d_overlay);
    return hipGetLastError();
}

hipError_t launch_overlay(const unsigned char* d_small, int frames, int small_w, int small_h, int out_w, int out_h, unsigned char* d_prev,
                          const unsigned char* d_camera, unsigned char* d_out, float w_prev, float w_new, float w_cam, float w_heat,
                          hipStream_t stream)
{
    const int px = out_w * out_h;
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
