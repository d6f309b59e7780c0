// nms_kernels.hip -- detector post-processing on the device: anchor decode + confidence filter, then greedy NMS.
//
// The reference calls `ultralytics.YOLO(path).predict(frame)` (image-detection/src/yolo_smooth_tracking.py:9-23); the
// arithmetic lives in that third-party package (unpinned, weights missing -- SURVEY.md fact 2), so this is the published
// YOLOv5 head decode and non-maximum suppression, restated:
//   decode  y = sigmoid(raw);  xy = (y[0:2] * 2 - 0.5 + grid) * stride;  wh = (y[2:4] * 2)^2 * anchor;  score = obj * max(cls)
//   nms     candidates sorted by score; a box is kept unless an earlier KEPT box overlaps it with IoU > iou_thres
// Three kernels: decode (one thread per anchor box, all three scales), overlap bit-matrix (64 x 64 tiles, one
// 64-bit word per (row, column-block)), and a one-wave scan that walks the sorted list and emits the kept boxes.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

namespace bf {

namespace {

struct Level { const void* raw; int h, w, stride; float aw[3], ah[3]; long long box0; };
struct DecodeArgs { Level lv[3]; int n_levels, batch, nc, total, is_half; float conf_thres; float* boxes; float* scores; int* cls; };

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ float ld(const void* p, size_t i, int is_half) { return is_half ? __half2float(static_cast<const __half*>(p)[i]) : static_cast<const float*>(p)[i]; }

// raw head output of a level: [B][3*(5+nc)][H][W] (NCHW, as the 1x1 detect convs produce it)
__global__ void __launch_bounds__(256) decode_kernel(DecodeArgs a)
{
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (long long)a.batch * a.total) return;
    const int b = (int)(gid / a.total);
    long long r = gid - (long long)b * a.total;
    int l = 0;
    while (l + 1 < a.n_levels && r >= a.lv[l + 1].box0) ++l;
    const Level& L = a.lv[l];
    r -= L.box0;
    const int hw = L.h * L.w, anchor = (int)(r / hw), cell = (int)(r - (long long)anchor * hw), gy = cell / L.w, gx = cell - gy * L.w;
    const int no = 5 + a.nc;
    const size_t base = ((size_t)b * 3 * no + (size_t)anchor * no) * hw + cell;
    const float tx = sigmoidf_(ld(L.raw, base + 0 * (size_t)hw, a.is_half)), ty = sigmoidf_(ld(L.raw, base + 1 * (size_t)hw, a.is_half));
    const float tw = sigmoidf_(ld(L.raw, base + 2 * (size_t)hw, a.is_half)), th = sigmoidf_(ld(L.raw, base + 3 * (size_t)hw, a.is_half));
    const float obj = sigmoidf_(ld(L.raw, base + 4 * (size_t)hw, a.is_half));
    float best = 0.0f; int bc = 0;
    for (int c = 0; c < a.nc; ++c) {
        const float p = sigmoidf_(ld(L.raw, base + (size_t)(5 + c) * hw, a.is_half));
        if (p > best) { best = p; bc = c; }
    }
    const float cx = (tx * 2.0f - 0.5f + (float)gx) * (float)L.stride, cy = (ty * 2.0f - 0.5f + (float)gy) * (float)L.stride;
    const float w = (tw * 2.0f) * (tw * 2.0f) * L.aw[anchor], h = (th * 2.0f) * (th * 2.0f) * L.ah[anchor];
    const float score = obj * best;
    float* o = a.boxes + (size_t)gid * 4;
    o[0] = cx - w * 0.5f; o[1] = cy - h * 0.5f; o[2] = cx + w * 0.5f; o[3] = cy + h * 0.5f;
    a.scores[gid] = (obj > a.conf_thres && score > a.conf_thres) ? score : -1.0f;   // ultralytics filters on obj, then on obj*cls
    a.cls[gid] = bc;
}

__device__ __forceinline__ float iou(const float4 a, const float4 b)
{
    const float iw = fminf(a.z, b.z) - fmaxf(a.x, b.x), ih = fminf(a.w, b.w) - fmaxf(a.y, b.y);
    const float inter = fmaxf(iw, 0.0f) * fmaxf(ih, 0.0f);
    const float uni = (a.z - a.x) * (a.w - a.y) + (b.z - b.x) * (b.w - b.y) - inter;
    return inter / (uni + 1e-7f);     // box_iou of YOLOv5's utils (eps in the denominator)
}

// boxes: [B][K][4] sorted by descending score; counts[b] = valid entries.  mask[b][i][j/64] bit j%64 = IoU(i, j) > thr, j > i
__global__ void __launch_bounds__(64) nms_mask_kernel(const float* __restrict__ boxes, const int* __restrict__ counts, int K, float thr,
                                                      unsigned long long* __restrict__ mask)
{
    const int b = blockIdx.z, rb = blockIdx.y, cb = blockIdx.x, n = counts[b];
    const int words = (K + 63) / 64;
    if (rb * 64 >= n || cb * 64 >= n || cb < rb) {
        const int i = rb * 64 + threadIdx.x;
        if (i < K) mask[((size_t)b * K + i) * words + cb] = 0ull;
        return;
    }
    __shared__ float4 col[64];
    const float4* bx = reinterpret_cast<const float4*>(boxes) + (size_t)b * K;
    const int j = cb * 64 + threadIdx.x;
    if (j < n) col[threadIdx.x] = bx[j];
    __syncthreads();
    const int i = rb * 64 + threadIdx.x;
    unsigned long long bits = 0ull;
    if (i < n) {
        const float4 me = bx[i];
        const int jmax = min(64, n - cb * 64);
        for (int t = (rb == cb ? threadIdx.x + 1 : 0); t < jmax; ++t)
            if (iou(me, col[t]) > thr) bits |= 1ull << t;
    }
    if (i < K) mask[((size_t)b * K + i) * words + cb] = bits;
}

// one wave per image: walk the sorted candidates, keep those not suppressed by an earlier kept box
__global__ void __launch_bounds__(64) nms_scan_kernel(const float* __restrict__ boxes, const float* __restrict__ scores, const int* __restrict__ cls,
                                                      const int* __restrict__ counts, const unsigned long long* __restrict__ mask, int K, int max_det,
                                                      float* __restrict__ out, int* __restrict__ out_count)
{
    const int b = blockIdx.x, lane = threadIdx.x, n = counts[b];
    const int words = (K + 63) / 64;               // <= 64: lane w owns word w of the `removed` bit set
    unsigned long long removed = 0ull;
    int kept = 0;
    for (int i = 0; i < n && kept < max_det; ++i) {
        const unsigned long long w = __shfl(removed, i >> 6, 64);
        if ((w >> (i & 63)) & 1ull) continue;       // wave-uniform
        if (lane < words) removed |= mask[((size_t)b * K + i) * words + lane];
        if (lane < 4) out[((size_t)b * max_det + kept) * 6 + lane] = boxes[((size_t)b * K + i) * 4 + lane];
        if (lane == 4) out[((size_t)b * max_det + kept) * 6 + 4] = scores[(size_t)b * K + i];
        if (lane == 5) out[((size_t)b * max_det + kept) * 6 + 5] = (float)cls[(size_t)b * K + i];
        ++kept;
    }
    if (lane == 0) out_count[b] = kept;
}

}  // namespace

hipError_t launch_yolo_decode(const void* const raw[3], const int hs[3], const int ws[3], const int strides[3], const float* anchors /*[3][3][2]*/,
                              int batch, int nc, int is_half, float conf_thres, float* d_boxes, float* d_scores, int* d_cls, hipStream_t stream)
{
    DecodeArgs a{};
    long long off = 0;
    for (int l = 0; l < 3; ++l) {
        a.lv[l].raw = raw[l]; a.lv[l].h = hs[l]; a.lv[l].w = ws[l]; a.lv[l].stride = strides[l]; a.lv[l].box0 = off;
        for (int k = 0; k < 3; ++k) { a.lv[l].aw[k] = anchors[(l * 3 + k) * 2 + 0]; a.lv[l].ah[k] = anchors[(l * 3 + k) * 2 + 1]; }
        off += 3LL * hs[l] * ws[l];
    }
    a.n_levels = 3; a.batch = batch; a.nc = nc; a.total = (int)off; a.is_half = is_half; a.conf_thres = conf_thres;
    a.boxes = d_boxes; a.scores = d_scores; a.cls = d_cls;
    const long long n = (long long)batch * off;
    hipLaunchKernelGGL(decode_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_nms(const float* d_boxes, const float* d_scores, const int* d_cls, const int* d_counts, int batch, int K, float iou_thres,
                      int max_det, unsigned long long* d_mask, float* d_out, int* d_out_count, hipStream_t stream)
{
    if (K > 4096) return hipErrorInvalidValue;
    const int blocks = (K + 63) / 64;
    hipLaunchKernelGGL(nms_mask_kernel, dim3(blocks, blocks, batch), dim3(64), 0, stream, d_boxes, d_counts, K, iou_thres, d_mask);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(nms_scan_kernel, dim3(batch), dim3(64), 0, stream, d_boxes, d_scores, d_cls, d_counts, d_mask, K, max_det, d_out, d_out_count);
    return hipGetLastError();
}

}  // namespace bf
