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
struct DecodeArgs { Level lv[3]; int n_levels, batch, nc, total, is_half, nhwc; float conf_thres; float* boxes; float* scores; int* cls; };

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ float ld(const void* p, size_t i, int is_half) { return is_half ? __half2float(static_cast<const __half*>(p)[i]) : static_cast<const float*>(p)[i]; }

// raw head output of a level: [B][3*(5+nc)][H][W] (NCHW), or with a.nhwc [B][H][W][3*(5+nc)] (the channels_last memory the detect
// convolutions of csrc/conv_kernels.hip write: no relayout pass in between)
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
    const size_t base = a.nhwc ? ((size_t)b * hw + cell) * (3 * no) + (size_t)anchor * no : ((size_t)b * 3 * no + (size_t)anchor * no) * hw + cell;
    const size_t ch = a.nhwc ? 1 : (size_t)hw;                 // elements between consecutive channels of one box
    const float tx = sigmoidf_(ld(L.raw, base + 0 * ch, a.is_half)), ty = sigmoidf_(ld(L.raw, base + 1 * ch, a.is_half));
    const float tw = sigmoidf_(ld(L.raw, base + 2 * ch, a.is_half)), th = sigmoidf_(ld(L.raw, base + 3 * ch, a.is_half));
    const float obj = sigmoidf_(ld(L.raw, base + 4 * ch, a.is_half));
    float best = 0.0f; int bc = 0;
    for (int c = 0; c < a.nc; ++c) {
        const float p = sigmoidf_(ld(L.raw, base + (size_t)(5 + c) * ch, a.is_half));
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
    for (int j = kept * 6 + lane; j < max_det * 6; j += 64) out[(size_t)b * max_det * 6 + j] = 0.0f;      // rows past the kept boxes: zeros
    if (lane == 0) out_count[b] = kept;
}

// ---- candidate selection: the K best-scoring boxes of every image, in descending score order (ties: lower box index first),
// with their boxes and classes gathered -- what feeds the greedy pass.  One 1024-thread workgroup per image:
//   1. radix select (four 8-bit passes over a monotone integer image of the float scores) finds the K-th largest score;
//   2. boxes above it are compacted into LDS in any order, boxes equal to it are admitted in index order until K are in;
//   3. a bitonic sort of the (score key, ~index) pairs in LDS orders them;
//   4. scores / boxes / classes are gathered to the candidate arrays, counts[b] = candidates with a positive score.
// K <= 1024.  No index ever leaves [0, T): slots beyond the image's boxes are emitted as score -1.
__device__ __forceinline__ unsigned score_key(float x)
{
    unsigned u = __float_as_uint(x);
    if (x != x) return 0u;                                       // NaN: below everything
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);           // monotone: a > b  <=>  key(a) > key(b)
}

__global__ void __launch_bounds__(1024) topk_select_kernel(const float* __restrict__ scores, const float* __restrict__ boxes, const int* __restrict__ cls, int T,
                                                           int K, float* __restrict__ top_scores, float* __restrict__ top_boxes, int* __restrict__ top_cls,
                                                           int* __restrict__ counts)
{
    __shared__ unsigned hist[256];
    __shared__ unsigned long long cand[1024];
    __shared__ unsigned s_prefix, s_remaining, s_fill, s_running, s_wave_tot[16], s_positive;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* sc = scores + (size_t)b * T;
    const int keff = min(K, T);
    cand[tid] = 0ull;
    if (tid == 0) { s_prefix = 0u; s_remaining = (unsigned)keff; s_fill = 0u; s_running = 0u; s_positive = 0u; }
    __syncthreads();
    unsigned mask = 0u;
    for (int pass = 3; pass >= 0; --pass) {
        if (tid < 256) hist[tid] = 0u;
        __syncthreads();
        const unsigned prefix = s_prefix;
        for (int i = tid; i < T; i += 1024) {
            const unsigned k = score_key(sc[i]);
            if ((k & mask) == prefix) atomicAdd(&hist[(k >> (8 * pass)) & 255u], 1u);
        }
        __syncthreads();
        if (tid == 0) {
            unsigned rem = s_remaining, cum = 0u;
            int bin = 255;
            for (; bin > 0; --bin) {
                if (cum + hist[bin] >= rem) break;
                cum += hist[bin];
            }
            s_remaining = rem - cum;                             // how many of this bin's elements are still needed
            s_prefix = prefix | ((unsigned)bin << (8 * pass));
        }
        mask |= 0xFFu << (8 * pass);
        __syncthreads();
    }
    const unsigned kth = s_prefix, need_eq = s_remaining;       // the K-th largest key; how many boxes equal to it are admitted
    // boxes above the K-th: any order
    for (int i = tid; i < T; i += 1024) {
        const unsigned k = score_key(sc[i]);
        if (k > kth) {
            const unsigned slot = atomicAdd(&s_fill, 1u);
            cand[slot] = ((unsigned long long)k << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)i);
        }
    }
    __syncthreads();
    // boxes equal to the K-th: the first `need_eq` in index order (a workgroup-wide ordered count per 1024 boxes)
    const unsigned base_fill = s_fill;
    for (int i0 = 0; i0 < T; i0 += 1024) {
        if (s_running >= need_eq) break;                         // (uniform: s_running is read after the previous round's barrier)
        const int i = i0 + tid;
        const bool eq = i < T && score_key(sc[i]) == kth;
        const unsigned long long bal = __ballot(eq);
        if (lane == 0) s_wave_tot[wave] = (unsigned)__popcll(bal);
        __syncthreads();
        unsigned before = s_running;
        for (int w = 0; w < wave; ++w) before += s_wave_tot[w];
        const unsigned rank = before + (unsigned)__popcll(bal & ((1ull << lane) - 1ull));
        if (eq && rank < need_eq) cand[base_fill + rank] = ((unsigned long long)kth << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)i);
        __syncthreads();
        if (tid == 0) { unsigned tot = 0u; for (int w = 0; w < 16; ++w) tot += s_wave_tot[w]; s_running += tot; }
        __syncthreads();
    }
    __syncthreads();
    // bitonic sort, descending, of the 1024 pairs (empty slots are 0 = below every real pair: a real pair's low word is >= 1 only
    // if its index is < 2^32 - 1, and its high word is >= 1 unless the score is NaN or -inf... both sort last together with the padding)
    for (unsigned k2 = 2; k2 <= 1024; k2 <<= 1) {
        for (unsigned j = k2 >> 1; j > 0; j >>= 1) {
            const unsigned partner = tid ^ j;
            if (partner > (unsigned)tid) {
                const unsigned long long x = cand[tid], y = cand[partner];
                const bool desc = (tid & k2) == 0;
                if (desc ? x < y : x > y) { cand[tid] = y; cand[partner] = x; }
            }
            __syncthreads();
        }
    }
    if (tid < K) {
        const unsigned long long c = cand[tid];
        const bool real = tid < keff;
        const unsigned idx = real ? 0xFFFFFFFFu - (unsigned)(c & 0xFFFFFFFFull) : 0u;
        const bool ok = real && idx < (unsigned)T;
        const float s = ok ? sc[idx] : -1.0f;
        top_scores[(size_t)b * K + tid] = s;
        const float4 bx = ok ? reinterpret_cast<const float4*>(boxes)[(size_t)b * T + idx] : make_float4(0.f, 0.f, 0.f, 0.f);
        reinterpret_cast<float4*>(top_boxes)[(size_t)b * K + tid] = bx;
        top_cls[(size_t)b * K + tid] = ok ? cls[(size_t)b * T + idx] : 0;
        if (s > 0.0f) atomicAdd(&s_positive, 1u);
    }
    __syncthreads();
    if (tid == 0) counts[b] = (int)s_positive;
}

// The same selection for T <= 1024 * kPer boxes with the score keys held in registers (one global read of the scores instead of five), the
// lanes of a wave that share the first lane's bin folded into one LDS atomic (the boxes below the confidence threshold all carry the same score, so most
// of a wave lands on one address, and same-address atomics serialise), the 256-bin scan done by one wave with shuffles instead of one thread, and the bitonic network run in
// registers: the 45 of its 55 stages whose partner is within the wave are shuffles, the other 10 go through LDS.  The pairs are distinct (the index is
// part of them; the padding zeros are below every real pair), so the sorted order -- and with it every output -- equals topk_select_kernel's.
// 25,200 boxes x 64 images, K = 1024: 120 us -> 46 us.
#ifndef BF_TOPK_ROUNDS
#define BF_TOPK_ROUNDS 1     // measured 0 / 1 / 4 / 8 on one box: 108 / 46 / 57 / 57 us with 2 % of the boxes above the threshold, 58 / 64 / 92 / 113 us for uniform scores
#endif
template <int kPer>
__global__ void __launch_bounds__(1024) topk_select_reg_kernel(const float* __restrict__ scores, const float* __restrict__ boxes, const int* __restrict__ cls,
                                                               int T, int K, float* __restrict__ top_scores, float* __restrict__ top_boxes,
                                                               int* __restrict__ top_cls, int* __restrict__ counts)
{
    __shared__ unsigned hist[256];
    __shared__ unsigned long long cand[1024];
    __shared__ unsigned s_prefix, s_remaining, s_fill, s_running, s_wave_tot[16], s_positive;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* sc = scores + (size_t)b * T;
    const int keff = min(K, T);
    unsigned key[kPer];
#pragma unroll
    for (int j = 0; j < kPer; ++j) key[j] = j * 1024 + tid < T ? score_key(sc[j * 1024 + tid]) : 0u;
    cand[tid] = 0ull;
    if (tid == 0) { s_prefix = 0u; s_remaining = (unsigned)keff; s_fill = 0u; s_running = 0u; s_positive = 0u; }
    __syncthreads();
    unsigned mask = 0u;
    for (int pass = 3; pass >= 0; --pass) {
        if (tid < 256) hist[tid] = 0u;
        __syncthreads();
        const unsigned prefix = s_prefix;
#pragma unroll
        for (int j = 0; j < kPer; ++j) {
            const bool part = j * 1024 + tid < T && (key[j] & mask) == prefix;
            const unsigned bin = (key[j] >> (8 * pass)) & 255u;
            unsigned long long rest = __ballot(part);
            for (int round = 0; round < BF_TOPK_ROUNDS && rest != 0ull; ++round) {                // (wave-uniform) the lanes of the first open lane's bin: one atomic for all
                const int first = __ffsll((long long)rest) - 1;
                const unsigned fb = (unsigned)__builtin_amdgcn_readlane((int)bin, first);
                const unsigned long long same = __ballot(((rest >> lane) & 1ull) != 0ull && bin == fb);
                if (lane == first) atomicAdd(&hist[fb], (unsigned)__popcll(same));
                rest &= ~same;
            }
            if ((rest >> lane) & 1ull) atomicAdd(&hist[bin], 1u);                     // the other bins of the wave: one atomic per lane
        }
        __syncthreads();
        if (wave == 0) {
            // the serial rule "walk the bins from 255 down, stop at the first (and not below bin 1) whose running count reaches `rem`" on suffix sums
            // S(t) = hist[t] + ... + hist[255]: S falls with t, so the stop is the number of t in [1, 255] with S(t) >= rem, and the count taken
            // above it is S(stop + 1)
            const unsigned rem = s_remaining;
            const unsigned h0 = hist[4 * lane], h1 = hist[4 * lane + 1], h2 = hist[4 * lane + 2], h3 = hist[4 * lane + 3];
            unsigned v = h0 + h1 + h2 + h3;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const unsigned t = __shfl_down(v, off, 64);
                if (lane + off < 64) v += t;
            }
            const unsigned s3 = v - (h0 + h1 + h2), s2 = s3 + h2, s1 = s2 + h1, s0 = s1 + h0;     // S(4 lane + 3 .. 4 lane)
            unsigned c = (s3 >= rem) + (s2 >= rem) + (s1 >= rem) + (lane > 0 && s0 >= rem);
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off, 64);
            const int bin = (int)c, nb = bin + 1;                                     // S(nb) is what the bins above `bin` hold
            if (nb == 256) { if (lane == 0) s_remaining = rem; }
            else if ((nb >> 2) == lane) s_remaining = rem - ((nb & 3) == 0 ? s0 : (nb & 3) == 1 ? s1 : (nb & 3) == 2 ? s2 : s3);
            if (lane == 0) s_prefix = prefix | ((unsigned)bin << (8 * pass));
        }
        mask |= 0xFFu << (8 * pass);
        __syncthreads();
    }
    const unsigned kth = s_prefix, need_eq = s_remaining;
#pragma unroll
    for (int j = 0; j < kPer; ++j)
        if (j * 1024 + tid < T && key[j] > kth) {
            const unsigned slot = atomicAdd(&s_fill, 1u);
            cand[slot] = ((unsigned long long)key[j] << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)(j * 1024 + tid));
        }
    __syncthreads();
    const unsigned base_fill = s_fill;
#pragma unroll
    for (int j = 0; j < kPer; ++j) {
        if (j * 1024 < T && s_running < need_eq) {                                    // (uniform: s_running is read after the previous round's barrier)
            const int i = j * 1024 + tid;
            const bool eq = i < T && key[j] == kth;
            const unsigned long long bal = __ballot(eq);
            if (lane == 0) s_wave_tot[wave] = (unsigned)__popcll(bal);
            __syncthreads();
            unsigned before = s_running;
            for (int w = 0; w < wave; ++w) before += s_wave_tot[w];
            const unsigned rank = before + (unsigned)__popcll(bal & ((1ull << lane) - 1ull));
            if (eq && rank < need_eq) cand[base_fill + rank] = ((unsigned long long)kth << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)i);
            __syncthreads();
            if (tid == 0) { unsigned tot = 0u; for (int w = 0; w < 16; ++w) tot += s_wave_tot[w]; s_running += tot; }
            __syncthreads();
        }
    }
    __syncthreads();
    unsigned long long x = cand[tid];
    for (unsigned k2 = 2; k2 <= 1024; k2 <<= 1) {
        for (unsigned j = k2 >> 1; j > 0; j >>= 1) {
            unsigned long long y;
            if (j >= 64) {
                __syncthreads();                                                      // the previous LDS stage has been read
                cand[tid] = x;
                __syncthreads();
                y = cand[tid ^ j];
            } else {
                y = __shfl_xor(x, (int)j, 64);
            }
            const bool want_max = (((unsigned)tid & j) == 0) == (((unsigned)tid & k2) == 0);      // lower index of a descending pair, or upper of an ascending one
            x = want_max ? (x > y ? x : y) : (x < y ? x : y);
        }
    }
    if (tid < K) {
        const bool real = tid < keff;
        const unsigned idx = real ? 0xFFFFFFFFu - (unsigned)(x & 0xFFFFFFFFull) : 0u;
        const bool ok = real && idx < (unsigned)T;
        const float s = ok ? sc[idx] : -1.0f;
        top_scores[(size_t)b * K + tid] = s;
        const float4 bx = ok ? reinterpret_cast<const float4*>(boxes)[(size_t)b * T + idx] : make_float4(0.f, 0.f, 0.f, 0.f);
        reinterpret_cast<float4*>(top_boxes)[(size_t)b * K + tid] = bx;
        top_cls[(size_t)b * K + tid] = ok ? cls[(size_t)b * T + idx] : 0;
        const unsigned long long pos = __ballot(s > 0.0f);
        if (lane == 0 && pos) atomicAdd(&s_positive, (unsigned)__popcll(pos));
    }
    __syncthreads();
    if (tid == 0) counts[b] = (int)s_positive;
}

}  // namespace

hipError_t launch_topk_candidates(const float* d_scores, const float* d_boxes, const int* d_cls, int batch, int total, int K, float* d_top_scores,
                                  float* d_top_boxes, int* d_top_cls, int* d_counts, hipStream_t stream)
{
    if (K < 1 || K > 1024 || total < 1 || batch < 1) return hipErrorInvalidValue;
    auto go = [&](auto kernel) {
        hipLaunchKernelGGL(kernel, dim3((unsigned)batch), dim3(1024), 0, stream, d_scores, d_boxes, d_cls, total, K, d_top_scores, d_top_boxes, d_top_cls, d_counts);
    };
    if (total <= 8 * 1024) go(topk_select_reg_kernel<8>);
    else if (total <= 16 * 1024) go(topk_select_reg_kernel<16>);
    else if (total <= 32 * 1024) go(topk_select_reg_kernel<32>);        // 640 x 640: 25,200 boxes
    else go(topk_select_kernel);
    return hipGetLastError();
}

hipError_t launch_yolo_decode(const void* const raw[3], const int hs[3], const int ws[3], const int strides[3], const float* anchors /*[3][3][2]*/,
                              int batch, int nc, int format, float conf_thres, float* d_boxes, float* d_scores, int* d_cls, hipStream_t stream)
{
    // format: bit 0 = float16 maps (else float32), bit 1 = NHWC maps (else NCHW)
    DecodeArgs a{};
    long long off = 0;
    for (int l = 0; l < 3; ++l) {
        a.lv[l].raw = raw[l]; a.lv[l].h = hs[l]; a.lv[l].w = ws[l]; a.lv[l].stride = strides[l]; a.lv[l].box0 = off;
        for (int k = 0; k < 3; ++k) { a.lv[l].aw[k] = anchors[(l * 3 + k) * 2 + 0]; a.lv[l].ah[k] = anchors[(l * 3 + k) * 2 + 1]; }
        off += 3LL * hs[l] * ws[l];
    }
    a.n_levels = 3; a.batch = batch; a.nc = nc; a.total = (int)off; a.is_half = format & 1; a.nhwc = (format >> 1) & 1; a.conf_thres = conf_thres;
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
