// Convolution + bias + SiLU of the detector (image_detection/model/yolov5s.py) as one implicit-GEMM kernel on the f16 matrix cores.
//
// The reference calls `ultralytics.YOLO(...)` (image-detection/src/yolo_smooth_tracking.py:9-23); its network is the published
// YOLOv5s graph: 60 convolutions with 1x1, 3x3 and one 6x6 window, stride 1 or 2, each followed (BatchNorm folded) by a bias and SiLU,
// plus three plain 1x1 detect convolutions.  All of them are this kernel:
//
//   out[b, ho, wo, n] = act( bias[n] + sum_{kh, kw, c} in[b, ho s - p + kh, wo s - p + kw, c] * w[n, kh, kw, c] )
//
// as a GEMM  [M = B Ho Wo pixels] x [K = KH KW C] x [N = Cout]  on v_mfma_f32_32x32x16_f16 (f16 operands, f32 accumulation):
//   * activations are NHWC (torch channels_last), so for one kh the KW x C operands of an output pixel are ONE contiguous run of the
//     input row; the weights are packed [Cout][KH][KW][C], the same K order, each row zero-padded to whole stages.  C is a power of
//     two: from 8 up every 16-byte chunk of 8 halfs lies inside one input pixel; C = 4 (the 3-channel image padded by one) makes
//     a chunk two neighbouring pixels, which the 6x6 / stride 2 / pad 2 stem keeps inside or outside the image together (even
//     window start, even width).  Either way the zero padding of the window is a per-chunk mask;
//   * a workgroup of 4 waves owns a 128-pixel x 64-channel output tile; K advances 32 per stage through two LDS buffers (one barrier per
//     stage): every thread moves two 16-byte chunks of the pixel tile and one of the weight tile, global -> registers (issued a stage
//     ahead) -> LDS; rows are padded to 80 bytes, which spreads a wave's 16-byte fragment reads over all banks;
//   * a wave owns 32 pixels x 64 channels: per 16-deep step one A fragment and two B fragments (ds_read_b128) feed two MFMAs;
//   * the epilogue adds the bias, applies SiLU (x / (1 + exp(-x))), rounds to f16 and transposes the tile through LDS so that every
//     thread stores 16 contiguous bytes of a pixel's channels (the 1x1 layers are bound by these stores).  The output may be a
//     channel slice of a wider NHWC buffer (row stride ldy: the network's concatenations cost nothing) and a residual tensor may be
//     added to the rounded result (the bottlenecks' x + cv2(cv1(x)), added like torch adds two f16 tensors: in f32, rounded once).
// Parity: tests/test_detector.py compares against torch's fp32 convolution of the same f16 operands.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>

#include <cstdint>

#include "das_kernels.h"

namespace bf {
namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));

constexpr int kBM = 128, kBK = 32, kRow = kBK + 8;   // halfs per LDS row (80 bytes)
// kBN (template parameter): output channels per workgroup tile.
//   32: layers of up to 32 channels (the 6x6 stem among them); 64; 128 for layers of 128 channels and more -- there the four waves
//   form a 2 x 2 grid of 64-pixel x 64-channel tiles (two A and two B fragments feed four MFMAs: one LDS read per MFMA instead of 1.5,
//   a third less staging per MFMA), otherwise they stack 4 x 1 with 32-pixel x kBN tiles.

struct ConvArgs {
    int B, H, W, C, Ho, Wo, N, KH, KW, stride, pad, act;
    int ldy, ldr;         // halfs between consecutive pixels of the output / the residual
    int wide;
    const _Float16* res;  // optional residual, [M][ldr]
    int c_shift;          // log2(C)
    long long M;          // B * Ho * Wo
};

template <int kBN>
__global__ void __launch_bounds__(256) conv_igemm_kernel(const _Float16* __restrict__ x, const _Float16* __restrict__ w, const float* __restrict__ bias,
                                                         _Float16* __restrict__ y, ConvArgs a)
{
    constexpr int kWN = kBN == 128 ? 2 : 1, kWM = 4 / kWN;              // the waves' grid over the tile
    constexpr int kTM = kBM / kWM / 32, kTN = kBN / kWN / 32;           // 32 x 32 MFMA tiles per wave
    constexpr int kBRows = (kBN + 63) / 64;                             // weight rows a thread moves per stage
    __shared__ __attribute__((aligned(16))) _Float16 smem[2 * (kBM + kBN) * kRow];
    _Float16* const As = smem;                                          // [2][kBM][kRow]
    _Float16* const Bs = smem + 2 * kBM * kRow;                         // [2][kBN][kRow]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave % kWM, wn = wave / kWM;
    const long long m0 = (long long)blockIdx.x * kBM;
    const int n0 = blockIdx.y * kBN;
    // K is walked in 16-byte chunks of 8 halfs: chunk q = (kh, 8 (q % cpk) halfs into the kh run of KW * C); 4 chunks per stage
    const int cpk = (a.KW * a.C) >> 3, n_chunk = a.KH * cpk, n_stage = (n_chunk + 3) >> 2;

    // this thread's chunks: pixel rows r and r + 64, chunk `ck` of the stage; weight rows r (+ 64), same chunk
    const int r = tid >> 2, ck = tid & 3;
    int hi0[2], wi0[2];
    const _Float16* px[2];
    bool pv[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const long long m = m0 + r + 64 * i;
        pv[i] = m < a.M;
        const unsigned mm = pv[i] ? (unsigned)m : 0u;          // (M < 2^31: launch_conv2d_nhwc_f16)
        const unsigned b = mm / (unsigned)(a.Wo * a.Ho), rem = mm - b * (unsigned)(a.Wo * a.Ho);
        const unsigned ho = rem / (unsigned)a.Wo, wo = rem - ho * (unsigned)a.Wo;
        hi0[i] = (int)ho * a.stride - a.pad;
        wi0[i] = (int)wo * a.stride - a.pad;
        px[i] = x + (size_t)b * a.H * a.W * a.C;
    }
    bool wv[kBRows];
    const _Float16* wrow[kBRows];
#pragma unroll
    for (int i = 0; i < kBRows; ++i) {
        wv[i] = r + 64 * i < kBN && n0 + r + 64 * i < a.N;
        wrow[i] = w + (size_t)(wv[i] ? n0 + r + 64 * i : 0) * n_stage * kBK;      // weight rows are padded to whole stages
    }

    half8 ra[2], rb[kBRows];
    bool oka[2];          // (the zero padding is applied when the chunk goes to LDS: a select right behind the load would wait for it there)
    auto fetch = [&](int s) {
        const int q = 4 * s + ck;
        const int kh = q / cpk, kk = (q - kh * cpk) << 3;      // halfs into the kh run
        const int kw = kk >> a.c_shift, c = kk & (a.C - 1);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int hi = hi0[i] + kh, wi = wi0[i] + kw;
            const bool ok = pv[i] && q < n_chunk && (unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W;
            const size_t off = ok ? ((size_t)hi * a.W + wi) * a.C + c : 0;
            ra[i] = *reinterpret_cast<const half8*>(px[i] + off);
            oka[i] = ok;
        }
#pragma unroll
        for (int i = 0; i < kBRows; ++i) rb[i] = *reinterpret_cast<const half8*>(wrow[i] + (size_t)q * 8);
    };
    auto stash = [&](int buf) {
        const half8 z = {0, 0, 0, 0, 0, 0, 0, 0};
        *reinterpret_cast<half8*>(&As[(buf * kBM + r) * kRow + ck * 8]) = oka[0] ? ra[0] : z;
        *reinterpret_cast<half8*>(&As[(buf * kBM + r + 64) * kRow + ck * 8]) = oka[1] ? ra[1] : z;
#pragma unroll
        for (int i = 0; i < kBRows; ++i)
            if (r + 64 * i < kBN) *reinterpret_cast<half8*>(&Bs[(buf * kBN + r + 64 * i) * kRow + ck * 8]) = wv[i] ? rb[i] : z;
    };

    float16v acc[kTM][kTN];
#pragma unroll
    for (int i = 0; i < kTM; ++i)
#pragma unroll
        for (int t = 0; t < kTN; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][t][e] = 0.0f;

    fetch(0);
    stash(0);
    __syncthreads();
    for (int s = 0; s < n_stage; ++s) {
        const int buf = s & 1;
        if (s + 1 < n_stage) fetch(s + 1);                     // in flight under this stage's MFMAs
        const _Float16* A = &As[(buf * kBM + wm * 32 * kTM + (lane & 31)) * kRow + 8 * (lane >> 5)];
        const _Float16* Bp = &Bs[(buf * kBN + wn * 32 * kTN + (lane & 31)) * kRow + 8 * (lane >> 5)];
#pragma unroll
        for (int k16 = 0; k16 < kBK / 16; ++k16) {
            half8 af[kTM], bf[kTN];
#pragma unroll
            for (int i = 0; i < kTM; ++i) af[i] = *reinterpret_cast<const half8*>(A + 32 * i * kRow + 16 * k16);
#pragma unroll
            for (int t = 0; t < kTN; ++t) bf[t] = *reinterpret_cast<const half8*>(Bp + 32 * t * kRow + 16 * k16);
#pragma unroll
            for (int i = 0; i < kTM; ++i)
#pragma unroll
                for (int t = 0; t < kTN; ++t) acc[i][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i], bf[t], acc[i][t], 0, 0, 0);
        }
        if (s + 1 < n_stage) stash(buf ^ 1);                   // the other buffer: its readers passed the previous barrier
        __syncthreads();
    }

    // Epilogue.  C lane map of v_mfma_f32_32x32x16_f16: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).
    // The f16 tile goes through LDS (the stage buffers are free after the loop's last barrier): rows of kBN channels + 8.
    constexpr int kCRow = kBN + 8;
    _Float16* Cs = smem;
    static_assert(kBM * kCRow <= 2 * (kBM + kBN) * kRow, "the output tile fits the stage buffers");
#pragma unroll
    for (int t = 0; t < kTN; ++t) {
        const int col = wn * 32 * kTN + 32 * t + (lane & 31), n = n0 + col;
        const float bn = (bias && n < a.N) ? bias[n] : 0.0f;
#pragma unroll
        for (int i = 0; i < kTM; ++i) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                float v = acc[i][t][e] + bn;
                if (a.act) v = v * __builtin_amdgcn_rcpf(1.0f + __expf(-v));      // SiLU (1 ulp reciprocal, rounded to f16 next)
                Cs[(wm * 32 * kTM + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)) * kCRow + col] = (_Float16)v;
            }
        }
    }
    __syncthreads();
    const bool wide = a.wide != 0;                             // 16-byte stores need 16-byte rows and bases (launch_conv2d_nhwc_f16)
    constexpr int kCC = kBN / 8;                               // 16-byte chunks per tile row
#pragma unroll
    for (int j = 0; j < (kBM * kCC) / 256; ++j) {
        const int id = tid + 256 * j, row = id / kCC, cc = (id % kCC) * 8;
        const long long m = m0 + row;
        if (m >= a.M || n0 + cc >= a.N) continue;
        const _Float16* src = &Cs[row * kCRow + cc];
        _Float16* dst = y + (size_t)m * a.ldy + n0 + cc;
        if (wide) {
            half8 v = *reinterpret_cast<const half8*>(src);
            if (a.res) {
                const half8 rv = *reinterpret_cast<const half8*>(a.res + (size_t)m * a.ldr + n0 + cc);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = (_Float16)((float)v[e] + (float)rv[e]);
            }
            *reinterpret_cast<half8*>(dst) = v;
        } else {
            for (int e = 0; e < 8 && n0 + cc + e < a.N; ++e)
                dst[e] = a.res ? (_Float16)((float)src[e] + (float)a.res[(size_t)m * a.ldr + n0 + cc + e]) : src[e];
        }
    }
}

// Camera frames -> network input (yolo_smooth_tracking.py:9-23 hands ultralytics BGR uint8 frames; its letterbox-free part is
// BGR -> RGB, / 255): [B][H][W][3] uint8 BGR -> [B][H][W][cpad] float16 RGB in [0, 1], channels 3.. zero -- the NHWC buffer the stem
// convolution reads.  (float)u / 255 rounded to f16: what torch computes for half(u) / 255.
__global__ void __launch_bounds__(256) preprocess_kernel(const uint8_t* __restrict__ in, _Float16* __restrict__ out, long long pixels, int cpad)
{
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < pixels; p += (long long)gridDim.x * blockDim.x) {
        const uint8_t* s = in + p * 3;
        _Float16* d = out + p * cpad;
        const _Float16 r = (_Float16)((float)s[2] / 255.0f), g = (_Float16)((float)s[1] / 255.0f), b = (_Float16)((float)s[0] / 255.0f);
        if (cpad == 4) {
            typedef _Float16 half4 __attribute__((ext_vector_type(4)));
            *reinterpret_cast<half4*>(d) = half4{r, g, b, (_Float16)0.0f};
        } else {
            d[0] = r; d[1] = g; d[2] = b;
            for (int c = 3; c < cpad; ++c) d[c] = (_Float16)0.0f;
        }
    }
}

// SPPF (the network's spatial-pyramid block): y1 = maxpool5(x), y2 = maxpool5(y1), y3 = maxpool5(y2), stride 1, "same" padding, written
// next to x in the concatenation buffer [B][H][W][4c] (x = channels [0, c), written by the block's first convolution).  One workgroup
// per (image, 8-channel chunk): the H x W x 8 plane sits in LDS (16 bytes per pixel), each pool is a row pass and a column pass.
typedef _Float16 half8p __attribute__((ext_vector_type(8)));
__device__ __forceinline__ half8p max8(half8p a, half8p b)
{
    half8p r;
#pragma unroll
    for (int e = 0; e < 8; ++e) r[e] = a[e] > b[e] ? a[e] : b[e];
    return r;
}
__global__ void __launch_bounds__(256) sppf_pool_kernel(_Float16* __restrict__ buf, int H, int W, int c)
{
    extern __shared__ __attribute__((aligned(16))) _Float16 plane[];          // [2][H * W] half8: current plane, row-pass result
    half8p* cur = reinterpret_cast<half8p*>(plane);
    half8p* tmp = cur + H * W;
    const int chunks = c >> 3, b = blockIdx.x / chunks, ch = (blockIdx.x % chunks) * 8, ld = 4 * c, P = H * W;
    _Float16* img = buf + (size_t)b * P * ld + ch;
    for (int p = threadIdx.x; p < P; p += 256) cur[p] = *reinterpret_cast<const half8p*>(img + (size_t)p * ld);
    __syncthreads();
    for (int level = 1; level <= 3; ++level) {
        for (int p = threadIdx.x; p < P; p += 256) {          // max over the row window
            const int h = p / W, w = p - h * W;
            half8p m = cur[p];
            for (int d = 1; d <= 2; ++d) {
                if (w - d >= 0) m = max8(m, cur[p - d]);
                if (w + d < W) m = max8(m, cur[p + d]);
            }
            tmp[p] = m;
        }
        __syncthreads();
        for (int p = threadIdx.x; p < P; p += 256) {          // max over the column window, into the plane and the level's channel slice
            const int h = p / W;
            half8p m = tmp[p];
            for (int d = 1; d <= 2; ++d) {
                if (h - d >= 0) m = max8(m, tmp[p - d * W]);
                if (h + d < H) m = max8(m, tmp[p + d * W]);
            }
            cur[p] = m;       // (each thread rewrites only its own pixels of `cur`, which this pass does not read)
            *reinterpret_cast<half8p*>(img + (size_t)p * ld + level * c) = m;
        }
        __syncthreads();
    }
}

// The head's torch.cat((upsample2x(a), b), 1) as one pass: out[b][y][x] = (a[b][y / 2][x / 2][0..ca), b[b][y][x][0..cb)), NHWC f16, 16-byte chunks.
__global__ void __launch_bounds__(256) upsample_concat_kernel(const _Float16* __restrict__ a, const _Float16* __restrict__ b, _Float16* __restrict__ out,
                                                              int H, int W, int ca, int cb, long long chunks)
{
    const int cpa = ca >> 3, cpp = (ca + cb) >> 3;             // chunks of a pixel from a, chunks per output pixel
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < chunks; i += (long long)gridDim.x * blockDim.x) {
        const long long pix = i / cpp;
        const int k = (int)(i - pix * cpp);
        half8p v;
        if (k < cpa) {
            const int x = (int)(pix % W), y = (int)((pix / W) % H);
            const long long img = pix / ((long long)W * H);
            v = *reinterpret_cast<const half8p*>(a + ((img * (H >> 1) + (y >> 1)) * (W >> 1) + (x >> 1)) * ca + 8 * k);
        } else {
            v = *reinterpret_cast<const half8p*>(b + pix * cb + 8 * (k - cpa));
        }
        *reinterpret_cast<half8p*>(out + i * 8) = v;
    }
}

}  // namespace

hipError_t launch_upsample_concat(const void* a, const void* b, void* out, int B, int H, int W, int ca, int cb, hipStream_t stream)
{
    if (B <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1) || ca < 8 || cb < 8 || (ca & 7) || (cb & 7)) return hipErrorInvalidValue;
    const long long chunks = (long long)B * H * W * ((ca + cb) >> 3), blocks = (chunks + 255) / 256;
    hipLaunchKernelGGL(upsample_concat_kernel, dim3((unsigned)(blocks < 262144 ? blocks : 262144)), dim3(256), 0, stream, static_cast<const _Float16*>(a),
                       static_cast<const _Float16*>(b), static_cast<_Float16*>(out), H, W, ca, cb, chunks);
    return hipGetLastError();
}

hipError_t launch_sppf_pool(void* buf, int B, int H, int W, int c, hipStream_t stream)
{
    if (B <= 0 || H <= 0 || W <= 0 || c < 8 || (c & 7) != 0 || (long long)H * W > 2048) return hipErrorInvalidValue;
    const size_t lds = (size_t)2 * H * W * 16;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(sppf_pool_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(sppf_pool_kernel, dim3((unsigned)(B * (c >> 3))), dim3(256), lds, stream, static_cast<_Float16*>(buf), H, W, c);
    return hipGetLastError();
}

hipError_t launch_preprocess_bgr8(const void* frames, void* out, long long pixels, int cpad, hipStream_t stream)
{
    if (pixels <= 0 || cpad < 3) return hipErrorInvalidValue;
    const long long blocks = (pixels + 255) / 256;
    hipLaunchKernelGGL(preprocess_kernel, dim3((unsigned)(blocks < 65536 ? blocks : 65536)), dim3(256), 0, stream, static_cast<const uint8_t*>(frames),
                       static_cast<_Float16*>(out), pixels, cpad);
    return hipGetLastError();
}

hipError_t launch_conv2d_nhwc_f16(const void* x, const void* w, const float* bias, void* y, int B, int H, int W, int C, int N, int KH, int KW, int stride,
                                  int pad, int act, int ldy, const void* res, int ldr, hipStream_t stream)
{
    if (ldy < N || (res && ldr < N)) return hipErrorInvalidValue;
    if (B <= 0 || H <= 0 || W <= 0 || N <= 0 || KH <= 0 || KW <= 0 || stride <= 0 || pad < 0) return hipErrorInvalidValue;
    if (C < 4 || (C & (C - 1)) != 0 || ((KW * C) & 7) != 0) return hipErrorInvalidValue;       // whole 16-byte chunks per window row
    // C = 4: a chunk is two pixels, which must leave the image together -- even window starts, even width
    if (C == 4 && ((stride & 1) || (pad & 1) || (W & 1))) return hipErrorInvalidValue;
    ConvArgs a;
    a.B = B; a.H = H; a.W = W; a.C = C; a.N = N; a.KH = KH; a.KW = KW; a.stride = stride; a.pad = pad; a.act = act;
    a.Ho = (H + 2 * pad - KH) / stride + 1;
    a.Wo = (W + 2 * pad - KW) / stride + 1;
    if (a.Ho <= 0 || a.Wo <= 0) return hipErrorInvalidValue;
    a.ldy = ldy; a.ldr = ldr; a.res = static_cast<const _Float16*>(res);
    a.wide = ((N & 7) == 0 && (ldy & 7) == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0 &&
              (!res || ((ldr & 7) == 0 && (reinterpret_cast<uintptr_t>(res) & 15) == 0))) ? 1 : 0;
    a.c_shift = 0;
    while ((1 << a.c_shift) < C) ++a.c_shift;
    a.M = (long long)B * a.Ho * a.Wo;
    if (a.M > 0x7fffffffLL) return hipErrorInvalidValue;       // (the kernel splits pixel indices in 32 bits)
    const long long gx = (a.M + kBM - 1) / kBM;
    if (N >= 128) {
        hipLaunchKernelGGL(conv_igemm_kernel<128>, dim3((unsigned)gx, (unsigned)((N + 127) / 128)), dim3(256), 0, stream, static_cast<const _Float16*>(x),
                           static_cast<const _Float16*>(w), bias, static_cast<_Float16*>(y), a);
    } else if (N <= 32) {
        hipLaunchKernelGGL(conv_igemm_kernel<32>, dim3((unsigned)gx, 1u), dim3(256), 0, stream, static_cast<const _Float16*>(x), static_cast<const _Float16*>(w),
                           bias, static_cast<_Float16*>(y), a);
    } else {
        hipLaunchKernelGGL(conv_igemm_kernel<64>, dim3((unsigned)gx, (unsigned)((N + 63) / 64)), dim3(256), 0, stream, static_cast<const _Float16*>(x),
                           static_cast<const _Float16*>(w), bias, static_cast<_Float16*>(y), a);
    }
    return hipGetLastError();
}

}  // namespace bf
