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
#include <cstdlib>
#include <cstring>

#include "das_kernels.h"
#include "f32_split.h"

namespace bf {
namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float4v __attribute__((ext_vector_type(4)));
typedef float float16v __attribute__((ext_vector_type(16)));

// The element type of activations and weights: _Float16 (v_mfma_f32_32x32x16_f16) or float (v_mfma_f32_32x32x2_f32: exact f32, the
// precision ultralytics' predict runs at by default).  Everything is laid out in 16-byte chunks of E elements.
template <typename T> struct Elem;
template <> struct Elem<_Float16> { static constexpr int E = 8; typedef half8 vec; };
template <> struct Elem<float> { static constexpr int E = 4; typedef float4v vec; };

constexpr int kBM = 128, kStageBytes = 64, kRowBytes = kStageBytes + 16;   // a stage is 64 bytes of K per row, rows padded to 80
// kBN (template parameter): output channels per workgroup tile.
//   32: layers of up to 32 channels (the 6x6 stem among them); 64; 128 for layers of 128 channels and more -- there the four waves
//   form a 2 x 2 grid of 64-pixel x 64-channel tiles (two A and two B fragments feed four MFMAs: one LDS read per MFMA instead of 1.5,
//   a third less staging per MFMA), otherwise they stack 4 x 1 with 32-pixel x kBN tiles.

// n / d for n < 2^31 and a divisor fixed at launch: q = (mulhi(magic, n) + n) >> shift (Granlund / Montgomery round-up form; the sum cannot overflow
// below 2^31) -- three instructions where hipcc's unsigned division is about thirty.  The host fills it (make_fastdiv).
struct FastDiv { unsigned magic, shift; };
__device__ __forceinline__ unsigned fdiv(unsigned n, FastDiv d) { return (__umulhi(d.magic, n) + n) >> d.shift; }
inline FastDiv make_fastdiv(unsigned d)
{
    unsigned l = 0;
    while ((1ull << l) < d) ++l;
    FastDiv f;
    f.magic = (unsigned)((((1ull << l) - d) << 32) / d + 1ull);
    f.shift = l;
    return f;
}

struct ConvArgs {
    int B, H, W, C, Ho, Wo, N, KH, KW, stride, pad, act;
    int ldy, ldr;         // elements between consecutive pixels of the output / the residual
    int wide;
    const void* res;      // optional residual, [M][ldr]
    int c_shift;          // log2(C)
    long long M;          // B * Ho * Wo
    // kCat (1x1 window, stride 1, no padding, over a virtual concatenation): channels [0, c1) of a pixel come from x (ld1 elements
    // between pixels; up1: x is [B][H/2][W/2] and read through a nearest-neighbour 2x upsampling), channels [c1, C) from x2 (ld2)
    const void* x2;
    int c1, ld1, ld2, up1;
    int wld;              // elements between consecutive weight rows (rows are zero-padded to whole 128-byte stages)
    int fast_tap;         // LDS-DMA kernels, window path: every stage lies inside one (kh, kw) tap (C elements >= a stage row) and KH * KW <= 32
    FastDiv d_img, d_row; // by Ho * Wo (pixel -> image) and by Wo (-> row)
    FastDiv d_p0, d_p1, d_p2, d_p3;   // the patch kernel's: tiles per image, tiles per tile row, chunks per patch row, bytes per padded weight row
};

// SiLU, x * sigmoid(x), of an f32 accumulator: exp by v_exp_f32 (2^(-x log2 e)), the reciprocal by v_rcp_f32 (1 ulp).  For a float32 layer one
// Newton step follows (the quotient is then correct to an ulp: ~1e-7 of the value, two orders inside the 1e-5 bar the float32 path is held to);
// a float16 result is rounded to 11 bits right after.
__device__ __forceinline__ float silu_f(float v, bool refine)
{
    const float d = 1.0f + __expf(-v);
    float r = __builtin_amdgcn_rcpf(d);
    if (refine) r = __builtin_fmaf(r, __builtin_fmaf(-d, r, 1.0f), r);
    return v * r;
}

template <typename T, int kBN, bool kCat>
__global__ void __launch_bounds__(256) conv_igemm_kernel(const T* __restrict__ x, const T* __restrict__ w, const float* __restrict__ bias,
                                                         T* __restrict__ y, ConvArgs a)
{
    typedef typename Elem<T>::vec vec;
    constexpr int E = Elem<T>::E, kBK = kStageBytes / (int)sizeof(T), kRow = kRowBytes / (int)sizeof(T);
    constexpr bool kF32 = sizeof(T) == 4;
    constexpr int kWN = kBN == 128 ? 2 : 1, kWM = 4 / kWN;              // the waves' grid over the tile
    constexpr int kTM = kBM / kWM / 32, kTN = kBN / kWN / 32;           // 32 x 32 MFMA tiles per wave
    constexpr int kBRows = (kBN + 63) / 64;                             // weight rows a thread moves per stage
    constexpr int kCRow = kBN + E;                                      // the output tile's rows in LDS
    constexpr int kStage = 2 * (kBM + kBN) * kRow, kTile = kBM * kCRow;
    __shared__ __attribute__((aligned(16))) T smem[kStage > kTile ? kStage : kTile];
    T* const As = smem;                                                 // [2][kBM][kRow]
    T* const Bs = smem + 2 * kBM * kRow;                                // [2][kBN][kRow]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave % kWM, wn = wave / kWM;
    const long long m0 = (long long)blockIdx.x * kBM;
    const int n0 = blockIdx.y * kBN;
    // K is walked in 16-byte chunks of E elements: chunk q = (kh, E (q % cpk) elements into the kh run of KW * C); 4 chunks per stage
    const int cpk = (a.KW * a.C) / E, n_chunk = a.KH * cpk, n_stage = (n_chunk + 3) >> 2;

    // this thread's chunks: pixel rows r and r + 64, chunk `ck` of the stage; weight rows r (+ 64), same chunk
    const int r = tid >> 2, ck = tid & 3;
    int hi0[2], wi0[2];
    const T* px[2];
    const T* px2[2];
    bool pv[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const long long m = m0 + r + 64 * i;
        pv[i] = m < a.M;
        const unsigned mm = pv[i] ? (unsigned)m : 0u;          // (M < 2^31: launch_conv2d)
        if constexpr (kCat) {
            px2[i] = static_cast<const T*>(a.x2) + (size_t)mm * a.ld2 - a.c1;          // indexed by the channel of the concatenation
            if (a.up1) {
                const unsigned b = fdiv(mm, a.d_img), rem = mm - b * (unsigned)(a.Wo * a.Ho);
                const unsigned ho = fdiv(rem, a.d_row), wo = rem - ho * (unsigned)a.Wo;
                px[i] = x + ((size_t)(b * (unsigned)(a.Ho >> 1) + (ho >> 1)) * (unsigned)(a.Wo >> 1) + (wo >> 1)) * a.ld1;
            } else {
                px[i] = x + (size_t)mm * a.ld1;
            }
        } else {
            const unsigned b = fdiv(mm, a.d_img), rem = mm - b * (unsigned)(a.Wo * a.Ho);
            const unsigned ho = fdiv(rem, a.d_row), wo = rem - ho * (unsigned)a.Wo;
            hi0[i] = (int)ho * a.stride - a.pad;
            wi0[i] = (int)wo * a.stride - a.pad;
            px[i] = x + (size_t)b * a.H * a.W * a.C;
        }
    }
    bool wv[kBRows];
    const T* wrow[kBRows];
#pragma unroll
    for (int i = 0; i < kBRows; ++i) {
        wv[i] = r + 64 * i < kBN && n0 + r + 64 * i < a.N;
        wrow[i] = w + (size_t)(wv[i] ? n0 + r + 64 * i : 0) * a.wld;               // weight rows are padded to whole stages
    }

    vec ra[2], rb[kBRows];
    bool oka[2];          // (the zero padding is applied when the chunk goes to LDS: a select right behind the load would wait for it there)
    auto fetch = [&](int s) {
        const int q = 4 * s + ck;
        if constexpr (kCat) {
            const int c = q * E;                               // the window is one pixel: chunk q = channels [c, c + E) of the concatenation
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const bool ok = pv[i] && q < n_chunk;
                const T* src = c < a.c1 ? px[i] + c : px2[i] + c;
                ra[i] = *reinterpret_cast<const vec*>(ok ? src : x);
                oka[i] = ok;
            }
        } else {
            const int kh = q / cpk, kk = (q - kh * cpk) * E;       // elements into the kh run
            const int kw = kk >> a.c_shift, c = kk & (a.C - 1);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int hi = hi0[i] + kh, wi = wi0[i] + kw;
                const bool ok = pv[i] && q < n_chunk && (unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W;
                const size_t off = ok ? ((size_t)hi * a.W + wi) * a.C + c : 0;
                ra[i] = *reinterpret_cast<const vec*>(px[i] + off);
                oka[i] = ok;
            }
        }
#pragma unroll
        for (int i = 0; i < kBRows; ++i) rb[i] = *reinterpret_cast<const vec*>(wrow[i] + (size_t)q * E);
    };
    auto stash = [&](int buf) {
        vec z;
#pragma unroll
        for (int e = 0; e < E; ++e) z[e] = 0;
        *reinterpret_cast<vec*>(&As[(buf * kBM + r) * kRow + ck * E]) = oka[0] ? ra[0] : z;
        *reinterpret_cast<vec*>(&As[(buf * kBM + r + 64) * kRow + ck * E]) = oka[1] ? ra[1] : z;
#pragma unroll
        for (int i = 0; i < kBRows; ++i)
            if (r + 64 * i < kBN) *reinterpret_cast<vec*>(&Bs[(buf * kBN + r + 64 * i) * kRow + ck * E]) = wv[i] ? rb[i] : z;
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
        // a lane's fragment: 16 bytes of row (lane & 31) at byte 16 (lane >> 5) of each 32-byte step.  f16: the 8 halfs are the operand
        // of one 32x32x16 MFMA.  f32: the 4 floats feed four 32x32x2 MFMAs, whose two k slots (lane halves) then hold elements
        // e and e + 4 of the 8-deep step -- the same assignment on both operands, so each product pairs the right k.
        const T* A = &As[(buf * kBM + wm * 32 * kTM + (lane & 31)) * kRow + E * (lane >> 5)];
        const T* Bp = &Bs[(buf * kBN + wn * 32 * kTN + (lane & 31)) * kRow + E * (lane >> 5)];
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
            vec af[kTM], bf[kTN];
#pragma unroll
            for (int i = 0; i < kTM; ++i) af[i] = *reinterpret_cast<const vec*>(A + 32 * i * kRow + 2 * E * k2);
#pragma unroll
            for (int t = 0; t < kTN; ++t) bf[t] = *reinterpret_cast<const vec*>(Bp + 32 * t * kRow + 2 * E * k2);
            if constexpr (kF32) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int i = 0; i < kTM; ++i)
#pragma unroll
                        for (int t = 0; t < kTN; ++t) acc[i][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][e], bf[t][e], acc[i][t], 0, 0, 0);
            } else {
#pragma unroll
                for (int i = 0; i < kTM; ++i)
#pragma unroll
                    for (int t = 0; t < kTN; ++t) acc[i][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i], bf[t], acc[i][t], 0, 0, 0);
            }
        }
        if (s + 1 < n_stage) stash(buf ^ 1);                   // the other buffer: its readers passed the previous barrier
        __syncthreads();
    }

    // Epilogue.  C lane map of the 32x32 MFMAs: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).
    // The tile goes through LDS (the stage buffers are free after the loop's last barrier): rows of kBN channels + E.
    T* Cs = smem;
#pragma unroll
    for (int t = 0; t < kTN; ++t) {
        const int col = wn * 32 * kTN + 32 * t + (lane & 31), n = n0 + col;
        const float bn = (bias && n < a.N) ? bias[n] : 0.0f;
#pragma unroll
        for (int i = 0; i < kTM; ++i) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                float v = acc[i][t][e] + bn;
                if (a.act) v = silu_f(v, kF32);
                Cs[(wm * 32 * kTM + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)) * kCRow + col] = (T)v;
            }
        }
    }
    __syncthreads();
    const bool wide = a.wide != 0;                             // 16-byte stores need 16-byte rows and bases (launch_conv2d)
    constexpr int kCC = kBN / E;                               // 16-byte chunks per tile row
    const T* res = static_cast<const T*>(a.res);
#pragma unroll
    for (int j = 0; j < (kBM * kCC) / 256; ++j) {
        const int id = tid + 256 * j, row = id / kCC, cc = (id % kCC) * E;
        const long long m = m0 + row;
        if (m >= a.M || n0 + cc >= a.N) continue;
        const T* src = &Cs[row * kCRow + cc];
        T* dst = y + (size_t)m * a.ldy + n0 + cc;
        if (wide) {
            vec v = *reinterpret_cast<const vec*>(src);
            if (res) {                                         // added the way torch adds two tensors of the type: in f32, rounded once
                const vec rv = *reinterpret_cast<const vec*>(res + (size_t)m * a.ldr + n0 + cc);
#pragma unroll
                for (int e = 0; e < E; ++e) v[e] = (T)((float)v[e] + (float)rv[e]);
            }
            *reinterpret_cast<vec*>(dst) = v;
        } else {
            for (int e = 0; e < E && n0 + cc + e < a.N; ++e)
                dst[e] = res ? (T)((float)src[e] + (float)res[(size_t)m * a.ldr + n0 + cc + e]) : src[e];
        }
    }
}

// ---- The same GEMM with both operand tiles staged by LDS-DMA (buffer_load_dwordx4 ... lds: global memory -> LDS without a register hop).
//
// conv_igemm_kernel moves every operand global -> VGPR -> ds_write_b128 -> LDS one stage ahead; its staging stores alone cost about what
// its MFMAs cost (profiles/r02_pmc_mfma_detector_conv.csv: matrix pipes 0.17 busy).  Here:
//   * a ring of kDepth LDS stage buffers (64 bytes of K per row); every wave issues the 1-KiB pieces of a stage (two of the pixel tile, one or
//     two of the weight tile) kDepth - 1 stages ahead and waits for its own with a COUNTED s_waitcnt vmcnt, so that kDepth - 2 stages of loads stay
//     in flight across the stage's one s_barrier (raw: __syncthreads() would drain them);
//   * an LDS-DMA piece lands lane-linear (lane l at base + 16 l), so rows cannot be padded: a 64-byte row holds its four 16-byte chunks in
//     slots permuted by slot = chunk ^ ((row >> 2) & 3) -- applied to the SOURCE address of the lane that fills the slot and to the address of
//     the fragment read -- which spreads the 16 rows of every ds_read_b128 lane group over all 64 banks;
//   * the zero padding of the window, rows past the last pixel and output channels past the last are lanes whose offset is pushed past the
//     buffer descriptor's range: the hardware returns zeros, there is no select and no branch;
//   * (kh, position in the kh run) of a lane's chunk advance incrementally from stage to stage (no division in the loop); a 1x1 layer's
//     offsets are its pixel's base plus 64 bytes per stage in a scalar register.
// Operand tensors must lie below 2 GiB each (32-bit buffer offsets with the top bit as the out-of-range mark): launch_conv2d_nhwc falls
// back to conv_igemm_kernel otherwise.
typedef __attribute__((address_space(3))) void lds_void;
constexpr unsigned kOob = 0x80000000u;

template <typename T, int kBM_, int kBN, int kCPR, bool kSplit = false> struct DmaGeo {
    // kCPR: 16-byte chunks of K per row and stage -- 4 (64-byte rows) or 8 (128-byte rows: every row of a piece is one full cache line, half the
    // barriers and address computations per byte; float16).
    // Ring depth: a float32 stage carries 32 MFMAs of 64 cycles per wave (one stage of loads in flight covers the memory latency, and the smaller
    // ring admits a third workgroup per CU); a float16 stage of 64-byte rows is 8 MFMAs of 32 cycles, so two stages stay in flight; 128-byte rows
    // double the stage and run a ring of three where LDS allows two workgroups per CU with it, of two otherwise.
    static constexpr int kRowBytes_ = 16 * kCPR;
    static constexpr int kStageBytes_ = (kBM_ + kBN) * kRowBytes_;
#ifndef BF_F32_DEPTH
#define BF_F32_DEPTH 2
#endif
#ifndef BF_F16_DEPTH
#define BF_F16_DEPTH 3
#endif
    // (kSplit: float32 operands on the bfloat16 pipes, below -- a stage is 6 MFMAs of 32 cycles per tile, as short as a float16 one)
    static constexpr int kDepth = sizeof(T) == 4 && !kSplit ? BF_F32_DEPTH : (kCPR == 4 ? BF_F16_DEPTH : (3 * kStageBytes_ <= 80 * 1024 ? 3 : 2));
    static constexpr int kRingBytes = kDepth * kStageBytes_;
    static constexpr int kPieceRows = 64 / kCPR;                        // a 1-KiB piece = kPieceRows rows; piece p of a tile belongs to wave p % 4
    static constexpr bool kHalfB = kBN / kPieceRows < 4;                // (32 channels, 64-byte rows: two pieces, issued as four half pieces)
    static constexpr int kAPieces = kBM_ / kPieceRows / 4;
    static constexpr int kBPieces = kHalfB ? 1 : kBN / kPieceRows / 4;
    static constexpr int kPerStage = kAPieces + kBPieces;               // DMA instructions a wave issues per stage
    // the waves' grid over the tile, 32 x 32 MFMA tiles per wave
    static constexpr int kWN = kBN == 128 ? 2 : (kBM_ == 64 && kBN == 64 ? 2 : 1), kWM = 4 / kWN;
    static constexpr int kTM = kBM_ / kWM / 32, kTN = kBN / kWN / 32;
    static_assert(kTM >= 1 && kTN >= 1 && kAPieces >= 1, "every wave owns at least one MFMA tile and one piece");
    // the epilogue hands the tile to the stores one 32-column block per wave column at a time
    static constexpr int kCCols = kWN * 32;
    // slot of logical chunk c in row r: spreads the 16 rows of a ds_read_b128 lane group over all 64 banks
    __device__ static constexpr int swz(int r) { return kCPR == 4 ? (r >> 2) & 3 : (r >> 1) & 7; }
};

// (The body lives in a __device__ function: the buffer-descriptor type of the LDS-DMA builtins does not exist in the host pass, and a kernel whose
//  body the host pass cannot parse gets no launch stub; a __device__ function's host-side diagnostics are deferred and dropped.)
// kSplit (float32 only): the products run on the bfloat16 matrix pipes at float32 accuracy -- f32_split.h (three-way exact operand split, six part
// products per product, float32 accumulation).
using split::Split3;
using split::split3;
using split::bf16x8v;

template <typename T, int kBM_, int kBN, int kCPR, bool kCat, bool kSplit = false>
__device__ __forceinline__ void conv_dma_body(const T* __restrict__ x, const T* __restrict__ w, const float* __restrict__ bias, T* __restrict__ y, const ConvArgs& a,
                                              unsigned x_bytes, unsigned x2_bytes, unsigned w_bytes)
{
    typedef typename Elem<T>::vec vec;
    typedef DmaGeo<T, kBM_, kBN, kCPR, kSplit> G;
    constexpr int E = Elem<T>::E;
    constexpr bool kF32 = sizeof(T) == 4;
    static_assert(!kSplit || kF32, "the split is a float32 mode");
    constexpr int kWM = G::kWM, kTM = G::kTM, kTN = G::kTN;
    constexpr int kCRow = G::kCCols + E;
    constexpr int kTileBytes = kBM_ * kCRow * (int)sizeof(T);
    constexpr int D = G::kDepth, RB = G::kRowBytes_, PR = G::kPieceRows;
    constexpr int kMainBytes = G::kRingBytes > kTileBytes ? G::kRingBytes : kTileBytes;
    __shared__ __attribute__((aligned(16))) unsigned char smem[kMainBytes + 512];      // ring / output image, then the tile's kBN bias values
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % kWM, wn = wave / kWM;
    const long long m0 = (long long)blockIdx.x * kBM_;
    const int n0 = blockIdx.y * kBN;
    const int cpk = (a.KW * a.C) / E, n_chunk = a.KH * cpk, n_stage = (n_chunk + kCPR - 1) / kCPR;

    // buffer descriptors (wave-uniform): reads past num_records return zeros
    const auto rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(x), 0, (int)x_bytes, 0x00020000);
    const auto rx2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(kCat && a.x2 ? a.x2 : static_cast<const void*>(x)), 0, (int)(kCat && a.x2 ? x2_bytes : x_bytes), 0x00020000);
    const auto rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(w), 0, (int)w_bytes, 0x00020000);

    // ---- what this lane fills.  Piece p = wave + 4 j of the pixel tile = its rows [p PR, (p + 1) PR): lane l fills row p PR + l / kCPR, slot l % kCPR,
    // i.e. the row's logical chunk ck = slot ^ swz(row) -- the same for every piece of the wave (their rows differ by multiples of 4 PR).
    const int ck = (lane % kCPR) ^ G::swz(wave * PR + lane / kCPR);
    unsigned abase[G::kAPieces];                            // byte offset of the row's pixel (kCat: in source 1 / source 2), or kOob
    unsigned abase2[G::kAPieces];
    int hi0[G::kAPieces], wi0[G::kAPieces];
#pragma unroll
    for (int j = 0; j < G::kAPieces; ++j) {
        const int row = (wave + 4 * j) * PR + lane / kCPR;
        const long long m = m0 + row;
        const bool pv = m < a.M;
        const unsigned mm = pv ? (unsigned)m : 0u;
        abase2[j] = 0u; hi0[j] = 0; wi0[j] = 0;
        if constexpr (kCat) {
            unsigned p1 = mm;
            if (a.up1) {
                const unsigned b = fdiv(mm, a.d_img), rem = mm - b * (unsigned)(a.Wo * a.Ho);
                const unsigned ho = fdiv(rem, a.d_row), wo = rem - ho * (unsigned)a.Wo;
                p1 = (b * (unsigned)(a.Ho >> 1) + (ho >> 1)) * (unsigned)(a.Wo >> 1) + (wo >> 1);
            }
            abase[j] = pv ? p1 * (unsigned)a.ld1 * (unsigned)sizeof(T) : kOob;
            abase2[j] = pv ? (mm * (unsigned)a.ld2 - (unsigned)a.c1) * (unsigned)sizeof(T) : kOob;      // indexed by the channel of the concatenation
        } else {
            const unsigned b = fdiv(mm, a.d_img), rem = mm - b * (unsigned)(a.Wo * a.Ho);
            const unsigned ho = fdiv(rem, a.d_row), wo = rem - ho * (unsigned)a.Wo;
            hi0[j] = pv ? (int)ho * a.stride - a.pad : -(1 << 28);       // (a row past the last pixel is never inside the image)
            wi0[j] = (int)wo * a.stride - a.pad;
            abase[j] = b * (unsigned)(a.H * a.W * a.C) * (unsigned)sizeof(T);
        }
    }
    // the chunk's position in K: kh and the chunk index inside the kh run, advanced by kCPR chunks per stage
    int kh = ck / cpk, kk = ck - kh * cpk;
    // The cheap form of the per-stage addresses.  Window path with a.fast_tap: the kCPR chunks of a stage share their tap (kh, kw), so a lane's offset is
    //   [pixel base + ((hi0 W + wi0) C + ck E) bytes]  +  [((kh W + kw) C + c0) bytes]  =  lbase (per lane, fixed)  +  a wave-uniform term per stage,
    // and whether the tap falls inside the picture is bit (kh KW + kw) of a per-lane mask made once (nmask: 1 = outside): shifted to bit 31 and or-ed into
    // the offset it sends the lane past every tensor this kernel accepts -- three vector instructions per piece and stage where the general form (below,
    // kept for layers of fewer channels than a stage row) takes about twenty.  kCat: the chunk's 16 bytes fold into the bases, a stage adds kCPR * 16.
    unsigned lbase[G::kAPieces], nmask[G::kAPieces];
    int s_kh = 0, s_kw = 0, s_c0 = 0;                         // (wave-uniform) tap and first channel of the stage about to be issued
#pragma unroll
    for (int j = 0; j < G::kAPieces; ++j) { lbase[j] = 0u; nmask[j] = 0xffffffffu; }
    if constexpr (kCat) {
#pragma unroll
        for (int j = 0; j < G::kAPieces; ++j) { abase[j] += 16u * (unsigned)ck; abase2[j] += 16u * (unsigned)ck; }      // (kOob stays past 2^31)
    } else if (a.fast_tap) {
#pragma unroll
        for (int j = 0; j < G::kAPieces; ++j) {
            lbase[j] = abase[j] + (((unsigned)hi0[j] * (unsigned)a.W + (unsigned)wi0[j]) * (unsigned)a.C + (unsigned)(ck * E)) * (unsigned)sizeof(T);
            unsigned inside = 0u;
            for (int th = 0; th < a.KH; ++th)
                for (int tw = 0; tw < a.KW; ++tw)
                    if ((unsigned)(hi0[j] + th) < (unsigned)a.H && (unsigned)(wi0[j] + tw) < (unsigned)a.W) inside |= 1u << (th * a.KW + tw);
            nmask[j] = ~inside;
        }
    }
    // Weight tile: the same pieces over its kBN rows (kHalfB: waves fill rows [8 wave, 8 wave + 8) with their lanes 0..31)
    unsigned wbase[G::kBPieces];
#pragma unroll
    for (int j = 0; j < G::kBPieces; ++j) {
        const int row = G::kHalfB ? wave * 8 + (lane >> 2) : (wave + 4 * j) * PR + lane / kCPR;
        const int n = n0 + row;
        const int ckb = (lane % kCPR) ^ G::swz(row);
        wbase[j] = n < a.N ? ((unsigned)n * (unsigned)a.wld) * (unsigned)sizeof(T) + 16u * (unsigned)ckb : kOob;
    }
    const unsigned lds0 = (unsigned)(size_t)((__attribute__((address_space(3))) unsigned char*)smem);

    auto issue = [&](int s, int ring) {   // the DMA pieces of stage s into ring slot `ring` (past the last stage: out-of-range lanes only, zeros nobody reads)
#ifdef BF_DIAG_NO_ISSUE                                      // (timing experiment only: nothing is fetched, results wrong)
        return;
#endif
        const unsigned slot = lds0 + (unsigned)(ring * G::kStageBytes_);
        const bool live = s < n_stage;
        if constexpr (kCat) {
            // chunk kCPR s + ck = channels [(kCPR s + ck) E, + E) of the concatenation; a stage lies in one source (c1 % (kCPR E) == 0)
            const bool second = kCPR * s * E >= a.c1;
            const int left = n_chunk - kCPR * s;           // chunks of K from this stage on (only the last stage can have fewer than kCPR)
            const unsigned u = (unsigned)(s * RB);
#pragma unroll
            for (int j = 0; j < G::kAPieces; ++j) {
                unsigned off = live ? (second ? abase2[j] : abase[j]) + u : kOob;
                if (left < kCPR && ck >= left) off = kOob;
                lds_void* dst = (lds_void*)(size_t)(slot + (unsigned)((wave + 4 * j) * 1024));
                if (second) __builtin_amdgcn_raw_ptr_buffer_load_lds(rx2, dst, 16, off, 0, 0, 0);
                else __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, dst, 16, off, 0, 0, 0);
            }
        } else if (a.fast_tap) {
            const unsigned u = live ? (unsigned)(((s_kh * a.W + s_kw) * a.C + s_c0) * (int)sizeof(T)) : kOob;
            const int sh = live ? 31 - (s_kh * a.KW + s_kw) : 0;          // (past the last stage: bit 31 comes from u, whatever the mask says)
#pragma unroll
            for (int j = 0; j < G::kAPieces; ++j) {
                const unsigned off = live ? (((nmask[j] << sh) & 0x80000000u) | (lbase[j] + u)) : kOob;
                lds_void* dst = (lds_void*)(size_t)(slot + (unsigned)((wave + 4 * j) * 1024));
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, dst, 16, off, 0, 0, 0);
            }
            s_c0 += kCPR * E;
            if (s_c0 >= a.C) { s_c0 = 0; if (++s_kw == a.KW) { s_kw = 0; ++s_kh; } }
        } else {
            const int ke = kk * E, kw = ke >> a.c_shift, c = ke & (a.C - 1);
#pragma unroll
            for (int j = 0; j < G::kAPieces; ++j) {
                const int hi = hi0[j] + kh, wi = wi0[j] + kw;
#ifdef BF_DIAG_NO_A                                         // (timing experiment only: the pixel tile's pieces fetch nothing, results wrong)
                const bool ok = false;
#else
                const bool ok = live && kh < a.KH && (unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W;
#endif
                const unsigned off = ok ? abase[j] + (unsigned)((hi * a.W + wi) * a.C + c) * (unsigned)sizeof(T) : kOob;
                lds_void* dst = (lds_void*)(size_t)(slot + (unsigned)((wave + 4 * j) * 1024));
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, dst, 16, off, 0, 0, 0);
            }
            kk += kCPR;
            while (kk >= cpk) { kk -= cpk; ++kh; }
        }
#pragma unroll
        for (int j = 0; j < G::kBPieces; ++j) {
            const unsigned off = live ? wbase[j] + (unsigned)(s * RB) : kOob;
            lds_void* dst = (lds_void*)(size_t)(slot + (unsigned)(kBM_ * RB) + (unsigned)(G::kHalfB ? wave * 512 : (wave + 4 * j) * 1024));
            if (!G::kHalfB || lane < 32) __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, dst, 16, off, 0, 0, 0);
        }
    };

    float16v acc[kTM][kTN];
#pragma unroll
    for (int i = 0; i < kTM; ++i)
#pragma unroll
        for (int t = 0; t < kTN; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][t][e] = 0.0f;

    // fragment reads: row (lane & 31) of a 32-row block, logical chunk (lane >> 5) + 2 k2 -> slot ^ swz(row)
    const int sw = G::swz(lane & 31);
    const unsigned fa = (unsigned)((wm * 32 * kTM + (lane & 31)) * RB), fb = (unsigned)(kBM_ * RB + (wn * 32 * kTN + (lane & 31)) * RB);
    unsigned fo[kCPR / 2];
#pragma unroll
    for (int k2 = 0; k2 < kCPR / 2; ++k2) fo[k2] = (unsigned)((((lane >> 5) + 2 * k2) ^ sw) * 16);

    // The tile's bias values travel like the operands: one LDS-DMA piece of 4-byte lanes (waves 0 and 1, 64 channels each; channels past the last are
    // out of the descriptor's range: zeros), the oldest vector-memory operation of its wave, landed long before the epilogue reads it -- an ordinary
    // global load there would sit on its latency with nothing left to overlap it (a short layer's whole K is one or two stages), and one here would
    // make hipcc wait for it before the first DMA.
    if (bias != nullptr && wave * 64 < kBN) {
        const auto rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(bias), 0, a.N * 4, 0x00020000);
        lds_void* dst = (lds_void*)(size_t)(lds0 + (unsigned)kMainBytes + (unsigned)(wave * 256));
        if (kBN >= 64 || lane < kBN) __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, dst, 4, (unsigned)(n0 + wave * 64 + lane) * 4u, 0, 0, 0);
    }
#pragma unroll
    for (int s = 0; s < D - 1; ++s) issue(s, s);
    int rd = 0, wr = D - 1;                                   // ring slots: stage s is read from rd, stage s + D - 1 goes to wr (= the slot stage s - 1 left)
    for (int s = 0; s < n_stage; ++s) {
        // this wave's pieces of stage s have landed once at most (D - 2) younger stages' are outstanding; the barrier extends that to every wave's
        // pieces, and says that every wave has finished reading stage s - 1, whose slot the next issue refills
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"((D - 2) * G::kPerStage) : "memory");
#ifndef BF_DIAG_NO_BARRIER                                  // (timing experiment only: waves uncoupled, results wrong)
        __builtin_amdgcn_s_barrier();
#endif
        issue(s + D - 1, wr);
        const unsigned char* st = smem + rd * G::kStageBytes_;
        rd = rd + 1 == D ? 0 : rd + 1;
        wr = wr + 1 == D ? 0 : wr + 1;
#ifdef BF_CONV_PRIO
        __builtin_amdgcn_s_setprio(BF_CONV_PRIO);
#endif
        if constexpr (kSplit) {
#pragma unroll
            for (int k4 = 0; k4 < kCPR / 4; ++k4) {            // 16 values of K: this lane's two chunks of every row
                Split3 as[kTM], bs[kTN];
#pragma unroll
                for (int i = 0; i < kTM; ++i)
                    as[i] = split3(*reinterpret_cast<const float4v*>(st + fa + 32 * RB * i + fo[2 * k4]), *reinterpret_cast<const float4v*>(st + fa + 32 * RB * i + fo[2 * k4 + 1]));
#pragma unroll
                for (int t = 0; t < kTN; ++t)
                    bs[t] = split3(*reinterpret_cast<const float4v*>(st + fb + 32 * RB * t + fo[2 * k4]), *reinterpret_cast<const float4v*>(st + fb + 32 * RB * t + fo[2 * k4 + 1]));
#define BF_SPLIT_MFMA(P, Q)                                                                                                                          \
                _Pragma("unroll") for (int i = 0; i < kTM; ++i)                                                                                      \
                    _Pragma("unroll") for (int t = 0; t < kTN; ++t)                                                                                  \
                        acc[i][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8v, as[i].P), __builtin_bit_cast(bf16x8v, bs[t].Q), acc[i][t], 0, 0, 0)
                BF_SPLIT_MFMA(l, h); BF_SPLIT_MFMA(h, l); BF_SPLIT_MFMA(m, m); BF_SPLIT_MFMA(m, h); BF_SPLIT_MFMA(h, m); BF_SPLIT_MFMA(h, h);
#undef BF_SPLIT_MFMA
            }
        } else {
#pragma unroll
        for (int k2 = 0; k2 < kCPR / 2; ++k2) {
            vec af[kTM], bf[kTN];
#pragma unroll
            for (int i = 0; i < kTM; ++i) af[i] = *reinterpret_cast<const vec*>(st + fa + 32 * RB * i + fo[k2]);
#pragma unroll
            for (int t = 0; t < kTN; ++t) bf[t] = *reinterpret_cast<const vec*>(st + fb + 32 * RB * t + fo[k2]);
            if constexpr (kF32) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int i = 0; i < kTM; ++i)
#pragma unroll
                        for (int t = 0; t < kTN; ++t) acc[i][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][e], bf[t][e], acc[i][t], 0, 0, 0);
            } else {
#pragma unroll
                for (int i = 0; i < kTM; ++i)
#pragma unroll
                    for (int t = 0; t < kTN; ++t) acc[i][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i], bf[t], acc[i][t], 0, 0, 0);
            }
        }
        }
#ifdef BF_CONV_PRIO
        __builtin_amdgcn_s_setprio(0);
#endif
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the tail's dummy pieces: the ring is reused by the output tile

    // Epilogue: bias, SiLU, the tile transposed through LDS, 16-byte stores, optional slice pitch and residual.  One pass per MFMA tile column of a
    // wave (kTN passes): the LDS image of a pass is kBM rows of the kWN waves' 32 columns each, so that it never outgrows the ring.
    T* Cs = reinterpret_cast<T*>(smem);
    const bool wide = a.wide != 0;
    constexpr int kCC = G::kCCols / E;                         // 16-byte chunks per row of a pass
    const T* res = static_cast<const T*>(a.res);
#pragma unroll
    for (int t = 0; t < kTN; ++t) {
        __syncthreads();                                       // the stage buffers (pass 0) / the previous pass's image have been read
        {
            const int col = wn * 32 + (lane & 31);
            const float bn = bias != nullptr ? reinterpret_cast<const float*>(smem + kMainBytes)[wn * 32 * kTN + 32 * t + (lane & 31)] : 0.0f;
#pragma unroll
            for (int i = 0; i < kTM; ++i) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    float v = acc[i][t][e] + bn;
                    if (a.act) v = silu_f(v, kF32);
                    Cs[(wm * 32 * kTM + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)) * kCRow + col] = (T)v;
                }
            }
        }
        constexpr int kJ = (kBM_ * kCC + 255) / 256;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < kJ; ++j) {
            const int id = tid + 256 * j, row = id / kCC, cq = id % kCC;
            const int wcol = (cq * E) / 32, cin = (cq * E) % 32;       // the wave column the chunk came from, its place in that wave's 32 columns
            const int nn = n0 + wcol * 32 * kTN + 32 * t + cin;         // first output channel of the chunk
            const long long m = m0 + row;
            if (row >= kBM_ || m >= a.M || nn >= a.N) continue;
            const T* src = &Cs[row * kCRow + cq * E];
            T* dst = y + (size_t)m * a.ldy + nn;
            if (wide) {
                vec v = *reinterpret_cast<const vec*>(src);
                if (res) {
                    const vec rv = *reinterpret_cast<const vec*>(res + (size_t)m * a.ldr + nn);
#pragma unroll
                    for (int e = 0; e < E; ++e) v[e] = (T)((float)v[e] + (float)rv[e]);
                }
                *reinterpret_cast<vec*>(dst) = v;
            } else {
                for (int e = 0; e < E && nn + e < a.N; ++e)
                    dst[e] = res ? (T)((float)src[e] + (float)res[(size_t)m * a.ldr + nn + e]) : src[e];
            }
        }
    }
}

template <typename T, int kBM_, int kBN, int kCPR, bool kCat, bool kSplit = false>
__global__ void __launch_bounds__(256) conv_dma_kernel(const T* __restrict__ x, const T* __restrict__ w, const float* __restrict__ bias, T* __restrict__ y, ConvArgs a,
                                                       unsigned x_bytes, unsigned x2_bytes, unsigned w_bytes)
{
    conv_dma_body<T, kBM_, kBN, kCPR, kCat, kSplit>(x, w, bias, y, a, x_bytes, x2_bytes, w_bytes);
}

// ---- The stem (the network's first layer: 6x6 window, stride 2, 4 input channels after padding, 32 output channels) as a PATCH kernel.
//
// As an implicit GEMM the stem is the worst layer of the network (0.45 / 0.37 of its roofline in float32 / float16): its K walks 36 window pixels of 16
// (8) bytes each, so every 1-KiB DMA piece is 64 separate little requests, each input pixel is fetched nine times, and the address arithmetic per MFMA is
// the largest of any layer.  Here a workgroup owns an 8 x 16 block of output pixels of one image, loads the input PATCH under it once -- (7 s + KH) x
// (15 s + KW) pixels, whole rows of contiguous bytes -- and all 32 weight rows once, by LDS-DMA, and then builds the A fragments of the same MFMAs straight
// from the patch: the window pixel (kh, kw) of output pixel (ty, tx) is patch pixel (ty s + kh, tx s + kw).  K order, weight packing, the pairing of k
// slots with lane halves and the epilogue are those of conv_dma_kernel, so the results are bit-identical to it.
// Preconditions (launch_conv2d_nhwc checks them): C == 4, N <= 32, KW even, an even number of 16-byte chunks per window, patch + weights within LDS.
struct PatchGeo {
    static constexpr int kTH = 8, kTW = 16;                 // output pixels per tile
};

template <typename T>
__device__ __forceinline__ void conv_patch_body(const T* __restrict__ x, const T* __restrict__ w, const float* __restrict__ bias, T* __restrict__ y, const ConvArgs& a,
                                                unsigned x_bytes, unsigned w_bytes)
{
    typedef typename Elem<T>::vec vec;
    constexpr int E = Elem<T>::E;
    constexpr bool kF32 = sizeof(T) == 4;
    constexpr int kPix = 4 * (int)sizeof(T);                // bytes per (4-channel) pixel
    constexpr int kCRow = 32 + E;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int PH = (PatchGeo::kTH - 1) * a.stride + a.KH, PW = (PatchGeo::kTW - 1) * a.stride + a.KW;
    const int patch_bytes = (PH * PW * kPix + 1023) & ~1023;                    // whole pieces
    const int wrow = a.wld * (int)sizeof(T), wstride = wrow + 16;               // LDS row pitch of the weights: 36 dwords past a multiple of 64 -> conflict-free reads
    const int w_lds = 32 * wstride;
    const int tiles_x = (a.Wo + PatchGeo::kTW - 1) / PatchGeo::kTW, tiles_y = (a.Ho + PatchGeo::kTH - 1) / PatchGeo::kTH;
    const int b = (int)fdiv(blockIdx.x, a.d_p0), trem = (int)blockIdx.x - b * (tiles_x * tiles_y);
    const int trow = (int)fdiv((unsigned)trem, a.d_p1);
    const int oy0 = trow * PatchGeo::kTH, ox0 = (trem - trow * tiles_x) * PatchGeo::kTW;
    const int iy0 = oy0 * a.stride - a.pad, ix0 = ox0 * a.stride - a.pad;

    const auto rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(x), 0, (int)x_bytes, 0x00020000);
    const auto rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(w), 0, (int)w_bytes, 0x00020000);
    const unsigned lds0 = (unsigned)(size_t)((__attribute__((address_space(3))) unsigned char*)smem);
    const unsigned img = (unsigned)b * (unsigned)(a.H * a.W) * (unsigned)kPix;

    // the patch: 16-byte chunk g of the patch image (float32: pixel g; float16: pixels 2 g, 2 g + 1 -- PW, ix0 and W are even there)
    constexpr int kPPC = 16 / kPix;                          // pixels per chunk
    const int chunks = PH * PW / kPPC, cpr = PW / kPPC;      // chunks in the patch, per patch row
    for (int p = wave; p * 64 < chunks; p += 4) {
        const int g = p * 64 + lane;
        const int py = (int)fdiv((unsigned)g, a.d_p2), px = (g - py * cpr) * kPPC;
        const int iy = iy0 + py, ix = ix0 + px;
        const bool ok = g < chunks && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
        const unsigned off = ok ? img + (unsigned)(iy * a.W + ix) * (unsigned)kPix : kOob;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_void*)(size_t)(lds0 + (unsigned)(p * 1024)), 16, off, 0, 0, 0);
    }
    // the weights: LDS position -> (row, byte in the padded row); the pad and rows past the last channel are out-of-range lanes
    for (int p = wave; p * 1024 < w_lds; p += 4) {
        const int g = p * 1024 + lane * 16;
        const int row = (int)fdiv((unsigned)g, a.d_p3), col = g - row * wstride;
        const unsigned off = (g < w_lds && col < wrow && row < a.N) ? (unsigned)(row * wrow + col) : kOob;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_void*)(size_t)(lds0 + (unsigned)patch_bytes + (unsigned)(p * 1024)), 16, off, 0, 0, 0);
    }
    const int bias_at = patch_bytes + ((w_lds + 1023) & ~1023);
    if (bias != nullptr && wave == 0) {
        const auto rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(bias), 0, a.N * 4, 0x00020000);
        if (lane < 32) __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_void*)(size_t)(lds0 + (unsigned)bias_at), 4, (unsigned)lane * 4u, 0, 0, 0);
    }

    // this lane's fragment rows: output pixel (ty, tx) of the tile (a wave owns two tile rows = 32 pixels), weight row lane & 31
    const int r = lane & 31, half = lane >> 5;
    const int ty = wave * 2 + (r >> 4), tx = r & 15;
    const unsigned abase = (unsigned)((ty * a.stride) * PW + tx * a.stride) * (unsigned)kPix;
    const unsigned bbase = (unsigned)patch_bytes + (unsigned)(r * wstride);
    // chunk c of K = (kh, the c % cpk-th chunk of the kh run): lanes 0..31 take chunks 0, 2, .., lanes 32..63 chunks 1, 3, ..
    const int cpk = (a.KW * 4) / E, n_chunk = a.KH * cpk;
    int kh = half / cpk, kk = half - kh * cpk;
    float16v acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int c = half; c < n_chunk; c += 2) {
        const vec af = *reinterpret_cast<const vec*>(smem + abase + (unsigned)((kh * PW + kk * kPPC) * kPix));
        const vec bf = *reinterpret_cast<const vec*>(smem + bbase + (unsigned)(c * 16));
        if constexpr (kF32) {
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[e], bf[e], acc, 0, 0, 0);
        } else {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, bf, acc, 0, 0, 0);
        }
        kk += 2;
        while (kk >= cpk) { kk -= cpk; ++kh; }
    }

    // epilogue: bias, SiLU, transposed through LDS (the patch is done with), 16-byte stores of a pixel's 32 channels
    const float bn = bias != nullptr ? reinterpret_cast<const float*>(smem + bias_at)[r] : 0.0f;
    __syncthreads();
    T* Cs = reinterpret_cast<T*>(smem);
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        float v = acc[e] + bn;
        if (a.act) v = silu_f(v, kF32);
        Cs[(wave * 32 + (e & 3) + 8 * (e >> 2) + 4 * half) * kCRow + r] = (T)v;
    }
    __syncthreads();
    constexpr int kCC = 32 / E;
    const bool wide = a.wide != 0;
    const T* res = static_cast<const T*>(a.res);
#pragma unroll
    for (int j = 0; j < (128 * kCC) / 256; ++j) {
        const int id = tid + 256 * j, row = id / kCC, cc = (id % kCC) * E;
        const int oy = oy0 + (row >> 4), ox = ox0 + (row & 15);
        if (oy >= a.Ho || ox >= a.Wo || cc >= a.N) continue;
        const size_t m = ((size_t)b * a.Ho + oy) * a.Wo + ox;
        const T* src = &Cs[row * kCRow + cc];
        T* dst = y + m * a.ldy + cc;
        if (wide) {
            vec v = *reinterpret_cast<const vec*>(src);
            if (res) {
                const vec rv = *reinterpret_cast<const vec*>(res + m * a.ldr + cc);
#pragma unroll
                for (int e = 0; e < E; ++e) v[e] = (T)((float)v[e] + (float)rv[e]);
            }
            *reinterpret_cast<vec*>(dst) = v;
        } else {
            for (int e = 0; e < E && cc + e < a.N; ++e) dst[e] = res ? (T)((float)src[e] + (float)res[m * a.ldr + cc + e]) : src[e];
        }
    }
}

template <typename T>
__global__ void __launch_bounds__(256) conv_patch_kernel(const T* __restrict__ x, const T* __restrict__ w, const float* __restrict__ bias, T* __restrict__ y, ConvArgs a,
                                                         unsigned x_bytes, unsigned w_bytes)
{
    conv_patch_body<T>(x, w, bias, y, a, x_bytes, w_bytes);
}

// Camera frames -> network input (yolo_smooth_tracking.py:9-23 hands ultralytics BGR uint8 frames; its letterbox-free part is
// BGR -> RGB, / 255): [B][H][W][3] uint8 BGR -> [B][H][W][cpad] RGB in [0, 1] (float16 or float32), channels 3.. zero -- the NHWC
// buffer the stem convolution reads.  (float)u * (1.0f / 255.0f), for float16 rounded once more: bit for bit what torch computes for u.half() / 255 and
// u.float() / 255 on the GPU (ultralytics' `im /= 255`).
template <typename T>
__global__ void __launch_bounds__(256) preprocess_kernel(const uint8_t* __restrict__ in, T* __restrict__ out, long long pixels, int cpad)
{
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < pixels; p += (long long)gridDim.x * blockDim.x) {
        const uint8_t* s = in + p * 3;
        T* d = out + p * cpad;
        constexpr float inv = 1.0f / 255.0f;                   // torch divides a tensor by a scalar as a product with the scalar's float reciprocal
        const T r = (T)((float)s[2] * inv), g = (T)((float)s[1] * inv), b = (T)((float)s[0] * inv);
        if (cpad == 4) {
            typedef T vec4 __attribute__((ext_vector_type(4)));
            *reinterpret_cast<vec4*>(d) = vec4{r, g, b, (T)0.0f};
        } else {
            d[0] = r; d[1] = g; d[2] = b;
            for (int c = 3; c < cpad; ++c) d[c] = (T)0.0f;
        }
    }
}

// SPPF (the network's spatial-pyramid block): y1 = maxpool5(x), y2 = maxpool5(y1), y3 = maxpool5(y2), stride 1, "same" padding, written
// next to x in the concatenation buffer [B][H][W][4c] (x = channels [0, c), written by the block's first convolution).  One workgroup
// per (image, 16-byte channel chunk): the H x W plane of chunks sits in LDS (16 bytes per pixel), each pool is a row pass and a column pass.
template <typename V>
__device__ __forceinline__ V maxv(V a, V b)
{
    V r;
#pragma unroll
    for (int e = 0; e < (int)(sizeof(V) / sizeof(a[0])); ++e) r[e] = a[e] > b[e] ? a[e] : b[e];
    return r;
}
template <typename T>
__global__ void __launch_bounds__(256) sppf_pool_kernel(T* __restrict__ buf, int H, int W, int c, int q)
{
    // A workgroup owns q adjacent 16-byte channel chunks of one image: an item is (pixel, chunk of the group), chunk fastest, so that q consecutive
    // threads move 16 q contiguous bytes of a pixel (q = 4: a 64-byte run; one chunk per workgroup fetched a whole sector for every 16 bytes it used).
    typedef typename Elem<T>::vec vec;
    constexpr int E = Elem<T>::E;
    extern __shared__ __attribute__((aligned(16))) unsigned char plane[];     // [2][H * W * q] chunks: current planes, row-pass result
    const int groups = c / (E * q), b = blockIdx.x / groups, ch = (blockIdx.x % groups) * E * q, ld = 4 * c, P = H * W, items = P * q;
    vec* cur = reinterpret_cast<vec*>(plane);
    vec* tmp = cur + items;
    T* img = buf + (size_t)b * P * ld + ch;
    for (int it = threadIdx.x; it < items; it += 256) {
        const int p = it / q, j = it - p * q;
        cur[it] = *reinterpret_cast<const vec*>(img + (size_t)p * ld + j * E);
    }
    __syncthreads();
    for (int level = 1; level <= 3; ++level) {
        for (int it = threadIdx.x; it < items; it += 256) {   // max over the row window
            const int p = it / q, h = p / W, w = p - h * W;
            vec m = cur[it];
            for (int d = 1; d <= 2; ++d) {
                if (w - d >= 0) m = maxv(m, cur[it - d * q]);
                if (w + d < W) m = maxv(m, cur[it + d * q]);
            }
            tmp[it] = m;
        }
        __syncthreads();
        for (int it = threadIdx.x; it < items; it += 256) {   // max over the column window, into the planes and the level's channel slice
            const int p = it / q, j = it - p * q, h = p / W;
            vec m = tmp[it];
            for (int d = 1; d <= 2; ++d) {
                if (h - d >= 0) m = maxv(m, tmp[it - d * W * q]);
                if (h + d < H) m = maxv(m, tmp[it + d * W * q]);
            }
            cur[it] = m;      // (each thread rewrites only its own items of `cur`, which this pass does not read)
            *reinterpret_cast<vec*>(img + (size_t)p * ld + level * c + j * E) = m;
        }
        __syncthreads();
    }
}

// The head's torch.cat((upsample2x(a), b), 1) as one pass: out[b][y][x] = (a[b][y / 2][x / 2][0..ca), b[b][y][x][0..cb)), NHWC, 16-byte chunks
// (cpa, cpb: chunks per pixel of a and b -- the element type does not matter to a copy).
__global__ void __launch_bounds__(256) upsample_concat_kernel(const uint4* __restrict__ a, const uint4* __restrict__ b, uint4* __restrict__ out,
                                                              int H, int W, int cpa, int cpb, long long chunks)
{
    const int cpp = cpa + cpb;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < chunks; i += (long long)gridDim.x * blockDim.x) {
        const long long pix = i / cpp;
        const int k = (int)(i - pix * cpp);
        uint4 v;
        if (k < cpa) {
            const int x = (int)(pix % W), y = (int)((pix / W) % H);
            const long long img = pix / ((long long)W * H);
            v = a[((img * (H >> 1) + (y >> 1)) * (W >> 1) + (x >> 1)) * cpa + k];
        } else {
            v = b[pix * cpb + (k - cpa)];
        }
        out[i] = v;
    }
}

// Tile of the LDS-DMA kernel for a layer.  A launch lasts as long as its most loaded CU, i.e. ceil(workgroups / CUs) tiles' worth of matrix work
// (tile area x K is the same product whatever the shape), so among the shapes the output allows the one with the least
// ceil(workgroups / CUs) x area x (1 + thin / rows + thin / columns) wins.  `thin` prices what a smaller tile moves per MFMA: almost nothing
// next to a 64-cycle float32 MFMA (profiles/r03_conv_layers_f32.csv: 64 x 64 tiles win on most layers through the finer split alone), a lot next
// to a float16 one, whose 128 x 128 tile already keeps the LDS port 3/4 busy (measured: 64 x 64 tiles there run at 0.15 of the matrix peak).
#ifndef BF_CONV_F32_DEFAULT
#define BF_CONV_F32_DEFAULT 1
#endif
struct DmaTile { int bm, bn; };
inline DmaTile pick_dma_tile(long long M, int N, int n_cus, bool f32, bool split = false)
{
    static const DmaTile shapes[] = {{128, 128}, {128, 64}, {64, 128}, {64, 64}, {128, 32}};
    const double thin = f32 && !split ? 8.0 : 160.0;       // (the split float32 mode multiplies like float16: measured 8 / 24 / 64 / 160 -> 8.82 / 8.28 / 8.13 / 8.12 ms for the 60 layers)
    DmaTile best{128, N >= 128 ? 128 : (N <= 32 ? 32 : 64)};
    double best_cost = 1e300;
    for (const DmaTile& t : shapes) {
        if (t.bn == 32 ? N > 32 : N <= 32) continue;           // 32-channel tiles for layers of up to 32 channels, and only for those
        const long long wgs = ((M + t.bm - 1) / t.bm) * ((N + t.bn - 1) / t.bn);
        const double cost = (double)((wgs + n_cus - 1) / n_cus) * t.bm * t.bn * (1.0 + thin / t.bm + thin / t.bn);
        if (cost < best_cost) { best_cost = cost; best = t; }
    }
    return best;
}

template <typename T>
hipError_t launch_conv_t(const ConvArgs& a, const void* x, const void* w, const float* bias, void* y, bool cat, bool dma, unsigned x_bytes, unsigned x2_bytes,
                         unsigned w_bytes, hipStream_t stream)
{
    const T* xp = static_cast<const T*>(x);
    const T* wp = static_cast<const T*>(w);
    T* yp = static_cast<T*>(y);
    if (dma) {
        static int n_cus = 0;
        if (n_cus == 0) {
            int dev = 0, v = 0;
            if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) n_cus = v; else n_cus = 256;
        }
        // the stem's shape: a 4-channel input under a wide window -- the patch kernel, in float16 (0.27 -> 0.19 ms at batch 64; in float32 both kernels sit
        // on the 72 MFMAs per wave of the 4-channel K and the patch form measured 5 % slower: 0.68 against 0.64 ms).  $BF_CONV_PATCH=0 / 1: never / always (A/B).
        static const int patch_env = [] { const char* e = getenv("BF_CONV_PATCH"); return e ? atoi(e) : -1; }();
        const bool patch_wanted = conv_dma_switch(-1) == 2 ? true : (patch_env < 0 ? sizeof(T) == 2 : patch_env != 0);
        if (patch_wanted && !cat && a.C == 4 && a.N <= 32 && a.KH * a.KW >= 16 && (a.KW & 1) == 0 && ((a.KH * a.KW * 4 / (16 / (int)sizeof(T))) & 1) == 0 &&
            (sizeof(T) == 4 || ((a.stride & 1) == 0 && (a.pad & 1) == 0 && (a.W & 1) == 0))) {
            const int PH = 7 * a.stride + a.KH, PW = 15 * a.stride + a.KW;
            const int patch_bytes = (PH * PW * 4 * (int)sizeof(T) + 1023) & ~1023, w_lds = 32 * (a.wld * (int)sizeof(T) + 16);
            const int lds = patch_bytes + ((w_lds + 1023) & ~1023) + 256, tile = 128 * (32 + 16 / (int)sizeof(T)) * (int)sizeof(T);
            const int need = lds > tile ? lds : tile;
            if (need <= 64 * 1024) {
                auto kernel = conv_patch_kernel<T>;
                hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, need);
                if (e != hipSuccess) return e;
                const unsigned grid = (unsigned)a.B * (unsigned)((a.Ho + 7) / 8) * (unsigned)((a.Wo + 15) / 16);
                hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), (size_t)need, stream, xp, wp, bias, yp, a, x_bytes, w_bytes);
                return hipGetLastError();
            }
        }
        const bool split = sizeof(T) == 4 && conv_f32_mode(-1) == 1;
        const DmaTile t = pick_dma_tile(a.M, a.N, n_cus, sizeof(T) == 4 && !split, split);
        // Rows of 128 bytes (full cache lines per row, half the barriers per MFMA) pay where stages are many and the tile is 128 channels wide: float16
        // layers of 128+ channels with K >= 1024, or 1x1 layers with K >= 512 (profiles/r03_conv_layers_f16.csv; the stem and the 32- / 64-channel layers
        // lose to the coarser K padding and the smaller ring).  float32 stages are long as they are: 64-byte rows.  $BF_CONV_ROW=64|128 forces either.
        static const int row_env = [] { const char* e = getenv("BF_CONV_ROW"); return e ? atoi(e) : 0; }();
        const int K = a.KH * a.KW * a.C;
        bool wide_rows = row_env == 128 ? true : (row_env == 64 ? false : (sizeof(T) == 2 && a.N >= 128 && (K >= 1024 || (a.KH == 1 && K >= 512))));
        if (cat && a.x2 && (a.c1 % (8 * (16 / (int)sizeof(T)))) != 0) wide_rows = false;
        ConvArgs af = a;                                       // (see ConvArgs::fast_tap; $BF_CONV_TAP=0 keeps the general address form, for A/B runs)
        static const int tap_env = [] { const char* e = getenv("BF_CONV_TAP"); return e ? atoi(e) : 1; }();
        af.fast_tap = (tap_env != 0 && !cat && a.C / (16 / (int)sizeof(T)) >= ((wide_rows && !split) ? 8 : 4) && a.KH * a.KW <= 32) ? 1 : 0;
#define BF_DMA_LAUNCH(BM, BN, CPR, CAT, SPLIT)                                                                                                                          \
        hipLaunchKernelGGL((conv_dma_kernel<T, BM, BN, CPR, CAT, SPLIT>), dim3((unsigned)((a.M + BM - 1) / BM), (unsigned)((a.N + BN - 1) / BN)), dim3(256), 0, stream,  \
                           xp, wp, bias, yp, af, x_bytes, x2_bytes, w_bytes)
#define BF_DMA_PICK(BM, BN)                                                                                                    \
        do {                                                                                                                   \
            if constexpr (sizeof(T) == 4) {                                                                                    \
                if (split) { if (cat) BF_DMA_LAUNCH(BM, BN, 4, true, true); else BF_DMA_LAUNCH(BM, BN, 4, false, true); break; } \
            }                                                                                                                  \
            if (wide_rows) { if (cat) BF_DMA_LAUNCH(BM, BN, 8, true, false); else BF_DMA_LAUNCH(BM, BN, 8, false, false); }    \
            else { if (cat) BF_DMA_LAUNCH(BM, BN, 4, true, false); else BF_DMA_LAUNCH(BM, BN, 4, false, false); }              \
        } while (0)
        if (t.bm == 128 && t.bn == 128) BF_DMA_PICK(128, 128);
        else if (t.bm == 128 && t.bn == 64) BF_DMA_PICK(128, 64);
        else if (t.bm == 128) BF_DMA_PICK(128, 32);
        else if (t.bn == 128) BF_DMA_PICK(64, 128);
        else BF_DMA_PICK(64, 64);
#undef BF_DMA_PICK
#undef BF_DMA_LAUNCH
        return hipGetLastError();
    }
    const long long gx = (a.M + kBM - 1) / kBM;
#define BF_CONV_LAUNCH(BN, CAT) \
    hipLaunchKernelGGL((conv_igemm_kernel<T, BN, CAT>), dim3((unsigned)gx, (unsigned)((a.N + BN - 1) / BN)), dim3(256), 0, stream, xp, wp, bias, yp, a)
    if (a.N >= 128) { if (cat) BF_CONV_LAUNCH(128, true); else BF_CONV_LAUNCH(128, false); }
    else if (a.N <= 32) { if (cat) BF_CONV_LAUNCH(32, true); else BF_CONV_LAUNCH(32, false); }
    else { if (cat) BF_CONV_LAUNCH(64, true); else BF_CONV_LAUNCH(64, false); }
#undef BF_CONV_LAUNCH
    return hipGetLastError();
}

}  // namespace

hipError_t launch_upsample_concat(const void* a, const void* b, void* out, int B, int H, int W, int ca, int cb, int elem_bytes, hipStream_t stream)
{
    const int E = 16 / elem_bytes;
    if (B <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1) || ca < E || cb < E || (ca % E) || (cb % E)) return hipErrorInvalidValue;
    const long long chunks = (long long)B * H * W * ((ca + cb) / E), blocks = (chunks + 255) / 256;
    hipLaunchKernelGGL(upsample_concat_kernel, dim3((unsigned)(blocks < 262144 ? blocks : 262144)), dim3(256), 0, stream, static_cast<const uint4*>(a),
                       static_cast<const uint4*>(b), static_cast<uint4*>(out), H, W, ca / E, cb / E, chunks);
    return hipGetLastError();
}

hipError_t launch_sppf_pool(void* buf, int B, int H, int W, int c, int elem_bytes, hipStream_t stream)
{
    const int E = 16 / elem_bytes;
    if (B <= 0 || H <= 0 || W <= 0 || c < E || (c % E) != 0 || (long long)H * W > 2048) return hipErrorInvalidValue;
    const int chunks = c / E;
    int q = 4;                                                // chunks per workgroup: as many as divide the channel count and fit 64 KiB of LDS
    while (q > 1 && ((chunks % q) != 0 || (size_t)2 * H * W * q * 16 > 64 * 1024)) q >>= 1;
    const size_t lds = (size_t)2 * H * W * q * 16;
    const void* fn = elem_bytes == 4 ? reinterpret_cast<const void*>(sppf_pool_kernel<float>) : reinterpret_cast<const void*>(sppf_pool_kernel<_Float16>);
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    const dim3 grid((unsigned)(B * (chunks / q)));
    if (elem_bytes == 4) hipLaunchKernelGGL(sppf_pool_kernel<float>, grid, dim3(256), lds, stream, static_cast<float*>(buf), H, W, c, q);
    else hipLaunchKernelGGL(sppf_pool_kernel<_Float16>, grid, dim3(256), lds, stream, static_cast<_Float16*>(buf), H, W, c, q);
    return hipGetLastError();
}

hipError_t launch_preprocess_bgr8(const void* frames, void* out, long long pixels, int cpad, int elem_bytes, hipStream_t stream)
{
    if (pixels <= 0 || cpad < 3) return hipErrorInvalidValue;
    const long long blocks = (pixels + 255) / 256;
    const dim3 grid((unsigned)(blocks < 65536 ? blocks : 65536));
    if (elem_bytes == 4) hipLaunchKernelGGL(preprocess_kernel<float>, grid, dim3(256), 0, stream, static_cast<const uint8_t*>(frames), static_cast<float*>(out), pixels, cpad);
    else hipLaunchKernelGGL(preprocess_kernel<_Float16>, grid, dim3(256), 0, stream, static_cast<const uint8_t*>(frames), static_cast<_Float16*>(out), pixels, cpad);
    return hipGetLastError();
}

// Elements per packed weight row [KH][KW][C]: zero-padded to whole 128-byte stages (the longest stage any kernel here walks).
int conv_weight_row(int elem_bytes, int kh, int kw, int c)
{
    const int per = 128 / elem_bytes;
    return (kh * kw * c + per - 1) / per * per;
}

// Which convolution kernel launch_conv2d_nhwc picks: 1 = LDS-DMA staging where its conditions hold (the default; $BF_CONV_DMA=0 changes it),
// 0 = the register-staged kernel always.  value < 0 only reads.  Returns the previous setting.
int conv_dma_switch(int value)
{
    static int state = [] { const char* e = getenv("BF_CONV_DMA"); return e ? (atoi(e) != 0 ? 1 : 0) : 1; }();
    const int old = state;
    if (value >= 0) state = value > 2 ? 1 : value;         // 2: LDS-DMA kernels, and the patch kernel for every layer it can take (tests)
    return old;
}

// How the float32 LDS-DMA kernels multiply: 0 = v_mfma_f32_32x32x2_f32 on the operands as they are, 1 = three-way bfloat16 split of both operands and
// six v_mfma_f32_32x32x16_bf16 per product (conv_dma_body; same accuracy, 2.67 x the matrix rate).  $BF_CONV_F32=native|split sets the initial mode;
// value < 0 only reads.  Returns the previous setting.  (The register-staged kernel and the stem's patch kernel are native only.)
int conv_f32_mode(int value)
{
    static int state = [] { const char* e = getenv("BF_CONV_F32"); return e ? (strcmp(e, "split") == 0 ? 1 : 0) : BF_CONV_F32_DEFAULT; }();
    const int old = state;
    if (value >= 0) state = value != 0 ? 1 : 0;
    return old;
}

// elem_bytes: 2 (float16) or 4 (float32).  x2 != nullptr or ld1 != C or up1: the 1x1 window over a virtual concatenation (ConvArgs).
hipError_t launch_conv2d_nhwc(int elem_bytes, const void* x, const void* w, const float* bias, void* y, int B, int H, int W, int C, int N, int KH, int KW,
                              int stride, int pad, int act, int ldy, const void* res, int ldr, const void* x2, int c1, int ld1, int ld2, int up1,
                              hipStream_t stream)
{
    const int E = 16 / elem_bytes;
    if (elem_bytes != 2 && elem_bytes != 4) return hipErrorInvalidValue;
    if (ldy < N || (res && ldr < N)) return hipErrorInvalidValue;
    if (B <= 0 || H <= 0 || W <= 0 || N <= 0 || KH <= 0 || KW <= 0 || stride <= 0 || pad < 0) return hipErrorInvalidValue;
    if (C < 4 || (C & (C - 1)) != 0 || ((KW * C) % E) != 0) return hipErrorInvalidValue;       // whole 16-byte chunks per window row
    // float16, C = 4: a chunk is two pixels, which must leave the image together -- even window starts, even width
    if (C < E && ((stride & 1) || (pad & 1) || (W & 1))) return hipErrorInvalidValue;
    // every 1x1 / stride 1 / unpadded layer takes the "virtual concatenation" kernel, also with one dense source: its addresses are the pixel's base plus
    // a stage offset -- no pixel decomposition (two integer divisions per piece), no window arithmetic; the short layers are bound by exactly such
    // vector instructions (a 64 -> 32 channel 1x1 layer: 460 VALU instructions per wave around 4 MFMAs, vector issue 63 % busy)
    const bool cat = x2 != nullptr || up1 != 0 || ld1 != C || (KH == 1 && KW == 1 && stride == 1 && pad == 0);
    if (cat) {
        if (KH != 1 || KW != 1 || stride != 1 || pad != 0 || c1 < E || c1 > C || (c1 % E) || (ld1 % E) || ld1 < c1) return hipErrorInvalidValue;
        if (c1 < C && (!x2 || (ld2 % E) || ld2 < C - c1 || (reinterpret_cast<uintptr_t>(x2) & 15))) return hipErrorInvalidValue;
        if (up1 && ((H & 1) || (W & 1))) return hipErrorInvalidValue;
    }
    if ((reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(w) & 15)) return hipErrorInvalidValue;
    ConvArgs a;
    a.B = B; a.H = H; a.W = W; a.C = C; a.N = N; a.KH = KH; a.KW = KW; a.stride = stride; a.pad = pad; a.act = act;
    a.Ho = (H + 2 * pad - KH) / stride + 1;
    a.Wo = (W + 2 * pad - KW) / stride + 1;
    if (a.Ho <= 0 || a.Wo <= 0) return hipErrorInvalidValue;
    a.ldy = ldy; a.ldr = ldr; a.res = res; a.fast_tap = 0;
    a.wide = ((N % E) == 0 && (ldy % E) == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0 &&
              (!res || ((ldr % E) == 0 && (reinterpret_cast<uintptr_t>(res) & 15) == 0))) ? 1 : 0;
    a.c_shift = 0;
    while ((1 << a.c_shift) < C) ++a.c_shift;
    a.M = (long long)B * a.Ho * a.Wo;
    if (a.M > 0x7fffffffLL) return hipErrorInvalidValue;       // (the kernel splits pixel indices in 32 bits)
    a.x2 = x2; a.c1 = cat ? c1 : C; a.ld1 = cat ? ld1 : C; a.ld2 = ld2; a.up1 = up1;
    a.d_img = make_fastdiv((unsigned)(a.Ho * a.Wo)); a.d_row = make_fastdiv((unsigned)a.Wo);
    {
        const int tx = (a.Wo + 15) / 16, ty = (a.Ho + 7) / 8, ppc = 16 / (4 * elem_bytes), pw = 15 * stride + KW;
        a.d_p0 = make_fastdiv((unsigned)(tx * ty)); a.d_p1 = make_fastdiv((unsigned)tx);
        a.d_p2 = make_fastdiv((unsigned)(pw / ppc > 0 ? pw / ppc : 1));
    }
    // The LDS-DMA kernel addresses its operands through 32-bit buffer offsets with the top bit as the out-of-range mark: every operand tensor
    // below 2 GiB, and a two-source layer switching source on a stage boundary.  BF_CONV_DMA=0 selects the register-staged kernel (A/B runs).
    const int want_dma = conv_dma_switch(-1);
    const unsigned long long eb = (unsigned long long)elem_bytes;
    const unsigned long long src1_pixels = cat && up1 ? (unsigned long long)B * (H / 2) * (W / 2) : (unsigned long long)B * H * W;
    const unsigned long long xb = cat ? ((src1_pixels - 1) * (unsigned long long)ld1 + (unsigned long long)a.c1) * eb : src1_pixels * (unsigned long long)C * eb;
    const unsigned long long x2b = (cat && x2) ? (((unsigned long long)B * H * W - 1) * (unsigned long long)ld2 + (unsigned long long)(C - a.c1)) * eb : 0ull;
    a.wld = conv_weight_row(elem_bytes, KH, KW, C);
    a.d_p3 = make_fastdiv((unsigned)(a.wld * elem_bytes + 16));
    const unsigned long long wb = (unsigned long long)N * (unsigned long long)a.wld * eb;
    const bool dma = want_dma != 0 && xb < 0x7ffffff0ull && x2b < 0x7ffffff0ull && wb < 0x7ffffff0ull && (!(cat && x2) || (a.c1 % (4 * E)) == 0);
    return elem_bytes == 4 ? launch_conv_t<float>(a, x, w, bias, y, cat, dma, (unsigned)xb, (unsigned)x2b, (unsigned)wb, stream)
                           : launch_conv_t<_Float16>(a, x, w, bias, y, cat, dma, (unsigned)xb, (unsigned)x2b, (unsigned)wb, stream);
}

}  // namespace bf
