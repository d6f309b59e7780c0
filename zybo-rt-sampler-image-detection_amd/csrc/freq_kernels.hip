// freq_kernels.hip -- frequency-domain (phase-steered) beamformers on gfx950: delay-and-sum and MVDR.
//
// Reference for the delay-and-sum flavour: PC/application/realtime_scripts/beam_forming_algorithm.py:30-70 and
// calc_phase_shift_cartesian.py:37-50 (NumPy, complex128):
//     X = rfft(signal, axis=0)[bins];  B[k, d] = sum_m X[k, m] * exp(-j 2 pi f_k tau[d, m]);  P[d] = sum_k |B[k, d]|^2
// MVDR (BASELINE.json config 3) has NO counterpart in the reference; it is defined here as
//     R_k = (1/F) sum_frames x x^H + delta * tr(R_k)/M * I;   P[d] = sum_k 1 / (v_{k,d}^H R_k^{-1} v_{k,d})
// with v = conj(a), a_{k,d}[m] = exp(-j 2 pi f_k tau[d, m]) the reference's phase shifts (its beam is sum_m a_m x_m = v^H x).
//
// This is the only GEMM-shaped work in the repository, so it is the only place MFMA is used: one complex-GEMM
// kernel built on v_mfma_f32_32x32x2_f32 (f32 in, f32 accumulate: bit-wise an fmaf chain, no reduced precision),
// specialised by its epilogue:
//     EPI_POWER   C = X_k^T A_k      (frames x dirs)  ->  P[f, d] += |C|^2 over the bins      (phase-steer DAS)
//     EPI_STORE   C = X_k X_k^H / F  (mics x mics)    ->  R_k                                  (covariance)
//     EPI_MVDR    C = L_k^-1 conj(A_k) (mics x dirs)  ->  P[d] += 1 / sum_m |C[m, d]|^2         (MVDR quadratic form)
// Operands are stored as separate re / im planes, K-major with the tile index contiguous ([batch][k][i]), so that the
// MFMA lane map (A: lane l holds A[i = l & 31][k = l >> 5]; B: B[k = l >> 5][j = l & 31]) reads 128-byte rows.
// A complex product is four real MFMAs; conj(B) is a sign flip of the loaded imaginary part.
#include <hip/hip_runtime.h>
#include <type_traits>
#include <stdint.h>
#include <cstdlib>
#include <cstring>

#include "f32_split.h"

namespace bf {

// How the bin-reducing float32 GEMMs (phase-steer power map, MVDR quadratic form) multiply: 0 = v_mfma_f32_32x32x2_f32 on the operands as they are,
// 1 = three-way bfloat16 split of both operands and six v_mfma_f32_32x32x16_bf16 per product (f32_split.h: float32 accuracy, 2.67 x the matrix rate).
// $BF_GEMM_F32=native|split sets the initial mode; value < 0 only reads.  Returns the previous setting.
#ifndef BF_GEMM_F32_DEFAULT
#define BF_GEMM_F32_DEFAULT 1
#endif
int gemm_f32_mode(int value)
{
    static int state = [] { const char* e = getenv("BF_GEMM_F32"); return e ? (strcmp(e, "split") == 0 ? 1 : 0) : BF_GEMM_F32_DEFAULT; }();
    const int old = state;
    if (value >= 0) state = value != 0 ? 1 : 0;
    return old;
}

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

enum { EPI_POWER = 0, EPI_STORE = 1, EPI_MVDR = 2 };

struct GemmArgs {
    const float* a_re; const float* a_im;   // [batch][K][I]
    const float* b_re; const float* b_im;   // [batch][K][J]
    float* out0; float* out1;               // EPI_POWER: P[I][J] (out0);  EPI_STORE: C planes [batch][I][J];  EPI_MVDR: P[J] (out0)
    int I, J, K, batch;
    int conj_b;
    float scale;                            // EPI_STORE: C *= scale
};

// C/D lane map of v_mfma_f32_32x32x2_f32: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
__device__ __forceinline__ int acc_row(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

// One wave per 32 x 32 tile of C; 4 waves (4 column tiles) per workgroup.
template <int EPI>
__global__ void __launch_bounds__(256) cgemm_kernel(GemmArgs g)
{
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int i0 = blockIdx.y * 32;
    const int j0 = (blockIdx.x * 4 + wave) * 32;
    if (j0 >= g.J) return;
    const int li = lane & 31, lk = lane >> 5;
    const bool ai_ok = i0 + li < g.I, bj_ok = j0 + li < g.J;

    f32x16 acc_p;   // EPI_POWER: running |C|^2 over the batch (bins)
    float inv_sum = 0.0f;   // EPI_MVDR
#pragma unroll
    for (int r = 0; r < 16; ++r) acc_p[r] = 0.0f;

    const int b_begin = (EPI == EPI_STORE) ? blockIdx.z : 0;
    const int b_end = (EPI == EPI_STORE) ? blockIdx.z + 1 : g.batch;

    for (int b = b_begin; b < b_end; ++b) {
        // EPI_MVDR needs all I rows of a column: loop the row tiles inside (I <= 128 -> up to 4 tiles)
        const int row_tiles = (EPI == EPI_MVDR) ? (g.I + 31) / 32 : 1;
        float colsum = 0.0f;
        for (int rt = 0; rt < row_tiles; ++rt) {
            const int ib = (EPI == EPI_MVDR) ? rt * 32 : i0;
            const bool a_ok = (EPI == EPI_MVDR) ? (ib + li < g.I) : ai_ok;
            f32x16 cre, cim;
#pragma unroll
            for (int r = 0; r < 16; ++r) { cre[r] = 0.0f; cim[r] = 0.0f; }
            const float* are = g.a_re + ((size_t)b * g.K) * g.I + ib + li;
            const float* aim = g.a_im + ((size_t)b * g.K) * g.I + ib + li;
            const float* bre = g.b_re + ((size_t)b * g.K) * g.J + j0 + li;
            const float* bim = g.b_im + ((size_t)b * g.K) * g.J + j0 + li;
            for (int k = 0; k < g.K; k += 2) {
                const int kk = k + lk;
                const bool k_ok = kk < g.K;
                const float xr = (a_ok && k_ok) ? are[(size_t)kk * g.I] : 0.0f;
                const float xi = (a_ok && k_ok) ? aim[(size_t)kk * g.I] : 0.0f;
                const float yr = (bj_ok && k_ok) ? bre[(size_t)kk * g.J] : 0.0f;
                float yi = (bj_ok && k_ok) ? bim[(size_t)kk * g.J] : 0.0f;
                if (g.conj_b) yi = -yi;
                // (xr + j xi)(yr + j yi)
                cre = __builtin_amdgcn_mfma_f32_32x32x2f32(xr, yr, cre, 0, 0, 0);
                cre = __builtin_amdgcn_mfma_f32_32x32x2f32(-xi, yi, cre, 0, 0, 0);
                cim = __builtin_amdgcn_mfma_f32_32x32x2f32(xr, yi, cim, 0, 0, 0);
                cim = __builtin_amdgcn_mfma_f32_32x32x2f32(xi, yr, cim, 0, 0, 0);
            }
            if constexpr (EPI == EPI_POWER) {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc_p[r] += cre[r] * cre[r] + cim[r] * cim[r];
            } else if constexpr (EPI == EPI_STORE) {
                const int col = j0 + li;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = ib + acc_row(r, lane);
                    if (row < g.I && col < g.J) {
                        const size_t o = ((size_t)b * g.I + row) * g.J + col;
                        g.out0[o] = cre[r] * g.scale;
                        g.out1[o] = cim[r] * g.scale;
                    }
                }
            } else {
                float s = 0.0f;
#pragma unroll
                for (int r = 0; r < 16; ++r) s += cre[r] * cre[r] + cim[r] * cim[r];   // rows outside I were zero operands
                colsum += s;
            }
        }
        if constexpr (EPI == EPI_MVDR) {
            colsum += __shfl_xor(colsum, 32, 64);      // the other half of the rows of this column
            inv_sum += 1.0f / colsum;
        }
    }
    if constexpr (EPI == EPI_POWER) {
        const int col = j0 + li;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = i0 + acc_row(r, lane);
            if (row < g.I && col < g.J) g.out0[(size_t)row * g.J + col] = acc_p[r];
        }
    } else if constexpr (EPI == EPI_MVDR) {
        if (lane < 32 && j0 + li < g.J) g.out0[j0 + li] = inv_sum;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// The two big GEMMs (phase-steer DAS power, MVDR quadratic form) reduce over the bins, so they get a kernel that keeps
// the matrix cores fed: a wave owns one 32-column tile of B (32 directions) and RT row tiles of A, and walks its share
// of the bins.  Per bin it loads a 64-deep panel of B into registers once (64 VGPRs: re/im x 32 k-steps), then per row
// tile a panel of A (64 VGPRs) and issues 128 back-to-back MFMAs (4 real MFMAs per complex k-step).  Two such waves
// per SIMD (<= 256 VGPRs each) hide each other's panel loads.  The 4 waves of a workgroup take bins b0+w, b0+w+4, ...
// and are summed in wave order through LDS; bin groups (blockIdx.y) write partial planes that reduce_planes_kernel
// sums in group order -- no atomics, bit-reproducible.
struct Panel { float re[32], im[32]; };

// One 64-deep panel in MFMA operand order: step s2 holds row k0 + 2 s2 + (lane >> 5), column (lane & 31) of the tile.
// `re` / `im` are wave-uniform plane pointers (batch applied), `voff` the lane's element offset (lk * ld + column), so
// every load is "scalar base + 32-bit lane offset" and costs no vector address arithmetic.
__device__ __forceinline__ void load_panel(Panel& p, const float* __restrict__ re, const float* __restrict__ im, int ld, int k0, int K, int lk,
                                           int voff, bool ok, bool conj)
{
#pragma unroll
    for (int s2 = 0; s2 < 32; ++s2) { p.re[s2] = 0.0f; p.im[s2] = 0.0f; }
    if (ok) {
        if (k0 + 64 <= K) {                      // whole panel in range: straight loads
#pragma unroll
            for (int s2 = 0; s2 < 32; ++s2) {
                const size_t row = (size_t)(k0 + 2 * s2) * (size_t)ld;     // uniform
                p.re[s2] = (re + row)[voff];
                p.im[s2] = (im + row)[voff];
            }
        } else {
#pragma unroll
            for (int s2 = 0; s2 < 32; ++s2) {
                const int k = k0 + 2 * s2;       // uniform
                if (k + lk < K) {
                    const size_t row = (size_t)k * (size_t)ld;
                    p.re[s2] = (re + row)[voff];
                    p.im[s2] = (im + row)[voff];
                }
            }
        }
        if (conj) {
#pragma unroll
            for (int s2 = 0; s2 < 32; ++s2) p.im[s2] = -p.im[s2];
        }
    }
}

// SPLIT: the same products on the bfloat16 pipes (f32_split.h).  The panels' register order is already what v_mfma_f32_32x32x16_bf16 wants: a lane's
// registers [8 t, 8 t + 8) are its eight values of K step t (k = k0 + 16 t + 2 i + (lane >> 5)), for A and B alike.  A bin's B panel is split once
// (24 registers per K step instead of 16), an A panel's step right before its 24 MFMAs (4 real products x 6 part products; -Ai Bi by flipping the
// sign bits of Ai's parts in place after Ai Br has been issued).
template <int EPI, int RT, bool SPLIT>
__global__ void __launch_bounds__(256, 2) cgemm_bins_kernel(GemmArgs g, int row_groups, int bins_per_group)
{
    static_assert(EPI == EPI_POWER || EPI == EPI_MVDR, "bin-reducing epilogues");
    using split::Split3;
    extern __shared__ float red[];              // [4 waves][RT * 16][64 lanes] (POWER) or [4][32] (MVDR)
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int li = lane & 31, lk = lane >> 5;
    const int rg = (int)(blockIdx.x % (unsigned)row_groups);     // row group fastest: the groups that share a B tile run together
    const int j0 = (int)(blockIdx.x / (unsigned)row_groups) * 32;
    const int b0 = blockIdx.y * bins_per_group, b1 = min(b0 + bins_per_group, g.batch);
    const bool bj_ok = j0 + li < g.J;
    const bool single = g.K <= 64;              // one panel: B is loaded once per bin and shared by the row tiles
    const int row_tiles = (EPI == EPI_MVDR) ? (g.I + 31) / 32 : RT;
    const int rt_base = (EPI == EPI_MVDR) ? 0 : rg * RT;

    f32x16 acc_p[RT];
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc_p[t][r] = 0.0f;
    float inv_sum = 0.0f;

    const int b_voff = lk * g.J + j0 + li;
    for (int b = b0 + wave; b < b1; b += 4) {
        const float* bre = g.b_re + ((size_t)b * g.K) * g.J;      // wave-uniform plane pointers
        const float* bim = g.b_im + ((size_t)b * g.K) * g.J;
        const float* are = g.a_re + ((size_t)b * g.K) * g.I;
        const float* aim = g.a_im + ((size_t)b * g.K) * g.I;
        Panel Bp;
        Split3 Bs_re[SPLIT ? 4 : 1], Bs_im[SPLIT ? 4 : 1];
        auto split_b = [&]() {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                Bs_re[t] = split::split3(split::float4v{Bp.re[8 * t], Bp.re[8 * t + 1], Bp.re[8 * t + 2], Bp.re[8 * t + 3]},
                                         split::float4v{Bp.re[8 * t + 4], Bp.re[8 * t + 5], Bp.re[8 * t + 6], Bp.re[8 * t + 7]});
                Bs_im[t] = split::split3(split::float4v{Bp.im[8 * t], Bp.im[8 * t + 1], Bp.im[8 * t + 2], Bp.im[8 * t + 3]},
                                         split::float4v{Bp.im[8 * t + 4], Bp.im[8 * t + 5], Bp.im[8 * t + 6], Bp.im[8 * t + 7]});
            }
        };
        if (single) {
            load_panel(Bp, bre, bim, g.J, 0, g.K, lk, b_voff, bj_ok, g.conj_b != 0);
            if constexpr (SPLIT) split_b();
        }
        float colsum = 0.0f;

        auto row_tile = [&](int rt, f32x16* accp) {
            const int ib = (rt_base + rt) * 32;
            const bool a_ok = ib + li < g.I;
            const int a_voff = lk * g.I + ib + li;
            f32x16 cre, cim;
#pragma unroll
            for (int r = 0; r < 16; ++r) { cre[r] = 0.0f; cim[r] = 0.0f; }
            // MVDR: the A operand is L^-1 (transposed planes), lower triangular -- rows [ib, ib + 32) meet columns k < ib + 32 only, so the k-steps
            // beyond are products with zeros and are not issued (64 microphones: 192 instead of 256 MFMAs per bin and column tile).
            const int kmax = (EPI == EPI_MVDR) ? min(g.K, ib + 32) : g.K;
            if constexpr (SPLIT) {
                // the A operand arrives one 16-deep K step at a time (8 + 8 registers), the next step's loads in flight under this step's 24 MFMAs:
                // with the bin's split B panel resident (96 registers) a whole 64-deep A panel would not fit beside the accumulators
                auto load_a = [&](float (&re8)[8], float (&im8)[8], int kb) {
                    const bool whole = kb + 16 <= g.K;                                   // (uniform)
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const int k = kb + 2 * i;
                        const size_t row = (size_t)k * (size_t)g.I;
                        const bool ok = a_ok && (whole || k + lk < g.K);
                        re8[i] = ok ? (are + row)[a_voff] : 0.0f;
                        im8[i] = ok ? (aim + row)[a_voff] : 0.0f;
                    }
                };
                auto step = [&](const float (&re8)[8], const float (&im8)[8], const Split3& br, const Split3& bi) {
                    const Split3 ar = split::split3(split::float4v{re8[0], re8[1], re8[2], re8[3]}, split::float4v{re8[4], re8[5], re8[6], re8[7]});
                    Split3 ai = split::split3(split::float4v{im8[0], im8[1], im8[2], im8[3]}, split::float4v{im8[4], im8[5], im8[6], im8[7]});
                    split::mfma6(cre, ar, br);
                    split::mfma6(cim, ar, bi);
                    split::mfma6(cim, ai, br);
                    split::negate(ai);
                    split::mfma6(cre, ai, bi);
                };
                float r0[8], i0[8], r1[8], i1[8];
                for (int k0 = 0; k0 < kmax; k0 += 64) {
                    if (!single) {
                        load_panel(Bp, bre, bim, g.J, k0, g.K, lk, b_voff, bj_ok, g.conj_b != 0);
                        split_b();
                    }
                    load_a(r0, i0, k0);
                    if (k0 + 16 < kmax) load_a(r1, i1, k0 + 16);
                    step(r0, i0, Bs_re[0], Bs_im[0]);
                    if (k0 + 16 < kmax) {                                               // (uniform)
                        if (k0 + 32 < kmax) load_a(r0, i0, k0 + 32);
                        step(r1, i1, Bs_re[1], Bs_im[1]);
                        if (k0 + 32 < kmax) {
                            if (k0 + 48 < kmax) load_a(r1, i1, k0 + 48);
                            step(r0, i0, Bs_re[2], Bs_im[2]);
                            if (k0 + 48 < kmax) step(r1, i1, Bs_re[3], Bs_im[3]);
                        }
                    }
                }
            } else
            for (int k0 = 0; k0 < kmax; k0 += 64) {
                const bool both = k0 + 32 < kmax;              // (uniform) the panel's second half is needed
                Panel Ap;
                load_panel(Ap, are, aim, g.I, k0, both ? g.K : min(g.K, k0 + 32), lk, a_voff, a_ok, false);
                if (!single) load_panel(Bp, bre, bim, g.J, k0, g.K, lk, b_voff, bj_ok, g.conj_b != 0);
#pragma unroll
                for (int s2 = 0; s2 < 16; ++s2) {
                    // (xr + j xi)(yr + j yi)
                    cre = __builtin_amdgcn_mfma_f32_32x32x2f32(Ap.re[s2], Bp.re[s2], cre, 0, 0, 0);
                    cim = __builtin_amdgcn_mfma_f32_32x32x2f32(Ap.re[s2], Bp.im[s2], cim, 0, 0, 0);
                    cre = __builtin_amdgcn_mfma_f32_32x32x2f32(-Ap.im[s2], Bp.im[s2], cre, 0, 0, 0);
                    cim = __builtin_amdgcn_mfma_f32_32x32x2f32(Ap.im[s2], Bp.re[s2], cim, 0, 0, 0);
                }
                if (both) {
#pragma unroll
                    for (int s2 = 16; s2 < 32; ++s2) {
                        cre = __builtin_amdgcn_mfma_f32_32x32x2f32(Ap.re[s2], Bp.re[s2], cre, 0, 0, 0);
                        cim = __builtin_amdgcn_mfma_f32_32x32x2f32(Ap.re[s2], Bp.im[s2], cim, 0, 0, 0);
                        cre = __builtin_amdgcn_mfma_f32_32x32x2f32(-Ap.im[s2], Bp.im[s2], cre, 0, 0, 0);
                        cim = __builtin_amdgcn_mfma_f32_32x32x2f32(Ap.im[s2], Bp.re[s2], cim, 0, 0, 0);
                    }
                }
            }
            if constexpr (EPI == EPI_POWER) {
#pragma unroll
                for (int r = 0; r < 16; ++r) (*accp)[r] += cre[r] * cre[r] + cim[r] * cim[r];
            } else {
                float sq = 0.0f;
#pragma unroll
                for (int r = 0; r < 16; ++r) sq += cre[r] * cre[r] + cim[r] * cim[r];   // rows outside I were zero operands
                colsum += sq;
            }
        };
        if constexpr (EPI == EPI_POWER) {
#pragma unroll
            for (int t = 0; t < RT; ++t) row_tile(t, &acc_p[t]);
        } else {
            for (int t = 0; t < row_tiles; ++t) row_tile(t, nullptr);
            colsum += __shfl_xor(colsum, 32, 64);      // the other half of the rows of this column
            inv_sum += 1.0f / colsum;
        }
    }

    // ---- the 4 waves' partial sums, added in wave order
    if constexpr (EPI == EPI_POWER) {
#pragma unroll
        for (int t = 0; t < RT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) red[((wave * RT + t) * 16 + r) * 64 + lane] = acc_p[t][r];
        __syncthreads();
        float* out = g.out0 + (size_t)blockIdx.y * g.I * g.J;
        const int col = j0 + li;
        for (int e = wave; e < RT * 16; e += 4) {            // entry e = (row tile, register)
            const int t = e >> 4, r = e & 15;
            float v = red[((0 * RT + t) * 16 + r) * 64 + lane];
            v += red[((1 * RT + t) * 16 + r) * 64 + lane];
            v += red[((2 * RT + t) * 16 + r) * 64 + lane];
            v += red[((3 * RT + t) * 16 + r) * 64 + lane];
            const int row = (rt_base + t) * 32 + acc_row(r, lane);
            if (row < g.I && col < g.J) out[(size_t)row * g.J + col] = v;
        }
    } else {
        if (lane < 32) red[wave * 32 + lane] = inv_sum;
        __syncthreads();
        if (wave == 0 && lane < 32 && j0 + li < g.J)
            g.out0[(size_t)blockIdx.y * g.J + j0 + li] = ((red[lane] + red[32 + lane]) + red[64 + lane]) + red[96 + lane];
    }
}

// out[i] = sum over the planes, in plane order
__global__ void __launch_bounds__(256) reduce_planes_kernel(const float* __restrict__ planes, int n_planes, size_t plane, float* __restrict__ out)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < plane; i += (size_t)gridDim.x * blockDim.x) {
        float v = planes[i];
        for (int p = 1; p < n_planes; ++p) v += planes[(size_t)p * plane + i];
        out[i] = v;
    }
}

// ---- DFT on the matrix cores.  X[k][r] = sum_n W[n][k] s[r][n] over the rows r = (frame, mic): a real GEMM per plane
// (cos / -sin twiddles), 32 rows x all bins per wave.  The twiddles are built once per (N, bin range) in float64 with the
// exact angle reduction (k n mod N) and staged through LDS in chunks of 64 samples; the signal operand is read straight
// from the frames (each lane walks its own row).  Writes both layouts the GEMMs need: [K][M][F] and [K][F][M].
__global__ void __launch_bounds__(256) twiddle_kernel(float* __restrict__ wc, float* __restrict__ ws, int n_samples, int bin_lo, int n_bins, int kp)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_samples * kp; i += gridDim.x * blockDim.x) {
        const int n = i / kp, t = i - n * kp;
        float c = 0.0f, sn = 0.0f;
        if (t < n_bins) {
            const long long idx = ((long long)(bin_lo + t) * n) % n_samples;
            const double ang = -2.0 * 3.14159265358979323846 * (double)idx / (double)n_samples;
            c = (float)cos(ang);
            sn = (float)sin(ang);
        }
        wc[i] = c;
        ws[i] = sn;
    }
}

template <int KT>   // 32-bin tiles per wave; blockIdx.y walks groups of 32 KT bins of a table whose rows hold kp_total bins
__global__ void __launch_bounds__(256) dft_mfma_kernel(const float* __restrict__ frames, const int32_t* __restrict__ mics, int m_total, int n_samples,
                                                       int n_frames, int n_mics, int n_bins, int kp_total, const float* __restrict__ wc,
                                                       const float* __restrict__ ws, float* __restrict__ xre_mf, float* __restrict__ xim_mf,
                                                       float* __restrict__ xre_fm, float* __restrict__ xim_fm)
{
    constexpr int KP = 32 * KT, CH = 64;        // bins of this group; samples per staged twiddle chunk
    const int kofs = blockIdx.y * KP;
    extern __shared__ float tw[];               // [2][CH][KP]
    float* tc = tw;
    float* ts = tw + CH * KP;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 31, lk = lane >> 5;
    const long long rows = (long long)n_frames * n_mics;
    const long long r = ((long long)blockIdx.x * 4 + wave) * 32 + li;       // this lane's row (as the B operand's column)
    const bool r_ok = r < rows;
    const int f = r_ok ? (int)(r / n_mics) : 0, m = r_ok ? (int)(r - (long long)f * n_mics) : 0;
    const float* row = frames + ((size_t)f * m_total + (r_ok ? mics[m] : 0)) * n_samples;
    f32x16 are_[KT], aim_[KT];
#pragma unroll
    for (int t = 0; t < KT; ++t)
#pragma unroll
        for (int q = 0; q < 16; ++q) { are_[t][q] = 0.0f; aim_[t][q] = 0.0f; }
    for (int n0 = 0; n0 < n_samples; n0 += CH) {
        __syncthreads();
        for (int i = threadIdx.x; i < CH * KP; i += blockDim.x) {
            const int nn = i / KP, kk = i - nn * KP, n = n0 + nn;
            const bool ok = n < n_samples && kofs + kk < kp_total;
            tc[i] = ok ? wc[(size_t)n * kp_total + kofs + kk] : 0.0f;
            ts[i] = ok ? ws[(size_t)n * kp_total + kofs + kk] : 0.0f;
        }
        __syncthreads();
#pragma unroll 4
        for (int s2 = 0; s2 < CH / 2; ++s2) {
            const int n = n0 + 2 * s2 + lk;
            const float b = (r_ok && n < n_samples) ? row[n] : 0.0f;
#pragma unroll
            for (int t = 0; t < KT; ++t) {
                const float c = tc[(2 * s2 + lk) * KP + 32 * t + li];
                const float sn = ts[(2 * s2 + lk) * KP + 32 * t + li];
                are_[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(c, b, are_[t], 0, 0, 0);
                aim_[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(sn, b, aim_[t], 0, 0, 0);
            }
        }
    }
    if (!r_ok) return;
#pragma unroll
    for (int t = 0; t < KT; ++t)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int k = kofs + 32 * t + acc_row(q, lane);
            if (k < n_bins) {
                const size_t mf = ((size_t)k * n_mics + m) * n_frames + f, fm = ((size_t)k * n_frames + f) * n_mics + m;
                xre_mf[mf] = are_[t][q]; xim_mf[mf] = aim_[t][q];
                xre_fm[fm] = are_[t][q]; xim_fm[fm] = aim_[t][q];
            }
        }
}

// X[k][.] = sum_n s[n] exp(-2 pi j k n / N) for the bins [bin_lo, bin_hi) (numpy.fft.rfft semantics), direct DFT with an
// LDS twiddle table.  One workgroup per (frame, mic); thread t owns bin bin_lo + t.  Writes both operand layouts the
// GEMMs need: [K][M][F] (mic-major, frames contiguous) and [K][F][M] (frame-major, mics contiguous).
__global__ void __launch_bounds__(128) dft_kernel(const float* __restrict__ frames, const int32_t* __restrict__ mics, int m_total, int n_samples,
                                                  int n_frames, int n_mics, int bin_lo, int n_bins, float* __restrict__ xre_mf,
                                                  float* __restrict__ xim_mf, float* __restrict__ xre_fm, float* __restrict__ xim_fm)
{
    extern __shared__ float tw[];   // cos[N], sin[N], then the signal row [N]
    float* tc = tw;
    float* ts = tw + n_samples;
    float* sg = tw + 2 * n_samples;
    const int f = blockIdx.x / n_mics, m = blockIdx.x - f * n_mics;
    const float* row = frames + ((size_t)f * m_total + mics[m]) * n_samples;
    for (int n = threadIdx.x; n < n_samples; n += blockDim.x) {
        const double ang = -2.0 * 3.14159265358979323846 * (double)n / (double)n_samples;
        tc[n] = (float)cos(ang);
        ts[n] = (float)sin(ang);
        sg[n] = row[n];
    }
    __syncthreads();
    for (int t = threadIdx.x; t < n_bins; t += blockDim.x) {
        const int k = bin_lo + t;
        float re = 0.0f, im = 0.0f;
        int idx = 0;
        for (int n = 0; n < n_samples; ++n) {
            const float s = sg[n];
            re = fmaf(s, tc[idx], re);
            im = fmaf(s, ts[idx], im);
            idx += k;
            if (idx >= n_samples) idx -= n_samples * (idx / n_samples);
        }
        xre_mf[((size_t)t * n_mics + m) * n_frames + f] = re;
        xim_mf[((size_t)t * n_mics + m) * n_frames + f] = im;
        xre_fm[((size_t)t * n_frames + f) * n_mics + m] = re;
        xim_fm[((size_t)t * n_frames + f) * n_mics + m] = im;
    }
}

// a[k][m][d] = exp(-j 2 pi f_k tau[d][m]), evaluated in float64 and stored as float32 planes.
__global__ void __launch_bounds__(256) steering_kernel(const double* __restrict__ tau, const double* __restrict__ freq, int n_dirs, int n_mics,
                                                       int n_bins, float* __restrict__ are, float* __restrict__ aim)
{
    const size_t total = (size_t)n_bins * n_mics * n_dirs;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int d = (int)(i % n_dirs);
        const int m = (int)((i / n_dirs) % n_mics);
        const int k = (int)(i / ((size_t)n_dirs * n_mics));
        const double ph = -2.0 * 3.14159265358979323846 * freq[k] * tau[(size_t)d * n_mics + m];
        are[i] = (float)cos(ph);
        aim[i] = (float)sin(ph);
    }
}

// Per bin: R += delta * tr(R)/M * I;  R = L L^H (Cholesky, lower);  Linv = L^-1;  stores Linv TRANSPOSED planes
// [bin][col][row] (the A-operand layout of the MVDR GEMM).  One workgroup per bin, everything in LDS (M <= 128), rows
// padded to M + 1 floats so that column walks hit distinct banks.
//   * factorisation: right-looking, ONE barrier per column -- the column is left unscaled while the trailing block is
//     updated with a_ij conj(a_cj) / d_j, and all columns are scaled by 1 / sqrt(d_j) afterwards;
//   * inverse: column c of L^-1 by forward substitution, 4 lanes per column splitting the dot product, written into the
//     (free) upper triangle at the transposed position, which is exactly the output layout.
// Generalised for the blocked factorisation of larger matrices: the M x M input block sits in a matrix of leading
// dimension ld_in (one matrix of in_stride floats per bin), the transposed inverse goes to a block of leading dimension
// ld_out, and the diagonal loading is either relative to this block's trace (load_abs == nullptr) or an absolute per-bin
// value computed from the whole matrix.  `status_base` offsets the column index reported for a non-positive pivot.
__global__ void __launch_bounds__(256) cholesky_inverse_kernel(const float* __restrict__ rre, const float* __restrict__ rim, size_t in_stride, int ld_in, int M,
                                                               float loading, const float* __restrict__ load_abs, float* __restrict__ lire_t,
                                                               float* __restrict__ liim_t, size_t out_stride, int ld_out, int* __restrict__ status,
                                                               int status_base)
{
    extern __shared__ float sm[];
    const int LD = M + 1;
    float* Lr = sm;                 // M * LD
    float* Li = sm + M * LD;        // M * LD
    __shared__ float s_trace;
    __shared__ float rdiag[128];    // 1 / L[i][i]
    const int b = blockIdx.x, t = threadIdx.x, nt = blockDim.x;
    const float* Rr = rre + (size_t)b * in_stride;
    const float* Ri = rim + (size_t)b * in_stride;
    for (int i = t; i < M * M; i += nt) {
        const int r = i / M, c = i - r * M;
        Lr[r * LD + c] = Rr[(size_t)r * ld_in + c];
        Li[r * LD + c] = Ri[(size_t)r * ld_in + c];
    }
    __syncthreads();
    if (t == 0) {
        float tr = 0.0f;
        for (int i = 0; i < M; ++i) tr += Lr[i * LD + i];
        s_trace = load_abs ? load_abs[b] : loading * (tr / (float)M);
    }
    __syncthreads();
    if (t < M) { Lr[t * LD + t] += s_trace; Li[t * LD + t] = 0.0f; }

    const int ti = t >> 4, tc = t & 15;
    for (int j = 0; j < M; ++j) {
        __syncthreads();                       // column j and d_j are final
        const float d = Lr[j * LD + j];
        if (t == 0 && !(d > 0.0f)) status[b] = status_base + j + 1;
        const float invd = 1.0f / fmaxf(d, 1e-30f);
        // trailing update: A[i][c] -= a_ij conj(a_cj) / d_j,  j < c <= i
        for (int i = j + 1 + ti; i < M; i += 16) {
            const float sr = Lr[i * LD + j] * invd, si = Li[i * LD + j] * invd;
            for (int c = j + 1 + tc; c <= i; c += 16) {
                const float br = Lr[c * LD + j], bi = Li[c * LD + j];
                Lr[i * LD + c] -= sr * br + si * bi;
                Li[i * LD + c] -= si * br - sr * bi;
            }
        }
    }
    __syncthreads();
    if (t < M) rdiag[t] = 1.0f / sqrtf(fmaxf(Lr[t * LD + t], 1e-30f));     // 1 / L[t][t]
    __syncthreads();
    for (int e = t; e < M * M; e += nt) {       // L[i][j] = a_ij / sqrt(d_j), strictly lower part
        const int i = e / M, j = e - i * M;
        if (j < i) { Lr[i * LD + j] *= rdiag[j]; Li[i * LD + j] *= rdiag[j]; }
    }

    // x = column c of L^-1:  x[c] = 1 / L[c][c];  x[i] = -(sum_{p=c}^{i-1} L[i][p] x[p]) / L[i][i].   x[i] (i > c) lives at [c][i].
    const int q = t & 3;
    for (int i = 1; i < M; ++i) {
        __syncthreads();                        // the scaled L (first pass) / every x[p], p < i
        for (int c = t >> 2; c < i; c += nt >> 2) {
            float sr = 0.0f, si = 0.0f;
            for (int p = c + q; p < i; p += 4) {
                const float lr = Lr[i * LD + p], li = Li[i * LD + p];
                const float xr = (p == c) ? rdiag[c] : Lr[c * LD + p];
                const float xi = (p == c) ? 0.0f : Li[c * LD + p];
                sr += lr * xr - li * xi;
                si += lr * xi + li * xr;
            }
            sr += __shfl_xor(sr, 1, 64); si += __shfl_xor(si, 1, 64);
            sr += __shfl_xor(sr, 2, 64); si += __shfl_xor(si, 2, 64);
            if (q == 0) { Lr[c * LD + i] = -sr * rdiag[i]; Li[c * LD + i] = -si * rdiag[i]; }
        }
    }
    __syncthreads();
    float* Or = lire_t + (size_t)b * out_stride;
    float* Oi = liim_t + (size_t)b * out_stride;
    for (int e = t; e < M * M; e += nt) {       // transposed planes: [c][i] = Linv[i][c]
        const int c = e / M, i = e - c * M;
        Or[(size_t)c * ld_out + i] = i < c ? 0.0f : (i == c ? rdiag[c] : Lr[c * LD + i]);
        Oi[(size_t)c * ld_out + i] = i <= c ? 0.0f : Li[c * LD + i];
    }
}

// The same factorisation + inverse for M <= 64 with the matrix in REGISTERS (BASELINE config 3's 64-microphone array): cholesky_inverse_kernel
// walks the trailing block through LDS read-modify-write chains (137 us for 94 bins of 64 x 64, a fifth of the MVDR map).  Here thread
// (ti, tc) of a 16 x 16 grid owns the 4 x 4 elements (ti + 16 a, tc + 16 b) of A -- and later of L and of X = L^-1 -- as named registers; a step
// broadcasts one column (and, for the inverse, one row) through a double-buffered LDS vector, one barrier per step, and every thread applies the
// rank-1 update to its 16 elements:
//   factorisation, step j:  A[i][c] -= (a_ij / d_j) conj(a_cj)   (j < c <= i; the column stays unscaled, as in the LDS kernel: the same arithmetic);
//   L[i][j] = a_ij / sqrt(d_j);
//   inverse (forward elimination of [L | I]), step j:  X[j][:] *= 1 / L[j][j];  X[i][:] -= L[i][j] X[j][:]  (i > j).
// Elements past M behave as an identity block.  Output as cholesky_inverse_kernel: transposed planes out[c][i] = X[i][c], zeros above the diagonal.
__global__ void __launch_bounds__(256) cholesky_inverse_reg_kernel(const float* __restrict__ rre, const float* __restrict__ rim, size_t in_stride, int ld_in, int M,
                                                                   float loading, const float* __restrict__ load_abs, float* __restrict__ lire_t,
                                                                   float* __restrict__ liim_t, size_t out_stride, int ld_out, int* __restrict__ status,
                                                                   int status_base)
{
    __shared__ float colr[2][64], coli[2][64];          // the broadcast column (factorisation: raw a_ij; inverse: L[i][j])
    __shared__ float rowr[2][64], rowi[2][64];          // inverse: the scaled row X[j][:]
    __shared__ float s_red[4];
    __shared__ float rdiag[64];                         // 1 / L[j][j]
    const int b = blockIdx.x, t = threadIdx.x;
    const int ti = t & 15, tc = t >> 4;                 // rows fastest: 16 consecutive lanes hold 16 consecutive rows of a column
    const float* Rr = rre + (size_t)b * in_stride;
    const float* Ri = rim + (size_t)b * in_stride;
    float ar[4][4], ai[4][4];
    float tr = 0.0f;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int i = ti + 16 * a, j = tc + 16 * c;
            const bool in = i < M && j < M;
            ar[a][c] = in ? Rr[(size_t)i * ld_in + j] : (i == j ? 1.0f : 0.0f);
            ai[a][c] = in ? Ri[(size_t)i * ld_in + j] : 0.0f;
            if (in && i == j) tr += ar[a][c];
        }
    // diagonal loading: relative to the block's trace, or absolute per bin
    for (int off = 32; off > 0; off >>= 1) tr += __shfl_xor(tr, off, 64);
    if ((t & 63) == 0) s_red[t >> 6] = tr;
    __syncthreads();
    const float load = load_abs ? load_abs[b] : loading * ((s_red[0] + s_red[1] + s_red[2] + s_red[3]) / (float)M);
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (ti + 16 * a == tc + 16 * c && ti + 16 * a < M) { ar[a][c] += load; ai[a][c] = 0.0f; }

    // ---- factorisation.  Register arrays are indexed by constants only: the steps run as four instances of a 16-column block body whose block
    // number is a template constant (a select chain over the block gets folded back into a dynamically indexed array -- and into scratch).
    if (tc == 0) {
#pragma unroll
        for (int a = 0; a < 4; ++a) { colr[0][ti + 16 * a] = ar[a][0]; coli[0][ti + 16 * a] = ai[a][0]; }
    }
    __syncthreads();
    auto fact_block = [&](auto BLK) __attribute__((always_inline)) {
        constexpr int blk = decltype(BLK)::value;
        for (int jj = 0; jj < 16; ++jj) {
            const int j = 16 * blk + jj, cur = j & 1;
            if (j >= M) return;                             // (uniform)
            const float d = colr[cur][j];
            if (t == 0) { if (!(d > 0.0f)) status[b] = status_base + j + 1; rdiag[j] = 1.0f / sqrtf(fmaxf(d, 1e-30f)); }
            const float invd = 1.0f / fmaxf(d, 1e-30f);
            float sr[4], si[4], cr[4], ci[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) { sr[a] = colr[cur][ti + 16 * a] * invd; si[a] = coli[cur][ti + 16 * a] * invd; cr[a] = colr[cur][tc + 16 * a]; ci[a] = coli[cur][tc + 16 * a]; }
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int i = ti + 16 * a, cc = tc + 16 * c;
                    if (cc > j && cc <= i) {                // (i > j follows)
                        ar[a][c] -= sr[a] * cr[c] + si[a] * ci[c];
                        ai[a][c] -= si[a] * cr[c] - sr[a] * ci[c];
                    }
                }
            // the owners of column j + 1 publish it (updated through step j) into the other buffer
            const int jn = j + 1;
            if (jn < M && tc == (jn & 15)) {
                if (jj < 15) {
#pragma unroll
                    for (int a = 0; a < 4; ++a) { colr[cur ^ 1][ti + 16 * a] = ar[a][blk]; coli[cur ^ 1][ti + 16 * a] = ai[a][blk]; }
                } else if constexpr (blk < 3) {
#pragma unroll
                    for (int a = 0; a < 4; ++a) { colr[cur ^ 1][ti + 16 * a] = ar[a][blk + 1]; coli[cur ^ 1][ti + 16 * a] = ai[a][blk + 1]; }
                }
            }
            __syncthreads();
        }
    };
    fact_block(std::integral_constant<int, 0>{}); fact_block(std::integral_constant<int, 1>{});
    fact_block(std::integral_constant<int, 2>{}); fact_block(std::integral_constant<int, 3>{});
    // L[i][c] = a_ic / sqrt(d_c) below the diagonal (the diagonal lives on as rdiag), zero above; X starts as the identity
    float xr[4][4], xi[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int i = ti + 16 * a, cc = tc + 16 * c;
            const float rd = cc < M ? rdiag[cc] : 1.0f;
            ar[a][c] = i > cc ? ar[a][c] * rd : 0.0f;
            ai[a][c] = i > cc ? ai[a][c] * rd : 0.0f;
            xr[a][c] = i == cc ? 1.0f : 0.0f;
            xi[a][c] = 0.0f;
        }
    // ---- inverse: step j scales row j of X by 1 / L[j][j] and eliminates column j of L from the rows below
    auto publish = [&](int j, int buf, auto BLK) __attribute__((always_inline)) {      // column j of L (owners: tc == j % 16), scaled row j of X (owners: ti == j % 16)
        constexpr int blk = decltype(BLK)::value;
        if (tc == (j & 15)) {
#pragma unroll
            for (int a = 0; a < 4; ++a) { colr[buf][ti + 16 * a] = ar[a][blk]; coli[buf][ti + 16 * a] = ai[a][blk]; }
        }
        if (ti == (j & 15)) {
            const float rd = rdiag[j];
#pragma unroll
            for (int c = 0; c < 4; ++c) { rowr[buf][tc + 16 * c] = xr[blk][c] * rd; rowi[buf][tc + 16 * c] = xi[blk][c] * rd; }
        }
    };
    publish(0, 0, std::integral_constant<int, 0>{});
    __syncthreads();
    auto inv_block = [&](auto BLK) __attribute__((always_inline)) {
        constexpr int blk = decltype(BLK)::value;
        for (int jj = 0; jj < 16; ++jj) {
            const int j = 16 * blk + jj, cur = j & 1;
            if (j >= M) return;
            float lr[4], li[4], yr[4], yi[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) { lr[a] = colr[cur][ti + 16 * a]; li[a] = coli[cur][ti + 16 * a]; yr[a] = rowr[cur][tc + 16 * a]; yi[a] = rowi[cur][tc + 16 * a]; }
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int i = ti + 16 * a, cc = tc + 16 * c;
                    if (i == j) { xr[a][c] = yr[c]; xi[a][c] = yi[c]; }                 // the scaled row itself
                    else if (i > j && cc <= j) {
                        xr[a][c] -= lr[a] * yr[c] - li[a] * yi[c];
                        xi[a][c] -= lr[a] * yi[c] + li[a] * yr[c];
                    }
                }
            if (j + 1 < M) {
                if (jj < 15) publish(j + 1, cur ^ 1, std::integral_constant<int, blk>{});
                else if constexpr (blk < 3) publish(j + 1, cur ^ 1, std::integral_constant<int, blk + 1>{});
            }
            __syncthreads();
        }
    };
    inv_block(std::integral_constant<int, 0>{}); inv_block(std::integral_constant<int, 1>{});
    inv_block(std::integral_constant<int, 2>{}); inv_block(std::integral_constant<int, 3>{});
    float* Or = lire_t + (size_t)b * out_stride;
    float* Oi = liim_t + (size_t)b * out_stride;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int i = ti + 16 * a, cc = tc + 16 * c;
            if (i < M && cc < M) {
                Or[(size_t)cc * ld_out + i] = i < cc ? 0.0f : xr[a][c];
                Oi[(size_t)cc * ld_out + i] = i <= cc ? 0.0f : xi[a][c];
            }
        }
}

// ---- pieces of the blocked factorisation for 128 < M <= 256 (launch_fd_cholesky_inverse) -------------------------------
// load[b] = loading * tr(R_b) / M over the whole matrix
__global__ void __launch_bounds__(64) diag_load_kernel(const float* __restrict__ rre, int M, float loading, float* __restrict__ load)
{
    const float* R = rre + (size_t)blockIdx.x * M * M;
    float tr = 0.0f;
    for (int i = threadIdx.x; i < M; i += 64) tr += R[(size_t)i * M + i];
    for (int off = 32; off > 0; off >>= 1) tr += __shfl_xor(tr, off, 64);
    if (threadIdx.x == 0) load[blockIdx.x] = loading * (tr / (float)M);
}

// Small batched complex GEMM with arbitrary element strides (so that transposes and sub-blocks need no copies):
//   C(i, j) = alpha * sum_k opA(A(k, i)) opB(B(k, j)) + beta * Cin(i, j),   op = conj or identity.
// One wave per 32 x 32 tile, exact-f32 MFMA as everywhere in this file.  Sizes here are <= 128^3 per bin: not tuned.
struct StridedGemm {
    const float *a_re, *a_im; size_t a_batch; int a_sk, a_si;
    const float *b_re, *b_im; size_t b_batch; int b_sk, b_sj;
    const float *cin_re, *cin_im; size_t cin_batch; int cin_si, cin_sj;
    float *c_re, *c_im; size_t c_batch; int c_si, c_sj;
    int I, J, K, conj_a, conj_b;
    float alpha, beta;
};

__global__ void __launch_bounds__(64) cgemm_strided_kernel(StridedGemm g)
{
    const int lane = threadIdx.x, li = lane & 31, lk = lane >> 5;
    const int i0 = blockIdx.y * 32, j0 = blockIdx.x * 32, b = blockIdx.z;
    const bool ai_ok = i0 + li < g.I, bj_ok = j0 + li < g.J;
    const float* are = g.a_re + (size_t)b * g.a_batch + (size_t)(i0 + li) * g.a_si;
    const float* aim = g.a_im + (size_t)b * g.a_batch + (size_t)(i0 + li) * g.a_si;
    const float* bre = g.b_re + (size_t)b * g.b_batch + (size_t)(j0 + li) * g.b_sj;
    const float* bim = g.b_im + (size_t)b * g.b_batch + (size_t)(j0 + li) * g.b_sj;
    f32x16 cre, cim;
#pragma unroll
    for (int r = 0; r < 16; ++r) { cre[r] = 0.0f; cim[r] = 0.0f; }
    for (int k = 0; k < g.K; k += 2) {
        const int kk = k + lk;
        const bool k_ok = kk < g.K;
        const float xr = (ai_ok && k_ok) ? are[(size_t)kk * g.a_sk] : 0.0f;
        float xi = (ai_ok && k_ok) ? aim[(size_t)kk * g.a_sk] : 0.0f;
        const float yr = (bj_ok && k_ok) ? bre[(size_t)kk * g.b_sk] : 0.0f;
        float yi = (bj_ok && k_ok) ? bim[(size_t)kk * g.b_sk] : 0.0f;
        if (g.conj_a) xi = -xi;
        if (g.conj_b) yi = -yi;
        cre = __builtin_amdgcn_mfma_f32_32x32x2f32(xr, yr, cre, 0, 0, 0);
        cre = __builtin_amdgcn_mfma_f32_32x32x2f32(-xi, yi, cre, 0, 0, 0);
        cim = __builtin_amdgcn_mfma_f32_32x32x2f32(xr, yi, cim, 0, 0, 0);
        cim = __builtin_amdgcn_mfma_f32_32x32x2f32(xi, yr, cim, 0, 0, 0);
    }
    const int col = j0 + li;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = i0 + acc_row(r, lane);
        if (row < g.I && col < g.J) {
            float vr = g.alpha * cre[r], vi = g.alpha * cim[r];
            if (g.cin_re) {
                const size_t ci = (size_t)b * g.cin_batch + (size_t)row * g.cin_si + (size_t)col * g.cin_sj;
                vr += g.beta * g.cin_re[ci];
                vi += g.beta * g.cin_im[ci];
            }
            const size_t co = (size_t)b * g.c_batch + (size_t)row * g.c_si + (size_t)col * g.c_sj;
            g.c_re[co] = vr;
            g.c_im[co] = vi;
        }
    }
}

template <int EPI>
hipError_t run_gemm(const GemmArgs& g, hipStream_t stream)
{
    const dim3 grid((unsigned)((g.J + 127) / 128), (unsigned)(EPI == EPI_MVDR ? 1 : (g.I + 31) / 32), (unsigned)(EPI == EPI_STORE ? g.batch : 1));
    hipLaunchKernelGGL(cgemm_kernel<EPI>, grid, dim3(256), 0, stream, g);
    return hipGetLastError();
}

}  // namespace

hipError_t launch_fd_steering(const double* d_tau, const double* d_freq, int n_dirs, int n_mics, int n_bins, float* d_are, float* d_aim, hipStream_t stream)
{
    hipLaunchKernelGGL(steering_kernel, dim3(2048), dim3(256), 0, stream, d_tau, d_freq, n_dirs, n_mics, n_bins, d_are, d_aim);
    return hipGetLastError();
}

size_t fd_twiddle_floats(int n_samples, int n_bins) { return (size_t)2 * n_samples * ((n_bins + 31) / 32 * 32); }

hipError_t launch_fd_twiddles(int n_samples, int bin_lo, int n_bins, float* d_tw, hipStream_t stream)
{
    const int kp = (n_bins + 31) / 32 * 32;
    hipLaunchKernelGGL(twiddle_kernel, dim3(256), dim3(256), 0, stream, d_tw, d_tw + (size_t)n_samples * kp, n_samples, bin_lo, n_bins, kp);
    return hipGetLastError();
}

namespace {

// The same DFT with a finer split: a workgroup owns 32 rows (= 32 consecutive (frame, mic) pairs), its waves one 32-bin tile each.  A 64-sample chunk of the
// 32 signal rows is staged through LDS by coalesced loads (dft_mfma_kernel's lanes fetched their operand sample by sample from rows 1 KiB apart), three times as
// many workgroups fill the chip (config 3: 380 instead of 95), and only the [K][F][M] layout is written here -- rows are consecutive addresses of it --, the
// [K][M][F] one by transpose_planes_kernel (dft_mfma_kernel scattered it in 4-byte stores).
template <int KT>
__global__ void __launch_bounds__(256) dft_tile_kernel(const float* __restrict__ frames, const int32_t* __restrict__ mics, int m_total, int n_samples,
                                                       int n_frames, int n_mics, int n_bins, int kp_total, const float* __restrict__ wc,
                                                       const float* __restrict__ ws, float* __restrict__ xre_fm, float* __restrict__ xim_fm)
{
    constexpr int KP = 32 * KT, SP = 65;
    const int kofs = blockIdx.y * KP;
    __shared__ float sig[2][32 * SP];           // the signal rows' current 64-sample chunk, double-buffered
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lk = lane >> 5;
    const long long rows = (long long)n_frames * n_mics;
    const long long r0 = (long long)blockIdx.x * 32;
    // staging role: thread t fills samples [8 (t & 7), +8) of row t >> 3 of the chunk
    const long long sr = r0 + (tid >> 3);
    const bool s_ok = sr < rows;
    const int sf = s_ok ? (int)(sr / n_mics) : 0, sm = s_ok ? (int)(sr - (long long)sf * n_mics) : 0;
    const float* srow = frames + ((size_t)sf * m_total + (s_ok ? mics[sm] : 0)) * n_samples;
    float nx[8];                                 // the next chunk's samples on their way: requested before a chunk's MFMAs, stored to LDS after them
    auto fetch = [&](int n0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int n = n0 + (tid & 7) * 8 + j;
            nx[j] = (s_ok && n < n_samples) ? srow[n] : 0.0f;
        }
    };
    auto store = [&](int buf) {
#pragma unroll
        for (int j = 0; j < 8; ++j) sig[buf][(tid >> 3) * SP + (tid & 7) * 8 + j] = nx[j];
    };
    // this wave's twiddle operands come straight from the table (rows of kp_total bins, 32 consecutive bins per half wave: one 128-byte line; the table
    // stays in L2), no staging
    const int kb = kofs + 32 * wave + li;
    const bool k_ok = wave < KT && kb < kp_total;
    const float* wcp = wc + (k_ok ? kb : 0);
    const float* wsp = ws + (k_ok ? kb : 0);
    f32x16 are_, aim_;
#pragma unroll
    for (int q = 0; q < 16; ++q) { are_[q] = 0.0f; aim_[q] = 0.0f; }
    // Every workgroup walks the whole table; started together at sample 0 they would all ask one L2 channel for the same lines at the same time (measured:
    // 63 us for config 3, the table's traffic through one channel at a time).  Each workgroup therefore starts at its own chunk and its own place inside the
    // chunk -- a sum over the same terms in a rotated order.
    const int n_chunks = (n_samples + 63) / 64;
    const int c_first = (int)(blockIdx.x % (unsigned)n_chunks), s_first = (int)((blockIdx.x / (unsigned)n_chunks) & 3u) * 8;
    fetch(c_first * 64);
    store(0);
    int buf = 0;
    for (int ci = 0; ci < n_chunks; ++ci, buf ^= 1) {
        const int cc = ci + c_first, n0 = (cc >= n_chunks ? cc - n_chunks : cc) * 64;
        __syncthreads();                                          // chunk n0 is in sig[buf]; sig[buf ^ 1]'s readers are done
        const bool more = ci + 1 < n_chunks;
        if (more) fetch((cc + 1 >= n_chunks ? cc + 1 - n_chunks : cc + 1) * 64);
        if (wave < KT) {
#pragma unroll 8
            for (int si = 0; si < 32; ++si) {
                const int s2 = (si + s_first) & 31;
                // (unconditional loads: a sample past the last has a zero in `sig`, a bin past the table's last is never stored -- a conditional load
                //  is a branch around it, and 64 of those per chunk serialise the loop)
                const int n = min(n0 + 2 * s2 + lk, n_samples - 1);
                const float b = sig[buf][li * SP + 2 * s2 + lk];
                const float c = wcp[(size_t)n * kp_total];
                const float sn = wsp[(size_t)n * kp_total];
                are_ = __builtin_amdgcn_mfma_f32_32x32x2f32(c, b, are_, 0, 0, 0);
                aim_ = __builtin_amdgcn_mfma_f32_32x32x2f32(sn, b, aim_, 0, 0, 0);
            }
        }
        if (more) store(buf ^ 1);
    }
    const long long r = r0 + li;
    if (wave >= KT || r >= rows) return;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int k = kofs + 32 * wave + acc_row(q, lane);
        if (k < n_bins) {
            const size_t fm = (size_t)k * (size_t)rows + (size_t)r;                  // ((k F + f) M + m)
            xre_fm[fm] = are_[q];
            xim_fm[fm] = aim_[q];
        }
    }
}

// [K][F][M] -> [K][M][F], both planes: 32 x 32 tiles through LDS
__global__ void __launch_bounds__(256) transpose_planes_kernel(const float* __restrict__ in_re, const float* __restrict__ in_im, float* __restrict__ out_re,
                                                               float* __restrict__ out_im, int F, int M)
{
    __shared__ float tr[32][33], ti[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int k = blockIdx.z, m0 = blockIdx.x * 32, f0 = blockIdx.y * 32;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int f = f0 + ty + 8 * j, m = m0 + tx;
        if (f < F && m < M) {
            const size_t at = ((size_t)k * F + f) * M + m;
            tr[ty + 8 * j][tx] = in_re[at];
            ti[ty + 8 * j][tx] = in_im[at];
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int m = m0 + ty + 8 * j, f = f0 + tx;
        if (m < M && f < F) {
            const size_t at = ((size_t)k * M + m) * F + f;
            out_re[at] = tr[tx][ty + 8 * j];
            out_im[at] = ti[tx][ty + 8 * j];
        }
    }
}

}  // namespace

hipError_t launch_fd_dft(const float* d_frames, const int32_t* d_mics, int m_total, int n_samples, int n_frames, int n_mics, int bin_lo, int n_bins,
                         const float* d_tw, float* xre_mf, float* xim_mf, float* xre_fm, float* xim_fm, hipStream_t stream)
{
    const int kt_all = (n_bins + 31) / 32;
    if (d_tw == nullptr) {
        // no twiddle table: the plain kernel, one workgroup per (frame, mic)
        hipLaunchKernelGGL(dft_kernel, dim3((unsigned)(n_frames * n_mics)), dim3(128), (size_t)3 * n_samples * sizeof(float), stream, d_frames, d_mics, m_total,
                           n_samples, n_frames, n_mics, bin_lo, n_bins, xre_mf, xim_mf, xre_fm, xim_fm);
        return hipGetLastError();
    }
    const int kp = kt_all * 32;                      // row length of the twiddle table (launch_fd_twiddles)
    const int kt = kt_all > 4 ? 4 : kt_all;          // tiles per wave; more than 128 bins: several bin groups (blockIdx.y)
    const int groups = (kt_all + kt - 1) / kt;
    const float* wc = d_tw;
    const float* ws = d_tw + (size_t)n_samples * kp;
    const long long rows = (long long)n_frames * n_mics;
    static const int tile_env = [] { const char* e = getenv("BF_FD_DFT_TILE"); return e ? atoi(e) : 1; }();      // 0: dft_mfma_kernel (A/B)
    if (tile_env != 0) {
        const dim3 grid32((unsigned)((rows + 31) / 32), (unsigned)groups);
        const size_t lds32 = 0;
        auto go32 = [&](auto kernel) -> hipError_t {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds32);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL(kernel, grid32, dim3(256), lds32, stream, d_frames, d_mics, m_total, n_samples, n_frames, n_mics, n_bins, kp, wc, ws, xre_fm, xim_fm);
            e = hipGetLastError();
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL(transpose_planes_kernel, dim3((unsigned)((n_mics + 31) / 32), (unsigned)((n_frames + 31) / 32), (unsigned)n_bins), dim3(256), 0, stream,
                               xre_fm, xim_fm, xre_mf, xim_mf, n_frames, n_mics);
            return hipGetLastError();
        };
        switch (kt) {
            case 1: return go32(dft_tile_kernel<1>);
            case 2: return go32(dft_tile_kernel<2>);
            case 3: return go32(dft_tile_kernel<3>);
            default: return go32(dft_tile_kernel<4>);
        }
    }
    const dim3 grid((unsigned)((rows + 127) / 128), (unsigned)groups);
    const size_t lds = (size_t)2 * 64 * (32 * kt) * sizeof(float);
    auto go = [&](auto kernel) -> hipError_t {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kernel, grid, dim3(256), lds, stream, d_frames, d_mics, m_total, n_samples, n_frames, n_mics, n_bins, kp, wc, ws, xre_mf, xim_mf,
                           xre_fm, xim_fm);
        return hipGetLastError();
    };
    switch (kt) {
        case 1: return go(dft_mfma_kernel<1>);
        case 2: return go(dft_mfma_kernel<2>);
        case 3: return go(dft_mfma_kernel<3>);
        default: return go(dft_mfma_kernel<4>);
    }
}

namespace {

// Bin groups so that the launch fills the chip a few times over (two 4-wave workgroups per CU) while every wave keeps >= 2 bins.
int bin_groups(int col_tiles, int row_groups, int n_bins)
{
    const int per_wave_cap = (n_bins + 7) / 8;                    // at least 2 bins per wave, 8 per workgroup
    int want = (1536 + col_tiles * row_groups - 1) / (col_tiles * row_groups);
    if (want > per_wave_cap) want = per_wave_cap;
    return want < 1 ? 1 : want;
}

template <int EPI, int RT, bool SPLIT>
hipError_t run_bins_t(const GemmArgs& g0, int row_groups, float* d_work, size_t work_floats, size_t plane, hipStream_t stream)
{
    GemmArgs g = g0;
    const int col_tiles = (g.J + 31) / 32;
    int groups = bin_groups(col_tiles, row_groups, g.batch);
    if (groups > 1 && (d_work == nullptr || work_floats < (size_t)groups * plane)) groups = 1;   // no workspace: one group, direct output
    const int per_group = (g.batch + groups - 1) / groups;
    groups = (g.batch + per_group - 1) / per_group;
    float* final_out = g.out0;
    if (groups > 1) g.out0 = d_work;
    const size_t lds = (EPI == EPI_POWER) ? (size_t)4 * RT * 16 * 64 * sizeof(float) : (size_t)4 * 32 * sizeof(float);
    auto kernel = cgemm_bins_kernel<EPI, RT, SPLIT>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kernel, dim3((unsigned)(col_tiles * row_groups), (unsigned)groups), dim3(256), lds, stream, g, row_groups, per_group);
    e = hipGetLastError();
    if (e != hipSuccess || groups == 1) return e;
    hipLaunchKernelGGL(reduce_planes_kernel, dim3(1024), dim3(256), 0, stream, d_work, groups, plane, final_out);
    return hipGetLastError();
}

template <int EPI, int RT>
hipError_t run_bins(const GemmArgs& g, int row_groups, float* d_work, size_t work_floats, size_t plane, hipStream_t stream)
{
    return gemm_f32_mode(-1) == 1 ? run_bins_t<EPI, RT, true>(g, row_groups, d_work, work_floats, plane, stream)
                                  : run_bins_t<EPI, RT, false>(g, row_groups, d_work, work_floats, plane, stream);
}

}  // namespace

size_t fd_workspace_floats(int n_rows, int n_dirs, int n_bins)
{
    // the most bin groups run_bins can choose, times one output plane (n_rows = frames for the DAS power, 1 for MVDR)
    return (size_t)((n_bins + 7) / 8) * (size_t)n_rows * (size_t)n_dirs;
}

hipError_t launch_fd_das_power(const float* xre_mf, const float* xim_mf, const float* are, const float* aim, int n_frames, int n_mics, int n_dirs,
                               int n_bins, float* d_power, float* d_work, size_t work_floats, hipStream_t stream)
{
    GemmArgs g{xre_mf, xim_mf, are, aim, d_power, nullptr, n_frames, n_dirs, n_mics, n_bins, 0, 1.0f};
    // row tiles (32 frames each) per wave: at most 4 (its |C|^2 accumulators live in registers), balanced over the row groups
    // ($BF_FD_RT caps it for A/B runs; split mode, config 3: 1 / 2 / 3 / 4 -> 1.15 / 1.05 / 1.04 / 1.05 ms -- the few spilled registers of the 3- and 4-tile
    //  instantiations cost less than a bin's B panel split once more)
    static const int rt_env = [] { const char* e = getenv("BF_FD_RT"); return e ? atoi(e) : 0; }();
    const int rt_max = rt_env >= 1 && rt_env <= 4 ? rt_env : 4;
    const int tiles = (n_frames + 31) / 32;
    const int row_groups = (tiles + rt_max - 1) / rt_max;
    const int rt = (tiles + row_groups - 1) / row_groups;
    const size_t plane = (size_t)n_frames * n_dirs;
    switch (rt) {
        case 1: return run_bins<EPI_POWER, 1>(g, row_groups, d_work, work_floats, plane, stream);
        case 2: return run_bins<EPI_POWER, 2>(g, row_groups, d_work, work_floats, plane, stream);
        case 3: return run_bins<EPI_POWER, 3>(g, row_groups, d_work, work_floats, plane, stream);
        default: return run_bins<EPI_POWER, 4>(g, row_groups, d_work, work_floats, plane, stream);
    }
}

hipError_t launch_fd_covariance(const float* xre_fm, const float* xim_fm, int n_frames, int n_mics, int n_bins, float* rre, float* rim, hipStream_t stream)
{
    // R[i][j] = (1/F) sum_f x[f][i] conj(x[f][j]):  A = X ([K=frames][I=mics]), B = conj(X) ([K=frames][J=mics])
    GemmArgs g{xre_fm, xim_fm, xre_fm, xim_fm, rre, rim, n_mics, n_mics, n_frames, n_bins, 1, 1.0f / (float)n_frames};
    return run_gemm<EPI_STORE>(g, stream);
}

namespace {
hipError_t run_strided(const StridedGemm& g, int batch, hipStream_t stream)
{
    hipLaunchKernelGGL(cgemm_strided_kernel, dim3((unsigned)((g.J + 31) / 32), (unsigned)((g.I + 31) / 32), (unsigned)batch), dim3(64), 0, stream, g);
    return hipGetLastError();
}

hipError_t run_cholesky(const float* rre, const float* rim, size_t in_stride, int ld_in, int m, float loading, const float* load_abs, float* ore, float* oim,
                        size_t out_stride, int ld_out, int* d_status, int status_base, int n_bins, hipStream_t stream)
{
    if (m <= 64) {                                         // the register-resident form
        hipLaunchKernelGGL(cholesky_inverse_reg_kernel, dim3((unsigned)n_bins), dim3(256), 0, stream, rre, rim, in_stride, ld_in, m, loading, load_abs, ore, oim,
                           out_stride, ld_out, d_status, status_base);
        return hipGetLastError();
    }
    const size_t lds = (size_t)2 * m * (m + 1) * sizeof(float);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(cholesky_inverse_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(cholesky_inverse_kernel, dim3((unsigned)n_bins), dim3(256), lds, stream, rre, rim, in_stride, ld_in, m, loading, load_abs, ore, oim,
                       out_stride, ld_out, d_status, status_base);
    return hipGetLastError();
}
}  // namespace

size_t fd_cholesky_workspace_floats(int n_mics, int n_bins)
{
    if (n_mics <= 128) return 0;
    const size_t h = 128, m2 = (size_t)n_mics - h;
    // per bin: the absolute loading, then (re, im) of L21 [m2 x h], S [m2 x m2], T [m2 x h]
    return (size_t)n_bins * (1 + 2 * (m2 * h + m2 * m2 + m2 * h));
}

hipError_t launch_fd_cholesky_inverse(const float* rre, const float* rim, int n_mics, int n_bins, float loading, float* lire_t, float* liim_t,
                                      int* d_status, float* d_work, size_t work_floats, hipStream_t stream)
{
    const int M = n_mics;
    const size_t mm = (size_t)M * M;
    if (M <= 128) return run_cholesky(rre, rim, mm, M, M, loading, nullptr, lire_t, liim_t, mm, M, d_status, 0, n_bins, stream);
    if (M > 256 || d_work == nullptr || work_floats < fd_cholesky_workspace_floats(M, n_bins)) return hipErrorInvalidValue;

    // Two-by-two blocks, h = 128:   R = [R11 R21^H; R21 R22],  L = [L11 0; L21 L22],  L^-1 = [X11 0; X21 X22]
    //   X11 = L11^-1 (in-LDS kernel on R11);  L21 = R21 X11^H;  S = R22 - L21 L21^H;  X22 = chol-inverse(S);
    //   X21 = -X22 (L21 X11).            Output planes hold the TRANSPOSE: out[c][r] = L^-1[r][c].
    const int h = 128, m2 = M - h;
    float* load = d_work;
    float* l21_re = load + n_bins;                       float* l21_im = l21_re + (size_t)n_bins * m2 * h;
    float* s_re = l21_im + (size_t)n_bins * m2 * h;      float* s_im = s_re + (size_t)n_bins * m2 * m2;
    float* t_re = s_im + (size_t)n_bins * m2 * m2;       float* t_im = t_re + (size_t)n_bins * m2 * h;
    hipError_t e = hipMemsetAsync(lire_t, 0, (size_t)n_bins * mm * sizeof(float), stream);          // the upper-right block of L^-1 is zero
    if (e == hipSuccess) e = hipMemsetAsync(liim_t, 0, (size_t)n_bins * mm * sizeof(float), stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(diag_load_kernel, dim3((unsigned)n_bins), dim3(64), 0, stream, rre, M, loading, load);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    // X11^T -> out[c][r], c, r < h
    if ((e = run_cholesky(rre, rim, mm, M, h, 0.0f, load, lire_t, liim_t, mm, M, d_status, 0, n_bins, stream)) != hipSuccess) return e;
    // L21(i, j) = sum_k R21(i, k) conj(X11(j, k)):  A(k, i) = R[(h + i) M + k],  B(k, j) = X11(j, k) = out[k M + j]
    StridedGemm g1{rre + (size_t)h * M, rim + (size_t)h * M, mm, 1, M,   lire_t, liim_t, mm, M, 1,   nullptr, nullptr, 0, 0, 0,
                   l21_re, l21_im, (size_t)m2 * h, h, 1,   m2, h, h, 0, 1, 1.0f, 0.0f};
    if ((e = run_strided(g1, n_bins, stream)) != hipSuccess) return e;
    // S(i, j) = R22(i, j) - sum_k L21(i, k) conj(L21(j, k))
    StridedGemm g2{l21_re, l21_im, (size_t)m2 * h, 1, h,   l21_re, l21_im, (size_t)m2 * h, 1, h,   rre + (size_t)h * M + h, rim + (size_t)h * M + h, mm, M, 1,
                   s_re, s_im, (size_t)m2 * m2, m2, 1,   m2, m2, h, 0, 1, -1.0f, 1.0f};
    if ((e = run_strided(g2, n_bins, stream)) != hipSuccess) return e;
    // X22^T -> out[h + c][h + r]
    if ((e = run_cholesky(s_re, s_im, (size_t)m2 * m2, m2, m2, 0.0f, load, lire_t + (size_t)h * M + h, liim_t + (size_t)h * M + h, mm, M, d_status, h, n_bins,
                          stream)) != hipSuccess) return e;
    // T(i, j) = sum_k L21(i, k) X11(k, j):  B(k, j) = X11(k, j) = out[j M + k]
    StridedGemm g3{l21_re, l21_im, (size_t)m2 * h, 1, h,   lire_t, liim_t, mm, 1, M,   nullptr, nullptr, 0, 0, 0,
                   t_re, t_im, (size_t)m2 * h, h, 1,   m2, h, h, 0, 0, 1.0f, 0.0f};
    if ((e = run_strided(g3, n_bins, stream)) != hipSuccess) return e;
    // X21(i, j) = -sum_k X22(i, k) T(k, j), stored transposed at out[j M + h + i]:  A(k, i) = X22(i, k) = out[(h + k) M + h + i]
    StridedGemm g4{lire_t + (size_t)h * M + h, liim_t + (size_t)h * M + h, mm, M, 1,   t_re, t_im, (size_t)m2 * h, h, 1,   nullptr, nullptr, 0, 0, 0,
                   lire_t + h, liim_t + h, mm, 1, M,   m2, h, m2, 0, 0, -1.0f, 0.0f};
    return run_strided(g4, n_bins, stream);
}

hipError_t launch_fd_mvdr_power(const float* lire_t, const float* liim_t, const float* are, const float* aim, int n_mics, int n_dirs, int n_bins,
                                float* d_power, float* d_work, size_t work_floats, hipStream_t stream)
{
    // y = Linv a:  A = Linv ([K=col][I=row], i.e. the transposed planes), B = a ([K=mic][J=dir]);  P[d] = sum_k 1 / ||y||^2
    // the steering vector in the w^H x sense is v = conj(a) (the delay-and-sum output is sum_m a_m x_m = v^H x)
    GemmArgs g{lire_t, liim_t, are, aim, d_power, nullptr, n_mics, n_dirs, n_mics, n_bins, 1, 1.0f};
    return run_bins<EPI_MVDR, 1>(g, 1, d_work, work_floats, (size_t)n_dirs, stream);
}

}  // namespace bf
