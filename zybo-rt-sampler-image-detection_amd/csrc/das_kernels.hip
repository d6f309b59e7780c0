// das_kernels.hip -- time-domain delay-and-sum beamformers for gfx950 (MI355X, CDNA4, wave64).
//
// What one launch computes (reference: PC/src/algorithms/{pad,lerp,convolve,hybrid_convolve}_and_sum.c):
//   for every frame f and steering direction d:   out_d[k] = sum_m delay_{d,m}( signals[f][mic_m][.] )[k]
//                                                  image[f][d] = (1/N) * sum_k (out_d[k] / M)^2
// The reference walks directions, then mics, then samples on one CPU thread.  Here:
//   * one WORKGROUP owns a tile of directions of one frame and passes that frame's microphone rows through LDS
//     (zero-extended so that a delayed read never needs a bounds test);
//   * one WAVE carries a few directions' out_d in registers across the mics; the per-(direction, mic) delay /
//     interpolation weight / FIR taps are wave-uniform;
//   * mic order and operation order are the reference's, and the final sum over k is sequential in k (rows parked in
//     LDS, one lane per direction), so the images are bit-identical to the CPU result.
// Two families of kernels (DESIGN.md section 4.1 has the measurements behind each choice):
//   * das_mimo_kernel / das_miso_kernel ("strided": lane l owns samples l, l+64, ...): any N <= 1024, any tap count,
//     single beams; table entries by vector load + v_readlane.  The first implementation; now the general fallback.
//   * copies::das_copies_kernel ("shifted copies": every staged row kept in copies shifted by one sample each, lane l owns
//     the quad 4l..4l+3): pad / lerp for 128 < N <= 1024 and the 8-tap FIR flavours for N <= 256.  pad / lerp sweep a mic
//     over the wave's 8 directions and re-read its quad from LDS only when the delay changes from one direction to the
//     next (two copies, 8-byte reads); the FIR flavours and the direction-outer variant for tables without structure read
//     at every step (four copies, 16-byte reads).
//   * copies::das_pair_kernel: the same sweep with TWO frames per workgroup -- batched pad / lerp launches at N <= 256,
//     what the bench and every multi-frame caller run: the per-step scalar work is shared by both frames.
// Workgroup id -> (tile, frame): with large tables the tile count is padded to a multiple of 8 so that tile % 8 == id % 8,
// i.e. all frames' workgroups of one direction tile land on one XCD and re-read that tile's table slice from the XCD's
// own L2; tables that fit every L2 spread their tiles over all XCDs (plan_das sizes the tiles by the rounds either costs).
//
// Roofline: gather-accumulate, no MFMA, not HBM-bound (tables and samples are reused out of L2 / LDS); the binding
// resource is VALU issue.  See DESIGN.md section 5 for the byte / instruction accounting.
#include "das_kernels.h"

#include <algorithm>
#include <type_traits>

namespace bf {

namespace {

constexpr int kWave = 64;

// Scalars of one launch (kernel argument, lives in SGPRs).
struct KArgs {
    long long miso_row;                   // launch_miso only: flat table offset (the reference's `offset`)
    int n_mics, m_total, n_samples, n_taps;
    int dir_begin, dir_end, image_stride, image_origin;
    int lead, row_stride, mic_chunk, n_chunks, tile_dirs, n_tiles;
    int scratch_off, srow, pbw;           // per-wave power scratch: float offset in LDS, row stride, rows per wave
    int n_is_pow2;
    float inv_n;
    int debug;   // profiling only (BF_DEBUG): bit 0 = skip the ordered power sum (wrong images)
    int n_frames;   // frames of the launch (das_pair_kernel: whether a workgroup's second frame exists)
    int wg_frames, frame_inner;   // workgroup id -> (tile, frame [pair]): see tile_and_frame()
    long long digest_h_off;   // shifted-copies pad / lerp: where the grouped lerp weights start in the digest buffer (floats)
    long long digest_t_off;   // the 8-tap FIR pair kernel: where the taps regrouped per 8 directions start in the digest buffer (floats)
};

// Workgroup id -> (direction tile, frame or frame pair).  Ids go round-robin over the 8 XCDs.
//   frame_inner == 0:  tile = id % n_tiles, frame = id / n_tiles.  With n_tiles a multiple of 8 a tile's workgroups stay on
//                      one XCD (tile % 8 == id % 8); an XCD walks its tiles frame by frame, so a tile's table slice is
//                      re-used out of L2 only if the XCD's share of the whole table stays resident (cfg2: 650 KB).
//   frame_inner == 1:  tables beyond that (cfg5: 33 MB per XCD): XCD x = id % 8 walks tile x, x + 8, .. and runs ALL frames of
//                      a tile back to back (its 32 CUs hold 32 frames of the same tile at a time), so the slice comes from
//                      HBM once instead of once per frame.
__device__ __forceinline__ void tile_and_frame(const KArgs& a, int* tile, int* frame)
{
    const unsigned id = blockIdx.x;
    if (a.frame_inner) {
        const unsigned x = id & 7u, j = id >> 3;
        *tile = (int)(x + 8u * (j / (unsigned)a.wg_frames));
        *frame = (int)(j % (unsigned)a.wg_frames);
    } else {
        *tile = (int)(id % (unsigned)a.n_tiles);
        *frame = (int)(id / (unsigned)a.n_tiles);
    }
}

// The read-only tables are separate `const __restrict__` kernel parameters on purpose: only then can the
// compiler prove that no store in the kernel clobbers them and fetch the wave-uniform table entries with
// scalar loads (s_load_*) instead of 64-lane vector loads.
#define BF_TABLE_PARAMS                                                                                   \
    const float* __restrict__ signals, float* __restrict__ images, const int32_t* __restrict__ mics,     \
        const int32_t* __restrict__ whole, const float* __restrict__ frac, const float* __restrict__ taps
#define BF_TABLE_ARGS signals, images, mics, whole, frac, taps

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, kWave);
    return v;
}

// Copy mic rows [m0, m0+mc) of one frame into LDS rows 0..mc-1 at column `lead`.  One wave per row, lanes
// stride the row in 16-byte pieces (coalesced global_load_dwordx4 -> ds_write_b128).
__device__ __forceinline__ void stage_chunk(float* lds, const KArgs& a, const int32_t* __restrict__ mics,
                                            const float* __restrict__ frame, int m0, int mc, int wave, int nwaves, int lane)
{
    const int n = a.n_samples;
    for (int r = wave; r < mc; r += nwaves) {
        const int mic = mics[m0 + r];
        const float* src = frame + (size_t)mic * n;
        float* dst = lds + r * a.row_stride + a.lead;
        if ((n & 3) == 0) {
            const float4* s4 = reinterpret_cast<const float4*>(src);
            float4* d4 = reinterpret_cast<float4*>(dst);
            for (int i = lane; i < (n >> 2); i += kWave) d4[i] = s4[i];
        } else {
            for (int i = lane; i < n; i += kWave) dst[i] = src[i];
        }
    }
}

// First 64-entry block of a table row, requested one work item ahead (pad / lerp): lane m holds entry m.
struct RowHead {
    int p = 0;
    float h = 0.0f;
    bool valid = false;   // wave-uniform
};

template <int ALGO>
__device__ __forceinline__ RowHead request_row_head(const int32_t* __restrict__ whole, const float* __restrict__ frac, size_t row, int mc, int lane)
{
    RowHead r;
    if constexpr (ALGO == ALGO_PAD || ALGO == ALGO_LERP) {
        r.p = (lane < mc) ? whole[row + lane] : 0;
        if constexpr (ALGO == ALGO_LERP) r.h = (lane < mc) ? frac[row + lane] : 0.0f;
        r.valid = true;
    }
    return r;
}

// Accumulate mics [m0, m0+mc) of the table row starting at flat entry `row_base` (= d*M for direction d)
// into acc[NC] (lane l holds samples l + 64 c).
template <int ALGO, int NC>
__device__ __forceinline__ void accumulate(float (&acc)[NC], const float* lds, const KArgs& a, const int32_t* __restrict__ whole,
                                           const float* __restrict__ frac, const float* __restrict__ taps, size_t row_base, int m0,
                                           int mc, int lane, const RowHead head = RowHead())
{
    const size_t row = row_base + m0;
    const int rs = a.row_stride;

    // pad / lerp: the table row of a direction is fetched 64 mics at a time with ONE coalesced vector load (lane m
    // holds entry m; the next block is requested before the current one is consumed) and the wave-uniform entry of
    // each mic is then read out of that register with v_readlane.  Scalar loads would need no VALU slot, but
    // every new row misses the scalar cache and s_load shares the LDS wait counter, so their latency sat fully
    // exposed in front of every group of LDS reads (measured: 14 instead of 9 cycles per (direction, mic) per CU).
    if constexpr (ALGO == ALGO_PAD || ALGO == ALGO_LERP) {
        const int32_t* __restrict__ wrow = whole + row;
        const float* __restrict__ hrow = frac + row;
        int vp = head.p;
        float vh = head.h;
        if (!head.valid) {
            vp = (lane < mc) ? wrow[lane] : 0;
            if constexpr (ALGO == ALGO_LERP) vh = (lane < mc) ? hrow[lane] : 0.0f;
        }
        for (int b0 = 0; b0 < mc; b0 += kWave) {
            const int bn = min(kWave, mc - b0);
            int vp_next = 0;
            float vh_next = 0.0f;
            if (b0 + kWave < mc) {   // wave-uniform
                vp_next = (b0 + kWave + lane < mc) ? wrow[b0 + kWave + lane] : 0;
                if constexpr (ALGO == ALGO_LERP) vh_next = (b0 + kWave + lane < mc) ? hrow[b0 + kWave + lane] : 0.0f;
            }
            // (v_readlane is a convergent operation: the compiler will not unroll a runtime-trip loop around it, so the
            //  blocks of 8 / 4 mics are spelled out and a scalar remainder loop follows)
            auto pad_one = [&](int u) {
                // pad_and_sum.c:41-47,54-70   out[p + i] += s[i]
                const int p = __builtin_amdgcn_readlane(vp, u);
                const float* r = lds + (b0 + u) * rs + (a.lead - p) + lane;
#pragma unroll
                for (int c = 0; c < NC; ++c) acc[c] += r[c * kWave];
            };
            auto lerp_one = [&](int u) {
                // lerp_and_sum.c:50-56,67-92  out[p + i + 1] += s[i] + h * (s[i+1] - s[i]),  0 <= i < N - p - 1
                const int p = __builtin_amdgcn_readlane(vp, u);
                const float h = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(vh), u));
                const float* r = lds + (b0 + u) * rs + (a.lead - p - 1) + lane;
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    const float s0 = r[c * kWave];
                    const float s1 = r[c * kWave + 1];
                    float v = __fmaf_rn(h, s1 - s0, s0);      // gcc contracts s0 + h*(s1-s0) into one fma
                    if (c * kWave <= p) v = (lane + c * kWave > p) ? v : 0.0f;   // i >= 0 only (wave-uniform guard)
                    acc[c] += v;
                }
            };
            int u = 0;
            if constexpr (ALGO == ALGO_PAD) {
                constexpr int kU = NC <= 4 ? 8 : 4;
                for (; u + kU <= bn; u += kU) {
#pragma unroll
                    for (int i = 0; i < kU; ++i) pad_one(u + i);
                }
                for (; u < bn; ++u) pad_one(u);
            } else {
                constexpr int kU = NC <= 4 ? 4 : 2;
                for (; u + kU <= bn; u += kU) {
#pragma unroll
                    for (int i = 0; i < kU; ++i) lerp_one(u + i);
                }
                for (; u < bn; ++u) lerp_one(u);
            }
            vp = vp_next;
            vh = vh_next;
        }
    } else if constexpr (ALGO == ALGO_HYBRID) {
        // hybrid_convolve_and_sum.c:51-64  out[p + i + 1] += h[t] * padded[i + t], t = 0..T-1 in order
        const int T = a.n_taps;
        const int32_t* __restrict__ wrow = whole + row;
        const float* __restrict__ trow = taps + row * T;
        for (int ms = 0; ms < mc; ++ms) {
            const int p = wrow[ms];
            const float* __restrict__ h = trow + ms * T;
            const float* r = lds + ms * rs + (a.lead - p - 1 - T / 2) + lane;
            if (T == 8) {   // the reference's N_TAPS: taps in scalar registers, tap loop unrolled
                const float h0 = h[0], h1 = h[1], h2 = h[2], h3 = h[3], h4 = h[4], h5 = h[5], h6 = h[6], h7 = h[7];
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    const float* x = r + c * kWave;
                    float o = acc[c];
                    o = __fmaf_rn(h0, x[0], o); o = __fmaf_rn(h1, x[1], o); o = __fmaf_rn(h2, x[2], o); o = __fmaf_rn(h3, x[3], o);
                    o = __fmaf_rn(h4, x[4], o); o = __fmaf_rn(h5, x[5], o); o = __fmaf_rn(h6, x[6], o); o = __fmaf_rn(h7, x[7], o);
                    acc[c] = (c * kWave <= p && !(lane + c * kWave > p)) ? acc[c] : o;   // samples with i < 0 receive nothing
                }
                continue;
            }
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                float o = acc[c];
                if (c * kWave <= p) {
                    // this segment contains samples with i < 0: they must not receive anything
                    const bool live = lane + c * kWave > p;
                    for (int t = 0; t < T; ++t) o = live ? __fmaf_rn(h[t], r[c * kWave + t], o) : o;
                } else {
                    for (int t = 0; t < T; ++t) o = __fmaf_rn(h[t], r[c * kWave + t], o);
                }
                acc[c] = o;
            }
        }
    } else if constexpr (ALGO == ALGO_FIR_NAIVE) {
        // convolve_and_sum.c:197-211  out[i] += h[t] * padded[i + t], t in order (fma chain into out)
        const int T = a.n_taps;
        const float* __restrict__ trow = taps + row * T;
        for (int ms = 0; ms < mc; ++ms) {
            const float* __restrict__ h = trow + ms * T;
            const float* r = lds + ms * rs + (a.lead - T / 2) + lane;
            if (T == 8) {
                const float h0 = h[0], h1 = h[1], h2 = h[2], h3 = h[3], h4 = h[4], h5 = h[5], h6 = h[6], h7 = h[7];
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    const float* x = r + c * kWave;
                    float o = acc[c];
                    o = __fmaf_rn(h0, x[0], o); o = __fmaf_rn(h1, x[1], o); o = __fmaf_rn(h2, x[2], o); o = __fmaf_rn(h3, x[3], o);
                    o = __fmaf_rn(h4, x[4], o); o = __fmaf_rn(h5, x[5], o); o = __fmaf_rn(h6, x[6], o); o = __fmaf_rn(h7, x[7], o);
                    acc[c] = o;
                }
                continue;
            }
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                float o = acc[c];
                for (int t = 0; t < T; ++t) o = __fmaf_rn(h[t], r[c * kWave + t], o);
                acc[c] = o;
            }
        }
    } else {  // ALGO_FIR_VEC
        // convolve_and_sum.c:158-192 + sum8 :132-153: 8 independent fma lanes over tap blocks, fixed tree, out +=
        const int T = a.n_taps;
        const float* __restrict__ trow = taps + row * T;
        for (int ms = 0; ms < mc; ++ms) {
            const float* __restrict__ h = trow + ms * T;
            const float* r = lds + ms * rs + (a.lead - T / 2) + lane;
            if (T == 8) {   // one AVX block: the eight fma lanes start from 0, i.e. they are plain products
                const float h0 = h[0], h1 = h[1], h2 = h[2], h3 = h[3], h4 = h[4], h5 = h[5], h6 = h[6], h7 = h[7];
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    const float* x = r + c * kWave;
                    const float q0 = x[0] * h0 + x[4] * h4, q1 = x[1] * h1 + x[5] * h5, q2 = x[2] * h2 + x[6] * h6, q3 = x[3] * h3 + x[7] * h7;
                    acc[c] += (q0 + q2) + (q1 + q3);
                }
                continue;
            }
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                float l0 = 0.f, l1 = 0.f, l2 = 0.f, l3 = 0.f, l4 = 0.f, l5 = 0.f, l6 = 0.f, l7 = 0.f;
                for (int t = 0; t < T; t += 8) {
                    const float* x = r + c * kWave + t;
                    l0 = __fmaf_rn(x[0], h[t + 0], l0); l1 = __fmaf_rn(x[1], h[t + 1], l1);
                    l2 = __fmaf_rn(x[2], h[t + 2], l2); l3 = __fmaf_rn(x[3], h[t + 3], l3);
                    l4 = __fmaf_rn(x[4], h[t + 4], l4); l5 = __fmaf_rn(x[5], h[t + 5], l5);
                    l6 = __fmaf_rn(x[6], h[t + 6], l6); l7 = __fmaf_rn(x[7], h[t + 7], l7);
                }
                const float q0 = l0 + l4, q1 = l1 + l5, q2 = l2 + l6, q3 = l3 + l7;
                const float d0 = q0 + q2, d1 = q1 + q3;
                acc[c] += d0 + d1;
            }
        }
    }
}

// ---- "quad" layout for blocks of up to 256 samples (the reference's N_SAMPLES) ------------------------------
// Lane l owns the four CONSECUTIVE samples 4l .. 4l+3, so one conflict-free ds_read_b128 per (direction, mic)
// brings all 256 samples of a mic row into the wave: a quarter of the LDS cycles of the strided layout above.
// A delay p = 4q + r shifts the row by q whole quads (folded into the 16-byte-aligned LDS address) and by r
// samples inside the quad: those r leading values come from the previous lane's quad through DPP (wave_shr:1,
// lane 0 receives 0 = the zero prefix), and r is wave-uniform, so the four register alignments are four
// scalar-branch targets -- no per-lane select.  Mic order and operation order are unchanged.
__device__ __forceinline__ float lane_prev(float x)
{
    // wave_shr:1 (0x138), all rows/banks, bound_ctrl: lane 0 reads 0
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x138, 0xf, 0xf, true));
}

// One (direction, mic) step on a quad, as ONE inline-asm statement: a two-level scalar branch on r = p & 3
// selects one of four straight-line bodies.  Written in asm because (a) the previous lane's values are consumed
// through the DPP operand of the instruction that uses them (no v_mov), (b) the accumulators stay in the same
// four VGPRs on every path (hipcc's structurizer otherwise turns the wave-uniform switch into "flow" blocks
// with register copies), and (c) the branch is a plain s_cbranch_scc on an SGPR.
// DPP control: wave_shr:1 = src0 comes from lane-1; bound_ctrl:1 = lane 0 reads 0 (the row's zero prefix).
// The DPP sources are written by ds_read, not by a VALU instruction, so no VALU->DPP wait states are owed; a
// v_mov_b32_dpp result is consumed by ordinary (non-DPP) reads, which need none either.
#define BF_DPP " wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"

template <int ALGO>
__device__ __forceinline__ void quad_one(float (&acc)[4], const float4 Q, int p, float h, int lane)
{
    int tmp;
    if constexpr (ALGO == ALGO_PAD) {
        // out[k] += s[k - p]:   acc[t] += W[4 - r + t],  W = [previous lane's quad | Q]
        asm volatile(
            "s_and_b32 %[t], %[p], 3\n\t"
            "s_cmp_lt_u32 %[t], 2\n\t"
            "s_cbranch_scc1 .Lbf_lo_%=\n\t"
            "s_cmp_eq_u32 %[t], 2\n\t"
            "s_cbranch_scc1 .Lbf_r2_%=\n\t"
            /* r = 3 */
            "v_add_f32_dpp %[a0], %[qy], %[a0]" BF_DPP
            "v_add_f32_dpp %[a1], %[qz], %[a1]" BF_DPP
            "v_add_f32_dpp %[a2], %[qw], %[a2]" BF_DPP
            "v_add_f32 %[a3], %[a3], %[qx]\n\t"
            "s_branch .Lbf_end_%=\n"
            ".Lbf_r2_%=:\n\t"
            "v_add_f32_dpp %[a0], %[qz], %[a0]" BF_DPP
            "v_add_f32_dpp %[a1], %[qw], %[a1]" BF_DPP
            "v_add_f32 %[a2], %[a2], %[qx]\n\t"
            "v_add_f32 %[a3], %[a3], %[qy]\n\t"
            "s_branch .Lbf_end_%=\n"
            ".Lbf_lo_%=:\n\t"
            "s_cmp_eq_u32 %[t], 0\n\t"
            "s_cbranch_scc1 .Lbf_r0_%=\n\t"
            /* r = 1 */
            "v_add_f32_dpp %[a0], %[qw], %[a0]" BF_DPP
            "v_add_f32 %[a1], %[a1], %[qx]\n\t"
            "v_add_f32 %[a2], %[a2], %[qy]\n\t"
            "v_add_f32 %[a3], %[a3], %[qz]\n\t"
            "s_branch .Lbf_end_%=\n"
            ".Lbf_r0_%=:\n\t"
            "v_add_f32 %[a0], %[a0], %[qx]\n\t"
            "v_add_f32 %[a1], %[a1], %[qy]\n\t"
            "v_add_f32 %[a2], %[a2], %[qz]\n\t"
            "v_add_f32 %[a3], %[a3], %[qw]\n"
            ".Lbf_end_%=:"
            : [a0] "+v"(acc[0]), [a1] "+v"(acc[1]), [a2] "+v"(acc[2]), [a3] "+v"(acc[3]), [t] "=&s"(tmp)
            : [p] "s"(p), [qx] "v"(Q.x), [qy] "v"(Q.y), [qz] "v"(Q.z), [qw] "v"(Q.w)
            : "scc");
    } else {
        // out[k] += a + h * (b - a),  a = s[k - p - 1], b = s[k - p];  nothing for k <= p.
        // With p = 4q + r and W = [previous lane's quad | Q]:  a_t = W[3 - r + t], b_t = W[4 - r + t], always inside W.
        // With the zero prefix the only sample that would wrongly receive something is k == p (a = 0, b = s[0]):
        // lane q, element r -- cleared by the v_and with `keep`.
        // Per element, in the reference's order: d = b - a;  v = fma(h, d, a);  acc += v.
        const int keep = (lane != (p >> 2)) ? -1 : 0;
        float w0, w1, w2, w3, t0, t1, t2, t3;
        asm volatile(
            "s_and_b32 %[t], %[p], 3\n\t"
            "s_cmp_lt_u32 %[t], 2\n\t"
            "s_cbranch_scc1 .Lbf_lo_%=\n\t"
            "s_cmp_eq_u32 %[t], 2\n\t"
            "s_cbranch_scc1 .Lbf_r2_%=\n\t"
            /* r = 3: w = [pQx, pQy, pQz, pQw, Qx] */
            "v_mov_b32_dpp %[w0], %[qx]" BF_DPP
            "v_mov_b32_dpp %[w1], %[qy]" BF_DPP
            "v_mov_b32_dpp %[w2], %[qz]" BF_DPP
            "v_mov_b32_dpp %[w3], %[qw]" BF_DPP
            "v_sub_f32 %[t0], %[w1], %[w0]\n\tv_sub_f32 %[t1], %[w2], %[w1]\n\tv_sub_f32 %[t2], %[w3], %[w2]\n\tv_sub_f32 %[t3], %[qx], %[w3]\n\t"
            "v_fma_f32 %[t0], %[h], %[t0], %[w0]\n\tv_fma_f32 %[t1], %[h], %[t1], %[w1]\n\tv_fma_f32 %[t2], %[h], %[t2], %[w2]\n\tv_fma_f32 %[t3], %[h], %[t3], %[w3]\n\t"
            "v_and_b32 %[t3], %[t3], %[keep]\n\t"
            "s_branch .Lbf_end_%=\n"
            ".Lbf_r2_%=:\n\t"
            /* r = 2: w = [pQy, pQz, pQw, Qx, Qy] */
            "v_mov_b32_dpp %[w0], %[qy]" BF_DPP
            "v_mov_b32_dpp %[w1], %[qz]" BF_DPP
            "v_mov_b32_dpp %[w2], %[qw]" BF_DPP
            "v_sub_f32 %[t0], %[w1], %[w0]\n\tv_sub_f32 %[t1], %[w2], %[w1]\n\tv_sub_f32 %[t2], %[qx], %[w2]\n\tv_sub_f32 %[t3], %[qy], %[qx]\n\t"
            "v_fma_f32 %[t0], %[h], %[t0], %[w0]\n\tv_fma_f32 %[t1], %[h], %[t1], %[w1]\n\tv_fma_f32 %[t2], %[h], %[t2], %[w2]\n\tv_fma_f32 %[t3], %[h], %[t3], %[qx]\n\t"
            "v_and_b32 %[t2], %[t2], %[keep]\n\t"
            "s_branch .Lbf_end_%=\n"
            ".Lbf_lo_%=:\n\t"
            "s_cmp_eq_u32 %[t], 0\n\t"
            "s_cbranch_scc1 .Lbf_r0_%=\n\t"
            /* r = 1: w = [pQz, pQw, Qx, Qy, Qz] */
            "v_mov_b32_dpp %[w0], %[qz]" BF_DPP
            "v_mov_b32_dpp %[w1], %[qw]" BF_DPP
            "v_sub_f32 %[t0], %[w1], %[w0]\n\tv_sub_f32 %[t1], %[qx], %[w1]\n\tv_sub_f32 %[t2], %[qy], %[qx]\n\tv_sub_f32 %[t3], %[qz], %[qy]\n\t"
            "v_fma_f32 %[t0], %[h], %[t0], %[w0]\n\tv_fma_f32 %[t1], %[h], %[t1], %[w1]\n\tv_fma_f32 %[t2], %[h], %[t2], %[qx]\n\tv_fma_f32 %[t3], %[h], %[t3], %[qy]\n\t"
            "v_and_b32 %[t1], %[t1], %[keep]\n\t"
            "s_branch .Lbf_end_%=\n"
            ".Lbf_r0_%=:\n\t"
            /* r = 0: w = [pQw, Qx, Qy, Qz, Qw] */
            "v_mov_b32_dpp %[w0], %[qw]" BF_DPP
            "v_sub_f32 %[t0], %[qx], %[w0]\n\tv_sub_f32 %[t1], %[qy], %[qx]\n\tv_sub_f32 %[t2], %[qz], %[qy]\n\tv_sub_f32 %[t3], %[qw], %[qz]\n\t"
            "v_fma_f32 %[t0], %[h], %[t0], %[w0]\n\tv_fma_f32 %[t1], %[h], %[t1], %[qx]\n\tv_fma_f32 %[t2], %[h], %[t2], %[qy]\n\tv_fma_f32 %[t3], %[h], %[t3], %[qz]\n\t"
            "v_and_b32 %[t0], %[t0], %[keep]\n"
            ".Lbf_end_%=:\n\t"
            "v_add_f32 %[a0], %[a0], %[t0]\n\tv_add_f32 %[a1], %[a1], %[t1]\n\tv_add_f32 %[a2], %[a2], %[t2]\n\tv_add_f32 %[a3], %[a3], %[t3]"
            : [a0] "+v"(acc[0]), [a1] "+v"(acc[1]), [a2] "+v"(acc[2]), [a3] "+v"(acc[3]), [t] "=&s"(tmp),
              [w0] "=&v"(w0), [w1] "=&v"(w1), [w2] "=&v"(w2), [w3] "=&v"(w3), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3)
            : [p] "s"(p), [h] "s"(h), [keep] "v"(keep), [qx] "v"(Q.x), [qy] "v"(Q.y), [qz] "v"(Q.z), [qw] "v"(Q.w)
            : "scc");
    }
}

template <int ALGO>
__device__ __forceinline__ void accumulate_quad(float (&acc)[4], const float* lds, const KArgs& a, const int32_t* __restrict__ whole,
                                                const float* __restrict__ frac, size_t row_base, int m0, int mc, int lane,
                                                const RowHead head = RowHead())
{
    static_assert(ALGO == ALGO_PAD || ALGO == ALGO_LERP, "quad layout: pad and lerp");
    const size_t row = row_base + m0;
    const int rs = a.row_stride;
    const int32_t* __restrict__ wrow = whole + row;
    const float* __restrict__ hrow = frac + row;
    const float* base = lds + a.lead + 4 * lane;
    // table rows: one coalesced vector load per 64 mics, entries scalarised with v_readlane (see accumulate())
    int vp = head.p;
    float vh = head.h;
    if (!head.valid) {
        vp = (lane < mc) ? wrow[lane] : 0;
        if constexpr (ALGO == ALGO_LERP) vh = (lane < mc) ? hrow[lane] : 0.0f;
    }
    for (int b0 = 0; b0 < mc; b0 += kWave) {
        const int bn = min(kWave, mc - b0);
        int vp_next = 0;
        float vh_next = 0.0f;
        if (b0 + kWave < mc) {
            vp_next = (b0 + kWave + lane < mc) ? wrow[b0 + kWave + lane] : 0;
            if constexpr (ALGO == ALGO_LERP) vh_next = (b0 + kWave + lane < mc) ? hrow[b0 + kWave + lane] : 0.0f;
        }
        int u = 0;
        for (; u + 4 <= bn; u += 4) {
            int p[4];
            float h[4];
            float4 Q[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                p[i] = __builtin_amdgcn_readlane(vp, u + i);
                h[i] = 0.f;
                if constexpr (ALGO == ALGO_LERP) h[i] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(vh), u + i));
                Q[i] = *reinterpret_cast<const float4*>(base + (b0 + u + i) * rs - (p[i] & ~3));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) quad_one<ALGO>(acc, Q[i], p[i], h[i], lane);
        }
        for (; u < bn; ++u) {
            const int p = __builtin_amdgcn_readlane(vp, u);
            float h = 0.f;
            if constexpr (ALGO == ALGO_LERP) h = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(vh), u));
            const float4 Q = *reinterpret_cast<const float4*>(base + (b0 + u) * rs - (p & ~3));
            quad_one<ALGO>(acc, Q, p, h, lane);
        }
        vp = vp_next;
        vh = vh_next;
    }
}

// ---- mean power, in the reference's summation order ------------------------------------------------------
// The reference finishes a direction with (pad_and_sum.c:122-131)
//     for k: out[k] /= n; sum += out[k]^2          (gcc: vdivps, vmulps, then one vaddss per k, in k order)
//     image = sum / N
// A float32 sum of N squares taken in another order differs from that by up to ~N*2^-24 relative (2e-5 observed
// at N = 1024), which is more than the 1e-5 parity bar.  So the squares are summed in k order here too:
// every wave parks the squares of `pbw` finished directions as rows of a private LDS scratch, then lanes
// 0..pbw-1 each walk one row front to back (ds_read_b128, four ordered adds per read).
template <int NC, bool QUAD>
__device__ __forceinline__ void park_squares(const float (&acc)[NC], float* scratch_row, const KArgs& a, int d, int lane)
{
    float sq[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        // out[k] /= (float)n: for a power-of-two n the reciprocal multiply is exact; otherwise a true division
        const float o = a.n_is_pow2 ? acc[c] * a.inv_n : acc[c] / (float)a.n_mics;
        sq[c] = o * o;
    }
    if constexpr (QUAD) {
        // lane l owns samples 4l..4l+3: one aligned 16-byte store (entries past N are never summed)
        *reinterpret_cast<float4*>(scratch_row + 4 * lane) = make_float4(sq[0], sq[1], sq[2], sq[3]);
    } else {
#pragma unroll
        for (int c = 0; c < NC; ++c) scratch_row[lane + c * kWave] = sq[c];
    }
    if (lane == 0) scratch_row[a.srow - 4] = __int_as_float(d);   // the pad column carries the direction id
}

// Rows are 16-byte aligned and 4 (mod 64) dwords apart, so up to 16 lanes can each stream their own row with
// ds_read_b128 without sharing a bank; the additions stay strictly in k order.
__device__ __forceinline__ void flush_powers(const float* scratch, int filled, float* __restrict__ img, const KArgs& a, int lane)
{
    if (lane < filled) {
        const float* row = scratch + lane * a.srow;
        const float4* row4 = reinterpret_cast<const float4*>(row);
        const int n = a.n_samples;
        float sum = 0.0f;
        int k = 0;
#pragma unroll 4
        for (; k + 4 <= n; k += 4) {
            const float4 v = row4[k >> 2];
            sum += v.x; sum += v.y; sum += v.z; sum += v.w;
        }
        for (; k < n; ++k) sum += row[k];
        const int d = __float_as_int(row[a.srow - 4]);
        img[d - a.image_origin] = sum / (float)n;
    }
}

template <int ALGO, int NC, int DPW, bool QUAD>
__global__ void __launch_bounds__(1024) das_mimo_kernel(BF_TABLE_PARAMS, KArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int nwaves = (int)(blockDim.x >> 6);
    const int tile = (int)(blockIdx.x % (unsigned)a.n_tiles);
    const int frame = (int)(blockIdx.x / (unsigned)a.n_tiles);
    const int tile_begin = a.dir_begin + tile * a.tile_dirs;
    if (tile_begin >= a.dir_end) return;  // padding tile (n_tiles is rounded up to a multiple of 8)
    const int tile_end = min(tile_begin + a.tile_dirs, a.dir_end);

    // zero the whole LDS image once: the lead/tail columns and unused rows stay zero for the kernel's lifetime
    {
        const int total4 = (a.mic_chunk * a.row_stride) >> 2;
        float4* z = reinterpret_cast<float4*>(lds);
        for (int i = threadIdx.x; i < total4; i += blockDim.x) z[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();

    const float* __restrict__ frame_sig = signals + (size_t)frame * a.m_total * a.n_samples;
    float* __restrict__ img = images + (size_t)frame * a.image_stride;
    const int group = nwaves * DPW;
    float* scratch = lds + a.scratch_off + wave * (a.pbw * a.srow);
    int filled = 0;   // wave-uniform

    // Work items of this wave, in order: for g0 / for chunk / for j.  The table-row head of item i+1 is requested
    // (vector load, its own wait counter) before item i is computed, so its HBM/L2 latency hides behind ~64 mics of work.
    auto item_dir = [&](int g0_, int j_) { return g0_ + j_ * nwaves + wave; };
    RowHead head;
    if (item_dir(tile_begin, 0) < tile_end)
        head = request_row_head<ALGO>(whole, frac, (size_t)item_dir(tile_begin, 0) * a.n_mics, min(a.mic_chunk, a.n_mics), lane);

    for (int g0 = tile_begin; g0 < tile_end; g0 += group) {
        float acc[DPW][NC];
#pragma unroll
        for (int j = 0; j < DPW; ++j)
#pragma unroll
            for (int c = 0; c < NC; ++c) acc[j][c] = 0.0f;

        for (int ch = 0; ch < a.n_chunks; ++ch) {
            const int m0 = ch * a.mic_chunk;
            const int mc = min(a.mic_chunk, a.n_mics - m0);
            if (a.n_chunks > 1 || g0 == tile_begin) {
                if (a.n_chunks > 1 && (ch > 0 || g0 != tile_begin)) __syncthreads();  // previous readers done
                stage_chunk(lds, a, mics, frame_sig, m0, mc, wave, nwaves, lane);
                __syncthreads();
            }
#pragma unroll
            for (int j = 0; j < DPW; ++j) {
                const int d = g0 + j * nwaves + wave;  // wave-uniform
                // successor of (g0, ch, j)
                int ng0 = g0, nch = ch, nj = j + 1;
                if (nj == DPW) { nj = 0; nch = ch + 1; if (nch == a.n_chunks) { nch = 0; ng0 = g0 + group; } }
                const int nd = item_dir(ng0, nj);
                const RowHead cur = head;
                head = RowHead();
                if (nd < tile_end) {
                    const int nm0 = nch * a.mic_chunk;
                    head = request_row_head<ALGO>(whole, frac, (size_t)nd * a.n_mics + nm0, min(a.mic_chunk, a.n_mics - nm0), lane);
                }
                if (d < tile_end) {
                    if constexpr (QUAD) accumulate_quad<ALGO>(acc[j], lds, a, whole, frac, (size_t)d * a.n_mics, m0, mc, lane, cur);
                    else accumulate<ALGO, NC>(acc[j], lds, a, whole, frac, taps, (size_t)d * a.n_mics, m0, mc, lane, cur);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < DPW; ++j) {
            const int d = g0 + j * nwaves + wave;
            if (d < tile_end) {
                if (a.debug & 1) {   // profiling only
                    float t = 0.f;
#pragma unroll
                    for (int c = 0; c < NC; ++c) t += acc[j][c];
                    if (lane == 0) img[d - a.image_origin] = t;
                    continue;
                }
                park_squares<NC, QUAD>(acc[j], scratch + filled * a.srow, a, d, lane);
                if (++filled == a.pbw) { flush_powers(scratch, filled, img, a, lane); filled = 0; }
            }
        }
    }
    if (filled > 0) flush_powers(scratch, filled, img, a, lane);
}

// One direction, raw out[N] (no division): miso_pad / miso_lerp / miso_convolve_* (pad_and_sum.c:54-70 ...).
template <int ALGO, int NC>
__global__ void __launch_bounds__(64) das_miso_kernel(BF_TABLE_PARAMS, const float* __restrict__ miso_init, float* __restrict__ miso_out, KArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & (kWave - 1);
    {
        const int total4 = (a.mic_chunk * a.row_stride) >> 2;
        float4* z = reinterpret_cast<float4*>(lds);
        for (int i = threadIdx.x; i < total4; i += blockDim.x) z[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
    float acc[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c)
        acc[c] = (miso_init != nullptr && lane + c * kWave < a.n_samples) ? miso_init[lane + c * kWave] : 0.0f;
    for (int ch = 0; ch < a.n_chunks; ++ch) {
        const int m0 = ch * a.mic_chunk;
        const int mc = min(a.mic_chunk, a.n_mics - m0);
        if (ch > 0) __syncthreads();
        stage_chunk(lds, a, mics, signals, m0, mc, 0, 1, lane);
        __syncthreads();
        accumulate<ALGO, NC>(acc, lds, a, whole, frac, taps, (size_t)a.miso_row, m0, mc, lane);
    }
#pragma unroll
    for (int c = 0; c < NC; ++c)
        if (lane + c * kWave < a.n_samples) miso_out[lane + c * kWave] = acc[c];
}

// ==================================================================================================
// "Shifted-copies" layout (copies::das_copies_kernel below).
//
//   * every staged mic row is kept in C copies shifted by 0..C-1 samples, so a delay p = C q + r becomes an ALIGNED read
//     from copy r (C = 4: 16-byte ds_read_b128 for the kernels that read at every step; C = 2: pairs of 8-byte
//     ds_read_b64 for the pad / lerp sweep, which re-reads rarely -- half the staging and half the LDS per mic): no
//     sub-quad alignment work, no branches;
//   * lane l owns the four consecutive samples 4l..4l+3 of a 256-sample segment (one quad read = a whole segment row);
//   * for lerp the staged rows also carry D[i] = s[i+1] - s[i] (rounded exactly as the reference's subtraction,
//     D[-1] = 0), so a sample is fma(h, D[i], s[i]) -- the reference's own two roundings -- and the i < 0 guard
//     falls out of the zero prefix;
//   * the copies of a whole frame do not fit in LDS, so the mics are staged in chunks of up to 32 and every wave
//     carries its directions' accumulators across the chunks; the next chunk's samples are already in registers
//     (global loads issued a chunk ahead) when the buffer is rewritten;
//   * the chunk buffer doubles as the scratch for the k-ordered power sum.
// Mic order and operation order are unchanged, so the maps stay bit-identical to the CPU reference.
// Digest of a `whole` table for the shifted-copies layout: entry (d, m) -> LDS byte offset of the aligned quad row that
// direction d reads for staged mic m % mic_chunk (copy p mod C, shifted back by p - p mod C samples; lerp reads one sample
// earlier, p + 1).  Built once per (table, layout) so that the hot loop gets its addresses with scalar loads only.
__global__ void __launch_bounds__(256) digest_kernel(const int32_t* __restrict__ whole, int32_t* __restrict__ digest, long long entries, int n_mics,
                                                     int mic_chunk, int arrays, int row_stride, int lead, int bias, int ncopies)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < entries; i += (long long)gridDim.x * blockDim.x) {
        const int m = (int)(i % n_mics) % mic_chunk;
        const int pd = whole[i] + bias;
        digest[i] = ((m * arrays * ncopies + (pd & (ncopies - 1))) * row_stride + lead - (pd & ~(ncopies - 1))) * 4;
    }
}

// Grouped digest for pad / lerp: directions are taken in groups of `gdirs` (= the directions one wave carries) counted
// from dir_begin, and the group's entries of one mic sit together: entry ((g * M + m) * gdirs + j) = direction
// dir_begin + g * gdirs + j, mic m, so one s_load_dwordx8 brings the 8 directions' offsets of a mic.  The lerp weights
// follow in the same buffer (float, same order) at `h_off`.  Directions past dir_end repeat the last one (their results
// are never stored).
__global__ void __launch_bounds__(256) digest_grouped_kernel(const int32_t* __restrict__ whole, const float* __restrict__ frac, int32_t* __restrict__ digest,
                                                             long long entries, long long h_off, int n_mics, int gdirs, int dir_begin, int dir_end,
                                                             int mic_chunk, int arrays, int row_stride, int lead, int bias, int ncopies,
                                                             unsigned long long* __restrict__ reload_count, int pack_guards, int scale,
                                                             const float* __restrict__ taps, long long t_off)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < entries; i += (long long)gridDim.x * blockDim.x) {
        const int j = (int)(i % gdirs);
        const long long gm = i / gdirs;
        const int mic = (int)(gm % n_mics);
        const long long g = gm / n_mics;
        long long d = dir_begin + g * gdirs + j;
        if (d > dir_end - 1) d = dir_end - 1;
        if (taps != nullptr) {
            // the 8 taps of (direction, mic), regrouped like the entries: [group][mic][direction of the group][8] -- the FIR sweep
            // then walks ONE pointer per wave (eight row pointers cost it 14 scalar registers it does not have)
            const float4* src = reinterpret_cast<const float4*>(taps + ((size_t)d * n_mics + mic) * 8);
            float4* dst = reinterpret_cast<float4*>(reinterpret_cast<float*>(digest) + t_off + i * 8);
            dst[0] = src[0];
            dst[1] = src[1];
        }
        if (whole == nullptr) continue;                 // the plain FIRs have no whole-sample table
        const int m = mic % mic_chunk;
        const int pd = whole[d * n_mics + mic] + bias;
        // (scale = 2: rows hold two frames interleaved sample by sample, a sample is two floats wide)
        digest[i] = ((m * arrays * ncopies + (pd & (ncopies - 1))) * row_stride + scale * (lead - (pd & ~(ncopies - 1)))) * 4;
        if (frac != nullptr) reinterpret_cast<float*>(digest)[h_off + i] = frac[d * n_mics + mic];
        if (pack_guards) {
            // hybrid: output j of a lane (sample 4 lane + j) is live for 4 lane + j > p, i.e. from lane n_j = (p + 4 - j) >> 2 on;
            // the four n_j, one byte each (hybrid_convolve_and_sum.c:58: the loop starts at i = 0, i.e. sample p + 1)
            const int p = whole[d * n_mics + mic];
            digest[h_off + i] = ((p + 4) >> 2) | (((p + 3) >> 2) << 8) | (((p + 2) >> 2) << 16) | (((p + 1) >> 2) << 24);
        }
        if (reload_count != nullptr && j > 0) {
            // how often the sweep will have to re-read: this direction's delay differs from the previous direction's
            // (directions past the end repeat the last one: never a change)
            const long long dn = dir_begin + g * gdirs + j;
            if (dn <= dir_end - 1 && whole[(d - 1) * n_mics + mic] != whole[d * n_mics + mic]) atomicAdd(reload_count, 1ull);
        }
    }
}

namespace copies {

constexpr int kWaves = 16;       // waves per workgroup (8 for pad / lerp at N <= 256: two workgroups per CU cover each other's barriers)

__device__ __forceinline__ float dpp_prev(float x)   // lane-1's value, 0 in lane 0
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float dpp_next(float x)   // lane+1's value, 0 in lane 63
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x130, 0xf, 0xf, true));
}
__device__ __forceinline__ float lane_value(float x, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), l)); }

// Shifted copies kept per staged array.  The sweep of pad / lerp re-reads rarely and reads 8-byte halves: one copy per
// delay mod 2 is enough (half the staging writes and half the LDS per mic).  The kernels that read at every step -- the
// 8-tap FIR flavours and the direction-outer (DIRECT) variant of pad / lerp -- need ds_read_b128: one copy per delay mod 4.
__host__ __device__ constexpr int copies_of(int algo, bool direct) { return ((algo == ALGO_PAD || algo == ALGO_LERP) && !direct) ? 2 : 4; }

// Geometry of the shifted-copies layout for a block of NSEG x 256 samples (N <= 256: 1, <= 512: 2, <= 1024: 4).
//   * a wave owns DW directions x NSEG segments of 256 samples (one aligned quad per lane per segment);
//   * compile-time row stride (RS > 0) when the largest delay fits kLead: the D / segment reads become immediate offsets.
template <int NSEG>
struct Geo {
    static constexpr int kDw = NSEG == 1 ? 8 : NSEG == 2 ? 8 : 4;   // directions per wave
    static constexpr int kBatch = 4 / NSEG;                          // mics whose reads are in flight together
    static constexpr int kLead = NSEG == 1 ? 56 : 64;                // zero prefix of the fixed-stride variant (56: the as-shipped array's delays, up to 47 samples, still fit; 32 lerp mics x 4 rows x 312 floats = 156 KiB)
    static constexpr int kRs = NSEG * 256 + kLead;                   // its row stride
    static constexpr int kPark = NSEG * 256 + 4;                     // floats per parked row of squares
    static constexpr int kFirTail = 8;                               // 8-tap FIR rows: the reference's zero padding after the block
    static constexpr int kRsFir = kRs + kFirTail;
};

// One 8-tap FIR step of the hybrid beamformer on a quad (hybrid_convolve_and_sum.c:51-64): output j of the lane (sample
// k = 4 lane + j) takes  o_j = fma(h_t, W[j + t], o_j), t = 0..7 in order, but only where k > p (the reference starts at
// i = 0, i.e. k = p + 1).  The live lanes of output j are a wave-uniform suffix of the wave, so the four guards are four
// EXEC masks set by the scalar unit -- no per-lane compare / select on the VALU.  One asm statement: the compiler must
// not move anything between the EXEC writes.
__device__ __forceinline__ void fir8_masked(float (&o)[4], const float (&W)[12], const float (&h)[8], unsigned long long m0, unsigned long long m1,
                                            unsigned long long m2, unsigned long long m3)
{
    unsigned long long saved;
    asm volatile(
        "s_mov_b64 %[sv], exec\n\t"
        "s_mov_b64 exec, %[m0]\n\t"
        "v_fmac_f32 %[o0], %[h0], %[w0]\n\tv_fmac_f32 %[o0], %[h1], %[w1]\n\tv_fmac_f32 %[o0], %[h2], %[w2]\n\tv_fmac_f32 %[o0], %[h3], %[w3]\n\t"
        "v_fmac_f32 %[o0], %[h4], %[w4]\n\tv_fmac_f32 %[o0], %[h5], %[w5]\n\tv_fmac_f32 %[o0], %[h6], %[w6]\n\tv_fmac_f32 %[o0], %[h7], %[w7]\n\t"
        "s_mov_b64 exec, %[m1]\n\t"
        "v_fmac_f32 %[o1], %[h0], %[w1]\n\tv_fmac_f32 %[o1], %[h1], %[w2]\n\tv_fmac_f32 %[o1], %[h2], %[w3]\n\tv_fmac_f32 %[o1], %[h3], %[w4]\n\t"
        "v_fmac_f32 %[o1], %[h4], %[w5]\n\tv_fmac_f32 %[o1], %[h5], %[w6]\n\tv_fmac_f32 %[o1], %[h6], %[w7]\n\tv_fmac_f32 %[o1], %[h7], %[w8]\n\t"
        "s_mov_b64 exec, %[m2]\n\t"
        "v_fmac_f32 %[o2], %[h0], %[w2]\n\tv_fmac_f32 %[o2], %[h1], %[w3]\n\tv_fmac_f32 %[o2], %[h2], %[w4]\n\tv_fmac_f32 %[o2], %[h3], %[w5]\n\t"
        "v_fmac_f32 %[o2], %[h4], %[w6]\n\tv_fmac_f32 %[o2], %[h5], %[w7]\n\tv_fmac_f32 %[o2], %[h6], %[w8]\n\tv_fmac_f32 %[o2], %[h7], %[w9]\n\t"
        "s_mov_b64 exec, %[m3]\n\t"
        "v_fmac_f32 %[o3], %[h0], %[w3]\n\tv_fmac_f32 %[o3], %[h1], %[w4]\n\tv_fmac_f32 %[o3], %[h2], %[w5]\n\tv_fmac_f32 %[o3], %[h3], %[w6]\n\t"
        "v_fmac_f32 %[o3], %[h4], %[w7]\n\tv_fmac_f32 %[o3], %[h5], %[w8]\n\tv_fmac_f32 %[o3], %[h6], %[w9]\n\tv_fmac_f32 %[o3], %[h7], %[w10]\n\t"
        "s_mov_b64 exec, %[sv]"
        : [o0] "+v"(o[0]), [o1] "+v"(o[1]), [o2] "+v"(o[2]), [o3] "+v"(o[3]), [sv] "=&s"(saved)
        : [h0] "v"(h[0]), [h1] "v"(h[1]), [h2] "v"(h[2]), [h3] "v"(h[3]), [h4] "v"(h[4]), [h5] "v"(h[5]), [h6] "v"(h[6]), [h7] "v"(h[7]),
          [w0] "v"(W[0]), [w1] "v"(W[1]), [w2] "v"(W[2]), [w3] "v"(W[3]), [w4] "v"(W[4]), [w5] "v"(W[5]), [w6] "v"(W[6]), [w7] "v"(W[7]),
          [w8] "v"(W[8]), [w9] "v"(W[9]), [w10] "v"(W[10]), [m0] "s"(m0), [m1] "s"(m1), [m2] "s"(m2), [m3] "s"(m3));
}

// What one thread holds of a staged (mic, segment) pair between its global load and its LDS write.
struct Staged {
    float4 v;     // samples 4q .. 4q+3, q = 64 seg + lane
    float edge;   // NSEG > 1: lanes 0..2 hold s[4q-3 .. 4q-1] of the segment's first quad, lane 63 holds s[4q+4] of its last
};

// Write the NC shifted copies of a segment: copy c holds the row shifted right by c samples, i.e. its aligned
// quad i is (x[4i-c], ..., x[4i-c+3]); (py, pz, pw) are x[4q-3 .. 4q-1] (previous lane, or the segment edge).
// NC = 4 serves 16-byte reads at any delay (the FIR flavours' ds_read_b128), NC = 2 the 8-byte reads of pad / lerp.
template <int NC>
__device__ __forceinline__ void write_copies(float* row0, int rs, int col, int lane, float4 v, float py, float pz, float pw)
{
    float4* q0 = reinterpret_cast<float4*>(row0 + 0 * rs + col) + lane;
    float4* q1 = reinterpret_cast<float4*>(row0 + 1 * rs + col) + lane;
    *q0 = v;
    *q1 = make_float4(pw, v.x, v.y, v.z);
    if constexpr (NC == 4) {
        float4* q2 = reinterpret_cast<float4*>(row0 + 2 * rs + col) + lane;
        float4* q3 = reinterpret_cast<float4*>(row0 + 3 * rs + col) + lane;
        *q2 = make_float4(pz, pw, v.x, v.y);
        *q3 = make_float4(py, pz, pw, v.x);
    }
}


typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// The quad (two register pairs) of one 256-sample segment of a staged mic row, and of its difference row for lerp.
struct Quad { f32x2 lo, hi; };

// (Re)load the NSEG quads of a mic for LDS byte offset `e` -- unless `e` equals the offset they were loaded for (`ep`):
// a scalar compare and branch, no vector work and no wait when the quads are still valid; a reload waits for its reads,
// so the quads are usable afterwards.  One asm statement: the compiler sees an opaque in-place update of the registers and
// cannot hoist, duplicate or speculate the reads.
#define BF_RD(sg, off, off8)                                                   \
    "ds_read_b64 %[s" #sg "l], %[ad] offset:" #off "\n\t"                      \
    "ds_read_b64 %[s" #sg "h], %[ad] offset:" #off8 "\n\t"
#define BF_RDD(sg, off, off8)                                                  \
    "ds_read_b64 %[d" #sg "l], %[ad2] offset:" #off "\n\t"                     \
    "ds_read_b64 %[d" #sg "h], %[ad2] offset:" #off8 "\n\t"
// The reload path lives out of line (subsection 1 of the text section): the common case -- offset unchanged -- is a
// compare and a NOT-taken branch, which costs the wave nothing; a taken branch per step did (instruction refetch).
#define BF_RELOAD_HEAD "s_cmp_lg_u32 %[e], %[ep]\n\ts_cbranch_scc1 .Lreload_%=\n.Lback_%=:\n\t.subsection 1\n.Lreload_%=:\n\t"
#define BF_RELOAD_TAIL "s_waitcnt lgkmcnt(0)\n\ts_branch .Lback_%=\n\t.subsection 0"
#define BF_S_OPS(sg) [s##sg##l] "+v"(S[sg].lo), [s##sg##h] "+v"(S[sg].hi)
#define BF_D_OPS(sg) [d##sg##l] "+v"(D[sg].lo), [d##sg##h] "+v"(D[sg].hi)

template <int NSEG, bool LERP>
__device__ __forceinline__ void reload_quads(Quad (&S)[NSEG], Quad (&D)[NSEG], int e, int ep, int lbase, int d_off)
{
    int ad, ad2;
    if constexpr (NSEG == 1 && !LERP) {
        asm volatile(BF_RELOAD_HEAD "v_add_u32 %[ad], %[e], %[lb]\n\t" BF_RD(0, 0, 8) BF_RELOAD_TAIL
                     : BF_S_OPS(0), [ad] "=&v"(ad) : [e] "s"(e), [ep] "s"(ep), [lb] "v"(lbase) : "scc");
    } else if constexpr (NSEG == 1 && LERP) {
        asm volatile(BF_RELOAD_HEAD "v_add_u32 %[ad], %[e], %[lb]\n\tv_add_u32 %[ad2], %[ad], %[doff]\n\t" BF_RD(0, 0, 8)
                         BF_RDD(0, 0, 8) BF_RELOAD_TAIL
                     : BF_S_OPS(0), BF_D_OPS(0), [ad] "=&v"(ad), [ad2] "=&v"(ad2) : [e] "s"(e), [ep] "s"(ep), [lb] "v"(lbase), [doff] "s"(d_off) : "scc");
    } else if constexpr (NSEG == 2 && !LERP) {
        asm volatile(BF_RELOAD_HEAD "v_add_u32 %[ad], %[e], %[lb]\n\t" BF_RD(0, 0, 8) BF_RD(1, 1024, 1032)
                     BF_RELOAD_TAIL
                     : BF_S_OPS(0), BF_S_OPS(1), [ad] "=&v"(ad) : [e] "s"(e), [ep] "s"(ep), [lb] "v"(lbase) : "scc");
    } else if constexpr (NSEG == 2 && LERP) {
        asm volatile(BF_RELOAD_HEAD "v_add_u32 %[ad], %[e], %[lb]\n\tv_add_u32 %[ad2], %[ad], %[doff]\n\t" BF_RD(0, 0, 8)
                         BF_RDD(0, 0, 8) BF_RD(1, 1024, 1032) BF_RDD(1, 1024, 1032) BF_RELOAD_TAIL
                     : BF_S_OPS(0), BF_D_OPS(0), BF_S_OPS(1), BF_D_OPS(1), [ad] "=&v"(ad), [ad2] "=&v"(ad2)
                     : [e] "s"(e), [ep] "s"(ep), [lb] "v"(lbase), [doff] "s"(d_off) : "scc");
    } else if constexpr (NSEG == 4 && !LERP) {
        asm volatile(BF_RELOAD_HEAD "v_add_u32 %[ad], %[e], %[lb]\n\t" BF_RD(0, 0, 8) BF_RD(1, 1024, 1032) BF_RD(2, 2048, 2056)
                         BF_RD(3, 3072, 3080) BF_RELOAD_TAIL
                     : BF_S_OPS(0), BF_S_OPS(1), BF_S_OPS(2), BF_S_OPS(3), [ad] "=&v"(ad) : [e] "s"(e), [ep] "s"(ep), [lb] "v"(lbase) : "scc");
    } else {
        static_assert(NSEG == 4 && LERP, "segments: 1, 2 or 4");
        asm volatile(BF_RELOAD_HEAD "v_add_u32 %[ad], %[e], %[lb]\n\tv_add_u32 %[ad2], %[ad], %[doff]\n\t" BF_RD(0, 0, 8)
                         BF_RDD(0, 0, 8) BF_RD(1, 1024, 1032) BF_RDD(1, 1024, 1032) BF_RD(2, 2048, 2056) BF_RDD(2, 2048, 2056) BF_RD(3, 3072, 3080) BF_RDD(3, 3072, 3080)
                     BF_RELOAD_TAIL
                     : BF_S_OPS(0), BF_D_OPS(0), BF_S_OPS(1), BF_D_OPS(1), BF_S_OPS(2), BF_D_OPS(2), BF_S_OPS(3), BF_D_OPS(3), [ad] "=&v"(ad), [ad2] "=&v"(ad2)
                     : [e] "s"(e), [ep] "s"(ep), [lb] "v"(lbase), [doff] "s"(d_off) : "scc");
    }
}

// Request the NSEG quads of a mic for LDS byte offset `e` WITHOUT waiting: the caller consumes them one mic later, after
// an "s_waitcnt lgkmcnt(0)" of its own.  (The compiler does not know these reads are in flight: the registers are only
// touched again by asm statements that follow that wait.)
template <int NSEG, bool LERP>
__device__ __forceinline__ void issue_quads(Quad (&S)[NSEG], Quad (&D)[NSEG], int e, int lbase, int d_off)
{
    int ad, ad2;
    if constexpr (NSEG == 1 && !LERP) {
        asm volatile("v_add_u32 %[ad], %[e], %[lb]\n\t" BF_RD(0, 0, 8) : BF_S_OPS(0), [ad] "=&v"(ad) : [e] "s"(e), [lb] "v"(lbase));
    } else if constexpr (NSEG == 1 && LERP) {
        asm volatile("v_add_u32 %[ad], %[e], %[lb]\n\tv_add_u32 %[ad2], %[ad], %[doff]\n\t" BF_RD(0, 0, 8) BF_RDD(0, 0, 8)
                     : BF_S_OPS(0), BF_D_OPS(0), [ad] "=&v"(ad), [ad2] "=&v"(ad2) : [e] "s"(e), [lb] "v"(lbase), [doff] "s"(d_off));
    } else if constexpr (NSEG == 2 && !LERP) {
        asm volatile("v_add_u32 %[ad], %[e], %[lb]\n\t" BF_RD(0, 0, 8) BF_RD(1, 1024, 1032)
                     : BF_S_OPS(0), BF_S_OPS(1), [ad] "=&v"(ad) : [e] "s"(e), [lb] "v"(lbase));
    } else if constexpr (NSEG == 2 && LERP) {
        asm volatile("v_add_u32 %[ad], %[e], %[lb]\n\tv_add_u32 %[ad2], %[ad], %[doff]\n\t" BF_RD(0, 0, 8) BF_RDD(0, 0, 8) BF_RD(1, 1024, 1032)
                         BF_RDD(1, 1024, 1032)
                     : BF_S_OPS(0), BF_D_OPS(0), BF_S_OPS(1), BF_D_OPS(1), [ad] "=&v"(ad), [ad2] "=&v"(ad2) : [e] "s"(e), [lb] "v"(lbase), [doff] "s"(d_off));
    } else if constexpr (NSEG == 4 && !LERP) {
        asm volatile("v_add_u32 %[ad], %[e], %[lb]\n\t" BF_RD(0, 0, 8) BF_RD(1, 1024, 1032) BF_RD(2, 2048, 2056) BF_RD(3, 3072, 3080)
                     : BF_S_OPS(0), BF_S_OPS(1), BF_S_OPS(2), BF_S_OPS(3), [ad] "=&v"(ad) : [e] "s"(e), [lb] "v"(lbase));
    } else {
        asm volatile("v_add_u32 %[ad], %[e], %[lb]\n\tv_add_u32 %[ad2], %[ad], %[doff]\n\t" BF_RD(0, 0, 8) BF_RDD(0, 0, 8) BF_RD(1, 1024, 1032)
                         BF_RDD(1, 1024, 1032) BF_RD(2, 2048, 2056) BF_RDD(2, 2048, 2056) BF_RD(3, 3072, 3080) BF_RDD(3, 3072, 3080)
                     : BF_S_OPS(0), BF_D_OPS(0), BF_S_OPS(1), BF_D_OPS(1), BF_S_OPS(2), BF_D_OPS(2), BF_S_OPS(3), BF_D_OPS(3), [ad] "=&v"(ad), [ad2] "=&v"(ad2)
                     : [e] "s"(e), [lb] "v"(lbase), [doff] "s"(d_off));
    }
}
// One segment, lerp, compile-time row stride: the difference quads sit DOFF bytes after the sample quads -- one address.
template <int DOFF>
__device__ __forceinline__ void issue_quads_i(Quad (&S)[1], Quad (&D)[1], int e, int lbase)
{
    int ad;
    asm volatile("v_add_u32 %[ad], %[e], %[lb]\n\t" BF_RD(0, 0, 8)
                 "ds_read_b64 %[d0l], %[ad] offset:%[o0]\n\tds_read_b64 %[d0h], %[ad] offset:%[o1]\n\t"
                 : BF_S_OPS(0), BF_D_OPS(0), [ad] "=&v"(ad) : [e] "s"(e), [lb] "v"(lbase), [o0] "n"(DOFF), [o1] "n"(DOFF + 8));
}
// "the quads requested by issue_quads have arrived" (also drains outstanding scalar loads: same counter).  The wait is
// the compiler-visible builtin, so that the compiler knows every earlier scalar load has landed and does not insert a
// second, badly placed wait before the first use of a table entry; the empty asm ties the quad registers to this point.
template <int NSEG>
__device__ __forceinline__ void await_quads(Quad (&S)[NSEG], Quad (&D)[NSEG])
{
    __builtin_amdgcn_s_waitcnt(0xC07F);   // vmcnt(63) expcnt(7) lgkmcnt(0)
    if constexpr (NSEG == 1) {
        asm volatile("" : BF_S_OPS(0), BF_D_OPS(0));
    } else if constexpr (NSEG == 2) {
        asm volatile("" : BF_S_OPS(0), BF_D_OPS(0), BF_S_OPS(1), BF_D_OPS(1));
    } else {
        asm volatile("" : BF_S_OPS(0), BF_D_OPS(0), BF_S_OPS(1), BF_D_OPS(1), BF_S_OPS(2), BF_D_OPS(2), BF_S_OPS(3), BF_D_OPS(3));
    }
}
#undef BF_RD
#undef BF_RELOAD_HEAD
#undef BF_RELOAD_TAIL
#undef BF_RDD
#undef BF_S_OPS
#undef BF_D_OPS

// acc += quad (pad) / acc += fma(h, D, S) (lerp) on the two register pairs of a segment.  asm so that it stays between
// the reloads in program order (a C++ expression may be sunk below the next reload at the price of register copies).
__device__ __forceinline__ void add_quad(f32x2 (&ac)[2], const Quad& q)
{
    // pad_and_sum.c:41-47   out[k] += s[k - p]
    asm volatile("v_pk_add_f32 %[a0], %[a0], %[lo]\n\tv_pk_add_f32 %[a1], %[a1], %[hi]" : [a0] "+v"(ac[0]), [a1] "+v"(ac[1]) : [lo] "v"(q.lo), [hi] "v"(q.hi));
}
template <int HALF>   // which half of the SGPR pair `hp` holds this direction's weight
__device__ __forceinline__ void lerp_quad(f32x2 (&ac)[2], const Quad& q, const Quad& d, unsigned long long hp)
{
    // lerp_and_sum.c:50-56  out[k] += s[i] + h * (s[i+1] - s[i]),  i = k - p - 1   (gcc contracts it into one fma)
    // The weights of two neighbouring directions arrive as one aligned SGPR pair straight from s_load; op_sel picks the half.
    // The products go through v[124:127], named as clobbers instead of compiler-chosen outputs: with outputs the register
    // allocator reuses them for the address temporaries of the next asm statement, and the hazard recogniser then puts an
    // s_nop between the two statements at every step (an issue slot like any other instruction: 9 % of the kernel).
    if constexpr (HALF == 0) {
        asm volatile("v_pk_fma_f32 v[124:125], %[h], %[dl], %[ql] op_sel_hi:[0,1,1]\n\t"
                     "v_pk_fma_f32 v[126:127], %[h], %[dh], %[qh] op_sel_hi:[0,1,1]\n\t"
                     "v_pk_add_f32 %[a0], %[a0], v[124:125]\n\t"
                     "v_pk_add_f32 %[a1], %[a1], v[126:127]"
                     : [a0] "+v"(ac[0]), [a1] "+v"(ac[1])
                     : [h] "s"(hp), [dl] "v"(d.lo), [dh] "v"(d.hi), [ql] "v"(q.lo), [qh] "v"(q.hi)
                     : "v124", "v125", "v126", "v127");
    } else {
        asm volatile("v_pk_fma_f32 v[124:125], %[h], %[dl], %[ql] op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
                     "v_pk_fma_f32 v[126:127], %[h], %[dh], %[qh] op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
                     "v_pk_add_f32 %[a0], %[a0], v[124:125]\n\t"
                     "v_pk_add_f32 %[a1], %[a1], v[126:127]"
                     : [a0] "+v"(ac[0]), [a1] "+v"(ac[1])
                     : [h] "s"(hp), [dl] "v"(d.lo), [dh] "v"(d.hi), [ql] "v"(q.lo), [qh] "v"(q.hi)
                     : "v124", "v125", "v126", "v127");
    }
}

// Four direction steps (4 H .. 4 H + 3) of the one-segment sweep as ONE asm statement: per step the offset test with its
// out-of-line re-read (as reload_quads) and the step's arithmetic (as add_quad / lerp_quad).  Between two asm statements
// that pass registers to each other the compiler inserts an s_nop (it cannot see that the registers the first one "wrote"
// are ready): an issue slot per step when every step is its own pair of statements, 9 % of the kernel's instructions.
// lerp: the products go through v[124:127], named as clobbers (no operands left to spend, and no compiler-chosen
// temporaries that it could share with the next statement's address registers).
#define BF_Q_CHECK(j, ep, ec) "s_cmp_lg_u32 %[" #ec "], %[" #ep "]\n\ts_cbranch_scc1 .Lr" #j "_%=\n.Lb" #j "_%=:\n\t"
#define BF_Q_PAD(j) "v_pk_add_f32 %[a" #j "0], %[a" #j "0], %[sl]\n\tv_pk_add_f32 %[a" #j "1], %[a" #j "1], %[sh]\n\t"
#define BF_Q_LERP(j, h, mods)                                                    \
    "v_pk_fma_f32 v[124:125], %[" #h "], %[dl], %[sl] " mods "\n\t"              \
    "v_pk_fma_f32 v[126:127], %[" #h "], %[dh], %[sh] " mods "\n\t"              \
    "v_pk_add_f32 %[a" #j "0], %[a" #j "0], v[124:125]\n\t"                      \
    "v_pk_add_f32 %[a" #j "1], %[a" #j "1], v[126:127]\n\t"
#define BF_Q_PAD_RELOAD(j, ec)                                                   \
    ".Lr" #j "_%=:\n\tv_add_u32 %[ad], %[" #ec "], %[lb]\n\t"                    \
    "ds_read_b64 %[sl], %[ad] offset:0\n\tds_read_b64 %[sh], %[ad] offset:8\n\t" \
    "s_waitcnt lgkmcnt(0)\n\ts_branch .Lb" #j "_%=\n"
#define BF_Q_LERP_RELOAD(j, ec)                                                  \
    ".Lr" #j "_%=:\n\tv_add_u32 %[ad], %[" #ec "], %[lb]\n\tv_add_u32 %[ad2], %[ad], %[doff]\n\t" \
    "ds_read_b64 %[sl], %[ad] offset:0\n\tds_read_b64 %[sh], %[ad] offset:8\n\t" \
    "ds_read_b64 %[dl], %[ad2] offset:0\n\tds_read_b64 %[dh], %[ad2] offset:8\n\t" \
    "s_waitcnt lgkmcnt(0)\n\ts_branch .Lb" #j "_%=\n"
// (compile-time row stride: the difference copies sit at a fixed byte distance -- an immediate offset, no second address)
#define BF_Q_LERP_RELOAD_I(j, ec)                                                \
    ".Lr" #j "_%=:\n\tv_add_u32 %[ad], %[" #ec "], %[lb]\n\t"                    \
    "ds_read_b64 %[sl], %[ad] offset:0\n\tds_read_b64 %[sh], %[ad] offset:8\n\t" \
    "ds_read_b64 %[dl], %[ad] offset:%[o0]\n\tds_read_b64 %[dh], %[ad] offset:%[o1]\n\t" \
    "s_waitcnt lgkmcnt(0)\n\ts_branch .Lb" #j "_%=\n"
#define BF_Q_EVEN "op_sel_hi:[0,1,1]"
#define BF_Q_ODD "op_sel:[1,0,0] op_sel_hi:[1,1,1]"
#define BF_Q_ACC(j, i) [a##j##0] "+v"(acc[i][0][0]), [a##j##1] "+v"(acc[i][0][1])
template <bool LERP, int H, int DOFF>   // DOFF > 0: byte distance of the difference copies as a compile-time constant
__device__ __forceinline__ void steps4(f32x2 (&acc)[8][1][2], Quad& S, Quad& D, int ep, int e0, int e1, int e2, int e3,
                                       unsigned long long hlo, unsigned long long hhi, int lbase, int d_off)
{
    // e0..e3: LDS offsets of directions 4 H .. 4 H + 3; ep: the offset the quads hold on entry (H = 0: e0 itself -- the
    // mic's first quads were requested a mic ahead, so its first step has no test)
    int ad, ad2;
    constexpr int B = 4 * H;
    if constexpr (!LERP && H == 0) {
        asm volatile(BF_Q_PAD(0) BF_Q_CHECK(1, e0, e1) BF_Q_PAD(1) BF_Q_CHECK(2, e1, e2) BF_Q_PAD(2) BF_Q_CHECK(3, e2, e3) BF_Q_PAD(3)
                     ".subsection 1\n" BF_Q_PAD_RELOAD(1, e1) BF_Q_PAD_RELOAD(2, e2) BF_Q_PAD_RELOAD(3, e3) "\t.subsection 0"
                     : BF_Q_ACC(0, B), BF_Q_ACC(1, B + 1), BF_Q_ACC(2, B + 2), BF_Q_ACC(3, B + 3), [sl] "+v"(S.lo), [sh] "+v"(S.hi), [ad] "=&v"(ad)
                     : [e0] "s"(e0), [e1] "s"(e1), [e2] "s"(e2), [e3] "s"(e3), [lb] "v"(lbase) : "scc");
    } else if constexpr (!LERP) {
        asm volatile(BF_Q_CHECK(0, ep, e0) BF_Q_PAD(0) BF_Q_CHECK(1, e0, e1) BF_Q_PAD(1) BF_Q_CHECK(2, e1, e2) BF_Q_PAD(2) BF_Q_CHECK(3, e2, e3) BF_Q_PAD(3)
                     ".subsection 1\n" BF_Q_PAD_RELOAD(0, e0) BF_Q_PAD_RELOAD(1, e1) BF_Q_PAD_RELOAD(2, e2) BF_Q_PAD_RELOAD(3, e3) "\t.subsection 0"
                     : BF_Q_ACC(0, B), BF_Q_ACC(1, B + 1), BF_Q_ACC(2, B + 2), BF_Q_ACC(3, B + 3), [sl] "+v"(S.lo), [sh] "+v"(S.hi), [ad] "=&v"(ad)
                     : [ep] "s"(ep), [e0] "s"(e0), [e1] "s"(e1), [e2] "s"(e2), [e3] "s"(e3), [lb] "v"(lbase) : "scc");
    } else if constexpr (H == 0 && DOFF == 0) {
        asm volatile(BF_Q_LERP(0, h0, BF_Q_EVEN) BF_Q_CHECK(1, e0, e1) BF_Q_LERP(1, h0, BF_Q_ODD) BF_Q_CHECK(2, e1, e2) BF_Q_LERP(2, h1, BF_Q_EVEN)
                         BF_Q_CHECK(3, e2, e3) BF_Q_LERP(3, h1, BF_Q_ODD)
                     ".subsection 1\n" BF_Q_LERP_RELOAD(1, e1) BF_Q_LERP_RELOAD(2, e2) BF_Q_LERP_RELOAD(3, e3) "\t.subsection 0"
                     : BF_Q_ACC(0, B), BF_Q_ACC(1, B + 1), BF_Q_ACC(2, B + 2), BF_Q_ACC(3, B + 3), [sl] "+v"(S.lo), [sh] "+v"(S.hi), [dl] "+v"(D.lo),
                       [dh] "+v"(D.hi), [ad] "=&v"(ad), [ad2] "=&v"(ad2)
                     : [e0] "s"(e0), [e1] "s"(e1), [e2] "s"(e2), [e3] "s"(e3), [h0] "s"(hlo), [h1] "s"(hhi), [lb] "v"(lbase), [doff] "s"(d_off)
                     : "scc", "v124", "v125", "v126", "v127");
    } else if constexpr (H == 0) {
        asm volatile(BF_Q_LERP(0, h0, BF_Q_EVEN) BF_Q_CHECK(1, e0, e1) BF_Q_LERP(1, h0, BF_Q_ODD) BF_Q_CHECK(2, e1, e2) BF_Q_LERP(2, h1, BF_Q_EVEN)
                         BF_Q_CHECK(3, e2, e3) BF_Q_LERP(3, h1, BF_Q_ODD)
                     ".subsection 1\n" BF_Q_LERP_RELOAD_I(1, e1) BF_Q_LERP_RELOAD_I(2, e2) BF_Q_LERP_RELOAD_I(3, e3) "\t.subsection 0"
                     : BF_Q_ACC(0, B), BF_Q_ACC(1, B + 1), BF_Q_ACC(2, B + 2), BF_Q_ACC(3, B + 3), [sl] "+v"(S.lo), [sh] "+v"(S.hi), [dl] "+v"(D.lo),
                       [dh] "+v"(D.hi), [ad] "=&v"(ad)
                     : [e0] "s"(e0), [e1] "s"(e1), [e2] "s"(e2), [e3] "s"(e3), [h0] "s"(hlo), [h1] "s"(hhi), [lb] "v"(lbase), [o0] "n"(DOFF), [o1] "n"(DOFF + 8)
                     : "scc", "v124", "v125", "v126", "v127");
    } else if constexpr (DOFF == 0) {
        asm volatile(BF_Q_CHECK(0, ep, e0) BF_Q_LERP(0, h0, BF_Q_EVEN) BF_Q_CHECK(1, e0, e1) BF_Q_LERP(1, h0, BF_Q_ODD) BF_Q_CHECK(2, e1, e2)
                         BF_Q_LERP(2, h1, BF_Q_EVEN) BF_Q_CHECK(3, e2, e3) BF_Q_LERP(3, h1, BF_Q_ODD)
                     ".subsection 1\n" BF_Q_LERP_RELOAD(0, e0) BF_Q_LERP_RELOAD(1, e1) BF_Q_LERP_RELOAD(2, e2) BF_Q_LERP_RELOAD(3, e3) "\t.subsection 0"
                     : BF_Q_ACC(0, B), BF_Q_ACC(1, B + 1), BF_Q_ACC(2, B + 2), BF_Q_ACC(3, B + 3), [sl] "+v"(S.lo), [sh] "+v"(S.hi), [dl] "+v"(D.lo),
                       [dh] "+v"(D.hi), [ad] "=&v"(ad), [ad2] "=&v"(ad2)
                     : [ep] "s"(ep), [e0] "s"(e0), [e1] "s"(e1), [e2] "s"(e2), [e3] "s"(e3), [h0] "s"(hlo), [h1] "s"(hhi), [lb] "v"(lbase),
                       [doff] "s"(d_off)
                     : "scc", "v124", "v125", "v126", "v127");
    } else {
        asm volatile(BF_Q_CHECK(0, ep, e0) BF_Q_LERP(0, h0, BF_Q_EVEN) BF_Q_CHECK(1, e0, e1) BF_Q_LERP(1, h0, BF_Q_ODD) BF_Q_CHECK(2, e1, e2)
                         BF_Q_LERP(2, h1, BF_Q_EVEN) BF_Q_CHECK(3, e2, e3) BF_Q_LERP(3, h1, BF_Q_ODD)
                     ".subsection 1\n" BF_Q_LERP_RELOAD_I(0, e0) BF_Q_LERP_RELOAD_I(1, e1) BF_Q_LERP_RELOAD_I(2, e2) BF_Q_LERP_RELOAD_I(3, e3) "\t.subsection 0"
                     : BF_Q_ACC(0, B), BF_Q_ACC(1, B + 1), BF_Q_ACC(2, B + 2), BF_Q_ACC(3, B + 3), [sl] "+v"(S.lo), [sh] "+v"(S.hi), [dl] "+v"(D.lo),
                       [dh] "+v"(D.hi), [ad] "=&v"(ad)
                     : [ep] "s"(ep), [e0] "s"(e0), [e1] "s"(e1), [e2] "s"(e2), [e3] "s"(e3), [h0] "s"(hlo), [h1] "s"(hhi), [lb] "v"(lbase),
                       [o0] "n"(DOFF), [o1] "n"(DOFF + 8)
                     : "scc", "v124", "v125", "v126", "v127");
    }
}

// One workgroup (16 waves, one per CU: it owns the LDS) walks a tile of directions in groups of kGroup; for each group
// the frame's mics pass through LDS in chunks (each mic staged as 4 shifted copies, lerp also 4 copies of the first
// difference), every wave accumulating its directions in registers across the chunks.  A chunk is consumed between
// two workgroup barriers and the next chunk's global loads are in flight meanwhile.
//
// Table entries never touch the VALU: the `taps` slot carries the int32 digest [D][M] (digest_kernel: LDS byte offset
// per (direction, mic)), fetched with s_load_dwordx16 like the lerp weights in `frac`.  Per (direction, mic, segment)
// the wave issues  pad: 1 ds_read_b128 + 2 v_pk_add_f32;  lerp: 2 ds_read_b128 + 2 v_pk_fma_f32 + 2 v_pk_add_f32, plus
// one v_add per (direction, mic) for the address (two for lerp with a run-time row stride).
// DIRECT: direction-outer inner loop with a [D][M] digest, for pad / lerp tables without structure (see `directions`).
template <int ALGO, int NSEG, int RS, int W, bool DIRECT = false>
__global__ void __launch_bounds__(W * 64, 4) das_copies_kernel(BF_TABLE_PARAMS, KArgs a)
{
    constexpr bool FIR = ALGO == ALGO_HYBRID || ALGO == ALGO_FIR_NAIVE || ALGO == ALGO_FIR_VEC;   // 8 taps, N <= 256
    static_assert(!FIR || NSEG == 1, "the FIR flavours use the one-segment geometry");
    static_assert(RS == 0 || RS == (FIR ? Geo<NSEG>::kRsFir : Geo<NSEG>::kRs), "fixed row stride");
    constexpr int A = (ALGO == ALGO_LERP) ? 2 : 1;   // arrays per mic: s (and D)
    constexpr int C = copies_of(ALGO, DIRECT);        // shifted copies per array
    constexpr int DW = Geo<NSEG>::kDw, kGroup = DW * W, kPark = Geo<NSEG>::kPark;   // kGroup: directions per workgroup pass
    // (mic, segment) pairs a wave stages per chunk: 16 per workgroup, 32 where the two-copy rows leave room for them
    // (pad with several segments; the one-segment sweep with 16 waves: 32-mic chunks, half the barriers)
    constexpr int SP = (((NSEG > 1 && ALGO == ALGO_PAD) || (NSEG == 1 && W == 16 && !FIR && !DIRECT)) ? 32 : 16) / W;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    int tile, frame;
    tile_and_frame(a, &tile, &frame);
    const int tile_begin = a.dir_begin + tile * a.tile_dirs;
    if (tile_begin >= a.dir_end) return;
    const int tile_end = min(tile_begin + a.tile_dirs, a.dir_end);

    const int rs = RS > 0 ? RS : a.row_stride, lead = RS > 0 ? Geo<NSEG>::kLead : a.lead;   // (kRsFir has the same lead, plus the tail)
    const int mc = a.mic_chunk, M = a.n_mics, N = a.n_samples;
    const float* __restrict__ frame_sig = signals + (size_t)frame * a.m_total * N;
    float* __restrict__ img = images + (size_t)frame * a.image_stride;
    // the digest rides in a pointer slot the algorithm does not use: taps (pad, lerp) or frac (hybrid)
    const int32_t* __restrict__ dig = reinterpret_cast<const int32_t*>(FIR ? frac : taps);
    const int slot_floats = A * C * rs;   // floats per staged mic

    // Rows of `signals` this wave stages: pair index pr = wave + W i  <->  chunk mic pr / NSEG, segment pr % NSEG.
    // The mic ids of the first 64 chunks are loaded once (lane c holds chunk c's) so that the per-chunk prefetch is a
    // single independent load, not a load that waits for an index load.
    int vmic[SP];
#pragma unroll
    for (int i = 0; i < SP; ++i) {
        const int cm = (wave + W * i) / NSEG;
        vmic[i] = 0;
        if (lane < a.n_chunks && cm < mc && lane * mc + cm < M) vmic[i] = mics[lane * mc + cm];
    }
    auto fetch = [&](int ch, int mcc, Staged (&st)[SP]) {
#pragma unroll
        for (int i = 0; i < SP; ++i) {
            const int pr = wave + W * i, cm = pr / NSEG, seg = pr % NSEG;
            st[i].v = make_float4(0.f, 0.f, 0.f, 0.f);
            st[i].edge = 0.0f;
            if (cm < mcc) {
                const int mic = (ch < kWave) ? __builtin_amdgcn_readlane(vmic[i], ch) : mics[ch * mc + cm];
                const float* src = frame_sig + (size_t)mic * N;
                const int k = 4 * (64 * seg + lane);
                if ((N & 3) == 0) {
                    if (k < N) st[i].v = *reinterpret_cast<const float4*>(src + k);
                } else {
                    if (k < N) st[i].v.x = src[k];
                    if (k + 1 < N) st[i].v.y = src[k + 1];
                    if (k + 2 < N) st[i].v.z = src[k + 2];
                    if (k + 3 < N) st[i].v.w = src[k + 3];
                }
                if constexpr (NSEG > 1) {
                    // what DPP cannot reach: the three samples before the segment and the one after it
                    const int ke = lane < 3 ? 256 * seg - 3 + lane : 256 * seg + 256;
                    if ((lane < 3 || lane == 63) && ke >= 0 && ke < N) st[i].edge = src[ke];
                }
            }
        }
    };
    auto stage = [&](int mcc, const Staged (&st)[SP]) {
#pragma unroll
        for (int i = 0; i < SP; ++i) {
            const int pr = wave + W * i, cm = pr / NSEG, seg = pr % NSEG;
            if (cm >= mcc) continue;
            float* row0 = lds + cm * slot_floats;
            const int col = lead + 256 * seg;
            const float4 v = st[i].v;
            float py = dpp_prev(v.y), pz = dpp_prev(v.z), pw = dpp_prev(v.w), nx = dpp_next(v.x);
            float ey = 0.0f, ez = 0.0f, ew = 0.0f;
            if constexpr (NSEG > 1) {
                ey = lane_value(st[i].edge, 0); ez = lane_value(st[i].edge, 1); ew = lane_value(st[i].edge, 2);
                const float en = lane_value(st[i].edge, 63);
                if (lane == 0) { py = ey; pz = ez; pw = ew; }
                if (lane == 63) nx = en;
            }
            write_copies<C>(row0, rs, col, lane, v, py, pz, pw);
            if constexpr (FIR) {
                // the zero padding after the block (convolve_and_sum.c:199-203): quad 64 of copy c still holds the last c
                // samples, quad 65 is zero
                if (lane == kWave - 1) {
                    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
                    float4* t0 = reinterpret_cast<float4*>(row0 + col + 256);
                    t0[0] = z; t0[1] = z;
                    float4* t1 = reinterpret_cast<float4*>(row0 + rs + col + 256);
                    t1[0] = make_float4(v.w, 0.f, 0.f, 0.f); t1[1] = z;
                    float4* t2 = reinterpret_cast<float4*>(row0 + 2 * rs + col + 256);
                    t2[0] = make_float4(v.z, v.w, 0.f, 0.f); t2[1] = z;
                    float4* t3 = reinterpret_cast<float4*>(row0 + 3 * rs + col + 256);
                    t3[0] = make_float4(v.y, v.z, v.w, 0.f); t3[1] = z;
                }
            }
            if constexpr (ALGO == ALGO_LERP) {
                // D[i] = s[i+1] - s[i], the reference's own subtraction (lerp_and_sum.c:54); D[-1] stays 0 (prefix)
                const float4 dq = make_float4(v.y - v.x, v.z - v.y, v.w - v.z, nx - v.w);
                float dy = dpp_prev(dq.y), dz = dpp_prev(dq.z), dw = dpp_prev(dq.w);
                if constexpr (NSEG > 1) {
                    if (lane == 0 && seg > 0) { dy = ez - ey; dz = ew - ez; dw = v.x - ew; }
                }
                write_copies<C>(row0 + C * rs, rs, col, lane, dq, dy, dz, dw);
            }
            if (seg == 0) {   // the zero prefix (also wiped by the parked rows of the previous group); lead can exceed 256
                const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
                for (int q = lane; q < (lead >> 2); q += kWave) {
#pragma unroll
                    for (int c = 0; c < C * A; ++c) reinterpret_cast<float4*>(row0 + c * rs)[q] = z;
                }
            }
        }
    };

    Staged staged[SP];
    fetch(0, min(mc, M), staged);
    const char* lbase = reinterpret_cast<const char*>(lds) + 16 * lane;

    for (int g0 = tile_begin; g0 < tile_end; g0 += kGroup) {
        // v_pk_add_f32 / v_pk_fma_f32 keep the instruction count down (IEEE-identical to the scalar forms); the
        // accumulators are float2 register pairs matching the (x,y)/(z,w) halves of a quad read.
        f32x2 acc[DW][NSEG][2];
#pragma unroll
        for (int j = 0; j < DW; ++j)
#pragma unroll
            for (int sg = 0; sg < NSEG; ++sg) { acc[j][sg][0] = f32x2{0.0f, 0.0f}; acc[j][sg][1] = f32x2{0.0f, 0.0f}; }

        for (int ch = 0; ch < a.n_chunks; ++ch) {
            const int m0 = ch * mc;
            const int mcc = min(mc, M - m0);
            __syncthreads();   // every wave is done with the previous contents (chunk reads or parked rows)
            stage(mcc, staged);
            __syncthreads();
            {   // request the next chunk (or the next group's first) while this one is consumed
                int ng0 = g0, nch = ch + 1;
                if (nch == a.n_chunks) { nch = 0; ng0 = g0 + kGroup; }
                if (ng0 < tile_end) fetch(nch, min(mc, M - nch * mc), staged);
            }
            // 8-tap FIR flavours: per (direction, mic) the lane's 4 outputs need 11 consecutive samples = 3 aligned quads
            // of one copy (immediate offsets 0 / 16 / 32).  The 8 taps of a (direction, mic) are fetched with two broadcast
            // vector loads (every lane the same address): scalar loads would share the LDS wait counter and stall every
            // step, vector loads have their own (vmcnt) and are prefetched four mics ahead.  LDS offsets and delays of the
            // 16 mics come by s_load once per direction.
            auto fir_directions = [&](auto mcc_c) {
                constexpr int MCC = decltype(mcc_c)::value;     // > 0: compile-time mic count, software-pipelined; 0: any count
                int vz = 0;
                asm volatile("" : "+v"(vz));                    // an opaque zero: keeps the tap loads on the vector memory path
#pragma unroll
                for (int j = 0; j < DW; ++j) {
                    const int d = g0 + wave * DW + j;           // wave-uniform
                    if (d >= tile_end) continue;
                    const size_t idx = (size_t)d * M + m0;
                    const float4* __restrict__ tp = reinterpret_cast<const float4*>(taps + idx * 8) + vz;
                    int e[16], pw[16];
#pragma unroll
                    for (int m = 0; m < 16; ++m) {
                        if constexpr (ALGO == ALGO_HYBRID) {
                            e[m] = dig[idx + m];
                            pw[m] = whole[idx + m];
                        } else {
                            // convolve_and_sum.c:197-211: no whole-sample shift, window starts T/2 = 4 samples back: copy 0
                            e[m] = ((m * 4) * rs + lead - 4) * 4;
                            pw[m] = 0;
                        }
                    }
                    float o[4] = {acc[j][0][0].x, acc[j][0][0].y, acc[j][0][1].x, acc[j][0][1].y};
                    auto consume = [&](const float4 (&hq)[2], const float4 (&q)[3], int pp) {
                        const float h[8] = {hq[0].x, hq[0].y, hq[0].z, hq[0].w, hq[1].x, hq[1].y, hq[1].z, hq[1].w};
                        const float Wn[12] = {q[0].x, q[0].y, q[0].z, q[0].w, q[1].x, q[1].y, q[1].z, q[1].w, q[2].x, q[2].y, q[2].z, q[2].w};
                        if constexpr (ALGO == ALGO_HYBRID) {
                            // output j (sample 4 lane + j) is live for 4 lane + j > p: the first n_j = (p + 4 - j) >> 2 lanes are not
                            auto mask = [](int n) -> unsigned long long { return n >= 64 ? 0ull : (~0ull << n); };
                            fir8_masked(o, Wn, h, mask((pp + 4) >> 2), mask((pp + 3) >> 2), mask((pp + 2) >> 2), mask((pp + 1) >> 2));
                        } else if constexpr (ALGO == ALGO_FIR_NAIVE) {
                            // convolve_and_sum.c:197-211  out[i] += h[t] * padded[i + t], t in order (fma chain into out)
#pragma unroll
                            for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                                for (int t = 0; t < 8; ++t) o[jj] = __fmaf_rn(h[t], Wn[jj + t], o[jj]);
                        } else {
                            // convolve_and_sum.c:158-192 + sum8 :132-153 with one AVX block: eight plain products, fixed tree, out +=
#pragma unroll
                            for (int jj = 0; jj < 4; ++jj) {
                                const float* x = Wn + jj;
                                const float q0 = x[0] * h[0] + x[4] * h[4], q1 = x[1] * h[1] + x[5] * h[5];
                                const float q2 = x[2] * h[2] + x[6] * h[6], q3 = x[3] * h[3] + x[7] * h[7];
                                o[jj] += (q0 + q2) + (q1 + q3);
                            }
                        }
                    };
                    if constexpr (MCC > 0) {
                        constexpr int RH = 4, RQ = 2;           // mics of taps / of sample quads in flight
                        float4 H[RH][2], Q[RQ][3];
                        auto load_taps = [&](int i, int slot) { H[slot][0] = tp[2 * i]; H[slot][1] = tp[2 * i + 1]; };
                        auto load_quads = [&](int i, int slot) {
                            const char* sp = lbase + e[i];
                            Q[slot][0] = *reinterpret_cast<const float4*>(sp);
                            Q[slot][1] = *reinterpret_cast<const float4*>(sp + 16);
                            Q[slot][2] = *reinterpret_cast<const float4*>(sp + 32);
                        };
#pragma unroll
                        for (int i = 0; i < RH && i < MCC; ++i) load_taps(i, i);
#pragma unroll
                        for (int i = 0; i < RQ && i < MCC; ++i) load_quads(i, i);
#pragma unroll
                        for (int i = 0; i < MCC; ++i) {
                            consume(H[i % RH], Q[i % RQ], pw[i]);
                            if (i + RH < MCC) load_taps(i + RH, i % RH);
                            if (i + RQ < MCC) load_quads(i + RQ, i % RQ);
                            asm volatile("" ::: "memory");      // keeps the prefetch distance: later loads stay below this point
                        }
                    } else {
                        for (int i = 0; i < mcc; ++i) {
                            // (runtime index into e / pw: pick through a uniform select chain the compiler keeps in SGPRs)
                            int ei = e[0], pi = pw[0];
#pragma unroll
                            for (int m = 1; m < 16; ++m) { ei = (i == m) ? e[m] : ei; pi = (i == m) ? pw[m] : pi; }
                            float4 hq[2] = {tp[2 * i], tp[2 * i + 1]};
                            const char* sp = lbase + ei;
                            float4 q[3] = {*reinterpret_cast<const float4*>(sp), *reinterpret_cast<const float4*>(sp + 16),
                                           *reinterpret_cast<const float4*>(sp + 32)};
                            consume(hq, q, pi);
                        }
                    }
                    acc[j][0][0] = f32x2{o[0], o[1]};
                    acc[j][0][1] = f32x2{o[2], o[3]};
                }
            };
            // DIRECT variant (tables without structure: neighbouring directions do not share delays, so the sweep below would
            // reload -- and wait -- at every step): one direction at a time, its staged mics in order, U mics x NSEG segments
            // of reads kept in flight.  Digest [D][M] and the lerp weights straight from `frac`, 16 entries per s_load.
            auto directions = [&](auto mcc_c) {
                // MCC > 0: the chunk's mic count at compile time (a whole number of batches): straight-line code, no
                // per-mic conditionals.  MCC == 0: any count, one uniform branch per batch.
                constexpr int MCC = decltype(mcc_c)::value;
                constexpr bool FULL = MCC > 0;
                constexpr int U = Geo<NSEG>::kBatch;
                static_assert(MCC % U == 0, "whole batches");
#pragma unroll
                for (int j = 0; j < DW; ++j) {
                    const int d = g0 + wave * DW + j;           // wave-uniform
                    if (d >= tile_end) continue;
                    const size_t idx = (size_t)d * M + m0;
                    // 16 entries unconditionally (the tables carry 64 bytes of slack) so the loads merge into wide s_loads
                    int e[16];
                    float hh[16];
#pragma unroll
                    for (int m = 0; m < 16; ++m) {
                        e[m] = dig[idx + m];
                        hh[m] = 0.0f;
                        if constexpr (ALGO == ALGO_LERP) hh[m] = frac[idx + m];
                    }
                    auto consume = [&](const float4& Sq, const float4& Dv, float h, f32x2 (&ac)[2]) {
                        const f32x2 S01{Sq.x, Sq.y}, S23{Sq.z, Sq.w};
                        if constexpr (ALGO == ALGO_PAD) {
                            // pad_and_sum.c:41-47   out[k] += s[k - p]
                            ac[0] += S01; ac[1] += S23;
                        } else {
                            // lerp_and_sum.c:50-56  out[k] += s[i] + h * (s[i+1] - s[i]),  i = k - p - 1
                            const f32x2 h2{h, h}, D01{Dv.x, Dv.y}, D23{Dv.z, Dv.w};
                            ac[0] += __builtin_elementwise_fma(h2, D01, S01);
                            ac[1] += __builtin_elementwise_fma(h2, D23, S23);
                        }
                    };
                    if constexpr (FULL) {
                        // software pipeline over the (mic, segment) units of this direction: R units' reads in flight,
                        // each consumed unit's registers are refilled at once
                        constexpr int R = 4, UNITS = MCC * NSEG;
                        float4 S[R], Dq[R];
                        auto issue = [&](int i, int slot) {
                            const char* sp = lbase + e[i / NSEG] + 1024 * (i % NSEG);
                            S[slot] = *reinterpret_cast<const float4*>(sp);
                            if constexpr (ALGO == ALGO_LERP) Dq[slot] = *reinterpret_cast<const float4*>(sp + 4 * C * rs);   // D copies: C rows on
                        };
#pragma unroll
                        for (int i = 0; i < R; ++i) issue(i, i);
#pragma unroll
                        for (int i = 0; i < UNITS; ++i) {
                            consume(S[i % R], Dq[i % R], hh[i / NSEG], acc[j][i % NSEG]);
                            if (i + R < UNITS) issue(i + R, i % R);
                        }
                    } else {
#pragma unroll
                        for (int mb = 0; mb < 16; mb += U) {
                            if (mb >= mcc) break;
                            float4 S[U][NSEG], Dq[U][NSEG];
#pragma unroll
                            for (int u = 0; u < U; ++u) {
                                // a partial batch re-reads mic 0's row for the missing mics and drops the result
                                const int eo = (mb + u < mcc) ? e[mb + u] : e[0];
                                const char* sp = lbase + eo;
#pragma unroll
                                for (int sg = 0; sg < NSEG; ++sg) {
                                    S[u][sg] = *reinterpret_cast<const float4*>(sp + 1024 * sg);
                                    if constexpr (ALGO == ALGO_LERP) Dq[u][sg] = *reinterpret_cast<const float4*>(sp + 4 * C * rs + 1024 * sg);
                                }
                            }
#pragma unroll
                            for (int u = 0; u < U; ++u) {
                                if (mb + u >= mcc) break;
#pragma unroll
                                for (int sg = 0; sg < NSEG; ++sg) consume(S[u][sg], Dq[u][sg], hh[mb + u], acc[j][sg]);
                            }
                        }
                    }
                }
            };
            // pad / lerp: mic-outer sweep over the wave's DW directions.  Neighbouring directions mostly share a mic's
            // whole-sample delay (the delay changes by a fraction of a sample per grid step), and the LDS read depends on
            // nothing else: the quad (and its difference quad) is re-read only when the offset differs from the previous
            // direction's -- a wave-uniform test on two SGPRs.  On the reference's geometries that is 1.1 - 2.2 reads per
            // 8 directions, which takes the LDS pipe out of the picture and leaves the adds.  Two mics advance together
            // so that their reads overlap.
            auto sweep = [&]() {
                const int dw0 = g0 + wave * DW;                 // wave-uniform
                if (dw0 >= tile_end) return;                    // this wave has no directions in the tile
                const size_t grp = (size_t)(dw0 - a.dir_begin) / DW;
                const int32_t* __restrict__ eg = dig + (grp * M + m0) * DW;
                const float* __restrict__ hg = reinterpret_cast<const float*>(dig) + a.digest_h_off + (grp * M + m0) * DW;
                // LDS byte address of this lane's quad column (the asm reads need the raw 32-bit LDS address)
                const int lb = 16 * lane + (int)(unsigned)(size_t)((__attribute__((address_space(3))) char*)lds);
                const int d_off = 4 * C * rs;                   // D copies sit C rows after the s copies
                constexpr bool kLerp = ALGO == ALGO_LERP;
                constexpr int DOFF = (kLerp && NSEG == 1) ? 4 * C * RS : 0;   // ... a compile-time distance with the fixed row stride
                struct Entries { int e[DW]; unsigned long long hp[DW / 2]; };   // offsets; lerp weights as (even, odd) direction pairs
                // (et, ht): the table rows of the current six-mic trip's first mic; `m` counts from there, so that every request
                // inside a trip is an immediate offset off the same two pointers
                const int32_t* __restrict__ et = eg;
                const float* __restrict__ ht = hg;
                auto request = [&](Entries& t, int m) {
                    // (reads past the chunk's last mic stay inside the slack-padded table and are dropped)
#pragma unroll
                    for (int j = 0; j < DW; ++j) t.e[j] = et[m * DW + j];
#pragma unroll
                    for (int j = 0; j < DW / 2; ++j) {
                        t.hp[j] = 0;
                        if constexpr (kLerp) t.hp[j] = *reinterpret_cast<const unsigned long long*>(ht + m * DW + 2 * j);   // 8-byte aligned: DW is even
                    }
                };
                // A three-stage pipeline over the mics, so that no wave ever sits on an LDS or scalar-load round trip:
                //   mic m     its quads (requested one mic ago) are awaited, then consumed by the DW direction steps;
                //   mic m + 1 its table entries (requested two mics ago) are used to request its first quads;
                //   mic m + 2 its table entries are requested.
                // Two quad sets alternate by mic parity, three entry sets rotate: six mics per loop trip, no register copies.
                Entries E[3];
                // The cross-mic quad prefetch is used for one segment only: with more segments the 64 accumulators make the
                // compiler spill, and a register with a read in flight must not be moved behind the asm statements' back.
                constexpr bool kPipe = NSEG == 1;
                constexpr int NQ = kPipe ? 2 : 1;              // quad sets: one per mic parity with the prefetch, else one
                Quad S[NQ][NSEG], Dq[NQ][NSEG];
#pragma unroll
                for (int q = 0; q < NQ; ++q)
#pragma unroll
                    for (int sg = 0; sg < NSEG; ++sg) S[q][sg].lo = S[q][sg].hi = Dq[q][sg].lo = Dq[q][sg].hi = f32x2{0.0f, 0.0f};
                request(E[0], 0);
                request(E[1], 1);
                auto issue = [&](Quad (&sq)[NSEG], Quad (&dq)[NSEG], int e) {
                    if constexpr (DOFF > 0) issue_quads_i<DOFF>(sq, dq, e, lb);
                    else issue_quads<NSEG, kLerp>(sq, dq, e, lb, d_off);
                };
                auto mic = [&](int m, auto pc, auto kc) {
                    constexpr int P = kPipe ? decltype(pc)::value : 0, K = decltype(kc)::value, K1 = (K + 1) % 3, K2 = (K + 2) % 3;
                    const Entries& cur = E[K];
                    if constexpr (kPipe) {
                        await_quads<NSEG>(S[P], Dq[P]);
                        issue(S[P ^ 1], Dq[P ^ 1], E[K1].e[0]);
                        request(E[K2], m + 2);
                    } else {
                        request(E[K2], m + 2);
                        reload_quads<NSEG, kLerp>(S[P], Dq[P], cur.e[0], -1, lb, d_off);   // offsets are >= 0: -1 always loads (and waits)
                    }
                    if constexpr (NSEG == 1) {
                        static_assert(DW == 8, "two statements of four direction steps");
                        steps4<kLerp, 0, DOFF>(acc, S[P][0], Dq[P][0], cur.e[0], cur.e[0], cur.e[1], cur.e[2], cur.e[3], cur.hp[0], cur.hp[1], lb, d_off);
                        steps4<kLerp, 1, DOFF>(acc, S[P][0], Dq[P][0], cur.e[3], cur.e[4], cur.e[5], cur.e[6], cur.e[7], cur.hp[2], cur.hp[3], lb, d_off);
                    } else {
                        auto stepj = [&](auto jc) {
                            constexpr int j = decltype(jc)::value;
                            if constexpr (j > 0) reload_quads<NSEG, kLerp>(S[P], Dq[P], cur.e[j], j == 1 ? cur.e[0] : cur.e[j - 1], lb, d_off);
#pragma unroll
                            for (int sg = 0; sg < NSEG; ++sg) {
                                if constexpr (ALGO == ALGO_PAD) add_quad(acc[j][sg], S[P][sg]);
                                else lerp_quad<j & 1>(acc[j][sg], S[P][sg], Dq[P][sg], cur.hp[j / 2]);
                            }
                        };
                        stepj(std::integral_constant<int, 0>{});
                        stepj(std::integral_constant<int, 1>{});
                        stepj(std::integral_constant<int, 2>{});
                        stepj(std::integral_constant<int, 3>{});
                        if constexpr (DW == 8) {
                            stepj(std::integral_constant<int, 4>{});
                            stepj(std::integral_constant<int, 5>{});
                            stepj(std::integral_constant<int, 6>{});
                            stepj(std::integral_constant<int, 7>{});
                        }
                    }
                };
                using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
                // The planner's usual chunk sizes get the mic count at compile time: whole six-mic trips without the per-mic
                // "is there another mic" tests, table addresses as immediates off one pointer per trip.
                // (The first request of quads and the final wait live INSIDE each alternative: no read may be in flight across
                // the C++ branch that picks one -- the compiler copies live registers at such branches and merges, and a copy
                // of a register whose read has not landed yet carries the old value.)
                auto trips = [&](auto mcc_c) {
                    constexpr int MCC = decltype(mcc_c)::value;
                    if constexpr (kPipe) issue(S[0], Dq[0], E[0].e[0]);
                    if constexpr (MCC > 0) {
                        constexpr int R0 = MCC / 6 * 6;
#pragma unroll 1
                        for (int m = 0; m < R0; m += 6) {
                            mic(0, I0{}, I0{}); mic(1, I1{}, I1{}); mic(2, I0{}, I2{});
                            mic(3, I1{}, I0{}); mic(4, I0{}, I1{}); mic(5, I1{}, I2{});
                            et += 6 * DW; ht += 6 * DW;
                        }
                        if constexpr (MCC - R0 > 0) mic(0, I0{}, I0{});
                        if constexpr (MCC - R0 > 1) mic(1, I1{}, I1{});
                        if constexpr (MCC - R0 > 2) mic(2, I0{}, I2{});
                        if constexpr (MCC - R0 > 3) mic(3, I1{}, I0{});
                        if constexpr (MCC - R0 > 4) mic(4, I0{}, I1{});
                    } else {
                        for (int m = 0; m < mcc; m += 6) {     // (plain ifs, no early exit: keeps the register state of all paths identical)
                            mic(0, I0{}, I0{});
                            if (m + 1 < mcc) mic(1, I1{}, I1{});
                            if (m + 2 < mcc) mic(2, I0{}, I2{});
                            if (m + 3 < mcc) mic(3, I1{}, I0{});
                            if (m + 4 < mcc) mic(4, I0{}, I1{});
                            if (m + 5 < mcc) mic(5, I1{}, I2{});
                            et += 6 * DW; ht += 6 * DW;
                        }
                    }
                    // nothing may stay in flight into registers the compiler is about to move or reuse
                    await_quads<NSEG>(S[0], Dq[0]);
                    if constexpr (kPipe) await_quads<NSEG>(S[NQ - 1], Dq[NQ - 1]);
                };
                if constexpr (NSEG == 1) {
                    if (mcc == 32) trips(std::integral_constant<int, 32>{});
                    else if (mcc == 16) trips(std::integral_constant<int, 16>{});
                    else trips(I0{});
                } else {
                    trips(I0{});
                }
            };
            if constexpr (FIR) {
                if (mcc == 16) fir_directions(std::integral_constant<int, 16>{});
                else if (mcc == 8) fir_directions(std::integral_constant<int, 8>{});
                else fir_directions(std::integral_constant<int, 0>{});
            } else if constexpr (DIRECT) {
                // N <= 256: the chunk sizes the planner picks get straight-line code.  With more segments the 64 accumulator
                // registers leave no room for the deeper read pipelining that buys (it spills), so only the branchy form.
                if constexpr (NSEG == 1) {
                    if (mcc == 16) directions(std::integral_constant<int, 16>{});
                    else if (mcc == 8) directions(std::integral_constant<int, 8>{});
                    else directions(std::integral_constant<int, 0>{});
                } else {
                    directions(std::integral_constant<int, 0>{});
                }
            } else {
                sweep();
            }
        }

        // ---- k-ordered mean power (pad_and_sum.c:120-128): the waves park the squared means of their directions
        // (row = direction; the rows alias the chunk buffer; a.pbw waves at a time when a group's rows outgrow it),
        // then one direction per lane runs the sequential sum: 64 chains per instruction, coalesced image stores.
        const int pw_waves = NSEG == 1 ? W : a.pbw;   // N <= 256: the planner sizes LDS for the whole group's rows
        for (int w0 = 0; w0 < W; w0 += pw_waves) {
            __syncthreads();
            if (wave >= w0 && wave < w0 + pw_waves) {
                // mean over the mics: a power-of-two count multiplies (exact), anything else divides like the reference;
                // two separate code paths so that the division sequence is never executed speculatively
                auto park = [&](auto mul_c) {
#pragma unroll
                    for (int j = 0; j < DW; ++j) {
                        float* row = lds + ((wave - w0) * DW + j) * kPark;
#pragma unroll
                        for (int sg = 0; sg < NSEG; ++sg) {
                            const f32x2 a0 = acc[j][sg][0], a1 = acc[j][sg][1];
                            float o0, o1, o2, o3;
                            if constexpr (decltype(mul_c)::value) {
                                o0 = a0.x * a.inv_n; o1 = a0.y * a.inv_n; o2 = a1.x * a.inv_n; o3 = a1.y * a.inv_n;
                            } else {
                                float fm = (float)M;
                                asm volatile("" : "+v"(fm));   // not speculatable: keeps this path behind its branch
                                o0 = a0.x / fm; o1 = a0.y / fm; o2 = a1.x / fm; o3 = a1.y / fm;
                            }
                            reinterpret_cast<float4*>(row + 256 * sg)[lane] = make_float4(o0 * o0, o1 * o1, o2 * o2, o3 * o3);
                        }
                    }
                };
                if (__builtin_expect(a.n_is_pow2, 1)) park(std::true_type{}); else park(std::false_type{});
            }
            __syncthreads();
            const int g = wave * kWave + lane;            // parked row of this lane
            const int d = g0 + w0 * DW + g;
            if (g < pw_waves * DW && d < tile_end) {
                const float* row = lds + g * kPark;
                const float4* row4 = reinterpret_cast<const float4*>(row);
                float sum = 0.0f;
                int k = 0;
                for (; k + 32 <= N; k += 32) {
                    float4 v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) v[u] = row4[(k >> 2) + u];
#pragma unroll
                    for (int u = 0; u < 8; ++u) { sum += v[u].x; sum += v[u].y; sum += v[u].z; sum += v[u].w; }
                }
                for (; k < N; ++k) sum += row[k];
                img[d - a.image_origin] = sum / (float)N;
            }
        }
    }
}


// ==================================================================================================
// Two frames per workgroup (pad / lerp, N <= 256, fixed row stride, mic count a multiple of 16, two or more frames).
//
// The sweep above is bound by the number of instructions a SIMD issues, and per (direction, mic) step only 2 (pad) / 4
// (lerp) of them are arithmetic: the rest -- table loads, the address, the offset tests, waits -- depends on the tables
// alone.  A wave that carries its eight directions through TWO frames pays that part once per 4 / 8 packed operations.
// Same layout as das_copies_kernel<.., NSEG = 1, RS = kRs, W = 16> (two shifted copies, difference rows for lerp), with the
// two frames' rows of a mic next to each other: frame 1's quads sit kFoff bytes after frame 0's, an immediate offset off the
// same address.  16 mics x 2 frames per chunk; the 64 accumulator registers leave no room for quads in flight across a
// mic, so a mic's first quads are read (and waited for) in place -- the other three waves of the SIMD cover that.
// Mic order and operation order per frame are those of the one-frame kernel: bit-identical maps.
template <int ALGO>
struct PairGeo {
    static constexpr bool kLerp = ALGO == ALGO_LERP;
    static constexpr int kA = kLerp ? 2 : 1, kC = 2, kRs = Geo<1>::kRs, kLead = Geo<1>::kLead;
    static constexpr int kSlot = kA * kC * kRs;          // floats per staged (mic, frame)
    static constexpr int kFoff = kSlot * 4;              // bytes from a frame-0 quad to the same quad of frame 1
    static constexpr int kDoff = kC * kRs * 4;           // bytes from a sample quad to its difference quad
    static constexpr int kMc = 16;                       // mics per LDS image (the digest's slot count) ...
    static constexpr int kHalf = 8;                      // ... swept and re-staged in halves of 8
};

#define BF_P_ACC(n, j, f) [a##n##0] "+v"(acc[j][f][0]), [a##n##1] "+v"(acc[j][f][1])
// pad: two direction steps x two frames; a0/a1 = step A frame 0/1, a2/a3 = step B frame 0/1
#define BF_P_PAD_STEP(n0, n1)                                                                             \
    "v_pk_add_f32 %[a" #n0 "0], %[a" #n0 "0], %[s0l]\n\tv_pk_add_f32 %[a" #n0 "1], %[a" #n0 "1], %[s0h]\n\t" \
    "v_pk_add_f32 %[a" #n1 "0], %[a" #n1 "0], %[s1l]\n\tv_pk_add_f32 %[a" #n1 "1], %[a" #n1 "1], %[s1h]\n\t"
#define BF_P_PAD_READ                                                                                     \
    "ds_read_b64 %[s0l], %[ad] offset:0\n\tds_read_b64 %[s0h], %[ad] offset:8\n\t"                         \
    "ds_read_b64 %[s1l], %[ad] offset:%[f0]\n\tds_read_b64 %[s1h], %[ad] offset:%[f8]\n\t"                 \
    "s_waitcnt lgkmcnt(0)\n\t"
// lerp: frame 0's products through v[120:123], frame 1's through v[124:127] (clobbers), so that each add is four
// instructions behind its product
#define BF_P_LERP_STEP(n0, n1, mods)                                                                      \
    "v_pk_fma_f32 v[120:121], %[h], %[d0l], %[s0l] " mods "\n\tv_pk_fma_f32 v[122:123], %[h], %[d0h], %[s0h] " mods "\n\t" \
    "v_pk_fma_f32 v[124:125], %[h], %[d1l], %[s1l] " mods "\n\tv_pk_fma_f32 v[126:127], %[h], %[d1h], %[s1h] " mods "\n\t" \
    "v_pk_add_f32 %[a" #n0 "0], %[a" #n0 "0], v[120:121]\n\tv_pk_add_f32 %[a" #n0 "1], %[a" #n0 "1], v[122:123]\n\t"         \
    "v_pk_add_f32 %[a" #n1 "0], %[a" #n1 "0], v[124:125]\n\tv_pk_add_f32 %[a" #n1 "1], %[a" #n1 "1], v[126:127]\n\t"
#define BF_P_LERP_READ                                                                                    \
    "ds_read_b64 %[s0l], %[ad] offset:0\n\tds_read_b64 %[s0h], %[ad] offset:8\n\t"                         \
    "ds_read_b64 %[d0l], %[ad] offset:%[g0]\n\tds_read_b64 %[d0h], %[ad] offset:%[g8]\n\t"                 \
    "ds_read_b64 %[s1l], %[ad] offset:%[f0]\n\tds_read_b64 %[s1h], %[ad] offset:%[f8]\n\t"                 \
    "ds_read_b64 %[d1l], %[ad] offset:%[k0]\n\tds_read_b64 %[d1h], %[ad] offset:%[k8]\n\t"                 \
    "s_waitcnt lgkmcnt(0)\n\t"
#define BF_P_ADDR(e) "v_add_u32 %[ad], %[" #e "], %[lb]\n\t"
#define BF_P_CHECK(n, ep, ec) "s_cmp_lg_u32 %[" #ec "], %[" #ep "]\n\ts_cbranch_scc1 .Lr" #n "_%=\n.Lb" #n "_%=:\n\t"
#define BF_P_STUB(n, ec, READ) ".Lr" #n "_%=:\n\t" BF_P_ADDR(ec) READ "s_branch .Lb" #n "_%=\n"

// Direction steps 2 Q and 2 Q + 1 of a mic for both frames.  Q = 0 reads the mic's first quads in place (offset eb; ea is
// unused) before step 0; the others test ea -> eb before their first step; all test eb -> ec before their second.
template <int ALGO, int Q>
__device__ __forceinline__ void pair_steps(f32x2 (&acc)[8][2][2], Quad& S0, Quad& D0, Quad& S1, Quad& D1, int ea, int eb, int ec,
                                           unsigned long long h, int lbase)
{
    using G = PairGeo<ALGO>;
    constexpr int JA = 2 * Q, JB = 2 * Q + 1;
    int ad;
    if constexpr (ALGO == ALGO_PAD) {
        // pad_and_sum.c:41-47   out[k] += s[k - p]
        if constexpr (Q == 0) {
            // (step 0: each frame's adds wait only for that frame's two reads -- LDS returns in order)
            asm volatile(BF_P_ADDR(eb)
                         "ds_read_b64 %[s0l], %[ad] offset:0\n\tds_read_b64 %[s0h], %[ad] offset:8\n\t"
                         "ds_read_b64 %[s1l], %[ad] offset:%[f0]\n\tds_read_b64 %[s1h], %[ad] offset:%[f8]\n\ts_waitcnt lgkmcnt(2)\n\t"
                         "v_pk_add_f32 %[a00], %[a00], %[s0l]\n\tv_pk_add_f32 %[a01], %[a01], %[s0h]\n\ts_waitcnt lgkmcnt(0)\n\t"
                         "v_pk_add_f32 %[a10], %[a10], %[s1l]\n\tv_pk_add_f32 %[a11], %[a11], %[s1h]\n\t"
                         BF_P_CHECK(1, eb, ec) BF_P_PAD_STEP(2, 3)
                         ".subsection 1\n" BF_P_STUB(1, ec, BF_P_PAD_READ) "\t.subsection 0"
                         // (the quads are pure outputs here: as in-out operands they are carried around the mic loop -- and copied at its back-edge)
                         : BF_P_ACC(0, JA, 0), BF_P_ACC(1, JA, 1), BF_P_ACC(2, JB, 0), BF_P_ACC(3, JB, 1), [s0l] "=&v"(S0.lo), [s0h] "=&v"(S0.hi),
                           [s1l] "=&v"(S1.lo), [s1h] "=&v"(S1.hi), [ad] "=&v"(ad)
                         : [eb] "s"(eb), [ec] "s"(ec), [lb] "v"(lbase), [f0] "n"(G::kFoff), [f8] "n"(G::kFoff + 8) : "scc");
        } else {
            asm volatile(BF_P_CHECK(0, ea, eb) BF_P_PAD_STEP(0, 1) BF_P_CHECK(1, eb, ec) BF_P_PAD_STEP(2, 3)
                         ".subsection 1\n" BF_P_STUB(0, eb, BF_P_PAD_READ) BF_P_STUB(1, ec, BF_P_PAD_READ) "\t.subsection 0"
                         : BF_P_ACC(0, JA, 0), BF_P_ACC(1, JA, 1), BF_P_ACC(2, JB, 0), BF_P_ACC(3, JB, 1), [s0l] "+v"(S0.lo), [s0h] "+v"(S0.hi),
                           [s1l] "+v"(S1.lo), [s1h] "+v"(S1.hi), [ad] "=&v"(ad)
                         : [ea] "s"(ea), [eb] "s"(eb), [ec] "s"(ec), [lb] "v"(lbase), [f0] "n"(G::kFoff), [f8] "n"(G::kFoff + 8) : "scc");
        }
    } else {
        // lerp_and_sum.c:50-56  out[k] += s[i] + h * (s[i+1] - s[i]),  i = k - p - 1   (gcc contracts it into one fma)
#define BF_P_LERP_OUTS                                                                                                         \
        BF_P_ACC(0, JA, 0), BF_P_ACC(1, JA, 1), BF_P_ACC(2, JB, 0), BF_P_ACC(3, JB, 1), [s0l] "+v"(S0.lo), [s0h] "+v"(S0.hi),    \
        [d0l] "+v"(D0.lo), [d0h] "+v"(D0.hi), [s1l] "+v"(S1.lo), [s1h] "+v"(S1.hi), [d1l] "+v"(D1.lo), [d1h] "+v"(D1.hi), [ad] "=&v"(ad)
#define BF_P_LERP_IMMS                                                                                                         \
        [g0] "n"(G::kDoff), [g8] "n"(G::kDoff + 8), [f0] "n"(G::kFoff), [f8] "n"(G::kFoff + 8), [k0] "n"(G::kFoff + G::kDoff),   \
        [k8] "n"(G::kFoff + G::kDoff + 8)
        if constexpr (Q == 0) {
            asm volatile(BF_P_ADDR(eb) BF_P_LERP_READ BF_P_LERP_STEP(0, 1, "op_sel_hi:[0,1,1]") BF_P_CHECK(1, eb, ec)
                         BF_P_LERP_STEP(2, 3, "op_sel:[1,0,0] op_sel_hi:[1,1,1]")
                         ".subsection 1\n" BF_P_STUB(1, ec, BF_P_LERP_READ) "\t.subsection 0"
                         : BF_P_LERP_OUTS
                         : [eb] "s"(eb), [ec] "s"(ec), [h] "s"(h), [lb] "v"(lbase), BF_P_LERP_IMMS
                         : "scc", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127");
        } else {
            asm volatile(BF_P_CHECK(0, ea, eb) BF_P_LERP_STEP(0, 1, "op_sel_hi:[0,1,1]") BF_P_CHECK(1, eb, ec)
                         BF_P_LERP_STEP(2, 3, "op_sel:[1,0,0] op_sel_hi:[1,1,1]")
                         ".subsection 1\n" BF_P_STUB(0, eb, BF_P_LERP_READ) BF_P_STUB(1, ec, BF_P_LERP_READ) "\t.subsection 0"
                         : BF_P_LERP_OUTS
                         : [ea] "s"(ea), [eb] "s"(eb), [ec] "s"(ec), [h] "s"(h), [lb] "v"(lbase), BF_P_LERP_IMMS
                         : "scc", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127");
        }
#undef BF_P_LERP_OUTS
#undef BF_P_LERP_IMMS
    }
}

// pad, direction steps 2..7 of a mic for both frames as ONE statement (between two statements the hazard recogniser puts an s_nop:
// three issue slots per mic with one statement per pair of steps).
__device__ __forceinline__ void pair_pad_rest(f32x2 (&acc)[8][2][2], Quad& S0, Quad& S1, const int (&e)[8], int lbase)
{
    using G = PairGeo<ALGO_PAD>;
    int ad;
#define BF_P_ACC2(n, j) [a##n##0] "+v"(acc[j][0][0]), [a##n##1] "+v"(acc[j][0][1]), [b##n##0] "+v"(acc[j][1][0]), [b##n##1] "+v"(acc[j][1][1])
#define BF_P_PAD1(n) "v_pk_add_f32 %[a" #n "0], %[a" #n "0], %[s0l]\n\tv_pk_add_f32 %[a" #n "1], %[a" #n "1], %[s0h]\n\t" \
                     "v_pk_add_f32 %[b" #n "0], %[b" #n "0], %[s1l]\n\tv_pk_add_f32 %[b" #n "1], %[b" #n "1], %[s1h]\n\t"
    asm volatile(BF_P_CHECK(2, e1, e2) BF_P_PAD1(2) BF_P_CHECK(3, e2, e3) BF_P_PAD1(3) BF_P_CHECK(4, e3, e4) BF_P_PAD1(4)
                 BF_P_CHECK(5, e4, e5) BF_P_PAD1(5) BF_P_CHECK(6, e5, e6) BF_P_PAD1(6) BF_P_CHECK(7, e6, e7) BF_P_PAD1(7)
                 ".subsection 1\n" BF_P_STUB(2, e2, BF_P_PAD_READ) BF_P_STUB(3, e3, BF_P_PAD_READ) BF_P_STUB(4, e4, BF_P_PAD_READ)
                 BF_P_STUB(5, e5, BF_P_PAD_READ) BF_P_STUB(6, e6, BF_P_PAD_READ) BF_P_STUB(7, e7, BF_P_PAD_READ) "\t.subsection 0"
                 : BF_P_ACC2(2, 2), BF_P_ACC2(3, 3), BF_P_ACC2(4, 4), BF_P_ACC2(5, 5), BF_P_ACC2(6, 6), BF_P_ACC2(7, 7),
                   [s0l] "+v"(S0.lo), [s0h] "+v"(S0.hi), [s1l] "+v"(S1.lo), [s1h] "+v"(S1.hi), [ad] "=&v"(ad)
                 : [e1] "s"(e[1]), [e2] "s"(e[2]), [e3] "s"(e[3]), [e4] "s"(e[4]), [e5] "s"(e[5]), [e6] "s"(e[6]), [e7] "s"(e[7]), [lb] "v"(lbase),
                   [f0] "n"(G::kFoff), [f8] "n"(G::kFoff + 8)
                 : "scc");
#undef BF_P_ACC2
#undef BF_P_PAD1
}

// Profiling build only (-DBF_STAMPS, scripts/dev/phase_stamps.py): every wave sums the time it spends in each phase of
// das_pair_kernel / das_pair2_kernel (s_memtime at the phase boundaries, which are barrier neighbours anyway) and adds the totals to
// g_stamps[phase] when its workgroup ends.  BF_STAMP(k) closes the phase that was running and charges it to slot k:
//   0 sweep  1 wait (chunk free)  2 staging  3 wait (chunk staged)  4 wait (power: rows free)  5 parking  6 wait (rows parked)  7 ordered sum
#ifdef BF_STAMPS
__device__ unsigned long long g_stamps[64 * 16];         // (sums over all waves, in 64 replicas picked by workgroup id: nine atomics per wave on one
                                                         //  line would make the flush longer than the kernel)
#define BF_STAMP_DECL unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long st_prev = __builtin_amdgcn_s_memtime();
#define BF_STAMP(k) do { const unsigned long long st_now = __builtin_amdgcn_s_memtime(); st_acc[k] += st_now - st_prev; st_prev = st_now; } while (0)
#define BF_STAMP_FLUSH do { if (lane == 0) { unsigned long long* gs = g_stamps + 16 * (blockIdx.x & 63); for (int i = 0; i < 8; ++i) atomicAdd(&gs[i], st_acc[i]); atomicAdd(&gs[8], 1ull); } } while (0)
#else
#define BF_STAMP_DECL
#define BF_STAMP(k)
#define BF_STAMP_FLUSH
#endif

template <int ALGO>
__global__ void __launch_bounds__(1024, 4) das_pair_kernel(BF_TABLE_PARAMS, KArgs a)
{
    using G = PairGeo<ALGO>;
    constexpr bool kLerp = G::kLerp;
    constexpr int A = G::kA, C = G::kC, RS = G::kRs, LEAD = G::kLead, HC = G::kHalf, W = 16, DW = 8, kGroup = DW * W, kPark = Geo<1>::kPark;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    int tile, fpair;
    tile_and_frame(a, &tile, &fpair);
    const int f0 = 2 * fpair;
    const bool two = f0 + 1 < a.n_frames;                      // an odd frame count: the last workgroup row computes its frame twice
    const int f1 = two ? f0 + 1 : f0;
    const int tile_begin = a.dir_begin + tile * a.tile_dirs;
    if (tile_begin >= a.dir_end) return;
    const int tile_end = min(tile_begin + a.tile_dirs, a.dir_end);
    const int M = a.n_mics, N = a.n_samples;                   // M % 16 == 0, N % 4 == 0, N <= 256 (plan_das)
    const int n_half = M / HC;                                  // half chunks of 8 mics
    const float* __restrict__ sig0 = signals + (size_t)f0 * a.m_total * N;
    const float* __restrict__ sig1 = signals + (size_t)f1 * a.m_total * N;
    float* __restrict__ img0 = images + (size_t)f0 * a.image_stride;
    float* __restrict__ img1 = images + (size_t)f1 * a.image_stride;
    const int32_t* __restrict__ dig = reinterpret_cast<const int32_t*>(taps);   // the digest rides in the unused `taps` slot

    // The LDS image is the 16-mic chunk the digest was built for (mic m -> slot m % 16), used as TWO halves of 8 mics: while
    // the waves sweep half h, each of them also writes its row of half h + 1 into the other half -- ONE barrier per 8 mics,
    // and the staging stores (slow: 13 cycles per ds_write_b128 and wave on the LDS store path) run under other waves' adds
    // instead of between two barriers with every SIMD idle.
    // This wave stages row `wave` of every half: mic (wave >> 1) of the half, frame (wave & 1).
    // Lane c holds half c's mic id (first 64 halves): the per-half prefetch is then one independent load.
    const int vmic = (lane < n_half) ? mics[lane * HC + (wave >> 1)] : 0;
    auto fetch = [&](int h) -> float4 {
        const int mic = (h < kWave) ? __builtin_amdgcn_readlane(vmic, h) : mics[h * HC + (wave >> 1)];
        const float* src = ((wave & 1) ? sig1 : sig0) + (size_t)mic * N;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (4 * lane < N) v = *reinterpret_cast<const float4*>(src + 4 * lane);
        return v;
    };
    auto stage = [&](int h, const float4 v, bool wipe) {
        float* row0 = lds + (((h & 1) * W) + wave) * G::kSlot;  // slot (h & 1) * 8 + (wave >> 1), frame wave & 1
        const float py = dpp_prev(v.y), pz = dpp_prev(v.z), pw = dpp_prev(v.w), nx = dpp_next(v.x);
        write_copies<C>(row0, RS, LEAD, lane, v, py, pz, pw);
        if constexpr (kLerp) {
            // D[i] = s[i+1] - s[i], the reference's own subtraction (lerp_and_sum.c:54); D[-1] stays 0 (prefix)
            const float4 dq = make_float4(v.y - v.x, v.z - v.y, v.w - v.z, nx - v.w);
            const float dy = dpp_prev(dq.y), dz = dpp_prev(dq.z), dw = dpp_prev(dq.w);
            write_copies<C>(row0 + C * RS, RS, LEAD, lane, dq, dy, dz, dw);
        }
        if (wipe) {
            // the zero prefix: nothing but the parked rows of the power pass ever overwrites it, so only a group's first
            // visit of a half restores it -- one store: lane -> (copy row lane / 14, quad lane % 14) of the C * A rows
            static_assert((LEAD >> 2) * C * A <= kWave, "one lane per prefix quad");
            constexpr int PQ = LEAD >> 2;
            if (lane < PQ * C * A) reinterpret_cast<float4*>(row0 + (lane / PQ) * RS)[lane % PQ] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };

    float4 st = fetch(0);
    const int lb = 16 * lane + (int)(unsigned)(size_t)((__attribute__((address_space(3))) char*)lds);
    BF_STAMP_DECL

    for (int g0 = tile_begin; g0 < tile_end; g0 += kGroup) {
        f32x2 acc[DW][2][2];
#pragma unroll
        for (int j = 0; j < DW; ++j)
#pragma unroll
            for (int f = 0; f < 2; ++f) { acc[j][f][0] = f32x2{0.0f, 0.0f}; acc[j][f][1] = f32x2{0.0f, 0.0f}; }

        BF_STAMP(7);
        __syncthreads();   // the previous group's parked rows have been summed
        BF_STAMP(1);
        stage(0, st, true);
        st = fetch(1);
        BF_STAMP(2);
        __syncthreads();
        BF_STAMP(3);

        for (int h = 0; h < n_half; ++h) {
            if (h + 1 < n_half) {
                stage(h + 1, st, h == 0);                       // into the half whose sweeps ended before the last barrier
                // request what is staged an iteration from now: half h + 2, or the next group's first half (the same rows)
                if (h + 2 < n_half) st = fetch(h + 2);
                else if (g0 + kGroup < tile_end) st = fetch(0);
                BF_STAMP(2);
            }
            const int dw0 = g0 + wave * DW;                     // wave-uniform
            if (dw0 < tile_end) {
                const size_t grp = (size_t)(dw0 - a.dir_begin) / DW;
                const int32_t* __restrict__ et = dig + (grp * M + (size_t)h * HC) * DW;
                const float* __restrict__ ht = reinterpret_cast<const float*>(dig) + a.digest_h_off + (grp * M + (size_t)h * HC) * DW;
                struct Entries { int e[DW]; unsigned long long hp[DW / 2]; };
                auto request = [&](Entries& t, int m) {
                    // (reads past the half's last mic stay inside the slack-padded table and are dropped)
#pragma unroll
                    for (int j = 0; j < DW; ++j) t.e[j] = et[m * DW + j];
#pragma unroll
                    for (int j = 0; j < DW / 2; ++j) {
                        t.hp[j] = 0;
                        if constexpr (kLerp) t.hp[j] = *reinterpret_cast<const unsigned long long*>(ht + m * DW + 2 * j);
                    }
                };
                Entries E[3];
                Quad S0, D0, S1, D1;
                S0.lo = S0.hi = D0.lo = D0.hi = S1.lo = S1.hi = D1.lo = D1.hi = f32x2{0.0f, 0.0f};
                request(E[0], 0);
                request(E[1], 1);
                auto mic = [&](int m, auto kc) {
                    constexpr int K = decltype(kc)::value, K2 = (K + 2) % 3;
                    const Entries& cur = E[K];
                    pair_steps<ALGO, 0>(acc, S0, D0, S1, D1, cur.e[0], cur.e[0], cur.e[1], cur.hp[0], lb);
                    request(E[K2], m + 2);      // after the first statement's wait, so that it does not sit on these loads
                    if constexpr (ALGO == ALGO_PAD) {
                        pair_pad_rest(acc, S0, S1, cur.e, lb);
                        return;
                    }
                    pair_steps<ALGO, 1>(acc, S0, D0, S1, D1, cur.e[1], cur.e[2], cur.e[3], cur.hp[1], lb);
                    pair_steps<ALGO, 2>(acc, S0, D0, S1, D1, cur.e[3], cur.e[4], cur.e[5], cur.hp[2], lb);
                    pair_steps<ALGO, 3>(acc, S0, D0, S1, D1, cur.e[5], cur.e[6], cur.e[7], cur.hp[3], lb);
                };
                using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
                static_assert(HC == 8, "eight mics: two trips of three and two more");
#pragma unroll 1
                for (int t = 0; t < 2; ++t) {
                    mic(0, I0{}); mic(1, I1{}); mic(2, I2{});
                    et += 3 * DW; ht += 3 * DW;
                }
                mic(0, I0{});
                mic(1, I1{});
            }
            BF_STAMP(0);       // sweep -> waiting for the others
            __syncthreads();   // half h is free, half h + 1 is staged
            BF_STAMP(h + 1 < n_half ? 3 : 4);
        }

        // ---- k-ordered mean power (pad_and_sum.c:120-128), one frame at a time: the 16 waves park the squared means of their
        // directions (row = direction; the rows alias the chunk buffer), then one direction per lane runs the sequential sum.
#pragma unroll
        for (int f = 0; f < 2; ++f) {
            if (f == 1) {
                __syncthreads();        // frame 0's rows have been summed
                BF_STAMP(4);
            }
            auto park = [&](auto mul_c) {
#pragma unroll
                for (int j = 0; j < DW; ++j) {
                    float* row = lds + (wave * DW + j) * kPark;
                    const f32x2 a0 = acc[j][f][0], a1 = acc[j][f][1];
                    float o0, o1, o2, o3;
                    if constexpr (decltype(mul_c)::value) {
                        o0 = a0.x * a.inv_n; o1 = a0.y * a.inv_n; o2 = a1.x * a.inv_n; o3 = a1.y * a.inv_n;
                    } else {
                        float fm = (float)M;
                        asm volatile("" : "+v"(fm));   // not speculatable: keeps this path behind its branch
                        o0 = a0.x / fm; o1 = a0.y / fm; o2 = a1.x / fm; o3 = a1.y / fm;
                    }
                    reinterpret_cast<float4*>(row)[lane] = make_float4(o0 * o0, o1 * o1, o2 * o2, o3 * o3);
                }
            };
            if (__builtin_expect(a.n_is_pow2, 1)) park(std::true_type{}); else park(std::false_type{});
            BF_STAMP(5);                // -> waiting
            __syncthreads();
            BF_STAMP(6);                // -> ordered sum (two waves; the others go on to the next barrier)
            const int g = wave * kWave + lane;            // parked row of this lane
            const int d = g0 + g;
            if (g < kGroup && d < tile_end && (f == 0 || two)) {
                const float* row = lds + g * kPark;
                const float4* row4 = reinterpret_cast<const float4*>(row);
                float sum = 0.0f;
                int k = 0;
                for (; k + 32 <= N; k += 32) {
                    float4 v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) v[u] = row4[(k >> 2) + u];
#pragma unroll
                    for (int u = 0; u < 8; ++u) { sum += v[u].x; sum += v[u].y; sum += v[u].z; sum += v[u].w; }
                }
                for (; k < N; ++k) sum += row[k];
                (f == 0 ? img0 : img1)[d - a.image_origin] = sum / (float)N;
            }
        }
    }
    BF_STAMP(7);
    BF_STAMP_FLUSH;
}
#undef BF_P_ACC
#undef BF_P_PAD_STEP
#undef BF_P_PAD_READ
#undef BF_P_LERP_STEP
#undef BF_P_LERP_READ
#undef BF_P_ADDR
#undef BF_P_CHECK
#undef BF_P_STUB

// ==================================================================================================
// Two frames per workgroup, frames INTERLEAVED sample by sample in the LDS rows (pad / lerp, N <= 256; das_pair_kernel's successor).
//
// Every instruction costs a SIMD a quad-cycle, so what is left to gain on the sweep is instruction count.  With the
// row of a mic holding (f0 s0, f1 s0, f0 s1, f1 s1, ..):
//   * one ds_read_b128 brings two samples of BOTH frames: a (re)load is 2 (pad) / 4 (lerp) LDS instructions instead of 4 / 8, and
//     with lane l owning the sample pairs (2l, 2l+1) and (128+2l, 128+2l+1) every read covers 1 KiB of contiguous LDS (no bank
//     conflicts; the 16-byte lane stride of ds_read_b64 pairs was a two-way conflict on every read);
//   * a register pair is (frame 0, frame 1) of one sample, so the packed operations are the same 4 (pad) / 8 (lerp) per
//     direction step, the lerp weight still one scalar operand for both lanes;
//   * the quads live in HARD-WIRED registers v[96:111] (+ products v[112:119], address v120), named as clobbers: 16-byte reads
//     need 4-register tuples whose halves the packed operations address, which inline-asm operands cannot express.  A mic is
//     two statements -- S1: address, reads, wait, step 0;  S2: steps 1..7 with their tests and out-of-line re-reads -- and the
//     quads must survive from S1 to S2 across the table requests the compiler places between them (scalar instructions only;
//     tests/test_isa_hazards.py checks that nothing between the two markers touches a vector register).
// Halves of 8 mics staged under the sweep, power pass, digest: as das_pair_kernel (digest offsets scaled for the 2-float samples).
template <int ALGO>
struct Pair2Geo {
    static constexpr bool kLerp = ALGO == ALGO_LERP;
    static constexpr int kA = kLerp ? 2 : 1, kC = 2, kLead = Geo<1>::kLead, kRs = 2 * Geo<1>::kRs, kMc = 16, kHalf = 8;
    static constexpr int kSlot = kA * kC * kRs;          // floats per staged mic (both frames)
    static constexpr int kDoff = kC * kRs * 4;           // bytes from a sample quad to its difference quad
};

#define BF_I_ACC(n, j) [a##n##0] "+v"(acc[j][0]), [a##n##1] "+v"(acc[j][1]), [a##n##2] "+v"(acc[j][2]), [a##n##3] "+v"(acc[j][3])
#define BF_I_READ_PAD "ds_read_b128 v[96:99], v120\n\tds_read_b128 v[100:103], v120 offset:1024\n\ts_waitcnt lgkmcnt(0)\n\t"
#define BF_I_READ_LERP                                                                              \
    "ds_read_b128 v[96:99], v120\n\tds_read_b128 v[100:103], v120 offset:1024\n\t"                   \
    "ds_read_b128 v[104:107], v120 offset:%[g0]\n\tds_read_b128 v[108:111], v120 offset:%[g1]\n\ts_waitcnt lgkmcnt(0)\n\t"
#define BF_I_PAD_STEP(n)                                                                            \
    "v_pk_add_f32 %[a" #n "0], %[a" #n "0], v[96:97]\n\tv_pk_add_f32 %[a" #n "1], %[a" #n "1], v[98:99]\n\t" \
    "v_pk_add_f32 %[a" #n "2], %[a" #n "2], v[100:101]\n\tv_pk_add_f32 %[a" #n "3], %[a" #n "3], v[102:103]\n\t"
#define BF_I_LERP_STEP(n, h, mods)                                                                  \
    "v_pk_fma_f32 v[112:113], %[" #h "], v[104:105], v[96:97] " mods "\n\tv_pk_fma_f32 v[114:115], %[" #h "], v[106:107], v[98:99] " mods "\n\t" \
    "v_pk_fma_f32 v[116:117], %[" #h "], v[108:109], v[100:101] " mods "\n\tv_pk_fma_f32 v[118:119], %[" #h "], v[110:111], v[102:103] " mods "\n\t" \
    "v_pk_add_f32 %[a" #n "0], %[a" #n "0], v[112:113]\n\tv_pk_add_f32 %[a" #n "1], %[a" #n "1], v[114:115]\n\t" \
    "v_pk_add_f32 %[a" #n "2], %[a" #n "2], v[116:117]\n\tv_pk_add_f32 %[a" #n "3], %[a" #n "3], v[118:119]\n\t"
// A mic's first reads with direction step 0 behind them, each half of the step waiting only for its own reads (LDS returns in order;
// a counted wait bounds the outstanding operations of any kind, hence the outstanding reads).
#define BF_I_FIRST_PAD                                                                              \
    "ds_read_b128 v[96:99], v120\n\tds_read_b128 v[100:103], v120 offset:1024\n\ts_waitcnt lgkmcnt(1)\n\t"  \
    "v_pk_add_f32 %[a00], %[a00], v[96:97]\n\tv_pk_add_f32 %[a01], %[a01], v[98:99]\n\ts_waitcnt lgkmcnt(0)\n\t" \
    "v_pk_add_f32 %[a02], %[a02], v[100:101]\n\tv_pk_add_f32 %[a03], %[a03], v[102:103]\n\t"
#define BF_I_FIRST_LERP(h, mods)                                                                    \
    "ds_read_b128 v[96:99], v120\n\tds_read_b128 v[104:107], v120 offset:%[g0]\n\t"                \
    "ds_read_b128 v[100:103], v120 offset:1024\n\tds_read_b128 v[108:111], v120 offset:%[g1]\n\ts_waitcnt lgkmcnt(2)\n\t" \
    "v_pk_fma_f32 v[112:113], %[" #h "], v[104:105], v[96:97] " mods "\n\tv_pk_fma_f32 v[114:115], %[" #h "], v[106:107], v[98:99] " mods "\n\t" \
    "v_pk_add_f32 %[a00], %[a00], v[112:113]\n\tv_pk_add_f32 %[a01], %[a01], v[114:115]\n\ts_waitcnt lgkmcnt(0)\n\t" \
    "v_pk_fma_f32 v[116:117], %[" #h "], v[108:109], v[100:101] " mods "\n\tv_pk_fma_f32 v[118:119], %[" #h "], v[110:111], v[102:103] " mods "\n\t" \
    "v_pk_add_f32 %[a02], %[a02], v[116:117]\n\tv_pk_add_f32 %[a03], %[a03], v[118:119]\n\t"
#define BF_I_EVEN "op_sel_hi:[0,1,1]"
#define BF_I_ODD "op_sel:[1,0,0] op_sel_hi:[1,1,1]"
#define BF_I_CHECK(n, ep, ec) "s_cmp_lg_u32 %[" #ec "], %[" #ep "]\n\ts_cbranch_scc1 .Lr" #n "_%=\n.Lb" #n "_%=:\n\t"
#define BF_I_STUB(n, ec, READ) ".Lr" #n "_%=:\n\tv_add_u32 v120, %[" #ec "], %[lb]\n\t" READ "s_branch .Lb" #n "_%=\n"
#define BF_I_CLOB "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", \
                  "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120"

// S1: a mic's first quads, read in place, and direction step 0
template <int ALGO>
__device__ __forceinline__ void pair2_first(f32x2 (&acc)[8][4], int e0, unsigned long long h01, int lbase)
{
    using G = Pair2Geo<ALGO>;
    if constexpr (ALGO == ALGO_PAD) {
        // pad_and_sum.c:41-47   out[k] += s[k - p]
        asm volatile("v_add_u32 v120, %[e0], %[lb]\n\t" BF_I_FIRST_PAD ";BF_S1_END"
                     : BF_I_ACC(0, 0) : [e0] "s"(e0), [lb] "v"(lbase) : BF_I_CLOB);
    } else {
        // lerp_and_sum.c:50-56  out[k] += s[i] + h * (s[i+1] - s[i]),  i = k - p - 1   (gcc contracts it into one fma)
        asm volatile("v_add_u32 v120, %[e0], %[lb]\n\t" BF_I_FIRST_LERP(h01, BF_I_EVEN) ";BF_S1_END"
                     : BF_I_ACC(0, 0) : [e0] "s"(e0), [h01] "s"(h01), [lb] "v"(lbase), [g0] "n"(G::kDoff), [g1] "n"(G::kDoff + 1024) : BF_I_CLOB);
    }
}
// S2: direction steps 1..7, each behind the test of its LDS offset against the previous step's
template <int ALGO>
__device__ __forceinline__ void pair2_rest(f32x2 (&acc)[8][4], const int (&e)[8], const unsigned long long (&hp)[4], int lbase)
{
    using G = Pair2Geo<ALGO>;
#define BF_I_S2_ACCS BF_I_ACC(1, 1), BF_I_ACC(2, 2), BF_I_ACC(3, 3), BF_I_ACC(4, 4), BF_I_ACC(5, 5), BF_I_ACC(6, 6), BF_I_ACC(7, 7)
#define BF_I_S2_E [e0] "s"(e[0]), [e1] "s"(e[1]), [e2] "s"(e[2]), [e3] "s"(e[3]), [e4] "s"(e[4]), [e5] "s"(e[5]), [e6] "s"(e[6]), [e7] "s"(e[7]), [lb] "v"(lbase)
    if constexpr (ALGO == ALGO_PAD) {
        asm volatile(";BF_S2_BEGIN\n\t"
                     BF_I_CHECK(1, e0, e1) BF_I_PAD_STEP(1) BF_I_CHECK(2, e1, e2) BF_I_PAD_STEP(2) BF_I_CHECK(3, e2, e3) BF_I_PAD_STEP(3)
                     BF_I_CHECK(4, e3, e4) BF_I_PAD_STEP(4) BF_I_CHECK(5, e4, e5) BF_I_PAD_STEP(5) BF_I_CHECK(6, e5, e6) BF_I_PAD_STEP(6)
                     BF_I_CHECK(7, e6, e7) BF_I_PAD_STEP(7)
                     ".subsection 1\n" BF_I_STUB(1, e1, BF_I_READ_PAD) BF_I_STUB(2, e2, BF_I_READ_PAD) BF_I_STUB(3, e3, BF_I_READ_PAD)
                     BF_I_STUB(4, e4, BF_I_READ_PAD) BF_I_STUB(5, e5, BF_I_READ_PAD) BF_I_STUB(6, e6, BF_I_READ_PAD) BF_I_STUB(7, e7, BF_I_READ_PAD)
                     "\t.subsection 0"
                     : BF_I_S2_ACCS : BF_I_S2_E : "scc", BF_I_CLOB);
    } else {
        asm volatile(";BF_S2_BEGIN\n\t"
                     BF_I_CHECK(1, e0, e1) BF_I_LERP_STEP(1, h01, BF_I_ODD) BF_I_CHECK(2, e1, e2) BF_I_LERP_STEP(2, h23, BF_I_EVEN)
                     BF_I_CHECK(3, e2, e3) BF_I_LERP_STEP(3, h23, BF_I_ODD) BF_I_CHECK(4, e3, e4) BF_I_LERP_STEP(4, h45, BF_I_EVEN)
                     BF_I_CHECK(5, e4, e5) BF_I_LERP_STEP(5, h45, BF_I_ODD) BF_I_CHECK(6, e5, e6) BF_I_LERP_STEP(6, h67, BF_I_EVEN)
                     BF_I_CHECK(7, e6, e7) BF_I_LERP_STEP(7, h67, BF_I_ODD)
                     ".subsection 1\n" BF_I_STUB(1, e1, BF_I_READ_LERP) BF_I_STUB(2, e2, BF_I_READ_LERP) BF_I_STUB(3, e3, BF_I_READ_LERP)
                     BF_I_STUB(4, e4, BF_I_READ_LERP) BF_I_STUB(5, e5, BF_I_READ_LERP) BF_I_STUB(6, e6, BF_I_READ_LERP) BF_I_STUB(7, e7, BF_I_READ_LERP)
                     "\t.subsection 0"
                     : BF_I_S2_ACCS
                     : BF_I_S2_E, [h01] "s"(hp[0]), [h23] "s"(hp[1]), [h45] "s"(hp[2]), [h67] "s"(hp[3]), [g0] "n"(G::kDoff), [g1] "n"(G::kDoff + 1024)
                     : "scc", BF_I_CLOB);
    }
#undef BF_I_S2_ACCS
#undef BF_I_S2_E
}
#undef BF_I_ACC
#undef BF_I_READ_PAD
#undef BF_I_FIRST_PAD
#undef BF_I_FIRST_LERP
#undef BF_I_READ_LERP
#undef BF_I_PAD_STEP
#undef BF_I_LERP_STEP
#undef BF_I_EVEN
#undef BF_I_ODD
#undef BF_I_CHECK
#undef BF_I_STUB
#undef BF_I_CLOB

template <int ALGO>
__global__ void __launch_bounds__(1024, 4) das_pair2_kernel(BF_TABLE_PARAMS, KArgs a)
{
    using G = Pair2Geo<ALGO>;
    constexpr bool kLerp = G::kLerp;
    constexpr int A = G::kA, C = G::kC, RS = G::kRs, LEAD = G::kLead, HC = G::kHalf, W = 16, DW = 8, kGroup = DW * W, kPark = Geo<1>::kPark;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane_ = threadIdx.x & (kWave - 1), lane = lane_;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    int tile, fpair;
    tile_and_frame(a, &tile, &fpair);
    const int f0 = 2 * fpair;
    const bool two = f0 + 1 < a.n_frames;                      // an odd frame count: the last workgroup row computes its frame twice
    const int f1 = two ? f0 + 1 : f0;
    const int tile_begin = a.dir_begin + tile * a.tile_dirs;
    if (tile_begin >= a.dir_end) return;
    const int tile_end = min(tile_begin + a.tile_dirs, a.dir_end);
    const int M = a.n_mics, N = a.n_samples;                   // M % 16 == 0, N % 4 == 0, N <= 256 (plan_das)
    const int n_half = M / HC;
    const float* __restrict__ sig0 = signals + (size_t)f0 * a.m_total * N;
    const float* __restrict__ sig1 = signals + (size_t)f1 * a.m_total * N;
    float* __restrict__ img0 = images + (size_t)f0 * a.image_stride;
    float* __restrict__ img1 = images + (size_t)f1 * a.image_stride;
    const int32_t* __restrict__ dig = reinterpret_cast<const int32_t*>(taps);   // the digest rides in the unused `taps` slot

    // Staging: waves w and w + 8 share mic (w & 7) of every half (both fetch its two frames): part 0 writes the sample rows
    // (lerp) / copy 0 (pad), part 1 the difference rows (lerp) / copy 1 (pad).  Lane c holds half c's mic id (first 64 halves).
    const int my_mic = wave & 7, part = wave >> 3;
    const int vmic = (lane < n_half) ? mics[lane * HC + my_mic] : 0;
    struct Staged2 { float4 v0, v1; };
    auto fetch = [&](int h) -> Staged2 {
        const int mic = (h < kWave) ? __builtin_amdgcn_readlane(vmic, h) : mics[h * HC + my_mic];
        Staged2 st;
        st.v0 = make_float4(0.f, 0.f, 0.f, 0.f);
        st.v1 = st.v0;
        if (4 * lane < N) {
            if constexpr (kLerp) {
                // scalar row base + one 32-bit lane offset (global_load saddr form): the per-lane 64-bit pointers of the two frames, hoisted out
                // of the group loop, used to be spilled around the sweep (16 bytes of scratch per lane, stored once per workgroup and
                // reloaded per group: WRITE_SIZE 5x the image bytes)
                unsigned voff = 16u * (unsigned)lane;
                asm volatile("" : "+v"(voff));                  // (opaque: or hipcc folds it back into two hoisted 64-bit lane pointers)
                const char* r0 = reinterpret_cast<const char*>(sig0 + (size_t)mic * N);
                const char* r1 = reinterpret_cast<const char*>(sig1 + (size_t)mic * N);
                st.v0 = *reinterpret_cast<const float4*>(r0 + voff);
                st.v1 = *reinterpret_cast<const float4*>(r1 + voff);
            } else {
                st.v0 = *reinterpret_cast<const float4*>(sig0 + (size_t)mic * N + 4 * lane);
                st.v1 = *reinterpret_cast<const float4*>(sig1 + (size_t)mic * N + 4 * lane);
            }
        }
        return st;
    };
    // rows of a mic: [s copy 0][s copy 1] ([d copy 0][d copy 1]); copy c holds sample i - c at position i; position i = floats 2 i, 2 i + 1
    auto write_row = [&](float* row, const float4 x0, const float4 x1, float p0, float p1, bool shifted, int lane) {
        float4* q = reinterpret_cast<float4*>(row + 2 * LEAD) + 2 * lane;
        if (!shifted) {
            q[0] = make_float4(x0.x, x1.x, x0.y, x1.y);
            q[1] = make_float4(x0.z, x1.z, x0.w, x1.w);
        } else {
            q[0] = make_float4(p0, p1, x0.x, x1.x);
            q[1] = make_float4(x0.y, x1.y, x0.z, x1.z);
        }
    };
    auto stage = [&](int h, const Staged2& st, bool wipe) {
        int lane = lane_;                                       // (opaque copy: the per-lane addresses are recomputed here, not hoisted)
        asm volatile("" : "+v"(lane));
        float* slot = lds + ((h & 1) * HC + my_mic) * G::kSlot;
        float4 x0 = st.v0, x1 = st.v1;
        float* rows;                                            // the two rows this wave writes
        bool both_copies = true;
        if constexpr (kLerp) {
            rows = slot + part * C * RS;
            if (part == 1) {
                // D[i] = s[i+1] - s[i], the reference's own subtraction (lerp_and_sum.c:54); D[-1] stays 0 (prefix)
                const float n0 = dpp_next(x0.x), n1 = dpp_next(x1.x);
                x0 = make_float4(x0.y - x0.x, x0.z - x0.y, x0.w - x0.z, n0 - x0.w);
                x1 = make_float4(x1.y - x1.x, x1.z - x1.y, x1.w - x1.z, n1 - x1.w);
            }
        } else {
            rows = slot + part * RS;                            // pad: one copy per wave
            both_copies = false;
        }
        const float p0 = dpp_prev(x0.w), p1 = dpp_prev(x1.w);   // the previous lane's last sample (0 in lane 0: the prefix)
        if (both_copies) {
            write_row(rows, x0, x1, p0, p1, false, lane);
            write_row(rows + RS, x0, x1, p0, p1, true, lane);
        } else {
            write_row(rows, x0, x1, p0, p1, part == 1, lane);
        }
        if (wipe) {
            // the zero prefix (56 samples x 2 frames = 28 quads per row): only the parked rows of the power pass overwrite it
            constexpr int PQ = LEAD >> 1;
            static_assert(2 * PQ <= kWave, "one lane per prefix quad of two rows");
            if (both_copies) {
                if (lane < 2 * PQ) reinterpret_cast<float4*>(rows + (lane / PQ) * RS)[lane % PQ] = make_float4(0.f, 0.f, 0.f, 0.f);
            } else {
                if (lane < PQ) reinterpret_cast<float4*>(rows)[lane] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    };

    BF_STAMP_DECL
    Staged2 st = fetch(0);
    const int lb = 16 * lane + (int)(unsigned)(size_t)((__attribute__((address_space(3))) char*)lds);

    for (int g0 = tile_begin; g0 < tile_end; g0 += kGroup) {
        f32x2 acc[DW][4];                                       // (frame 0, frame 1) of samples 2l, 2l+1, 128+2l, 128+2l+1
#pragma unroll
        for (int j = 0; j < DW; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[j][q] = f32x2{0.0f, 0.0f};

        BF_STAMP(7);
        __syncthreads();   // the previous group's parked rows have been summed
        BF_STAMP(1);
        stage(0, st, true);
        st = fetch(1 % n_half);
        BF_STAMP(2);
        __syncthreads();
        BF_STAMP(3);

        const int dw0 = g0 + wave * DW;                         // wave-uniform
        const bool busy = dw0 < tile_end;
        const size_t grp = busy ? (size_t)(dw0 - a.dir_begin) / DW : 0;
        for (int h = 0; h < n_half; ++h) {
            if (h + 1 < n_half) {
                stage(h + 1, st, h == 0);                       // into the half whose sweeps ended before the last barrier
                if (h + 2 < n_half) st = fetch(h + 2);
                else if (g0 + kGroup < tile_end) st = fetch(0);
                BF_STAMP(2);
            }
            if (busy) {
                const int32_t* __restrict__ et = dig + (grp * M + (size_t)h * HC) * DW;
                const float* __restrict__ ht = reinterpret_cast<const float*>(dig) + a.digest_h_off + (grp * M + (size_t)h * HC) * DW;
                struct Entries { int e[DW]; unsigned long long hp[DW / 2]; };
                auto request = [&](Entries& t, int m) {
                    // (reads past the half's last mic stay inside the slack-padded table and are dropped)
#pragma unroll
                    for (int j = 0; j < DW; ++j) t.e[j] = et[m * DW + j];
#pragma unroll
                    for (int j = 0; j < DW / 2; ++j) {
                        t.hp[j] = 0;
                        if constexpr (kLerp) t.hp[j] = *reinterpret_cast<const unsigned long long*>(ht + m * DW + 2 * j);
                    }
                };
                Entries E[3];
                request(E[0], 0);
                request(E[1], 1);
                auto mic = [&](int m, auto kc) {
                    constexpr int K = decltype(kc)::value, K2 = (K + 2) % 3;
                    const Entries& cur = E[K];
                    pair2_first<ALGO>(acc, cur.e[0], cur.hp[0], lb);
                    request(E[K2], m + 2);      // after the first statement's wait, so that it does not sit on these loads
                    pair2_rest<ALGO>(acc, cur.e, cur.hp, lb);
                };
                using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
                static_assert(HC == 8, "eight mics: two trips of three and two more");
#pragma unroll 1
                for (int t = 0; t < 2; ++t) {
                    mic(0, I0{}); mic(1, I1{}); mic(2, I2{});
                    et += 3 * DW; ht += 3 * DW;
                }
                mic(0, I0{});
                mic(1, I1{});
                __builtin_amdgcn_s_waitcnt(0xC07F);             // the entries requested past the half's end have landed (and are dropped)
            }
            BF_STAMP(0);       // sweep -> waiting for the others
            __syncthreads();   // half h is free, half h + 1 is staged
            BF_STAMP(h + 1 < n_half ? 3 : 4);
        }

        // ---- k-ordered mean power (pad_and_sum.c:120-128), one frame at a time: the 16 waves park the squared means of their
        // directions (row = direction, k in order; the rows alias the LDS image), then one direction per lane runs the sequential sum.
#pragma unroll
        for (int f = 0; f < 2; ++f) {
            if (f == 1) {
                __syncthreads();        // frame 0's rows have been summed
                BF_STAMP(4);
            }
            auto park = [&](auto mul_c) __attribute__((always_inline)) {
#pragma unroll
                for (int j = 0; j < DW; ++j) {
                    float* row = lds + (wave * DW + j) * kPark;
                    const float x0 = f == 0 ? acc[j][0].x : acc[j][0].y, x1 = f == 0 ? acc[j][1].x : acc[j][1].y;
                    const float x2 = f == 0 ? acc[j][2].x : acc[j][2].y, x3 = f == 0 ? acc[j][3].x : acc[j][3].y;
                    float o0, o1, o2, o3;
                    if constexpr (decltype(mul_c)::value) {
                        o0 = x0 * a.inv_n; o1 = x1 * a.inv_n; o2 = x2 * a.inv_n; o3 = x3 * a.inv_n;
                    } else {
                        float fm = (float)M;
                        asm volatile("" : "+v"(fm));   // not speculatable: keeps this path behind its branch
                        o0 = x0 / fm; o1 = x1 / fm; o2 = x2 / fm; o3 = x3 / fm;
                    }
                    reinterpret_cast<float2*>(row)[lane] = make_float2(o0 * o0, o1 * o1);             // samples 2l, 2l+1
                    reinterpret_cast<float2*>(row + 128)[lane] = make_float2(o2 * o2, o3 * o3);       // samples 128+2l, 128+2l+1
                }
            };
            if (__builtin_expect(a.n_is_pow2, 1)) park(std::true_type{}); else park(std::false_type{});
            BF_STAMP(5);                // -> waiting
            __syncthreads();
            BF_STAMP(6);                // -> ordered sum (two waves; the others go on to the next barrier)
            int lane_o = lane;                            // (opaque: keeps the per-lane row address out of the registers the sweep needs)
            asm volatile("" : "+v"(lane_o));
            const int g = wave * kWave + lane_o;          // parked row of this lane
            const int d = g0 + g;
            if (g < kGroup && d < tile_end && (f == 0 || two)) {
                const float* row = lds + g * kPark;
                const float4* row4 = reinterpret_cast<const float4*>(row);
                float sum = 0.0f;
                int k = 0;
                for (; k + 32 <= N; k += 32) {
                    float4 v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) v[u] = row4[(k >> 2) + u];
#pragma unroll
                    for (int u = 0; u < 8; ++u) { sum += v[u].x; sum += v[u].y; sum += v[u].z; sum += v[u].w; }
                }
                for (; k < N; ++k) sum += row[k];
                (f == 0 ? img0 : img1)[d - a.image_origin] = sum / (float)N;
            }
        }
    }
    BF_STAMP(7);
    BF_STAMP_FLUSH;
}

// ==================================================================================================
// Hybrid beamformer (integer delay + 8-tap fractional FIR, hybrid_convolve_and_sum.c:51-121), two frames per workgroup.
//
// das_copies_kernel<hybrid> walks direction-outer: per (direction, mic) three ds_read_b128, two tap loads and the guard masks for
// 32 v_fmac_f32 -- and a plain (unpacked) VALU instruction holds its SIMD as long as a packed one (scripts/dev/fmac_probe.hip:
// v_fmac_f32 sustains 47 % of the fp32 peak, v_pk_fma_f32 91 %, even as one dependent chain).  Here:
//   * the two frames of a workgroup are INTERLEAVED sample by sample in the LDS rows (f0 s0, f1 s0, f0 s1, f1 s1, ..), so a
//     window read lands as register pairs (frame 0, frame 1) and every multiply-accumulate is a v_pk_fma_f32 with the tap as
//     its scalar operand: 32 packed instructions per (direction, mic) for both frames' 64 multiply-accumulates;
//   * the WINDOW a (direction, mic) reads depends on the whole-sample delay only, which neighbouring directions mostly share
//     (the structure the pad / lerp sweep lives on), only the taps differ: mic-outer sweep over the wave's 8 directions, the
//     11-sample windows re-read only when the LDS offset changes (scalar test, out-of-line stub);
//   * the 8 taps of a step arrive in SGPRs (s_load_dwordx8 two steps ahead); the guard ("output k receives nothing for k <= p")
//     is four EXEC masks, rebuilt only together with the window;
//   * interleaved rows need only TWO shifted copies for 16-byte-aligned window reads (the window start must fall on an even
//     sample), so the LDS image holds 32 mics: two halves of 16, staged under the sweep as in das_pair_kernel.
// Operation order per output is the reference's (taps 0..7 in order, mics in order): bit-identical maps.
struct HybridGeo {
    static constexpr int kC = 2, kLead = Geo<1>::kLead, kRs = 2 * Geo<1>::kRsFir;   // floats per row: two frames interleaved
    static constexpr int kHalf = 16, kMc = 2 * kHalf;
    static constexpr int kSlot = kC * kRs;               // floats per staged mic (both frames)
};

// The guard of a (direction, mic): output j of a lane (sample 4 lane + j) is live on the lanes n_j .. 63 (four bytes of `ng`,
// digest_grouped_kernel; n < 64 under the fixed row stride).  It depends on the whole-sample delay only -- like the window.
struct HybridMasks { unsigned long long m[4]; };
#define BF_H_MASKS                                                                                          \
    "s_lshl_b64 %[m0], -1, %[ng]\n\t"                                                                       \
    "s_lshr_b32 %[t], %[ng], 8\n\ts_lshl_b64 %[m1], -1, %[t]\n\t"                                            \
    "s_lshr_b32 %[t], %[ng], 16\n\ts_lshl_b64 %[m2], -1, %[t]\n\t"                                           \
    "s_lshr_b32 %[t], %[ng], 24\n\ts_lshl_b64 %[m3], -1, %[t]\n\t"
#define BF_H_READS                                                                                          \
    "ds_read_b128 %[q0], %[ad] offset:0\n\tds_read_b128 %[q1], %[ad] offset:16\n\tds_read_b128 %[q2], %[ad] offset:32\n\t" \
    "ds_read_b128 %[q3], %[ad] offset:48\n\tds_read_b128 %[q4], %[ad] offset:64\n\tds_read_b128 %[q5], %[ad] offset:80\n\t"
#define BF_H_QOPS [q0] "+v"(Q[0]), [q1] "+v"(Q[1]), [q2] "+v"(Q[2]), [q3] "+v"(Q[3]), [q4] "+v"(Q[4]), [q5] "+v"(Q[5])
// (re)read the window (11 samples x 2 frames + 1 = 6 quads) for LDS byte offset ec unless it equals ep
__device__ __forceinline__ void hybrid_window(f32x4 (&Q)[6], HybridMasks& k, int ep, int ec, int ng, int lbase)
{
    int ad, t;
    asm volatile("s_cmp_lg_u32 %[ec], %[ep]\n\ts_cbranch_scc1 .Lr_%=\n.Lb_%=:\n\t"
                 ".subsection 1\n.Lr_%=:\n\tv_add_u32 %[ad], %[ec], %[lb]\n\t" BF_H_READS BF_H_MASKS
                 "s_waitcnt lgkmcnt(0)\n\ts_branch .Lb_%=\n\t.subsection 0"
                 : BF_H_QOPS, [ad] "=&v"(ad), [m0] "+s"(k.m[0]), [m1] "+s"(k.m[1]), [m2] "+s"(k.m[2]), [m3] "+s"(k.m[3]), [t] "=&s"(t)
                 : [ep] "s"(ep), [ec] "s"(ec), [ng] "s"(ng), [lb] "v"(lbase)
                 : "scc");
}
// a mic's first window: read in place (no test)
__device__ __forceinline__ void hybrid_window_first(f32x4 (&Q)[6], HybridMasks& k, int ec, int ng, int lbase)
{
    int ad, t;
    asm volatile("v_add_u32 %[ad], %[ec], %[lb]\n\t" BF_H_READS BF_H_MASKS "s_waitcnt lgkmcnt(0)"
                 : BF_H_QOPS, [ad] "=&v"(ad), [m0] "=&s"(k.m[0]), [m1] "=&s"(k.m[1]), [m2] "=&s"(k.m[2]), [m3] "=&s"(k.m[3]), [t] "=&s"(t)
                 : [ec] "s"(ec), [ng] "s"(ng), [lb] "v"(lbase)
                 : "scc");
}
#undef BF_H_MASKS
#undef BF_H_READS
#undef BF_H_QOPS

// One (direction, mic) step for both frames: output j of a lane takes o_j = fma(h_t, W[j + t], o_j), t = 0..7 in order
// (hybrid_convolve_and_sum.c:58-63) under the guard mask of output j; o_j and W[.] are (frame 0, frame 1) pairs, the taps come
// as four aligned SGPR pairs (even tap: low half for both lanes, odd tap: high half).
#define BF_H_E(j, hh, w) "v_pk_fma_f32 %[o" #j "], %[" #hh "], %[w" #w "], %[o" #j "] op_sel_hi:[0,1,1]\n\t"
#define BF_H_O(j, hh, w) "v_pk_fma_f32 %[o" #j "], %[" #hh "], %[w" #w "], %[o" #j "] op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
#define BF_H_OUT(j, w0, w1, w2, w3, w4, w5, w6, w7)                                                          \
    BF_H_E(j, h01, w0) BF_H_O(j, h01, w1) BF_H_E(j, h23, w2) BF_H_O(j, h23, w3) BF_H_E(j, h45, w4) BF_H_O(j, h45, w5) BF_H_E(j, h67, w6) BF_H_O(j, h67, w7)
__device__ __forceinline__ void hybrid_step(f32x2 (&o)[4], const f32x4 (&Q)[6], const unsigned long long (&h)[4], const HybridMasks& k)
{
    // W[t] = (frame 0, frame 1) of the window's sample t: the two halves of the quads
    const f32x2 w0 = __builtin_shufflevector(Q[0], Q[0], 0, 1), w1 = __builtin_shufflevector(Q[0], Q[0], 2, 3);
    const f32x2 w2 = __builtin_shufflevector(Q[1], Q[1], 0, 1), w3 = __builtin_shufflevector(Q[1], Q[1], 2, 3);
    const f32x2 w4 = __builtin_shufflevector(Q[2], Q[2], 0, 1), w5 = __builtin_shufflevector(Q[2], Q[2], 2, 3);
    const f32x2 w6 = __builtin_shufflevector(Q[3], Q[3], 0, 1), w7 = __builtin_shufflevector(Q[3], Q[3], 2, 3);
    const f32x2 w8 = __builtin_shufflevector(Q[4], Q[4], 0, 1), w9 = __builtin_shufflevector(Q[4], Q[4], 2, 3);
    const f32x2 w10 = __builtin_shufflevector(Q[5], Q[5], 0, 1);
    asm volatile(
        // (EXEC is all ones on entry -- 1024-thread workgroups, the sweep sits behind wave-uniform branches only -- so it is
        //  restored from the constant instead of being saved first: one scalar move and one SGPR pair less per step)
        "s_mov_b64 exec, %[m0]\n\t"
        BF_H_OUT(0, 0, 1, 2, 3, 4, 5, 6, 7)
        "s_mov_b64 exec, %[m1]\n\t"
        BF_H_OUT(1, 1, 2, 3, 4, 5, 6, 7, 8)
        "s_mov_b64 exec, %[m2]\n\t"
        BF_H_OUT(2, 2, 3, 4, 5, 6, 7, 8, 9)
        "s_mov_b64 exec, %[m3]\n\t"
        BF_H_OUT(3, 3, 4, 5, 6, 7, 8, 9, 10)
        "s_mov_b64 exec, -1"
        : [o0] "+v"(o[0]), [o1] "+v"(o[1]), [o2] "+v"(o[2]), [o3] "+v"(o[3])
        : [h01] "s"(h[0]), [h23] "s"(h[1]), [h45] "s"(h[2]), [h67] "s"(h[3]), [m0] "s"(k.m[0]), [m1] "s"(k.m[1]), [m2] "s"(k.m[2]), [m3] "s"(k.m[3]),
          [w0] "v"(w0), [w1] "v"(w1), [w2] "v"(w2), [w3] "v"(w3), [w4] "v"(w4), [w5] "v"(w5), [w6] "v"(w6), [w7] "v"(w7), [w8] "v"(w8), [w9] "v"(w9),
          [w10] "v"(w10));
}
// The plain 8-tap FIR (convolve_and_sum.c:197-211, mimo_convolve_naive): the same chains without a guard
__device__ __forceinline__ void fir_naive_step(f32x2 (&o)[4], const f32x4 (&Q)[6], const unsigned long long (&h)[4])
{
    const f32x2 w0 = __builtin_shufflevector(Q[0], Q[0], 0, 1), w1 = __builtin_shufflevector(Q[0], Q[0], 2, 3);
    const f32x2 w2 = __builtin_shufflevector(Q[1], Q[1], 0, 1), w3 = __builtin_shufflevector(Q[1], Q[1], 2, 3);
    const f32x2 w4 = __builtin_shufflevector(Q[2], Q[2], 0, 1), w5 = __builtin_shufflevector(Q[2], Q[2], 2, 3);
    const f32x2 w6 = __builtin_shufflevector(Q[3], Q[3], 0, 1), w7 = __builtin_shufflevector(Q[3], Q[3], 2, 3);
    const f32x2 w8 = __builtin_shufflevector(Q[4], Q[4], 0, 1), w9 = __builtin_shufflevector(Q[4], Q[4], 2, 3);
    const f32x2 w10 = __builtin_shufflevector(Q[5], Q[5], 0, 1);
    asm volatile(
        BF_H_OUT(0, 0, 1, 2, 3, 4, 5, 6, 7) BF_H_OUT(1, 1, 2, 3, 4, 5, 6, 7, 8) BF_H_OUT(2, 2, 3, 4, 5, 6, 7, 8, 9) BF_H_OUT(3, 3, 4, 5, 6, 7, 8, 9, 10)
        : [o0] "+v"(o[0]), [o1] "+v"(o[1]), [o2] "+v"(o[2]), [o3] "+v"(o[3])
        : [h01] "s"(h[0]), [h23] "s"(h[1]), [h45] "s"(h[2]), [h67] "s"(h[3]),
          [w0] "v"(w0), [w1] "v"(w1), [w2] "v"(w2), [w3] "v"(w3), [w4] "v"(w4), [w5] "v"(w5), [w6] "v"(w6), [w7] "v"(w7), [w8] "v"(w8), [w9] "v"(w9),
          [w10] "v"(w10));
}
// The AVX2 flavour (convolve_and_sum.c:158-192 + sum8 :132-153, mimo_convolve_vectorized) with one block of 8 taps: eight plain
// products p_t = x[t] * h[t] (the fma lanes start from 0), then the fixed tree ((p0 + p4) + (p2 + p6)) + ((p1 + p5) + (p3 + p7)), out +=.
#define BF_V_MUL(p, hh, w, MODS) "v_pk_mul_f32 %[" #p "], %[" #hh "], %[w" #w "] " MODS "\n\t"
#define BF_V_OUT(j, w0, w1, w2, w3, w4, w5, w6, w7)                                                                             \
    BF_V_MUL(p0, h01, w0, "op_sel_hi:[0,1]") BF_V_MUL(p1, h01, w1, "op_sel:[1,0] op_sel_hi:[1,1]")                                \
    BF_V_MUL(p2, h23, w2, "op_sel_hi:[0,1]") BF_V_MUL(p3, h23, w3, "op_sel:[1,0] op_sel_hi:[1,1]")                                \
    BF_V_MUL(p4, h45, w4, "op_sel_hi:[0,1]") BF_V_MUL(p5, h45, w5, "op_sel:[1,0] op_sel_hi:[1,1]")                                \
    BF_V_MUL(p6, h67, w6, "op_sel_hi:[0,1]") BF_V_MUL(p7, h67, w7, "op_sel:[1,0] op_sel_hi:[1,1]")                                \
    "v_pk_add_f32 %[p0], %[p0], %[p4]\n\tv_pk_add_f32 %[p1], %[p1], %[p5]\n\tv_pk_add_f32 %[p2], %[p2], %[p6]\n\tv_pk_add_f32 %[p3], %[p3], %[p7]\n\t" \
    "v_pk_add_f32 %[p0], %[p0], %[p2]\n\tv_pk_add_f32 %[p1], %[p1], %[p3]\n\tv_pk_add_f32 %[p0], %[p0], %[p1]\n\t"                   \
    "v_pk_add_f32 %[o" #j "], %[o" #j "], %[p0]\n\t"
__device__ __forceinline__ void fir_vec_step(f32x2 (&o)[4], const f32x4 (&Q)[6], const unsigned long long (&h)[4])
{
    const f32x2 w0 = __builtin_shufflevector(Q[0], Q[0], 0, 1), w1 = __builtin_shufflevector(Q[0], Q[0], 2, 3);
    const f32x2 w2 = __builtin_shufflevector(Q[1], Q[1], 0, 1), w3 = __builtin_shufflevector(Q[1], Q[1], 2, 3);
    const f32x2 w4 = __builtin_shufflevector(Q[2], Q[2], 0, 1), w5 = __builtin_shufflevector(Q[2], Q[2], 2, 3);
    const f32x2 w6 = __builtin_shufflevector(Q[3], Q[3], 0, 1), w7 = __builtin_shufflevector(Q[3], Q[3], 2, 3);
    const f32x2 w8 = __builtin_shufflevector(Q[4], Q[4], 0, 1), w9 = __builtin_shufflevector(Q[4], Q[4], 2, 3);
    const f32x2 w10 = __builtin_shufflevector(Q[5], Q[5], 0, 1);
    f32x2 p0, p1, p2, p3, p4, p5, p6, p7;
    asm volatile(
        BF_V_OUT(0, 0, 1, 2, 3, 4, 5, 6, 7) BF_V_OUT(1, 1, 2, 3, 4, 5, 6, 7, 8) BF_V_OUT(2, 2, 3, 4, 5, 6, 7, 8, 9) BF_V_OUT(3, 3, 4, 5, 6, 7, 8, 9, 10)
        : [o0] "+v"(o[0]), [o1] "+v"(o[1]), [o2] "+v"(o[2]), [o3] "+v"(o[3]), [p0] "=&v"(p0), [p1] "=&v"(p1), [p2] "=&v"(p2), [p3] "=&v"(p3),
          [p4] "=&v"(p4), [p5] "=&v"(p5), [p6] "=&v"(p6), [p7] "=&v"(p7)
        : [h01] "s"(h[0]), [h23] "s"(h[1]), [h45] "s"(h[2]), [h67] "s"(h[3]),
          [w0] "v"(w0), [w1] "v"(w1), [w2] "v"(w2), [w3] "v"(w3), [w4] "v"(w4), [w5] "v"(w5), [w6] "v"(w6), [w7] "v"(w7), [w8] "v"(w8), [w9] "v"(w9),
          [w10] "v"(w10));
}
#undef BF_V_MUL
#undef BF_V_OUT
#undef BF_H_OUT
#undef BF_H_E
#undef BF_H_O
// a mic's window for the plain FIRs: read in place, no guard
__device__ __forceinline__ void fir_window_first(f32x4 (&Q)[6], int ec, int lbase)
{
    int ad;
    asm volatile("v_add_u32 %[ad], %[ec], %[lb]\n\t"
                 "ds_read_b128 %[q0], %[ad] offset:0\n\tds_read_b128 %[q1], %[ad] offset:16\n\tds_read_b128 %[q2], %[ad] offset:32\n\t"
                 "ds_read_b128 %[q3], %[ad] offset:48\n\tds_read_b128 %[q4], %[ad] offset:64\n\tds_read_b128 %[q5], %[ad] offset:80\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : [q0] "=&v"(Q[0]), [q1] "=&v"(Q[1]), [q2] "=&v"(Q[2]), [q3] "=&v"(Q[3]), [q4] "=&v"(Q[4]), [q5] "=&v"(Q[5]), [ad] "=&v"(ad)
                 : [ec] "s"(ec), [lb] "v"(lbase));
}

template <int ALGO>
__global__ void __launch_bounds__(1024, 4) das_hybrid_pair_kernel(BF_TABLE_PARAMS, KArgs a)
{
    constexpr bool kHybrid = ALGO == ALGO_HYBRID;
    using G = HybridGeo;
    constexpr int C = G::kC, RS = G::kRs, LEAD = G::kLead, HC = G::kHalf, W = 16, DW = 8, kGroup = DW * W, kPark = Geo<1>::kPark;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    int tile, fpair;
    tile_and_frame(a, &tile, &fpair);
    const int f0 = 2 * fpair;
    const bool two = f0 + 1 < a.n_frames;                      // an odd frame count: the last workgroup row computes its frame twice
    const int f1 = two ? f0 + 1 : f0;
    const int tile_begin = a.dir_begin + tile * a.tile_dirs;
    if (tile_begin >= a.dir_end) return;
    const int tile_end = min(tile_begin + a.tile_dirs, a.dir_end);
    const int M = a.n_mics, N = a.n_samples;                   // M % 16 == 0, N % 4 == 0, N <= 256, 8 taps (plan_das)
    const int n_half = M / HC;
    const float* __restrict__ sig0 = signals + (size_t)f0 * a.m_total * N;
    const float* __restrict__ sig1 = signals + (size_t)f1 * a.m_total * N;
    float* __restrict__ img0 = images + (size_t)f0 * a.image_stride;
    float* __restrict__ img1 = images + (size_t)f1 * a.image_stride;
    const int32_t* __restrict__ dig = reinterpret_cast<const int32_t*>(frac);   // the digest (offsets, guards, regrouped taps) rides in the `frac` slot

    // staging: wave w stages mic w of every half, both frames.  Lane c holds half c's mic id (first 64 halves).
    const int vmic = (lane < n_half) ? mics[lane * HC + wave] : 0;
    struct Staged2 { float4 v0, v1; };
    auto fetch = [&](int h) -> Staged2 {
        const int mic = (h < kWave) ? __builtin_amdgcn_readlane(vmic, h) : mics[h * HC + wave];
        Staged2 st;
        st.v0 = make_float4(0.f, 0.f, 0.f, 0.f);
        st.v1 = st.v0;
        if (4 * lane < N) {
            if constexpr (ALGO == ALGO_FIR_VEC) {
                // (this flavour's eight product registers leave no room for two hoisted 64-bit lane pointers: scalar row base + one lane offset)
                unsigned voff = 16u * (unsigned)lane;
                asm volatile("" : "+v"(voff));
                st.v0 = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(sig0 + (size_t)mic * N) + voff);
                st.v1 = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(sig1 + (size_t)mic * N) + voff);
            } else {
                st.v0 = *reinterpret_cast<const float4*>(sig0 + (size_t)mic * N + 4 * lane);
                st.v1 = *reinterpret_cast<const float4*>(sig1 + (size_t)mic * N + 4 * lane);
            }
        }
        return st;
    };
    auto stage = [&](int h, const Staged2& st, bool wipe) {
        float* row0 = lds + (((h & 1) * HC) + wave) * G::kSlot;        // copy 0; copy 1 (shifted right by one sample) RS floats on
        const float4 a0 = st.v0, a1 = st.v1;
        float4* c0 = reinterpret_cast<float4*>(row0 + 2 * LEAD) + 2 * lane;
        c0[0] = make_float4(a0.x, a1.x, a0.y, a1.y);
        c0[1] = make_float4(a0.z, a1.z, a0.w, a1.w);
        const float p0 = dpp_prev(a0.w), p1 = dpp_prev(a1.w);           // the previous lane's last sample (0 in lane 0: the prefix)
        float4* c1 = reinterpret_cast<float4*>(row0 + RS + 2 * LEAD) + 2 * lane;
        c1[0] = make_float4(p0, p1, a0.x, a1.x);
        c1[1] = make_float4(a0.y, a1.y, a0.z, a1.z);
        // the zero padding after the block (hybrid_convolve_and_sum.c:98-104), 8 samples = 4 quads per copy; copy 1 still holds the last sample
        float zf = 0.0f;
        asm volatile("" : "+v"(zf));                            // (a zero made here: hipcc otherwise keeps a zero quad live through the sweep -- or spills it)
        if (lane == kWave - 1) {
            const float4 z = make_float4(zf, zf, zf, zf);
            float4* t0 = reinterpret_cast<float4*>(row0 + 2 * (LEAD + 256));
            t0[0] = z; t0[1] = z; t0[2] = z; t0[3] = z;
            float4* t1 = reinterpret_cast<float4*>(row0 + RS + 2 * (LEAD + 256));
            t1[0] = make_float4(a0.w, a1.w, 0.f, 0.f); t1[1] = z; t1[2] = z; t1[3] = z;
        }
        if (wipe) {
            // the zero prefix (56 samples x 2 frames = 28 quads per copy): only the parked rows of the power pass overwrite it
            static_assert((LEAD >> 1) * C <= kWave, "one lane per prefix quad");
            constexpr int PQ = LEAD >> 1;
            if (lane < PQ * C) reinterpret_cast<float4*>(row0 + (lane / PQ) * RS)[lane % PQ] = make_float4(zf, zf, zf, zf);
        }
    };

    Staged2 st = fetch(0);
    const int lb = 32 * lane + (int)(unsigned)(size_t)((__attribute__((address_space(3))) char*)lds);   // a lane's window starts 4 samples x 2 frames on

    for (int g0 = tile_begin; g0 < tile_end; g0 += kGroup) {
        f32x2 acc[DW][4];                                       // (frame 0, frame 1) of output j of direction d
#pragma unroll
        for (int j = 0; j < DW; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[j][q] = f32x2{0.0f, 0.0f};

        __syncthreads();   // the previous group's parked rows have been summed
        stage(0, st, true);
        st = fetch(1 % n_half);
        __syncthreads();

        const int dw0 = g0 + wave * DW;                         // wave-uniform
        const bool busy = dw0 < tile_end;
        const size_t grp = busy ? (size_t)(dw0 - a.dir_begin) / DW : 0;
        for (int h = 0; h < n_half; ++h) {
            if (h + 1 < n_half) {
                stage(h + 1, st, h == 0);
                if (h + 2 < n_half) st = fetch(h + 2);
                else if (g0 + kGroup < tile_end) st = fetch(0);
            }
            if (busy) {
                const int m0 = h * HC;
                const int32_t* __restrict__ et = dig + (kHybrid ? (grp * M + m0) * DW : 0);       // LDS offsets, 8 per mic (hybrid only)
                const int32_t* __restrict__ nt = dig + (kHybrid ? a.digest_h_off + (grp * M + m0) * DW : 0);   // packed guards, 8 per mic
                // taps regrouped by digest_grouped_kernel: [group of 8 directions][mic][direction][8] -- one pointer per wave
                const unsigned long long* __restrict__ tg =
                    reinterpret_cast<const unsigned long long*>(reinterpret_cast<const float*>(dig) + a.digest_t_off + (grp * M + m0) * DW * 8);
                struct Entries { int e[DW]; int n[DW]; };
                struct Taps { unsigned long long h[4]; };
                auto request = [&](Entries& t, int m) {
#pragma unroll
                    for (int j = 0; j < DW; ++j) { t.e[j] = et[m * DW + j]; t.n[j] = nt[m * DW + j]; }
                };
                auto request_taps = [&](Taps& t, auto jc, int m) {
                    constexpr int j = decltype(jc)::value;
#pragma unroll
                    for (int k = 0; k < 4; ++k) t.h[k] = tg[(m * DW + j) * 4 + k];
                };
                Entries E[2];
                Taps T[3];
                f32x4 Q[6];
                HybridMasks KM;
#pragma unroll
                for (int q = 0; q < 6; ++q) Q[q] = f32x4{0.f, 0.f, 0.f, 0.f};
                KM.m[0] = KM.m[1] = KM.m[2] = KM.m[3] = 0ull;
                if constexpr (kHybrid) request(E[0], 0);
                request_taps(T[0], std::integral_constant<int, 0>{}, 0);
                request_taps(T[1], std::integral_constant<int, 1>{}, 0);
                // steps (m, j) in order; the taps of step s = 8 m + j live in set s % 3 and are requested two steps ahead;
                // 8 % 3 == 2, so the sets of mic m start at (2 m) % 3: three mics until the pattern repeats
                int m0s = 0;                                    // mic (inside the half) the current trip starts at
                auto mic = [&](int m, auto pc, auto rc) {
                    constexpr int P = decltype(pc)::value, R = decltype(rc)::value;    // entry set, first taps set
                    const Entries& cur = E[P];
                    __builtin_amdgcn_s_waitcnt(0xC07F);
                    if constexpr (kHybrid) {
                        hybrid_window_first(Q, KM, cur.e[0], cur.n[0], lb);
                        request(E[P ^ 1], m + 1);               // (past the half's last mic: inside the slack-padded table, dropped)
                    } else {
                        // no whole-sample delay: every direction reads the window that starts T/2 = 4 samples back (copy 0)
                        fir_window_first(Q, (((h & 1) * HC + (m0s + m)) * G::kSlot + 2 * (LEAD - 4)) * 4, lb);
                    }
                    auto stepj = [&](auto jc) {
                        constexpr int j = decltype(jc)::value;
                        // Scalar loads return out of order, so "this step's taps have landed" can only be a full wait -- placed
                        // BEFORE the request for the step after next: what it covers was issued a whole step ago, and the new
                        // request then has this step's arithmetic to land behind.
                        if constexpr (j > 0) __builtin_amdgcn_s_waitcnt(0xC07F);
                        if constexpr (j + 2 < DW) request_taps(T[(R + j + 2) % 3], std::integral_constant<int, j + 2>{}, m);
                        else request_taps(T[(R + j + 2) % 3], std::integral_constant<int, j + 2 - DW>{}, m + 1);
                        if constexpr (kHybrid) {
                            if constexpr (j > 0) hybrid_window(Q, KM, cur.e[j - 1], cur.e[j], cur.n[j], lb);
                            hybrid_step(acc[j], Q, T[(R + j) % 3].h, KM);
                        } else if constexpr (ALGO == ALGO_FIR_NAIVE) {
                            fir_naive_step(acc[j], Q, T[(R + j) % 3].h);
                        } else {
                            fir_vec_step(acc[j], Q, T[(R + j) % 3].h);
                        }
                    };
                    stepj(std::integral_constant<int, 0>{}); stepj(std::integral_constant<int, 1>{});
                    stepj(std::integral_constant<int, 2>{}); stepj(std::integral_constant<int, 3>{});
                    stepj(std::integral_constant<int, 4>{}); stepj(std::integral_constant<int, 5>{});
                    stepj(std::integral_constant<int, 6>{}); stepj(std::integral_constant<int, 7>{});
                };
                using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
                static_assert(HC == 16, "sixteen mics per half");
#pragma unroll 1
                for (int t = 0; t < 2; ++t) {                   // 6 mics per trip: entry sets alternate, taps sets repeat after 3 mics
                    mic(0, I0{}, I0{}); mic(1, I1{}, I2{}); mic(2, I0{}, I1{}); mic(3, I1{}, I0{}); mic(4, I0{}, I2{}); mic(5, I1{}, I1{});
                    et += 6 * DW; nt += 6 * DW; m0s += 6;
                    tg += 6 * DW * 4;
                }
                mic(0, I0{}, I0{}); mic(1, I1{}, I2{}); mic(2, I0{}, I1{}); mic(3, I1{}, I0{});
                __builtin_amdgcn_s_waitcnt(0xC07F);             // the requests made past the half's end have landed (and are dropped)
            }
            __syncthreads();   // half h is free, half h + 1 is staged
        }

        // ---- k-ordered mean power (hybrid_convolve_and_sum.c:108-116), one frame at a time (see das_pair_kernel)
#pragma unroll
        for (int f = 0; f < 2; ++f) {
            if (f == 1) __syncthreads();        // frame 0's rows have been summed
            auto park = [&](auto mul_c) __attribute__((always_inline)) {
#pragma unroll
                for (int j = 0; j < DW; ++j) {
                    float* row = lds + (wave * DW + j) * kPark;
                    const float x0 = f == 0 ? acc[j][0].x : acc[j][0].y, x1 = f == 0 ? acc[j][1].x : acc[j][1].y;
                    const float x2 = f == 0 ? acc[j][2].x : acc[j][2].y, x3 = f == 0 ? acc[j][3].x : acc[j][3].y;
                    float o0, o1, o2, o3;
                    if constexpr (decltype(mul_c)::value) {
                        o0 = x0 * a.inv_n; o1 = x1 * a.inv_n; o2 = x2 * a.inv_n; o3 = x3 * a.inv_n;
                    } else {
                        float fm = (float)M;
                        asm volatile("" : "+v"(fm));   // not speculatable: keeps this path behind its branch
                        o0 = x0 / fm; o1 = x1 / fm; o2 = x2 / fm; o3 = x3 / fm;
                    }
                    reinterpret_cast<float4*>(row)[lane] = make_float4(o0 * o0, o1 * o1, o2 * o2, o3 * o3);
                }
            };
            if (__builtin_expect(a.n_is_pow2, 1)) park(std::true_type{}); else park(std::false_type{});
            __syncthreads();
            int lane_o = lane;                            // (opaque: keeps the per-lane row address out of the registers the sweep needs)
            asm volatile("" : "+v"(lane_o));
            const int g = wave * kWave + lane_o;          // parked row of this lane
            const int d = g0 + g;
            if (g < kGroup && d < tile_end && (f == 0 || two)) {
                const float* row = lds + g * kPark;
                const float4* row4 = reinterpret_cast<const float4*>(row);
                float sum = 0.0f;
                int k = 0;
#pragma unroll 1
                for (; k + 32 <= N; k += 32) {                  // (not unrolled further: frame 1's accumulators are live during frame 0's sum)
                    float4 v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) v[u] = row4[(k >> 2) + u];
#pragma unroll
                    for (int u = 0; u < 8; ++u) { sum += v[u].x; sum += v[u].y; sum += v[u].z; sum += v[u].w; }
                }
                for (; k < N; ++k) sum += row[k];
                (f == 0 ? img0 : img1)[d - a.image_origin] = sum / (float)N;
            }
        }
    }
}

// ==================================================================================================
// Long blocks: pad / lerp at 256 < N <= 1024 (2 or 4 segments of 256 samples per row; BASELINE config 5: 256 mics x 1024).
//
// The sweep of das_copies_kernel with the chunk handling of das_pair_kernel and a conflict-free lane mapping:
//   * a wave carries DW = 8 / 4 directions x NSEG = 2 / 4 segments (64 accumulator registers); per direction step the offset
//     test is shared by 4 NSEG (lerp) packed operations;
//   * the LDS image holds 2 halves of HC = 16 / NSEG mics; while the waves sweep one half each of them stages ONE (mic, segment)
//     pair of the next half into the other -- one barrier per HC mics (the old kernel: two per 4 mics, staging between them with
//     every SIMD idle), and the table-entry pipeline runs on across the barrier;
//   * lane l owns the sample pairs (2l, 2l+1) and (128+2l, 128+2l+1) of a segment: every ds_read_b64 covers 512 contiguous bytes;
//   * nothing is in flight across statements the compiler cannot see (a mic's first quads are read in place), and the kernel
//     must not use scratch (tests/test_isa_hazards.py reads the code-object metadata).
// Mic order and operation order are the reference's; the power is summed in k order from parked rows: bit-identical maps.
// lerp on long rows: a staged row holds sample PAIRS with their differences, (s[2j], s[2j+1], D[2j], D[2j+1]) per 16 bytes, so that one
// ds_read_b128 brings both operand pairs of two packed operations (8 LDS instructions per re-read of a 1024-sample row instead of 16;
// two shifted copies still serve any delay).  The quads live in hard-wired registers -- v[92:123], address v90, products v[124:127] --
// because a 16-byte read needs a 4-register tuple whose halves the packed operations address, which asm operands cannot express.
// A mic is two statements: S1 (address, reads, wait, the first direction) and S2 (the other directions with their offset tests and
// out-of-line re-reads); the compiler's table requests for the mic after next sit between them, and nothing there may name
// v90..v127 (tests/test_isa_hazards.py).  One statement per group of steps also keeps the hazard recogniser's s_nop out of the sweep.
#define BF_L_LO "op_sel_hi:[0,1,1]"
#define BF_L_HI "op_sel:[1,0,0] op_sel_hi:[1,1,1]"
// lerp_and_sum.c:50-56  out[k] += s[i] + h * (s[i+1] - s[i]),  i = k - p - 1   (gcc contracts it into one fma)
#define BF_L_SEG(n, sg, h, sel, S0, D0, S1, D1)                                                      \
    "v_pk_fma_f32 v[124:125], %[" #h "], v[" D0 "], v[" S0 "] " sel "\n\t"                           \
    "v_pk_fma_f32 v[126:127], %[" #h "], v[" D1 "], v[" S1 "] " sel "\n\t"                           \
    "v_pk_add_f32 %[a" #n #sg "0], %[a" #n #sg "0], v[124:125]\n\t"                                  \
    "v_pk_add_f32 %[a" #n #sg "1], %[a" #n #sg "1], v[126:127]\n\t"
#define BF_L_STEP2(n, h, sel) BF_L_SEG(n, 0, h, sel, "92:93", "94:95", "96:97", "98:99") BF_L_SEG(n, 1, h, sel, "100:101", "102:103", "104:105", "106:107")
#define BF_L_STEP4(n, h, sel) BF_L_STEP2(n, h, sel) BF_L_SEG(n, 2, h, sel, "108:109", "110:111", "112:113", "114:115") \
                              BF_L_SEG(n, 3, h, sel, "116:117", "118:119", "120:121", "122:123")
// (step 0 of a mic: each segment waits only for its own two reads -- LDS returns in order, and a counted wait holds whatever else,
//  scalar loads included, is still outstanding: at most that many operations of any kind, hence at most that many reads)
#define BF_L_STEP2_W(n, h, sel) "s_waitcnt lgkmcnt(2)\n\t" BF_L_SEG(n, 0, h, sel, "92:93", "94:95", "96:97", "98:99") \
                                "s_waitcnt lgkmcnt(0)\n\t" BF_L_SEG(n, 1, h, sel, "100:101", "102:103", "104:105", "106:107")
#define BF_L_STEP4_W(n, h, sel) "s_waitcnt lgkmcnt(6)\n\t" BF_L_SEG(n, 0, h, sel, "92:93", "94:95", "96:97", "98:99") \
                                "s_waitcnt lgkmcnt(4)\n\t" BF_L_SEG(n, 1, h, sel, "100:101", "102:103", "104:105", "106:107") \
                                "s_waitcnt lgkmcnt(2)\n\t" BF_L_SEG(n, 2, h, sel, "108:109", "110:111", "112:113", "114:115") \
                                "s_waitcnt lgkmcnt(0)\n\t" BF_L_SEG(n, 3, h, sel, "116:117", "118:119", "120:121", "122:123")
// (the last step of a mic that has a successor in its half: each segment's quads are dead once the step has used them, and the next
//  mic's reads for that segment are issued right behind -- they fly during the rest of the step and the compiler's code between the mics)
#define BF_L_RD(q, off) "ds_read_b128 v[" q "], v90 offset:" #off "\n\t"
#define BF_L_STEP2_PF(n, h, sel) BF_L_SEG(n, 0, h, sel, "92:93", "94:95", "96:97", "98:99") BF_L_RD("92:95", 0) BF_L_RD("96:99", 1024) \
                                 BF_L_SEG(n, 1, h, sel, "100:101", "102:103", "104:105", "106:107") BF_L_RD("100:103", 2048) BF_L_RD("104:107", 3072)
#define BF_L_STEP4_PF(n, h, sel) BF_L_STEP2_PF(n, h, sel)                                                                                \
                                 BF_L_SEG(n, 2, h, sel, "108:109", "110:111", "112:113", "114:115") BF_L_RD("108:111", 4096) BF_L_RD("112:115", 5120) \
                                 BF_L_SEG(n, 3, h, sel, "116:117", "118:119", "120:121", "122:123") BF_L_RD("116:119", 6144) BF_L_RD("120:123", 7168)
#define BF_L_READ2 "ds_read_b128 v[92:95], v90\n\tds_read_b128 v[96:99], v90 offset:1024\n\t"          \
                   "ds_read_b128 v[100:103], v90 offset:2048\n\tds_read_b128 v[104:107], v90 offset:3072\n\t"
#define BF_L_READ4 BF_L_READ2 "ds_read_b128 v[108:111], v90 offset:4096\n\tds_read_b128 v[112:115], v90 offset:5120\n\t" \
                   "ds_read_b128 v[116:119], v90 offset:6144\n\tds_read_b128 v[120:123], v90 offset:7168\n\t"
#define BF_L_ACC2(n, j) [a##n##00] "+v"(acc[j][0][0]), [a##n##01] "+v"(acc[j][0][1]), [a##n##10] "+v"(acc[j][1][0]), [a##n##11] "+v"(acc[j][1][1])
#define BF_L_ACC4(n, j) BF_L_ACC2(n, j), [a##n##20] "+v"(acc[j][2][0]), [a##n##21] "+v"(acc[j][2][1]), [a##n##30] "+v"(acc[j][3][0]), [a##n##31] "+v"(acc[j][3][1])
#define BF_L_CHECK(n, ep, ec) "s_cmp_lg_u32 %[" #ec "], %[" #ep "]\n\ts_cbranch_scc1 .Lr" #n "_%=\n.Lb" #n "_%=:\n\t"
#define BF_L_STUB(n, ec, READ) ".Lr" #n "_%=:\n\tv_add_u32 v90, %[" #ec "], %[lb]\n\t" READ "s_waitcnt lgkmcnt(0)\n\ts_branch .Lb" #n "_%=\n"
#define BF_L_CLOB "v90", "v92", "v93", "v94", "v95", "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", \
                  "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127"
template <int NSEG>
__device__ __forceinline__ void long_lerp_first(f32x2 (&acc)[Geo<NSEG>::kDw][NSEG][2], int e0, unsigned long long h0, int lbase)
{
    if constexpr (NSEG == 2) {
        asm volatile("v_add_u32 v90, %[e0], %[lb]\n\t" BF_L_READ2 BF_L_STEP2_W(0, h0, BF_L_LO) ";BF_S1_END 90 127"
                     : BF_L_ACC2(0, 0) : [e0] "s"(e0), [h0] "s"(h0), [lb] "v"(lbase) : BF_L_CLOB);
    } else {
        asm volatile("v_add_u32 v90, %[e0], %[lb]\n\t" BF_L_READ4 BF_L_STEP4_W(0, h0, BF_L_LO) ";BF_S1_END 90 127"
                     : BF_L_ACC4(0, 0) : [e0] "s"(e0), [h0] "s"(h0), [lb] "v"(lbase) : BF_L_CLOB);
    }
}
// a mic whose reads the previous mic's last step has issued: direction step 0 alone
template <int NSEG>
__device__ __forceinline__ void long_lerp_cont(f32x2 (&acc)[Geo<NSEG>::kDw][NSEG][2], unsigned long long h0)
{
    if constexpr (NSEG == 2) {
        asm volatile(";BF_S2_BEGIN\n\t" BF_L_STEP2_W(0, h0, BF_L_LO) ";BF_S1_END 90 127" : BF_L_ACC2(0, 0) : [h0] "s"(h0) : BF_L_CLOB);
    } else {
        asm volatile(";BF_S2_BEGIN\n\t" BF_L_STEP4_W(0, h0, BF_L_LO) ";BF_S1_END 90 127" : BF_L_ACC4(0, 0) : [h0] "s"(h0) : BF_L_CLOB);
    }
}
// PF: the mic has a successor in its half, whose first offset is `n0`
template <int NSEG, bool PF>
__device__ __forceinline__ void long_lerp_rest(f32x2 (&acc)[Geo<NSEG>::kDw][NSEG][2], const int (&e)[Geo<NSEG>::kDw], const unsigned long long (&h)[Geo<NSEG>::kDw / 2], int lbase,
                                               int n0)
{
#define BF_L_LAST2(n, h) "v_add_u32 v90, %[n0], %[lb]\n\t" BF_L_STEP2_PF(n, h, BF_L_HI) ";BF_S1_END 90 127"
#define BF_L_LAST4(n, h) "v_add_u32 v90, %[n0], %[lb]\n\t" BF_L_STEP4_PF(n, h, BF_L_HI) ";BF_S1_END 90 127"
#define BF_L_STUBS2 ".subsection 1\n" BF_L_STUB(1, e1, BF_L_READ2) BF_L_STUB(2, e2, BF_L_READ2) BF_L_STUB(3, e3, BF_L_READ2) BF_L_STUB(4, e4, BF_L_READ2) \
                    BF_L_STUB(5, e5, BF_L_READ2) BF_L_STUB(6, e6, BF_L_READ2) BF_L_STUB(7, e7, BF_L_READ2) "\t.subsection 0\n\t"
#define BF_L_STUBS4 ".subsection 1\n" BF_L_STUB(1, e1, BF_L_READ4) BF_L_STUB(2, e2, BF_L_READ4) BF_L_STUB(3, e3, BF_L_READ4) "\t.subsection 0\n\t"
#define BF_L_BODY2 BF_L_CHECK(1, e0, e1) BF_L_STEP2(1, h0, BF_L_HI) BF_L_CHECK(2, e1, e2) BF_L_STEP2(2, h1, BF_L_LO) BF_L_CHECK(3, e2, e3) BF_L_STEP2(3, h1, BF_L_HI) \
                   BF_L_CHECK(4, e3, e4) BF_L_STEP2(4, h2, BF_L_LO) BF_L_CHECK(5, e4, e5) BF_L_STEP2(5, h2, BF_L_HI) BF_L_CHECK(6, e5, e6) BF_L_STEP2(6, h3, BF_L_LO) \
                   BF_L_CHECK(7, e6, e7)
#define BF_L_BODY4 BF_L_CHECK(1, e0, e1) BF_L_STEP4(1, h0, BF_L_HI) BF_L_CHECK(2, e1, e2) BF_L_STEP4(2, h1, BF_L_LO) BF_L_CHECK(3, e2, e3)
#define BF_L_IN2 [e0] "s"(e[0]), [e1] "s"(e[1]), [e2] "s"(e[2]), [e3] "s"(e[3]), [e4] "s"(e[4]), [e5] "s"(e[5]), [e6] "s"(e[6]), [e7] "s"(e[7]), \
                 [h0] "s"(h[0]), [h1] "s"(h[1]), [h2] "s"(h[2]), [h3] "s"(h[3]), [lb] "v"(lbase)
#define BF_L_IN4 [e0] "s"(e[0]), [e1] "s"(e[1]), [e2] "s"(e[2]), [e3] "s"(e[3]), [h0] "s"(h[0]), [h1] "s"(h[1]), [lb] "v"(lbase)
#define BF_L_OUT2 BF_L_ACC2(1, 1), BF_L_ACC2(2, 2), BF_L_ACC2(3, 3), BF_L_ACC2(4, 4), BF_L_ACC2(5, 5), BF_L_ACC2(6, 6), BF_L_ACC2(7, 7)
#define BF_L_OUT4 BF_L_ACC4(1, 1), BF_L_ACC4(2, 2), BF_L_ACC4(3, 3)
    if constexpr (NSEG == 2 && !PF) {
        asm volatile(";BF_S2_BEGIN\n\t" BF_L_BODY2 BF_L_STEP2(7, h3, BF_L_HI) BF_L_STUBS2 : BF_L_OUT2 : BF_L_IN2 : "scc", BF_L_CLOB);
    } else if constexpr (NSEG == 2 && PF) {
        asm volatile(";BF_S2_BEGIN\n\t" BF_L_BODY2 BF_L_STUBS2 BF_L_LAST2(7, h3) : BF_L_OUT2 : BF_L_IN2, [n0] "s"(n0) : "scc", BF_L_CLOB);
    } else if constexpr (NSEG == 4 && !PF) {
        asm volatile(";BF_S2_BEGIN\n\t" BF_L_BODY4 BF_L_STEP4(3, h1, BF_L_HI) BF_L_STUBS4 : BF_L_OUT4 : BF_L_IN4 : "scc", BF_L_CLOB);
    } else {
        asm volatile(";BF_S2_BEGIN\n\t" BF_L_BODY4 BF_L_STUBS4 BF_L_LAST4(3, h1) : BF_L_OUT4 : BF_L_IN4, [n0] "s"(n0) : "scc", BF_L_CLOB);
    }
#undef BF_L_LAST2
#undef BF_L_LAST4
#undef BF_L_STUBS2
#undef BF_L_STUBS4
#undef BF_L_BODY2
#undef BF_L_BODY4
#undef BF_L_IN2
#undef BF_L_IN4
#undef BF_L_OUT2
#undef BF_L_OUT4
}
// ---- pad on long rows, the same way: quads (8-byte pairs, no differences) in hard-wired v[92:107], address v90; S1 / continuation /
// S2 statements, counted waits in a mic's first step, the next mic's reads behind the last step's segments.
// pad_and_sum.c:41-47   out[k] += s[k - p]
#define BF_LP_SEG(n, sg, LO, HI) "v_pk_add_f32 %[a" #n #sg "0], %[a" #n #sg "0], v[" LO "]\n\tv_pk_add_f32 %[a" #n #sg "1], %[a" #n #sg "1], v[" HI "]\n\t"
#define BF_LP_RD(q, off) "ds_read_b64 v[" q "], v90 offset:" #off "\n\t"
#define BF_LP_STEP2(n) BF_LP_SEG(n, 0, "92:93", "94:95") BF_LP_SEG(n, 1, "96:97", "98:99")
#define BF_LP_STEP4(n) BF_LP_STEP2(n) BF_LP_SEG(n, 2, "100:101", "102:103") BF_LP_SEG(n, 3, "104:105", "106:107")
#define BF_LP_STEP2_W(n) "s_waitcnt lgkmcnt(2)\n\t" BF_LP_SEG(n, 0, "92:93", "94:95") "s_waitcnt lgkmcnt(0)\n\t" BF_LP_SEG(n, 1, "96:97", "98:99")
#define BF_LP_STEP4_W(n) "s_waitcnt lgkmcnt(6)\n\t" BF_LP_SEG(n, 0, "92:93", "94:95") "s_waitcnt lgkmcnt(4)\n\t" BF_LP_SEG(n, 1, "96:97", "98:99") \
                         "s_waitcnt lgkmcnt(2)\n\t" BF_LP_SEG(n, 2, "100:101", "102:103") "s_waitcnt lgkmcnt(0)\n\t" BF_LP_SEG(n, 3, "104:105", "106:107")
#define BF_LP_STEP2_PF(n) BF_LP_SEG(n, 0, "92:93", "94:95") BF_LP_RD("92:93", 0) BF_LP_RD("94:95", 512) \
                          BF_LP_SEG(n, 1, "96:97", "98:99") BF_LP_RD("96:97", 1024) BF_LP_RD("98:99", 1536)
#define BF_LP_STEP4_PF(n) BF_LP_STEP2_PF(n) BF_LP_SEG(n, 2, "100:101", "102:103") BF_LP_RD("100:101", 2048) BF_LP_RD("102:103", 2560) \
                          BF_LP_SEG(n, 3, "104:105", "106:107") BF_LP_RD("104:105", 3072) BF_LP_RD("106:107", 3584)
#define BF_LP_READ2 BF_LP_RD("92:93", 0) BF_LP_RD("94:95", 512) BF_LP_RD("96:97", 1024) BF_LP_RD("98:99", 1536)
#define BF_LP_READ4 BF_LP_READ2 BF_LP_RD("100:101", 2048) BF_LP_RD("102:103", 2560) BF_LP_RD("104:105", 3072) BF_LP_RD("106:107", 3584)
#define BF_LP_CLOB "v90", "v92", "v93", "v94", "v95", "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107"
template <int NSEG>
__device__ __forceinline__ void long_pad_first(f32x2 (&acc)[Geo<NSEG>::kDw][NSEG][2], int e0, int lbase)
{
    if constexpr (NSEG == 2) {
        asm volatile("v_add_u32 v90, %[e0], %[lb]\n\t" BF_LP_READ2 BF_LP_STEP2_W(0) ";BF_S1_END 90 107" : BF_L_ACC2(0, 0) : [e0] "s"(e0), [lb] "v"(lbase) : BF_LP_CLOB);
    } else {
        asm volatile("v_add_u32 v90, %[e0], %[lb]\n\t" BF_LP_READ4 BF_LP_STEP4_W(0) ";BF_S1_END 90 107" : BF_L_ACC4(0, 0) : [e0] "s"(e0), [lb] "v"(lbase) : BF_LP_CLOB);
    }
}
template <int NSEG>
__device__ __forceinline__ void long_pad_cont(f32x2 (&acc)[Geo<NSEG>::kDw][NSEG][2])
{
    if constexpr (NSEG == 2) {
        asm volatile(";BF_S2_BEGIN\n\t" BF_LP_STEP2_W(0) ";BF_S1_END 90 107" : BF_L_ACC2(0, 0) : : BF_LP_CLOB);
    } else {
        asm volatile(";BF_S2_BEGIN\n\t" BF_LP_STEP4_W(0) ";BF_S1_END 90 107" : BF_L_ACC4(0, 0) : : BF_LP_CLOB);
    }
}
template <int NSEG, bool PF>
__device__ __forceinline__ void long_pad_rest(f32x2 (&acc)[Geo<NSEG>::kDw][NSEG][2], const int (&e)[Geo<NSEG>::kDw], int lbase, int n0)
{
#define BF_LP_STUBS2 ".subsection 1\n" BF_L_STUB(1, e1, BF_LP_READ2) BF_L_STUB(2, e2, BF_LP_READ2) BF_L_STUB(3, e3, BF_LP_READ2) BF_L_STUB(4, e4, BF_LP_READ2) \
                     BF_L_STUB(5, e5, BF_LP_READ2) BF_L_STUB(6, e6, BF_LP_READ2) BF_L_STUB(7, e7, BF_LP_READ2) "\t.subsection 0\n\t"
#define BF_LP_STUBS4 ".subsection 1\n" BF_L_STUB(1, e1, BF_LP_READ4) BF_L_STUB(2, e2, BF_LP_READ4) BF_L_STUB(3, e3, BF_LP_READ4) "\t.subsection 0\n\t"
#define BF_LP_BODY2 BF_L_CHECK(1, e0, e1) BF_LP_STEP2(1) BF_L_CHECK(2, e1, e2) BF_LP_STEP2(2) BF_L_CHECK(3, e2, e3) BF_LP_STEP2(3) BF_L_CHECK(4, e3, e4) BF_LP_STEP2(4) \
                    BF_L_CHECK(5, e4, e5) BF_LP_STEP2(5) BF_L_CHECK(6, e5, e6) BF_LP_STEP2(6) BF_L_CHECK(7, e6, e7)
#define BF_LP_BODY4 BF_L_CHECK(1, e0, e1) BF_LP_STEP4(1) BF_L_CHECK(2, e1, e2) BF_LP_STEP4(2) BF_L_CHECK(3, e2, e3)
#define BF_LP_IN2 [e0] "s"(e[0]), [e1] "s"(e[1]), [e2] "s"(e[2]), [e3] "s"(e[3]), [e4] "s"(e[4]), [e5] "s"(e[5]), [e6] "s"(e[6]), [e7] "s"(e[7]), [lb] "v"(lbase)
#define BF_LP_IN4 [e0] "s"(e[0]), [e1] "s"(e[1]), [e2] "s"(e[2]), [e3] "s"(e[3]), [lb] "v"(lbase)
#define BF_LP_OUT2 BF_L_ACC2(1, 1), BF_L_ACC2(2, 2), BF_L_ACC2(3, 3), BF_L_ACC2(4, 4), BF_L_ACC2(5, 5), BF_L_ACC2(6, 6), BF_L_ACC2(7, 7)
#define BF_LP_OUT4 BF_L_ACC4(1, 1), BF_L_ACC4(2, 2), BF_L_ACC4(3, 3)
    if constexpr (NSEG == 2 && !PF) {
        asm volatile(";BF_S2_BEGIN\n\t" BF_LP_BODY2 BF_LP_STEP2(7) BF_LP_STUBS2 : BF_LP_OUT2 : BF_LP_IN2 : "scc", BF_LP_CLOB);
    } else if constexpr (NSEG == 2 && PF) {
        asm volatile(";BF_S2_BEGIN\n\t" BF_LP_BODY2 BF_LP_STUBS2 "v_add_u32 v90, %[n0], %[lb]\n\t" BF_LP_STEP2_PF(7) ";BF_S1_END 90 107"
                     : BF_LP_OUT2 : BF_LP_IN2, [n0] "s"(n0) : "scc", BF_LP_CLOB);
    } else if constexpr (NSEG == 4 && !PF) {
        asm volatile(";BF_S2_BEGIN\n\t" BF_LP_BODY4 BF_LP_STEP4(3) BF_LP_STUBS4 : BF_LP_OUT4 : BF_LP_IN4 : "scc", BF_LP_CLOB);
    } else {
        asm volatile(";BF_S2_BEGIN\n\t" BF_LP_BODY4 BF_LP_STUBS4 "v_add_u32 v90, %[n0], %[lb]\n\t" BF_LP_STEP4_PF(3) ";BF_S1_END 90 107"
                     : BF_LP_OUT4 : BF_LP_IN4, [n0] "s"(n0) : "scc", BF_LP_CLOB);
    }
#undef BF_LP_STUBS2
#undef BF_LP_STUBS4
#undef BF_LP_BODY2
#undef BF_LP_BODY4
#undef BF_LP_IN2
#undef BF_LP_IN4
#undef BF_LP_OUT2
#undef BF_LP_OUT4
}
#undef BF_LP_SEG
#undef BF_LP_RD
#undef BF_LP_STEP2
#undef BF_LP_STEP4
#undef BF_LP_STEP2_W
#undef BF_LP_STEP4_W
#undef BF_LP_STEP2_PF
#undef BF_LP_STEP4_PF
#undef BF_LP_READ2
#undef BF_LP_READ4
#undef BF_LP_CLOB
#undef BF_L_LO
#undef BF_L_HI
#undef BF_L_SEG
#undef BF_L_STEP2
#undef BF_L_STEP4
#undef BF_L_STEP2_W
#undef BF_L_RD
#undef BF_L_STEP2_PF
#undef BF_L_STEP4_PF
#undef BF_L_STEP4_W
#undef BF_L_READ2
#undef BF_L_READ4
#undef BF_L_ACC2
#undef BF_L_ACC4
#undef BF_L_CHECK
#undef BF_L_STUB
#undef BF_L_CLOB

template <int ALGO, int NSEG>
struct LongGeo {
    static constexpr bool kLerp = ALGO == ALGO_LERP;
    static constexpr int kA = kLerp ? 2 : 1, kC = 2, kDw = Geo<NSEG>::kDw, kHalf = 16 / NSEG, kMc = 2 * kHalf;
    static constexpr int kPark = Geo<NSEG>::kPark;
};

template <int ALGO, int NSEG, int RS>
__global__ void __launch_bounds__(1024, 4) das_long_kernel(BF_TABLE_PARAMS, KArgs a)
{
    using G = LongGeo<ALGO, NSEG>;
    constexpr bool kLerp = G::kLerp;
    constexpr int A = G::kA, C = G::kC, HC = G::kHalf, W = 16, DW = G::kDw, kGroup = DW * W, kPark = G::kPark;
    static_assert(RS == 0 || RS == Geo<NSEG>::kRs, "fixed row stride");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane_ = threadIdx.x & (kWave - 1), lane = lane_;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    int tile, frame;
    tile_and_frame(a, &tile, &frame);
    const int tile_begin = a.dir_begin + tile * a.tile_dirs;
    if (tile_begin >= a.dir_end) return;
    const int tile_end = min(tile_begin + a.tile_dirs, a.dir_end);
    // rs: samples per staged row (lerp: the plan's row stride counts floats, two per sample -- see `stage`)
    const int rs = RS > 0 ? RS : (kLerp ? a.row_stride >> 1 : a.row_stride), lead = RS > 0 ? Geo<NSEG>::kLead : a.lead;
    const int M = a.n_mics, N = a.n_samples;                   // M % HC == 0 (plan_das)
    const int n_half = M / HC;
    const float* __restrict__ frame_sig = signals + (size_t)frame * a.m_total * N;
    float* __restrict__ img = images + (size_t)frame * a.image_stride;
    const int32_t* __restrict__ dig = reinterpret_cast<const int32_t*>(taps);   // the digest rides in the unused `taps` slot
    const int slot_floats = A * C * rs;                        // floats per staged mic

    // This wave stages the pair `wave` of every half: mic (wave / NSEG) of the half, segment (wave % NSEG).
    // Staging is straight-line code (it runs once per half of HC mics beside 4 HC direction steps per wave, and every instruction --
    // a taken branch more than ten times over -- costs the SIMD an issue slot like the sweep's own): N % 4 == 0 (plan_das), the
    // mic id by a scalar load, the two samples DPP cannot reach through one masked load, the prefix wiped at the group's start.
    const int my_mic = wave / NSEG, my_seg = wave % NSEG, k0 = 256 * my_seg;
    // (fetch / stage take the lane id from v_mbcnt in a volatile asm: their per-lane addresses are then recomputed where they are used --
    //  a handful of integer operations per half -- instead of being hoisted out of the mic loop into registers the sweep needs)
    auto lane_id = []() -> int {
        int l;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
        return l;
    };
    // The quad load is unconditional from a clamped address (a load under a lane mask is merged with zeros -- and waited for -- on the
    // spot).  The two samples DPP cannot reach -- the one before the segment and the one after it -- are wave-uniform.
    // lerp: they come by scalar loads, what lands beyond sample N stays (it only reaches outputs k >= N, which the power sum never
    // reads), the quad is pinned to its load registers until `stage`.  pad: lane 0 / lane 63 fetch them with a second vector load
    // and `stage` zeroes what lies outside the block -- its staging is 30 instructions either way, and measured 3.6 % faster so
    // (1089 against 1050 frames/s at cfg5), the leaner form 1 % faster for lerp.
    struct StagedL { float4 v; float before, after; };
    int hf = 0, mic_a = mics[my_mic];                           // the half the next fetch reads, and its mic id (wave-uniform: scalar loads)
    auto fetch_next = [&]() -> StagedL {
        StagedL st;
        const float* src = frame_sig + (size_t)mic_a * N;
        const int lane = lane_id();
        const int k = min(k0 + 4 * lane, N - 4);
        st.v = *reinterpret_cast<const float4*>(src + k);
        if constexpr (kLerp) {
            st.before = src[min(max(k0 - 1, 0), N - 1)];        // (clamped both ways: a segment may begin beyond a short block, N <= k0)
            st.after = src[min(k0 + 256, N - 1)];
        } else {
            // (any address for the lanes between; plain arithmetic: a nested conditional becomes two EXEC-masked branches)
            const int ke = k - (lane == 0 ? 1 : 0) + (lane == 63 ? 4 : 0);
            st.before = st.after = src[min(max(ke, 0), N - 1)];
        }
        hf = hf + 1 < n_half ? hf + 1 : 0;
        mic_a = mics[hf * HC + my_mic];                         // consumed by the next fetch, half a sweep from now
        return st;
    };
    auto stage = [&](int h, const StagedL& st) {
        const int lane = lane_id();
        float* row0 = lds + ((h & 1) * HC + my_mic) * slot_floats;
        const int col = lead + k0;
        float4 v;
        if constexpr (kLerp) {
            // (the loaded quad stays in the registers the load wrote until here: otherwise the compiler rearranges it for the 16-byte
            //  stores below right behind the load -- and waits for it there, half a sweep early)
            f32x4 t = {st.v.x, st.v.y, st.v.z, st.v.w};
            asm volatile("" : "+v"(t));
            v = make_float4(t.x, t.y, t.z, t.w);
        } else {
            const bool in = k0 + 4 * lane < N;
            v = make_float4(in ? st.v.x : 0.0f, in ? st.v.y : 0.0f, in ? st.v.z : 0.0f, in ? st.v.w : 0.0f);
        }
        float pw = dpp_prev(v.w), nx = dpp_next(v.x);          // (0 in lane 0 / lane 63)
        const float before = my_seg > 0 ? st.before : 0.0f, after = k0 + 256 < N ? st.after : 0.0f;
        pw = lane == 0 ? before : pw;
        nx = lane == 63 ? after : nx;
        if constexpr (kLerp) {
            // D[i] = s[i+1] - s[i], the reference's own subtraction (lerp_and_sum.c:54); D[-1] stays 0 (prefix)
            const float4 dq = make_float4(v.y - v.x, v.z - v.y, v.w - v.z, nx - v.w);
            float dw = dpp_prev(dq.w);
            if (my_seg > 0) dw = lane == 0 ? v.x - before : dw;
            // a row of sample pairs with their differences, (s[2j], s[2j+1], D[2j], D[2j+1]): 2 rs floats; copy 1 holds sample i - 1 at position i
            float4* c0 = reinterpret_cast<float4*>(row0 + 2 * col) + 2 * lane;
            c0[0] = make_float4(v.x, v.y, dq.x, dq.y);
            c0[1] = make_float4(v.z, v.w, dq.z, dq.w);
            float4* c1 = reinterpret_cast<float4*>(row0 + 2 * rs + 2 * col) + 2 * lane;
            c1[0] = make_float4(pw, v.x, dw, dq.x);
            c1[1] = make_float4(v.y, v.z, dq.y, dq.z);
        } else {
            write_copies<C>(row0, rs, col, lane, v, 0.0f, 0.0f, pw);
        }
    };
    // the zero prefix of both halves' rows of this wave's mic: only the parked rows of the power pass ever overwrite it
    // (pad: C rows of rs floats; lerp: C rows of 2 rs floats -- `lead` samples are lead / 4 resp. lead / 2 quads)
    auto wipe_prefix = [&]() {
        const int lane = lane_id();
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            float* row0 = lds + (hh * HC + my_mic) * slot_floats;
            for (int q = lane; q < ((A * lead) >> 2); q += kWave) {
#pragma unroll
                for (int c = 0; c < C; ++c) reinterpret_cast<float4*>(row0 + c * A * rs)[q] = z;
            }
        }
    };

    const int lb = (kLerp ? 16 : 8) * lane + (int)(unsigned)(size_t)((__attribute__((address_space(3))) char*)lds);

    for (int g0 = tile_begin; g0 < tile_end; g0 += kGroup) {
        f32x2 acc[DW][NSEG][2];
#pragma unroll
        for (int j = 0; j < DW; ++j)
#pragma unroll
            for (int sg = 0; sg < NSEG; ++sg) { acc[j][sg][0] = f32x2{0.0f, 0.0f}; acc[j][sg][1] = f32x2{0.0f, 0.0f}; }

        __syncthreads();   // the previous group's parked rows have been summed
        if (my_seg == 0) wipe_prefix();
        // (half 0 is fetched here, not under the previous group's last half: five registers less to carry through the power pass, which
        //  is where the kernel used to spill; the exposed load is one in M / HC halves)
        StagedL st = fetch_next();
        stage(0, st);
        st = fetch_next();                                      // half 1 (a single half: half 0 again, for nothing)
        const int dw0 = g0 + wave * DW;                         // wave-uniform
        const bool busy = dw0 < tile_end;                       // this wave has directions in the tile
        const size_t grp = busy ? (size_t)(dw0 - a.dir_begin) / DW : 0;
        const int32_t* __restrict__ et = dig + grp * M * DW;
        const float* __restrict__ ht = reinterpret_cast<const float*>(dig) + a.digest_h_off + grp * M * DW;
        struct Entries { int e[DW]; unsigned long long hp[DW / 2]; };
        auto request = [&](Entries& t, int m) {
            // (reads past the last mic stay inside the slack-padded table and are dropped)
#pragma unroll
            for (int j = 0; j < DW; ++j) t.e[j] = et[m * DW + j];
#pragma unroll
            for (int j = 0; j < DW / 2; ++j) {
                t.hp[j] = 0;
                if constexpr (kLerp) t.hp[j] = *reinterpret_cast<const unsigned long long*>(ht + m * DW + 2 * j);
            }
        };
        // table entries two mics ahead, KE sets in rotation; the pipeline runs on across the halves' barriers
        // (4 mics per half: four sets, a mic's set is m % 4 in every half; 8 mics per half: three sets and a move at the half's end)
        constexpr int KE = HC == 4 ? 4 : 3;
        Entries E[KE];
        request(E[0], 0);
        request(E[1], 1);
        __syncthreads();

        for (int h = 0; h < n_half; ++h) {
            if (h + 1 < n_half) {
                stage(h + 1, st);                               // into the half whose sweeps ended before the last barrier
                if (h + 2 < n_half) st = fetch_next();          // half h + 2
            }
            if (busy) {
                auto mic = [&](int m, auto kc, auto pc) {
                    constexpr int K = decltype(kc)::value, K2 = (K + 2) % KE, POS = decltype(pc)::value;   // POS: the mic's place in its half
                    const Entries& cur = E[K];
                    if constexpr (kLerp) {
                        // The first mic of a half reads its quads itself; every other mic's reads were issued by its predecessor's last
                        // step.  Step 0 ends with everything landed (its last wait is lgkmcnt(0)); the s_waitcnt behind it costs nothing
                        // and tells the compiler so -- the entries of the next two mics are usable from here on.
                        if constexpr (POS == 0) long_lerp_first<NSEG>(acc, cur.e[0], cur.hp[0], lb);
                        else long_lerp_cont<NSEG>(acc, cur.hp[0]);
                        __builtin_amdgcn_s_waitcnt(0xC07F);
                        request(E[K2], m + 2);
                        if constexpr (POS == HC - 1) long_lerp_rest<NSEG, false>(acc, cur.e, cur.hp, lb, 0);
                        else long_lerp_rest<NSEG, true>(acc, cur.e, cur.hp, lb, E[(K + 1) % KE].e[0]);
                    } else {                                    // pad, the same structure
                        if constexpr (POS == 0) long_pad_first<NSEG>(acc, cur.e[0], lb);
                        else long_pad_cont<NSEG>(acc);
                        __builtin_amdgcn_s_waitcnt(0xC07F);
                        request(E[K2], m + 2);
                        if constexpr (POS == HC - 1) long_pad_rest<NSEG, false>(acc, cur.e, lb, 0);
                        else long_pad_rest<NSEG, true>(acc, cur.e, lb, E[(K + 1) % KE].e[0]);
                    }
                };
                using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
                using I3 = std::integral_constant<int, 3>;
                const int m0 = h * HC;
                if constexpr (HC == 4) {
                    mic(m0 + 0, I0{}, I0{}); mic(m0 + 1, I1{}, I1{}); mic(m0 + 2, I2{}, I2{}); mic(m0 + 3, I3{}, I3{});
                } else {
                    // entry sets rotate K = m % 3 from 0 at the start of every half ...
                    using I4 = std::integral_constant<int, 4>; using I5 = std::integral_constant<int, 5>; using I6 = std::integral_constant<int, 6>;
                    using I7 = std::integral_constant<int, 7>;
                    mic(m0 + 0, I0{}, I0{}); mic(m0 + 1, I1{}, I1{}); mic(m0 + 2, I2{}, I2{}); mic(m0 + 3, I0{}, I3{});
                    mic(m0 + 4, I1{}, I4{}); mic(m0 + 5, I2{}, I5{}); mic(m0 + 6, I0{}, I6{}); mic(m0 + 7, I1{}, I7{});
                    // ... so the two sets already requested for the next half's first mics (8 % 3, 9 % 3) = (2, 0) move to slots 0 and 1
                    const Entries t = E[0]; E[0] = E[2]; E[1] = t;
                }
            }
            __syncthreads();   // half h is free, half h + 1 is staged
        }

        // ---- k-ordered mean power (pad_and_sum.c:120-128): a.pbw waves at a time park the squared means of their directions
        // (row = direction, k in order; the rows alias the LDS image), then one direction per lane runs the sequential sum.
        // The squares are computed where they are stored, behind an opaque pass of the accumulators through an empty asm: without it
        // they are loop-invariant in the rounds loop and the compiler hoists all 64 of them out of it -- into scratch.
        const int pw_waves = a.pbw;
        for (int w0 = 0; w0 < W; w0 += pw_waves) {
            if (w0 > 0) __syncthreads();                        // the previous round's rows have been summed
            if (wave >= w0 && wave < w0 + pw_waves) {
                auto park = [&](auto mul_c) __attribute__((always_inline)) {   // (a call would force the accumulators into memory)
#pragma unroll
                    for (int j = 0; j < DW; ++j) {
                        float* row = lds + ((wave - w0) * DW + j) * kPark;
#pragma unroll
                        for (int sg = 0; sg < NSEG; ++sg) {
                            f32x2 a0 = acc[j][sg][0], a1 = acc[j][sg][1];
                            asm volatile("" : "+v"(a0), "+v"(a1));
                            // mean over the mics: a power-of-two count multiplies (exact), anything else divides like the reference
                            float o0, o1, o2, o3;
                            if constexpr (decltype(mul_c)::value) {
                                o0 = a0.x * a.inv_n; o1 = a0.y * a.inv_n; o2 = a1.x * a.inv_n; o3 = a1.y * a.inv_n;
                            } else {
                                float fm = (float)M;
                                asm volatile("" : "+v"(fm));   // not speculatable: keeps this path behind its branch
                                o0 = a0.x / fm; o1 = a0.y / fm; o2 = a1.x / fm; o3 = a1.y / fm;
                                __builtin_amdgcn_sched_barrier(0);   // one quad's divisions at a time: interleaved, their temporaries spill
                            }
                            // samples (2l, 2l+1) and (128+2l, 128+2l+1) of the segment
                            reinterpret_cast<float2*>(row + 256 * sg)[lane] = make_float2(o0 * o0, o1 * o1);
                            reinterpret_cast<float2*>(row + 256 * sg + 128)[lane] = make_float2(o2 * o2, o3 * o3);
                        }
                    }
                };
                if (__builtin_expect(a.n_is_pow2, 1)) park(std::true_type{}); else park(std::false_type{});
            }
            __syncthreads();
            const int g = wave * kWave + lane;                  // parked row of this lane
            const int d = g0 + w0 * DW + g;
            if (g < pw_waves * DW && d < tile_end) {
                const float* row = lds + g * kPark;
                const float4* row4 = reinterpret_cast<const float4*>(row);
                float sum = 0.0f;
                int k = 0;
                for (; k + 32 <= N; k += 32) {
                    float4 v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) v[u] = row4[(k >> 2) + u];
#pragma unroll
                    for (int u = 0; u < 8; ++u) { sum += v[u].x; sum += v[u].y; sum += v[u].z; sum += v[u].w; }
                }
                for (; k < N; ++k) sum += row[k];
                img[d - a.image_origin] = sum / (float)N;
            }
        }
    }
}

}  // namespace copies

// The sweeps keep LDS reads in flight in hard-wired registers across asm statements: sound only in a build that does not spill
// (tests/test_isa_hazards.py checks the build the tests run on; this checks the one that is about to launch).
template <typename K>
static hipError_t refuse_scratch(K kernel, int* cached)
{
    if (*cached < 0) {
        hipFuncAttributes fa{};
        hipError_t e = hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(kernel));
        if (e != hipSuccess) return e;
        *cached = (int)fa.localSizeBytes;
    }
#ifdef BF_STAMPS
    return hipSuccess;              // (the profiling build's stamp registers may spill: its phase shares are read, never its images)
#else
    return *cached != 0 ? hipErrorInvalidDeviceFunction : hipSuccess;
#endif
}

template <int ALGO, int NC>
hipError_t launch_nc(const DasLaunch& L, const KArgs& a, const DasPlan& plan, int frames, hipStream_t stream)
{
    const dim3 grid((unsigned)plan.n_tiles * (unsigned)frames);
    const dim3 block((unsigned)plan.waves * kWave);
    auto go = [&](auto kernel) -> hipError_t {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)plan.lds_bytes);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kernel, grid, block, plan.lds_bytes, stream, L.signals, L.images, L.mics, L.tab.whole, L.tab.frac, L.tab.taps, a);
        return hipGetLastError();
    };
    constexpr bool kFir = ALGO == ALGO_HYBRID || ALGO == ALGO_FIR_NAIVE || ALGO == ALGO_FIR_VEC;
    if constexpr ((NC >= 4 && !kFir) || (NC == 4 && kFir)) {
        if (plan.layout == 2) {
            constexpr bool kNeedsDigest = ALGO != ALGO_FIR_NAIVE && ALGO != ALGO_FIR_VEC;
            if (kNeedsDigest && L.tab.digest == nullptr) return hipErrorInvalidValue;   // launch_digest first
            if (kFir && L.n_taps != 8) return hipErrorInvalidValue;
            constexpr int NSEG = NC / 4;
            if constexpr (!kFir && NSEG == 1) {
                if (plan.nf == 2) {
                    if (L.tab.digest == nullptr || L.tab.digest_direct || plan.waves != copies::kWaves || plan.mic_chunk != copies::PairGeo<ALGO>::kMc ||
                        (!plan.interleaved && plan.row_stride != copies::PairGeo<ALGO>::kRs) || plan.lead != copies::PairGeo<ALGO>::kLead || (L.n_mics % 16) != 0)
                        return hipErrorInvalidValue;
                    if (plan.interleaved) {
                        if (plan.row_stride != copies::Pair2Geo<ALGO>::kRs) return hipErrorInvalidValue;
                        auto kernel2 = copies::das_pair2_kernel<ALGO>;
                        hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel2), hipFuncAttributeMaxDynamicSharedMemorySize, (int)plan.lds_bytes);
                        if (e2 != hipSuccess) return e2;
                        static int pair2_scratch = -1;
                        if ((e2 = refuse_scratch(kernel2, &pair2_scratch)) != hipSuccess) return e2;
                        const dim3 grid2((unsigned)plan.n_tiles * (unsigned)((frames + 1) / 2));
                        hipLaunchKernelGGL(kernel2, grid2, block, plan.lds_bytes, stream, L.signals, L.images, L.mics, L.tab.whole, L.tab.frac,
                                           reinterpret_cast<const float*>(L.tab.digest), a);
                        return hipGetLastError();
                    }
                    auto kernel = copies::das_pair_kernel<ALGO>;
                    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)plan.lds_bytes);
                    if (e != hipSuccess) return e;
                    static int pair_scratch = -1;             // (asm statements pass registers to each other: refuse a build that spills)
                    if (pair_scratch < 0) {
                        hipFuncAttributes fa{};
                        e = hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(kernel));
                        if (e != hipSuccess) return e;
                        pair_scratch = (int)fa.localSizeBytes;
                    }
                    if (pair_scratch != 0) return hipErrorInvalidDeviceFunction;
                    const dim3 pair_grid((unsigned)plan.n_tiles * (unsigned)((frames + 1) / 2));
                    hipLaunchKernelGGL(kernel, pair_grid, block, plan.lds_bytes, stream, L.signals, L.images, L.mics, L.tab.whole, L.tab.frac,
                                       reinterpret_cast<const float*>(L.tab.digest), a);
                    return hipGetLastError();
                }
            }
            if constexpr (kFir) {
                if (plan.nf == 2) {                             // das_hybrid_pair_kernel<hybrid | fir_naive | fir_vec>
                    using HG = copies::HybridGeo;
                    if (L.tab.digest == nullptr || plan.waves != copies::kWaves || plan.mic_chunk != HG::kMc || plan.row_stride != HG::kRs ||
                        plan.lead != HG::kLead || (L.n_mics % 16) != 0 || L.n_taps != 8)
                        return hipErrorInvalidValue;
                    auto kernel = copies::das_hybrid_pair_kernel<ALGO>;
                    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)plan.lds_bytes);
                    if (e != hipSuccess) return e;
                    static int hybrid_scratch = -1;
                    if ((e = refuse_scratch(kernel, &hybrid_scratch)) != hipSuccess) return e;
                    const dim3 pair_grid((unsigned)plan.n_tiles * (unsigned)((frames + 1) / 2));
                    hipLaunchKernelGGL(kernel, pair_grid, block, plan.lds_bytes, stream, L.signals, L.images, L.mics, L.tab.whole,
                                       reinterpret_cast<const float*>(L.tab.digest), L.tab.taps, a);
                    return hipGetLastError();
                }
            }
            if constexpr (!kFir && (NSEG == 2 || NSEG == 4)) {
                if (plan.long_rows) {                           // das_long_kernel
                    using LG = copies::LongGeo<ALGO, NSEG>;
                    if (L.tab.digest == nullptr || L.tab.digest_direct || plan.waves != copies::kWaves || plan.mic_chunk != LG::kMc ||
                        (L.n_mics % LG::kHalf) != 0 || plan.dpw != LG::kDw)
                        return hipErrorInvalidValue;
                    const bool fixed_rs = plan.row_stride == (LG::kLerp ? 2 : 1) * copies::Geo<NSEG>::kRs && plan.lead == copies::Geo<NSEG>::kLead;
                    auto kernel = fixed_rs ? copies::das_long_kernel<ALGO, NSEG, copies::Geo<NSEG>::kRs> : copies::das_long_kernel<ALGO, NSEG, 0>;
                    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)plan.lds_bytes);
                    if (e != hipSuccess) return e;
                    static int long_scratch[2] = {-1, -1};
                    if ((e = refuse_scratch(kernel, &long_scratch[fixed_rs ? 1 : 0])) != hipSuccess) return e;
                    hipLaunchKernelGGL(kernel, grid, block, plan.lds_bytes, stream, L.signals, L.images, L.mics, L.tab.whole, L.tab.frac,
                                       reinterpret_cast<const float*>(L.tab.digest), a);
                    return hipGetLastError();
                }
            }
            using G = copies::Geo<NSEG>;
            constexpr int kRs = kFir ? G::kRsFir : G::kRs;
            const bool fixed = plan.row_stride == kRs && plan.lead == G::kLead;
            auto kernel = fixed ? copies::das_copies_kernel<ALGO, NSEG, kRs, copies::kWaves> : copies::das_copies_kernel<ALGO, NSEG, 0, copies::kWaves>;
            if constexpr (!kFir && NSEG == 1) {
                if (plan.waves == 8) kernel = fixed ? copies::das_copies_kernel<ALGO, NSEG, kRs, 8> : copies::das_copies_kernel<ALGO, NSEG, 0, 8>;
            }
            bool direct = false;
            if constexpr (!kFir) {
                if (L.tab.digest_direct) {
                    if (plan.waves != copies::kWaves) return hipErrorInvalidValue;      // the DIRECT variant exists for 16 waves only
                    kernel = fixed ? copies::das_copies_kernel<ALGO, NSEG, kRs, copies::kWaves, true> : copies::das_copies_kernel<ALGO, NSEG, 0, copies::kWaves, true>;
                    direct = true;
                }
            }
            if (plan.waves != copies::kWaves && !(plan.waves == 8 && !kFir && NSEG == 1)) return hipErrorInvalidValue;
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)plan.lds_bytes);
            if (e != hipSuccess) return e;
            if constexpr (!kFir && NSEG == 1) if (!direct) {
                // This variant keeps LDS reads in flight across asm statements (issue_quads / await_quads): sound only while
                // the compiler neither spills nor copies those registers.  Spilling is checkable: refuse to run a build that
                // uses scratch (copies would show in the bit-exact parity tests).
                static int scratch_bytes[4] = {-1, -1, -1, -1};
                int& sb = scratch_bytes[(fixed ? 1 : 0) + (plan.waves == 8 ? 2 : 0)];
                if (sb < 0) {
                    hipFuncAttributes fa{};
                    e = hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(kernel));
                    if (e != hipSuccess) return e;
                    sb = (int)fa.localSizeBytes;
                }
                if (sb != 0) return hipErrorInvalidDeviceFunction;
            }
            // the digest rides in a pointer slot the algorithm does not use: taps (pad, lerp) or frac (hybrid)
            const float* dig = reinterpret_cast<const float*>(L.tab.digest);
            hipLaunchKernelGGL(kernel, grid, block, plan.lds_bytes, stream, L.signals, L.images, L.mics, L.tab.whole, kFir ? dig : L.tab.frac,
                               kFir ? L.tab.taps : dig, a);
            return hipGetLastError();
        }
    }
    if constexpr (NC == 4 && (ALGO == ALGO_PAD || ALGO == ALGO_LERP)) {
        if (plan.quad) {
            switch (plan.dpw) {
                case 1: return go(das_mimo_kernel<ALGO, 4, 1, true>);
                case 4: return go(das_mimo_kernel<ALGO, 4, 4, true>);
                default: return hipErrorInvalidValue;
            }
        }
    }
    switch (plan.dpw) {
        case 1: return go(das_mimo_kernel<ALGO, NC, 1, false>);
        case 4: return go(das_mimo_kernel<ALGO, NC, 4, false>);
        default: return hipErrorInvalidValue;
    }
}

template <int ALGO>
hipError_t launch_algo(const DasLaunch& L, const KArgs& a, const DasPlan& plan, int frames, hipStream_t stream)
{
    switch (plan.nc) {
        case 1: return launch_nc<ALGO, 1>(L, a, plan, frames, stream);
        case 2: return launch_nc<ALGO, 2>(L, a, plan, frames, stream);
        case 4: return launch_nc<ALGO, 4>(L, a, plan, frames, stream);
        case 8: return launch_nc<ALGO, 8>(L, a, plan, frames, stream);
        case 16: return launch_nc<ALGO, 16>(L, a, plan, frames, stream);
        default: return hipErrorInvalidValue;
    }
}

template <int ALGO>
hipError_t launch_miso_algo(const DasLaunch& L, const KArgs& a, const DasPlan& plan, const float* init_dev, float* out_dev, hipStream_t stream)
{
    auto go = [&](auto kernel) -> hipError_t {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)plan.lds_bytes);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kernel, dim3(1), dim3(kWave), plan.lds_bytes, stream, L.signals, L.images, L.mics, L.tab.whole, L.tab.frac, L.tab.taps,
                           init_dev, out_dev, a);
        return hipGetLastError();
    };
    switch (plan.nc) {
        case 1: return go(das_miso_kernel<ALGO, 1>);
        case 2: return go(das_miso_kernel<ALGO, 2>);
        case 4: return go(das_miso_kernel<ALGO, 4>);
        case 8: return go(das_miso_kernel<ALGO, 8>);
        case 16: return go(das_miso_kernel<ALGO, 16>);
        default: return hipErrorInvalidValue;
    }
}

long long grouped_entries_for_args(const DasLaunch& L, const DasPlan& plan)
{
    const long long groups = ((long long)(L.dir_end - L.dir_begin) + plan.dpw - 1) / plan.dpw;
    return groups * L.n_mics * plan.dpw;
}

KArgs make_args(const DasLaunch& L, const DasPlan& plan)
{
    KArgs a{};
    a.miso_row = 0;
    a.n_mics = L.n_mics; a.m_total = L.m_total; a.n_samples = L.n_samples; a.n_taps = L.n_taps;
    a.dir_begin = L.dir_begin; a.dir_end = L.dir_end; a.image_stride = L.image_stride; a.image_origin = L.image_origin;
    a.lead = plan.lead; a.row_stride = plan.row_stride; a.mic_chunk = plan.mic_chunk; a.n_chunks = plan.n_chunks;
    a.tile_dirs = plan.tile_dirs; a.n_tiles = plan.n_tiles;
    a.scratch_off = plan.scratch_off; a.srow = plan.srow; a.pbw = plan.pbw;
    a.n_is_pow2 = (L.n_mics & (L.n_mics - 1)) == 0;
    a.inv_n = 1.0f / (float)L.n_mics;
    a.debug = L.debug;
    a.n_frames = L.frames;
    a.wg_frames = plan.nf == 2 ? (L.frames + 1) / 2 : L.frames;
    a.frame_inner = plan.frame_inner;
    a.digest_h_off = (plan.layout == 2 && (L.algo == ALGO_LERP || (L.algo == ALGO_HYBRID && plan.nf == 2))) ? grouped_entries_for_args(L, plan) : 0;
    a.digest_t_off = (plan.layout == 2 && L.algo == ALGO_HYBRID && plan.nf == 2) ? 2 * grouped_entries_for_args(L, plan) : 0;
    return a;
}

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

}  // namespace

int plan_das(const DasLaunch& L, int n_cus, DasPlan* plan, const char** why)
{
    static const char* kWhy[] = {"", "N_SAMPLES must be in [1, 1024]", "N_TAPS must be in [1, 64] (multiple of 8 for the vectorized FIR)",
                                 "one microphone row does not fit in LDS", "empty launch"};
    auto fail = [&](int i) { if (why) *why = kWhy[i]; return -i; };
    if (L.n_samples < 1 || L.n_samples > 1024) return fail(1);
    const bool fir = L.algo == ALGO_HYBRID || L.algo == ALGO_FIR_NAIVE || L.algo == ALGO_FIR_VEC;
    if (fir && (L.n_taps < 1 || L.n_taps > 64 || (L.algo == ALGO_FIR_VEC && (L.n_taps % 8) != 0))) return fail(2);
    if (L.n_mics < 1 || L.frames < 1 || L.dir_end <= L.dir_begin) return fail(4);

    DasPlan p{};
    p.nf = 1;
    p.frame_inner = 0;
    int nc = (L.n_samples + kWave - 1) / kWave;
    p.nc = nc <= 1 ? 1 : nc <= 2 ? 2 : nc <= 4 ? 4 : nc <= 8 ? 8 : 16;
    const int T = fir ? L.n_taps : 0;
    const int shift = (L.algo == ALGO_FIR_NAIVE || L.algo == ALGO_FIR_VEC) ? 0 : L.tab.max_whole;
    p.lead = round_up(shift + 1 + T / 2, 4);
    const int tail = round_up(T, 4);
    p.row_stride = p.lead + p.nc * kWave + tail;
    const size_t row_bytes = (size_t)p.row_stride * sizeof(float);

    // One 1024-thread workgroup (16 waves) per CU owns the whole 160 KiB LDS:
    //   [ mic rows of one frame (or one chunk of them) | per-wave power scratch: waves x pbw rows of 64*nc+4 floats ]
    // When the frame's mic block does not fit beside the scratch, the mics are staged in chunks and every wave
    // carries DPW directions' accumulators across the chunks.
    const size_t lds_budget = 160 * 1024;
    p.waves = 16;
    p.srow = p.nc * kWave + 4;   // +4: keeps rows 16-byte aligned and 4 banks apart; column nc*64 holds the direction id
    p.pbw = p.nc <= 4 ? 4 : p.nc <= 8 ? 2 : 1;
    const size_t scratch_bytes = (size_t)p.waves * p.pbw * p.srow * sizeof(float);
    const size_t sig_budget = lds_budget - scratch_bytes - 16;
    if (row_bytes * (size_t)L.n_mics <= sig_budget) {
        p.mic_chunk = L.n_mics; p.n_chunks = 1; p.dpw = 1;
    } else {
        int mc = (int)(sig_budget / row_bytes);
        if (mc < 1) return fail(3);
        if (mc >= 4) mc &= ~3;
        if (mc > L.n_mics) mc = L.n_mics;
        p.mic_chunk = mc; p.n_chunks = (L.n_mics + mc - 1) / mc;
        p.dpw = p.n_chunks > 1 ? 4 : 1;
    }
    // Layout for pad / lerp at 128 < N <= 1024 and for the 8-tap FIR flavours at 128 < N <= 256: 2 = shifted copies
    // (default), 0 = strided; pad / lerp at N <= 256 only: 1 = quad + DPP.
    const bool plain = L.algo == ALGO_PAD || L.algo == ALGO_LERP;
    const bool copies_ok = plain ? p.nc >= 4 : (p.nc == 4 && L.n_taps == 8);
    p.layout = copies_ok ? (L.force_layout >= 0 ? L.force_layout : 2) : 0;
    if (p.layout == 1 && !(plain && p.nc == 4)) p.layout = 0;
    p.quad = p.layout == 1 ? 1 : 0;
    if (p.layout == 2) {
        const int nseg = p.nc / 4, arrays = (L.algo == ALGO_LERP) ? 2 : 1;
        const int fixed_lead = nseg == 1 ? copies::Geo<1>::kLead : copies::Geo<4>::kLead;   // Geo<2> == Geo<4> here
        const int dw = nseg == 4 ? copies::Geo<4>::kDw : copies::Geo<1>::kDw;               // Geo<2> == Geo<1> here
        // zero prefix: the furthest look-back is the delay (+1 for lerp, +1 + T/2 for hybrid, T/2 for the plain FIRs)
        const int back = L.algo == ALGO_HYBRID ? L.tab.max_whole + 1 + L.n_taps / 2 : fir ? L.n_taps / 2 : L.tab.max_whole + 1;
        p.lead = round_up(back + 1, 4);
        if (p.lead <= fixed_lead && !(L.debug & 2)) p.lead = fixed_lead;   // compile-time row stride
        p.row_stride = p.lead + nseg * 256 + (fir ? copies::Geo<1>::kFirTail : 0);
        p.copies = copies::copies_of(L.algo, L.tab.digest_direct);
        const size_t slot_bytes = (size_t)arrays * p.copies * p.row_stride * sizeof(float);
        // a chunk: as many mics as fit beside nothing else in 156 KiB, at most 16 (one s_load of table entries) and at
        // most what the 16 waves stage in one go (one (mic, segment) pair each; two for pad with several segments)
        int stage_pairs = 16 * ((nseg > 1 && L.algo == ALGO_PAD) ? 2 : 1);
        // One 16-wave workgroup per CU with (nearly) the whole LDS.  pad / lerp at N <= 256 also come as 8-wave workgroups
        // (two per CU, 78 KiB each): twice the staging per direction, so only for grids too coarse to fill 16 waves' 128
        // directions (cfg1: 121 directions, 637K -> 961K frames/s).  (cfg2, 190 frames: 16 waves 80.0K, 8 waves 72.0K.)
        const int waves = (plain && nseg == 1 && ((L.dir_end - L.dir_begin) < 256 || (L.debug & 4))) ? 8 : copies::kWaves;   // debug bit 2: A/B switch
        const size_t budget = waves == 8 ? (size_t)78 * 1024 : (size_t)156 * 1024;
        if (plain && nseg == 1 && waves == copies::kWaves && !L.tab.digest_direct && !(L.debug & 8)) stage_pairs = 32;   // debug bit 3: A/B switch
        int mc = (int)(budget / slot_bytes);
        if (mc > stage_pairs / nseg) mc = stage_pairs / nseg;
        mc = mc >= 32 ? 32 : mc >= 16 ? 16 : mc >= 8 ? 8 : mc >= 4 ? 4 : mc >= 2 ? 2 : mc;
        if (mc < 1) return fail(3);
        if (mc > L.n_mics) mc = L.n_mics;
        // Two frames per workgroup (das_pair_kernel) where its fixed geometry applies: the per-step scalar work is then shared
        // by both frames.  (debug bit 4: A/B switch)
        p.nf = 1;
        if (plain && nseg == 1 && waves == copies::kWaves && !L.tab.digest_direct && p.lead == fixed_lead && (L.n_mics % 16) == 0 &&
            (L.n_samples % 4) == 0 && L.frames >= 2 && !(L.debug & 16)) {
            p.nf = 2;
            mc = 16;
            // frames interleaved in the rows (das_pair2_kernel) for lerp: 4 instead of 8 LDS reads per (re)load, +1.5 %; pad reads
            // half as much to begin with and measured 2 % slower that way  (debug bit 15: A/B switch, the other kernel)
            if ((L.algo == ALGO_LERP) != ((L.debug & 32768) != 0)) {
                p.interleaved = 1;
                p.row_stride = 2 * copies::Geo<1>::kRs;         // the two frames of a mic share a row, sample by sample
            }
        }
        // The hybrid beamformer's two-frame sweep (das_hybrid_pair_kernel) under the same conditions.  (debug bit 13: A/B switch)
        if (fir && nseg == 1 && L.n_taps == 8 && waves == copies::kWaves && p.lead == fixed_lead && (L.n_mics % 16) == 0 &&
            (L.n_samples % 4) == 0 && L.frames >= 2 && !(L.debug & 8192)) {
            p.nf = 2;
            mc = copies::HybridGeo::kMc;                        // 32 mic slots: two frames interleaved per row, two shifted copies
            p.copies = copies::HybridGeo::kC;
            p.row_stride = copies::HybridGeo::kRs;
            p.interleaved = 1;
        }
        const bool hybrid_pair = fir && p.nf == 2;
        // Long rows (2 / 4 segments): das_long_kernel where its LDS image -- two halves of 16 / nseg mics -- fits and the mic count is
        // a whole number of halves.  (debug bit 12: A/B switch back to das_copies_kernel)
        p.long_rows = 0;
        if (plain && nseg > 1 && !L.tab.digest_direct && !(L.debug & 4096)) {
            const int half = 16 / nseg;
            if ((L.n_mics % half) == 0 && (L.n_samples % 4) == 0 && slot_bytes * (size_t)(2 * half) <= (size_t)160 * 1024) {
                p.long_rows = 1;
                mc = 2 * half;
                if (L.algo == ALGO_LERP) {                      // rows of (sample pair, difference pair) quads: two floats per sample, no separate difference rows
                    p.interleaved = 1;
                    p.row_stride *= 2;
                }
            }
        }
        p.mic_chunk = mc; p.n_chunks = (L.n_mics + mc - 1) / mc;
        p.waves = waves; p.dpw = dw; p.srow = nseg * 256 + 4;
        p.scratch_off = 0;
        const size_t buf = hybrid_pair ? (size_t)mc * copies::HybridGeo::kSlot * sizeof(float) : slot_bytes * (size_t)mc * (size_t)p.nf;
        const size_t wave_rows = (size_t)dw * p.srow * sizeof(float);          // the parked rows of one wave
        p.lds_bytes = buf > 2 * wave_rows ? buf : 2 * wave_rows;
        if (p.long_rows && p.lds_bytes < 8 * wave_rows) p.lds_bytes = 8 * wave_rows;             // eight waves park together (two rounds)
        if (nseg == 1 && p.lds_bytes < p.waves * wave_rows) p.lds_bytes = p.waves * wave_rows;   // N <= 256: the whole group parks at once
        int pw = (int)(p.lds_bytes / wave_rows);                                 // waves that park together (power of two)
        p.pbw = pw >= 16 ? 16 : pw >= 8 ? 8 : pw >= 4 ? 4 : 2;
        if (p.pbw > p.waves) p.pbw = p.waves;
    }
    if (p.layout != 2) {
        p.scratch_off = round_up(p.mic_chunk * p.row_stride, 4);
        p.lds_bytes = (size_t)p.scratch_off * sizeof(float) + scratch_bytes;
    }

    // Tile size: enough workgroups to fill the chip a few times over, but as many directions per staged block
    // as possible.  A tile is a whole number of wave groups -- except for small launches (a single frame through the
    // host-pointer API), where latency matters: then every CU gets a tile, even if that leaves waves of a group idle.
    const int group = p.waves * p.dpw;
    const long long dirs = (long long)(L.dir_end - L.dir_begin);
    bool spread = false;   // no XCD affinity of the tiles (see below)
    const long long target_wgs = (long long)n_cus * 4;
    const long long wg_frames = p.nf == 2 ? (L.frames + 1) / 2 : L.frames;   // frames (frame pairs) a column of the grid walks
    long long td = (dirs * wg_frames + target_wgs - 1) / target_wgs;
    if (dirs * wg_frames < (long long)n_cus * group) {
        td = (dirs * wg_frames + n_cus - 1) / n_cus;
        td = round_up((int)(td < 1 ? 1 : td), p.dpw);
    } else {
        // A whole number of wave groups per tile, chosen by what the grid costs.  Workgroup ids go round-robin over the 8 XCDs.
        //   * Large tables: workgroup id -> (tile, frame) with the tile count padded to a multiple of 8 keeps
        //     tile % 8 == id % 8, so every XCD's L2 serves only its own tiles' table rows for all frames; XCD x then runs
        //     the tiles with tile % 8 == x on its 32 CUs and the launch takes  max_x ceil(tiles_x * frames / 32)  rounds of
        //     k groups.  (cfg2, 95 frame pairs, lerp: k = 2 or 5 -> 30 units, 122K frames/s; k = 4 -> 36, 106K; k = 8 -> 48,
        //     80K: measured.)
        //   * Tables that fit every XCD's L2 whole (a rank's direction shard of bench.py --gpus 4 / 8): no padding, every
        //     tile's workgroups spread over the XCDs, ceil(tiles * frames / CUs) rounds -- pinning 10 tiles to 8 XCDs left
        //     a rank of the 8-GPU shape at 63 % of the one-GPU rate.
        // Ties go to the first of 2, 3, .., 8, 1.
        const size_t table_bytes = (size_t)dirs * (size_t)L.n_mics * 4u * ((L.algo == ALGO_LERP ? 2u : 1u) + (fir ? (size_t)L.n_taps : 0u));
        spread = (table_bytes <= ((size_t)3 << 20) && !(L.debug & 32)) || (L.debug & 64);   // debug bits 5 / 6: A/B switches (never / always)
        // an XCD's share of the table beyond its L2: all frames of a tile back to back (tile_and_frame); debug bit 7: never
        p.frame_inner = (!spread && p.layout == 2 && table_bytes > ((size_t)16 << 20) && wg_frames > 1 && !(L.debug & 128)) ? 1 : 0;
        const int wg_per_cu = 1;
        const int xcds = 8, cus_per_xcd = (n_cus >= xcds ? n_cus / xcds : 1) * wg_per_cu;
        long long best_cost = -1;
        int best_k = 4;
        for (int i = 0; i < 8; ++i) {
            const int k = i < 7 ? i + 2 : 1;
            const long long tiles = (dirs + (long long)k * group - 1) / ((long long)k * group);
            const long long tiles_x = tiles / xcds + (tiles % xcds ? 1 : 0);        // the busiest XCD's share
            const long long slots = (long long)n_cus * wg_per_cu;
            const long long rounds = spread ? (tiles * wg_frames + slots - 1) / slots : (tiles_x * wg_frames + cus_per_xcd - 1) / cus_per_xcd;
            const long long cost = rounds * k;
            if (best_cost < 0 || cost < best_cost) { best_cost = cost; best_k = k; }
        }
        td = (long long)best_k * group;
        if ((L.debug >> 8) & 15) td = (long long)((L.debug >> 8) & 15) * group;   // debug bits 8..11: tile size in groups (A/B)
    }
    p.tile_dirs = (int)td;
    p.n_tiles = spread ? (int)((dirs + td - 1) / td) : round_up((int)((dirs + td - 1) / td), 8);
    *plan = p;
    if (why) *why = kWhy[0];
    return 0;
}

namespace {
long long grouped_entries(const DasLaunch& L, const DasPlan& plan) { return grouped_entries_for_args(L, plan); }
}  // namespace

size_t digest_elements(const DasLaunch& L, const DasPlan& plan)
{
    if (plan.layout != 2) return 0;
    const size_t direct = (size_t)L.n_dirs * (size_t)L.n_mics;                 // the [D][M] layout of the DIRECT variant
    if (L.algo == ALGO_PAD) return std::max((size_t)grouped_entries(L, plan), direct);
    if (L.algo == ALGO_LERP) return std::max((size_t)(2 * grouped_entries(L, plan)), direct);    // offsets, then the lerp weights in the same order
    // the FIR pair kernel: offsets and packed guards (hybrid), then the taps regrouped per 8 directions
    if (L.algo == ALGO_HYBRID) return std::max((size_t)L.n_dirs * (size_t)L.n_mics, plan.nf == 2 ? (size_t)(10 * grouped_entries(L, plan)) : (size_t)0);
    if ((L.algo == ALGO_FIR_NAIVE || L.algo == ALGO_FIR_VEC) && plan.nf == 2) return (size_t)(8 * grouped_entries(L, plan));
    return 0;
}

hipError_t launch_digest(const DasLaunch& L, const DasPlan& plan, int32_t* d_digest, unsigned long long* d_reload_count, bool direct, hipStream_t stream)
{
    // what the kernel looks back by beyond the whole-sample delay: lerp reads s[k - p - 1], hybrid starts its window at
    // s[k - p - 1 - T/2] (T = 8)
    // (`arrays`: rows of one staged mic in units of its shifted copies -- samples, lerp's differences, and both frames of the pair kernel)
    // (the hybrid pair kernel interleaves its two frames inside a row: one array per mic)
    // (das_long_kernel's lerp rows carry their differences inside: one array)
    const int arrays = ((L.algo == ALGO_LERP && !(plan.long_rows && plan.interleaved)) ? 2 : 1) * ((plan.nf == 2 && !plan.interleaved) ? 2 : 1), bias = (L.algo == ALGO_LERP) ? 1 : (L.algo == ALGO_HYBRID) ? 5 : 0;
    const bool fir_pair = plan.nf == 2 && (L.algo == ALGO_HYBRID || L.algo == ALGO_FIR_NAIVE || L.algo == ALGO_FIR_VEC);
    if ((L.algo == ALGO_HYBRID && plan.nf != 2) || direct) {
        hipLaunchKernelGGL(digest_kernel, dim3(1024), dim3(256), 0, stream, L.tab.whole, d_digest, (long long)L.n_dirs * L.n_mics, L.n_mics, plan.mic_chunk,
                           arrays, plan.row_stride, plan.lead, bias, plan.copies);
    } else {
        const long long entries = grouped_entries(L, plan);
        hipLaunchKernelGGL(digest_grouped_kernel, dim3(1024), dim3(256), 0, stream, L.algo == ALGO_HYBRID || !fir_pair ? L.tab.whole : nullptr,
                           L.algo == ALGO_LERP ? L.tab.frac : nullptr, d_digest,
                           entries, entries, L.n_mics, plan.dpw, L.dir_begin, L.dir_end, plan.mic_chunk, arrays, plan.row_stride, plan.lead, bias,
                           plan.copies, d_reload_count, L.algo == ALGO_HYBRID ? 1 : 0, plan.interleaved ? 2 : 1, fir_pair ? L.tab.taps : nullptr,
                           fir_pair ? (L.algo == ALGO_HYBRID ? 2 * entries : 0) : 0);
    }
    return hipGetLastError();
}

// Steps of a launch over which the sweep can share reads at all (all but the first direction of every group).
long long digest_shareable_steps(const DasLaunch& L, const DasPlan& plan)
{
    return plan.dpw > 1 ? grouped_entries(L, plan) / plan.dpw * (plan.dpw - 1) : 0;
}

// Profiling build only: read and clear the phase totals of das_pair_kernel (16 counters; zeros in a production build).
hipError_t read_phase_stamps(unsigned long long* out16, bool clear)
{
#ifdef BF_STAMPS
    static unsigned long long all[64 * 16];
    hipError_t e = hipMemcpyFromSymbol(all, HIP_SYMBOL(copies::g_stamps), sizeof(all));
    if (e != hipSuccess) return e;
    for (int i = 0; i < 16; ++i) { out16[i] = 0; for (int r = 0; r < 64; ++r) out16[i] += all[16 * r + i]; }
    if (!clear) return e;
    for (int i = 0; i < 64 * 16; ++i) all[i] = 0;
    return hipMemcpyToSymbol(HIP_SYMBOL(copies::g_stamps), all, sizeof(all));
#else
    for (int i = 0; i < 16; ++i) out16[i] = 0;
    (void)clear;
    return hipSuccess;
#endif
}

hipError_t launch_das(const DasLaunch& L, const DasPlan& plan, hipStream_t stream)
{
    const KArgs a = make_args(L, plan);
    switch (L.algo) {
        case ALGO_PAD: return launch_algo<ALGO_PAD>(L, a, plan, L.frames, stream);
        case ALGO_LERP: return launch_algo<ALGO_LERP>(L, a, plan, L.frames, stream);
        case ALGO_HYBRID: return launch_algo<ALGO_HYBRID>(L, a, plan, L.frames, stream);
        case ALGO_FIR_NAIVE: return launch_algo<ALGO_FIR_NAIVE>(L, a, plan, L.frames, stream);
        case ALGO_FIR_VEC: return launch_algo<ALGO_FIR_VEC>(L, a, plan, L.frames, stream);
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_miso(const DasLaunch& L, const DasPlan& plan, long long row_offset, const float* init_dev, float* out_dev,
                       hipStream_t stream)
{
    KArgs a = make_args(L, plan);
    a.miso_row = row_offset;
    switch (L.algo) {
        case ALGO_PAD: return launch_miso_algo<ALGO_PAD>(L, a, plan, init_dev, out_dev, stream);
        case ALGO_LERP: return launch_miso_algo<ALGO_LERP>(L, a, plan, init_dev, out_dev, stream);
        case ALGO_HYBRID: return launch_miso_algo<ALGO_HYBRID>(L, a, plan, init_dev, out_dev, stream);
        case ALGO_FIR_NAIVE: return launch_miso_algo<ALGO_FIR_NAIVE>(L, a, plan, init_dev, out_dev, stream);
        case ALGO_FIR_VEC: return launch_miso_algo<ALGO_FIR_VEC>(L, a, plan, init_dev, out_dev, stream);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace bf
