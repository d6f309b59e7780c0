// Internal interface between the C-ABI layer (beamformer_api.cpp) and the gfx950 kernels (das_kernels.hip).
// Not installed; the public boundary is include/beamformer_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bf {

// Delay flavours of the reference's four beamformers (PC/src/algorithms/*.c).
enum Algo : int {
    ALGO_PAD = 0,        // pad_and_sum.c        integer shift
    ALGO_LERP = 1,       // lerp_and_sum.c       integer shift + linear interpolation
    ALGO_HYBRID = 2,     // hybrid_convolve_and_sum.c  integer shift + T-tap fractional FIR
    ALGO_FIR_NAIVE = 3,  // convolve_and_sum.c   T-tap FIR, sequential tap order (mimo_convolve_naive)
    ALGO_FIR_VEC = 4,    // convolve_and_sum.c   T-tap FIR, AVX2 lane/tree order  (mimo_convolve_vectorized)
    ALGO_COUNT = 5
};

// One steering-table set resident in HBM (what load_coefficients_* uploads).
struct DeviceTables {
    const int32_t* whole = nullptr;  // [D][M]      integer delays            (pad, lerp, hybrid)
    const float* frac = nullptr;     // [D][M]      h = 1 - frac              (lerp)
    const float* taps = nullptr;     // [D][M][T]   FIR taps                  (hybrid, fir)
    int max_whole = 0;               // max over the table, clamped to N (sizes the zero prefix in LDS)
    bool digest_direct = false;      // digest is in the [D][M] layout: run the direction-outer (DIRECT) kernel variant (plan_das sizes the chunk for its four copies)
    const int32_t* digest = nullptr; // LDS byte offsets for the shifted-copies layout (launch_digest): grouped by the wave's directions for pad / lerp
                                     // (+ the lerp weights in the same order), [D][M] for hybrid; null when not built
};

// Geometry of one launch.  Directions [dir_begin, dir_end) of every frame in [0, frames).
struct DasLaunch {
    const float* signals;    // [frames][m_total][N] device, mic-major (receiver.c:94-151 layout)
    float* images;           // [frames][image_stride] device; direction d lands at images[f*image_stride + d - image_origin]
    const int32_t* mics;     // [M] device copy of adaptive_array (row of `signals` for each active slot)
    DeviceTables tab;
    int algo;
    int n_mics;              // M  = n of the reference call
    int m_total;             // rows in one frame of `signals`
    int n_samples;           // N
    int n_taps;              // T
    int n_dirs;              // D  = MAX_RES_X * MAX_RES_Y
    int dir_begin, dir_end;  // shard of the direction grid handled by this launch
    int image_stride, image_origin;
    int frames;
    int force_layout;        // tests/bench ($BF_LAYOUT): -1 = planner's choice, else 0 / 1 / 2 for pad and lerp at N <= 256
    int debug;               // profiling / A-B switches ($BF_DEBUG), 0 in production: bit 0 skip the power sum (strided kernel), bit 1 run-time row stride (copies kernel),
                             // bit 2 8-wave workgroups, bit 3 16-mic chunks, bit 4 one frame per workgroup, bits 8..11 tile size in wave groups
};

// Plan chosen on the host for a launch (exposed so tests can check LDS sizing without a GPU).
struct DasPlan {
    int nc;          // 64-sample segments per row (N rounded up to 64*nc)
    int lead;        // zero floats in front of every LDS row
    int row_stride;  // floats per LDS row
    int mic_chunk;   // mics staged per pass
    int n_chunks;
    int waves;       // waves per workgroup
    int scratch_off; // float offset of the per-wave power scratch in LDS
    int srow;        // scratch row stride in floats (64*nc + 4)
    int pbw;         // scratch rows (finished directions) per wave
    int quad;        // 1: lane owns 4 consecutive samples (ds_read_b128 + DPP), 0: lane-strided samples (ds_read_b32)
    int layout;      // 0 strided, 1 quad + DPP, 2 shifted copies (pad / lerp, N <= 256)
    int dpw;         // directions a wave carries across mic chunks
    int nf;          // frames a workgroup carries (2: das_pair_kernel -- pad / lerp at N <= 256 with the fixed row stride, a multiple of 16 mics and two or more frames)
    int interleaved; // 1: two-frame kernels whose LDS rows hold both frames interleaved sample by sample (das_pair2_kernel, das_hybrid_pair_kernel)
    int long_rows;   // 1: das_long_kernel (pad / lerp at 256 < N <= 1024: LDS image in two halves, conflict-free lane mapping)
    int frame_inner; // workgroup id -> (tile, frame): 1 = an XCD runs all frames of a tile back to back (tables beyond the L2s)
    int copies;      // layout 2: shifted copies per staged array (2: the sweep of pad / lerp, 4: FIR flavours and the DIRECT variant)
    int tile_dirs;   // directions per workgroup
    int n_tiles;     // padded to a multiple of 8 (XCD affinity: tile % 8 == workgroup id % 8) unless the launch's table fits every XCD's L2
    size_t lds_bytes;
};

// Returns 0 and fills `plan`, or a negative value when the shape is unsupported (message in `why`).
int plan_das(const DasLaunch& L, int n_cus, DasPlan* plan, const char** why);

// Build DeviceTables::digest for the layout `plan` describes (shifted-copies layout only).
size_t digest_elements(const DasLaunch& L, const DasPlan& plan);   // 4-byte elements the digest of this launch needs (0: none)
// `direct`: the [D][M] layout of the direction-outer kernel variant (tables without structure) instead of the grouped one;
// d_reload_count (grouped build, optional) receives the number of direction steps whose delay differs from the previous one's.
hipError_t launch_digest(const DasLaunch& L, const DasPlan& plan, int32_t* d_digest, unsigned long long* d_reload_count, bool direct, hipStream_t stream);
long long digest_shareable_steps(const DasLaunch& L, const DasPlan& plan);

// Enqueue on `stream`; no host synchronisation, no allocation (graph-capturable).
hipError_t launch_das(const DasLaunch& L, const DasPlan& plan, hipStream_t stream);

// One steered beam -> raw out[N] (miso_pad / miso_lerp / miso_convolve_*).  `row_offset` is the reference's flat
// table `offset` (d*M); `init_dev` (may be null) seeds the accumulators, which turns the launch into the
// single-signal helpers `out += delay(signal)` (pad_delay, lerp_delay, convolve_*_delay*).
hipError_t launch_miso(const DasLaunch& L, const DasPlan& plan, long long row_offset, const float* init_dev, float* out_dev,
                       hipStream_t stream);

// FPGA protocol-v2 datagrams (one per sample instant) -> float32 [n_mics_out][n_samples] mic-major frame (receiver.c:94-151).
hipError_t launch_ingest(const void* d_packets, int packet_stride, int header_bytes, int n_samples, int n_mics_out, int stream_len,
                         int rows, int columns, float* d_frame, hipStream_t stream);

// heat-map post-processing (PC/src/visual.py:143-188, 295-322, 450-452); see heatmap_kernels.hip
hipError_t launch_colorize(const float* d_power, int frames, int res_x, int res_y, float threshold, float amount, float exponent,
                           unsigned char* d_small, int* d_overlay, hipStream_t stream);
hipError_t launch_overlay(const unsigned char* d_small, int frames, int small_w, int small_h, int out_w, int out_h, unsigned char* d_prev,
                          const unsigned char* d_camera, unsigned char* d_out, float w_prev, float w_new, float w_cam, float w_heat,
                          hipStream_t stream);
// uint8 BGR frame [sh][sw][3] -> [oh][ow][3]: cv2.resize(INTER_LINEAR) to new_w x new_h at (top, left), the rest = value (heatmap_kernels.hip)
hipError_t launch_letterbox(const unsigned char* d_src, int sh, int sw, unsigned char* d_out, int oh, int ow, int new_h, int new_w, int top, int left, int value,
                            hipStream_t stream);
hipError_t launch_power_center(const float* d_power, int frames, int rows, int cols, float* d_centers, float* d_workspace, hipStream_t stream);

// frequency-domain beamformers (freq_kernels.hip): steering phasors, DFT of the selected bins, and the MFMA complex GEMM
// with its three epilogues (phase-steer DAS power, covariance, MVDR quadratic form) plus the per-bin Cholesky inverse.
hipError_t launch_fd_steering(const double* d_tau, const double* d_freq, int n_dirs, int n_mics, int n_bins, float* d_are, float* d_aim, hipStream_t stream);
// Twiddle table of the MFMA DFT for (N, bin range): fd_twiddle_floats floats, built by launch_fd_twiddles.  launch_fd_dft
// without a table runs the plain one-workgroup-per-row kernel.
size_t fd_twiddle_floats(int n_samples, int n_bins);
hipError_t launch_fd_twiddles(int n_samples, int bin_lo, int n_bins, float* d_tw, hipStream_t stream);
hipError_t launch_fd_dft(const float* d_frames, const int32_t* d_mics, int m_total, int n_samples, int n_frames, int n_mics, int bin_lo, int n_bins,
                         const float* d_tw, float* xre_mf, float* xim_mf, float* xre_fm, float* xim_fm, hipStream_t stream);
// Workspace (floats) that lets the two bin-reducing GEMMs split the bins over several workgroup groups (optional: without
// it they run one group).  n_rows = frames for the delay-and-sum power, 1 for MVDR.
size_t fd_workspace_floats(int n_rows, int n_dirs, int n_bins);
hipError_t launch_fd_das_power(const float* xre_mf, const float* xim_mf, const float* are, const float* aim, int n_frames, int n_mics, int n_dirs,
                               int n_bins, float* d_power, float* d_work, size_t work_floats, hipStream_t stream);
hipError_t launch_fd_covariance(const float* xre_fm, const float* xim_fm, int n_frames, int n_mics, int n_bins, float* rre, float* rim, hipStream_t stream);
// Up to 128 mics: one in-LDS factorisation per bin, no workspace.  129..256 mics: two-by-two blocks of 128 (in-LDS kernel on
// the diagonal blocks, small strided MFMA GEMMs in between), needs fd_cholesky_workspace_floats(n_mics, n_bins) floats.
size_t fd_cholesky_workspace_floats(int n_mics, int n_bins);
hipError_t launch_fd_cholesky_inverse(const float* rre, const float* rim, int n_mics, int n_bins, float loading, float* lire_t, float* liim_t,
                                      int* d_status, float* d_work, size_t work_floats, hipStream_t stream);
hipError_t launch_fd_mvdr_power(const float* lire_t, const float* liim_t, const float* are, const float* aim, int n_mics, int n_dirs, int n_bins,
                                float* d_power, float* d_work, size_t work_floats, hipStream_t stream);

// Profiling build only (-DBF_STAMPS): phase totals of das_pair_kernel, see das_kernels.hip (zeros in a production build).
hipError_t read_phase_stamps(unsigned long long* out16, bool clear);

// detector post-processing (nms_kernels.hip): YOLOv5 head decode + confidence filter, greedy NMS over score-sorted candidates
hipError_t launch_yolo_decode(const void* const raw[3], const int hs[3], const int ws[3], const int strides[3], const float* anchors,
                              int batch, int nc, int format, float conf_thres, float* d_boxes, float* d_scores, int* d_cls, hipStream_t stream);
// The detector's tensor kernels (conv_kernels.hip).  elem_bytes: 2 = float16, 4 = float32 activations / weights.
// out [B][H][W][ca + cb] = (nearest 2x upsampling of a [B][H/2][W/2][ca], b [B][H][W][cb]), NHWC
hipError_t launch_upsample_concat(const void* a, const void* b, void* out, int B, int H, int W, int ca, int cb, int elem_bytes, hipStream_t stream);
// SPPF pooling inside the concatenation buffer [B][H][W][4c]: channels [c, 4c) = the three cascaded 5x5 max pools of channels [0, c)
hipError_t launch_sppf_pool(void* buf, int B, int H, int W, int c, int elem_bytes, hipStream_t stream);
// uint8 BGR frames [pixels][3] -> RGB / 255 [pixels][cpad], channels 3.. zero
hipError_t launch_preprocess_bgr8(const void* frames, void* out, long long pixels, int cpad, int elem_bytes, hipStream_t stream);
// Convolution + bias + optional SiLU, NHWC in / out, weights [N][KH][KW][C] in rows padded to whole 64-byte stages.  C a power of two >= 4,
// KW * C a whole number of 16-byte chunks.  ldy: elements between consecutive output pixels (>= N: the output may be a channel slice of a
// wider buffer); res / ldr: optional residual added to the result.  x2 / c1 / ld1 / ld2 / up1: a 1x1 window over the virtual
// concatenation (x [.., c1) at pixel pitch ld1, optionally 2x-upsampled; x2 [c1, C) at pitch ld2) -- pass nullptr, C, C, 0, 0 otherwise.
int conv_dma_switch(int value);
int conv_f32_mode(int value);
int gemm_f32_mode(int value);
int conv_weight_row(int elem_bytes, int kh, int kw, int c);
hipError_t launch_conv2d_nhwc(int elem_bytes, const void* x, const void* w, const float* bias, void* y, int B, int H, int W, int C, int N, int KH, int KW,
                              int stride, int pad, int act, int ldy, const void* res, int ldr, const void* x2, int c1, int ld1, int ld2, int up1,
                              hipStream_t stream);
hipError_t launch_topk_candidates(const float* d_scores, const float* d_boxes, const int* d_cls, int batch, int total, int K, float* d_top_scores,
                                  float* d_top_boxes, int* d_top_cls, int* d_counts, hipStream_t stream);
hipError_t launch_nms(const float* d_boxes, const float* d_scores, const int* d_cls, const int* d_counts, int batch, int K, float iou_thres,
                      int max_det, unsigned long long* d_mask, float* d_out, int* d_out_count, hipStream_t stream);

}  // namespace bf
