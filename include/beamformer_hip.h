/*
 * beamformer_hip.h -- C-ABI of libbeamformer_hip.so, the MI355X (gfx950) drop-in for the reference's
 * CPU delay-and-sum path.
 *
 * PART 1 re-exports, with identical names, argument order and ownership rules, the plain-C symbols the
 * reference links into its Cython extensions (`beamformer` via PC/setup.py:22-23, `tests` via
 * PC/src/benchmark.pyx:28-55).  Each prototype cites the reference declaration it replaces.
 *
 * Differences a maintainer must know:
 *   - Sizes.  The reference bakes N_SAMPLES / MAX_RES_X / MAX_RES_Y / N_TAPS / N_MICROPHONES into config.h
 *     at build time.  Here they are run-time state: defaults are the as-shipped PC/src/config.json, replaced
 *     by bf_configure(...) or, on first use, by the JSON file named in $BF_CONFIG.
 *   - Errors.  Reference functions return void and never check anything.  These keep the void signatures;
 *     a failure (no GPU, bad size, HIP error) is printed to stderr, recorded for bf_last_error() and the
 *     output buffer is filled with NaN so it cannot pass for a result.  There is NO CPU fallback.
 *   - Process model.  The HIP context is created lazily by the first load_* / mimo_* call in the CALLING
 *     process (the reference forks its workers before loading tables: PC/src/main.pyx:172-181,707).
 *   - Tables are copied to the GPU by load_*; the caller's buffer may be freed afterwards (as in the
 *     reference, which malloc+memcpy's: PC/src/algorithms/pad_and_sum.c:147-151).
 *
 * PART 2 (bf_* names) is the extension surface: run-time configuration, device-resident / batched entry
 * points that take HIP device pointers and a stream (what bench.py and the multi-GPU path use), the steering
 * table generators of PC/src/directions.pyx, and error reporting.
 */
#ifndef BEAMFORMER_HIP_H
#define BEAMFORMER_HIP_H

#include <stdbool.h>   /* load(bool), as PC/src/api.h:4 */

#ifdef __cplusplus
extern "C" {
#endif

/* ===================================================================== PART 1: reference symbols */

/* ---- PC/src/algorithms/pad_and_sum.h:5-13 ---- */
void pad_delay(float *signal, float *out, int pos_pad);                                   /* :5  */
void miso_pad(float *signals, float *out, int *adaptive_array, int n, int offset);        /* :6  */
void miso_pad2(float *signals, float *out, int *adaptive_array, int n, int offset);       /* :7  */
void mimo_pad(float *signals, float *image, int *adaptive_array, int n);                  /* :8  */
void load_coefficients_pad(int *whole_samples, int n);                                    /* :10 */
void load_coefficients_pad2(int *whole_miso, int n);                                      /* :11 */
void unload_coefficients_pad(void);                                                       /* :12 */
void unload_coefficients_pad2(void);                                                      /* :13 */

/* ---- PC/src/algorithms/lerp_and_sum.h:4-12 ---- */
void lerp_delay(float *signal, float *out, float h, int pad);                             /* :4  */
void miso_lerp(float *signals, float *out, int *adaptive_array, int n, int offset);       /* :6  */
void mimo_lerp(float *signals, float *image, int *adaptive_array, int n);                 /* :8  */
void load_coefficients_lerp(float *delays, int n);                                        /* :10 */
void unload_coefficients_lerp(void);                                                      /* :12 */

/* ---- PC/src/algorithms/convolve_and_sum.h:4-22 (convolve_naive, :12, is declared but never defined there) ---- */
void convolve_delay_naive_add(float *signal, float *h, float *out);                       /* :4  */
void convolve_delay_vectorized(float *signal, float *h, float *out);                      /* :6  */
void convolve_delay_vectorized_add(float *signal, float *h, float *out);                  /* :8  */
void convolve_delay_naive(float *signal, float *out, float *h);                           /* :10 */
void mimo_convolve_naive(float *signals, float *image, int *adaptive_array, int n);       /* :14 */
void miso_convolve_vectorized(float *signals, float *out, int *adaptive_array, int n, int offset); /* :16 */
void mimo_convolve_vectorized(float *signals, float *image, int *adaptive_array, int n);  /* :18 */
void load_coefficients_convolve(float *h, int n);                                         /* :20 */
void unload_coefficients_convolve(void);                                                  /* :22 */

/* ---- PC/src/algorithms/hybrid_convolve_and_sum.h:4-12 ---- */
void convolve_hybrid_delay_add(float *signal, float *h, int pad, float *out);             /* :4  */
void miso_convolve_hybrid(float *signals, float *out, int *adaptive_array, int n, int offset); /* :6 */
void mimo_convolve_hybrid(float *signals, float *image, int *adaptive_array, int n);      /* :8  */
void load_coefficients_convolve_hybrid(float *h, int n);                                  /* :10 */
void unload_coefficients_convolve_hybrid(void);                                           /* :12 */

/* ---- PC/src/api.h:8-21, the shims that pair get_data() with an algorithm (PC/src/api.c:951-1104).
 * The UDP receiver / SysV ring buffer behind get_data() is out of scope; here get_data() copies the frame
 * most recently published with bf_publish_frame() (the hook a receiver process calls once per window). ---- */
void get_data(float *signals);                                                            /* api.h:7  */
void pad_mimo(float *image, int *adaptive_array, int n);                                  /* api.h:10 */
void lerp_mimo(float *image, int *adaptive_array, int n);                                 /* api.h:11 */
void convolve_mimo_naive(float *image, int *adaptive_array, int n);                       /* api.h:12 */
void convolve_mimo_vectorized(float *image, int *adaptive_array, int n);                  /* api.h:13 */
void mimo_truncated(float *image, int *adaptive_array, int n);                            /* api.h:16 */
void load_coefficients2(int *whole_samples, int n);                                       /* api.h:17 */
void miso_steer_listen(float *out, int *adaptive_array, int n, int steer_offset);         /* api.h:19 */

/* ---- PC/src/api.h:6-9,41-45: the process management around the path.  Exported so that the reference's main.pyx
 * (cdef extern block, PC/src/main.pyx:33-60) links against this library unchanged.  What they manage in the reference
 * -- the forked UDP receiver with its SysV ring buffer (api.c:874-939) and the forked PortAudio playback child
 * (api.c:491-543, 583-640) -- is live-hardware I/O outside the beamforming path:
 *   load            no receiver is forked: records an error and returns -1 (frames arrive through bf_publish_frame);
 *   stop_receiving  drops the published frame (get_data then reports "no frame published");
 *   signal_handler  no-op;
 *   load_miso       initialises the listen state exactly as miso_init_shared_memory does (api.c:461-489: n = 1,
 *                   adaptive_array all zero, steer_offset 0) and returns 0; no playback child is started --
 *                   bf_miso_listen_block below is the body of its loop;
 *   load_pa         the microphone set of the listening beam (api.c:553-567);
 *   steer           the flat table offset of the listening beam (api.c:576-581);
 *   stop_miso       clears the listen state. ---- */
int load(bool replay_mode);                                                               /* api.h:6  */
void stop_receiving(void);                                                                /* api.h:8  */
void signal_handler(void);                                                                /* api.h:9  */
int load_miso(void);                                                                      /* api.h:42 */
void load_pa(int *adaptive_array, int n);                                                 /* api.h:43 */
void stop_miso(void);                                                                     /* api.h:44 */
void steer(int offset);                                                                   /* api.h:45 */

/* ===================================================================== PART 2: extensions */

enum bf_algo {
    BF_PAD = 0,        /* mimo_pad                  */
    BF_LERP = 1,       /* mimo_lerp                 */
    BF_HYBRID = 2,     /* mimo_convolve_hybrid      */
    BF_FIR_NAIVE = 3,  /* mimo_convolve_naive       */
    BF_FIR_VEC = 4     /* mimo_convolve_vectorized  */
};

/* Run-time replacement of the config.h size macros (PC/src/config.json:3-11).  Returns 0, or -1 (see
 * bf_last_error).  Changing sizes drops every loaded table. */
int bf_configure(int n_microphones, int n_samples, int max_res_x, int max_res_y, int n_taps);
/* Same, reading the keys N_MICROPHONES, N_SAMPLES, MAX_RES_X, MAX_RES_Y, N_TAPS of section "general" from a
 * file laid out like PC/src/config.json. */
int bf_configure_from_json(const char *path);
/* Current sizes: out[0..4] = N_MICROPHONES, N_SAMPLES, MAX_RES_X, MAX_RES_Y, N_TAPS. */
void bf_get_config(int out[5]);

/* Last failure of any entry point on this thread's process ("" when none); bf_clear_error resets it. */
const char *bf_last_error(void);
void bf_clear_error(void);

/* 1 when a gfx9xx GPU is usable by this process, else 0 (never initialises a context as a side effect of
 * loading the library). */
int bf_gpu_available(void);
/* Kernel family of the last delay-and-sum launch: 0 strided, 1 quad + DPP, 2 shifted copies (sweep), 3 shifted copies
 * (direction-outer, chosen for tables without structure), 4 shifted copies (8-tap FIR), 5 shifted copies (sweep, two
 * frames per workgroup: batched launches of pad / lerp), 6 shifted copies for long blocks (256 < N_SAMPLES <= 1024: LDS image in
 * two halves, conflict-free lane mapping), 7 hybrid sweep with shared windows, two frames per workgroup (batched launches of the
 * 8-tap FIR flavours), 8 the two-frame sweep on frame-interleaved rows (batched lerp); -1 before the first launch. */
int bf_last_das_variant(void);
/* Planner A/B switches (the bits of $BF_DEBUG, das_kernels.hip plan_das) at run time, for tests and profiling; -1 returns to
 * the environment's value. */
void bf_set_debug(int flags);
/* Profiling builds only (hipcc -DBF_STAMPS, scripts/dev/phase_stamps.py): per-phase wave time of the batched pad / lerp kernel,
 * summed over all waves since the last clear: out16[0..7] = sweep, wait, staging, wait, wait, parking, wait, ordered power sum
 * (s_memtime ticks), out16[8] = waves counted.  All zero in the production build.  Returns 0 or -1. */
int bf_read_phase_stamps(unsigned long long *out16, int clear);
/* Select the HIP device (default 0, or $BF_DEVICE) before the first load_* call. */
int bf_set_device(int device);

/* Frame hand-off for the api.h shims: copies n_microphones*n_samples floats (mic-major). */
void bf_publish_frame(const float *signals);
/* One trip of the reference's playback loop (miso_loop, PC/src/api.c:505-531): get_data(), miso_pad at the offset set by
 * steer() over the microphones set by load_pa(), then out[i] = out[i] / n * mic_gain (MIC_GAIN, config.json:62).
 * out = float32 [N_SAMPLES], host pointer.  Returns 0 or -1. */
int bf_miso_listen_block(float *out, float mic_gain);
/* The listen state as steer() / load_pa() left it: returns the steer offset, *n_out = microphone count. */
int bf_get_steer(int *n_out);

/* ---- device-resident, batched delay-and-sum (the throughput path) ----
 * d_signals : HIP device pointer, float32 [frames][m_total][N_SAMPLES], mic-major
 * d_images  : HIP device pointer, float32 [frames][image_stride]; direction d of the launched range is
 *             written to d_images[f*image_stride + d - dir_begin]
 * adaptive_array / n : HOST array of the active mic rows, as in the reference calls
 * [dir_begin, dir_end) : shard of the flat direction grid 0..MAX_RES_X*MAX_RES_Y handled by this call
 * stream    : hipStream_t (0 = null stream).  Enqueue only, graph-capturable -- except the FIRST call for a (table, launch
 *             geometry) pair, which builds that geometry's digest of the table and waits for it, and a call with a new
 *             adaptive array, which synchronises the device before replacing the uploaded copy.  Up to four geometries per
 *             table stay cached (one-frame and batched calls, direction shards), so alternating callers do not rebuild.
 * Returns 0 or -1. */
int bf_das_device(int algo, const float *d_signals, int m_total, float *d_images, int image_stride, int frames,
                  const int *adaptive_array, int n, int dir_begin, int dir_end, void *stream);

/* ---- ingest: FPGA protocol-v2 datagrams -> the mic-major float32 frame the beamformers read (PC/src/receiver.c:94-151,
 * `receive_and_write_to_buffer`).  `packets` holds N_SAMPLES datagrams back to back, each
 * { u16 frequency; i8 n_arrays; i8 protocol_ver; i32 counter; i32 stream[N_MICROPHONES]; } (receiver.h:51-59).
 * Writes n_arrays*rows*columns mic rows of N_SAMPLES floats; rows/columns are the 8 x 8 tile of config.json:7-8.
 * bf_ingest takes host pointers (copies both ways); bf_ingest_device takes HIP device pointers and only enqueues.
 * The one element the reference reads past the end of the datagram (last array, last row, x = 0) is defined as 0. */
int bf_ingest(const void *packets, int n_arrays, int rows, int columns, float *frame);
int bf_ingest_device(const void *d_packets, int n_arrays, int rows, int columns, float *d_frame, void *stream);

/* bf_jet_lut: visual.py:26-49 generate_color_map("jet") -- the colour table the colourise kernel uses, uint8 [256][3]. */
void bf_jet_lut(unsigned char *out768);

/* ---- heat-map post-processing on the device (PC/src/visual.py; display side of the path) ----
 * bf_heatmap_colorize_device: visual.py:143-185 -- d_power float32 [frames][MAX_RES_X*MAX_RES_Y] -> d_small uint8
 *   [frames][MAX_RES_Y][MAX_RES_X][3] (reversed-jet colours, flipped as the reference indexes it) and should_overlay flags.
 *   Reference defaults: threshold 1e-7, amount 0.5, exponent 5.
 * bf_heatmap_overlay_device: cv2.resize(INTER_LINEAR) to out_w x out_h (:186), res = w_prev*prev + w_new*new (:450, 0.5/0.5;
 *   d_prev uint8 [out_h][out_w][3] is the carried state, updated in place), then onto the camera frames
 *   w_cam*frame + w_heat*res (:452, 0.9/0.9) when d_camera is not NULL.  d_out uint8 [frames][out_h][out_w][3].
 * bf_power_center_device: find_power_center (:295-322) -> d_centers float32 [frames][2] = (center_x, center_y);
 *   d_workspace float32 [frames][MAX_RES_X*MAX_RES_Y].
 * All take HIP device pointers and a stream and only enqueue. */
int bf_heatmap_colorize_device(const float *d_power, int frames, float threshold, float amount, float exponent,
                               unsigned char *d_small, int *d_should_overlay, void *stream);
int bf_heatmap_overlay_device(const unsigned char *d_small, int frames, int out_w, int out_h, unsigned char *d_prev,
                              const unsigned char *d_camera, unsigned char *d_out, float w_prev, float w_new, float w_cam,
                              float w_heat, void *stream);
int bf_power_center_device(const float *d_power, int frames, float *d_centers, float *d_workspace, void *stream);

/* ---- frequency-domain beamformers (device pointers, enqueue only) ----
 * Delay-and-sum by phase steering: PC/application/realtime_scripts/beam_forming_algorithm.py:30-70 with the steering
 * phasors of calc_phase_shift_cartesian.py:37-50.  MVDR has no counterpart in the reference (builder-defined, see
 * DESIGN.md).  n_bins rFFT bins starting at bin_lo; planes are float32, complex values as separate re / im arrays.
 *   bf_fd_steering_device   a[k][m][d] = exp(-j 2 pi freq[k] tau[d][m])   (tau seconds, float64 [D][M]; freq Hz, float64 [K])
 *   bf_fd_dft_device        rfft of every active mic row of every frame -> X as [K][M][F] and as [K][F][M]
 *   bf_fd_das_power_device  P[f][d] = sum_k | sum_m X[k][m][f] a[k][m][d] |^2                                    -> float32 [F][D]
 *   bf_fd_covariance_device R[k] = (1/F) sum_f x x^H                                                              -> [K][M][M]
 *   bf_fd_cholesky_inverse_device  R += loading*tr(R)/M*I = L L^H;  writes inverse(L) transposed [K][col][row];  M <= 256 (two blocks of 128 above 128);
 *                           d_status int32 [K] (zeroed by the caller) receives j+1 where a pivot was not positive
 *   bf_fd_mvdr_power_device P[d] = sum_k 1 / || inverse(L_k) a[k][:, d] ||^2                                     -> float32 [D]
 *                           d_lire_t / d_liim_t are the planes bf_fd_cholesky_inverse_device writes, [K][col][row] of a LOWER-TRIANGULAR inverse: entries with
 *                           col > row are taken to be zero and whole 32 x 32 blocks of them are not multiplied at all
 *   bf_fd_gemm_f32_mode     how the two bin-reducing GEMMs (bf_fd_das_power_device, bf_fd_mvdr_power_device) multiply: 0 = v_mfma_f32_32x32x2_f32,
 *                           1 = exact three-way bfloat16 split of the float32 operands, six part products per product on v_mfma_f32_32x32x16_bf16,
 *                           float32 accumulation (float32 accuracy: tests/test_freqdomain.py holds both modes to the same bounds).  mode < 0 only
 *                           asks; returns the previous setting ($BF_GEMM_F32=native|split sets the initial one)  */
int bf_fd_gemm_f32_mode(int mode);
int bf_fd_steering_device(const double *d_tau, const double *d_freq, int n_dirs, int n_mics, int n_bins, float *d_are, float *d_aim, void *stream);
int bf_fd_dft_device(const float *d_frames, int m_total, int frames, const int *adaptive_array, int n, int bin_lo, int n_bins,
                     float *d_xre_mf, float *d_xim_mf, float *d_xre_fm, float *d_xim_fm, void *stream);
int bf_fd_das_power_device(const float *d_xre_mf, const float *d_xim_mf, const float *d_are, const float *d_aim, int frames, int n_mics,
                           int n_dirs, int n_bins, float *d_power, void *stream);
int bf_fd_covariance_device(const float *d_xre_fm, const float *d_xim_fm, int frames, int n_mics, int n_bins, float *d_rre, float *d_rim, void *stream);
int bf_fd_cholesky_inverse_device(const float *d_rre, const float *d_rim, int n_mics, int n_bins, float loading, float *d_lire_t, float *d_liim_t,
                                  int *d_status, void *stream);
int bf_fd_mvdr_power_device(const float *d_lire_t, const float *d_liim_t, const float *d_are, const float *d_aim, int n_mics, int n_dirs,
                            int n_bins, float *d_power, void *stream);

/* ---- detector post-processing (device pointers, enqueue only).  The reference obtains boxes from
 * ultralytics.YOLO(...).predict (image-detection/src/yolo_smooth_tracking.py:9-23); these are the published YOLOv5 head
 * decode and non-maximum suppression it performs internally.
 *   bf_yolo_decode_device: raw[l] = head output of level l, [batch][3*(5+nc)][h[l]][w[l]] (format bit 0: float16 maps, else float32;
 *       bit 1: the maps are NHWC, [batch][h][w][3*(5+nc)] -- the channels_last memory the detect convolutions write);
 *       anchors float32 [3][3][2] (pixels, HOST pointer); writes xyxy boxes [batch][T][4], scores [batch][T] (obj*cls, or -1
 *       when under conf_thres) and class ids [batch][T], T = 3 * sum(h*w).
 *   bf_nms_device: boxes / scores / cls of the K best candidates per image, already sorted by descending score, counts[b] valid
 *       entries; d_mask workspace uint64 [batch][K][ceil(K/64)]; writes up to max_det rows [x1,y1,x2,y2,score,cls] per image
 *       and the number kept.  K <= 4096. */
int bf_yolo_decode_device(const void *const raw[3], const int h[3], const int w[3], const int strides[3], const float *anchors, int batch, int nc,
                          int format, float conf_thres, float *d_boxes, float *d_scores, int *d_cls, void *stream);
/*   bf_topk_candidates_device: the k (<= 1024) best-scoring boxes of every image in descending score order (ties: lower box index
 *       first) -- d_top_scores [batch][k], d_top_boxes [batch][k][4], d_top_cls [batch][k], d_counts [batch] = entries with a
 *       positive score (decode marks rejected boxes with -1).  One workgroup per image: radix select, ordered tie admission,
 *       bitonic sort in LDS.  This is the candidate list bf_nms_device walks. */
int bf_topk_candidates_device(const float *d_scores, const float *d_boxes, const int *d_cls, int batch, int total, int k, float *d_top_scores,
                              float *d_top_boxes, int *d_top_cls, int *d_counts, void *stream);
/*   bf_upsample_concat_device: the head's torch.cat((nn.Upsample(2, "nearest")(a), b), 1) in one pass -- a float16 NHWC [batch][h/2][w/2][ca],
 *       b [batch][h][w][cb], out [batch][h][w][ca + cb]; h, w even, ca, cb multiples of 8. */
int bf_upsample_concat_device(const void *d_a, const void *d_b, void *d_out, int batch, int h, int w, int ca, int cb, void *stream);
/*   bf_sppf_pool_device: the SPPF block's pooling inside its concatenation buffer, float16 NHWC [batch][h][w][4*c]: channels [c, 2c),
 *       [2c, 3c), [3c, 4c) become the 5x5 / stride 1 / pad 2 max pool of channels [0, c) applied once, twice and three times
 *       (nn.MaxPool2d(5, 1, 2) cascaded; exact).  c a multiple of 8, h * w <= 2048. */
int bf_sppf_pool_device(void *d_buf, int batch, int h, int w, int c, void *stream);
/*   bf_preprocess_bgr8_device: camera frames uint8 [batch][h][w][3] BGR (as OpenCV delivers them, main.pyx:632) -> the network's input,
 *       float16 NHWC [batch][h][w][cpad], RGB / 255 in channels 0..2, zeros above (cpad = 4: what the stem convolution reads). */
int bf_preprocess_bgr8_device(const void *d_frames, void *d_out, int batch, int h, int w, int cpad, void *stream);
/*   bf_conv2d_nhwc_f16_device: the detector's convolutions (the network behind ultralytics.YOLO, yolo_smooth_tracking.py:9-23) as an
 *       implicit GEMM on the f16 matrix cores: y[b][ho][wo][n] = act(bias[n] + sum x[b][ho*stride-pad+i][wo*stride-pad+j][c] * w[n][i][j][c]),
 *       x float16 NHWC [batch][h][w][c]; w float16 [n][kh][kw][c], every output channel's kh*kw*c values followed by zeros up to a
 *       multiple of 64 = whole 128-byte stages (bf_conv2d_weight_row(kh, kw, c) halfs per row); bias float32 [n] or NULL; y float16 NHWC
 *       [batch][(h+2*pad-kh)/stride+1][(w+2*pad-kw)/stride+1][n]; silu != 0 applies x*sigmoid(x) (f32 accumulation throughout).
 *       c a power of two >= 4 with kw * c a multiple of 8 (pad a 3-channel image with one zero channel; 4 channels need even
 *       stride, pad and width). */
int bf_conv2d_weight_row(int kh, int kw, int c);
/*   bf_conv2d_use_dma_kernel: the convolution entry points have two kernels behind them -- operand tiles staged by LDS-DMA (the default wherever
 *       every operand tensor is smaller than 2 GiB) or through registers.  enable = 1 / 0 selects (2: as 1, and the stem's patch kernel in either precision), < 0 only asks; returns the previous setting
 *       ($BF_CONV_DMA=0 sets the initial one).  Results of the two are bit-identical (same products, same summation order). */
int bf_conv2d_use_dma_kernel(int enable);
/*   bf_conv2d_f32_mode: how the float32 LDS-DMA convolution kernels multiply.  0 = the float32 matrix instruction on the operands as they are
 *       (v_mfma_f32_32x32x2_f32); 1 = every operand value split exactly into three bfloat16 parts and the product accumulated (in float32) from the
 *       six part products that matter (v_mfma_f32_32x32x16_bf16): float32 accuracy -- measured against float64 in tests/test_detector.py -- at
 *       2.67 x the matrix rate.  Differences from the float32 instruction: the summation order, an infinite operand gives NaN (inf - inf in the split)
 *       where the instruction gives an infinity, and operands below 2^-118 lose their low parts to bfloat16's exponent range (float32's own
 *       subnormal neighbourhood).  mode < 0 only asks; returns the previous setting ($BF_CONV_F32=native|split sets the initial one). */
int bf_conv2d_f32_mode(int mode);
int bf_conv2d_nhwc_f16_device(const void *d_x, const void *d_w, const float *d_bias, void *d_y, int batch, int h, int w, int c, int n, int kh, int kw,
                              int stride, int pad, int silu, void *stream);
/*   bf_conv2d_nhwc_f16_into_device: the same, writing into a channel slice of a wider NHWC buffer -- d_y points at the slice's first
 *       channel of pixel 0, ldy = halfs between consecutive pixels of that buffer (the network's torch.cat of convolution outputs
 *       becomes free) -- and optionally adding a residual tensor (d_res, row stride ldr; NULL for none) to the rounded result the
 *       way two float16 tensors are added (the bottlenecks' x + cv2(cv1(x))). */
int bf_conv2d_nhwc_f16_into_device(const void *d_x, const void *d_w, const float *d_bias, void *d_y, int ldy, const void *d_res, int ldr, int batch, int h,
                                   int w, int c, int n, int kh, int kw, int stride, int pad, int silu, void *stream);
/*   float32 forms of the detector's tensor kernels -- the precision ultralytics' predict runs at by default
 *       (yolo_smooth_tracking.py:13-23 passes no half=): the same kernels on float32 NHWC tensors, the convolution's products exact and its sums in f32 in either
 *       float32 mode (bf_conv2d_f32_mode above: three-way bfloat16 operand split on v_mfma_f32_32x32x16_bf16, the default, or v_mfma_f32_32x32x2_f32),
 *       SiLU as x / (1 + expf(-x)).
 *       Weight rows are padded to a multiple of 32 floats (bf_conv2d_weight_row_f32); c a power of two >= 4. */
int bf_conv2d_weight_row_f32(int kh, int kw, int c);
int bf_conv2d_nhwc_f32_device(const void *d_x, const void *d_w, const float *d_bias, void *d_y, int batch, int h, int w, int c, int n, int kh, int kw,
                              int stride, int pad, int silu, void *stream);
int bf_conv2d_nhwc_f32_into_device(const void *d_x, const void *d_w, const float *d_bias, void *d_y, int ldy, const void *d_res, int ldr, int batch, int h,
                                   int w, int c, int n, int kh, int kw, int stride, int pad, int silu, void *stream);
int bf_upsample_concat_f32_device(const void *d_a, const void *d_b, void *d_out, int batch, int h, int w, int ca, int cb, void *stream);
int bf_sppf_pool_f32_device(void *d_buf, int batch, int h, int w, int c, void *stream);
int bf_preprocess_bgr8_f32_device(const void *d_frames, void *d_out, int batch, int h, int w, int cpad, void *stream);
/*   bf_conv1x1_cat_nhwc_{f16,f32}_device: a 1x1 convolution (+ bias, SiLU, slice output, residual as above) whose input is a VIRTUAL
 *       concatenation, so that the network's torch.cat((a, b), 1) and torch.cat((upsample(a), b), 1) in front of a 1x1 layer cost no
 *       pass of their own: channels [0, c1) of pixel (b, y, x) are read from d_x1 (ld1 elements between consecutive pixels: a dense
 *       tensor or a channel slice of a wider NHWC buffer; up1 != 0: d_x1 is [batch][h/2][w/2] and pixel (y/2, x/2) is read =
 *       nn.Upsample(2, "nearest")), channels [c1, c) from d_x2 (pitch ld2; may be NULL when c1 == c).  c1, ld1, ld2 whole 16-byte
 *       chunks, pointers 16-byte aligned.  d_w: [n][c] rows as bf_conv2d_weight_row(_f32)(1, 1, c). */
int bf_conv1x1_cat_nhwc_f16_device(const void *d_x1, int ld1, int c1, int up1, const void *d_x2, int ld2, const void *d_w, const float *d_bias, void *d_y,
                                   int ldy, const void *d_res, int ldr, int batch, int h, int w, int c, int n, int silu, void *stream);
int bf_conv1x1_cat_nhwc_f32_device(const void *d_x1, int ld1, int c1, int up1, const void *d_x2, int ld2, const void *d_w, const float *d_bias, void *d_y,
                                   int ldy, const void *d_res, int ldr, int batch, int h, int w, int c, int n, int silu, void *stream);
/*   bf_letterbox_bgr8_device: what ultralytics' predict does to a frame before the network (yolo_smooth_tracking.py:13-23): d_frame uint8 [h][w][3]
 *       resized with cv2.resize(INTER_LINEAR) semantics (8-bit fixed-point bilinear, half-pixel centres) to new_w x new_h and placed at (top, left) of
 *       d_out uint8 [out_h][out_w][3], whose other pixels get the border value (114 in ultralytics).  new_h == h and new_w == w copies. */
int bf_letterbox_bgr8_device(const unsigned char *d_frame, int h, int w, unsigned char *d_out, int out_h, int out_w, int new_h, int new_w, int top, int left,
                             int value, void *stream);
int bf_nms_device(const float *d_boxes, const float *d_scores, const int *d_cls, const int *d_counts, int batch, int k, float iou_thres, int max_det,
                  unsigned long long *d_mask, float *d_out, int *d_out_count, void *stream);

/* Launch geometry the planner picks for a call like the above (no GPU needed): out[0..9] = nc, lead,
 * row_stride, mic_chunk, n_chunks, waves, dpw, tile_dirs, n_tiles, lds_bytes.  Returns 0 or -1. */
int bf_plan_das(int algo, int n, int frames, int dir_begin, int dir_end, int max_whole, int n_cus, long long out[10]);

/* Copies of the tables as resident on the GPU after load_coefficients_lerp / _convolve_hybrid (the reference
 * keeps them in file-scope globals: lerp_and_sum.c:33-34, hybrid_convolve_and_sum.c:40-41). */
int bf_get_lerp_tables(int *whole, float *h, int n);
int bf_get_hybrid_tables(int *whole, float *taps, int n);

/* ---- steering tables: PC/src/directions.pyx, bit-exact, host C++ (float64 with the float32-typed
 * constants of config.pxd:14-23) ---- */
typedef struct bf_geometry {
    int rows, columns;          /* ROWS, COLUMNS                    config.json:7-8   */
    int arrays;                 /* _ACTIVE_MICS (8x8 tiles)         directions.pyx:16 */
    int skip_n_mics;            /* SKIP_N_MICS                      config.json:20    */
    float sample_rate;          /* SAMPLE_RATE                      config.json:16    */
    float propagation_speed;    /* PROPAGATION_SPEED                config.json:21    */
    float element_distance;     /* ELEMENT_DISTANCE                 config.json:17    */
    float view_angle;           /* VIEW_ANGLE                       config.json:13    */
    float z;                    /* Z                                config.json:11    */
} bf_geometry;
void bf_default_geometry(bf_geometry *g);                                   /* as-shipped config.json values */
/* directions.pyx:35-87; `unused`/`n_unused` = contents of unused_mics.npy (already offset), may be NULL/0.
 * Writes up to rows*columns*arrays sorted indices, returns how many. */
int bf_active_microphones(const bf_geometry *g, const int *unused, int n_unused, int *active_out);
/* directions.pyx:17-32; r_prime_out = float64 [2][n_active]. Returns n_active. */
int bf_calc_r_prime(const bf_geometry *g, const int *unused, int n_unused, double *r_prime_out);
/* directions.pyx:90-124; delays_out = float64 [max_res_x][max_res_y][n_active]. Returns n_active or -1. */
int bf_calculate_delays(const bf_geometry *g, int max_res_x, int max_res_y, const int *unused, int n_unused, double *delays_out);
/* directions.pyx:189-205 / :207-226; taps_out = float64[8] / float32[n_taps]. */
void bf_get_h(double frac, double *taps_out);
void bf_get_h2(double delay, int n_taps, float *taps_out);
/* The Python loops over those two (directions.pyx:240-243, :272-275) as one call: float32 [n][8] / [n][n_taps]. */
void bf_get_h_batch(const double *frac, long long n, float *taps_out);
void bf_get_h2_batch(const double *delay, long long n, int n_taps, float *taps_out);

#ifdef __cplusplus
}
#endif
#endif /* BEAMFORMER_HIP_H */
