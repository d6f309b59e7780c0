/*
 * das_oracle.c -- CPU restatement of the reference's time-domain delay-and-sum kernels.
 *
 * TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may
 * load this library; the product path (HIP kernels behind include/beamformer_hip.h) never links or calls it.
 *
 * Plain C, single thread, run-time sizes (the reference bakes N_SAMPLES / MAX_RES_X / MAX_RES_Y / N_TAPS
 * into config.h macros).  Every function names the reference lines it restates.  Loop nests, operand order
 * and the place of every rounding follow the reference so that, built with the reference's contraction
 * default (gcc -O3, -ffp-contract=fast, FMA available), the images are bit-identical to the compiled
 * reference -- tests/test_oracle_vs_ref.py checks that against oracle/_ref/ and tests/golden/.
 *
 * Status: pinned (golden vectors generated from the compiled reference, oracle/gen_golden.py).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

static int g_nsamp = 256, g_resx = 57, g_resy = 32, g_taps = 8;

/* tables, one set per process like the reference's file-scope globals */
static int *g_pad_whole;     /* pad_and_sum.c:29  */
static int *g_lerp_whole;    /* lerp_and_sum.c:33 */
static float *g_lerp_h;      /* lerp_and_sum.c:34 */
static float *g_fir_taps;    /* convolve_and_sum.c:40 */
static int *g_hyb_whole;     /* hybrid_convolve_and_sum.c:40 */
static float *g_hyb_taps;    /* hybrid_convolve_and_sum.c:41 */

void oracle_configure(int n_samples, int max_res_x, int max_res_y, int n_taps)
{
    g_nsamp = n_samples; g_resx = max_res_x; g_resy = max_res_y; g_taps = n_taps;
}

/* mean power of one steered block: pad_and_sum.c:122-131 (identical tail in all four variants) */
static float block_power(float *out, int n_mics)
{
    float sum = 0.0;
    for (int k = 0; k < g_nsamp; k++) {
        out[k] /= (float)n_mics;
        sum += powf(out[k], 2);
    }
    sum /= (float)g_nsamp;
    return sum;
}

/* ---------------------------------------------------------------- pad (integer delay) */

void oracle_load_coefficients_pad(const int *whole, int n)           /* pad_and_sum.c:147-151 */
{
    free(g_pad_whole);
    g_pad_whole = (int *)malloc((size_t)n * sizeof(int));
    memcpy(g_pad_whole, whole, (size_t)n * sizeof(int));
}

void oracle_pad_delay(const float *sig, float *out, int shift)       /* pad_and_sum.c:41-47 */
{
    for (int i = 0; i < g_nsamp - shift; i++)
        out[shift + i] += sig[i];
}

void oracle_miso_pad(const float *signals, float *out, const int *mics, int n, int offset)  /* :54-70 */
{
    memset(out, 0, (size_t)g_nsamp * sizeof(float));
    for (int m = 0; m < n; m++)
        oracle_pad_delay(signals + (size_t)mics[m] * g_nsamp, out, g_pad_whole[offset + m]);
}

void oracle_mimo_pad(const float *signals, float *image, const int *mics, int n)           /* :100-143 */
{
    float *out = (float *)malloc((size_t)g_nsamp * sizeof(float));
    for (int y = 0; y < g_resy; y++)
        for (int x = 0; x < g_resx; x++) {
            int d = y * g_resx + x;                       /* flat direction; table row d*n */
            oracle_miso_pad(signals, out, mics, n, d * n);
            image[d] = block_power(out, n);
        }
    free(out);
}

/* ---------------------------------------------------------------- lerp (integer + linear fraction) */

void oracle_load_coefficients_lerp(const float *delays, int n)       /* lerp_and_sum.c:139-153 */
{
    free(g_lerp_whole); free(g_lerp_h);
    g_lerp_whole = (int *)malloc((size_t)n * sizeof(int));
    g_lerp_h = (float *)malloc((size_t)n * sizeof(float));
    for (int i = 0; i < n; i++) {
        double ip;
        g_lerp_h[i] = 1.0 - (float)modf(delays[i], &ip);  /* h := 1 - frac, float->double->float */
        g_lerp_whole[i] = (int)ip;
    }
}

void oracle_get_lerp_tables(int *whole, float *h, int n)
{
    memcpy(whole, g_lerp_whole, (size_t)n * sizeof(int));
    memcpy(h, g_lerp_h, (size_t)n * sizeof(float));
}

void oracle_lerp_delay(const float *sig, float *out, float h, int shift)   /* lerp_and_sum.c:50-56 */
{
    for (int i = 0; i < g_nsamp - shift - 1; i++)
        out[shift + i + 1] += sig[i] + h * (sig[i + 1] - sig[i]);
}

void oracle_miso_lerp(const float *signals, float *out, const int *mics, int n, int offset)  /* :67-92 */
{
    memset(out, 0, (size_t)g_nsamp * sizeof(float));
    for (int m = 0; m < n; m++)
        oracle_lerp_delay(signals + (size_t)mics[m] * g_nsamp, out, g_lerp_h[offset + m], g_lerp_whole[offset + m]);
}

void oracle_mimo_lerp(const float *signals, float *image, const int *mics, int n)           /* :103-136 */
{
    float *out = (float *)malloc((size_t)g_nsamp * sizeof(float));
    for (int y = 0; y < g_resy; y++)
        for (int x = 0; x < g_resx; x++) {
            int d = y * g_resx + x;
            oracle_miso_lerp(signals, out, mics, n, d * n);
            image[d] = block_power(out, n);
        }
    free(out);
}

/* ---------------------------------------------------------------- convolve (T-tap FIR per direction,mic) */

void oracle_load_coefficients_convolve(const float *h, int n)        /* convolve_and_sum.c:327-331 */
{
    free(g_fir_taps);
    g_fir_taps = (float *)malloc((size_t)n * sizeof(float));
    memcpy(g_fir_taps, h, (size_t)n * sizeof(float));
}

/* zero-extended copy: T/2 zeros, the block, T - T/2 zeros (convolve_and_sum.c:199-201) */
static float *padded_copy(const float *sig)
{
    float *p = (float *)calloc((size_t)(g_nsamp + g_taps), sizeof(float));
    memcpy(p + g_taps / 2, sig, (size_t)g_nsamp * sizeof(float));
    return p;
}

void oracle_convolve_delay_naive(const float *sig, float *out, const float *h)   /* :197-211 */
{
    float *p = padded_copy(sig);
    for (int i = 0; i < g_nsamp; i++)
        for (int k = 0; k < g_taps; k++)
            out[i] += h[k] * p[i + k];
    free(p);
}

/* AVX2 form, convolve_and_sum.c:158-192: eight independent FMA lanes over the taps (lane j sees taps
 * j, j+8, ...), then the fixed `sum8` tree :132-153  ((x0+x4)+(x2+x6)) + ((x1+x5)+(x3+x7)), then out += .
 * Restated in scalar C with explicit fmaf so that no other contraction can happen. */
__attribute__((optimize("fp-contract=off")))
void oracle_convolve_delay_vectorized_add(const float *sig, const float *h, float *out)
{
    float *p = padded_copy(sig);
    for (int i = 0; i < g_nsamp; i++) {
        float lane[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int k = 0; k < g_taps; k += 8)
            for (int j = 0; j < 8; j++)
                lane[j] = fmaf(p[i + k + j], h[k + j], lane[j]);
        float q0 = lane[0] + lane[4], q1 = lane[1] + lane[5], q2 = lane[2] + lane[6], q3 = lane[3] + lane[7];
        float d0 = q0 + q2, d1 = q1 + q3;
        out[i] += d0 + d1;
    }
    free(p);
}

static void mimo_convolve(const float *signals, float *image, const int *mics, int n, int vectorized)
{
    float *out = (float *)malloc((size_t)g_nsamp * sizeof(float));
    for (int y = 0; y < g_resy; y++)
        for (int x = 0; x < g_resx; x++) {
            int d = y * g_resx + x;
            const float *taps = g_fir_taps + (size_t)d * n * g_taps;      /* :246-250 */
            memset(out, 0, (size_t)g_nsamp * sizeof(float));
            for (int m = 0; m < n; m++) {
                const float *sig = signals + (size_t)mics[m] * g_nsamp;
                if (vectorized) oracle_convolve_delay_vectorized_add(sig, taps + m * g_taps, out);
                else            oracle_convolve_delay_naive(sig, out, taps + m * g_taps);
            }
            image[d] = block_power(out, n);
        }
    free(out);
}

void oracle_mimo_convolve_naive(const float *s, float *img, const int *mics, int n)      { mimo_convolve(s, img, mics, n, 0); }  /* :231-272 */
void oracle_mimo_convolve_vectorized(const float *s, float *img, const int *mics, int n) { mimo_convolve(s, img, mics, n, 1); }  /* :295-324 */

/* ---------------------------------------------------------------- hybrid (integer pad + fractional FIR) */

#define REF_PI 3.14159265359                                         /* hybrid_convolve_and_sum.c:123 */

void oracle_compute_h_convolve(float *h, double delay)               /* hybrid_convolve_and_sum.c:124-157 */
{
    double sum = 0.0, eps = 1e-9;
    double tau = 0.5 - delay + eps;
    for (int i = 0; i < g_taps; i++) {
        double v = (double)i - ((double)g_taps - 1.0) / 2.0 - tau;
        v = sin(v * REF_PI) / (v * REF_PI);
        double n = (double)(i * 2 - g_taps + 1);
        double w = 0.42 + 0.5 * cos(REF_PI * n / ((double)(g_taps - 1)) + eps)
                        + 0.08 * cos(2.0 * REF_PI * n / ((double)(g_taps - 1) + eps));
        v *= w;
        sum += v;
        h[i] = (float)v;
    }
    for (int i = 0; i < g_taps; i++)
        h[i] /= (float)sum;
}

void oracle_load_coefficients_convolve_hybrid(const float *delays, int n)   /* :161-180 */
{
    free(g_hyb_whole); free(g_hyb_taps);
    g_hyb_taps = (float *)malloc((size_t)n * g_taps * sizeof(float));
    g_hyb_whole = (int *)malloc((size_t)n * sizeof(int));
    for (int i = 0; i < n; i++) {
        double ip, fraction = 1.0 - modf((double)delays[i], &ip);
        g_hyb_whole[i] = (int)ip;
        oracle_compute_h_convolve(g_hyb_taps + (size_t)i * g_taps, fraction);  /* the reference repeats this T times; idempotent */
    }
}

void oracle_get_hybrid_tables(int *whole, float *taps, int n)
{
    memcpy(whole, g_hyb_whole, (size_t)n * sizeof(int));
    memcpy(taps, g_hyb_taps, (size_t)n * g_taps * sizeof(float));
}

void oracle_convolve_hybrid_delay_add(const float *sig, const float *h, int shift, float *out)   /* :51-64 */
{
    float *p = padded_copy(sig);
    for (int i = 0; i < g_nsamp - shift - 1; i++)
        for (int k = 0; k < g_taps; k++)
            out[shift + i + 1] += h[k] * p[i + k];
    free(p);
}

void oracle_mimo_convolve_hybrid(const float *signals, float *image, const int *mics, int n)   /* :88-121 */
{
    float *out = (float *)malloc((size_t)g_nsamp * sizeof(float));
    for (int y = 0; y < g_resy; y++)
        for (int x = 0; x < g_resx; x++) {
            int d = y * g_resx + x;
            memset(out, 0, (size_t)g_nsamp * sizeof(float));
            for (int m = 0; m < n; m++)
                oracle_convolve_hybrid_delay_add(signals + (size_t)mics[m] * g_nsamp,
                                                 g_hyb_taps + (size_t)(d * n + m) * g_taps,
                                                 g_hyb_whole[d * n + m], out);
            image[d] = block_power(out, n);
        }
    free(out);
}

/* ---------------------------------------------------------------- ingest (PC/src/receiver.c:94-151) */

/* FPGA protocol-v2 datagrams -> mic-major float frame.  `packets` = n_samples datagrams of (8 + 4*n_microphones) bytes
 * (receiver.h:51-59).  Restates receive_and_write_to_buffer / receive_to_buffer without the socket: datagram `step` is
 * what the step-th recv() returned.  The reference's odd-row index `row + COLUMNS - x` reads stream[n_microphones] (one
 * int past the datagram) for the very last row when every array is present; that element is written as 0 here. */
void oracle_ingest(const unsigned char *packets, int n_samples, int n_microphones, int n_arrays, int rows, int columns, float *data)
{
    const size_t stride = 8 + 4 * (size_t)n_microphones;
    for (int step = 0; step < n_samples; step++) {
        const int *stream = (const int *)(packets + step * stride + 8);
        int s = 0;
        for (int n = 0; n < n_arrays; n++)
            for (int y = 0; y < rows; y++) {
                int row = n * rows * columns + y * columns;
                for (int x = 0; x < columns; x++) {
                    int idx = (y % 2) == 0 ? row + x : row + columns - x;
                    int v = idx < n_microphones ? stream[idx] : 0;
                    data[step + (size_t)n_samples * s] = (float)((double)v / 16777216.0);   /* NORM_FACTOR, config.json:52 */
                    s++;
                }
            }
    }
}

/* ---------------------------------------------------------------- helpers for tests / the bench's cpu_baseline leg */

/* mean power of a raw steered block (exposes block_power; `out` is overwritten with out/n as in the reference) */
float oracle_block_power(float *out, int n_mics) { return block_power(out, n_mics); }

/* images of the flat directions [d0, d1) only: image[d - d0].  algo: 0 pad, 1 lerp, 2 hybrid, 3 fir naive, 4 fir vectorized.
 * Same code path as the full mimo_* loops; lets tests sample directions of sizes whose full image takes minutes on a CPU. */
void oracle_mimo_range(int algo, const float *signals, float *image, const int *mics, int n, int d0, int d1)
{
    float *out = (float *)malloc((size_t)g_nsamp * sizeof(float));
    for (int d = d0; d < d1; d++) {
        switch (algo) {
        case 0: oracle_miso_pad(signals, out, mics, n, d * n); break;
        case 1: oracle_miso_lerp(signals, out, mics, n, d * n); break;
        case 2:
            memset(out, 0, (size_t)g_nsamp * sizeof(float));
            for (int m = 0; m < n; m++)
                oracle_convolve_hybrid_delay_add(signals + (size_t)mics[m] * g_nsamp, g_hyb_taps + ((size_t)d * n + m) * g_taps,
                                                 g_hyb_whole[(size_t)d * n + m], out);
            break;
        default:
            memset(out, 0, (size_t)g_nsamp * sizeof(float));
            for (int m = 0; m < n; m++) {
                const float *sig = signals + (size_t)mics[m] * g_nsamp;
                const float *h = g_fir_taps + ((size_t)d * n + m) * g_taps;
                if (algo == 4) oracle_convolve_delay_vectorized_add(sig, h, out);
                else           oracle_convolve_delay_naive(sig, out, h);
            }
        }
        image[d - d0] = block_power(out, n);
    }
    free(out);
}

void oracle_unload_all(void)
{
    free(g_pad_whole); free(g_lerp_whole); free(g_lerp_h); free(g_fir_taps); free(g_hyb_whole); free(g_hyb_taps);
    g_pad_whole = g_lerp_whole = g_hyb_whole = NULL;
    g_lerp_h = g_fir_taps = g_hyb_taps = NULL;
}
