// directions.cpp -- steering geometry and delay tables (host side of the drop-in; PC/src/directions.pyx).
//
// The reference computes these once per process in Cython/NumPy before the frame loop
// (PC/src/main.pyx:172-181); the integer/fractional tables derived from them are what the GPU kernels index,
// so they must match the reference bit for bit.  That pins three things this file reproduces on purpose:
//   * the C `float` typing of the config macros seen through config.pxd:14-23 (fs/c is a float32 division,
//     d = double(float32(ELEMENT_DISTANCE)), z*z is a float32 product, alpha/2 is double(float32)/2.0);
//   * NumPy's linspace arithmetic (step = (stop-start)/(num-1); y[i] = i*step + start; y[last] = stop);
//   * NumPy's left-to-right, one-rounding-per-operation float64 broadcasting -- so this translation unit is
//     compiled with -ffp-contract=off and uses no fused operations.
// libm's tan/sin/cos are used where NumPy calls its own; they agree for the as-shipped VIEW_ANGLE and for
// every fixture in tests/golden (NumPy's AVX-512 tan differs from glibc's in the last ulp for ~0.5 % of
// arguments -- see DESIGN.md "Known numerical caveats").
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "../../include/beamformer_hip.h"

namespace {

constexpr double kPi = 3.141592653589793;  // numpy.pi

std::vector<int> active_list(const bf_geometry& g, const int* unused, int n_unused)
{
    // directions.pyx:35-87.  The index matrix is [rows][columns*arrays]: tile a occupies columns
    // a*columns .. (a+1)*columns-1 and numbers its elements row-major from a*rows*columns.
    const int per = g.rows * g.columns;
    const int skip = g.skip_n_mics > 0 ? g.skip_n_mics : 1;
    std::vector<int> out;
    for (int r = 0; r < g.rows; r += skip)
        for (int c = 0; c < g.columns * g.arrays; c += skip) {
            const int tile = c / g.columns, col = c % g.columns;
            const int mic = tile * per + r * g.columns + col;
            bool drop = false;
            for (int u = 0; u < n_unused; ++u) drop |= (unused[u] == mic);
            if (!drop) out.push_back(mic);
        }
    std::sort(out.begin(), out.end());
    return out;
}

// directions.pyx:17-32 for ALL rows*columns*arrays mics (selection by the active list happens in the callers).
void all_positions(const bf_geometry& g, double d, std::vector<double>& x, std::vector<double>& y)
{
    const double half = d / 2;
    const int total = g.rows * g.columns * g.arrays;
    x.assign(total, 0.0);
    y.assign(total, 0.0);
    int e = 0;
    for (int a = 0; a < g.arrays; ++a) {
        const int tile = -a;
        for (int row = 0; row < g.rows; ++row)
            for (int col = 0; col < g.columns; ++col, ++e) {
                // -col*d - half + tile*COLUMNS*d + tile*0 + COLUMNS*arrays*half, evaluated left to right
                double v = (double)(-col) * d;
                v = v - half;
                v = v + (double)(tile * g.columns) * d;
                v = v + 0.0;
                v = v + (double)(g.columns * g.arrays) * half;
                x[e] = v - 0.0;                                            // r_prime[0,:] -= arrays*0/2
                double w = (double)row * d;
                w = w - (double)g.rows * half;
                y[e] = w + half;
            }
    }
}

void linspace(double start, double stop, int num, std::vector<double>& out)
{
    out.resize(num);
    if (num <= 0) return;
    const int div = num - 1;
    const double delta = stop - start;
    if (div > 0) {
        const double step = delta / (double)div;
        if (step == 0.0) {
            for (int i = 0; i < num; ++i) out[i] = ((double)i / (double)div) * delta + start;
        } else {
            for (int i = 0; i < num; ++i) out[i] = (double)i * step + start;
        }
        out[num - 1] = stop;
    } else {
        out[0] = 0.0 * delta + start;
    }
}

}  // namespace

extern "C" {

void bf_default_geometry(bf_geometry* g)
{
    g->rows = 8; g->columns = 8; g->arrays = 4; g->skip_n_mics = 1;
    g->sample_rate = 48828.0f; g->propagation_speed = 340.0f; g->element_distance = 0.02f;
    g->view_angle = 59.0f; g->z = 1.0f;
}

int bf_active_microphones(const bf_geometry* g, const int* unused, int n_unused, int* active_out)
{
    const std::vector<int> a = active_list(*g, unused, n_unused);
    if (active_out) std::memcpy(active_out, a.data(), a.size() * sizeof(int));
    return (int)a.size();
}

int bf_calc_r_prime(const bf_geometry* g, const int* unused, int n_unused, double* r_prime_out)
{
    const std::vector<int> a = active_list(*g, unused, n_unused);
    std::vector<double> x, y;
    all_positions(*g, (double)g->element_distance, x, y);
    const size_t n = a.size();
    for (size_t i = 0; i < n; ++i) {
        r_prime_out[i] = x[a[i]];
        r_prime_out[n + i] = y[a[i]];
    }
    return (int)n;
}

int bf_calculate_delays(const bf_geometry* g, int X, int Y, const int* unused, int n_unused, double* out)
{
    if (X < 1 || Y < 1) return -1;
    const std::vector<int> act = active_list(*g, unused, n_unused);
    const int M = (int)act.size();
    if (M == 0) return 0;
    std::vector<double> px, py;
    all_positions(*g, (double)g->element_distance, px, py);
    std::vector<double> xi(M), yi(M);
    for (int m = 0; m < M; ++m) { xi[m] = px[act[m]]; yi[m] = py[act[m]]; }

    const double aspect = 16.0 / 9.0;                                                   // directions.pyx:101
    const double x_max = (double)g->z * std::tan(((double)g->view_angle / 2.0) * kPi / 180);  // :109
    const double y_max = x_max / aspect;
    std::vector<double> xs, ys;
    linspace(-x_max, x_max, X, xs);
    linspace(-y_max, y_max, Y, ys);
    const float zf = g->z * g->z;                       // powf(z_scan, 2) in float
    const double zz = (double)zf;
    const float kf = g->sample_rate / g->propagation_speed;  // float32 division
    const double k = (double)kf;

    std::vector<double> xprod(M);
    for (int ix = 0; ix < X; ++ix) {
        const double x = xs[ix];
        for (int m = 0; m < M; ++m) xprod[m] = x * xi[m];
        const double xx = x * x;
        for (int iy = 0; iy < Y; ++iy) {
            const double y = ys[iy];
            double r = xx + y * y;
            r = std::sqrt(r + zz);
            double* row = out + ((size_t)ix * Y + iy) * M;
            double lo = 0.0;
            for (int m = 0; m < M; ++m) {
                double v = xprod[m] + y * yi[m];
                v = k * v;
                v = v / r;
                row[m] = v;
                lo = (m == 0 || v < lo) ? v : lo;
            }
            for (int m = 0; m < M; ++m) row[m] = row[m] - lo;
        }
    }
    return M;
}

void bf_get_h(double frac, double* taps)
{
    // directions.pyx:189-205 (window length hard-coded to 8 there)
    const double eps = 1e-9, tau = -frac;
    double h[8], sum = 0.0;
    for (int n = 0; n < 8; ++n) {
        double a = (double)n - (8 - 1) / 2.0;
        a = a - (0.5 + tau);
        a = a + eps;
        const double s = std::sin(a * kPi) / (a * kPi);
        double w = 0.42 - 0.5 * std::cos(2 * kPi * (double)n / 8);
        w = w + 0.08 * std::cos(4 * kPi * (double)n / 8);
        h[n] = s * w;
    }
    // numpy.sum over 8 doubles uses pairwise summation with an 8-way unrolled head: for n == 8 that is
    // ((h0+h1)+(h2+h3)) + ((h4+h5)+(h6+h7))
    sum = ((h[0] + h[1]) + (h[2] + h[3])) + ((h[4] + h[5]) + (h[6] + h[7]));
    for (int n = 0; n < 8; ++n) taps[n] = h[n] / sum;
}

void bf_get_h2(double delay, int T, float* taps)
{
    // directions.pyx:207-226
    const double eps = 1e-9;
    const double tau = 0.5 - delay + eps;
    double sum = 0.0;
    for (int i = 0; i < T; ++i) {
        double v = (double)i - (double)(T - 1) / 2;
        v = v - tau;
        v = std::sin(v * kPi) / (v * kPi);
        const double n = (double)(i * 2 - T + 1);
        double w = 0.42 + 0.5 * std::cos(kPi * n / ((double)(T - 1) + eps));
        w = w + 0.08 * std::cos(2 * kPi * n / ((double)(T - 1) + eps));
        v = v * w;
        sum = sum + v;
        taps[i] = (float)v;
    }
    // `h /= sum_`: h is float32, sum_ a numpy.float64 scalar -> NumPy divides in float64 and rounds the
    // quotient back to float32 (the C twin, hybrid_convolve_and_sum.c:153-156, divides in float32 instead).
    for (int i = 0; i < T; ++i) taps[i] = (float)((double)taps[i] / sum);
}

// Whole-table forms of the two tap generators (the reference loops over them in Python,
// directions.pyx:240-243,272-275).  taps_out is float32 [n][8] / [n][n_taps].
void bf_get_h_batch(const double* frac, long long n, float* taps_out)
{
    double h[8];
    for (long long i = 0; i < n; ++i) {
        bf_get_h(frac[i], h);
        for (int k = 0; k < 8; ++k) taps_out[i * 8 + k] = (float)h[k];
    }
}

void bf_get_h2_batch(const double* delay, long long n, int n_taps, float* taps_out)
{
    for (long long i = 0; i < n; ++i) bf_get_h2(delay[i], n_taps, taps_out + i * n_taps);
}

}  // extern "C"
