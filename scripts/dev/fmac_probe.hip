// Dev-only microbenchmark (MI355X): how fast does a SIMD retire FIR-style fma chains?  16 waves per CU, each wave runs ITER x 64
// scalar-tap multiply-accumulates on NCH independent accumulator chains (the hybrid kernel has 2 per EXEC mask), as v_fmac_f32 with
// an SGPR tap or as v_pk_fma_f32 on register pairs, with or without EXEC mask switches every 16 instructions.
// Build: hipcc --offload-arch=gfx950 -O3 scripts/dev/fmac_probe.hip -o scripts/dev/bin/fmac_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int NCH, bool PK, bool MASK>
__global__ void __launch_bounds__(1024) k(float* out, int iters, float seed, int hbits, int sh)
{
    float a[8];
    f32x2 p[8];
    for (int i = 0; i < 8; ++i) { a[i] = seed + i + threadIdx.x; p[i] = f32x2{a[i], a[i] + 1.0f}; }
    float w[4] = {seed * 0.5f + threadIdx.x, seed * 0.25f, seed * 0.125f + threadIdx.x, seed};
    f32x2 wp[4] = {f32x2{w[0], w[1]}, f32x2{w[1], w[2]}, f32x2{w[2], w[3]}, f32x2{w[3], w[0]}};
    const int hb = __builtin_amdgcn_readfirstlane(hbits), s0 = __builtin_amdgcn_readfirstlane(sh);
    unsigned long long m0 = ~0ull << s0, m1 = ~0ull << (s0 + 1), sv;
    const unsigned long long hp = (unsigned long long)(unsigned)hb;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int blk = 0; blk < 4; ++blk) {
            if constexpr (MASK) asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, %1" : "=&s"(sv) : "s"(blk & 1 ? m1 : m0));
#pragma unroll
            for (int t = 0; t < 16 / (PK ? 2 : 1); ++t) {
                const int c = t % NCH;
                if constexpr (PK) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(p[c]) : "s"(hp), "v"(wp[t & 3]));
                else asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[c]) : "s"(hb), "v"(w[t & 3]));
            }
            if constexpr (MASK) asm volatile("s_mov_b64 exec, %0" :: "s"(sv));
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NCH, bool PK, bool MASK>
int run(const char* name, float* d_out)
{
    const int iters = 40000, blocks = 256;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k<NCH, PK, MASK>), dim3(blocks), dim3(1024), 0, 0, d_out, iters, 1.0f, 0x3a83126f, 1);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    // multiply-accumulates per SIMD: 4 waves x iters x 64
    const double macs = 4.0 * iters * 64;
    printf("%-44s %8.3f ms  %.3f ns per wave-MAC-instruction-equivalent per SIMD  (%.1f%% of 2 cycles @ 2.4 GHz)\n", name, best, best * 1e6 / macs,
           100.0 * (2.0 / 2.4) / (best * 1e6 / macs));
    return 0;
}

int main()
{
    float* d_out; CK(hipMalloc(&d_out, 256 * 1024 * sizeof(float)));
    if (run<8, false, false>("v_fmac sgpr, 8 chains", d_out)) return 1;
    if (run<4, false, false>("v_fmac sgpr, 4 chains", d_out)) return 1;
    if (run<2, false, false>("v_fmac sgpr, 2 chains", d_out)) return 1;
    if (run<1, false, false>("v_fmac sgpr, 1 chain", d_out)) return 1;
    if (run<2, false, true>("v_fmac sgpr, 2 chains, EXEC switch per 16", d_out)) return 1;
    if (run<4, true, false>("v_pk_fma sgpr, 4 chains", d_out)) return 1;
    if (run<2, true, false>("v_pk_fma sgpr, 2 chains", d_out)) return 1;
    if (run<1, true, false>("v_pk_fma sgpr, 1 chain", d_out)) return 1;
    if (run<1, true, true>("v_pk_fma sgpr, 1 chain, EXEC switch per 8", d_out)) return 1;
    return 0;
}
