// Dev-only microbenchmark: VALU issue cost of wave64 f32 instructions on gfx950, in shader-clock cycles (s_memtime).
// Each wave runs ITER x 32 independent instructions of one kind (8 accumulator chains); 16 waves per CU (4 per SIMD).
// Build: hipcc --offload-arch=gfx950 -O3 scripts/dev/valu_probe.hip -o gpurun_out/valu_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void __launch_bounds__(1024) probe(float* out, long long* cycles, int iters, float seed)
{
    float a[8];
    f32x2 p[8];
    for (int i = 0; i < 8; ++i) { a[i] = seed + i + threadIdx.x; p[i] = f32x2{a[i], a[i] + 1.0f}; }
    const float k = seed * 0.5f, m = seed * 0.25f;
    const f32x2 k2{k, k}, m2{m, m};
    __syncthreads();
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if constexpr (MODE == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(k));
                if constexpr (MODE == 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(k), "v"(m));
                if constexpr (MODE == 2) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(k2));
                if constexpr (MODE == 3) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(k2), "v"(m2));
                if constexpr (MODE == 4) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(k2));
                if constexpr (MODE == 5) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(k));
                if constexpr (MODE == 6) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "s"(k), "v"(m));
                if constexpr (MODE == 7) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(k));
                if constexpr (MODE == 8) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(k));
            }
        }
    }
    const long long t1 = clock64();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}


// The consume patterns of the delay-and-sum sweep: 8 directions per trip, per direction  pad: 2 v_pk_add  /  lerp: 2 v_pk_fma
// (SGPR weight) + 2 dependent v_pk_add.  MODE 0 pad, 1 lerp as written, 2 lerp with the adds of direction j issued after
// the fmas of direction j + 1, 3 = mode 1 plus the scalar compare / not-taken branch / s_nop filler of the real loop.
template <int MODE>
__global__ void __launch_bounds__(1024) pattern(float* out, int iters, float seed, int never)
{
    f32x2 a0[8], a1[8];
    for (int i = 0; i < 8; ++i) { a0[i] = f32x2{seed + i, seed}; a1[i] = f32x2{seed, seed + i + threadIdx.x}; }
    f32x2 S0{seed, seed * 2}, S1{seed * 3, seed}, D0{seed, seed}, D1{seed * 0.5f, seed};
    unsigned long long hp = __builtin_amdgcn_readfirstlane((int)__float_as_uint(seed * 0.25f));
    int e0 = __builtin_amdgcn_readfirstlane(never), e1 = e0;
    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE == 2) {
            f32x2 t[2][2];
            asm volatile("v_pk_fma_f32 %0, %2, %3, %4 op_sel_hi:[0,1,1]\n\tv_pk_fma_f32 %1, %2, %5, %6 op_sel_hi:[0,1,1]" : "=&v"(t[0][0]), "=&v"(t[0][1]) : "s"(hp), "v"(D0), "v"(S0), "v"(D1), "v"(S1));
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (j + 1 < 8)
                    asm volatile("v_pk_fma_f32 %0, %2, %3, %4 op_sel_hi:[0,1,1]\n\tv_pk_fma_f32 %1, %2, %5, %6 op_sel_hi:[0,1,1]" : "=&v"(t[(j + 1) & 1][0]), "=&v"(t[(j + 1) & 1][1]) : "s"(hp), "v"(D0), "v"(S0), "v"(D1), "v"(S1));
                asm volatile("v_pk_add_f32 %0, %0, %2\n\tv_pk_add_f32 %1, %1, %3" : "+v"(a0[j]), "+v"(a1[j]) : "v"(t[j & 1][0]), "v"(t[j & 1][1]));
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if constexpr (MODE == 3) asm volatile("s_cmp_lg_u32 %0, %1\n\ts_cbranch_scc1 .Lx_%=\n.Ly_%=:\n\t.subsection 1\n.Lx_%=:\n\ts_nop 0\n\ts_branch .Ly_%=\n\t.subsection 0" :: "s"(e0), "s"(e1) : "scc");
                if constexpr (MODE == 0) {
                    asm volatile("v_pk_add_f32 %0, %0, %2\n\tv_pk_add_f32 %1, %1, %3" : "+v"(a0[j]), "+v"(a1[j]) : "v"(S0), "v"(S1));
                } else {
                    f32x2 t0, t1;
                    asm volatile("v_pk_fma_f32 %2, %4, %5, %6 op_sel_hi:[0,1,1]\n\tv_pk_fma_f32 %3, %4, %7, %8 op_sel_hi:[0,1,1]\n\t"
                                 "v_pk_add_f32 %0, %0, %2\n\tv_pk_add_f32 %1, %1, %3"
                                 : "+v"(a0[j]), "+v"(a1[j]), "=&v"(t0), "=&v"(t1) : "s"(hp), "v"(D0), "v"(S0), "v"(D1), "v"(S1));
                }
            }
        }
    }
    float s2 = 0;
    for (int i = 0; i < 8; ++i) s2 += a0[i].x + a0[i].y + a1[i].x + a1[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s2;
}

template <int MODE>
int run_pattern(const char* name, float* d_out)
{
    const int iters = 4000, blocks = 256, waves = 16;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(pattern<MODE>, dim3(blocks), dim3(waves * 64), 0, 0, d_out, iters, 1.0f, 7);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    }
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-34s %.3f ms -> %.2f ns per (direction, mic) step per CU\n", name, ms, ms * 1e6 / ((double)iters * 8 * waves));
    return 0;
}

template <int MODE>
int run(const char* name, float* d_out, long long* d_cyc, int waves)
{
    const int iters = 2000, blocks = 256;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(waves * 64), 0, 0, d_out, d_cyc, iters, 1.0f);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    }
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<long long> c(blocks * 16);
    CK(hipMemcpy(c.data(), d_cyc, c.size() * sizeof(long long), hipMemcpyDeviceToHost));
    double avg = 0; for (int b = 0; b < blocks; ++b) for (int w = 0; w < waves; ++w) avg += (double)c[b * 16 + w];
    avg /= (double)blocks * waves;
    const double instr = (double)iters * 32;
    // s_memtime ticks at a fixed 100 MHz on this part, so convert through the wall time instead: cycles/instr/SIMD at 2.4 GHz and the implied rate
    printf("%-14s waves/CU %2d: %.3f ms; s_memtime ticks/wave %.0f; wave-instr per SIMD per us: %.1f  -> %.2f ns per instr-slot\n", name, waves, ms, avg,
           instr * (waves / 4.0) / (ms * 1e3), ms * 1e6 / (instr * (waves / 4.0)));
    return 0;
}

int main()
{
    float* d_out; long long* d_cyc;
    CK(hipMalloc(&d_out, 256 * 1024 * sizeof(float))); CK(hipMalloc(&d_cyc, 256 * 16 * sizeof(long long)));
    if (run_pattern<0>("pad consume (2 pk_add)", d_out)) return 1;
    if (run_pattern<1>("lerp consume (2 pk_fma + 2 pk_add)", d_out)) return 1;
    if (run_pattern<2>("lerp consume, adds one step late", d_out)) return 1;
    if (run_pattern<3>("lerp consume + cmp/branch filler", d_out)) return 1;
    for (int waves : {16}) {
        if (run<0>("v_add_f32", d_out, d_cyc, waves)) return 1;
        if (run<1>("v_fma_f32", d_out, d_cyc, waves)) return 1;
        if (run<5>("v_mul_f32", d_out, d_cyc, waves)) return 1;
        if (run<6>("v_fmac sgpr", d_out, d_cyc, waves)) return 1;
        if (run<2>("v_pk_add_f32", d_out, d_cyc, waves)) return 1;
        if (run<3>("v_pk_fma_f32", d_out, d_cyc, waves)) return 1;
        if (run<4>("v_pk_mul_f32", d_out, d_cyc, waves)) return 1;
        if (run<7>("v_mov_b32", d_out, d_cyc, waves)) return 1;
        if (run<8>("v_add_u32", d_out, d_cyc, waves)) return 1;
    }
    return 0;
}
