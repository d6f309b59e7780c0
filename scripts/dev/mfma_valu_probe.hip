// Do vector instructions run in the shadow of a multi-pass MFMA on gfx950?  (dev probe; hipcc --offload-arch=gfx950 -O3 mfma_valu_probe.hip -o bin/mfma_valu_probe)
// Four loops of the same instruction counts per iteration -- 24 v_mfma_f32_32x32x16_bf16 (four accumulators in rotation) and 144 v_pk_add_f32 --
// as: MFMAs only, vector only, "run of MFMAs then run of vector", "one MFMA then six vector", at one and at two waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef float f2 __attribute__((ext_vector_type(2)));

#define MFMA(C) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(C) : "v"(a), "v"(b))
template <int kOp> __device__ __forceinline__ void valu(f2& x, const f2& one)
{
    if (kOp == 0) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(x) : "v"(one));
    if (kOp == 1) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[0]) : "v"(one[0]));
    if (kOp == 2) asm volatile("v_and_b32 %0, %0, %1" : "+v"(x[0]) : "v"(one[0]));
    if (kOp == 3) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(x[0]) : "v"(one[0]));
    if (kOp == 4) asm volatile("v_lshlrev_b32 %0, 16, %0" : "+v"(x[0]));
}
#define VALU(X) valu<kOp>(X, one)
#define V6 VALU(x0); VALU(x1); VALU(x2); VALU(x3); VALU(x4); VALU(x5)

template <int kMode, int kOp> __global__ void __launch_bounds__(256) probe(float* out, int iters)
{
    f16v c0 = {}, c1 = {}, c2 = {}, c3 = {};
    bf8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x + i); b[i] = (__bf16)1.0f; }
    f2 x0 = {1, 2}, x1 = {3, 4}, x2 = {5, 6}, x3 = {7, 8}, x4 = {9, 10}, x5 = {11, 12}, one = {1.0f, 1.0f};
    for (int it = 0; it < iters; ++it) {
        if (kMode == 0) { for (int n = 0; n < 6; ++n) { MFMA(c0); MFMA(c1); MFMA(c2); MFMA(c3); } }
        if (kMode == 1) { for (int n = 0; n < 24; ++n) { V6; } }
        if (kMode == 2) { for (int n = 0; n < 6; ++n) { MFMA(c0); MFMA(c1); MFMA(c2); MFMA(c3); } for (int n = 0; n < 24; ++n) { V6; } }
        if (kMode == 3) { for (int n = 0; n < 6; ++n) { MFMA(c0); V6; MFMA(c1); V6; MFMA(c2); V6; MFMA(c3); V6; } }
    }
    float s = x0[0] + x1[0] + x2[0] + x3[0] + x4[0] + x5[0];
    for (int e = 0; e < 16; ++e) s += c0[e] + c1[e] + c2[e] + c3[e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main()
{
    float* out; hipMalloc(&out, 2048 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    const char* names[4] = {"24 MFMA", "144 VALU", "24 MFMA then 144 VALU", "24 x (MFMA, 6 VALU)"};
    const char* ops[5] = {"v_pk_add_f32", "v_add_f32", "v_and_b32", "v_cvt_pk_bf16_f32", "v_lshlrev_b32"};
    typedef void (*K)(float*, int);
    K kernels[5][4] = {{probe<0, 0>, probe<1, 0>, probe<2, 0>, probe<3, 0>}, {probe<0, 1>, probe<1, 1>, probe<2, 1>, probe<3, 1>}, {probe<0, 2>, probe<1, 2>, probe<2, 2>, probe<3, 2>},
                       {probe<0, 3>, probe<1, 3>, probe<2, 3>, probe<3, 3>}, {probe<0, 4>, probe<1, 4>, probe<2, 4>, probe<3, 4>}};
    for (int op = 0; op < 5; ++op)
        for (int blocks : {256, 512}) {
            for (int mode = 0; mode < 4; ++mode) {
                float ms = 0;
                for (int rep = 0; rep < 2; ++rep) {
                    hipEventRecord(e0);
                    hipLaunchKernelGGL(kernels[op][mode], dim3(blocks), dim3(256), 0, 0, out, iters);
                    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
                }
                printf("%-18s %d wave/SIMD  %-24s %8.1f ns per iteration  (%.0f cycles at 2.4 GHz)\n", ops[op], blocks / 256, names[mode], ms * 1e6 / iters, ms * 1e6 / iters * 2.4);
            }
        }
    return 0;
}
