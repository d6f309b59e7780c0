// Dev-only microbenchmark (not part of the product): what does the CU sustain for the delay-and-sum inner loop
// shapes?  Each wave repeats: UNROLL x { 1 address add, one LDS read group, 4 dependent-per-accumulator adds }.
//   mode 0: 4 x ds_read_b32 (lane-strided, via 2 x ds_read2st64_b32)      mode 1: 1 x ds_read_b128 (quad)
//   mode 2: quad + DPP adds (r = 3 body, no branch)                         mode 3: quad + asm scalar branch tree (r varies)
// Build: hipcc --offload-arch=gfx950 -O3 scripts/dev/lds_probe.hip -o /tmp/lds_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
#define DPP " wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"

template <int MODE, int UNROLL>
__global__ void __launch_bounds__(1024) probe(const int* __restrict__ tab, float* out, int iters, int rs)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    for (int i = threadIdx.x; i < 16 * 1024; i += blockDim.x) lds[i] = (float)(i & 255) * 0.001f;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    float a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    int lane_b = lane + 64;                       // second 128-sample segment; opaque so the two ds_read_b64 are not fused into ds_read2_b64
    asm volatile("" : "+v"(lane_b));
    const int* row = tab + (blockIdx.x * 16 + wave) * 64;
    for (int it = 0; it < iters; ++it) {
#pragma unroll 1
        for (int m0 = 0; m0 < 64; m0 += UNROLL) {
            int p[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) p[u] = row[m0 + u];
            if constexpr (MODE == 0) {
                float v[UNROLL][4];
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) {
                    const float* r = lds + (m0 + u) * rs + 48 - p[u] + lane;
                    v[u][0] = r[0]; v[u][1] = r[64]; v[u][2] = r[128]; v[u][3] = r[192];
                }
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) { a0 += v[u][0]; a1 += v[u][1]; a2 += v[u][2]; a3 += v[u][3]; }
            } else if constexpr (MODE == 7 || MODE == 8 || MODE == 9) {
                // product-like: table entries via v_readlane from a VGPR (7, 8) or s_load (9); packed adds (7) or plain adds (8, 9)
                typedef float f32x2 __attribute__((ext_vector_type(2)));
                static f32x2 dummy;
                int vtab = row[(m0 + lane) & 63];
                float4 q[UNROLL];
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) {
                    const int pp = (MODE == 9) ? p[u] : __builtin_amdgcn_readlane(vtab, u);
                    q[u] = reinterpret_cast<const float4*>(lds)[(m0 + u) * (rs >> 2) + 12 - (pp >> 2) + lane];
                }
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) {
                    if constexpr (MODE == 7) {
                        f32x2 x{a0, a1}, y{a2, a3};
                        x += f32x2{q[u].x, q[u].y}; y += f32x2{q[u].z, q[u].w};
                        a0 = x.x; a1 = x.y; a2 = y.x; a3 = y.y;
                    } else { a0 += q[u].x; a1 += q[u].y; a2 += q[u].z; a3 += q[u].w; }
                }
            } else if constexpr (MODE == 6) {
                // quad + DPP + VGPR index mode: window W_u = v[64+8u : 71+8u] (W[4:7] = ds_read_b128 result, W[1:3] = previous lane's
                // y,z,w via DPP); acc[t] += W[idx + t], idx = 4 - r in M0 (s_set_gpr_idx_on, SRC0 relative).  No branches.
                static_assert(UNROLL == 4, "mode 6 is written for 4 mics per block");
                int off[4], idx[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { off[u] = ((m0 + u) * rs + 48 - (p[u] & ~3)) * 4; idx[u] = 4 - (p[u] & 3); }
                const int lane16 = lane * 16;
                asm volatile(
                    "v_add_u32 v96, %[o0], %[l16]\n\tv_add_u32 v97, %[o1], %[l16]\n\tv_add_u32 v98, %[o2], %[l16]\n\tv_add_u32 v99, %[o3], %[l16]\n\t"
                    "ds_read_b128 v[68:71], v96\n\tds_read_b128 v[76:79], v97\n\tds_read_b128 v[84:87], v98\n\tds_read_b128 v[92:95], v99\n\t"
                    "s_waitcnt lgkmcnt(3)\n\t"
                    "v_mov_b32_dpp v65, v69" DPP "v_mov_b32_dpp v66, v70" DPP "v_mov_b32_dpp v67, v71" DPP
                    "s_set_gpr_idx_on %[i0], 1\n\t"
                    "v_add_f32 %[a0], v64, %[a0]\n\tv_add_f32 %[a1], v65, %[a1]\n\tv_add_f32 %[a2], v66, %[a2]\n\tv_add_f32 %[a3], v67, %[a3]\n\t"
                    "s_set_gpr_idx_off\n\t"
                    "s_waitcnt lgkmcnt(2)\n\t"
                    "v_mov_b32_dpp v73, v77" DPP "v_mov_b32_dpp v74, v78" DPP "v_mov_b32_dpp v75, v79" DPP
                    "s_set_gpr_idx_on %[i1], 1\n\t"
                    "v_add_f32 %[a0], v72, %[a0]\n\tv_add_f32 %[a1], v73, %[a1]\n\tv_add_f32 %[a2], v74, %[a2]\n\tv_add_f32 %[a3], v75, %[a3]\n\t"
                    "s_set_gpr_idx_off\n\t"
                    "s_waitcnt lgkmcnt(1)\n\t"
                    "v_mov_b32_dpp v81, v85" DPP "v_mov_b32_dpp v82, v86" DPP "v_mov_b32_dpp v83, v87" DPP
                    "s_set_gpr_idx_on %[i2], 1\n\t"
                    "v_add_f32 %[a0], v80, %[a0]\n\tv_add_f32 %[a1], v81, %[a1]\n\tv_add_f32 %[a2], v82, %[a2]\n\tv_add_f32 %[a3], v83, %[a3]\n\t"
                    "s_set_gpr_idx_off\n\t"
                    "s_waitcnt lgkmcnt(0)\n\t"
                    "v_mov_b32_dpp v89, v93" DPP "v_mov_b32_dpp v90, v94" DPP "v_mov_b32_dpp v91, v95" DPP
                    "s_set_gpr_idx_on %[i3], 1\n\t"
                    "v_add_f32 %[a0], v88, %[a0]\n\tv_add_f32 %[a1], v89, %[a1]\n\tv_add_f32 %[a2], v90, %[a2]\n\tv_add_f32 %[a3], v91, %[a3]\n\t"
                    "s_set_gpr_idx_off"
                    : [a0] "+v"(a0), [a1] "+v"(a1), [a2] "+v"(a2), [a3] "+v"(a3)
                    : [o0] "s"(off[0]), [o1] "s"(off[1]), [o2] "s"(off[2]), [o3] "s"(off[3]), [i0] "s"(idx[0]), [i1] "s"(idx[1]), [i2] "s"(idx[2]),
                      [i3] "s"(idx[3]), [l16] "v"(lane16)
                    : "m0", "memory", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79",
                      "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97",
                      "v98", "v99");
            } else if constexpr (MODE == 4) {
                // pairs: lane owns samples 2l,2l+1 (+128): table entry = byte offset (8-aligned), 2 x ds_read_b64
                float2 q0[UNROLL], q1[UNROLL];
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) {
                    const float2* b = reinterpret_cast<const float2*>(lds) + ((m0 + u) * (rs >> 1) + 24 - (p[u] >> 1)) + lane;
                    q0[u] = b[0];
                    q1[u] = b[lane_b - lane];
                }
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) { a0 += q0[u].x; a1 += q0[u].y; a2 += q1[u].x; a3 += q1[u].y; }
            } else if constexpr (MODE == 5) {
                // lerp on pairs: a-pair and b-pair from two addresses, 4 x ds_read_b64, sub/fma/add per sample, h-mask on segment 0
                float2 A0[UNROLL], A1[UNROLL], B0[UNROLL], B1[UNROLL];
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) {
                    const float2* ba = reinterpret_cast<const float2*>(lds) + ((m0 + u) * (rs >> 1) + 24 - (p[u] >> 1)) + lane;
                    const float2* bb = reinterpret_cast<const float2*>(lds) + ((m0 + u) * (rs >> 1) + 25 - (p[u] >> 1)) + lane;
                    A0[u] = ba[0]; A1[u] = ba[lane_b - lane];
                    B0[u] = bb[0]; B1[u] = bb[lane_b - lane];
                }
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) {
                    const float h = __int_as_float(0x3e800000 + (p[u] << 8));
                    const float h0 = (2 * lane != p[u]) ? h : 0.0f, h1 = (2 * lane + 1 != p[u]) ? h : 0.0f;
                    a0 += __fmaf_rn(h0, B0[u].x - A0[u].x, A0[u].x); a1 += __fmaf_rn(h1, B0[u].y - A0[u].y, A0[u].y);
                    a2 += __fmaf_rn(h, B1[u].x - A1[u].x, A1[u].x); a3 += __fmaf_rn(h, B1[u].y - A1[u].y, A1[u].y);
                }
            } else {
                float4 q[UNROLL];
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) q[u] = reinterpret_cast<const float4*>(lds)[(m0 + u) * (rs >> 2) + 12 - (p[u] >> 2) + lane];
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) {
                    if constexpr (MODE == 1) { a0 += q[u].x; a1 += q[u].y; a2 += q[u].z; a3 += q[u].w; }
                    if constexpr (MODE == 2) {
                        asm volatile("v_add_f32_dpp %0, %5, %0" DPP "v_add_f32_dpp %1, %6, %1" DPP "v_add_f32_dpp %2, %7, %2" DPP "v_add_f32 %3, %3, %4"
                                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(q[u].x), "v"(q[u].y), "v"(q[u].z), "v"(q[u].w));
                    }
                    if constexpr (MODE == 3) {
                        int t;
                        asm volatile(
                            "s_and_b32 %[t], %[p], 3\n\ts_cmp_lt_u32 %[t], 2\n\ts_cbranch_scc1 .Llo_%=\n\ts_cmp_eq_u32 %[t], 2\n\ts_cbranch_scc1 .Lr2_%=\n\t"
                            "v_add_f32_dpp %[a0], %[qy], %[a0]" DPP "v_add_f32_dpp %[a1], %[qz], %[a1]" DPP "v_add_f32_dpp %[a2], %[qw], %[a2]" DPP "v_add_f32 %[a3], %[a3], %[qx]\n\ts_branch .Lend_%=\n"
                            ".Lr2_%=:\n\tv_add_f32_dpp %[a0], %[qz], %[a0]" DPP "v_add_f32_dpp %[a1], %[qw], %[a1]" DPP "v_add_f32 %[a2], %[a2], %[qx]\n\tv_add_f32 %[a3], %[a3], %[qy]\n\ts_branch .Lend_%=\n"
                            ".Llo_%=:\n\ts_cmp_eq_u32 %[t], 0\n\ts_cbranch_scc1 .Lr0_%=\n\t"
                            "v_add_f32_dpp %[a0], %[qw], %[a0]" DPP "v_add_f32 %[a1], %[a1], %[qx]\n\tv_add_f32 %[a2], %[a2], %[qy]\n\tv_add_f32 %[a3], %[a3], %[qz]\n\ts_branch .Lend_%=\n"
                            ".Lr0_%=:\n\tv_add_f32 %[a0], %[a0], %[qx]\n\tv_add_f32 %[a1], %[a1], %[qy]\n\tv_add_f32 %[a2], %[a2], %[qz]\n\tv_add_f32 %[a3], %[a3], %[qw]\n.Lend_%=:"
                            : [a0] "+v"(a0), [a1] "+v"(a1), [a2] "+v"(a2), [a3] "+v"(a3), [t] "=&s"(t)
                            : [p] "s"(p[u]), [qx] "v"(q[u].x), [qy] "v"(q[u].y), [qz] "v"(q[u].z), [qw] "v"(q[u].w) : "scc");
                    }
                }
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3;
}

template <int MODE, int UNROLL>
int run(const char* name, const int* d_tab, float* d_out, int waves, int rmode, int lds_kb = 80)
{
    const int iters = 400, blocks = 256 * 2, rs = 256 + 48;
    const size_t lds = 64 * 1024 + 4096;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(probe<MODE, UNROLL>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024)));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const size_t shmem = (size_t)lds_kb * 1024;   // 16 waves: one WG per CU... (two 80 KB WGs fit a CU, so use 2 blocks/CU when waves == 8)
    (void)lds;
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((probe<MODE, UNROLL>), dim3(blocks), dim3(waves * 64), shmem, 0, d_tab, d_out, iters, rs);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    }
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double dm = (double)blocks * waves * iters * 64;           // (direction, mic) steps
    const double cyc_per_dm_cu = ms * 1e-3 * 2.4e9 * 256 / dm;
    std::vector<float> ho(1024); (void)hipMemcpy(ho.data(), d_out, 1024 * sizeof(float), hipMemcpyDeviceToHost);
    double chk = 0; for (float v : ho) chk += v;
    printf("[chk %.6e] ", chk);
    printf("%-28s lds %3d KB waves/WG %2d tablemode %d: %.3f ms  %.2f cycles per (d,m) per CU  -> %.1f MAC/clk/CU\n", name, lds_kb, waves, rmode, ms, cyc_per_dm_cu, 256.0 / cyc_per_dm_cu);
    return 0;
}

int main()
{
    const int entries = 512 * 16 * 64;
    std::vector<int> h(entries);
    int* d_tab; float* d_out;
    CK(hipMalloc(&d_tab, entries * sizeof(int))); CK(hipMalloc(&d_out, 512 * 1024 * sizeof(float)));
    for (int rmode = 0; rmode < 2; ++rmode) {
        unsigned s = 12345;
        for (int i = 0; i < entries; ++i) { s = s * 1664525u + 1013904223u; int p = (s >> 16) % 44; h[i] = rmode == 0 ? (p & ~3) : p; }
        CK(hipMemcpy(d_tab, h.data(), entries * sizeof(int), hipMemcpyHostToDevice));
        for (int waves : {16}) {
            if (run<0, 4>("b32x4 unroll4", d_tab, d_out, waves, rmode, 136)) return 1;
            if (run<0, 8>("b32x4 unroll8", d_tab, d_out, waves, rmode, 136)) return 1;
            if (run<1, 8>("b128 unroll8", d_tab, d_out, waves, rmode, 136)) return 1;
            if (run<3, 8>("b128+branch unroll8", d_tab, d_out, waves, rmode, 136)) return 1;
            if (run<0, 4>("b32x4 unroll4", d_tab, d_out, waves, rmode)) return 1;
            if (run<0, 8>("b32x4 unroll8", d_tab, d_out, waves, rmode)) return 1;
            if (run<1, 4>("b128 unroll4", d_tab, d_out, waves, rmode)) return 1;
            if (run<1, 8>("b128 unroll8", d_tab, d_out, waves, rmode)) return 1;
            if (run<1, 16>("b128 unroll16", d_tab, d_out, waves, rmode)) return 1;
            if (run<2, 8>("b128+dpp unroll8", d_tab, d_out, waves, rmode)) return 1;
            if (run<3, 8>("b128+branch unroll8", d_tab, d_out, waves, rmode)) return 1;
            if (run<7, 4>("b128 readlane pk_add u4", d_tab, d_out, waves, rmode, 136)) return 1;
            if (run<8, 4>("b128 readlane add u4", d_tab, d_out, waves, rmode, 136)) return 1;
            if (run<9, 4>("b128 sload add u4", d_tab, d_out, waves, rmode, 136)) return 1;
            if (run<7, 8>("b128 readlane pk_add u8", d_tab, d_out, waves, rmode, 136)) return 1;
            if (run<8, 8>("b128 readlane add u8", d_tab, d_out, waves, rmode, 136)) return 1;
            if (run<4, 4>("pairs b64x2 unroll4", d_tab, d_out, waves, rmode)) return 1;
            if (run<4, 8>("pairs b64x2 unroll8", d_tab, d_out, waves, rmode)) return 1;
            if (run<5, 4>("lerp pairs b64x4 unroll4", d_tab, d_out, waves, rmode)) return 1;
            if (run<5, 8>("lerp pairs b64x4 unroll8", d_tab, d_out, waves, rmode)) return 1;
        }
    }
    return 0;
}
