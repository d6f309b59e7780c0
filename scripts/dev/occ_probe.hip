// Dev-only: what residency does the runtime report / the hardware give for big-LDS workgroups on gfx950?
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(1024) k1024(float* o, long long* slot) {
    extern __shared__ float lds[];
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    if (threadIdx.x == 0) {
        long long t0 = __builtin_amdgcn_s_memrealtime();
        while (__builtin_amdgcn_s_memrealtime() - t0 < 20000) {}                       // 200 us at 100 MHz
        slot[blockIdx.x * 2 + 0] = t0; slot[blockIdx.x * 2 + 1] = __builtin_amdgcn_s_memrealtime();
    }
    __syncthreads();
    o[blockIdx.x] = lds[(threadIdx.x + 1) & 1023];
}
int main() {
    hipFuncSetAttribute(reinterpret_cast<const void*>(k1024), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int threads : {1024, 512}) for (int kb : {32, 64, 70, 76, 80}) {
        int nb = 0;
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k1024, threads, (size_t)kb * 1024);
        printf("API: threads %4d lds %3d KB -> %d blocks/CU\n", threads, kb, nb);
    }
    for (int threads : {1024, 512}) for (int kb : {70, 40, 20}) {
        float* o; long long* slot; const int blocks = 1024;
        hipMalloc(&o, blocks * 4); hipMalloc(&slot, blocks * 16);
        hipLaunchKernelGGL(k1024, dim3(blocks), dim3(threads), (size_t)kb * 1024, 0, o, slot);
        hipDeviceSynchronize();
        static long long h[1024 * 2]; hipMemcpy(h, slot, sizeof(h), hipMemcpyDeviceToHost);
        long long tmin = h[0]; for (int b = 0; b < blocks; ++b) if (h[b * 2] < tmin) tmin = h[b * 2];
        int early = 0; for (int b = 0; b < blocks; ++b) if (h[b * 2] - tmin < 10000) ++early;   // started within the first 100 us
        printf("census threads %4d lds %d KB: %d of %d blocks started within the first 100 us (256 CUs) -> %.2f per CU\n", threads, kb, early, blocks, early / 256.0);
        hipFree(o); hipFree(slot);
    }
    return 0;
}
