// LDS read rates of one CU on gfx950: ds_read_b128, ds_read_b64 and the transposing ds_read_b64_tr_b16, each alone (conflict-free
// addresses, 8 waves issuing back to back) — bytes per clock per CU.  The weight-gradient kernels form their MFMA fragments with the
// transposing read; if it runs at half the plain rate, the fragment reads (not the MFMAs) bound those kernels.
// Build: hipcc --offload-arch=gfx950 -O3 lds_rate.hip -o lds_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(2))) int i32x2;
typedef __attribute__((ext_vector_type(4))) int i32x4;

template <int MODE>
__global__ __launch_bounds__(512) void lds_kernel(int iters, int* __restrict__ sink, long long* __restrict__ clocks)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    for (int i = threadIdx.x; i < 65536 / 4; i += 512) reinterpret_cast<int*>(smem)[i] = i;
    __syncthreads();
    const uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // 16 B (b128) or 8 B (b64) per lane, consecutive lanes consecutive addresses: conflict-free
    uint32_t addr = base + wave * 4096 + lane * (MODE == 0 ? 16 : 8);
    i32x4 a4 = {0, 0, 0, 0};
    i32x2 a2 = {0, 0};
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (MODE == 0) { i32x4 v; asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(u * 1024 % 4096)); a4 += v; }
            else if (MODE == 1) { i32x2 v; asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(u * 512 % 4096)); a2 += v; }
            else { i32x2 v; asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(u * 512 % 4096)); a2 += v; }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    const long long t1 = __builtin_readcyclecounter();
    if (a4.x + a4.y + a4.z + a4.w + a2.x + a2.y == 0x12345678) sink[0] = 1;
    if (threadIdx.x == 0) clocks[blockIdx.x] = t1 - t0;
}

int main()
{
    int* sink; long long* clocks;
    hipMalloc(&sink, 4); hipMalloc(&clocks, 256 * 8);
    const int iters = 2000;
    const char* names[3] = {"ds_read_b128", "ds_read_b64", "ds_read_b64_tr_b16"};
    for (int mode = 0; mode < 3; ++mode) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(lds_kernel<0>, dim3(256), dim3(512), 65536, 0, iters, sink, clocks);
            else if (mode == 1) hipLaunchKernelGGL(lds_kernel<1>, dim3(256), dim3(512), 65536, 0, iters, sink, clocks);
            else hipLaunchKernelGGL(lds_kernel<2>, dim3(256), dim3(512), 65536, 0, iters, sink, clocks);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long c[256]; hipMemcpy(c, clocks, sizeof(c), hipMemcpyDeviceToHost);
        const double bytes_per_cu = (double)iters * 16 * 8 * 64 * (mode == 0 ? 16 : 8);    // iterations x reads x waves x lanes x bytes
        // s_memtime / readcyclecounter ticks at a fixed 100 MHz on this part: use the event time and the nominal 2.4 GHz for B/clk
        printf("%-20s %8.3f ms  %7.1f GB/s per CU  = %6.1f B/clk at 2.4 GHz (%.0f TB/s chip)\n", names[mode], ms, bytes_per_cu / ms * 1e-6,
               bytes_per_cu / (ms * 1e-3) / 2.4e9, bytes_per_cu * 256 / ms * 1e-9);
    }
    return 0;
}
