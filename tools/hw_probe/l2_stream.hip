// How fast can every CU of an MI355X stream L2-resident data — the weight stream of the implicit-GEMM / halo convolutions — (a) into
// LDS with buffer_load ... lds (1 KiB per wave-instruction) and (b) into VGPRs with buffer_load_dwordx4?  All 256 workgroups
// (512 threads, one per CU) read either the SAME region (the convolution pattern: every tile walks the same 3.5 MB of weights)
// or disjoint regions.  Build: hipcc --offload-arch=gfx950 -O3 l2_stream.hip -o l2_stream
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <vector>
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef __attribute__((ext_vector_type(4))) int i32x4;

template <int DEPTH, bool TO_LDS>
__global__ __launch_bounds__(512) void stream_kernel(const char* __restrict__ buf, uint32_t region_bytes, uint32_t block_stride, int iters, int* __restrict__ sink)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const char* base = buf + (size_t)blockIdx.x * block_stride;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, region_bytes, 0x00020000);
    const uint32_t units = region_bytes / 16384;          // a unit = 16 KiB = 2 pieces per wave
    i32x4 acc = {0, 0, 0, 0};
    uint32_t u = 0;
    for (int it = 0; it < iters; ++it) {
        const uint32_t off = u * 16384 + wave * 2048;
        if constexpr (TO_LDS) {
            char* dst = smem + (it % DEPTH) * 16384 + wave * 2048;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)dst, 16, lane * 16, off, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(dst + 1024), 16, lane * 16, off + 1024, 0, 0);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (DEPTH - 1)) : "memory");
        } else {
            i32x4 a = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, off, 0);
            i32x4 b = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, off + 1024, 0);
            acc += a; acc += b;       // the compiler's own counted waits keep DEPTH-ish loads in flight through unrolling
        }
        if (++u == units) u = 0;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (acc.x + acc.y + acc.z + acc.w == 0x12345678) sink[0] = 1;
}

template <int DEPTH, bool TO_LDS>
static void run(const char* name, const char* buf, uint32_t region, uint32_t stride, int iters, int* sink, int nblk)
{
    const size_t lds = TO_LDS ? DEPTH * 16384 : 0;
    hipFuncSetAttribute((const void*)stream_kernel<DEPTH, TO_LDS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((stream_kernel<DEPTH, TO_LDS>), dim3(nblk), dim3(512), lds, 0, buf, region, stride, iters, sink);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep == 2) {
            const double bytes = (double)nblk * iters * 16384.0;
            printf("%-34s depth %d region %6.2f MB %s: %7.3f ms  %7.2f TB/s chip  %6.1f GB/s per CU\n", name, DEPTH, region / 1048576.0,
                   stride ? "disjoint" : "shared  ", ms, bytes / ms / 1e9, bytes / ms / 1e6 / nblk);
        }
    }
}

int main(int argc, char** argv)
{
    const int nblk = 256, iters = 4000;
    const uint32_t shared_region = 3538944;               // 27 * 8 units of 16 KiB: the 256 -> 256 weight pack
    const uint32_t own_region = 131072;                   // 128 KiB per block: 32 CUs x 128 KiB = one XCD's 4 MiB L2
    char* buf; int* sink;
    hipMalloc(&buf, (size_t)256 * 1048576);
    hipMemset(buf, 1, (size_t)256 * 1048576);
    hipMalloc(&sink, 4);
    run<2, true>("LDS-DMA", buf, shared_region, 0, iters, sink, nblk);
    run<4, true>("LDS-DMA", buf, shared_region, 0, iters, sink, nblk);
    run<8, true>("LDS-DMA", buf, shared_region, 0, iters, sink, nblk);
    run<4, true>("LDS-DMA", buf, own_region, own_region, iters, sink, nblk);
    run<8, true>("LDS-DMA", buf, own_region, 1048576, iters, sink, nblk);
    run<4, false>("buffer_load_dwordx4 -> VGPR", buf, shared_region, 0, iters, sink, nblk);
    run<4, false>("buffer_load_dwordx4 -> VGPR", buf, own_region, own_region, iters, sink, nblk);
    run<4, true>("LDS-DMA 64 blocks", buf, shared_region, 0, iters, sink, 64);
    run<4, false>("VGPR 64 blocks", buf, shared_region, 0, iters, sink, 64);
    hipDeviceSynchronize();
    return 0;
}
