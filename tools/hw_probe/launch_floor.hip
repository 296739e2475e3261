// Dependent-launch floor on one HIP stream: N short kernels back to back, plain launches against the same chain replayed from a hipGraph.
// build: hipcc --offload-arch=gfx950 -O2 -o launch_floor launch_floor.hip ;  run: ./launch_floor [work]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
__global__ void tiny(float* p, int work) {
    float v = p[threadIdx.x];
    for (int i = 0; i < work; ++i) v = v * 1.0001f + 0.5f;
    p[threadIdx.x] = v;
}
#define HC(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main(int argc, char** argv) {
    const int work = argc > 1 ? atoi(argv[1]) : 0, N = 2000, blocks = argc > 2 ? atoi(argv[2]) : 256;
    float* p; HC(hipMalloc(&p, 4096)); HC(hipMemset(p, 0, 4096));
    hipStream_t st; HC(hipStreamCreate(&st));
    auto run_plain = [&]() { for (int i = 0; i < N; ++i) hipLaunchKernelGGL(tiny, dim3(blocks), dim3(256), 0, st, p, work); };
    run_plain(); HC(hipStreamSynchronize(st));
    for (int rep = 0; rep < 3; ++rep) {
        auto t0 = std::chrono::steady_clock::now();
        run_plain();
        auto t1 = std::chrono::steady_clock::now();
        HC(hipStreamSynchronize(st));
        auto t2 = std::chrono::steady_clock::now();
        printf("plain: issue %.2f us/launch, complete %.2f us/launch\n", std::chrono::duration<double, std::micro>(t1 - t0).count() / N,
               std::chrono::duration<double, std::micro>(t2 - t0).count() / N);
    }
    hipGraph_t g; hipGraphExec_t ge;
    HC(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    run_plain();
    HC(hipStreamEndCapture(st, &g));
    HC(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    HC(hipGraphLaunch(ge, st)); HC(hipStreamSynchronize(st));
    for (int rep = 0; rep < 3; ++rep) {
        auto t0 = std::chrono::steady_clock::now();
        HC(hipGraphLaunch(ge, st));
        auto t1 = std::chrono::steady_clock::now();
        HC(hipStreamSynchronize(st));
        auto t2 = std::chrono::steady_clock::now();
        printf("graph: issue %.2f us/launch, complete %.2f us/launch\n", std::chrono::duration<double, std::micro>(t1 - t0).count() / N,
               std::chrono::duration<double, std::micro>(t2 - t0).count() / N);
    }
    // one kernel's own duration (events around a single launch after a sync)
    hipEvent_t e0, e1; HC(hipEventCreate(&e0)); HC(hipEventCreate(&e1));
    HC(hipEventRecord(e0, st)); for (int i = 0; i < 100; ++i) hipLaunchKernelGGL(tiny, dim3(blocks), dim3(256), 0, st, p, work); HC(hipEventRecord(e1, st));
    HC(hipStreamSynchronize(st));
    float ms; HC(hipEventElapsedTime(&ms, e0, e1));
    printf("events around 100 launches: %.2f us/launch\n", ms * 10.f);
    return 0;
}
