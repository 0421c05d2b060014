// One-off diagnosis: the device-side gap between two dependent kernels -- plain stream launches against a captured hipGraph.
// build: hipcc -O2 --offload-arch=gfx950 -o tools/gap_probe tools/gap_probe.hip ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void work(unsigned long long *stamps, int slot, float *p, int iters)
{
    const unsigned long long t0 = wall_clock64();
    float a = p[threadIdx.x & 63];
    for (int i = 0; i < iters; ++i) a = a * 1.0001f + 0.5f;
    p[(blockIdx.x * blockDim.x + threadIdx.x) & 1023] = a;
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicMin(&stamps[slot * 2], t0);                                   // first entry of the launch
        atomicMax(&stamps[slot * 2 + 1], wall_clock64());                   // last exit
    }
}
int main()
{
    unsigned long long *d_st; float *d_p;
    const int N = 40;                      // kernel pairs per run
    CK(hipMalloc((void **)&d_st, sizeof(unsigned long long) * 4 * N)); CK(hipMalloc((void **)&d_p, 4096 * 4));
    CK(hipMemset(d_p, 0, 4096 * 4));
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    std::vector<unsigned long long> h(4 * N);
    auto reset = [&]() { for (int i = 0; i < 2 * N; ++i) { h[2 * i] = ~0ull; h[2 * i + 1] = 0ull; } return hipMemcpy(d_st, h.data(), h.size() * 8, hipMemcpyHostToDevice); };
    auto report = [&](const char *what) {
        if (hipMemcpy(h.data(), d_st, h.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) return;
        std::vector<double> gaps;
        for (int i = 0; i + 1 < 2 * N; ++i) gaps.push_back(((double)h[2 * (i + 1)] - (double)h[2 * i + 1]) / 100.0);     // next entry - this exit, us (100 MHz)
        std::sort(gaps.begin(), gaps.end());
        printf("%s: gap between dependent kernels, us: min %.2f median %.2f max %.2f\n", what, gaps.front(), gaps[gaps.size() / 2], gaps.back());
    };
    for (int grid : {64, 1536}) {
        // plain stream launches
        CK(reset());
        for (int i = 0; i < 2 * N; ++i) hipLaunchKernelGGL(work, dim3(grid), dim3(256), 0, st, d_st, i, d_p, 2000);
        CK(hipStreamSynchronize(st));
        printf("grid %d ", grid); report("stream");
        // the same chain captured into a graph
        CK(reset());
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
        for (int i = 0; i < 2 * N; ++i) hipLaunchKernelGGL(work, dim3(grid), dim3(256), 0, st, d_st, i, d_p, 2000);
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, st));
        CK(hipStreamSynchronize(st));
        printf("grid %d ", grid); report("graph ");
        (void)hipGraphExecDestroy(ge); (void)hipGraphDestroy(g);
    }
    return 0;
}
