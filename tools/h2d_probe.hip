// One-off diagnosis: what an asynchronous frame upload (3 images, 2.8 MB) costs on this box, by form.
// build: hipcc -O2 --offload-arch=gfx950 -o tools/h2d_probe tools/h2d_probe.hip ; run on the GPU box
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void busy(float *p, int n) { float a = p[0]; for (int i = 0; i < n; ++i) a = a * 1.0001f + 0.5f; p[threadIdx.x] = a; }
int main()
{
    const size_t P = 1242 * 375, sz[3] = {P * 3, P * 2, P};
    for (int flags_i = 0; flags_i < 2; ++flags_i) {
        const unsigned flags = flags_i == 0 ? hipHostMallocDefault : hipHostMallocNonCoherent;
        unsigned char *h[3][3], *d[3][3];
        for (int s = 0; s < 3; ++s) for (int k = 0; k < 3; ++k) { CK(hipHostMalloc((void **)&h[s][k], sz[k], flags)); CK(hipMalloc((void **)&d[s][k], sz[k])); }
        hipStream_t main_s, c1, c2;
        CK(hipStreamCreateWithFlags(&main_s, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&c1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&c2, hipStreamNonBlocking));
        hipEvent_t e1[3], e2[3], ef[3];
        for (int s = 0; s < 3; ++s) { CK(hipEventCreateWithFlags(&e1[s], hipEventDisableTiming)); CK(hipEventCreateWithFlags(&e2[s], hipEventDisableTiming)); CK(hipEventCreateWithFlags(&ef[s], hipEventDisableTiming)); }
        float *scratch; CK(hipMalloc((void **)&scratch, 4096));
        // calibrate the stand-in for a frame's kernels to ~50 us
        int iters = 20000;
        {
            hipLaunchKernelGGL(busy, dim3(1), dim3(64), 0, main_s, scratch, iters); CK(hipDeviceSynchronize());
            const double b0 = now();
            hipLaunchKernelGGL(busy, dim3(1), dim3(64), 0, main_s, scratch, iters); CK(hipDeviceSynchronize());
            const double bt = now() - b0;
            iters = (int)(iters * 50e-6 / bt) + 1;
            const double b1 = now();
            hipLaunchKernelGGL(busy, dim3(1), dim3(64), 0, main_s, scratch, iters); CK(hipDeviceSynchronize());
            if (flags_i == 0) printf("stand-in kernel: %d iterations = %.1f us (incl. launch + sync)\n", iters, (now() - b1) * 1e6);
        }
        for (int form = 0; form < 4; ++form) {     // 0: two copy streams + events (the core's form); 1: one copy stream; 2: all on the main stream; 3: as 0, one buffer set of 2.8 MB in ONE copy
            for (int s = 0; s < 3; ++s) CK(hipEventRecord(ef[s], main_s));
            CK(hipDeviceSynchronize());
            const int N = 60;
            std::vector<double> call(N);
            const double t0 = now();
            for (int f = 0; f < N; ++f) {
                const int s = f % 3;
                const double c0 = now();
                CK(hipEventSynchronize(ef[s]));
                hipStream_t sa = form == 2 ? main_s : c1, sb = form == 0 ? c2 : sa;
                if (form == 3) { sa = c1; sb = c1; CK(hipMemcpyAsync(d[s][0], h[s][0], sz[0], hipMemcpyHostToDevice, sa)); }
                else {
                    CK(hipMemcpyAsync(d[s][0], h[s][0], sz[0], hipMemcpyHostToDevice, sa));
                    CK(hipMemcpyAsync(d[s][1], h[s][1], sz[1], hipMemcpyHostToDevice, sb));
                    CK(hipMemcpyAsync(d[s][2], h[s][2], sz[2], hipMemcpyHostToDevice, sb));
                }
                if (form != 2) {
                    CK(hipEventRecord(e1[s], sa)); CK(hipStreamWaitEvent(main_s, e1[s], 0));
                    if (form == 0) { CK(hipEventRecord(e2[s], sb)); CK(hipStreamWaitEvent(main_s, e2[s], 0)); }
                }
                hipLaunchKernelGGL(busy, dim3(1), dim3(64), 0, main_s, scratch, iters);      // ~50 us of "frame"
                CK(hipEventRecord(ef[s], main_s));
                call[f] = now() - c0;
            }
            CK(hipDeviceSynchronize());
            const double el = now() - t0;
            double mx = 0, sum = 0; for (double c : call) { mx = c > mx ? c : mx; sum += c; }
            printf("flags %s form %d: %.1f us/frame, host per call avg %.1f max %.1f us\n", flags_i ? "noncoherent" : "default", form, el / N * 1e6, sum / N * 1e6, mx * 1e6);
        }
        for (int s = 0; s < 3; ++s) for (int k = 0; k < 3; ++k) { (void)hipHostFree(h[s][k]); (void)hipFree(d[s][k]); }
    }
    return 0;
}
