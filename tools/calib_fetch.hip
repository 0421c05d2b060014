// calib_fetch.hip -- known-size streaming kernels for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950
// (MI355X_MICROARCH.md "HBM": FETCH_SIZE reports half of the bytes of a 16 B/lane streaming read; other widths are
// uncalibrated).  Every kernel touches exactly `bytes` bytes of a 1 GiB buffer (four times the 256 MiB Infinity Cache),
// once, so  (bytes the counter reports) / bytes  is the factor for that access shape:
//   k_calib_r4 / r8 / r16    coalesced streaming reads, 4 / 8 / 16 bytes per lane
//   k_calib_g8               random 8-byte gathers (the conflict test's (depth, class) lookups), 128 Mi of them
//   k_calib_w4 / w8 / w16    coalesced streaming writes
//   k_calib_a8               64-bit atomicMin to random slots (the index-map splat), 64 Mi of them
// Build: hipcc -O3 --offload-arch=gfx950 -o tools/calib_fetch tools/calib_fetch.hip      (__graft_entry__.build() does)
// Run:   rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/calib/fetch -- ./tools/calib_fetch
//        rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/calib/write -- ./tools/calib_fetch
//        python tools/prof_summary.py --calibrate gpurun_out/calib --out profiles/fetch_calibration.json
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t x)
{
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull;
    return x ^ (x >> 33);
}

#define STRIDE_LOOP(i, n) for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < (n); i += (size_t)gridDim.x * 256)

__global__ __launch_bounds__(256) void k_calib_r4(const uint32_t *__restrict__ p, size_t n, uint32_t *__restrict__ sink)
{
    uint32_t acc = 0;
    STRIDE_LOOP(i, n) acc ^= p[i];
    if (acc == 0x12345678u) sink[0] = acc;
}
__global__ __launch_bounds__(256) void k_calib_r8(const uint2 *__restrict__ p, size_t n, uint32_t *__restrict__ sink)
{
    uint32_t acc = 0;
    STRIDE_LOOP(i, n) { const uint2 v = p[i]; acc ^= v.x ^ v.y; }
    if (acc == 0x12345678u) sink[0] = acc;
}
__global__ __launch_bounds__(256) void k_calib_r16(const uint4 *__restrict__ p, size_t n, uint32_t *__restrict__ sink)
{
    uint32_t acc = 0;
    STRIDE_LOOP(i, n) { const uint4 v = p[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) sink[0] = acc;
}
__global__ __launch_bounds__(256) void k_calib_g8(const uint2 *__restrict__ p, size_t n_slots, size_t n_loads, uint32_t *__restrict__ sink)
{
    uint32_t acc = 0;
    STRIDE_LOOP(i, n_loads) { const uint2 v = p[mix(i) % n_slots]; acc ^= v.x ^ v.y; }
    if (acc == 0x12345678u) sink[0] = acc;
}
__global__ __launch_bounds__(256) void k_calib_w4(uint32_t *__restrict__ p, size_t n) { STRIDE_LOOP(i, n) p[i] = (uint32_t)i; }
__global__ __launch_bounds__(256) void k_calib_w8(uint2 *__restrict__ p, size_t n) { STRIDE_LOOP(i, n) p[i] = make_uint2((uint32_t)i, 1u); }
__global__ __launch_bounds__(256) void k_calib_w16(uint4 *__restrict__ p, size_t n) { STRIDE_LOOP(i, n) p[i] = make_uint4((uint32_t)i, 1u, 2u, 3u); }
__global__ __launch_bounds__(256) void k_calib_a8(unsigned long long *__restrict__ p, size_t n_slots, size_t n_ops)
{
    STRIDE_LOOP(i, n_ops) atomicMin(&p[mix(i) % n_slots], (unsigned long long)i);
}

int main()
{
    const size_t bytes = 1ull << 30;
    void *buf = nullptr;
    uint32_t *sink = nullptr;
    CK(hipMalloc(&buf, bytes));
    CK(hipMalloc((void **)&sink, 64));
    CK(hipMemset(buf, 0x7f, bytes));
    CK(hipDeviceSynchronize());
    const dim3 grid(2048), block(256);
    for (int rep = 0; rep < 3; ++rep) {          // three launches each: the averages in the summary show the spread
        hipLaunchKernelGGL(k_calib_r4, grid, block, 0, 0, (const uint32_t *)buf, bytes / 4, sink);
        hipLaunchKernelGGL(k_calib_r8, grid, block, 0, 0, (const uint2 *)buf, bytes / 8, sink);
        hipLaunchKernelGGL(k_calib_r16, grid, block, 0, 0, (const uint4 *)buf, bytes / 16, sink);
        hipLaunchKernelGGL(k_calib_g8, grid, block, 0, 0, (const uint2 *)buf, bytes / 8, (size_t)1 << 27, sink);
        hipLaunchKernelGGL(k_calib_w4, grid, block, 0, 0, (uint32_t *)buf, bytes / 4);
        hipLaunchKernelGGL(k_calib_w8, grid, block, 0, 0, (uint2 *)buf, bytes / 8);
        hipLaunchKernelGGL(k_calib_w16, grid, block, 0, 0, (uint4 *)buf, bytes / 16);
        hipLaunchKernelGGL(k_calib_a8, grid, block, 0, 0, (unsigned long long *)buf, bytes / 8, (size_t)1 << 26);
        CK(hipGetLastError());
        CK(hipDeviceSynchronize());
    }
    printf("calib_fetch: r4/r8/r16/w4/w8/w16 move %zu bytes each; g8 = %zu random 8-byte loads (%zu bytes of lanes, <= %zu bytes of 64-byte sectors);"
           " a8 = %zu random 8-byte atomics\n", bytes, (size_t)1 << 27, (size_t)8 << 27, (size_t)64 << 27, (size_t)1 << 26);
    CK(hipFree(buf));
    CK(hipFree(sink));
    return 0;
}
