// divconst_check.hip -- exhaustive check (all 2^32 float inputs x) of the two-instruction division by a known constant
//   q = fma(x, zh, RN(x * zl)),  zh = RN(1/c), zl = RN(1/c - zh)        (Brisebarre, Muller, Raina 2004)
// against the correctly rounded x / c, for the per-frame constants the surfel kernels divide by.
// Build: hipcc -O2 --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-flush-denormals-to-zero -o tools/divconst_check tools/divconst_check.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

__global__ void k_check(float c, float zh, float zl, unsigned long long *bad /* [4]: total, |x| normal & result normal, first bad x, - */)
{
    const uint64_t base = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 256ull;
    unsigned long long nb = 0, nbn = 0;
    for (uint32_t i = 0; i < 256u; ++i) {
        const uint32_t bits = (uint32_t)(base + i);
        const float x = __uint_as_float(bits);
        const float ref = x / c;
        const float q = __builtin_fmaf(x, zh, x * zl);
        const uint32_t a = __float_as_uint(ref), b = __float_as_uint(q);
        const bool same = a == b || (ref != ref && q != q);
        if (!same) {
            ++nb;
            const float ar = fabsf(ref);
            if (ar >= 1.17549435e-38f && ar <= 3.0e38f) { ++nbn; atomicMin(&bad[2], (unsigned long long)bits); atomicMax(&bad[3], (unsigned long long)(bits & 0x7FFFFFFFu)); }
        }
    }
    if (nb) atomicAdd(&bad[0], nb);
    if (nbn) atomicAdd(&bad[1], nbn);
}

int main(int argc, char **argv)
{
    unsigned long long *d_bad, h[4];
    hipMalloc((void **)&d_bad, 32);
    for (int a = 1; a < argc; ++a) {
        const float c = strtof(argv[a], nullptr);
        const float zh = 1.0f / c;
        const float zl = (float)(1.0 / (double)c - (double)zh);
        h[0] = h[1] = 0; h[2] = ~0ull; h[3] = 0;
        hipMemcpy(d_bad, h, 32, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_check, dim3(1u << 16), dim3(256), 0, 0, c, zh, zl, d_bad);
        hipDeviceSynchronize();
        hipMemcpy(h, d_bad, 32, hipMemcpyDeviceToHost);
        float fx, fm; uint32_t fb = (uint32_t)h[2], fmb = (uint32_t)h[3]; memcpy(&fx, &fb, 4); memcpy(&fm, &fmb, 4);
        printf("c = %-12.9g zh = %-14.9g zl = %-14.9g  mismatches: %llu (with a normal quotient: %llu, smallest x = %g, largest |x| = %g)\n", c, zh, zl, h[0], h[1], h[1] ? fx : 0.0f, h[1] ? fm : 0.0f);
    }
    return 0;
}
