// Microbenchmark: 32x32->64 multiply as v_mul_lo_u32 + v_mul_hi_u32 versus one v_mad_u64_u32 (Philox round cost).
// build: hipcc --offload-arch=gfx950 -O3 -o scripts/ubench/mulwide scripts/ubench/mulwide.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ void wide_pair(uint32_t a, uint32_t b, uint32_t& hi, uint32_t& lo) { hi = __umulhi(a, b); lo = a * b; }
__device__ __forceinline__ void wide_mad(uint32_t a, uint32_t b, uint32_t& hi, uint32_t& lo) {
    uint64_t r;
    asm("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(r) : "s"(a), "v"(b) : "vcc");
    lo = (uint32_t)r; hi = (uint32_t)(r >> 32);
}
template <int MODE>
__global__ void __launch_bounds__(256) k(uint32_t* out, int iters, uint32_t seed) {
    uint32_t c0 = threadIdx.x + seed, c1 = blockIdx.x, c2 = 7u * threadIdx.x, c3 = 1u, k0 = seed, k1 = 0;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 10; i++) {
            uint32_t hi0, lo0, hi1, lo1;
            if (MODE == 0) { wide_pair(0xD2511F53u, c0, hi0, lo0); wide_pair(0xCD9E8D57u, c2, hi1, lo1); }
            else { wide_mad(0xD2511F53u, c0, hi0, lo0); wide_mad(0xCD9E8D57u, c2, hi1, lo1); }
            uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
            c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
            k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0 ^ c1 ^ c2 ^ c3;
}
int main() {
    const int blocks = 256 * 16, iters = 2000;
    uint32_t* d; (void)hipMalloc(&d, blocks * 256 * 4);
    uint32_t h[2][4];
    for (int mode = 0; mode < 2; mode++) {
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(a);
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, d, iters, 12345u);
            else hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, d, iters, 12345u);
            hipEventRecord(b); hipEventSynchronize(b);
        }
        float ms; hipEventElapsedTime(&ms, a, b);
        hipMemcpy(h[mode], d, 16, hipMemcpyDeviceToHost);
        double blocks_per_s = (double)blocks * 256 * iters / (ms * 1e-3);
        printf("mode %d (%s): %.3f ms, %.2f G Philox blocks/s, out %08x %08x\n", mode, mode ? "v_mad_u64_u32" : "mul_lo+mul_hi", ms, blocks_per_s * 1e-9, h[mode][0], h[mode][1]);
    }
    printf("%s\n", (h[0][0] == h[1][0] && h[0][1] == h[1][1]) ? "results identical" : "RESULTS DIFFER");
    return 0;
}
