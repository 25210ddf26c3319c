// Does a wave64 VALU instruction with only lanes 0-31 (or every other lane, or lanes 0-15) enabled issue faster than one with all
// 64? k_path runs at lane utilisation 0.5: if a half-empty EXEC mask halved the issue cost, compacting live lanes would pay.
// build: hipcc --offload-arch=gfx950 -O3 -o halfexec halfexec.hip ; run: ./halfexec
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(256) k(float* out, int iters, unsigned long long mask) {
    const unsigned lane = threadIdx.x & 63u;
    float a = threadIdx.x * 1e-3f, b = 1.0001f, c = 0.5f, d = 0.25f, e = 0.125f, f = 2.f, g = 3.f, h = 4.f;
    if ((mask >> lane) & 1ull) {
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int j = 0; j < 16; j++) {
                a = __builtin_fmaf(a, b, c); d = __builtin_fmaf(d, b, e); f = __builtin_fmaf(f, b, g); h = __builtin_fmaf(h, b, a);
                c = __builtin_fmaf(c, b, d); e = __builtin_fmaf(e, b, f); g = __builtin_fmaf(g, b, h); b = __builtin_fmaf(b, 0.99999f, 1e-6f);
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + c + d + e + f + g + h + b;
}
int main() {
    float* out;
    const int grid = 256 * 8;  // 8 workgroups of 4 waves per CU: 8 waves per SIMD
    hipMalloc(&out, grid * 256 * sizeof(float));
    const unsigned long long masks[] = {~0ull, 0xffffffffull, 0xffffffff00000000ull, 0x5555555555555555ull, 0xffffull, 0x00000000ffff0000ull, 0x1ull};
    const char* names[] = {"all 64", "lanes 0-31", "lanes 32-63", "even lanes", "lanes 0-15", "lanes 16-31", "lane 0"};
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int m = 0; m < 7; m++) {
        hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, out, 200, masks[m]);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, out, 2000, masks[m]);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double inst = (double)grid * 4 * 2000.0 * 16 * 8;  // wave-level v_fma
        printf("%-12s %8.3f ms  %.1f G wave-instr/s\n", names[m], ms, inst / ms * 1e-6);
    }
    return 0;
}
