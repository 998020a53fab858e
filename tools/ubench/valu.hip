// Micro-benchmark: sustained wave-level VALU issue rate of gfx950 with 1..8 waves per SIMD (f32 fma,
// int32 add, 64-bit mad), to read SQ_ACTIVE_INST_VALU correctly.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters)
{
    float a[8]; int b[8]; long long c[4];
    for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 0.001f + i; b[i] = threadIdx.x + i; }
    for (int i = 0; i < 4; ++i) c[i] = threadIdx.x + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            if (MODE == 0) { for (int i = 0; i < 8; ++i) a[i] = fmaf(a[i], 1.0001f, 0.5f); }
            if (MODE == 1) { for (int i = 0; i < 8; ++i) b[i] = b[i] * 3 + it; }
            if (MODE == 2) { for (int i = 0; i < 4; ++i) c[i] = c[i] * (long long)(int)b[i] + it; }
        }
    }
    float s = 0; for (int i = 0; i < 8; ++i) s += a[i] + b[i]; for (int i = 0; i < 4; ++i) s += (float)c[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main()
{
    float* d; hipMalloc(&d, 256 * 256 * 8 * 4 * sizeof(float));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int mode = 0; mode < 3; ++mode)
        for (int bpc : {1, 2, 4, 8}) {   // blocks of 256 threads per CU -> waves per SIMD
            const int blocks = 256 * bpc;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (mode == 0) k<0><<<blocks, 256>>>(d, iters);
                if (mode == 1) k<1><<<blocks, 256>>>(d, iters);
                if (mode == 2) k<2><<<blocks, 256>>>(d, iters);
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double per_iter = mode == 2 ? 32.0 : 64.0;   // wave-instructions of the measured kind per loop iteration
            const double winst = (double)blocks * 4 * iters * per_iter;
            printf("mode %d waves/SIMD %d: %.3f ms, %.1f G wave-instr/s, %.2f cycles per instr per SIMD at 2.4 GHz\n", mode, bpc, ms, winst / ms / 1e6,
                   1024.0 * 2.4e9 / (winst / (ms * 1e-3)));
        }
    return 0;
}
