// Micro-benchmark: sustained wave-level vector-instruction issue rate of gfx950 with 1, 2, 4 and 8 waves per SIMD -- f32 fma, int32
// multiply-add, 64-bit mad, transcendental (exp2), 32 x 32-bit multiply -- to know what a kernel's SQ_INSTS_VALU count means in time:
// how many cycles of a SIMD one wave-instruction takes when enough waves are resident.
//     valu [iterations]        (the CPU pre-check of this file, tools/ubench/check_on_cpu.py, runs it with 2)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kMaxBlocks = 256 * 8;     // eight workgroups of 256 work-items per CU of 256: eight waves per SIMD

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters)
{
    float a[8]; int b[8]; long long c[4];
    for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 0.001f + i; b[i] = (int)threadIdx.x + i; }
    for (int i = 0; i < 4; ++i) c[i] = threadIdx.x + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            if (MODE == 0) { for (int i = 0; i < 8; ++i) a[i] = fmaf(a[i], 1.0001f, 0.5f); }
            if (MODE == 1) { for (int i = 0; i < 8; ++i) b[i] = b[i] * 3 + it; }
            if (MODE == 2) { for (int i = 0; i < 4; ++i) c[i] = c[i] * (long long)(int)b[i] + it; }
            if (MODE == 3) { for (int i = 0; i < 8; ++i) a[i] = __builtin_amdgcn_exp2f(a[i]) * 0.25f; }     // one transcendental + one multiply
            if (MODE == 4) { for (int i = 0; i < 8; ++i) b[i] = b[i] * b[(i + 1) & 7]; }                    // full 32-bit multiply
        }
    }
    float s = 0; for (int i = 0; i < 8; ++i) s += a[i] + (float)b[i]; for (int i = 0; i < 4; ++i) s += (float)c[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main(int argc, char** argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    float* d;
    CHECK(hipMalloc((void**)&d, (size_t)kMaxBlocks * 256 * sizeof(float)));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const char* name[] = {"v_fma_f32", "int32 multiply-add", "64-bit mad", "v_exp_f32 + v_mul_f32", "32 x 32-bit multiply"};
    const int n_waves[] = {1, 2, 4, 8};
    for (int mode = 0; mode < 5; ++mode)
        for (int w = 0; w < 4; ++w) {   // workgroups of 256 work-items per CU = waves per SIMD
            const int bpc = n_waves[w], blocks = 256 * bpc;     // <= kMaxBlocks
            float ms = 0;
            for (int rep = 0; rep < 2; ++rep) {
                CHECK(hipEventRecord(e0, 0));
                if (mode == 0) k<0><<<blocks, 256>>>(d, iters);
                if (mode == 1) k<1><<<blocks, 256>>>(d, iters);
                if (mode == 2) k<2><<<blocks, 256>>>(d, iters);
                if (mode == 3) k<3><<<blocks, 256>>>(d, iters);
                if (mode == 4) k<4><<<blocks, 256>>>(d, iters);
                CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
                CHECK(hipEventElapsedTime(&ms, e0, e1));
            }
            const double per_iter = mode == 2 ? 32.0 : mode == 3 ? 128.0 : 64.0;   // wave-instructions per loop iteration as written (what the compiler makes of them: see the ISA)
            const double winst = (double)blocks * 4 * iters * per_iter;
            printf("%-22s waves/SIMD %d: %8.3f ms, %8.1f G wave-instructions/s, %.2f cycles of a SIMD per instruction at 2.4 GHz\n", name[mode], bpc, ms, winst / ms / 1e6,
                   1024.0 * 2.4e9 / (winst / (ms * 1e-3)));
        }
    CHECK(hipGetLastError());
    CHECK(hipFree(d));
    return 0;
}
