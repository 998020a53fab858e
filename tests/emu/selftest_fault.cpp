// TEST INFRASTRUCTURE ONLY: self-test of the CPU emulation's memory fence.  A kernel that indexes by a
// 64-wide block is launched once correctly and once with 128 work-items over a 64-element buffer; the
// second launch must end in a reported fault, a sticky error, and no crash.
#include <hip/hip_runtime.h>

#include <cstdio>

extern "C" void hipemu_reset_error(void);

__global__ void copy_kernel(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst)
{
    __shared__ uint32_t stage[64];
    stage[threadIdx.x & 63u] = src[threadIdx.x];
    __syncthreads();
    const uint64_t m = __ballot(stage[threadIdx.x & 63u] & 1u);
    dst[threadIdx.x] = stage[(threadIdx.x + 1u) & 63u] + (uint32_t)__popcll(m) + (uint32_t)__shfl((int)threadIdx.x, 3);
}

// a wave-wide operation inside divergent control flow is refused, not guessed at
__global__ void divergent_kernel(uint64_t* __restrict__ out)
{
    const uint32_t lane = threadIdx.x & 63u;
    uint64_t inner = 0;
    if (lane < 20u) inner = __ballot(lane & 1u);
    out[threadIdx.x] = inner + __ballot(lane >= 60u);
}

int main()
{
    uint32_t *src = nullptr, *dst = nullptr, host[64];
    if (hipMalloc((void**)&src, sizeof host) != hipSuccess || hipMalloc((void**)&dst, sizeof host) != hipSuccess) return 2;
    for (uint32_t i = 0; i < 64; ++i) host[i] = i;
    hipMemcpy(src, host, sizeof host, hipMemcpyHostToDevice);
    ::hipemu::Launch(dim3(1), dim3(64)).go("copy_kernel", copy_kernel, src, dst);
    if (hipGetLastError() != hipSuccess) { std::printf("unexpected error in the valid launch\n"); return 3; }
    hipMemcpy(host, dst, sizeof host, hipMemcpyDeviceToHost);
    for (uint32_t i = 0; i < 64; ++i)
        if (host[i] != ((i + 1u) & 63u) + 32u + 3u) { std::printf("wrong result at %u: %u\n", i, host[i]); return 4; }
    ::hipemu::Launch(dim3(1), dim3(128)).go("copy_kernel", copy_kernel, src, dst);
    const hipError_t e = hipGetLastError();
    if (e != hipErrorIllegalAddress) { std::printf("the out-of-range launch was not caught (error %d)\n", e); return 5; }
    std::printf("%s\n", hipGetErrorString(e));
    void* more = nullptr;
    if (hipMalloc(&more, 16) == hipSuccess) { std::printf("the error is not sticky\n"); return 6; }
    hipemu_reset_error();
    uint64_t* d = nullptr;
    if (hipMalloc((void**)&d, 128 * 8) != hipSuccess) return 2;
    ::hipemu::Launch(dim3(1), dim3(128)).go("divergent_kernel", divergent_kernel, d);
    if (hipGetLastError() != hipErrorLaunchFailure) { std::printf("a divergent wave-wide operation was not refused\n"); return 7; }
    std::printf("%s\n", hipGetErrorString(hipErrorLaunchFailure));
    // a block larger than the kernel's launch bounds is refused at launch, as the GPU refuses it
    hipemu_reset_error();
    ::hipemu::Launch(dim3(1), dim3(128)).bounds(64).go("copy_kernel", copy_kernel, src, dst);
    if (hipGetLastError() != hipErrorLaunchFailure) { std::printf("a launch beyond the kernel's launch bounds was not refused\n"); return 8; }
    std::printf("%s\n", hipGetErrorString(hipErrorLaunchFailure));
    return 0;
}
