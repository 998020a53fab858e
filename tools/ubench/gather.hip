// Micro-benchmark: what rocprofv3's FETCH_SIZE reports on gfx950 for gathers, against read volumes known by construction.
// Every kernel reads a 1 GiB buffer (four times the die-level cache) in which each 128-byte line is touched exactly once per launch:
//   stream16   16 bytes per lane, consecutive lanes consecutive addresses: every byte of the buffer, the reference point
//   gather8x1  one 8-byte load per line, lines in a scrambled order (a bijection): 8 of every 128 bytes are used
//   gather8x2  two 8-byte loads per line (adjacent lanes, at +0 and +64)
//   gather8x4  four (at +0, +32, +64, +96)
// If FETCH_SIZE per line is the same for x1, x2 and x4 the memory side moves whole 128-byte lines whatever part of them is used; if it
// grows with the touches it moves sectors.  Compared with stream16 (where the true figure is 128 bytes per line) that gives the factor
// to apply to FETCH_SIZE for gather-heavy kernels.  Run under `rocprofv3 --pmc FETCH_SIZE` (tools/calibrate_fetch_size.sh).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr uint32_t kLineBits = 23;                 // 2^23 lines of 128 bytes = 1 GiB
constexpr uint32_t kLines = 1u << kLineBits;

__global__ __launch_bounds__(256) void stream16(const uint4* __restrict__ buf, uint32_t n16, uint32_t* __restrict__ sink)
{
    uint32_t acc = 0;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n16; i += gridDim.x * 256u) { const uint4 v = buf[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x9e3779b9u) sink[0] = acc;         // (never: keeps the loads alive)
}

// PER_LINE lanes share a line; the line of a group of lanes comes from an odd multiplier modulo 2^kLineBits (a bijection)
template <uint32_t PER_LINE>
__global__ __launch_bounds__(256) void gather8(const unsigned char* __restrict__ buf, uint32_t* __restrict__ sink)
{
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;            // t < kLines * PER_LINE, checked by the host
    const uint32_t line = ((t / PER_LINE) * 2654435761u) & (kLines - 1u);
    const uint32_t within = (t % PER_LINE) * (128u / PER_LINE);
    const uint2 v = *reinterpret_cast<const uint2*>(buf + (size_t)line * 128u + within);
    if ((v.x ^ v.y) == 0x9e3779b9u) sink[1] = v.x;
}

int main()
{
    unsigned char* buf; uint32_t* sink;
    const size_t bytes = (size_t)kLines * 128u;
    CHECK(hipMalloc(&buf, bytes));
    CHECK(hipMalloc(&sink, 64));
    CHECK(hipMemset(buf, 1, bytes));
    CHECK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) {
        float ms[4];
        CHECK(hipEventRecord(e0));
        stream16<<<8192, 256>>>(reinterpret_cast<const uint4*>(buf), (uint32_t)(bytes / 16), sink);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms[0], e0, e1));
        CHECK(hipEventRecord(e0));
        gather8<1><<<kLines / 256u, 256>>>(buf, sink);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms[1], e0, e1));
        CHECK(hipEventRecord(e0));
        gather8<2><<<kLines / 128u, 256>>>(buf, sink);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms[2], e0, e1));
        CHECK(hipEventRecord(e0));
        gather8<4><<<kLines / 64u, 256>>>(buf, sink);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms[3], e0, e1));
        printf("rep %d: stream16 %.3f ms (%.0f GB/s)  gather8x1 %.3f ms  gather8x2 %.3f ms  gather8x4 %.3f ms  (%u lines of 128 bytes each)\n", rep, ms[0],
               bytes / ms[0] / 1e6, ms[1], ms[2], ms[3], kLines);
    }
    CHECK(hipGetLastError());
    CHECK(hipFree(buf)); CHECK(hipFree(sink));
    return 0;
}
