// Micro-benchmark: what 64-bit atomic minimums into a frame-sized buffer cost on gfx950 against plain 8-byte stores, for address patterns
// of different coherence.  N operations per launch (one per work-item) into a 3840 x 2160 buffer of 8-byte words:
//   rows       operation i goes to pixel (x0 + i % w, y0 + i / w) of a w-pixel-wide window: consecutive lanes, consecutive words
//   pairs      the same, two operations per word (lanes 2k and 2k + 1 share a word)
//   scrambled  a bijection of the operation number onto the window's words: every lane of a wave in a line of its own
//   strided    consecutive lanes 16 words (one 128-byte line) apart inside a row
// Kinds: st (plain store), min64 (atomicMin, result unused), min64r (result used), min32 (32-bit atomicMin on the word's upper half).
// The times tell how far the set-up kernel's resolve is from what the memory system can do, and whether sorting fragments by address
// before the atomics could pay.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr uint32_t kW = 3840, kH = 2160;
constexpr uint32_t kWinW = 1600, kWinX = 1120, kWinY = 280;   // a 1600 x 1600 window in the middle: 2.56 M words

enum Pattern { kRows, kPairs, kScrambled, kStrided };
enum Kind { kStore, kMin64, kMin64Ret, kMin32 };

__device__ inline uint32_t word_of(uint32_t i, int pattern, uint32_t n)
{
    uint32_t j = i;
    if (pattern == kPairs) j = i >> 1;
    else if (pattern == kScrambled) j = (uint32_t)(((uint64_t)i * 2654435761ull) % n);     // n is chosen odd-free below: a bijection needs gcd = 1
    else if (pattern == kStrided) { const uint32_t row = i / kWinW, c = i % kWinW; j = row * kWinW + (c % 100u) * 16u + c / 100u; }
    return (kWinY + j / kWinW) * kW + kWinX + j % kWinW;
}

template <int KIND>
__global__ __launch_bounds__(256) void ops(unsigned long long* __restrict__ buf, uint32_t n, int pattern, uint32_t salt, unsigned long long* __restrict__ sink)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const uint32_t w = word_of(i, pattern, n);
    const unsigned long long key = ((unsigned long long)(0x3f000000u - salt * 4096u - (i & 1023u)) << 32) | i;
    if (KIND == kStore) buf[w] = key;
    else if (KIND == kMin64) atomicMin(&buf[w], key);
    else if (KIND == kMin64Ret) { const unsigned long long old = atomicMin(&buf[w], key); if (old == 12345ull) sink[0] = old; }
    else atomicMin(reinterpret_cast<uint32_t*>(&buf[w]) + 1, (uint32_t)(key >> 32));
}

// What the set-up kernel's resolve does: every lane owns a 2 x 2-pixel box (neighbouring lanes neighbouring boxes) and walks it in four
// turns; in each turn about a third of the lanes have a covered sample and fire.  Against `rows` (the same number of operations, every
// lane of every instruction active) this prices the partly filled atomic instructions.
template <int KIND>
__global__ __launch_bounds__(256) void sparse_boxes(unsigned long long* __restrict__ buf, uint32_t n_boxes, uint32_t salt, uint32_t* __restrict__ fired)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n_boxes) return;
    const uint32_t bw = kWinW / 2u, bx = i % bw, by = i / bw;     // by < kWinW / 2: the host launches (kWinW / 2)^2 boxes
    uint32_t count = 0;
    for (uint32_t turn = 0; turn < 4u; ++turn) {
        const uint32_t h = (i * 4u + turn) * 2654435761u + salt * 40503u;
        if ((h >> 16) % 100u >= 30u) continue;
        const uint32_t w = (kWinY + by * 2u + (turn >> 1)) * kW + kWinX + bx * 2u + (turn & 1u);
        const unsigned long long key = ((unsigned long long)(0x3f000000u - salt * 4096u - (i & 1023u)) << 32) | i;
        if (KIND == kStore) buf[w] = key; else atomicMin(&buf[w], key);
        ++count;
    }
    if (fired && count) atomicAdd(&fired[(i >> 8) & 63u], count);
}

int main(int argc, char** argv)
{
    const int reps = argc > 1 ? atoi(argv[1]) : 6;     // (the CPU pre-check of this file runs it with 2)
    unsigned long long *buf, *sink;
    const size_t words = (size_t)kW * kH;
    CHECK(hipMalloc((void**)&buf, words * 8));
    CHECK(hipMalloc((void**)&sink, 64));
    CHECK(hipMemsetAsync(buf, 0x7f, words * 8, 0));
    CHECK(hipStreamSynchronize(0));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const uint32_t n = 1999993u;    // a prime close to 2 M: the multiplier of the scrambled pattern is a bijection modulo it
    const char* pname[] = {"rows", "pairs", "scrambled", "strided"};
    const char* kname[] = {"st", "min64", "min64r", "min32"};
    uint32_t salt = 0;
    for (int pattern = 0; pattern < 4; ++pattern) {
        for (int kind = 0; kind < 4; ++kind) {
            float best = 1e9f;
            for (int rep = 0; rep < reps; ++rep) {
                ++salt;
                CHECK(hipEventRecord(e0, 0));
                const uint32_t grid = (n + 255u) / 256u;
                if (kind == kStore) ops<kStore><<<grid, 256>>>(buf, n, pattern, salt, sink);
                else if (kind == kMin64) ops<kMin64><<<grid, 256>>>(buf, n, pattern, salt, sink);
                else if (kind == kMin64Ret) ops<kMin64Ret><<<grid, 256>>>(buf, n, pattern, salt, sink);
                else ops<kMin32><<<grid, 256>>>(buf, n, pattern, salt, sink);
                CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
                float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (rep > 0 && ms < best) best = ms;
            }
            printf("%-10s %-7s %8.2f us  (%.1f G operations/s)\n", pname[pattern], kname[kind], best * 1e3f, n / best / 1e6f);
        }
    }
    {
        const uint32_t n_boxes = (kWinW / 2u) * (kWinW / 2u);     // the 1600 x 1600-pixel window in 2 x 2 boxes: 640 000 lanes, about 770 000 operations
        uint32_t* fired; CHECK(hipMalloc((void**)&fired, 256)); CHECK(hipMemsetAsync(fired, 0, 256, 0)); CHECK(hipStreamSynchronize(0));
        sparse_boxes<kMin64><<<(n_boxes + 255u) / 256u, 256>>>(buf, n_boxes, 1u, fired);
        uint32_t h[64]; CHECK(hipMemcpy(h, fired, 256, hipMemcpyDeviceToHost));
        unsigned long long total = 0; for (int k = 0; k < 64; ++k) total += h[k];
        for (int kind = 0; kind < 2; ++kind) {
            float best = 1e9f;
            for (int rep = 0; rep < reps; ++rep) {
                ++salt;
                CHECK(hipEventRecord(e0, 0));
                if (kind == 0) sparse_boxes<kStore><<<(n_boxes + 255u) / 256u, 256>>>(buf, n_boxes, 1u, nullptr);
                else sparse_boxes<kMin64><<<(n_boxes + 255u) / 256u, 256>>>(buf, n_boxes, 1u, nullptr);
                CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
                float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (rep > 0 && ms < best) best = ms;
            }
            printf("%-10s %-7s %8.2f us  (%llu operations from %u lanes in four turns: %.1f G operations/s)\n", "boxes2x2", kind ? "min64" : "st", best * 1e3f, total, n_boxes,
                   total / best / 1e6f);
        }
        CHECK(hipFree(fired));
    }
    CHECK(hipGetLastError());
    CHECK(hipFree(buf)); CHECK(hipFree(sink));
    return 0;
}
