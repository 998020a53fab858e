// TEST INFRASTRUCTURE ONLY.  A stand-in for <hip/hip_runtime.h> that lets the rasteriser's kernel
// sources be compiled by the HOST compiler and executed on the CPU, one workgroup at a time, with
// every work-item on its own fibre.  Purpose: run the kernels' indexing and synchronisation under
// guard pages before they are ever launched on a real GPU (an out-of-bounds access on the GPU takes
// the whole machine down), and check their integer results against the CPU oracle without a GPU.
// Nothing under tests/emu is part of the shipped library: the library is built by hipcc for gfx950
// from the same sources and has no CPU path.
//
// Semantics kept: grid/block indexing, wave64 lanes, __syncthreads, __ballot/__shfl* in wave-uniform
// control flow (reported as an error otherwise), sequentially consistent atomics, uninitialised device memory (filled with a
// pattern), no-access pages on both sides of every device allocation.
// Semantics NOT kept: concurrency between workgroups and between waves (lanes run one after another
// between synchronisation points, order selectable for race hunting), timing, hardware rounding of
// rcp/rsq/exp2/log (IEEE host functions are used), fp contraction pragmas.
#pragma once

#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <functional>
#include <tuple>
#include <type_traits>
#include <utility>

// ---- qualifiers ------------------------------------------------------------------------------------
#define __global__
#define __device__
#define __host__
#define __shared__ static
#define __align__(n) __attribute__((aligned(n)))
#define __launch_bounds__(...)
#define __forceinline__ inline __attribute__((always_inline))

// ---- vector types ----------------------------------------------------------------------------------
struct dim3 {
    uint32_t x, y, z;
    constexpr dim3(uint32_t x_ = 1, uint32_t y_ = 1, uint32_t z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint3 { uint32_t x, y, z; };
struct int2 { int x, y; };
struct uint2 { uint32_t x, y; };
struct float2 { float x, y; };
struct alignas(16) int4 { int x, y, z, w; };
struct alignas(16) uint4 { uint32_t x, y, z, w; };
struct alignas(16) float4 { float x, y, z, w; };
struct alignas(16) ulonglong2 { unsigned long long x, y; };
static inline int2 make_int2(int x, int y) { return int2{x, y}; }
static inline uint2 make_uint2(uint32_t x, uint32_t y) { return uint2{x, y}; }
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
static inline int4 make_int4(int x, int y, int z, int w) { return int4{x, y, z, w}; }
static inline uint4 make_uint4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { return uint4{x, y, z, w}; }
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
static inline ulonglong2 make_ulonglong2(unsigned long long x, unsigned long long y) { return ulonglong2{x, y}; }

// ---- runtime API (the subset the library calls) ----------------------------------------------------
typedef int hipError_t;
enum : int {
    hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorInvalidSymbol = 13, hipErrorNoDevice = 100,
    hipErrorNotReady = 600, hipErrorIllegalAddress = 700, hipErrorLaunchFailure = 719,
};
typedef struct hipemu_stream* hipStream_t;
typedef struct hipemu_event* hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };
enum : unsigned { hipStreamDefault = 0, hipStreamNonBlocking = 1, hipHostMallocDefault = 0, hipHostMallocMapped = 2, hipEventDefault = 0, hipEventDisableTiming = 2 };
struct hipDeviceProp_t {
    char name[256];
    char gcnArchName[256];
    size_t totalGlobalMem;
    int multiProcessorCount;
    int warpSize;
};

extern "C" {
hipError_t hipGetDeviceCount(int* n);
// a __device__ variable is an ordinary host variable here; its "device address" is its address
#define HIP_SYMBOL(x) (&(x))
static inline hipError_t hipGetSymbolAddress(void** ptr, const void* symbol) { *ptr = const_cast<void*>(symbol); return 0; }
hipError_t hipSetDevice(int ordinal);
hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int ordinal);
hipError_t hipGetLastError(void);
const char* hipGetErrorString(hipError_t e);
hipError_t hipMalloc(void** p, size_t bytes);
hipError_t hipFree(void* p);
hipError_t hipHostMalloc(void** p, size_t bytes, unsigned flags);
hipError_t hipHostFree(void* p);
hipError_t hipHostGetDevicePointer(void** dev, void* host, unsigned flags);
hipError_t hipMemcpy(void* dst, const void* src, size_t bytes, hipMemcpyKind kind);
hipError_t hipMemcpyAsync(void* dst, const void* src, size_t bytes, hipMemcpyKind kind, hipStream_t st);
hipError_t hipMemsetAsync(void* dst, int value, size_t bytes, hipStream_t st);
hipError_t hipStreamCreateWithFlags(hipStream_t* st, unsigned flags);
hipError_t hipStreamDestroy(hipStream_t st);
hipError_t hipStreamSynchronize(hipStream_t st);
hipError_t hipStreamQuery(hipStream_t st);
hipError_t hipEventCreate(hipEvent_t* e);
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned flags);
hipError_t hipEventDestroy(hipEvent_t e);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t st);
hipError_t hipEventSynchronize(hipEvent_t e);
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b);
}

// ---- execution engine ------------------------------------------------------------------------------
namespace hipemu {

struct Lane {                // what a work-item knows about itself
    uint3 thread_idx;
    uint32_t flat;           // x + y*dim.x + z*dim.x*dim.y
    uint32_t lane;           // flat & 63
};
extern Lane* cur;            // the work-item that is running now
extern int g_fuzz_ulp;       // HIPEMU_FUZZ_ULP=n: results of rcp/rsq/exp2/log move by up to n ulp
extern uint3 g_block_idx;
extern dim3 g_block_dim, g_grid_dim;

struct WaveResult {          // outcome of one wave-wide exchange
    uint64_t value[64];
    uint64_t pred_mask;      // lanes of the group whose predicate was true
    uint64_t group_mask;     // lanes that took part
};
// Blocks the calling lane until the lanes of its wave that reach the same call site have all arrived.
const WaveResult& wave_exchange(uint64_t value, bool pred, const void* site);
void barrier();
void run_grid(const char* name, dim3 grid, dim3 block, const std::function<void()>& body);
void note_error(const char* what);
void refuse_launch(const char* kernel, unsigned max_block);

struct Launch {
    dim3 grid, block;
    unsigned max_block = 0;   // the kernel's __launch_bounds__, when the build script could read it from the source
    Launch(dim3 g, dim3 b, size_t /*dynamic_lds*/ = 0, hipStream_t /*stream*/ = nullptr) : grid(g), block(b) {}
    Launch& bounds(unsigned n) { max_block = n; return *this; }
    template <class... P, class... A>
    void go(const char* name, void (*kernel)(P...), A&&... a) const
    {
        // what the GPU refuses at launch: more work-items in a block than the kernel was compiled for, or more
        // than 2^32 work-items along x
        if ((max_block && block.x * block.y * block.z > max_block) || (uint64_t)grid.x * block.x > 0xFFFFFFFFull) {
            refuse_launch(name, max_block);
            return;
        }
        std::tuple<std::decay_t<P>...> params(std::forward<A>(a)...);
        run_grid(name, grid, block, [&]() { std::apply(kernel, params); });
    }
};

template <class T> inline uint64_t to_bits(T v) { static_assert(sizeof(T) <= 8, "shuffle of a wide type"); uint64_t b = 0; std::memcpy(&b, &v, sizeof(T)); return b; }
template <class T> inline T from_bits(uint64_t b) { T v; std::memcpy(&v, &b, sizeof(T)); return v; }

}  // namespace hipemu

#define threadIdx (::hipemu::cur->thread_idx)
#define blockIdx (::hipemu::g_block_idx)
#define blockDim (::hipemu::g_block_dim)
#define gridDim (::hipemu::g_grid_dim)

// ---- device intrinsics -----------------------------------------------------------------------------
// A call site of a wave-wide operation is identified by its source line, not by its address: the host compiler clones
// code (loop unrolling, jump threading on a lane-dependent condition), so that lanes at the same point of the kernel
// can sit at different addresses of the same source-level call.
#define HIPEMU_SITE_OF_LINE(line) reinterpret_cast<const void*>(static_cast<uintptr_t>(line))

static inline void __syncthreads() { ::hipemu::barrier(); }
static inline void __threadfence() {}   // one work-item runs at a time: every store is visible at once

__attribute__((noinline)) static uint64_t hipemu_ballot(bool pred, int line) { return ::hipemu::wave_exchange(0, pred, HIPEMU_SITE_OF_LINE(line)).pred_mask; }
#define __ballot(pred) hipemu_ballot((pred) != 0, __LINE__)

// reads lane `src` of the exchange; `line` is the source line of the __shfl* call in the kernel
template <class T>
__attribute__((noinline)) static T hipemu_shfl_abs(T v, int src, int line)
{
    const ::hipemu::WaveResult& r = ::hipemu::wave_exchange(::hipemu::to_bits(v), false, HIPEMU_SITE_OF_LINE(line));
    const int s = src & 63;
    if (!((r.group_mask >> s) & 1ull)) { ::hipemu::note_error("__shfl reads a lane that is not active at the call"); return T(); }
    return ::hipemu::from_bits<T>(r.value[s]);
}
template <class T> __attribute__((always_inline)) static inline T __shfl(T v, int src, int line = __builtin_LINE()) { return hipemu_shfl_abs(v, src, line); }
template <class T> __attribute__((always_inline)) static inline T __shfl_down(T v, unsigned delta, int line = __builtin_LINE())
{
    const int me = (int)::hipemu::cur->lane;
    return hipemu_shfl_abs(v, me + (int)delta < 64 ? me + (int)delta : me, line);
}
template <class T> __attribute__((always_inline)) static inline T __shfl_up(T v, unsigned delta, int line = __builtin_LINE())
{
    const int me = (int)::hipemu::cur->lane;
    return hipemu_shfl_abs(v, me >= (int)delta ? me - (int)delta : me, line);
}

static inline int __mul24(int a, int b) { return (int)((uint32_t)((int32_t)((uint32_t)a << 8) >> 8) * (uint32_t)((int32_t)((uint32_t)b << 8) >> 8)); }   // low 32 bits of the product of the low 24 bits, sign-extended
static inline unsigned __umul24(unsigned a, unsigned b) { return (a & 0xFFFFFFu) * (b & 0xFFFFFFu); }
static inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
static inline int __popc(unsigned x) { return __builtin_popcount(x); }
static inline int __ffsll(long long x) { return __builtin_ffsll(x); }
static inline int __ffs(int x) { return __builtin_ffs(x); }
static inline int __clz(int x) { return x ? __builtin_clz((unsigned)x) : 32; }
static inline uint32_t __float_as_uint(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
static inline float __uint_as_float(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }

static inline uint32_t hipemu_mbcnt_lo(uint32_t mask, uint32_t base) { const uint32_t l = ::hipemu::cur->lane; return base + (uint32_t)__builtin_popcount(mask & (l >= 32 ? 0xFFFFFFFFu : ((1u << l) - 1u))); }
static inline uint32_t hipemu_mbcnt_hi(uint32_t mask, uint32_t base) { const uint32_t l = ::hipemu::cur->lane; return base + (uint32_t)__builtin_popcount(mask & (l <= 32 ? 0u : ((1u << (l - 32)) - 1u))); }
#define __builtin_amdgcn_mbcnt_lo(m, b) hipemu_mbcnt_lo((m), (b))
#define __builtin_amdgcn_mbcnt_hi(m, b) hipemu_mbcnt_hi((m), (b))
// The hardware's rcp/rsq/exp2/log are 1-ulp approximations.  With HIPEMU_FUZZ_ULP=n every result is
// moved by -n..+n ulp (a fixed pseudo-random function of the result), which shows whether code that
// uses them tolerates any such rounding, not just the host's.
static inline float hipemu_approx(float v)
{
    if (!::hipemu::g_fuzz_ulp || !std::isfinite(v) || v == 0.0f) return v;
    uint32_t b; std::memcpy(&b, &v, 4);
    uint32_t h = b * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    const uint32_t span = (uint32_t)::hipemu::g_fuzz_ulp;   // 1: -1..+1 ulp, 2: -2..+2 ulp, ...
    b += (h % (2u * span + 1u)) - span;
    std::memcpy(&v, &b, 4);
    return v;
}
#define __builtin_amdgcn_rcpf(x) hipemu_approx(1.0f / (x))
#define __builtin_amdgcn_rsqf(x) hipemu_approx(1.0f / sqrtf(x))
#define __builtin_amdgcn_exp2f(x) hipemu_approx(exp2f(x))
#define __builtin_amdgcn_logf(x) hipemu_approx(log2f(x))

// atomics: lanes never run concurrently here, so plain read-modify-write is exact
template <class T> static inline T atomicAdd(T* p, T v) { const T o = *p; *p = (T)(o + v); return o; }
template <class T> static inline T atomicSub(T* p, T v) { const T o = *p; *p = (T)(o - v); return o; }
template <class T> static inline T atomicMin(T* p, T v) { const T o = *p; if (v < o) *p = v; return o; }
template <class T> static inline T atomicMax(T* p, T v) { const T o = *p; if (v > o) *p = v; return o; }
template <class T> static inline T atomicCAS(T* p, T expect, T v) { const T o = *p; if (o == expect) *p = v; return o; }
template <class T> static inline T atomicExch(T* p, T v) { const T o = *p; *p = v; return o; }
template <class T> static inline T atomicOr(T* p, T v) { const T o = *p; *p = (T)(o | v); return o; }

// min / max as device code sees them (same-type overloads; mixed signedness promotes like HIP's)
static inline int min(int a, int b) { return a < b ? a : b; }
static inline int max(int a, int b) { return a > b ? a : b; }
static inline unsigned min(unsigned a, unsigned b) { return a < b ? a : b; }
static inline unsigned max(unsigned a, unsigned b) { return a > b ? a : b; }
static inline long long min(long long a, long long b) { return a < b ? a : b; }
static inline long long max(long long a, long long b) { return a > b ? a : b; }
static inline unsigned long long min(unsigned long long a, unsigned long long b) { return a < b ? a : b; }
static inline unsigned long long max(unsigned long long a, unsigned long long b) { return a > b ? a : b; }
static inline unsigned long min(unsigned long a, unsigned long b) { return a < b ? a : b; }
static inline unsigned long max(unsigned long a, unsigned long b) { return a > b ? a : b; }
static inline long min(long a, long b) { return a < b ? a : b; }
static inline long max(long a, long b) { return a > b ? a : b; }
static inline unsigned min(int a, unsigned b) { return (unsigned)a < b ? (unsigned)a : b; }
static inline unsigned min(unsigned a, int b) { return a < (unsigned)b ? a : (unsigned)b; }
static inline unsigned max(int a, unsigned b) { return (unsigned)a > b ? (unsigned)a : b; }
static inline unsigned max(unsigned a, int b) { return a > (unsigned)b ? a : (unsigned)b; }
static inline float min(float a, float b) { return fminf(a, b); }
static inline float max(float a, float b) { return fmaxf(a, b); }
using std::isinf;
using std::isnan;
