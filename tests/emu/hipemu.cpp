// TEST INFRASTRUCTURE ONLY: the CPU execution engine behind tests/emu/hip/hip_runtime.h.
// One OS thread; every work-item of the workgroup being executed is a fibre (own stack, hand-written
// x86-64 register switch: swapcontext costs two system calls per switch).  A fibre runs
// until it reaches __syncthreads, a wave-wide exchange (__ballot/__shfl*) or the end of the kernel;
// when every live fibre is parked the scheduler releases the waves whose lanes all wait at one call
// site (a wave-wide operation that only part of a wave's live lanes reaches is reported as an error
// rather than emulated), and when only barrier waiters are left it releases the barrier.  Device and pinned allocations sit between
// no-access pages, so an out-of-range load or store of a kernel raises SIGSEGV here instead of taking
// a GPU down; the handler turns it into a sticky hipErrorIllegalAddress that names the kernel, the
// workgroup, the work-item and the allocation next to the address.
#include "hip/hip_runtime.h"

#include <signal.h>
#include <sys/mman.h>
#include <ucontext.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <string>
#include <vector>

struct hipemu_stream { int unused; };
struct hipemu_event { std::chrono::steady_clock::time_point t; bool recorded; };

// hipemu_switch(&save_sp, load_sp): park the caller (callee-saved registers on its stack, stack
// pointer to *save_sp) and continue the context whose stack pointer is load_sp.
extern "C" void hipemu_switch(void** save_sp, void* load_sp);
asm(R"(
    .text
    .globl hipemu_switch
    .hidden hipemu_switch
    .type hipemu_switch,@function
hipemu_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
    .size hipemu_switch, .-hipemu_switch
)");

namespace hipemu {

Lane* cur = nullptr;
int g_fuzz_ulp = [] { const char* e = std::getenv("HIPEMU_FUZZ_ULP"); return e ? std::max(0, std::atoi(e)) : 0; }();
uint3 g_block_idx = {0, 0, 0};
dim3 g_block_dim, g_grid_dim;

namespace {

constexpr size_t kPage = 4096;
constexpr size_t kStackBytes = 256 * 1024;
constexpr uint32_t kMaxBlock = 1024;

enum State { RUNNABLE, AT_BARRIER, AT_WAVE, DONE };
struct Fiber {
    Lane lane;
    void* sp;          // saved stack pointer while the fibre is not running
    State state;
    const void* site;
    uint64_t value;
    bool pred;
    const WaveResult* result;
};

std::vector<Fiber>* g_fibers = nullptr;
char* g_stacks = nullptr;
ucontext_t g_fault_ctx;   // where the fault handler returns to (run_grid)
void* g_sched_sp = nullptr;
volatile int g_fault_flag = 0;
Fiber* g_running = nullptr;
const std::function<void()>* g_body = nullptr;
const char* g_kernel_name = "";
bool g_in_kernel = false;
WaveResult g_wave_result[kMaxBlock / 64];

int g_error = hipSuccess;       // sticky once a kernel has faulted, like a dead GPU context
std::string g_error_text;
struct sigaction g_old_segv, g_old_bus;
bool g_handlers_installed = false;

struct Alloc { size_t bytes; void* map_base; size_t map_len; bool host; };
std::map<uintptr_t, Alloc> g_allocs;   // keyed by user pointer

enum Order { FORWARD, REVERSE, SHUFFLE };
Order schedule_order()
{
    static const Order o = [] {
        const char* e = std::getenv("HIPEMU_ORDER");
        if (e && !std::strcmp(e, "reverse")) return REVERSE;
        if (e && !std::strcmp(e, "shuffle")) return SHUFFLE;
        return FORWARD;
    }();
    return o;
}

std::string where_am_i()
{
    char buf[256];
    if (g_running)
        std::snprintf(buf, sizeof buf, "kernel %s, block (%u,%u,%u), thread (%u,%u,%u)", g_kernel_name, g_block_idx.x, g_block_idx.y, g_block_idx.z,
                      g_running->lane.thread_idx.x, g_running->lane.thread_idx.y, g_running->lane.thread_idx.z);
    else
        std::snprintf(buf, sizeof buf, "kernel %s", g_kernel_name);
    return buf;
}

std::string describe_address(uintptr_t a)
{
    char buf[256];
    const Alloc* best = nullptr;
    uintptr_t best_ptr = 0, best_dist = ~(uintptr_t)0;
    for (const auto& kv : g_allocs) {
        const uintptr_t lo = kv.first, hi = kv.first + kv.second.bytes;
        const uintptr_t d = a < lo ? lo - a : (a >= hi ? a - hi + 1 : 0);
        if (d < best_dist) { best_dist = d; best = &kv.second; best_ptr = lo; }
    }
    if (!best) { std::snprintf(buf, sizeof buf, "address %p (no allocation known)", (void*)a); return buf; }
    if (a < best_ptr)
        std::snprintf(buf, sizeof buf, "address %p is %zu bytes BEFORE the %s allocation of %zu bytes at %p", (void*)a, (size_t)(best_ptr - a),
                      best->host ? "pinned host" : "device", best->bytes, (void*)best_ptr);
    else
        std::snprintf(buf, sizeof buf, "address %p is at offset %zu of the %s allocation of %zu bytes at %p (past its end)", (void*)a,
                      (size_t)(a - best_ptr), best->host ? "pinned host" : "device", best->bytes, (void*)best_ptr);
    return buf;
}

void fault_handler(int sig, siginfo_t* info, void* uctx)
{
    if (!g_in_kernel) {   // not ours: hand the signal back to whoever had it before
        sigaction(SIGSEGV, &g_old_segv, nullptr);
        sigaction(SIGBUS, &g_old_bus, nullptr);
        g_handlers_installed = false;
        (void)uctx;
        return;
    }
    g_error = hipErrorIllegalAddress;
    g_error_text = "emulated GPU memory access fault (signal " + std::to_string(sig) + "): " + where_am_i() + "; " + describe_address((uintptr_t)info->si_addr);
    std::fprintf(stderr, "[hipemu] %s\n", g_error_text.c_str());
    g_fault_flag = 1;
    setcontext(&g_fault_ctx);
}

void install_handlers()
{
    if (g_handlers_installed) return;
    struct sigaction sa;
    std::memset(&sa, 0, sizeof sa);
    sa.sa_sigaction = fault_handler;
    sa.sa_flags = SA_SIGINFO | SA_NODEFER;
    sigemptyset(&sa.sa_mask);
    sigaction(SIGSEGV, &sa, &g_old_segv);
    sigaction(SIGBUS, &sa, &g_old_bus);
    g_handlers_installed = true;
}

void park(State s)
{
    Fiber* f = g_running;
    f->state = s;
    hipemu_switch(&f->sp, g_sched_sp);
}

void fiber_entry()
{
    (*g_body)();
    park(DONE);   // never resumed
    std::abort();
}

void resume(Fiber& f)
{
    g_running = &f;
    cur = &f.lane;
    hipemu_switch(&g_sched_sp, f.sp);
}

bool run_block(uint32_t n)
{
    std::vector<Fiber>& fb = *g_fibers;
    for (uint32_t t = 0; t < n; ++t) {
        Fiber& f = fb[t];
        f.lane.flat = t;
        f.lane.lane = t & 63u;
        f.lane.thread_idx.x = t % g_block_dim.x;
        f.lane.thread_idx.y = (t / g_block_dim.x) % g_block_dim.y;
        f.lane.thread_idx.z = t / (g_block_dim.x * g_block_dim.y);
        f.state = RUNNABLE;
        f.result = nullptr;
        // fresh stack (its lowest page is a no-access page): six zeroed callee-saved registers, then
        // the entry point as the return address of the first switch, then a null return address
        void** top = (void**)(g_stacks + (size_t)(t + 1) * kStackBytes);
        top[-1] = nullptr;
        top[-2] = (void*)&fiber_entry;
        for (int r = 3; r <= 8; ++r) top[-r] = nullptr;
        f.sp = (void*)(top - 8);
    }
    std::vector<uint32_t> order(n);
    for (uint32_t t = 0; t < n; ++t) order[t] = schedule_order() == REVERSE ? n - 1 - t : t;
    if (schedule_order() == SHUFFLE) {
        uint32_t s = 0x9E3779B9u ^ (g_block_idx.x * 7919u + g_block_idx.y * 104729u);
        for (uint32_t t = n; t > 1; --t) { s = s * 1664525u + 1013904223u; std::swap(order[t - 1], order[(s >> 8) % t]); }
    }
    const uint32_t n_waves = (n + 63u) / 64u;
    for (;;) {
        bool ran = false;
        for (uint32_t k = 0; k < n; ++k) {
            Fiber& f = fb[order[k]];
            if (f.state == RUNNABLE) { resume(f); ran = true; }
        }
        g_running = nullptr;
        (void)ran;
        // everyone is parked or done
        uint32_t live = 0, at_barrier = 0, released = 0;
        for (uint32_t t = 0; t < n; ++t) { live += fb[t].state != DONE; at_barrier += fb[t].state == AT_BARRIER; }
        if (!live) return true;
        for (uint32_t w = 0; w < n_waves; ++w) {
            const uint32_t lo = w * 64u, hi = std::min(lo + 64u, n);
            const void* site = nullptr;
            for (uint32_t t = lo; t < hi; ++t)
                if (fb[t].state == AT_WAVE && (!site || (uintptr_t)fb[t].site < (uintptr_t)site)) site = fb[t].site;
            if (!site) continue;
            WaveResult& r = g_wave_result[w];
            r.pred_mask = r.group_mask = 0;
            // Every live lane of the wave must be at this call: which lanes a wave-wide operation inside divergent
            // control flow sees depends on how the hardware reconverges, which cannot be reproduced from call
            // addresses alone.  The kernels keep their wave-wide operations in uniform control flow; anything
            // else is reported instead of being emulated approximately.
            for (uint32_t t = lo; t < hi; ++t)
                if (fb[t].state != DONE && (fb[t].state != AT_WAVE || fb[t].site != site)) {
                    g_running = &fb[t];
                    note_error("a wave-wide operation (__ballot/__shfl*) is not reached by every live lane of its wave");
                    g_running = nullptr;
                    return false;
                }
            for (uint32_t t = lo; t < hi; ++t) {
                Fiber& f = fb[t];
                r.value[t - lo] = 0;
                if (f.state != AT_WAVE || f.site != site) continue;
                r.value[t - lo] = f.value;
                r.group_mask |= 1ull << (t - lo);
                if (f.pred) r.pred_mask |= 1ull << (t - lo);
                f.result = &r;
                f.state = RUNNABLE;
                ++released;
            }
        }
        if (released) continue;
        if (at_barrier == live) {
            for (uint32_t t = 0; t < n; ++t) if (fb[t].state == AT_BARRIER) fb[t].state = RUNNABLE;
            continue;
        }
        note_error("deadlock: live work-items are neither runnable nor waiting");
        return false;
    }
}

void* guarded_alloc(size_t bytes, bool host, unsigned char fill)
{
    const size_t user = std::max<size_t>((bytes + 15) & ~(size_t)15, 16);
    const size_t body = (user + kPage - 1) & ~(kPage - 1);
    const size_t len = body + 2 * kPage;
    char* base = (char*)mmap(nullptr, len, PROT_NONE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (base == (char*)MAP_FAILED) return nullptr;
    if (mprotect(base + kPage, body, PROT_READ | PROT_WRITE) != 0) { munmap(base, len); return nullptr; }
    char* p = base + kPage + body - user;   // the allocation ends right at the trailing no-access page
    std::memset(base + kPage, fill, body);
    g_allocs[(uintptr_t)p] = Alloc{bytes, base, len, host};
    return p;
}

int guarded_free(void* p)
{
    if (!p) return hipSuccess;
    auto it = g_allocs.find((uintptr_t)p);
    if (it == g_allocs.end()) return hipErrorInvalidValue;
    munmap(it->second.map_base, it->second.map_len);
    g_allocs.erase(it);
    return hipSuccess;
}

// is [p, p+n) inside one registered allocation?
bool inside_alloc(const void* p, size_t n)
{
    if (n == 0) return true;
    auto it = g_allocs.upper_bound((uintptr_t)p);
    if (it == g_allocs.begin()) return false;
    --it;
    return (uintptr_t)p + n <= it->first + it->second.bytes;
}

}  // namespace

void note_error(const char* what)
{
    if (g_error == hipSuccess) {
        g_error = hipErrorLaunchFailure;
        g_error_text = std::string("emulated launch failure: ") + what + "; " + where_am_i();
        std::fprintf(stderr, "[hipemu] %s\n", g_error_text.c_str());
    }
}

void refuse_launch(const char* kernel, unsigned max_block)
{
    if (g_error == hipSuccess) {
        g_error = hipErrorLaunchFailure;
        g_error_text = std::string("emulated launch failure: kernel ") + kernel + " launched with a block larger than its __launch_bounds__(" +
                       std::to_string(max_block) + ") or with more than 2^32 work-items along x";
        std::fprintf(stderr, "[hipemu] %s\n", g_error_text.c_str());
    }
}

const WaveResult& wave_exchange(uint64_t value, bool pred, const void* site)
{
    Fiber* f = g_running;
    f->value = value;
    f->pred = pred;
    f->site = site;
    park(AT_WAVE);
    return *f->result;
}

void barrier() { park(AT_BARRIER); }

void run_grid(const char* name, dim3 grid, dim3 block, const std::function<void()>& body)
{
    if (g_error != hipSuccess) return;   // a faulted context runs nothing more
    g_kernel_name = name;
    g_running = nullptr;
    const uint64_t n = (uint64_t)block.x * block.y * block.z;
    if (n == 0 || n > kMaxBlock || grid.x == 0 || grid.y == 0 || grid.z == 0 || grid.y > 65535u || grid.z > 65535u) {
        g_error = hipErrorInvalidValue;
        g_error_text = std::string("invalid launch configuration for kernel ") + name;
        return;
    }
    if (!g_fibers) {
        g_fibers = new std::vector<Fiber>(kMaxBlock);
        g_stacks = (char*)mmap(nullptr, kStackBytes * kMaxBlock, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (g_stacks == (char*)MAP_FAILED) { std::fprintf(stderr, "[hipemu] cannot map fibre stacks\n"); std::abort(); }
        for (uint32_t t = 0; t < kMaxBlock; ++t) mprotect(g_stacks + (size_t)t * kStackBytes, kPage, PROT_NONE);
    }
    install_handlers();
    g_body = &body;
    g_block_dim = block;
    g_grid_dim = grid;
    g_fault_flag = 0;
    getcontext(&g_fault_ctx);
    if (g_fault_flag) {   // arrived here from the fault handler
        g_fault_flag = 0;
        g_in_kernel = false;
        g_running = nullptr;
        return;
    }
    g_in_kernel = true;
    bool ok = true;
    for (uint32_t bz = 0; bz < grid.z && ok; ++bz)
        for (uint32_t by = 0; by < grid.y && ok; ++by)
            for (uint32_t bx = 0; bx < grid.x && ok; ++bx) {
                g_block_idx = uint3{bx, by, bz};
                ok = run_block((uint32_t)n);
            }
    g_in_kernel = false;
    g_running = nullptr;
}

}  // namespace hipemu

using namespace hipemu;

extern "C" {

// test hooks
__attribute__((visibility("default"))) void hipemu_reset_error(void) { g_error = hipSuccess; g_error_text.clear(); }
__attribute__((visibility("default"))) size_t hipemu_live_allocations(void) { return g_allocs.size(); }

hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
hipError_t hipSetDevice(int ordinal) { return ordinal == 0 ? hipSuccess : hipErrorInvalidValue; }
hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int ordinal)
{
    if (ordinal != 0) return hipErrorInvalidValue;
    std::memset(p, 0, sizeof *p);
    std::snprintf(p->name, sizeof p->name, "hipemu (CPU execution of the kernels, tests only)");
    std::snprintf(p->gcnArchName, sizeof p->gcnArchName, "gfx950:emulated");
    p->totalGlobalMem = (size_t)8 << 30;
    p->multiProcessorCount = 1;
    p->warpSize = 64;
    return hipSuccess;
}
hipError_t hipGetLastError(void) { return g_error; }
const char* hipGetErrorString(hipError_t e)
{
    if (e != hipSuccess && e == g_error && !g_error_text.empty()) return g_error_text.c_str();
    switch (e) {
    case hipSuccess: return "no error";
    case hipErrorInvalidValue: return "invalid argument";
    case hipErrorOutOfMemory: return "out of memory";
    case hipErrorIllegalAddress: return "an illegal memory access was encountered";
    case hipErrorLaunchFailure: return "unspecified launch failure";
    default: return "unknown error";
    }
}
hipError_t hipMalloc(void** p, size_t bytes)
{
    if (g_error != hipSuccess) return g_error;
    *p = guarded_alloc(bytes, false, 0xA5);
    return *p ? hipSuccess : hipErrorOutOfMemory;
}
hipError_t hipFree(void* p) { return guarded_free(p); }
hipError_t hipHostMalloc(void** p, size_t bytes, unsigned)
{
    if (g_error != hipSuccess) return g_error;
    *p = guarded_alloc(bytes, true, 0x5A);
    return *p ? hipSuccess : hipErrorOutOfMemory;
}
hipError_t hipHostFree(void* p) { return guarded_free(p); }
hipError_t hipHostGetDevicePointer(void** dev, void* host, unsigned) { *dev = host; return hipSuccess; }
hipError_t hipMemcpy(void* dst, const void* src, size_t bytes, hipMemcpyKind kind)
{
    if (g_error != hipSuccess) return g_error;
    const bool dst_dev = kind == hipMemcpyHostToDevice || kind == hipMemcpyDeviceToDevice;
    const bool src_dev = kind == hipMemcpyDeviceToHost || kind == hipMemcpyDeviceToDevice;
    if ((dst_dev && !inside_alloc(dst, bytes)) || (src_dev && !inside_alloc(src, bytes))) {
        g_error = hipErrorInvalidValue;
        g_error_text = "hipMemcpy range is not inside a device allocation: " + describe_address((uintptr_t)(dst_dev && !inside_alloc(dst, bytes) ? dst : src));
        std::fprintf(stderr, "[hipemu] %s\n", g_error_text.c_str());
        return hipErrorInvalidValue;
    }
    std::memmove(dst, src, bytes);
    return hipSuccess;
}
hipError_t hipMemcpyAsync(void* dst, const void* src, size_t bytes, hipMemcpyKind kind, hipStream_t) { return hipMemcpy(dst, src, bytes, kind); }
hipError_t hipMemsetAsync(void* dst, int value, size_t bytes, hipStream_t)
{
    if (g_error != hipSuccess) return g_error;
    if (!inside_alloc(dst, bytes)) {
        g_error = hipErrorInvalidValue;
        g_error_text = "hipMemsetAsync range is not inside a device allocation: " + describe_address((uintptr_t)dst);
        return hipErrorInvalidValue;
    }
    std::memset(dst, value, bytes);
    return hipSuccess;
}
hipError_t hipStreamCreateWithFlags(hipStream_t* st, unsigned) { *st = new hipemu_stream{0}; return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t st) { delete st; return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return g_error; }
hipError_t hipStreamQuery(hipStream_t) { return g_error; }
hipError_t hipEventCreate(hipEvent_t* e) { *e = new hipemu_event{{}, false}; return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->t = std::chrono::steady_clock::now(); e->recorded = true; return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return g_error; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b)
{
    if (!a->recorded || !b->recorded) return hipErrorInvalidValue;
    *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count();
    return hipSuccess;
}

}  // extern "C"
