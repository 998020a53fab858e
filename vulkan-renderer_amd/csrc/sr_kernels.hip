// HIP kernels of the draw path for gfx950 (wave64): vertex transform, triangle set-up + tile binning,
// per-tile rasterisation into an LDS visibility tile, deferred fragment shading, mip-chain builder.
// Host launchers at the bottom are what sr_api.hip calls.
#include "sr_device.h"
#include "sr_launch.h"

#include <algorithm>
#include <cmath>
#include <cstring>

namespace sr {

// =================================================================================================
// mesh upload helper: split the reference's 32-byte vertex records into position planes
// =================================================================================================
__global__ __launch_bounds__(256) void split_positions_kernel(const float4* __restrict__ aos /* 2 float4 per vertex */, uint32_t n,
                                                              float* __restrict__ px, float* __restrict__ py, float* __restrict__ pz)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= plane_stride(n)) return;
    const float4 a = i < n ? aos[2 * (size_t)i] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);   // the padding of the planes is zeroed
    px[i] = a.x; py[i] = a.y; pz[i] = a.z;
}

// =================================================================================================
// stage 0: start of a pass.  Clears the visibility buffer of the views of the pass (depth 1.0, no triangle), zeroes
// the binning counters, and on the first pass of a draw pulls the per-view transforms from pinned host memory -- so a
// draw issues kernels only, no small copies or memsets.
// Only the parts of the visibility buffer that hold anything are cleared: the shading pass, which reads every word of
// it anyway, leaves one flag per kFlagBlockW x kFlagBlockH pixels (its workgroup shape) saying whether any sample
// of the block is covered; a block whose flag is clear holds nothing but the clear value.  In the 10 M-triangle 4K frame
// 77 % of the frame is background: 15 MB of stores instead of 66.  One work-item per two pixels of a row.
// =================================================================================================
struct BeginWork {
    unsigned long long* vis;     // the target's whole visibility buffer
    BlockFlags flags;
    int rw, rh, S;
    uint32_t view0, n_views;     // views of the pass
    uint4* zero4; size_t n_zero4;   // binning counters to zero
    uint32_t clear_group;           // flag blocks a wave looks at per turn (clear_flagged_blocks)
};
// The visibility words of the flagged blocks of the pass's views are cleared wave by wave: a wave looks at the flags of
// w.clear_group consecutive blocks per turn (one lane each; 1 for a small frame, whose few blocks should be spread over many waves,
// up to 8 for a large one) and clears the flagged ones of them one after the other, 64 words (512 contiguous bytes of a row) per
// store instruction.  A frame that is mostly background costs little more than reading its flags.
__device__ inline void clear_flagged_blocks(const BeginWork& w, size_t wave_id, size_t n_waves, uint32_t lane)
{
    const uint32_t per_view = (uint32_t)(w.flags.nbx * w.flags.nby);
    const size_t n_blocks = (size_t)per_view * w.n_views;
    const uint32_t log2_s = w.S == 4 ? 2u : 0u;                                   // 1 or 4 samples
    const uint32_t row_shift = 5u + log2_s, block_words = (kFlagBlockW * kFlagBlockH) << log2_s;
    static_assert(kFlagBlockW == 32, "a block row is 32 pixels");
    for (size_t first = wave_id * w.clear_group; first < n_blocks; first += n_waves * w.clear_group) {
        const size_t mine = first + lane;
        const uint32_t flag = lane < w.clear_group && mine < n_blocks ? w.flags.flags[(size_t)w.view0 * per_view + mine] : 0u;
        for (uint64_t todo = __ballot(flag != 0u); todo; todo &= todo - 1ull) {
            const size_t blk = first + (uint32_t)(__ffsll((long long)todo) - 1);
            const uint32_t view = w.view0 + (uint32_t)(blk / per_view), in_view = (uint32_t)(blk % per_view);
            const int x0 = (int)(in_view % (uint32_t)w.flags.nbx) * (int)kFlagBlockW, y0 = (int)(in_view / (uint32_t)w.flags.nbx) * (int)kFlagBlockH;
            unsigned long long* origin = w.vis + ((((size_t)view * w.rh + y0) * w.rw + x0) << log2_s);
            for (uint32_t word = lane; word < block_words; word += 64u) {
                const uint32_t row = word >> row_shift, col = word & ((1u << row_shift) - 1u);
                if (x0 + (int)(col >> log2_s) < w.rw && y0 + (int)row < w.rh) origin[(((size_t)row * w.rw) << log2_s) + col] = kVisClear;
            }
        }
        if (flag != 0u) w.flags.flags[(size_t)w.view0 * per_view + mine] = 0u;   // down again: the kernels of this draw set it where they write
    }
}
__device__ inline void begin_work(const BeginWork& w, size_t gtid, size_t gstride)
{
    for (size_t i = gtid; i < w.n_zero4; i += gstride) w.zero4[i] = make_uint4(0u, 0u, 0u, 0u);
    clear_flagged_blocks(w, gtid >> 6, gstride >> 6, (uint32_t)(gtid & 63u));   // (workgroups are whole waves: 256 work-items)
}

__global__ __launch_bounds__(256) void begin_pass_kernel(BeginWork w, const uint4* __restrict__ views_src, uint4* __restrict__ views_dst, uint32_t n_views4)
{
    const size_t gtid = (size_t)blockIdx.x * 256u + threadIdx.x, gstride = (size_t)gridDim.x * 256u;
    if (blockIdx.x == 0)
        for (uint32_t i = threadIdx.x; i < n_views4; i += 256u) views_dst[i] = views_src[i];
    begin_work(w, gtid, gstride);
}

// The per-view transforms of a draw that may not touch staging memory: they arrive as kernel arguments.
struct ViewPack { ViewXform v[kMaxPackedViews]; };
__global__ __launch_bounds__(64) void store_views_kernel(ViewPack pack, uint32_t n, ViewXform* __restrict__ dst)
{
    constexpr uint32_t kWords = sizeof(ViewXform) / 4;
    const uint32_t* src = reinterpret_cast<const uint32_t*>(&pack);
    for (uint32_t i = threadIdx.x; i < n * kWords; i += 64u) reinterpret_cast<uint32_t*>(dst)[i] = src[i];
}

// =================================================================================================
// stage 1: vertex shader (shaders/shader.vert:20-27) + perspective divide + viewport + snapping
// One work-item per G consecutive vertices, for a range of the views of the launch: the positions are loaded once (one load of
// G floats from each of the three planes) and projected with one view's transform after the other, so a batch of views reads the
// mesh once instead of once per view (the 128-view share of the 50 M-triangle batch: 14.4 -> 8.6 ms).  All plane strides are
// multiples of four vertices; what is computed for the padding of the last group lands in the padding of the output planes and is
// never read.  Measured on MI355X: G = 4 is the fastest for the one-view frame (fused with the start of the pass: 34 -> 29 us for
// 5 M vertices), G = 2 for the view batch.
// =================================================================================================
constexpr uint32_t kFusedVertexGroup = 4, kBatchVertexGroup = 2;
constexpr uint32_t kVertexViewsPerBlock = 8;   // views a workgroup of the batch vertex kernel goes through with its positions loaded once
template <uint32_t G> struct VecOf;
template <> struct VecOf<1> { typedef float F; typedef int2 I; };
template <> struct VecOf<2> { typedef float2 F; typedef int4 I; };
template <> struct VecOf<4> { typedef float4 F; struct __attribute__((aligned(16))) I { int4 a, b; }; };
template <uint32_t G, class Views>
__device__ inline void vertex_group(const Views& views, uint32_t view_begin, uint32_t view_end, const float* __restrict__ px, const float* __restrict__ py,
                                    const float* __restrict__ pz, uint32_t n, uint32_t group, int W, int H, const ScreenPlanes& out,
                                    ScreenVertex* __restrict__ debug_out, float4* __restrict__ clip_out)
{
    typedef typename VecOf<G>::F FV;
    typedef typename VecOf<G>::I IV;
    if (G * group >= n) return;
    union { FV v; float f[G]; } X, Y, Z;
    X.v = reinterpret_cast<const FV*>(px)[group]; Y.v = reinterpret_cast<const FV*>(py)[group]; Z.v = reinterpret_cast<const FV*>(pz)[group];
    for (uint32_t view = view_begin; view < view_end; ++view) {
        union { IV v; int i[2 * G]; } xy;
        union { FV v; float f[G]; } z;
#pragma unroll
        for (uint32_t k = 0; k < G; ++k) {
            float clip[4];
            const ScreenVertex sv = project_vertex(views[view], X.f[k], Y.f[k], Z.f[k], W, H, clip);
            xy.i[2 * k] = sv.rw == 0.0f ? kBadVertexX : sv.X; xy.i[2 * k + 1] = sv.Y; z.f[k] = sv.z;
            const uint32_t i = G * group + k;
            if ((debug_out || clip_out) && i < n) {   // (the debugging entry point only)
                if (debug_out) debug_out[(size_t)view * n + i] = sv;
                if (clip_out) clip_out[(size_t)view * n + i] = make_float4(clip[0], clip[1], clip[2], clip[3]);
            }
        }
        const size_t o = (size_t)view * out.stride + G * (size_t)group;
        *reinterpret_cast<IV*>(out.xy + o) = xy.v;
        if (out.z) *reinterpret_cast<FV*>(out.z + o) = z.v;      // (the same for the whole launch)
    }
}

// The start of a pass folded into the vertex kernel (a draw of a few views, whose transforms fit the kernel arguments):
// the same launch clears the flagged blocks of the visibility buffer, zeroes the binning counters and leaves the
// transforms in device memory for the later kernels -- its stores ride along with the vertex stage's loads, and the pass
// is one launch shorter.  Everything it writes is first read by the kernels that follow.
__global__ __launch_bounds__(256) void vertex_begin_kernel(const float* __restrict__ px, const float* __restrict__ py, const float* __restrict__ pz,
                                                           uint32_t n, ViewPack pack, ViewXform* __restrict__ views_dst, int W, int H,
                                                           ScreenPlanes out, BeginWork w, CullUse cull)
{
    const size_t gtid = (size_t)blockIdx.x * 256u + threadIdx.x, gstride = (size_t)gridDim.x * 256u;
    if (blockIdx.x == 0) {
        constexpr uint32_t kWords = sizeof(ViewXform) / 4;
        const uint32_t* src = reinterpret_cast<const uint32_t*>(&pack);
        for (uint32_t i = threadIdx.x; i < w.n_views * kWords; i += 256u) reinterpret_cast<uint32_t*>(views_dst)[i] = src[i];
    }
    begin_work(w, gtid, gstride);
    const uint32_t group = blockIdx.x * 256u + threadIdx.x;
    // chunk culling (a one-view draw into a restricted rectangle): a wave's kCullVertexGroup vertices are transformed only if a chunk
    // that can touch the rectangle refers to the group -- the same answer for the whole wave
    static_assert(kCullVertexGroup == 64u * kFusedVertexGroup, "one flag per wave of this kernel");
    if (cull.stamp && kFusedVertexGroup * group < n && cull.stamp[0] != cull.epoch && cull.stamp[1u + group / 64u] != cull.epoch) return;
    vertex_group<kFusedVertexGroup>(pack.v, 0u, w.n_views, px, py, pz, n, group, W, H, out, nullptr, nullptr);
}

__global__ __launch_bounds__(256) void vertex_kernel(const float* __restrict__ px, const float* __restrict__ py, const float* __restrict__ pz,
                                                     uint32_t n, const ViewXform* __restrict__ views, uint32_t n_views, uint32_t views_per_block,
                                                     int W, int H, ScreenPlanes out, ScreenVertex* __restrict__ debug_out, float4* __restrict__ clip_out)
{
    const uint32_t view_begin = blockIdx.y * views_per_block;
    vertex_group<kBatchVertexGroup>(views, view_begin, min(view_begin + views_per_block, n_views), px, py, pz, n, blockIdx.x * 256u + threadIdx.x, W, H,
                                    out, debug_out, clip_out);
}

// =================================================================================================
// stage 2a: triangle set-up, culling, tile counting and compaction of the survivors
// =================================================================================================
__device__ inline uint32_t lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// Binning.  Same-address global atomics are expensive (they serialise at the memory side), and a
// coherent mesh makes neighbouring blocks hit the same few tiles at the same time.  So each block
// first builds a histogram of its (triangle, tile) pairs in an LDS hash table, then reserves the
// places of each group in the tile's list with ONE returning global atomic per distinct tile, and
// writes one record per pair that already knows its final position.  The scan then only has to turn
// per-tile totals into offsets and the fill pass is a plain scatter.
#ifndef SR_SETUP_TRIS
#define SR_SETUP_TRIS 3
#endif
// Build-time tuning knobs.  tools/exp_build.py builds variants of the library with them for A/B runs on the GPU, and refuses to write
// a variant whose sources, built for the CPU with the same flags, do not pass the emulated kernel tests first; each knob's legal
// range is asserted where it is used.  The defaults are what was measured best: triangles per lane of the set-up kernel (measured on the 10 M-triangle frame with the
// depths gathered and the samples resolved by the lane that set the triangle up: 1 -> 210 us, 2 -> 131 us, 3 -> 82 us,
// 4 -> 83 us, 6 -> 88 us, 8 -> 97 us -- what counts is the number of independent gathers a lane has in flight, until the
// registers they need cost occupancy; with the covered triangles queued for a second phase: 3 -> 77.8 us, 4 -> 78.0 us,
// 6 -> 79 us, 8 -> 81 us, and the 50 M-triangle view batch 22.3 instead of 29.1 ms), and optional occupancy targets (waves per SIMD) that make the compiler fit the set-up / shading kernel into fewer registers.
#ifdef SR_SETUP_WAVES
static_assert(SR_SETUP_WAVES >= 1 && SR_SETUP_WAVES <= 8, "SR_SETUP_WAVES: waves per SIMD, 1..8");
#define SR_SETUP_ATTR __attribute__((amdgpu_waves_per_eu(SR_SETUP_WAVES, SR_SETUP_WAVES)))
#else
#define SR_SETUP_ATTR
#endif
#ifdef SR_FRAG_WAVES
static_assert(SR_FRAG_WAVES >= 1 && SR_FRAG_WAVES <= 8, "SR_FRAG_WAVES: waves per SIMD, 1..8");
#define SR_FRAG_ATTR __attribute__((amdgpu_waves_per_eu(SR_FRAG_WAVES, SR_FRAG_WAVES)))
#else
#define SR_FRAG_ATTR
#endif
constexpr uint32_t kSetupTris = SR_SETUP_TRIS;               // triangles per lane of the set-up kernel
static_assert(kSetupTris >= 1u && kSetupTris <= 8u, "SR_SETUP_TRIS: 1..8 triangles per lane (the LDS queue holds 256 * kSetupTris entries of 16 bytes, twice in the view loop)");
constexpr uint32_t kSetupBlockTris = 256u * kSetupTris;      // triangles per set-up block
// (one triangle per lane: measured against two and four on the reference's 800 x 600 frame -- binning 15, 21, 31 us --, on the 64-view batch
// and on 8 views of 390 000 fifty-pixel triangles at 4K -- 28, 30, 40 us: a lane's triangles go through the hash table one after the other)
constexpr uint32_t kTrisPerThread = 1;                       // triangles per lane of the binning kernel
constexpr uint32_t kBlockTris = 256u * kTrisPerThread;       // triangles per chunk of the pending list
constexpr uint32_t kBlockRecords = kBlockTris * kMaxSmallTiles;
constexpr uint32_t kHashBits = 11;
constexpr uint32_t kHashSize = 1u << kHashBits;
// The table must be able to hold a chunk's (triangle, tile) pairs at a load that keeps the probe sequences short (pairs that find no
// slot fall back to the per-tile path, which is correct but slow), and the loops that initialise and walk it below move whole
// 16-byte groups per work-item.
static_assert(kHashSize >= 4u && (kHashSize & (kHashSize - 1u)) == 0u, "the binning hash table is a power of two of at least one uint4");
static_assert(kHashSize >= kBlockRecords, "the binning hash table holds every (triangle, tile) pair of a chunk");
constexpr uint32_t kHashEmpty = 0xFFFFFFFFu;
constexpr uint32_t kHashProbes = 24;
// slot of `tile` in the block's table, or kHashSize if the table is too crowded (the caller then
// treats the triangle as a large one)
__device__ inline uint32_t hash_slot(uint32_t* keys, uint32_t tile)
{
    uint32_t h = (tile * 2654435761u) >> (32 - kHashBits);
    for (uint32_t p = 0; p < kHashProbes; ++p) {
        const uint32_t old = atomicCAS(&keys[h], kHashEmpty, tile);
        if (old == kHashEmpty || old == tile) return h;
        h = (h + 1u) & (kHashSize - 1u);
    }
    return kHashSize;
}

// Micro-triangles (bounding box of at most micro_pixels<S>() pixels, under 64 pixels before clipping to
// the stored rectangle) never enter a tile list: the set-up kernel tests their few samples itself and
// resolves the covered ones with a 64-bit atomic min straight into the visibility buffer in HBM.
// Same integer edge functions, same depth arithmetic as the tile rasteriser, so it makes no
// difference to the result which path a triangle takes.
// (32 pixels at one sample -- the 10 M-triangle frame then leaves nothing at all for the tile lists at 4K, its largest boxes being
// 3 x 3, nor at 8K -- and 8 pixels at four)
// (measured: at one sample a limit of 32 instead of 16 pixels leaves the 7680 x 4320 frame of the 10 M-triangle mesh nothing for the tile
// lists -- 0.6 % of its triangles had larger boxes and cost four launches, 0.116 ms of 0.644 -- and makes that frame lean: 0.554 ms; the
// 4K frame is unchanged and the 64-view batch of 90-pixel triangles 0.7 % slower.  At four samples 16 instead of 8 costs the reference's
// own frame 2 %.)
template <int S> __device__ constexpr int micro_pixels() { return S == 1 ? 32 : 8; }

struct MicroEdges {
    int32_t ax, ay, bx, by, cx, cy;   // a = v0, b = v2, c = v1
    int32_t bias_a, bias_b, bias_c;
};
__device__ inline MicroEdges micro_edges(const int2& v0, const int2& v1, const int2& v2)
{
    MicroEdges e;
    e.ax = v0.x; e.ay = v0.y; e.bx = v2.x; e.by = v2.y; e.cx = v1.x; e.cy = v1.y;
    e.bias_a = is_top_left(e.cx - e.bx, e.cy - e.by) ? 0 : -1;
    e.bias_b = is_top_left(e.ax - e.cx, e.ay - e.cy) ? 0 : -1;
    e.bias_c = is_top_left(e.bx - e.ax, e.by - e.ay) ? 0 : -1;
    return e;
}
// Product of two factors of the edge functions of a triangle whose box is under 64 pixels (each below 2^15, the product
// below 2^31).  (The 24-bit multiplier, __mul24, would be exact here too; measured, it made the sample walks longer in
// instructions and the 32-bit tile path 16 % slower.  The facing test of the set-up kernel, outside any loop, does use it.)
__device__ inline int32_t mul_small(int32_t a, int32_t b) { return a * b; }

__device__ inline void micro_eval(const MicroEdges& e, int32_t px, int32_t py, int32_t& Ea, int32_t& Eb, int32_t& Ec)
{
    Ea = mul_small(e.cx - e.bx, py - e.by) - mul_small(e.cy - e.by, px - e.bx);
    Eb = mul_small(e.ax - e.cx, py - e.cy) - mul_small(e.ay - e.cy, px - e.cx);
    Ec = mul_small(e.bx - e.ax, py - e.ay) - mul_small(e.by - e.ay, px - e.ax);
}
// Draws a micro-triangle: every sample of its box is tested with the edge functions, stepped incrementally (exact in integers) from
// their value at the box's first pixel, and each covered sample goes into the visibility buffer with a 64-bit atomic min.  Depth is
// the tile rasteriser's expression of the unbiased edge values.  Returns whether any sample was covered.
// The box is walked as ONE loop over its npx * npy pixels, the step at the end of a row chosen by a select: lanes with boxes of
// different shapes (1 x 4 beside 4 x 1) then take max(npx * npy) turns together instead of max(npx) * max(npy) -- set-up of the
// 800 x 600 four-sample frame 15.2 -> 12.6 us, of the 64-view batch 21.3 -> 17.5 us, frames of sub-pixel triangles unchanged.
template <int S>
__device__ inline bool micro_draw(const MicroEdges& e, int x0, int y0, int npx, int npy, float z0, float z1, float z2, float rA,
                                  uint32_t tri, unsigned long long* vis_view, const FrameGeom& g)
{
    bool any = false;
    const int32_t sxa = -(e.cy - e.by) * kSubpix, sya = (e.cx - e.bx) * kSubpix;   // change per pixel step in x / y
    const int32_t sxb = -(e.ay - e.cy) * kSubpix, syb = (e.ax - e.cx) * kSubpix;
    const int32_t sxc = -(e.by - e.ay) * kSubpix, syc = (e.bx - e.ax) * kSubpix;
    const int32_t wxa = sya - (npx - 1) * sxa, wxb = syb - (npx - 1) * sxb, wxc = syc - (npx - 1) * sxc;   // from the end of a row to the start of the next
    const uint32_t step_row = (uint32_t)(g.rw - (npx - 1)) * S;
#pragma unroll
    for (int s = 0; s < S; ++s) {
        int32_t Ea, Eb, Ec;
        micro_eval(e, x0 * kSubpix + sample_dx(S, s), y0 * kSubpix + sample_dy(S, s), Ea, Eb, Ec);
        Ea += e.bias_a; Eb += e.bias_b; Ec += e.bias_c;
        unsigned long long* p = vis_view + ((size_t)(y0 - g.ry) * g.rw + (x0 - g.rx)) * S + s;
        int i = 0;
        for (int n = npx * npy; n > 0; --n) {
            if ((Ea | Eb | Ec) >= 0) {
                any = true;
                const float l0 = (float)(Ea - e.bias_a) * rA, l2 = (float)(Eb - e.bias_b) * rA, l1 = (float)(Ec - e.bias_c) * rA;
                float z = fmaf(l2, z2, fmaf(l1, z1, l0 * z0));
                z = z + 0.0f;
                if (z >= 0.0f && z < 1.0f) atomicMin(p, ((unsigned long long)__float_as_uint(z) << 32) | (unsigned long long)tri);
            }
            const bool wrap = ++i == npx;
            Ea += wrap ? wxa : sxa; Eb += wrap ? wxb : sxb; Ec += wrap ? wxc : sxc;
            p += wrap ? step_row : (uint32_t)S;
            i = wrap ? 0 : i;
        }
    }
    return any;
}

// The same walk without the resolve: bit n of the result is set when sample n of the box (row by row) is covered.  One sample per
// pixel, at most 32 pixels (micro_pixels<1>()).
__device__ inline uint32_t micro_coverage_mask(const MicroEdges& e, int x0, int y0, int npx, int npy)
{
    const int32_t sxa = -(e.cy - e.by) * kSubpix, sya = (e.cx - e.bx) * kSubpix;
    const int32_t sxb = -(e.ay - e.cy) * kSubpix, syb = (e.ax - e.cx) * kSubpix;
    const int32_t sxc = -(e.by - e.ay) * kSubpix, syc = (e.bx - e.ax) * kSubpix;
    const int32_t wxa = sya - (npx - 1) * sxa, wxb = syb - (npx - 1) * sxb, wxc = syc - (npx - 1) * sxc;
    int32_t Ea, Eb, Ec;
    micro_eval(e, x0 * kSubpix + kHalf, y0 * kSubpix + kHalf, Ea, Eb, Ec);
    Ea += e.bias_a; Eb += e.bias_b; Ec += e.bias_c;
    uint32_t mask = 0, bit = 1u;
    int i = 0;
    for (int n = npx * npy; n > 0; --n) {
        mask |= (Ea | Eb | Ec) >= 0 ? bit : 0u;
        bit <<= 1;
        const bool wrap = ++i == npx;
        Ea += wrap ? wxa : sxa; Eb += wrap ? wxb : sxb; Ec += wrap ? wxc : sxc;
        i = wrap ? 0 : i;
    }
    return mask;
}

// SR_WAVE_SYNC(): the lanes of a wave have written LDS words other lanes of the same wave read next.  On the GPU a wave's LDS
// instructions execute in program order for all its lanes, so this is a fence for the compiler only; where the kernels are executed
// lane by lane on the CPU (the tests) it is the point at which the wave's lanes wait for each other.
#if defined(__HIP_DEVICE_COMPILE__)
#define SR_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
#else
#define SR_WAVE_SYNC() ((void)__ballot(1))
#endif

// The pixels whose sample points the bounding box of a triangle can hold, clipped to the stored rectangle: false if there are none.
template <int S>
__device__ inline bool sample_box(int mnx, int mxx, int mny, int mxy, const FrameGeom& g, int& x0, int& y0, int& npx, int& npy)
{
    int x1, y1;
    pixel_span<S>(mnx, mxx, x0, x1);
    pixel_span<S>(mny, mxy, y0, y1);
    x0 = max(x0, g.rx); x1 = min(x1, g.rx + g.rw - 1);
    y0 = max(y0, g.ry); y1 = min(y1, g.ry + g.rh - 1);
    npx = x1 - x0 + 1; npy = y1 - y0 + 1;
    return npx > 0 && npy > 0;
}

// The same with the spans computed in unsigned arithmetic (which wraps): for callers that evaluate it unconditionally, also on the
// INT32_MIN coordinates of vertices that cannot be projected, and discard the answer for those.
template <int S>
__device__ inline bool sample_box_wrapping(int mnx, int mxx, int mny, int mxy, const FrameGeom& g, int& x0, int& y0, int& npx, int& npy)
{
    int x1, y1;
    if (S == 1) {
        x0 = (int32_t)((uint32_t)mnx + (uint32_t)(kSubpix - 1 - kHalf)) >> 8; x1 = (int32_t)((uint32_t)mxx - (uint32_t)kHalf) >> 8;
        y0 = (int32_t)((uint32_t)mny + (uint32_t)(kSubpix - 1 - kHalf)) >> 8; y1 = (int32_t)((uint32_t)mxy - (uint32_t)kHalf) >> 8;
    } else {
        x0 = mnx >> 8; x1 = mxx >> 8; y0 = mny >> 8; y1 = mxy >> 8;
    }
    x0 = max(x0, g.rx); x1 = min(x1, g.rx + g.rw - 1);
    y0 = max(y0, g.ry); y1 = min(y1, g.ry + g.rh - 1);
    npx = (int32_t)((uint32_t)x1 - (uint32_t)x0 + 1u); npy = (int32_t)((uint32_t)y1 - (uint32_t)y0 + 1u);
    return (npx > 0) & (npy > 0);
}

// The hardware deals workgroups to the eight XCDs round robin by linear workgroup id, and every XCD has an L2 of its own.  Neighbouring
// parts of a mesh share vertices and land on neighbouring pixels, so with the plain order every vertex line is fetched by several
// L2s and the lines of the visibility buffer travel between them under the atomic mins.  xcd_runs renumbers the workgroups so that
// each XCD works through runs of kXcdRun consecutive logical ids (short enough that the eight stay evenly loaded).  A bijection on
// [0, n): the ids past the last whole group of 8 * kXcdRun keep their number.
constexpr uint32_t kXcdRun = 64;
__device__ inline uint32_t xcd_runs(uint32_t b, uint32_t n)
{
    if (kXcdRun <= 1u || b >= n / (8u * kXcdRun) * (8u * kXcdRun)) return b;
    const uint32_t x = b & 7u, j = b >> 3;
    return ((j / kXcdRun) * 8u + x) * kXcdRun + j % kXcdRun;
}

// SR_IN_LOOP(x): x, a 32-bit byte offset that does not change from view to view, is declared opaque at the place of its use.  Without
// it the compiler widens the offsets to 64 bits once, before the view loop, and adds the view's plane base to each with a 64-bit
// vector addition per gather and view (nine register pairs, nine additions); with it every gather is `base in scalar registers + 32-bit
// offset` (the view-loop kernel: 84 -> 65 registers).  Emits no instruction.
#if defined(__HIP_DEVICE_COMPILE__)
#define SR_IN_LOOP(x) asm volatile("" : "+v"(x))
#else
#define SR_IN_LOOP(x) ((void)0)
#endif
// element of a per-view plane at a 32-bit byte offset from the plane's base (the base is the same for the whole wave)
template <class T>
__device__ inline T plane_at(const T* plane, uint32_t byte_offset)
{
    return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(plane) + byte_offset);
}

// Set-up: one lane handles kSetupTris triangles (all index loads first, then all vertex loads, so
// nine independent gathers are in flight per lane).  Back faces, degenerate triangles and boxes
// that contain no sample point are dropped; micro-triangles are resolved here, in a second phase; triangles with
// one or two vertices that cannot be projected go on the clip list; whatever is left is handed to the
// binning kernel through the view's pending list (ballots and prefix counts inside the block, one
// atomic add per block that has any).
// ZCALC: the pass has no depth plane (a lean draw of a batch of views: the plane is a third of what the vertex stage writes, and only the
// triangles that reach phase two -- a few in a hundred of a mesh much finer than the frame -- ever read it); phase two works the three
// depths out again from the positions, with the vertex stage's own operations.
// LOCAL (view loop, a mesh with chunk vertex lists -- sr_device.h): for every view the workgroup first loads the window-space x, y of its
// chunk's distinct vertices into LDS, side by side, and the triangles take their vertices from there.  A queued triangle carries its three
// list positions; phase two, which runs for entries of earlier views too, looks the vertex numbers up in the list and gathers as before.
template <int S, bool VIEW_LOOP, bool STAGED, bool ZCALC = false, bool LOCAL = false>
__global__ __launch_bounds__(256) SR_SETUP_ATTR void setup_kernel(const uint32_t* __restrict__ idx, uint32_t n_tris, ScreenPlanes sp,
                                                    FrameGeom g, BinBuffers b, ClipList clip, unsigned long long* __restrict__ vis, BlockFlags flags,
                                                    uint32_t view0, uint32_t n_views, uint32_t views_per_block, const uint32_t* __restrict__ chunk_visible,
                                                    DepthSource ds, ChunkVerts cv)
{
    static_assert(!ZCALC || !STAGED, "depths are worked out again in the form of phase two that resolves inside the walk");
    static_assert(!LOCAL || VIEW_LOOP, "the chunk's vertices are kept in LDS by the workgroups that go through many views");
    // The queue of phase two.  One view per workgroup: filled in phase one, a barrier, emptied in phase two.  In the view loop a view of
    // a mesh much finer than the frame queues a few dozen triangles per workgroup -- half a wave's worth, and phase two would run for
    // them at half a wave's width once per view.  So the queue is kept over the views (every entry carries its view) and emptied when
    // it holds more than kDrainAt entries or the views are through: whole waves, all four of them.  One barrier per view (after the
    // pushes) and one more after each emptying.  Two counters take the pushes of even and odd views in turn and are never reset: what a
    // counter reads after a view's barrier is final until the view after the next one starts pushing, which no wave does before every
    // wave has passed the next barrier -- so all waves derive the same queue length from it, and where the next view's entries go.
    constexpr uint32_t kDrainAt = 512u;
    constexpr uint32_t kQueueCap = VIEW_LOOP ? kSetupBlockTris + kDrainAt : kSetupBlockTris;     // (a view queues kSetupBlockTris entries at most)
    constexpr uint32_t kQueueWide = LOCAL ? 1u : kQueueCap;      // (LOCAL: a queued triangle's three list positions share one word)
    __shared__ uint32_t s_q_tri[kQueueCap], s_q_i0[kQueueCap], s_q_i1[kQueueWide], s_q_i2[kQueueWide];
    constexpr uint32_t kLocalPer = kChunkVertsMax / 256u;        // vertices of the chunk's list per work-item
    __shared__ int2 s_vxy[LOCAL ? kChunkVertsMax : 1u];            // LOCAL: window-space x, y of the chunk's vertices in the current view
    __shared__ uint32_t s_q_view[VIEW_LOOP ? kQueueCap : 1u];
    __shared__ uint32_t s_q_count[2];
    // Staged resolve (STAGED: one sample, one view per workgroup, a launch of thousands of workgroups).  An atomic minimum costs what
    // the 128-byte lines its instruction touches cost (measured: 64 lanes in 4 lines 155 G operations/s, in 64 lines 25 G/s), and a
    // lane that walks its own box fires in a few of its turns only: instructions with a fifth of the lanes active, each in a line of
    // its own.  So the lanes first only mark their covered samples (a 32-bit mask), the wave counts them and numbers them triangle by
    // triangle, every covered sample is written to the wave's stretch of LDS in that order, and the wave then resolves the stretch
    // with all its lanes: full instructions whose neighbouring lanes are neighbouring samples.  Set-up of the 10 M-triangle frame:
    // 74.6 -> 59.7 us at 3840 x 2160 (1.9 M samples), 268 -> 117 us at 7680 x 4320 (7.8 M).  A launch of a few hundred workgroups is
    // bound by the latency of a workgroup's chain of steps instead, which the two passes lengthen (124 000 triangles at 1080p:
    // 22 -> 33 us): such launches keep the resolve inside the walk.
    static_assert(!STAGED || (S == 1 && !VIEW_LOOP), "the staged resolve is built for one sample and one view per workgroup");
    constexpr bool kStaged = STAGED;
    constexpr uint32_t kStageCap = kStaged ? 160u : 1u;            // covered samples a wave resolves per round
    __shared__ uint32_t s_f_off[4][kStageCap], s_f_z[4][kStageCap], s_f_tri[4][kStageCap];
    __shared__ uint32_t s_inv[33];                                  // s_inv[n] = 65535 / n + 1: bit / n == (bit * s_inv[n]) >> 16 for bit < 64, n <= 32
    if (kStaged && threadIdx.x >= 1u && threadIdx.x <= 32u) s_inv[threadIdx.x] = 65535u / threadIdx.x + 1u;
    const uint32_t turn = xcd_runs(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
    const uint32_t blk = turn % gridDim.x, view_begin = turn / gridDim.x * views_per_block;
    const uint32_t tid = threadIdx.x, lane = lane_id(), wave = tid >> 6;
    // chunk culling (one-view draws into a restricted rectangle only, so never with the view loop): none of the workgroup's triangles
    // can touch a stored pixel -- and their vertices may not even have been transformed
    if (!VIEW_LOOP && chunk_visible && chunk_visible[blk] == 0u) return;   // (the same for the whole workgroup, before any barrier)
    // i0, i1, i2: the three vertices of a triangle as BYTE offsets into a view's xy plane (index * 8; half of it is the offset into the
    // z plane).  A plane holds fewer than 2^29 vertices (checked when the mesh is made), so the offsets fit 32 bits and every gather is
    // a plane's base address, the same for the whole wave, plus a 32-bit offset per lane: no 64-bit address arithmetic per view.
    uint32_t tri[kSetupTris], i0[kSetupTris], i1[kSetupTris], i2[kSetupTris];
    bool in_range[kSetupTris];
#pragma unroll
    for (uint32_t k = 0; k < kSetupTris; ++k) {
        tri[k] = blk * kSetupBlockTris + k * 256u + tid;
        in_range[k] = tri[k] < n_tris;
        const uint32_t tt = in_range[k] ? tri[k] : 0u;
        if constexpr (LOCAL) {      // the same offsets, into s_vxy: list positions * 8
            const uint32_t loc = in_range[k] ? cv.local[tt] : 0u;
            i0[k] = (loc << 3) & 0x1FF8u; i1[k] = (loc >> 7) & 0x1FF8u; i2[k] = (loc >> 17) & 0x1FF8u;
        } else {
            i0[k] = idx[3 * (size_t)tt] << 3; i1[k] = idx[3 * (size_t)tt + 1] << 3; i2[k] = idx[3 * (size_t)tt + 2] << 3;
        }
    }
    // LOCAL: the chunk's distinct vertices, kLocalPer per work-item, as offsets into a view's xy plane
    uint32_t n_local = 0, voff[kLocalPer];
    if constexpr (LOCAL) {
        n_local = cv.count[blk];
#pragma unroll
        for (uint32_t j = 0; j < kLocalPer; ++j) {
            const uint32_t v = j * 256u + tid;
            voff[j] = (v < n_local ? cv.list[(size_t)blk * kChunkVertsMax + v] : 0u) << 3;
        }
    }
    // the indices are loaded once for all the views the workgroup handles (every view of a pass, unless the mesh is small): a batch of
    // views streams the index list once instead of once per view
    // (VIEW_LOOP false = one view per workgroup: without the loop around it the code needs 63 registers instead of 90, which is worth
    // 5 % on a one-view frame of 10 M triangles)
    const uint32_t view_end = VIEW_LOOP ? min(view_begin + views_per_block, n_views) : view_begin + 1u;
    if (tid < 2u) s_q_count[tid] = 0u;
    // LOCAL: a view's vertices are loaded into registers (vnext) while the view before it is worked on, and copied to LDS at the start of
    // their view: with the loads issued only then, every wave of the workgroup waited for them at the barrier that follows
    // (50 M triangles x 16 views: set-up 20.5 ms; loaded a view ahead 13.1 ms; without the lists 15.9 ms).
    int2 vnext[kLocalPer];
    if constexpr (LOCAL) {
#pragma unroll
        for (uint32_t j = 0; j < kLocalPer; ++j)
            vnext[j] = j * 256u + tid < n_local ? plane_at(sp.xy + (size_t)view_begin * sp.stride, voff[j]) : make_int2(0, 0);
    }
    __syncthreads();
    uint32_t q_seen0 = 0, q_seen1 = 0, q_total = 0;   // the two counters as last read after a barrier; entries in the queue (the same in every wave)
    for (uint32_t view = view_begin; view < view_end; ++view) {
    const uint32_t qp = VIEW_LOOP ? (view - view_begin) & 1u : 0u;
    const int2* xy = sp.xy + (size_t)view * sp.stride;
    int2 r0[kSetupTris], r1[kSetupTris], r2[kSetupTris];
    if constexpr (LOCAL) {
        // (the previous view's readers of s_vxy are through: they passed that view's barrier after phase one)
#pragma unroll
        for (uint32_t j = 0; j < kLocalPer; ++j) {
            const uint32_t v = j * 256u + tid;
            if (v < n_local) s_vxy[v] = vnext[j];
        }
        __syncthreads();
        if (view + 1u < view_end) {
            const int2* xy_next = xy + sp.stride;
#pragma unroll
            for (uint32_t j = 0; j < kLocalPer; ++j) {
                SR_IN_LOOP(voff[j]);
                if (j * 256u + tid < n_local) vnext[j] = plane_at(xy_next, voff[j]);
            }
        }
        const char* here = reinterpret_cast<const char*>(s_vxy);
#pragma unroll
        for (uint32_t k = 0; k < kSetupTris; ++k) {
            r0[k] = *reinterpret_cast<const int2*>(here + i0[k]);
            r1[k] = *reinterpret_cast<const int2*>(here + i1[k]);
            r2[k] = *reinterpret_cast<const int2*>(here + i2[k]);
        }
    } else {
#pragma unroll
        for (uint32_t k = 0; k < kSetupTris; ++k) {
            SR_IN_LOOP(i0[k]); SR_IN_LOOP(i1[k]); SR_IN_LOOP(i2[k]);
            r0[k] = plane_at(xy, i0[k]); r1[k] = plane_at(xy, i1[k]); r2[k] = plane_at(xy, i2[k]);
        }
    }

    // Phase one: the cheap tests, for every triangle.  Micro-triangles whose box holds a sample point are queued in LDS (their id and
    // their three vertex offsets); phase two takes them one per work-item, every lane busy, fetches positions and depths (cache hits: the
    // workgroup touched those lines a moment ago), walks the few samples and does the atomic mins.  Run in place, at the one lane in
    // three or ten that holds such a triangle, the sample walk was most of the kernel's instructions.
    uint32_t n_unproj = 0, n_micro = 0;
    bool pend[kSetupTris], queued[kSetupTris];
    // a vertex that cannot be projected has x = INT32_MIN, which then is the smallest x of its triangle.  Triangles with one or two
    // such vertices need the near-plane clipper (a full list drops and counts the triangle): rare, one test for all the wave's triangles
    int mnx_of[kSetupTris];
    bool bad_any = false;
#pragma unroll
    for (uint32_t k = 0; k < kSetupTris; ++k) {
        mnx_of[k] = min(r0[k].x, min(r1[k].x, r2[k].x));
        bad_any |= (mnx_of[k] == kBadVertexX) & in_range[k];   // (no short circuit: branch-free)
    }
    if (__ballot(bad_any) != 0ull) {
#pragma unroll
        for (uint32_t k = 0; k < kSetupTris; ++k) {
            const int bad = (r0[k].x == kBadVertexX) + (r1[k].x == kBadVertexX) + (r2[k].x == kBadVertexX);
            if (in_range[k] && bad > 0 && bad < 3) {
                const uint32_t slot = atomicAdd(&b.counters[kCtrClipListed], 1u);
                if (slot < clip.capacity) clip.entries[slot] = make_uint2(view, tri[k]);
                else ++n_unproj;
            }
        }
    }
    // facing: the sign of the doubled area.  A box under 64 pixels on both axes (differences in unsigned arithmetic: x is INT32_MIN
    // for a vertex that cannot be projected) keeps every factor below 2^15: the 24-bit multiplier is exact and the products fit 32
    // bits; else 64 bits -- a path the wave skips, for its three triangles at once, unless one of them needs it.
    int mxx_of[kSetupTris], mny_of[kSetupTris], mxy_of[kSetupTris];
    bool small_box[kSetupTris], front[kSetupTris];
    bool big_any = false;
#pragma unroll
    for (uint32_t k = 0; k < kSetupTris; ++k) {
        mxx_of[k] = max(r0[k].x, max(r1[k].x, r2[k].x));
        mny_of[k] = min(r0[k].y, min(r1[k].y, r2[k].y)); mxy_of[k] = max(r0[k].y, max(r1[k].y, r2[k].y));
        small_box[k] = (((uint32_t)mxx_of[k] - (uint32_t)mnx_of[k]) | ((uint32_t)mxy_of[k] - (uint32_t)mny_of[k])) < 16384u;
        // (differences in unsigned arithmetic, which wraps: for a larger box the 32-bit value means nothing and is replaced below)
        front[k] = (int32_t)((uint32_t)__mul24((int)((uint32_t)r1[k].x - (uint32_t)r0[k].x), (int)((uint32_t)r2[k].y - (uint32_t)r0[k].y)) -
                             (uint32_t)__mul24((int)((uint32_t)r2[k].x - (uint32_t)r0[k].x), (int)((uint32_t)r1[k].y - (uint32_t)r0[k].y))) < 0;
        big_any |= !small_box[k];
    }
    if (__ballot(big_any) != 0ull) {
#pragma unroll
        for (uint32_t k = 0; k < kSetupTris; ++k)
            if (!small_box[k]) front[k] = (int64_t)(int32_t)((uint32_t)r1[k].x - (uint32_t)r0[k].x) * (int64_t)(r2[k].y - r0[k].y) -
                                          (int64_t)(int32_t)((uint32_t)r2[k].x - (uint32_t)r0[k].x) * (int64_t)(r1[k].y - r0[k].y) < 0;
    }
    // No branches from here on (the kernel is bound by the instructions it issues, scalar ones included, and every nested
    // condition costs a wave several of those whether or not it skips anything): the sample box is worked out for every
    // triangle, in arithmetic that wraps for the coordinates of unprojectable vertices, and the conditions are combined.
#pragma unroll
    for (uint32_t k = 0; k < kSetupTris; ++k) {
        const bool candidate = in_range[k] & (mnx_of[k] != kBadVertexX) & front[k];
        int x0, y0, npx, npy;
        const bool holds_sample = sample_box_wrapping<S>(mnx_of[k], mxx_of[k], mny_of[k], mxy_of[k], g, x0, y0, npx, npy);
        const bool micro = (__mul24(npx, npy) <= micro_pixels<S>()) & small_box[k];   // (npx, npy <= 16384 where it counts)
        queued[k] = candidate & holds_sample & micro;
        pend[k] = candidate & holds_sample & !micro;
    }
    {   // the wave's queued triangles take their places in the workgroup's queue: one LDS atomic per wave
        uint64_t qm[kSetupTris];
        uint32_t n = 0;
#pragma unroll
        for (uint32_t k = 0; k < kSetupTris; ++k) { qm[k] = __ballot(queued[k]); n += (uint32_t)__popcll(qm[k]); }
        uint32_t base = 0;
        if (lane == 0 && n) base = atomicAdd(&s_q_count[qp], n);
        base = (uint32_t)__shfl((int)base, 0) - (qp ? q_seen1 : q_seen0) + q_total;
#pragma unroll
        for (uint32_t k = 0; k < kSetupTris; ++k) {
            if (queued[k]) {
                const uint32_t q = base + (uint32_t)__popcll(qm[k] & ((1ull << lane) - 1ull));
                s_q_tri[q] = tri[k];
                if constexpr (LOCAL) s_q_i0[q] = (i0[k] >> 3) | (i1[k] << 7) | (i2[k] << 17);     // (the three list positions again, 10 bits each)
                else { s_q_i0[q] = i0[k]; s_q_i1[q] = i1[k]; s_q_i2[q] = i2[k]; }
                if constexpr (VIEW_LOOP) s_q_view[q] = view;
            }
            base += (uint32_t)__popcll(qm[k]);
        }
    }
    __syncthreads();
    {   // (all the same in every wave)
        const uint32_t cur = s_q_count[qp];
        q_total += cur - (qp ? q_seen1 : q_seen0);
        if (qp) q_seen1 = cur; else q_seen0 = cur;
    }
    if (!VIEW_LOOP || q_total > kDrainAt || view + 1u == view_end) {
        const float* zp = sp.z + (size_t)view * sp.stride;
        unsigned long long* vis_view = vis + (size_t)(view0 + view) * g.rw * g.rh * S;
        const uint32_t n_q = q_total;
        for (uint32_t q0 = 0; q0 < n_q; q0 += 256u) {   // (the same number of turns for every work-item: the ballot below is reached by all)
            const uint32_t q = q0 + tid;
            bool drawn = false;
            uint32_t qv = view;                         // the entry's view
            uint32_t bx0 = 0, bx1 = 0, by0 = 0, by1 = 0;   // blocks of flags the drawn triangle's box reaches into
            if constexpr (kStaged) {
                uint32_t mask = 0, t = 0, inv = 0;
                int x0 = 0, y0 = 0, npx = 1, npy = 1;
                MicroEdges e{};
                float za = 0.0f, zb = 0.0f, zc = 0.0f, rA = 0.0f;
                if (q < n_q) {
                    t = s_q_tri[q];
                    const uint32_t j0 = s_q_i0[q], j1 = s_q_i1[q], j2 = s_q_i2[q];
                    const int2 a0 = plane_at(xy, j0), a1 = plane_at(xy, j1), a2 = plane_at(xy, j2);
                    za = plane_at(zp, j0 >> 1); zb = plane_at(zp, j1 >> 1); zc = plane_at(zp, j2 >> 1);
                    sample_box<S>(min(a0.x, min(a1.x, a2.x)), max(a0.x, max(a1.x, a2.x)), min(a0.y, min(a1.y, a2.y)), max(a0.y, max(a1.y, a2.y)), g, x0, y0, npx, npy);
                    const int32_t A = -(mul_small(a1.x - a0.x, a2.y - a0.y) - mul_small(a2.x - a0.x, a1.y - a0.y));
                    rA = 1.0f / (float)A;
                    e = micro_edges(a0, a1, a2);
                    mask = micro_coverage_mask(e, x0, y0, npx, npy);
                    inv = s_inv[npx];                          // (1 <= npx <= 32: the box holds a sample point and at most micro_pixels<1>() pixels)
                    drawn = mask != 0u;
                    bx0 = (uint32_t)(x0 - g.rx) / kFlagBlockW; bx1 = (uint32_t)(x0 + npx - 1 - g.rx) / kFlagBlockW;
                    by0 = (uint32_t)(y0 - g.ry) / kFlagBlockH; by1 = (uint32_t)(y0 + npy - 1 - g.ry) / kFlagBlockH;
                }
                // number the wave's covered samples triangle by triangle (every lane of the wave is here: the loop over the queue takes
                // the same number of turns for all)
                const uint32_t cnt = (uint32_t)__popc(mask);
                uint32_t incl = cnt;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) { const uint32_t v = (uint32_t)__shfl_up((int)incl, o); if (lane >= (uint32_t)o) incl += v; }
                const uint32_t total = (uint32_t)__shfl((int)incl, 63);
                uint32_t next = incl - cnt;                    // number of this lane's next covered sample
                // first sample of the box: unbiased edge values, and their steps per pixel in x and y
                int32_t Ea0, Eb0, Ec0;
                micro_eval(e, x0 * kSubpix + kHalf, y0 * kSubpix + kHalf, Ea0, Eb0, Ec0);
                const int32_t sxa = -(e.cy - e.by) * kSubpix, sya = (e.cx - e.bx) * kSubpix;
                const int32_t sxb = -(e.ay - e.cy) * kSubpix, syb = (e.ax - e.cx) * kSubpix;
                const int32_t sxc = -(e.by - e.ay) * kSubpix, syc = (e.bx - e.ax) * kSubpix;
                const uint32_t first_off = (uint32_t)(y0 - g.ry) * (uint32_t)g.rw + (uint32_t)(x0 - g.rx);
                for (uint32_t w0 = 0; w0 < total; w0 += kStageCap) {          // rounds: the same number for every lane of the wave
                    const uint32_t w1 = min(w0 + kStageCap, total);
                    while (mask != 0u && next < w1) {
                        const uint32_t bit = (uint32_t)__ffs((int)mask) - 1u;
                        mask &= mask - 1u;
                        const uint32_t dy = (bit * inv) >> 16, dx = bit - dy * (uint32_t)npx;
                        // (every factor is small: |step| < 2^22, dx, dy < 32; the sums are the walk's values, exact in 32 bits)
                        const int32_t Ea = Ea0 + __mul24((int)dx, sxa) + __mul24((int)dy, sya);
                        const int32_t Eb = Eb0 + __mul24((int)dx, sxb) + __mul24((int)dy, syb);
                        const int32_t Ec = Ec0 + __mul24((int)dx, sxc) + __mul24((int)dy, syc);
                        const float l0 = (float)Ea * rA, l2 = (float)Eb * rA, l1 = (float)Ec * rA;
                        float z = fmaf(l2, zc, fmaf(l1, zb, l0 * za));
                        z = z + 0.0f;
                        const uint32_t slot = next - w0;
                        s_f_off[wave][slot] = z >= 0.0f && z < 1.0f ? first_off + dy * (uint32_t)g.rw + dx : 0xFFFFFFFFu;   // outside the clip volume in z: not drawn
                        s_f_z[wave][slot] = __float_as_uint(z);
                        s_f_tri[wave][slot] = t;
                        ++next;
                    }
                    SR_WAVE_SYNC();
                    for (uint32_t j = lane; j < w1 - w0; j += 64u) {
                        const uint32_t off = s_f_off[wave][j];
                        if (off != 0xFFFFFFFFu) atomicMin(vis_view + off, ((unsigned long long)s_f_z[wave][j] << 32) | (unsigned long long)s_f_tri[wave][j]);
                    }
                    SR_WAVE_SYNC();                             // (the stretch is read before the next round overwrites it)
                }
            } else
            if (q < n_q) {
                const uint32_t t = s_q_tri[q];
                uint32_t j0 = s_q_i0[q], j1, j2;
                if constexpr (LOCAL) {          // list positions -> vertex numbers -> offsets into a view's xy plane
                    const uint32_t* mine = cv.list + (size_t)blk * kChunkVertsMax;
                    j1 = mine[(j0 >> 10) & 1023u] << 3; j2 = mine[(j0 >> 20) & 1023u] << 3; j0 = mine[j0 & 1023u] << 3;
                } else { j1 = s_q_i1[q]; j2 = s_q_i2[q]; }
                const int2* xy_q = xy;
                const float* zp_q = zp;
                unsigned long long* vis_q = vis_view;
                if constexpr (VIEW_LOOP) {          // the planes and the visibility buffer of the entry's view
                    qv = s_q_view[q];
                    xy_q = sp.xy + (size_t)qv * sp.stride; zp_q = sp.z + (size_t)qv * sp.stride;
                    vis_q = vis + (size_t)(view0 + qv) * g.rw * g.rh * S;
                }
                const int2 a0 = plane_at(xy_q, j0), a1 = plane_at(xy_q, j1), a2 = plane_at(xy_q, j2);
                float za, zb, zc;
                if constexpr (ZCALC) {
                    const ViewXform& vx = ds.views[qv];
                    za = project_depth(vx, plane_at(ds.px, j0 >> 1), plane_at(ds.py, j0 >> 1), plane_at(ds.pz, j0 >> 1));
                    zb = project_depth(vx, plane_at(ds.px, j1 >> 1), plane_at(ds.py, j1 >> 1), plane_at(ds.pz, j1 >> 1));
                    zc = project_depth(vx, plane_at(ds.px, j2 >> 1), plane_at(ds.py, j2 >> 1), plane_at(ds.pz, j2 >> 1));
                } else { za = plane_at(zp_q, j0 >> 1); zb = plane_at(zp_q, j1 >> 1); zc = plane_at(zp_q, j2 >> 1); }
                int x0, y0, npx, npy;
                sample_box<S>(min(a0.x, min(a1.x, a2.x)), max(a0.x, max(a1.x, a2.x)), min(a0.y, min(a1.y, a2.y)), max(a0.y, max(a1.y, a2.y)), g, x0, y0, npx, npy);
                const int32_t A = -(mul_small(a1.x - a0.x, a2.y - a0.y) - mul_small(a2.x - a0.x, a1.y - a0.y));
                drawn = micro_draw<S>(micro_edges(a0, a1, a2), x0, y0, npx, npy, za, zb, zc, 1.0f / (float)A, t, vis_q, g);
                // every block the box reaches into may hold something now.  One block as a rule; a box of up to micro_pixels
                // pixels can be a sliver one pixel wide and 32 rows tall (five rows of blocks) or the transpose (two columns).
                static_assert(micro_pixels<S>() <= (int)kFlagBlockW, "a micro-triangle's box spans two block columns at most");
                bx0 = (uint32_t)(x0 - g.rx) / kFlagBlockW; bx1 = (uint32_t)(x0 + npx - 1 - g.rx) / kFlagBlockW;
                by0 = (uint32_t)(y0 - g.ry) / kFlagBlockH; by1 = (uint32_t)(y0 + npy - 1 - g.ry) / kFlagBlockH;
            }
            if (drawn)
                for (uint32_t by = by0; by <= by1; ++by) {
                    flags.at(view0 + qv, bx0, by) = 1u;
                    if (bx1 != bx0) flags.at(view0 + qv, bx1, by) = 1u;
                }
            n_micro += (uint32_t)__popcll(__ballot(drawn));   // of the wave
        }
        if constexpr (VIEW_LOOP) {
            __syncthreads();        // (the entries are read: the next view's pushes may take their places)
            q_total = 0;
        }
    }
    if (n_unproj) atomicAdd(&b.counters[kCtrUnprojectable], n_unproj);   // rare
    // the count of micro-triangles drawn (instrumentation): one atomic per wave that drew any, spread over kMicroSpread counters a
    // cache line apart (ten thousand atomic adds on ONE address cost the kernel 15 % of its time)
    if (lane == 0 && n_micro) atomicAdd(&b.micro_spread[((blk * 4u + wave) % kMicroSpread) * kMicroSpreadStride], n_micro);
    // hand the remaining triangles to the binning kernel: every wave that has any takes its own stretch of the view's pending list
    // (ballots and prefix counts inside the wave, one atomic add)
    uint64_t m[kSetupTris];
    uint32_t wave_total = 0;
#pragma unroll
    for (uint32_t k = 0; k < kSetupTris; ++k) { m[k] = __ballot(pend[k]); wave_total += (uint32_t)__popcll(m[k]); }
    if (wave_total != 0) {   // uniform over the wave
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(&b.pending_total[view * kPendingTotalStride], wave_total);
        base = (uint32_t)__shfl((int)base, 0);
        uint32_t* list = b.pending + (size_t)view * b.pending_stride + base;
        uint32_t run = 0;
#pragma unroll
        for (uint32_t k = 0; k < kSetupTris; ++k) {
            if (pend[k]) list[run + (uint32_t)__popcll(m[k] & ((1ull << lane) - 1ull))] = tri[k];
            run += (uint32_t)__popcll(m[k]);
        }
    }
    }
}

// Binning of the triangles set-up left pending.  The view's pending list is taken in chunks of kBlockTris
// triangles by a fixed number of workgroups; the chunks are visited in a scrambled order (odd multiplier
// modulo a power of two) so that workgroups running at the same time work on distant parts of the list --
// neighbouring chunks come from neighbouring parts of the mesh and would pile their per-tile atomics on
// the same counters.  A frame of sub-pixel triangles leaves next to nothing pending: all but one
// workgroup then leave at once.
template <int S>
__global__ __launch_bounds__(256) void bin_kernel(const uint32_t* __restrict__ idx, ScreenPlanes sp, FrameGeom g, BinBuffers b)
{
    __shared__ __attribute__((aligned(16))) uint32_t s_key[kHashSize];
    __shared__ __attribute__((aligned(16))) uint32_t s_cnt[kHashSize];
    __shared__ uint32_t s_total;
    const uint32_t view = blockIdx.y;
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = lane_id();
    const uint64_t lanes_below = (1ull << lane) - 1ull;
    const uint32_t n_tiles = (uint32_t)(g.tiles_x * g.tiles_y);
    const uint32_t n_pending = min(b.pending_total[view * kPendingTotalStride], b.pending_stride);
    const uint32_t n_chunks = (n_pending + kBlockTris - 1u) / kBlockTris;
    uint32_t span_chunks = 1u;
    while (span_chunks < n_chunks) span_chunks <<= 1;
    const uint32_t scramble = ((uint32_t)((float)span_chunks * 0.6180339887f) | 1u);   // odd: a bijection modulo span_chunks
    for (uint32_t turn = blockIdx.x; turn < span_chunks; turn += gridDim.x) {
    const uint32_t blk = (turn * scramble) & (span_chunks - 1u);
    if (blk >= n_chunks) continue;   // uniform over the block
    const uint32_t n = min(kBlockTris, n_pending - blk * kBlockTris);
    __syncthreads();   // the tables of the previous chunk are no longer in use
    uint32_t* count_small = b.count_small + (size_t)view * n_tiles;
    uint32_t* count_large = b.count_large + (size_t)view * n_tiles;
    for (uint32_t e = tid; e < kHashSize / 4u; e += 256u) {   // (any table size: no trip count that rounds down to nothing)
        reinterpret_cast<uint4*>(s_key)[e] = make_uint4(kHashEmpty, kHashEmpty, kHashEmpty, kHashEmpty);
        reinterpret_cast<uint4*>(s_cnt)[e] = make_uint4(0u, 0u, 0u, 0u);
    }
    if (tid == 0) s_total = 0u;

    const uint32_t* slice = b.pending + (size_t)view * b.pending_stride + (size_t)blk * kBlockTris;
    const int2* xy = sp.xy + (size_t)view * sp.stride;
    uint32_t tri[kTrisPerThread], i0[kTrisPerThread], i1[kTrisPerThread], i2[kTrisPerThread];
    bool have[kTrisPerThread];
#pragma unroll
    for (uint32_t k = 0; k < kTrisPerThread; ++k) {
        have[k] = k * 256u + tid < n;
        tri[k] = have[k] ? slice[k * 256u + tid] : 0u;
    }
#pragma unroll
    for (uint32_t k = 0; k < kTrisPerThread; ++k) {
        i0[k] = i1[k] = i2[k] = 0u;
        if (have[k]) { i0[k] = idx[3 * (size_t)tri[k]]; i1[k] = idx[3 * (size_t)tri[k] + 1]; i2[k] = idx[3 * (size_t)tri[k] + 2]; }
    }
    int2 r0[kTrisPerThread], r1[kTrisPerThread], r2[kTrisPerThread];
#pragma unroll
    for (uint32_t k = 0; k < kTrisPerThread; ++k) {
        r0[k] = r1[k] = r2[k] = make_int2(0, 0);
        if (have[k]) { r0[k] = xy[i0[k]]; r1[k] = xy[i1[k]]; r2[k] = xy[i2[k]]; }
    }
    __syncthreads();   // hash table initialised

    // per triangle: tile span; for small triangles the hash slot and the rank inside the slot's group
    // of every covered tile
    uint32_t nrec[kTrisPerThread], span[kTrisPerThread];
    uint32_t slot[kTrisPerThread][kMaxSmallTiles], local[kTrisPerThread][kMaxSmallTiles];
    uint32_t is_large = 0;
#pragma unroll
    for (uint32_t k = 0; k < kTrisPerThread; ++k) {
        nrec[k] = 0; span[k] = 0;
        int tx0 = 0, ty0 = 0, nx = 0, ny = 0;
        bool live = false;
        if (have[k]) {   // set-up already established: projectable, front-facing, non-empty pixel span
            int x0, x1, y0, y1;
            pixel_span<S>(min(r0[k].x, min(r1[k].x, r2[k].x)), max(r0[k].x, max(r1[k].x, r2[k].x)), x0, x1);
            pixel_span<S>(min(r0[k].y, min(r1[k].y, r2[k].y)), max(r0[k].y, max(r1[k].y, r2[k].y)), y0, y1);
            x0 = max(x0, g.rx); x1 = min(x1, g.rx + g.rw - 1);
            y0 = max(y0, g.ry); y1 = min(y1, g.ry + g.rh - 1);
            if (x0 <= x1 && y0 <= y1) {
                live = true;
                tx0 = (x0 - g.rx) >> g.tile_log2; ty0 = (y0 - g.ry) >> g.tile_log2;
                const int tx1 = (x1 - g.rx) >> g.tile_log2, ty1 = (y1 - g.ry) >> g.tile_log2;
                span[k] = (uint32_t)tx0 | ((uint32_t)ty0 << 8) | ((uint32_t)tx1 << 16) | ((uint32_t)ty1 << 24);
                nx = tx1 - tx0 + 1; ny = ty1 - ty0 + 1;
            }
        }
        bool small = live && nx <= 2 && ny <= 2;
        // first covered tile of every small triangle: lanes of the wave that share a tile are
        // counted with ballots, one lane per distinct tile touches the block's table
        {
            const uint32_t tile0 = (uint32_t)(ty0 * g.tiles_x + tx0);
            slot[k][0] = kHashSize; local[k][0] = 0;
            bool pend = small;
            uint64_t m = __ballot(pend);
            while (m) {
                const int leader = __ffsll((long long)m) - 1;
                const uint32_t lt = (uint32_t)__shfl((int)tile0, leader);
                const uint64_t same = __ballot(pend && tile0 == lt);
                uint32_t sl = kHashSize, base = 0;
                if (lane == (uint32_t)leader) {
                    sl = hash_slot(s_key, lt);
                    if (sl < kHashSize) base = atomicAdd(&s_cnt[sl], (uint32_t)__popcll(same));
                }
                sl = (uint32_t)__shfl((int)sl, leader); base = (uint32_t)__shfl((int)base, leader);
                if (pend && tile0 == lt) {
                    slot[k][0] = sl; local[k][0] = base + (uint32_t)__popcll(same & lanes_below);
                    if (sl == kHashSize) small = false;   // table too crowded: handle as a large triangle
                    pend = false;
                }
                m &= ~same;
            }
        }
        if (small) {
            nrec[k] = (uint32_t)(nx * ny);
#pragma unroll
            for (uint32_t j = 1; j < kMaxSmallTiles; ++j) {   // further tiles of triangles that straddle a tile border
                slot[k][j] = kHashSize; local[k][j] = 0;
                if (j < nrec[k]) {
                    const uint32_t tile = (uint32_t)((ty0 + (int)(j / (uint32_t)nx)) * g.tiles_x + tx0 + (int)(j % (uint32_t)nx));
                    slot[k][j] = hash_slot(s_key, tile);
                    if (slot[k][j] < kHashSize) local[k][j] = atomicAdd(&s_cnt[slot[k][j]], 1u);
                    else atomicAdd(&count_large[tile], 1u);   // crowded table: this (triangle, tile) pair takes the slow path
                }
            }
        } else if (live) {
            // large triangle, or a small one whose first tile found no room in the table: one record
            // with the whole tile span, appended by the fill pass
            nrec[k] = 1;
            is_large |= 1u << k;
            const int tx1 = (int)((span[k] >> 16) & 255u), ty1 = (int)(span[k] >> 24);
            for (int ty = ty0; ty <= ty1; ++ty)
                for (int tx = tx0; tx <= tx1; ++tx) atomicAdd(&count_large[ty * g.tiles_x + tx], 1u);
        }
    }
    // place of this lane's records inside the block's segment (any order will do): wave prefix sum
    uint32_t first[kTrisPerThread];
    {
        uint32_t mine = 0;
#pragma unroll
        for (uint32_t k = 0; k < kTrisPerThread; ++k) mine += nrec[k];
        uint32_t incl = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t v = (uint32_t)__shfl_up((int)incl, o); if (lane >= (uint32_t)o) incl += v; }
        uint32_t wave_base = 0;
        if (lane == 63u) wave_base = atomicAdd(&s_total, incl);
        wave_base = (uint32_t)__shfl((int)wave_base, 63);
        uint32_t run = wave_base + incl - mine;
#pragma unroll
        for (uint32_t k = 0; k < kTrisPerThread; ++k) { first[k] = run; run += nrec[k]; }
    }
    __syncthreads();
    // one returning global atomic per distinct tile of the block reserves the group's place in the
    // tile's list; the slot then holds the group's first position
    for (uint32_t e = tid; e < kHashSize; e += 256u) {
        const uint32_t c = s_cnt[e];
        if (c) s_cnt[e] = atomicAdd(&count_small[s_key[e]], c);
    }
    __syncthreads();
    BinRecord* seg = b.records + ((size_t)view * b.n_blocks + blk) * kBlockRecords;
#pragma unroll
    for (uint32_t k = 0; k < kTrisPerThread; ++k) {
        if (nrec[k] == 0) continue;
        if (is_large & (1u << k)) {
            BinRecord rec; rec.tri = tri[k]; rec.where = span[k]; rec.rel = kLargeTri;
            seg[first[k]] = rec;
        } else {
            const uint32_t tx0 = span[k] & 255u, ty0 = (span[k] >> 8) & 255u, nx = ((span[k] >> 16) & 255u) - tx0 + 1u;
#pragma unroll
            for (uint32_t j = 0; j < kMaxSmallTiles; ++j)
                if (j < nrec[k]) {
                    BinRecord rec; rec.tri = tri[k];
                    if (slot[k][j] < kHashSize) { rec.where = s_key[slot[k][j]]; rec.rel = s_cnt[slot[k][j]] + local[k][j]; }
                    else { const uint32_t tx = tx0 + j % nx, ty = ty0 + j / nx; rec.where = tx | (ty << 8) | (tx << 16) | (ty << 24); rec.rel = kLargeTri; }
                    seg[first[k] + j] = rec;
                }
        }
    }
    if (tid == 0) b.block_count[(size_t)view * b.n_blocks + blk] = s_total;
    }
}

// =================================================================================================
// stage 2b: exclusive scan of the per-tile counts (all views of the pass at once)
// =================================================================================================
__global__ __launch_bounds__(1024) void scan_kernel(const uint32_t* __restrict__ count_a, const uint32_t* __restrict__ count_b, uint32_t n,
                                                    uint32_t* __restrict__ offset, uint32_t* __restrict__ counters,
                                                    const uint32_t* __restrict__ pending_total, uint32_t n_views,
                                                    const uint32_t* __restrict__ micro_spread,
                                                    uint32_t* __restrict__ host_counters)
{
    // chunks of kScanChunk counts: loaded side by side into LDS (one word of padding per 32, so that work-items reading eight consecutive
    // counts each do not meet in a bank), eight consecutive ones summed by each work-item, the sums scanned inside every wave with
    // shuffles and the sixteen wave totals by every work-item for itself, offsets written side by side; the running total carries over
    // to the next chunk.  Four barriers per chunk (a scan of the 1024 sums through LDS took twenty-three: 64 views of 1080p in 32 x 32
    // tiles, 130 560 counts, 0.110 -> 0.03 ms).
    constexpr uint32_t kPer = 8, kScanChunk = 1024u * kPer;
    __shared__ uint32_t s_cnt[kScanChunk + kScanChunk / 32u];
    __shared__ uint32_t s_wave_total[16];
    __shared__ uint32_t s_micro_total;
    if (threadIdx.x == 0) s_micro_total = 0u;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    // One workgroup walks the chunks one after the other -- or, launched as one workgroup per chunk, each first adds up everything
    // before its chunk for itself (reads that stay in the L2; a chunk's turn in the walk costs two trips to memory, 6 us, and sixteen
    // chunks 0.1 ms, while the sixteen sums run side by side).  The launcher takes the second form for 2..256 chunks.
    const bool one_chunk_each = gridDim.x > 1u;
    uint32_t carry = 0;
    if (one_chunk_each) {
        const uint32_t before = blockIdx.x * kScanChunk;
        uint32_t acc = 0;   // (`before` is a multiple of 8192: eight loads from each array in flight per work-item)
        for (uint32_t i0 = tid; i0 < before; i0 += 8192u) {
            uint32_t a[8], c[8];
#pragma unroll
            for (uint32_t k = 0; k < 8u; ++k) { a[k] = count_a[i0 + k * 1024u]; c[k] = count_b[i0 + k * 1024u]; }
#pragma unroll
            for (uint32_t k = 0; k < 8u; ++k) acc += a[k] + c[k];
        }
        for (int o = 32; o > 0; o >>= 1) acc += (uint32_t)__shfl_down((int)acc, o);
        if (lane == 0u) s_wave_total[wave] = acc;
        __syncthreads();
#pragma unroll
        for (uint32_t w = 0; w < 16u; ++w) carry += s_wave_total[w];
    }
    const bool reports = !one_chunk_each || blockIdx.x == gridDim.x - 1u;   // the workgroup that ends up with the grand total
    for (uint32_t base = one_chunk_each ? blockIdx.x * kScanChunk : 0u; base < n; base += one_chunk_each ? n : kScanChunk) {
        const uint32_t m = min(kScanChunk, n - base);
        __syncthreads();   // the previous chunk's tables (or the sums above) are no longer in use
        for (uint32_t i = tid; i < kScanChunk; i += 1024u) s_cnt[i + (i >> 5)] = i < m ? count_a[base + i] + count_b[base + i] : 0u;
        __syncthreads();
        uint32_t s = 0;
#pragma unroll
        for (uint32_t k = 0; k < kPer; ++k) { const uint32_t i = tid * kPer + k; s += s_cnt[i + (i >> 5)]; }
        uint32_t inclusive = s;   // of the wave's sums up to this lane
        for (uint32_t o = 1; o < 64u; o <<= 1) {
            const uint32_t v = (uint32_t)__shfl_up((int)inclusive, o);
            if (lane >= o) inclusive += v;
        }
        if (lane == 63u) s_wave_total[wave] = inclusive;
        __syncthreads();
        uint32_t before = 0, total = 0;
#pragma unroll
        for (uint32_t w = 0; w < 16u; ++w) { const uint32_t t = s_wave_total[w]; before += w < wave ? t : 0u; total += t; }
        uint32_t run = carry + before + inclusive - s;
#pragma unroll
        for (uint32_t k = 0; k < kPer; ++k) { const uint32_t i = tid * kPer + k; const uint32_t c = s_cnt[i + (i >> 5)]; s_cnt[i + (i >> 5)] = run; run += c; }
        carry += total;
        __syncthreads();
        for (uint32_t i = tid; i < m; i += 1024u) offset[base + i] = s_cnt[i + (i >> 5)];
    }
    if (!reports) return;
    if (tid == 1023u) { offset[n] = carry; counters[kCtrRefs] = carry; }
    // micro-triangles resolved by the set-up blocks (instrumentation)
    uint32_t micro = tid < kMicroSpread ? micro_spread[tid * kMicroSpreadStride] : 0u;
    for (int o = 32; o > 0; o >>= 1) micro += (uint32_t)__shfl_down((int)micro, o);
    if ((tid & 63u) == 0u && micro) atomicAdd(&s_micro_total, micro);
    __syncthreads();
    if (tid == 0) {
        // results of the pass for the host, written straight into pinned host memory (read after the
        // stream has been synchronised): total references, -, unprojectable, binned, direct, -, clipped
        uint32_t binned = 0;
        for (uint32_t v = 0; v < n_views; ++v) binned += pending_total[v * kPendingTotalStride];
        host_counters[0] = carry; host_counters[1] = counters[kCtrDirectTooBig]; host_counters[2] = counters[kCtrUnprojectable];
        host_counters[3] = binned; host_counters[4] = s_micro_total; host_counters[5] = counters[kCtrClipListed];
        host_counters[6] = counters[kCtrClipped];
    }
}

// =================================================================================================
// stage 2c: scatter the records into per-tile reference lists.  Small triangles already know their
// place (tile offset + position reserved by set-up): a plain scatter.  Large triangles are appended
// behind them, one atomic per covered tile.
// =================================================================================================
__global__ __launch_bounds__(256) void fill_kernel(FrameGeom g, BinBuffers b)
{
    const uint32_t view = blockIdx.y, tid = threadIdx.x;
    const uint32_t n_tiles = (uint32_t)(g.tiles_x * g.tiles_y);
    const uint32_t* offset = b.tile_offset + (size_t)view * n_tiles;
    const uint32_t* count_small = b.count_small + (size_t)view * n_tiles;
    uint32_t* cursor_large = b.cursor_large + (size_t)view * n_tiles;
    const uint32_t n_chunks = (min(b.pending_total[view * kPendingTotalStride], b.pending_stride) + kBlockTris - 1u) / kBlockTris;
    for (uint32_t chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
        const uint32_t n = min(b.block_count[(size_t)view * b.n_blocks + chunk], kBlockRecords);
        const BinRecord* seg = b.records + ((size_t)view * b.n_blocks + chunk) * kBlockRecords;
        for (uint32_t i = tid; i < n; i += 256u) {
            const BinRecord rec = seg[i];
            if (rec.rel != kLargeTri) {
                const uint32_t pos = offset[rec.where] + rec.rel;
                if (pos < b.ref_capacity) b.refs[pos] = rec.tri; else atomicAdd(&b.counters[kCtrRefsDropped], 1u);
            } else {
                const int tx0 = rec.where & 255, ty0 = (rec.where >> 8) & 255, tx1 = (rec.where >> 16) & 255, ty1 = rec.where >> 24;
                for (int ty = ty0; ty <= ty1; ++ty)
                    for (int tx = tx0; tx <= tx1; ++tx) {
                        const uint32_t tile = (uint32_t)(ty * g.tiles_x + tx);
                        const uint32_t pos = offset[tile] + count_small[tile] + atomicAdd(&cursor_large[tile], 1u);
                        if (pos < b.ref_capacity) b.refs[pos] = rec.tri; else atomicAdd(&b.counters[kCtrRefsDropped], 1u);
                    }
            }
        }
    }
}

// =================================================================================================
// stage 3: per-tile rasterisation.  One workgroup per (tile, view): the tile's visibility words live
// in LDS, every covered sample does a 64-bit LDS atomic min on (depth bits << 32 | triangle id), the
// finished tile is stored to HBM with plain coalesced stores (clear value included, so there is no
// separate clear pass).  Small triangles: one lane each.  Large ones: the wave takes them in turn
// and scans the bounding box in 8x8-pixel blocks.
// =================================================================================================
struct LaneTri {
    TriSetup ts;
    int x0, x1, y0, y1;
    uint32_t tri;
};

// one sample s of pixel (x, y): coverage by the 64-bit edge functions, depth, 64-bit minimum into `word`
template <int S>
__device__ inline bool raster_sample(const TriSetup& ts, uint32_t tri, int x, int y, int s, unsigned long long* word)   // true: a depth went into `word`
{
    const int64_t px = (int64_t)x * kSubpix + sample_dx(S, s), py = (int64_t)y * kSubpix + sample_dy(S, s);
    int64_t Ea, Eb, Ec;
    edge_eval(ts, px, py, Ea, Eb, Ec);
    if (((Ea + ts.bias_a) | (Eb + ts.bias_b) | (Ec + ts.bias_c)) < 0) return false;
    uint64_t key;
    if (!depth_key(ts, Ea, Eb, Ec, tri, key)) return false;
    atomicMin(word, (unsigned long long)key);
    return true;
}
template <int S>
__device__ inline void raster_pixel(const TriSetup& ts, uint32_t tri, int x, int y, unsigned long long* cell /* S words */)
{
#pragma unroll
    for (int s = 0; s < S; ++s) (void)raster_sample<S>(ts, tri, x, y, s, cell + s);
}

// Same arithmetic in 32 bits, valid when the triangle's bounding box is under 64 pixels on both axes:
// every factor is then below 2^15 + 2^9 and every edge function below 2^31, so the integers (and the
// floats converted from them) are the same as in the 64-bit form.
__device__ inline bool fits_32bit(const TriSetup& ts) { return (ts.maxX - ts.minX) < 16384 && (ts.maxY - ts.minY) < 16384; }

template <int S>
__device__ inline bool raster_sample32(const TriSetup& ts, uint32_t tri, int x, int y, int s, unsigned long long* word)   // true: a depth went into `word`
{
    const int32_t px = x * kSubpix + sample_dx(S, s), py = y * kSubpix + sample_dy(S, s);
    const int32_t Ea = mul_small(ts.cx - ts.bx, py - ts.by) - mul_small(ts.cy - ts.by, px - ts.bx);
    const int32_t Eb = mul_small(ts.ax - ts.cx, py - ts.cy) - mul_small(ts.ay - ts.cy, px - ts.cx);
    const int32_t Ec = mul_small(ts.bx - ts.ax, py - ts.ay) - mul_small(ts.by - ts.ay, px - ts.ax);
    if (((Ea + ts.bias_a) | (Eb + ts.bias_b) | (Ec + ts.bias_c)) < 0) return false;
    const float l0 = (float)Ea * ts.rA, l2 = (float)Eb * ts.rA, l1 = (float)Ec * ts.rA;
    float z = fmaf(l2, ts.z2, fmaf(l1, ts.z1, l0 * ts.z0));
    z = z + 0.0f;
    if (!(z >= 0.0f && z < 1.0f)) return false;
    atomicMin(word, ((unsigned long long)__float_as_uint(z) << 32) | (unsigned long long)tri);
    return true;
}
template <int S>
__device__ inline void raster_pixel32(const TriSetup& ts, uint32_t tri, int x, int y, unsigned long long* cell /* S words */)
{
#pragma unroll
    for (int s = 0; s < S; ++s) (void)raster_sample32<S>(ts, tri, x, y, s, cell + s);
}

template <int TL2, int S, int TPB>
__global__ __launch_bounds__(TPB) void raster_kernel(const uint32_t* __restrict__ idx, uint32_t n_tris, ScreenPlanes sp,
                                                     FrameGeom g, const uint32_t* __restrict__ tile_offset, const uint32_t* __restrict__ refs,
                                                     uint32_t ref_capacity, unsigned long long* __restrict__ vis, BlockFlags flags, uint32_t view0)
{
    constexpr int T = 1 << TL2;
    constexpr int kSmall = 16;   // bounding boxes up to this many pixels are walked by one lane
    __shared__ unsigned long long tile[T * T * S];
    const uint32_t tile_id = blockIdx.x, view = blockIdx.y;
    const uint32_t n_tiles = (uint32_t)(g.tiles_x * g.tiles_y);
    const int tx = (int)(tile_id % (uint32_t)g.tiles_x), ty = (int)(tile_id / (uint32_t)g.tiles_x);
    const int px0 = g.rx + tx * T, py0 = g.ry + ty * T;                     // absolute pixel origin of the tile
    const int px1 = min(px0 + T, g.rx + g.rw) - 1, py1 = min(py0 + T, g.ry + g.rh) - 1;
    const uint32_t tid = threadIdx.x, lane = lane_id();

    const uint32_t beg = tile_offset[(size_t)view * n_tiles + tile_id];
    uint32_t end = tile_offset[(size_t)view * n_tiles + tile_id + 1];
    end = min(end, ref_capacity);
    if (beg >= end) return;   // nothing binned here: the visibility buffer was cleared before set-up (uniform over the block)

    for (int i = tid; i < T * T * S; i += TPB) tile[i] = kVisClear;
    __syncthreads();
    const int2* xy = sp.xy + (size_t)view * sp.stride;
    const float* zp = sp.z + (size_t)view * sp.stride;

    // Software pipeline over the tile's list: reference -> indices -> vertices are three dependent
    // loads, so while one triangle is rasterised the vertices of the next one, the indices of the one
    // after and the reference of the third are already in flight.
    struct Fetched { int2 a0, a1, a2; float z0, z1, z2; };
    auto load_ref = [&](uint32_t r) -> uint32_t {
        uint32_t t = kNoTri;
        if (r < end) { t = refs[r]; if (t >= n_tris) t = kNoTri; }   // (lists only ever hold valid ids; cheap insurance against a fault)
        return t;
    };
    auto load_idx = [&](uint32_t t, uint32_t& i0, uint32_t& i1, uint32_t& i2) {
        i0 = i1 = i2 = 0;
        if (t != kNoTri) { i0 = idx[3 * (size_t)t]; i1 = idx[3 * (size_t)t + 1]; i2 = idx[3 * (size_t)t + 2]; }
    };
    auto load_verts = [&](uint32_t t, uint32_t i0, uint32_t i1, uint32_t i2) -> Fetched {
        Fetched f{};
        if (t != kNoTri) {
            f.a0 = xy[i0]; f.a1 = xy[i1]; f.a2 = xy[i2];
            f.z0 = zp[i0]; f.z1 = zp[i1]; f.z2 = zp[i2];
        }
        return f;
    };
    // Work-item -> position in the list: consecutive references go to different waves (reference r of a turn to lane
    // r / waves of wave r % waves).  A tile of a scene of large triangles holds a few dozen references: dealt out in
    // list order they would all land in the first wave, which scans one large triangle at a time while the others idle.
    constexpr uint32_t kWaves = TPB / 64u;
    const uint32_t slot = (tid & 63u) * kWaves + (tid >> 6);
    uint32_t tri_cur = load_ref(beg + slot), tri_n1 = load_ref(beg + TPB + slot), tri_n2 = load_ref(beg + 2 * TPB + slot);
    uint32_t c0, c1, c2, n0, n1, n2;
    load_idx(tri_cur, c0, c1, c2);
    load_idx(tri_n1, n0, n1, n2);
    Fetched f_cur = load_verts(tri_cur, c0, c1, c2);

    for (uint32_t base = beg; base < end; base += TPB) {
        // next stages first, so that their latency overlaps this iteration's work
        const Fetched f_n1 = load_verts(tri_n1, n0, n1, n2);
        uint32_t m0, m1, m2;
        load_idx(tri_n2, m0, m1, m2);
        const uint32_t tri_n3 = load_ref(base + 3 * TPB + slot);

        bool have = tri_cur != kNoTri;
        LaneTri L;
        L.tri = tri_cur;
        int npix = 0;
        if (have) {
            ScreenVertex v0, v1, v2;
            v0.X = f_cur.a0.x; v0.Y = f_cur.a0.y; v0.z = f_cur.z0; v0.rw = f_cur.a0.x == kBadVertexX ? 0.0f : 1.0f;
            v1.X = f_cur.a1.x; v1.Y = f_cur.a1.y; v1.z = f_cur.z1; v1.rw = f_cur.a1.x == kBadVertexX ? 0.0f : 1.0f;
            v2.X = f_cur.a2.x; v2.Y = f_cur.a2.y; v2.z = f_cur.z2; v2.rw = f_cur.a2.x == kBadVertexX ? 0.0f : 1.0f;
            have = setup_triangle(v0, v1, v2, L.ts) == 1;
            if (have) {
                pixel_span<S>(L.ts.minX, L.ts.maxX, L.x0, L.x1);
                pixel_span<S>(L.ts.minY, L.ts.maxY, L.y0, L.y1);
                L.x0 = max(L.x0, px0); L.x1 = min(L.x1, px1);
                L.y0 = max(L.y0, py0); L.y1 = min(L.y1, py1);
                have = L.x0 <= L.x1 && L.y0 <= L.y1;
                if (have) npix = (L.x1 - L.x0 + 1) * (L.y1 - L.y0 + 1);
            }
        }
        tri_cur = tri_n1; f_cur = f_n1;
        tri_n1 = tri_n2; n0 = m0; n1 = m1; n2 = m2;
        tri_n2 = tri_n3;
        if (have && npix <= kSmall) {
            if (fits_32bit(L.ts)) {
                for (int y = L.y0; y <= L.y1; ++y)
                    for (int x = L.x0; x <= L.x1; ++x)
                        raster_pixel32<S>(L.ts, L.tri, x, y, &tile[((y - py0) * T + (x - px0)) * S]);
            } else {
                for (int y = L.y0; y <= L.y1; ++y)
                    for (int x = L.x0; x <= L.x1; ++x)
                        raster_pixel<S>(L.ts, L.tri, x, y, &tile[((y - py0) * T + (x - px0)) * S]);
            }
        }
        uint64_t big = __ballot(have && npix > kSmall);
        while (big) {
            const int src = __ffsll((long long)big) - 1;
            big &= big - 1;
            TriSetup ts;
            ts.ax = __shfl(L.ts.ax, src); ts.ay = __shfl(L.ts.ay, src);
            ts.bx = __shfl(L.ts.bx, src); ts.by = __shfl(L.ts.by, src);
            ts.cx = __shfl(L.ts.cx, src); ts.cy = __shfl(L.ts.cy, src);
            ts.bias_a = __shfl(L.ts.bias_a, src); ts.bias_b = __shfl(L.ts.bias_b, src); ts.bias_c = __shfl(L.ts.bias_c, src);
            ts.z0 = __shfl(L.ts.z0, src); ts.z1 = __shfl(L.ts.z1, src); ts.z2 = __shfl(L.ts.z2, src);
            ts.rA = __shfl(L.ts.rA, src);
            const int x0 = __shfl(L.x0, src), x1 = __shfl(L.x1, src), y0 = __shfl(L.y0, src), y1 = __shfl(L.y1, src);
            const uint32_t tri = (uint32_t)__shfl((int)L.tri, src);
            const bool narrow = __shfl((int)fits_32bit(L.ts), src) != 0;
            const int lx = (int)(lane & 7u), ly = (int)(lane >> 3);
            for (int by = y0; by <= y1; by += 8)
                for (int bx = x0; bx <= x1; bx += 8) {
                    const int x = bx + lx, y = by + ly;
                    if (x <= x1 && y <= y1) {
                        if (narrow) raster_pixel32<S>(ts, tri, x, y, &tile[((y - py0) * T + (x - px0)) * S]);
                        else raster_pixel<S>(ts, tri, x, y, &tile[((y - py0) * T + (x - px0)) * S]);
                    }
                }
        }
    }
    __syncthreads();

    const int tw = px1 - px0 + 1, th = py1 - py0 + 1;
    unsigned long long* out = vis + (size_t)(view0 + view) * g.rw * g.rh * S;
    for (int i = tid; i < T * T * S; i += TPB) {
        const int s = i % S, p = i / S, lx = p & (T - 1), ly = p >> TL2;
        // merge with what micro-triangles already resolved in HBM; untouched samples cost nothing
        const bool covered = lx < tw && ly < th && tile[i] != kVisClear;
        if (covered) atomicMin(&out[((size_t)(py0 - g.ry + ly) * g.rw + (px0 - g.rx + lx)) * S + s], tile[i]);
        // the flag of the block: stored by the first lane of every run of covered samples of one block (a wave's 64 samples lie in
        // one row of the tile, in one or two blocks)
        const int bid = ((py0 - g.ry + ly) / (int)kFlagBlockH) * flags.nbx + (px0 - g.rx + lx) / (int)kFlagBlockW;
        const int prev_bid = __shfl_up(covered ? bid : -1, 1);
        if (covered && (lane == 0u || prev_bid != bid)) flags.flags[(size_t)(view0 + view) * flags.nbx * flags.nby + bid] = 1u;
    }
}

// =================================================================================================
// stage 3, small frames: the pending triangles resolved straight into the visibility buffer
// =================================================================================================
// The tile path (binning, scan, fill, one workgroup per tile) is four dependent launches, each a chain of dependent loads: some 90 us
// on the reference's own 800 x 600 frame of 15 000 triangles, whatever little they have to do.  When a target's previous draw left
// few triangles pending, the next one resolves them with this kernel instead: one wave per pending triangle, the clipped box walked
// in steps of 64 samples (8 x 8 pixels at one sample, 8 x 2 pixels x 4 samples at four: neighbouring lanes are neighbouring words of
// the visibility buffer, so a step's atomic minimums touch few lines), the same edge functions, depth expression and 64-bit minimum
// as every other path.  A triangle whose box is larger than kDirectMaxBoxPixels is left alone and counted: the host then repeats the
// draw with the tile path before anything of it can be observed (a wave would walk a screen-filling triangle for a millisecond).
constexpr int kDirectMaxBoxPixels = 64 * 64;
template <int S>
__global__ __launch_bounds__(256) void direct_raster_kernel(const uint32_t* __restrict__ idx, uint32_t n_tris, ScreenPlanes sp, FrameGeom g, BinBuffers b,
                                                            unsigned long long* __restrict__ vis, BlockFlags flags, uint32_t view0)
{
    const uint32_t view = blockIdx.y, lane = lane_id();
    const uint32_t n_pending = min(b.pending_total[view * kPendingTotalStride], b.pending_stride);
    const uint32_t* list = b.pending + (size_t)view * b.pending_stride;
    const int2* xy = sp.xy + (size_t)view * sp.stride;
    const float* zp = sp.z + (size_t)view * sp.stride;
    unsigned long long* vis_view = vis + (size_t)(view0 + view) * g.rw * g.rh * S;
    constexpr int kStepH = S == 1 ? 8 : 2;                         // a step is 8 pixels wide and kStepH tall
    const int lx = S == 1 ? (int)(lane & 7u) : (int)((lane >> 2) & 7u), ly = S == 1 ? (int)(lane >> 3) : (int)(lane >> 5), ls = S == 1 ? 0 : (int)(lane & 3u);
    uint32_t too_big = 0;
    for (uint32_t i = blockIdx.x * 4u + (threadIdx.x >> 6); i < n_pending; i += gridDim.x * 4u) {     // the same triangle for the whole wave
        const uint32_t tri = list[i];
        if (tri >= n_tris) continue;   // (the list only ever holds valid ids; cheap insurance against a fault)
        const uint32_t i0 = idx[3 * (size_t)tri], i1 = idx[3 * (size_t)tri + 1], i2 = idx[3 * (size_t)tri + 2];
        const int2 a0 = xy[i0], a1 = xy[i1], a2 = xy[i2];
        ScreenVertex v0, v1, v2;
        v0.X = a0.x; v0.Y = a0.y; v0.z = zp[i0]; v0.rw = a0.x == kBadVertexX ? 0.0f : 1.0f;
        v1.X = a1.x; v1.Y = a1.y; v1.z = zp[i1]; v1.rw = a1.x == kBadVertexX ? 0.0f : 1.0f;
        v2.X = a2.x; v2.Y = a2.y; v2.z = zp[i2]; v2.rw = a2.x == kBadVertexX ? 0.0f : 1.0f;
        TriSetup ts;
        if (setup_triangle(v0, v1, v2, ts) != 1) continue;          // (set-up established: projectable, front-facing)
        int x0, x1, y0, y1;
        pixel_span<S>(ts.minX, ts.maxX, x0, x1);
        pixel_span<S>(ts.minY, ts.maxY, y0, y1);
        x0 = max(x0, g.rx); x1 = min(x1, g.rx + g.rw - 1);
        y0 = max(y0, g.ry); y1 = min(y1, g.ry + g.rh - 1);
        if (x0 > x1 || y0 > y1) continue;
        if ((x1 - x0 + 1) > kDirectMaxBoxPixels / (y1 - y0 + 1)) { ++too_big; continue; }   // (the same for every lane of the wave)
        const bool narrow = fits_32bit(ts);
        for (int by = y0; by <= y1; by += kStepH)
            for (int bx = x0; bx <= x1; bx += 8) {
                const int x = bx + lx, y = by + ly;
                bool wrote = false;
                if (x <= x1 && y <= y1) {
                    unsigned long long* word = &vis_view[((size_t)(y - g.ry) * g.rw + (x - g.rx)) * S + ls];
                    wrote = narrow ? raster_sample32<S>(ts, tri, x, y, ls, word) : raster_sample<S>(ts, tri, x, y, ls, word);
                }
                // a step that wrote anything: the blocks under its four corners (all the blocks it reaches into) may hold something now
                // (every lane of the wave is here: the triangle and the loop bounds are the wave's)
                if (__ballot(wrote) != 0ull && lane < 4u)
                    flags.of(view0 + view, min(bx + 7 * (int)(lane & 1u), x1) - g.rx, min(by + (kStepH - 1) * (int)(lane >> 1), y1) - g.ry) = 1u;
            }
    }
    if (lane == 0u && too_big) atomicAdd(&b.counters[kCtrDirectTooBig], too_big);
}

// =================================================================================================
// near-plane clipping (see sr_device.h)
// =================================================================================================
// One kernel cuts the listed triangles and resolves the pieces into the visibility buffer in HBM, 8x8 pixels per step.
// A work item is (listed triangle, slab): the rows of 8x8 blocks of a piece's box are dealt out to kClipSlabs waves, so
// that a piece that fills the frame does not hang on one wave; every lane of a wave computes the same polygon (a few
// hundred instructions, nothing next to the steps of a large piece).  Positions come from the three planes px, py, pz
// and are projected with the vertex kernel's own function, so a projectable vertex gets the window position its
// neighbours were rasterised with.  The list is empty in any ordinary frame: every wave then leaves at once.
constexpr uint32_t kClipSlabs = 16;
template <int S>
__global__ __launch_bounds__(256) void near_clip_kernel(const uint32_t* __restrict__ idx, uint32_t n_tris,
                                                        const float* __restrict__ px, const float* __restrict__ py, const float* __restrict__ pz,
                                                        const ViewXform* __restrict__ views, FrameGeom g, ClipList clip, uint32_t* __restrict__ counters,
                                                        unsigned long long* __restrict__ vis, BlockFlags flags, uint32_t view0)
{
    const uint32_t n = min(counters[kCtrClipListed], clip.capacity);
    const uint32_t lane = lane_id();
    const uint32_t waves_total = gridDim.x * 4u, first = blockIdx.x * 4u + (threadIdx.x >> 6);
    const int lx = (int)(lane & 7u), ly = (int)(lane >> 3);
    uint32_t n_clipped = 0, n_dropped = 0;   // counted by the wave that holds a triangle's slab 0
    for (uint32_t item = first; item < n * kClipSlabs; item += waves_total) {
        const uint2 entry = clip.entries[item / kClipSlabs];
        const uint32_t slab = item % kClipSlabs, view = entry.x, tri = entry.y;
        if (tri >= n_tris) continue;   // (the list only ever holds valid ids; cheap insurance against a fault)
        float c[3][4];
        ScreenVertex sv[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const uint32_t i = idx[3 * (size_t)tri + k];
            sv[k] = project_vertex(views[view], px[i], py[i], pz[i], g.W, g.H, c[k]);
        }
        ScreenVertex poly[kClipMaxVertices];
        const int np = near_clip_polygon(c, sv, g.W, g.H, poly);
        if (np < 0) { if (slab == 0u) ++n_dropped; continue; }   // cannot be clipped: dropped and counted
        if (np == 0) continue;
        if (slab == 0u) ++n_clipped;
        unsigned long long* vis_view = vis + (size_t)(view0 + view) * g.rw * g.rh * S;
        for (int k = 2; k < np; ++k) {    // fan around the first polygon vertex
            ScreenVertex v0 = poly[0], v1 = poly[k - 1], v2 = poly[k];
            v0.rw = v1.rw = v2.rw = 1.0f;
            TriSetup ts;
            if (setup_triangle(v0, v1, v2, ts) != 1) continue;
            int x0, x1, y0, y1;
            pixel_span<S>(ts.minX, ts.maxX, x0, x1);
            pixel_span<S>(ts.minY, ts.maxY, y0, y1);
            x0 = max(x0, g.rx); x1 = min(x1, g.rx + g.rw - 1);
            y0 = max(y0, g.ry); y1 = min(y1, g.ry + g.rh - 1);
            const bool narrow = fits_32bit(ts);
            for (int by = y0 + 8 * (int)slab; by <= y1; by += 8 * (int)kClipSlabs)
                for (int bx = x0; bx <= x1; bx += 8) {
                    const int x = bx + lx, y = by + ly;
                    if (x <= x1 && y <= y1) {
                        unsigned long long* cell = &vis_view[((size_t)(y - g.ry) * g.rw + (x - g.rx)) * S];
                        if (narrow) raster_pixel32<S>(ts, tri, x, y, cell);
                        else raster_pixel<S>(ts, tri, x, y, cell);
                    }
                    // the blocks under the four corners of the 8 x 8 step (all the blocks it reaches into) may hold something now
                    if (lane < 4u) flags.of(view0 + view, min(bx + 7 * (int)(lane & 1u), x1) - g.rx, min(by + 7 * (int)(lane >> 1), y1) - g.ry) = 1u;
                }
        }
    }
    if (lane == 0u) {
        if (n_dropped) atomicAdd(&counters[kCtrUnprojectable], n_dropped);
        if (n_clipped) atomicAdd(&counters[kCtrClipped], n_clipped);
    }
}

// =================================================================================================
// stage 4: deferred fragment shading (shaders/shader.frag:102-152) from the visibility buffer
// =================================================================================================
struct ShadeTables {
    const float* decode;   // 256: sRGB code -> linear
    const float* thresh;   // 255: linear value where the code steps from c to c+1
};

// The code of x is the number of thresholds not above it (thresh[c] = the linear value at which the code steps from c to c + 1,
// ascending).  The transfer function, evaluated with the hardware's exp2 (log2_x: log2(x), accurate to a few ulp), gives the code
// to within one; the three thresholds around that estimate then decide exactly -- no loop, no branch.  A NaN or a negative x gives 0.
__device__ inline int srgb_encode8(float x, const float* thresh /* LDS or global, 255 entries */, float log2_x)
{
    x = fmaxf(x, 0.0f);                                     // (a NaN becomes 0)
    const float e = x <= 0.0031308f ? 12.92f * x : 1.055f * __builtin_amdgcn_exp2f(log2_x * (1.0f / 2.4f)) - 0.055f;
    int c = (int)(e * 255.0f + 0.5f);                       // within one of the code (the estimate is good to 1e-5 of a step)
    c = min(max(c, 1), 253);                                // so that c - 1 .. c + 1 are thresholds: the result spans 0 .. 255
    return c - 1 + (x >= thresh[c - 1]) + (x >= thresh[c]) + (x >= thresh[c + 1]);
}
__device__ inline int srgb_encode8(float x, const float* thresh) { return srgb_encode8(x, thresh, __builtin_amdgcn_logf(x)); }
__device__ inline int unorm_encode8(float x)
{
    if (!(x > 0.0f)) return 0;
    if (x > 1.0f) x = 1.0f;
    return (int)rintf(x * 255.0f);
}

__device__ inline int wrap_repeat(int i, int n) { const int m = i % n; return m < 0 ? m + n : m; }

// Texel offsets (in texels from the start of the chain) and weights of one bilinear footprint.  A footprint depends
// on the level's size and the coordinates only, so two maps of the same size sampled at the same place (the
// reference's metallic and roughness maps) share it.
struct Taps {
    uint32_t o00, o10, o01, o11;
    float w00, w10, w01, w11;
};
// SR_COLOUR_MATH: the functions of the shading pass that feed nothing but the colour (allowed 1 LSB of the 8-bit result against the
// CPU oracle) let the compiler fuse multiplies and adds; everything that feeds coverage, depth or triangle ids is compiled unfused
// (-ffp-contract=off), like the oracle.
#define SR_COLOUR_MATH _Pragma("clang fp contract(fast)")
__device__ inline Taps level_taps(const TexDesc& tex, const uint32_t* lvl /* level offsets, LDS */, int level, float s, float tt)
{
    SR_COLOUR_MATH
    const uint32_t base = lvl[level];   // (an LDS copy: a per-lane index into kernel arguments would be a memory load)
    const float lim = 1073741824.0f;
    Taps t;
    float a, b;
    uint32_t r0, r1, i0, i1;   // a chain holds fewer than 2^30 texels (16384^2 * 4/3)
    if (tex.log2_w >= 0) {     // both sizes powers of two (the same for the whole launch): scale by an exponent, wrap with a mask, rows by a shift
        const int sw = max(tex.log2_w - level, 0), sh = max(tex.log2_h - level, 0);
        const float u = ldexpf(s, sw) - 0.5f, v = ldexpf(tt, sh) - 0.5f;              // s * 2^sw, exactly what the product with the level's width gives
        const float flu = floorf(u), flv = floorf(v);
        a = u - flu; b = v - flv;
        const uint32_t mw = (1u << sw) - 1u, mh = (1u << sh) - 1u;
        i0 = (uint32_t)(int)fminf(fmaxf(flu, -lim), lim) & mw;                          // (a NaN becomes -lim)
        const uint32_t j0 = (uint32_t)(int)fminf(fmaxf(flv, -lim), lim) & mh;
        i1 = (i0 + 1u) & mw;
        r0 = base + (j0 << sw); r1 = base + (((j0 + 1u) & mh) << sw);
    } else {
        const int wl = max(tex.w >> level, 1), hl = max(tex.h >> level, 1);
        const float u = s * (float)wl - 0.5f, v = tt * (float)hl - 0.5f;
        const float flu = floorf(u), flv = floorf(v);
        a = u - flu; b = v - flv;
        const int x0 = wrap_repeat((int)fminf(fmaxf(flu, -lim), lim), wl), y0 = wrap_repeat((int)fminf(fmaxf(flv, -lim), lim), hl);
        i0 = (uint32_t)x0; i1 = (uint32_t)wrap_repeat(x0 + 1, wl);
        r0 = base + (uint32_t)y0 * (uint32_t)wl; r1 = base + (uint32_t)wrap_repeat(y0 + 1, hl) * (uint32_t)wl;
    }
    t.o00 = r0 + i0; t.o10 = r0 + i1; t.o01 = r1 + i0; t.o11 = r1 + i1;
    t.w00 = (1.0f - a) * (1.0f - b); t.w10 = a * (1.0f - b); t.w01 = (1.0f - a) * b; t.w11 = a * b;
    return t;
}
// texel at a 32-bit byte offset from a base that is the same for the whole launch (scalar base + per-lane offset)
__device__ inline uint32_t texel_at(const uint32_t* texels, uint32_t index)
{
    return *reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(texels) + (index << 2));
}
// bilinear fetch of NCH leading channels through a footprint
template <int NCH>
__device__ inline void fetch_taps(const uint32_t* texels, const Taps& t, const float* dec, float* o)
{
    SR_COLOUR_MATH
    const uint32_t c00 = texel_at(texels, t.o00), c10 = texel_at(texels, t.o10), c01 = texel_at(texels, t.o01), c11 = texel_at(texels, t.o11);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int sh = 8 * c;
        const float t00 = dec[(c00 >> sh) & 255u], t10 = dec[(c10 >> sh) & 255u];
        const float t01 = dec[(c01 >> sh) & 255u], t11 = dec[(c11 >> sh) & 255u];
        o[c] = ((t.w00 * t00 + t.w10 * t10) + t.w01 * t01) + t.w11 * t11;
    }
}
template <int NCH>
__device__ inline void sample_level(const TexDesc& t, const uint32_t* lvl, int level, float s, float tt, const float* dec, float* o)
{
    fetch_taps<NCH>(t.texels, level_taps(t, lvl, level, s, tt), dec, o);
}

// The two footprints of a trilinear fetch and the weight of the coarser one (0: the finer level alone).
struct TrilinearTaps {
    Taps hi, lo;
    float delta;
};
// log2 of the isotropic scale factor rho of a texture under the 2x2-quad differences of its coordinates: the level of detail before
// the sampler's bias and clamps
__device__ inline float lod_of_footprint(const TexDesc& t, float dsdx, float dtdx, float dsdy, float dtdy)
{
    SR_COLOUR_MATH
    const float mux = dsdx * (float)t.w, mvx = dtdx * (float)t.h;
    const float muy = dsdy * (float)t.w, mvy = dtdy * (float)t.h;
    const float rx2 = mux * mux + mvx * mvx, ry2 = muy * muy + mvy * mvy;
    const float r2 = rx2 > ry2 ? rx2 : ry2;
    return 0.5f * __builtin_amdgcn_logf(r2);
}
// texture() with log2(rho) supplied; trilinear, REPEAT
__device__ inline TrilinearTaps trilinear_taps_at(const TexDesc& t, const uint32_t* lvl, float s, float tt, float log2_rho)
{
    SR_COLOUR_MATH
    float lambda = log2_rho + t.bias;
    if (!(lambda > t.min_lod)) lambda = t.min_lod;
    if (lambda > t.max_lod) lambda = t.max_lod;
    const float q = (float)(t.levels - 1);
    float d = lambda; if (d < 0.0f) d = 0.0f; if (d > q) d = q;
    const float dhi = floorf(d);
    const int lhi = (int)dhi, llo = min(lhi + 1, t.levels - 1);
    TrilinearTaps tp;
    tp.delta = d - dhi;
    tp.hi = level_taps(t, lvl, lhi, s, tt);
    tp.lo = tp.hi;
    if (tp.delta != 0.0f) tp.lo = level_taps(t, lvl, llo, s, tt);
    return tp;
}
// texture() with the 2x2-quad differences supplied; isotropic LOD, trilinear, REPEAT
__device__ inline TrilinearTaps trilinear_taps(const TexDesc& t, const uint32_t* lvl, float s, float tt, float dsdx, float dtdx, float dsdy, float dtdy)
{
    return trilinear_taps_at(t, lvl, s, tt, lod_of_footprint(t, dsdx, dtdx, dsdy, dtdy));
}
template <int NCH>
__device__ inline void fetch_trilinear(const uint32_t* texels, const TrilinearTaps& tp, const float* dec, float* o)
{
    SR_COLOUR_MATH
    float chi[NCH];
    fetch_taps<NCH>(texels, tp.hi, dec, chi);
    if (tp.delta == 0.0f) {
#pragma unroll
        for (int c = 0; c < NCH; ++c) o[c] = chi[c];
        return;
    }
    float clo[NCH];
    fetch_taps<NCH>(texels, tp.lo, dec, clo);
#pragma unroll
    for (int c = 0; c < NCH; ++c) o[c] = (1.0f - tp.delta) * chi[c] + tp.delta * clo[c];
}
template <int NCH>
__device__ inline void sample_texture(const TexDesc& t, const uint32_t* lvl, float s, float tt, float dsdx, float dtdx, float dsdy, float dtdy,
                                      const float* dec, float* o)
{
    fetch_trilinear<NCH>(t.texels, trilinear_taps(t, lvl, s, tt, dsdx, dtdx, dsdy, dtdy), dec, o);
}

// texture() with the sampler's anisotropy applied, by the method the Vulkan specification spells out (Scale
// Factor Operation / Texel Anisotropic Filtering): eta = min(rho_max / rho_min, max_aniso), N = ceil(eta),
// level of detail from rho_max / eta, and the mean of N trilinear samples placed at i / (N + 1) - 1/2,
// i = 1..N, along the derivative of the major axis.  A sampler with maxAnisotropy <= 1 gets the isotropic footprint.
template <int NCH>
__device__ inline void sample_texture_aniso(const TexDesc& t, const uint32_t* lvl, float s, float tt, float dsdx, float dtdx,
                                            float dsdy, float dtdy, const float* dec, float* o)
{
    SR_COLOUR_MATH
    const float max_aniso = t.max_aniso;
    if (!(max_aniso > 1.0f)) { sample_texture<NCH>(t, lvl, s, tt, dsdx, dtdx, dsdy, dtdy, dec, o); return; }   // uniform over the launch
    const float mux = dsdx * (float)t.w, mvx = dtdx * (float)t.h;
    const float muy = dsdy * (float)t.w, mvy = dtdy * (float)t.h;
    const float rx2 = mux * mux + mvx * mvx, ry2 = muy * muy + mvy * mvy;
    const bool x_major = rx2 > ry2;
    const float rmax = sqrtf(x_major ? rx2 : ry2), rmin = sqrtf(x_major ? ry2 : rx2);
    float eta = 1.0f;
    if (rmax > 0.0f) {
        eta = rmin > 0.0f ? rmax * __builtin_amdgcn_rcpf(rmin) : max_aniso;
        if (!(eta < max_aniso)) eta = max_aniso;
        if (!(eta > 1.0f)) eta = 1.0f;
    }
    const int n = (int)ceilf(eta);
    float lambda = __builtin_amdgcn_logf(rmax * __builtin_amdgcn_rcpf(eta)) + t.bias;
    if (!(lambda > t.min_lod)) lambda = t.min_lod;
    if (lambda > t.max_lod) lambda = t.max_lod;
    const float q = (float)(t.levels - 1);
    float d = lambda; if (d < 0.0f) d = 0.0f; if (d > q) d = q;
    const float dhi = floorf(d), delta = d - dhi;
    const int lhi = (int)dhi, llo = min(lhi + 1, t.levels - 1);
    const float ds = x_major ? dsdx : dsdy, dt = x_major ? dtdx : dtdy;
    const float step = __builtin_amdgcn_rcpf((float)(n + 1));
    float acc[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) acc[c] = 0.0f;
    for (int i = 1; i <= n; ++i) {
        const float off = (float)i * step - 0.5f, si = s + off * ds, ti = tt + off * dt;
        float chi[NCH];
        sample_level<NCH>(t, lvl, lhi, si, ti, dec, chi);
        if (delta != 0.0f) {
            float clo[NCH];
            sample_level<NCH>(t, lvl, llo, si, ti, dec, clo);
#pragma unroll
            for (int c = 0; c < NCH; ++c) chi[c] = (1.0f - delta) * chi[c] + delta * clo[c];
        }
#pragma unroll
        for (int c = 0; c < NCH; ++c) acc[c] += chi[c];
    }
    const float inv = __builtin_amdgcn_rcpf((float)n);
#pragma unroll
    for (int c = 0; c < NCH; ++c) o[c] = acc[c] * inv;
}

// Colour arithmetic below is allowed 1 LSB of the 8-bit result against the CPU oracle, so it uses
// the hardware reciprocal / reciprocal-square-root / exp2 / log2 (1 ulp) instead of IEEE division.
// Nothing here feeds coverage, depth or triangle ids.
__device__ inline float max0(float x) { return x > 0.0f ? x : 0.0f; }
__device__ inline float frcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ inline float frsq(float x) { return __builtin_amdgcn_rsqf(x); }
__device__ inline float dot3(const float* a, const float* b) { return fmaf(a[2], b[2], fmaf(a[1], b[1], a[0] * b[0])); }
__device__ inline void normalize3(const float* a, float* o)
{
    const float inv = frsq(dot3(a, a));
    o[0] = a[0] * inv; o[1] = a[1] * inv; o[2] = a[2] * inv;
}
__device__ inline float fast_pow(float x, float y) { return __builtin_amdgcn_exp2f(y * __builtin_amdgcn_logf(x)); }   // pow() of the shader; x >= 0

__device__ inline float geometry_schlick_ggx(float NdotV, float roughness)
{
    const float r = roughness + 1.0f, k = (r * r) * 0.125f;
    return NdotV * frcp(fmaf(NdotV, 1.0f - k, k));
}

// main() of shaders/shader.frag with the three texture() results passed in
__device__ inline void fragment_main(const float* wp, const float* N, const float* cam, const float* albedo_tex,
                                     float metallic, float roughness, float* out, float* log2_out /* log2 of out */)
{
    SR_COLOUR_MATH
    const float kPI = 3.14159265359f, kInvPI = 1.0f / 3.14159265359f;
    const float albedo[3] = { fast_pow(albedo_tex[0], 2.2f), fast_pow(albedo_tex[1], 2.2f), fast_pow(albedo_tex[2], 2.2f) };
    const float ao = 0.01f;
    float V[3], t[3];
    t[0] = cam[0] - wp[0]; t[1] = cam[1] - wp[1]; t[2] = cam[2] - wp[2];
    normalize3(t, V);
    float F0[3], diff[3];
    const float one_m = 1.0f - metallic;
#pragma unroll
    for (int c = 0; c < 3; ++c) { F0[c] = fmaf(albedo[c], metallic, 0.04f * one_m); diff[c] = albedo[c] * kInvPI * one_m; }
    float Lo[3] = { 0.0f, 0.0f, 0.0f };
    const float power[3] = { 23.47f, 21.31f, 20.79f };
    const float NdV = dot3(N, V), NdotV = max0(NdV);
    const float a = roughness * roughness, a2 = a * a;
    const float gv = geometry_schlick_ggx(NdotV, roughness);
    const float rp1 = roughness + 1.0f, gk = (rp1 * rp1) * 0.125f;   // k of GeometrySchlickGGX
    const float a2gv = a2 * gv;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float lp[3] = { (i & 1) ? 10.0f : -10.0f, (i & 2) ? -10.0f : 10.0f, 10.0f };
        const float D[3] = { lp[0] - wp[0], lp[1] - wp[1], lp[2] - wp[2] };
        const float dd = dot3(D, D);
        const float invd = frsq(dd);
        // L = D * invd and H = normalize(V + L) are never formed: with |V| = |L| = 1, |V + L|^2 = 2 + 2 V.L, so every dot product
        // with H follows from the dot products with V and D (N is NOT a unit vector here -- the shader never normalises it -- and
        // need not be)
        const float NdL = dot3(N, D) * invd, VdL = dot3(V, D) * invd;
        const float inv_t = frsq(fmaf(2.0f, VdL, 2.0f));
        const float attenuation = invd * invd;           // 1 / dot(D, D)
        const float NdotH = max0((NdV + NdL) * inv_t), NdotH2 = NdotH * NdotH;
        float denom = fmaf(NdotH2, a2 - 1.0f, 1.0f);
        denom = kPI * denom * denom;                      // NDF = a2 / denom
        const float NdotL = max0(NdL);
        const float gl = fmaf(NdotL, 1.0f - gk, gk);      // G = NdotL / gl * gv
        const float x1 = 1.0f - max0((1.0f + VdL) * inv_t);   // 1 - max(dot(H, V), 0)
        const float x2 = x1 * x1;
        const float fc = x2 * x2 * x1;                    // pow(x, 5.0)
        const float denominator = fmaf(4.0f * NdotV, NdotL, 0.001f);
        // NDF * G / max(denominator, 0.001) over one reciprocal of the three denominators' product (each lies between 1e-9 and 4)
        const float ndg = (a2gv * NdotL) * frcp((denom * gl) * (denominator > 0.001f ? denominator : 0.001f));
        const float att_ndl = attenuation * NdotL;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float F = fmaf(1.0f - F0[c], fc, F0[c]);
            const float radiance = power[c] * att_ndl;
            Lo[c] = fmaf(fmaf(1.0f - F, diff[c], ndg * F), radiance, Lo[c]);   // (kD*albedo/PI + specular) * radiance * NdotL
        }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float ambient = 0.03f * albedo[c] * ao;
        float color = ambient + Lo[c];
        color = color * frcp(color + 1.0f);
        log2_out[c] = __builtin_amdgcn_logf(color) * (1.0f / 2.2f);   // pow(color, 1.0 / 2.2)
        out[c] = __builtin_amdgcn_exp2f(log2_out[c]);
    }
}

struct FragParams {
    const uint32_t* idx;
    const float4* verts;       // 2 float4 per vertex: (pos.xyz, n.x) (n.yz, uv)
    const ViewXform* views;    // views of this pass
    const unsigned long long* vis;
    uint32_t* color;
    FrameGeom g;
    TexDesc albedo, metallic, roughness;
    TexDesc normal;            // read only by the kernels compiled with normal mapping
    ShadeTables tab;
    uint32_t view0;            // first view of this pass inside the target
    const uint32_t* packed_mr; // the metallic and roughness maps have the same size and sampler state and the device holds their red channels packed into
                               // one chain (launch_pack_red_channels): one set of footprints and one set of texel loads serves both.  Null: two fetches
    uint32_t lod_shared;       // bit 0 / 1: the metallic / roughness map has the albedo map's proportions (w_m * h_a == h_m * w_a), so its
    float lod_shift[2];        // log2(rho) is the albedo map's + lod_shift = log2(w_m / w_a)
    BlockFlags flags;          // written here, read by the start of the next draw
    PassReport report;         // host_counters != null: the first workgroup reports the counters of the pass to the host (a lean
                               // draw, which has no scan kernel to do it)
};

// shades triangle `tri` at pixel (x,y): packed B,G,R,A.  The window-space vertices are recomputed from
// the positions just loaded instead of gathered from the vertex stage's planes (six divergent loads cost
// the CU's address/cache path far more than the arithmetic), with the vertex stage's own function
// (IEEE divide, no contraction), so they are bit-identical to the ones the triangle was rasterised from.
// Each barycentric weight comes from its own cross product of the two opposite vertices seen from the
// pixel, so its error is relative to its own size (a form like l0 = 1 - l1 - l2 around vertex 0 loses l0
// when vertex 0 is far away and close to the eye plane: its huge 1/w then multiplies the rounding
// error of l1 + l2); the values at the other pixels of the 2x2 quad follow from the screen-space
// gradients.
// IDENT: the model matrix of every view of the launch is the identity (every configuration of the reference,
// src/main.cpp:991): the products with it are skipped.  The general product with an identity matrix returns p + 0 for a
// finite position, bit for bit, so the window-space vertices stay those of the vertex stage.
// NMAP: the normal comes from the normal map (getNormalFromMap of shaders/shader.frag:35-53 with its body, which the reference has
// commented out, restored): off unless the caller asks for it, because the reference itself shades with the interpolated normal.
template <bool ANISO, bool IDENT, bool NMAP>
__device__ inline uint32_t shade_pixel(const FragParams& p, const ViewXform& vx, uint32_t tri, int x, int y,
                                       const float* dec, const float* thr, const uint32_t* lvl /* [4][kMaxMipLevels] */)
{
    const uint32_t i0 = p.idx[3 * (size_t)tri], i1 = p.idx[3 * (size_t)tri + 1], i2 = p.idx[3 * (size_t)tri + 2];
    const float4 a0 = p.verts[2 * (size_t)i0], b0 = p.verts[2 * (size_t)i0 + 1];
    const float4 a1 = p.verts[2 * (size_t)i1], b1 = p.verts[2 * (size_t)i1 + 1];
    const float4 a2 = p.verts[2 * (size_t)i2], b2 = p.verts[2 * (size_t)i2 + 1];
    float c0[4], c1[4], c2[4];
    float w0[4], w1[4], w2[4], n0[3], n1[3], n2[3];   // world-space positions and normals of the three vertices
    ScreenVertex v0, v1, v2;
    if constexpr (IDENT) {
        // (the general product with the identity returns p + 0, which differs from p for a negative zero only; the sign of a zero
        // coordinate changes neither a window position -- every clip coordinate is a sum that contains the non-zero w term or is
        // a zero of either sign, which the viewport transform takes to the same value -- nor an interpolated value)
        w0[0] = a0.x; w0[1] = a0.y; w0[2] = a0.z;
        w1[0] = a1.x; w1[1] = a1.y; w1[2] = a1.z;
        w2[0] = a2.x; w2[1] = a2.y; w2[2] = a2.z;
        n0[0] = a0.w; n0[1] = b0.x; n0[2] = b0.y;
        n1[0] = a1.w; n1[1] = b1.x; n1[2] = b1.y;
        n2[0] = a2.w; n2[1] = b2.x; n2[2] = b2.y;
        mat4_mul_vec4(vx.pv, w0[0], w0[1], w0[2], 1.0f, c0);
        mat4_mul_vec4(vx.pv, w1[0], w1[1], w1[2], 1.0f, c1);
        mat4_mul_vec4(vx.pv, w2[0], w2[1], w2[2], 1.0f, c2);
        v0 = project_clip_coords_select(c0, p.g.W, p.g.H);
        v1 = project_clip_coords_select(c1, p.g.W, p.g.H);
        v2 = project_clip_coords_select(c2, p.g.W, p.g.H);
    } else {
        v0 = project_vertex(vx, a0.x, a0.y, a0.z, p.g.W, p.g.H, c0);
        v1 = project_vertex(vx, a1.x, a1.y, a1.z, p.g.W, p.g.H, c1);
        v2 = project_vertex(vx, a2.x, a2.y, a2.z, p.g.W, p.g.H, c2);
        mat4_mul_vec4(vx.model, a0.x, a0.y, a0.z, 1.0f, w0);
        mat4_mul_vec4(vx.model, a1.x, a1.y, a1.z, 1.0f, w1);
        mat4_mul_vec4(vx.model, a2.x, a2.y, a2.z, 1.0f, w2);
        mat3_mul_vec3(vx.model, a0.w, b0.x, b0.y, n0);
        mat3_mul_vec3(vx.model, a1.w, b1.x, b1.y, n1);
        mat3_mul_vec3(vx.model, a2.w, b2.x, b2.y, n2);
    }
    // (from here on nothing but the colour depends on the arithmetic)
    {
    SR_COLOUR_MATH
    float q0, q1, q2, q0x, q1x, q2x, q0y, q1y, q2y;
    if (v0.rw != 0.0f && v1.rw != 0.0f && v2.rw != 0.0f) {
        const int32_t px = x * kSubpix + kHalf, py = y * kSubpix + kHalf;
        const float e1x = (float)(v1.X - v0.X), e1y = (float)(v1.Y - v0.Y);
        const float e2x = (float)(v2.X - v0.X), e2y = (float)(v2.Y - v0.Y);
        const float a0x = (float)(v0.X - px), a0y = (float)(v0.Y - py), a1x = (float)(v1.X - px), a1y = (float)(v1.Y - py);
        const float a2x = (float)(v2.X - px), a2y = (float)(v2.Y - py);
        const float E0 = a1x * a2y - a1y * a2x, E1 = a2x * a0y - a2y * a0x, E2 = a0x * a1y - a0y * a1x;
        const float rA = frcp((E0 + E1) + E2);
        const float l0 = E0 * rA, l1 = E1 * rA, l2 = E2 * rA;
        // gradients per pixel step (256 sub-pixels)
        const float s = (float)kSubpix * rA;
        const float l1x = e2y * s, l1y = -e2x * s, l2x = -e1y * s, l2y = e1x * s;
        // perspective: attribute = sum(q_i a_i) / sum(q_i), q_i = l_i / w_i, all affine in screen space
        q0 = l0 * v0.rw; q1 = l1 * v1.rw; q2 = l2 * v2.rw;
        q0x = -(l1x + l2x) * v0.rw; q1x = l1x * v1.rw; q2x = l2x * v2.rw;
        q0y = -(l1y + l2y) * v0.rw; q1y = l1y * v1.rw; q2y = l2y * v2.rw;
    } else {
        // A triangle that reached the frame through the near-plane clipper: a vertex has no window position, so
        // the weights come from the clip-space coordinates (x, y, w).  With h_i = (x_i - w_i X, y_i - w_i Y) for
        // the pixel's normalised device coordinates (X, Y), q_i is proportional to cross(h_j, h_k), affine in
        // (X, Y) like the q_i above (any common factor cancels in the divisions by D below).
        const float hw = (float)p.g.W * 0.5f, hh = (float)p.g.H * 0.5f, ihw = 1.0f / hw, ihh = 1.0f / hh;
        const float X = ((float)x + 0.5f - hw) * ihw, Y = (hh - ((float)y + 0.5f)) * ihh;
        const float h0x = c0[0] - c0[3] * X, h0y = c0[1] - c0[3] * Y, h1x = c1[0] - c1[3] * X, h1y = c1[1] - c1[3] * Y;
        const float h2x = c2[0] - c2[3] * X, h2y = c2[1] - c2[3] * Y;
        q0 = h1x * h2y - h1y * h2x; q1 = h2x * h0y - h2y * h0x; q2 = h0x * h1y - h0y * h1x;
        q0x = -(c1[3] * h2y - h1y * c2[3]) * ihw; q0y = (h1x * c2[3] - c1[3] * h2x) * ihh;   // one pixel to the right: X + 1/hw; down: Y - 1/hh
        q1x = -(c2[3] * h0y - h2y * c0[3]) * ihw; q1y = (h2x * c0[3] - c2[3] * h0x) * ihh;
        q2x = -(c0[3] * h1y - h0y * c1[3]) * ihw; q2y = (h0x * c1[3] - c0[3] * h1x) * ihh;
    }
    const float D = q0 + q1 + q2, Dx = q0x + q1x + q2x, Dy = q0y + q1y + q2y;
    const float Nu = fmaf(q2, b2.z, fmaf(q1, b1.z, q0 * b0.z)), Nv = fmaf(q2, b2.w, fmaf(q1, b1.w, q0 * b0.w));
    const float Nux = fmaf(q2x, b2.z, fmaf(q1x, b1.z, q0x * b0.z)), Nvx = fmaf(q2x, b2.w, fmaf(q1x, b1.w, q0x * b0.w));
    const float Nuy = fmaf(q2y, b2.z, fmaf(q1y, b1.z, q0y * b0.z)), Nvy = fmaf(q2y, b2.w, fmaf(q1y, b1.w, q0y * b0.w));
    const float rD = frcp(D);
    const float u = Nu * rD, v = Nv * rD;
    const float p0 = q0 * rD, p1 = q1 * rD, p2 = q2 * rD;
    // coarse derivatives over the pixel's 2x2 quad: this triangle's uv at the quad's top-left,
    // top-right and bottom-left pixel centres (what helper invocations would compute)
    const float ox = (float)((x & ~1) - x), oy = (float)((y & ~1) - y);
    const float D00 = fmaf(oy, Dy, fmaf(ox, Dx, D)), Nu00 = fmaf(oy, Nuy, fmaf(ox, Nux, Nu)), Nv00 = fmaf(oy, Nvy, fmaf(ox, Nvx, Nv));
    const float r00 = frcp(D00), r10 = frcp(D00 + Dx), r01 = frcp(D00 + Dy);
    const float u00 = Nu00 * r00, v00 = Nv00 * r00;
    const float dudx = (Nu00 + Nux) * r10 - u00, dvdx = (Nv00 + Nvx) * r10 - v00;
    const float dudy = (Nu00 + Nuy) * r01 - u00, dvdy = (Nv00 + Nvy) * r01 - v00;

    float wp[3], N[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { wp[k] = fmaf(p2, w2[k], fmaf(p1, w1[k], p0 * w0[k])); N[k] = fmaf(p2, n2[k], fmaf(p1, n1[k], p0 * n0[k])); }
    if constexpr (NMAP) {
        // dFdx / dFdy of WorldPos: this triangle's world position at the quad's top-left, top-right and bottom-left pixel centres
        const float a0q = fmaf(oy, q0y, fmaf(ox, q0x, q0)), a1q = fmaf(oy, q1y, fmaf(ox, q1x, q1)), a2q = fmaf(oy, q2y, fmaf(ox, q2x, q2));
        float Q1[3], Q2[3], tn[3], Nn[3], T[3], B[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float w00 = fmaf(a2q, w2[k], fmaf(a1q, w1[k], a0q * w0[k])) * r00;
            const float w10 = fmaf(a2q + q2x, w2[k], fmaf(a1q + q1x, w1[k], (a0q + q0x) * w0[k])) * r10;
            const float w01 = fmaf(a2q + q2y, w2[k], fmaf(a1q + q1y, w1[k], (a0q + q0y) * w0[k])) * r01;
            Q1[k] = w10 - w00; Q2[k] = w01 - w00;
        }
        if constexpr (ANISO) sample_texture_aniso<3>(p.normal, lvl + 3 * kMaxMipLevels, u, v, dudx, dvdx, dudy, dvdy, dec, tn);
        else sample_texture<3>(p.normal, lvl + 3 * kMaxMipLevels, u, v, dudx, dvdx, dudy, dvdy, dec, tn);
#pragma unroll
        for (int k = 0; k < 3; ++k) { tn[k] = tn[k] * 2.0f - 1.0f; T[k] = Q1[k] * dvdy - Q2[k] * dvdx; }
        normalize3(N, Nn);
        normalize3(T, T);
        B[0] = Nn[1] * T[2] - Nn[2] * T[1]; B[1] = Nn[2] * T[0] - Nn[0] * T[2]; B[2] = Nn[0] * T[1] - Nn[1] * T[0];
        normalize3(B, B);
#pragma unroll
        for (int k = 0; k < 3; ++k) N[k] = T[k] * tn[0] - B[k] * tn[1] + Nn[k] * tn[2];   // mat3(T, -normalize(cross(N, T)), N) * tangentNormal
        normalize3(N, N);
    }
    float alb[3], met[1], rgh[1], rgb[3];
    if constexpr (ANISO) {
        sample_texture_aniso<3>(p.albedo, lvl, u, v, dudx, dvdx, dudy, dvdy, dec, alb);
        if (p.packed_mr) {     // (the same for the whole launch) both maps from the chain that packs them: one set of footprints and loads
            TexDesc both = p.metallic;
            both.texels = p.packed_mr;
            float mr[2];
            sample_texture_aniso<2>(both, lvl + kMaxMipLevels, u, v, dudx, dvdx, dudy, dvdy, dec, mr);
            met[0] = mr[0]; rgh[0] = mr[1];
        } else {
            sample_texture_aniso<1>(p.metallic, lvl + kMaxMipLevels, u, v, dudx, dvdx, dudy, dvdy, dec, met);
            sample_texture_aniso<1>(p.roughness, lvl + 2 * kMaxMipLevels, u, v, dudx, dvdx, dudy, dvdy, dec, rgh);
        }
    } else {
        // A map with the albedo map's proportions scales its footprint by a constant: its log2(rho) is the albedo map's plus the
        // logarithm of the ratio of the widths (FragParams::lod_shift) -- one logarithm per pixel instead of three.
        const float la = lod_of_footprint(p.albedo, dudx, dvdx, dudy, dvdy);
        fetch_trilinear<3>(p.albedo.texels, trilinear_taps_at(p.albedo, lvl, u, v, la), dec, alb);
        const float lm = p.lod_shared & 1u ? la + p.lod_shift[0] : lod_of_footprint(p.metallic, dudx, dvdx, dudy, dvdy);   // (the same choice for the whole launch)
        const TrilinearTaps tm = trilinear_taps_at(p.metallic, lvl + kMaxMipLevels, u, v, lm);
        if (p.packed_mr) {     // (the same for the whole launch) one map holds both: a texel's red byte is the metallic map's, its green byte the roughness map's
            float mr[2];
            fetch_trilinear<2>(p.packed_mr, tm, dec, mr);
            met[0] = mr[0]; rgh[0] = mr[1];
        } else {
            fetch_trilinear<1>(p.metallic.texels, tm, dec, met);
            const float lr = p.lod_shared & 2u ? la + p.lod_shift[1] : lod_of_footprint(p.roughness, dudx, dvdx, dudy, dvdy);
            fetch_trilinear<1>(p.roughness.texels, trilinear_taps_at(p.roughness, lvl + 2 * kMaxMipLevels, u, v, lr), dec, rgh);
        }
    }
    float lg[3];
    fragment_main(wp, N, vx.cam, alb, met[0], rgh[0], rgb, lg);
    const uint32_t R = (uint32_t)srgb_encode8(rgb[0] > 1.0f ? 1.0f : rgb[0], thr, lg[0]);
    const uint32_t G = (uint32_t)srgb_encode8(rgb[1] > 1.0f ? 1.0f : rgb[1], thr, lg[1]);
    const uint32_t B = (uint32_t)srgb_encode8(rgb[2] > 1.0f ? 1.0f : rgb[2], thr, lg[2]);
    return B | (G << 8) | (R << 16) | (255u << 24);
    }
}

// One work-item per pixel, workgroups of kFragBlockW x kFragBlockH pixels.  The launch shape and the
// indexing of the 256-entry tables below are tied together by these two constants and the assertion.
constexpr uint32_t kFragBlockW = kFlagBlockW, kFragBlockH = kFlagBlockH;   // one flag per workgroup of the shading pass
static_assert(kFragBlockW * kFragBlockH == 256, "the tables of the shading kernel are filled by 256 work-items");

// the counters of the pass, as the scan kernel reports them (every kernel that writes them has finished); one wave
__device__ inline void report_pass(const PassReport& r, uint32_t lane)
{
    static_assert(kMicroSpread == 64, "one counter per lane");
    uint32_t micro = r.micro_spread[lane * kMicroSpreadStride];
    for (int o = 32; o > 0; o >>= 1) micro += (uint32_t)__shfl_down((int)micro, o);
    if (lane == 0) {
        uint32_t binned = 0;
        for (uint32_t v = 0; v < r.n_views; ++v) binned += r.pending_total[v * kPendingTotalStride];
        uint32_t* h = r.host_counters;
        h[0] = 0u; h[1] = r.counters[kCtrDirectTooBig]; h[2] = r.counters[kCtrUnprojectable]; h[3] = binned; h[4] = micro;
        h[5] = r.counters[kCtrClipListed]; h[6] = r.counters[kCtrClipped];
    }
}

template <int S, bool ANISO, bool IDENT, bool NMAP>
__device__ __forceinline__ void fragment_body(const FragParams& p)
{
    __shared__ float s_dec[256];
    __shared__ float s_thr[256];
    __shared__ uint32_t s_lvl[4 * kMaxMipLevels];
    const uint32_t tid = threadIdx.y * blockDim.x + threadIdx.x;
    const int lx = (int)(blockIdx.x * blockDim.x + threadIdx.x), ly = (int)(blockIdx.y * blockDim.y + threadIdx.y);
    const uint32_t view = blockIdx.z;
    const bool inside = lx < p.g.rw && ly < p.g.rh;
    const size_t pix = ((size_t)(p.view0 + view) * p.g.rh + (inside ? ly : 0)) * p.g.rw + (inside ? lx : 0);
    // A workgroup over a block that no kernel of this draw wrote a visibility word of (most of them, in a frame like the
    // benchmark's) reads nothing: it stores the clear colour, if the block's colour is not known to be that already, and leaves.
    uint32_t& dirty = p.flags.dirty_at(p.view0 + view, blockIdx.x, blockIdx.y);
    if (p.flags.at(p.view0 + view, blockIdx.x, blockIdx.y) == 0u) {   // the same for the whole workgroup
        if (dirty != 0u) {
            if (inside) p.color[pix] = kColorClear;
            __syncthreads();   // (every work-item has read the flag)
            if (tid == 0) dirty = 0u;
        }
        return;
    }
    if (tid == 0) dirty = 1u;
    uint32_t ids[S];
#pragma unroll
    for (int s = 0; s < S; ++s) ids[s] = inside ? (uint32_t)p.vis[pix * S + s] : kNoTri;
    if (tid < 256u) {   // (always: the launcher uses kFragBlockW x kFragBlockH; a different shape must not index past the tables)
        s_dec[tid] = p.tab.decode[tid];
        s_thr[tid] = tid < 255u ? p.tab.thresh[tid] : 3.0e38f;
    }
    if (tid < (NMAP ? 4u : 3u) * kMaxMipLevels) {
        const uint32_t m = tid / kMaxMipLevels, l = tid % kMaxMipLevels;
        s_lvl[tid] = m == 0 ? p.albedo.level_off[l] : m == 1 ? p.metallic.level_off[l] : m == 2 ? p.roughness.level_off[l] : p.normal.level_off[l];
    }
    __syncthreads();
    const ViewXform& vx = p.views[view];
    uint32_t col[S];
    if constexpr (S == 1) {
        if (!inside) return;
        col[0] = ids[0] == kNoTri ? kColorClear : shade_pixel<ANISO, IDENT, NMAP>(p, vx, ids[0], p.g.rx + lx, p.g.ry + ly, s_dec, s_thr, s_lvl);
    } else {
        // Several samples per pixel: a pixel is shaded once per DISTINCT triangle among its samples.  Done pixel by pixel, a wave takes
        // as many turns through the shading code as its worst pixel has triangles (four, wherever a vertex lies inside a pixel), most
        // lanes idle in the later ones.  So the workgroup first lists its (pixel, triangle) pairs in LDS, shades the list with every
        // lane busy -- kFragTurns turns of 256 pairs at most, as many as the list needs -- and every pixel then picks up the colours
        // of its samples' triangles.
        constexpr uint32_t kFragTurns = S;
        __shared__ uint32_t s_pair_tri[kFragTurns * 256u], s_pair_px[kFragTurns * 256u], s_pair_col[kFragTurns * 256u];
        __shared__ uint32_t s_wave_pairs[4];
        uint32_t uniq[S];
        int slot_of[S];     // which of the pixel's distinct triangles covers sample s; -1: none
        uint32_t n = 0;
#pragma unroll
        for (int s = 0; s < S; ++s) {
            slot_of[s] = -1;
            if (ids[s] == kNoTri) continue;
#pragma unroll
            for (int k = 0; k < s; ++k) if (slot_of[s] < 0 && slot_of[k] >= 0 && ids[k] == ids[s]) slot_of[s] = slot_of[k];
            if (slot_of[s] < 0) {
#pragma unroll
                for (int k = 0; k < S; ++k) if ((uint32_t)k == n) uniq[k] = ids[s];     // (no indexing of registers by a variable)
                slot_of[s] = (int)n++;
            }
        }
        const uint32_t lane = lane_id(), wave = tid >> 6;
        uint32_t incl = n;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t v = (uint32_t)__shfl_up((int)incl, o); if (lane >= (uint32_t)o) incl += v; }
        if (lane == 63u) s_wave_pairs[wave] = incl;
        __syncthreads();
        uint32_t first = incl - n, total = 0;
#pragma unroll
        for (uint32_t w = 0; w < 4u; ++w) { if (w < wave) first += s_wave_pairs[w]; total += s_wave_pairs[w]; }
#pragma unroll
        for (int k = 0; k < S; ++k)
            if ((uint32_t)k < n) { s_pair_tri[first + k] = uniq[k]; s_pair_px[first + k] = tid | ((uint32_t)k << 8); }
        __syncthreads();
#pragma unroll
        for (uint32_t turn = 0; turn < kFragTurns; ++turn) {
            const uint32_t w = turn * 256u + tid;
            if (w < total) {
                const uint32_t px = s_pair_px[w], t = px & 255u;
                // (the pixel of work-item t of this workgroup: the launch is kFragBlockW wide)
                const int wx = p.g.rx + (int)(blockIdx.x * kFlagBlockW + (t % kFlagBlockW)), wy = p.g.ry + (int)(blockIdx.y * kFlagBlockH + (t / kFlagBlockW));
                s_pair_col[(px >> 8) * 256u + t] = shade_pixel<ANISO, IDENT, NMAP>(p, vx, s_pair_tri[w], wx, wy, s_dec, s_thr, s_lvl);
            }
        }
        __syncthreads();
        if (!inside) return;
#pragma unroll
        for (int s = 0; s < S; ++s) col[s] = slot_of[s] < 0 ? kColorClear : s_pair_col[(uint32_t)slot_of[s] * 256u + tid];
    }
    uint32_t outc;
    if (S == 1) outc = col[0];
    else {
        outc = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float acc = 0.0f;
#pragma unroll
            for (int s = 0; s < S; ++s) {
                const uint32_t b = (col[s] >> (8 * k)) & 255u;
                acc += k == 3 ? (float)b / 255.0f : s_dec[b];
            }
            acc = acc / (float)S;
            const uint32_t ch = k == 3 ? (uint32_t)unorm_encode8(acc) : (uint32_t)srgb_encode8(acc, s_thr);
            outc |= ch << (8 * k);
        }
    }
    p.color[pix] = outc;
}

template <int S, bool ANISO, bool IDENT, bool NMAP>
__global__ __launch_bounds__(256) SR_FRAG_ATTR void fragment_kernel(FragParams p)
{
    fragment_body<S, ANISO, IDENT, NMAP>(p);
}
// the same, for a lean draw: the launch's first workgroup reports the counters of the pass before it turns to its pixels
template <int S, bool ANISO, bool IDENT>
__global__ __launch_bounds__(256) SR_FRAG_ATTR void fragment_report_kernel(FragParams p)
{
    if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.y * blockDim.x + threadIdx.x < 64u) report_pass(p.report, lane_id());
    fragment_body<S, ANISO, IDENT, false>(p);
}

// =================================================================================================
// mip-chain builder: one linear blit per level (src/texture.cpp:119-164), sRGB-correct
// =================================================================================================
// a workgroup of kMipBlockW * kMipBlockH work-items (one-dimensional) covers kMipBlockW x kMipBlockH texels of the destination
// Two maps of the same size of which the shader reads the red channel only (the reference's metallic and roughness maps): texel i of
// the result holds a's red byte and, above it, b's.
__global__ __launch_bounds__(256) void pack_red_channels_kernel(const uint32_t* __restrict__ a, const uint32_t* __restrict__ b, size_t n, uint32_t* __restrict__ out)
{
    for (size_t i = (size_t)blockIdx.x * 256u + threadIdx.x; i < n; i += (size_t)gridDim.x * 256u) out[i] = (a[i] & 255u) | ((b[i] & 255u) << 8);
}

constexpr uint32_t kMipBlockW = 32, kMipBlockH = 8;
__global__ __launch_bounds__(256) void mip_blit_kernel(const uint32_t* __restrict__ src, int sw, int sh, uint32_t* __restrict__ dst, int dw, int dh,
                                                       ShadeTables tab)
{
    static_assert(kMipBlockW * kMipBlockH == 256, "launch bound of mip_blit_kernel");
    const int i = (int)(blockIdx.x * kMipBlockW + (threadIdx.x % kMipBlockW)), j = (int)(blockIdx.y * kMipBlockH + (threadIdx.x / kMipBlockW));
    if (i >= dw || j >= dh) return;
    const float sx = (float)sw / (float)dw, sy = (float)sh / (float)dh;
    const float u = ((float)i + 0.5f) * sx - 0.5f, v = ((float)j + 0.5f) * sy - 0.5f;
    const float fu = floorf(u), fv = floorf(v);
    const float a = u - fu, b = v - fv;
    const int i0 = min(max((int)fu, 0), sw - 1), i1 = min(max((int)fu + 1, 0), sw - 1);
    const int j0 = min(max((int)fv, 0), sh - 1), j1 = min(max((int)fv + 1, 0), sh - 1);
    const uint32_t c00 = src[(size_t)j0 * sw + i0], c10 = src[(size_t)j0 * sw + i1];
    const uint32_t c01 = src[(size_t)j1 * sw + i0], c11 = src[(size_t)j1 * sw + i1];
    const float w00 = (1.0f - a) * (1.0f - b), w10 = a * (1.0f - b), w01 = (1.0f - a) * b, w11 = a * b;
    uint32_t outc = 0;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int sh8 = 8 * c;
        const uint32_t b00 = (c00 >> sh8) & 255u, b10 = (c10 >> sh8) & 255u, b01 = (c01 >> sh8) & 255u, b11 = (c11 >> sh8) & 255u;
        float t00, t10, t01, t11;
        if (c < 3) { t00 = tab.decode[b00]; t10 = tab.decode[b10]; t01 = tab.decode[b01]; t11 = tab.decode[b11]; }
        else { t00 = (float)b00 / 255.0f; t10 = (float)b10 / 255.0f; t01 = (float)b01 / 255.0f; t11 = (float)b11 / 255.0f; }
        const float r = ((w00 * t00 + w10 * t10) + w01 * t01) + w11 * t11;
        const uint32_t e = c < 3 ? (uint32_t)srgb_encode8(r, tab.thresh) : (uint32_t)unorm_encode8(r);
        outc |= e << sh8;
    }
    dst[(size_t)j * dw + i] = outc;
}

// =================================================================================================
// chunk culling (sr_device.h): bounds per chunk at mesh creation, one projection test per chunk and draw
// =================================================================================================
constexpr uint32_t kChunkCullTris = kSetupBlockTris;   // a chunk is what one set-up workgroup handles
__device__ inline float block_reduce(float v, bool take_max, float* s_tmp /* 4 floats */, uint32_t tid)
{
    for (int o = 32; o > 0; o >>= 1) { const float u = __shfl_down(v, o); v = take_max ? fmaxf(v, u) : fminf(v, u); }   // lane 0 ends up with the wave's
    __syncthreads();   // s_tmp's last readers are through
    if ((tid & 63u) == 0u) s_tmp[tid >> 6] = v;
    __syncthreads();
    const float a = s_tmp[0], b = s_tmp[1], c = s_tmp[2], d = s_tmp[3];
    return take_max ? fmaxf(fmaxf(a, b), fmaxf(c, d)) : fminf(fminf(a, b), fminf(c, d));
}
// One workgroup per chunk: bounding sphere (centre of the bounding box, largest distance of a referenced position from it, a few ulp
// added) and the range of vertex numbers.  A position that is not finite makes the radius infinite: such a chunk is never culled.
__global__ __launch_bounds__(256) void chunk_bounds_kernel(const uint32_t* __restrict__ idx, uint32_t n_tris, const float4* __restrict__ verts /* 2 float4 per vertex */,
                                                           float4* __restrict__ sphere, uint4* __restrict__ vrange)
{
    __shared__ float s_tmp[4];
    __shared__ uint32_t s_lo, s_hi, s_groups[2];
    const uint32_t tid = threadIdx.x, first = blockIdx.x * kChunkCullTris;
    if (tid == 0) { s_lo = 0xFFFFFFFFu; s_hi = 0u; s_groups[0] = s_groups[1] = 0u; }
    uint32_t vid[3 * kSetupTris];
    float p[3 * kSetupTris][3];
    bool have[kSetupTris];
    float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
    uint32_t vlo = 0xFFFFFFFFu, vhi = 0u;
    bool finite = true;
#pragma unroll
    for (uint32_t k = 0; k < kSetupTris; ++k) {
        const uint32_t t = first + k * 256u + tid;
        have[k] = t < n_tris;
#pragma unroll
        for (uint32_t c = 0; c < 3u; ++c) {
            const uint32_t i = have[k] ? idx[3 * (size_t)t + c] : 0u;
            vid[3 * k + c] = i;
            const float4 a = have[k] ? verts[2 * (size_t)i] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            p[3 * k + c][0] = a.x; p[3 * k + c][1] = a.y; p[3 * k + c][2] = a.z;
            if (have[k]) {
                vlo = min(vlo, i); vhi = max(vhi, i);
                finite = finite && fabsf(a.x) <= 3.0e38f && fabsf(a.y) <= 3.0e38f && fabsf(a.z) <= 3.0e38f;   // (false for infinities and NaNs)
                lo[0] = fminf(lo[0], a.x); lo[1] = fminf(lo[1], a.y); lo[2] = fminf(lo[2], a.z);
                hi[0] = fmaxf(hi[0], a.x); hi[1] = fmaxf(hi[1], a.y); hi[2] = fmaxf(hi[2], a.z);
            }
        }
    }
    float centre[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float l = block_reduce(lo[c], false, s_tmp, tid), h = block_reduce(hi[c], true, s_tmp, tid);
        centre[c] = 0.5f * l + 0.5f * h;
    }
    float r2 = 0.0f;
#pragma unroll
    for (uint32_t k = 0; k < kSetupTris; ++k)
#pragma unroll
        for (uint32_t c = 0; c < 3u; ++c)
            if (have[k]) {
                const float dx = p[3 * k + c][0] - centre[0], dy = p[3 * k + c][1] - centre[1], dz = p[3 * k + c][2] - centre[2];
                r2 = fmaxf(r2, dx * dx + dy * dy + dz * dz);
            }
    r2 = block_reduce(r2, true, s_tmp, tid);
    const float all_finite = block_reduce(finite ? 1.0f : 0.0f, false, s_tmp, tid);
    atomicMin(&s_lo, vlo); atomicMax(&s_hi, vhi);
    __syncthreads();
    // which of the groups of kCullVertexGroup vertices from the lowest one's on the chunk's triangles use (the first 64 of them: a chunk
    // that spans more has every vertex transformed anyway)
    const uint32_t group0 = s_lo / kCullVertexGroup;
#pragma unroll
    for (uint32_t k = 0; k < kSetupTris; ++k)
#pragma unroll
        for (uint32_t c = 0; c < 3u; ++c)
            if (have[k]) {
                const uint32_t gi = vid[3 * k + c] / kCullVertexGroup - group0;
                if (gi < 64u) atomicOr(&s_groups[gi >> 5], 1u << (gi & 31u));
            }
    __syncthreads();
    if (tid == 0) {
        float r = sqrtf(r2) * 1.00001f + 1.0e-30f;            // (rounding of the squares and the root: a few ulp; this adds a thousand)
        if (!(all_finite > 0.5f) || !(r <= 3.0e38f)) r = __builtin_inff();
        sphere[blockIdx.x] = make_float4(centre[0], centre[1], centre[2], r);
        vrange[blockIdx.x] = make_uint4(s_lo, s_hi, s_groups[0], s_groups[1]);
    }
}

// Chunk vertex lists (sr_device.h), one workgroup per chunk: a bitmap over the chunk's span of vertex numbers in LDS gives the distinct
// vertices in ascending order (the same lists whatever order the work-items run in); a triangle's vertex then stands at the number of
// set bits before its own.  A chunk that does not fit counts in `unfit` (the mesh then has no lists).
__global__ __launch_bounds__(256) void chunk_verts_kernel(const uint32_t* __restrict__ idx, uint32_t n_tris, const uint4* __restrict__ vrange,
                                                          uint32_t* __restrict__ list, uint32_t* __restrict__ count, uint32_t* __restrict__ local,
                                                          uint32_t* __restrict__ unfit)
{
    constexpr uint32_t kWords = kChunkVertsSpan / 32u, kPer = kWords / 256u;     // words of the bitmap, and of one work-item's stretch of it
    static_assert(kWords % 256u == 0u, "every work-item takes the same number of consecutive words");
    static_assert(kChunkVertsMax <= 1024u, "three list positions share a 32-bit word");
    __shared__ uint32_t s_bits[kWords], s_before[kWords];
    __shared__ uint32_t s_wave[4];
    const uint32_t tid = threadIdx.x, lane = lane_id(), wave = tid >> 6, chunk = blockIdx.x, first = chunk * kChunkCullTris;
    const uint4 r = vrange[chunk];
    if (r.x > r.y || r.y - r.x >= kChunkVertsSpan) {      // (the same for the whole workgroup)
        if (tid == 0) { count[chunk] = 0u; atomicAdd(unfit, 1u); }
        return;
    }
    for (uint32_t w = tid; w < kWords; w += 256u) s_bits[w] = 0u;
    __syncthreads();
    uint32_t v[kSetupTris][3];
    bool have[kSetupTris];
#pragma unroll
    for (uint32_t k = 0; k < kSetupTris; ++k) {
        const uint32_t t = first + k * 256u + tid;
        have[k] = t < n_tris;
#pragma unroll
        for (uint32_t c = 0; c < 3u; ++c) {
            v[k][c] = have[k] ? idx[3 * (size_t)t + c] - r.x : 0u;
            if (have[k]) atomicOr(&s_bits[v[k][c] >> 5], 1u << (v[k][c] & 31u));
        }
    }
    __syncthreads();
    uint32_t mine = 0;
#pragma unroll
    for (uint32_t j = 0; j < kPer; ++j) mine += (uint32_t)__popc(s_bits[tid * kPer + j]);
    uint32_t incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t u = (uint32_t)__shfl_up((int)incl, o); if (lane >= (uint32_t)o) incl += u; }
    if (lane == 63u) s_wave[wave] = incl;
    __syncthreads();
    uint32_t before = incl - mine;
    for (uint32_t w = 0; w < wave; ++w) before += s_wave[w];
    const uint32_t total = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
    if (total > kChunkVertsMax) {                          // (the same for the whole workgroup)
        if (tid == 0) { count[chunk] = 0u; atomicAdd(unfit, 1u); }
        return;
    }
    for (uint32_t j = 0; j < kPer; ++j) {
        const uint32_t word = tid * kPer + j;
        s_before[word] = before;
        for (uint32_t bits = s_bits[word]; bits != 0u; bits &= bits - 1u)
            list[(size_t)chunk * kChunkVertsMax + before++] = r.x + word * 32u + (uint32_t)__ffs((int)bits) - 1u;
    }
    __syncthreads();
#pragma unroll
    for (uint32_t k = 0; k < kSetupTris; ++k) {
        if (!have[k]) continue;
        uint32_t packed = 0;
#pragma unroll
        for (uint32_t c = 0; c < 3u; ++c) {
            const uint32_t word = v[k][c] >> 5;
            packed |= (s_before[word] + (uint32_t)__popc(s_bits[word] & ((1u << (v[k][c] & 31u)) - 1u))) << (10u * c);
        }
        local[first + k * 256u + tid] = packed;
    }
    if (tid == 0) count[chunk] = total;
}

// One work-item per chunk.  With A = proj * view * model, every position p of the chunk has clip coordinate
// A_r . (c, 1) + A_r[0..2] . (p - c), the second term at most radius * |A_r[0..2]| in size.  If w stays positive over the sphere,
// x / w and y / w over it are bounded by the quotients of the ends, and the chunk's window coordinates by the viewport transform of
// those; slack: kCullSlackPixels plus the rounding the device's own float evaluation of a vertex can differ by.  Anything not
// provably outside stays (comparisons with a NaN are false).
__global__ __launch_bounds__(256) void chunk_cull_kernel(ChunkCull cc, CullView v, FrameGeom g)
{
    const uint32_t chunk = blockIdx.x * 256u + threadIdx.x;
    bool culled = false;
    if (chunk < cc.n_chunks) {
        const float4 s = cc.sphere[chunk];
        const float reach = fabsf(s.x) + fabsf(s.y) + fabsf(s.z) + s.w + 1.0f;
        const float round_off = 1.0e-5f * v.scale * reach;          // (the device's clip coordinates carry a few ulp of the terms' sizes: 1e-7 of them)
        const float cx = fmaf(v.ax[2], s.z, fmaf(v.ax[1], s.y, fmaf(v.ax[0], s.x, v.ax[3])));
        const float cy = fmaf(v.ay[2], s.z, fmaf(v.ay[1], s.y, fmaf(v.ay[0], s.x, v.ay[3])));
        const float cw = fmaf(v.aw[2], s.z, fmaf(v.aw[1], s.y, fmaf(v.aw[0], s.x, v.aw[3])));
        const float ex = s.w * v.nx * 1.0001f + round_off, ey = s.w * v.ny * 1.0001f + round_off, ew = s.w * v.nw * 1.0001f + round_off;
        const float w_lo = cw - ew, w_hi = cw + ew;
        if (w_lo > 1.0e-6f * (fabsf(cw) + ew) && w_lo > 0.0f) {     // every vertex in front of the eye, by a margin
            const float x_lo = cx - ex, x_hi = cx + ex, y_lo = cy - ey, y_hi = cy + ey;
            const float ndc_x_max = fmaxf(x_hi / w_lo, x_hi / w_hi), ndc_x_min = fminf(x_lo / w_lo, x_lo / w_hi);
            const float ndc_y_max = fmaxf(y_hi / w_lo, y_hi / w_hi), ndc_y_min = fminf(y_lo / w_lo, y_lo / w_hi);
            const float hw = 0.5f * (float)g.W, hh = 0.5f * (float)g.H;
            const float slack_x = kCullSlackPixels + 1.0e-5f * (float)g.W, slack_y = kCullSlackPixels + 1.0e-5f * (float)g.H;
            const float wx_min = hw + ndc_x_min * hw - slack_x, wx_max = hw + ndc_x_max * hw + slack_x;
            const float wy_min = hh - ndc_y_max * hh - slack_y, wy_max = hh - ndc_y_min * hh + slack_y;    // (the viewport is flipped: src/main.cpp:564-571)
            culled = wx_max < (float)g.rx || wx_min > (float)(g.rx + g.rw) || wy_max < (float)g.ry || wy_min > (float)(g.ry + g.rh);
        }
        cc.visible[chunk] = culled ? 0u : 1u;
        if (!culled) {
            const uint4 r = cc.vrange[chunk];
            const uint32_t g0 = r.x / kCullVertexGroup, g1 = r.y / kCullVertexGroup;
            const uint64_t used = (uint64_t)r.z | ((uint64_t)r.w << 32);
            if (r.x > r.y) { /* a chunk without triangles refers to no vertex */ }
            else if (g1 - g0 >= kCullMaxGroupsPerChunk) cc.stamp[0] = cc.epoch;      // scattered indices: every group (same value from every writer)
            else for (uint32_t k = g0; k <= g1; ++k) if ((used >> (k - g0)) & 1ull) cc.stamp[1u + k] = cc.epoch;    // (a strip of a grid mesh: two short runs of vertices a row apart)
        }
    }
    const uint64_t m = __ballot(culled), all = __ballot(chunk < cc.n_chunks);
    if (lane_id() == 0u && all) {
        atomicAdd(&cc.counts[0], (unsigned long long)__popcll(all));
        if (m) atomicAdd(&cc.counts[1], (unsigned long long)__popcll(m));
    }
}

// =================================================================================================
// read-back helpers and work counters (not on the timed path)
// =================================================================================================
__global__ __launch_bounds__(256) void split_vis_kernel(const unsigned long long* __restrict__ vis, size_t n, float* __restrict__ depth, uint32_t* __restrict__ tri)
{
    const size_t i = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const unsigned long long v = vis[i];
    if (depth) depth[i] = __uint_as_float((uint32_t)(v >> 32));
    if (tri) tri[i] = (uint32_t)v;
}

// marks the triangles that own a sample and counts pixels with any covered sample
__global__ __launch_bounds__(256) void stats_kernel(const unsigned long long* __restrict__ vis, size_t n_pixels, int S, uint32_t* __restrict__ tri_flag,
                                                    unsigned long long* __restrict__ shaded_pixels)
{
    const size_t i = (size_t)blockIdx.x * 256u + threadIdx.x;
    bool any = false;
    if (i < n_pixels)
        for (int s = 0; s < S; ++s) {
            const uint32_t id = (uint32_t)vis[i * S + s];
            if (id != kNoTri) { any = true; tri_flag[id] = 1u; }
        }
    const uint64_t m = __ballot(any);
    if (m && lane_id() == (uint32_t)(__ffsll((long long)m) - 1)) atomicAdd(shaded_pixels, (unsigned long long)__popcll(m));
}
__global__ __launch_bounds__(256) void count_flags_kernel(const uint32_t* __restrict__ flag, uint32_t n, unsigned long long* __restrict__ total)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    const bool f = i < n && flag[i] != 0u;
    const uint64_t m = __ballot(f);
    if (m && lane_id() == (uint32_t)(__ffsll((long long)m) - 1)) atomicAdd(total, (unsigned long long)__popcll(m));
}

// =================================================================================================
// host launchers
// =================================================================================================
static inline uint32_t div_up(uint64_t a, uint32_t b) { return (uint32_t)((a + b - 1) / b); }

void launch_split_positions(hipStream_t st, const void* aos, uint32_t n, float* px, float* py, float* pz)
{
    if (!n) return;
    split_positions_kernel<<<div_up(plane_stride(n), 256), 256, 0, st>>>((const float4*)aos, n, px, py, pz);
}

static uint32_t clear_group(const BlockFlags& flags, uint32_t n_views)
{
    return (uint32_t)std::min<size_t>(std::max<size_t>((size_t)flags.nbx * flags.nby * n_views / 4096, 1), 8);
}
static BeginWork make_begin_work(uint64_t* vis, const BlockFlags& flags, const FrameGeom& g, uint32_t view0, uint32_t n_views, void* zero16, size_t zero_bytes)
{
    BeginWork w;
    w.vis = (unsigned long long*)vis; w.flags = flags; w.rw = g.rw; w.rh = g.rh; w.S = g.samples; w.view0 = view0; w.n_views = n_views;
    w.zero4 = (uint4*)zero16; w.n_zero4 = zero_bytes / 16;
    w.clear_group = clear_group(flags, n_views);
    return w;
}

// workgroups for the start of a pass: a wave per group of flag blocks (clear_flagged_blocks), a work-item per 16 bytes to zero
static uint32_t begin_blocks(const BlockFlags& flags, uint32_t n_views, size_t zero_bytes)
{
    const uint32_t group = clear_group(flags, n_views);
    const size_t wave_turns = ((size_t)flags.nbx * flags.nby * n_views + group - 1) / group;
    return (uint32_t)std::min<size_t>(std::max<size_t>({(wave_turns + 3) / 4, (zero_bytes / 16 + 255) / 256, (size_t)1}), 65536);
}

void launch_begin_pass(hipStream_t st, uint64_t* vis, const BlockFlags& flags, const FrameGeom& g, uint32_t view0, uint32_t n_views,
                       void* zero16, size_t zero_bytes, const void* views_src, void* views_dst, size_t views_bytes)
{
    const uint32_t blocks = begin_blocks(flags, n_views, zero_bytes);
    begin_pass_kernel<<<blocks, 256, 0, st>>>(make_begin_work(vis, flags, g, view0, n_views, zero16, zero_bytes),
                                              (const uint4*)views_src, (uint4*)views_dst, (uint32_t)(views_bytes / 16));
}

bool launch_vertex_and_begin(hipStream_t st, const float* px, const float* py, const float* pz, uint32_t n, const ViewXform* host_views, uint32_t n_views,
                             ViewXform* views_dst, int W, int H, const ScreenPlanes& out, uint64_t* vis, const BlockFlags& flags, const FrameGeom& g, uint32_t view0,
                             void* zero16, size_t zero_bytes, const CullUse& cull)
{
    if (!n || !n_views || n_views > kMaxPackedViews) return false;
    ViewPack pack;
    std::memcpy(pack.v, host_views, (size_t)n_views * sizeof(ViewXform));
    // enough workgroups for the clearing part too when the mesh is small (a workgroup past the last vertex group only clears)
    const uint32_t for_clearing = begin_blocks(flags, n_views, zero_bytes);
    vertex_begin_kernel<<<std::max(div_up(n, kFusedVertexGroup * 256u), for_clearing), 256, 0, st>>>(px, py, pz, n, pack, views_dst, W, H, out,
                                                                       make_begin_work(vis, flags, g, view0, n_views, zero16, zero_bytes), cull);
    return true;
}

void launch_store_views(hipStream_t st, const ViewXform* host_views, uint32_t n, ViewXform* dst)
{
    ViewPack pack;
    n = std::min(n, kMaxPackedViews);
    std::memcpy(pack.v, host_views, (size_t)n * sizeof(ViewXform));
    if (n) store_views_kernel<<<1, 64, 0, st>>>(pack, n, dst);
}

void launch_vertex(hipStream_t st, const float* px, const float* py, const float* pz, uint32_t n, const ViewXform* views, uint32_t n_views,
                   int W, int H, const ScreenPlanes& out, ScreenVertex* debug_out, float* clip_out, uint32_t fill_blocks)
{
    if (!n || !n_views) return;
    // a workgroup takes up to kVertexViewsPerBlock views of the launch for its vertices (fewer if the mesh is too small to fill the
    // chip that way).  Not all of a long pass: every view adds two output streams (xy and z plane) per workgroup, and with several
    // dozen streams per workgroup the stores lose more in the memory system than reading the positions once more costs (the
    // 50 M-triangle batch, 34 views per pass: every view per workgroup 11.2 ms, sixteen 11.0 ms, eight 10.0 ms, four 9.9 ms).
    const uint32_t grid_x = div_up(n, kBatchVertexGroup * 256u);
    // (without a depth plane a view is one output stream, not two: the same batch, 34 views per pass, 8 views per workgroup 6.3 ms, 16 -- three
    // rows of 12 -- 5.9 ms, 32 -- two rows of 17 -- 6.3 ms)
    const uint32_t most_views = out.z ? kVertexViewsPerBlock : 2u * kVertexViewsPerBlock;
    uint32_t views_per_block = std::min<uint32_t>({std::max<uint32_t>((uint32_t)((uint64_t)n_views * grid_x / std::max(fill_blocks, 1u)), 1u), n_views, most_views});
    views_per_block = div_up(n_views, div_up(n_views, views_per_block));   // the same number of rows of workgroups, evenly filled (17 views: 6 + 6 + 5, not 8 + 8 + 1)
    if (grid_x < std::max(fill_blocks, 1u)) {   // a small mesh: one launch, a row of workgroups per group of views
        vertex_kernel<<<dim3(grid_x, div_up(n_views, views_per_block)), 256, 0, st>>>(px, py, pz, n, views, n_views, views_per_block, W, H, out, debug_out,
                                                                                       (float4*)clip_out);
        return;
    }
    // a large one: one launch per row (rows of one launch run into each other, which costs the 50 M-triangle batch 4 % of this kernel)
    for (uint32_t v = 0; v < n_views; v += views_per_block) {
        ScreenPlanes o = out; o.xy += (size_t)v * out.stride; if (o.z) o.z += (size_t)v * out.stride;
        const uint32_t nv = std::min(views_per_block, n_views - v);
        vertex_kernel<<<dim3(grid_x, 1), 256, 0, st>>>(px, py, pz, n, views + v, nv, nv, W, H, o, debug_out ? debug_out + (size_t)v * n : nullptr,
                                                       clip_out ? (float4*)clip_out + (size_t)v * n : nullptr);
    }
}

// workgroups of the kernels that walk the pending list in chunks: enough to fill the chip when the list is long, few
// enough to cost next to nothing when it is empty
static inline uint32_t chunk_grid(const BinBuffers& b) { return std::min<uint32_t>(std::max<uint32_t>(b.n_blocks, 1u), 1024u); }

void launch_setup(hipStream_t st, const uint32_t* idx, uint32_t n_tris, const ScreenPlanes& sp, const FrameGeom& g,
                  const BinBuffers& b, const ClipList& clip, uint32_t n_views, uint64_t* vis, const BlockFlags& flags, uint32_t view0, hipEvent_t between,
                  bool bin, uint32_t fill_blocks, const uint32_t* chunk_visible, uint32_t staged_min_blocks, const DepthSource* depth_source, const ChunkVerts* chunk_verts)
{
    if (!n_tris || !n_views) return;
    const ChunkVerts cv = chunk_verts ? *chunk_verts : ChunkVerts{};
    if (depth_source) {   // no depth plane: the view-loop form, whatever the number of views per workgroup
        const uint32_t gx = div_up(n_tris, kSetupBlockTris);
        const uint32_t vpb = std::min<uint32_t>(std::max<uint32_t>((uint32_t)((uint64_t)n_views * gx / std::max(fill_blocks, 1u)), 1u), n_views);
        const dim3 grid(gx, div_up(n_views, vpb));
        // (a workgroup with several views loads its chunk's vertices once per view, where the mesh has the lists for that)
        const bool local = chunk_verts && vpb >= 2u;
        if (local && g.samples == 1) setup_kernel<1, true, false, true, true><<<grid, 256, 0, st>>>(idx, n_tris, sp, g, b, clip, (unsigned long long*)vis, flags, view0, n_views, vpb, nullptr, *depth_source, cv);
        else if (local) setup_kernel<4, true, false, true, true><<<grid, 256, 0, st>>>(idx, n_tris, sp, g, b, clip, (unsigned long long*)vis, flags, view0, n_views, vpb, nullptr, *depth_source, cv);
        else if (g.samples == 1) setup_kernel<1, true, false, true><<<grid, 256, 0, st>>>(idx, n_tris, sp, g, b, clip, (unsigned long long*)vis, flags, view0, n_views, vpb, nullptr, *depth_source, cv);
        else setup_kernel<4, true, false, true><<<grid, 256, 0, st>>>(idx, n_tris, sp, g, b, clip, (unsigned long long*)vis, flags, view0, n_views, vpb, nullptr, *depth_source, cv);
        if (between) (void)hipEventRecord(between, st);
        return;           // (such a pass is a lean one: no binning)
    }
    // a workgroup takes every view of the pass for its triangles, unless the mesh is too small to fill the chip that way
    const uint32_t grid_x = div_up(n_tris, kSetupBlockTris);
    const uint32_t views_per_block = std::min<uint32_t>(std::max<uint32_t>((uint32_t)((uint64_t)n_views * grid_x / std::max(fill_blocks, 1u)), 1u), n_views);
    const dim3 grid(grid_x, div_up(n_views, views_per_block));
    if (views_per_block == 1) {
        // (the staged resolve where the launch is long enough to be bound by what it does, not by how long one workgroup takes)
        const bool staged = g.samples == 1 && (uint64_t)grid.x * grid.y >= staged_min_blocks;
        if (staged) setup_kernel<1, false, true><<<grid, 256, 0, st>>>(idx, n_tris, sp, g, b, clip, (unsigned long long*)vis, flags, view0, n_views, 1u, chunk_visible, DepthSource{}, cv);
        else if (g.samples == 1) setup_kernel<1, false, false><<<grid, 256, 0, st>>>(idx, n_tris, sp, g, b, clip, (unsigned long long*)vis, flags, view0, n_views, 1u, chunk_visible, DepthSource{}, cv);
        else setup_kernel<4, false, false><<<grid, 256, 0, st>>>(idx, n_tris, sp, g, b, clip, (unsigned long long*)vis, flags, view0, n_views, 1u, chunk_visible, DepthSource{}, cv);
    } else {
        if (chunk_verts && g.samples == 1) setup_kernel<1, true, false, false, true><<<grid, 256, 0, st>>>(idx, n_tris, sp, g, b, clip, (unsigned long long*)vis, flags, view0, n_views, views_per_block, nullptr, DepthSource{}, cv);
        else if (chunk_verts) setup_kernel<4, true, false, false, true><<<grid, 256, 0, st>>>(idx, n_tris, sp, g, b, clip, (unsigned long long*)vis, flags, view0, n_views, views_per_block, nullptr, DepthSource{}, cv);
        else if (g.samples == 1) setup_kernel<1, true, false><<<grid, 256, 0, st>>>(idx, n_tris, sp, g, b, clip, (unsigned long long*)vis, flags, view0, n_views, views_per_block, nullptr, DepthSource{}, cv);
        else setup_kernel<4, true, false><<<grid, 256, 0, st>>>(idx, n_tris, sp, g, b, clip, (unsigned long long*)vis, flags, view0, n_views, views_per_block, nullptr, DepthSource{}, cv);
    }
    if (between) (void)hipEventRecord(between, st);
    if (!bin) return;
    const dim3 bin_grid(chunk_grid(b), n_views);
    if (g.samples == 1) bin_kernel<1><<<bin_grid, 256, 0, st>>>(idx, sp, g, b);
    else bin_kernel<4><<<bin_grid, 256, 0, st>>>(idx, sp, g, b);
}

void launch_near_clip(hipStream_t st, const uint32_t* idx, uint32_t n_tris, const float* px, const float* py, const float* pz,
                      const ViewXform* views, const FrameGeom& g, const ClipList& clip, const BinBuffers& b, uint64_t* vis, const BlockFlags& flags,
                      uint32_t view0)
{
    if (!n_tris) return;
    // the list is empty in any ordinary frame; five hundred waves share the pieces of a long one
    if (g.samples == 1) near_clip_kernel<1><<<128, 256, 0, st>>>(idx, n_tris, px, py, pz, views, g, clip, b.counters, (unsigned long long*)vis, flags, view0);
    else near_clip_kernel<4><<<128, 256, 0, st>>>(idx, n_tris, px, py, pz, views, g, clip, b.counters, (unsigned long long*)vis, flags, view0);
}

void launch_direct_raster(hipStream_t st, const uint32_t* idx, uint32_t n_tris, const ScreenPlanes& sp, const FrameGeom& g, const BinBuffers& b,
                          uint64_t* vis, const BlockFlags& flags, uint32_t view0, uint32_t n_views, uint32_t expected_per_view)
{
    if (!n_tris || !n_views) return;
    // about one wave per pending triangle, going by what the previous draw left pending (a wave's triangle is a chain of three dependent
    // loads before any work: waves that take several in turn pay it several times); every wave takes triangles until the view's list is through
    const dim3 grid(std::min(std::max((expected_per_view + 3u) / 4u, 64u), 4096u), n_views);
    if (g.samples == 1) direct_raster_kernel<1><<<grid, 256, 0, st>>>(idx, n_tris, sp, g, b, (unsigned long long*)vis, flags, view0);
    else direct_raster_kernel<4><<<grid, 256, 0, st>>>(idx, n_tris, sp, g, b, (unsigned long long*)vis, flags, view0);
}

void launch_scan(hipStream_t st, const BinBuffers& b, uint32_t n_entries, uint32_t n_views, uint32_t* host_counters)
{
    const uint32_t chunks = div_up(n_entries, 8192u);   // (kScanChunk of the kernel)
    scan_kernel<<<chunks >= 2u && chunks <= 256u ? chunks : 1u, 1024, 0, st>>>(b.count_small, b.count_large, n_entries, b.tile_offset, b.counters, b.pending_total, n_views,
                                    b.micro_spread, host_counters);
}

void launch_fill(hipStream_t st, const FrameGeom& g, const BinBuffers& b, uint32_t n_tris, uint32_t n_views)
{
    if (!n_tris || !n_views) return;
    fill_kernel<<<dim3(chunk_grid(b), n_views), 256, 0, st>>>(g, b);
}

// work-items per workgroup of the one-sample rasteriser on 32 x 32 tiles (measured on the 64-view 1080p batch: 512 -> 0.64 ms, 256 -> 0.46,
// 128 -> 0.40, 64 -> 0.40: a tile's few dozen references keep two waves busy, more only wait at its barriers)
constexpr int kRasterTpb51 = 128;
void launch_raster(hipStream_t st, const uint32_t* idx, uint32_t n_tris, const ScreenPlanes& sp, const FrameGeom& g, const BinBuffers& b,
                   uint64_t* vis, const BlockFlags& flags, uint32_t view0, uint32_t n_views)
{
    if (!n_views) return;
    const dim3 grid((uint32_t)(g.tiles_x * g.tiles_y), n_views);
    unsigned long long* v = (unsigned long long*)vis;
    if (g.samples == 1 && g.tile_log2 == 6) raster_kernel<6, 1, 1024><<<grid, 1024, 0, st>>>(idx, n_tris, sp, g, b.tile_offset, b.refs, b.ref_capacity, v, flags, view0);
    else if (g.samples == 1 && g.tile_log2 == 5) raster_kernel<5, 1, kRasterTpb51><<<grid, kRasterTpb51, 0, st>>>(idx, n_tris, sp, g, b.tile_offset, b.refs, b.ref_capacity, v, flags, view0);
    // (512 work-items, not 1024: at 1024 the kernel is capped at 128 registers and spills 500 bytes per work-item; measured 133 vs 386 us
    // on the reference's own 800x600 4-sample frame)
    else if (g.samples == 4 && g.tile_log2 == 5) raster_kernel<5, 4, 512><<<grid, 512, 0, st>>>(idx, n_tris, sp, g, b.tile_offset, b.refs, b.ref_capacity, v, flags, view0);
    else if (g.samples == 4 && g.tile_log2 == 4) raster_kernel<4, 4, 256><<<grid, 256, 0, st>>>(idx, n_tris, sp, g, b.tile_offset, b.refs, b.ref_capacity, v, flags, view0);
}

void launch_chunk_bounds(hipStream_t st, const uint32_t* idx, uint32_t n_tris, const void* verts_aos, float4* sphere, uint4* vrange)
{
    if (!n_tris) return;
    chunk_bounds_kernel<<<div_up(n_tris, kChunkCullTris), 256, 0, st>>>(idx, n_tris, (const float4*)verts_aos, sphere, vrange);
}

void launch_chunk_verts(hipStream_t st, const uint32_t* idx, uint32_t n_tris, const uint4* vrange, uint32_t* list, uint32_t* count, uint32_t* local, uint32_t* unfit)
{
    if (!n_tris) return;
    chunk_verts_kernel<<<div_up(n_tris, kChunkCullTris), 256, 0, st>>>(idx, n_tris, vrange, list, count, local, unfit);
}

void launch_chunk_cull(hipStream_t st, const ChunkCull& cc, const CullView& v, const FrameGeom& g)
{
    if (!cc.n_chunks) return;
    chunk_cull_kernel<<<div_up(cc.n_chunks, 256u), 256, 0, st>>>(cc, v, g);
}

uint32_t setup_block_triangles() { return kSetupBlockTris; }
uint32_t chunk_triangles() { return kBlockTris; }
uint32_t setup_block_records() { return kBlockRecords; }

bool raster_config_supported(int samples, int tile_log2)
{
    return (samples == 1 && (tile_log2 == 5 || tile_log2 == 6)) || (samples == 4 && (tile_log2 == 4 || tile_log2 == 5));
}

void launch_fragment(hipStream_t st, const FragLaunch& f)
{
    if (!f.n_views) return;
    FragParams p;
    p.idx = f.idx; p.verts = (const float4*)f.verts_aos; p.views = f.views; p.vis = (const unsigned long long*)f.vis; p.color = f.color;
    p.g = f.g; p.albedo = f.albedo; p.metallic = f.metallic; p.roughness = f.roughness; p.normal = f.normal;
    p.tab.decode = f.decode; p.tab.thresh = f.thresh; p.view0 = f.view0; p.flags = f.flags; p.report = f.report;
    const TexDesc &m = f.metallic, &r = f.roughness;
    const bool same_sampling = m.w == r.w && m.h == r.h && m.levels == r.levels && m.min_lod == r.min_lod && m.max_lod == r.max_lod && m.bias == r.bias && m.max_aniso == r.max_aniso &&
                               std::memcmp(m.level_off, r.level_off, sizeof m.level_off) == 0;
    p.packed_mr = same_sampling ? f.packed_mr : nullptr;
    p.lod_shared = 0u; p.lod_shift[0] = p.lod_shift[1] = 0.0f;
    const TexDesc* other[2] = {&f.metallic, &f.roughness};
    for (int k = 0; k < 2; ++k)
        if ((int64_t)other[k]->w * f.albedo.h == (int64_t)other[k]->h * f.albedo.w) {
            p.lod_shared |= 1u << k;
            p.lod_shift[k] = (float)std::log2((double)other[k]->w / (double)f.albedo.w);
        }
    const dim3 grid(div_up((uint32_t)f.g.rw, kFragBlockW), div_up((uint32_t)f.g.rh, kFragBlockH), f.n_views), block(kFragBlockW, kFragBlockH);
    // the anisotropic footprint is compiled in only when a bound sampler asks for it
    const bool aniso = f.albedo.max_aniso > 1.0f || f.metallic.max_aniso > 1.0f || f.roughness.max_aniso > 1.0f ||
                       (f.normal_mapping && f.normal.max_aniso > 1.0f);
    const bool report = f.report.host_counters != nullptr;
#define SR_LAUNCH_FRAGMENT(S, A, I)                                                          \
    do {                                                                                      \
        if (report) fragment_report_kernel<S, A, I><<<grid, block, 0, st>>>(p);               \
        else fragment_kernel<S, A, I, false><<<grid, block, 0, st>>>(p);                      \
    } while (0)
    if (f.normal_mapping) {   // (never a lean draw; the general model-matrix form)
        if (f.g.samples == 1) { if (aniso) fragment_kernel<1, true, false, true><<<grid, block, 0, st>>>(p); else fragment_kernel<1, false, false, true><<<grid, block, 0, st>>>(p); }
        else { if (aniso) fragment_kernel<4, true, false, true><<<grid, block, 0, st>>>(p); else fragment_kernel<4, false, false, true><<<grid, block, 0, st>>>(p); }
    } else if (f.g.samples == 1) {
        if (aniso && f.identity_model) SR_LAUNCH_FRAGMENT(1, true, true);
        else if (aniso) SR_LAUNCH_FRAGMENT(1, true, false);
        else if (f.identity_model) SR_LAUNCH_FRAGMENT(1, false, true);
        else SR_LAUNCH_FRAGMENT(1, false, false);
    } else {
        if (aniso && f.identity_model) SR_LAUNCH_FRAGMENT(4, true, true);
        else if (aniso) SR_LAUNCH_FRAGMENT(4, true, false);
        else if (f.identity_model) SR_LAUNCH_FRAGMENT(4, false, true);
        else SR_LAUNCH_FRAGMENT(4, false, false);
    }
#undef SR_LAUNCH_FRAGMENT
}

void launch_pack_red_channels(hipStream_t st, const uint32_t* a, const uint32_t* b, size_t n_texels, uint32_t* out)
{
    if (n_texels) pack_red_channels_kernel<<<(uint32_t)std::min<size_t>((n_texels + 255) / 256, 1u << 20), 256, 0, st>>>(a, b, n_texels, out);
}

void launch_mip_blit(hipStream_t st, const uint32_t* src, int sw, int sh, uint32_t* dst, int dw, int dh, const float* decode, const float* thresh)
{
    ShadeTables tab; tab.decode = decode; tab.thresh = thresh;
    mip_blit_kernel<<<dim3(div_up((uint32_t)dw, kMipBlockW), div_up((uint32_t)dh, kMipBlockH)), kMipBlockW * kMipBlockH, 0, st>>>(src, sw, sh, dst, dw, dh, tab);
}

void launch_split_vis(hipStream_t st, const uint64_t* vis, size_t n, float* depth, uint32_t* tri)
{
    if (!n) return;
    split_vis_kernel<<<div_up(n, 256), 256, 0, st>>>((const unsigned long long*)vis, n, depth, tri);
}

void launch_stats(hipStream_t st, const uint64_t* vis, size_t n_pixels, int S, uint32_t* tri_flag, uint32_t n_tris, uint64_t* shaded, uint64_t* visible)
{
    if (n_pixels) stats_kernel<<<div_up(n_pixels, 256), 256, 0, st>>>((const unsigned long long*)vis, n_pixels, S, tri_flag, (unsigned long long*)shaded);
    if (n_tris) count_flags_kernel<<<div_up(n_tris, 256), 256, 0, st>>>(tri_flag, n_tris, (unsigned long long*)visible);
}

}  // namespace sr
