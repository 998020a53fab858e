// C ABI of the rasteriser (include/softraster.h): handle management, uploads, the draw sequence
// and read-backs.  Thin by design: everything per-vertex, per-triangle or per-pixel runs in the
// kernels of sr_kernels.hip; there is no CPU rendering path in this library.
#include "../../include/softraster.h"
#include "sr_assets.h"
#include "sr_launch.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <string>
#include <vector>

using namespace sr;

// Scratch of sr_draw_no_alloc (include/softraster.h): one array in the device image of this library instead of
// allocations.  Each GPU the library runs on holds its own copy at its own address; sr_device_create looks that address up
// once (hipGetSymbolAddress, with the device current) and from then on only the pointer is used: in host code the
// array's name is a stand-in for the symbol, not a device address.  Every draw through that entry point on one GPU
// shares the array and finds the previous draw's contents in it; the begin kernel zeroes what must start at zero
// (counters, per-tile counts), everything else is written before it is read.
constexpr size_t kStaticScratchBytes = (size_t)64 << 20;
constexpr uint32_t kStaticScratchViews = kMaxPackedViews;     // views of a draw whose transforms travel as kernel arguments
static __device__ __align__(256) unsigned char scratch[kStaticScratchBytes];


namespace {

thread_local std::string g_last_error;

int fail(int status, const std::string& msg)
{
    g_last_error = msg;
    return status;
}

#define SR_HIP(expr)                                                                                   \
    do {                                                                                               \
        hipError_t e__ = (expr);                                                                       \
        if (e__ != hipSuccess)                                                                         \
            return fail(e__ == hipErrorOutOfMemory ? SR_ERR_OUT_OF_MEMORY : SR_ERR_DEVICE,             \
                        std::string(#expr) + ": " + hipGetErrorString(e__));                           \
    } while (0)

struct DeviceBuffer {
    void* ptr = nullptr;
    size_t bytes = 0;
    hipError_t ensure(size_t need)
    {
        if (need <= bytes) return hipSuccess;
        if (ptr) { hipError_t e = hipFree(ptr); ptr = nullptr; bytes = 0; if (e != hipSuccess) return e; }
        hipError_t e = hipMalloc(&ptr, need);
        if (e == hipSuccess) bytes = need;
        return e;
    }
    void release() { if (ptr) (void)hipFree(ptr); ptr = nullptr; bytes = 0; }
};

constexpr int kNumStageEvents = 9;   // start, clear, vertex, setup, bin, scan, fill, raster, fragment
constexpr int kCountersPerPass = 8;  // words per pass the scan kernel writes for the host: total refs, -, unprojectable, binned, micro, -, clipped, -

}  // namespace

struct sr_device_t {
    int ordinal = 0;
    hipStream_t stream = nullptr;
    float* d_decode = nullptr;   // 256
    float* d_thresh = nullptr;   // 255 (+1 pad)
    // scratch reused by every draw on this device (in-order on `stream`)
    DeviceBuffer screen, tile_count, tile_offset, records, pending, refs, views;
    DeviceBuffer clip;           // entries of the near-plane clip list: capacity x 8 bytes
    uint32_t clip_capacity = 65536;   // triangles per pass that can be clipped; more are dropped and counted
    uint32_t fill_blocks = kFillBlocks;   // see sr_launch.h
    uint32_t staged_min_blocks = kStagedResolveMinBlocks;   // see sr_launch.h
    bool normal_mapping = false;          // sr_device_set_normal_mapping: applies to the draws issued after it
    // pinned staging for the per-view transforms: a small ring, so that back-to-back draws do not
    // wait for each other on the host
    struct Staging { ViewXform* h = nullptr; void* d = nullptr; size_t cap = 0 /* bytes */; hipEvent_t ev = nullptr; bool in_flight = false; };   // h: pinned + mapped, d: its device address
    Staging staging[4];
    unsigned staging_next = 0;
    uint64_t ref_capacity = 0;
    size_t scratch_budget = (size_t)16 << 30;
    int tile_log2_override = 0;
    bool profiling = false;
    std::vector<hipEvent_t> events;   // kNumStageEvents per pass of the last profiled draw
    size_t events_used = 0;
    std::vector<sr_target_t*> targets;   // live targets of this device (they may hold a reference to a mesh and textures)
    void* static_scratch = nullptr;      // this GPU's copy of the array `scratch` (sr_api.hip), looked up at creation
    sr_draw_counts counts{};             // sr_device_get_draw_counts
    // chunk culling (sr_device.h): per-draw flags, allocated on first use; the stamps compare against an epoch that goes up with
    // every culled draw, so they never need clearing
    DeviceBuffer cull_visible, cull_stamp;
    unsigned long long* d_cull_counts = nullptr;   // [2] chunks tested / culled, cumulative
    uint32_t cull_epoch = 0;
    uint32_t cull_min_chunks = 8;                  // meshes of fewer chunks are not worth the extra launch (SR_CULL_MIN_CHUNKS)
    bool cull_enabled = true;                      // SR_NO_CULL=1 turns it off
    bool chunk_vertex_lists = true;                 // meshes get chunk vertex lists for batches of views (SR_NO_CHUNK_VERTS=1: none)
    bool recompute_depth = true;                    // lean draws of view batches store no depth plane (SR_STORE_DEPTH=1: always store it)
    // Pairs of maps of the same size whose red channels have been packed into one chain for the shading pass (the reference's metallic and
    // roughness maps: eight texel loads per pixel instead of sixteen).  Made by the first sr_draw that uses the pair; gone with either map.
    struct PackedPair { const sr_texture_t* a; const sr_texture_t* b; uint32_t* d_chain; };
    std::vector<PackedPair> packed_pairs;
    bool pack_pairs = true;                         // (SR_NO_PACKED_MAPS=1: never)
    uint32_t direct_max_triangles = 500000;         // pending triangles of a draw (all its views) up to which it skips the tile path (SR_DIRECT_MAX_TRIANGLES; 0: never)
};

struct sr_mesh_t {
    void* d_aos = nullptr;
    float *px = nullptr, *py = nullptr, *pz = nullptr;
    uint32_t* d_idx = nullptr;
    uint32_t n_verts = 0, n_tris = 0;
    float4* d_chunk_sphere = nullptr;   // chunk culling: bounds of every chunk of setup_block_triangles() triangles ([chunks] spheres, then
    uint4* d_chunk_vrange = nullptr;    // [chunks] vertex ranges in the same allocation)
    uint32_t* d_chunk_verts = nullptr;  // chunk vertex lists (one allocation: lists, counts, the triangles' list positions), or null:
    ChunkVerts chunk_verts{};           // a chunk with too many or too widely spread vertices, or lists switched off
    uint32_t n_chunks = 0;
};

struct sr_texture_t {
    uint32_t* d_chain = nullptr;
    int w = 0, h = 0, levels = 0;
    sr_sampler_desc samp{};
    uint32_t level_off[kMaxMipLevels] = {};
    size_t total_texels = 0;
};

struct sr_model_t {
    std::vector<sr_assets::HostMesh> host;   // as read from the file
    std::vector<sr_mesh> meshes;             // uploaded copies (sr_model_load), same order
};

struct sr_target_t {
    FrameGeom g{};
    uint32_t n_views = 0;
    uint64_t* d_vis = nullptr;
    uint32_t* d_color_own = nullptr;
    uint32_t* d_color = nullptr;
    BlockFlags flags{nullptr, nullptr, 0, 0};   // per block of every view: visibility words written / colour not clear (sr_device.h); all set at creation
    // the last draw, kept so that it can be re-issued if the reference list was too small; mesh and textures are
    // referenced only while check_pending is set (destroying one of them resolves the draw first)
    bool drawn = false, check_pending = false;
    sr_mesh mesh = nullptr;
    sr_texture tex[4] = {nullptr, nullptr, nullptr, nullptr};
    uint32_t mesh_verts = 0, mesh_tris = 0;   // of the last draw's mesh (it may be gone when the counters are asked for)
    std::vector<ViewXform> views;
    uint32_t passes = 0;
    uint32_t* h_counters = nullptr; void* hd_counters = nullptr; size_t h_counters_cap = 0;   // pinned + mapped, [passes][8], written by the scan kernel of the last draw
    uint64_t capacity_at_draw = 0;
    uint32_t clip_capacity_at_draw = 0;
    uint64_t total_refs = 0, unprojectable = 0, binned = 0, micro = 0, clipped = 0;
    // A frame of nothing but micro-triangles leaves the binning, clipping, scan, fill and tile kernels without work, yet
    // each costs a launch and a chain of dependent loads (30 us of a 240 us frame).  When the previous draw of this
    // target (same mesh) left them nothing, the next one is issued without them ("lean"): set-up is followed by the
    // one-workgroup kernel that reports the counters.  If the counters then say that there was work after all, the
    // draw is issued again in full before anything of it can be observed (finish_target), like a draw whose tile
    // reference list overflowed.
    bool lean = false;
    // A frame that leaves a FEW triangles pending (the reference's own 800 x 600 frame) is issued the same way plus one kernel that
    // resolves the pending triangles straight into the visibility buffer (`direct`), and the near-plane clipper if the previous
    // draw listed anything (`direct_clip`): two or three launches in place of the tile path's five.  Falls back like a lean draw.
    bool direct = false, direct_clip = false;
    uint32_t direct_expected = 0;  // pending triangles per view of the previous draw: sizes the direct rasteriser's launch
    bool direct_banned = false;    // a direct draw of this mesh met a triangle too large for it: the tile path until the mesh changes
    bool normal_mapping = false;   // as the device said when the draw was issued (a redraw repeats it)
    bool no_alloc = false;      // the last draw came through sr_draw_no_alloc: if it is issued again, then the same way
};

namespace {

double srgb_decode_d(double x) { return x <= 0.04045 ? x / 12.92 : std::pow((x + 0.055) / 1.055, 2.4); }

void mat4_mul_mat4_host(const float* A, const float* B, float* R)
{
    for (int c = 0; c < 4; ++c) mat4_mul_vec4(A, B[4 * c], B[4 * c + 1], B[4 * c + 2], B[4 * c + 3], R + 4 * c);
}

// rows x, y, w of proj * view * model in double precision, rounded once (chunk culling, sr_device.h)
CullView make_cull_view(const ViewXform& x)
{
    CullView v{};
    const int rows[3] = {0, 1, 3};
    float* out[3] = {v.ax, v.ay, v.aw};
    float* norm[3] = {&v.nx, &v.ny, &v.nw};
    double scale = 0.0;
    for (int k = 0; k < 3; ++k) {
        double a[4], n2 = 0.0;
        for (int c = 0; c < 4; ++c) {
            a[c] = 0.0;
            for (int j = 0; j < 4; ++j) a[c] += (double)x.pv[4 * j + rows[k]] * (double)x.model[4 * c + j];   // column-major: element (r, c) at [4 * c + r]
            out[k][c] = (float)a[c];
            scale += std::fabs(a[c]);
            for (int j = 0; j < 4; ++j) scale += std::fabs((double)x.pv[4 * j + rows[k]] * (double)x.model[4 * c + j]) * 0.25;   // (the terms' sizes, not only their sum's)
        }
        for (int c = 0; c < 3; ++c) n2 += a[c] * a[c];
        *norm[k] = (float)(std::sqrt(n2) * 1.000001);
    }
    v.scale = (float)scale;
    return v;
}

// the chain that holds the red channels of a and b (same size), made now unless the caller may not allocate; null if there is none
const uint32_t* packed_pair(sr_device_t* dev, const sr_texture_t* a, const sr_texture_t* b, bool may_allocate)
{
    if (!dev->pack_pairs || !a || !b || a->w != b->w || a->h != b->h || a->levels != b->levels || a->total_texels != b->total_texels ||
        std::memcmp(a->level_off, b->level_off, sizeof a->level_off) != 0)
        return nullptr;
    for (const auto& p : dev->packed_pairs) if (p.a == a && p.b == b) return p.d_chain;
    if (!may_allocate) return nullptr;
    uint32_t* chain = nullptr;
    if (hipMalloc((void**)&chain, std::max<size_t>(a->total_texels, 1) * 4) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    launch_pack_red_channels(dev->stream, a->d_chain, b->d_chain, a->total_texels, chain);
    dev->packed_pairs.push_back({a, b, chain});
    return chain;
}

TexDesc make_tex_desc(const sr_texture_t* t)
{
    TexDesc d{};
    d.texels = t->d_chain; d.w = t->w; d.h = t->h; d.levels = t->levels;
    d.log2_w = d.log2_h = -1;
    if ((t->w & (t->w - 1)) == 0 && (t->h & (t->h - 1)) == 0) {
        d.log2_w = d.log2_h = 0;
        while ((1 << d.log2_w) < t->w) ++d.log2_w;
        while ((1 << d.log2_h) < t->h) ++d.log2_h;
    }
    d.min_lod = t->samp.min_lod;
    d.max_lod = t->samp.max_lod < 0.0f ? (float)t->levels : t->samp.max_lod;
    d.bias = t->samp.mip_lod_bias;
    d.max_aniso = t->samp.max_anisotropy;
    std::memcpy(d.level_off, t->level_off, sizeof d.level_off);
    return d;
}

int choose_tile_log2(const sr_device_t* dev, int samples, int rect_w, int rect_h)
{
    if (dev->tile_log2_override && raster_config_supported(samples, dev->tile_log2_override)) return dev->tile_log2_override;
    // 32 x 32 pixels at one sample and at four (the 64-view 1080p batch rasterises in 0.46 instead of 0.62 ms with them); at one sample
    // 64 x 64 for a rectangle of more than 256 of the smaller tiles along an axis (tile coordinates are eight bits)
    if (samples == 1 && (rect_w > (256 << 5) || rect_h > (256 << 5))) return 6;
    return 5;
}

// bytes per vertex and view of the window-space planes of a pass: int2 xy, float z
constexpr size_t kPlaneBytesPerVertex = sizeof(int2) + sizeof(float);

// per-view scratch bytes of a pass.  A lean draw launches no binning kernel and so needs no binning records, by far the largest
// buffer (48 of 64 bytes per triangle): its passes take several times as many views, and a pass reads the mesh once.
size_t per_view_bytes(const sr_mesh_t* m, const FrameGeom& g, bool lean, bool depth_plane)
{
    return (size_t)plane_stride(m->n_verts) * (depth_plane ? kPlaneBytesPerVertex : sizeof(int2)) + ((size_t)m->n_tris + 1024) * ((lean ? 0 : 4 * sizeof(BinRecord)) + 4) + (size_t)m->n_tris / 64 +
           (size_t)g.tiles_x * g.tiles_y * 16 + 64 + kPendingTotalStride * 4;
}

uint32_t views_per_pass(const sr_device_t* dev, const sr_mesh_t* m, const FrameGeom& g, uint32_t n_views, bool lean, bool depth_plane)
{
    size_t pv = per_view_bytes(m, g, lean, depth_plane);
    uint64_t by_mem = std::max<uint64_t>(1, dev->scratch_budget / std::max<size_t>(pv, 1));
    uint64_t by_refs = std::max<uint64_t>(1, (uint64_t)0x7FFFFFFF / std::max<uint32_t>(m->n_tris, 1));
    uint64_t v = std::min<uint64_t>({(uint64_t)n_views, by_mem, by_refs, (uint64_t)65535});
    return (uint32_t)std::max<uint64_t>(v, 1);
}

// where the per-draw scratch of a pass lives
struct PassScratch {
    void* screen = nullptr; uint32_t* tile_count = nullptr; uint32_t* pending = nullptr; uint32_t* tile_offset = nullptr;
    BinRecord* records = nullptr; uint32_t* refs = nullptr; uint64_t ref_capacity = 0; uint2* clip = nullptr; uint32_t clip_capacity = 0;
    ViewXform* views = nullptr;
};

// issues every kernel of one draw (all passes) on the device's stream
int issue_draw(sr_device_t* dev, sr_target_t* tg)
{
    const sr_mesh_t* m = tg->mesh;
    const FrameGeom& g = tg->g;
    const uint32_t n_views = (uint32_t)tg->views.size();
    const bool no_alloc = tg->no_alloc;
    // A lean draw of a batch of views (more than travel as kernel arguments) stores no depth plane: of its kernels only set-up reads
    // window-space depths, and only for the triangles that reach its second phase; it works them out again from the positions.
    const bool depth_plane = !(tg->lean && !tg->direct && !no_alloc && dev->recompute_depth && n_views > kMaxPackedViews);
    const uint32_t vpp = no_alloc ? n_views : views_per_pass(dev, m, g, n_views, tg->lean, depth_plane);
    const uint32_t passes = (n_views + vpp - 1) / vpp;
    const uint32_t n_tiles = (uint32_t)(g.tiles_x * g.tiles_y);
    hipStream_t st = dev->stream;

    const uint32_t bt = chunk_triangles();
    const uint32_t n_blocks = std::max((m->n_tris + bt - 1u) / bt, 1u);            // chunks of a view's pending list
    const uint32_t pending_stride = n_blocks * bt;
    const size_t count_words = (size_t)vpp * n_tiles * 3;
    // layout of the binning state: [count_small | count_large | cursor_large | pending_total per view | counters of this pass |
    // spread micro-triangle counters] (zeroed at the start of every pass, padded to 16 bytes) [block_count: written by the binning kernel]
    const size_t zero_words = (count_words + (size_t)vpp * kPendingTotalStride + kNumPassCounters + kMicroSpread * kMicroSpreadStride + 3) & ~(size_t)3;
    const size_t view_bytes = (size_t)n_views * sizeof(ViewXform);
    const uint32_t v_stride = std::max<uint32_t>(plane_stride(m->n_verts), 4);
    const size_t bytes_screen = (size_t)vpp * v_stride * (depth_plane ? kPlaneBytesPerVertex : sizeof(int2));   // the int2 plane of every view of the pass, then the float plane
    const size_t bytes_tile_count = (zero_words + (size_t)vpp * n_blocks) * 4;
    const size_t bytes_pending = (size_t)vpp * pending_stride * 4;
    const size_t bytes_tile_offset = ((size_t)vpp * n_tiles + 1) * 4;
    const size_t bytes_records = tg->lean && !no_alloc ? 0 : (size_t)vpp * n_blocks * setup_block_records() * sizeof(BinRecord);   // (a lean draw has no binning kernel)
    PassScratch ps;
    sr_device_t::Staging* staging = nullptr;
    if (no_alloc) {
        // carve the static array; the tile reference list takes what is left
        size_t used = 0;
        auto take = [&](size_t bytes) { void* p = (char*)dev->static_scratch + used; used += (bytes + 255) & ~(size_t)255; return p; };
        ps.screen = take(bytes_screen); ps.tile_count = (uint32_t*)take(bytes_tile_count); ps.pending = (uint32_t*)take(bytes_pending);
        ps.tile_offset = (uint32_t*)take(bytes_tile_offset); ps.records = (BinRecord*)take(bytes_records);
        ps.views = (ViewXform*)take(view_bytes);
        ps.clip_capacity = 1024; ps.clip = (uint2*)take((size_t)ps.clip_capacity * sizeof(uint2));
        const size_t min_refs = (size_t)1 << 16;
        if (used + min_refs * 4 > kStaticScratchBytes)
            return fail(SR_ERR_UNSUPPORTED, "sr_draw_no_alloc: this draw needs " + std::to_string(used + min_refs * 4) + " bytes of scratch, the static array holds " +
                                                std::to_string(kStaticScratchBytes) + " (fewer triangles, vertices or views, or sr_draw)");
        ps.ref_capacity = std::min<uint64_t>((kStaticScratchBytes - used) / 4, 0xFFFFFFFFull);
        ps.refs = (uint32_t*)((char*)dev->static_scratch + used);
        dev->events_used = 0;
    } else {
        if (dev->ref_capacity == 0) dev->ref_capacity = std::max<uint64_t>((uint64_t)m->n_tris * vpp, (uint64_t)1 << 20);
        dev->ref_capacity = std::min<uint64_t>(dev->ref_capacity, 0xFFFFFFFFull);
        SR_HIP(dev->screen.ensure(bytes_screen));
        SR_HIP(dev->tile_count.ensure(bytes_tile_count));
        SR_HIP(dev->pending.ensure(bytes_pending));
        SR_HIP(dev->tile_offset.ensure(bytes_tile_offset));
        SR_HIP(dev->records.ensure(bytes_records));
        SR_HIP(dev->refs.ensure((size_t)dev->ref_capacity * 4));
        SR_HIP(dev->clip.ensure((size_t)dev->clip_capacity * sizeof(uint2)));
        SR_HIP(dev->views.ensure(view_bytes));
        ps.screen = dev->screen.ptr; ps.tile_count = (uint32_t*)dev->tile_count.ptr; ps.pending = (uint32_t*)dev->pending.ptr;
        ps.tile_offset = (uint32_t*)dev->tile_offset.ptr; ps.records = (BinRecord*)dev->records.ptr; ps.refs = (uint32_t*)dev->refs.ptr;
        ps.ref_capacity = dev->ref_capacity; ps.clip = (uint2*)dev->clip.ptr; ps.clip_capacity = dev->clip_capacity; ps.views = (ViewXform*)dev->views.ptr;
        if (n_views > kMaxPackedViews) {    // too many transforms for kernel arguments: through pinned staging memory
            sr_device_t::Staging& sg = dev->staging[dev->staging_next++ & 3u];
            staging = &sg;
            if (!sg.ev) SR_HIP(hipEventCreateWithFlags(&sg.ev, hipEventDisableTiming));
            if (sg.in_flight) { SR_HIP(hipEventSynchronize(sg.ev)); sg.in_flight = false; }
            if (sg.cap < view_bytes) {
                if (sg.h) SR_HIP(hipHostFree(sg.h));
                sg.h = nullptr; sg.cap = 0;
                SR_HIP(hipHostMalloc((void**)&sg.h, view_bytes, hipHostMallocMapped));
                SR_HIP(hipHostGetDevicePointer(&sg.d, sg.h, 0));
                sg.cap = view_bytes;
            }
            std::memcpy(sg.h, tg->views.data(), view_bytes);
        }
        if (tg->h_counters_cap < passes) {
            SR_HIP(hipStreamSynchronize(st));
            if (tg->h_counters) SR_HIP(hipHostFree(tg->h_counters));
            tg->h_counters = nullptr; tg->h_counters_cap = 0;
            SR_HIP(hipHostMalloc((void**)&tg->h_counters, (size_t)passes * kCountersPerPass * 4, hipHostMallocMapped));
            SR_HIP(hipHostGetDevicePointer(&tg->hd_counters, tg->h_counters, 0));
            tg->h_counters_cap = passes;
        }
        if (dev->profiling) {
            while (dev->events.size() < (size_t)passes * kNumStageEvents) {
                hipEvent_t e; SR_HIP(hipEventCreate(&e));
                dev->events.push_back(e);
            }
            dev->events_used = (size_t)passes * kNumStageEvents;
        } else dev->events_used = 0;
    }
    const bool profiling = dev->profiling && !no_alloc;

    ScreenPlanes sp{};
    sp.xy = (int2*)ps.screen;
    sp.z = depth_plane ? (float*)((char*)ps.screen + (size_t)vpp * v_stride * sizeof(int2)) : nullptr;
    sp.stride = v_stride;
    BinBuffers b{};
    b.count_small = ps.tile_count;
    b.count_large = b.count_small + (size_t)vpp * n_tiles;
    b.cursor_large = b.count_large + (size_t)vpp * n_tiles;
    b.pending_total = b.count_small + count_words;
    b.counters = b.pending_total + (size_t)vpp * kPendingTotalStride;
    b.block_count = b.count_small + zero_words;
    b.micro_spread = b.counters + kNumPassCounters;
    b.pending = ps.pending;
    b.pending_stride = pending_stride;
    b.n_blocks = n_blocks;
    b.tile_offset = ps.tile_offset;
    b.refs = ps.refs;
    b.ref_capacity = (uint32_t)ps.ref_capacity;
    b.records = ps.records;
    ClipList clip{ps.clip, ps.clip_capacity};
    // the transforms of a draw of a few views travel as kernel arguments (of the kernel that starts the pass, or, for a mesh
    // without vertices, of a kernel of their own): no staging memory
    const bool views_by_argument = n_views <= kMaxPackedViews;
    if (views_by_argument && m->n_verts == 0) launch_store_views(st, tg->views.data(), n_views, ps.views);

    const sr_texture_t* albedo = tg->tex[0];
    const sr_texture_t* metallic = tg->tex[2];
    const sr_texture_t* roughness = tg->tex[3];

    // Chunk culling: a one-view draw into a target restricted to a rectangle (a band of a frame split over several GPUs).  One small
    // launch decides which chunks of the mesh can touch the rectangle; the vertex stage then skips the groups of vertices no such
    // chunk uses and set-up's workgroups for the other chunks leave at once.
    CullUse cull_use{nullptr, nullptr, 0u};
    if (dev->cull_enabled && !no_alloc && n_views == 1 && views_by_argument && m->n_verts && m->d_chunk_sphere && m->n_chunks >= dev->cull_min_chunks &&
        (g.rw < g.W || g.rh < g.H)) {
        const size_t stamp_words = 1 + (size_t)(v_stride + kCullVertexGroup - 1) / kCullVertexGroup;
        const size_t had = dev->cull_stamp.bytes;
        SR_HIP(dev->cull_visible.ensure((size_t)m->n_chunks * 4));
        SR_HIP(dev->cull_stamp.ensure(stamp_words * 4));
        if (dev->cull_stamp.bytes != had || dev->cull_epoch == 0xFFFFFFFFu) {   // new memory (or the epoch about to wrap): no stamp may equal a coming epoch
            SR_HIP(hipMemsetAsync(dev->cull_stamp.ptr, 0, dev->cull_stamp.bytes, st));
            dev->cull_epoch = 0;
        }
        ChunkCull cc{};
        cc.sphere = m->d_chunk_sphere; cc.vrange = m->d_chunk_vrange; cc.n_chunks = m->n_chunks;
        cc.visible = (uint32_t*)dev->cull_visible.ptr; cc.stamp = (uint32_t*)dev->cull_stamp.ptr; cc.epoch = ++dev->cull_epoch; cc.counts = dev->d_cull_counts;
        launch_chunk_cull(st, cc, make_cull_view(tg->views[0]), g);
        cull_use = CullUse{cc.visible, cc.stamp, cc.epoch};
    }

    for (uint32_t pass = 0; pass < passes; ++pass) {
        const uint32_t v0 = pass * vpp, nv = std::min(vpp, n_views - v0);
        const ViewXform* d_views = ps.views + v0;
        hipEvent_t* ev = profiling ? &dev->events[(size_t)pass * kNumStageEvents] : nullptr;
        if (ev) SR_HIP(hipEventRecord(ev[0], st));
        // one launch starts the pass and runs the vertex stage when the transforms fit its arguments (then n_views == nv: one pass)
        const bool fused = views_by_argument &&
                           launch_vertex_and_begin(st, m->px, m->py, m->pz, m->n_verts, tg->views.data() + v0, nv, ps.views + v0, g.W, g.H, sp, tg->d_vis, tg->flags, g, v0,
                                                   b.count_small, zero_words * 4, cull_use);
        if (!fused) {
            const bool copy_views = pass == 0 && staging && !views_by_argument;
            launch_begin_pass(st, tg->d_vis, tg->flags, g, v0, nv, b.count_small, zero_words * 4, copy_views ? staging->d : nullptr, ps.views,
                              copy_views ? view_bytes : 0);
            if (copy_views) { SR_HIP(hipEventRecord(staging->ev, st)); staging->in_flight = true; }
        }
        if (ev) SR_HIP(hipEventRecord(ev[1], st));
        if (!fused) launch_vertex(st, m->px, m->py, m->pz, m->n_verts, d_views, nv, g.W, g.H, sp, nullptr, nullptr, dev->fill_blocks);
        if (ev) SR_HIP(hipEventRecord(ev[2], st));
        const DepthSource depth_source{m->px, m->py, m->pz, d_views};
        if (pass == 0 && n_views >= 2 && m->d_chunk_verts) ++dev->counts.chunk_list_draws;
        launch_setup(st, m->d_idx, m->n_tris, sp, g, b, clip, nv, tg->d_vis, tg->flags, v0, ev ? ev[3] : nullptr, !tg->lean, dev->fill_blocks,
                     fused ? cull_use.visible : nullptr /* (culled vertex groups were skipped by the fused vertex launch only) */, dev->staged_min_blocks,
                     depth_plane ? nullptr : &depth_source, m->d_chunk_verts ? &m->chunk_verts : nullptr);
        if (ev && !m->n_tris) SR_HIP(hipEventRecord(ev[3], st));
        if (ev) SR_HIP(hipEventRecord(ev[4], st));
        uint32_t* pass_counters = (uint32_t*)tg->hd_counters + (size_t)pass * kCountersPerPass;
        if (tg->lean) {   // the shading pass reports the counters of the pass
            if (tg->direct && tg->direct_clip) launch_near_clip(st, m->d_idx, m->n_tris, m->px, m->py, m->pz, d_views, g, clip, b, tg->d_vis, tg->flags, v0);
            if (ev) { SR_HIP(hipEventRecord(ev[5], st)); SR_HIP(hipEventRecord(ev[6], st)); }
            if (tg->direct) launch_direct_raster(st, m->d_idx, m->n_tris, sp, g, b, tg->d_vis, tg->flags, v0, nv, tg->direct_expected);
        } else {
            // (before the scan kernel, which reports the counters to the host)
            launch_near_clip(st, m->d_idx, m->n_tris, m->px, m->py, m->pz, d_views, g, clip, b, tg->d_vis, tg->flags, v0);
            launch_scan(st, b, nv * n_tiles, nv, pass_counters);
            if (ev) SR_HIP(hipEventRecord(ev[5], st));
            launch_fill(st, g, b, m->n_tris, nv);
            if (ev) SR_HIP(hipEventRecord(ev[6], st));
            launch_raster(st, m->d_idx, m->n_tris, sp, g, b, tg->d_vis, tg->flags, v0, nv);
        }
        if (ev) SR_HIP(hipEventRecord(ev[7], st));
        FragLaunch f{};
        f.idx = m->d_idx; f.verts_aos = m->d_aos; f.views = d_views; f.vis = tg->d_vis; f.color = tg->d_color; f.g = g;
        f.albedo = make_tex_desc(albedo); f.metallic = make_tex_desc(metallic); f.roughness = make_tex_desc(roughness);
        f.normal_mapping = tg->normal_mapping && tg->tex[1] != nullptr;
        if (f.normal_mapping) f.normal = make_tex_desc(tg->tex[1]);
        f.decode = dev->d_decode; f.thresh = dev->d_thresh; f.view0 = v0; f.n_views = nv; f.flags = tg->flags;
        f.report = PassReport{b.counters, b.pending_total, nv, b.micro_spread, tg->lean ? pass_counters : nullptr};
        f.packed_mr = packed_pair(dev, metallic, roughness, !no_alloc);
        f.identity_model = true;
        for (uint32_t v = v0; v < v0 + nv; ++v) f.identity_model = f.identity_model && tg->views[v].model_is_identity != 0.0f;
        launch_fragment(st, f);
        if (ev) SR_HIP(hipEventRecord(ev[8], st));
    }
    SR_HIP(hipGetLastError());
    tg->passes = passes;
    tg->capacity_at_draw = ps.ref_capacity;
    tg->clip_capacity_at_draw = ps.clip_capacity;
    tg->check_pending = true;
    return SR_OK;
}

// waits for the target's last draw; re-issues it once with a larger reference list if it overflowed
int finish_target(sr_device_t* dev, sr_target_t* tg)
{
    SR_HIP(hipStreamSynchronize(dev->stream));
    if (!tg->check_pending) return SR_OK;
    for (int attempt = 0; attempt < 3; ++attempt) {
        uint64_t worst = 0, total = 0, unproj = 0, binned = 0, micro = 0, clipped = 0, listed = 0, too_big = 0;
        for (uint32_t p = 0; p < tg->passes; ++p) {
            const uint32_t* c = tg->h_counters + (size_t)p * kCountersPerPass;
            worst = std::max<uint64_t>(worst, c[0]); total += c[0]; too_big += c[1]; unproj += c[2]; binned += c[3]; micro += c[4];
            listed += c[5]; clipped += c[6];
        }
        tg->total_refs = total; tg->unprojectable = unproj; tg->binned = binned; tg->micro = micro; tg->clipped = clipped;
        // drawn without the kernels it turned out to need: pending triangles without the direct rasteriser (or some too large for it),
        // clip-listed ones without the clipper
        const bool lean_missed = tg->lean && ((binned > 0 && !tg->direct) || too_big > 0 || (listed > 0 && !(tg->direct && tg->direct_clip)));
        if (tg->no_alloc && listed > tg->clip_capacity_at_draw)   // the header promises sr_draw's frame or an error, never a frame with triangles missing
            return fail(SR_ERR_UNSUPPORTED, "sr_draw_no_alloc: " + std::to_string(listed) + " triangles cross the near plane or the guard band, the clip list of this entry point holds " +
                                                std::to_string(tg->clip_capacity_at_draw) + " (use sr_draw)");
        if (!lean_missed && worst <= tg->capacity_at_draw) { tg->check_pending = false; return SR_OK; }
        if (attempt == 2) break;
        if (lean_missed) { tg->direct_banned = tg->direct_banned || too_big > 0; tg->lean = tg->direct = tg->direct_clip = false; ++dev->counts.lean_redraws; }
        else if (tg->no_alloc) return fail(SR_ERR_UNSUPPORTED, "sr_draw_no_alloc: the frame needs " + std::to_string(worst) + " tile references, the static scratch holds " +
                                                                    std::to_string(tg->capacity_at_draw) + " after this draw's other buffers (use sr_draw)");
        else { dev->ref_capacity = std::min<uint64_t>(worst + worst / 4 + 1024, 0xFFFFFFFFull); ++dev->counts.overflow_redraws; }
        int rc = issue_draw(dev, tg);
        if (rc != SR_OK) return rc;
        SR_HIP(hipStreamSynchronize(dev->stream));
    }
    return fail(SR_ERR_UNSUPPORTED, "tile reference list exceeds 2^32 entries in one pass");
}

// A mesh or a texture is about to be destroyed: resolve every draw of this device that still refers to it (a draw
// whose tile reference list overflowed is re-issued when its target is next read, and needs them) and drop the
// references.  The reference's buffers and images may be destroyed as soon as the device is idle (src/main.cpp:170,
// 300-347); this keeps that order legal here.
void release_references(sr_device_t* dev, const sr_mesh_t* mesh, const sr_texture_t* tex)
{
    for (sr_target_t* t : dev->targets) {
        bool uses = mesh && t->mesh == mesh;
        for (int k = 0; k < 4 && tex; ++k) uses = uses || t->tex[k] == tex;
        if (!uses) continue;
        if (t->check_pending) (void)finish_target(dev, t);
        t->check_pending = false;   // (if the re-issued draw failed, its counters are what they are)
        t->mesh = nullptr;
        for (int k = 0; k < 4; ++k) t->tex[k] = nullptr;
    }
}

}  // namespace

// =================================================================================================
extern "C" {

const char* sr_status_string(int status)
{
    switch (status) {
    case SR_OK: return "ok";
    case SR_ERR_INVALID_ARGUMENT: return "invalid argument";
    case SR_ERR_OUT_OF_MEMORY: return "out of device memory";
    case SR_ERR_DEVICE: return "HIP call failed";
    case SR_ERR_INDEX_OUT_OF_RANGE: return "vertex index out of range";
    case SR_ERR_UNSUPPORTED: return "unsupported configuration";
    case SR_ERR_NO_DEVICE: return "no gfx950 device";
    case SR_ERR_FILE: return "file cannot be read";
    default: return "unknown status";
    }
}

const char* sr_last_error_string(void) { return g_last_error.c_str(); }

int sr_device_create(int hip_ordinal, sr_device* out)
{
    if (!out) return fail(SR_ERR_INVALID_ARGUMENT, "out is NULL");
    *out = nullptr;
    int count = 0;
    const hipError_t count_err = hipGetDeviceCount(&count);
    if (count_err != hipSuccess || count <= 0) {
        (void)hipGetLastError();
        /* a GPU left faulted by an earlier process shows up here as an error, not as zero devices */
        return fail(SR_ERR_NO_DEVICE, std::string("no HIP device visible (hipGetDeviceCount: ") +
                                          hipGetErrorString(count_err) + ", count " + std::to_string(count) + ")");
    }
    if (hip_ordinal < 0 || hip_ordinal >= count) return fail(SR_ERR_NO_DEVICE, "HIP ordinal out of range");
    hipDeviceProp_t prop;
    SR_HIP(hipGetDeviceProperties(&prop, hip_ordinal));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(SR_ERR_NO_DEVICE, std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950 only");
    SR_HIP(hipSetDevice(hip_ordinal));
    sr_device_t* dev = new sr_device_t();
    dev->ordinal = hip_ordinal;
    float dec[256], thr[256];
    for (int c = 0; c < 256; ++c) dec[c] = (float)srgb_decode_d(c / 255.0);
    for (int c = 0; c < 255; ++c) thr[c] = (float)srgb_decode_d((c + 0.5) / 255.0);
    thr[255] = 3.0e38f;
    hipError_t e = hipStreamCreateWithFlags(&dev->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipMalloc((void**)&dev->d_decode, sizeof dec);
    if (e == hipSuccess) e = hipMalloc((void**)&dev->d_thresh, sizeof thr);
    if (e == hipSuccess) e = hipMemcpy(dev->d_decode, dec, sizeof dec, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(dev->d_thresh, thr, sizeof thr, hipMemcpyHostToDevice);
    // this GPU's copy of the static scratch array (the device is current: hipSetDevice above)
    if (e == hipSuccess) e = hipGetSymbolAddress(&dev->static_scratch, HIP_SYMBOL(scratch));
    if (e == hipSuccess && dev->static_scratch == nullptr) e = hipErrorInvalidSymbol;
    if (e == hipSuccess) e = hipMalloc((void**)&dev->d_cull_counts, 16);
    if (e == hipSuccess) e = hipMemsetAsync(dev->d_cull_counts, 0, 16, dev->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(dev->stream);
    if (e != hipSuccess) { (void)sr_device_destroy(dev); SR_HIP(e); }
    if (const char* v = std::getenv("SR_TILE_LOG2")) dev->tile_log2_override = std::atoi(v);
    if (const char* v = std::getenv("SR_SCRATCH_GIB")) dev->scratch_budget = (size_t)std::max(1, std::atoi(v)) << 30;
    if (const char* v = std::getenv("SR_REF_CAPACITY")) dev->ref_capacity = (uint64_t)std::max(1, std::atoi(v));   // initial size of the tile reference list (tests: force the grow-and-redraw path)
    if (const char* v = std::getenv("SR_CLIP_CAPACITY")) dev->clip_capacity = (uint32_t)std::max(1, std::atoi(v));   // tests: a full clip list
    if (const char* v = std::getenv("SR_FILL_BLOCKS")) dev->fill_blocks = (uint32_t)std::max(1, std::atoi(v));   // tests: 1 = a workgroup always takes all views
    if (const char* v = std::getenv("SR_CULL_MIN_CHUNKS")) dev->cull_min_chunks = (uint32_t)std::max(1, std::atoi(v));   // tests: cull small meshes too
    if (std::getenv("SR_NO_CULL")) dev->cull_enabled = false;
    if (const char* v = std::getenv("SR_NO_CHUNK_VERTS")) dev->chunk_vertex_lists = std::atoi(v) == 0;
    if (const char* v = std::getenv("SR_STORE_DEPTH")) dev->recompute_depth = std::atoi(v) == 0;
    if (const char* v = std::getenv("SR_NO_PACKED_MAPS")) dev->pack_pairs = std::atoi(v) == 0;
    if (const char* v = std::getenv("SR_DIRECT_MAX_TRIANGLES")) dev->direct_max_triangles = (uint32_t)std::max(0, std::atoi(v));
    if (const char* v = std::getenv("SR_STAGED_MIN_BLOCKS")) dev->staged_min_blocks = (uint32_t)std::max(1, std::atoi(v));   // tests: 1 = the staged resolve for small launches too
    if (const char* v = std::getenv("SR_SCRATCH_MIB")) dev->scratch_budget = (size_t)std::max(1, std::atoi(v)) << 20;   // small budgets: many passes per draw
    *out = dev;
    return SR_OK;
}

int sr_device_destroy(sr_device dev)
{
    if (!dev) return SR_OK;
    (void)hipSetDevice(dev->ordinal);
    if (dev->stream) (void)hipStreamSynchronize(dev->stream);
    for (hipEvent_t e : dev->events) (void)hipEventDestroy(e);
    dev->screen.release(); dev->tile_count.release(); dev->tile_offset.release(); dev->records.release();
    dev->pending.release(); dev->refs.release(); dev->views.release(); dev->clip.release();
    dev->cull_visible.release(); dev->cull_stamp.release(); (void)hipFree(dev->d_cull_counts);
    for (auto& sg : dev->staging) { if (sg.h) (void)hipHostFree(sg.h); if (sg.ev) (void)hipEventDestroy(sg.ev); }
    for (auto& p : dev->packed_pairs) (void)hipFree(p.d_chain);
    (void)hipFree(dev->d_decode); (void)hipFree(dev->d_thresh);
    if (dev->stream) (void)hipStreamDestroy(dev->stream);
    for (sr_target_t* t : dev->targets) { t->mesh = nullptr; t->check_pending = false; }   // targets outliving their device can only be destroyed
    delete dev;
    return SR_OK;
}

int sr_device_synchronize(sr_device dev)
{
    if (!dev) return fail(SR_ERR_INVALID_ARGUMENT, "device is NULL");
    SR_HIP(hipSetDevice(dev->ordinal));
    SR_HIP(hipStreamSynchronize(dev->stream));
    // idle means: every draw is complete, including those that have to be issued a second time (a tile reference list
    // that was too small, a lean draw that needed the full pipeline) -- a caller who bound his own colour memory to a
    // target reads it after this call
    for (sr_target_t* t : dev->targets)
        if (t->check_pending) { const int rc = finish_target(dev, t); if (rc != SR_OK) return rc; }
    return SR_OK;
}

// ---- mesh ---------------------------------------------------------------------------------------
int sr_mesh_create(sr_device dev, const void* vertices, uint32_t n_vertices, const uint32_t* indices, uint32_t n_indices, sr_mesh* out)
{
    if (!dev || !out) return fail(SR_ERR_INVALID_ARGUMENT, "device or out is NULL");
    *out = nullptr;
    if ((n_vertices && !vertices) || (n_indices && !indices)) return fail(SR_ERR_INVALID_ARGUMENT, "NULL data with a non-zero count");
    // (the set-up kernel addresses a view's window-space planes with 32-bit byte offsets, 8 bytes per vertex)
    if (n_vertices > (1u << 29) - 4u) return fail(SR_ERR_UNSUPPORTED, "a mesh holds at most 2^29 - 4 vertices");
    const uint32_t n_tris = n_indices / 3;   // an incomplete trailing primitive is not drawn
    for (size_t i = 0; i < (size_t)n_tris * 3; ++i)
        if (indices[i] >= n_vertices) return fail(SR_ERR_INDEX_OUT_OF_RANGE, "index " + std::to_string(i) + " = " + std::to_string(indices[i]) + " >= " + std::to_string(n_vertices));
    SR_HIP(hipSetDevice(dev->ordinal));
    sr_mesh_t* m = new sr_mesh_t();
    m->n_verts = n_vertices; m->n_tris = n_tris;
    const size_t nv = std::max<uint32_t>(n_vertices, 1), ni = std::max<uint32_t>(n_tris * 3, 1);
    hipError_t e = hipMalloc(&m->d_aos, nv * 32);
    const size_t ps = std::max<uint32_t>(plane_stride(n_vertices), 4);   // the three position planes, each padded to whole groups of four
    if (e == hipSuccess) e = hipMalloc((void**)&m->px, ps * 12);
    if (e == hipSuccess) e = hipMalloc((void**)&m->d_idx, ni * 4);
    if (e != hipSuccess) { (void)hipFree(m->d_aos); (void)hipFree(m->px); (void)hipFree(m->d_idx); delete m; SR_HIP(e); }
    m->py = m->px + ps; m->pz = m->py + ps;
    if (n_vertices) e = hipMemcpyAsync(m->d_aos, vertices, (size_t)n_vertices * 32, hipMemcpyHostToDevice, dev->stream);
    if (e == hipSuccess && n_tris) e = hipMemcpyAsync(m->d_idx, indices, (size_t)n_tris * 12, hipMemcpyHostToDevice, dev->stream);
    if (e == hipSuccess) launch_split_positions(dev->stream, m->d_aos, n_vertices, m->px, m->py, m->pz);
    if (e == hipSuccess && n_tris) {   // bounds of every chunk, for draws into a restricted rectangle
        m->n_chunks = (n_tris + setup_block_triangles() - 1) / setup_block_triangles();
        e = hipMalloc((void**)&m->d_chunk_sphere, (size_t)m->n_chunks * (sizeof(float4) + sizeof(uint4)));
        if (e == hipSuccess) {
            m->d_chunk_vrange = (uint4*)(m->d_chunk_sphere + m->n_chunks);
            launch_chunk_bounds(dev->stream, m->d_idx, n_tris, m->d_aos, m->d_chunk_sphere, m->d_chunk_vrange);
        }
    }
    uint32_t* d_unfit = nullptr;
    if (e == hipSuccess && n_tris && dev->chunk_vertex_lists) {   // the chunks' distinct vertices, for batches of views
        const size_t list_words = (size_t)m->n_chunks * kChunkVertsMax, words = list_words + m->n_chunks + n_tris + 1;
        e = hipMalloc((void**)&m->d_chunk_verts, words * 4);
        if (e == hipSuccess) {
            uint32_t* count = m->d_chunk_verts + list_words;
            uint32_t* local = count + m->n_chunks;
            d_unfit = local + n_tris;
            m->chunk_verts = ChunkVerts{m->d_chunk_verts, count, local};
            e = hipMemsetAsync(d_unfit, 0, 4, dev->stream);
            if (e == hipSuccess) launch_chunk_verts(dev->stream, m->d_idx, n_tris, m->d_chunk_vrange, m->d_chunk_verts, count, local, d_unfit);
        }
    }
    if (e == hipSuccess) e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(dev->stream);   // uploaded on return, like the reference's blocking staging copy
    if (e == hipSuccess && d_unfit) {
        uint32_t unfit = 0;
        e = hipMemcpy(&unfit, d_unfit, 4, hipMemcpyDeviceToHost);
        if (e == hipSuccess && unfit) { (void)hipFree(m->d_chunk_verts); m->d_chunk_verts = nullptr; m->chunk_verts = ChunkVerts{}; }
    }
    if (e != hipSuccess) { (void)sr_mesh_destroy(dev, m); SR_HIP(e); }
    *out = m;
    return SR_OK;
}

int sr_mesh_destroy(sr_device dev, sr_mesh mesh)
{
    if (!mesh) return SR_OK;
    if (dev) {
        (void)hipSetDevice(dev->ordinal);
        release_references(dev, mesh, nullptr);
        (void)hipStreamSynchronize(dev->stream);
    }
    (void)hipFree(mesh->d_aos); (void)hipFree(mesh->px); (void)hipFree(mesh->d_idx); (void)hipFree(mesh->d_chunk_sphere);
    (void)hipFree(mesh->d_chunk_verts);
    delete mesh;
    return SR_OK;
}

// ---- texture ------------------------------------------------------------------------------------
void sr_sampler_default(sr_sampler_desc* out)
{
    if (!out) return;
    out->min_lod = 0.0f; out->max_lod = -1.0f; out->mip_lod_bias = 0.0f; out->max_anisotropy = 16.0f;
}

int sr_texture_create_rgba8_srgb(sr_device dev, const void* rgba8, int width, int height, const sr_sampler_desc* sampler, sr_texture* out)
{
    if (!dev || !out || !rgba8) return fail(SR_ERR_INVALID_ARGUMENT, "NULL argument");
    *out = nullptr;
    if (width <= 0 || height <= 0 || width > 16384 || height > 16384) return fail(SR_ERR_INVALID_ARGUMENT, "texture size out of range (1..16384)");
    SR_HIP(hipSetDevice(dev->ordinal));
    sr_texture_t* t = new sr_texture_t();
    t->w = width; t->h = height;
    int m = std::max(width, height), levels = 1;
    while (m > 1) { m >>= 1; ++levels; }   // floor(log2(max(w,h))) + 1, src/texture.cpp:187-189
    t->levels = levels;
    size_t off = 0;
    for (int l = 0; l < levels; ++l) { t->level_off[l] = (uint32_t)off; off += (size_t)std::max(width >> l, 1) * std::max(height >> l, 1); }
    t->total_texels = off;
    if (sampler) t->samp = *sampler; else sr_sampler_default(&t->samp);
    hipError_t e = hipMalloc((void**)&t->d_chain, off * 4);
    if (e != hipSuccess) { delete t; SR_HIP(e); }
    e = hipMemcpyAsync(t->d_chain, rgba8, (size_t)width * height * 4, hipMemcpyHostToDevice, dev->stream);
    for (int l = 1; l < levels && e == hipSuccess; ++l)
        launch_mip_blit(dev->stream, t->d_chain + t->level_off[l - 1], std::max(width >> (l - 1), 1), std::max(height >> (l - 1), 1),
                        t->d_chain + t->level_off[l], std::max(width >> l, 1), std::max(height >> l, 1), dev->d_decode, dev->d_thresh);
    if (e == hipSuccess) e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(dev->stream);
    if (e != hipSuccess) { (void)hipFree(t->d_chain); delete t; SR_HIP(e); }
    *out = t;
    return SR_OK;
}

int sr_texture_destroy(sr_device dev, sr_texture tex)
{
    if (!tex) return SR_OK;
    if (dev) {
        (void)hipSetDevice(dev->ordinal);
        release_references(dev, nullptr, tex);
        (void)hipStreamSynchronize(dev->stream);
        for (size_t k = dev->packed_pairs.size(); k-- > 0;)        // the chains that pack this map with another
            if (dev->packed_pairs[k].a == tex || dev->packed_pairs[k].b == tex) {
                (void)hipFree(dev->packed_pairs[k].d_chain);
                dev->packed_pairs.erase(dev->packed_pairs.begin() + (long)k);
            }
    }
    (void)hipFree(tex->d_chain);
    delete tex;
    return SR_OK;
}

int sr_texture_mip_levels(sr_texture tex) { return tex ? tex->levels : 0; }

int sr_texture_size(sr_texture tex, int* width, int* height)
{
    if (!tex || !width || !height) return fail(SR_ERR_INVALID_ARGUMENT, "NULL argument");
    *width = tex->w; *height = tex->h;
    return SR_OK;
}

int sr_texture_read_level(sr_device dev, sr_texture tex, int level, void* host_rgba8, size_t bytes)
{
    if (!dev || !tex || !host_rgba8) return fail(SR_ERR_INVALID_ARGUMENT, "NULL argument");
    if (level < 0 || level >= tex->levels) return fail(SR_ERR_INVALID_ARGUMENT, "mip level out of range");
    const size_t need = (size_t)std::max(tex->w >> level, 1) * std::max(tex->h >> level, 1) * 4;
    if (bytes != need) return fail(SR_ERR_INVALID_ARGUMENT, "host buffer must hold exactly " + std::to_string(need) + " bytes");
    SR_HIP(hipSetDevice(dev->ordinal));
    SR_HIP(hipStreamSynchronize(dev->stream));
    SR_HIP(hipMemcpy(host_rgba8, tex->d_chain + tex->level_off[level], need, hipMemcpyDeviceToHost));
    return SR_OK;
}

// ---- files ---------------------------------------------------------------------------------------
int sr_image_read_file(const char* path, void** rgba8_out, int* width, int* height)
{
    if (!path || !rgba8_out || !width || !height) return fail(SR_ERR_INVALID_ARGUMENT, "NULL argument");
    *rgba8_out = nullptr;
    sr_assets::Image img;
    std::string err;
    try {
        if (!sr_assets::load_image(path, img, err)) return fail(SR_ERR_FILE, err);
    } catch (const std::exception& e) { return fail(SR_ERR_OUT_OF_MEMORY, std::string("reading ") + path + ": " + e.what()); }
    void* p = std::malloc(img.rgba.size());
    if (!p) return fail(SR_ERR_OUT_OF_MEMORY, "host allocation failed");
    std::memcpy(p, img.rgba.data(), img.rgba.size());
    *rgba8_out = p; *width = img.width; *height = img.height;
    return SR_OK;
}

void sr_image_free(void* rgba8) { std::free(rgba8); }

int sr_texture_create_from_file(sr_device dev, const char* path, const sr_sampler_desc* sampler, sr_texture* out)
{
    if (!dev || !path || !out) return fail(SR_ERR_INVALID_ARGUMENT, "NULL argument");
    *out = nullptr;
    sr_assets::Image img;
    std::string err;
    try {
        if (!sr_assets::load_image(path, img, err)) return fail(SR_ERR_FILE, "failed to load texture image: " + err);   // src/texture.cpp:281-284
    } catch (const std::exception& e) { return fail(SR_ERR_OUT_OF_MEMORY, std::string("reading ") + path + ": " + e.what()); }
    return sr_texture_create_rgba8_srgb(dev, img.rgba.data(), img.width, img.height, sampler, out);
}

int sr_model_read_file(const char* path, sr_model* out)
{
    if (!path || !out) return fail(SR_ERR_INVALID_ARGUMENT, "NULL argument");
    *out = nullptr;
    sr_model_t* m = new sr_model_t();
    std::string err;
    try {
        if (!sr_assets::load_model(path, m->host, err)) { delete m; return fail(SR_ERR_FILE, err); }
    } catch (const std::exception& e) { delete m; return fail(SR_ERR_OUT_OF_MEMORY, std::string("reading ") + path + ": " + e.what()); }
    *out = m;
    return SR_OK;
}

int sr_model_load(sr_device dev, const char* path, sr_model* out)
{
    if (!dev || !path || !out) return fail(SR_ERR_INVALID_ARGUMENT, "NULL argument");
    sr_model m = nullptr;
    int rc = sr_model_read_file(path, &m);
    if (rc != SR_OK) return rc;
    for (const sr_assets::HostMesh& h : m->host) {
        sr_mesh mesh = nullptr;
        rc = sr_mesh_create(dev, h.vertices.data(), (uint32_t)(h.vertices.size() / 8), h.indices.data(), (uint32_t)h.indices.size(), &mesh);
        if (rc != SR_OK) { const std::string keep = g_last_error; (void)sr_model_destroy(dev, m); g_last_error = keep; *out = nullptr; return rc; }
        m->meshes.push_back(mesh);
    }
    *out = m;
    return SR_OK;
}

uint32_t sr_model_mesh_count(sr_model model) { return model ? (uint32_t)model->host.size() : 0u; }

sr_mesh sr_model_mesh(sr_model model, uint32_t index) { return model && index < model->meshes.size() ? model->meshes[index] : nullptr; }

int sr_model_mesh_data(sr_model model, uint32_t index, const void** vertices, uint32_t* n_vertices, const uint32_t** indices, uint32_t* n_indices)
{
    if (!model || index >= model->host.size()) return fail(SR_ERR_INVALID_ARGUMENT, "mesh index out of range");
    const sr_assets::HostMesh& h = model->host[index];
    if (vertices) *vertices = h.vertices.data();
    if (n_vertices) *n_vertices = (uint32_t)(h.vertices.size() / 8);
    if (indices) *indices = h.indices.data();
    if (n_indices) *n_indices = (uint32_t)h.indices.size();
    return SR_OK;
}

int sr_model_destroy(sr_device dev, sr_model model)
{
    if (!model) return SR_OK;
    for (sr_mesh mesh : model->meshes) (void)sr_mesh_destroy(dev, mesh);
    delete model;
    return SR_OK;
}

// ---- target -------------------------------------------------------------------------------------
int sr_target_create(sr_device dev, int width, int height, int samples, uint32_t n_views, const sr_rect* rect, sr_target* out)
{
    if (!dev || !out) return fail(SR_ERR_INVALID_ARGUMENT, "NULL argument");
    *out = nullptr;
    if (width <= 0 || height <= 0 || width > 16384 || height > 16384) return fail(SR_ERR_INVALID_ARGUMENT, "frame size out of range (1..16384)");
    if (samples != 1 && samples != 4) return fail(SR_ERR_UNSUPPORTED, "samples must be 1 or 4");
    if (n_views == 0) return fail(SR_ERR_INVALID_ARGUMENT, "n_views must be > 0");
    sr_rect r = {0, 0, width, height};
    if (rect) r = *rect;
    if (r.x < 0 || r.y < 0 || r.width <= 0 || r.height <= 0 || r.x + r.width > width || r.y + r.height > height)
        return fail(SR_ERR_INVALID_ARGUMENT, "rect must lie inside the frame");
    SR_HIP(hipSetDevice(dev->ordinal));
    sr_target_t* t = new sr_target_t();
    FrameGeom& g = t->g;
    g.W = width; g.H = height; g.rx = r.x; g.ry = r.y; g.rw = r.width; g.rh = r.height; g.samples = samples;
    g.tile_log2 = choose_tile_log2(dev, samples, g.rw, g.rh);
    g.tiles_x = (g.rw + (1 << g.tile_log2) - 1) >> g.tile_log2;
    g.tiles_y = (g.rh + (1 << g.tile_log2) - 1) >> g.tile_log2;
    if (g.tiles_x > 256 || g.tiles_y > 256) { delete t; return fail(SR_ERR_UNSUPPORTED, "more than 256 tiles along one axis"); }
    t->n_views = n_views;
    const size_t px = (size_t)g.rw * g.rh;
    hipError_t e = hipMalloc((void**)&t->d_vis, px * samples * 8 * n_views);
    if (e == hipSuccess) e = hipMalloc((void**)&t->d_color_own, px * 4 * n_views);
    if (e != hipSuccess) { (void)hipFree(t->d_vis); delete t; SR_HIP(e); }
    t->d_color = t->d_color_own;
    // every block of the visibility buffer counts as holding something until a first draw has cleared it
    t->flags.nbx = (g.rw + (int)kFlagBlockW - 1) / (int)kFlagBlockW;
    t->flags.nby = (g.rh + (int)kFlagBlockH - 1) / (int)kFlagBlockH;
    const size_t flag_bytes = (size_t)n_views * t->flags.nbx * t->flags.nby * 4;
    e = hipMalloc((void**)&t->flags.flags, 2 * flag_bytes);   // both planes; everything flagged: the first draw clears every word and stores every pixel
    if (e == hipSuccess) { t->flags.color_dirty = t->flags.flags + flag_bytes / 4; e = hipMemsetAsync(t->flags.flags, 0x01, 2 * flag_bytes, dev->stream); }
    if (e == hipSuccess) e = hipStreamSynchronize(dev->stream);
    if (e != hipSuccess) { (void)hipFree(t->d_vis); (void)hipFree(t->d_color_own); (void)hipFree(t->flags.flags); delete t; SR_HIP(e); }
    // room for the counters of a one-pass draw, so that such a draw allocates nothing here
    e = hipHostMalloc((void**)&t->h_counters, (size_t)kCountersPerPass * 4, hipHostMallocMapped);
    if (e == hipSuccess) e = hipHostGetDevicePointer(&t->hd_counters, t->h_counters, 0);
    if (e != hipSuccess) { (void)hipFree(t->d_vis); (void)hipFree(t->d_color_own); (void)hipFree(t->flags.flags); if (t->h_counters) (void)hipHostFree(t->h_counters); delete t; SR_HIP(e); }
    t->h_counters_cap = 1;
    t->views.reserve(n_views);      // a later draw does not allocate host memory for them either
    dev->targets.push_back(t);
    *out = t;
    return SR_OK;
}

int sr_target_destroy(sr_device dev, sr_target target)
{
    if (!target) return SR_OK;
    if (dev) {
        (void)hipSetDevice(dev->ordinal); (void)hipStreamSynchronize(dev->stream);
        dev->targets.erase(std::remove(dev->targets.begin(), dev->targets.end(), target), dev->targets.end());
    }
    (void)hipFree(target->d_vis); (void)hipFree(target->d_color_own); (void)hipFree(target->flags.flags);
    if (target->h_counters) (void)hipHostFree(target->h_counters);
    delete target;
    return SR_OK;
}

int sr_target_bind_color(sr_device dev, sr_target target, void* device_ptr)
{
    if (!dev || !target) return fail(SR_ERR_INVALID_ARGUMENT, "NULL argument");
    SR_HIP(hipSetDevice(dev->ordinal));
    int rc = finish_target(dev, target);
    if (rc != SR_OK) return rc;
    target->d_color = device_ptr ? (uint32_t*)device_ptr : target->d_color_own;
    // nothing is known about what this memory holds: every block's colour has to be stored by the next draw
    SR_HIP(hipMemsetAsync(target->flags.color_dirty, 0x01, (size_t)target->n_views * target->flags.nbx * target->flags.nby * 4, dev->stream));
    return SR_OK;
}

int sr_target_readback(sr_device dev, sr_target target, int kind, uint32_t view, void* host, size_t bytes)
{
    if (!dev || !target || !host) return fail(SR_ERR_INVALID_ARGUMENT, "NULL argument");
    if (view >= target->n_views) return fail(SR_ERR_INVALID_ARGUMENT, "view out of range");
    if (!target->drawn) return fail(SR_ERR_INVALID_ARGUMENT, "nothing has been drawn into this target");
    SR_HIP(hipSetDevice(dev->ordinal));
    int rc = finish_target(dev, target);
    if (rc != SR_OK) return rc;
    const FrameGeom& g = target->g;
    const size_t px = (size_t)g.rw * g.rh, ns = px * g.samples;
    size_t need = 0;
    switch (kind) {
    case SR_READ_COLOR_BGRA8: need = px * 4; break;
    case SR_READ_DEPTH_F32: case SR_READ_TRI_ID_U32: need = ns * 4; break;
    case SR_READ_VIS_U64: need = ns * 8; break;
    default: return fail(SR_ERR_INVALID_ARGUMENT, "unknown read-back kind");
    }
    if (bytes != need) return fail(SR_ERR_INVALID_ARGUMENT, "host buffer must hold exactly " + std::to_string(need) + " bytes");
    if (kind == SR_READ_COLOR_BGRA8) {
        SR_HIP(hipMemcpy(host, target->d_color + px * view, need, hipMemcpyDeviceToHost));
    } else if (kind == SR_READ_VIS_U64) {
        SR_HIP(hipMemcpy(host, target->d_vis + ns * view, need, hipMemcpyDeviceToHost));
    } else {
        void* tmp = nullptr;
        SR_HIP(hipMalloc(&tmp, need));
        launch_split_vis(dev->stream, target->d_vis + ns * view, ns, kind == SR_READ_DEPTH_F32 ? (float*)tmp : nullptr,
                         kind == SR_READ_TRI_ID_U32 ? (uint32_t*)tmp : nullptr);
        hipError_t e = hipStreamSynchronize(dev->stream);
        if (e == hipSuccess) e = hipMemcpy(host, tmp, need, hipMemcpyDeviceToHost);
        (void)hipFree(tmp);
        SR_HIP(e);
    }
    return SR_OK;
}

// ---- draw ---------------------------------------------------------------------------------------
static int draw_common(sr_device dev, sr_target target, sr_mesh mesh, const void* ubo_blocks, uint32_t n_views, size_t ubo_stride, const sr_texture textures[4],
                       bool no_alloc)
{
    if (!dev || !target || !mesh || !ubo_blocks || !textures) return fail(SR_ERR_INVALID_ARGUMENT, "NULL argument");
    if (no_alloc && n_views > kStaticScratchViews)
        return fail(SR_ERR_UNSUPPORTED, "sr_draw_no_alloc draws at most " + std::to_string(kStaticScratchViews) + " views per call");
    if (n_views == 0 || n_views > target->n_views) return fail(SR_ERR_INVALID_ARGUMENT, "n_views must be in 1..target views");
    if (ubo_stride < 204 || (ubo_stride & 3)) return fail(SR_ERR_INVALID_ARGUMENT, "ubo_stride must be >= 204 and a multiple of 4");
    if (!textures[0] || !textures[2] || !textures[3]) return fail(SR_ERR_INVALID_ARGUMENT, "albedo, metallic and roughness textures are required");
    if (dev->normal_mapping && !textures[1]) return fail(SR_ERR_INVALID_ARGUMENT, "normal mapping is on: the normal texture is required");
    SR_HIP(hipSetDevice(dev->ordinal));
    // a finished earlier draw tells how many tile references the scene needs: grow before reuse
    if (!no_alloc && target->check_pending && hipStreamQuery(dev->stream) == hipSuccess) {
        for (uint32_t p = 0; p < target->passes; ++p)
            if (target->h_counters[(size_t)p * kCountersPerPass] > dev->ref_capacity) {
                const uint64_t w = target->h_counters[(size_t)p * kCountersPerPass];
                dev->ref_capacity = std::min<uint64_t>(w + w / 4 + 1024, 0xFFFFFFFFull);
            }
    }
    (void)hipGetLastError();
    {   // lean or full (see sr_target_t::lean): from the counters of this target's previous draw.  They live in pinned
        // memory the kernels write; that draw may still be in flight, in which case the figures are a frame older --
        // either way only speed depends on them.
        const bool history = target->passes > 0 && target->mesh == mesh && target->views.size() == n_views && std::getenv("SR_NO_LEAN") == nullptr &&
                             !dev->normal_mapping;   // (the kernel that reports a lean draw's counters has no normal-mapping form)
        uint64_t binned = 0, listed = 0, refs = 0;
        for (uint32_t p = 0; history && p < target->passes; ++p) {
            const volatile uint32_t* c = target->h_counters + (size_t)p * kCountersPerPass;
            refs += c[0]; binned += c[3]; listed += c[5];
        }
        const bool was_direct = target->direct;
        if (target->mesh != mesh) target->direct_banned = false;
        target->lean = history && binned == 0 && listed == 0;
        // few pending triangles, and small ones -- a previous draw through the tile path says so by its references per triangle, one through
        // the direct rasteriser by not having been repeated
        target->direct_expected = (uint32_t)std::min<uint64_t>(binned / std::max(n_views, 1u) + 1, 0xFFFFFFFFull);
        target->direct = history && !target->lean && dev->direct_max_triangles > 0 && binned <= dev->direct_max_triangles &&
                         (was_direct || refs <= 4 * binned) && !target->direct_banned;
        target->direct_clip = target->direct && listed > 0;
        if (target->direct) target->lean = true;       // (issued like a lean draw, plus the direct rasteriser)
    }
    target->normal_mapping = dev->normal_mapping;
    target->mesh = mesh; target->mesh_verts = mesh->n_verts; target->mesh_tris = mesh->n_tris;
    for (int k = 0; k < 4; ++k) target->tex[k] = textures[k];
    target->views.resize(n_views);
    for (uint32_t v = 0; v < n_views; ++v) {
        const float* u = (const float*)((const char*)ubo_blocks + (size_t)v * ubo_stride);
        ViewXform& x = target->views[v];
        std::memcpy(x.model, u, 64);
        mat4_mul_mat4_host(u + 32, u + 16, x.pv);   // proj * view
        std::memcpy(x.cam, u + 48, 12);
        static const float identity[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
        x.model_is_identity = std::memcmp(x.model, identity, 64) == 0 ? 1.0f : 0.0f;
    }
    target->drawn = true;
    target->no_alloc = no_alloc;
    ++dev->counts.draws;
    if (target->lean) ++dev->counts.lean_draws;
    if (target->direct) ++dev->counts.direct_draws;
    return issue_draw(dev, target);
}

int sr_draw(sr_device dev, sr_target target, sr_mesh mesh, const void* ubo_blocks, uint32_t n_views, size_t ubo_stride, const sr_texture textures[4])
{
    return draw_common(dev, target, mesh, ubo_blocks, n_views, ubo_stride, textures, false);
}

int sr_draw_no_alloc(sr_device dev, sr_target target, sr_mesh mesh, const void* ubo_blocks, uint32_t n_views, size_t ubo_stride, const sr_texture textures[4])
{
    return draw_common(dev, target, mesh, ubo_blocks, n_views, ubo_stride, textures, true);
}

size_t sr_no_alloc_scratch_bytes(void) { return kStaticScratchBytes; }

int sr_device_set_normal_mapping(sr_device dev, int enabled)
{
    if (!dev) return fail(SR_ERR_INVALID_ARGUMENT, "NULL argument");
    dev->normal_mapping = enabled != 0;
    return SR_OK;
}

// ---- instrumentation ----------------------------------------------------------------------------
int sr_draw_get_stats(sr_device dev, sr_target target, sr_draw_stats* out)
{
    if (!dev || !target || !out) return fail(SR_ERR_INVALID_ARGUMENT, "NULL argument");
    if (!target->drawn) return fail(SR_ERR_INVALID_ARGUMENT, "nothing has been drawn into this target");
    SR_HIP(hipSetDevice(dev->ordinal));
    int rc = finish_target(dev, target);
    if (rc != SR_OK) return rc;
    const FrameGeom& g = target->g;
    const uint32_t n_views = (uint32_t)target->views.size(), mesh_tris = target->mesh_tris;
    std::memset(out, 0, sizeof *out);
    out->vertices = (uint64_t)target->mesh_verts * n_views;
    out->triangles = (uint64_t)mesh_tris * n_views;
    out->tile_refs = target->total_refs;
    out->triangles_unprojectable = target->unprojectable;
    out->pixels = (uint64_t)g.rw * g.rh * n_views;
    uint32_t* flags = nullptr; uint64_t* sums = nullptr;
    SR_HIP(hipMalloc((void**)&flags, (size_t)std::max<uint32_t>(mesh_tris, 1) * 4));
    hipError_t e = hipMalloc((void**)&sums, 16);
    if (e != hipSuccess) { (void)hipFree(flags); SR_HIP(e); }
    e = hipMemsetAsync(sums, 0, 16, dev->stream);
    const size_t px = (size_t)g.rw * g.rh;
    for (uint32_t v = 0; v < n_views && e == hipSuccess; ++v) {
        e = hipMemsetAsync(flags, 0, (size_t)std::max<uint32_t>(mesh_tris, 1) * 4, dev->stream);
        launch_stats(dev->stream, target->d_vis + px * g.samples * v, px, g.samples, flags, mesh_tris, sums, sums + 1);
    }
    uint64_t h[2] = {0, 0};
    if (e == hipSuccess) e = hipStreamSynchronize(dev->stream);
    if (e == hipSuccess) e = hipMemcpy(h, sums, 16, hipMemcpyDeviceToHost);
    (void)hipFree(flags); (void)hipFree(sums);
    SR_HIP(e);
    out->shaded_pixels = h[0];
    out->visible_triangles = h[1];
    out->triangles_binned = target->binned;
    out->triangles_direct = target->micro;
    out->triangles_clipped = target->clipped;
    out->passes = target->passes;
    return SR_OK;
}

int sr_device_get_draw_counts(sr_device dev, sr_draw_counts* out)
{
    if (!dev || !out) return fail(SR_ERR_INVALID_ARGUMENT, "NULL argument");
    *out = dev->counts;
    unsigned long long c[2] = {0, 0};
    SR_HIP(hipSetDevice(dev->ordinal));
    SR_HIP(hipStreamSynchronize(dev->stream));     // (the two culling counters live in device memory)
    SR_HIP(hipMemcpy(c, dev->d_cull_counts, sizeof c, hipMemcpyDeviceToHost));
    out->chunks_tested = c[0]; out->chunks_culled = c[1];
    return SR_OK;
}

int sr_profile_enable(sr_device dev, int enabled)
{
    if (!dev) return fail(SR_ERR_INVALID_ARGUMENT, "device is NULL");
    dev->profiling = enabled != 0;
    return SR_OK;
}

int sr_profile_get(sr_device dev, sr_stage_times* out)
{
    if (!dev || !out) return fail(SR_ERR_INVALID_ARGUMENT, "NULL argument");
    std::memset(out, 0, sizeof *out);
    if (!dev->events_used) return fail(SR_ERR_INVALID_ARGUMENT, "no profiled draw: call sr_profile_enable before sr_draw");
    SR_HIP(hipSetDevice(dev->ordinal));
    SR_HIP(hipStreamSynchronize(dev->stream));
    float* f = &out->clear_ms;
    for (size_t p = 0; p < dev->events_used / kNumStageEvents; ++p) {
        hipEvent_t* ev = &dev->events[p * kNumStageEvents];
        for (int k = 0; k < 8; ++k) {
            float ms = 0.0f;
            SR_HIP(hipEventElapsedTime(&ms, ev[k], ev[k + 1]));
            f[k] += ms;
        }
        float ms = 0.0f;
        SR_HIP(hipEventElapsedTime(&ms, ev[0], ev[8]));
        out->total_ms += ms;
    }
    return SR_OK;
}

int sr_debug_vertex_stage(sr_device dev, sr_mesh mesh, const void* ubo_block, int width, int height, float* clip_out, void* screen_out)
{
    if (!dev || !mesh || !ubo_block) return fail(SR_ERR_INVALID_ARGUMENT, "NULL argument");
    if (width <= 0 || height <= 0) return fail(SR_ERR_INVALID_ARGUMENT, "bad frame size");
    SR_HIP(hipSetDevice(dev->ordinal));
    const uint32_t n = mesh->n_verts;
    if (!n) return SR_OK;
    ViewXform x{};
    const float* u = (const float*)ubo_block;
    std::memcpy(x.model, u, 64);
    mat4_mul_mat4_host(u + 32, u + 16, x.pv);
    std::memcpy(x.cam, u + 48, 12);
    void *d_view = nullptr, *d_screen = nullptr, *d_clip = nullptr, *d_planes = nullptr;
    hipError_t e = hipMalloc(&d_view, sizeof x);
    if (e == hipSuccess) e = hipMalloc(&d_screen, (size_t)n * 16);
    if (e == hipSuccess) e = hipMalloc(&d_planes, (size_t)plane_stride(n) * 12);
    if (e == hipSuccess) e = hipMalloc(&d_clip, (size_t)n * 16);
    if (e == hipSuccess) e = hipMemcpy(d_view, &x, sizeof x, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        ScreenPlanes sp{};
        sp.stride = plane_stride(n); sp.xy = (int2*)d_planes; sp.z = (float*)((char*)d_planes + (size_t)sp.stride * 8);
        launch_vertex(dev->stream, mesh->px, mesh->py, mesh->pz, n, (const ViewXform*)d_view, 1, width, height, sp, (ScreenVertex*)d_screen, (float*)d_clip);
        e = hipStreamSynchronize(dev->stream);
    }
    if (e == hipSuccess && clip_out) e = hipMemcpy(clip_out, d_clip, (size_t)n * 16, hipMemcpyDeviceToHost);
    if (e == hipSuccess && screen_out) e = hipMemcpy(screen_out, d_screen, (size_t)n * 16, hipMemcpyDeviceToHost);
    (void)hipFree(d_view); (void)hipFree(d_screen); (void)hipFree(d_clip); (void)hipFree(d_planes);
    SR_HIP(e);
    return SR_OK;
}

}  // extern "C"
