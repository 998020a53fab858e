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
constexpr int kCountersPerPass = 8;  // total refs, dropped refs, unprojectable triangles, binned triangles, micro-triangles, spare

// Pipeline variants that can be switched on per device with the environment variable SR_VARIANT
// (comma-separated names).  They produce the same frames as the default pipeline (the parity tests
// run with each of them); they exist so that alternatives can be timed against each other on the GPU.
enum : unsigned {
    kVariantFusedGeometry = 1u << 0,   // "fused_geometry": triangle set-up transforms the vertices itself, no vertex pass
    kVariantDirtyBlocks = 1u << 1,     // "dirty_blocks": per-block dirty map; only written blocks are cleared and read back by shading
                                       //                 (the marking lives in the fused set-up kernel, so this implies fused_geometry)
    kVariantExactShading = 1u << 2,    // "exact_shading": the shading pass recomputes window-space vertices bit-identically to the vertex stage
    kVariantSparseTail = 1u << 3,      // "sparse_tail": a target whose last frame left few triangles for the tile lists gets one kernel
                                       //                in place of bin + scan + fill + raster (implies fused_geometry)
    kVariantSparseTailForce = 1u << 4, // "sparse_tail_force": always take that kernel, whatever the last frame was (for tests)
    kVariantClusterCull = 1u << 5,     // "cluster_cull": skip 64-triangle clusters that provably draw nothing (sr_device.h; implies fused_geometry)
    kVariantAnisotropic = 1u << 7,     // "anisotropic": the sampler's maxAnisotropy (src/texture.cpp:255-256) is applied, by the method the
                                       //                Vulkan specification spells out (implies dirty_blocks, exact_shading, fused_geometry)
    kVariantRaster512 = 1u << 8,       // "raster_512": the 4-sample tile kernel runs 512 work-items per 32x32 tile; at 1024 it is capped at
                                       //               128 registers and spills 500 bytes per work-item
    kVariantNearClip = 1u << 9,        // "near_clip": triangles with a vertex that cannot be projected are clipped against the near plane
                                       //              (sr_device.h) instead of dropped (implies exact_shading)
    kVariantClusterVerify = 1u << 6,   // "cluster_cull_verify": evaluate the cluster test but skip nothing; count what skipped clusters
                                       //                        would have drawn into sr_draw_stats.cull_violations (for tests)
};
// a frame qualifies for the sparse tail when its predecessor stayed under these
constexpr uint32_t kSparseMaxPending = 4096, kSparseMaxRefs = 16384, kSparseMaxSteps = 65536;

unsigned parse_variants(const char* text)
{
    unsigned v = 0;
    const std::string s = text ? text : "";
    size_t pos = 0;
    while (pos <= s.size()) {
        const size_t end = std::min(s.find(',', pos), s.size());
        const std::string name = s.substr(pos, end - pos);
        if (name == "fused_geometry") v |= kVariantFusedGeometry;
        if (name == "dirty_blocks") v |= kVariantDirtyBlocks | kVariantFusedGeometry;
        if (name == "exact_shading") v |= kVariantExactShading;
        if (name == "raster_512") v |= kVariantRaster512;
        if (name == "near_clip") v |= kVariantNearClip | kVariantExactShading;
        if (name == "anisotropic") v |= kVariantAnisotropic | kVariantExactShading | kVariantDirtyBlocks | kVariantFusedGeometry;
        if (name == "cluster_cull") v |= kVariantClusterCull | kVariantFusedGeometry;
        if (name == "cluster_cull_verify") v |= kVariantClusterVerify | kVariantClusterCull | kVariantFusedGeometry;
        if (name == "sparse_tail") v |= kVariantSparseTail | kVariantFusedGeometry;
        if (name == "sparse_tail_force") v |= kVariantSparseTailForce | kVariantSparseTail | kVariantFusedGeometry;
        pos = end + 1;
    }
    return v;
}

}  // namespace

struct sr_device_t {
    int ordinal = 0;
    hipStream_t stream = nullptr;
    float* d_decode = nullptr;   // 256
    float* d_thresh = nullptr;   // 255 (+1 pad)
    // scratch reused by every draw on this device (in-order on `stream`)
    DeviceBuffer screen, tile_count, tile_offset, survivors, survivor_count, refs, views, counters;
    DeviceBuffer clip;           // kVariantNearClip: [state: 4 words][entries: capacity x 8 bytes]
    uint32_t clip_capacity = 65536;   // triangles per pass that can be clipped; more are dropped and counted
    // pinned staging for the per-view transforms: a small ring, so that back-to-back draws do not
    // wait for each other on the host
    struct Staging { ViewXform* h = nullptr; void* d = nullptr; size_t cap = 0 /* bytes */; hipEvent_t ev = nullptr; bool in_flight = false; };   // h: pinned + mapped, d: its device address
    Staging staging[4];
    unsigned staging_next = 0;
    uint64_t ref_capacity = 0;
    size_t scratch_budget = (size_t)16 << 30;
    int tile_log2_override = 0;
    unsigned variants = 0;       // kVariant* bits, from SR_VARIANT
    uint32_t* d_cull_violations = nullptr;   // kVariantClusterVerify: one counter, accumulated over the passes of a draw
    bool profiling = false;
    std::vector<hipEvent_t> events;   // kNumStageEvents per pass of the last profiled draw
    size_t events_used = 0;
};

struct sr_mesh_t {
    void* d_aos = nullptr;
    void* d_pos4 = nullptr;      // packed (x, y, z, 1) positions; only with kVariantFusedGeometry
    ClusterRec* d_clusters = nullptr;   // kVariantClusterCull: one record per kClusterTris consecutive triangles
    uint32_t n_clusters = 0;
    float *px = nullptr, *py = nullptr, *pz = nullptr;
    uint32_t* d_idx = nullptr;
    uint32_t n_verts = 0, n_tris = 0;
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
    DirtyMap dirty{nullptr, 0, 0};   // kVariantDirtyBlocks: flags for all views of the target
    uint32_t* d_color_own = nullptr;
    uint32_t* d_color = nullptr;
    // the last draw, kept so that it can be re-issued if the reference list was too small
    bool drawn = false, check_pending = false;
    sr_mesh mesh = nullptr;
    sr_texture tex[4] = {nullptr, nullptr, nullptr, nullptr};
    std::vector<ViewXform> views;
    std::vector<ViewCull> cull_views;   // kVariantClusterCull: one per view, staged behind the transforms
    uint32_t passes = 0;
    uint32_t* h_counters = nullptr; void* hd_counters = nullptr; size_t h_counters_cap = 0;   // pinned + mapped, [passes][8], written by the scan kernel of the last draw
    uint64_t capacity_at_draw = 0;
    uint64_t total_refs = 0, unprojectable = 0, binned = 0, micro = 0, clipped = 0;
    bool sparse = false;        // kVariantSparseTail: the draw being issued (and then: the last draw) uses the sparse tail
    uint64_t cull_violations = 0;   // kVariantClusterVerify: of the last draw
};

namespace {

double srgb_decode_d(double x) { return x <= 0.04045 ? x / 12.92 : std::pow((x + 0.055) / 1.055, 2.4); }

void mat4_mul_mat4_host(const float* A, const float* B, float* R)
{
    for (int c = 0; c < 4; ++c) mat4_mul_vec4(A, B[4 * c], B[4 * c + 1], B[4 * c + 2], B[4 * c + 3], R + 4 * c);
}

TexDesc make_tex_desc(const sr_texture_t* t)
{
    TexDesc d{};
    d.texels = t->d_chain; d.w = t->w; d.h = t->h; d.levels = t->levels;
    d.min_lod = t->samp.min_lod;
    d.max_lod = t->samp.max_lod < 0.0f ? (float)t->levels : t->samp.max_lod;
    d.bias = t->samp.mip_lod_bias;
    std::memcpy(d.level_off, t->level_off, sizeof d.level_off);
    return d;
}

int choose_tile_log2(const sr_device_t* dev, int samples)
{
    if (dev->tile_log2_override && raster_config_supported(samples, dev->tile_log2_override)) return dev->tile_log2_override;
    return samples == 1 ? 6 : 5;
}

// per-view scratch bytes of a pass
size_t per_view_bytes(const sr_mesh_t* m, const FrameGeom& g)
{
    return (size_t)m->n_verts * sizeof(ScreenVertex) + ((size_t)m->n_tris + 1024) * (4 * sizeof(BinRecord) + 4) + (size_t)m->n_tris / 64 + (size_t)g.tiles_x * g.tiles_y * 16 + 64;
}

uint32_t views_per_pass(const sr_device_t* dev, const sr_mesh_t* m, const FrameGeom& g, uint32_t n_views)
{
    size_t pv = per_view_bytes(m, g);
    uint64_t by_mem = std::max<uint64_t>(1, dev->scratch_budget / std::max<size_t>(pv, 1));
    uint64_t by_refs = std::max<uint64_t>(1, (uint64_t)0x7FFFFFFF / std::max<uint32_t>(m->n_tris, 1));
    uint64_t v = std::min<uint64_t>({(uint64_t)n_views, by_mem, by_refs, (uint64_t)65535});
    return (uint32_t)std::max<uint64_t>(v, 1);
}

// One record per kClusterTris consecutive triangles (sr_device.h).  Double precision; the float results are
// widened on the safe side (larger sphere, wider cone, smaller inradius and perimeter).
std::vector<ClusterRec> build_clusters(const float* verts /* 8 floats per vertex */, const uint32_t* idx, uint32_t n_tris)
{
    const uint32_t n_clusters = (n_tris + kClusterTris - 1) / kClusterTris;
    std::vector<ClusterRec> out(n_clusters);
    std::vector<double> normals;
    for (uint32_t c = 0; c < n_clusters; ++c) {
        const uint32_t t0 = c * kClusterTris, t1 = std::min(n_tris, t0 + kClusterTris);
        double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
        for (uint32_t t = t0; t < t1; ++t)
            for (int k = 0; k < 3; ++k) {
                const float* p = verts + 8 * (size_t)idx[3 * (size_t)t + k];
                for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], (double)p[a]); hi[a] = std::max(hi[a], (double)p[a]); }
            }
        const double ctr[3] = {0.5 * (lo[0] + hi[0]), 0.5 * (lo[1] + hi[1]), 0.5 * (lo[2] + hi[2])};
        double r2 = 0.0, sum[3] = {0, 0, 0}, rho_min = 1e300, p_min = 1e300;
        bool usable = true;
        normals.clear();
        for (uint32_t t = t0; t < t1; ++t) {
            const float* pf[3];
            double p[3][3];
            for (int k = 0; k < 3; ++k) {
                pf[k] = verts + 8 * (size_t)idx[3 * (size_t)t + k];
                for (int a = 0; a < 3; ++a) p[k][a] = pf[k][a];
                const double d2 = (p[k][0] - ctr[0]) * (p[k][0] - ctr[0]) + (p[k][1] - ctr[1]) * (p[k][1] - ctr[1]) + (p[k][2] - ctr[2]) * (p[k][2] - ctr[2]);
                r2 = std::max(r2, d2);
            }
            auto same = [&](int i, int j) { return pf[i][0] == pf[j][0] && pf[i][1] == pf[j][1] && pf[i][2] == pf[j][2]; };
            if (same(0, 1) || same(1, 2) || same(0, 2)) continue;   // two corners coincide: the snapped triangle has zero area whatever the view
            const double e1[3] = {p[1][0] - p[0][0], p[1][1] - p[0][1], p[1][2] - p[0][2]}, e2[3] = {p[2][0] - p[0][0], p[2][1] - p[0][1], p[2][2] - p[0][2]};
            const double e3[3] = {p[2][0] - p[1][0], p[2][1] - p[1][1], p[2][2] - p[1][2]};
            const double n[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
            const double a2 = std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
            const double per = std::sqrt(e1[0] * e1[0] + e1[1] * e1[1] + e1[2] * e1[2]) + std::sqrt(e2[0] * e2[0] + e2[1] * e2[1] + e2[2] * e2[2]) +
                               std::sqrt(e3[0] * e3[0] + e3[1] * e3[1] + e3[2] * e3[2]);
            if (!(a2 > 1e-12 * per * per)) { usable = false; continue; }   // collinear corners: facing is decided by rounding, never skip
            for (int a = 0; a < 3; ++a) { normals.push_back(n[a] / a2); sum[a] += n[a] / a2; }
            rho_min = std::min(rho_min, a2 / per);
            p_min = std::min(p_min, per);
        }
        ClusterRec& rec = out[c];
        rec.cx = (float)ctr[0]; rec.cy = (float)ctr[1]; rec.cz = (float)ctr[2];
        rec.r = (float)(std::sqrt(r2) * 1.00001 + 1e-30) + 1e-6f * (std::fabs(rec.cx) + std::fabs(rec.cy) + std::fabs(rec.cz));   // covers the rounding of the centre
        rec.ax = rec.ay = rec.az = 0.0f; rec.cos_alpha = -2.0f; rec.sin_alpha = 0.0f; rec.rho_min = 0.0f; rec.p_min = 1.0f; rec.pad = 0.0f;
        const double sl = std::sqrt(sum[0] * sum[0] + sum[1] * sum[1] + sum[2] * sum[2]);
        if (usable && !normals.empty() && sl > 1e-3 * (double)(normals.size() / 3)) {
            const double ax[3] = {sum[0] / sl, sum[1] / sl, sum[2] / sl};
            double cmin = 1.0;
            for (size_t i = 0; i < normals.size(); i += 3) cmin = std::min(cmin, normals[i] * ax[0] + normals[i + 1] * ax[1] + normals[i + 2] * ax[2]);
            cmin -= 1e-5;   // the float axis and the device's float arithmetic
            if (cmin > 0.05) {
                rec.ax = (float)ax[0]; rec.ay = (float)ax[1]; rec.az = (float)ax[2];
                rec.cos_alpha = (float)cmin; rec.sin_alpha = (float)std::sqrt(1.0 - cmin * cmin) * 1.00001f;
                rec.rho_min = (float)(rho_min * 0.9999); rec.p_min = (float)(p_min * 0.9999);
            }
        }
    }
    return out;
}

// per-view constants of the cluster test, from the same float matrices the kernels use
ViewCull make_view_cull(const ViewXform& x)
{
    double M[16], PV[16], T[16];
    for (int i = 0; i < 16; ++i) { M[i] = x.model[i]; PV[i] = x.pv[i]; }
    for (int c = 0; c < 4; ++c)
        for (int r = 0; r < 4; ++r) { double s = 0; for (int k = 0; k < 4; ++k) s += PV[4 * k + r] * M[4 * c + k]; T[4 * c + r] = s; }   // column-major PV * M
    auto row = [&](int r, double* v, double& t) { v[0] = T[r]; v[1] = T[4 + r]; v[2] = T[8 + r]; t = T[12 + r]; };
    double R0[3], R1[3], R3[3], t0, t1, t3;
    row(0, R0, t0); row(1, R1, t1); row(3, R3, t3);
    const double det = R0[0] * (R1[1] * R3[2] - R1[2] * R3[1]) - R0[1] * (R1[0] * R3[2] - R1[2] * R3[0]) + R0[2] * (R1[0] * R3[1] - R1[1] * R3[0]);
    auto len = [](const double* v) { return std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); };
    ViewCull c{};
    for (int a = 0; a < 3; ++a) { c.R0[a] = (float)R0[a]; c.R1[a] = (float)R1[a]; c.R3[a] = (float)R3[a]; }
    c.t0 = (float)t0; c.t1 = (float)t1; c.t3 = (float)t3;
    c.nR0 = (float)(len(R0) * 1.000001); c.nR1 = (float)(len(R1) * 1.000001); c.nR3 = (float)(len(R3) * 1.000001);
    c.sigma = det < 0 ? -1.0f : 1.0f;
    c.absdet = (float)(std::fabs(det) * 0.999999);
    c.cone_ok = 0.0f;
    const double scale = len(R0) * len(R1) * len(R3);
    if (std::fabs(det) > 1e-9 * scale && scale > 0) {
        // centre of projection: P e + t = 0 (Cramer)
        const double b0 = -t0, b1 = -t1, b3 = -t3;
        const double ex = (b0 * (R1[1] * R3[2] - R1[2] * R3[1]) - R0[1] * (b1 * R3[2] - R1[2] * b3) + R0[2] * (b1 * R3[1] - R1[1] * b3)) / det;
        const double ey = (R0[0] * (b1 * R3[2] - R1[2] * b3) - b0 * (R1[0] * R3[2] - R1[2] * R3[0]) + R0[2] * (R1[0] * b3 - b1 * R3[0])) / det;
        const double ez = (R0[0] * (R1[1] * b3 - b1 * R3[1]) - R0[1] * (R1[0] * b3 - b1 * R3[0]) + b0 * (R1[0] * R3[1] - R1[1] * R3[0])) / det;
        if (std::isfinite(ex) && std::isfinite(ey) && std::isfinite(ez) && std::fabs(ex) + std::fabs(ey) + std::fabs(ez) < 1e15) {
            c.e[0] = (float)ex; c.e[1] = (float)ey; c.e[2] = (float)ez; c.cone_ok = 1.0f;
        }
    }
    return c;
}

// issues every kernel of one draw (all passes) on the device's stream
int issue_draw(sr_device_t* dev, sr_target_t* tg)
{
    const sr_mesh_t* m = tg->mesh;
    const FrameGeom& g = tg->g;
    const uint32_t n_views = (uint32_t)tg->views.size();
    const uint32_t vpp = views_per_pass(dev, m, g, n_views);
    const uint32_t passes = (n_views + vpp - 1) / vpp;
    const uint32_t n_tiles = (uint32_t)(g.tiles_x * g.tiles_y);
    hipStream_t st = dev->stream;

    if (dev->ref_capacity == 0) dev->ref_capacity = std::max<uint64_t>((uint64_t)m->n_tris * vpp, (uint64_t)1 << 20);
    dev->ref_capacity = std::min<uint64_t>(dev->ref_capacity, 0xFFFFFFFFull);

    // three per-tile counter arrays of a pass, cleared by one memset; [block_count: vpp*n_blocks]
    // follows and is fully written by the set-up kernel
    const uint32_t bt = setup_block_triangles();
    const uint32_t n_blocks = (m->n_tris + bt - 1u) / bt;
    const size_t count_words = (size_t)vpp * n_tiles * 3;
    SR_HIP(dev->screen.ensure((size_t)vpp * std::max<uint32_t>(m->n_verts, 1) * sizeof(ScreenVertex)));
    // layout of the binning state: [count_small | count_large | cursor_large | counters of this pass] (zeroed at the
    // start of every pass, padded to 16 bytes) [block_count | block_live | pending_count]
    const size_t zero_words = (count_words + kCountersPerPass + 3) & ~(size_t)3;
    SR_HIP(dev->tile_count.ensure((zero_words + 3 * (size_t)vpp * std::max(n_blocks, 1u)) * 4));
    SR_HIP(dev->survivor_count.ensure((size_t)vpp * std::max(n_blocks, 1u) * bt * 4));
    SR_HIP(dev->tile_offset.ensure(((size_t)vpp * n_tiles + 1) * 4));
    SR_HIP(dev->survivors.ensure((size_t)vpp * std::max(n_blocks, 1u) * setup_block_records() * sizeof(BinRecord)));
    SR_HIP(dev->refs.ensure((size_t)dev->ref_capacity * 4));
    const bool near_clip = (dev->variants & kVariantNearClip) != 0;
    if (near_clip) SR_HIP(dev->clip.ensure(16 + (size_t)dev->clip_capacity * sizeof(uint2)));
    const bool culling = (dev->variants & kVariantClusterCull) && m->d_clusters && m->d_pos4 && tg->cull_views.size() == n_views;
    const size_t view_bytes = (size_t)n_views * (sizeof(ViewXform) + (culling ? sizeof(ViewCull) : 0));   // [transforms][cull constants]
    SR_HIP(dev->views.ensure(view_bytes));
    sr_device_t::Staging& sg = dev->staging[dev->staging_next++ & 3u];
    if (!sg.ev) SR_HIP(hipEventCreateWithFlags(&sg.ev, hipEventDisableTiming));
    if (sg.in_flight) { SR_HIP(hipEventSynchronize(sg.ev)); sg.in_flight = false; }
    if (sg.cap < view_bytes) {
        if (sg.h) SR_HIP(hipHostFree(sg.h));
        sg.h = nullptr; sg.cap = 0;
        SR_HIP(hipHostMalloc((void**)&sg.h, view_bytes, hipHostMallocMapped));
        SR_HIP(hipHostGetDevicePointer(&sg.d, sg.h, 0));
        sg.cap = view_bytes;
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

    std::memcpy(sg.h, tg->views.data(), (size_t)n_views * sizeof(ViewXform));
    if (culling) std::memcpy((char*)sg.h + (size_t)n_views * sizeof(ViewXform), tg->cull_views.data(), (size_t)n_views * sizeof(ViewCull));
    if (culling && dev->d_cull_violations) SR_HIP(hipMemsetAsync(dev->d_cull_violations, 0, 4, st));

    ScreenPlanes sp{};
    sp.xy = (int2*)dev->screen.ptr;
    sp.zw = (float2*)((char*)dev->screen.ptr + (size_t)vpp * std::max<uint32_t>(m->n_verts, 1) * sizeof(int2));
    sp.n_verts = m->n_verts;
    BinBuffers b{};
    b.count_small = (uint32_t*)dev->tile_count.ptr;
    b.count_large = b.count_small + (size_t)vpp * n_tiles;
    b.cursor_large = b.count_large + (size_t)vpp * n_tiles;
    b.counters = b.count_small + count_words;
    b.block_count = b.count_small + zero_words;
    b.block_live = b.block_count + (size_t)vpp * std::max(n_blocks, 1u);
    b.pending_count = b.block_live + (size_t)vpp * std::max(n_blocks, 1u);
    b.pending = (uint32_t*)dev->survivor_count.ptr;
    b.n_blocks = n_blocks;
    {   // stride close to the golden ratio of n_blocks and coprime with it
        auto gcd = [](uint32_t a, uint32_t c) { while (c) { uint32_t t = a % c; a = c; c = t; } return a; };
        uint32_t stride = (uint32_t)((double)n_blocks * 0.6180339887498949) | 1u;
        while (n_blocks > 2 && gcd(stride, n_blocks) != 1u) stride += 2u;
        b.block_stride = n_blocks > 2 ? stride % n_blocks : 1u;
        if (b.block_stride == 0) b.block_stride = 1u;
    }
    b.tile_offset = (uint32_t*)dev->tile_offset.ptr;
    b.refs = (uint32_t*)dev->refs.ptr;
    b.ref_capacity = (uint32_t)dev->ref_capacity;
    b.records = (BinRecord*)dev->survivors.ptr;

    const sr_texture_t* albedo = tg->tex[0];
    const sr_texture_t* metallic = tg->tex[2];
    const sr_texture_t* roughness = tg->tex[3];

    for (uint32_t pass = 0; pass < passes; ++pass) {
        const uint32_t v0 = pass * vpp, nv = std::min(vpp, n_views - v0);
        const ViewXform* d_views = (const ViewXform*)dev->views.ptr + v0;
        hipEvent_t* ev = dev->profiling ? &dev->events[(size_t)pass * kNumStageEvents] : nullptr;
        if (ev) SR_HIP(hipEventRecord(ev[0], st));
        // the dirty map needs the set-up kernel that marks it: meshes created without the packed positions fall back to the full clear
        const bool fused = (dev->variants & kVariantFusedGeometry) && m->d_pos4;
        const DirtyMap dirty = fused && tg->dirty.flags ? tg->dirty : DirtyMap{nullptr, 0, 0};
        if (dirty.flags)
            launch_begin_pass_dirty(st, tg->d_vis, g, v0, nv, dirty, b.count_small, zero_words * 4,
                                    sg.d, dev->views.ptr, pass == 0 ? view_bytes : 0);
        else
            launch_begin_pass(st, tg->d_vis + (size_t)v0 * g.rw * g.rh * g.samples, (size_t)nv * g.rw * g.rh * g.samples, b.count_small, zero_words * 4,
                              sg.d, dev->views.ptr, pass == 0 ? view_bytes : 0);
        if (pass == 0) { SR_HIP(hipEventRecord(sg.ev, st)); sg.in_flight = true; }
        if (ev) SR_HIP(hipEventRecord(ev[1], st));
        if (!fused) launch_vertex(st, m->px, m->py, m->pz, m->n_verts, d_views, nv, g.W, g.H, sp, nullptr, nullptr);
        if (ev) SR_HIP(hipEventRecord(ev[2], st));
        const bool sparse = fused && tg->sparse && m->n_tris > 0;
        ClusterCull cull{nullptr, nullptr, nullptr};
        if (culling) {
            cull.clusters = m->d_clusters;
            cull.views = (const ViewCull*)((const char*)dev->views.ptr + (size_t)n_views * sizeof(ViewXform)) + v0;
            cull.violations = (dev->variants & kVariantClusterVerify) ? dev->d_cull_violations : nullptr;
        }
        ClipList clip{nullptr, nullptr, 0, nullptr};
        if (near_clip) {
            clip.state = (uint32_t*)dev->clip.ptr;
            clip.entries = (uint2*)((char*)dev->clip.ptr + 16);
            clip.capacity = dev->clip_capacity;
            clip.count = b.counters + 7;   // zeroed with the other counters at the start of the pass
            SR_HIP(hipMemsetAsync(clip.state, 0, 16, st));
        }
        launch_setup(st, m->d_idx, m->n_tris, sp, g, b, nv, tg->d_vis, v0, ev ? ev[3] : nullptr, fused ? m->d_pos4 : nullptr, d_views, dirty, !sparse, cull,
                     fused ? clip : ClipList{nullptr, nullptr, 0, nullptr});
        if (ev && !m->n_tris) SR_HIP(hipEventRecord(ev[3], st));
        if (ev) SR_HIP(hipEventRecord(ev[4], st));
        uint32_t* pass_counters = (uint32_t*)tg->hd_counters + (size_t)pass * kCountersPerPass;
        // (before the scan / sparse-tail kernel, which reports the number of dropped triangles to the host)
        if (clip.entries) launch_near_clip(st, m->d_idx, m->n_tris, fused ? m->d_pos4 : nullptr, m->px, m->py, m->pz, sp, nv, d_views, g, clip, b, tg->d_vis, v0, dirty, pass_counters);
        if (sparse) {   // one kernel resolves whatever set-up left pending; its time is reported as the raster stage
            if (ev) { SR_HIP(hipEventRecord(ev[5], st)); SR_HIP(hipEventRecord(ev[6], st)); }
            launch_sparse_tail(st, m->d_idx, m->n_tris, sp, g, b, tg->d_vis, v0, nv, dirty, pass_counters);
        } else {
            launch_scan(st, b, nv * n_tiles, nv, pass_counters);
            if (ev) SR_HIP(hipEventRecord(ev[5], st));
            launch_fill(st, g, b, m->n_tris, nv);
            if (ev) SR_HIP(hipEventRecord(ev[6], st));
            launch_raster(st, m->d_idx, m->n_tris, sp, g, b, tg->d_vis, v0, nv, dirty, (dev->variants & kVariantRaster512) != 0);
        }
        if (ev) SR_HIP(hipEventRecord(ev[7], st));
        FragLaunch f{};
        f.idx = m->d_idx; f.verts_aos = m->d_aos; f.screen = sp; f.views = d_views; f.vis = tg->d_vis; f.color = tg->d_color; f.g = g;
        f.albedo = make_tex_desc(albedo); f.metallic = make_tex_desc(metallic); f.roughness = make_tex_desc(roughness);
        f.decode = dev->d_decode; f.thresh = dev->d_thresh; f.view0 = v0; f.n_views = nv; f.dirty = dirty;
        f.exact_vertices = (dev->variants & kVariantExactShading) != 0;
        f.max_anisotropy = (dev->variants & kVariantAnisotropic) ? std::max(albedo->samp.max_anisotropy, 1.0f) : 1.0f;
        launch_fragment(st, f);
        if (ev) SR_HIP(hipEventRecord(ev[8], st));
    }
    SR_HIP(hipGetLastError());
    tg->passes = passes;
    tg->capacity_at_draw = dev->ref_capacity;
    tg->check_pending = true;
    return SR_OK;
}

// waits for the target's last draw; re-issues it once with a larger reference list if it overflowed
int finish_target(sr_device_t* dev, sr_target_t* tg)
{
    SR_HIP(hipStreamSynchronize(dev->stream));
    if (!tg->check_pending) return SR_OK;
    for (int attempt = 0; attempt < 2; ++attempt) {
        uint64_t worst = 0, total = 0, unproj = 0, binned = 0, micro = 0, clipped = 0;
        const bool near_clip = (dev->variants & kVariantNearClip) != 0;
        for (uint32_t p = 0; p < tg->passes; ++p) {
            const uint32_t* c = tg->h_counters + (size_t)p * kCountersPerPass;
            worst = std::max<uint64_t>(worst, c[0]); total += c[0]; unproj += c[2]; binned += c[3]; micro += c[4];
            if (near_clip) clipped += c[6];
        }
        tg->total_refs = total; tg->unprojectable = unproj; tg->binned = binned; tg->micro = micro; tg->clipped = clipped;
        if (worst <= tg->capacity_at_draw) { tg->check_pending = false; return SR_OK; }
        if (attempt == 1) break;
        dev->ref_capacity = std::min<uint64_t>(worst + worst / 4 + 1024, 0xFFFFFFFFull);
        int rc = issue_draw(dev, tg);
        if (rc != SR_OK) return rc;
        SR_HIP(hipStreamSynchronize(dev->stream));
    }
    return fail(SR_ERR_UNSUPPORTED, "tile reference list exceeds 2^32 entries in one pass");
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
    SR_HIP(hipStreamCreateWithFlags(&dev->stream, hipStreamNonBlocking));
    float dec[256], thr[256];
    for (int c = 0; c < 256; ++c) dec[c] = (float)srgb_decode_d(c / 255.0);
    for (int c = 0; c < 255; ++c) thr[c] = (float)srgb_decode_d((c + 0.5) / 255.0);
    thr[255] = 3.0e38f;
    SR_HIP(hipMalloc((void**)&dev->d_decode, sizeof dec));
    SR_HIP(hipMalloc((void**)&dev->d_thresh, sizeof thr));
    SR_HIP(hipMemcpy(dev->d_decode, dec, sizeof dec, hipMemcpyHostToDevice));
    SR_HIP(hipMemcpy(dev->d_thresh, thr, sizeof thr, hipMemcpyHostToDevice));
    if (const char* e = std::getenv("SR_TILE_LOG2")) dev->tile_log2_override = std::atoi(e);
    if (const char* e = std::getenv("SR_SCRATCH_GIB")) dev->scratch_budget = (size_t)std::max(1, std::atoi(e)) << 30;
    if (const char* e = std::getenv("SR_REF_CAPACITY")) dev->ref_capacity = (uint64_t)std::max(1, std::atoi(e));   // initial size of the tile reference list (tests: force the grow-and-redraw path)
    if (const char* e = std::getenv("SR_CLIP_CAPACITY")) dev->clip_capacity = (uint32_t)std::max(1, std::atoi(e));   // tests: a full clip list
    if (const char* e = std::getenv("SR_SCRATCH_MIB")) dev->scratch_budget = (size_t)std::max(1, std::atoi(e)) << 20;   // small budgets: many passes per draw
    dev->variants = parse_variants(std::getenv("SR_VARIANT"));
    if (dev->variants & kVariantClusterVerify) SR_HIP(hipMalloc((void**)&dev->d_cull_violations, 4));
    *out = dev;
    return SR_OK;
}

int sr_device_destroy(sr_device dev)
{
    if (!dev) return SR_OK;
    (void)hipSetDevice(dev->ordinal);
    (void)hipStreamSynchronize(dev->stream);
    for (hipEvent_t e : dev->events) (void)hipEventDestroy(e);
    dev->screen.release(); dev->tile_count.release(); dev->tile_offset.release(); dev->survivors.release();
    dev->survivor_count.release(); dev->refs.release(); dev->views.release(); dev->counters.release(); dev->clip.release();
    for (auto& sg : dev->staging) { if (sg.h) (void)hipHostFree(sg.h); if (sg.ev) (void)hipEventDestroy(sg.ev); }
    (void)hipFree(dev->d_decode); (void)hipFree(dev->d_thresh); (void)hipFree(dev->d_cull_violations);
    (void)hipStreamDestroy(dev->stream);
    delete dev;
    return SR_OK;
}

int sr_device_synchronize(sr_device dev)
{
    if (!dev) return fail(SR_ERR_INVALID_ARGUMENT, "device is NULL");
    SR_HIP(hipSetDevice(dev->ordinal));
    SR_HIP(hipStreamSynchronize(dev->stream));
    return SR_OK;
}

// ---- mesh ---------------------------------------------------------------------------------------
int sr_mesh_create(sr_device dev, const void* vertices, uint32_t n_vertices, const uint32_t* indices, uint32_t n_indices, sr_mesh* out)
{
    if (!dev || !out) return fail(SR_ERR_INVALID_ARGUMENT, "device or out is NULL");
    *out = nullptr;
    if ((n_vertices && !vertices) || (n_indices && !indices)) return fail(SR_ERR_INVALID_ARGUMENT, "NULL data with a non-zero count");
    const uint32_t n_tris = n_indices / 3;   // an incomplete trailing primitive is not drawn
    for (size_t i = 0; i < (size_t)n_tris * 3; ++i)
        if (indices[i] >= n_vertices) return fail(SR_ERR_INDEX_OUT_OF_RANGE, "index " + std::to_string(i) + " = " + std::to_string(indices[i]) + " >= " + std::to_string(n_vertices));
    SR_HIP(hipSetDevice(dev->ordinal));
    sr_mesh_t* m = new sr_mesh_t();
    m->n_verts = n_vertices; m->n_tris = n_tris;
    const size_t nv = std::max<uint32_t>(n_vertices, 1), ni = std::max<uint32_t>(n_tris * 3, 1);
    hipError_t e = hipMalloc(&m->d_aos, nv * 32);
    if (e == hipSuccess) e = hipMalloc((void**)&m->px, nv * 12);
    if (e == hipSuccess) e = hipMalloc((void**)&m->d_idx, ni * 4);
    if (e == hipSuccess && (dev->variants & kVariantFusedGeometry)) e = hipMalloc(&m->d_pos4, nv * 16);
    if (e != hipSuccess) { (void)hipFree(m->d_aos); (void)hipFree(m->px); (void)hipFree(m->d_idx); (void)hipFree(m->d_pos4); delete m; SR_HIP(e); }
    m->py = m->px + nv; m->pz = m->py + nv;
    if (n_vertices) e = hipMemcpyAsync(m->d_aos, vertices, (size_t)n_vertices * 32, hipMemcpyHostToDevice, dev->stream);
    if (e == hipSuccess && n_tris) e = hipMemcpyAsync(m->d_idx, indices, (size_t)n_tris * 12, hipMemcpyHostToDevice, dev->stream);
    if (e == hipSuccess) {
        launch_split_positions(dev->stream, m->d_aos, n_vertices, m->px, m->py, m->pz);
        if (m->d_pos4) launch_pack_positions(dev->stream, m->d_aos, n_vertices, m->d_pos4);
    }
    if (e == hipSuccess && (dev->variants & kVariantClusterCull) && n_tris) {
        const std::vector<ClusterRec> recs = build_clusters((const float*)vertices, indices, n_tris);
        m->n_clusters = (uint32_t)recs.size();
        e = hipMalloc((void**)&m->d_clusters, recs.size() * sizeof(ClusterRec));
        if (e == hipSuccess) e = hipMemcpy(m->d_clusters, recs.data(), recs.size() * sizeof(ClusterRec), hipMemcpyHostToDevice);
    }
    if (e == hipSuccess) e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(dev->stream);   // uploaded on return, like the reference's blocking staging copy
    if (e != hipSuccess) { (void)sr_mesh_destroy(dev, m); SR_HIP(e); }
    *out = m;
    return SR_OK;
}

int sr_mesh_destroy(sr_device dev, sr_mesh mesh)
{
    if (!mesh) return SR_OK;
    if (dev) { (void)hipSetDevice(dev->ordinal); (void)hipStreamSynchronize(dev->stream); }
    (void)hipFree(mesh->d_aos); (void)hipFree(mesh->px); (void)hipFree(mesh->d_idx); (void)hipFree(mesh->d_pos4); (void)hipFree(mesh->d_clusters);
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
    if (width <= 0 || height <= 0 || width > 32768 || height > 32768) return fail(SR_ERR_INVALID_ARGUMENT, "texture size out of range");
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
    if (dev) { (void)hipSetDevice(dev->ordinal); (void)hipStreamSynchronize(dev->stream); }
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
    g.tile_log2 = choose_tile_log2(dev, samples);
    g.tiles_x = (g.rw + (1 << g.tile_log2) - 1) >> g.tile_log2;
    g.tiles_y = (g.rh + (1 << g.tile_log2) - 1) >> g.tile_log2;
    if (g.tiles_x > 256 || g.tiles_y > 256) { delete t; return fail(SR_ERR_UNSUPPORTED, "more than 256 tiles along one axis"); }
    t->n_views = n_views;
    const size_t px = (size_t)g.rw * g.rh;
    hipError_t e = hipMalloc((void**)&t->d_vis, px * samples * 8 * n_views);
    if (e == hipSuccess) e = hipMalloc((void**)&t->d_color_own, px * 4 * n_views);
    if (e != hipSuccess) { (void)hipFree(t->d_vis); delete t; SR_HIP(e); }
    t->d_color = t->d_color_own;
    if (dev->variants & kVariantDirtyBlocks) {
        // the map's invariant -- an unmarked block holds only the clear value -- starts with one full clear
        t->dirty.nbx = (g.rw + (1 << kDirtyLog2W) - 1) >> kDirtyLog2W;
        t->dirty.nby = (g.rh + (1 << kDirtyLog2H) - 1) >> kDirtyLog2H;
        const size_t flag_bytes = (size_t)n_views * t->dirty.nbx * t->dirty.nby * 4;
        e = hipMalloc((void**)&t->dirty.flags, flag_bytes);
        if (e == hipSuccess) e = hipMemsetAsync(t->dirty.flags, 0, flag_bytes, dev->stream);
        if (e == hipSuccess) {
            launch_begin_pass(dev->stream, t->d_vis, px * samples * n_views, nullptr, 0, nullptr, nullptr, 0);
            e = hipStreamSynchronize(dev->stream);
        }
        if (e != hipSuccess) { (void)hipFree(t->d_vis); (void)hipFree(t->d_color_own); (void)hipFree(t->dirty.flags); delete t; SR_HIP(e); }
    }
    *out = t;
    return SR_OK;
}

int sr_target_destroy(sr_device dev, sr_target target)
{
    if (!target) return SR_OK;
    if (dev) { (void)hipSetDevice(dev->ordinal); (void)hipStreamSynchronize(dev->stream); }
    (void)hipFree(target->d_vis); (void)hipFree(target->d_color_own); (void)hipFree(target->dirty.flags);
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
int sr_draw(sr_device dev, sr_target target, sr_mesh mesh, const void* ubo_blocks, uint32_t n_views, size_t ubo_stride, const sr_texture textures[4])
{
    if (!dev || !target || !mesh || !ubo_blocks || !textures) return fail(SR_ERR_INVALID_ARGUMENT, "NULL argument");
    if (n_views == 0 || n_views > target->n_views) return fail(SR_ERR_INVALID_ARGUMENT, "n_views must be in 1..target views");
    if (ubo_stride < 204 || (ubo_stride & 3)) return fail(SR_ERR_INVALID_ARGUMENT, "ubo_stride must be >= 204 and a multiple of 4");
    if (!textures[0] || !textures[2] || !textures[3]) return fail(SR_ERR_INVALID_ARGUMENT, "albedo, metallic and roughness textures are required");
    SR_HIP(hipSetDevice(dev->ordinal));
    // a finished earlier draw tells how many tile references the scene needs: grow before reuse
    if (target->check_pending && hipStreamQuery(dev->stream) == hipSuccess) {
        for (uint32_t p = 0; p < target->passes; ++p)
            if (target->h_counters[(size_t)p * kCountersPerPass] > dev->ref_capacity) {
                const uint64_t w = target->h_counters[(size_t)p * kCountersPerPass];
                dev->ref_capacity = std::min<uint64_t>(w + w / 4 + 1024, 0xFFFFFFFFull);
            }
    }
    (void)hipGetLastError();
    if (dev->variants & kVariantSparseTail) {
        // Decide from the counters of this target's previous draw (pinned memory the kernels write; it may still be
        // in flight, in which case the figures are one frame older -- either way only speed depends on them).
        bool next = (dev->variants & kVariantSparseTailForce) != 0;
        if (!next && target->passes > 0 && target->mesh == mesh) {
            next = true;
            for (uint32_t p = 0; p < target->passes; ++p) {
                const uint32_t* c = target->h_counters + (size_t)p * kCountersPerPass;
                if (c[3] > kSparseMaxPending || (target->sparse ? c[5] > kSparseMaxSteps : c[0] > kSparseMaxRefs)) next = false;
            }
        }
        target->sparse = next;
    }
    target->mesh = mesh;
    for (int k = 0; k < 4; ++k) target->tex[k] = textures[k];
    target->views.resize(n_views);
    for (uint32_t v = 0; v < n_views; ++v) {
        const float* u = (const float*)((const char*)ubo_blocks + (size_t)v * ubo_stride);
        ViewXform& x = target->views[v];
        std::memcpy(x.model, u, 64);
        mat4_mul_mat4_host(u + 32, u + 16, x.pv);   // proj * view
        std::memcpy(x.cam, u + 48, 12);
        static const float identity[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
        x.model_is_identity = std::memcmp(x.model, identity, 64) == 0 ? 1.0f : 0.0f;   // lets the variants' kernels skip the product with it
    }
    target->cull_views.clear();
    if (dev->variants & kVariantClusterCull)
        for (uint32_t v = 0; v < n_views; ++v) target->cull_views.push_back(make_view_cull(target->views[v]));
    target->drawn = true;
    return issue_draw(dev, target);
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
    const sr_mesh_t* m = target->mesh;
    const uint32_t n_views = (uint32_t)target->views.size();
    std::memset(out, 0, sizeof *out);
    out->vertices = (uint64_t)m->n_verts * n_views;
    out->triangles = (uint64_t)m->n_tris * n_views;
    out->tile_refs = target->total_refs;
    out->triangles_unprojectable = target->unprojectable;
    out->pixels = (uint64_t)g.rw * g.rh * n_views;
    uint32_t* flags = nullptr; uint64_t* sums = nullptr;
    SR_HIP(hipMalloc((void**)&flags, (size_t)std::max<uint32_t>(m->n_tris, 1) * 4));
    hipError_t e = hipMalloc((void**)&sums, 16);
    if (e != hipSuccess) { (void)hipFree(flags); SR_HIP(e); }
    e = hipMemsetAsync(sums, 0, 16, dev->stream);
    const size_t px = (size_t)g.rw * g.rh;
    for (uint32_t v = 0; v < n_views && e == hipSuccess; ++v) {
        e = hipMemsetAsync(flags, 0, (size_t)std::max<uint32_t>(m->n_tris, 1) * 4, dev->stream);
        launch_stats(dev->stream, target->d_vis + px * g.samples * v, px, g.samples, flags, m->n_tris, sums, sums + 1);
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
    if ((dev->variants & kVariantClusterCull) && m->d_clusters && target->cull_views.size() == n_views) {
        // re-evaluate the cluster test for the draw's views (its device copy may have been overwritten by a later draw)
        ViewCull* d_cv = nullptr; uint64_t* d_total = nullptr; uint64_t total = 0;
        hipError_t ce = hipMalloc((void**)&d_cv, (size_t)n_views * sizeof(ViewCull));
        if (ce == hipSuccess) ce = hipMalloc((void**)&d_total, 8);
        if (ce == hipSuccess) ce = hipMemcpy(d_cv, target->cull_views.data(), (size_t)n_views * sizeof(ViewCull), hipMemcpyHostToDevice);
        if (ce == hipSuccess) ce = hipMemsetAsync(d_total, 0, 8, dev->stream);
        if (ce == hipSuccess) {
            launch_count_culled_clusters(dev->stream, ClusterCull{m->d_clusters, d_cv, nullptr}, g, m->n_clusters, n_views, d_total);
            ce = hipStreamSynchronize(dev->stream);
        }
        if (ce == hipSuccess) ce = hipMemcpy(&total, d_total, 8, hipMemcpyDeviceToHost);
        (void)hipFree(d_cv); (void)hipFree(d_total);
        SR_HIP(ce);
        out->clusters_culled = total;
        if (dev->d_cull_violations) {   // of the most recent draw on this device
            uint32_t v = 0;
            SR_HIP(hipMemcpy(&v, dev->d_cull_violations, 4, hipMemcpyDeviceToHost));
            out->cull_violations = v;
        }
    }
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
    if (e == hipSuccess) e = hipMalloc(&d_planes, (size_t)n * 16);
    if (e == hipSuccess) e = hipMalloc(&d_clip, (size_t)n * 16);
    if (e == hipSuccess) e = hipMemcpy(d_view, &x, sizeof x, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        ScreenPlanes sp{};
        sp.xy = (int2*)d_planes; sp.zw = (float2*)((char*)d_planes + (size_t)n * 8); sp.n_verts = n;
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
