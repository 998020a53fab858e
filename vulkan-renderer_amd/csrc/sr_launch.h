// Host-callable launchers of the kernels in sr_kernels.hip.
#pragma once

#include "sr_device.h"

namespace sr {

struct FragLaunch {
    const uint32_t* idx;
    const void* verts_aos;     // the reference's 32-byte vertex records
    ScreenPlanes screen;       // window-space vertices of this pass
    const ViewXform* views;
    const uint64_t* vis;
    uint32_t* color;
    FrameGeom g;
    TexDesc albedo, metallic, roughness;
    const float* decode;
    const float* thresh;
    uint32_t view0, n_views;
    DirtyMap dirty;            // flags == null: every block is shaded from the visibility buffer
    bool exact_vertices;       // recompute window-space vertices with the vertex stage's exact arithmetic
    float max_anisotropy;      // > 1 (together with dirty + exact_vertices): the sampler's anisotropic footprint is applied
};

void launch_split_positions(hipStream_t st, const void* aos, uint32_t n, float* px, float* py, float* pz);
void launch_vertex(hipStream_t st, const float* px, const float* py, const float* pz, uint32_t n, const ViewXform* views, uint32_t n_views,
                   int W, int H, const ScreenPlanes& out, ScreenVertex* debug_out, float* clip_out);
// vis: n words to clear; zero16/zero_bytes: 16-byte aligned block to zero; views_*: optional 16-byte aligned copy (0 bytes = none)
void launch_begin_pass(hipStream_t st, uint64_t* vis, size_t n, void* zero16, size_t zero_bytes, const void* views_src, void* views_dst, size_t views_bytes);
void launch_pack_positions(hipStream_t st, const void* aos, uint32_t n, void* pos4);
// pos4 != null selects the set-up kernel that transforms vertices itself (packed positions, transforms of the pass's views);
// the caller then skips launch_vertex
void launch_setup(hipStream_t st, const uint32_t* idx, uint32_t n_tris, const ScreenPlanes& sp, const FrameGeom& g,
                  const BinBuffers& b, uint32_t n_views, uint64_t* vis, uint32_t view0, hipEvent_t between /* recorded between set-up and binning, may be null */,
                  const void* pos4 = nullptr, const ViewXform* views = nullptr, const DirtyMap& dirty = DirtyMap{nullptr, 0, 0},
                  bool bin = true /* false: set-up only, the caller resolves the pending triangles with launch_sparse_tail */,
                  const ClusterCull& cull = ClusterCull{nullptr, nullptr, nullptr},
                  const ClipList& clip = ClipList{nullptr, nullptr, 0, nullptr} /* with pos4: where triangles that need near-plane clipping are listed */);
// near-plane clipping of the triangles that need it (clip.count must be b.counters + 7, zeroed with the pass; clip.state
// zeroed by the caller): cuts them, resolves the pieces into the visibility buffer, writes the number of triangles
// clipped to host_counters[6].  With pos4 the fused set-up kernel has listed them; without, they are found by a scan of
// the vertex kernel's planes (sp) and positions are read from px, py, pz.  Runs after launch_setup and before
// launch_scan / launch_sparse_tail.
void launch_near_clip(hipStream_t st, const uint32_t* idx, uint32_t n_tris, const void* pos4, const float* px, const float* py, const float* pz,
                      const ScreenPlanes& sp, uint32_t n_views, const ViewXform* views, const FrameGeom& g,
                      const ClipList& clip, const BinBuffers& b, uint64_t* vis, uint32_t view0, const DirtyMap& dirty, uint32_t* host_counters);
void launch_count_culled_clusters(hipStream_t st, const ClusterCull& cull, const FrameGeom& g, uint32_t n_clusters, uint32_t n_views, uint64_t* total);
// resolves every pending triangle of the pass straight into the visibility buffer (in place of bin + scan + fill + raster)
// and writes the pass's counters to host_counters: [2] unprojectable, [3] pending triangles, [4] micro-triangles, [5] 8x8 steps
void launch_sparse_tail(hipStream_t st, const uint32_t* idx, uint32_t n_tris, const ScreenPlanes& sp, const FrameGeom& g, const BinBuffers& b,
                        uint64_t* vis, uint32_t view0, uint32_t n_views, const DirtyMap& dirty, uint32_t* host_counters);
// start of a pass when the target keeps a dirty map: vis is the target's whole visibility buffer
void launch_begin_pass_dirty(hipStream_t st, uint64_t* vis, const FrameGeom& g, uint32_t view0, uint32_t n_views, const DirtyMap& dirty,
                             void* zero16, size_t zero_bytes, const void* views_src, void* views_dst, size_t views_bytes);
void launch_scan(hipStream_t st, const BinBuffers& b, uint32_t n_entries, uint32_t n_views, uint32_t* host_counters);
void launch_fill(hipStream_t st, const FrameGeom& g, const BinBuffers& b, uint32_t n_tris, uint32_t n_views);
void launch_raster(hipStream_t st, const uint32_t* idx, uint32_t n_tris, const ScreenPlanes& sp, const FrameGeom& g, const BinBuffers& b,
                   uint64_t* vis, uint32_t view0, uint32_t n_views, const DirtyMap& dirty = DirtyMap{nullptr, 0, 0},
                   bool half_workgroup = false /* 4-sample 32x32 tiles with 512 work-items instead of 1024 (no register spills) */);
bool raster_config_supported(int samples, int tile_log2);
uint32_t setup_block_triangles();   // triangles handled by one set-up block
uint32_t setup_block_records();     // size of the record segment one set-up block owns
void launch_fragment(hipStream_t st, const FragLaunch& f);
void launch_mip_blit(hipStream_t st, const uint32_t* src, int sw, int sh, uint32_t* dst, int dw, int dh, const float* decode, const float* thresh);
void launch_split_vis(hipStream_t st, const uint64_t* vis, size_t n, float* depth, uint32_t* tri);
void launch_stats(hipStream_t st, const uint64_t* vis, size_t n_pixels, int S, uint32_t* tri_flag, uint32_t n_tris, uint64_t* shaded, uint64_t* visible);

}  // namespace sr
