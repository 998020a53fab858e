// Host-callable launchers of the kernels in sr_kernels.hip.
#pragma once

#include "sr_device.h"

namespace sr {

struct FragLaunch {
    const uint32_t* idx;
    const void* verts_aos;     // the reference's 32-byte vertex records
    const ViewXform* views;
    const uint64_t* vis;
    uint32_t* color;
    FrameGeom g;
    TexDesc albedo, metallic, roughness;
    TexDesc normal;            // used when normal_mapping is set
    bool normal_mapping;       // shade with the normal map's normals (the reference's getNormalFromMap with its body restored)
    const float* decode;
    const float* thresh;
    uint32_t view0, n_views;
    BlockFlags flags;          // of the target
    PassReport report;         // host_counters != null: the shading pass reports the counters of the pass (lean draws)
    bool identity_model;       // the model matrix of every one of these views is the identity
    const uint32_t* packed_mr; // launch_pack_red_channels of the metallic and the roughness chain (used when their sampling is the same too), or null
};

constexpr uint32_t kMaxPackedViews = 8;   // views whose transforms fit the kernel arguments of launch_store_views
// writes n (<= kMaxPackedViews) per-view transforms from host memory to dst without any copy or staging buffer
void launch_store_views(hipStream_t st, const ViewXform* host_views, uint32_t n, ViewXform* dst);
void launch_split_positions(hipStream_t st, const void* aos, uint32_t n, float* px, float* py, float* pz);
// fill_blocks (here and in launch_setup): a workgroup handles its vertices / triangles for EVERY view of the launch, loading positions /
// indices once -- unless that leaves fewer than about fill_blocks workgroups, in which case the views are spread over more of them
constexpr uint32_t kFillBlocks = 4096;
// set-up workgroups (one view each, one sample) from which a launch resolves its micro-triangles' samples through LDS, whole waves at a
// time, instead of inside each lane's walk: two per workgroup slot of the chip -- a launch long enough to be bound by what it does
constexpr uint32_t kStagedResolveMinBlocks = 4096;
void launch_vertex(hipStream_t st, const float* px, const float* py, const float* pz, uint32_t n, const ViewXform* views, uint32_t n_views,
                   int W, int H, const ScreenPlanes& out, ScreenVertex* debug_out, float* clip_out, uint32_t fill_blocks = kFillBlocks);
// vis: the target's whole visibility buffer, of which the flagged blocks of views [view0, view0 + n_views) are cleared;
// zero16/zero_bytes: 16-byte aligned block to zero; views_*: optional 16-byte aligned copy (0 bytes = none)
void launch_begin_pass(hipStream_t st, uint64_t* vis, const BlockFlags& flags, const FrameGeom& g, uint32_t view0, uint32_t n_views,
                       void* zero16, size_t zero_bytes, const void* views_src, void* views_dst, size_t views_bytes);
// the start of a pass and its vertex stage in one launch, the transforms of the pass's views (host memory) travelling as
// kernel arguments and being left in views_dst for the later kernels.  Returns false, having launched nothing, when
// that does not apply (no vertices, or more than kMaxPackedViews views): the caller then uses launch_begin_pass + launch_vertex.
bool launch_vertex_and_begin(hipStream_t st, const float* px, const float* py, const float* pz, uint32_t n, const ViewXform* host_views, uint32_t n_views,
                             ViewXform* views_dst, int W, int H, const ScreenPlanes& out, uint64_t* vis, const BlockFlags& flags, const FrameGeom& g, uint32_t view0,
                             void* zero16, size_t zero_bytes, const CullUse& cull = CullUse{nullptr, nullptr, 0u});
// triangle set-up (micro-triangles resolved, clip list and pending lists filled) followed by the binning of the pending lists
void launch_setup(hipStream_t st, const uint32_t* idx, uint32_t n_tris, const ScreenPlanes& sp, const FrameGeom& g,
                  const BinBuffers& b, const ClipList& clip, uint32_t n_views, uint64_t* vis, const BlockFlags& flags /* of the target */, uint32_t view0,
                  hipEvent_t between /* recorded between set-up and binning, may be null */, bool bin = true /* false: set-up only */,
                  uint32_t fill_blocks = kFillBlocks, const uint32_t* chunk_visible = nullptr /* chunk culling: one-view draws only */,
                  uint32_t staged_min_blocks = kStagedResolveMinBlocks,
                  const DepthSource* depth_source = nullptr /* set: sp.z is null, set-up works depths out itself; set-up only, no binning */,
                  const ChunkVerts* chunk_verts = nullptr /* set: the mesh's chunk vertex lists, for workgroups that take several views */);
// chunk culling (sr_device.h): bounds of every chunk of setup_block_triangles() triangles, once per mesh; the per-draw test
void launch_chunk_bounds(hipStream_t st, const uint32_t* idx, uint32_t n_tris, const void* verts_aos, float4* sphere, uint4* vrange);
void launch_chunk_cull(hipStream_t st, const ChunkCull& cc, const CullView& v, const FrameGeom& g);
// chunk vertex lists (sr_device.h), once per mesh, after launch_chunk_bounds: list [chunks][kChunkVertsMax], count [chunks], local [triangles];
// *unfit (zeroed by the caller) counts the chunks that do not fit
void launch_chunk_verts(hipStream_t st, const uint32_t* idx, uint32_t n_tris, const uint4* vrange, uint32_t* list, uint32_t* count, uint32_t* local, uint32_t* unfit);
// near-plane clipping of the triangles set-up listed: cuts them and resolves the pieces into the visibility buffer.
// Runs after launch_setup and before launch_scan (which reports the counters to the host).
void launch_near_clip(hipStream_t st, const uint32_t* idx, uint32_t n_tris, const float* px, const float* py, const float* pz,
                      const ViewXform* views, const FrameGeom& g, const ClipList& clip, const BinBuffers& b, uint64_t* vis, const BlockFlags& flags,
                      uint32_t view0);
// the pending triangles of a draw issued without the tile path, resolved straight into the visibility buffer (a few thousand of them)
void launch_direct_raster(hipStream_t st, const uint32_t* idx, uint32_t n_tris, const ScreenPlanes& sp, const FrameGeom& g, const BinBuffers& b,
                          uint64_t* vis, const BlockFlags& flags, uint32_t view0, uint32_t n_views, uint32_t expected_per_view);
void launch_scan(hipStream_t st, const BinBuffers& b, uint32_t n_entries, uint32_t n_views, uint32_t* host_counters);
void launch_fill(hipStream_t st, const FrameGeom& g, const BinBuffers& b, uint32_t n_tris, uint32_t n_views);
void launch_raster(hipStream_t st, const uint32_t* idx, uint32_t n_tris, const ScreenPlanes& sp, const FrameGeom& g, const BinBuffers& b,
                   uint64_t* vis, const BlockFlags& flags, uint32_t view0, uint32_t n_views);
bool raster_config_supported(int samples, int tile_log2);
uint32_t setup_block_triangles();   // triangles handled by one set-up block
uint32_t chunk_triangles();         // triangles of one chunk of the pending list (what one turn of a binning workgroup takes)
uint32_t setup_block_records();     // size of the record segment one chunk owns
void launch_fragment(hipStream_t st, const FragLaunch& f);
// out[i] = red byte of a[i] | red byte of b[i] << 8, for whole mip chains of two maps of the same size
void launch_pack_red_channels(hipStream_t st, const uint32_t* a, const uint32_t* b, size_t n_texels, uint32_t* out);
void launch_mip_blit(hipStream_t st, const uint32_t* src, int sw, int sh, uint32_t* dst, int dw, int dh, const float* decode, const float* thresh);
void launch_split_vis(hipStream_t st, const uint64_t* vis, size_t n, float* depth, uint32_t* tri);
void launch_stats(hipStream_t st, const uint64_t* vis, size_t n_pixels, int S, uint32_t* tri_flag, uint32_t n_tris, uint64_t* shaded, uint64_t* visible);

}  // namespace sr
