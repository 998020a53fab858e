// Shared device-side definitions of the gfx950 rasteriser: records that live in HBM, the fixed-point
// and floating-point conventions every kernel follows, and the per-triangle set-up.
//
// Numerical conventions (every kernel that touches visibility obeys them, which is what makes the
// depth / triangle-id result independent of tile size, bin order, view batching and screen split):
//   * mat * vec = last column times its component, then fused multiply-adds down to column 0
//     (shaders/shader.vert:21-26 of the reference; proj*view is formed first, then applied);
//   * perspective divide multiplies by 1/w; viewport transform is multiply then add, unfused
//     (viewport of src/main.cpp:564-571: x_f = x_d*W/2 + W/2, y_f = -y_d*H/2 + H/2, z_f = z_d);
//   * positions snap to 1/256 pixel, round to nearest even; coverage uses 64-bit integer edge
//     functions with the top-left rule; sample points are the pixel centre (1 sample) or the Vulkan
//     standard 4-sample pattern;
//   * depth at a sample is l0*z0 + l1*z1 + l2*z2, l_i = float(E_i) * (1 / float(2*area)),
//     accumulated as fma(l2, z2, fma(l1, z1, l0*z0));
//   * the build uses -ffp-contract=off: only the fmaf() calls written here are fused.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sr {

constexpr int kSubpix = 256;
constexpr int kHalf = 128;
constexpr float kGuardPx = 1048576.0f;  // 2^20 pixels
constexpr uint64_t kVisClear = 0x3F800000FFFFFFFFull;  // depth 1.0, no triangle
constexpr uint32_t kNoTri = 0xFFFFFFFFu;
constexpr uint32_t kColorClear = 0xFF000000u;  // B,G,R,A = 0,0,0,255 (clear colour of src/main.cpp:920)
constexpr int kMaxMipLevels = 16;

struct ViewXform {       // one per view, derived on the host from the reference's uniform block
    float model[16];     // column-major
    float pv[16];        // proj * view, column-major
    float cam[3];
    float model_is_identity;   // 1 when model is exactly the identity matrix (every configuration of the reference, src/main.cpp:991), else 0
};

struct ScreenVertex {    // 16 bytes; rw == 0 marks a vertex that cannot be projected
    int32_t X, Y;        // window coordinates, 1/256 pixel
    float z;             // depth after the perspective divide
    float rw;            // 1 / w_clip
};

// Window-space vertices of a pass are stored as two planes, so that a stage only pulls what it
// needs through the caches: triangle set-up reads xy only, and z only for the triangles that cover a sample.
// (1/w is not stored: the shading pass recomputes the vertices it needs from the positions.)
constexpr int32_t kBadVertexX = INT32_MIN;   // xy.x of a vertex that cannot be projected
// Vertex planes are padded to whole groups of four vertices, so that the vertex stage can move four vertices per work-item
// with 16-byte loads and stores whatever the vertex count and the view.
__host__ __device__ inline uint32_t plane_stride(uint32_t n_verts) { return (n_verts + 3u) & ~3u; }
struct ScreenPlanes {
    int2* xy;      // [views][stride]: window x, y in 1/256 pixel
    float* z;      // [views][stride]: depth; null in a pass that stores none (launch_setup with a DepthSource)
    uint32_t stride;   // vertices per view in both planes: plane_stride(number of vertices)
};

struct TexDesc {
    const uint32_t* texels;  // RGBA8, all levels back to back
    int w, h, levels;
    int log2_w, log2_h;      // both sizes powers of two: their logarithms (every level then is one too: wrap with a mask, rows by a shift); else -1
    float min_lod, max_lod, bias;
    float max_aniso;         // the sampler's maxAnisotropy (src/texture.cpp:255-256); <= 1: isotropic footprint
    uint32_t level_off[kMaxMipLevels];  // in texels
};

struct FrameGeom {
    int W, H;                // frame size (viewport)
    int rx, ry, rw, rh;      // stored rectangle
    int tiles_x, tiles_y;    // tiles over the stored rectangle
    int tile_log2;           // tile edge in pixels = 1 << tile_log2
    int samples;             // 1 or 4
};

// Two flags per kFlagBlockW x kFlagBlockH pixels of every view of a target.  `flags`: the block's part of the visibility buffer
// may hold something other than the clear value -- every kernel that writes a visibility word sets it; the shading pass of the
// draw neither reads nor shades a block without it (77 % of the benchmark's frame), and the start of the next draw clears only
// the flagged blocks and takes the flags down again.  `color_dirty`: the block's colour is not known to be the clear colour, so
// a shading pass that finds the block empty has to store the clear colour once.
constexpr uint32_t kFlagBlockW = 32, kFlagBlockH = 8;
struct BlockFlags {
    uint32_t* flags;         // [views of the target][nby][nbx]: covered -- set by every kernel that writes a visibility word of the block
                             // (always before the shading pass of the same draw), reset by the clear at the start of the next draw
    uint32_t* color_dirty;   // same shape: the block's colour is not the clear colour (set and reset by the shading pass)
    int nbx, nby;
    __device__ uint32_t& at(uint32_t view, uint32_t bx, uint32_t by) const { return flags[((size_t)view * nby + by) * nbx + bx]; }
    __device__ uint32_t& of(uint32_t view, int x /* pixel inside the stored rectangle */, int y) const { return at(view, (uint32_t)x / kFlagBlockW, (uint32_t)y / kFlagBlockH); }
    __device__ uint32_t& dirty_at(uint32_t view, uint32_t bx, uint32_t by) const { return color_dirty[((size_t)view * nby + by) * nbx + bx]; }
};

// ---- vertex shader arithmetic -------------------------------------------------------------------
__host__ __device__ inline void mat4_mul_vec4(const float* M, float x, float y, float z, float w, float* r)
{
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float acc = M[12 + i] * w;
        acc = fmaf(M[8 + i], z, acc);
        acc = fmaf(M[4 + i], y, acc);
        acc = fmaf(M[i], x, acc);
        r[i] = acc;
    }
}

__device__ inline void mat3_mul_vec3(const float* M, float x, float y, float z, float* r)
{
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        float acc = M[8 + i] * z;
        acc = fmaf(M[4 + i], y, acc);
        acc = fmaf(M[i], x, acc);
        r[i] = acc;
    }
}

// gl_Position and the window-space record of one vertex
__device__ inline ScreenVertex project_vertex(const ViewXform& vx, float px, float py, float pz, int W, int H, float* clip)
{
    float wp[4];
    mat4_mul_vec4(vx.model, px, py, pz, 1.0f, wp);
    mat4_mul_vec4(vx.pv, wp[0], wp[1], wp[2], 1.0f, clip);
    ScreenVertex sv;
    sv.X = 0; sv.Y = 0; sv.z = 0.0f; sv.rw = 0.0f;
    const float w = clip[3];
    if (!(w > 0.0f)) return sv;
    const float rw = 1.0f / w;
    const float xd = clip[0] * rw, yd = clip[1] * rw, zd = clip[2] * rw;
    const float hw = (float)W * 0.5f, hh = (float)H * 0.5f;
    const float xf = xd * hw + hw;
    const float yf = yd * (-hh) + hh;
    if (!(fabsf(xf) <= kGuardPx) || !(fabsf(yf) <= kGuardPx)) return sv;
    if (!(rw > 0.0f) || isinf(rw)) return sv;
    sv.X = (int32_t)rintf(xf * (float)kSubpix);
    sv.Y = (int32_t)rintf(yf * (float)kSubpix);
    sv.z = zd;
    sv.rw = rw;
    return sv;
}

// the depth project_vertex stores for a vertex it can project, by the same operations in the same order (a pass whose vertex stage
// stores no depth plane: set-up works it out again for the few triangles that cover a sample)
__device__ inline float project_depth(const ViewXform& vx, float px, float py, float pz)
{
    float wp[4], clip[4];
    mat4_mul_vec4(vx.model, px, py, pz, 1.0f, wp);
    mat4_mul_vec4(vx.pv, wp[0], wp[1], wp[2], 1.0f, clip);
    const float rw = 1.0f / clip[3];
    return clip[2] * rw;
}
struct DepthSource {         // what project_depth needs: the mesh's position planes and the transforms of the pass
    const float *px, *py, *pz;
    const ViewXform* views;
};

// ---- triangle set-up -----------------------------------------------------------------------------
// a = v0, b = v2, c = v1: a front-facing triangle (counter-clockwise on the displayed image) is
// traversed clockwise in y-down window coordinates, which makes all edge functions >= 0 inside.
struct TriSetup {
    int32_t ax, ay, bx, by, cx, cy;
    int32_t bias_a, bias_b, bias_c;   // 0 on top/left edges, -1 otherwise (edge opposite a, b, c)
    float z0, z1, z2;                 // depth of v0, v1, v2
    float rA;                         // 1 / float(2*area)
    int64_t A;                        // 2*area > 0
    int32_t minX, maxX, minY, maxY;
};

__device__ inline bool is_top_left(int32_t dx, int32_t dy) { return dy < 0 || (dy == 0 && dx > 0); }

// 0: culled (back face / degenerate), 1: ok, 2: dropped because a vertex cannot be projected while
// another can (the triangle would need clipping against w > 0 or the guard band)
__device__ inline int setup_triangle(const ScreenVertex& v0, const ScreenVertex& v1, const ScreenVertex& v2, TriSetup& t)
{
    const int bad = (v0.rw == 0.0f) + (v1.rw == 0.0f) + (v2.rw == 0.0f);
    if (bad) return bad < 3 ? 2 : 0;
    const int64_t area2 = (int64_t)(v1.X - v0.X) * (int64_t)(v2.Y - v0.Y) - (int64_t)(v2.X - v0.X) * (int64_t)(v1.Y - v0.Y);
    if (area2 >= 0) return 0;
    t.A = -area2;
    t.ax = v0.X; t.ay = v0.Y; t.bx = v2.X; t.by = v2.Y; t.cx = v1.X; t.cy = v1.Y;
    t.bias_a = is_top_left(t.cx - t.bx, t.cy - t.by) ? 0 : -1;
    t.bias_b = is_top_left(t.ax - t.cx, t.ay - t.cy) ? 0 : -1;
    t.bias_c = is_top_left(t.bx - t.ax, t.by - t.ay) ? 0 : -1;
    t.z0 = v0.z; t.z1 = v1.z; t.z2 = v2.z;
    t.rA = 1.0f / (float)t.A;
    t.minX = min(v0.X, min(v1.X, v2.X)); t.maxX = max(v0.X, max(v1.X, v2.X));
    t.minY = min(v0.Y, min(v1.Y, v2.Y)); t.maxY = max(v0.Y, max(v1.Y, v2.Y));
    return 1;
}

// edge functions (weights of v0 = a, v2 = b, v1 = c) at a 1/256-pixel position
__device__ inline void edge_eval(const TriSetup& t, int64_t px, int64_t py, int64_t& Ea, int64_t& Eb, int64_t& Ec)
{
    Ea = (int64_t)(t.cx - t.bx) * (py - t.by) - (int64_t)(t.cy - t.by) * (px - t.bx);
    Eb = (int64_t)(t.ax - t.cx) * (py - t.cy) - (int64_t)(t.ay - t.cy) * (px - t.cx);
    Ec = (int64_t)(t.bx - t.ax) * (py - t.ay) - (int64_t)(t.by - t.ay) * (px - t.ax);
}

// sample offsets inside a pixel, 1/256 pixel units (Vulkan standard sample locations)
__device__ inline int sample_dx(int S, int s) { return S == 1 ? kHalf : (s == 0 ? 96 : s == 1 ? 224 : s == 2 ? 32 : 160); }
__device__ inline int sample_dy(int S, int s) { return S == 1 ? kHalf : (s == 0 ? 32 : s == 1 ? 96 : s == 2 ? 160 : 224); }

// pixel columns/rows whose samples can lie inside [lo, hi] (1/256 pixel, inclusive)
template <int S>
__device__ inline void pixel_span(int32_t lo, int32_t hi, int& p0, int& p1)
{
    if (S == 1) {           // sample at +128: lo <= 256p+128 <= hi
        p0 = (lo - kHalf + kSubpix - 1) >> 8;
        p1 = (hi - kHalf) >> 8;
    } else {                // conservative: any pixel the box touches
        p0 = lo >> 8;
        p1 = hi >> 8;
    }
}

// depth key of a covered sample, or false if the fragment falls outside the clip volume in z -- or lies exactly on the
// far plane: the depth test is LESS against a buffer cleared to 1.0 (src/main.cpp:602-610, :921), so z = 1.0 never passes
__device__ inline bool depth_key(const TriSetup& t, int64_t Ea, int64_t Eb, int64_t Ec, uint32_t tri, uint64_t& key)
{
    const float l0 = (float)Ea * t.rA, l2 = (float)Eb * t.rA, l1 = (float)Ec * t.rA;
    float z = fmaf(l2, t.z2, fmaf(l1, t.z1, l0 * t.z0));
    z = z + 0.0f;
    if (!(z >= 0.0f && z < 1.0f)) return false;
    key = ((uint64_t)__float_as_uint(z) << 32) | (uint64_t)tri;
    return true;
}

// Culling of whole chunks of the mesh against the stored rectangle (draws of one view into a target restricted to a
// rectangle: the bands of a frame split over several GPUs).  A chunk is the kChunkCullTris consecutive triangles one set-up
// workgroup handles; at mesh creation every chunk gets a bounding sphere of the positions it references and the range of vertex
// numbers it uses.  Per draw, one small kernel projects the spheres conservatively: a chunk whose every vertex provably lands
// outside the rectangle (by more than the slack below) cannot touch a stored pixel, its set-up workgroup leaves at once, and the
// groups of kCullVertexGroup vertices no surviving chunk refers to are not transformed.  Nothing else changes: what is drawn is
// computed exactly as without culling, so a band stays bit-identical to the same rows of the whole frame.
constexpr uint32_t kCullVertexGroup = 256;        // vertices per flag of the vertex stage: one wave of the fused vertex kernel
constexpr uint32_t kCullMaxGroupsPerChunk = 64;   // a chunk whose vertex numbers span more groups marks "every vertex" instead
constexpr float kCullSlackPixels = 2.0f;          // on every side of the projected bound
struct CullView {            // rows x, y, w of (proj * view * model), their 3-vectors' lengths, made on the host in double precision
    float ax[4], ay[4], aw[4];
    float nx, ny, nw;
    float scale;             // |entries| of the rows summed: the rounding allowance of the device's own float evaluation is a multiple of it
};
struct ChunkCull {
    const float4* sphere;    // [chunks]: centre, radius (infinite when a position is not finite)
    const uint4* vrange;     // [chunks]: lowest and highest vertex number the chunk's triangles use; z, w: bit k set = the chunk uses a vertex of
                             // group (lowest / kCullVertexGroup + k), k < 64
    uint32_t n_chunks;
    uint32_t* visible;       // [chunks], written per draw: 0 = the chunk cannot touch the stored rectangle
    uint32_t* stamp;         // [1 + vertex groups]: == epoch when the group (word 0: every group) must be transformed in this draw
    uint32_t epoch;
    unsigned long long* counts;   // [2], cumulative: chunks tested, chunks culled
};
// what the vertex and set-up kernels get (null pointers: no culling)
struct CullUse {
    const uint32_t* visible;
    const uint32_t* stamp;
    uint32_t epoch;
};

// Chunk vertex lists ------------------------------------------------------------------------------
// Made once per mesh: for every chunk of setup_block_triangles() consecutive triangles the distinct vertices it uses (ascending vertex
// numbers) and, per triangle, where its three vertices stand in that list.  A set-up workgroup that goes through many views of a pass
// then loads its chunk's window-space vertices of a view ONCE, side by side, into LDS, and its triangles take their vertices from
// there: a strip of a grid mesh uses about one vertex per triangle, so a third of the bytes cross the L1 that three gathers per
// triangle take -- and the L1 is what bounds that kernel.  A mesh with a chunk of more than kChunkVertsMax distinct vertices, or whose
// vertex numbers inside a chunk span more than kChunkVertsSpan, has no lists and keeps the gathers.
constexpr uint32_t kChunkVertsMax = 1024;         // distinct vertices of a chunk (three 10-bit list positions per triangle)
constexpr uint32_t kChunkVertsSpan = 1u << 16;    // highest minus lowest vertex number of a chunk, plus one: a bitmap in LDS when the lists are made
struct ChunkVerts {
    const uint32_t* list;    // [chunks][kChunkVertsMax]: vertex numbers, ascending; `count[chunk]` of them are valid.  Null: the mesh has no lists
    const uint32_t* count;   // [chunks]
    const uint32_t* local;   // [triangles]: list positions of the three vertices, 10 bits each (vertex 0 lowest)
};

// binning records ---------------------------------------------------------------------------------
struct BinRecord {            // 12 bytes, written by set-up, read by fill
    uint32_t tri;             // triangle id
    uint32_t where;           // small triangle: tile index; large triangle: tx0 | ty0<<8 | tx1<<16 | ty1<<24
    uint32_t rel;             // small triangle: position inside the tile's list; large triangle: kLargeTri
};
constexpr uint32_t kLargeTri = 0xFFFFFFFFu;
constexpr uint32_t kMaxSmallTiles = 4;   // a "small" triangle overlaps at most 2x2 tiles and gets one record per tile

// counters of one pass (all its views summed), zeroed by the begin kernel
enum : uint32_t {
    kCtrRefs = 0,          // total tile references (written by the scan kernel)
    kCtrRefsDropped = 1,   // references that found no room in the list
    kCtrUnprojectable = 2, // triangles that would need clipping and could not be clipped (clip list full, degenerate projection)
    kCtrClipListed = 7,    // triangles put on the clip list by set-up (may exceed its capacity)
    kCtrClipped = 8,       // triangles the near-plane clipper cut and drew
    kCtrDirectTooBig = 9,  // pending triangles the direct rasteriser left alone because their boxes are too large for it (the draw is then repeated with the tile path)
    kNumPassCounters = 16,
};
constexpr uint32_t kPendingTotalStride = 32;   // words between the pending-list lengths of two views
constexpr uint32_t kMicroSpread = 64, kMicroSpreadStride = 64;   // counters, words between them (256 bytes)

struct BinBuffers {
    // per (view, tile) counters, zeroed at the start of a pass: [3][views*tiles]
    uint32_t* count_small;    // references of small triangles (positions handed out by the binning kernel)
    uint32_t* count_large;    // references of large triangles (appended after the small ones by fill)
    uint32_t* cursor_large;   // fill-pass cursor of the large part
    uint32_t* tile_offset;    // [views*tiles + 1]: exclusive scan of count_small + count_large
    uint32_t* refs;           // [capacity]: triangle ids grouped by (view, tile)
    uint32_t ref_capacity;
    uint32_t* pending;        // [views][pending_stride]: ids of the triangles set-up hands to the binning kernel, one list per view;
                              // a wave of a set-up block reserves its place with one atomic add on the view's pending_total
    uint32_t* pending_total;  // [views * kPendingTotalStride], zeroed at the start of a pass: length of the view's list, one counter per
                              // 128 bytes (with the counters of 64 views side by side in four cache lines, the waves' atomic adds
                              // queued up behind one another: set-up of a 64-view batch 21 -> 31 us)
    uint32_t pending_stride;  // >= triangles of the mesh
    BinRecord* records;       // [views][chunks][chunk triangles * kMaxSmallTiles]: the binning kernel takes the pending list in chunks,
                              // each chunk owns one segment
    uint32_t* block_count;    // [views][chunks]: records written for each chunk
    uint32_t n_blocks;        // chunks per view at most
    uint32_t* micro_spread;   // [kMicroSpread * kMicroSpreadStride], zeroed at the start of a pass: micro-triangles the set-up blocks resolved
                              // themselves (instrumentation).  Block b adds to counter b % kMicroSpread; the counters sit a cache line
                              // apart, so that ten thousand blocks do not queue up on one address of the memory-side atomic units.
    uint32_t* counters;       // [kNumPassCounters]
};

// What the kernel that closes the geometry part of a pass tells the host (pinned memory, 8 words: tile references, -,
// unprojectable, binned, micro, clip-listed, clipped, -): the scan kernel does, or, in a draw issued without it, the first
// workgroup of the shading pass.
struct PassReport {
    const uint32_t* counters;        // [kNumPassCounters]
    const uint32_t* pending_total;   // [n_views * kPendingTotalStride]
    uint32_t n_views;
    const uint32_t* micro_spread;    // [kMicroSpread * kMicroSpreadStride]
    uint32_t* host_counters;         // null: nothing to report
};

// Near-plane clipping.  A triangle with one or two vertices that cannot be projected (w <= 0, or outside the guard
// band) cannot be rasterised from window coordinates.  The fixed-function clipper of the reference's pipeline keeps the
// part of such a triangle inside 0 <= z_c <= w_c (src/main.cpp:586 leaves depth clipping on), so the set-up kernel puts
// these triangles on a list and near_clip_kernel clips each of them in clip space against the near plane and four side
// planes inside the guard band (near_clip_polygon below) and resolves the convex polygon that is left, as a fan, into
// the visibility buffer under the id of the triangle it came from -- with the edge functions, depth expression and
// 64-bit min of every other path.
// The shading pass recognises such a triangle by its unprojectable vertex and interpolates from clip-space
// coordinates (shade_pixel).
struct ClipList {
    uint2* entries;            // (view inside the pass, triangle)
    uint32_t capacity;         // entries beyond it are dropped and counted as unprojectable
};

// window-space record of a clip-space position produced by the clipper: the divide, viewport transform, guard band
// and snapping of project_vertex
__device__ inline ScreenVertex project_clip_coords(const float* clip, int W, int H)
{
    ScreenVertex sv;
    sv.X = 0; sv.Y = 0; sv.z = 0.0f; sv.rw = 0.0f;
    const float w = clip[3];
    if (!(w > 0.0f)) return sv;
    const float rw = 1.0f / w;
    const float xd = clip[0] * rw, yd = clip[1] * rw, zd = clip[2] * rw;
    const float hw = (float)W * 0.5f, hh = (float)H * 0.5f;
    const float xf = xd * hw + hw;
    const float yf = yd * (-hh) + hh;
    if (!(fabsf(xf) <= kGuardPx) || !(fabsf(yf) <= kGuardPx)) return sv;
    if (!(rw > 0.0f) || isinf(rw)) return sv;
    sv.X = (int32_t)rintf(xf * (float)kSubpix);
    sv.Y = (int32_t)rintf(yf * (float)kSubpix);
    sv.z = zd;
    sv.rw = rw;
    return sv;
}

// The same record without branches, for callers inside long straight-line code (the shading pass): everything is computed
// whatever w is and the conditions select at the end.  Every operation on the path of a projectable vertex is the one
// project_clip_coords performs, in the same order, so the record is the same bit for bit; a vertex that fails a test gets
// the same zeros.
__device__ inline ScreenVertex project_clip_coords_select(const float* clip, int W, int H)
{
    const float w = clip[3];
    const float rw = 1.0f / w;
    const float xd = clip[0] * rw, yd = clip[1] * rw;
    const float hw = (float)W * 0.5f, hh = (float)H * 0.5f;
    const float xf = xd * hw + hw;
    const float yf = yd * (-hh) + hh;
    const bool ok = (w > 0.0f) & (fabsf(xf) <= kGuardPx) & (fabsf(yf) <= kGuardPx) & (rw > 0.0f) & (rw <= 3.4028234e38f);   // (false for NaNs; the last: rw is not infinite)
    ScreenVertex sv;
    sv.X = ok ? (int32_t)rintf(xf * (float)kSubpix) : 0;
    sv.Y = ok ? (int32_t)rintf(yf * (float)kSubpix) : 0;
    sv.z = 0.0f;                 // (not needed by the callers of this form)
    sv.rw = ok ? rw : 0.0f;
    return sv;
}

// The clipper proper.  The triangle is clipped in clip space against the near plane (z >= 0) and then against four
// side planes |x| <= K_x w, |y| <= K_y w that keep every cut point 2^19 pixels or less from the frame (inside the
// guard band).  Two rules make the result fit the triangles around it:
//   * a projectable vertex of the triangle counts as inside every plane, whatever its coordinates: the edges between
//     two such vertices are never cut and stay bit-identical to the neighbouring triangles' edges (on the near plane
//     the depth range test removes, per sample, what lies beyond it; the side planes are outside the guard band such
//     a vertex is inside of);
//   * a cut point is always computed from the inside end g of its edge towards the outside end o,
//         t = d_g / (d_g - d_o),   (x, y, z, w) = fma(t, o - g, g)     (z = 0 exactly on the near plane),
//     with d the signed distance to the plane, so both triangles that share the edge get the same point; when the
//     inside end is inside only by the first rule (d_g <= 0) no point is produced -- the polygon goes on from g.
// out: up to kClipMaxVertices window-space vertices of a convex polygon, to be drawn as a fan.  Returns their number
// (0: nothing to draw), or -1 when a vertex that is left cannot be projected (a degenerate projection): dropped and counted.
constexpr int kClipMaxVertices = 8;     // 3 + one per plane
struct ClipVertex { float c[4]; ScreenVertex sv; bool original; };   // original: a projectable vertex of the triangle, sv valid

__device__ inline float clip_plane_distance(int plane, const float* c, float kx, float ky)
{
    switch (plane) {
    case 0: return c[2];
    case 1: return kx * c[3] - c[0];
    case 2: return kx * c[3] + c[0];
    case 3: return ky * c[3] - c[1];
    default: return ky * c[3] + c[1];
    }
}

__device__ inline int near_clip_polygon(const float (*c)[4] /* clip coordinates of the 3 vertices */, const ScreenVertex* sv /* [3] */, int W, int H, ScreenVertex* out)
{
    ClipVertex A[kClipMaxVertices], B[kClipMaxVertices];
    ClipVertex* src = A;
    ClipVertex* dst = B;
    int n = 3;
    for (int i = 0; i < 3; ++i) {
        for (int k = 0; k < 4; ++k) src[i].c[k] = c[i][k];
        src[i].sv = sv[i];
        src[i].original = sv[i].rw != 0.0f;
    }
    const float kx = 524288.0f / ((float)W * 0.5f), ky = 524288.0f / ((float)H * 0.5f);
    for (int plane = 0; plane < 5 && n >= 3; ++plane) {
        int m = 0;
        for (int i = 0; i < n; ++i) {
            const ClipVertex& a = src[i];
            const ClipVertex& b = src[i + 1 == n ? 0 : i + 1];
            const float da = clip_plane_distance(plane, a.c, kx, ky), db = clip_plane_distance(plane, b.c, kx, ky);
            const bool in_a = a.original || da >= 0.0f, in_b = b.original || db >= 0.0f;
            if (in_a) dst[m++] = a;
            if (in_a == in_b) continue;
            const ClipVertex& g = in_a ? a : b;
            const ClipVertex& o = in_a ? b : a;
            const float dg = in_a ? da : db, dout = in_a ? db : da;
            if (!(dg > 0.0f) || !(dout < 0.0f)) continue;
            const float t = dg / (dg - dout);
            ClipVertex& cut = dst[m++];
            for (int k = 0; k < 4; ++k) cut.c[k] = fmaf(t, o.c[k] - g.c[k], g.c[k]);
            if (plane == 0) cut.c[2] = 0.0f;
            cut.original = false;
        }
        ClipVertex* tmp = src; src = dst; dst = tmp;
        n = m;
    }
    if (n < 3) return 0;
    for (int i = 0; i < n; ++i) {
        out[i] = src[i].original ? src[i].sv : project_clip_coords(src[i].c, W, H);
        if (out[i].rw == 0.0f) return -1;
    }
    return n;
}

}  // namespace sr
