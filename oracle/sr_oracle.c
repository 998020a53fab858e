/*
 * sr_oracle.c -- CPU restatement of the per-frame draw path of LesleyLai/Vulkan-Renderer.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity checker for the HIP rasteriser and the
 * "port" CPU baseline of the benchmark.  Nothing in the shipped library includes, links or
 * calls it; only tests/, the smoke check and the benchmark's cpu_baseline leg load it.
 *
 * PARITY UNPINNED: the reference has no test, golden image or read-back for this path
 * (reference tests/main.cpp:1-2 is an empty main), and the arithmetic of the path lives in
 * the Vulkan driver the reference runs on (Mesa lavapipe on its CPU path; not vendored,
 * version unpinned), which is not available where this was written.  The restatement follows
 * the reference's own sources for everything they define and the Vulkan 1.2 specification
 * for what the driver defines; it is pinned only by hand-derivable known answers
 * (tests/golden/).
 *
 * What each part follows (paths relative to the reference root):
 *   UBO layout                      src/main.cpp:50-55, shaders/shader.vert:4-9
 *   camera matrices                 src/main.cpp:988-1004 (GLM lookAt RH / perspective RH, depth 0..1)
 *   vertex format                   src/mesh.hpp:18-54
 *   vertex shader                   shaders/shader.vert:20-27
 *   primitive / viewport / cull     src/main.cpp:559-593
 *   multisampling                   src/main.cpp:595-600, src/vulkan_helper/gpu_device.cpp:16-30
 *   depth test                      src/main.cpp:602-610, clear values src/main.cpp:919-921
 *   colour target, blend off        src/vulkan_helper/swapchain.cpp:17-21, src/main.cpp:612-624
 *   draw (mesh 0, uint32 indices)   src/main.cpp:938-955
 *   texture image, mip count        src/texture.cpp:185-214
 *   mip generation (linear blits)   src/texture.cpp:87-176
 *   sampler                         src/texture.cpp:246-263
 *   fragment shader                 shaders/shader.frag:16-152
 *
 * Conventions chosen where the specification leaves freedom (the HIP path makes the same
 * choices; they are listed once more in the design notes of the repository):
 *   - mat*vec is a multiply by the last column followed by fused multiply-adds down to
 *     column 0; proj*view is formed first (GLSL '*' is left-associative).
 *   - perspective divide multiplies by 1/w; viewport transform is multiply then add.
 *   - 8 sub-pixel bits, round-to-nearest-even snapping, 64-bit integer edge functions,
 *     top-left fill rule, pixel centre at +0.5, Vulkan standard 4x sample positions.
 *   - depth at a sample = l0*z0 + l1*z1 + l2*z2 with l_i = float(E_i) * (1/float(2A)).
 *   - a triangle with one or two vertices at w <= 0 or outside a 2^20-pixel guard band is clipped, as the
 *     reference's fixed-function clipper does (depthClampEnable = VK_FALSE, src/main.cpp:586): cut at the near
 *     plane and at four side planes inside the guard band, the pieces drawn as a fan under the triangle's own
 *     id (near_clip_polygon).  sro_set_near_clip(1) selects this; the checker's callers always do.  With
 *     sro_set_near_clip(0) such a triangle is dropped and counted instead -- the behaviour of a library
 *     whose clip list is full, kept so that case can be checked too.  A triangle with all three vertices
 *     unprojectable draws nothing either way.  Fragments with depth outside [0,1) are dropped (clip volume,
 *     no depth clamp; 1.0 never passes LESS against the cleared 1.0).
 *   - texture LOD from coarse 2x2-quad differences.  The sampler's maxAnisotropy (src/texture.cpp:255-256:
 *     16) is applied by the method the Vulkan specification spells out when sro_set_anisotropic(1) is set:
 *     eta = min(rho_max / rho_min, maxAniso), N = ceil(eta) trilinear taps at i/(N+1) - 1/2 along the major
 *     axis' derivative, lambda = log2(rho_max / eta).  With sro_set_anisotropic(0) the footprint is the
 *     isotropic one (a sampler with maxAnisotropy 1).  The footprint is implementation-defined in the
 *     specification, so what a particular driver does stays unpinned either way.
 *   - sRGB encode to 8 bits is defined by 255 thresholds (decode of the half-way codes).
 *   - multisample resolve decodes the stored 8-bit samples, averages, re-encodes.
 */
#define _GNU_SOURCE
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define SRO_API __attribute__((visibility("default")))

enum { SUBPIX = 256, HALF = 128 };
static const float GUARD_PX = 1048576.0f; /* 2^20 pixels */

/* ------------------------------------------------------------------------------------ */
/* sRGB tables                                                                          */
/* ------------------------------------------------------------------------------------ */
static float g_srgb_decode[256];  /* 8-bit code -> linear */
static float g_srgb_thresh[255];  /* linear value at which the code goes from c to c+1 */
static pthread_once_t g_once = PTHREAD_ONCE_INIT;

static double srgb_decode_d(double x)
{
    return x <= 0.04045 ? x / 12.92 : pow((x + 0.055) / 1.055, 2.4);
}
static void init_tables(void)
{
    for (int c = 0; c < 256; ++c) g_srgb_decode[c] = (float)srgb_decode_d(c / 255.0);
    for (int c = 0; c < 255; ++c) g_srgb_thresh[c] = (float)srgb_decode_d((c + 0.5) / 255.0);
}
/* smallest code c with x < thresh[c]; 255 if none; NaN and negatives give 0 */
static int srgb_encode8(float x)
{
    if (!(x > 0.0f)) return 0;
    int lo = 0, hi = 255; /* answer in [lo,hi] */
    while (lo < hi) {
        int mid = (lo + hi) >> 1;
        if (x < g_srgb_thresh[mid]) hi = mid; else lo = mid + 1;
    }
    return lo;
}
/* linear (non-sRGB) channel to 8 bits: clamp, round half to even of 255*x */
static int unorm_encode8(float x)
{
    if (!(x > 0.0f)) return 0;
    if (x > 1.0f) x = 1.0f;
    return (int)rintf(x * 255.0f);
}

SRO_API int sro_srgb_encode8(float x)
{
    pthread_once(&g_once, init_tables);
    return srgb_encode8(x);
}

SRO_API void sro_srgb_tables(float* decode256, float* thresh255)
{
    pthread_once(&g_once, init_tables);
    memcpy(decode256, g_srgb_decode, sizeof g_srgb_decode);
    memcpy(thresh255, g_srgb_thresh, sizeof g_srgb_thresh);
}

/* ------------------------------------------------------------------------------------ */
/* Camera: GLM lookAt (right-handed) and perspective (right-handed, depth 0..1),        */
/* as called at src/main.cpp:992-997.  Column-major float[16].                          */
/* ------------------------------------------------------------------------------------ */
static void v3_sub(const float* a, const float* b, float* o) { o[0]=a[0]-b[0]; o[1]=a[1]-b[1]; o[2]=a[2]-b[2]; }
static float v3_dot(const float* a, const float* b) { return (a[0]*b[0] + a[1]*b[1]) + a[2]*b[2]; }
static void v3_cross(const float* a, const float* b, float* o)
{
    o[0] = a[1]*b[2] - b[1]*a[2];
    o[1] = a[2]*b[0] - b[2]*a[0];
    o[2] = a[0]*b[1] - b[0]*a[1];
}
static void v3_normalize(const float* a, float* o)
{
    float inv = 1.0f / sqrtf(v3_dot(a, a));
    o[0] = a[0]*inv; o[1] = a[1]*inv; o[2] = a[2]*inv;
}

SRO_API void sro_look_at(const float eye[3], const float center[3], const float up[3], float m[16])
{
    float f[3], s[3], u[3], t[3];
    v3_sub(center, eye, t); v3_normalize(t, f);
    v3_cross(f, up, t);     v3_normalize(t, s);
    v3_cross(s, f, u);
    m[0] = s[0]; m[4] = s[1]; m[8]  = s[2]; m[12] = -v3_dot(s, eye);
    m[1] = u[0]; m[5] = u[1]; m[9]  = u[2]; m[13] = -v3_dot(u, eye);
    m[2] = -f[0]; m[6] = -f[1]; m[10] = -f[2]; m[14] = v3_dot(f, eye);
    m[3] = 0; m[7] = 0; m[11] = 0; m[15] = 1;
}

SRO_API void sro_perspective(float fovy_rad, float aspect, float z_near, float z_far, float m[16])
{
    float t = tanf(fovy_rad / 2.0f);
    memset(m, 0, 16 * sizeof(float));
    m[0]  = 1.0f / (aspect * t);
    m[5]  = 1.0f / t;
    m[10] = z_far / (z_near - z_far);
    m[11] = -1.0f;
    m[14] = -(z_far * z_near) / (z_far - z_near);
}

/* ------------------------------------------------------------------------------------ */
/* Vertex stage (shaders/shader.vert:20-27)                                             */
/* ------------------------------------------------------------------------------------ */
/* r = M * v, M column-major 4x4: last column first, then fused multiply-adds */
static void mat4_mul_vec4(const float* M, const float* v, float* r)
{
    for (int i = 0; i < 4; ++i) {
        float acc = M[12 + i] * v[3];
        acc = fmaf(M[8 + i], v[2], acc);
        acc = fmaf(M[4 + i], v[1], acc);
        acc = fmaf(M[i],     v[0], acc);
        r[i] = acc;
    }
}
static void mat4_mul_mat4(const float* A, const float* B, float* R)
{
    for (int c = 0; c < 4; ++c) mat4_mul_vec4(A, B + 4 * c, R + 4 * c);
}
static void mat3_mul_vec3(const float* M /* 4x4 */, const float* v, float* r)
{
    for (int i = 0; i < 3; ++i) {
        float acc = M[8 + i] * v[2];
        acc = fmaf(M[4 + i], v[1], acc);
        acc = fmaf(M[i],     v[0], acc);
        r[i] = acc;
    }
}

typedef struct { int32_t X, Y; float z, rw; } screen_vertex; /* rw == 0 marks "not projectable" */

typedef struct {
    float model[16], pv[16], cam[3];
} view_xform;

static void view_from_ubo(const float* ubo /* 52 floats: model, view, proj, cam */, view_xform* vx)
{
    memcpy(vx->model, ubo, 64);
    mat4_mul_mat4(ubo + 32, ubo + 16, vx->pv); /* proj * view */
    memcpy(vx->cam, ubo + 48, 12);
}

static void world_pos(const view_xform* vx, const float* p, float* w3)
{
    float v[4] = { p[0], p[1], p[2], 1.0f }, r[4];
    mat4_mul_vec4(vx->model, v, r);
    w3[0] = r[0]; w3[1] = r[1]; w3[2] = r[2];
}

static void project_vertex(const view_xform* vx, const float* p, int W, int H, float* clip, screen_vertex* sv)
{
    float wp[4];
    world_pos(vx, p, wp);
    wp[3] = 1.0f;
    mat4_mul_vec4(vx->pv, wp, clip);
    sv->X = 0; sv->Y = 0; sv->z = 0.0f; sv->rw = 0.0f;
    float w = clip[3];
    if (!(w > 0.0f)) return;
    float rw = 1.0f / w;
    float xd = clip[0] * rw, yd = clip[1] * rw, zd = clip[2] * rw;
    float hw = (float)W * 0.5f, hh = (float)H * 0.5f;
    float xf = xd * hw + hw;
    float yf = yd * (-hh) + hh;      /* viewport y = H, height = -H (src/main.cpp:564-571) */
    if (!(fabsf(xf) <= GUARD_PX) || !(fabsf(yf) <= GUARD_PX)) return;
    if (!(rw > 0.0f) || isinf(rw)) return;
    sv->X = (int32_t)rintf(xf * (float)SUBPIX);
    sv->Y = (int32_t)rintf(yf * (float)SUBPIX);
    sv->z = zd;
    sv->rw = rw;
}

/* Window-space record of a clip-space position that the near-plane clipper produced: the same divide, viewport
 * transform, guard band and snapping as project_vertex. */
static void project_clip(const float* clip, int W, int H, screen_vertex* sv)
{
    sv->X = 0; sv->Y = 0; sv->z = 0.0f; sv->rw = 0.0f;
    float w = clip[3];
    if (!(w > 0.0f)) return;
    float rw = 1.0f / w;
    float xd = clip[0] * rw, yd = clip[1] * rw, zd = clip[2] * rw;
    float hw = (float)W * 0.5f, hh = (float)H * 0.5f;
    float xf = xd * hw + hw;
    float yf = yd * (-hh) + hh;
    if (!(fabsf(xf) <= GUARD_PX) || !(fabsf(yf) <= GUARD_PX)) return;
    if (!(rw > 0.0f) || isinf(rw)) return;
    sv->X = (int32_t)rintf(xf * (float)SUBPIX);
    sv->Y = (int32_t)rintf(yf * (float)SUBPIX);
    sv->z = zd;
    sv->rw = rw;
}

/* verts: n x 8 floats (pos3, normal3, uv2).  clip_out: n x 4 floats (may be NULL).
 * screen_out: n x 16 bytes {int32 X, int32 Y, float z, float 1/w} (may be NULL). */
SRO_API void sro_vertex_stage(const float* ubo52, const float* verts, uint32_t n, int W, int H,
                              float* clip_out, void* screen_out)
{
    view_xform vx; view_from_ubo(ubo52, &vx);
    screen_vertex* so = (screen_vertex*)screen_out;
    for (uint32_t i = 0; i < n; ++i) {
        float clip[4]; screen_vertex sv;
        project_vertex(&vx, verts + 8 * (size_t)i, W, H, clip, &sv);
        if (clip_out) memcpy(clip_out + 4 * (size_t)i, clip, 16);
        if (so) so[i] = sv;
    }
}

/* ------------------------------------------------------------------------------------ */
/* Rasterisation (src/main.cpp:559-610)                                                 */
/* ------------------------------------------------------------------------------------ */
static const int g_samp4[4][2] = { {96, 32}, {224, 96}, {32, 160}, {160, 224} }; /* x16/256 of the standard 4x pattern */
static const int g_samp1[1][2] = { {128, 128} };

typedef struct {
    int64_t A;              /* twice the area, > 0 for front faces */
    int32_t ax, ay, bx, by, cx, cy; /* a = v0, b = v2, c = v1: clockwise on the y-down screen */
    int bias_a, bias_b, bias_c;     /* 0 for top/left edges, -1 otherwise; bias of the edge opposite a,b,c */
    float z0, z1, z2, rw0, rw1, rw2, rA;
    int32_t minX, maxX, minY, maxY;
} tri_setup;

static int is_top_left(int32_t dx, int32_t dy) { return dy < 0 || (dy == 0 && dx > 0); }

/* returns 1 if the triangle survives (front facing, non-degenerate, all vertices projectable) */
static int setup_triangle(const screen_vertex* v0, const screen_vertex* v1, const screen_vertex* v2, tri_setup* t, int* needs_clip)
{
    *needs_clip = 0;
    int bad = (v0->rw == 0.0f) + (v1->rw == 0.0f) + (v2->rw == 0.0f);
    if (bad) { *needs_clip = bad < 3; return 0; }
    int64_t area2 = (int64_t)(v1->X - v0->X) * (int64_t)(v2->Y - v0->Y) - (int64_t)(v2->X - v0->X) * (int64_t)(v1->Y - v0->Y);
    if (area2 >= 0) return 0; /* >0: clockwise on the displayed image = back face; 0: degenerate */
    t->A = -area2;
    t->ax = v0->X; t->ay = v0->Y; t->bx = v2->X; t->by = v2->Y; t->cx = v1->X; t->cy = v1->Y;
    t->bias_a = is_top_left(t->cx - t->bx, t->cy - t->by) ? 0 : -1; /* edge b->c, weight of a */
    t->bias_b = is_top_left(t->ax - t->cx, t->ay - t->cy) ? 0 : -1; /* edge c->a, weight of b */
    t->bias_c = is_top_left(t->bx - t->ax, t->by - t->ay) ? 0 : -1; /* edge a->b, weight of c */
    t->z0 = v0->z; t->z1 = v1->z; t->z2 = v2->z;
    t->rw0 = v0->rw; t->rw1 = v1->rw; t->rw2 = v2->rw;
    t->rA = 1.0f / (float)t->A;
    int32_t mnx = v0->X, mxx = v0->X, mny = v0->Y, mxy = v0->Y;
    if (v1->X < mnx) mnx = v1->X; if (v1->X > mxx) mxx = v1->X;
    if (v2->X < mnx) mnx = v2->X; if (v2->X > mxx) mxx = v2->X;
    if (v1->Y < mny) mny = v1->Y; if (v1->Y > mxy) mxy = v1->Y;
    if (v2->Y < mny) mny = v2->Y; if (v2->Y > mxy) mxy = v2->Y;
    t->minX = mnx; t->maxX = mxx; t->minY = mny; t->maxY = mxy;
    return 1;
}

/* edge functions at fixed-point position (px,py): weights of v0 (a), v2 (b), v1 (c) */
static void edge_eval(const tri_setup* t, int64_t px, int64_t py, int64_t* Ea, int64_t* Eb, int64_t* Ec)
{
    *Ea = (int64_t)(t->cx - t->bx) * (py - t->by) - (int64_t)(t->cy - t->by) * (px - t->bx);
    *Eb = (int64_t)(t->ax - t->cx) * (py - t->cy) - (int64_t)(t->ay - t->cy) * (px - t->cx);
    *Ec = (int64_t)(t->bx - t->ax) * (py - t->ay) - (int64_t)(t->by - t->ay) * (px - t->ax);
}

static int floor_div256(int32_t v) { return v >> 8; } /* arithmetic shift: floor */

/* Near-plane clipping (option, off by default: a triangle with a vertex that cannot be projected is then dropped
 * and counted).  The fixed-function clipper of the reference's pipeline keeps the part of a primitive inside
 * 0 <= z_c <= w_c (src/main.cpp:559-600 leaves depth clipping on).  Here a triangle with one or two vertices that
 * cannot be projected is clipped in clip space against the near plane (z >= 0) and then against four side planes
 * |x| <= K_x w, |y| <= K_y w that keep every cut point 2^19 pixels or less from the frame (inside the guard band):
 *   - a projectable vertex of the triangle counts as inside every plane, whatever its coordinates: edges between two
 *     such vertices are never cut and stay identical to the neighbouring triangles' edges (what lies beyond the near
 *     plane on them is removed per sample by the depth range test);
 *   - a cut point is always computed from the inside end g of its edge towards the outside end o,
 *         t = d_g / (d_g - d_o),   (x, y, z, w) = fma(t, o - g, g)     (z = 0 exactly on the near plane),
 *     with d the signed distance to the plane, so that the two triangles sharing the edge get the same point; when the
 *     inside end is inside only by the first rule (d_g <= 0) no point is produced.
 * The convex polygon of projectable vertices that is left is drawn as a fan, under the id of the triangle it came from. */
static int g_near_clip = 0;
SRO_API void sro_set_near_clip(int on) { g_near_clip = on; }

#define CLIP_MAX_VERTICES 8 /* 3 + one per plane */
typedef struct { float c[4]; screen_vertex sv; int original; } clip_vertex; /* original: a projectable vertex of the triangle, sv valid */

static float clip_plane_distance(int plane, const float* c, float kx, float ky)
{
    switch (plane) {
    case 0: return c[2];
    case 1: return kx * c[3] - c[0];
    case 2: return kx * c[3] + c[0];
    case 3: return ky * c[3] - c[1];
    default: return ky * c[3] + c[1];
    }
}

/* returns the number of polygon vertices written to out[CLIP_MAX_VERTICES] (0: nothing to draw), or -1 when a vertex that
 * is left cannot be projected (a degenerate projection): the triangle is dropped and counted */
static int near_clip_polygon(const float* c0, const float* c1, const float* c2, const screen_vertex* s0, const screen_vertex* s1,
                             const screen_vertex* s2, int W, int H, screen_vertex* out)
{
    clip_vertex A[CLIP_MAX_VERTICES], B[CLIP_MAX_VERTICES];
    clip_vertex *src = A, *dst = B;
    const float* c[3] = { c0, c1, c2 };
    const screen_vertex* sv[3] = { s0, s1, s2 };
    int n = 3;
    for (int i = 0; i < 3; ++i) {
        memcpy(src[i].c, c[i], 16);
        src[i].sv = *sv[i];
        src[i].original = sv[i]->rw != 0.0f;
    }
    const float kx = 524288.0f / ((float)W * 0.5f), ky = 524288.0f / ((float)H * 0.5f);
    for (int plane = 0; plane < 5 && n >= 3; ++plane) {
        int m = 0;
        for (int i = 0; i < n; ++i) {
            const clip_vertex* a = &src[i];
            const clip_vertex* b = &src[i + 1 == n ? 0 : i + 1];
            float da = clip_plane_distance(plane, a->c, kx, ky), db = clip_plane_distance(plane, b->c, kx, ky);
            int in_a = a->original || da >= 0.0f, in_b = b->original || db >= 0.0f;
            if (in_a) dst[m++] = *a;
            if (in_a == in_b) continue;
            const clip_vertex* g = in_a ? a : b;
            const clip_vertex* o = in_a ? b : a;
            float dg = in_a ? da : db, dout = in_a ? db : da;
            if (!(dg > 0.0f) || !(dout < 0.0f)) continue;
            float t = dg / (dg - dout);
            clip_vertex* cut = &dst[m++];
            for (int k = 0; k < 4; ++k) cut->c[k] = fmaf(t, o->c[k] - g->c[k], g->c[k]);
            if (plane == 0) cut->c[2] = 0.0f;
            cut->original = 0;
        }
        clip_vertex* tmp = src; src = dst; dst = tmp;
        n = m;
    }
    if (n < 3) return 0;
    for (int i = 0; i < n; ++i) {
        if (src[i].original) out[i] = src[i].sv;
        else project_clip(src[i].c, W, H, &out[i]);
        if (out[i].rw == 0.0f) return -1;
    }
    return n;
}

typedef struct {
    int W, H, S;            /* full frame size and samples per pixel */
    int rx, ry, rw, rh;     /* stored rectangle */
    float* depth;           /* rh*rw*S */
    uint32_t* tri;          /* rh*rw*S, 0xFFFFFFFF = background */
} vis_target;

typedef struct {
    uint64_t tris_in, tris_culled_clip, tris_front, samples_tested, samples_passed, tris_clipped;
} raster_stats;

static void raster_triangle(const tri_setup* tp, uint32_t ti, const vis_target* vt, int band_y0, int band_y1, raster_stats* st)
{
    const tri_setup t = *tp;
    const int S = vt->S;
    const int (*samp)[2] = S == 4 ? g_samp4 : g_samp1;
    int x0 = floor_div256(t.minX), x1 = floor_div256(t.maxX);
    int y0 = floor_div256(t.minY), y1 = floor_div256(t.maxY);
    if (x0 < vt->rx) x0 = vt->rx; if (x1 > vt->rx + vt->rw - 1) x1 = vt->rx + vt->rw - 1;
    if (y0 < band_y0) y0 = band_y0; if (y1 > band_y1 - 1) y1 = band_y1 - 1;
    for (int y = y0; y <= y1; ++y) for (int x = x0; x <= x1; ++x) for (int s = 0; s < S; ++s) {
        int64_t px = (int64_t)x * SUBPIX + samp[s][0], py = (int64_t)y * SUBPIX + samp[s][1];
        int64_t Ea, Eb, Ec; edge_eval(&t, px, py, &Ea, &Eb, &Ec);
        if (st) st->samples_tested++;
        if (Ea + t.bias_a < 0 || Eb + t.bias_b < 0 || Ec + t.bias_c < 0) continue;
        float l0 = (float)Ea * t.rA, l2 = (float)Eb * t.rA, l1 = (float)Ec * t.rA;
        float z = fmaf(l2, t.z2, fmaf(l1, t.z1, l0 * t.z0));
        z = z + 0.0f;
        if (!(z >= 0.0f && z <= 1.0f)) continue;
        size_t o = ((size_t)(y - vt->ry) * vt->rw + (x - vt->rx)) * S + s;
        if (z < vt->depth[o]) { vt->depth[o] = z; vt->tri[o] = ti; if (st) st->samples_passed++; }
    }
}

static void raster_band(const screen_vertex* sv, const float* clip /* 4 per vertex */, const uint32_t* idx, uint32_t n_tris, const vis_target* vt,
                        int band_y0, int band_y1 /* absolute pixel rows */, raster_stats* st)
{
    for (uint32_t ti = 0; ti < n_tris; ++ti) {
        tri_setup t; int needs_clip;
        const uint32_t i0 = idx[3*(size_t)ti], i1 = idx[3*(size_t)ti+1], i2 = idx[3*(size_t)ti+2];
        const screen_vertex *v0 = &sv[i0], *v1 = &sv[i1], *v2 = &sv[i2];
        if (st) st->tris_in++;
        if (!setup_triangle(v0, v1, v2, &t, &needs_clip)) {
            if (!needs_clip) continue;
            if (!g_near_clip) { if (st) st->tris_culled_clip++; continue; }
            screen_vertex poly[CLIP_MAX_VERTICES];
            int n = near_clip_polygon(clip + 4*(size_t)i0, clip + 4*(size_t)i1, clip + 4*(size_t)i2, v0, v1, v2, vt->W, vt->H, poly);
            if (n < 0) { if (st) st->tris_culled_clip++; continue; }
            if (n >= 3 && st) st->tris_clipped++;
            for (int k = 2; k < n; ++k) {
                int nc;
                if (!setup_triangle(&poly[0], &poly[k - 1], &poly[k], &t, &nc)) continue;
                if (st && k == 2) st->tris_front++;
                raster_triangle(&t, ti, vt, band_y0, band_y1, st);
            }
            continue;
        }
        if (st) st->tris_front++;
        raster_triangle(&t, ti, vt, band_y0, band_y1, st);
    }
}

/* ------------------------------------------------------------------------------------ */
/* Textures (src/texture.cpp:87-176, 185-263)                                           */
/* ------------------------------------------------------------------------------------ */
SRO_API int sro_mip_levels(int w, int h)
{
    int m = w > h ? w : h, l = 0;
    while (m > 1) { m >>= 1; ++l; }
    return l + 1; /* floor(log2(max(w,h))) + 1 */
}
static int mip_dim(int d, int l) { int v = d >> l; return v > 0 ? v : 1; }
SRO_API size_t sro_mip_offset(int w, int h, int level) /* in texels */
{
    size_t o = 0;
    for (int l = 0; l < level; ++l) o += (size_t)mip_dim(w, l) * mip_dim(h, l);
    return o;
}

static void decode_texel(const uint8_t* p, float* o)
{
    o[0] = g_srgb_decode[p[0]]; o[1] = g_srgb_decode[p[1]]; o[2] = g_srgb_decode[p[2]];
    o[3] = (float)p[3] / 255.0f;
}
static void bilerp4(const float* t00, const float* t10, const float* t01, const float* t11, float a, float b, float* o)
{
    float w00 = (1.0f - a) * (1.0f - b), w10 = a * (1.0f - b), w01 = (1.0f - a) * b, w11 = a * b;
    for (int c = 0; c < 4; ++c) o[c] = ((w00 * t00[c] + w10 * t10[c]) + w01 * t01[c]) + w11 * t11[c];
}

/* One vkCmdBlitImage with VK_FILTER_LINEAR from level (sw x sh) to (dw x dh), whole image to
 * whole image, R8G8B8A8_SRGB: unnormalised source coordinate (i+0.5)*sw/dw, clamp to edge. */
static void blit_linear(const uint8_t* src, int sw, int sh, uint8_t* dst, int dw, int dh)
{
    float sx = (float)sw / (float)dw, sy = (float)sh / (float)dh;
    for (int j = 0; j < dh; ++j) for (int i = 0; i < dw; ++i) {
        float u = ((float)i + 0.5f) * sx - 0.5f, v = ((float)j + 0.5f) * sy - 0.5f;
        float fu = floorf(u), fv = floorf(v);
        float a = u - fu, b = v - fv;
        int i0 = (int)fu, j0 = (int)fv, i1 = i0 + 1, j1 = j0 + 1;
        if (i0 < 0) i0 = 0; if (i0 > sw - 1) i0 = sw - 1; if (i1 < 0) i1 = 0; if (i1 > sw - 1) i1 = sw - 1;
        if (j0 < 0) j0 = 0; if (j0 > sh - 1) j0 = sh - 1; if (j1 < 0) j1 = 0; if (j1 > sh - 1) j1 = sh - 1;
        float t00[4], t10[4], t01[4], t11[4], r[4];
        decode_texel(src + 4 * ((size_t)j0 * sw + i0), t00);
        decode_texel(src + 4 * ((size_t)j0 * sw + i1), t10);
        decode_texel(src + 4 * ((size_t)j1 * sw + i0), t01);
        decode_texel(src + 4 * ((size_t)j1 * sw + i1), t11);
        bilerp4(t00, t10, t01, t11, a, b, r);
        uint8_t* d = dst + 4 * ((size_t)j * dw + i);
        d[0] = (uint8_t)srgb_encode8(r[0]); d[1] = (uint8_t)srgb_encode8(r[1]);
        d[2] = (uint8_t)srgb_encode8(r[2]); d[3] = (uint8_t)unorm_encode8(r[3]);
    }
}

/* chain: all levels back to back, RGBA8; level 0 is copied from rgba8 */
SRO_API void sro_build_mip_chain(const uint8_t* rgba8, int w, int h, uint8_t* chain)
{
    pthread_once(&g_once, init_tables);
    int levels = sro_mip_levels(w, h);
    memcpy(chain, rgba8, (size_t)w * h * 4);
    for (int l = 1; l < levels; ++l)
        blit_linear(chain + 4 * sro_mip_offset(w, h, l - 1), mip_dim(w, l - 1), mip_dim(h, l - 1),
                    chain + 4 * sro_mip_offset(w, h, l), mip_dim(w, l), mip_dim(h, l));
}

typedef struct {
    const uint8_t* chain; int w, h, levels;
    float min_lod, max_lod, lod_bias, max_aniso; /* sampler of src/texture.cpp:246-263 */
} texture;

static int wrap_repeat(int i, int n) { int m = i % n; return m < 0 ? m + n : m; }

static void sample_level(const texture* t, int level, float s, float tt, float* o)
{
    int wl = mip_dim(t->w, level), hl = mip_dim(t->h, level);
    const uint8_t* base = t->chain + 4 * sro_mip_offset(t->w, t->h, level);
    float u = s * (float)wl - 0.5f, v = tt * (float)hl - 0.5f;
    float fu = floorf(u), fv = floorf(v);
    float a = u - fu, b = v - fv;
    const float lim = 1073741824.0f;
    if (!(fu > -lim)) fu = -lim; if (fu > lim) fu = lim;
    if (!(fv > -lim)) fv = -lim; if (fv > lim) fv = lim;
    int i0 = wrap_repeat((int)fu, wl), j0 = wrap_repeat((int)fv, hl);
    int i1 = wrap_repeat(i0 + 1, wl), j1 = wrap_repeat(j0 + 1, hl);
    float t00[4], t10[4], t01[4], t11[4];
    decode_texel(base + 4 * ((size_t)j0 * wl + i0), t00);
    decode_texel(base + 4 * ((size_t)j0 * wl + i1), t10);
    decode_texel(base + 4 * ((size_t)j1 * wl + i0), t01);
    decode_texel(base + 4 * ((size_t)j1 * wl + i1), t11);
    bilerp4(t00, t10, t01, t11, a, b, o);
}

/* Anisotropic filtering is off unless a test switches it on: the reference's sampler asks for 16x
 * (src/texture.cpp:255-256), the footprint is implementation-defined, and the kernels apply it only in
 * their "anisotropic" pipeline variant.  When on, this is the method the Vulkan specification spells
 * out (Scale Factor Operation / Texel Anisotropic Filtering): eta = min(rho_max / rho_min, maxAnisotropy),
 * N = ceil(eta), lambda from rho_max / eta, and the mean of N trilinear samples placed at
 * (i / (N + 1) - 1/2), i = 1..N, along the derivative of the major axis. */
static int g_anisotropic = 0;
SRO_API void sro_set_anisotropic(int on) { g_anisotropic = on; }

static void sample_trilinear(const texture* t, float lambda, float s, float tt, float* o, float* lod_out)
{
    if (!(lambda > t->min_lod)) lambda = t->min_lod;  /* also catches -inf and NaN */
    if (lambda > t->max_lod) lambda = t->max_lod;
    float q = (float)(t->levels - 1);
    float d = lambda; if (d < 0.0f) d = 0.0f; if (d > q) d = q;
    float dhi = floorf(d), delta = d - dhi;
    int lhi = (int)dhi, llo = lhi + 1 > t->levels - 1 ? t->levels - 1 : lhi + 1;
    if (lod_out) *lod_out = d;
    float chi[4], clo[4];
    sample_level(t, lhi, s, tt, chi);
    if (delta == 0.0f) { memcpy(o, chi, 16); return; }
    sample_level(t, llo, s, tt, clo);
    for (int c = 0; c < 4; ++c) o[c] = (1.0f - delta) * chi[c] + delta * clo[c];
}

static void sample_texture_anisotropic(const texture* t, float s, float tt, float dsdx, float dtdx, float dsdy, float dtdy, float* o, float* lod_out)
{
    float mux = dsdx * (float)t->w, mvx = dtdx * (float)t->h;
    float muy = dsdy * (float)t->w, mvy = dtdy * (float)t->h;
    float rx = sqrtf(mux * mux + mvx * mvx), ry = sqrtf(muy * muy + mvy * mvy);
    int x_major = rx > ry;
    float rmax = x_major ? rx : ry, rmin = x_major ? ry : rx;
    float eta = 1.0f;
    if (rmax > 0.0f) {
        eta = rmin > 0.0f ? rmax / rmin : t->max_aniso;
        if (!(eta < t->max_aniso)) eta = t->max_aniso;
        if (!(eta > 1.0f)) eta = 1.0f;
    }
    int n = (int)ceilf(eta);
    float lambda = log2f(rmax / eta) + t->lod_bias;
    float ds = x_major ? dsdx : dsdy, dt = x_major ? dtdx : dtdy;
    float acc[4] = { 0.0f, 0.0f, 0.0f, 0.0f };
    for (int i = 1; i <= n; ++i) {
        float off = (float)i / (float)(n + 1) - 0.5f, tap[4];
        sample_trilinear(t, lambda, s + off * ds, tt + off * dt, tap, lod_out);
        for (int c = 0; c < 4; ++c) acc[c] += tap[c];
    }
    for (int c = 0; c < 4; ++c) o[c] = acc[c] / (float)n;
}

/* texture(sampler2D, uv) with implicit derivatives given explicitly */
static void sample_texture(const texture* t, float s, float tt, float dsdx, float dtdx, float dsdy, float dtdy, float* o, float* lod_out)
{
    if (g_anisotropic && t->max_aniso > 1.0f) { sample_texture_anisotropic(t, s, tt, dsdx, dtdx, dsdy, dtdy, o, lod_out); return; }
    float mux = dsdx * (float)t->w, mvx = dtdx * (float)t->h;
    float muy = dsdy * (float)t->w, mvy = dtdy * (float)t->h;
    float rx2 = mux * mux + mvx * mvx, ry2 = muy * muy + mvy * mvy;
    float r2 = rx2 > ry2 ? rx2 : ry2;
    float lambda = 0.5f * log2f(r2) + t->lod_bias;
    if (!(lambda > t->min_lod)) lambda = t->min_lod;  /* also catches -inf and NaN */
    if (lambda > t->max_lod) lambda = t->max_lod;
    float q = (float)(t->levels - 1);
    float d = lambda; if (d < 0.0f) d = 0.0f; if (d > q) d = q;
    float dhi = floorf(d), delta = d - dhi;
    int lhi = (int)dhi, llo = lhi + 1 > t->levels - 1 ? t->levels - 1 : lhi + 1;
    if (lod_out) *lod_out = d;
    float chi[4], clo[4];
    sample_level(t, lhi, s, tt, chi);
    if (delta == 0.0f) { memcpy(o, chi, 16); return; }
    sample_level(t, llo, s, tt, clo);
    for (int c = 0; c < 4; ++c) o[c] = (1.0f - delta) * chi[c] + delta * clo[c];
}

/* direct access for tests: sample with explicit derivatives; chain laid out as by sro_build_mip_chain */
SRO_API void sro_sample(const uint8_t* chain, int w, int h, float s, float t, float dsdx, float dtdx, float dsdy, float dtdy,
                        float* rgba_out, float* lod_out)
{
    pthread_once(&g_once, init_tables);
    texture tx = { chain, w, h, sro_mip_levels(w, h), 0.0f, (float)sro_mip_levels(w, h), 0.0f, 16.0f };
    sample_texture(&tx, s, t, dsdx, dtdx, dsdy, dtdy, rgba_out, lod_out);
}

/* ------------------------------------------------------------------------------------ */
/* Fragment shader (shaders/shader.frag:16-152)                                         */
/* ------------------------------------------------------------------------------------ */
static float max0(float x) { return x > 0.0f ? x : 0.0f; }
static float glsl_pow(float x, float y) { return powf(x, y); }
static void glsl_normalize(const float* a, float* o) { v3_normalize(a, o); }

static const float kPI = 3.14159265359f;
static const float kLightPos[4][3] = { {-10.0f, 10.0f, 10.0f}, {10.0f, 10.0f, 10.0f}, {-10.0f, -10.0f, 10.0f}, {10.0f, -10.0f, 10.0f} };
static const float kLightPower[3] = { 23.47f, 21.31f, 20.79f };

static float distribution_ggx(const float* N, const float* Hv, float roughness)
{
    float a = roughness * roughness, a2 = a * a;
    float NdotH = max0(v3_dot(N, Hv)), NdotH2 = NdotH * NdotH;
    float denom = NdotH2 * (a2 - 1.0f) + 1.0f;
    denom = kPI * denom * denom;
    return a2 / denom;
}
static float geometry_schlick_ggx(float NdotV, float roughness)
{
    float r = roughness + 1.0f, k = (r * r) / 8.0f;
    return NdotV / (NdotV * (1.0f - k) + k);
}
static float geometry_smith(const float* N, const float* V, const float* L, float roughness)
{
    float ggx2 = geometry_schlick_ggx(max0(v3_dot(N, V)), roughness);
    float ggx1 = geometry_schlick_ggx(max0(v3_dot(N, L)), roughness);
    return ggx1 * ggx2;
}

/* inputs: interpolated WorldPos, Normal (not normalised, as in the shader), camera position and the
 * three filtered texture values.  output: linear-space RGB written to the sRGB attachment */
static void fragment_main(const float* wp, const float* N, const float* cam, const float* albedo_tex,
                          float metallic, float roughness, float* out)
{
    float albedo[3] = { glsl_pow(albedo_tex[0], 2.2f), glsl_pow(albedo_tex[1], 2.2f), glsl_pow(albedo_tex[2], 2.2f) };
    const float ao = 0.01f;
    float V[3], t[3];
    v3_sub(cam, wp, t); glsl_normalize(t, V);
    float F0[3];
    for (int c = 0; c < 3; ++c) F0[c] = 0.04f * (1.0f - metallic) + albedo[c] * metallic;
    float Lo[3] = { 0, 0, 0 };
    for (int i = 0; i < 4; ++i) {
        float L[3], Hv[3];
        v3_sub(kLightPos[i], wp, t); glsl_normalize(t, L);
        t[0] = V[0] + L[0]; t[1] = V[1] + L[1]; t[2] = V[2] + L[2]; glsl_normalize(t, Hv);
        float D[3]; v3_sub(kLightPos[i], wp, D);
        float attenuation = 1.0f / v3_dot(D, D);
        float NDF = distribution_ggx(N, Hv, roughness);
        float G = geometry_smith(N, V, L, roughness);
        float fc = glsl_pow(1.0f - max0(v3_dot(Hv, V)), 5.0f);
        float NdotL = max0(v3_dot(N, L));
        float denominator = 4.0f * max0(v3_dot(N, V)) * NdotL + 0.001f;
        float dn = denominator > 0.001f ? denominator : 0.001f;
        for (int c = 0; c < 3; ++c) {
            float F = F0[c] + (1.0f - F0[c]) * fc;
            float specular = (NDF * G * F) / dn;
            float kD = (1.0f - F) * (1.0f - metallic);
            float radiance = kLightPower[c] * attenuation;
            Lo[c] += (kD * albedo[c] / kPI + specular) * radiance * NdotL;
        }
    }
    for (int c = 0; c < 3; ++c) {
        float ambient = 0.03f * albedo[c] * ao;
        float color = ambient + Lo[c];
        color = color / (color + 1.0f);
        out[c] = glsl_pow(color, 1.0f / 2.2f);
    }
}

/* exported for closed-form tests of the shader alone */
SRO_API void sro_fragment_main(const float* wp, const float* N, const float* cam, const float* albedo_tex,
                               float metallic, float roughness, float* out_rgb)
{
    fragment_main(wp, N, cam, albedo_tex, metallic, roughness, out_rgb);
}

/* ------------------------------------------------------------------------------------ */
/* Deferred shading of the visibility result                                            */
/* ------------------------------------------------------------------------------------ */
typedef struct {
    const float* verts; const uint32_t* idx; const screen_vertex* sv; const float* clip; const view_xform* vx; int W, H;
    texture tex[4]; /* albedo, normal, metallic, roughness (binding order of src/main.cpp:461-516) */
} shade_ctx;

/* perspective-correct weights of v0,v1,v2 at a fixed-point position */
static void persp_weights(const tri_setup* t, int64_t px, int64_t py, float* p0, float* p1, float* p2)
{
    int64_t Ea, Eb, Ec; edge_eval(t, px, py, &Ea, &Eb, &Ec);
    float l0 = (float)Ea * t->rA, l2 = (float)Eb * t->rA, l1 = (float)Ec * t->rA;
    float q0 = l0 * t->rw0, q1 = l1 * t->rw1, q2 = l2 * t->rw2;
    float rden = 1.0f / ((q0 + q1) + q2);
    *p0 = q0 * rden; *p1 = q1 * rden; *p2 = q2 * rden;
}
/* The same weights for a triangle that reached the frame through the near-plane clipper: one of its vertices has no
 * window position, so the weights come from the clip-space coordinates (x, y, w) of the original triangle.  With
 * h_i = (x_i - w_i X, y_i - w_i Y) for the pixel's normalised device coordinates (X, Y), the perspective-correct
 * weight of vertex i is proportional to cross(h_j, h_k).  Double precision. */
static void persp_weights_clip(const float* c0, const float* c1, const float* c2, int W, int H, int64_t px, int64_t py,
                               float* p0, float* p1, float* p2)
{
    double hw = 0.5 * W, hh = 0.5 * H;
    double X = ((double)px / SUBPIX - hw) / hw, Y = (hh - (double)py / SUBPIX) / hh;
    double h0x = c0[0] - c0[3] * X, h0y = c0[1] - c0[3] * Y, h1x = c1[0] - c1[3] * X, h1y = c1[1] - c1[3] * Y;
    double h2x = c2[0] - c2[3] * X, h2y = c2[1] - c2[3] * Y;
    double q0 = h1x * h2y - h1y * h2x, q1 = h2x * h0y - h2y * h0x, q2 = h0x * h1y - h0y * h1x;
    double rden = 1.0 / ((q0 + q1) + q2);
    *p0 = (float)(q0 * rden); *p1 = (float)(q1 * rden); *p2 = (float)(q2 * rden);
}
static float interp(float p0, float p1, float p2, float a0, float a1, float a2)
{
    return fmaf(p2, a2, fmaf(p1, a1, p0 * a0));
}

/* getNormalFromMap of shaders/shader.frag:35-53 with its body (commented out in the reference) restored; off by default */
static int g_normal_mapping = 0;
SRO_API void sro_set_normal_mapping(int on) { g_normal_mapping = on; }

/* shade triangle ti at pixel (x,y): returns packed B,G,R,A bytes */
static uint32_t shade_pixel(const shade_ctx* c, uint32_t ti, int x, int y)
{
    const uint32_t i0 = c->idx[3*(size_t)ti], i1 = c->idx[3*(size_t)ti+1], i2 = c->idx[3*(size_t)ti+2];
    tri_setup t; int nc;
    setup_triangle(&c->sv[i0], &c->sv[i1], &c->sv[i2], &t, &nc);
    const float *k0 = c->clip + 4*(size_t)i0, *k1 = c->clip + 4*(size_t)i1, *k2 = c->clip + 4*(size_t)i2;
    const float *a0 = c->verts + 8*(size_t)i0, *a1 = c->verts + 8*(size_t)i1, *a2 = c->verts + 8*(size_t)i2;
    float p0, p1, p2;
    if (nc) persp_weights_clip(k0, k1, k2, c->W, c->H, (int64_t)x * SUBPIX + HALF, (int64_t)y * SUBPIX + HALF, &p0, &p1, &p2);
    else persp_weights(&t, (int64_t)x * SUBPIX + HALF, (int64_t)y * SUBPIX + HALF, &p0, &p1, &p2);
    float w0[3], w1[3], w2[3], n0[3], n1[3], n2[3];
    world_pos(c->vx, a0, w0); world_pos(c->vx, a1, w1); world_pos(c->vx, a2, w2);
    mat3_mul_vec3(c->vx->model, a0 + 3, n0); mat3_mul_vec3(c->vx->model, a1 + 3, n1); mat3_mul_vec3(c->vx->model, a2 + 3, n2);
    float wp[3], N[3];
    for (int k = 0; k < 3; ++k) { wp[k] = interp(p0, p1, p2, w0[k], w1[k], w2[k]); N[k] = interp(p0, p1, p2, n0[k], n1[k], n2[k]); }
    float u = interp(p0, p1, p2, a0[6], a1[6], a2[6]), v = interp(p0, p1, p2, a0[7], a1[7], a2[7]);
    /* coarse derivatives over the 2x2 quad, evaluated for this triangle (helper invocations) */
    int qx = x & ~1, qy = y & ~1;
    float uq[3], vq[3], wq[3][3];
    const int off[3][2] = { {0, 0}, {1, 0}, {0, 1} };
    for (int k = 0; k < 3; ++k) {
        float r0, r1, r2;
        if (nc) persp_weights_clip(k0, k1, k2, c->W, c->H, (int64_t)(qx + off[k][0]) * SUBPIX + HALF, (int64_t)(qy + off[k][1]) * SUBPIX + HALF, &r0, &r1, &r2);
        else persp_weights(&t, (int64_t)(qx + off[k][0]) * SUBPIX + HALF, (int64_t)(qy + off[k][1]) * SUBPIX + HALF, &r0, &r1, &r2);
        uq[k] = interp(r0, r1, r2, a0[6], a1[6], a2[6]);
        vq[k] = interp(r0, r1, r2, a0[7], a1[7], a2[7]);
        for (int j = 0; j < 3; ++j) wq[k][j] = interp(r0, r1, r2, w0[j], w1[j], w2[j]);
    }
    float dudx = uq[1] - uq[0], dvdx = vq[1] - vq[0], dudy = uq[2] - uq[0], dvdy = vq[2] - vq[0];
    if (g_normal_mapping) {
        float tn[4], Q1[3], Q2[3], Nn[3], T[3], B[3], M[3];
        sample_texture(&c->tex[1], u, v, dudx, dvdx, dudy, dvdy, tn, NULL);       /* texture(normalMapSampler, TexCoord).xyz * 2.0 - 1.0 */
        for (int j = 0; j < 3; ++j) {
            tn[j] = tn[j] * 2.0f - 1.0f;
            Q1[j] = wq[1][j] - wq[0][j]; Q2[j] = wq[2][j] - wq[0][j];             /* dFdx(WorldPos), dFdy(WorldPos) */
            T[j] = Q1[j] * dvdy - Q2[j] * dvdx;                                   /* Q1*st2.t - Q2*st1.t */
        }
        glsl_normalize(N, Nn);
        glsl_normalize(T, T);
        v3_cross(Nn, T, B);
        glsl_normalize(B, B);                                                     /* B = -normalize(cross(N, T)) */
        for (int j = 0; j < 3; ++j) M[j] = (T[j] * tn[0] - B[j] * tn[1]) + Nn[j] * tn[2];   /* mat3(T, B, N) * tangentNormal */
        glsl_normalize(M, N);
    }
    float alb[4], met[4], rgh[4], rgb[3];
    sample_texture(&c->tex[0], u, v, dudx, dvdx, dudy, dvdy, alb, NULL);
    sample_texture(&c->tex[2], u, v, dudx, dvdx, dudy, dvdy, met, NULL);
    sample_texture(&c->tex[3], u, v, dudx, dvdx, dudy, dvdy, rgh, NULL);
    fragment_main(wp, N, c->vx->cam, alb, met[0], rgh[0], rgb);
    uint32_t R = (uint32_t)srgb_encode8(rgb[0] > 1.0f ? 1.0f : rgb[0]);
    uint32_t G = (uint32_t)srgb_encode8(rgb[1] > 1.0f ? 1.0f : rgb[1]);
    uint32_t B = (uint32_t)srgb_encode8(rgb[2] > 1.0f ? 1.0f : rgb[2]);
    return B | (G << 8) | (R << 16) | (255u << 24); /* B8G8R8A8 in memory order */
}

static void shade_band(const shade_ctx* c, const vis_target* vt, uint8_t* bgra /* rh*rw*4 */, int y0, int y1, uint64_t* shaded_px)
{
    const int S = vt->S;
    const uint32_t clear = 0xFF000000u; /* (0,0,0,1) */
    for (int y = y0; y < y1; ++y) for (int x = vt->rx; x < vt->rx + vt->rw; ++x) {
        size_t po = (size_t)(y - vt->ry) * vt->rw + (x - vt->rx);
        const uint32_t* ids = vt->tri + po * S;
        uint32_t col[4]; int any = 0;
        for (int s = 0; s < S; ++s) {
            if (ids[s] == 0xFFFFFFFFu) { col[s] = clear; continue; }
            any = 1;
            int prev = -1;
            for (int k = 0; k < s; ++k) if (ids[k] == ids[s]) { prev = k; break; }
            col[s] = prev >= 0 ? col[prev] : shade_pixel(c, ids[s], x, y);
        }
        if (any && shaded_px) (*shaded_px)++;
        uint32_t outc;
        if (S == 1) outc = col[0];
        else {
            uint32_t ch[4];
            for (int k = 0; k < 4; ++k) {
                float acc = 0.0f;
                for (int s = 0; s < S; ++s) {
                    uint32_t b = (col[s] >> (8 * k)) & 255u;
                    acc += k == 3 ? (float)b / 255.0f : g_srgb_decode[b];
                }
                acc = acc / (float)S;
                ch[k] = k == 3 ? (uint32_t)unorm_encode8(acc) : (uint32_t)srgb_encode8(acc);
            }
            outc = ch[0] | (ch[1] << 8) | (ch[2] << 16) | (ch[3] << 24);
        }
        memcpy(bgra + 4 * po, &outc, 4);
    }
}

/* ------------------------------------------------------------------------------------ */
/* Whole draw                                                                           */
/* ------------------------------------------------------------------------------------ */
typedef struct {
    /* inputs */
    const float* ubo52;          /* model, view, proj (column-major), camera_pos */
    const float* verts; uint32_t n_verts;   /* 8 floats per vertex */
    const uint32_t* indices; uint32_t n_indices;
    const uint8_t* tex_chain[4]; int tex_w[4], tex_h[4];  /* chains from sro_build_mip_chain */
    int W, H, S;                 /* frame size, samples (1 or 4) */
    int rx, ry, rw, rh;          /* stored rectangle (full frame: 0,0,W,H) */
    int threads;                 /* >= 1 */
    /* outputs (any may be NULL) */
    float* depth_out;            /* rh*rw*S */
    uint32_t* tri_out;           /* rh*rw*S */
    uint8_t* bgra_out;           /* rh*rw*4 */
    uint64_t* stats_out;         /* 8 values: tris_in, tris_front, tris_clip_culled, samples_tested, samples_passed, shaded_px, tris_clipped, 0 */
} sro_draw_args;

typedef struct {
    const sro_draw_args* a; const screen_vertex* sv; const float* clip; const vis_target* vt; const shade_ctx* sc;
    int y0, y1; raster_stats rs; uint64_t shaded; int phase;
} band_job;

static void* band_main(void* p)
{
    band_job* j = (band_job*)p;
    if (j->phase == 0) raster_band(j->sv, j->clip, j->a->indices, j->a->n_indices / 3, j->vt, j->y0, j->y1, &j->rs);
    else if (j->a->bgra_out) shade_band(j->sc, j->vt, j->a->bgra_out, j->y0, j->y1, &j->shaded);
    return NULL;
}

SRO_API int sro_draw(const sro_draw_args* a)
{
    pthread_once(&g_once, init_tables);
    if (!(a->S == 1 || a->S == 4) || a->W <= 0 || a->H <= 0 || a->n_indices % 3) return -1;
    if (a->rx < 0 || a->ry < 0 || a->rw <= 0 || a->rh <= 0 || a->rx + a->rw > a->W || a->ry + a->rh > a->H) return -1;
    for (uint32_t i = 0; i < a->n_indices; ++i) if (a->indices[i] >= a->n_verts) return -2;
    view_xform vx; view_from_ubo(a->ubo52, &vx);
    screen_vertex* sv = (screen_vertex*)malloc(sizeof(screen_vertex) * (size_t)(a->n_verts ? a->n_verts : 1));
    float* clip = (float*)malloc(16 * (size_t)(a->n_verts ? a->n_verts : 1));
    for (uint32_t i = 0; i < a->n_verts; ++i) project_vertex(&vx, a->verts + 8*(size_t)i, a->W, a->H, clip + 4*(size_t)i, &sv[i]);
    size_t ns = (size_t)a->rw * a->rh * a->S;
    vis_target vt = { a->W, a->H, a->S, a->rx, a->ry, a->rw, a->rh, NULL, NULL };
    vt.depth = a->depth_out ? a->depth_out : (float*)malloc(ns * 4);
    vt.tri = a->tri_out ? a->tri_out : (uint32_t*)malloc(ns * 4);
    for (size_t i = 0; i < ns; ++i) { vt.depth[i] = 1.0f; vt.tri[i] = 0xFFFFFFFFu; } /* clear, src/main.cpp:919-921 */
    shade_ctx sc; memset(&sc, 0, sizeof sc);
    sc.verts = a->verts; sc.idx = a->indices; sc.sv = sv; sc.clip = clip; sc.vx = &vx; sc.W = a->W; sc.H = a->H;
    for (int k = 0; k < 4; ++k) {
        sc.tex[k].chain = a->tex_chain[k]; sc.tex[k].w = a->tex_w[k]; sc.tex[k].h = a->tex_h[k];
        sc.tex[k].levels = a->tex_chain[k] ? sro_mip_levels(a->tex_w[k], a->tex_h[k]) : 0;
        sc.tex[k].min_lod = 0.0f; sc.tex[k].max_lod = (float)sc.tex[k].levels; sc.tex[k].lod_bias = 0.0f; sc.tex[k].max_aniso = 16.0f;
    }
    int nt = a->threads < 1 ? 1 : a->threads; if (nt > 256) nt = 256; if (nt > a->rh) nt = a->rh;
    band_job* jobs = (band_job*)calloc((size_t)nt, sizeof(band_job));
    pthread_t* th = (pthread_t*)calloc((size_t)nt, sizeof(pthread_t));
    uint64_t st[8] = { 0 };
    for (int phase = 0; phase < 2; ++phase) {
        for (int k = 0; k < nt; ++k) {
            band_job* j = &jobs[k];
            j->a = a; j->sv = sv; j->clip = clip; j->vt = &vt; j->sc = &sc; j->phase = phase;
            j->y0 = a->ry + (int)((int64_t)a->rh * k / nt); j->y1 = a->ry + (int)((int64_t)a->rh * (k + 1) / nt);
            if (nt == 1) band_main(j); else pthread_create(&th[k], NULL, band_main, j);
        }
        if (nt > 1) for (int k = 0; k < nt; ++k) pthread_join(th[k], NULL);
    }
    for (int k = 0; k < nt; ++k) {
        st[3] += jobs[k].rs.samples_tested; st[4] += jobs[k].rs.samples_passed; st[5] += jobs[k].shaded;
    }
    /* per-triangle counts are identical in every band; take them from band 0 */
    st[0] = jobs[0].rs.tris_in; st[1] = jobs[0].rs.tris_front; st[2] = jobs[0].rs.tris_culled_clip; st[6] = jobs[0].rs.tris_clipped;
    if (a->stats_out) memcpy(a->stats_out, st, sizeof st);
    free(jobs); free(th); free(sv); free(clip);
    if (!a->depth_out) free(vt.depth);
    if (!a->tri_out) free(vt.tri);
    return 0;
}

SRO_API size_t sro_draw_args_size(void) { return sizeof(sro_draw_args); }
