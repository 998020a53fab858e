/*
 * softraster.h -- C ABI of the MI355X (gfx950) compute rasteriser that stands behind the draw path
 * of LesleyLai/Vulkan-Renderer.
 *
 * The reference has no plugin or FFI interface; the seam is its own host API.  Each entry point
 * below names the reference call site it replaces (paths relative to the reference root).  All
 * functions return SR_OK (0) or a negative sr_status; nothing throws or aborts across this boundary
 * (the reference's own convention, beyond::expected<T,VkResult> in src/vulkan_helper/buffer.hpp:84-90
 * and VKH_CHECK/panic in src/vulkan_helper/check.hpp:8-15, can be rebuilt on top: see the C++ shim).
 *
 * Threading: like the reference (one host thread, src/main.cpp:163-171) calls on one device handle
 * are not thread-safe.  Every handle owns one HIP stream; sr_draw is asynchronous on it,
 * read-backs synchronise.
 */
#ifndef SOFTRASTER_H
#define SOFTRASTER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SR_API __attribute__((visibility("default")))

typedef enum sr_status {
    SR_OK = 0,
    SR_ERR_INVALID_ARGUMENT = -1,
    SR_ERR_OUT_OF_MEMORY = -2,
    SR_ERR_DEVICE = -3,         /* a HIP call failed; sr_last_error_string() has the HIP message */
    SR_ERR_INDEX_OUT_OF_RANGE = -4,
    SR_ERR_UNSUPPORTED = -5,
    SR_ERR_NO_DEVICE = -6,      /* no gfx950 device visible: the library never falls back to the CPU */
    SR_ERR_FILE = -7            /* a file cannot be read, or is not what the reader understands; sr_last_error_string() says why */
} sr_status;

typedef struct sr_device_t* sr_device;
typedef struct sr_mesh_t* sr_mesh;
typedef struct sr_texture_t* sr_texture;
typedef struct sr_target_t* sr_target;
typedef struct sr_model_t* sr_model;

/* Sampler state; sr_sampler_default() returns the one src/texture.cpp:246-263 creates. */
typedef struct sr_sampler_desc {
    float min_lod;        /* 0 */
    float max_lod;        /* (float)mip_levels; a negative value means "use mip_levels" */
    float mip_lod_bias;   /* 0 */
    float max_anisotropy; /* 16 (anisotropyEnable, src/texture.cpp:255-256).  > 1: the footprint the Vulkan specification
                             describes (N = ceil(min(rho_max/rho_min, max)) trilinear taps along the major axis);
                             <= 1: isotropic trilinear filtering */
} sr_sampler_desc;

typedef struct sr_rect { int32_t x, y, width, height; } sr_rect;

/* What sr_target_readback returns. */
typedef enum sr_readback_kind {
    SR_READ_COLOR_BGRA8 = 0, /* height*width*4 bytes, B,G,R,A: the swap-chain format, src/vulkan_helper/swapchain.cpp:17-21 */
    SR_READ_DEPTH_F32 = 1,   /* height*width*samples floats; 1.0 where nothing was drawn (clear, src/main.cpp:921) */
    SR_READ_TRI_ID_U32 = 2,  /* height*width*samples; 0xFFFFFFFF where nothing was drawn */
    SR_READ_VIS_U64 = 3      /* height*width*samples; (depth bits << 32) | triangle id */
} sr_readback_kind;

/* Work counters of the last sr_draw on a target (all views summed). */
typedef struct sr_draw_stats {
    uint64_t vertices;            /* V: vertices transformed */
    uint64_t triangles;           /* T: triangles submitted */
    uint64_t triangles_binned;    /* surviving triangles that were not resolved by the set-up kernel itself */
    uint64_t tile_refs;           /* (triangle, tile) references emitted for them; 0 if they were resolved without tile lists */
    uint64_t triangles_unprojectable; /* dropped: triangles with a vertex at w <= 0 or outside the guard band that the near-plane
                                         clipper could not handle (more than 65536 of them in one pass, or a degenerate projection);
                                         a frame with a non-zero count is incomplete */
    uint64_t shaded_pixels;       /* Pv: pixels with at least one covered sample */
    uint64_t visible_triangles;   /* F: distinct triangles owning at least one sample */
    uint64_t pixels;              /* P: pixels of the stored rectangle */
    uint64_t triangles_direct;    /* micro-triangles resolved by the set-up kernel itself (they cover >= 1 sample) */
    uint64_t triangles_clipped;   /* triangles with a vertex behind the eye (or outside the guard band) that were cut at the near
                                     plane and drawn: the fixed-function clipping of the reference's pipeline
                                     (depthClampEnable = VK_FALSE, src/main.cpp:586) */
    uint64_t passes;              /* passes the draw ran as: the views of a batch are drawn as many at a time as the per-view scratch
                                     buffers fit the budget; positions and indices are read once per pass, not once per view */
} sr_draw_stats;

/* How the draws of a device were issued since it was created (cumulative).  A draw whose target's previous draw of the same
 * mesh and view count left nothing for the tile lists and the clipper is issued "lean" (without those kernels); if its own
 * counters then show that it needed them, it is issued again in full before anything of it can be observed. */
typedef struct sr_draw_counts {
    uint64_t draws;               /* sr_draw / sr_draw_no_alloc calls that issued kernels */
    uint64_t lean_draws;          /* of them, issued lean (including those that also ran the direct rasteriser for a few pending triangles) */
    uint64_t direct_draws;        /* lean draws that resolved their pending triangles straight into the visibility buffer instead of through tile lists */
    uint64_t lean_redraws;        /* lean draws that had to be issued again in full */
    uint64_t overflow_redraws;    /* draws issued again because the tile reference list was too small */
    uint64_t chunks_tested;       /* one-view draws into a target restricted to a rectangle test every chunk of 768 consecutive triangles */
    uint64_t chunks_culled;       /* against the rectangle: chunks (and the vertices only they use) that were skipped because no */
                                  /* triangle of theirs can touch a stored pixel.  The frame is the same with and without. */
    uint64_t chunk_list_draws;    /* draws of two views or more of a mesh that has chunk vertex lists: set-up workgroups that take several */
                                  /* views load the distinct vertices of their 768 triangles once per view (about 450 loads instead of */
                                  /* 2304 on a regular mesh).  A mesh has no lists when a chunk uses more than 1024 distinct vertices or */
                                  /* vertex numbers 65536 apart.  The frame is the same with and without. */
} sr_draw_counts;

/* Durations of the last profiled sr_draw, milliseconds, from HIP events on the device's stream. */
typedef struct sr_stage_times {
    float clear_ms, vertex_ms, setup_ms, bin_ms, scan_ms, fill_ms, raster_ms, fragment_ms, total_ms;
} sr_stage_times;

/* ---- device: replaces vkh::GPUDevice, src/vulkan_helper/gpu_device.hpp:42-48 (constructed at
 * src/main.cpp:144).  Fails with SR_ERR_NO_DEVICE when the ordinal is not a gfx950 GPU. */
SR_API int sr_device_create(int hip_ordinal, sr_device* out);
SR_API int sr_device_destroy(sr_device dev);
/* vkDeviceWaitIdle, src/main.cpp:170 */
SR_API int sr_device_synchronize(sr_device dev);
SR_API const char* sr_status_string(int status);
SR_API const char* sr_last_error_string(void);

/* ---- mesh: replaces create_mesh_from_data, src/mesh.hpp:69-72 / src/mesh.cpp:79-87.
 * vertices: n_vertices records of the reference's Vertex (32 bytes: pos f32x3, normal f32x3,
 * uv f32x2; src/mesh.hpp:18-54).  indices: uint32 triangle list (src/main.cpp:561, 947-948).
 * Data is fully uploaded on return, as with the reference's blocking staging copies.
 * At most 2^29 - 4 vertices (SR_ERR_UNSUPPORTED beyond; 17 GB of vertex records).
 * A mesh (and a texture) may be destroyed at any time after the draws that use it were issued: sr_mesh_destroy and
 * sr_texture_destroy first complete those draws, as the reference waits for the device to be idle before it
 * destroys its buffers (src/main.cpp:170, 300-347). */
SR_API int sr_mesh_create(sr_device dev, const void* vertices, uint32_t n_vertices,
                          const uint32_t* indices, uint32_t n_indices, sr_mesh* out);
SR_API int sr_mesh_destroy(sr_device dev, sr_mesh mesh);

/* ---- texture: replaces create_texture / destroy_texture, src/texture.hpp:17-24.
 * rgba8: width*height*4 bytes interpreted as VK_FORMAT_R8G8B8A8_SRGB (src/texture.cpp:209).
 * Builds floor(log2(max(w,h)))+1 levels in device memory by successive linear blits
 * (src/texture.cpp:87-176).  sampler == NULL selects sr_sampler_default(). */
SR_API void sr_sampler_default(sr_sampler_desc* out);
SR_API int sr_texture_create_rgba8_srgb(sr_device dev, const void* rgba8, int width, int height,
                                        const sr_sampler_desc* sampler, sr_texture* out);
SR_API int sr_texture_destroy(sr_device dev, sr_texture tex);
SR_API int sr_texture_mip_levels(sr_texture tex);
/* size of level 0 (no reference counterpart: the reference's Texture does not keep it) */
SR_API int sr_texture_size(sr_texture tex, int* width, int* height);
/* copies one level (RGBA8, max(1,w>>level) * max(1,h>>level) * 4 bytes) to the host; no reference
 * counterpart (the reference never reads an image back); used to check the mip chain */
SR_API int sr_texture_read_level(sr_device dev, sr_texture tex, int level, void* host_rgba8, size_t bytes);

/* ---- files.  The reference reads its assets through stb_image and Assimp; neither exists here, so the
 * library reads the formats the reference's data directory uses: PNG, JPEG and glTF 2.0 (.gltf / .glb).
 *
 * sr_texture_create_from_file replaces create_texture_from_file, src/texture.hpp:20 / src/texture.cpp:273-291:
 * the file is expanded to 8-bit RGBA as stbi_load(path, ..., STBI_rgb_alpha) does (gray -> R=G=B, no alpha
 * -> 255, 16-bit -> high byte), then handed to sr_texture_create_rgba8_srgb.  PNG and
 * Huffman-coded 8-bit JPEG (baseline and progressive; gray, YCbCr, RGB).  JPEG decoders differ in the last bit
 * of the inverse DCT, the chroma filter and the colour conversion: the default follows stb_image's arithmetic,
 * the environment variable SR_JPEG_DECODER=libjpeg selects libjpeg's. */
SR_API int sr_texture_create_from_file(sr_device dev, const char* path, const sr_sampler_desc* sampler, sr_texture* out);
/* the decoding step alone (host only, no device): *rgba8_out is width*height*4 bytes, released with sr_image_free */
SR_API int sr_image_read_file(const char* path, void** rgba8_out, int* width, int* height);
SR_API void sr_image_free(void* rgba8);

/* sr_model_load replaces Model::load, src/model.hpp:24 / src/model.cpp:81-98: one mesh per primitive met on a
 * depth-first walk of the default scene's nodes; node transforms ignored (src/model.cpp:61-77); position, normal
 * and first uv set with v flipped (aiProcess_FlipUVs); strips and fans triangulated (aiProcess_Triangulate);
 * fails like the reference when a primitive has no material ("No material!") or the material's base-colour
 * texture is not embedded in the file ("Cannot load diffuse texture", src/model.cpp:36-54).
 * The meshes are owned by the model: sr_model_mesh(model, i) stays valid until sr_model_destroy. */
SR_API int sr_model_load(sr_device dev, const char* path, sr_model* out);
SR_API uint32_t sr_model_mesh_count(sr_model model);
SR_API sr_mesh sr_model_mesh(sr_model model, uint32_t index);
SR_API int sr_model_destroy(sr_device dev, sr_model model);
/* the reading step alone (host only: dev may be NULL in sr_model_destroy): arrays as sr_mesh_create takes them */
SR_API int sr_model_read_file(const char* path, sr_model* out);
SR_API int sr_model_mesh_data(sr_model model, uint32_t index, const void** vertices, uint32_t* n_vertices,
                              const uint32_t** indices, uint32_t* n_indices);

/* ---- render target: replaces the swap-chain colour image, the multisampled colour and depth
 * attachments and the clear values (src/main.cpp:377-431, 723-738, 919-921).
 * width/height: frame size used by the viewport transform.  samples: 1 or 4 (the reference takes
 * the device maximum, src/vulkan_helper/gpu_device.cpp:16-30).  n_views: layers drawn by one
 * sr_draw.  rect: the part of the frame this target stores and rasterises (NULL = whole frame);
 * used to split one frame over several GPUs by screen region. */
SR_API int sr_target_create(sr_device dev, int width, int height, int samples, uint32_t n_views,
                            const sr_rect* rect, sr_target* out);
SR_API int sr_target_destroy(sr_device dev, sr_target target);
/* Render colour into caller-owned device memory (n_views * rect.height * rect.width * 4 bytes)
 * instead of the target's own buffer, e.g. a tensor handed to a collective.  NULL restores it.
 * A draw stores only the pixels whose colour changes: parts of the frame that were background in the last
 * draw and still are keep the clear colour they already hold.  A caller that writes into the bound memory itself
 * binds it again before the next draw (the draw after a bind stores every pixel). */
SR_API int sr_target_bind_color(sr_device dev, sr_target target, void* device_ptr);
/* No reference counterpart (the reference only presents): copy one view's result to the host. */
SR_API int sr_target_readback(sr_device dev, sr_target target, int kind, uint32_t view,
                              void* host, size_t bytes);

/* ---- draw: replaces update_uniform_buffer + the recorded render pass + vkQueueSubmit
 * (src/main.cpp:988-1004, 897-959, 1051): clear, vertex shader, triangle setup and binning,
 * rasterisation with depth test LESS, fragment shader, sRGB store, multisample resolve.
 * ubo_blocks: n_views uniform blocks of ubo_stride bytes each (>= 204; 208 for std140), laid out
 * as the reference's UniformBufferObject (src/main.cpp:50-55): model, view, proj column-major
 * mat4 at 0/64/128, camera_pos vec3 at 192.  Host memory; it is consumed before the call returns.
 * textures: albedo, normal, metallic, roughness in the binding order of src/main.cpp:461-516
 * (the normal map is bound but never sampled, shaders/shader.frag:35-52; it may be NULL).
 * Draws the whole index buffer as one instance (vkCmdDrawIndexed(indices_size,1,0,0,0),
 * src/main.cpp:954).  Asynchronous. */
SR_API int sr_draw(sr_device dev, sr_target target, sr_mesh mesh, const void* ubo_blocks,
                   uint32_t n_views, size_t ubo_stride, const sr_texture textures[4]);

/* ---- sr_draw for callers that cannot allocate.  No counterpart in the reference.  For callers that must not trigger
 * a device allocation while they draw: tight serving loops, harnesses that forbid hipMalloc, code that holds a HIP graph
 * capture open on ANOTHER stream (the kernels go to the library's own stream and the call reads the previous draw's counters on
 * the host, so the call itself cannot be captured into a graph).
 * Same arguments, same frame and same asynchrony as sr_draw, but every intermediate buffer of the draw (window-space
 * vertices, binning state, pending and reference lists, clip list, per-view transforms) lives in one array of
 * sr_no_alloc_scratch_bytes() = 64 MiB inside the library's device image -- each GPU the library runs on holds its own
 * copy until the process ends -- and the call issues kernels only: no hipMalloc, hipFree, hipHostMalloc, event creation or
 * copy from host memory (the transforms travel as kernel arguments).
 * Limits: at most 8 views per call; the draw must fit the array in one pass (about 1 M triangles for one view: 48 bytes
 * per triangle and 12 per vertex, per view), else SR_ERR_UNSUPPORTED with the sizes in sr_last_error_string(); a frame
 * whose tile reference list does not fit what is left of the array reports the same when its target is next read, and
 * so does a frame with more than 1024 triangles per call that cross the near plane or the guard band (the clip list of this
 * entry point holds 1024 entries where sr_draw's holds 65536): such a frame would be incomplete and is not handed over.
 * All calls on one GPU share that GPU's array, which still holds the previous call's contents (what must start at zero is
 * zeroed by the draw itself): calls on one device handle are ordered by its stream; calls through DIFFERENT device handles
 * of the same GPU that overlap in time race on the array and must not be made.
 * The target, mesh and textures are created beforehand with the ordinary entry points (which do allocate). */
SR_API int sr_draw_no_alloc(sr_device dev, sr_target target, sr_mesh mesh, const void* ubo_blocks,
                            uint32_t n_views, size_t ubo_stride, const sr_texture textures[4]);
SR_API size_t sr_no_alloc_scratch_bytes(void);

/* ---- normal mapping: getNormalFromMap of shaders/shader.frag:35-53 with the body the reference has commented out restored
 * (tangent frame from the screen-space derivatives of WorldPos and TexCoord, tangent-space normal from textures[1] through the
 * same sampler and sRGB image format as every other map, src/texture.cpp:209).  Off by default: the reference returns the
 * interpolated normal, and so does this library unless asked.  enabled != 0 makes every draw issued on the device afterwards
 * shade with the mapped normal (textures[1] is then required). */
SR_API int sr_device_set_normal_mapping(sr_device dev, int enabled);

/* ---- instrumentation (no reference counterpart) */
SR_API int sr_draw_get_stats(sr_device dev, sr_target target, sr_draw_stats* out);
/* Instrumentation, no reference counterpart: see sr_draw_counts.  Waits for the device's stream (the culling counters live in
 * device memory); draws that turn out to need a second issue are counted when they are resolved (read-back, stats, synchronize). */
SR_API int sr_device_get_draw_counts(sr_device dev, sr_draw_counts* out);
/* When enabled, sr_draw brackets every stage with HIP events on the device's stream. */
SR_API int sr_profile_enable(sr_device dev, int enabled);
SR_API int sr_profile_get(sr_device dev, sr_stage_times* out);
/* Per-vertex results of the vertex stage for one uniform block, for parity checks:
 * clip_out: n_vertices*4 floats (gl_Position); screen_out: n_vertices*16 bytes
 * {int32 x_fixed, int32 y_fixed, float depth, float inv_w}, inv_w == 0 for unprojectable vertices.
 * Either may be NULL. */
SR_API int sr_debug_vertex_stage(sr_device dev, sr_mesh mesh, const void* ubo_block, int width,
                                 int height, float* clip_out, void* screen_out);

#ifdef __cplusplus
}
#endif
#endif /* SOFTRASTER_H */
