// C++ host API of the draw path, shaped like the reference's so that its application code ports by
// changing includes: vkh::GPUDevice (src/vulkan_helper/gpu_device.hpp:23-110), Vertex / StaticMesh /
// create_mesh_from_data (src/mesh.hpp:18-72), Texture / create_texture / destroy_texture
// (src/texture.hpp:9-24), UniformBufferObject (src/main.cpp:50-55) and the GLM camera helpers the
// reference calls (src/main.cpp:992-997).  Everything forwards to the C ABI in include/softraster.h;
// failures follow the reference's application-level convention (VKH_CHECK, src/vulkan_helper/check.hpp:8-15):
// print file:line and abort.
#pragma once

#include <softraster.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <utility>
#include <vector>

#define SRH_CHECK(expr)                                                                                   \
    do {                                                                                                  \
        const int srh_status_ = (expr);                                                                   \
        if (srh_status_ != SR_OK) {                                                                       \
            std::fprintf(stderr, "[%s:%d] %s failed: %s (%s)\n", __FILE__, __LINE__, #expr,               \
                         sr_status_string(srh_status_), sr_last_error_string());                          \
            std::abort();                                                                                 \
        }                                                                                                 \
    } while (0)

// ---- the small part of GLM the reference uses -----------------------------------------------------
struct vec2 { float x, y; };
struct vec3 { float x, y, z; };
struct mat4 { float m[16]; };   // column-major, like glm::mat4

inline vec3 operator-(vec3 a, vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline float dot(vec3 a, vec3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
inline vec3 cross(vec3 a, vec3 b) { return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y}; }
inline vec3 normalize(vec3 a) { const float i = 1.0f / std::sqrt(dot(a, a)); return {a.x * i, a.y * i, a.z * i}; }
inline float radians(float deg) { return deg * 0.01745329251994329576923690768489f; }
inline mat4 identity() { mat4 r{}; r.m[0] = r.m[5] = r.m[10] = r.m[15] = 1.0f; return r; }

// glm::lookAt, right-handed
inline mat4 look_at(vec3 eye, vec3 center, vec3 up)
{
    const vec3 f = normalize(center - eye), s = normalize(cross(f, up)), u = cross(s, f);
    mat4 r = identity();
    r.m[0] = s.x; r.m[4] = s.y; r.m[8] = s.z; r.m[12] = -dot(s, eye);
    r.m[1] = u.x; r.m[5] = u.y; r.m[9] = u.z; r.m[13] = -dot(u, eye);
    r.m[2] = -f.x; r.m[6] = -f.y; r.m[10] = -f.z; r.m[14] = dot(f, eye);
    return r;
}
// glm::perspective with GLM_FORCE_DEPTH_ZERO_TO_ONE, right-handed (src/main.cpp:4-5)
inline mat4 perspective(float fovy, float aspect, float z_near, float z_far)
{
    const float t = std::tan(fovy / 2.0f);
    mat4 r{};
    r.m[0] = 1.0f / (aspect * t);
    r.m[5] = 1.0f / t;
    r.m[10] = z_far / (z_near - z_far);
    r.m[11] = -1.0f;
    r.m[14] = -(z_far * z_near) / (z_far - z_near);
    return r;
}

// glm::rotate(vec3, angle, normal) of glm/gtx/rotate_vector.hpp: mat3(glm::rotate(angle, normal)) * v
inline vec3 rotate(vec3 v, float angle, vec3 normal)
{
    const float c = std::cos(angle), s = std::sin(angle);
    const vec3 a = normalize(normal), t{(1.0f - c) * a.x, (1.0f - c) * a.y, (1.0f - c) * a.z};
    const float r00 = c + t.x * a.x, r01 = t.x * a.y + s * a.z, r02 = t.x * a.z - s * a.y;   // first column
    const float r10 = t.y * a.x - s * a.z, r11 = c + t.y * a.y, r12 = t.y * a.z + s * a.x;
    const float r20 = t.z * a.x + s * a.y, r21 = t.z * a.y - s * a.x, r22 = c + t.z * a.z;
    return {(r00 * v.x + r10 * v.y) + r20 * v.z, (r01 * v.x + r11 * v.y) + r21 * v.z, (r02 * v.x + r12 * v.y) + r22 * v.z};
}

struct UniformBufferObject {   // 204 bytes, memcpy'd per frame by the reference (src/main.cpp:1002)
    mat4 model;
    mat4 view;
    mat4 proj;
    vec3 camera_pos;
};
static_assert(sizeof(UniformBufferObject) == 204, "layout of src/main.cpp:50-55");

// The camera of the reference's Application: eye and up vector (src/main.cpp:210-211), moved by the W/S/A/D keys
// in steps of 0.1 rad (key_callback, src/main.cpp:223-267), turned into the per-frame block by
// update_uniform_buffer (src/main.cpp:988-1004).
struct OrbitCamera {
    vec3 camera_pos{2.0f, 2.0f, 2.0f};
    vec3 up{0.0f, 0.0f, 1.0f};

    void key(char k)   // GLFW_PRESS or GLFW_REPEAT of that key; other keys do nothing
    {
        const float step = 0.1f;
        if (k == 'W' || k == 'S') {            // over the top: eye and up turn about the axis perpendicular to both
            const vec3 axis = normalize(cross(camera_pos, up));
            const float angle = k == 'W' ? step : -step;
            camera_pos = rotate(camera_pos, angle, axis);
            up = rotate(up, angle, axis);
        } else if (k == 'A' || k == 'D') camera_pos = rotate(camera_pos, k == 'A' ? step : -step, up);
    }
    UniformBufferObject ubo(int width, int height) const
    {
        return {identity(), look_at(camera_pos, {0.0f, 0.0f, 0.0f}, up),
                perspective(radians(45.0f), static_cast<float>(width) / static_cast<float>(height), 0.1f, 10.0f), camera_pos};
    }
};

struct Vertex {                // 32 bytes: the vertex input layout of src/mesh.hpp:18-54
    vec3 pos;
    vec3 normal;
    vec2 tex_coord;
};
static_assert(sizeof(Vertex) == 32, "stride of src/mesh.hpp:28");

namespace vkh {

class GPUDevice {
public:
    explicit GPUDevice(int hip_ordinal = 0) { SRH_CHECK(sr_device_create(hip_ordinal, &device_)); }
    ~GPUDevice() noexcept { sr_device_destroy(device_); }
    GPUDevice(const GPUDevice&) = delete;
    GPUDevice& operator=(const GPUDevice&) = delete;
    GPUDevice(GPUDevice&&) = delete;
    GPUDevice& operator=(GPUDevice&&) = delete;
    void wait_idle() noexcept { sr_device_synchronize(device_); }
    // shade with the normal map (shaders/shader.frag:35-53 with its commented-out body restored); off by default, as in the reference
    void set_normal_mapping(bool enabled) noexcept { sr_device_set_normal_mapping(device_, enabled ? 1 : 0); }
    [[nodiscard]] sr_device device() const noexcept { return device_; }

private:
    sr_device device_ = nullptr;
};

}  // namespace vkh

class StaticMesh {             // move-only owner, like the reference's UniqueBuffer pair
public:
    StaticMesh() = default;
    StaticMesh(sr_device d, sr_mesh m, std::uint32_t n) : indices_size(n), device_(d), mesh_(m) {}
    ~StaticMesh() { if (mesh_) sr_mesh_destroy(device_, mesh_); }
    StaticMesh(StaticMesh&& o) noexcept : indices_size(o.indices_size), device_(o.device_), mesh_(std::exchange(o.mesh_, nullptr)) {}
    StaticMesh& operator=(StaticMesh&& o) noexcept
    {
        if (this != &o) { if (mesh_) sr_mesh_destroy(device_, mesh_); device_ = o.device_; mesh_ = std::exchange(o.mesh_, nullptr); indices_size = o.indices_size; }
        return *this;
    }
    StaticMesh(const StaticMesh&) = delete;
    StaticMesh& operator=(const StaticMesh&) = delete;
    [[nodiscard]] sr_mesh handle() const noexcept { return mesh_; }
    std::uint32_t indices_size = 0;

private:
    sr_device device_ = nullptr;
    sr_mesh mesh_ = nullptr;
};

[[nodiscard]] inline StaticMesh create_mesh_from_data(vkh::GPUDevice& device, const std::vector<Vertex>& vertices,
                                                      const std::vector<std::uint32_t>& indices)
{
    sr_mesh m = nullptr;
    SRH_CHECK(sr_mesh_create(device.device(), vertices.data(), static_cast<std::uint32_t>(vertices.size()), indices.data(),
                             static_cast<std::uint32_t>(indices.size()), &m));
    return StaticMesh{device.device(), m, static_cast<std::uint32_t>(indices.size())};
}

struct Texture {               // plain struct with manual destroy, as in src/texture.hpp:9-24
    std::uint32_t mip_levels{};
    sr_texture image{};
};

[[nodiscard]] inline Texture create_texture(vkh::GPUDevice& device, void* data, int tex_width, int tex_height)
{
    Texture t;
    SRH_CHECK(sr_texture_create_rgba8_srgb(device.device(), data, tex_width, tex_height, nullptr, &t.image));
    t.mip_levels = static_cast<std::uint32_t>(sr_texture_mip_levels(t.image));
    return t;
}

// src/texture.hpp:20 / src/texture.cpp:273-291 (PNG and JPEG files; stb's STBI_rgb_alpha expansion)
[[nodiscard]] inline Texture create_texture_from_file(vkh::GPUDevice& device, const char* texture_path)
{
    Texture t;
    SRH_CHECK(sr_texture_create_from_file(device.device(), texture_path, nullptr, &t.image));
    t.mip_levels = static_cast<std::uint32_t>(sr_texture_mip_levels(t.image));
    return t;
}

inline void destroy_texture(vkh::GPUDevice& device, Texture& texture)
{
    sr_texture_destroy(device.device(), texture.image);
    texture.image = nullptr;
}

// src/model.hpp:9-26: the meshes of a glTF file, one per primitive in node order.  The library owns the
// uploaded meshes; Mesh is the non-owning view the draw call takes.
class Model {
public:
    struct Mesh {
        sr_mesh mesh = nullptr;
        std::uint32_t indices_size = 0;
        [[nodiscard]] sr_mesh handle() const noexcept { return mesh; }
    };
    Model() = default;
    ~Model() { if (model_) sr_model_destroy(device_, model_); }
    Model(Model&& o) noexcept : device_(o.device_), model_(std::exchange(o.model_, nullptr)), meshes_(std::move(o.meshes_)) {}
    Model& operator=(Model&& o) noexcept
    {
        if (this != &o) { if (model_) sr_model_destroy(device_, model_); device_ = o.device_; model_ = std::exchange(o.model_, nullptr); meshes_ = std::move(o.meshes_); }
        return *this;
    }
    Model(const Model&) = delete;
    Model& operator=(const Model&) = delete;
    [[nodiscard]] const std::vector<Mesh>& meshes() const { return meshes_; }
    [[nodiscard]] static Model load(vkh::GPUDevice& device, const char* path)
    {
        Model m;
        m.device_ = device.device();
        SRH_CHECK(sr_model_load(device.device(), path, &m.model_));
        for (std::uint32_t i = 0; i < sr_model_mesh_count(m.model_); ++i) {
            std::uint32_t n_indices = 0;
            SRH_CHECK(sr_model_mesh_data(m.model_, i, nullptr, nullptr, nullptr, &n_indices));
            m.meshes_.push_back({sr_model_mesh(m.model_, i), n_indices});
        }
        return m;
    }

private:
    sr_device device_ = nullptr;
    sr_model model_ = nullptr;
    std::vector<Mesh> meshes_;
};

// Stands for the swap-chain image plus the multisampled colour and depth attachments
// (src/main.cpp:723-738) and for draw_frame's submit (src/main.cpp:1006-1076).
class RenderTarget {
public:
    RenderTarget(vkh::GPUDevice& device, int width, int height, int samples = 1, std::uint32_t n_views = 1)
        : device_(device.device()), width_(width), height_(height)
    {
        SRH_CHECK(sr_target_create(device_, width, height, samples, n_views, nullptr, &target_));
    }
    ~RenderTarget() { sr_target_destroy(device_, target_); }
    RenderTarget(const RenderTarget&) = delete;
    RenderTarget& operator=(const RenderTarget&) = delete;

    // albedo, normal, metallic, roughness: the descriptor bindings 1-4 of src/main.cpp:461-516
    template <class MeshT>   // StaticMesh or Model::Mesh
    void draw(const MeshT& mesh, const UniformBufferObject& ubo, const Texture& albedo, const Texture& normal,
              const Texture& metallic, const Texture& roughness)
    {
        const sr_texture tex[4] = {albedo.image, normal.image, metallic.image, roughness.image};
        SRH_CHECK(sr_draw(device_, target_, mesh.handle(), &ubo, 1, sizeof(UniformBufferObject), tex));
    }
    [[nodiscard]] std::vector<std::uint8_t> read_color_bgra()
    {
        std::vector<std::uint8_t> out(static_cast<std::size_t>(width_) * height_ * 4);
        SRH_CHECK(sr_target_readback(device_, target_, SR_READ_COLOR_BGRA8, 0, out.data(), out.size()));
        return out;
    }
    [[nodiscard]] sr_draw_stats stats()
    {
        sr_draw_stats s{};
        SRH_CHECK(sr_draw_get_stats(device_, target_, &s));
        return s;
    }

private:
    sr_device device_;
    sr_target target_ = nullptr;
    int width_, height_;
};
