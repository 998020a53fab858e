// Headless counterpart of the reference's Application (src/main.cpp:142-1076): same start-up order
// (device, textures, mesh, uniform buffer) and the same per-frame work (update_uniform_buffer, draw),
// with the window, swap chain and presentation replaced by a read-back into a PPM file.
//
//   headless_app [width height [x_segments y_segments [out.ppm]]] [--model file.gltf|.glb] [--textures albedo.png normal.png metallic.png roughness.png] [--keys WASD...]
// Without --model the reference's generated UV sphere is drawn (src/main.cpp:94-115), without --textures
// procedural maps; with them the files go through Model::load (src/main.cpp:289) and
// create_texture_from_file (src/main.cpp:283-288) and every mesh of the model is drawn.
#include "renderer_host.hpp"

#include <chrono>
#include <cstring>
#include <string>

namespace {

constexpr int init_width = 800;    // src/main.cpp:45-46
constexpr int init_height = 600;

// The UV sphere of the reference's generate_uv_sphere (src/main.cpp:94-115), indexed as a triangle list.
StaticMesh generate_uv_sphere(vkh::GPUDevice& device, std::uint32_t x_segments, std::uint32_t y_segments)
{
    std::vector<Vertex> vertices;
    std::vector<std::uint32_t> indices;
    const float pi = 3.14159265358979323846f;
    for (std::uint32_t y = 0; y <= y_segments; ++y)
        for (std::uint32_t x = 0; x <= x_segments; ++x) {
            const float xs = static_cast<float>(x) / static_cast<float>(x_segments), ys = static_cast<float>(y) / static_cast<float>(y_segments);
            const float theta = ys * pi, phi = xs * 2.0f * pi;
            const vec3 p{std::cos(phi) * std::sin(theta), std::sin(phi) * std::sin(theta), std::cos(theta)};
            vertices.push_back({p, p, {xs, ys}});
        }
    for (std::uint32_t y = 0; y < y_segments; ++y)
        for (std::uint32_t x = 0; x < x_segments; ++x) {
            const std::uint32_t i00 = y * (x_segments + 1) + x, i10 = i00 + 1, i01 = i00 + x_segments + 1, i11 = i01 + 1;
            for (std::uint32_t i : {i00, i01, i11, i00, i11, i10}) indices.push_back(i);
        }
    return create_mesh_from_data(device, vertices, indices);
}

Texture procedural_texture(vkh::GPUDevice& device, int size, int kind)
{
    std::vector<std::uint8_t> px(static_cast<std::size_t>(size) * size * 4);
    for (int y = 0; y < size; ++y)
        for (int x = 0; x < size; ++x) {
            std::uint8_t* p = &px[(static_cast<std::size_t>(y) * size + x) * 4];
            if (kind == 0) { p[0] = (x ^ y) & 255; p[1] = (x * 3 + y * 5) & 255; p[2] = (((x >> 4) ^ (y >> 4)) * 17) & 255; }
            else if (kind == 1) { p[0] = 128; p[1] = 128; p[2] = 255; }
            else { const int lo = kind == 2 ? 0 : 72, hi = kind == 2 ? 254 : 255; p[0] = p[1] = p[2] = static_cast<std::uint8_t>(lo + (x + 2 * y) % (hi - lo + 1)); }
            p[3] = 255;
        }
    return create_texture(device, px.data(), size, size);
}

void write_ppm(const std::string& path, const std::vector<std::uint8_t>& bgra, int width, int height)
{
    if (FILE* f = std::fopen(path.c_str(), "wb")) {
        std::fprintf(f, "P6\n%d %d\n255\n", width, height);
        for (std::size_t i = 0; i < bgra.size(); i += 4) { const std::uint8_t rgb[3] = {bgra[i + 2], bgra[i + 1], bgra[i]}; std::fwrite(rgb, 1, 3, f); }
        std::fclose(f);
    }
}

}  // namespace

int main(int argc, char** argv)
{
    int npos = 1;   // positional arguments end at the first option
    while (npos < argc && std::strncmp(argv[npos], "--", 2) != 0) ++npos;
    const int width = npos > 2 ? std::atoi(argv[1]) : init_width, height = npos > 2 ? std::atoi(argv[2]) : init_height;
    const std::uint32_t xs = npos > 4 ? std::atoi(argv[3]) : 88, ys = npos > 4 ? std::atoi(argv[4]) : 88;
    const std::string out = npos > 5 ? argv[5] : "frame.ppm";

    const char* model_path = nullptr;
    const char* texture_paths[4] = {nullptr, nullptr, nullptr, nullptr};
    const char* keys = "";
    for (int a = 1; a < argc; ++a) {
        if (!std::strcmp(argv[a], "--model") && a + 1 < argc) model_path = argv[a + 1];
        if (!std::strcmp(argv[a], "--textures") && a + 4 < argc) for (int k = 0; k < 4; ++k) texture_paths[k] = argv[a + 1 + k];
        if (!std::strcmp(argv[a], "--keys") && a + 1 < argc) keys = argv[a + 1];
    }
    OrbitCamera camera;                                      // src/main.cpp:210-211
    for (const char* k = keys; *k; ++k) camera.key(*k);      // what key_callback does on each press
    if (*keys) std::printf("camera %.9g %.9g %.9g up %.9g %.9g %.9g\n", camera.camera_pos.x, camera.camera_pos.y, camera.camera_pos.z, camera.up.x, camera.up.y, camera.up.z);

    vkh::GPUDevice device;                                   // Application(): device_{window_}
    // init_vulkan(): four textures, then the model
    Texture albedo = texture_paths[0] ? create_texture_from_file(device, texture_paths[0]) : procedural_texture(device, 2048, 0);
    Texture normal = texture_paths[1] ? create_texture_from_file(device, texture_paths[1]) : procedural_texture(device, 64, 1);
    Texture metallic = texture_paths[2] ? create_texture_from_file(device, texture_paths[2]) : procedural_texture(device, 2048, 2);
    Texture roughness = texture_paths[3] ? create_texture_from_file(device, texture_paths[3]) : procedural_texture(device, 2048, 3);
    RenderTarget target(device, width, height);
    if (model_path) {
        // The reference loads the model (src/main.cpp:289) but records the draw of its generated sphere; drawing the
        // model's meshes is what a user of Model::load wants to see.  One frame per mesh, last one kept.
        Model model = Model::load(device, model_path);
        const UniformBufferObject view = camera.ubo(width, height);
        for (const Model::Mesh& m : model.meshes()) {
            target.draw(m, view, albedo, normal, metallic, roughness);
            const sr_draw_stats st = target.stats();
            std::printf("mesh with %u indices: %llu shaded pixels, %llu visible triangles\n", m.indices_size,
                        static_cast<unsigned long long>(st.shaded_pixels), static_cast<unsigned long long>(st.visible_triangles));
        }
        write_ppm(out, target.read_color_bgra(), width, height);
        destroy_texture(device, albedo); destroy_texture(device, normal); destroy_texture(device, metallic); destroy_texture(device, roughness);
        return 0;
    }
    StaticMesh mesh = generate_uv_sphere(device, xs, ys);

    const UniformBufferObject ubo = camera.ubo(width, height);        // update_uniform_buffer, src/main.cpp:988-1004

    const int frames = 20;
    target.draw(mesh, ubo, albedo, normal, metallic, roughness);      // warm-up: sizes the per-tile lists
    device.wait_idle();
    const auto t0 = std::chrono::steady_clock::now();
    for (int f = 0; f < frames; ++f) target.draw(mesh, ubo, albedo, normal, metallic, roughness);
    device.wait_idle();
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / frames;

    const sr_draw_stats st = target.stats();
    std::printf("%dx%d, %llu triangles: %.3f ms/frame, %llu shaded pixels, %llu visible triangles\n", width, height,
                static_cast<unsigned long long>(st.triangles), ms, static_cast<unsigned long long>(st.shaded_pixels),
                static_cast<unsigned long long>(st.visible_triangles));
    write_ppm(out, target.read_color_bgra(), width, height);
    destroy_texture(device, albedo); destroy_texture(device, normal); destroy_texture(device, metallic); destroy_texture(device, roughness);
    return 0;
}
