// Headless counterpart of the reference's Application (src/main.cpp:142-1076): same start-up order
// (device, textures, mesh, uniform buffer) and the same per-frame work (update_uniform_buffer, draw),
// with the window, swap chain and presentation replaced by a read-back into a PPM file.
//
//   headless_app [width height [x_segments y_segments [out.ppm]]]
#include "renderer_host.hpp"

#include <chrono>
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

}  // namespace

int main(int argc, char** argv)
{
    const int width = argc > 2 ? std::atoi(argv[1]) : init_width, height = argc > 2 ? std::atoi(argv[2]) : init_height;
    const std::uint32_t xs = argc > 4 ? std::atoi(argv[3]) : 88, ys = argc > 4 ? std::atoi(argv[4]) : 88;
    const std::string out = argc > 5 ? argv[5] : "frame.ppm";

    vkh::GPUDevice device;                                   // Application(): device_{window_}
    Texture albedo = procedural_texture(device, 2048, 0);    // init_vulkan(): four textures, then the model
    Texture normal = procedural_texture(device, 64, 1);
    Texture metallic = procedural_texture(device, 2048, 2);
    Texture roughness = procedural_texture(device, 2048, 3);
    StaticMesh mesh = generate_uv_sphere(device, xs, ys);
    RenderTarget target(device, width, height);

    const vec3 camera_pos{2.0f, 2.0f, 2.0f}, up{0.0f, 0.0f, 1.0f};   // src/main.cpp:210-211
    const UniformBufferObject ubo = {                                 // update_uniform_buffer, src/main.cpp:988-1004
        identity(), look_at(camera_pos, {0.0f, 0.0f, 0.0f}, up),
        perspective(radians(45.0f), static_cast<float>(width) / static_cast<float>(height), 0.1f, 10.0f), camera_pos};

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
    const std::vector<std::uint8_t> bgra = target.read_color_bgra();
    if (FILE* f = std::fopen(out.c_str(), "wb")) {
        std::fprintf(f, "P6\n%d %d\n255\n", width, height);
        for (std::size_t i = 0; i < bgra.size(); i += 4) { const std::uint8_t rgb[3] = {bgra[i + 2], bgra[i + 1], bgra[i]}; std::fwrite(rgb, 1, 3, f); }
        std::fclose(f);
    }
    destroy_texture(device, albedo); destroy_texture(device, normal); destroy_texture(device, metallic); destroy_texture(device, roughness);
    return 0;
}
