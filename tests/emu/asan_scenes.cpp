// TEST INFRASTRUCTURE ONLY.  Drives the C ABI through a set of scenes; linked together with the kernel
// sources compiled for the host (tests/emu) and AddressSanitizer.  The guard pages of the emulation
// catch out-of-range accesses to device allocations; this build adds the workgroup-local arrays
// (__shared__ becomes static, which AddressSanitizer fences like any global) and the host side of the
// library.  Results are not checked here (the parity tests do that); the run must simply be clean.
#include "../../include/softraster.h"

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace {

struct Rng {
    uint64_t s;
    double next() { s = s * 6364136223846793005ull + 1442695040888963407ull; return (double)(s >> 11) / 9007199254740992.0; }
    float range(float lo, float hi) { return lo + (float)next() * (hi - lo); }
};

#define CHECK(expr)                                                                                              \
    do {                                                                                                         \
        const int rc__ = (expr);                                                                                 \
        if (rc__ != SR_OK) { std::printf("%s -> %d (%s)\n", #expr, rc__, sr_last_error_string()); std::exit(1); } \
    } while (0)

void look_at_ubo(float* u, int W, int H, float ex, float ey, float ez)
{
    std::memset(u, 0, 52 * 4);
    u[0] = u[5] = u[10] = u[15] = 1.0f;   // model
    // view: eye looking at the origin, +z up
    float f[3] = {-ex, -ey, -ez};
    const float fl = std::sqrt(f[0] * f[0] + f[1] * f[1] + f[2] * f[2]);
    for (float& v : f) v /= fl;
    float s[3] = {f[1] * 1.0f - f[2] * 0.0f, f[2] * 0.0f - f[0] * 1.0f, 0.0f};   // f x up
    const float sl = std::sqrt(s[0] * s[0] + s[1] * s[1] + s[2] * s[2]);
    for (float& v : s) v /= sl;
    const float t[3] = {s[1] * f[2] - s[2] * f[1], s[2] * f[0] - s[0] * f[2], s[0] * f[1] - s[1] * f[0]};   // s x f
    float* V = u + 16;
    V[0] = s[0]; V[4] = s[1]; V[8] = s[2]; V[12] = -(s[0] * ex + s[1] * ey + s[2] * ez);
    V[1] = t[0]; V[5] = t[1]; V[9] = t[2]; V[13] = -(t[0] * ex + t[1] * ey + t[2] * ez);
    V[2] = -f[0]; V[6] = -f[1]; V[10] = -f[2]; V[14] = f[0] * ex + f[1] * ey + f[2] * ez;
    V[15] = 1.0f;
    float* P = u + 32;
    const float th = std::tan(0.5f * 0.7853982f), zn = 0.1f, zf = 10.0f;
    P[0] = 1.0f / ((float)W / (float)H * th); P[5] = 1.0f / th; P[10] = zf / (zn - zf); P[11] = -1.0f; P[14] = -(zf * zn) / (zf - zn);
    u[48] = ex; u[49] = ey; u[50] = ez;
}

void sphere(int X, int Y, std::vector<float>& v, std::vector<uint32_t>& idx)
{
    for (int y = 0; y <= Y; ++y)
        for (int x = 0; x <= X; ++x) {
            const float th = 3.14159265f * y / Y, ph = 6.2831853f * x / X;
            const float p[3] = {std::cos(ph) * std::sin(th), std::sin(ph) * std::sin(th), std::cos(th)};
            v.insert(v.end(), {p[0], p[1], p[2], p[0], p[1], p[2], (float)x / X, (float)y / Y});
        }
    for (int y = 0; y < Y; ++y)
        for (int x = 0; x < X; ++x) {
            const uint32_t a = y * (X + 1) + x, b = a + 1, c = a + X + 1, d = c + 1;
            idx.insert(idx.end(), {a, c, d, a, d, b});
        }
}

void soup(Rng& r, int n, float size, std::vector<float>& v, std::vector<uint32_t>& idx)
{
    for (int t = 0; t < n; ++t) {
        const float c[3] = {r.range(-1.2f, 1.2f), r.range(-1.2f, 1.2f), r.range(-1.2f, 1.2f)};
        const float sc = size * (float)std::pow(r.next(), 3.0);
        for (int k = 0; k < 3; ++k) {
            v.insert(v.end(), {c[0] + r.range(-1, 1) * sc, c[1] + r.range(-1, 1) * sc, c[2] + r.range(-1, 1) * sc, 0.0f, 0.0f, 1.0f, r.range(-2, 3), r.range(-2, 3)});
            idx.push_back((uint32_t)idx.size());
        }
    }
}

}  // namespace

int main()
{
    sr_device dev = nullptr;
    CHECK(sr_device_create(0, &dev));
    Rng rng{12345};
    sr_texture tex[4];
    const int sizes[4] = {64, 4, 33, 17};   // power of two, tiny, and two sizes that are not powers of two
    for (int k = 0; k < 4; ++k) {
        std::vector<uint8_t> px((size_t)sizes[k] * sizes[k] * 4);
        for (uint8_t& b : px) b = (uint8_t)(rng.next() * 256.0);
        CHECK(sr_texture_create_rgba8_srgb(dev, px.data(), sizes[k], sizes[k], nullptr, &tex[k]));
    }
    struct Scene { std::vector<float> v; std::vector<uint32_t> idx; sr_mesh mesh = nullptr; } scenes[4];
    sphere(40, 30, scenes[0].v, scenes[0].idx);              // ordinary triangles
    sphere(300, 200, scenes[1].v, scenes[1].idx);            // sub-pixel triangles: the direct path
    soup(rng, 600, 1.5f, scenes[2].v, scenes[2].idx);        // a few huge ones among many small ones: tile lists, wave-cooperative rasterisation
    soup(rng, 3000, 0.05f, scenes[3].v, scenes[3].idx);
    for (Scene& s : scenes) CHECK(sr_mesh_create(dev, s.v.data(), (uint32_t)(s.v.size() / 8), s.idx.data(), (uint32_t)s.idx.size(), &s.mesh));

    struct Frame { int W, H, samples; uint32_t views; sr_rect rect; bool use_rect; };
    const Frame frames[] = {{200, 150, 1, 1, {}, false}, {130, 97, 4, 2, {}, false}, {256, 192, 1, 3, {33, 17, 150, 101}, true}, {64, 64, 4, 1, {1, 1, 62, 62}, true}};
    int draws = 0;
    for (const Frame& f : frames) {
        sr_target tgt = nullptr;
        CHECK(sr_target_create(dev, f.W, f.H, f.samples, f.views, f.use_rect ? &f.rect : nullptr, &tgt));
        const int rw = f.use_rect ? f.rect.width : f.W, rh = f.use_rect ? f.rect.height : f.H;
        std::vector<uint8_t> color((size_t)rw * rh * 4);
        std::vector<uint64_t> vis((size_t)rw * rh * f.samples);
        for (int repeat = 0; repeat < 2; ++repeat)   // twice: the second frame of a target reuses every scratch buffer
            for (Scene& s : scenes) {
                std::vector<float> ubo(52 * f.views);
                for (uint32_t v = 0; v < f.views; ++v) look_at_ubo(&ubo[52 * v], f.W, f.H, 2.0f + 0.7f * v, 2.0f - 0.9f * v, 1.0f + v);
                // the last view of a multi-view frame from inside the scene: triangles behind and across the eye plane
                // (cut at the near plane by the clipper)
                if (f.views > 1) look_at_ubo(&ubo[52 * (f.views - 1)], f.W, f.H, 0.3f, 0.2f, 0.1f);
                CHECK(sr_draw(dev, tgt, s.mesh, ubo.data(), f.views, 208, tex));
                for (uint32_t v = 0; v < f.views; ++v) {
                    CHECK(sr_target_readback(dev, tgt, SR_READ_COLOR_BGRA8, v, color.data(), color.size()));
                    CHECK(sr_target_readback(dev, tgt, SR_READ_VIS_U64, v, vis.data(), vis.size() * 8));
                }
                sr_draw_stats st;
                CHECK(sr_draw_get_stats(dev, tgt, &st));
                ++draws;
            }
        CHECK(sr_target_destroy(dev, tgt));
    }
    for (Scene& s : scenes) CHECK(sr_mesh_destroy(dev, s.mesh));
    for (sr_texture t : tex) CHECK(sr_texture_destroy(dev, t));
    CHECK(sr_device_destroy(dev));
    std::printf("ok: %d draws\n", draws);
    return 0;
}
