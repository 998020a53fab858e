// Host-side asset readers: PNG (sr_assets.cpp) and JPEG (sr_jpeg.cpp) -> RGBA8 the way
// stbi_load(..., STBI_rgb_alpha) expands them, glTF 2.0 (sr_assets.cpp) -> one vertex/index array pair per
// primitive in node order.
#pragma once

#include <stdint.h>

#include <string>
#include <vector>

namespace sr_assets {

struct Image {
    int width = 0, height = 0;
    std::vector<uint8_t> rgba;   // width * height * 4
};

struct HostMesh {
    std::vector<float> vertices;     // 8 floats per vertex: position, normal, uv (v flipped) -- the reference's Vertex, src/mesh.hpp:18-54
    std::vector<uint32_t> indices;   // triangle list
};

// which decoder's integer arithmetic the steps after entropy decoding follow (they differ in the last bit)
enum class JpegFlavor { Stb, LibJpeg };

bool decode_png(const uint8_t* data, size_t size, Image& img, std::string& err);
bool decode_jpeg(const uint8_t* data, size_t size, Image& img, std::string& err, JpegFlavor flavor = JpegFlavor::Stb);
// PNG or JPEG by the file's first bytes; the environment variable SR_JPEG_DECODER=libjpeg selects JpegFlavor::LibJpeg
bool load_image(const std::string& path, Image& img, std::string& err);
bool load_model(const std::string& path, std::vector<HostMesh>& meshes, std::string& err);

}  // namespace sr_assets
