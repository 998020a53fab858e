// Host-side asset readers behind the two file-taking calls of the reference's host API:
//   create_texture_from_file (src/texture.cpp:273-291): stbi_load(path, ..., STBI_rgb_alpha) -> create_texture
//   Model::load              (src/model.cpp:81-98):     Assimp ReadFile(Triangulate | FlipUVs), one StaticMesh per
//                                                        mesh found by a depth-first walk of the nodes
// Neither stb nor Assimp exists here, so this file reads the formats the reference's data directory
// uses itself: PNG (every colour type and bit depth, Adam7 interlacing), JPEG (sr_jpeg.cpp) and glTF 2.0 (.gltf
// with data: or file buffers, .glb).  Pure CPU code; nothing here touches the GPU.
#include "sr_assets.h"

#include <zlib.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>

namespace sr_assets {

namespace {

bool read_file(const std::string& path, std::vector<uint8_t>& out, std::string& err)
{
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) { err = "cannot open " + path; return false; }
    std::fseek(f, 0, SEEK_END);
    const long n = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    out.resize(n > 0 ? (size_t)n : 0);
    const size_t got = out.empty() ? 0 : std::fread(out.data(), 1, out.size(), f);
    std::fclose(f);
    if (got != out.size()) { err = "short read on " + path; return false; }
    return true;
}

uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

// ---- PNG ------------------------------------------------------------------------------------------
int paeth(int a, int b, int c)
{
    const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

}  // namespace

bool decode_png(const uint8_t* data, size_t size, Image& img, std::string& err)
{
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    if (size < 8 || std::memcmp(data, sig, 8) != 0) { err = "not a PNG file"; return false; }
    uint32_t w = 0, h = 0;
    int depth = 0, ctype = -1, interlace = 0;
    std::vector<uint8_t> idat, plte, trns;
    size_t pos = 8;
    bool end = false;
    while (!end && pos + 12 <= size) {
        const uint32_t len = be32(data + pos);
        const uint8_t* type = data + pos + 4;
        const uint8_t* body = data + pos + 8;
        if (pos + 12 + (size_t)len > size) { err = "truncated PNG chunk"; return false; }
        if (be32(body + len) != (uint32_t)crc32(crc32(0L, type, 4), body, len)) { err = "PNG chunk checksum mismatch"; return false; }
        if (!std::memcmp(type, "IHDR", 4)) {
            if (len != 13) { err = "bad IHDR"; return false; }
            w = be32(body); h = be32(body + 4); depth = body[8]; ctype = body[9]; interlace = body[12];
            if (body[10] != 0 || body[11] != 0) { err = "unknown PNG compression or filter method"; return false; }
        } else if (!std::memcmp(type, "PLTE", 4)) plte.assign(body, body + len);
        else if (!std::memcmp(type, "tRNS", 4)) trns.assign(body, body + len);
        else if (!std::memcmp(type, "IDAT", 4)) idat.insert(idat.end(), body, body + len);
        else if (!std::memcmp(type, "IEND", 4)) end = true;
        pos += 12 + (size_t)len;
    }
    if (ctype < 0 || !end) { err = "PNG without IHDR or IEND"; return false; }
    if (w == 0 || h == 0 || w > 32768 || h > 32768) { err = "PNG size out of range"; return false; }
    if (interlace > 1) { err = "unknown PNG interlace method"; return false; }
    int channels;
    switch (ctype) {
    case 0: channels = 1; break;
    case 2: channels = 3; break;
    case 3: channels = 1; break;
    case 4: channels = 2; break;
    case 6: channels = 4; break;
    default: err = "unknown PNG colour type"; return false;
    }
    const bool depth_ok = ctype == 0 ? (depth == 1 || depth == 2 || depth == 4 || depth == 8 || depth == 16)
                        : ctype == 3 ? (depth == 1 || depth == 2 || depth == 4 || depth == 8)
                                     : (depth == 8 || depth == 16);
    if (!depth_ok) { err = "bit depth not allowed for this PNG colour type"; return false; }
    if (ctype == 3 && plte.size() < 3) { err = "palette PNG without PLTE"; return false; }
    const size_t bits_pp = (size_t)channels * depth, bpp = std::max<size_t>(bits_pp / 8, 1);
    // the image is stored as one pass, or as the seven passes of Adam7 interlacing: sub-images of every 8th, 4th, 2nd
    // pixel, each with its own filtered scan lines
    struct Pass { uint32_t x0, y0, dx, dy; };
    static const Pass adam7[7] = {{0, 0, 8, 8}, {4, 0, 8, 8}, {0, 4, 4, 8}, {2, 0, 4, 4}, {0, 2, 2, 4}, {1, 0, 2, 2}, {0, 1, 1, 2}};
    static const Pass whole = {0, 0, 1, 1};
    const int n_passes = interlace ? 7 : 1;
    size_t raw_size = 0;
    for (int p = 0; p < n_passes; ++p) {
        const Pass& ps = interlace ? adam7[p] : whole;
        const size_t pw = w > ps.x0 ? (w - ps.x0 + ps.dx - 1) / ps.dx : 0, ph = h > ps.y0 ? (h - ps.y0 + ps.dy - 1) / ps.dy : 0;
        if (pw && ph) raw_size += ((pw * bits_pp + 7) / 8 + 1) * ph;
    }
    // deflate cannot expand by more than about 1032:1: a header that promises more than the data can hold is refused
    // before anything of that size is allocated
    if (raw_size > idat.size() * 1032 + 65536) { err = "PNG image data is too short for the size in its header"; return false; }
    std::vector<uint8_t> raw(raw_size);
    uLongf raw_len = (uLongf)raw.size();
    const int zr = uncompress(raw.data(), &raw_len, idat.data(), (uLong)idat.size());
    if (zr != Z_OK || raw_len != raw.size()) { err = "PNG image data does not inflate to the expected size"; return false; }
    // expansion to 8-bit RGBA the way stbi_load(..., STBI_rgb_alpha) does: gray -> r=g=b, no alpha -> 255,
    // 16-bit -> high byte, low bit depths scaled to 0..255, palette looked up, tRNS applied
    img.width = (int)w; img.height = (int)h;
    img.rgba.assign((size_t)w * h * 4, 255);
    auto sample = [&](const uint8_t* row, size_t index) -> uint32_t {   // index-th sample of the row, raw value
        if (depth == 8) return row[index];
        if (depth == 16) return ((uint32_t)row[2 * index] << 8) | row[2 * index + 1];
        const size_t bit = index * depth;
        return (row[bit >> 3] >> (8 - depth - (bit & 7))) & ((1u << depth) - 1u);
    };
    const uint32_t maxv = (1u << depth) - 1u;
    auto to8 = [&](uint32_t v) -> uint8_t { return depth == 16 ? (uint8_t)(v >> 8) : depth == 8 ? (uint8_t)v : (uint8_t)(v * 255u / maxv); };
    uint32_t key[3] = {0, 0, 0};
    const bool has_key = (ctype == 0 && trns.size() >= 2) || (ctype == 2 && trns.size() >= 6);
    if (has_key) for (int c = 0; c < (ctype == 0 ? 1 : 3); ++c) key[c] = ((uint32_t)trns[2 * c] << 8) | trns[2 * c + 1];
    uint8_t* pass_data = raw.data();
    for (int p = 0; p < n_passes; ++p) {
        const Pass& ps = interlace ? adam7[p] : whole;
        const size_t pw = w > ps.x0 ? (w - ps.x0 + ps.dx - 1) / ps.dx : 0, ph = h > ps.y0 ? (h - ps.y0 + ps.dy - 1) / ps.dy : 0;
        if (!pw || !ph) continue;
        const size_t stride = (pw * bits_pp + 7) / 8;
        // undo the scan-line filters in place (row r of the pass starts at r*(stride+1)+1)
        std::vector<uint8_t> zero(stride, 0);
        for (size_t y = 0; y < ph; ++y) {
            uint8_t* cur = pass_data + y * (stride + 1) + 1;
            const uint8_t* up = y ? cur - (stride + 1) : zero.data();
            switch (cur[-1]) {
            case 0: break;
            case 1: for (size_t i = bpp; i < stride; ++i) cur[i] = (uint8_t)(cur[i] + cur[i - bpp]); break;
            case 2: for (size_t i = 0; i < stride; ++i) cur[i] = (uint8_t)(cur[i] + up[i]); break;
            case 3: for (size_t i = 0; i < stride; ++i) cur[i] = (uint8_t)(cur[i] + (((i >= bpp ? cur[i - bpp] : 0) + up[i]) >> 1)); break;
            case 4: for (size_t i = 0; i < stride; ++i) cur[i] = (uint8_t)(cur[i] + paeth(i >= bpp ? cur[i - bpp] : 0, up[i], i >= bpp ? up[i - bpp] : 0)); break;
            default: err = "unknown PNG filter type"; return false;
            }
        }
        for (size_t y = 0; y < ph; ++y) {
            const uint8_t* row = pass_data + y * (stride + 1) + 1;
            for (size_t x = 0; x < pw; ++x) {
                uint8_t* o = img.rgba.data() + (((size_t)ps.y0 + y * ps.dy) * w + ps.x0 + x * ps.dx) * 4;
                if (ctype == 3) {
                    const uint32_t i = sample(row, x);
                    if (3 * (size_t)i + 2 >= plte.size()) { err = "PNG palette index out of range"; return false; }
                    o[0] = plte[3 * i]; o[1] = plte[3 * i + 1]; o[2] = plte[3 * i + 2];
                    o[3] = i < trns.size() ? trns[i] : 255;
                } else if (ctype == 0 || ctype == 4) {
                    const uint32_t g = sample(row, x * channels);
                    o[0] = o[1] = o[2] = to8(g);
                    if (ctype == 4) o[3] = to8(sample(row, x * 2 + 1));
                    else if (has_key && g == key[0]) o[3] = 0;
                } else {
                    const uint32_t r = sample(row, x * channels), g = sample(row, x * channels + 1), b = sample(row, x * channels + 2);
                    o[0] = to8(r); o[1] = to8(g); o[2] = to8(b);
                    if (ctype == 6) o[3] = to8(sample(row, x * 4 + 3));
                    else if (has_key && r == key[0] && g == key[1] && b == key[2]) o[3] = 0;
                }
            }
        }
        pass_data += (stride + 1) * ph;
    }
    return true;
}

bool load_image(const std::string& path, Image& img, std::string& err)
{
    std::vector<uint8_t> bytes;
    if (!read_file(path, bytes, err)) return false;
    if (bytes.size() >= 3 && bytes[0] == 0xFF && bytes[1] == 0xD8) {
        const char* which = std::getenv("SR_JPEG_DECODER");
        const JpegFlavor flavor = which && !std::strcmp(which, "libjpeg") ? JpegFlavor::LibJpeg : JpegFlavor::Stb;
        if (!decode_jpeg(bytes.data(), bytes.size(), img, err, flavor)) { err += ": " + path; return false; }
        return true;
    }
    if (!decode_png(bytes.data(), bytes.size(), img, err)) { err += ": " + path; return false; }
    return true;
}

// ---- JSON (just enough for glTF) -------------------------------------------------------------------
namespace {

struct Json {
    enum Kind { Null, Bool, Number, String, Array, Object } kind = Null;
    bool b = false;
    double num = 0.0;
    std::string str;
    std::vector<Json> arr;
    std::map<std::string, Json> obj;
    const Json* get(const char* key) const
    {
        if (kind != Object) return nullptr;
        auto it = obj.find(key);
        return it == obj.end() ? nullptr : &it->second;
    }
    bool is_index() const { return kind == Number && num >= 0 && num == std::floor(num) && num < 4294967296.0; }
};

struct JsonParser {
    const char* p;
    const char* end;
    std::string err;
    int depth = 0;
    void ws() { while (p < end && (*p == ' ' || *p == '\n' || *p == '\r' || *p == '\t')) ++p; }
    bool fail(const char* what) { if (err.empty()) err = what; return false; }
    bool parse_string(std::string& out)
    {
        if (p >= end || *p != '"') return fail("JSON: string expected");
        ++p;
        while (p < end && *p != '"') {
            if (*p == '\\') {
                if (++p >= end) return fail("JSON: bad escape");
                switch (*p) {
                case '"': out += '"'; break; case '\\': out += '\\'; break; case '/': out += '/'; break;
                case 'b': out += '\b'; break; case 'f': out += '\f'; break; case 'n': out += '\n'; break;
                case 'r': out += '\r'; break; case 't': out += '\t'; break;
                case 'u': {
                    if (end - p < 5) return fail("JSON: bad \\u escape");
                    unsigned cp = 0;
                    for (int i = 1; i <= 4; ++i) {
                        const char c = p[i];
                        cp = cp * 16 + (c >= '0' && c <= '9' ? c - '0' : c >= 'a' && c <= 'f' ? c - 'a' + 10 : c >= 'A' && c <= 'F' ? c - 'A' + 10 : 99);
                    }
                    p += 4;
                    if (cp < 0x80) out += (char)cp;
                    else if (cp < 0x800) { out += (char)(0xC0 | (cp >> 6)); out += (char)(0x80 | (cp & 63)); }
                    else { out += (char)(0xE0 | (cp >> 12)); out += (char)(0x80 | ((cp >> 6) & 63)); out += (char)(0x80 | (cp & 63)); }
                    break;
                }
                default: return fail("JSON: unknown escape");
                }
                ++p;
            } else out += *p++;
        }
        if (p >= end) return fail("JSON: unterminated string");
        ++p;
        return true;
    }
    bool parse(Json& v)
    {
        if (++depth > 200) return fail("JSON: nesting too deep");
        ws();
        if (p >= end) return fail("JSON: unexpected end");
        bool ok = true;
        if (*p == '{') {
            v.kind = Json::Object; ++p; ws();
            if (p < end && *p == '}') ++p;
            else for (;;) {
                std::string key;
                ws();
                if (!parse_string(key)) { ok = false; break; }
                ws();
                if (p >= end || *p != ':') { ok = fail("JSON: ':' expected"); break; }
                ++p;
                if (!parse(v.obj[key])) { ok = false; break; }
                ws();
                if (p < end && *p == ',') { ++p; continue; }
                if (p < end && *p == '}') { ++p; break; }
                ok = fail("JSON: ',' or '}' expected"); break;
            }
        } else if (*p == '[') {
            v.kind = Json::Array; ++p; ws();
            if (p < end && *p == ']') ++p;
            else for (;;) {
                v.arr.emplace_back();
                if (!parse(v.arr.back())) { ok = false; break; }
                ws();
                if (p < end && *p == ',') { ++p; continue; }
                if (p < end && *p == ']') { ++p; break; }
                ok = fail("JSON: ',' or ']' expected"); break;
            }
        } else if (*p == '"') { v.kind = Json::String; ok = parse_string(v.str); }
        else if (end - p >= 4 && !std::strncmp(p, "true", 4)) { v.kind = Json::Bool; v.b = true; p += 4; }
        else if (end - p >= 5 && !std::strncmp(p, "false", 5)) { v.kind = Json::Bool; v.b = false; p += 5; }
        else if (end - p >= 4 && !std::strncmp(p, "null", 4)) { v.kind = Json::Null; p += 4; }
        else {
            const std::string text(p, std::min<size_t>((size_t)(end - p), 64));
            char* stop = nullptr;
            v.num = std::strtod(text.c_str(), &stop);
            if (stop == text.c_str()) ok = fail("JSON: value expected");
            else { v.kind = Json::Number; p += stop - text.c_str(); }
        }
        --depth;
        return ok;
    }
};

bool base64_decode(const char* s, size_t n, std::vector<uint8_t>& out)
{
    uint32_t acc = 0; int bits = 0;
    for (size_t i = 0; i < n; ++i) {
        const char c = s[i];
        int v;
        if (c >= 'A' && c <= 'Z') v = c - 'A'; else if (c >= 'a' && c <= 'z') v = c - 'a' + 26; else if (c >= '0' && c <= '9') v = c - '0' + 52;
        else if (c == '+' || c == '-') v = 62; else if (c == '/' || c == '_') v = 63; else if (c == '=' || c == '\n' || c == '\r') continue; else return false;
        acc = (acc << 6) | (uint32_t)v; bits += 6;
        if (bits >= 8) { bits -= 8; out.push_back((uint8_t)(acc >> bits)); }
    }
    return true;
}

std::string directory_of(const std::string& path)
{
    const size_t s = path.find_last_of("/\\");
    return s == std::string::npos ? std::string() : path.substr(0, s + 1);
}

struct Gltf {
    Json root;
    std::vector<std::vector<uint8_t>> buffers;
    std::string dir, err;
    std::vector<uint8_t> glb_bin;

    bool fail(const std::string& what) { if (err.empty()) err = what; return false; }
    const Json* element(const char* list, const Json* index)
    {
        const Json* l = root.get(list);
        if (!l || l->kind != Json::Array || !index || !index->is_index() || (size_t)index->num >= l->arr.size()) return nullptr;
        return &l->arr[(size_t)index->num];
    }
    bool load_buffers()
    {
        const Json* l = root.get("buffers");
        if (!l) return true;
        if (l->kind != Json::Array) return fail("glTF: buffers is not an array");
        for (size_t i = 0; i < l->arr.size(); ++i) {
            buffers.emplace_back();
            const Json* uri = l->arr[i].get("uri");
            if (!uri) {   // the binary chunk of a .glb
                if (i != 0 || glb_bin.empty()) return fail("glTF: buffer without uri");
                buffers.back() = glb_bin;
            } else if (uri->kind != Json::String) return fail("glTF: buffer uri is not a string");
            else if (!uri->str.compare(0, 5, "data:")) {
                const size_t comma = uri->str.find(',');
                if (comma == std::string::npos || uri->str.find(";base64") == std::string::npos || uri->str.find(";base64") > comma) return fail("glTF: data uri is not base64");
                if (!base64_decode(uri->str.c_str() + comma + 1, uri->str.size() - comma - 1, buffers.back())) return fail("glTF: bad base64 data");
            } else {
                std::string e;
                if (!read_file(dir + uri->str, buffers.back(), e)) return fail("glTF: " + e);
            }
            const Json* len = l->arr[i].get("byteLength");
            if (!len || !len->is_index() || (size_t)len->num > buffers.back().size()) return fail("glTF: buffer shorter than its byteLength");
        }
        return true;
    }
    // reads accessor `index` as `want` floats per element (normalised integers converted) or as indices
    bool read_accessor(const Json* index, int want_components, std::vector<float>* floats, std::vector<uint32_t>* ints, size_t& count)
    {
        const Json* a = element("accessors", index);
        if (!a) return fail("glTF: accessor index out of range");
        if (a->get("sparse")) return fail("glTF: sparse accessors are not supported");
        const Json* ct = a->get("componentType"); const Json* cn = a->get("count"); const Json* ty = a->get("type");
        if (!ct || !ct->is_index() || !cn || !cn->is_index() || !ty || ty->kind != Json::String) return fail("glTF: malformed accessor");
        const int comp = (int)ct->num;
        const int csize = comp == 5120 || comp == 5121 ? 1 : comp == 5122 || comp == 5123 ? 2 : comp == 5125 || comp == 5126 ? 4 : 0;
        const int ncomp = ty->str == "SCALAR" ? 1 : ty->str == "VEC2" ? 2 : ty->str == "VEC3" ? 3 : ty->str == "VEC4" ? 4 : 0;
        if (!csize || !ncomp) return fail("glTF: unsupported accessor type");
        if (ncomp < want_components) return fail("glTF: accessor has too few components");
        count = (size_t)cn->num;
        const bool normalized = a->get("normalized") && a->get("normalized")->kind == Json::Bool && a->get("normalized")->b;
        const Json* bvi = a->get("bufferView");
        if (!bvi) {   // all zeros by definition
            if (count > ((size_t)1 << 26)) return fail("glTF: accessor without a buffer view is unreasonably large");   // nothing in the file backs this size
            if (floats) floats->assign(count * want_components, 0.0f);
            if (ints) ints->assign(count, 0u);
            return true;
        }
        const Json* bv = element("bufferViews", bvi);
        if (!bv) return fail("glTF: bufferView index out of range");
        const Json* bi = bv->get("buffer");
        if (!bi || !bi->is_index() || (size_t)bi->num >= buffers.size()) return fail("glTF: buffer index out of range");
        const std::vector<uint8_t>& buf = buffers[(size_t)bi->num];
        auto opt = [&](const Json* o, const char* k) -> size_t { const Json* v = o->get(k); return v && v->is_index() ? (size_t)v->num : 0; };
        const size_t elem = (size_t)csize * ncomp;
        size_t stride = opt(bv, "byteStride");
        if (stride == 0) stride = elem;
        const size_t start = opt(bv, "byteOffset") + opt(a, "byteOffset");
        if (count && (stride < elem || start + (count - 1) * stride + elem > buf.size())) return fail("glTF: accessor reaches past its buffer");
        if (floats) floats->resize(count * want_components);
        if (ints) ints->resize(count);
        for (size_t i = 0; i < count; ++i) {
            const uint8_t* e = buf.data() + start + i * stride;
            for (int c = 0; c < (ints ? 1 : want_components); ++c) {
                const uint8_t* q = e + (size_t)c * csize;
                double v;
                switch (comp) {
                case 5120: { int8_t t; std::memcpy(&t, q, 1); v = normalized ? std::max(t / 127.0, -1.0) : t; break; }
                case 5121: v = normalized ? q[0] / 255.0 : q[0]; break;
                case 5122: { int16_t t; std::memcpy(&t, q, 2); v = normalized ? std::max(t / 32767.0, -1.0) : t; break; }
                case 5123: { uint16_t t; std::memcpy(&t, q, 2); v = normalized ? t / 65535.0 : t; break; }
                case 5125: { uint32_t t; std::memcpy(&t, q, 4); v = t; break; }
                default: { float t; std::memcpy(&t, q, 4); v = t; break; }
                }
                if (ints) { if (comp == 5126 || v < 0) return fail("glTF: indices must be unsigned integers"); (*ints)[i] = (uint32_t)v; }
                else (*floats)[i * want_components + c] = (float)v;
            }
        }
        return true;
    }
    // is the base-colour texture of `material` stored inside the file (data: uri or buffer view)?
    bool diffuse_is_embedded(const Json* material)
    {
        const Json* pbr = material->get("pbrMetallicRoughness");
        const Json* bct = pbr ? pbr->get("baseColorTexture") : nullptr;
        const Json* tex = bct ? element("textures", bct->get("index")) : nullptr;
        const Json* image = tex ? element("images", tex->get("source")) : nullptr;
        if (!image) return false;
        if (image->get("bufferView")) return true;
        const Json* uri = image->get("uri");
        return uri && uri->kind == Json::String && !uri->str.compare(0, 5, "data:");
    }
    bool primitive_to_mesh(const Json& prim, HostMesh& out)
    {
        // src/model.cpp:36-54: a mesh must have a material, and the material an embedded diffuse texture
        const Json* material = element("materials", prim.get("material"));
        if (!material) return fail("No material!");
        if (!diffuse_is_embedded(material)) return fail("Cannot load diffuse texture");
        const Json* attrs = prim.get("attributes");
        if (!attrs || attrs->kind != Json::Object) return fail("glTF: primitive without attributes");
        // src/model.cpp:22-28 reads positions, normals and the first uv set unconditionally
        if (!attrs->get("POSITION") || !attrs->get("NORMAL") || !attrs->get("TEXCOORD_0")) return fail("glTF: primitive needs POSITION, NORMAL and TEXCOORD_0");
        std::vector<float> pos, nrm, uv;
        size_t n = 0, n2 = 0, n3 = 0;
        if (!read_accessor(attrs->get("POSITION"), 3, &pos, nullptr, n) || !read_accessor(attrs->get("NORMAL"), 3, &nrm, nullptr, n2) ||
            !read_accessor(attrs->get("TEXCOORD_0"), 2, &uv, nullptr, n3)) return false;
        if (n2 != n || n3 != n) return fail("glTF: attribute counts differ");
        out.vertices.resize(n * 8);
        for (size_t i = 0; i < n; ++i) {
            float* v = &out.vertices[i * 8];
            v[0] = pos[3 * i]; v[1] = pos[3 * i + 1]; v[2] = pos[3 * i + 2];
            v[3] = nrm[3 * i]; v[4] = nrm[3 * i + 1]; v[5] = nrm[3 * i + 2];
            v[6] = uv[2 * i]; v[7] = 1.0f - uv[2 * i + 1];   // aiProcess_FlipUVs
        }
        std::vector<uint32_t> raw;
        if (prim.get("indices")) { size_t ni = 0; if (!read_accessor(prim.get("indices"), 1, nullptr, &raw, ni)) return false; }
        else { raw.resize(n); for (size_t i = 0; i < n; ++i) raw[i] = (uint32_t)i; }
        for (uint32_t i : raw) if (i >= n) return fail("glTF: index out of range");
        const Json* mode_j = prim.get("mode");
        const int mode = mode_j && mode_j->is_index() ? (int)mode_j->num : 4;
        out.indices.clear();
        if (mode == 4) { raw.resize(raw.size() / 3 * 3); out.indices = raw; }
        else if (mode == 5) {   // strip, winding alternates
            for (size_t i = 2; i < raw.size(); ++i) {
                if (i & 1) { out.indices.push_back(raw[i - 1]); out.indices.push_back(raw[i - 2]); out.indices.push_back(raw[i]); }
                else { out.indices.push_back(raw[i - 2]); out.indices.push_back(raw[i - 1]); out.indices.push_back(raw[i]); }
            }
        } else if (mode == 6) {
            for (size_t i = 2; i < raw.size(); ++i) { out.indices.push_back(raw[0]); out.indices.push_back(raw[i - 1]); out.indices.push_back(raw[i]); }
        } else return fail("glTF: only triangle primitives can be drawn");
        return true;
    }
    bool walk(const Json* node_index, std::vector<HostMesh>& meshes, int depth)
    {
        if (depth > 256) return fail("glTF: node hierarchy too deep (cycle?)");
        const Json* node = element("nodes", node_index);
        if (!node) return fail("glTF: node index out of range");
        if (node->get("mesh")) {   // node transforms are ignored, as in src/model.cpp:61-77
            const Json* mesh = element("meshes", node->get("mesh"));
            const Json* prims = mesh ? mesh->get("primitives") : nullptr;
            if (!prims || prims->kind != Json::Array) return fail("glTF: mesh without primitives");
            for (const Json& prim : prims->arr) {
                meshes.emplace_back();
                if (!primitive_to_mesh(prim, meshes.back())) return false;
            }
        }
        if (const Json* kids = node->get("children")) {
            if (kids->kind != Json::Array) return fail("glTF: children is not an array");
            for (const Json& k : kids->arr) if (!walk(&k, meshes, depth + 1)) return false;
        }
        return true;
    }
};

}  // namespace

bool load_model(const std::string& path, std::vector<HostMesh>& meshes, std::string& err)
{
    std::vector<uint8_t> bytes;
    if (!read_file(path, bytes, err)) return false;
    Gltf g;
    g.dir = directory_of(path);
    const char* json_begin = (const char*)bytes.data();
    size_t json_len = bytes.size();
    if (bytes.size() >= 20 && !std::memcmp(bytes.data(), "glTF", 4)) {   // .glb: header, JSON chunk, optional BIN chunk
        auto le32 = [&](size_t o) { uint32_t v; std::memcpy(&v, bytes.data() + o, 4); return v; };
        if (le32(4) != 2) { err = "glb: only container version 2 is supported"; return false; }
        const size_t jl = le32(12);
        if (le32(16) != 0x4E4F534Au || 20 + jl > bytes.size()) { err = "glb: first chunk is not JSON"; return false; }
        json_begin = (const char*)bytes.data() + 20; json_len = jl;
        const size_t b = 20 + jl;
        if (b + 8 <= bytes.size() && le32(b + 4) == 0x004E4942u && b + 8 + le32(b) <= bytes.size()) g.glb_bin.assign(bytes.begin() + b + 8, bytes.begin() + b + 8 + le32(b));
    }
    JsonParser jp{json_begin, json_begin + json_len, {}, 0};
    if (!jp.parse(g.root) || g.root.kind != Json::Object) { err = (jp.err.empty() ? std::string("glTF: not a JSON object") : jp.err) + ": " + path; return false; }
    if (!g.load_buffers()) { err = g.err + ": " + path; return false; }
    // the default scene's root nodes, depth first (src/model.cpp:61-77)
    const Json* scenes = g.root.get("scenes");
    const Json* scene_index = g.root.get("scene");
    const Json* scene = nullptr;
    if (scenes && scenes->kind == Json::Array && !scenes->arr.empty())
        scene = scene_index && scene_index->is_index() && (size_t)scene_index->num < scenes->arr.size() ? &scenes->arr[(size_t)scene_index->num] : &scenes->arr[0];
    const Json* roots = scene ? scene->get("nodes") : nullptr;
    if (!roots || roots->kind != Json::Array) { err = "Error, cannot read scene from " + path; return false; }   // src/model.cpp:88-91
    meshes.clear();
    for (const Json& r : roots->arr)
        if (!g.walk(&r, meshes, 0)) { err = g.err + ": " + path; return false; }
    return true;
}

}  // namespace sr_assets
