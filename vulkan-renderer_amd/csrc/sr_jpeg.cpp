// JPEG reader for create_texture_from_file (src/texture.cpp:273-291: stbi_load(path, ..., STBI_rgb_alpha)); the
// reference's own albedo map data/textures/chalet.jpg is a baseline 4:2:0 JPEG.  Huffman-coded 8-bit JPEG:
// baseline, extended sequential and progressive frames, any integral sampling ratios, restart intervals,
// interleaved and per-component scans, gray / YCbCr / RGB (Adobe transform 0) colour.  Arithmetic coding,
// lossless, 12-bit and four-component (CMYK/YCCK) files are refused.
//
// Everything after entropy decoding (inverse DCT, chroma upsampling, colour conversion) exists in two flavours,
// because decoders differ there in the last bit:
//   JpegFlavor::Stb     restates the integer arithmetic published in stb_image 2.19 (the version the reference
//                       pins, conanfile.txt:3): its 12-bit-constant IDCT, its 3:1 "hv_2" chroma filter rounded
//                       +8>>4 on both phases, its 20-bit fixed-point YCbCr.  stb_image is not present on this
//                       machine, so this flavour is not checked against stb itself; the tests hold it within a
//                       few code values of the other flavour.
//   JpegFlavor::LibJpeg restates libjpeg's defaults (jidctint "islow" IDCT, "fancy" triangle upsampling with
//                       alternating 8/7 or 1/2 rounding bias, 16-bit fixed-point colour tables); checked bit for
//                       bit against libjpeg-turbo through Pillow in tests/test_asset_files.py.
// Pure CPU code.
#include "sr_assets.h"

#include <algorithm>
#include <cstring>

namespace sr_assets {

namespace {

// zig-zag position -> index in the natural (row-major) 8x8 block
const uint8_t kNatural[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                              41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                              30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct HuffTable {
    bool present = false;
    uint8_t vals[256];
    int32_t maxcode[18];     // largest code of each length, -1 if none
    int32_t valoffset[17];   // index of the first value of that length minus its first code
    uint16_t look[512];      // codes of up to 9 bits: (length << 8) | value, 0 = longer code

    bool build(const uint8_t* counts /* [16] */, const uint8_t* values, int n_values)
    {
        std::memcpy(vals, values, (size_t)n_values);
        std::memset(look, 0, sizeof look);
        int code = 0, k = 0;
        for (int len = 1; len <= 16; ++len) {
            valoffset[len] = k - code;
            for (int i = 0; i < counts[len - 1]; ++i, ++k, ++code) {
                if (code >= (1 << len)) return false;   // more codes than the length can hold
                if (len <= 9)
                    for (int fill = 0; fill < (1 << (9 - len)); ++fill) look[(code << (9 - len)) | fill] = (uint16_t)((len << 8) | vals[k]);
            }
            maxcode[len] = counts[len - 1] ? code - 1 : -1;
            code <<= 1;
        }
        maxcode[17] = 0x7fffffff;
        present = true;
        return true;
    }
};

// Entropy-coded segment reader: removes the 00 stuffed after FF, stops in front of a marker and hands out zero
// bits from there on (a damaged file then decodes to something instead of reading past its end).
struct BitReader {
    const uint8_t* p;
    const uint8_t* end;
    uint64_t acc = 0;
    int n = 0;
    bool at_marker = false;

    void fill()
    {
        while (n <= 56) {
            unsigned b = 0;
            if (!at_marker) {
                if (p >= end) at_marker = true;
                else if (*p != 0xFF) b = *p++;
                else if (p + 1 < end && p[1] == 0x00) { b = 0xFF; p += 2; }
                else at_marker = true;
            }
            acc = (acc << 8) | b;
            n += 8;
        }
    }
    unsigned peek(int k) { return (unsigned)(acc >> (n - k)) & ((1u << k) - 1u); }   // needs n >= k
    unsigned get(int k)
    {
        if (k == 0) return 0;
        if (n < k) fill();
        n -= k;
        return (unsigned)(acc >> n) & ((1u << k) - 1u);
    }
    int decode(const HuffTable& t)   // -1: no such code
    {
        if (n < 16) fill();
        const uint16_t e = t.look[peek(9)];
        if (e) { n -= e >> 8; return e & 255; }
        for (int len = 10; len <= 16; ++len) {
            const int code = (int)peek(len);
            if (code <= t.maxcode[len]) { n -= len; return t.vals[(code + t.valoffset[len]) & 255]; }
        }
        return -1;
    }
    // drops the rest of the current byte and steps over an RSTn marker if one is next
    void restart()
    {
        acc = 0; n = 0; at_marker = false;
        while (p + 1 < end && !(p[0] == 0xFF && p[1] != 0x00 && p[1] != 0xFF)) ++p;   // to the next marker
        if (p + 1 < end && p[1] >= 0xD0 && p[1] <= 0xD7) p += 2;
        else at_marker = true;
    }
};

inline int extend(unsigned v, int s) { return (int)v < (1 << (s - 1)) ? (int)v - (1 << s) + 1 : (int)v; }

struct Component {
    int id = 0, h = 1, v = 1, tq = 0;
    int dw = 0, dh = 0;              // size in samples at this component's resolution: ceil(W*h/hmax), ceil(H*v/vmax)
    int bw = 0, bh = 0;              // allocated blocks: whole MCUs of an interleaved scan
    std::vector<int16_t> coef;       // bw*bh blocks of 64, natural order
    uint16_t quant[64];              // natural order, taken when the component's first scan starts
    bool quant_latched = false;
    std::vector<uint8_t> plane;      // bw*8 x bh*8 samples after the inverse DCT
    int dc_pred = 0;
};

struct Decoder {
    const uint8_t* data;
    size_t size;
    std::string& err;
    JpegFlavor flavor;
    int W = 0, H = 0, hmax = 1, vmax = 1;
    bool progressive = false, have_frame = false, jfif = false;
    int adobe_transform = -1;
    int restart_interval = 0;
    std::vector<Component> comps;
    uint16_t qt[4][64];
    bool qt_present[4] = {false, false, false, false};
    HuffTable dc_tab[4], ac_tab[4];
    int scans = 0;

    Decoder(const uint8_t* d, size_t n, std::string& e, JpegFlavor f) : data(d), size(n), err(e), flavor(f) {}
    bool fail(const char* what) { err = what; return false; }

    bool read_dqt(const uint8_t* b, size_t len)
    {
        size_t i = 0;
        while (i < len) {
            const int pq = b[i] >> 4, tq = b[i] & 15;
            ++i;
            if (tq > 3 || pq > 1) return fail("bad JPEG quantisation table header");
            if (i + (pq ? 128 : 64) > len) return fail("truncated JPEG quantisation table");
            for (int k = 0; k < 64; ++k, i += pq ? 2 : 1) qt[tq][kNatural[k]] = pq ? (uint16_t)((b[i] << 8) | b[i + 1]) : b[i];
            qt_present[tq] = true;
        }
        return true;
    }

    bool read_dht(const uint8_t* b, size_t len)
    {
        size_t i = 0;
        while (i < len) {
            if (i + 17 > len) return fail("truncated JPEG Huffman table");
            const int tc = b[i] >> 4, th = b[i] & 15;
            if (tc > 1 || th > 3) return fail("bad JPEG Huffman table header");
            int total = 0;
            for (int k = 0; k < 16; ++k) total += b[i + 1 + k];
            if (total > 256 || i + 17 + (size_t)total > len) return fail("bad JPEG Huffman table size");
            if (!(tc ? ac_tab : dc_tab)[th].build(b + i + 1, b + i + 17, total)) return fail("bad JPEG Huffman code lengths");
            i += 17 + (size_t)total;
        }
        return true;
    }

    bool read_sof(const uint8_t* b, size_t len, bool prog)
    {
        if (have_frame) return fail("JPEG with more than one frame");
        if (len < 6) return fail("truncated JPEG frame header");
        if (b[0] != 8) return fail("only 8-bit JPEG is supported");
        H = (b[1] << 8) | b[2];
        W = (b[3] << 8) | b[4];
        const int nc = b[5];
        if (W == 0 || H == 0) return fail("JPEG size out of range");
        if (nc != 1 && nc != 3) return fail("only gray and three-component JPEG is supported");
        if (len < 6 + 3 * (size_t)nc) return fail("truncated JPEG frame header");
        // every 8x8 block costs at least two bits (a DC code and an end-of-block code): a header that promises far
        // more than the file can hold is refused before anything of that size is allocated
        if ((uint64_t)W * H > (uint64_t)size * 1024 + 65536) return fail("JPEG data is too short for the size in its header");
        comps.resize(nc);
        for (int c = 0; c < nc; ++c) {
            Component& k = comps[c];
            k.id = b[6 + 3 * c];
            k.h = b[7 + 3 * c] >> 4;
            k.v = b[7 + 3 * c] & 15;
            k.tq = b[8 + 3 * c];
            if (k.h < 1 || k.h > 4 || k.v < 1 || k.v > 4 || k.tq > 3) return fail("bad JPEG component description");
            hmax = std::max(hmax, k.h);
            vmax = std::max(vmax, k.v);
        }
        const int mcux = (W + 8 * hmax - 1) / (8 * hmax), mcuy = (H + 8 * vmax - 1) / (8 * vmax);
        for (Component& k : comps) {
            if (hmax % k.h || vmax % k.v) return fail("fractional JPEG sampling ratios are not supported");
            k.dw = (W * k.h + hmax - 1) / hmax;
            k.dh = (H * k.v + vmax - 1) / vmax;
            k.bw = mcux * k.h;
            k.bh = mcuy * k.v;
            k.coef.assign((size_t)k.bw * k.bh * 64, 0);
        }
        progressive = prog;
        have_frame = true;
        return true;
    }

    // ---- one scan ---------------------------------------------------------------------------------
    struct ScanComp { Component* c; int td, ta; };

    bool block_baseline(BitReader& br, const ScanComp& sc, int16_t* blk)
    {
        const int s = br.decode(dc_tab[sc.td]);
        if (s < 0 || s > 15) return fail("bad JPEG Huffman code");
        sc.c->dc_pred = (int)((unsigned)sc.c->dc_pred + (unsigned)(s ? extend(br.get(s), s) : 0));
        blk[0] = (int16_t)sc.c->dc_pred;
        const HuffTable& at = ac_tab[sc.ta];
        for (int k = 1; k < 64;) {
            const int rs = br.decode(at);
            if (rs < 0) return fail("bad JPEG Huffman code");
            const int r = rs >> 4, s2 = rs & 15;
            if (s2) {
                k += r;
                if (k > 63) return fail("JPEG coefficient index out of range");
                blk[kNatural[k++]] = (int16_t)extend(br.get(s2), s2);
            } else if (r == 15) k += 16;
            else break;
        }
        return true;
    }

    bool block_dc_progressive(BitReader& br, const ScanComp& sc, int16_t* blk, int ah, int al)
    {
        if (ah == 0) {
            const int s = br.decode(dc_tab[sc.td]);
            if (s < 0 || s > 15) return fail("bad JPEG Huffman code");
            sc.c->dc_pred = (int)((unsigned)sc.c->dc_pred + (unsigned)(s ? extend(br.get(s), s) : 0));
            blk[0] = (int16_t)((int64_t)sc.c->dc_pred * (1 << al));
        } else if (br.get(1)) blk[0] = (int16_t)(blk[0] | (1 << al));
        return true;
    }

    bool block_ac_first(BitReader& br, const ScanComp& sc, int16_t* blk, int ss, int se, int al, unsigned& eobrun)
    {
        if (eobrun) { --eobrun; return true; }
        const HuffTable& at = ac_tab[sc.ta];
        for (int k = ss; k <= se; ++k) {
            const int rs = br.decode(at);
            if (rs < 0) return fail("bad JPEG Huffman code");
            const int r = rs >> 4, s = rs & 15;
            if (s) {
                k += r;
                if (k > 63) return fail("JPEG coefficient index out of range");
                blk[kNatural[k]] = (int16_t)(extend(br.get(s), s) * (1 << al));
            } else if (r == 15) k += 15;
            else {
                eobrun = (1u << r) + (r ? br.get(r) : 0u) - 1u;
                break;
            }
        }
        return true;
    }

    bool block_ac_refine(BitReader& br, const ScanComp& sc, int16_t* blk, int ss, int se, int al, unsigned& eobrun)
    {
        const int p1 = 1 << al, m1 = -(1 << al);
        const HuffTable& at = ac_tab[sc.ta];
        auto correct = [&](int16_t& c) {   // one more bit for a coefficient that is already non-zero
            if (br.get(1) && (c & p1) == 0) c = (int16_t)(c + (c >= 0 ? p1 : m1));
        };
        int k = ss;
        if (eobrun == 0) {
            for (; k <= se; ++k) {
                const int rs = br.decode(at);
                if (rs < 0) return fail("bad JPEG Huffman code");
                int r = rs >> 4, s = rs & 15;
                if (s) s = br.get(1) ? p1 : m1;   // the size is always 1 here: a newly non-zero coefficient
                else if (r != 15) {
                    eobrun = (1u << r) + (r ? br.get(r) : 0u);
                    break;
                }
                // pass r coefficients that are still zero; every non-zero one on the way takes a correction bit
                for (; k <= se; ++k) {
                    int16_t& c = blk[kNatural[k]];
                    if (c != 0) correct(c);
                    else if (--r < 0) break;
                }
                if (s && k <= 63) blk[kNatural[k]] = (int16_t)s;
            }
        }
        if (eobrun) {
            for (; k <= se; ++k) {
                int16_t& c = blk[kNatural[k]];
                if (c != 0) correct(c);
            }
            --eobrun;
        }
        return true;
    }

    // b points behind the SOS length field; returns the position behind the scan's entropy-coded data
    bool read_scan(const uint8_t* b, size_t len, size_t& pos)
    {
        if (!have_frame) return fail("JPEG scan before the frame header");
        if (len < 1) return fail("truncated JPEG scan header");
        const int ns = b[0];
        if (ns < 1 || ns > (int)comps.size() || len < 4 + 2 * (size_t)ns) return fail("bad JPEG scan header");
        ScanComp sc[3];
        for (int i = 0; i < ns; ++i) {
            sc[i].c = nullptr;
            for (Component& k : comps)
                if (k.id == b[1 + 2 * i]) sc[i].c = &k;
            for (int j = 0; j < i; ++j)
                if (sc[j].c == sc[i].c) return fail("JPEG scan names a component twice");
            if (!sc[i].c) return fail("JPEG scan names an unknown component");
            sc[i].td = b[2 + 2 * i] >> 4;
            sc[i].ta = b[2 + 2 * i] & 15;
            if (sc[i].td > 3 || sc[i].ta > 3) return fail("bad JPEG scan header");
        }
        int ss = b[1 + 2 * ns], se = b[2 + 2 * ns];
        const int ah = b[3 + 2 * ns] >> 4, al = b[3 + 2 * ns] & 15;
        if (progressive) {
            if (ss > se || se > 63 || al > 13 || ah > 13 || (ss == 0 && se != 0) || (ss > 0 && ns != 1)) return fail("bad JPEG progressive scan parameters");
        } else { ss = 0; se = 63; }
        const bool need_dc = ss == 0 && ah == 0, need_ac = se > 0;
        for (int i = 0; i < ns; ++i) {
            if (need_dc && !dc_tab[sc[i].td].present) return fail("JPEG scan uses a Huffman table that was not defined");
            if (need_ac && !ac_tab[sc[i].ta].present) return fail("JPEG scan uses a Huffman table that was not defined");
            Component& k = *sc[i].c;
            if (!k.quant_latched) {
                if (!qt_present[k.tq]) return fail("JPEG scan uses a quantisation table that was not defined");
                std::memcpy(k.quant, qt[k.tq], sizeof k.quant);
                k.quant_latched = true;
            }
            k.dc_pred = 0;
        }
        BitReader br;
        br.p = b + len;
        br.end = data + size;
        unsigned eobrun = 0;
        // a scan of one component covers just the blocks that hold image samples; an interleaved scan covers
        // whole MCUs of h x v blocks per component
        const int mcux = ns == 1 ? (sc[0].c->dw + 7) / 8 : (W + 8 * hmax - 1) / (8 * hmax);
        const int mcuy = ns == 1 ? (sc[0].c->dh + 7) / 8 : (H + 8 * vmax - 1) / (8 * vmax);
        int until_restart = restart_interval;
        for (int my = 0; my < mcuy; ++my)
            for (int mx = 0; mx < mcux; ++mx) {
                if (restart_interval && until_restart == 0) {
                    br.restart();
                    for (int i = 0; i < ns; ++i) sc[i].c->dc_pred = 0;
                    eobrun = 0;
                    until_restart = restart_interval;
                }
                --until_restart;
                for (int i = 0; i < ns; ++i) {
                    Component& k = *sc[i].c;
                    const int nh = ns == 1 ? 1 : k.h, nv = ns == 1 ? 1 : k.v;
                    for (int by = 0; by < nv; ++by)
                        for (int bx = 0; bx < nh; ++bx) {
                            int16_t* blk = k.coef.data() + ((size_t)(my * nv + by) * k.bw + (size_t)(mx * nh + bx)) * 64;
                            bool ok;
                            if (!progressive) ok = block_baseline(br, sc[i], blk);
                            else if (ss == 0) ok = block_dc_progressive(br, sc[i], blk, ah, al);
                            else if (ah == 0) ok = block_ac_first(br, sc[i], blk, ss, se, al, eobrun);
                            else ok = block_ac_refine(br, sc[i], blk, ss, se, al, eobrun);
                            if (!ok) return false;
                        }
                }
            }
        pos = (size_t)(br.p - data);
        ++scans;
        return true;
    }

    // ---- inverse DCT ------------------------------------------------------------------------------
    static uint8_t clamp8(int64_t v) { return (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v); }

    // libjpeg jidctint.c (13-bit constants, 2 extra bits kept between the passes); dequantised values are plain ints
    static void idct_islow(const int16_t* in, const uint16_t* q, uint8_t* out, int stride)
    {
        constexpr int kConst = 13, kPass1 = 2;
        constexpr int64_t F0_298 = 2446, F0_390 = 3196, F0_541 = 4433, F0_765 = 6270, F0_899 = 7373, F1_175 = 9633, F1_501 = 12299,
                          F1_847 = 15137, F1_961 = 16069, F2_053 = 16819, F2_562 = 20995, F3_072 = 25172;
        int64_t ws[64];
        auto pass = [&](int64_t s0, int64_t s1, int64_t s2, int64_t s3, int64_t s4, int64_t s5, int64_t s6, int64_t s7, int shift, int64_t* o, int ostep) {
            int64_t z1 = (s2 + s6) * F0_541;
            const int64_t e2 = z1 - s6 * F1_847, e3 = z1 + s2 * F0_765;
            const int64_t e0 = (s0 + s4) * (1 << kConst), e1 = (s0 - s4) * (1 << kConst);
            const int64_t t10 = e0 + e3, t13 = e0 - e3, t11 = e1 + e2, t12 = e1 - e2;
            int64_t t0 = s7, t1 = s5, t2 = s3, t3 = s1;
            z1 = t0 + t3;
            int64_t z2 = t1 + t2, z3 = t0 + t2, z4 = t1 + t3;
            const int64_t z5 = (z3 + z4) * F1_175;
            t0 *= F0_298; t1 *= F2_053; t2 *= F3_072; t3 *= F1_501;
            z1 *= -F0_899; z2 *= -F2_562; z3 *= -F1_961; z4 *= -F0_390;
            z3 += z5; z4 += z5;
            t0 += z1 + z3; t1 += z2 + z4; t2 += z2 + z3; t3 += z1 + z4;
            const int64_t half = (int64_t)1 << (shift - 1);
            o[0 * ostep] = (t10 + t3 + half) >> shift;
            o[7 * ostep] = (t10 - t3 + half) >> shift;
            o[1 * ostep] = (t11 + t2 + half) >> shift;
            o[6 * ostep] = (t11 - t2 + half) >> shift;
            o[2 * ostep] = (t12 + t1 + half) >> shift;
            o[5 * ostep] = (t12 - t1 + half) >> shift;
            o[3 * ostep] = (t13 + t0 + half) >> shift;
            o[4 * ostep] = (t13 - t0 + half) >> shift;
        };
        for (int c = 0; c < 8; ++c)
            pass(in[c] * (int)q[c], in[8 + c] * (int)q[8 + c], in[16 + c] * (int)q[16 + c], in[24 + c] * (int)q[24 + c], in[32 + c] * (int)q[32 + c],
                 in[40 + c] * (int)q[40 + c], in[48 + c] * (int)q[48 + c], in[56 + c] * (int)q[56 + c], kConst - kPass1, ws + c, 8);
        for (int r = 0; r < 8; ++r) {
            int64_t o[8];
            const int64_t* w = ws + 8 * r;
            pass(w[0], w[1], w[2], w[3], w[4], w[5], w[6], w[7], kConst + kPass1 + 3, o, 1);
            for (int c = 0; c < 8; ++c) out[r * stride + c] = clamp8(o[c] + 128);
        }
    }

    // stb_image's stbi__idct_block: 12-bit constants, the coefficient was multiplied by its quantiser into a
    // 16-bit short when it was decoded
    static void idct_stb(const int16_t* in, const uint16_t* q, uint8_t* out, int stride)
    {
        auto f2f = [](double x) { return (int)(x * 4096 + 0.5); };
        const int c0_541 = f2f(0.5411961), c1_847 = -f2f(1.847759065), c0_765 = f2f(0.765366865), c1_175 = f2f(1.175875602), c0_298 = f2f(0.298631336),
                  c2_053 = f2f(2.053119869), c3_072 = f2f(3.072711026), c1_501 = f2f(1.501321110), c0_899 = -f2f(0.899976223), c2_562 = -f2f(2.562915447),
                  c1_961 = -f2f(1.961570560), c0_390 = -f2f(0.390180644);
        // (64-bit intermediates: in a valid file everything fits 32 bits as in stb_image, in a damaged one nothing overflows)
        typedef int64_t I;
        I d[64], val[64];
        for (int i = 0; i < 64; ++i) d[i] = (int16_t)(in[i] * (int)q[i]);
        struct R { I x0, x1, x2, x3, t0, t1, t2, t3; };
        auto one_d = [&](I s0, I s1, I s2, I s3, I s4, I s5, I s6, I s7) {
            I p2 = s2, p3 = s6;
            I p1 = (p2 + p3) * c0_541;
            I t2 = p1 + p3 * c1_847, t3 = p1 + p2 * c0_765;
            p2 = s0; p3 = s4;
            I t0 = (p2 + p3) * 4096, t1 = (p2 - p3) * 4096;
            R r;
            r.x0 = t0 + t3; r.x3 = t0 - t3; r.x1 = t1 + t2; r.x2 = t1 - t2;
            t0 = s7; t1 = s5; t2 = s3; t3 = s1;
            p3 = t0 + t2;
            I p4 = t1 + t3;
            p1 = t0 + t3; p2 = t1 + t2;
            const I p5 = (p3 + p4) * c1_175;
            t0 *= c0_298; t1 *= c2_053; t2 *= c3_072; t3 *= c1_501;
            p1 = p5 + p1 * c0_899; p2 = p5 + p2 * c2_562;
            p3 *= c1_961; p4 *= c0_390;
            r.t3 = t3 + p1 + p4; r.t2 = t2 + p2 + p3; r.t1 = t1 + p2 + p4; r.t0 = t0 + p1 + p3;
            return r;
        };
        for (int c = 0; c < 8; ++c) {
            const I* s = d + c;
            I* v = val + c;
            if (s[8] == 0 && s[16] == 0 && s[24] == 0 && s[32] == 0 && s[40] == 0 && s[48] == 0 && s[56] == 0) {
                const I dc = s[0] * 4;
                v[0] = v[8] = v[16] = v[24] = v[32] = v[40] = v[48] = v[56] = dc;
            } else {
                R r = one_d(s[0], s[8], s[16], s[24], s[32], s[40], s[48], s[56]);
                r.x0 += 512; r.x1 += 512; r.x2 += 512; r.x3 += 512;
                v[0] = (r.x0 + r.t3) >> 10; v[56] = (r.x0 - r.t3) >> 10;
                v[8] = (r.x1 + r.t2) >> 10; v[48] = (r.x1 - r.t2) >> 10;
                v[16] = (r.x2 + r.t1) >> 10; v[40] = (r.x2 - r.t1) >> 10;
                v[24] = (r.x3 + r.t0) >> 10; v[32] = (r.x3 - r.t0) >> 10;
            }
        }
        for (int row = 0; row < 8; ++row) {
            const I* v = val + 8 * row;
            uint8_t* o = out + row * stride;
            R r = one_d(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
            const I bias = 65536 + (128 << 17);   // rounding, and the level shift of 128 in the same add
            r.x0 += bias; r.x1 += bias; r.x2 += bias; r.x3 += bias;
            o[0] = clamp8((r.x0 + r.t3) >> 17); o[7] = clamp8((r.x0 - r.t3) >> 17);
            o[1] = clamp8((r.x1 + r.t2) >> 17); o[6] = clamp8((r.x1 - r.t2) >> 17);
            o[2] = clamp8((r.x2 + r.t1) >> 17); o[5] = clamp8((r.x2 - r.t1) >> 17);
            o[3] = clamp8((r.x3 + r.t0) >> 17); o[4] = clamp8((r.x3 - r.t0) >> 17);
        }
    }

    // ---- chroma upsampling to full resolution --------------------------------------------------------
    // src: the component's plane (row pitch sp); only its dw x dh real samples are used, the edge sample repeats.
    // dst: (dw*hx) x (dh*vx) samples, pitch dw*hx.
    void upsample(const Component& k, std::vector<uint8_t>& dst) const
    {
        const int hx = hmax / k.h, vx = vmax / k.v, dw = k.dw, dh = k.dh, sp = k.bw * 8, ow = dw * hx;
        const uint8_t* src = k.plane.data();
        dst.resize((size_t)ow * dh * vx);
        auto row = [&](int y) { return src + (size_t)std::min(std::max(y, 0), dh - 1) * sp; };
        const bool lib = flavor == JpegFlavor::LibJpeg;
        if (hx == 1 && vx == 1) {
            for (int y = 0; y < dh; ++y) std::memcpy(dst.data() + (size_t)y * ow, row(y), (size_t)dw);
        } else if (hx == 2 && vx == 1 && (lib ? dw > 2 : true)) {
            for (int y = 0; y < dh; ++y) {
                const uint8_t* in = row(y);
                uint8_t* o = dst.data() + (size_t)y * ow;
                if (dw == 1) { o[0] = o[1] = in[0]; continue; }
                o[0] = in[0];
                o[1] = (uint8_t)((in[0] * 3 + in[1] + 2) >> 2);
                for (int i = 1; i < dw - 1; ++i) {
                    // libjpeg rounds the left phase with +1 and the right phase with +2, stb both with +2
                    o[2 * i] = (uint8_t)((in[i] * 3 + in[i - 1] + (lib ? 1 : 2)) >> 2);
                    o[2 * i + 1] = (uint8_t)((in[i] * 3 + in[i + 1] + 2) >> 2);
                }
                // stb_image weighs the last pair the other way round (its stbi__resample_row_h_2)
                o[2 * dw - 2] = lib ? (uint8_t)((in[dw - 1] * 3 + in[dw - 2] + 1) >> 2) : (uint8_t)((in[dw - 2] * 3 + in[dw - 1] + 2) >> 2);
                o[2 * dw - 1] = in[dw - 1];
            }
        } else if (hx == 1 && vx == 2) {
            for (int y = 0; y < 2 * dh; ++y) {
                const uint8_t* near = row(y >> 1);
                const uint8_t* far = row((y & 1) ? (y >> 1) + 1 : (y >> 1) - 1);
                uint8_t* o = dst.data() + (size_t)y * ow;
                const int bias = lib ? ((y & 1) ? 2 : 1) : 2;
                for (int i = 0; i < dw; ++i) o[i] = (uint8_t)((near[i] * 3 + far[i] + bias) >> 2);
            }
        } else if (hx == 2 && vx == 2 && (lib ? dw > 2 : true)) {
            for (int y = 0; y < 2 * dh; ++y) {
                const uint8_t* near = row(y >> 1);
                const uint8_t* far = row((y & 1) ? (y >> 1) + 1 : (y >> 1) - 1);
                uint8_t* o = dst.data() + (size_t)y * ow;
                auto colsum = [&](int i) { return near[i] * 3 + far[i]; };
                if (dw == 1) { o[0] = o[1] = (uint8_t)((colsum(0) + 2) >> 2); continue; }
                // libjpeg: +8 on the left phase, +7 on the right phase; stb: +8 on both
                const int rb = lib ? 7 : 8;
                int last, cur = colsum(0), next = colsum(1);
                o[0] = (uint8_t)((cur * 4 + 8) >> 4);
                o[1] = (uint8_t)((cur * 3 + next + rb) >> 4);
                for (int i = 1; i < dw - 1; ++i) {
                    last = cur; cur = next; next = colsum(i + 1);
                    o[2 * i] = (uint8_t)((cur * 3 + last + 8) >> 4);
                    o[2 * i + 1] = (uint8_t)((cur * 3 + next + rb) >> 4);
                }
                last = cur; cur = next;
                o[2 * dw - 2] = (uint8_t)((cur * 3 + last + 8) >> 4);
                o[2 * dw - 1] = (uint8_t)((cur * 4 + rb) >> 4);
            }
        } else {   // every other ratio: each sample repeated hx times across and vx times down
            for (int y = 0; y < dh * vx; ++y) {
                const uint8_t* in = row(y / vx);
                uint8_t* o = dst.data() + (size_t)y * ow;
                for (int i = 0; i < dw; ++i)
                    for (int r = 0; r < hx; ++r) *o++ = in[i];
            }
        }
    }

    // ---- colour ---------------------------------------------------------------------------------
    static void ycc_libjpeg(int y, int cb, int cr, uint8_t* o)
    {
        const int x = cb - 128, z = cr - 128;   // 16-bit fixed point, arithmetic shifts (jdcolor.c build_ycc_rgb_table)
        o[0] = clamp8(y + (int)((91881LL * z + 32768) >> 16));
        o[1] = clamp8(y + (int)((-22554LL * x + 32768 - 46802LL * z) >> 16));
        o[2] = clamp8(y + (int)((116130LL * x + 32768) >> 16));
    }
    static void ycc_stb(int y, int cb, int cr, uint8_t* o)
    {
        auto fixed = [](float x) { return ((int)(x * 4096.0f + 0.5f)) << 8; };
        const int yf = (y << 20) + (1 << 19), r = cr - 128, b = cb - 128;
        o[0] = clamp8((yf + r * fixed(1.40200f)) >> 20);
        o[1] = clamp8((int)(yf + r * -fixed(0.71414f) + (int)((unsigned)(b * -fixed(0.34414f)) & 0xffff0000u)) >> 20);
        o[2] = clamp8((yf + b * fixed(1.77200f)) >> 20);
    }

    bool finish(Image& img)
    {
        if (!have_frame || scans == 0) return fail("JPEG without image data");
        std::vector<uint8_t> full[3];
        for (size_t c = 0; c < comps.size(); ++c) {
            Component& k = comps[c];
            if (!k.quant_latched) return fail("JPEG component without a scan");
            const int pitch = k.bw * 8;
            k.plane.resize((size_t)pitch * k.bh * 8);
            for (int by = 0; by < k.bh; ++by)
                for (int bx = 0; bx < k.bw; ++bx) {
                    const int16_t* blk = k.coef.data() + ((size_t)by * k.bw + bx) * 64;
                    uint8_t* o = k.plane.data() + (size_t)by * 8 * pitch + (size_t)bx * 8;
                    if (flavor == JpegFlavor::LibJpeg) idct_islow(blk, k.quant, o, pitch);
                    else idct_stb(blk, k.quant, o, pitch);
                }
            std::vector<int16_t>().swap(k.coef);
            upsample(k, full[c]);
            std::vector<uint8_t>().swap(k.plane);
        }
        img.width = W;
        img.height = H;
        img.rgba.assign((size_t)W * H * 4, 255);
        // three components are YCbCr unless, in a file without a JFIF marker, an Adobe marker says "no transform"
        // or, without either marker, the components are named R, G, B
        // (stb_image takes components named R, G, B as RGB whatever the markers say)
        const bool named_rgb = comps.size() == 3 && comps[0].id == 'R' && comps[1].id == 'G' && comps[2].id == 'B';
        const bool rgb = comps.size() == 3 && ((adobe_transform == 0 && !jfif) || (named_rgb && (flavor == JpegFlavor::Stb || (adobe_transform < 0 && !jfif))));
        for (int y = 0; y < H; ++y) {
            uint8_t* o = img.rgba.data() + (size_t)y * W * 4;
            if (comps.size() == 1) {
                const uint8_t* g = full[0].data() + (size_t)y * comps[0].dw * (hmax / comps[0].h);
                for (int x = 0; x < W; ++x, o += 4) o[0] = o[1] = o[2] = g[x];
                continue;
            }
            const uint8_t* a = full[0].data() + (size_t)y * comps[0].dw * (hmax / comps[0].h);
            const uint8_t* b = full[1].data() + (size_t)y * comps[1].dw * (hmax / comps[1].h);
            const uint8_t* c = full[2].data() + (size_t)y * comps[2].dw * (hmax / comps[2].h);
            for (int x = 0; x < W; ++x, o += 4) {
                if (rgb) { o[0] = a[x]; o[1] = b[x]; o[2] = c[x]; }
                else if (flavor == JpegFlavor::LibJpeg) ycc_libjpeg(a[x], b[x], c[x], o);
                else ycc_stb(a[x], b[x], c[x], o);
            }
        }
        return true;
    }

    bool run(Image& img)
    {
        if (size < 4 || data[0] != 0xFF || data[1] != 0xD8) return fail("not a JPEG file");
        size_t pos = 2;
        for (;;) {
            while (pos < size && data[pos] != 0xFF) ++pos;       // to the next marker
            while (pos < size && data[pos] == 0xFF) ++pos;       // any number of fill bytes
            if (pos >= size) break;                              // no end-of-image marker: what was decoded counts
            const int m = data[pos++];
            if (m == 0xD9) break;
            if (m == 0x00 || m == 0x01 || (m >= 0xD0 && m <= 0xD8)) continue;   // stuffed byte or a marker without a body
            if (pos + 2 > size) break;
            const size_t len = ((size_t)data[pos] << 8) | data[pos + 1];
            if (len < 2 || pos + len > size) return fail("truncated JPEG segment");
            const uint8_t* b = data + pos + 2;
            const size_t blen = len - 2;
            pos += len;
            switch (m) {
            case 0xDB: if (!read_dqt(b, blen)) return false; break;
            case 0xC4: if (!read_dht(b, blen)) return false; break;
            case 0xC0: case 0xC1: if (!read_sof(b, blen, false)) return false; break;
            case 0xC2: if (!read_sof(b, blen, true)) return false; break;
            case 0xC3: case 0xC5: case 0xC6: case 0xC7: case 0xC9: case 0xCA: case 0xCB: case 0xCD: case 0xCE: case 0xCF:
                return fail("lossless, hierarchical and arithmetic-coded JPEG is not supported");
            case 0xDD:
                if (blen < 2) return fail("truncated JPEG restart interval");
                restart_interval = (b[0] << 8) | b[1];
                break;
            case 0xDA: if (!read_scan(b, blen, pos)) return false; break;
            case 0xE0: if (blen >= 5 && !std::memcmp(b, "JFIF", 5)) jfif = true; break;
            case 0xEE: if (blen >= 12 && !std::memcmp(b, "Adobe", 5)) adobe_transform = b[11]; break;
            default: break;
            }
        }
        return finish(img);
    }
};

}  // namespace

bool decode_jpeg(const uint8_t* data, size_t size, Image& img, std::string& err, JpegFlavor flavor)
{
    Decoder d(data, size, err, flavor);
    return d.run(img);
}

}  // namespace sr_assets
