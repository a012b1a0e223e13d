// Output side of capture_image: From<Vec3> for Rgb<u8> (vec3.rs:223-231) and
// RgbImage::save (main.rs:55) as a minimal PNG writer (stored deflate blocks).
#include <cmath>
#include <cstdio>
#include <cstring>

#include "scene.h"

namespace rtamd {

// floor(clamp(sqrt(c), 0, 1) * 255) as u8 ; f64::clamp keeps NaN, `as u8` saturates and maps NaN to 0 (Q13)
uint8_t tonemap_channel(double c) {
    double s = std::sqrt(c);
    if (s < 0.) s = 0.;
    else if (s > 1.) s = 1.;
    double f = std::floor(s * 255.);
    if (!(f == f)) return 0;
    if (f <= 0.) return 0;
    if (f >= 255.) return 255;
    return (uint8_t)f;
}

static uint32_t crc_table[256];
static bool crc_ready = false;
static uint32_t crc32(uint32_t c, const uint8_t* p, size_t n) {
    if (!crc_ready) {
        for (uint32_t i = 0; i < 256; i++) {
            uint32_t k = i;
            for (int j = 0; j < 8; j++) k = (k & 1) ? (0xEDB88320u ^ (k >> 1)) : (k >> 1);
            crc_table[i] = k;
        }
        crc_ready = true;
    }
    c = ~c;
    for (size_t i = 0; i < n; i++) c = crc_table[(c ^ p[i]) & 0xFF] ^ (c >> 8);
    return ~c;
}
static void be32(std::vector<uint8_t>& v, uint32_t x) {
    v.push_back((uint8_t)(x >> 24)); v.push_back((uint8_t)(x >> 16)); v.push_back((uint8_t)(x >> 8)); v.push_back((uint8_t)x);
}
static void chunk(std::vector<uint8_t>& out, const char* tag, const std::vector<uint8_t>& data) {
    be32(out, (uint32_t)data.size());
    std::vector<uint8_t> td(tag, tag + 4);
    td.insert(td.end(), data.begin(), data.end());
    out.insert(out.end(), td.begin(), td.end());
    be32(out, crc32(0, td.data(), td.size()));
}

void write_png(const char* path, int w, int h, const uint8_t* rgb) {
    if (w <= 0 || h <= 0 || !rgb) throw RtError(RT_ERR_ARG, "write_png: empty image");
    std::vector<uint8_t> raw;
    raw.reserve((size_t)h * (w * 3 + 1));
    for (int y = 0; y < h; y++) {
        raw.push_back(0);  // filter: none
        raw.insert(raw.end(), rgb + (size_t)y * w * 3, rgb + (size_t)(y + 1) * w * 3);
    }
    std::vector<uint8_t> z = {0x78, 0x01};
    uint32_t a = 1, b = 0;
    size_t pos = 0;
    while (pos < raw.size() || pos == 0) {
        size_t n = raw.size() - pos;
        if (n > 65535) n = 65535;
        bool last = pos + n >= raw.size();
        z.push_back(last ? 1 : 0);
        z.push_back((uint8_t)(n & 0xFF)); z.push_back((uint8_t)(n >> 8));
        z.push_back((uint8_t)(~n & 0xFF)); z.push_back((uint8_t)((~n >> 8) & 0xFF));
        for (size_t i = 0; i < n; i++) {
            a = (a + raw[pos + i]) % 65521;
            b = (b + a) % 65521;
        }
        z.insert(z.end(), raw.begin() + pos, raw.begin() + pos + n);
        pos += n;
        if (last) break;
    }
    be32(z, (b << 16) | a);
    std::vector<uint8_t> out = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    std::vector<uint8_t> ihdr;
    be32(ihdr, (uint32_t)w);
    be32(ihdr, (uint32_t)h);
    ihdr.push_back(8); ihdr.push_back(2); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);
    chunk(out, "IHDR", ihdr);
    chunk(out, "IDAT", z);
    chunk(out, "IEND", {});
    FILE* f = fopen(path, "wb");
    if (!f) throw RtError(RT_ERR_IO, std::string("cannot write ") + path);
    size_t wr = fwrite(out.data(), 1, out.size(), f);
    fclose(f);
    if (wr != out.size()) throw RtError(RT_ERR_IO, std::string("short write to ") + path);
}

}  // namespace rtamd
