// sm_png.h -- minimal PNG writer (8-bit grey or RGB, stored/uncompressed deflate blocks) so that
// SurfelMapping::acquireImages can write image/%06d.png and semantic/%06d.png like the reference does with
// cv::imwrite (src/SurfelMapping.cpp:408-422) without OpenCV, libpng or zlib.
#pragma once
#include <cstdint>
#include <cstdio>
#include <vector>

namespace sm_png {

inline uint32_t crc32(const uint8_t *p, size_t n, uint32_t crc = 0)
{
    static uint32_t table[256];
    static bool init = false;
    if (!init) {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            table[i] = c;
        }
        init = true;
    }
    crc = ~crc;
    for (size_t i = 0; i < n; ++i) crc = table[(crc ^ p[i]) & 0xFFu] ^ (crc >> 8);
    return ~crc;
}

inline void put32(std::vector<uint8_t> &v, uint32_t x)
{
    v.push_back((uint8_t)(x >> 24)); v.push_back((uint8_t)(x >> 16)); v.push_back((uint8_t)(x >> 8)); v.push_back((uint8_t)x);
}

inline void chunk(std::vector<uint8_t> &out, const char type[4], const std::vector<uint8_t> &data)
{
    put32(out, (uint32_t)data.size());
    const size_t start = out.size();
    out.insert(out.end(), type, type + 4);
    out.insert(out.end(), data.begin(), data.end());
    put32(out, crc32(out.data() + start, out.size() - start));
}

// pixels: h rows of w*channels bytes; channels = 1 (grey) or 3 (RGB, in the byte order given)
inline bool write(const char *path, const uint8_t *pixels, int w, int h, int channels)
{
    if (w <= 0 || h <= 0 || (channels != 1 && channels != 3)) return false;
    std::vector<uint8_t> raw;
    raw.reserve((size_t)h * ((size_t)w * channels + 1));
    for (int y = 0; y < h; ++y) {
        raw.push_back(0);                                      // filter type 0
        raw.insert(raw.end(), pixels + (size_t)y * w * channels, pixels + (size_t)(y + 1) * w * channels);
    }
    std::vector<uint8_t> z;
    z.push_back(0x78); z.push_back(0x01);                      // zlib header, no compression
    uint32_t a = 1, b = 0;                                     // adler32
    size_t pos = 0;
    while (pos < raw.size() || raw.empty()) {
        const size_t n = std::min<size_t>(65535, raw.size() - pos);
        const bool last = pos + n >= raw.size();
        z.push_back(last ? 1 : 0);
        z.push_back((uint8_t)(n & 0xFF)); z.push_back((uint8_t)(n >> 8));
        z.push_back((uint8_t)(~n & 0xFF)); z.push_back((uint8_t)((~n >> 8) & 0xFF));
        z.insert(z.end(), raw.begin() + pos, raw.begin() + pos + n);
        for (size_t i = 0; i < n; ++i) { a = (a + raw[pos + i]) % 65521u; b = (b + a) % 65521u; }
        pos += n;
        if (last) break;
    }
    put32(z, (b << 16) | a);
    std::vector<uint8_t> out = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    std::vector<uint8_t> ihdr;
    put32(ihdr, (uint32_t)w); put32(ihdr, (uint32_t)h);
    ihdr.push_back(8); ihdr.push_back(channels == 3 ? 2 : 0); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);
    chunk(out, "IHDR", ihdr);
    chunk(out, "IDAT", z);
    chunk(out, "IEND", {});
    FILE *f = std::fopen(path, "wb");
    if (!f) return false;
    const bool ok = std::fwrite(out.data(), 1, out.size(), f) == out.size();
    return (std::fclose(f) == 0) && ok;
}

}  // namespace sm_png
