// Minimal PNG writer (8-bit RGB, zlib "stored" blocks: no compression, no dependency) for the driver's
// saveDir/image/<t>.png (the reference writes CirclesEventFrame::image() with cv::imwrite,
// event_camera_calib/test/eventCameraCalib.cpp:214-227).
#ifndef ECAL_HOST_PNG_WRITER_H_
#define ECAL_HOST_PNG_WRITER_H_

#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

namespace ecal_host {

inline uint32_t png_crc(const uint8_t *p, size_t n, uint32_t crc = 0xFFFFFFFFu) {
    static uint32_t table[256];
    static bool made = false;
    if (!made) {
        for (uint32_t i = 0; i < 256; i++) {
            uint32_t c = i;
            for (int k = 0; k < 8; k++) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            table[i] = c;
        }
        made = true;
    }
    for (size_t i = 0; i < n; i++) crc = table[(crc ^ p[i]) & 0xFFu] ^ (crc >> 8);
    return crc;
}

// rgb: height rows of width * 3 bytes.  Returns false when the file cannot be written.
inline bool write_png_rgb(const std::string &path, int width, int height, const std::vector<uint8_t> &rgb) {
    if (width <= 0 || height <= 0 || rgb.size() < (size_t) width * height * 3) return false;
    std::vector<uint8_t> raw;   // filter byte 0 + the row
    raw.reserve((size_t) height * (1 + 3 * (size_t) width));
    for (int y = 0; y < height; y++) {
        raw.push_back(0);
        raw.insert(raw.end(), rgb.begin() + (size_t) y * width * 3, rgb.begin() + (size_t) (y + 1) * width * 3);
    }
    std::vector<uint8_t> z = {0x78, 0x01};   // zlib header, then stored blocks of <= 65535 bytes
    uint32_t a = 1, b = 0;                   // Adler-32 of the raw data
    for (size_t at = 0; at < raw.size() || at == 0;) {
        const size_t n = raw.size() - at < 65535 ? raw.size() - at : 65535;
        z.push_back(at + n >= raw.size() ? 1 : 0);
        z.push_back((uint8_t) (n & 0xFF));
        z.push_back((uint8_t) (n >> 8));
        z.push_back((uint8_t) (~n & 0xFF));
        z.push_back((uint8_t) ((~n >> 8) & 0xFF));
        for (size_t i = 0; i < n; i++) {
            a = (a + raw[at + i]) % 65521u;
            b = (b + a) % 65521u;
        }
        z.insert(z.end(), raw.begin() + at, raw.begin() + at + n);
        at += n;
        if (n == 0) break;
    }
    const uint32_t adler = (b << 16) | a;
    for (int k = 3; k >= 0; k--) z.push_back((uint8_t) (adler >> (8 * k)));
    FILE *f = std::fopen(path.c_str(), "wb");
    if (!f) return false;
    auto be32 = [](uint32_t v, uint8_t *o) {
        o[0] = (uint8_t) (v >> 24), o[1] = (uint8_t) (v >> 16), o[2] = (uint8_t) (v >> 8), o[3] = (uint8_t) v;
    };
    auto chunk = [&](const char *type, const uint8_t *data, size_t n) {
        uint8_t hdr[8];
        be32((uint32_t) n, hdr);
        for (int i = 0; i < 4; i++) hdr[4 + i] = (uint8_t) type[i];
        std::fwrite(hdr, 1, 8, f);
        if (n) std::fwrite(data, 1, n, f);
        uint32_t crc = png_crc(hdr + 4, 4);
        crc = png_crc(data, n, crc) ^ 0xFFFFFFFFu;
        uint8_t c[4];
        be32(crc, c);
        std::fwrite(c, 1, 4, f);
    };
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    std::fwrite(sig, 1, 8, f);
    uint8_t ihdr[13];
    be32((uint32_t) width, ihdr);
    be32((uint32_t) height, ihdr + 4);
    ihdr[8] = 8, ihdr[9] = 2, ihdr[10] = 0, ihdr[11] = 0, ihdr[12] = 0;   // 8 bit, RGB
    chunk("IHDR", ihdr, 13);
    chunk("IDAT", z.data(), z.size());
    chunk("IEND", nullptr, 0);
    return std::fclose(f) == 0;
}

}  // namespace ecal_host
#endif
