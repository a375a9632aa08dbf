// pt_image.cpp -- image write-out of the headless driver: the reference's end-of-render path
// (ref: src/main.cpp:116-141 x-flip + gamma 1.0; src/image.cpp:41-88 clamp(v*255, 0, 255) truncation and
// 24-bit BMP through stb).  Post-render host I/O, not accelerated.
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <vector>

#include "../../include/pt_abi.h"

namespace {

unsigned char quantize(float v)
{
    // image::applyGamma with gamma 1.0 / divisor 1.0 is pow(f/1.0f, 1.0f) == f; then clamp(f*255, 0, 255)
    float s = v * 255.0f;
    if (s < 0.0f) s = 0.0f;
    else if (s > 255.0f) s = 255.0f;
    return (unsigned char)s;
}

void put_u16(unsigned char *p, unsigned v) { p[0] = v & 255; p[1] = (v >> 8) & 255; }
void put_u32(unsigned char *p, uint32_t v) { p[0] = v & 255; p[1] = (v >> 8) & 255; p[2] = (v >> 16) & 255; p[3] = (v >> 24) & 255; }

}  // namespace

extern "C" {

// 8-bit RGB, row 0 = top of the picture, with the reference's horizontal flip:
// buffer (x, y) -> picture (W-1-x, y)   (ref: src/main.cpp:120-125)
int pt_image_to_rgb8(const float *host_rgb, int W, int H, int flip_x, unsigned char *rgb8_out)
{
    if (!host_rgb || !rgb8_out || W < 1 || H < 1) return PT_ERR_INVALID;
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            const float *src = host_rgb + 3 * ((size_t)x + (size_t)y * (size_t)W);
            const int ox = flip_x ? (W - 1 - x) : x;
            unsigned char *dst = rgb8_out + 3 * ((size_t)ox + (size_t)y * (size_t)W);
            dst[0] = quantize(src[0]);
            dst[1] = quantize(src[1]);
            dst[2] = quantize(src[2]);
        }
    return PT_OK;
}

int pt_save_image_bmp(const char *path, const float *host_rgb, int W, int H, int flip_x)
{
    if (!path) return PT_ERR_INVALID;
    std::vector<unsigned char> rgb((size_t)W * (size_t)H * 3);
    int rc = pt_image_to_rgb8(host_rgb, W, H, flip_x, rgb.data());
    if (rc != PT_OK) return rc;
    const int pad = (4 - (W * 3) % 4) % 4;
    const uint32_t data_bytes = (uint32_t)((W * 3 + pad) * H);
    unsigned char hdr[54] = {0};
    hdr[0] = 'B'; hdr[1] = 'M';
    put_u32(hdr + 2, 54 + data_bytes);
    put_u32(hdr + 10, 54);
    put_u32(hdr + 14, 40);
    put_u32(hdr + 18, (uint32_t)W);
    put_u32(hdr + 22, (uint32_t)H);       // positive height: rows stored bottom-up
    put_u16(hdr + 26, 1);
    put_u16(hdr + 28, 24);
    put_u32(hdr + 34, data_bytes);
    FILE *f = fopen(path, "wb");
    if (!f) return PT_ERR_INVALID;
    fwrite(hdr, 1, sizeof hdr, f);
    const unsigned char zeros[3] = {0, 0, 0};
    std::vector<unsigned char> row((size_t)W * 3);
    for (int y = H - 1; y >= 0; --y) {
        for (int x = 0; x < W; ++x) {
            const unsigned char *s = rgb.data() + 3 * ((size_t)x + (size_t)y * (size_t)W);
            row[3 * x + 0] = s[2];
            row[3 * x + 1] = s[1];
            row[3 * x + 2] = s[0];
        }
        fwrite(row.data(), 1, row.size(), f);
        if (pad) fwrite(zeros, 1, (size_t)pad, f);
    }
    fclose(f);
    return PT_OK;
}


// PNG, 8-bit RGB, rows top-down (ref: src/image.cpp:86 stbi_write_png(..., xSize*3)).  The zlib stream uses stored
// (uncompressed) deflate blocks: any decoder reads it, and a rendered frame is noise that deflate would not shrink much.
namespace {
uint32_t crc32_update(uint32_t crc, const unsigned char *p, size_t n)
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
    for (size_t i = 0; i < n; ++i) crc = table[(crc ^ p[i]) & 0xFFu] ^ (crc >> 8);
    return crc;
}
void put_be32(std::vector<unsigned char> &v, uint32_t x)
{
    v.push_back((unsigned char)(x >> 24)); v.push_back((unsigned char)(x >> 16));
    v.push_back((unsigned char)(x >> 8)); v.push_back((unsigned char)x);
}
void write_chunk(FILE *f, const char type[4], const std::vector<unsigned char> &data)
{
    std::vector<unsigned char> head;
    put_be32(head, (uint32_t)data.size());
    fwrite(head.data(), 1, 4, f);
    fwrite(type, 1, 4, f);
    if (!data.empty()) fwrite(data.data(), 1, data.size(), f);
    uint32_t crc = crc32_update(0xFFFFFFFFu, reinterpret_cast<const unsigned char *>(type), 4);
    if (!data.empty()) crc = crc32_update(crc, data.data(), data.size());
    std::vector<unsigned char> tail;
    put_be32(tail, crc ^ 0xFFFFFFFFu);
    fwrite(tail.data(), 1, 4, f);
}
}  // namespace

int pt_save_image_png(const char *path, const float *host_rgb, int W, int H, int flip_x)
{
    if (!path || W < 1 || H < 1) return PT_ERR_INVALID;
    std::vector<unsigned char> rgb((size_t)W * (size_t)H * 3);
    int rc = pt_image_to_rgb8(host_rgb, W, H, flip_x, rgb.data());
    if (rc != PT_OK) return rc;
    // raw scanlines, each preceded by filter type 0
    std::vector<unsigned char> raw;
    raw.reserve(((size_t)W * 3 + 1) * (size_t)H);
    for (int y = 0; y < H; ++y) {
        raw.push_back(0);
        raw.insert(raw.end(), rgb.begin() + (long)((size_t)y * (size_t)W * 3), rgb.begin() + (long)(((size_t)y + 1) * (size_t)W * 3));
    }
    // zlib container: header, stored blocks of <= 65535 bytes, Adler-32 of the raw data
    std::vector<unsigned char> z;
    z.push_back(0x78); z.push_back(0x01);
    uint32_t a = 1, b = 0;
    for (size_t pos = 0; pos < raw.size();) {
        const size_t n = raw.size() - pos < 65535 ? raw.size() - pos : 65535;
        z.push_back(pos + n == raw.size() ? 1 : 0);              // BFINAL, BTYPE = 00
        z.push_back((unsigned char)(n & 0xFF)); z.push_back((unsigned char)(n >> 8));
        z.push_back((unsigned char)(~n & 0xFF)); z.push_back((unsigned char)((~n >> 8) & 0xFF));
        z.insert(z.end(), raw.begin() + (long)pos, raw.begin() + (long)(pos + n));
        for (size_t i = pos; i < pos + n; ++i) { a = (a + raw[i]) % 65521u; b = (b + a) % 65521u; }
        pos += n;
    }
    put_be32(z, (b << 16) | a);
    FILE *f = fopen(path, "wb");
    if (!f) return PT_ERR_INVALID;
    static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    fwrite(sig, 1, 8, f);
    std::vector<unsigned char> ihdr;
    put_be32(ihdr, (uint32_t)W); put_be32(ihdr, (uint32_t)H);
    ihdr.push_back(8); ihdr.push_back(2); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);   // 8-bit RGB, no interlace
    write_chunk(f, "IHDR", ihdr);
    write_chunk(f, "IDAT", z);
    write_chunk(f, "IEND", std::vector<unsigned char>());
    const bool ok = fclose(f) == 0;
    return ok ? PT_OK : PT_ERR_INVALID;
}

// the reference's choice of container (ref: src/image.cpp:68-87): BMP when the name ends in "bmp", PNG otherwise
int pt_save_image(const char *path, const float *host_rgb, int W, int H, int flip_x)
{
    if (!path) return PT_ERR_INVALID;
    const size_t n = strlen(path);
    if (n >= 3 && path[n - 3] == 'b' && path[n - 2] == 'm' && path[n - 1] == 'p') return pt_save_image_bmp(path, host_rgb, W, H, flip_x);
    return pt_save_image_png(path, host_rgb, W, H, flip_x);
}

}  // extern "C"
