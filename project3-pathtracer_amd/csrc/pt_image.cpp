// pt_image.cpp -- image write-out of the headless driver: the reference's end-of-render path
// (ref: src/main.cpp:116-141 x-flip + gamma 1.0; src/image.cpp:41-88 clamp(v*255, 0, 255) truncation and
// 24-bit BMP through stb).  Post-render host I/O, not accelerated.
#include <math.h>
#include <stdint.h>
#include <stdio.h>

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

}  // extern "C"
