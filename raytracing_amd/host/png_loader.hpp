// png_loader.hpp — minimal PNG -> RGB8 decoder for the texture loader.
//
// Mirrors what the reference gets from `ImageReader::open(path).decode()` followed by
// `get_pixel(x, y).to_rgb()` (reference src/texture.rs:141-161) for the PNG kinds its asset
// directory holds: 8-bit grayscale, 1/2/4/8-bit palette, 8-bit RGB, 8-bit gray+alpha, 8-bit RGBA,
// non-interlaced.  `to_rgb()` drops alpha, replicates gray and expands palette indices; no gamma,
// no premultiplication.  libpng is not in the image, zlib is: inflate comes from zlib, chunk
// parsing and scanline unfiltering are done here.
#pragma once

#include <zlib.h>

#include <cstdint>
#include <cstdlib>
#include <fstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "raytracer.hpp"

namespace raytracer {

struct PngImage {
    uint32_t width = 0, height = 0;
    std::vector<uint8_t> rgb;  // width*height*3, row-major
};

inline uint32_t png_be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | (uint32_t)p[3]; }

inline PngImage png_decode(const std::vector<uint8_t>& file, const std::string& name = "<memory>")
{
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    auto bad = [&](const std::string& why) { return std::runtime_error("png " + name + ": " + why); };
    if (file.size() < 8 || std::memcmp(file.data(), sig, 8) != 0) throw bad("not a PNG file");
    uint32_t w = 0, h = 0;
    int depth = 0, ctype = -1, interlace = 0;
    std::vector<uint8_t> idat, plte;
    bool have_ihdr = false, have_iend = false;
    size_t pos = 8;
    while (pos + 12 <= file.size() && !have_iend) {
        uint32_t len = png_be32(&file[pos]);
        if (pos + 12 + (size_t)len > file.size()) throw bad("truncated chunk");
        const uint8_t* type = &file[pos + 4];
        const uint8_t* data = &file[pos + 8];
        uint32_t crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), type, 4 + len);
        if (crc != png_be32(&file[pos + 8 + len])) throw bad("chunk CRC mismatch");
        if (std::memcmp(type, "IHDR", 4) == 0) {
            if (len != 13) throw bad("bad IHDR");
            w = png_be32(data);
            h = png_be32(data + 4);
            depth = data[8];
            ctype = data[9];
            if (data[10] != 0 || data[11] != 0) throw bad("unknown compression/filter method");
            interlace = data[12];
            have_ihdr = true;
        } else if (std::memcmp(type, "PLTE", 4) == 0) {
            plte.assign(data, data + len);
        } else if (std::memcmp(type, "IDAT", 4) == 0) {
            idat.insert(idat.end(), data, data + len);
        } else if (std::memcmp(type, "IEND", 4) == 0) {
            have_iend = true;
        }
        pos += 12 + (size_t)len;
    }
    if (!have_ihdr || !have_iend) throw bad("missing IHDR or IEND");
    if (w == 0 || h == 0 || w > 16384 || h > 16384) throw bad("unsupported dimensions");
    if (interlace != 0) throw bad("interlaced PNG not supported");
    int channels;
    switch (ctype) {
    case 0: channels = 1; if (depth != 1 && depth != 2 && depth != 4 && depth != 8) throw bad("unsupported gray depth"); break;
    case 2: channels = 3; if (depth != 8) throw bad("unsupported RGB depth"); break;
    case 3: channels = 1; if (depth != 1 && depth != 2 && depth != 4 && depth != 8) throw bad("unsupported palette depth"); if (plte.empty()) throw bad("palette image without PLTE"); break;
    case 4: channels = 2; if (depth != 8) throw bad("unsupported gray+alpha depth"); break;
    case 6: channels = 4; if (depth != 8) throw bad("unsupported RGBA depth"); break;
    default: throw bad("unknown colour type");
    }
    const size_t bits_pp = (size_t)channels * (size_t)depth;
    const size_t stride = ((size_t)w * bits_pp + 7) / 8;
    const size_t bpp = bits_pp >= 8 ? bits_pp / 8 : 1;  // filter byte distance
    std::vector<uint8_t> raw((stride + 1) * (size_t)h);
    uLongf raw_len = (uLongf)raw.size();
    int zr = uncompress(raw.data(), &raw_len, idat.data(), (uLong)idat.size());
    if (zr != Z_OK || raw_len != raw.size()) throw bad("inflate failed or wrong decompressed size");

    std::vector<uint8_t> prev(stride, 0), cur(stride, 0);
    PngImage img;
    img.width = w;
    img.height = h;
    img.rgb.resize((size_t)w * h * 3);
    for (uint32_t y = 0; y < h; ++y) {
        const uint8_t* line = &raw[(stride + 1) * (size_t)y];
        const uint8_t ft = line[0];
        for (size_t i = 0; i < stride; ++i) {
            const int x = line[1 + i];
            const int a = i >= bpp ? cur[i - bpp] : 0;
            const int b = prev[i];
            const int c = i >= bpp ? prev[i - bpp] : 0;
            int v;
            switch (ft) {
            case 0: v = x; break;
            case 1: v = x + a; break;
            case 2: v = x + b; break;
            case 3: v = x + ((a + b) >> 1); break;
            case 4: {
                int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
                int pr = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
                v = x + pr;
                break;
            }
            default: throw bad("unknown scanline filter");
            }
            cur[i] = (uint8_t)v;
        }
        uint8_t* out = &img.rgb[(size_t)y * w * 3];
        for (uint32_t x = 0; x < w; ++x) {
            uint8_t r, g, bl;
            if (ctype == 2) { r = cur[3 * x]; g = cur[3 * x + 1]; bl = cur[3 * x + 2]; }
            else if (ctype == 6) { r = cur[4 * x]; g = cur[4 * x + 1]; bl = cur[4 * x + 2]; }
            else if (ctype == 4) { r = g = bl = cur[2 * x]; }
            else {
                // 1 sample per pixel, `depth` bits, MSB first
                uint32_t s;
                if (depth == 8) s = cur[x];
                else {
                    const size_t bit = (size_t)x * (size_t)depth;
                    s = (cur[bit >> 3] >> (8 - depth - (int)(bit & 7))) & ((1u << depth) - 1u);
                }
                if (ctype == 3) {
                    if ((size_t)s * 3 + 2 >= plte.size()) throw bad("palette index out of range");
                    r = plte[3 * s]; g = plte[3 * s + 1]; bl = plte[3 * s + 2];
                } else {
                    // gray of depth < 8 scales to 8 bits by bit replication (s * 255 / max)
                    const uint32_t mx = (1u << depth) - 1u;
                    r = g = bl = (uint8_t)(depth == 8 ? s : (s * 255u) / mx);
                }
            }
            out[3 * x] = r; out[3 * x + 1] = g; out[3 * x + 2] = bl;
        }
        prev.swap(cur);
    }
    return img;
}

inline PngImage png_decode_file(const std::string& path)
{
    std::ifstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("cannot open " + path);
    std::vector<uint8_t> buf((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    return png_decode(buf, path);
}

// Texture::new (reference src/texture.rs:141-169)
inline Texture load_texture_png(const std::string& file_path, size_t sprite_size)
{
    PngImage img = png_decode_file(file_path);
    return Texture::from_rgb8(img.rgb.data(), img.width, img.height, sprite_size);
}

// GameTextures::new (reference src/texture.rs:48-80): `asset_dir` is a plain string prefix, so it
// must end with '/' exactly as in the reference.
inline GameTextures load_game_textures_png(const std::string& asset_dir)
{
    GameTextures g;
    for (int i = 0; i < kTextureCount; ++i) g.tex[i] = load_texture_png(asset_dir + texture_file_stem((Textures)i) + ".png", 16);
    return g;
}

}  // namespace raytracer
