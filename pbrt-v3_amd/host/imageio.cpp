// Image input for ImageTexture (core/imageio.cpp:60-79, :216-440) and the MIPMap pyramid it is turned into
// (core/mipmap.h:93-187, textures/imagemap.cpp:52-101).  Readers written for this front end: PFM, TGA (uncompressed / RLE,
// true-colour, monochrome and colour-mapped) and PNG (zlib inflate + the five scanline filters; what lodepng_decode24
// hands the reference: 8-bit RGB).  OpenEXR is not available (nor is it in the reference build used as the oracle).
#include <zlib.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include "error.h"
#include "scene.h"

namespace pbrt {

static bool HasExtension(const std::string &value, const std::string &ending) {  // fileutil.h
    if (ending.size() > value.size()) return false;
    return std::equal(ending.rbegin(), ending.rend(), value.rbegin(), [](char a, char b) { return std::tolower(a) == std::tolower(b); });
}
static bool readFile(const std::string &name, std::vector<uint8_t> *out) {
    FILE *fp = fopen(name.c_str(), "rb");
    if (!fp) return false;
    fseek(fp, 0, SEEK_END);
    long n = ftell(fp);
    fseek(fp, 0, SEEK_SET);
    out->resize(n > 0 ? (size_t)n : 0);
    bool ok = n >= 0 && fread(out->data(), 1, out->size(), fp) == out->size();
    fclose(fp);
    return ok;
}

// ---- PFM (imageio.cpp:352-431): rows bottom-up in the file, flipped while reading; |scale| multiplies, sign = endianness
static bool ReadImagePFM(const std::string &filename, int *xres, int *yres, std::vector<RGB> *rgb) {
    std::vector<uint8_t> d;
    if (!readFile(filename, &d)) return false;
    size_t pos = 0;
    auto word = [&](std::string *w) {
        w->clear();
        while (pos < d.size()) {
            char c = (char)d[pos++];
            if (c == ' ' || c == '\n' || c == '\t') return true;
            w->push_back(c);
        }
        return false;
    };
    std::string w;
    if (!word(&w)) return false;
    int nChannels = w == "Pf" ? 1 : (w == "PF" ? 3 : 0);
    if (!nChannels) return false;
    if (!word(&w)) return false;
    int width = atoi(w.c_str());
    if (!word(&w)) return false;
    int height = atoi(w.c_str());
    if (!word(&w)) return false;
    float scale = 0;
    sscanf(w.c_str(), "%f", &scale);
    if (width <= 0 || height <= 0) return false;
    size_t nFloats = (size_t)nChannels * width * height;
    if (d.size() - pos < nFloats * 4) return false;
    std::vector<float> data(nFloats);
    for (int y = height - 1; y >= 0; --y) { memcpy(&data[(size_t)y * nChannels * width], &d[pos], (size_t)nChannels * width * 4); pos += (size_t)nChannels * width * 4; }
    const bool fileLittleEndian = scale < 0.f;
    if (!fileLittleEndian)  // the host is little endian
        for (size_t i = 0; i < nFloats; ++i) { uint8_t b[4]; memcpy(b, &data[i], 4); std::swap(b[0], b[3]); std::swap(b[1], b[2]); memcpy(&data[i], b, 4); }
    if (std::abs(scale) != 1.f) for (size_t i = 0; i < nFloats; ++i) data[i] *= std::abs(scale);
    rgb->resize((size_t)width * height);
    for (size_t i = 0; i < rgb->size(); ++i)
        (*rgb)[i] = nChannels == 1 ? RGB{{data[i], data[i], data[i]}} : RGB{{data[3 * i], data[3 * i + 1], data[3 * i + 2]}};
    *xres = width; *yres = height;
    return true;
}

// ---- TGA (imageio.cpp:216-256 over ext/targa.c): 8-bit values / 255, BGR(A) order, top-to-bottom left-to-right result
static bool ReadImageTGA(const std::string &name, int *width, int *height, std::vector<RGB> *rgb) {
    std::vector<uint8_t> d;
    if (!readFile(name, &d) || d.size() < 18) return false;
    const int idLen = d[0], cmapType = d[1], imgType = d[2];
    const int cmapFirst = d[3] | (d[4] << 8), cmapLen = d[5] | (d[6] << 8), cmapDepth = d[7];
    const int w = d[12] | (d[13] << 8), h = d[14] | (d[15] << 8), depth = d[16], desc = d[17];
    const bool rle = imgType == 9 || imgType == 10 || imgType == 11;
    const int base = rle ? imgType - 8 : imgType;  // 1 colour-mapped, 2 true-colour, 3 mono
    if (base < 1 || base > 3 || w <= 0 || h <= 0) return false;
    const int bpp = depth / 8;
    if (bpp < 1 || bpp > 4) return false;
    size_t pos = 18 + idLen;
    std::vector<uint8_t> cmap;
    const int cbpp = cmapDepth / 8;
    if (cmapType == 1) {
        if (pos + (size_t)cmapLen * cbpp > d.size()) return false;
        cmap.assign(d.begin() + pos, d.begin() + pos + (size_t)cmapLen * cbpp);
        pos += (size_t)cmapLen * cbpp;
    }
    std::vector<uint8_t> px((size_t)w * h * bpp);
    if (!rle) {
        if (pos + px.size() > d.size()) return false;
        memcpy(px.data(), &d[pos], px.size());
    } else {
        size_t o = 0;
        while (o < px.size()) {
            if (pos >= d.size()) return false;
            const int c = d[pos++], n = (c & 127) + 1;
            if (c & 128) {
                if (pos + bpp > d.size()) return false;
                for (int k = 0; k < n && o < px.size(); ++k) { memcpy(&px[o], &d[pos], bpp); o += bpp; }
                pos += bpp;
            } else {
                const size_t bytes = (size_t)n * bpp;
                if (pos + bytes > d.size() || o + bytes > px.size()) return false;
                memcpy(&px[o], &d[pos], bytes); o += bytes; pos += bytes;
            }
        }
    }
    const bool rightToLeft = (desc & 0x10) != 0, topToBottom = (desc & 0x20) != 0;
    rgb->resize((size_t)w * h);
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const int sx = rightToLeft ? w - 1 - x : x, sy = topToBottom ? y : h - 1 - y;
            const uint8_t *src = &px[((size_t)sy * w + sx) * bpp];
            uint8_t tmp[4];
            int sb = bpp;
            if (base == 1) {  // tga_color_unmap
                const int idx = (bpp == 1 ? src[0] : (src[0] | (src[1] << 8))) - cmapFirst;
                if (idx < 0 || idx >= cmapLen || cbpp < 3) return false;
                memcpy(tmp, &cmap[(size_t)idx * cbpp], cbpp);
                src = tmp; sb = cbpp;
            }
            RGB c;
            if (base == 3) c = RGB{{src[0] / 255.f, src[0] / 255.f, src[0] / 255.f}};
            else {
                if (sb < 3) return false;  // 15/16-bit true colour: not handled
                c.c[2] = src[0] / 255.f; c.c[1] = src[1] / 255.f; c.c[0] = src[2] / 255.f;
            }
            (*rgb)[(size_t)y * w + x] = c;
        }
    *width = w; *height = h;
    return true;
}

// ---- PNG (imageio.cpp:258-286 over lodepng_decode24_file): 8-bit RGB, alpha dropped, 16-bit samples keep the high byte
static bool ReadImagePNG(const std::string &name, int *width, int *height, std::vector<RGB> *rgb) {
    std::vector<uint8_t> d;
    static const uint8_t sig[8] = {137, 80, 78, 71, 13, 10, 26, 10};
    if (!readFile(name, &d) || d.size() < 8 || memcmp(d.data(), sig, 8) != 0) return false;
    size_t pos = 8;
    uint32_t w = 0, h = 0;
    int depth = 0, ctype = 0, interlace = 0;
    std::vector<uint8_t> idat, plte;
    auto be32 = [&](size_t p) { return ((uint32_t)d[p] << 24) | ((uint32_t)d[p + 1] << 16) | ((uint32_t)d[p + 2] << 8) | d[p + 3]; };
    while (pos + 12 <= d.size()) {
        const uint32_t len = be32(pos);
        const std::string type((const char *)&d[pos + 4], 4);
        if (pos + 12 + len > d.size()) return false;
        const uint8_t *body = &d[pos + 8];
        if (type == "IHDR" && len >= 13) { w = be32(pos + 8); h = be32(pos + 12); depth = body[8]; ctype = body[9]; interlace = body[12]; }
        else if (type == "PLTE") plte.assign(body, body + len);
        else if (type == "IDAT") idat.insert(idat.end(), body, body + len);
        else if (type == "IEND") break;
        pos += 12 + len;
    }
    if (!w || !h || interlace != 0) return false;
    const int channels = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
    if (!channels || (depth != 8 && depth != 16 && !(depth < 8 && (ctype == 0 || ctype == 3)))) return false;
    const size_t bitsPerPixel = (size_t)channels * depth, stride = (w * bitsPerPixel + 7) / 8, bppF = std::max<size_t>(1, bitsPerPixel / 8);
    // the header's size must be plausible for the data that follows (deflate expands by at most ~1032:1): a corrupt IHDR
    // must not turn into a multi-gigabyte allocation
    if (w > (1u << 20) || h > (1u << 20) || (stride + 1) * h > idat.size() * 1100 + 4096) return false;
    std::vector<uint8_t> raw((stride + 1) * h);
    uLongf rawLen = raw.size();
    if (uncompress(raw.data(), &rawLen, idat.data(), idat.size()) != Z_OK || rawLen != raw.size()) return false;
    std::vector<uint8_t> img(stride * h);
    for (uint32_t y = 0; y < h; ++y) {  // undo the scanline filters
        const uint8_t ft = raw[(stride + 1) * y];
        const uint8_t *in = &raw[(stride + 1) * y + 1];
        uint8_t *out = &img[stride * y];
        const uint8_t *up = y ? &img[stride * (y - 1)] : nullptr;
        for (size_t i = 0; i < stride; ++i) {
            const int a = i >= bppF ? out[i - bppF] : 0, b = up ? up[i] : 0, c = (up && i >= bppF) ? up[i - bppF] : 0;
            int v = in[i];
            if (ft == 1) v += a;
            else if (ft == 2) v += b;
            else if (ft == 3) v += (a + b) / 2;
            else if (ft == 4) { const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c); v += (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); }
            else if (ft != 0) return false;
            out[i] = (uint8_t)v;
        }
    }
    rgb->resize((size_t)w * h);
    for (uint32_t y = 0; y < h; ++y)
        for (uint32_t x = 0; x < w; ++x) {
            uint8_t s[4] = {0, 0, 0, 255};
            const uint8_t *row = &img[stride * y];
            if (depth >= 8) {
                const int bytes = depth / 8;
                for (int c = 0; c < channels; ++c) s[c] = row[((size_t)x * channels + c) * bytes];  // 16-bit: the most significant byte
            } else {
                const int v = (row[(x * depth) / 8] >> (8 - depth - (x * depth) % 8)) & ((1 << depth) - 1);
                s[0] = ctype == 3 ? (uint8_t)v : (uint8_t)(v * 255 / ((1 << depth) - 1));
            }
            uint8_t r, g, b;
            if (ctype == 3) { if ((size_t)s[0] * 3 + 2 >= plte.size()) return false; r = plte[s[0] * 3]; g = plte[s[0] * 3 + 1]; b = plte[s[0] * 3 + 2]; }
            else if (ctype == 0 || ctype == 4) r = g = b = s[0];
            else { r = s[0]; g = s[1]; b = s[2]; }
            (*rgb)[(size_t)y * w + x] = RGB{{r / 255.f, g / 255.f, b / 255.f}};
        }
    *width = (int)w; *height = (int)h;
    return true;
}

bool ReadImage(const std::string &name, int *xres, int *yres, std::vector<RGB> *rgb) try {  // imageio.cpp:60-79
    bool ok = false;
    if (HasExtension(name, ".tga")) { ok = ReadImageTGA(name, xres, yres, rgb); if (!ok) Error("Unable to read from TGA file \"%s\"", name.c_str()); }
    else if (HasExtension(name, ".png")) { ok = ReadImagePNG(name, xres, yres, rgb); if (!ok) Error("Error reading PNG \"%s\"", name.c_str()); }
    else if (HasExtension(name, ".pfm")) { ok = ReadImagePFM(name, xres, yres, rgb); if (!ok) Error("Error reading PFM file \"%s\"", name.c_str()); }
    else if (HasExtension(name, ".exr")) Error("Unable to read \"%s\": OpenEXR is not available in this build.", name.c_str());
    else Error("Unable to load image stored in format \"%s\" for filename \"%s\".", strrchr(name.c_str(), '.') ? (strrchr(name.c_str(), '.') + 1) : "(unknown)", name.c_str());
    return ok;
} catch (const std::bad_alloc &) {  // no exception leaves the library (it is driven through a C ABI)
    Error("Out of memory reading image \"%s\"", name.c_str());
    return false;
}
bool ImageGammaDefault(const std::string &filename) { return HasExtension(filename, ".tga") || HasExtension(filename, ".png"); }

// ---- MIPMap<T> construction (mipmap.h:93-187), T = Float (nc = 1) or RGBSpectrum (nc = 3) ------------------------------
static Float Lanczos(Float x, Float tau) {  // texture.cpp:254-262
    x = std::abs(x);
    if (x < 1e-5f) return 1;
    if (x > 1.f) return 0;
    x *= Pi;
    Float s = std::sin(x * tau) / (x * tau);
    Float lanczos = std::sin(x) / x;
    return s * lanczos;
}
static inline int Mod(int a, int b) { int result = a - (a / b) * b; return (int)((result < 0) ? result + b : result); }  // pbrt.h:314-317
struct ResampleWeight { int firstTexel; Float weight[4]; };
static std::vector<ResampleWeight> resampleWeights(int oldRes, int newRes) {  // mipmap.h:76-92
    std::vector<ResampleWeight> wt(newRes);
    Float filterwidth = 2.f;
    for (int i = 0; i < newRes; ++i) {
        Float center = (i + .5f) * oldRes / newRes;
        wt[i].firstTexel = std::floor((center - filterwidth) + 0.5f);
        for (int j = 0; j < 4; ++j) {
            Float pos = wt[i].firstTexel + j + .5f;
            wt[i].weight[j] = Lanczos((pos - center) / filterwidth, 2);
        }
        Float invSumWts = 1 / (wt[i].weight[0] + wt[i].weight[1] + wt[i].weight[2] + wt[i].weight[3]);
        for (int j = 0; j < 4; ++j) wt[i].weight[j] *= invSumWts;
    }
    return wt;
}
// Builds the pyramid into `pool` (appending) and fills img's level offsets.  data: nc floats per texel, row-major, the image
// already flipped / converted by the caller (imagemap.cpp:70-91).
void BuildMIPMap(int resX, int resY, int nc, const std::vector<float> &data, int wrap, PgImage *img, std::vector<float> *pool) {
    int res[2] = {resX, resY};
    std::vector<float> resampled;
    const float *level0 = data.data();
    auto isPow2 = [](int v) { return v && !(v & (v - 1)); };
    auto roundUpPow2 = [](int32_t v) { v--; v |= v >> 1; v |= v >> 2; v |= v >> 4; v |= v >> 8; v |= v >> 16; return v + 1; };
    if (!isPow2(res[0]) || !isPow2(res[1])) {
        const int p2[2] = {roundUpPow2(res[0]), roundUpPow2(res[1])};
        std::vector<ResampleWeight> sWeights = resampleWeights(res[0], p2[0]);
        resampled.assign((size_t)p2[0] * p2[1] * nc, 0.f);
        for (int t = 0; t < res[1]; ++t)  // resample in s
            for (int s = 0; s < p2[0]; ++s)
                for (int c = 0; c < nc; ++c) {
                    float acc = 0.f;
                    for (int j = 0; j < 4; ++j) {
                        int origS = sWeights[s].firstTexel + j;
                        if (wrap == 0) origS = Mod(origS, res[0]);
                        else if (wrap == 2) origS = std::min(std::max(origS, 0), res[0] - 1);
                        if (origS >= 0 && origS < res[0]) acc += sWeights[s].weight[j] * data[((size_t)t * res[0] + origS) * nc + c];
                    }
                    resampled[((size_t)t * p2[0] + s) * nc + c] = acc;
                }
        std::vector<ResampleWeight> tWeights = resampleWeights(res[1], p2[1]);
        std::vector<float> work((size_t)p2[1] * nc);
        for (int s = 0; s < p2[0]; ++s) {  // resample in t
            for (int t = 0; t < p2[1]; ++t)
                for (int c = 0; c < nc; ++c) {
                    float acc = 0.f;
                    for (int j = 0; j < 4; ++j) {
                        int offset = tWeights[t].firstTexel + j;
                        if (wrap == 0) offset = Mod(offset, res[1]);
                        else if (wrap == 2) offset = std::min(std::max(offset, 0), res[1] - 1);
                        if (offset >= 0 && offset < res[1]) acc += tWeights[t].weight[j] * resampled[((size_t)offset * p2[0] + s) * nc + c];
                    }
                    work[(size_t)t * nc + c] = acc;
                }
            for (int t = 0; t < p2[1]; ++t)
                for (int c = 0; c < nc; ++c) { float v = work[(size_t)t * nc + c]; resampled[((size_t)t * p2[0] + s) * nc + c] = v < 0.f ? 0.f : v; }  // clamp(v, 0, Infinity)
        }
        res[0] = p2[0]; res[1] = p2[1];
        level0 = resampled.data();
    }
    int nLevels = 1 + (31 - __builtin_clz((uint32_t)std::max(res[0], res[1])));
    static_assert(PG_MAX_MIP_LEVELS >= 32, "PgImage::level_offset holds the pyramid of any int resolution");
    img->n_levels = nLevels; img->width = res[0]; img->height = res[1];
    int sPrev = res[0], tPrev = res[1];
    size_t prevOff = pool->size();
    img->level_offset[0] = (int64_t)prevOff;
    pool->insert(pool->end(), level0, level0 + (size_t)res[0] * res[1] * nc);
    auto texel = [&](size_t off, int sRes, int tRes, int s, int t, int c) -> float {  // MIPMap::Texel, mipmap.h:189-212
        if (wrap == 0) { s = Mod(s, sRes); t = Mod(t, tRes); }
        else if (wrap == 2) { s = std::min(std::max(s, 0), sRes - 1); t = std::min(std::max(t, 0), tRes - 1); }
        else if (s < 0 || s >= sRes || t < 0 || t >= tRes) return 0.f;
        return (*pool)[off + ((size_t)t * sRes + s) * nc + c];
    };
    for (int i = 1; i < nLevels; ++i) {
        const int sRes = std::max(1, sPrev / 2), tRes = std::max(1, tPrev / 2);
        const size_t off = pool->size();
        img->level_offset[i] = (int64_t)off;
        pool->resize(off + (size_t)sRes * tRes * nc);
        for (int t = 0; t < tRes; ++t)
            for (int s = 0; s < sRes; ++s)
                for (int c = 0; c < nc; ++c)
                    (*pool)[off + ((size_t)t * sRes + s) * nc + c] =
                        .25f * (texel(prevOff, sPrev, tPrev, 2 * s, 2 * t, c) + texel(prevOff, sPrev, tPrev, 2 * s + 1, 2 * t, c) +
                                texel(prevOff, sPrev, tPrev, 2 * s, 2 * t + 1, c) + texel(prevOff, sPrev, tPrev, 2 * s + 1, 2 * t + 1, c));
        prevOff = off; sPrev = sRes; tPrev = tRes;
    }
}
// MIPMap<T>::Lookup(st, width) on the host (mipmap.h:214-243): InfiniteAreaLight's constructor and Power() use it
static void mipTexel(const PgImage &im, const std::vector<float> &pool, int level, int s, int t, float out[3]) {
    const int nc = im.is_float ? 1 : 3;
    const int sRes = std::max(1, im.width >> level), tRes = std::max(1, im.height >> level);
    out[0] = out[1] = out[2] = 0;
    if (im.wrap == 0) { s = Mod(s, sRes); t = Mod(t, tRes); }
    else if (im.wrap == 2) { s = std::min(std::max(s, 0), sRes - 1); t = std::min(std::max(t, 0), tRes - 1); }
    else if (s < 0 || s >= sRes || t < 0 || t >= tRes) return;
    const float *p = &pool[(size_t)im.level_offset[level] + ((size_t)t * sRes + s) * nc];
    for (int c = 0; c < nc; ++c) out[c] = p[c];
}
static void mipTriangle(const PgImage &im, const std::vector<float> &pool, int level, const float st[2], float out[3]) {
    level = std::min(std::max(level, 0), im.n_levels - 1);
    const int sRes = std::max(1, im.width >> level), tRes = std::max(1, im.height >> level);
    Float s = st[0] * sRes - 0.5f, t = st[1] * tRes - 0.5f;
    int s0 = std::floor(s), t0 = std::floor(t);
    Float ds = s - s0, dt = t - t0;
    float a[3], b[3], c[3], d[3];
    mipTexel(im, pool, level, s0, t0, a); mipTexel(im, pool, level, s0, t0 + 1, b);
    mipTexel(im, pool, level, s0 + 1, t0, c); mipTexel(im, pool, level, s0 + 1, t0 + 1, d);
    for (int k = 0; k < 3; ++k) out[k] = (1 - ds) * (1 - dt) * a[k] + (1 - ds) * dt * b[k] + ds * (1 - dt) * c[k] + ds * dt * d[k];
}
void MIPMapLookup(const PgImage &im, const std::vector<float> &pool, const float st[2], float width, float out[3]) {
    const Float invLog2 = 1.442695040888963387004650940071;
    Float level = im.n_levels - 1 + std::log(std::max(width, (Float)1e-8)) * invLog2;  // Log2, pbrt.h:328-331
    if (level < 0) { mipTriangle(im, pool, 0, st, out); return; }
    if (level >= im.n_levels - 1) { mipTexel(im, pool, im.n_levels - 1, 0, 0, out); return; }
    int iLevel = std::floor(level);
    Float delta = level - iLevel;
    float a[3], b[3];
    mipTriangle(im, pool, iLevel, st, a); mipTriangle(im, pool, iLevel + 1, st, b);
    for (int k = 0; k < 3; ++k) out[k] = (1 - delta) * a[k] + delta * b[k];
}
void EWAWeightLut(float lut[128]) {  // mipmap.h:178-184
    for (int i = 0; i < 128; ++i) {
        Float alpha = 2;
        Float r2 = Float(i) / Float(128 - 1);
        lut[i] = std::exp(-alpha * r2) - std::exp(-alpha);
    }
}

}  // namespace pbrt
