#include "HipRenderer.h"

#include <algorithm>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include <zlib.h>

#define V_LOG_ERROR(msg) fprintf(stderr, "[VHipRenderer][error] %s\n", (msg))
#define V_LOG_WARNING(msg) fprintf(stderr, "[VHipRenderer][warning] %s\n", (msg))

namespace VolumeRaytracer {

VObjectPtr<VTexture2D> VTexture2D::LoadPPM(const std::string& path) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return nullptr;
    auto token = [&](char* buf, size_t n) { /* whitespace-separated header token, '#' comments skipped */
        int c = fgetc(f);
        while (c != EOF) {
            if (c == '#') {
                while (c != EOF && c != '\n') c = fgetc(f);
            } else if (c == ' ' || c == '\t' || c == '\n' || c == '\r') {
                c = fgetc(f);
            } else {
                break;
            }
        }
        size_t i = 0;
        while (c != EOF && c != ' ' && c != '\t' && c != '\n' && c != '\r' && i + 1 < n) {
            buf[i++] = (char)c;
            c = fgetc(f);
        }
        buf[i] = 0;
        return i > 0;
    };
    char magic[8], ws[16], hs[16], ms[16];
    VObjectPtr<VTexture2D> out;
    if (token(magic, sizeof magic) && !strcmp(magic, "P6") && token(ws, sizeof ws) && token(hs, sizeof hs) && token(ms, sizeof ms)) {
        const long w = atol(ws), h = atol(hs), maxv = atol(ms);
        if (w > 0 && h > 0 && w <= 16384 && h <= 16384 && maxv == 255) {
            std::vector<uint8_t> rgb((size_t)w * h * 3), rgba((size_t)w * h * 4);
            if (fread(rgb.data(), 1, rgb.size(), f) == rgb.size()) {
                for (size_t i = 0; i < (size_t)w * h; i++) {
                    rgba[i * 4 + 0] = rgb[i * 3 + 0];
                    rgba[i * 4 + 1] = rgb[i * 3 + 1];
                    rgba[i * 4 + 2] = rgb[i * 3 + 2];
                    rgba[i * 4 + 3] = 255;
                }
                out = std::make_shared<VTexture2D>((size_t)w, (size_t)h, std::move(rgba));
            }
        }
    }
    fclose(f);
    return out;
}

namespace {
uint32_t be32(const unsigned char* p) { return (uint32_t)p[0] << 24 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 8 | (uint32_t)p[3]; }
int paeth(int a, int b, int c) {
    const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}
}  // namespace

VObjectPtr<VTexture2D> VTexture2D::LoadPNG(const std::string& path) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return nullptr;
    std::vector<unsigned char> file;
    unsigned char buf[65536];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) file.insert(file.end(), buf, buf + n);
    fclose(f);
    static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (file.size() < 8 + 25 || memcmp(file.data(), sig, 8) != 0) return nullptr;
    uint32_t w = 0, h = 0;
    int colour = -1;
    std::vector<unsigned char> idat, palette;
    for (size_t at = 8; at + 12 <= file.size();) {
        const uint32_t len = be32(&file[at]);
        const unsigned char* type = &file[at + 4];
        if (at + 12 + (size_t)len > file.size()) return nullptr;
        const unsigned char* data = &file[at + 8];
        if (!memcmp(type, "IHDR", 4) && len >= 13) {
            w = be32(data);
            h = be32(data + 4);
            if (data[8] != 8 || data[10] != 0 || data[11] != 0 || data[12] != 0) return nullptr; /* 8 bit, no interlace */
            colour = data[9];
        } else if (!memcmp(type, "PLTE", 4)) {
            palette.assign(data, data + len);
        } else if (!memcmp(type, "IDAT", 4)) {
            idat.insert(idat.end(), data, data + len);
        } else if (!memcmp(type, "IEND", 4)) {
            break;
        }
        at += 12 + (size_t)len;
    }
    const int channels = colour == 0 ? 1 : colour == 2 ? 3 : colour == 3 ? 1 : colour == 4 ? 2 : colour == 6 ? 4 : 0;
    if (!channels || w == 0 || h == 0 || w > 16384 || h > 16384 || (colour == 3 && palette.size() < 3)) return nullptr;
    const size_t stride = (size_t)w * channels;
    std::vector<unsigned char> raw((stride + 1) * h);
    uLongf rawLen = (uLongf)raw.size();
    if (uncompress(raw.data(), &rawLen, idat.data(), (uLong)idat.size()) != Z_OK || rawLen != raw.size()) return nullptr;
    std::vector<unsigned char> img(stride * h);
    for (uint32_t y = 0; y < h; y++) { /* undo the per-row filters (PNG spec §9) */
        const unsigned char* src = &raw[(stride + 1) * y];
        unsigned char* row = &img[stride * y];
        const unsigned char* up = y ? &img[stride * (y - 1)] : nullptr;
        const int filter = src[0];
        if (filter > 4) return nullptr;
        for (size_t x = 0; x < stride; x++) {
            const int a = x >= (size_t)channels ? row[x - channels] : 0, b = up ? up[x] : 0, c = (up && x >= (size_t)channels) ? up[x - channels] : 0;
            const int pred = filter == 0 ? 0 : filter == 1 ? a : filter == 2 ? b : filter == 3 ? (a + b) / 2 : paeth(a, b, c);
            row[x] = (unsigned char)(src[1 + x] + pred);
        }
    }
    std::vector<uint8_t> rgba((size_t)w * h * 4);
    for (size_t i = 0; i < (size_t)w * h; i++) {
        const unsigned char* p = &img[i * channels];
        uint8_t r, g, b, a = 255;
        if (colour == 0) { r = g = b = p[0]; }
        else if (colour == 4) { r = g = b = p[0]; a = p[1]; }
        else if (colour == 3) {
            const size_t k = (size_t)p[0] * 3;
            if (k + 2 >= palette.size()) return nullptr;
            r = palette[k]; g = palette[k + 1]; b = palette[k + 2];
        } else { r = p[0]; g = p[1]; b = p[2]; if (colour == 6) a = p[3]; }
        rgba[i * 4] = r; rgba[i * 4 + 1] = g; rgba[i * 4 + 2] = b; rgba[i * 4 + 3] = a;
    }
    return std::make_shared<VTexture2D>((size_t)w, (size_t)h, std::move(rgba));
}

VObjectPtr<VTextureCube> VTextureCube::LoadFromFaceDirectory(const std::string& dir) {
    static const char* names[6] = {"XP", "XM", "YP", "YM", "ZP", "ZM"};
    std::vector<uint8_t> all;
    size_t size = 0;
    for (int f = 0; f < 6; f++) {
        const std::string path = dir + (dir.empty() || dir.back() == '/' ? "" : "/") + names[f] + ".png";
        const VObjectPtr<VTexture2D> face = VTexture2D::LoadFromFile(path);
        if (!face || face->GetWidth() != face->GetHeight() || (f > 0 && face->GetWidth() != size)) return nullptr;
        size = face->GetWidth();
        all.insert(all.end(), face->GetPixels().begin(), face->GetPixels().end());
    }
    return std::make_shared<VTextureCube>(size, std::move(all));
}

/* One 4x4 block of BC1 / BC2 / BC3 (DXT1 / DXT3 / DXT5; "Texture Block Compression in Direct3D 11", the format every DDS sky box
 * exported with the reference's toolchain — DirectXTex `texassemble` / `texconv` — is most likely to be in) into 16 RGBA8 texels,
 * row-major.  Colours: two R5G6B5 end points expanded to [0,1] (c/31, c/63), palette entries 2 and 3 at 1/3 and 2/3 (BC1 with
 * c0 <= c1: the midpoint and transparent black); bytes = round(c * 255).  BC2: explicit 4-bit alpha; BC3: two 8-bit alpha end
 * points with six or four interpolated steps; BC4 / BC5: one / two such blocks as the red / red and green channel. */
/* One 8-byte interpolated-alpha block (BC3's alpha half, a BC4 block, each half of a BC5 block): two 8-bit end points with six
 * or four interpolated steps, sixteen 3-bit indices. */
static void decode_alpha_block(const uint8_t* b, uint8_t out[16]) {
    uint8_t alpha[8];
    alpha[0] = b[0];
    alpha[1] = b[1];
    for (int k = 2; k < 8; k++) {
        float a;
        if (alpha[0] > alpha[1]) a = ((float)(8 - k) * (float)alpha[0] + (float)(k - 1) * (float)alpha[1]) / 7.0f;
        else if (k < 6) a = ((float)(6 - k) * (float)alpha[0] + (float)(k - 1) * (float)alpha[1]) / 5.0f;
        else a = k == 6 ? 0.0f : 255.0f;
        alpha[k] = (uint8_t)(a + 0.5f);
    }
    unsigned long long bits = 0;
    for (int k = 0; k < 6; k++) bits |= (unsigned long long)b[2 + k] << (8 * k);
    for (int i = 0; i < 16; i++) out[i] = alpha[(bits >> (3 * i)) & 7u];
}

static void decode_bc_block(const uint8_t* b, int bc /* 1 ... 5 */, uint8_t out[16][4]) {
    if (bc == 4 || bc == 5) { /* BC4: one channel (red), BC5: two (red, green); the others read 0, alpha 1 */
        uint8_t r[16], g[16] = {};
        decode_alpha_block(b, r);
        if (bc == 5) decode_alpha_block(b + 8, g);
        for (int i = 0; i < 16; i++) {
            out[i][0] = r[i];
            out[i][1] = g[i];
            out[i][2] = 0;
            out[i][3] = 255;
        }
        return;
    }
    const uint8_t* col = bc == 1 ? b : b + 8;
    const unsigned c0 = col[0] | col[1] << 8, c1 = col[2] | col[3] << 8;
    float pal[4][4];
    auto expand = [](unsigned c, float* o) {
        o[0] = (float)((c >> 11) & 31u) / 31.0f;
        o[1] = (float)((c >> 5) & 63u) / 63.0f;
        o[2] = (float)(c & 31u) / 31.0f;
        o[3] = 1.0f;
    };
    expand(c0, pal[0]);
    expand(c1, pal[1]);
    const bool four = bc != 1 || c0 > c1;
    for (int k = 0; k < 3; k++) {
        if (four) {
            pal[2][k] = pal[0][k] + (pal[1][k] - pal[0][k]) * (1.0f / 3.0f);
            pal[3][k] = pal[0][k] + (pal[1][k] - pal[0][k]) * (2.0f / 3.0f);
        } else {
            pal[2][k] = pal[0][k] + (pal[1][k] - pal[0][k]) * 0.5f;
            pal[3][k] = 0.0f;
        }
    }
    pal[2][3] = 1.0f;
    pal[3][3] = four ? 1.0f : 0.0f;
    const unsigned idx = col[4] | col[5] << 8 | col[6] << 16 | (unsigned)col[7] << 24;
    uint8_t alpha3[16] = {};
    if (bc == 3) decode_alpha_block(b, alpha3);
    for (int i = 0; i < 16; i++) {
        const float* p = pal[(idx >> (2 * i)) & 3u];
        for (int k = 0; k < 3; k++) out[i][k] = (uint8_t)(p[k] * 255.0f + 0.5f);
        if (bc == 1) out[i][3] = (uint8_t)(p[3] * 255.0f + 0.5f);
        else if (bc == 2) out[i][3] = (uint8_t)(((b[i >> 1] >> ((i & 1) * 4)) & 15u) * 17u);
        else out[i][3] = alpha3[i];
    }
}

VObjectPtr<VTextureCube> VTextureCube::LoadFromDDSFile(const std::string& path) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return nullptr;
    std::vector<uint8_t> file;
    uint8_t chunk[1 << 16];
    size_t n;
    while ((n = fread(chunk, 1, sizeof chunk, f)) > 0) file.insert(file.end(), chunk, chunk + n);
    fclose(f);
    auto u32 = [&](size_t off) -> uint32_t {
        return off + 4 <= file.size() ? (uint32_t)file[off] | (uint32_t)file[off + 1] << 8 | (uint32_t)file[off + 2] << 16 | (uint32_t)file[off + 3] << 24 : 0u;
    };
    if (file.size() < 128 || u32(0) != 0x20534444u /* "DDS " */ || u32(4) != 124 || u32(76) != 32) return nullptr;
    const uint32_t hdr_flags = u32(8), height = u32(12), width = u32(16);
    /* dwMipMapCount is only defined with DDSD_MIPMAPCOUNT, and a square texture has at most 1 + floor(log2(width)) levels: a
       damaged header must neither make the size loop below run for seconds nor shift by >= 32 */
    uint32_t mips = (hdr_flags & 0x20000u) ? u32(28) : 1u;
    const uint32_t pf_flags = u32(80), fourcc = u32(84), bits = u32(88), rmask = u32(92), gmask = u32(96), bmask = u32(100), amask = u32(104);
    const uint32_t caps2 = u32(112);
    if (mips == 0) mips = 1;
    size_t data = 128;
    int bpp = 0;          /* bytes per pixel in the file */
    int bc = 0;           /* block-compressed: 1 = BC1 (DXT1), 2 = BC2 (DXT3), 3 = BC3 (DXT5), 4 = BC4 (ATI1 / BC4U), 5 = BC5 (ATI2 / BC5U) */
    int ri = 0, gi = 1, bi = 2, ai = 3; /* byte index of each channel; ai < 0: opaque */
    bool cube = false;
    const bool ati1 = fourcc == 0x31495441u /* "ATI1" */ || fourcc == 0x55344342u /* "BC4U" */;
    const bool ati2 = fourcc == 0x32495441u /* "ATI2" */ || fourcc == 0x55354342u /* "BC5U" */;
    if ((pf_flags & 0x4u) && (fourcc == 0x31545844u /* "DXT1" */ || fourcc == 0x33545844u /* "DXT3" */ || fourcc == 0x35545844u /* "DXT5" */ || ati1 || ati2)) {
        cube = (caps2 & 0x200u) && (caps2 & 0xfc00u) == 0xfc00u;
        bc = fourcc == 0x31545844u ? 1 : fourcc == 0x33545844u ? 2 : fourcc == 0x35545844u ? 3 : ati1 ? 4 : 5;
    } else if ((pf_flags & 0x4u) && fourcc == 0x30315844u /* "DX10" */) {
        if (file.size() < 148) return nullptr;
        const uint32_t dxgi = u32(128), dim = u32(132), misc = u32(136);
        data = 148;
        cube = dim == 3 /* TEXTURE2D */ && (misc & 0x4u) /* TEXTURECUBE */;
        bpp = 4;
        if (dxgi == 28 || dxgi == 29 || dxgi == 27) { /* R8G8B8A8 (typeless, unorm, srgb) */
        } else if (dxgi == 87 || dxgi == 91 || dxgi == 90) { ri = 2; bi = 0; /* B8G8R8A8 */
        } else if (dxgi == 88 || dxgi == 93 || dxgi == 92) { ri = 2; bi = 0; ai = -1; /* B8G8R8X8 */
        } else if (dxgi >= 70 && dxgi <= 78) { bc = 1 + (int)(dxgi - 70) / 3; /* BC1 / BC2 / BC3 (typeless, unorm, srgb) */
        } else if (dxgi == 79 || dxgi == 80) { bc = 4; /* BC4 (typeless, unorm; signed: not read) */
        } else if (dxgi == 82 || dxgi == 83) { bc = 5; /* BC5 (typeless, unorm) */
        } else return nullptr;
    } else if (pf_flags & 0x40u /* DDPF_RGB */) {
        cube = (caps2 & 0x200u) && (caps2 & 0xfc00u) == 0xfc00u; /* DDSCAPS2_CUBEMAP with all six faces */
        auto byte_of = [](uint32_t m) { return m == 0xffu ? 0 : m == 0xff00u ? 1 : m == 0xff0000u ? 2 : m == 0xff000000u ? 3 : -1; };
        ri = byte_of(rmask); gi = byte_of(gmask); bi = byte_of(bmask);
        ai = (pf_flags & 0x1u /* DDPF_ALPHAPIXELS */) ? byte_of(amask) : -1;
        if (bits == 32) bpp = 4;
        else if (bits == 24) { bpp = 3; ai = -1; }
        else return nullptr;
        if (ri < 0 || gi < 0 || bi < 0 || ri >= bpp || gi >= bpp || bi >= bpp) return nullptr;
    } else {
        return nullptr; /* BC6H / BC7, signed, float or exotic formats: not supported */
    }
    if (!cube || width == 0 || width != height || width > 16384) return nullptr;
    uint32_t max_mips = 1;
    while ((width >> max_mips) != 0) max_mips++;
    mips = std::min(mips, max_mips);
    size_t face_bytes = 0; /* all mip levels of one face */
    const size_t block_bytes = (bc == 1 || bc == 4) ? 8 : 16;
    for (uint32_t m = 0; m < mips; m++) {
        const size_t w = std::max(1u, width >> m), h = std::max(1u, height >> m);
        face_bytes += bc ? ((w + 3) / 4) * ((h + 3) / 4) * block_bytes : w * h * (size_t)bpp;
    }
    if (file.size() < data + 6 * face_bytes) return nullptr; /* truncated */
    std::vector<uint8_t> rgba((size_t)6 * width * height * 4);
    for (int face = 0; face < 6; face++) {
        const uint8_t* src = file.data() + data + (size_t)face * face_bytes;
        uint8_t* dst = rgba.data() + (size_t)face * width * height * 4;
        if (bc) { /* top mip level: rows of 4x4 blocks */
            const size_t bw = (width + 3) / 4;
            uint8_t texels[16][4];
            for (size_t by = 0; by < (height + 3) / 4; by++)
                for (size_t bx = 0; bx < bw; bx++) {
                    decode_bc_block(src + (by * bw + bx) * block_bytes, bc, texels);
                    for (size_t y = 0; y < 4 && by * 4 + y < height; y++)
                        for (size_t x = 0; x < 4 && bx * 4 + x < width; x++)
                            memcpy(dst + ((by * 4 + y) * width + bx * 4 + x) * 4, texels[y * 4 + x], 4);
                }
            continue;
        }
        for (size_t i = 0; i < (size_t)width * height; i++) {
            dst[i * 4 + 0] = src[i * bpp + ri];
            dst[i * 4 + 1] = src[i * bpp + gi];
            dst[i * 4 + 2] = src[i * bpp + bi];
            dst[i * 4 + 3] = ai >= 0 ? src[i * bpp + ai] : 255;
        }
    }
    return std::make_shared<VTextureCube>((size_t)width, std::move(rgba));
}

VObjectPtr<VTexture2D> VTexture2D::LoadFromFile(const std::string& path) {
    if (VObjectPtr<VTexture2D> t = LoadPNG(path)) return t;
    if (VObjectPtr<VTexture2D> t = LoadJPEG(path)) return t;
    return LoadPPM(path);
}

namespace Renderer {

std::shared_ptr<VRenderer> VRendererFactory::NewRenderer() { return std::make_shared<Hip::VHipRenderer>(); }

/* ---- VTextureFactory (Renderer/Public/TextureFactory.h:32-41) over the build's own file readers ------------------------- */
namespace {
/* the reference's asset paths are std::wstring (Windows); file names here are UTF-8 */
std::string narrow(const std::wstring& w) {
    std::string out;
    for (wchar_t wc : w) {
        unsigned long c = (unsigned long)wc;
        if (c < 0x80) out.push_back((char)c);
        else if (c < 0x800) { out.push_back((char)(0xC0 | (c >> 6))); out.push_back((char)(0x80 | (c & 0x3F))); }
        else if (c < 0x10000) { out.push_back((char)(0xE0 | (c >> 12))); out.push_back((char)(0x80 | ((c >> 6) & 0x3F))); out.push_back((char)(0x80 | (c & 0x3F))); }
        else { out.push_back((char)(0xF0 | (c >> 18))); out.push_back((char)(0x80 | ((c >> 12) & 0x3F))); out.push_back((char)(0x80 | ((c >> 6) & 0x3F))); out.push_back((char)(0x80 | (c & 0x3F))); }
    }
    return out;
}
template <typename T>
VObjectPtr<T> initialized(std::weak_ptr<VRenderer> renderer, VObjectPtr<T> texture) {
    if (texture)
        if (const std::shared_ptr<VRenderer> r = renderer.lock()) r->InitializeTexture(texture); /* TextureFactory.cpp:58,114,125,134,143 */
    return texture;
}
}  // namespace

VObjectPtr<VTextureCube> VTextureFactory::LoadTextureCubeFromFile(std::weak_ptr<VRenderer> renderer, const std::wstring& path) {
    const std::string file = narrow(path);
    /* the reference reads a .dds cube map (TextureFactory.cpp:28-67); a folder of six face images (its Resources/Skybox/ layout) is accepted too */
    VObjectPtr<VTextureCube> t = VTextureCube::IsDDSPath(file) ? VTextureCube::LoadFromDDSFile(file) : VTextureCube::LoadFromFaceDirectory(file);
    if (!t) V_LOG_ERROR(("Texture loading failed! " + file).c_str());
    return initialized(renderer, t);
}

VObjectPtr<VTexture2D> VTextureFactory::LoadTexture2DFromFile(std::weak_ptr<VRenderer> renderer, const std::wstring& path) {
    const std::string file = narrow(path);
    VObjectPtr<VTexture2D> t = VTexture2D::LoadFromFile(file); /* R8G8B8A8, one mip (TextureFactory.cpp:69-119 forces RGB and rejects other formats) */
    if (!t) V_LOG_ERROR(("Texture loading failed! " + file).c_str());
    return initialized(renderer, t);
}

VObjectPtr<VTexture3D> VTextureFactory::CreateTexture3D(std::weak_ptr<VRenderer> renderer, const size_t& width, const size_t& height, const size_t& depth, const size_t& mipLevels) {
    return initialized(renderer, std::make_shared<VTexture3D>(width, height, depth, mipLevels));
}

VObjectPtr<VTexture2D> VTextureFactory::CreateTexture2D(std::weak_ptr<VRenderer> renderer, const size_t& width, const size_t& height, const size_t& mipLevels) {
    (void)mipLevels; /* one mip: what the closest-hit shader samples (SampleLevel(.., 0)) */
    return initialized(renderer, std::make_shared<VTexture2D>(width, height, std::vector<uint8_t>(width * height * 4)));
}

VObjectPtr<VTexture3DFloat> VTextureFactory::CreateTexture3DFloat(std::weak_ptr<VRenderer> renderer, const size_t& width, const size_t& height, const size_t& depth, const size_t& mipLevels) {
    return initialized(renderer, std::make_shared<VTexture3DFloat>(width, height, depth, mipLevels));
}

namespace Hip {

namespace {
bool ok(int rc, const char* what) {
    if (rc == VRT_OK) return true;
    fprintf(stderr, "[VHipRenderer][error] %s: %s (%d)\n", what, vrt_strerror(rc), rc);
    return false;
}
}  // namespace

VHipRenderer::VHipRenderer() = default;
VHipRenderer::~VHipRenderer() { Stop(); }

bool VHipRenderer::Start() {
    if (Ctx) return true;
    if (!ok(vrt_create(&Ctx, (int)Devices.size(), Devices.data()), "vrt_create")) {
        Ctx = nullptr;
        return false; /* the engine shows its "no suitable GPU" message and shuts down (Engine.cpp:54-58) */
    }
    return true;
}

void VHipRenderer::Stop() {
    if (!Ctx) return;
    vrt_destroy(Ctx);
    Ctx = nullptr;
    Uploaded.clear();
    UploadedEnv = nullptr;
    for (bool& b : SlotBusy) b = false;
    FramePixels = nullptr;
    FrameBytes = 0;
    BlockFrames = nullptr;
    BlockFrameCount = 0;
    TextureIds.clear();
    for (auto& kv : Textures) kv.second.Id = -1;
}

void VHipRenderer::ResizeRenderOutput(unsigned int width, unsigned int height) {
    if (width == 0 || height == 0) return;
    Width = width;
    Height = height;
}

/* VRenderer::InitializeTexture (Renderer.h:56): the DX backend creates the texture's GPU resource and descriptors here
   (DXRenderer.cpp:115-143); this backend allocates at upload time, so there is nothing to create ahead of it. */
void VHipRenderer::InitializeTexture(VObjectPtr<VTexture>) {}

/* VRenderer::UploadToGPU (Renderer.h:57): by dynamic type, as VDXRenderer::UploadToGPU queues its VDXTexture* kinds (DXRenderer.cpp:145-160). */
void VHipRenderer::UploadToGPU(VObjectPtr<VTexture> texture) {
    if (!texture) return;
    if (auto cube = std::dynamic_pointer_cast<VTextureCube>(texture)) return UploadToGPU(cube);
    if (auto tex2d = std::dynamic_pointer_cast<VTexture2D>(texture)) return UploadToGPU(tex2d);
    /* VTexture3D / VTexture3DFloat: host containers only (volumes reach the device through vrt_volume_upload*) */
}

void VHipRenderer::UploadToGPU(VObjectPtr<VTextureCube> texture) {
    if (!Ctx) {
        V_LOG_WARNING("UploadToGPU on an inactive renderer");
        return;
    }
    if (!texture) {
        if (ok(vrt_env_upload(Ctx, 0, nullptr), "vrt_env_upload")) UploadedEnv = nullptr;
        return;
    }
    if (texture->GetPixels().size() != texture->GetWidth() * texture->GetWidth() * 6 * 4) {
        V_LOG_ERROR("cube map must hold 6 square RGBA8 faces");
        return;
    }
    if (ok(vrt_env_upload(Ctx, (int)texture->GetWidth(), texture->GetPixels().data()), "vrt_env_upload")) UploadedEnv = texture.get();
}

void VHipRenderer::UploadToGPU(VObjectPtr<VTexture2D> texture) {
    if (!Ctx) {
        V_LOG_WARNING("UploadToGPU on an inactive renderer");
        return;
    }
    if (!texture || TextureIds.count(texture.get())) return;
    if (texture->GetPixels().size() != texture->GetWidth() * texture->GetHeight() * 4 || texture->GetWidth() == 0) {
        V_LOG_ERROR("2D texture must hold width*height RGBA8 texels");
        return;
    }
    const int id = (int)TextureIds.size();
    if (id >= VRT_MAX_TEXTURES) {
        V_LOG_ERROR("too many material textures");
        return;
    }
    if (ok(vrt_texture_upload(Ctx, id, (int)texture->GetWidth(), (int)texture->GetHeight(), texture->GetPixels().data()), "vrt_texture_upload"))
        TextureIds[texture.get()] = id;
}

void VHipRenderer::RegisterTexture(const std::string& path, VObjectPtr<VTexture2D> texture) {
    TextureEntry e;
    e.Texture = std::move(texture);
    Textures[path] = e;
}

/* Device texture id of the image a material path names, -1 = unbound. */
int VHipRenderer::ResolveTexture(const std::string& path) {
    if (path.empty()) return -1;
    auto it = Textures.find(path);
    if (it == Textures.end()) {
        TextureEntry e;
        e.Texture = VTexture2D::LoadFromFile(path);
        if (!e.Texture) fprintf(stderr, "[VHipRenderer][warning] material texture '%s' is neither registered nor a readable PNG / binary PPM; left unbound\n", path.c_str());
        it = Textures.emplace(path, e).first;
    }
    if (!it->second.Texture) return -1;
    UploadToGPU(it->second.Texture);
    auto id = TextureIds.find(it->second.Texture.get());
    return id == TextureIds.end() ? -1 : id->second;
}

bool VHipRenderer::SyncWithScene(Scene::VScene& scene) {
    /* geometry: upload new / dirty volumes, free vanished slots (VRDXScene::UpdateSceneGeometry) */
    const auto volumes = scene.GetAllRegisteredVolumes();
    if (volumes.size() > VRT_MAX_VOLUMES) {
        V_LOG_ERROR("more than 20 distinct volumes (MaxAllowedObjectData)");
        return false;
    }
    if (Uploaded.size() < volumes.size()) Uploaded.resize(volumes.size(), nullptr);
    MinCell = 0.f;
    MaxResolution = 0;
    if (!ok(vrt_set_volume_format(Ctx, VolumeFormat), "vrt_set_volume_format")) return false;
    const bool reformat = UploadedFormat != VolumeFormat; /* format switched: every volume is uploaded again */
    UploadedFormat = VolumeFormat;
    for (size_t slot = 0; slot < volumes.size(); slot++) {
        const Voxel::VVoxelVolume& v = *volumes[slot];
        MaxResolution = std::max(MaxResolution, (int)v.GetResolution());
        if (Uploaded[slot] != &v || v.IsDirty() || reformat) {
            static_assert(sizeof(Voxel::VVoxel) == sizeof(vrt_voxel), "VVoxel must match the wire record");
            if (!ok(vrt_volume_upload_voxels(Ctx, (int)slot, v.GetResolution(), v.GetVolumeExtends(),
                                             reinterpret_cast<const vrt_voxel*>(v.GetVoxels().data())), "vrt_volume_upload_voxels"))
                return false;
            Uploaded[slot] = &v;
        }
        const VMaterial& m = v.GetMaterial();
        vrt_material mat = {{m.AlbedoColor.R, m.AlbedoColor.G, m.AlbedoColor.B, m.AlbedoColor.A}, m.Roughness, m.Metallic};
        if (!ok(vrt_volume_set_material(Ctx, (int)slot, &mat), "vrt_volume_set_material")) return false;
        if (!ok(vrt_volume_set_metric(Ctx, (int)slot, v.DensityScale, v.StepMax), "vrt_volume_set_metric")) return false;
        const int ta = ResolveTexture(m.AlbedoTexturePath), tr = ResolveTexture(m.RMTexturePath);
        int tn = ResolveTexture(m.NormalTexturePath);
        if (tn < 0 && ReferenceDefaultTextures) { /* DefaultNormalTexture->SetPixel(VColor(0.5, 0.5, 1, 1)): 255 * 0.5 truncates to 127 (DXTexture2D.cpp:63-71) */
            if (!DefaultNormalTexture) DefaultNormalTexture = std::make_shared<VTexture2D>(1, 1, std::vector<uint8_t>{127, 127, 255, 255});
            UploadToGPU(DefaultNormalTexture);
            const auto id = TextureIds.find(DefaultNormalTexture.get());
            tn = id == TextureIds.end() ? -1 : id->second;
        }
        const float su = m.TextureScale.X != 0.f ? m.TextureScale.X : 100.f, sv = m.TextureScale.Y != 0.f ? m.TextureScale.Y : 100.f;
        if (!ok(vrt_volume_set_textures(Ctx, (int)slot, ta, tn, tr, su, sv), "vrt_volume_set_textures")) return false;
        MinCell = (MinCell == 0.f || v.GetCellSize() < MinCell) ? v.GetCellSize() : MinCell;
    }
    for (size_t slot = volumes.size(); slot < Uploaded.size(); slot++)
        if (Uploaded[slot]) {
            ok(vrt_volume_free(Ctx, (int)slot), "vrt_volume_free");
            Uploaded[slot] = nullptr;
        }
    if (MinCell == 0.f) MinCell = 1.f;

    const VTextureCube* env = scene.GetEnvironmentTexture().get();
    if (env != UploadedEnv) UploadToGPU(scene.GetEnvironmentTexture());

    vrt_scene s;
    if (!FillSceneStruct(scene, s)) return false;
    return ok(vrt_scene_set(Ctx, &s), "vrt_scene_set");
}

/* The scene's constants, lights and instance list as the C-ABI takes them (UpdateSceneConstantBuffer, UpdateLights,
   BuildTopLevelAccelerationStructures: RDXScene.cpp:703-755, 454-545) — for vrt_scene_set, or as one frame of vrt_block::scenes.
   Volume slots are the indices of scene.GetAllRegisteredVolumes(), which SyncWithScene uploaded. */
bool VHipRenderer::FillSceneStruct(Scene::VScene& scene, vrt_scene& s) {
    const auto volumes = scene.GetAllRegisteredVolumes();
    memset(&s, 0, sizeof s);
    const VObjectPtr<Scene::VCamera> cam = scene.GetActiveCamera();
    if (!cam) {
        V_LOG_ERROR("scene has no active camera");
        return false;
    }
    s.cam_position[0] = cam->Position.X; s.cam_position[1] = cam->Position.Y; s.cam_position[2] = cam->Position.Z;
    s.cam_rotation[0] = cam->Rotation.x; s.cam_rotation[1] = cam->Rotation.y; s.cam_rotation[2] = cam->Rotation.z; s.cam_rotation[3] = cam->Rotation.w;
    s.cam_fov_deg = cam->FOVAngle;
    s.cam_near = cam->NearClipPlane;
    s.cam_far = cam->FarClipPlane;
    if (const VObjectPtr<Scene::VLight> dl = scene.GetActiveDirectionalLight()) {
        const VVector f = dl->Rotation.GetForwardVector(); /* RDXScene.cpp:720 */
        s.light_dir[0] = f.X; s.light_dir[1] = f.Y; s.light_dir[2] = f.Z;
        s.light_strength = dl->IlluminationStrength;
    } else {
        s.light_dir[0] = 1.f; /* no directional light: zero strength */
    }
    for (const auto& o : scene.GetAllPlacedObjects()) {
        if (auto pl = std::dynamic_pointer_cast<Scene::VPointLight>(o)) {
            if (s.n_point_lights >= VRT_MAX_POINT_LIGHTS) continue;
            vrt_point_light& L = s.point_lights[s.n_point_lights++];
            L.position[0] = pl->Position.X; L.position[1] = pl->Position.Y; L.position[2] = pl->Position.Z;
            L.color[0] = pl->Color.R; L.color[1] = pl->Color.G; L.color[2] = pl->Color.B;
            L.intensity = pl->IlluminationStrength;
            L.att_linear = pl->AttenuationLinear;
            L.att_exp = pl->AttenuationExp;
        } else if (auto sl = std::dynamic_pointer_cast<Scene::VSpotLight>(o)) {
            if (s.n_spot_lights >= VRT_MAX_SPOT_LIGHTS) continue;
            vrt_spot_light& L = s.spot_lights[s.n_spot_lights++];
            const VVector f = sl->Rotation.GetForwardVector();
            L.position[0] = sl->Position.X; L.position[1] = sl->Position.Y; L.position[2] = sl->Position.Z;
            L.forward[0] = f.X; L.forward[1] = f.Y; L.forward[2] = f.Z;
            L.color[0] = sl->Color.R; L.color[1] = sl->Color.G; L.color[2] = sl->Color.B;
            L.intensity = sl->IlluminationStrength;
            L.att_linear = sl->AttenuationLinear;
            L.att_exp = sl->AttenuationExp;
            L.cos_angle = std::cos(VMathHelpers::ToRadians(sl->Angle * 0.5f));           /* DXLightFactory.cpp:46 */
            L.cos_falloff_angle = std::cos(VMathHelpers::ToRadians(sl->FalloffAngle * 0.5f)); /* :47 */
        } else if (auto vo = std::dynamic_pointer_cast<Scene::VVoxelObject>(o)) {
            if (!vo->GetVoxelVolume()) continue;
            if (s.n_instances >= VRT_MAX_INSTANCES) {
                V_LOG_WARNING("more placed objects than the renderer supports; extra objects are not drawn");
                continue;
            }
            vrt_instance& I = s.instances[s.n_instances++];
            for (size_t slot = 0; slot < volumes.size(); slot++)
                if (volumes[slot] == vo->GetVoxelVolume()) I.volume_slot = (int)slot;
            I.position[0] = vo->Position.X; I.position[1] = vo->Position.Y; I.position[2] = vo->Position.Z;
            I.rotation[0] = vo->Rotation.x; I.rotation[1] = vo->Rotation.y; I.rotation[2] = vo->Rotation.z; I.rotation[3] = vo->Rotation.w;
            I.scale[0] = vo->Scale.X; I.scale[1] = vo->Scale.Y; I.scale[2] = vo->Scale.Z;
        }
    }
    return true;
}

/* March parameters of a frame of `scene` (DESIGN.md §3 defaults from the scene's smallest cell) in the adaptor's frame format. */
vrt_params VHipRenderer::MakeParams(Scene::VScene& scene) const {
    vrt_params p;
    memset(&p, 0, sizeof p);
    p.width = (int)Width;
    p.height = (int)Height;
    p.max_steps = MaxSteps << std::max(0, MaxResolution - 8);
    p.shadow = Shadows ? 1 : 0;
    p.mode = (int)RenderMode;
    p.path = DataPath;
    p.max_bounces = MaxBounces;
    p.eps_hit = std::min(0.004f * MinCell, 0.02f); /* at most a fifth of the secondary rays' 0.1 offset (Raytracing.hlsl:52): coarse volumes */
    p.eps_in = 0.01f; /* Raytracing.hlsl:178 */
    p.step_min = std::min(0.004f * MinCell, 0.02f);
    p.k_relax = Relaxation;
    const VObjectPtr<Scene::VCamera> cam = scene.GetActiveCamera();
    p.cone_eps = std::tan((cam ? cam->FOVAngle : 60.f) * (3.14159265358979323846f / 180.0f) * 0.5f) / (float)Height;
    if (FrameFormat != EFrameFormat::Float4) p.flags |= VRT_FLAG_OUTPUT_RGBA8;
    if (FrameFormat == EFrameFormat::BGRA8) p.flags |= VRT_FLAG_OUTPUT_BGRA8;
    if (ReferenceViewVector) p.flags |= VRT_FLAG_REFERENCE_VIEW_VECTOR;
    if (ReferenceBoundaryTexels) p.flags |= VRT_FLAG_REFERENCE_BOUNDARY_TEXELS;
    return p;
}

void VHipRenderer::Render() {
    if (!IsActive()) {
        V_LOG_WARNING("Render() on an inactive renderer"); /* DXRenderer.cpp:62-65 */
        return;
    }
    const VObjectPtr<Scene::VScene> scene = SceneRef.lock();
    if (!scene) return;
    if (const VObjectPtr<Scene::VCamera> cam = scene->GetActiveCamera()) cam->AspectRatio = (float)Width / (float)Height; /* :47 */
    if (!SyncWithScene(*scene)) return;
    const vrt_params p = MakeParams(*scene);
    const size_t bytes = (size_t)Width * Height * BytesPerPixel();
    if (FramesInFlight <= 1 || Devices.size() != 1) {
        Frame.resize((bytes + sizeof(float) - 1) / sizeof(float));
        if (ok(vrt_render(Ctx, &p, Frame.data()), "vrt_render")) {
            FramePixels = Frame.data();
            FrameBytes = bytes;
        }
        return;
    }
    /* pipelined: like the reference's swap chain, Render() returns once the frame is enqueued; GetFrame() holds the
       newest frame that has been collected (the one begun FramesInFlight calls ago) until Flush() */
    const int k = FramesInFlight > VRT_FRAMES_IN_FLIGHT ? VRT_FRAMES_IN_FLIGHT : FramesInFlight;
    const int slot = (int)(FrameIndex % (unsigned long long)k);
    if (SlotBusy[slot]) Collect(slot);
    if (ok(vrt_render_begin(Ctx, &p, slot), "vrt_render_begin")) {
        SlotBusy[slot] = true;
        SlotBytes[slot] = bytes;
        FrameIndex++;
    }
}

/* A stretch of the application's animation as ONE march launch (vrt_render_block_host over vrt_block::scenes): tick(f) is the
   application's per-frame update (RendererEngineInstance::OnEngineUpdate: objects, lights and camera move; volumes must not
   change inside a block), the scene is mirrored after every tick like SyncWithScene does per frame, then the block is marched
   and its frames land in pinned host memory: GetBlockFrame(f).  What VEngine::EngineLoop does frame by frame
   (tick -> Render), for n frames at once — the reference's TLAS rebuild per frame (DXRenderer.cpp:809-825) becomes one BVH per
   frame of the block, built on the host. */
bool VHipRenderer::RenderBlock(int n_frames, const std::function<void(int)>& tick) {
    if (!IsActive()) {
        V_LOG_WARNING("RenderBlock() on an inactive renderer");
        return false;
    }
    const VObjectPtr<Scene::VScene> scene = SceneRef.lock();
    if (!scene || n_frames < 1 || n_frames > 256 || Devices.size() != 1) return false;
    Flush();
    if (const VObjectPtr<Scene::VCamera> cam = scene->GetActiveCamera()) cam->AspectRatio = (float)Width / (float)Height;
    if (!SyncWithScene(*scene)) return false; /* volumes, materials, textures, sky: as they are at the block's start */
    for (const auto& v : scene->GetAllRegisteredVolumes()) v->PostRender(); /* uploaded: clean (what the engine's post-render tick does, Engine.cpp:214) */
    BlockScenes.resize((size_t)n_frames);
    for (int f = 0; f < n_frames; f++) {
        if (tick) tick(f);
        for (const auto& v : scene->GetAllRegisteredVolumes())
            if (v->IsDirty()) {
                V_LOG_ERROR("RenderBlock: a volume changed inside the block; render such frames with Render()");
                return false;
            }
        if (!FillSceneStruct(*scene, BlockScenes[(size_t)f])) return false;
    }
    const vrt_params p = MakeParams(*scene);
    vrt_block b;
    memset(&b, 0, sizeof b);
    b.n_frames = n_frames;
    b.rows = (int)Height;
    b.scenes = BlockScenes.data();
    const void* frames = nullptr;
    if (!ok(vrt_render_block_host(Ctx, &p, &b, &frames), "vrt_render_block_host")) return false;
    BlockFrames = static_cast<const unsigned char*>(frames);
    BlockFrameCount = n_frames;
    BlockFrameBytes = (size_t)Width * Height * BytesPerPixel();
    FramePixels = BlockFrames + (size_t)(n_frames - 1) * BlockFrameBytes; /* the newest frame, as after Render() */
    FrameBytes = BlockFrameBytes;
    return true;
}

void VHipRenderer::Collect(int slot) {
    const void* px = nullptr;
    if (ok(vrt_render_end(Ctx, slot, &px), "vrt_render_end") && px) {
        /* no copy: the pinned frame of the slot stays valid until the slot is begun again, FramesInFlight - 1 frames on */
        FramePixels = px;
        FrameBytes = SlotBytes[slot];
    }
    SlotBusy[slot] = false;
}

void VHipRenderer::Flush() {
    if (!Ctx) return;
    const int k = FramesInFlight > VRT_FRAMES_IN_FLIGHT ? VRT_FRAMES_IN_FLIGHT : FramesInFlight;
    for (int i = 0; i < k; i++) { /* oldest first, so that GetFrame() ends up with the newest frame */
        const int slot = (int)((FrameIndex + (unsigned long long)i) % (unsigned long long)(k > 0 ? k : 1));
        if (slot < VRT_FRAMES_IN_FLIGHT && SlotBusy[slot]) Collect(slot);
    }
}

bool VHipRenderer::GetLastTiming(vrt_timing& out) const { return Ctx && vrt_last_timing(Ctx, &out) == VRT_OK; }

}  // namespace Hip
}  // namespace Renderer
}  // namespace VolumeRaytracer
