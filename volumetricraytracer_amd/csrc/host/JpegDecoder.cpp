/* JpegDecoder.cpp — baseline, extended-sequential and progressive JPEG (ITU-T T.81: Huffman, 8-bit samples, 1 or 3 components,
 * sampling factors up to 2x2 per component, any number of scans, restart intervals) to RGBA8, for material textures and sky-box
 * faces.  The reference decodes image files through WIC (Renderer/Private/TextureFactory.cpp:58-125, DirectXTex: a Windows codec);
 * this is the part of that a material texture needs.  Arithmetic-coded, lossless, 12-bit and four-component files are refused
 * (nullptr), like a damaged file.
 *
 * The arithmetic is the published one of the Independent JPEG Group's decoder with its default settings — the accurate integer
 * inverse DCT (13-bit constants, two passes), "fancy" triangle-filter chroma upsampling for 2:1 horizontal and 2x2 subsampling,
 * the 16-bit fixed-point YCbCr -> RGB tables of JFIF — so that a file decodes to the bytes every libjpeg-based reader gives
 * (tests/test_voxelizer.py compares with Pillow's, byte for byte). */
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "HostScene.h"

namespace VolumeRaytracer {
namespace {

struct HuffTable {
    bool present = false;
    uint8_t bits[17] = {};
    uint8_t vals[256] = {};
    int mincode[17] = {}, maxcode[18] = {}, valptr[17] = {};
    bool build() {
        int code = 0, k = 0;
        for (int l = 1; l <= 16; l++) {
            valptr[l] = k;
            mincode[l] = code;
            k += bits[l];
            code += bits[l];
            maxcode[l] = bits[l] ? code - 1 : -1;
            if (code > (1 << l)) return false; /* over-subscribed */
            code <<= 1;
        }
        maxcode[17] = 0x7fffffff;
        return k <= 256;
    }
};

struct Component {
    int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
    int blocks_w = 0, blocks_h = 0;   /* blocks in the padded (MCU-aligned) plane */
    int width = 0, height = 0;        /* downsampled size of the real image */
    int pred = 0;
    std::vector<int16_t> coef;        /* blocks_w x blocks_h blocks of 64 coefficients, natural order, as the scans leave them */
    std::vector<uint8_t> plane;       /* blocks_w*8 x blocks_h*8 samples */
};

struct BitReader {
    const uint8_t* p;
    const uint8_t* end;
    uint32_t acc = 0;
    int n = 0;
    bool hit_marker = false;
    void fill() {
        while (n <= 24) {
            int byte = 0;
            if (!hit_marker && p < end) {
                byte = *p;
                if (byte == 0xFF) {
                    if (p + 1 < end && p[1] == 0x00) {
                        p += 2;
                    } else { /* a marker: the entropy-coded segment ends here, zeros follow */
                        hit_marker = true;
                        byte = 0;
                    }
                } else {
                    p++;
                }
            }
            acc |= (uint32_t)byte << (24 - n);
            n += 8;
        }
    }
    int bits(int count) {
        if (count == 0) return 0;
        if (n < count) fill();
        const int v = (int)(acc >> (32 - count));
        acc <<= count;
        n -= count;
        return v;
    }
    void reset() {
        acc = 0;
        n = 0;
        hit_marker = false;
    }
};

int decode_symbol(BitReader& br, const HuffTable& t) {
    int code = 0;
    for (int l = 1; l <= 16; l++) {
        code = (code << 1) | br.bits(1);
        if (t.maxcode[l] >= 0 && code <= t.maxcode[l] && code >= t.mincode[l]) return t.vals[t.valptr[l] + code - t.mincode[l]];
    }
    return -1;
}

inline int extend(int v, int s) { return v < (1 << (s - 1)) ? v - (1 << s) + 1 : v; }

const int kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                         41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                         30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

inline uint8_t clamp8(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

/* The accurate integer inverse DCT (Loeffler / Ligtenberg / Moschytz, 13-bit constants): columns, then rows; +128 and clamp. */
void idct_block(const int* coef /* dequantised, natural order */, uint8_t* out, int stride) {
    constexpr int CB = 13, P1 = 2;
    constexpr int64_t F_0_298 = 2446, F_0_390 = 3196, F_0_541 = 4433, F_0_765 = 6270, F_0_899 = 7373, F_1_175 = 9633,
                      F_1_501 = 12299, F_1_847 = 15137, F_1_961 = 16069, F_2_053 = 16819, F_2_562 = 20995, F_3_072 = 25172;
    /* (64-bit intermediates: a damaged file can hold coefficients far beyond what 8-bit samples produce; valid files give the
       32-bit results) */
    auto descale = [](int64_t x, int n) { return (x + ((int64_t)1 << (n - 1))) >> n; };
    auto pix = [](int64_t v) { return clamp8((int)(v < -1024 ? -1024 : (v > 1024 ? 1024 : v))); };
    int64_t ws[64];
    for (int c = 0; c < 8; c++) {
        const int* in = coef + c;
        int64_t* w = ws + c;
        if (in[8] == 0 && in[16] == 0 && in[24] == 0 && in[32] == 0 && in[40] == 0 && in[48] == 0 && in[56] == 0) {
            const int64_t dc = (int64_t)in[0] * (1 << P1);
            for (int r = 0; r < 8; r++) w[8 * r] = dc;
            continue;
        }
        int64_t z2 = in[16], z3 = in[48];
        int64_t z1 = (z2 + z3) * F_0_541;
        int64_t tmp2 = z1 + z3 * (-F_1_847);
        int64_t tmp3 = z1 + z2 * F_0_765;
        z2 = in[0];
        z3 = in[32];
        int64_t tmp0 = (z2 + z3) * (1 << CB);
        int64_t tmp1 = (z2 - z3) * (1 << CB);
        const int64_t tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
        tmp0 = in[56];
        tmp1 = in[40];
        tmp2 = in[24];
        tmp3 = in[8];
        z1 = tmp0 + tmp3;
        z2 = tmp1 + tmp2;
        z3 = tmp0 + tmp2;
        int64_t z4 = tmp1 + tmp3;
        const int64_t z5 = (z3 + z4) * F_1_175;
        tmp0 *= F_0_298;
        tmp1 *= F_2_053;
        tmp2 *= F_3_072;
        tmp3 *= F_1_501;
        z1 *= -F_0_899;
        z2 *= -F_2_562;
        z3 *= -F_1_961;
        z4 *= -F_0_390;
        z3 += z5;
        z4 += z5;
        tmp0 += z1 + z3;
        tmp1 += z2 + z4;
        tmp2 += z2 + z3;
        tmp3 += z1 + z4;
        w[0] = descale(tmp10 + tmp3, CB - P1);
        w[56] = descale(tmp10 - tmp3, CB - P1);
        w[8] = descale(tmp11 + tmp2, CB - P1);
        w[48] = descale(tmp11 - tmp2, CB - P1);
        w[16] = descale(tmp12 + tmp1, CB - P1);
        w[40] = descale(tmp12 - tmp1, CB - P1);
        w[24] = descale(tmp13 + tmp0, CB - P1);
        w[32] = descale(tmp13 - tmp0, CB - P1);
    }
    for (int r = 0; r < 8; r++) {
        const int64_t* w = ws + 8 * r;
        uint8_t* o = out + (size_t)r * stride;
        int64_t z2 = w[2], z3 = w[6];
        int64_t z1 = (z2 + z3) * F_0_541;
        int64_t tmp2 = z1 + z3 * (-F_1_847);
        int64_t tmp3 = z1 + z2 * F_0_765;
        int64_t tmp0 = (w[0] + w[4]) * (1 << CB);
        int64_t tmp1 = (w[0] - w[4]) * (1 << CB);
        const int64_t tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
        tmp0 = w[7];
        tmp1 = w[5];
        tmp2 = w[3];
        tmp3 = w[1];
        z1 = tmp0 + tmp3;
        z2 = tmp1 + tmp2;
        z3 = tmp0 + tmp2;
        int64_t z4 = tmp1 + tmp3;
        const int64_t z5 = (z3 + z4) * F_1_175;
        tmp0 *= F_0_298;
        tmp1 *= F_2_053;
        tmp2 *= F_3_072;
        tmp3 *= F_1_501;
        z1 *= -F_0_899;
        z2 *= -F_2_562;
        z3 *= -F_1_961;
        z4 *= -F_0_390;
        z3 += z5;
        z4 += z5;
        tmp0 += z1 + z3;
        tmp1 += z2 + z4;
        tmp2 += z2 + z3;
        tmp3 += z1 + z4;
        constexpr int S = CB + P1 + 3;
        o[0] = pix(descale(tmp10 + tmp3, S) + 128);
        o[7] = pix(descale(tmp10 - tmp3, S) + 128);
        o[1] = pix(descale(tmp11 + tmp2, S) + 128);
        o[6] = pix(descale(tmp11 - tmp2, S) + 128);
        o[2] = pix(descale(tmp12 + tmp1, S) + 128);
        o[5] = pix(descale(tmp12 - tmp1, S) + 128);
        o[3] = pix(descale(tmp13 + tmp0, S) + 128);
        o[4] = pix(descale(tmp13 - tmp0, S) + 128);
    }
}

struct Decoder {
    const uint8_t* data;
    size_t size;
    uint16_t qt[4][64] = {};
    bool have_qt[4] = {};
    HuffTable dc[4], ac[4];
    Component comp[3];
    int ncomp = 0, width = 0, height = 0, hmax = 1, vmax = 1, restart_interval = 0;
    int adobe_transform = -1;
    bool have_frame = false, progressive = false;

    /* One block of a scan.  Sequential (Ss = 0, Se = 63, Ah = Al = 0): DC difference and the AC run-lengths; progressive: the DC
       first / refinement bit, or a band of AC coefficients first / refined (T.81 G.1.2; end-of-band runs span blocks). */
    bool decode_block(BitReader& br, Component& c, int16_t* blk, int Ss, int Se, int Ah, int Al, int& eobrun) {
        if (Ss == 0) {
            if (Ah == 0) {
                const int s = decode_symbol(br, dc[c.td]);
                if (s < 0 || s > 11) return false;
                c.pred += s ? extend(br.bits(s), s) : 0;
                /* a DC value is an 11-bit difference accumulated over the blocks: T.81 keeps it within 12 bits for 8-bit samples; a
                   damaged stream that walks it out of 16 bits is refused before the shift below could overflow */
                if (c.pred < -32768 || c.pred > 32767) return false;
                blk[0] = (int16_t)(c.pred * (1 << Al));
            } else if (br.bits(1)) {
                blk[0] = (int16_t)(blk[0] | (1 << Al));
            }
            if (Se == 0) return true;
        }
        const HuffTable& ha = ac[c.ta];
        int k = Ss == 0 ? 1 : Ss;
        if (Ah == 0) {
            if (eobrun > 0) {
                eobrun--;
                return true;
            }
            for (; k <= Se; k++) {
                const int rs = decode_symbol(br, ha);
                if (rs < 0) return false;
                const int r = rs >> 4, sz = rs & 15;
                if (sz) {
                    k += r;
                    if (k > Se) return false;
                    blk[kZigzag[k]] = (int16_t)(extend(br.bits(sz), sz) * (1 << Al));
                } else if (r == 15) {
                    k += 15;
                } else {
                    eobrun = (1 << r) - 1;
                    if (r) eobrun += br.bits(r);
                    break;
                }
            }
            return true;
        }
        /* refinement of a band */
        const int p1 = 1 << Al, m1 = -(1 << Al);
        auto correct = [&](int16_t& v) {
            if (br.bits(1) && (v & p1) == 0) v = (int16_t)(v >= 0 ? v + p1 : v + m1);
        };
        if (eobrun == 0) {
            for (; k <= Se; k++) {
                const int rs = decode_symbol(br, ha);
                if (rs < 0) return false;
                int r = rs >> 4, sz = rs & 15;
                if (sz) {
                    if (sz != 1) return false;
                    sz = br.bits(1) ? p1 : m1;
                } else if (r != 15) {
                    eobrun = 1 << r;
                    if (r) eobrun += br.bits(r);
                    break;
                }
                do { /* over the already non-zero coefficients (a correction bit each) and r still-zero ones */
                    int16_t& v = blk[kZigzag[k]];
                    if (v != 0) {
                        correct(v);
                    } else if (--r < 0) {
                        break;
                    }
                    k++;
                } while (k <= Se);
                if (sz) {
                    if (k > Se) return false;
                    blk[kZigzag[k]] = (int16_t)sz;
                }
            }
        }
        if (eobrun > 0) {
            for (; k <= Se; k++) {
                int16_t& v = blk[kZigzag[k]];
                if (v != 0) correct(v);
            }
            eobrun--;
        }
        return true;
    }

    bool decode_scan(const uint8_t* p, const uint8_t* end, const int* sel, int nsel, int Ss, int Se, int Ah, int Al) {
        BitReader br{p, end};
        int until_restart = restart_interval, next_rst = 0, eobrun = 0;
        for (int i = 0; i < nsel; i++) comp[sel[i]].pred = 0;
        auto restart = [&]() -> bool {
            br.reset();
            const uint8_t* q = br.p;
            while (q + 1 < end && !(q[0] == 0xFF && q[1] >= 0xD0 && q[1] <= 0xD7)) q++;
            if (q + 1 >= end || q[1] != 0xD0 + next_rst) return false;
            br.p = q + 2;
            next_rst = (next_rst + 1) & 7;
            until_restart = restart_interval;
            eobrun = 0;
            for (int i = 0; i < nsel; i++) comp[sel[i]].pred = 0;
            return true;
        };
        if (nsel == 1) { /* not interleaved: the component's own blocks, row by row (those that hold image samples) */
            Component& c = comp[sel[0]];
            const int bw = (c.width + 7) / 8, bh = (c.height + 7) / 8;
            for (int by = 0; by < bh; by++)
                for (int bx = 0; bx < bw; bx++) {
                    if (restart_interval > 0 && until_restart == 0 && !restart()) return false;
                    if (!decode_block(br, c, c.coef.data() + ((size_t)by * c.blocks_w + bx) * 64, Ss, Se, Ah, Al, eobrun)) return false;
                    if (restart_interval > 0) until_restart--;
                }
            return true;
        }
        const int mcus_x = (width + 8 * hmax - 1) / (8 * hmax), mcus_y = (height + 8 * vmax - 1) / (8 * vmax);
        for (int my = 0; my < mcus_y; my++)
            for (int mx = 0; mx < mcus_x; mx++) {
                if (restart_interval > 0 && until_restart == 0 && !restart()) return false;
                for (int i = 0; i < nsel; i++) {
                    Component& c = comp[sel[i]];
                    for (int by = 0; by < c.v; by++)
                        for (int bx = 0; bx < c.h; bx++) {
                            const size_t blk = (size_t)(my * c.v + by) * c.blocks_w + (size_t)(mx * c.h + bx);
                            if (!decode_block(br, c, c.coef.data() + blk * 64, Ss, Se, Ah, Al, eobrun)) return false;
                        }
                }
                if (restart_interval > 0) until_restart--;
            }
        return true;
    }

    /* After the last scan: dequantise and inverse-transform every block. */
    void reconstruct() {
        int blk[64];
        for (int i = 0; i < ncomp; i++) {
            Component& c = comp[i];
            int q[64];
            for (int k = 0; k < 64; k++) q[kZigzag[k]] = qt[c.tq][k];
            for (int by = 0; by < c.blocks_h; by++)
                for (int bx = 0; bx < c.blocks_w; bx++) {
                    const int16_t* src = c.coef.data() + ((size_t)by * c.blocks_w + bx) * 64;
                    for (int k = 0; k < 64; k++) blk[k] = src[k] * q[k];
                    idct_block(blk, c.plane.data() + (size_t)by * 8 * (c.blocks_w * 8) + (size_t)bx * 8, c.blocks_w * 8);
                }
        }
    }

    /* One component as a full-resolution plane (width x height): replication, or the triangle filters for 2:1 / 2x2. */
    void upsample(const Component& c, std::vector<uint8_t>& out) const {
        out.resize((size_t)width * height);
        const int stride = c.blocks_w * 8;
        const int hs = hmax / c.h, vs = vmax / c.v;
        const uint8_t* src = c.plane.data();
        const int cw = c.width;
        if (hs == 1 && vs == 1) {
            for (int y = 0; y < height; y++) memcpy(&out[(size_t)y * width], src + (size_t)y * stride, (size_t)width);
            return;
        }
        std::vector<uint8_t> row((size_t)cw * 2 + 2);
        if (hs == 2 && vs == 1 && cw > 2) {
            for (int y = 0; y < height; y++) {
                const uint8_t* in = src + (size_t)y * stride;
                row[0] = in[0];
                row[1] = (uint8_t)((in[0] * 3 + in[1] + 2) >> 2);
                for (int i = 1; i < cw - 1; i++) {
                    const int v = in[i] * 3;
                    row[2 * i] = (uint8_t)((v + in[i - 1] + 1) >> 2);
                    row[2 * i + 1] = (uint8_t)((v + in[i + 1] + 2) >> 2);
                }
                row[2 * (cw - 1)] = (uint8_t)((in[cw - 1] * 3 + in[cw - 2] + 1) >> 2);
                row[2 * (cw - 1) + 1] = in[cw - 1];
                memcpy(&out[(size_t)y * width], row.data(), (size_t)width);
            }
            return;
        }
        if (hs == 2 && vs == 2 && cw > 2) {
            for (int y = 0; y < height; y++) {
                const int iy = y >> 1;
                int oy = (y & 1) ? iy + 1 : iy - 1; /* the further input row: above for the upper output row, below for the lower */
                oy = oy < 0 ? 0 : oy;
                oy = oy > c.height - 1 ? c.height - 1 : oy; /* (above the first and below the last real row: that row again) */
                const uint8_t* in0 = src + (size_t)iy * stride;
                const uint8_t* in1 = src + (size_t)oy * stride;
                int thiscol = in0[0] * 3 + in1[0], nextcol = in0[1] * 3 + in1[1], lastcol;
                row[0] = (uint8_t)((thiscol * 4 + 8) >> 4);
                row[1] = (uint8_t)((thiscol * 3 + nextcol + 7) >> 4);
                lastcol = thiscol;
                thiscol = nextcol;
                for (int i = 1; i < cw - 1; i++) {
                    nextcol = in0[i + 1] * 3 + in1[i + 1];
                    row[2 * i] = (uint8_t)((thiscol * 3 + lastcol + 8) >> 4);
                    row[2 * i + 1] = (uint8_t)((thiscol * 3 + nextcol + 7) >> 4);
                    lastcol = thiscol;
                    thiscol = nextcol;
                }
                row[2 * (cw - 1)] = (uint8_t)((thiscol * 3 + lastcol + 8) >> 4);
                row[2 * (cw - 1) + 1] = (uint8_t)((thiscol * 4 + 7) >> 4);
                memcpy(&out[(size_t)y * width], row.data(), (size_t)width);
            }
            return;
        }
        for (int y = 0; y < height; y++) /* any other ratio (and planes too narrow for the filters): replication — 4:4:0 (chroma 1x2) among them, where
                                            libjpeg-turbo uses its h1v2 triangle filter: byte parity with libjpeg is UNPINNED for that layout */
            for (int x = 0; x < width; x++) out[(size_t)y * width + x] = src[(size_t)(y / vs) * stride + x / hs];
    }

    bool run(size_t& w_out, size_t& h_out, std::vector<uint8_t>& rgba) {
        if (size < 4 || data[0] != 0xFF || data[1] != 0xD8) return false;
        size_t pos = 2;
        int scans = 0;
        for (;;) {
            while (pos < size && data[pos] != 0xFF) pos++; /* (garbage between segments is skipped) */
            while (pos < size && data[pos] == 0xFF) pos++;
            if (pos >= size) {
                if (scans > 0) break; /* no EOI marker: what the scans gave is the image */
                return false;
            }
            const int m = data[pos++];
            if (m == 0xD9) break;
            if (m == 0x01 || (m >= 0xD0 && m <= 0xD7)) continue;
            if (pos + 2 > size) return false;
            const size_t len = (size_t)data[pos] << 8 | data[pos + 1];
            if (len < 2 || pos + len > size) return false;
            const uint8_t* seg = data + pos + 2;
            const size_t n = len - 2;
            switch (m) {
                case 0xDB: { /* DQT */
                    size_t i = 0;
                    while (i < n) {
                        const int pq = seg[i] >> 4, tq = seg[i] & 15;
                        i++;
                        if (tq > 3 || pq > 1 || i + (size_t)64 * (pq + 1) > n) return false;
                        for (int k = 0; k < 64; k++) {
                            qt[tq][k] = pq ? (uint16_t)(seg[i] << 8 | seg[i + 1]) : seg[i];
                            i += pq + 1;
                        }
                        have_qt[tq] = true;
                    }
                    break;
                }
                case 0xC4: { /* DHT */
                    size_t i = 0;
                    while (i < n) {
                        if (i + 17 > n) return false;
                        const int tc = seg[i] >> 4, th = seg[i] & 15;
                        if (tc > 1 || th > 3) return false;
                        HuffTable& t = tc ? ac[th] : dc[th];
                        int total = 0;
                        t.bits[0] = 0;
                        for (int l = 1; l <= 16; l++) {
                            t.bits[l] = seg[i + l];
                            total += t.bits[l];
                        }
                        i += 17;
                        if (total > 256 || i + (size_t)total > n) return false;
                        memcpy(t.vals, seg + i, (size_t)total);
                        i += (size_t)total;
                        if (!t.build()) return false;
                        t.present = true;
                    }
                    break;
                }
                case 0xC0:
                case 0xC1:
                case 0xC2: { /* baseline / extended sequential / progressive, Huffman */
                    progressive = m == 0xC2;
                    if (have_frame || n < 6 || seg[0] != 8) return false;
                    height = seg[1] << 8 | seg[2];
                    width = seg[3] << 8 | seg[4];
                    ncomp = seg[5];
                    if ((ncomp != 1 && ncomp != 3) || width < 1 || height < 1 || width > 16384 || height > 16384 || (long long)width * height > (64ll << 20) || n < (size_t)6 + 3 * ncomp) return false;
                    for (int i = 0; i < ncomp; i++) {
                        comp[i].id = seg[6 + 3 * i];
                        comp[i].h = seg[7 + 3 * i] >> 4;
                        comp[i].v = seg[7 + 3 * i] & 15;
                        comp[i].tq = seg[8 + 3 * i];
                        if (comp[i].h < 1 || comp[i].h > 2 || comp[i].v < 1 || comp[i].v > 2 || comp[i].tq > 3) return false;
                        hmax = comp[i].h > hmax ? comp[i].h : hmax;
                        vmax = comp[i].v > vmax ? comp[i].v : vmax;
                    }
                    if (ncomp == 1) comp[0].h = comp[0].v = hmax = vmax = 1;
                    for (int i = 0; i < ncomp; i++) {
                        Component& c = comp[i];
                        if (hmax % c.h || vmax % c.v) return false;
                        const int mcus_x = (width + 8 * hmax - 1) / (8 * hmax), mcus_y = (height + 8 * vmax - 1) / (8 * vmax);
                        c.blocks_w = mcus_x * c.h;
                        c.blocks_h = mcus_y * c.v;
                        c.width = (width * c.h + hmax - 1) / hmax;
                        c.height = (height * c.v + vmax - 1) / vmax;
                        c.plane.assign((size_t)c.blocks_w * 8 * c.blocks_h * 8, 0);
                        c.coef.assign((size_t)c.blocks_w * c.blocks_h * 64, 0);
                    }
                    have_frame = true;
                    break;
                }
                case 0xC3: case 0xC5: case 0xC6: case 0xC7: case 0xC9: case 0xCA: case 0xCB: case 0xCD: case 0xCE: case 0xCF:
                    return false; /* lossless, differential, arithmetic: not read */
                case 0xDD:
                    if (n < 2) return false;
                    restart_interval = seg[0] << 8 | seg[1];
                    break;
                case 0xEE: /* Adobe: the colour transform flag */
                    if (n >= 12 && memcmp(seg, "Adobe", 5) == 0) adobe_transform = seg[11];
                    break;
                case 0xDA: { /* SOS: a scan of one or several components; the entropy-coded data follows the header */
                    if (!have_frame || n < 1) return false;
                    const int nsel = seg[0];
                    if (nsel < 1 || nsel > ncomp || n < (size_t)1 + 2 * nsel + 3) return false;
                    int sel[3];
                    for (int i = 0; i < nsel; i++) {
                        int ci = -1;
                        for (int j = 0; j < ncomp; j++)
                            if (comp[j].id == seg[1 + 2 * i]) ci = j;
                        if (ci < 0) return false;
                        for (int j = 0; j < i; j++)
                            if (sel[j] == ci) return false;
                        sel[i] = ci;
                        comp[ci].td = seg[2 + 2 * i] >> 4;
                        comp[ci].ta = seg[2 + 2 * i] & 15;
                        if (comp[ci].td > 3 || comp[ci].ta > 3 || !have_qt[comp[ci].tq]) return false;
                    }
                    int Ss = seg[1 + 2 * nsel], Se = seg[2 + 2 * nsel], Ah = seg[3 + 2 * nsel] >> 4, Al = seg[3 + 2 * nsel] & 15;
                    if (!progressive) {
                        Ss = 0;
                        Se = 63;
                        Ah = Al = 0;
                    } else if (Ss > Se || Se > 63 || Al > 13 || Ah > 13 || (Ss == 0 && Se != 0) || (Ss > 0 && nsel != 1)) {
                        return false;
                    }
                    for (int i = 0; i < nsel; i++) {
                        if (Ss == 0 && Ah == 0 && !dc[comp[sel[i]].td].present) return false;
                        if (Se > 0 && !ac[comp[sel[i]].ta].present) return false;
                    }
                    /* every scan runs over the whole image: a crafted file with millions of tiny SOS segments would be a CPU-time bomb
                       (libjpeg-turbo caps scans for the same reason; real progressive files have about ten) */
                    if (scans >= 256) return false;
                    if (!decode_scan(data + pos + len, data + size, sel, nsel, Ss, Se, Ah, Al)) return false;
                    scans++;
                    /* on to the next marker behind the entropy-coded data (stuffed 0xFF00 and restart markers are part of it) */
                    size_t q = pos + len;
                    while (q + 1 < size && !(data[q] == 0xFF && data[q + 1] != 0x00 && !(data[q + 1] >= 0xD0 && data[q + 1] <= 0xD7))) q++;
                    pos = q;
                    continue;
                }
                default: break; /* APPn, COM, ...: skipped */
            }
            pos += len;
        }
        if (!have_frame || scans == 0) return false;
        reconstruct();
        w_out = (size_t)width;
        h_out = (size_t)height;
        rgba.resize((size_t)width * height * 4);
        if (ncomp == 1) {
            const int stride = comp[0].blocks_w * 8;
            for (int y = 0; y < height; y++)
                for (int x = 0; x < width; x++) {
                    const uint8_t g = comp[0].plane[(size_t)y * stride + x];
                    uint8_t* o = &rgba[((size_t)y * width + x) * 4];
                    o[0] = o[1] = o[2] = g;
                    o[3] = 255;
                }
            return true;
        }
        std::vector<uint8_t> p0, p1, p2;
        upsample(comp[0], p0);
        upsample(comp[1], p1);
        upsample(comp[2], p2);
        const bool rgb = adobe_transform == 0; /* Adobe marker with transform 0: the three components ARE R, G, B */
        /* YCbCr -> RGB, JFIF, 16-bit fixed point */
        constexpr int SB = 16;
        constexpr int32_t HALF = 1 << (SB - 1);
        auto fix = [](double v) { return (int32_t)(v * 65536.0 + 0.5); };
        const int32_t f1402 = fix(1.40200), f1772 = fix(1.77200), f0714 = fix(0.71414), f0344 = fix(0.34414);
        for (size_t i = 0; i < (size_t)width * height; i++) {
            uint8_t* o = &rgba[i * 4];
            if (rgb) {
                o[0] = p0[i];
                o[1] = p1[i];
                o[2] = p2[i];
            } else {
                const int y = p0[i], cb = p1[i] - 128, cr = p2[i] - 128;
                o[0] = clamp8(y + (int)((f1402 * cr + HALF) >> SB));
                o[1] = clamp8(y + (int)((-f0344 * cb + HALF - f0714 * cr) >> SB));
                o[2] = clamp8(y + (int)((f1772 * cb + HALF) >> SB));
            }
            o[3] = 255;
        }
        return true;
    }
};

}  // namespace

VObjectPtr<VTexture2D> VTexture2D::LoadJPEG(const std::string& path) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return nullptr;
    std::vector<uint8_t> file;
    uint8_t buf[65536];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) {
        file.insert(file.end(), buf, buf + n);
        if (file.size() > ((size_t)1 << 30)) break;
    }
    fclose(f);
    auto dec = std::make_unique<Decoder>();
    dec->data = file.data();
    dec->size = file.size();
    size_t w = 0, h = 0;
    std::vector<uint8_t> rgba;
    if (!dec->run(w, h, rgba)) return nullptr;
    return std::make_shared<VTexture2D>(w, h, std::move(rgba));
}

}  // namespace VolumeRaytracer
