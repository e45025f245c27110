// JPEG decoder (ITU-T T.81 baseline and progressive Huffman DCT, 8 bit, JFIF YCbCr or greyscale).
//
// Stands in for what the `image` crate gives the reference when a texture is a JPEG:
//   image::open(path).into_rgb8() / .into_luma8()    src/scene/internal/texture_bank.rs:33,49 (ISF textures)
//   easy-gltf's decoded images                        src/scene/gltf.rs:27-45,78-129 (glTF textures: real assets ship JPEG)
// A JPEG decode is specified only up to the accuracy of the inverse DCT and the choice of the chroma
// interpolation, so - unlike PNG - no two decoders need agree bit for bit and the reference pins no JPEG texture:
// parity unpinned.  This one follows the Independent JPEG Group's defaults (the 13-bit "slow integer" inverse DCT of
// Loeffler, Ligtenberg and Moschytz, triangle-filter "fancy" chroma upsampling, 16-bit fixed-point BT.601 colour
// conversion), which makes PIL (libjpeg-turbo) a bit-exact yardstick for the tests (tests/test_host_loader.py).
// Not supported (PT_ERR_UNSUPPORTED): arithmetic coding, lossless and hierarchical processes, 12-bit samples, CMYK.
#include <cstdlib>
#include <cstring>
#include <memory>
#include <vector>

#include "host_common.hpp"

namespace pth {
namespace {

struct Huff {            // one Huffman table, decoded bit by bit against the canonical code ranges (T.81 F.2.2.3)
    bool present = false;
    uint8_t vals[256];
    int32_t mincode[17], maxcode[18], valptr[17];
};

struct Comp {
    int id = 0, h = 1, v = 1, tq = 0;
    int td = 0, ta = 0;               // tables of the current scan
    uint32_t bw = 0, bh = 0;          // blocks per row / column in the coefficient array (padded to whole MCUs)
    uint32_t cw = 0, ch = 0;          // true size in samples: ceil(W * h / hmax), ceil(H * v / vmax)
    std::vector<int16_t> coef;        // bw * bh * 64, zigzag order undone (natural order)
    int32_t pred = 0;
};

const uint8_t kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                             41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                             30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct Decoder {
    const uint8_t* d;
    size_t n, pos = 0;
    uint32_t W = 0, H = 0;
    int ncomp = 0, hmax = 1, vmax = 1;
    bool progressive = false, have_frame = false, adobe = false;
    int adobe_transform = -1;
    Comp comp[4];
    uint16_t qt[4][64];
    bool qt_present[4] = {false, false, false, false};
    Huff dc[4], ac[4];
    uint32_t restart_interval = 0;
    // entropy-coded segment reader
    uint32_t bitbuf = 0;
    int bitcnt = 0;
    bool hit_marker = false;
    uint32_t eobrun = 0;

    [[noreturn]] void bad(const char* what) { fail(PT_ERR_PARSE, "JPEG: %s", what); }

    uint16_t be16(size_t p) {
        if (p + 2 > n) bad("truncated file");
        return (uint16_t)((d[p] << 8) | d[p + 1]);
    }

    // ---------------------------------------------------------------- bit reader (T.81 F.2.2.5: 0xFF00 is a stuffed 0xFF)
    void reset_bits() {
        bitbuf = 0;
        bitcnt = 0;
        hit_marker = false;
    }
    int get_bit() {
        if (bitcnt == 0) {
            uint8_t b = 0;
            if (!hit_marker && pos < n) {
                b = d[pos];
                if (b == 0xff) {
                    const uint8_t b2 = pos + 1 < n ? d[pos + 1] : 0xd9;
                    if (b2 == 0x00) pos += 2;
                    else {          // a marker ends the segment: feed zeros (a damaged file decodes to something, like libjpeg)
                        hit_marker = true;
                        b = 0;
                    }
                } else {
                    ++pos;
                }
            } else {
                hit_marker = true;
            }
            bitbuf = b;
            bitcnt = 8;
        }
        --bitcnt;
        return (int)((bitbuf >> bitcnt) & 1u);
    }
    int get_bits(int k) {
        int v = 0;
        for (int i = 0; i < k; ++i) v = (v << 1) | get_bit();
        return v;
    }
    static int extend(int v, int t) { return t == 0 ? 0 : (v < (1 << (t - 1)) ? v - (1 << t) + 1 : v); }
    int decode_huff(const Huff& h) {
        if (!h.present) bad("scan uses an undefined Huffman table");
        int code = 0;
        for (int len = 1; len <= 16; ++len) {
            code = (code << 1) | get_bit();
            if (h.maxcode[len] >= 0 && code <= h.maxcode[len] && code >= h.mincode[len]) return h.vals[h.valptr[len] + code - h.mincode[len]];
        }
        bad("bad Huffman code");
    }

    // ---------------------------------------------------------------- tables
    void read_dqt(size_t p, size_t end) {
        while (p < end) {
            const int pq = d[p] >> 4, tq = d[p] & 15;
            ++p;
            if (tq > 3 || pq > 1) bad("bad quantisation table");
            if (p + (pq ? 128 : 64) > end) bad("truncated quantisation table");
            for (int i = 0; i < 64; ++i) {
                qt[tq][kZigzag[i]] = pq ? be16(p) : d[p];
                p += pq ? 2 : 1;
            }
            qt_present[tq] = true;
        }
    }
    void read_dht(size_t p, size_t end) {
        while (p < end) {
            if (p + 17 > end) bad("truncated Huffman table");
            const int tc = d[p] >> 4, th = d[p] & 15;
            if (tc > 1 || th > 3) bad("bad Huffman table id");
            Huff& h = tc ? ac[th] : dc[th];
            int counts[17], total = 0;
            for (int i = 1; i <= 16; ++i) {
                counts[i] = d[p + i];
                total += counts[i];
            }
            p += 17;
            if (total > 256 || p + (size_t)total > end) bad("bad Huffman table");
            memcpy(h.vals, d + p, (size_t)total);
            p += (size_t)total;
            int code = 0, k = 0;
            for (int len = 1; len <= 16; ++len) {
                h.valptr[len] = k;
                h.mincode[len] = code;
                code += counts[len];
                k += counts[len];
                h.maxcode[len] = counts[len] ? code - 1 : -1;
                if (code > (1 << len)) bad("over-subscribed Huffman table");
                code <<= 1;
            }
            h.present = true;
        }
    }
    void read_sof(size_t p, size_t end, bool prog) {
        if (have_frame) bad("more than one frame");
        if (end - p < 6) bad("truncated frame header");
        if (d[p] != 8) fail(PT_ERR_UNSUPPORTED, "JPEG: %d-bit samples are not supported", d[p]);
        H = be16(p + 1);
        W = be16(p + 3);
        ncomp = d[p + 5];
        if (W == 0 || H == 0) bad("empty image (or a DNL-defined height: not supported)");
        if (ncomp != 1 && ncomp != 3) fail(PT_ERR_UNSUPPORTED, "JPEG: %d components (only greyscale and YCbCr are supported)", ncomp);
        if (end - p < 6 + 3 * (size_t)ncomp) bad("truncated frame header");
        for (int i = 0; i < ncomp; ++i) {
            Comp& c = comp[i];
            c.id = d[p + 6 + 3 * i];
            c.h = d[p + 7 + 3 * i] >> 4;
            c.v = d[p + 7 + 3 * i] & 15;
            c.tq = d[p + 8 + 3 * i];
            if (c.h < 1 || c.h > 4 || c.v < 1 || c.v > 4 || c.tq > 3) bad("bad component parameters");
            hmax = std::max(hmax, c.h);
            vmax = std::max(vmax, c.v);
        }
        if (ncomp == 1) comp[0].h = comp[0].v = hmax = vmax = 1;   // (a single component is never interleaved: factors are moot)
        const uint32_t mcux = (W + 8 * hmax - 1) / (8 * hmax), mcuy = (H + 8 * vmax - 1) / (8 * vmax);
        for (int i = 0; i < ncomp; ++i) {
            Comp& c = comp[i];
            c.bw = mcux * c.h;
            c.bh = mcuy * c.v;
            c.cw = (W * c.h + hmax - 1) / hmax;
            c.ch = (H * c.v + vmax - 1) / vmax;
            // the header must not drive the allocation: an entropy-coded block takes at least a bit
            if ((uint64_t)c.bw * c.bh > (uint64_t)n * 8 + 64) bad("image size exceeds what the file can hold");
            c.coef.assign((size_t)c.bw * c.bh * 64, 0);
        }
        progressive = prog;
        have_frame = true;
    }

    // ---------------------------------------------------------------- block decoders
    // The DC predictor of a component: a sum of up to 2^26 differences of 11 bits each.  A conforming stream keeps it within the
    // 16-bit coefficient range (T.81 F.1.2.1); a crafted one would walk a plain `int` into signed overflow - rejected instead.
    int dc_pred(int pred, int diff) {
        const long long p = (long long)pred + diff;
        if (p < -32768 || p > 32767) bad("DC coefficient out of range");
        return (int)p;
    }
    void block_baseline(Comp& c, int16_t* b) {
        const int t = decode_huff(dc[c.td]);
        if (t > 11) bad("bad DC magnitude");
        c.pred = dc_pred(c.pred, extend(get_bits(t), t));
        b[0] = (int16_t)c.pred;
        for (int k = 1; k < 64;) {
            const int rs = decode_huff(ac[c.ta]), r = rs >> 4, s = rs & 15;
            if (s == 0) {
                if (r != 15) break;   // EOB
                k += 16;
                continue;
            }
            k += r;
            if (k > 63) bad("coefficient index out of range");
            b[kZigzag[k]] = (int16_t)extend(get_bits(s), s);
            ++k;
        }
    }
    void block_dc_first(Comp& c, int16_t* b, int al) {
        const int t = decode_huff(dc[c.td]);
        if (t > 11) bad("bad DC magnitude");
        c.pred = dc_pred(c.pred, extend(get_bits(t), t));
        b[0] = (int16_t)((int64_t)c.pred * ((int64_t)1 << al));   // (al <= 13; the product in 64 bits, truncated like libjpeg's cast)
    }
    void block_dc_refine(int16_t* b, int al) {
        if (get_bit()) b[0] = (int16_t)(b[0] | (1 << al));
    }
    void block_ac_first(Comp& c, int16_t* b, int ss, int se, int al) {
        if (eobrun > 0) {
            --eobrun;
            return;
        }
        for (int k = ss; k <= se;) {
            const int rs = decode_huff(ac[c.ta]), r = rs >> 4, s = rs & 15;
            if (s == 0) {
                if (r < 15) {   // EOBn: this block and 2^r - 1 + extra more are finished
                    eobrun = (1u << r) - 1u;
                    if (r) eobrun += (uint32_t)get_bits(r);
                    break;
                }
                k += 16;
                continue;
            }
            k += r;
            if (k > se) bad("coefficient index out of range");
            b[kZigzag[k]] = (int16_t)(extend(get_bits(s), s) * (1 << al));
            ++k;
        }
    }
    void block_ac_refine(Comp& c, int16_t* b, int ss, int se, int al) {   // T.81 G.1.2.3
        const int p1 = 1 << al, m1 = -(1 << al);
        int k = ss;
        if (eobrun == 0) {
            for (; k <= se;) {
                const int rs = decode_huff(ac[c.ta]);
                int r = rs >> 4;
                const int s = rs & 15;
                int value = 0;
                if (s == 0) {
                    if (r < 15) {
                        eobrun = (1u << r);
                        if (r) eobrun += (uint32_t)get_bits(r);
                        break;   // the rest of the block is handled by the EOB logic below
                    }
                } else {
                    if (s != 1) bad("bad refinement code");
                    value = get_bit() ? p1 : m1;
                }
                // advance over already-nonzero coefficients (each takes a correction bit) and r zero-valued ones
                for (; k <= se; ++k) {
                    int16_t& co = b[kZigzag[k]];
                    if (co != 0) {
                        if (get_bit() && (co & p1) == 0) co = (int16_t)(co >= 0 ? co + p1 : co + m1);
                    } else {
                        if (--r < 0) break;
                    }
                }
                if (value && k <= se) b[kZigzag[k]] = (int16_t)value;
                ++k;
            }
        }
        if (eobrun > 0) {   // correction bits for the nonzero coefficients of the rest of the band
            for (; k <= se; ++k) {
                int16_t& co = b[kZigzag[k]];
                if (co != 0 && get_bit() && (co & p1) == 0) co = (int16_t)(co >= 0 ? co + p1 : co + m1);
            }
            --eobrun;
        }
    }

    // ---------------------------------------------------------------- one scan
    void read_scan(size_t p, size_t end) {
        if (!have_frame) bad("scan before the frame header");
        if (end - p < 1) bad("truncated scan header");
        const int ns = d[p];
        if (ns < 1 || ns > ncomp || end - p < 4 + 2 * (size_t)ns) bad("bad scan header");
        int sel[4];
        for (int i = 0; i < ns; ++i) {
            const int id = d[p + 1 + 2 * i];
            int ci = -1;
            for (int j = 0; j < ncomp; ++j)
                if (comp[j].id == id) ci = j;
            if (ci < 0) bad("scan names an unknown component");
            for (int j = 0; j < i; ++j)
                if (sel[j] == ci) bad("scan names a component twice");
            comp[ci].td = d[p + 2 + 2 * i] >> 4;
            comp[ci].ta = d[p + 2 + 2 * i] & 15;
            if (comp[ci].td > 3 || comp[ci].ta > 3) bad("bad table selector");
            sel[i] = ci;
        }
        const int ss = d[p + 1 + 2 * ns], se = d[p + 2 + 2 * ns], ah = d[p + 3 + 2 * ns] >> 4, al = d[p + 3 + 2 * ns] & 15;
        if (progressive) {
            if (ss > se || se > 63 || al > 13 || ah > 13 || (ss == 0 && se != 0) || (ss != 0 && ns != 1)) bad("bad progressive scan parameters");
        } else if (ss != 0 || se != 63 || ah != 0 || al != 0) {
            bad("bad sequential scan parameters");
        }
        pos = end;
        reset_bits();
        eobrun = 0;
        for (int i = 0; i < ncomp; ++i) comp[i].pred = 0;
        // MCU geometry: interleaved (ns > 1) - h x v blocks per component and MCU; one component - block by block over
        // its TRUE extent (T.81 A.2.2)
        uint32_t mcux, mcuy;
        if (ns == 1) {
            const Comp& c = comp[sel[0]];
            mcux = (c.cw + 7) / 8;
            mcuy = (c.ch + 7) / 8;
        } else {
            mcux = (W + 8 * hmax - 1) / (8 * hmax);
            mcuy = (H + 8 * vmax - 1) / (8 * vmax);
        }
        uint32_t until_restart = restart_interval, next_rst = 0;
        for (uint32_t my = 0; my < mcuy; ++my) {
            for (uint32_t mx = 0; mx < mcux; ++mx) {
                if (restart_interval && until_restart == 0) {
                    // RSTn: byte-align, expect the marker, reset the predictions (T.81 F.2.2.4 / G.1.2)
                    while (pos + 1 < n && !(d[pos] == 0xff && d[pos + 1] >= 0xd0 && d[pos + 1] <= 0xd7)) {
                        if (d[pos] == 0xff && d[pos + 1] != 0x00 && d[pos + 1] != 0xff) break;   // some other marker: give up aligning
                        ++pos;
                    }
                    if (pos + 1 < n && d[pos] == 0xff && d[pos + 1] == 0xd0 + (next_rst & 7)) pos += 2;
                    ++next_rst;
                    reset_bits();
                    eobrun = 0;
                    for (int i = 0; i < ncomp; ++i) comp[i].pred = 0;
                    until_restart = restart_interval;
                }
                for (int i = 0; i < ns; ++i) {
                    Comp& c = comp[sel[i]];
                    const int bh_ = ns == 1 ? 1 : c.h, bv_ = ns == 1 ? 1 : c.v;
                    for (int by = 0; by < bv_; ++by)
                        for (int bx = 0; bx < bh_; ++bx) {
                            const uint32_t col = mx * (uint32_t)bh_ + (uint32_t)bx, row = my * (uint32_t)bv_ + (uint32_t)by;
                            if (col >= c.bw || row >= c.bh) bad("block outside the image");
                            int16_t* b = &c.coef[((size_t)row * c.bw + col) * 64];
                            if (!progressive) block_baseline(c, b);
                            else if (ss == 0) {
                                if (ah == 0) block_dc_first(c, b, al);
                                else block_dc_refine(b, al);
                            } else {
                                if (ah == 0) block_ac_first(c, b, ss, se, al);
                                else block_ac_refine(c, b, ss, se, al);
                            }
                        }
                }
                if (restart_interval) --until_restart;
            }
        }
        // skip to the next marker
        while (pos + 1 < n && !(d[pos] == 0xff && d[pos + 1] != 0x00 && !(d[pos + 1] >= 0xd0 && d[pos + 1] <= 0xd7) && d[pos + 1] != 0xff)) ++pos;
    }

    // ---------------------------------------------------------------- inverse DCT (jidctint.c: LL&M, CONST_BITS 13, PASS1_BITS 2)
    static inline uint8_t clamp8(int v) { return (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v); }
    static void idct(const int16_t* in, const uint16_t* q, uint8_t* out, size_t stride) {
        constexpr int CB = 13, P1 = 2;
        constexpr int32_t F0_298631336 = 2446, F0_390180644 = 3196, F0_541196100 = 4433, F0_765366865 = 6270, F0_899976223 = 7373,
                          F1_175875602 = 9633, F1_501321110 = 12299, F1_847759065 = 15137, F1_961570560 = 16069,
                          F2_053119869 = 16819, F2_562915447 = 20995, F3_072711026 = 25172;
        int32_t ws[64];
        auto descale = [](int64_t x, int nbits) { return (int32_t)((x + ((int64_t)1 << (nbits - 1))) >> nbits); };
        for (int c = 0; c < 8; ++c) {
            const int32_t d0 = in[c] * q[c], d1 = in[8 + c] * q[8 + c], d2 = in[16 + c] * q[16 + c], d3 = in[24 + c] * q[24 + c],
                          d4 = in[32 + c] * q[32 + c], d5 = in[40 + c] * q[40 + c], d6 = in[48 + c] * q[48 + c], d7 = in[56 + c] * q[56 + c];
            if ((d1 | d2 | d3 | d4 | d5 | d6 | d7) == 0) {
                // (d0 = coefficient x a 16-bit quantiser entry can reach 2^31: the shift in 64 bits, the store truncated like the
                // general path's descale())
                const int32_t dcv = (int32_t)((int64_t)d0 * (1 << P1));
                for (int r = 0; r < 8; ++r) ws[8 * r + c] = dcv;
                continue;
            }
            int64_t z2 = d2, z3 = d6;
            int64_t z1 = (z2 + z3) * F0_541196100;
            int64_t tmp2 = z1 + z3 * (-F1_847759065);
            int64_t tmp3 = z1 + z2 * F0_765366865;
            z2 = d0;
            z3 = d4;
            int64_t tmp0 = (z2 + z3) * (1 << CB), tmp1 = (z2 - z3) * (1 << CB);
            const int64_t tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
            tmp0 = d7;
            tmp1 = d5;
            tmp2 = d3;
            tmp3 = d1;
            z1 = tmp0 + tmp3;
            z2 = tmp1 + tmp2;
            z3 = tmp0 + tmp2;
            int64_t z4 = tmp1 + tmp3;
            const int64_t z5 = (z3 + z4) * F1_175875602;
            tmp0 *= F0_298631336;
            tmp1 *= F2_053119869;
            tmp2 *= F3_072711026;
            tmp3 *= F1_501321110;
            z1 *= -F0_899976223;
            z2 *= -F2_562915447;
            z3 = z3 * (-F1_961570560) + z5;
            z4 = z4 * (-F0_390180644) + z5;
            tmp0 += z1 + z3;
            tmp1 += z2 + z4;
            tmp2 += z2 + z3;
            tmp3 += z1 + z4;
            ws[c] = descale(tmp10 + tmp3, CB - P1);
            ws[56 + c] = descale(tmp10 - tmp3, CB - P1);
            ws[8 + c] = descale(tmp11 + tmp2, CB - P1);
            ws[48 + c] = descale(tmp11 - tmp2, CB - P1);
            ws[16 + c] = descale(tmp12 + tmp1, CB - P1);
            ws[40 + c] = descale(tmp12 - tmp1, CB - P1);
            ws[24 + c] = descale(tmp13 + tmp0, CB - P1);
            ws[32 + c] = descale(tmp13 - tmp0, CB - P1);
        }
        for (int r = 0; r < 8; ++r) {
            const int32_t* w = ws + 8 * r;
            uint8_t* o = out + stride * (size_t)r;
            int64_t z2 = w[2], z3 = w[6];
            int64_t z1 = (z2 + z3) * F0_541196100;
            int64_t tmp2 = z1 + z3 * (-F1_847759065);
            int64_t tmp3 = z1 + z2 * F0_765366865;
            int64_t tmp0 = ((int64_t)w[0] + w[4]) * (1 << CB), tmp1 = ((int64_t)w[0] - w[4]) * (1 << CB);
            const int64_t tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
            tmp0 = w[7];
            tmp1 = w[5];
            tmp2 = w[3];
            tmp3 = w[1];
            z1 = tmp0 + tmp3;
            z2 = tmp1 + tmp2;
            z3 = tmp0 + tmp2;
            int64_t z4 = tmp1 + tmp3;
            const int64_t z5 = (z3 + z4) * F1_175875602;
            tmp0 *= F0_298631336;
            tmp1 *= F2_053119869;
            tmp2 *= F3_072711026;
            tmp3 *= F1_501321110;
            z1 *= -F0_899976223;
            z2 *= -F2_562915447;
            z3 = z3 * (-F1_961570560) + z5;
            z4 = z4 * (-F0_390180644) + z5;
            tmp0 += z1 + z3;
            tmp1 += z2 + z4;
            tmp2 += z2 + z3;
            tmp3 += z1 + z4;
            constexpr int SH = CB + P1 + 3;
            o[0] = clamp8(descale(tmp10 + tmp3, SH) + 128);
            o[7] = clamp8(descale(tmp10 - tmp3, SH) + 128);
            o[1] = clamp8(descale(tmp11 + tmp2, SH) + 128);
            o[6] = clamp8(descale(tmp11 - tmp2, SH) + 128);
            o[2] = clamp8(descale(tmp12 + tmp1, SH) + 128);
            o[5] = clamp8(descale(tmp12 - tmp1, SH) + 128);
            o[3] = clamp8(descale(tmp13 + tmp0, SH) + 128);
            o[4] = clamp8(descale(tmp13 - tmp0, SH) + 128);
        }
    }

    // ---------------------------------------------------------------- component plane -> full resolution (jdsample.c, fancy upsampling)
    // plane: c.cw x c.ch true samples inside a (bw*8)-wide buffer.  Returns a W x H plane.
    std::vector<uint8_t> upsample(const Comp& c, const std::vector<uint8_t>& plane) {
        const size_t ps = (size_t)c.bw * 8;
        std::vector<uint8_t> out((size_t)W * H);
        const int hx = hmax / c.h, vx = vmax / c.v;
        const bool exact = hmax % c.h == 0 && vmax % c.v == 0;
        if (!exact) fail(PT_ERR_UNSUPPORTED, "JPEG: fractional sampling ratios are not supported");
        auto row = [&](int64_t y) -> const uint8_t* {   // rows above / below the image repeat the first / last true row
            if (y < 0) y = 0;
            if (y >= (int64_t)c.ch) y = (int64_t)c.ch - 1;
            return &plane[(size_t)y * ps];
        };
        if (hx == 1 && vx == 1) {
            for (uint32_t y = 0; y < H; ++y) memcpy(&out[(size_t)y * W], row(y), W);
        } else if (hx == 2 && vx == 1) {   // h2v1_fancy_upsample
            std::vector<uint8_t> line((size_t)c.cw * 2 + 2);
            for (uint32_t y = 0; y < H; ++y) {
                const uint8_t* in = row(y);
                const uint32_t w = c.cw;
                if (w == 1) {
                    line[0] = line[1] = in[0];
                } else {
                    line[0] = in[0];
                    line[1] = (uint8_t)((in[0] * 3 + in[1] + 2) >> 2);
                    for (uint32_t x = 1; x + 1 < w; ++x) {
                        const int v = in[x] * 3;
                        line[2 * x] = (uint8_t)((v + in[x - 1] + 1) >> 2);
                        line[2 * x + 1] = (uint8_t)((v + in[x + 1] + 2) >> 2);
                    }
                    line[2 * (w - 1)] = (uint8_t)((in[w - 1] * 3 + in[w - 2] + 1) >> 2);
                    line[2 * (w - 1) + 1] = in[w - 1];
                }
                memcpy(&out[(size_t)y * W], line.data(), W);
            }
        } else if (hx == 2 && vx == 2) {   // h2v2_fancy_upsample
            std::vector<uint8_t> line((size_t)c.cw * 2 + 2);
            for (uint32_t y = 0; y < H; ++y) {
                const int64_t iy = y / 2;
                const uint8_t* in0 = row(iy);
                const uint8_t* in1 = row((y & 1) ? iy + 1 : iy - 1);   // the nearer neighbour row
                const uint32_t w = c.cw;
                auto colsum = [&](uint32_t x) { return in0[x] * 3 + in1[x]; };
                if (w == 1) {
                    const int t = colsum(0);
                    line[0] = (uint8_t)((t * 4 + 8) >> 4);
                    line[1] = (uint8_t)((t * 4 + 7) >> 4);
                } else {
                    int last, cur = colsum(0), next = colsum(1);
                    line[0] = (uint8_t)((cur * 4 + 8) >> 4);
                    line[1] = (uint8_t)((cur * 3 + next + 7) >> 4);
                    for (uint32_t x = 1; x + 1 < w; ++x) {
                        last = cur;
                        cur = next;
                        next = colsum(x + 1);
                        line[2 * x] = (uint8_t)((cur * 3 + last + 8) >> 4);
                        line[2 * x + 1] = (uint8_t)((cur * 3 + next + 7) >> 4);
                    }
                    last = cur;
                    cur = next;
                    line[2 * (w - 1)] = (uint8_t)((cur * 3 + last + 8) >> 4);
                    line[2 * (w - 1) + 1] = (uint8_t)((cur * 4 + 7) >> 4);
                }
                memcpy(&out[(size_t)y * W], line.data(), W);
            }
        } else if (hx == 1 && vx == 2) {   // h1v2: the same triangle filter, vertically
            for (uint32_t y = 0; y < H; ++y) {
                const int64_t iy = y / 2;
                const uint8_t* in0 = row(iy);
                const uint8_t* in1 = row((y & 1) ? iy + 1 : iy - 1);
                const int bias = (y & 1) ? 2 : 1;
                for (uint32_t x = 0; x < W; ++x) out[(size_t)y * W + x] = (uint8_t)((in0[x] * 3 + in1[x] + bias) >> 2);
            }
        } else {   // int_upsample: replication
            for (uint32_t y = 0; y < H; ++y) {
                const uint8_t* in = row(y / (uint32_t)vx);
                for (uint32_t x = 0; x < W; ++x) out[(size_t)y * W + x] = in[std::min<uint32_t>(x / (uint32_t)hx, c.cw - 1)];
            }
        }
        return out;
    }

    void run(uint32_t want, uint32_t* ow, uint32_t* oh, uint8_t** opx) {
        if (n < 4 || d[0] != 0xff || d[1] != 0xd8) bad("not a JPEG file");
        pos = 2;
        bool seen_eoi = false, any_scan = false;
        while (!seen_eoi) {
            while (pos < n && d[pos] != 0xff) ++pos;   // (garbage between segments is skipped, as libjpeg does)
            while (pos < n && d[pos] == 0xff) ++pos;   // fill bytes
            if (pos >= n) break;
            const uint8_t m = d[pos++];
            if (m == 0xd9) {
                seen_eoi = true;
                break;
            }
            if (m == 0x01 || (m >= 0xd0 && m <= 0xd7)) continue;   // TEM, stray RSTn: no payload
            const size_t len = be16(pos);
            if (len < 2 || pos + len > n) bad("truncated segment");
            const size_t p = pos + 2, end = pos + len;
            switch (m) {
                case 0xdb: read_dqt(p, end); break;
                case 0xc4: read_dht(p, end); break;
                case 0xc0: case 0xc1: read_sof(p, end, false); break;
                case 0xc2: read_sof(p, end, true); break;
                case 0xc3: case 0xc5: case 0xc6: case 0xc7: case 0xc9: case 0xca: case 0xcb: case 0xcd: case 0xce: case 0xcf:
                    fail(PT_ERR_UNSUPPORTED, "JPEG: coding process SOF%d (lossless / hierarchical / arithmetic) is not supported", m - 0xc0);
                case 0xdd:
                    if (end - p < 2) bad("bad DRI");
                    restart_interval = be16(p);
                    break;
                case 0xee:   // Adobe: transform 0 = the three components are RGB, not YCbCr
                    if (end - p >= 12 && !memcmp(d + p, "Adobe", 5)) {
                        adobe = true;
                        adobe_transform = d[p + 11];
                    }
                    break;
                case 0xda:
                    read_scan(p, end);   // (moves pos past the entropy-coded data)
                    any_scan = true;
                    continue;
                default: break;          // APPn, COM, ...
            }
            pos = end;
        }
        if (!have_frame || !any_scan) bad("no image data");
        // dequantise + inverse DCT, component by component
        std::vector<uint8_t> full[3];
        for (int i = 0; i < ncomp; ++i) {
            Comp& c = comp[i];
            if (!qt_present[c.tq]) bad("frame uses an undefined quantisation table");
            const size_t ps = (size_t)c.bw * 8;
            std::vector<uint8_t> plane(ps * (size_t)c.bh * 8);
            for (uint32_t by = 0; by < c.bh; ++by)
                for (uint32_t bx = 0; bx < c.bw; ++bx)
                    idct(&c.coef[((size_t)by * c.bw + bx) * 64], qt[c.tq], &plane[(size_t)by * 8 * ps + (size_t)bx * 8], ps);
            std::vector<int16_t>().swap(c.coef);
            full[i] = upsample(c, plane);
        }
        if (want != 1 && want != 3 && want != 4) fail(PT_ERR_INVALID, "want_channels must be 1, 3 or 4");
        std::unique_ptr<uint8_t, void (*)(void*)> owner((uint8_t*)malloc((size_t)W * H * want), free);
        uint8_t* out = owner.get();
        if (!out) throw std::bad_alloc();
        const bool rgb_direct = ncomp == 3 && adobe && adobe_transform == 0;
        for (size_t i = 0; i < (size_t)W * H; ++i) {
            int r, g, b;
            bool grey = false;
            if (ncomp == 1) {
                r = g = b = full[0][i];
                grey = true;
            } else if (rgb_direct) {
                r = full[0][i];
                g = full[1][i];
                b = full[2][i];
            } else {   // jdcolor.c ycc_rgb_convert: 16-bit fixed point, the G terms summed before the shift
                const int y = full[0][i], cb = full[1][i] - 128, cr = full[2][i] - 128;
                r = clamp8(y + (int)((91881 * (int64_t)cr + 32768) >> 16));
                g = clamp8(y + (int)((-22554 * (int64_t)cb + 32768 - 46802 * (int64_t)cr) >> 16));
                b = clamp8(y + (int)((116130 * (int64_t)cb + 32768) >> 16));
            }
            uint8_t* o = out + i * want;
            if (want >= 3) {
                o[0] = (uint8_t)r;
                o[1] = (uint8_t)g;
                o[2] = (uint8_t)b;
                if (want == 4) o[3] = 255;
            } else {   // into_luma8 of a colour image: the same integer weights as for PNG (png_codec.cpp)
                o[0] = grey ? (uint8_t)r : (uint8_t)((2126u * (unsigned)r + 7152u * (unsigned)g + 722u * (unsigned)b) / 10000u);
            }
        }
        *ow = W;
        *oh = H;
        *opx = owner.release();
    }
};

}  // namespace

void decode_jpeg(const uint8_t* data, size_t len, uint32_t want, uint32_t* ow, uint32_t* oh, uint8_t** opx) {
    auto dec = std::make_unique<Decoder>();
    dec->d = data;
    dec->n = len;
    dec->run(want, ow, oh, opx);
}

}  // namespace pth
