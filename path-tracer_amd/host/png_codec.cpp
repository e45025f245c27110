// Minimal PNG codec on top of zlib (libpng is not in the image); pth_png_read / pth_png_decode also take JPEG files
// (sniffed by content, decoded by jpeg_codec.cpp): `image::open` decodes whatever format a texture comes in.
//
// Stands in for the `image` crate calls of the reference:
//   image::open(path).into_rgb8()  / .into_luma8()   src/scene/internal/texture_bank.rs:33,49
//   RgbImage::save(path)                              src/main.rs:50
// Decoding supports every non-interlaced PNG colour type / bit depth and
// converts like `image` 0.25: gray -> rgb replicates, rgb -> luma uses the
// integer weights (2126, 7152, 722)/10000, alpha is dropped, 16-bit samples
// are reduced with (v + 128) / 257, sub-byte gray is scaled to 0..255,
// palette entries expand to rgb.  Only the 8-bit gray / rgb / palette cases
// are pinned by reference fixtures (head, alpha_transparency); the rest is
// "parity unpinned" (SURVEY §4).
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <thread>
#include <fstream>
#include <vector>

#include "host_common.hpp"

namespace pth {
void decode_jpeg(const uint8_t* data, size_t len, uint32_t want, uint32_t* ow, uint32_t* oh, uint8_t** opx);   // jpeg_codec.cpp
namespace {

uint32_t be32(const uint8_t* p) {
    return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3];
}

uint8_t paeth(int a, int b, int c) {
    int p = a + b - c;
    int pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
    if (pa <= pb && pa <= pc) return (uint8_t)a;
    if (pb <= pc) return (uint8_t)b;
    return (uint8_t)c;
}

void decode(const uint8_t* data, size_t len, uint32_t want, uint32_t* ow, uint32_t* oh,
            uint8_t** opx) {
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};
    if (len >= 3 && data[0] == 0xff && data[1] == 0xd8 && data[2] == 0xff) {   // a JPEG: the other format textures come in
        decode_jpeg(data, len, want, ow, oh, opx);                             // (host/jpeg_codec.cpp)
        return;
    }
    if (len < 8 || memcmp(data, sig, 8) != 0) fail(PT_ERR_PARSE, "not a PNG or JPEG file");
    if (want != 1 && want != 3 && want != 4) fail(PT_ERR_INVALID, "want_channels must be 1, 3 or 4");
    size_t pos = 8;
    uint32_t w = 0, h = 0;
    int depth = 0, ctype = -1, interlace = 0;
    std::vector<uint8_t> idat, plte;
    std::vector<uint8_t> trns;   // tRNS: palette alphas (type 3) / the transparent grey or colour (types 0, 2), big-endian u16
    bool seen_ihdr = false, seen_iend = false;
    while (pos + 12 <= len && !seen_iend) {
        uint32_t clen = be32(data + pos);
        const uint8_t* type = data + pos + 4;
        if (pos + 12 + (size_t)clen > len) fail(PT_ERR_PARSE, "truncated PNG chunk");
        const uint8_t* body = data + pos + 8;
        if (!memcmp(type, "IHDR", 4)) {
            if (clen != 13) fail(PT_ERR_PARSE, "bad IHDR");
            w = be32(body);
            h = be32(body + 4);
            depth = body[8];
            ctype = body[9];
            interlace = body[12];
            seen_ihdr = true;
        } else if (!memcmp(type, "PLTE", 4)) {
            plte.assign(body, body + clen);
        } else if (!memcmp(type, "tRNS", 4)) {
            trns.assign(body, body + clen);
        } else if (!memcmp(type, "IDAT", 4)) {
            idat.insert(idat.end(), body, body + clen);
        } else if (!memcmp(type, "IEND", 4)) {
            seen_iend = true;
        }
        pos += 12 + (size_t)clen;
    }
    if (!seen_ihdr || w == 0 || h == 0) fail(PT_ERR_PARSE, "PNG without IHDR");
    if (interlace) fail(PT_ERR_UNSUPPORTED, "interlaced PNG is not supported");
    int channels;
    switch (ctype) {
        case 0: channels = 1; break;
        case 2: channels = 3; break;
        case 3: channels = 1; break;
        case 4: channels = 2; break;
        case 6: channels = 4; break;
        default: fail(PT_ERR_PARSE, "bad PNG colour type %d", ctype);
    }
    if (!(depth == 1 || depth == 2 || depth == 4 || depth == 8 || depth == 16) ||
        ((ctype == 2 || ctype == 4 || ctype == 6) && depth < 8) || (ctype == 3 && depth > 8))
        fail(PT_ERR_PARSE, "bad PNG bit depth %d for colour type %d", depth, ctype);
    if (ctype == 3 && plte.empty()) fail(PT_ERR_PARSE, "palette PNG without PLTE");

    size_t bpp_bits = (size_t)channels * depth;
    size_t stride = ((size_t)w * bpp_bits + 7) / 8;
    size_t bpp = bpp_bits < 8 ? 1 : bpp_bits / 8;  // filter byte distance
    // A damaged header must not drive the allocation: deflate expands at most ~1032 : 1, so the pixel data
    // the IDAT chunks can hold bounds the image; 2^31 bytes of raw scanlines is the format limit kept here.
    const size_t raw_size = (stride + 1) * (size_t)h;
    if (w > (1u << 24) || h > (1u << 24) || raw_size > ((size_t)1 << 31) || raw_size / 1100 > idat.size() + 1)
        fail(PT_ERR_PARSE, "PNG header claims %u x %u pixels, which its %zu bytes of image data cannot hold", w, h,
             idat.size());
    std::vector<uint8_t> raw(raw_size);
    uLongf rawlen = raw.size();
    int zr = uncompress(raw.data(), &rawlen, idat.data(), idat.size());
    if (zr != Z_OK || rawlen != raw.size()) fail(PT_ERR_PARSE, "PNG inflate failed (%d)", zr);

    // unfilter in place
    std::vector<uint8_t> zero(stride, 0);
    for (uint32_t y = 0; y < h; ++y) {
        uint8_t* row = raw.data() + (stride + 1) * (size_t)y;
        uint8_t ft = row[0];
        uint8_t* cur = row + 1;
        const uint8_t* up = y ? row - stride : zero.data();
        switch (ft) {
            case 0: break;
            case 1:
                for (size_t i = bpp; i < stride; ++i) cur[i] = (uint8_t)(cur[i] + cur[i - bpp]);
                break;
            case 2:
                for (size_t i = 0; i < stride; ++i) cur[i] = (uint8_t)(cur[i] + up[i]);
                break;
            case 3:
                for (size_t i = 0; i < stride; ++i) {
                    int a = i >= bpp ? cur[i - bpp] : 0;
                    cur[i] = (uint8_t)(cur[i] + ((a + up[i]) >> 1));
                }
                break;
            case 4:
                for (size_t i = 0; i < stride; ++i) {
                    int a = i >= bpp ? cur[i - bpp] : 0;
                    int c = i >= bpp ? up[i - bpp] : 0;
                    cur[i] = (uint8_t)(cur[i] + paeth(a, up[i], c));
                }
                break;
            default: fail(PT_ERR_PARSE, "bad PNG filter type %d", ft);
        }
    }

    // (owned until the end: a malformed palette index below throws)
    std::unique_ptr<uint8_t, void (*)(void*)> out_owner((uint8_t*)malloc((size_t)w * h * want), free);
    uint8_t* out = out_owner.get();
    if (!out) throw std::bad_alloc();
    auto sample = [&](const uint8_t* row, size_t idx) -> uint32_t {  // idx = sample index in row
        if (depth == 8) return row[idx];
        if (depth == 16) return ((uint32_t)row[2 * idx] << 8) | row[2 * idx + 1];
        size_t bit = idx * depth;
        uint32_t v = (row[bit >> 3] >> (8 - depth - (bit & 7))) & ((1u << depth) - 1);
        return v;
    };
    auto to8 = [&](uint32_t v) -> uint8_t {
        if (depth == 8) return (uint8_t)v;
        if (depth == 16) return (uint8_t)((v + 128) / 257);
        return (uint8_t)(v * 255 / ((1u << depth) - 1));
    };
    for (uint32_t y = 0; y < h; ++y) {
        const uint8_t* row = raw.data() + (stride + 1) * (size_t)y + 1;
        for (uint32_t x = 0; x < w; ++x) {
            uint8_t r, g, b, a = 255;
            bool is_gray = false;
            if (ctype == 4) a = to8(sample(row, (size_t)x * channels + 1));
            if (ctype == 6) a = to8(sample(row, (size_t)x * channels + 3));
            // tRNS (what the image crate's into_rgba8 carries for files without an alpha channel): one alpha per palette
            // entry, or ONE sample value (at the file's bit depth) that is fully transparent
            if (ctype == 0 || ctype == 4) {
                const uint32_t v = sample(row, (size_t)x * channels);
                r = g = b = to8(v);
                is_gray = true;
                if (ctype == 0 && trns.size() >= 2 && v == (((uint32_t)trns[0] << 8) | trns[1])) a = 0;
            } else if (ctype == 3) {
                uint32_t idx = sample(row, x);
                if ((size_t)idx * 3 + 2 >= plte.size()) fail(PT_ERR_PARSE, "palette index out of range");
                r = plte[idx * 3];
                g = plte[idx * 3 + 1];
                b = plte[idx * 3 + 2];
                if (idx < trns.size()) a = trns[idx];
            } else {
                const uint32_t vr = sample(row, (size_t)x * channels), vg = sample(row, (size_t)x * channels + 1),
                               vb = sample(row, (size_t)x * channels + 2);
                r = to8(vr);
                g = to8(vg);
                b = to8(vb);
                if (ctype == 2 && trns.size() >= 6 && vr == (((uint32_t)trns[0] << 8) | trns[1]) &&
                    vg == (((uint32_t)trns[2] << 8) | trns[3]) && vb == (((uint32_t)trns[4] << 8) | trns[5]))
                    a = 0;
            }
            uint8_t* o = out + ((size_t)y * w + x) * want;
            if (want >= 3) {
                o[0] = r;
                o[1] = g;
                o[2] = b;
                if (want == 4) o[3] = a;   // (into_rgba8: opaque when the file has neither an alpha channel nor tRNS)
            } else {
                o[0] = is_gray ? r : (uint8_t)((2126u * r + 7152u * g + 722u * b) / 10000u);
            }
        }
    }
    *ow = w;
    *oh = h;
    *opx = out_owner.release();
}

void put32(std::vector<uint8_t>& v, uint32_t x) {
    v.push_back((uint8_t)(x >> 24));
    v.push_back((uint8_t)(x >> 16));
    v.push_back((uint8_t)(x >> 8));
    v.push_back((uint8_t)x);
}

void chunk(std::vector<uint8_t>& out, const char* type, const uint8_t* body, size_t n) {
    put32(out, (uint32_t)n);
    size_t start = out.size();
    out.insert(out.end(), type, type + 4);
    out.insert(out.end(), body, body + n);
    uint32_t crc = (uint32_t)crc32(0L, out.data() + start, (uInt)(n + 4));
    put32(out, crc);
}

}  // namespace
}  // namespace pth

extern "C" {

int pth_png_decode(const uint8_t* data, size_t len, uint32_t want_channels, uint32_t* w,
                   uint32_t* h, uint8_t** pixels) {
    return pth::guarded([&] {
        if (!data || !w || !h || !pixels) pth::fail(PT_ERR_INVALID, "pth_png_decode: null argument");
        pth::decode(data, len, want_channels, w, h, pixels);
    });
}

int pth_png_read(const char* path, uint32_t want_channels, uint32_t* w, uint32_t* h,
                 uint8_t** pixels) {
    return pth::guarded([&] {
        if (!path) pth::fail(PT_ERR_INVALID, "pth_png_read: null path");
        std::ifstream f(path, std::ios::binary);
        if (!f) pth::fail(PT_ERR_IO, "%s: %s", path, strerror(errno));
        std::vector<uint8_t> buf((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
        pth::decode(buf.data(), buf.size(), want_channels, w, h, pixels);
    });
}

int pth_png_write_rgb8(const char* path, uint32_t w, uint32_t h, const uint8_t* rgb) {
    return pth::guarded([&] {
        if (!path || !rgb || !w || !h) pth::fail(PT_ERR_INVALID, "pth_png_write_rgb8: bad argument");
        size_t stride = (size_t)w * 3;
        std::vector<uint8_t> raw((stride + 1) * h);
        for (uint32_t y = 0; y < h; ++y) {
            raw[(stride + 1) * y] = 0;
            memcpy(&raw[(stride + 1) * y + 1], rgb + stride * y, stride);
        }
        // The zlib stream of IDAT, deflated in parallel (as pigz does): bands of rows compressed independently as raw
        // deflate, every band but the last ended with a sync flush (an empty stored block: byte-aligned), concatenated
        // behind one zlib header, the Adler-32 of the whole combined from the bands'.  A noisy 1080p render took 380 ms at
        // level 6 on one core - ten frames' worth of rendering; level 1 in bands takes ~15 ms on the MI355X host.
        const size_t band_rows = 32, n_bands = (h + band_rows - 1) / band_rows;
        std::vector<std::vector<uint8_t>> parts(n_bands);
        std::vector<uLong> adlers(n_bands);
        std::atomic<size_t> next{0};
        std::atomic<bool> failed{false};
        auto work = [&]() {
            try {
            for (size_t b; (b = next.fetch_add(1)) < n_bands;) {
                const size_t r0 = b * band_rows, r1 = std::min<size_t>(h, r0 + band_rows);
                const uint8_t* src = raw.data() + (stride + 1) * r0;
                const size_t len = (stride + 1) * (r1 - r0);
                adlers[b] = adler32(adler32(0L, Z_NULL, 0), src, (uInt)len);
                z_stream z{};
                if (deflateInit2(&z, 1, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) {
                    failed = true;
                    continue;
                }
                std::vector<uint8_t>& out_b = parts[b];
                out_b.resize(deflateBound(&z, (uLong)len) + 16);
                z.next_in = const_cast<Bytef*>(src);
                z.avail_in = (uInt)len;
                z.next_out = out_b.data();
                z.avail_out = (uInt)out_b.size();
                const int rc = deflate(&z, b + 1 == n_bands ? Z_FINISH : Z_SYNC_FLUSH);
                if ((b + 1 == n_bands ? rc != Z_STREAM_END : rc != Z_OK) || z.avail_in != 0) failed = true;
                out_b.resize(out_b.size() - z.avail_out);
                deflateEnd(&z);
            }
            } catch (...) {   // (out of memory in a band's buffer: a thread must not end the process)
                failed = true;
            }
        };
        {
            const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
            const size_t n_threads = std::min<size_t>({(size_t)hw, (size_t)32, n_bands});
            std::vector<std::thread> pool;
            for (size_t t = 1; t < n_threads; ++t) pool.emplace_back(work);
            work();
            for (auto& t : pool) t.join();
        }
        if (failed) pth::fail(PT_ERR_IO, "PNG deflate failed");
        std::vector<uint8_t> comp = {0x78, 0x01};
        uLong adler = adler32(0L, Z_NULL, 0);
        for (size_t b = 0; b < n_bands; ++b) {
            comp.insert(comp.end(), parts[b].begin(), parts[b].end());
            const size_t r0 = b * band_rows, r1 = std::min<size_t>(h, r0 + band_rows);
            adler = adler32_combine(adler, adlers[b], (z_off_t)((stride + 1) * (r1 - r0)));
        }
        for (int k = 3; k >= 0; --k) comp.push_back((uint8_t)(adler >> (8 * k)));
        const size_t clen = comp.size();
        std::vector<uint8_t> out = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};
        std::vector<uint8_t> ihdr;
        pth::put32(ihdr, w);
        pth::put32(ihdr, h);
        ihdr.insert(ihdr.end(), {8, 2, 0, 0, 0});
        pth::chunk(out, "IHDR", ihdr.data(), ihdr.size());
        pth::chunk(out, "IDAT", comp.data(), clen);
        pth::chunk(out, "IEND", nullptr, 0);
        std::ofstream f(path, std::ios::binary);
        if (!f) pth::fail(PT_ERR_IO, "%s: %s", path, strerror(errno));
        f.write((const char*)out.data(), (std::streamsize)out.size());
        if (!f) pth::fail(PT_ERR_IO, "%s: write failed", path);
    });
}

}  // extern "C"
