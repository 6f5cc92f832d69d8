// PNG codec of the clip I/O edge: counterpart of cv2.imread (utils.py:583-593) and
// cv2.imwrite (main.py:1165-1178) for the frames of a clip, on zlib only (libpng / OpenCV are not in the image).
// Host code; thread-safe and GIL-free (the Python side fans frames out over a thread pool: 8 GPUs x 70 frames/s is ~550
// PNGs/s to write).  Pixel layout on both sides is cv2's: uint8 [h, w, 3] in B, G, R order.
//   decode: 8/16-bit, colour types 0 (gray), 2 (RGB), 3 (palette), 4 (gray+alpha), 6 (RGBA), non-interlaced -> BGR8
//           (what cv2.imread(path) = IMREAD_COLOR returns: alpha dropped, 16-bit reduced to the high byte, gray replicated)
//   encode: 8-bit RGB, non-interlaced, one IDAT; filter per scanline by the minimum-sum-of-absolute-differences
//           heuristic (libpng's default strategy) or a fixed filter; lossless, so the decoded pixels are what cv2 would write.
#include "common.h"
#include <string.h>
#include <exception>
#include <vector>
#include <zlib.h>

namespace {

const uint8_t PNG_SIG[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
// limits of both directions of the codec: 32k x 32k covers any video frame, 2^28 pixels keeps every intermediate buffer (<= 8 bytes
// per pixel + one filter byte per row) below 4 GiB, the range zlib's one-shot uInt / uLong interfaces are used in
const int PNG_MAX_DIM = 32768;
const int64_t PNG_MAX_PIXELS = (int64_t)1 << 28;

inline uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
inline void put32(uint8_t* p, uint32_t v) { p[0] = v >> 24; p[1] = v >> 16; p[2] = v >> 8; p[3] = v; }

inline int paeth(int a, int b, int c)
{
    const int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

// filter one scanline (row, prev: raw bytes; prev == NULL for the first line) into out; returns sum |signed byte|
int64_t filter_row(int type, const uint8_t* row, const uint8_t* prev, int n, int bpp, uint8_t* out)
{
    int64_t sum = 0;
    for (int i = 0; i < n; ++i) {
        const int a = i >= bpp ? row[i - bpp] : 0, b = prev ? prev[i] : 0, c = (prev && i >= bpp) ? prev[i - bpp] : 0;
        int pred;
        switch (type) {
        case 0: pred = 0; break;
        case 1: pred = a; break;
        case 2: pred = b; break;
        case 3: pred = (a + b) >> 1; break;
        default: pred = paeth(a, b, c); break;
        }
        const uint8_t v = (uint8_t)(row[i] - pred);
        out[i] = v;
        sum += v < 128 ? v : 256 - v;
    }
    return sum;
}

void write_chunk(std::vector<uint8_t>& o, const char* type, const uint8_t* data, size_t n)
{
    const size_t at = o.size();
    o.resize(at + 12 + n);
    put32(&o[at], (uint32_t)n);
    memcpy(&o[at + 4], type, 4);
    if (n) memcpy(&o[at + 8], data, n);
    put32(&o[at + 8 + n], (uint32_t)crc32(crc32(0, nullptr, 0), &o[at + 4], (uInt)(n + 4)));
}

}  // namespace

extern "C" int64_t demfi_png_encode_bound(int h, int w)
{
    if (h <= 0 || w <= 0 || h > PNG_MAX_DIM || w > PNG_MAX_DIM || (int64_t)h * w > PNG_MAX_PIXELS) return 0;
    const uLong raw = (uLong)h * ((uLong)w * 3 + 1);
    return (int64_t)compressBound(raw) + 4096;
}

// bgr: uint8 [h,w,3] with row stride `stride` bytes.  level: zlib 0..9; filter: -1 = adaptive (min-sum heuristic over the 5
// filters, libpng's default), 0..4 = fixed (1 = Sub is what OpenCV's writer sets); strategy: -1 = Z_RLE for level <= 3 (OpenCV's
// default), else a zlib strategy constant.  level 1 / Sub / RLE: ~15 ms per 720p frame and core.
static int png_encode_impl(const uint8_t* bgr, int h, int w, int64_t stride, int level, int filter, int strategy, uint8_t* out,
                           int64_t out_cap, int64_t* out_bytes)
{
    if (h > PNG_MAX_DIM || w > PNG_MAX_DIM || (int64_t)h * w > PNG_MAX_PIXELS)
        return demfi_set_error(DEMFI_ERR_ARG, "demfi_png_encode: %d x %d is outside the supported range", h, w);
    if (!bgr || !out || !out_bytes || h <= 0 || w <= 0 || stride < (int64_t)w * 3 || level < 0 || level > 9 || filter < -1 || filter > 4 ||
        strategy > 4)
        return demfi_set_error(DEMFI_ERR_ARG, "demfi_png_encode: bad arguments");
    const int n = w * 3, bpp = 3;
    std::vector<uint8_t> raw((size_t)h * (n + 1));
    std::vector<uint8_t> cur(n), prev(n), cand(n), best(n);
    for (int y = 0; y < h; ++y) {
        const uint8_t* src = bgr + (int64_t)y * stride;
        for (int x = 0; x < w; ++x) { cur[3 * x] = src[3 * x + 2]; cur[3 * x + 1] = src[3 * x + 1]; cur[3 * x + 2] = src[3 * x]; }   // BGR -> RGB
        const uint8_t* pv = y ? prev.data() : nullptr;
        uint8_t* dst = &raw[(size_t)y * (n + 1)];
        if (filter == 1) {                                       // Sub: OpenCV's default; branch-free, vectorisable
            dst[0] = 1;
            const uint8_t* r = cur.data();
            uint8_t* q = dst + 1;
            q[0] = r[0]; q[1] = r[1]; q[2] = r[2];
            for (int i = 3; i < n; ++i) q[i] = (uint8_t)(r[i] - r[i - 3]);
        } else if (filter == 2 && pv) {
            dst[0] = 2;
            const uint8_t* r = cur.data();
            uint8_t* q = dst + 1;
            for (int i = 0; i < n; ++i) q[i] = (uint8_t)(r[i] - pv[i]);
        } else if (filter >= 0) {
            dst[0] = (uint8_t)filter;
            filter_row(filter, cur.data(), pv, n, bpp, dst + 1);
        } else {
            int64_t bs = -1;
            int bt = 0;
            for (int t = 0; t < 5; ++t) {
                const int64_t s = filter_row(t, cur.data(), pv, n, bpp, cand.data());
                if (bs < 0 || s < bs) { bs = s; bt = t; best.swap(cand); }
            }
            dst[0] = (uint8_t)bt;
            memcpy(dst + 1, best.data(), n);
        }
        prev.swap(cur);
    }
    // deflate: Z_RLE at a low level is what OpenCV's PNG writer uses by default (fast: the filtered rows are run-friendly);
    // strategy < 0 = Z_RLE for level <= 3, Z_DEFAULT_STRATEGY above
    const int strat = strategy >= 0 ? strategy : (level <= 3 ? Z_RLE : Z_DEFAULT_STRATEGY);
    z_stream zs;
    memset(&zs, 0, sizeof(zs));
    if (deflateInit2(&zs, level, Z_DEFLATED, 15, 8, strat) != Z_OK) return demfi_set_error(DEMFI_ERR_ARG, "demfi_png_encode: deflateInit2 failed");
    uLongf zn = deflateBound(&zs, (uLong)raw.size());
    std::vector<uint8_t> z(zn);
    zs.next_in = raw.data(); zs.avail_in = (uInt)raw.size();
    zs.next_out = z.data(); zs.avail_out = (uInt)zn;
    const int zr = deflate(&zs, Z_FINISH);
    zn = zs.total_out;
    deflateEnd(&zs);
    if (zr != Z_STREAM_END) return demfi_set_error(DEMFI_ERR_ARG, "demfi_png_encode: deflate failed (%d)", zr);
    std::vector<uint8_t> o(PNG_SIG, PNG_SIG + 8);
    uint8_t ihdr[13];
    put32(ihdr, (uint32_t)w);
    put32(ihdr + 4, (uint32_t)h);
    ihdr[8] = 8; ihdr[9] = 2; ihdr[10] = 0; ihdr[11] = 0; ihdr[12] = 0;       // 8-bit, RGB, deflate, adaptive filtering, no interlace
    write_chunk(o, "IHDR", ihdr, 13);
    write_chunk(o, "IDAT", z.data(), zn);
    write_chunk(o, "IEND", nullptr, 0);
    *out_bytes = (int64_t)o.size();
    if ((int64_t)o.size() > out_cap) return demfi_set_error(DEMFI_ERR_ARG, "demfi_png_encode: output buffer of %lld B < %lld B", (long long)out_cap, (long long)o.size());
    memcpy(out, o.data(), o.size());
    return DEMFI_OK;
}

// Header only: height / width of a PNG (so the caller can size the output of demfi_png_decode).
extern "C" int demfi_png_info(const uint8_t* data, int64_t n, int* h, int* w)
{
    if (!data || n < 33 || memcmp(data, PNG_SIG, 8) != 0 || memcmp(data + 12, "IHDR", 4) != 0)
        return demfi_set_error(DEMFI_ERR_ARG, "demfi_png_info: not a PNG");
    // the header is untrusted input: a crafted IHDR must come back as an error, not as an allocation the size of the address space
    const uint32_t uw = be32(data + 16), uh = be32(data + 20);
    if (uw == 0 || uh == 0 || uw > (uint32_t)PNG_MAX_DIM || uh > (uint32_t)PNG_MAX_DIM || (int64_t)uw * uh > PNG_MAX_PIXELS)
        return demfi_set_error(DEMFI_ERR_ARG, "demfi_png_info: image of %u x %u pixels is outside the supported range (<= %d per side, <= %lld pixels)",
                               uw, uh, PNG_MAX_DIM, (long long)PNG_MAX_PIXELS);
    if (w) *w = (int)uw;
    if (h) *h = (int)uh;
    return DEMFI_OK;
}

static int png_decode_impl(const uint8_t* data, int64_t n, uint8_t* bgr, int64_t stride, int h_expect, int w_expect)
{
    int h = 0, w = 0;
    int st = demfi_png_info(data, n, &h, &w);
    if (st < 0) return st;
    if (!bgr || h != h_expect || w != w_expect || stride < (int64_t)w * 3)
        return demfi_set_error(DEMFI_ERR_ARG, "demfi_png_decode: image is %dx%d, caller expects %dx%d", h, w, h_expect, w_expect);
    const int depth = data[24], ctype = data[25], interlace = data[28];
    if (interlace != 0 || (depth != 8 && depth != 16) || !(ctype == 0 || ctype == 2 || ctype == 3 || ctype == 4 || ctype == 6) ||
        (ctype == 3 && depth != 8))
        return demfi_set_error(DEMFI_ERR_ARG, "demfi_png_decode: unsupported PNG (depth %d, colour type %d, interlace %d)", depth, ctype, interlace);
    const int chans = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : 4;
    const int bps = depth / 8, bpp = chans * bps;
    const int64_t rowb = (int64_t)w * bpp;
    std::vector<uint8_t> z, pal;
    int64_t pos = 8;
    bool end = false;
    while (!end && pos + 12 <= n) {
        const uint32_t len = be32(data + pos);
        const uint8_t* type = data + pos + 4;
        if (pos + 12 + (int64_t)len > n) return demfi_set_error(DEMFI_ERR_ARG, "demfi_png_decode: truncated chunk");
        if (be32(data + pos + 8 + len) != (uint32_t)crc32(crc32(0, nullptr, 0), type, len + 4))
            return demfi_set_error(DEMFI_ERR_ARG, "demfi_png_decode: CRC mismatch in %.4s", (const char*)type);
        if (!memcmp(type, "IDAT", 4)) z.insert(z.end(), data + pos + 8, data + pos + 8 + len);
        else if (!memcmp(type, "PLTE", 4)) pal.assign(data + pos + 8, data + pos + 8 + len);
        else if (!memcmp(type, "IEND", 4)) end = true;
        pos += 12 + len;
    }
    if (!end || z.empty()) return demfi_set_error(DEMFI_ERR_ARG, "demfi_png_decode: no IDAT / IEND");
    if (ctype == 3 && pal.size() < 3) return demfi_set_error(DEMFI_ERR_ARG, "demfi_png_decode: palette image without PLTE");
    std::vector<uint8_t> raw((size_t)h * (rowb + 1));
    uLongf rn = (uLongf)raw.size();
    if (uncompress(raw.data(), &rn, z.data(), (uLong)z.size()) != Z_OK || rn != raw.size())
        return demfi_set_error(DEMFI_ERR_ARG, "demfi_png_decode: inflate failed or size mismatch");
    const uint8_t* prev = nullptr;
    for (int y = 0; y < h; ++y) {
        uint8_t* line = &raw[(size_t)y * (rowb + 1)];
        const int ft = line[0];
        uint8_t* r = line + 1;
        if (ft > 4) return demfi_set_error(DEMFI_ERR_ARG, "demfi_png_decode: bad filter type %d", ft);
        for (int64_t i = 0; i < rowb; ++i) {
            const int a = i >= bpp ? r[i - bpp] : 0, b = prev ? prev[i] : 0, c = (prev && i >= bpp) ? prev[i - bpp] : 0;
            int pred = 0;
            if (ft == 1) pred = a;
            else if (ft == 2) pred = b;
            else if (ft == 3) pred = (a + b) >> 1;
            else if (ft == 4) pred = paeth(a, b, c);
            r[i] = (uint8_t)(r[i] + pred);
        }
        prev = r;
        uint8_t* dst = bgr + (int64_t)y * stride;
        for (int x = 0; x < w; ++x) {
            const uint8_t* px = r + (int64_t)x * bpp;                // 16-bit samples are big endian: the high byte comes first
            uint8_t R, G, B;
            if (ctype == 2 || ctype == 6) { R = px[0]; G = px[bps]; B = px[2 * bps]; }
            else if (ctype == 3) {
                const size_t k = (size_t)px[0] * 3;
                if (k + 2 >= pal.size()) return demfi_set_error(DEMFI_ERR_ARG, "demfi_png_decode: palette index out of range");
                R = pal[k]; G = pal[k + 1]; B = pal[k + 2];
            } else { R = G = B = px[0]; }
            dst[3 * x] = B; dst[3 * x + 1] = G; dst[3 * x + 2] = R;
        }
    }
    return DEMFI_OK;
}

// The C ABI never lets a C++ exception out (std::bad_alloc / std::length_error of the std::vectors above would otherwise reach
// std::terminate through the extern "C" frame and kill the host process).
extern "C" int demfi_png_encode(const uint8_t* bgr, int h, int w, int64_t stride, int level, int filter, int strategy, uint8_t* out,
                                int64_t out_cap, int64_t* out_bytes)
{
    try {
        return png_encode_impl(bgr, h, w, stride, level, filter, strategy, out, out_cap, out_bytes);
    } catch (const std::exception& e) {
        return demfi_set_error(DEMFI_ERR_ARG, "demfi_png_encode: %s", e.what());
    } catch (...) {
        return demfi_set_error(DEMFI_ERR_ARG, "demfi_png_encode: unknown exception");
    }
}

extern "C" int demfi_png_decode(const uint8_t* data, int64_t n, uint8_t* bgr, int64_t stride, int h_expect, int w_expect)
{
    try {
        return png_decode_impl(data, n, bgr, stride, h_expect, w_expect);
    } catch (const std::exception& e) {
        return demfi_set_error(DEMFI_ERR_ARG, "demfi_png_decode: %s", e.what());
    } catch (...) {
        return demfi_set_error(DEMFI_ERR_ARG, "demfi_png_decode: unknown exception");
    }
}
