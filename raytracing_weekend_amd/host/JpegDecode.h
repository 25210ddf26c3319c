// JpegDecode.h — baseline JPEG decoder for image textures (the reference reads assets/earthmap.jpg through stb_image,
// texture/ioTexture.h:225-262, external/stb_image.h; this is an independent implementation written from the standard,
// ITU-T T.81, not a copy of stb):
//   * baseline sequential DCT (SOF0), 8-bit samples, Huffman coding, 1 component (grey) or 3 (YCbCr, JFIF)
//   * any sampling factors 1..4 per component; chroma is brought to full resolution with the usual triangle filter
//     ("fancy upsampling": 3/4 nearer + 1/4 farther sample per axis), edge samples replicated
//   * restart intervals (DRI / RSTn), 8- and 16-bit quantisation tables, padding bytes and fill 0xFF before markers
//   * inverse DCT straight from the definition in double precision (an image texture is decoded once per scene)
// Progressive, arithmetic-coded, lossless and 12-bit files are refused with a message.
// Output: width, height, tightly packed RGB8, rows top to bottom.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

namespace rtwhost {
namespace jpeg {

struct Huff {
    // canonical code tables, T.81 annex C / F.2.2.3: mincode / maxcode / valptr per code length
    int mincode[17], maxcode[18], valptr[17];
    uint8_t vals[256];
    bool present = false;
    void build(const uint8_t counts[16], const uint8_t* symbols, int n) {
        memcpy(vals, symbols, static_cast<size_t>(n));
        int code = 0, k = 0;
        for (int len = 1; len <= 16; len++) {
            valptr[len] = k;
            mincode[len] = code;
            code += counts[len - 1];
            k += counts[len - 1];
            maxcode[len] = counts[len - 1] ? code - 1 : -1;
            code <<= 1;
        }
        maxcode[17] = 0x7fffffff;
        present = true;
    }
};

struct Component {
    int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
    int pred = 0;                 // DC predictor
    int bw = 0, bh = 0;           // size in blocks (padded to whole MCUs)
    std::vector<uint8_t> plane;   // bw*8 x bh*8 samples
};

class Decoder {
public:
    bool decode(const uint8_t* data, size_t size, int& width, int& height, std::vector<uint8_t>& rgb, std::string& err) {
        d_ = data; n_ = size; pos_ = 0;
        if (size < 4 || data[0] != 0xFF || data[1] != 0xD8) return fail(err, "not a JPEG file (no SOI)");
        pos_ = 2;
        bool have_frame = false;
        for (;;) {
            int m = nextMarker();
            if (m < 0) return fail(err, "truncated file (no SOS)");
            if (m == 0xD9) return fail(err, "EOI before any scan");
            if (m == 0x01 || (m >= 0xD0 && m <= 0xD7)) continue;
            if (pos_ + 2 > n_) return fail(err, "truncated segment");
            const size_t len = (size_t(d_[pos_]) << 8) | d_[pos_ + 1];
            if (len < 2 || pos_ + len > n_) return fail(err, "bad segment length");
            const uint8_t* seg = d_ + pos_ + 2;
            const size_t sl = len - 2;
            if (m == 0xDB) { if (!readDQT(seg, sl)) return fail(err, "bad DQT"); }
            else if (m == 0xC4) { if (!readDHT(seg, sl)) return fail(err, "bad DHT"); }
            else if (m == 0xC0 || m == 0xC1) {
                // (SOF1, extended sequential, differs from baseline only in table limits when the precision is 8)
                if (!readSOF(seg, sl, err)) return false;
                have_frame = true;
            } else if (m == 0xC2) return fail(err, "progressive JPEG is not supported (baseline only)");
            else if (m == 0xC3 || (m >= 0xC5 && m <= 0xCF && m != 0xC8 && m != 0xCC)) return fail(err, "unsupported JPEG process (lossless / hierarchical / arithmetic)");
            else if (m == 0xCC) return fail(err, "arithmetic coding is not supported");
            else if (m == 0xDD) { if (sl < 2) return fail(err, "bad DRI"); restart_ = (seg[0] << 8) | seg[1]; }
            else if (m == 0xDA) {
                if (!have_frame) return fail(err, "SOS before SOF");
                if (!readSOS(seg, sl, err)) return false;
                pos_ += len;
                if (!decodeScan(err)) return false;
                break;  // baseline: one scan carries every component (non-interleaved multi-scan files are refused in readSOS)
            }
            pos_ += len;
        }
        toRGB(rgb);
        width = w_; height = h_;
        return true;
    }

private:
    const uint8_t* d_ = nullptr;
    size_t n_ = 0, pos_ = 0;
    int w_ = 0, h_ = 0, ncomp_ = 0, hmax_ = 1, vmax_ = 1, restart_ = 0;
    uint16_t qt_[4][64];
    bool qt_ok_[4] = {false, false, false, false};
    Huff dc_[4], ac_[4];
    Component comp_[3];
    // entropy decoder state
    uint32_t bits_ = 0;
    int nbits_ = 0;
    int pad_bits_ = 0;  // bits at the tail of the buffer that did not come from the file (fed after its end or after a marker)
    bool hit_marker_ = false;

    static bool fail(std::string& err, const char* msg) { err = msg; return false; }
    int nextMarker() {
        while (pos_ + 1 < n_) {
            if (d_[pos_] != 0xFF) { pos_++; continue; }
            while (pos_ < n_ && d_[pos_] == 0xFF) pos_++;  // fill bytes
            if (pos_ >= n_) return -1;
            const int m = d_[pos_++];
            if (m != 0) return m;
        }
        return -1;
    }
    bool readDQT(const uint8_t* s, size_t n) {
        static const uint8_t zz[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                                       35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
        size_t i = 0;
        while (i < n) {
            const int pq = s[i] >> 4, tq = s[i] & 15;
            i++;
            if (tq > 3 || pq > 1 || i + (pq ? 128u : 64u) > n) return false;
            for (int k = 0; k < 64; k++) {
                qt_[tq][zz[k]] = pq ? static_cast<uint16_t>((s[i] << 8) | s[i + 1]) : s[i];  // stored in natural (row-major) order
                i += pq ? 2 : 1;
            }
            qt_ok_[tq] = true;
        }
        return true;
    }
    bool readDHT(const uint8_t* s, size_t n) {
        size_t i = 0;
        while (i < n) {
            if (i + 17 > n) return false;
            const int tc = s[i] >> 4, th = s[i] & 15;
            if (tc > 1 || th > 3) return false;
            int total = 0;
            for (int k = 0; k < 16; k++) total += s[i + 1 + k];
            if (total > 256 || i + 17 + static_cast<size_t>(total) > n) return false;
            (tc ? ac_[th] : dc_[th]).build(s + i + 1, s + i + 17, total);
            i += 17 + static_cast<size_t>(total);
        }
        return true;
    }
    bool readSOF(const uint8_t* s, size_t n, std::string& err) {
        if (n < 6) return fail(err, "bad SOF");
        if (s[0] != 8) return fail(err, "only 8-bit JPEG samples are supported");
        h_ = (s[1] << 8) | s[2];
        w_ = (s[3] << 8) | s[4];
        ncomp_ = s[5];
        if (w_ <= 0 || h_ <= 0 || w_ > 32768 || h_ > 32768) return fail(err, "unsupported JPEG size");
        if ((ncomp_ != 1 && ncomp_ != 3) || n < 6 + 3 * static_cast<size_t>(ncomp_)) return fail(err, "only 1- or 3-component JPEG files are supported");
        hmax_ = vmax_ = 1;
        for (int c = 0; c < ncomp_; c++) {
            Component& k = comp_[c];
            k.id = s[6 + 3 * c]; k.h = s[7 + 3 * c] >> 4; k.v = s[7 + 3 * c] & 15; k.tq = s[8 + 3 * c];
            if (k.h < 1 || k.h > 4 || k.v < 1 || k.v > 4 || k.tq > 3) return fail(err, "bad component description");
            if (k.h > hmax_) hmax_ = k.h;
            if (k.v > vmax_) vmax_ = k.v;
        }
        const int mcux = (w_ + 8 * hmax_ - 1) / (8 * hmax_), mcuy = (h_ + 8 * vmax_ - 1) / (8 * vmax_);
        // A coded 8x8 block takes at least two bits (a DC and an end-of-block code of one bit each), so a file of n bytes
        // cannot describe more than 4 n blocks: a header that declares more is refused before anything is allocated for it
        size_t blocks = 0;
        for (int c = 0; c < ncomp_; c++) {
            Component& k = comp_[c];
            if (hmax_ % k.h || vmax_ % k.v) return fail(err, "fractional sampling ratios are not supported");
            k.bw = mcux * k.h; k.bh = mcuy * k.v;
            blocks += static_cast<size_t>(k.bw) * k.bh;
        }
        if (blocks > 4 * n_) return fail(err, "JPEG header declares more blocks than the file can hold");
        for (int c = 0; c < ncomp_; c++) comp_[c].plane.assign(static_cast<size_t>(comp_[c].bw) * 8 * comp_[c].bh * 8, 0);
        return true;
    }
    bool readSOS(const uint8_t* s, size_t n, std::string& err) {
        if (n < 1 || s[0] != ncomp_ || n < 1 + 2 * static_cast<size_t>(ncomp_) + 3) return fail(err, "scans that carry only some of the components are not supported");
        for (int i = 0; i < ncomp_; i++) {
            const int id = s[1 + 2 * i];
            int c = -1;
            for (int j = 0; j < ncomp_; j++) if (comp_[j].id == id) c = j;
            if (c != i) return fail(err, "unexpected component order in SOS");
            comp_[c].td = s[2 + 2 * i] >> 4; comp_[c].ta = s[2 + 2 * i] & 15;
            if (comp_[c].td > 3 || comp_[c].ta > 3 || !dc_[comp_[c].td].present || !ac_[comp_[c].ta].present || !qt_ok_[comp_[c].tq]) return fail(err, "scan refers to a missing table");
        }
        return true;
    }
    // ---- entropy-coded segment
    void fillBits() {
        while (nbits_ <= 24) {
            int b = 0;
            bool real = false;
            if (!hit_marker_ && pos_ < n_) {
                b = d_[pos_];
                real = true;
                if (b == 0xFF) {
                    const int b2 = pos_ + 1 < n_ ? d_[pos_ + 1] : 0xD9;
                    if (b2 == 0) pos_ += 2;              // stuffed zero
                    else { hit_marker_ = true; b = 0; real = false; }  // a marker: zeros from here on
                } else pos_++;
            }
            if (!real) pad_bits_ += 8;  // not from the file; a valid stream never consumes these (decodeBlock checks)
            bits_ |= static_cast<uint32_t>(b) << (24 - nbits_);
            nbits_ += 8;
        }
    }
    int getBits(int n) {
        if (n == 0) return 0;
        if (nbits_ < n) fillBits();
        const int v = static_cast<int>(bits_ >> (32 - n));
        bits_ <<= n; nbits_ -= n;
        return v;
    }
    int decodeSymbol(const Huff& h) {
        int code = 0;
        for (int len = 1; len <= 16; len++) {
            code = (code << 1) | getBits(1);
            if (h.maxcode[len] >= 0 && code <= h.maxcode[len] && code >= h.mincode[len]) return h.vals[h.valptr[len] + code - h.mincode[len]];
        }
        return -1;
    }
    static int extend(int v, int t) { return t == 0 ? 0 : (v < (1 << (t - 1)) ? v - (1 << t) + 1 : v); }  // T.81 F.2.2.1

    bool decodeBlock(Component& k, int coef[64]) {
        static const uint8_t zz[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                                       35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
        memset(coef, 0, 64 * sizeof(int));
        const int t = decodeSymbol(dc_[k.td]);
        if (t < 0 || t > 11) return false;
        k.pred += extend(getBits(t), t);
        if (k.pred < -16384 || k.pred > 16384) return false;  // 8-bit samples keep the quantised DC within +-2048
        coef[0] = k.pred * qt_[k.tq][0];
        for (int i = 1; i < 64;) {
            const int rs = decodeSymbol(ac_[k.ta]);
            if (rs < 0) return false;
            const int r = rs >> 4, s = rs & 15;
            if (s == 0) {
                if (r == 15) { i += 16; continue; }
                break;  // end of block
            }
            i += r;
            if (i > 63 || s > 10) return false;  // (8-bit samples: AC magnitude categories 1..10, T.81 F.1.2.2)
            coef[zz[i]] = extend(getBits(s), s) * qt_[k.tq][zz[i]];
            i++;
        }
        return nbits_ >= pad_bits_;  // false: the block consumed bits the file does not hold (truncated or a marker inside the data)
    }
    static void idct(const int coef[64], uint8_t* out, int stride) {
        // s(y,x) = 1/4 sum_u sum_v C(u) C(v) S(v,u) cos((2x+1)u pi/16) cos((2y+1)v pi/16) + 128   (T.81 A.3.3)
        struct Basis {  // function-local static with a constructor: initialised once, thread-safe (two scenes may load at once)
            double b[8][8];
            Basis() {
                for (int x = 0; x < 8; x++)
                    for (int u = 0; u < 8; u++) b[x][u] = (u == 0 ? std::sqrt(0.5) : 1.0) * std::cos((2 * x + 1) * u * 3.14159265358979323846 / 16.0) * 0.5;
            }
        };
        static const Basis table;
        const double (*basis)[8] = table.b;
        double tmp[64];
        for (int v = 0; v < 8; v++)
            for (int x = 0; x < 8; x++) {
                double a = 0;
                for (int u = 0; u < 8; u++) a += basis[x][u] * coef[v * 8 + u];
                tmp[v * 8 + x] = a;
            }
        for (int y = 0; y < 8; y++)
            for (int x = 0; x < 8; x++) {
                double a = 0;
                for (int v = 0; v < 8; v++) a += basis[y][v] * tmp[v * 8 + x];
                const long r = std::lround(a + 128.0);
                out[y * stride + x] = static_cast<uint8_t>(r < 0 ? 0 : (r > 255 ? 255 : r));
            }
    }
    bool decodeScan(std::string& err) {
        const int mcux = comp_[0].bw / comp_[0].h, mcuy = comp_[0].bh / comp_[0].v;
        bits_ = 0; nbits_ = 0; pad_bits_ = 0; hit_marker_ = false;
        int until_restart = restart_;
        int coef[64];
        for (int my = 0; my < mcuy; my++)
            for (int mx = 0; mx < mcux; mx++) {
                if (restart_ && until_restart == 0) {
                    // byte-align, expect RSTn, reset the predictors (T.81 F.2.1.3.1 / E.2.4)
                    bits_ = 0; nbits_ = 0; pad_bits_ = 0; hit_marker_ = false;
                    const int m = nextMarker();
                    if (m < 0xD0 || m > 0xD7) return fail(err, "missing restart marker");
                    for (int c = 0; c < ncomp_; c++) comp_[c].pred = 0;
                    until_restart = restart_;
                }
                for (int c = 0; c < ncomp_; c++) {
                    Component& k = comp_[c];
                    for (int by = 0; by < k.v; by++)
                        for (int bx = 0; bx < k.h; bx++) {
                            if (!decodeBlock(k, coef)) return fail(err, "corrupt or truncated entropy-coded data");
                            const int stride = k.bw * 8;
                            idct(coef, k.plane.data() + static_cast<size_t>(my * k.v + by) * 8 * stride + static_cast<size_t>(mx * k.h + bx) * 8, stride);
                        }
                }
                if (restart_) until_restart--;
            }
        return true;
    }
    // full-resolution sample of component k at pixel (x, y): triangle filter over the component's own grid
    // (sample centres of a component subsampled by f sit at (i + 0.5) * f - 0.5 in pixel units)
    static double sampleAt(const Component& k, int fx, int fy, int x, int y, int w, int h) {
        const int cw = (w + fx - 1) / fx, ch = (h + fy - 1) / fy, stride = k.bw * 8;
        auto axis = [](int p, int f, int n, int& i0, int& i1, double& t) {
            if (f == 1) { i0 = i1 = p; t = 0; return; }
            const double c = (p + 0.5) / f - 0.5;
            const double fl = std::floor(c);
            i0 = static_cast<int>(fl); i1 = i0 + 1; t = c - fl;
            if (i0 < 0) i0 = 0;
            if (i1 < 0) i1 = 0;
            if (i0 > n - 1) i0 = n - 1;
            if (i1 > n - 1) i1 = n - 1;
        };
        int x0, x1, y0, y1;
        double tx, ty;
        axis(x, fx, cw, x0, x1, tx);
        axis(y, fy, ch, y0, y1, ty);
        const uint8_t* p = k.plane.data();
        const double a = p[static_cast<size_t>(y0) * stride + x0] * (1 - tx) + p[static_cast<size_t>(y0) * stride + x1] * tx;
        const double b = p[static_cast<size_t>(y1) * stride + x0] * (1 - tx) + p[static_cast<size_t>(y1) * stride + x1] * tx;
        return a * (1 - ty) + b * ty;
    }
    void toRGB(std::vector<uint8_t>& rgb) const {
        rgb.resize(static_cast<size_t>(w_) * h_ * 3);
        auto clamp8 = [](double v) { const long r = std::lround(v); return static_cast<uint8_t>(r < 0 ? 0 : (r > 255 ? 255 : r)); };
        for (int y = 0; y < h_; y++)
            for (int x = 0; x < w_; x++) {
                uint8_t* o = rgb.data() + (static_cast<size_t>(y) * w_ + x) * 3;
                const double Y = sampleAt(comp_[0], hmax_ / comp_[0].h, vmax_ / comp_[0].v, x, y, w_, h_);
                if (ncomp_ == 1) { o[0] = o[1] = o[2] = clamp8(Y); continue; }
                const double cb = sampleAt(comp_[1], hmax_ / comp_[1].h, vmax_ / comp_[1].v, x, y, w_, h_) - 128.0;
                const double cr = sampleAt(comp_[2], hmax_ / comp_[2].h, vmax_ / comp_[2].v, x, y, w_, h_) - 128.0;
                o[0] = clamp8(Y + 1.402 * cr);                          // JFIF 1.02, "Conversion to and from RGB"
                o[1] = clamp8(Y - 0.344136286 * cb - 0.714136286 * cr);
                o[2] = clamp8(Y + 1.772 * cb);
            }
    }
};

}  // namespace jpeg

inline bool decodeJpeg(const uint8_t* data, size_t size, int& width, int& height, std::vector<uint8_t>& rgb, std::string& err) {
    jpeg::Decoder d;
    return d.decode(data, size, width, height, rgb, err);
}

}  // namespace rtwhost
