// FLAC decoding for libnisqa_ingest.so: what lb.load(path, sr=None) (reference nisqa/NISQA_lib.py:2299-2304) gets from
// soundfile / libsndfile for a .flac file -- the stream's integer samples, which soundfile then scales by 1 / 2^(bits - 1).
// The format is a third-party specification (xiph.org "FLAC format", the one libFLAC 1.3 under libsndfile 1.0.31 implements;
// neither is in /root/reference): STREAMINFO, frame headers (CRC-8), subframes CONSTANT / VERBATIM / FIXED (orders 0-4) / LPC
// (orders 1-32) with partitioned Rice residuals (4- or 5-bit parameters, escape partitions), wasted bits, the three stereo
// decorrelations, the CRC-16 of every frame and the MD5 of the decoded samples.  A stream is accepted only when EVERY frame's
// CRC-8 and CRC-16 hold, the decoded length equals STREAMINFO's and, when STREAMINFO carries one, the MD5 of the decoded
// samples matches: a decoder that misreads a file fails ('Could not load file ...', like any unreadable file) instead of
// feeding the network wrong audio.  Limits: 4-24 bits per sample, 1-8 channels, one sample format per stream.
#pragma once
#include <cstring>
#include <stdint.h>
#include <vector>

namespace nqflac {

// ---- MD5 (RFC 1321) of the decoded samples, as the encoder computed it: interleaved, little-endian, (bits + 7) / 8 bytes each ----
class Md5 {
public:
    Md5() { s_[0] = 0x67452301u; s_[1] = 0xefcdab89u; s_[2] = 0x98badcfeu; s_[3] = 0x10325476u; }
    void add(const uint8_t* p, size_t n) {
        len_ += n;
        while (n > 0) {
            if (fill_ == 0 && n >= 64) { block(p); p += 64; n -= 64; continue; }
            const size_t take = 64 - fill_ < n ? 64 - fill_ : n;
            std::memcpy(buf_ + fill_, p, take);
            fill_ += take; p += take; n -= take;
            if (fill_ == 64) { block(buf_); fill_ = 0; }
        }
    }
    void finish(uint8_t out[16]) {
        const uint64_t bits = len_ * 8;
        const uint8_t one = 0x80, zero = 0;
        add(&one, 1);
        while (fill_ != 56) add(&zero, 1);
        uint8_t lb[8];
        for (int i = 0; i < 8; ++i) lb[i] = (uint8_t)(bits >> (8 * i));
        add(lb, 8);
        for (int i = 0; i < 16; ++i) out[i] = (uint8_t)(s_[i >> 2] >> (8 * (i & 3)));
    }

private:
    static inline uint32_t rol(uint32_t x, int c) { return (x << c) | (x >> (32 - c)); }
    void block(const uint8_t* p) {
        uint32_t m[16];
        std::memcpy(m, p, 64);                       // (little-endian host: x86-64)
        uint32_t a = s_[0], b = s_[1], c = s_[2], d = s_[3];
#define NQ_MD5_F(x, y, z) ((z) ^ ((x) & ((y) ^ (z))))
#define NQ_MD5_G(x, y, z) ((y) ^ ((z) & ((x) ^ (y))))
#define NQ_MD5_H(x, y, z) ((x) ^ (y) ^ (z))
#define NQ_MD5_I(x, y, z) ((y) ^ ((x) | ~(z)))
#define NQ_MD5_STEP(f, a, b, c, d, g, k, r) a = b + rol(a + f(b, c, d) + (k) + m[g], r)
        NQ_MD5_STEP(NQ_MD5_F, a, b, c, d, 0, 0xd76aa478u, 7);   NQ_MD5_STEP(NQ_MD5_F, d, a, b, c, 1, 0xe8c7b756u, 12);
        NQ_MD5_STEP(NQ_MD5_F, c, d, a, b, 2, 0x242070dbu, 17);  NQ_MD5_STEP(NQ_MD5_F, b, c, d, a, 3, 0xc1bdceeeu, 22);
        NQ_MD5_STEP(NQ_MD5_F, a, b, c, d, 4, 0xf57c0fafu, 7);   NQ_MD5_STEP(NQ_MD5_F, d, a, b, c, 5, 0x4787c62au, 12);
        NQ_MD5_STEP(NQ_MD5_F, c, d, a, b, 6, 0xa8304613u, 17);  NQ_MD5_STEP(NQ_MD5_F, b, c, d, a, 7, 0xfd469501u, 22);
        NQ_MD5_STEP(NQ_MD5_F, a, b, c, d, 8, 0x698098d8u, 7);   NQ_MD5_STEP(NQ_MD5_F, d, a, b, c, 9, 0x8b44f7afu, 12);
        NQ_MD5_STEP(NQ_MD5_F, c, d, a, b, 10, 0xffff5bb1u, 17); NQ_MD5_STEP(NQ_MD5_F, b, c, d, a, 11, 0x895cd7beu, 22);
        NQ_MD5_STEP(NQ_MD5_F, a, b, c, d, 12, 0x6b901122u, 7);  NQ_MD5_STEP(NQ_MD5_F, d, a, b, c, 13, 0xfd987193u, 12);
        NQ_MD5_STEP(NQ_MD5_F, c, d, a, b, 14, 0xa679438eu, 17); NQ_MD5_STEP(NQ_MD5_F, b, c, d, a, 15, 0x49b40821u, 22);
        NQ_MD5_STEP(NQ_MD5_G, a, b, c, d, 1, 0xf61e2562u, 5);   NQ_MD5_STEP(NQ_MD5_G, d, a, b, c, 6, 0xc040b340u, 9);
        NQ_MD5_STEP(NQ_MD5_G, c, d, a, b, 11, 0x265e5a51u, 14); NQ_MD5_STEP(NQ_MD5_G, b, c, d, a, 0, 0xe9b6c7aau, 20);
        NQ_MD5_STEP(NQ_MD5_G, a, b, c, d, 5, 0xd62f105du, 5);   NQ_MD5_STEP(NQ_MD5_G, d, a, b, c, 10, 0x02441453u, 9);
        NQ_MD5_STEP(NQ_MD5_G, c, d, a, b, 15, 0xd8a1e681u, 14); NQ_MD5_STEP(NQ_MD5_G, b, c, d, a, 4, 0xe7d3fbc8u, 20);
        NQ_MD5_STEP(NQ_MD5_G, a, b, c, d, 9, 0x21e1cde6u, 5);   NQ_MD5_STEP(NQ_MD5_G, d, a, b, c, 14, 0xc33707d6u, 9);
        NQ_MD5_STEP(NQ_MD5_G, c, d, a, b, 3, 0xf4d50d87u, 14);  NQ_MD5_STEP(NQ_MD5_G, b, c, d, a, 8, 0x455a14edu, 20);
        NQ_MD5_STEP(NQ_MD5_G, a, b, c, d, 13, 0xa9e3e905u, 5);  NQ_MD5_STEP(NQ_MD5_G, d, a, b, c, 2, 0xfcefa3f8u, 9);
        NQ_MD5_STEP(NQ_MD5_G, c, d, a, b, 7, 0x676f02d9u, 14);  NQ_MD5_STEP(NQ_MD5_G, b, c, d, a, 12, 0x8d2a4c8au, 20);
        NQ_MD5_STEP(NQ_MD5_H, a, b, c, d, 5, 0xfffa3942u, 4);   NQ_MD5_STEP(NQ_MD5_H, d, a, b, c, 8, 0x8771f681u, 11);
        NQ_MD5_STEP(NQ_MD5_H, c, d, a, b, 11, 0x6d9d6122u, 16); NQ_MD5_STEP(NQ_MD5_H, b, c, d, a, 14, 0xfde5380cu, 23);
        NQ_MD5_STEP(NQ_MD5_H, a, b, c, d, 1, 0xa4beea44u, 4);   NQ_MD5_STEP(NQ_MD5_H, d, a, b, c, 4, 0x4bdecfa9u, 11);
        NQ_MD5_STEP(NQ_MD5_H, c, d, a, b, 7, 0xf6bb4b60u, 16);  NQ_MD5_STEP(NQ_MD5_H, b, c, d, a, 10, 0xbebfbc70u, 23);
        NQ_MD5_STEP(NQ_MD5_H, a, b, c, d, 13, 0x289b7ec6u, 4);  NQ_MD5_STEP(NQ_MD5_H, d, a, b, c, 0, 0xeaa127fau, 11);
        NQ_MD5_STEP(NQ_MD5_H, c, d, a, b, 3, 0xd4ef3085u, 16);  NQ_MD5_STEP(NQ_MD5_H, b, c, d, a, 6, 0x04881d05u, 23);
        NQ_MD5_STEP(NQ_MD5_H, a, b, c, d, 9, 0xd9d4d039u, 4);   NQ_MD5_STEP(NQ_MD5_H, d, a, b, c, 12, 0xe6db99e5u, 11);
        NQ_MD5_STEP(NQ_MD5_H, c, d, a, b, 15, 0x1fa27cf8u, 16); NQ_MD5_STEP(NQ_MD5_H, b, c, d, a, 2, 0xc4ac5665u, 23);
        NQ_MD5_STEP(NQ_MD5_I, a, b, c, d, 0, 0xf4292244u, 6);   NQ_MD5_STEP(NQ_MD5_I, d, a, b, c, 7, 0x432aff97u, 10);
        NQ_MD5_STEP(NQ_MD5_I, c, d, a, b, 14, 0xab9423a7u, 15); NQ_MD5_STEP(NQ_MD5_I, b, c, d, a, 5, 0xfc93a039u, 21);
        NQ_MD5_STEP(NQ_MD5_I, a, b, c, d, 12, 0x655b59c3u, 6);  NQ_MD5_STEP(NQ_MD5_I, d, a, b, c, 3, 0x8f0ccc92u, 10);
        NQ_MD5_STEP(NQ_MD5_I, c, d, a, b, 10, 0xffeff47du, 15); NQ_MD5_STEP(NQ_MD5_I, b, c, d, a, 1, 0x85845dd1u, 21);
        NQ_MD5_STEP(NQ_MD5_I, a, b, c, d, 8, 0x6fa87e4fu, 6);   NQ_MD5_STEP(NQ_MD5_I, d, a, b, c, 15, 0xfe2ce6e0u, 10);
        NQ_MD5_STEP(NQ_MD5_I, c, d, a, b, 6, 0xa3014314u, 15);  NQ_MD5_STEP(NQ_MD5_I, b, c, d, a, 13, 0x4e0811a1u, 21);
        NQ_MD5_STEP(NQ_MD5_I, a, b, c, d, 4, 0xf7537e82u, 6);   NQ_MD5_STEP(NQ_MD5_I, d, a, b, c, 11, 0xbd3af235u, 10);
        NQ_MD5_STEP(NQ_MD5_I, c, d, a, b, 2, 0x2ad7d2bbu, 15);  NQ_MD5_STEP(NQ_MD5_I, b, c, d, a, 9, 0xeb86d391u, 21);
#undef NQ_MD5_STEP
#undef NQ_MD5_F
#undef NQ_MD5_G
#undef NQ_MD5_H
#undef NQ_MD5_I
        s_[0] += a; s_[1] += b; s_[2] += c; s_[3] += d;
    }
    uint32_t s_[4];
    uint8_t buf_[64];
    size_t fill_ = 0;
    uint64_t len_ = 0;
};

// ---- the two CRCs of a frame: CRC-8 (x^8 + x^2 + x + 1) over the header, CRC-16 (x^16 + x^15 + x^2 + 1) over the whole frame; both
//      most-significant bit first, initial value 0 ----
struct CrcTables {
    uint8_t c8[256];
    uint16_t c16[256];
    CrcTables() {
        for (int i = 0; i < 256; ++i) {
            uint8_t a = (uint8_t)i;
            uint16_t b = (uint16_t)(i << 8);
            for (int k = 0; k < 8; ++k) {
                a = (uint8_t)((a & 0x80) ? (a << 1) ^ 0x07 : a << 1);
                b = (uint16_t)((b & 0x8000) ? (b << 1) ^ 0x8005 : b << 1);
            }
            c8[i] = a;
            c16[i] = b;
        }
    }
};
inline const CrcTables& crc_tables() {
    static const CrcTables t;
    return t;
}
inline uint8_t crc8(const uint8_t* p, size_t n) {
    const CrcTables& t = crc_tables();
    uint8_t c = 0;
    for (size_t i = 0; i < n; ++i) c = t.c8[c ^ p[i]];
    return c;
}
inline uint16_t crc16(const uint8_t* p, size_t n) {
    const CrcTables& t = crc_tables();
    uint16_t c = 0;
    for (size_t i = 0; i < n; ++i) c = (uint16_t)((c << 8) ^ t.c16[(c >> 8) ^ p[i]]);
    return c;
}

// ---- bit reader, most-significant bit first; reads past the end deliver zeros and set `over` ----
struct Bits {
    const uint8_t* p;
    size_t n, byte = 0;
    uint64_t acc = 0;
    int cnt = 0;
    bool over = false;
    Bits(const uint8_t* d, size_t len) : p(d), n(len) {}
    inline void refill() {
        if (byte + 8 <= n) {                         // away from the end: four bytes at once (big-endian load) keep >= 32 bits in the window
            if (cnt <= 32) {
                uint32_t w;
                std::memcpy(&w, p + byte, 4);
                acc |= (uint64_t)__builtin_bswap32(w) << (32 - cnt);
                byte += 4;
                cnt += 32;
            }
            return;
        }
        while (cnt <= 56) {
            uint64_t v = 0;
            if (byte < n) v = p[byte];
            else if (byte >= n + 8) over = true;
            ++byte;
            acc |= v << (56 - cnt);
            cnt += 8;
        }
    }
    inline uint32_t get(int k) {                     // 0 <= k <= 32
        if (k == 0) return 0;
        refill();
        const uint32_t v = (uint32_t)(acc >> (64 - k));
        acc <<= k;
        cnt -= k;
        return v;
    }
    inline int32_t gets(int k) {                     // two's complement field of k bits, 1 <= k <= 32
        const uint32_t v = get(k);
        return k >= 32 ? (int32_t)v : (int32_t)(v << (32 - k)) >> (32 - k);
    }
    inline uint32_t unary() {                        // number of 0 bits before the next 1 bit
        uint32_t q = 0;
        for (;;) {
            refill();
            if (acc == 0) {
                q += (uint32_t)cnt;
                cnt = 0;
                if (over || q > (1u << 26)) { over = true; return 0; }
                continue;
            }
            const int z = __builtin_clzll(acc);
            q += (uint32_t)z;
            acc <<= z;                               // (two steps: z + 1 may be 64)
            acc <<= 1;
            cnt -= z + 1;
            return q;
        }
    }
    inline void align() { const int k = cnt & 7; acc <<= k; cnt -= k; }
    inline size_t byte_pos() const { return byte - (size_t)(cnt >> 3); }     // after align()
    inline bool bad() const { return over || byte_pos() > n; }
};

struct Stream {
    int32_t sample_rate = 0, channels = 0, bits = 0, min_block = 0, max_block = 0;
    int64_t total = 0;              // samples per channel; 0 = unknown (the decoder then runs to the end of the file)
    uint8_t md5[16] = {0};
    bool have_md5 = false;
    size_t marker = 0;              // offset of "fLaC" (behind an ID3v2 tag, if any)
    size_t first_frame = 0;         // offset of the first audio frame (0: metadata not walked yet)
};

// "fLaC" + STREAMINFO from the first bytes of a file (STREAMINFO is always the first metadata block).  `d` must hold at least
// 42 bytes from the marker on; an ID3v2 tag in front of the marker is skipped (its size returned in s.marker so that a caller with
// a short buffer can re-read from there).  Returns 0 = ok, 1 = need `s.marker + 42` bytes, 2 = not FLAC.
inline int stream_info(const uint8_t* d, size_t n, Stream& s) {
    size_t pos = 0;
    if (n >= 10 && !std::memcmp(d, "ID3", 3)) {
        const size_t size = ((size_t)(d[6] & 0x7f) << 21) | ((size_t)(d[7] & 0x7f) << 14) | ((size_t)(d[8] & 0x7f) << 7) | (size_t)(d[9] & 0x7f);
        pos = 10 + size + ((d[5] & 0x10) ? 10 : 0);
    }
    s.marker = pos;
    if (pos + 42 > n) return n >= 4 && pos == 0 && std::memcmp(d, "fLaC", 4) ? 2 : 1;
    if (std::memcmp(d + pos, "fLaC", 4)) return 2;
    const uint8_t* b = d + pos + 4;
    const size_t len = ((size_t)b[1] << 16) | ((size_t)b[2] << 8) | b[3];
    if ((b[0] & 0x7f) != 0 || len < 34) return 2;
    const uint8_t* q = b + 4;
    s.min_block = (q[0] << 8) | q[1];
    s.max_block = (q[2] << 8) | q[3];
    s.sample_rate = (q[10] << 12) | (q[11] << 4) | (q[12] >> 4);
    s.channels = ((q[12] >> 1) & 7) + 1;
    s.bits = (((q[12] & 1) << 4) | (q[13] >> 4)) + 1;
    s.total = ((int64_t)(q[13] & 0x0f) << 32) | ((int64_t)q[14] << 24) | ((int64_t)q[15] << 16) | ((int64_t)q[16] << 8) | (int64_t)q[17];
    std::memcpy(s.md5, q + 18, 16);
    s.have_md5 = false;
    for (int i = 0; i < 16; ++i) s.have_md5 |= s.md5[i] != 0;
    if (s.sample_rate <= 0 || s.bits < 4 || s.bits > 24 || s.max_block < 16) return 2;
    return 0;
}

// Walk the metadata blocks of a whole file in memory -> s.first_frame.
inline bool walk_metadata(const uint8_t* d, size_t n, Stream& s) {
    size_t pos = s.marker + 4;
    for (;;) {
        if (pos + 4 > n) return false;
        const bool last = d[pos] >> 7;
        const size_t len = ((size_t)d[pos + 1] << 16) | ((size_t)d[pos + 2] << 8) | d[pos + 3];
        pos += 4 + len;
        if (pos > n) return false;
        if (last) break;
    }
    s.first_frame = pos;
    return true;
}

// Sample sink: receives every decoded block as `ch` channel arrays of `count` samples (already clamped to the stream length).
struct Sink {
    virtual void block(const int32_t* const* chan, int ch, int count) = 0;
    virtual ~Sink() {}
};

namespace detail {

inline bool residual(Bits& br, int32_t* out, int bs, int order) {
    const uint32_t method = br.get(2);
    if (method > 1) return false;
    const int pbits = method ? 5 : 4;
    const uint32_t esc = (1u << pbits) - 1;
    const int porder = (int)br.get(4);
    const int parts = 1 << porder;
    if (porder > 0 && (bs & (parts - 1))) return false;
    const int per = bs >> porder;
    if (per < order && porder > 0) return false;
    if (bs < order) return false;
    int i = order;
    for (int p = 0; p < parts; ++p) {
        const int cnt = per - (p == 0 ? order : 0);
        if (cnt < 0) return false;
        const uint32_t k = br.get(pbits);
        if (k == esc) {
            const int nb = (int)br.get(5);
            for (int j = 0; j < cnt; ++j) out[i++] = nb ? br.gets(nb) : 0;
        } else {
            for (int j = 0; j < cnt; ++j) {
                br.refill();
                uint32_t u;
                const int z = br.acc ? __builtin_clzll(br.acc) : 64;
                if (z + 1 + (int)k <= br.cnt && z + 1 + (int)k < 64) {      // quotient, stop bit and remainder are all in the window
                    const uint64_t rest = br.acc << (z + 1);
                    u = ((uint32_t)z << k) | (k ? (uint32_t)(rest >> (64 - k)) : 0u);
                    br.acc = rest << k;
                    br.cnt -= z + 1 + (int)k;
                } else {
                    const uint32_t q = br.unary();
                    u = (q << k) | br.get((int)k);
                }
                out[i++] = (int32_t)(u >> 1) ^ -(int32_t)(u & 1);
            }
        }
        if (br.over) return false;
    }
    return i == bs;
}

inline bool subframe(Bits& br, int32_t* out, int bs, int bits) {
    if (br.get(1)) return false;
    const int type = (int)br.get(6);
    int wasted = 0;
    if (br.get(1)) wasted = (int)br.unary() + 1;
    const int b = bits - wasted;
    if (b < 1 || b > 32) return false;
    if (type == 0) {
        const int32_t v = br.gets(b);
        for (int i = 0; i < bs; ++i) out[i] = v;
    } else if (type == 1) {
        for (int i = 0; i < bs; ++i) out[i] = br.gets(b);
    } else if (type >= 8 && type <= 12) {
        const int order = type - 8;
        if (order > bs) return false;
        for (int i = 0; i < order; ++i) out[i] = br.gets(b);
        if (!residual(br, out, bs, order)) return false;
        switch (order) {
            case 1: for (int i = 1; i < bs; ++i) out[i] = (int32_t)(uint32_t)((int64_t)out[i] + out[i - 1]); break;
            case 2: for (int i = 2; i < bs; ++i) out[i] = (int32_t)(uint32_t)((int64_t)out[i] + 2 * (int64_t)out[i - 1] - out[i - 2]); break;
            case 3: for (int i = 3; i < bs; ++i) out[i] = (int32_t)(uint32_t)((int64_t)out[i] + 3 * (int64_t)out[i - 1] - 3 * (int64_t)out[i - 2] + out[i - 3]); break;
            case 4: for (int i = 4; i < bs; ++i) out[i] = (int32_t)(uint32_t)((int64_t)out[i] + 4 * (int64_t)out[i - 1] - 6 * (int64_t)out[i - 2] + 4 * (int64_t)out[i - 3] - out[i - 4]); break;
            default: break;
        }
    } else if (type >= 32) {
        const int order = (type & 31) + 1;
        if (order > bs) return false;
        for (int i = 0; i < order; ++i) out[i] = br.gets(b);
        const int prec = (int)br.get(4) + 1;
        if (prec == 16) return false;
        const int shift = br.gets(5);
        if (shift < 0) return false;
        int32_t coef[32];
        for (int j = 0; j < order; ++j) coef[j] = br.gets(prec);
        if (!residual(br, out, bs, order)) return false;
        int lg = 0;
        while ((1 << lg) < order) ++lg;
        if (b + prec + lg <= 32) {                   // every partial sum fits 32 bits (what libFLAC's 32-bit restore assumes too)
            for (int i = order; i < bs; ++i) {
                uint32_t acc = 0;                    // (unsigned: a damaged stream may wrap, a valid one cannot)
                for (int j = 0; j < order; ++j) acc += (uint32_t)coef[j] * (uint32_t)out[i - 1 - j];
                out[i] = (int32_t)((uint32_t)out[i] + (uint32_t)((int32_t)acc >> shift));
            }
        } else {
            for (int i = order; i < bs; ++i) {
                int64_t acc = 0;
                for (int j = 0; j < order; ++j) acc += (int64_t)coef[j] * out[i - 1 - j];
                out[i] = (int32_t)(uint32_t)((int64_t)out[i] + (acc >> shift));
            }
        }
    } else {
        return false;                                // reserved subframe types
    }
    if (wasted)
        for (int i = 0; i < bs; ++i) out[i] = (int32_t)((uint32_t)out[i] << wasted);
    return !br.over;
}

}  // namespace detail

// Decode a whole file image.  Returns 0 = ok, 2 = malformed / unsupported / a checksum does not hold, 3 = the stream ends before
// (or runs beyond) STREAMINFO's length.  *decoded = samples per channel delivered to the sink.
inline int decode(const uint8_t* d, size_t n, Stream& s, Sink* sink, int64_t* decoded) {
    if (!s.first_frame && !walk_metadata(d, n, s)) return 2;
    const int ch = s.channels, bytes = (s.bits + 7) / 8;
    std::vector<int32_t> store((size_t)ch * 65536);
    int32_t* chan[8];
    for (int c = 0; c < ch; ++c) chan[c] = store.data() + (size_t)c * 65536;
    std::vector<uint8_t> raw;
    Md5 md5;
    size_t pos = s.first_frame;
    int64_t done = 0;
    while (s.total ? done < s.total : pos < n) {
        if (pos + 6 > n) return 3;
        const uint8_t* h = d + pos;
        if (h[0] != 0xFF || (h[1] & 0xFE) != 0xF8 || (h[3] & 1)) return 2;
        const int bs_code = h[2] >> 4, sr_code = h[2] & 15, ca = h[3] >> 4, ss = (h[3] >> 1) & 7;
        size_t q = 4;
        {                                             // the frame / sample number, coded like UTF-8 (1-7 bytes)
            const uint8_t f = h[q];
            int extra = 0;
            if (f & 0x80) {
                int ones = 0;
                while (ones < 8 && (f & (0x80 >> ones))) ++ones;
                if (ones < 2 || ones > 7) return 2;
                extra = ones - 1;
            }
            if (pos + q + 1 + extra + 5 > n) return 3;
            for (int i = 1; i <= extra; ++i)
                if ((h[q + i] & 0xC0) != 0x80) return 2;
            q += 1 + extra;
        }
        int bs;
        if (bs_code == 0) return 2;
        else if (bs_code == 1) bs = 192;
        else if (bs_code <= 5) bs = 576 << (bs_code - 2);
        else if (bs_code == 6) { bs = h[q] + 1; q += 1; }
        else if (bs_code == 7) { bs = ((h[q] << 8) | h[q + 1]) + 1; q += 2; }
        else bs = 256 << (bs_code - 8);
        if (sr_code == 12) q += 1;
        else if (sr_code == 13 || sr_code == 14) q += 2;
        else if (sr_code == 15) return 2;
        if (pos + q + 1 > n) return 3;
        if (crc8(h, q) != h[q]) return 2;
        q += 1;
        static const int ss_bits[8] = {0, 8, 12, -1, 16, 20, 24, -1};
        const int fbits = ss == 0 ? s.bits : ss_bits[ss];
        const int fch = ca < 8 ? ca + 1 : (ca <= 10 ? 2 : -1);
        if (fbits != s.bits || fch != ch) return 2;  // one sample format per stream
        Bits br(h + q, n - pos - q);
        for (int c = 0; c < ch; ++c) {
            const bool side = (ca == 8 && c == 1) || (ca == 9 && c == 0) || (ca == 10 && c == 1);
            if (!detail::subframe(br, chan[c], bs, fbits + (side ? 1 : 0))) return br.over ? 3 : 2;
        }
        br.align();
        if (br.bad()) return 3;
        const size_t body = q + br.byte_pos();
        if (pos + body + 2 > n) return 3;
        if (crc16(h, body) != (uint16_t)((h[body] << 8) | h[body + 1])) return 2;
        pos += body + 2;
        if (ca == 8) for (int i = 0; i < bs; ++i) chan[1][i] = (int32_t)((uint32_t)chan[0][i] - (uint32_t)chan[1][i]);
        else if (ca == 9) for (int i = 0; i < bs; ++i) chan[0][i] = (int32_t)((uint32_t)chan[0][i] + (uint32_t)chan[1][i]);
        else if (ca == 10)
            for (int i = 0; i < bs; ++i) {
                const int32_t sd = chan[1][i];
                const int32_t m = (int32_t)(((uint32_t)chan[0][i] << 1) | (uint32_t)(sd & 1));
                chan[0][i] = (int32_t)((uint32_t)m + (uint32_t)sd) >> 1;
                chan[1][i] = (int32_t)((uint32_t)m - (uint32_t)sd) >> 1;
            }
        int count = bs;
        if (s.total && done + count > s.total) return 3;      // an encoder never pads the last block: a longer stream is a broken one
        if (s.have_md5) {
            raw.resize((size_t)count * ch * bytes);
            uint8_t* w = raw.data();
            if (bytes == 2 && ch == 1) {
                int16_t* w16 = (int16_t*)w;
                for (int i = 0; i < count; ++i) w16[i] = (int16_t)chan[0][i];
            } else
            for (int i = 0; i < count; ++i)
                for (int c = 0; c < ch; ++c) {
                    const uint32_t v = (uint32_t)chan[c][i];
                    for (int k = 0; k < bytes; ++k) *w++ = (uint8_t)(v >> (8 * k));
                }
            md5.add(raw.data(), raw.size());
        }
        if (sink) sink->block(chan, ch, count);
        done += count;
    }
    if (decoded) *decoded = done;
    if (s.total && done != s.total) return 3;
    if (s.have_md5) {
        uint8_t got[16];
        md5.finish(got);
        if (std::memcmp(got, s.md5, 16)) return 2;
    }
    return 0;
}

}  // namespace nqflac
