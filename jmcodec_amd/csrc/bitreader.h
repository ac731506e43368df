// jmcodec_amd/csrc/bitreader.h -- MSB-first bit reader over an RBSP with a 64-bit cache.
// Host half of what the closed CUVID parser does inside cuvidParseVideoData
// (/root/reference/nv_dec/nv_dec.cpp:394).
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <string.h>

namespace jmamd {

class BitReader {
public:
    BitReader() {}
    BitReader(const uint8_t *p, size_t n) { reset(p, n); }
    // The buffer must have >= 8 readable bytes of slack after p+n (see Rbsp::kSlack).
    void reset(const uint8_t *p, size_t n) {
        base_ = p; size_ = n; pos_ = 0; refill();
    }
    inline void refill() {
        if ((pos_ >> 3) > size_) { cache_ = 0; avail_ = 64 - (int)(pos_ & 7); error_ = true; return; }
        uint64_t v;
        memcpy(&v, base_ + (pos_ >> 3), 8);
        cache_ = __builtin_bswap64(v) << (pos_ & 7);
        avail_ = 64 - (int)(pos_ & 7);
    }
    // peek up to 32 bits (n in 1..32)
    inline uint32_t peek(int n) {
        if (avail_ < n) refill();
        return (uint32_t)(cache_ >> (64 - n));
    }
    inline void skip(int n) { pos_ += n; cache_ <<= n; avail_ -= n; }
    inline uint32_t u(int n) {
        if (n == 0) return 0;
        uint32_t v = peek(n);
        skip(n);
        return v;
    }
    inline uint32_t u1() {
        if (avail_ < 1) refill();
        uint32_t v = (uint32_t)(cache_ >> 63);
        skip(1);
        return v;
    }
    inline uint32_t ue() {
        if (avail_ < 32) refill();
        uint32_t top = (uint32_t)(cache_ >> 32);
        if (top == 0) { skip(32); error_ = true; return 0; }
        int lz = __builtin_clz(top);
        if (lz <= 15) {                 // the whole codeword (2 * lz + 1 <= 31 bits) is inside the 32 bits just checked
            const int len = 2 * lz + 1;
            uint32_t v = top >> (32 - len);
            skip(len);
            return v - 1;
        }
        skip(lz);                       // now positioned at the leading 1
        if (avail_ < lz + 1) refill();
        uint32_t v = (uint32_t)(cache_ >> (63 - lz));   // lz+1 bits including the leading 1
        skip(lz + 1);
        return v - 1;
    }
    inline int32_t se() {
        uint32_t k = ue();
        return (k & 1) ? (int32_t)((k + 1) >> 1) : -(int32_t)(k >> 1);
    }
    inline int te(int range_max) { return range_max > 1 ? (int)ue() : (int)(u1() ^ 1); }
    inline size_t bitpos() const { return pos_; }
    inline bool overrun() const { return error_ || pos_ > size_ * 8; }
    inline bool aligned() const { return (pos_ & 7) == 0; }
    inline void align_zero() { int r = (int)(pos_ & 7); if (r) skip(8 - r); }
    // raw byte access for I_PCM (reader must be byte aligned)
    inline const uint8_t *byte_ptr() const { return base_ + (pos_ >> 3); }
    inline void skip_bytes(size_t n) { pos_ += n * 8; refill(); }
    // 7.2 more_rbsp_data(): position of the stop bit is precomputed by set_end()
    void set_end_from_trailing() {
        size_t n = size_;
        while (n > 0 && base_[n - 1] == 0) n--;
        if (n == 0) { last_bit_ = 0; return; }
        uint8_t b = base_[n - 1];
        int tz = __builtin_ctz((unsigned)b);
        last_bit_ = (n - 1) * 8 + (7 - tz);       // bit index of the rbsp_stop_one_bit
    }
    inline bool more_rbsp_data() const { return pos_ < last_bit_; }
    const uint8_t *base() const { return base_; }
    size_t size() const { return size_; }

private:
    const uint8_t *base_ = nullptr;
    size_t size_ = 0, pos_ = 0, last_bit_ = 0;
    uint64_t cache_ = 0;
    int avail_ = 0;
    bool error_ = false;
};

// Emulation-prevention removal (7.4.1).  out must hold len + kSlack bytes.
struct Rbsp {
    static constexpr size_t kSlack = 16;
    static size_t unescape(const uint8_t *in, size_t len, uint8_t *out) {
        size_t n = 0, i = 0;
        while (i < len) {
            // fast path: copy until a 0x00 0x00 0x03 pattern can start
            const uint8_t *z = (const uint8_t *)memchr(in + i, 0, len - i);
            if (!z) { memcpy(out + n, in + i, len - i); n += len - i; break; }
            size_t run = (size_t)(z - (in + i));
            memcpy(out + n, in + i, run); n += run; i += run;
            if (i + 2 < len && in[i + 1] == 0 && in[i + 2] == 3) { out[n++] = 0; out[n++] = 0; i += 3; }
            else { out[n++] = in[i++]; }
        }
        memset(out + n, 0, kSlack);
        return n;
    }
};

}  // namespace jmamd
