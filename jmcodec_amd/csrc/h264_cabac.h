// jmcodec_amd/csrc/h264_cabac.h -- CABAC arithmetic decoding engine and context variables (H.264 9.3.1, 9.3.3.2).
//
// Host half of the entropy stage the reference leaves to the NVDEC ASIC behind cuvidDecodePicture
// (/root/reference/nv_dec/nv_dec.cpp:33-41; entropy_coding_mode_flag travels in CUVIDH264PICPARAMS,
// nv_sdk/inc/dynlink_cuviddec.h:243-298).  Syntax-element binarisations and ctxIdxInc derivations live with the
// macroblock layer in h264_cavlc.cpp; this file is only the engine.
//
// Representation: `val` holds codIOffset followed by look-ahead bits of the stream, `pos` is the bit index of the offset's least
// significant bit inside `val` (codIOffset == val >> pos).  Renormalising by n bits is `pos -= n` (no shift of val); 32 more bits
// are appended when pos drops below 16.  Comparisons use codIRange << pos.  Exactly the decisions of 9.3.3.2 (the oracle's
// bit-serial restatement decodes the same bins), just without touching the bitstream per bin.
#pragma once
#include "bitreader.h"
#include "cabac_tables.h"

// developer diagnostic (make -C tools host_bench HB_FLAGS=-DJM_COUNT_BINS, single-threaded runs): number of bins decoded
#ifdef JM_COUNT_BINS
namespace jmamd { extern long g_cabac_bins; }
#define JM_BIN(n) (jmamd::g_cabac_bins += (n))
#else
#define JM_BIN(n) ((void)0)
#endif

// (a CabacRegs member that is called out of line takes the object's address, and with it the registers the object exists for)
#define JM_ALWAYS_INLINE __attribute__((always_inline)) inline

namespace jmamd {

// next state by ((pStateIdx << 1 | valMPS) << 1 | "the bin was the least probable symbol") (Table 9-45 transIdxLps / transIdxMps)
struct CabacTrans {
    uint8_t t[256];
    CabacTrans() : t() {
        for (int s = 0; s < 128; s++) {
            const int st = s >> 1, mps = s & 1;
            t[2 * s] = (uint8_t)(((st < 62 ? st + 1 : st) << 1) | mps);
            t[2 * s + 1] = (uint8_t)((cabac_trans_lps[st] << 1) | (st == 0 ? mps ^ 1 : mps));
        }
    }
};
static const CabacTrans kCabacTrans;       // 256 bytes per translation unit, filled before main

// Everything one decision needs from the tables, per context state (pStateIdx << 1 | valMPS): a 64-bit word with rangeTabLps for the four quantised
// ranges (Table 9-44) in the low byte of its four 16-bit fields, and a 16-bit word with the next state after the most probable symbol (low byte) and
// after the least probable one (high byte).  Their addresses depend on the state only, so the loads are off the dependency chain that links one bin to
// the next (range -> LPS range -> new range): on that chain the range picks its field with a shift instead of a second, range-addressed table load.
// (Tried and measured slower on Xeon and EPYC alike: renormalising BOTH candidate ranges before the comparison -- the LPS shift from bits 8-10 of the
// field -- so that one conditional move follows it; the shorter chain costs ten instructions more per bin.)
struct CabacRows {
    uint64_t r[128]; uint16_t next[128];
    CabacRows() : r(), next() {
        for (int s = 0; s < 128; s++) {
            uint64_t w = 0;
            for (int q = 0; q < 4; q++) w |= (uint64_t)cabac_range_lps[s >> 1][q] << (16 * q);
            r[s] = w; next[s] = (uint16_t)(kCabacTrans.t[2 * s] | kCabacTrans.t[2 * s + 1] << 8);
        }
    }
};
static const CabacRows kCabacRows;         // (defined after kCabacTrans in every translation unit: initialised after it)

// n bypass bins at once are the quotient of the offset by the range (9.3.3.2.3 is a long division, one bit per bin): x / range for x < range << 16, by
// multiplication with ceil(2^34 / range).  Exact for 256 <= range <= 510 and x < 2^25: range * M = 2^34 + e with 0 <= e < 2^9, so x * M / 2^34 =
// x / range + x * e / (range * 2^34), and the second term is below 2^25 * 2^9 / (range * 2^34) = 1 / range -- too small to carry the sum past the next
// integer, which x / range misses by at least 1 / range (tests/test_abi.py runs every range against the division).
struct CabacRecip {
    uint32_t m[256];
    CabacRecip() : m() { for (int r = 256; r < 512; r++) m[r - 256] = (uint32_t)((((uint64_t)1 << 34) + r - 1) / r); }
};
static const CabacRecip kCabacRecip;
static inline uint32_t cabac_quotient(uint64_t x, uint32_t range) {
    return range >= 256 ? (uint32_t)((x * kCabacRecip.m[range - 256]) >> 34) : (uint32_t)(x / range);       // (the range is normalised: always >= 256 here)
}
// Number of 1 bins at the head of a run of bypass bins, from the engine's variables alone: with S = range << pos the first bin is 1 when val >= S / 2,
// the second when the remainder reaches S / 4, ...: k ones followed by a zero <=> S / 2^k >= S - val > S / 2^(k + 1), so k is found by counting
// leading zeros instead of k + 1 compare-and-subtract steps that end in a mispredicted branch.  Returns -1 when the caller has to take the bins one by
// one (more than 15 ones, or a state no well-formed stream reaches).
static inline int cabac_leading_ones(uint64_t val, uint64_t s) {
    if (val >= s) return -1;
    const uint64_t t = s - val;
    int a = __builtin_clzll(t) - __builtin_clzll(s);
    a -= (t << a) > s;
    return a <= 15 ? a : -1;
}

// the next 32 bits of slice data when fewer than four bytes are left (a few bytes of look-ahead past the end are normal; bit 32 of the result: the
// decoder has run past the end of the data).  Out of line: it runs once per slice, and inlined its loop takes registers from every decoding loop.
__attribute__((noinline, cold)) static uint64_t cabac_tail_word(const uint8_t *ptr, const uint8_t *end) {
    uint64_t w = 0;
    for (int i = 0; i < 4; i++) w = (w << 8) | (ptr + i < end ? ptr[i] : 0);
    return w | (ptr >= end + 8 ? (uint64_t)1 << 32 : 0);
}

struct Cabac {
    uint64_t val = 0; int pos = 0;
    uint32_t range = 510;
    const uint8_t *ptr = nullptr, *start = nullptr, *end = nullptr;
    bool overrun = false;                    // the arithmetic decoder ran past the end of the slice data
    // pStateIdx << 1 | valMPS.  16-bit on purpose: a store through a character type may alias anything, so with uint8_t every context update made the
    // compiler reload val / pos / range from memory before the next bin (and keep them there); int16 stores cannot alias them.
    typedef uint16_t State;
    State state[CABAC_N_CTX];

    // 9.3.1.1: context variables from (m, n) and SliceQPY; table 0 = I slices, 1 + cabac_init_idc otherwise
    void init_contexts(int table, int slice_qp) {
        int qp = slice_qp < 0 ? 0 : (slice_qp > 51 ? 51 : slice_qp);
        for (int i = 0; i < CABAC_N_CTX; i++) {
            int pre = ((cabac_init_mn[table][i][0] * qp) >> 4) + cabac_init_mn[table][i][1];
            pre = pre < 1 ? 1 : (pre > 126 ? 126 : pre);
            state[i] = pre <= 63 ? (State)((63 - pre) << 1) : (State)(((pre - 64) << 1) | 1);
        }
    }
    inline void refill() {
        uint32_t w;
        if (__builtin_expect(ptr + 4 <= end, 1)) { w = ((uint32_t)ptr[0] << 24) | ((uint32_t)ptr[1] << 16) | ((uint32_t)ptr[2] << 8) | ptr[3]; }
        else { const uint64_t t = cabac_tail_word(ptr, end); w = (uint32_t)t; overrun |= (t >> 32) != 0; }
        ptr += 4;
        val = (val << 32) | w; pos += 32;
    }
    // 9.3.1.2: `p` is the first byte of the arithmetic code (byte aligned), `e` the end of the slice data
    void init_engine(const uint8_t *p, const uint8_t *e) {
        start = ptr = p; end = e; val = 0; pos = -9; range = 510; overrun = false;
        refill();                                              // val = first 32 bits, pos = 23: codIOffset = first 9 bits
    }
    // bits the arithmetic decoder of 9.3.1.2 / 9.3.3.2 has read so far (9 at initialisation + 1 per renormalisation shift)
    inline size_t bits_consumed() const { return (size_t)(ptr - start) * 8 - (size_t)pos; }

    // 9.3.3.2.1, without a branch on the decoded symbol (it is the least predictable branch of the whole parser): `m` is all ones when the
    // offset lies in the LPS sub-interval and selects offset, range, next state and bin value arithmetically
    __attribute__((always_inline)) inline int decision(int ctx) {
        JM_BIN(1);
        const uint32_t s = state[ctx];
        const uint64_t row = kCabacRows.r[s];
        const uint32_t lps = (uint32_t)(row >> ((range >> 2) & 48)) & 0xff;
        const uint32_t rm = range - lps;
        const uint64_t scaled = (uint64_t)rm << pos;
        const uint64_t m = (uint64_t)((int64_t)(scaled - val - 1) >> 63);
        const uint32_t m32 = (uint32_t)m;
        val -= scaled & m;
        const uint32_t r = rm ^ ((rm ^ lps) & m32);
        state[ctx] = (Cabac::State)((kCabacRows.next[s] >> (m32 & 8)) & 0xff);
        const int sh = __builtin_clz(r) - 23;                   // MPS: 0 or 1 (range stays >= 128); LPS: range in [6, 240] -> back into [256, 511]
        range = r << sh; pos -= sh;
        if (pos < 16) refill();
        return (int)((s ^ m32) & 1);
    }
    inline int bypass() {
        JM_BIN(1);
        pos--;
        uint64_t scaled = (uint64_t)range << pos;
        int bin = 0;
        if (val >= scaled) { val -= scaled; bin = 1; }
        if (pos < 16) refill();
        return bin;
    }
    // n bypass bins at once (1 <= n <= 16), first bin in the most significant bit: the n compare-and-subtract steps of 9.3.3.2.3 are one
    // division of the offset by the (scaled) range
    inline uint32_t bypass_bits(int n) {
        JM_BIN(n);
        if (pos < 16) refill();
        pos -= n;
        const uint32_t q = cabac_quotient(val >> pos, range);
        val -= ((uint64_t)q * range) << pos;
        if (pos < 16) refill();
        return q;
    }
    // bypass bins up to and including the first 0, at most `limit` of them when they are all 1: the number of 1 bins (a truncated unary prefix)
    inline int unary(int limit) {
        const uint64_t s = (uint64_t)range << pos;                  // pos >= 16: every operation leaves the engine refilled
        const int a = cabac_leading_ones(val, s);
        if (a < 0) { int q = 0; while (q < limit && bypass()) q++; return q; }
        JM_BIN(a < limit ? a + 1 : limit);
        if (a >= limit) { val -= s - (s >> limit); pos -= limit; if (pos < 16) refill(); return limit; }
        val -= s - (s >> a); pos -= a + 1;
        if (pos < 16) refill();
        return a;
    }
    inline int terminate() {
        JM_BIN(1);
        range -= 2;
        if (val >= ((uint64_t)range << pos)) return 1;          // no renormalisation: parsing of the slice / before I_PCM ends
        if (range < 256) { range <<= 1; pos--; if (pos < 16) refill(); }
        return 0;
    }
};

// The engine's hot variables as LOCALS of the calling function.  Inside `Cabac` they are members next to `state[]`, and every context update
// is a uint8_t store -- which may alias anything, so the compiler reloads val / pos / range from memory after each bin and the store-to-load
// forwarding sits on the dependency chain of the arithmetic decoder.  A residual block decodes dozens of bins in a row: it takes a CabacRegs
// (`CabacRegs r(cb);` ... `r.commit();`), whose scalars never have their address taken and therefore live in registers.  Same arithmetic as above.
struct CabacRegs {
    uint64_t val; int pos; uint32_t range; const uint8_t *ptr; const uint8_t *const end; Cabac::State *const state; bool overrun; Cabac &home;
    explicit CabacRegs(Cabac &c) : val(c.val), pos(c.pos), range(c.range), ptr(c.ptr), end(c.end), state(c.state), overrun(c.overrun), home(c) {}
    JM_ALWAYS_INLINE void commit() { home.val = val; home.pos = pos; home.range = range; home.ptr = ptr; home.overrun = overrun; }
    JM_ALWAYS_INLINE void refill() {
        uint32_t w;
        if (__builtin_expect(ptr + 4 <= end, 1)) { w = ((uint32_t)ptr[0] << 24) | ((uint32_t)ptr[1] << 16) | ((uint32_t)ptr[2] << 8) | ptr[3]; }
        else { const uint64_t t = cabac_tail_word(ptr, end); w = (uint32_t)t; overrun |= (t >> 32) != 0; }
        ptr += 4;
        val = (val << 32) | w; pos += 32;
    }
    JM_ALWAYS_INLINE int decision(int ctx) {
        JM_BIN(1);
        const uint32_t s = state[ctx];
        const uint64_t row = kCabacRows.r[s];
        const uint32_t lps = (uint32_t)(row >> ((range >> 2) & 48)) & 0xff;
        const uint32_t rm = range - lps;
        const uint64_t scaled = (uint64_t)rm << pos;
        const uint64_t m = (uint64_t)((int64_t)(scaled - val - 1) >> 63);
        const uint32_t m32 = (uint32_t)m;
        val -= scaled & m;
        const uint32_t r = rm ^ ((rm ^ lps) & m32);
        state[ctx] = (Cabac::State)((kCabacRows.next[s] >> (m32 & 8)) & 0xff);
        const int sh = __builtin_clz(r) - 23;                   // MPS: 0 or 1 (range stays >= 128); LPS: range in [6, 240] -> back into [256, 511]
        range = r << sh; pos -= sh;
        if (pos < 16) refill();
        return (int)((s ^ m32) & 1);
    }
    JM_ALWAYS_INLINE int bypass() {
        JM_BIN(1);
        pos--;
        const uint64_t scaled = (uint64_t)range << pos;
        int bin = 0;
        if (val >= scaled) { val -= scaled; bin = 1; }
        if (pos < 16) refill();
        return bin;
    }
    JM_ALWAYS_INLINE uint32_t bypass_bits(int n) {
        JM_BIN(n);
        if (pos < 16) refill();
        pos -= n;
        const uint32_t q = cabac_quotient(val >> pos, range);
        val -= ((uint64_t)q * range) << pos;
        if (pos < 16) refill();
        return q;
    }
    // bypass bins up to and including the first 0, at most `limit` of them when they are all 1: the number of 1 bins (a truncated unary prefix)
    JM_ALWAYS_INLINE int unary(int limit) {
        const uint64_t s = (uint64_t)range << pos;                  // pos >= 16: every operation leaves the engine refilled
        const int a = cabac_leading_ones(val, s);
        if (a < 0) { int q = 0; while (q < limit && bypass()) q++; return q; }
        JM_BIN(a < limit ? a + 1 : limit);
        if (a >= limit) { val -= s - (s >> limit); pos -= limit; if (pos < 16) refill(); return limit; }
        val -= s - (s >> a); pos -= a + 1;
        if (pos < 16) refill();
        return a;
    }
};

}  // namespace jmamd
