// jmcodec_amd/csrc/h264_cavlc.cpp -- CAVLC macroblock layer parser producing the device job list.
// H.264 7.3.4 (slice_data), 7.3.5 (macroblock_layer), 9.2 (CAVLC), 8.3.1.1 (intra mode
// prediction), 8.4.1 (motion vector prediction).  See h264_cavlc.h for what it replaces.
#include "h264_cavlc.h"
#include "h264_cabac.h"
#include <mutex>
#include <stddef.h>
#include <emmintrin.h>

namespace jmamd {

// ----------------------------------------------------------------------------------------
// VLC tables: {length, codeword} source arrays (Table 9-5, 9-7..9-10) -> lookup tables
// ----------------------------------------------------------------------------------------
static const uint8_t kTokLen[3][68] = {
{ 1,0,0,0, 6,2,0,0, 8,6,3,0, 9,8,7,5, 10,9,8,6, 11,10,9,7, 13,11,10,8, 13,13,11,9, 13,13,13,10,
 14,14,13,11, 14,14,14,13, 15,15,14,14, 15,15,15,14, 16,15,15,15, 16,16,16,15, 16,16,16,16, 16,16,16,16 },
{ 2,0,0,0, 6,2,0,0, 6,5,3,0, 7,6,6,4, 8,6,6,4, 8,7,7,5, 9,8,8,6, 11,9,9,6, 11,11,11,7,
 12,11,11,9, 12,12,12,11, 12,12,12,11, 13,13,13,12, 13,13,13,13, 13,14,13,13, 14,14,14,13, 14,14,14,14 },
{ 4,0,0,0, 6,4,0,0, 6,5,4,0, 6,5,5,4, 7,5,5,4, 7,5,5,4, 7,6,6,4, 7,6,6,4, 8,7,7,5,
 8,8,7,6, 9,8,8,7, 9,9,8,8, 9,9,9,8, 10,9,9,9, 10,10,10,10, 10,10,10,10, 10,10,10,10 } };
static const uint8_t kTokBits[3][68] = {
{ 1,0,0,0, 5,1,0,0, 7,4,1,0, 7,6,5,3, 7,6,5,3, 7,6,5,4, 15,6,5,4, 11,14,5,4, 8,10,13,4,
 15,14,9,4, 11,10,13,12, 15,14,9,12, 11,10,13,8, 15,1,9,12, 11,14,13,8, 7,10,9,12, 4,6,5,8 },
{ 3,0,0,0, 11,2,0,0, 7,7,3,0, 7,10,9,5, 7,6,5,4, 4,6,5,6, 7,6,5,8, 15,6,5,4, 11,14,13,4,
 15,10,9,4, 11,14,13,12, 8,10,9,8, 15,14,13,12, 11,10,9,12, 7,11,6,8, 9,8,10,1, 7,6,5,4 },
{ 15,0,0,0, 15,14,0,0, 11,15,13,0, 8,12,14,12, 15,10,11,11, 11,8,9,10, 9,14,13,9, 8,10,9,8, 15,14,13,13,
 11,14,10,12, 15,10,13,12, 11,14,9,12, 8,10,13,8, 13,7,9,12, 9,12,11,10, 5,8,7,6, 1,4,3,2 } };
static const uint8_t kCdcLen[20] = { 2,0,0,0, 6,1,0,0, 6,6,3,0, 6,7,7,6, 6,8,8,7 };
static const uint8_t kCdcBits[20] = { 1,0,0,0, 7,1,0,0, 4,6,1,0, 3,3,2,5, 2,3,2,0 };
static const uint8_t kTzLen[15][16] = {
 {1,3,3,4,4,5,5,6,6,7,7,8,8,9,9,9},{3,3,3,3,3,4,4,4,4,5,5,6,6,6,6,0},{4,3,3,3,4,4,3,3,4,5,5,6,5,6,0,0},
 {5,3,4,4,3,3,3,4,3,4,5,5,5,0,0,0},{4,4,4,3,3,3,3,3,4,5,4,5,0,0,0,0},{6,5,3,3,3,3,3,3,4,3,6,0,0,0,0,0},
 {6,5,3,3,3,2,3,4,3,6,0,0,0,0,0,0},{6,4,5,3,2,2,3,3,6,0,0,0,0,0,0,0},{6,6,4,2,2,3,2,5,0,0,0,0,0,0,0,0},
 {5,5,3,2,2,2,4,0,0,0,0,0,0,0,0,0},{4,4,3,3,1,3,0,0,0,0,0,0,0,0,0,0},{4,4,2,1,3,0,0,0,0,0,0,0,0,0,0,0},
 {3,3,1,2,0,0,0,0,0,0,0,0,0,0,0,0},{2,2,1,0,0,0,0,0,0,0,0,0,0,0,0,0},{1,1,0,0,0,0,0,0,0,0,0,0,0,0,0,0} };
static const uint8_t kTzBits[15][16] = {
 {1,3,2,3,2,3,2,3,2,3,2,3,2,3,2,1},{7,6,5,4,3,5,4,3,2,3,2,3,2,1,0,0},{5,7,6,5,4,3,4,3,2,3,2,1,1,0,0,0},
 {3,7,5,4,6,5,4,3,3,2,2,1,0,0,0,0},{5,4,3,7,6,5,4,3,2,1,1,0,0,0,0,0},{1,1,7,6,5,4,3,2,1,1,0,0,0,0,0,0},
 {1,1,5,4,3,3,2,1,1,0,0,0,0,0,0,0},{1,1,1,3,3,2,2,1,0,0,0,0,0,0,0,0},{1,0,1,3,2,1,1,1,0,0,0,0,0,0,0,0},
 {1,0,1,3,2,1,1,0,0,0,0,0,0,0,0,0},{0,1,1,2,1,3,0,0,0,0,0,0,0,0,0,0},{0,1,1,1,1,0,0,0,0,0,0,0,0,0,0,0},
 {0,1,1,1,0,0,0,0,0,0,0,0,0,0,0,0},{0,1,1,0,0,0,0,0,0,0,0,0,0,0,0,0},{0,1,0,0,0,0,0,0,0,0,0,0,0,0,0,0} };
static const uint8_t kCtzLen[3][4] = { {1,2,3,3},{1,2,2,0},{1,1,0,0} };
static const uint8_t kCtzBits[3][4] = { {1,1,1,0},{1,1,0,0},{1,0,0,0} };
static const uint8_t kRunLen[6][7] = { {1,1,0,0,0,0,0},{1,2,2,0,0,0,0},{2,2,2,2,0,0,0},{2,2,2,3,3,0,0},{2,2,3,3,3,3,0},{2,3,3,3,3,3,3} };
static const uint8_t kRunBits[6][7] = { {1,0,0,0,0,0,0},{1,1,0,0,0,0,0},{3,2,1,0,0,0,0},{3,2,1,1,0,0,0},{3,2,3,2,1,0,0},{3,0,1,3,2,5,4} };
static const uint8_t kCbpIntra[48] = {
    47,31,15,0,23,27,29,30,7,11,13,14,39,43,45,46,16,3,5,10,12,19,21,26,28,35,37,42,44,1,2,4,8,17,18,20,24,6,9,22,25,32,33,34,36,40,38,41 };
static const uint8_t kCbpInter[48] = {
    0,16,1,2,4,8,32,3,5,10,12,15,47,7,11,13,14,6,9,31,35,37,42,44,33,34,36,40,39,43,45,46,17,18,20,24,19,21,26,28,23,27,29,30,22,25,38,41 };
static const uint8_t kZigzag4[16] = {0,1,4,8,5,2,3,6,9,12,13,10,7,11,14,15};
// zig-zag scan of a luma block coded with the large transform, frame macroblocks (8.5.7): scan position -> raster index
static const uint8_t kZigzag8[64] = {
    0, 1, 8,16, 9, 2, 3,10, 17,24,32,25,18,11, 4, 5, 12,19,26,33,40,48,41,34, 27,20,13, 6, 7,14,21,28,
   35,42,49,56,57,50,43,36, 29,22,15,23,30,37,44,51, 58,59,52,45,38,31,39,46, 53,60,61,54,47,55,62,63 };

// Field scans (Table 8-2 / Table 8-3): the coefficients of a field macroblock run down the columns first.  Built from the (column, row) pairs of the
// tables -- two digits per scan position -- into the same "scan position -> raster index" form as the zig-zag tables above.
struct FieldScans {
    uint8_t s4[16], s8[64];
    FieldScans() {
        static const char *xy4 = "00 01 10 02 03 11 12 13 20 21 22 23 30 31 32 33";
        static const char *xy8 = "00 01 02 10 11 03 04 12 20 13 05 06 07 14 21 30 22 15 16 17 23 31 40 32 24 25 26 27 33 41 50 42 "
                                 "34 35 36 37 43 51 60 52 44 45 46 47 53 61 62 54 55 56 57 63 70 71 64 65 66 67 72 73 74 75 76 77";
        for (int i = 0; i < 16; i++) s4[i] = (uint8_t)((xy4[3 * i + 1] - '0') * 4 + (xy4[3 * i] - '0'));
        for (int i = 0; i < 64; i++) s8[i] = (uint8_t)((xy8[3 * i + 1] - '0') * 8 + (xy8[3 * i] - '0'));
    }
};
static const FieldScans kFieldScan;

// two-level table for coeff_token: primary on the top 8 bits, secondary on the next 8
struct TokTable { uint16_t t[256 * 24]; };           // entry = sym << 8 | len ; len 0xFF => sym = subtable number
static TokTable g_tok[4];                           // [3]: nC >= 8, the 6-bit fixed-length code in the same format (fast path)
static const uint8_t kTokClass[17] = {0, 0, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 3, 3, 3, 3, 3};
// ... straight from the sum s of the two neighbouring counts in the fast path's window (64 = not available): s < 64 -> nC = (s + 1) >> 1, else nC = s & 31
struct TokClassOfSum { uint8_t t[129]; TokClassOfSum() { for (int s = 0; s <= 128; s++) { const int nc = s < 64 ? (s + 1) >> 1 : s & 31;
    t[s] = kTokClass[nc > 16 ? 16 : nc]; } } };
static const TokClassOfSum kTokClassOfSum;
static uint16_t g_tok_flc[64];                        // nC >= 8: 6-bit FLC
static uint16_t g_cdc[256];                           // chroma DC token, max 8 bits
static uint16_t g_tz[15][512];                        // total_zeros, max 9 bits
static uint8_t  g_ctz[3][8];                          // chroma DC total_zeros, max 3 bits: val << 4 | len
static uint8_t  g_run[6][8];                          // run_before for zerosLeft 1..6: val << 4 | len
static std::once_flag g_once;

static void build_tok(TokTable &T, const uint8_t *len, const uint8_t *bits) {
    memset(T.t, 0, sizeof T.t);
    int nsub = 0;
    for (int s = 0; s < 68; s++) {
        int l = len[s]; if (!l) continue;
        uint32_t code = bits[s];
        if (l <= 8) {
            uint32_t base = code << (8 - l);
            for (uint32_t k = 0; k < (1u << (8 - l)); k++) T.t[base + k] = (uint16_t)(s << 8 | l);
        } else {
            uint32_t top = code >> (l - 8);
            int sub;
            if ((T.t[top] & 0xff) == 0xff) sub = T.t[top] >> 8;
            else { sub = ++nsub; T.t[top] = (uint16_t)(sub << 8 | 0xff); }
            uint32_t low = (code & ((1u << (l - 8)) - 1)) << (16 - l);
            for (uint32_t k = 0; k < (1u << (16 - l)); k++) T.t[256 * sub + low + k] = (uint16_t)(s << 8 | l);
        }
    }
}
void cavlc_init_tables() {
    std::call_once(g_once, [] {
        for (int i = 0; i < 3; i++) build_tok(g_tok[i], kTokLen[i], kTokBits[i]);
        // nC >= 8: 0000 11 -> (0,0); else total_coeff = (code >> 2) + 1, trailing_ones = code & 3
        for (int c = 0; c < 64; c++) { int tc = (c >> 2) + 1, t1 = c & 3; if (c == 3) { tc = 0; t1 = 0; } g_tok_flc[c] = (uint16_t)((4 * tc + t1) << 8 | 6); }
        memset(g_tok[3].t, 0, sizeof g_tok[3].t);
        for (int c = 0; c < 256; c++) g_tok[3].t[c] = g_tok_flc[c >> 2];
        memset(g_cdc, 0, sizeof g_cdc);
        for (int s = 0; s < 20; s++) if (kCdcLen[s]) { int l = kCdcLen[s]; uint32_t b = (uint32_t)kCdcBits[s] << (8 - l);
            for (uint32_t k = 0; k < (1u << (8 - l)); k++) g_cdc[b + k] = (uint16_t)(s << 8 | l); }
        memset(g_tz, 0, sizeof g_tz);
        for (int t = 0; t < 15; t++) for (int z = 0; z < 16 - t; z++) { int l = kTzLen[t][z]; uint32_t b = (uint32_t)kTzBits[t][z] << (9 - l);
            for (uint32_t k = 0; k < (1u << (9 - l)); k++) g_tz[t][b + k] = (uint16_t)(z << 8 | l); }
        memset(g_ctz, 0, sizeof g_ctz);
        for (int t = 0; t < 3; t++) for (int z = 0; z < 4 - t; z++) { int l = kCtzLen[t][z]; uint32_t b = (uint32_t)kCtzBits[t][z] << (3 - l);
            for (uint32_t k = 0; k < (1u << (3 - l)); k++) g_ctz[t][b + k] = (uint8_t)(z << 4 | l); }
        memset(g_run, 0, sizeof g_run);
        for (int t = 0; t < 6; t++) for (int r = 0; r <= t + 1; r++) { int l = kRunLen[t][r]; uint32_t b = (uint32_t)kRunBits[t][r] << (3 - l);
            for (uint32_t k = 0; k < (1u << (3 - l)); k++) g_run[t][b + k] = (uint8_t)(r << 4 | l); }
    });
}

void ParseScratch::resize(int w, int h) {
    if (w == mb_w && h == mb_h) return;
    mb_w = w; mb_h = h;
    size_t n = (size_t)w * h;
    tc.assign(n * 24, 0); mv.assign(n * 32, 0); refidx.assign(n * 4, -1); i4.assign(n * 16, 2); info.assign(n, 0); slice_of.assign(n, -1);
    cbp.assign(n, 0); cmode.assign(n, 0); cbf.assign(n, 0); mvd.assign(n * 32, 0);
    mv1.assign(n * 32, 0); refidx1.assign(n * 4, -1); mvd1.assign(n * 32, 0); direct8.assign(n, 0); uid0.assign(n * 4, -1); uid1.assign(n * 4, -1);
}
void ParseScratch::begin_picture() { std::fill(slice_of.begin(), slice_of.end(), (int16_t)-1); }

// ----------------------------------------------------------------------------------------
namespace {

struct Canon {                     // canonical per-MB serialisation for SyntaxDigest
    uint32_t addr; uint8_t kind /* | 16 when transform_size_8x8_flag: luma[] then holds 4 x 64 levels */, qp, cmode, i16mode; uint8_t i4[16]; int8_t ref[4];
    int16_t mv[16][2];
    int16_t i16dc[16], luma[16][16], cdc[2][4], cac[2][4][16];
    int8_t ref1[4]; int16_t mv1[16][2];          // list 1 (B slices)
};
static_assert(sizeof(Canon) == 4 + 4 + 16 + 4 + 64 + 32 + 512 + 16 + 256 + 4 + 64, "Canon must be packed");

struct P {
    const SeqParams &sps; const PicParamSet &pps; const SliceHeader &sh;
    BitReader &br; ParseScratch &cx; JobWriter &out; SyntaxDigest *dg;
    int slice_num; const SliceRefs &rf;
    int mb_w, mb_x = 0, mb_y = 0, addr = 0, qp;
    // neighbour MB indices or -1
    int nA = -1, nB = -1, nC = -1, nD = -1;
    uint8_t *tc; int16_t *mv; int8_t *ref; uint8_t *i4m;
    int16_t *mvl[2]; int8_t *refl[2]; uint8_t *mvdl[2];      // per list: [0] aliases mv / ref / mvd
    uint32_t decoded_mask = 0;
    Canon *canon = nullptr;
    Cabac *cb = nullptr;                 // non-null: entropy_coding_mode_flag = 1
    // a field picture (PAFF): every macroblock is a field macroblock -- field scans (8.5.6, 8.5.7) and the field contexts of significant_coeff_flag /
    // last_significant_coeff_flag (Table 9-34: 277.. / 338.. instead of 105.. / 166..)
    const uint8_t *zz4 = kZigzag4, *zz8 = kZigzag8; int fld_ctx = 0;
    bool last_dqp = false;               // mb_qp_delta of the previous macroblock in decoding order != 0
    uint8_t *mvd = nullptr;              // |mvd| per 4x4 and component of the current macroblock (CABAC ctxIdxInc)
    const char *err = nullptr;

    void locate(int a) {
        if (a == addr + 1 && mb_x + 1 < mb_w) mb_x++; else { mb_x = a % mb_w; mb_y = a / mb_w; }      // raster successor: no division
        addr = a;
        const int16_t *so = cx.slice_of.data();
        const int16_t sn = (int16_t)slice_num;
        nA = (mb_x > 0 && so[a - 1] == sn) ? a - 1 : -1;
        if (mb_y > 0) {
            const int b = a - mb_w;
            nB = so[b] == sn ? b : -1; nC = (mb_x + 1 < mb_w && so[b + 1] == sn) ? b + 1 : -1; nD = (mb_x > 0 && so[b - 1] == sn) ? b - 1 : -1;
        } else nB = nC = nD = -1;
        tc = &cx.tc[(size_t)a * 24]; mv = &cx.mv[(size_t)a * 32]; ref = &cx.refidx[(size_t)a * 4]; i4m = &cx.i4[(size_t)a * 16]; mvd = &cx.mvd[(size_t)a * 32];
        mvl[0] = mv; mvl[1] = &cx.mv1[(size_t)a * 32]; refl[0] = ref; refl[1] = &cx.refidx1[(size_t)a * 4]; mvdl[0] = mvd; mvdl[1] = &cx.mvd1[(size_t)a * 32];
    }
    // the same for the fast loop, which walks a slice's macroblocks in raster order: without FMO a slice is one contiguous run [first, ...), so a
    // neighbour belongs to this slice exactly when its address is not below the slice's first macroblock (no loads from slice_of)
    __attribute__((always_inline)) inline void locate_next(int a, int first) {
        if (a == addr + 1 && mb_x + 1 < mb_w) mb_x++; else { mb_x = a % mb_w; mb_y = a / mb_w; }
        addr = a;
        const int b = a - mb_w;
        nA = (mb_x > 0 && a - 1 >= first) ? a - 1 : -1;
        nB = b >= first ? b : -1;
        nC = (mb_x + 1 < mb_w && b + 1 >= first) ? b + 1 : -1;
        nD = (mb_x > 0 && b - 1 >= first) ? b - 1 : -1;
        tc = &cx.tc[(size_t)a * 24]; mv = &cx.mv[(size_t)a * 32]; ref = &cx.refidx[(size_t)a * 4]; i4m = &cx.i4[(size_t)a * 16]; mvd = &cx.mvd[(size_t)a * 32];
        mvl[0] = mv; mvl[1] = &cx.mv1[(size_t)a * 32]; refl[0] = ref; refl[1] = &cx.refidx1[(size_t)a * 4]; mvdl[0] = mvd; mvdl[1] = &cx.mvd1[(size_t)a * 32];
    }
    bool intra_usable(int n) const { return n >= 0 && (!pps.constrained_intra || (cx.info[n] & 1)); }

    // ---- 9.2.1 -------------------------------------------------------------------------
    int nc_luma(int bx, int by) const {
        int a = -1, b = -1;
        if (bx > 0) a = tc[by * 4 + bx - 1]; else if (nA >= 0) a = cx.tc[(size_t)nA * 24 + by * 4 + 3];
        if (by > 0) b = tc[(by - 1) * 4 + bx]; else if (nB >= 0) b = cx.tc[(size_t)nB * 24 + 12 + bx];
        if (a >= 0 && b >= 0) return (a + b + 1) >> 1;
        return a >= 0 ? a : (b >= 0 ? b : 0);
    }
    int nc_chroma(int pl, int bx, int by) const {
        int o = 16 + 4 * pl, a = -1, b = -1;
        if (bx > 0) a = tc[o + by * 2]; else if (nA >= 0) a = cx.tc[(size_t)nA * 24 + o + by * 2 + 1];
        if (by > 0) b = tc[o + bx]; else if (nB >= 0) b = cx.tc[(size_t)nB * 24 + o + 2 + bx];
        if (a >= 0 && b >= 0) return (a + b + 1) >> 1;
        return a >= 0 ? a : (b >= 0 ? b : 0);
    }

    // residual_block_cavlc: levels written at dst[map[scan_pos + first]]; returns total_coeff or -1.  dst is zeroed here, and only when the
    // block has levels (an empty block -- the common case inside a coded 8x8 -- costs the coeff_token lookup and nothing else)
    __attribute__((always_inline)) inline int residual_block(int nCtx, int max_num, int first, int16_t *dst, const uint8_t *map) {
        uint32_t e;
        if (nCtx < 0) { e = g_cdc[br.peek(8)]; }
        else if (nCtx >= 8) { e = g_tok_flc[br.peek(6)]; }
        else {
            const TokTable &T = g_tok[nCtx < 2 ? 0 : (nCtx < 4 ? 1 : 2)];
            uint32_t v = br.peek(16);
            e = T.t[v >> 8];
            if ((e & 0xff) == 0xff) e = T.t[256 * (e >> 8) + (v & 0xff)];
        }
        int len = e & 0xff;
        if (len == 0) return -1;
        br.skip(len);
        int total = (int)(e >> 10), t1 = (e >> 8) & 3;
        if (total == 0) return 0;
        if (total > max_num) return -1;
        return residual_levels(total, t1, max_num, first, dst, map);
    }
    __attribute__((noinline)) int residual_levels(int total, int t1, int max_num, int first, int16_t *dst, const uint8_t *map) {
        return levels_of(br, total, t1, max_num, first, dst, map); }
    // (static, explicit reader: the fast path runs it on a local copy of the reader that lives in registers)
    __attribute__((always_inline)) static inline int levels_of(BitReader &br, int total, int t1, int max_num, int first, int16_t *dst, const uint8_t *map) {
        memset(dst, 0, max_num == 4 ? 8 : 32);
        if (total == 1 && t1 == 1) {                          // the most frequent block by far: one level of +-1
            const int v = 1 - 2 * (int)br.u1();
            int z;
            if (max_num == 4) { const uint8_t e = g_ctz[0][br.peek(3)]; if (!(e & 15)) return -1; br.skip(e & 15); z = e >> 4; }
            else { const uint16_t e = g_tz[0][br.peek(9)]; if (!(e & 0xff)) return -1; br.skip(e & 0xff); z = e >> 8; }
            if (z + 1 > max_num) return -1;
            dst[map[z + first]] = (int16_t)v;
            return 1;
        }
        int level[16];
        int suffix_len = (total > 10 && t1 < 3) ? 1 : 0;
        int i = 0;
        {   // trailing_ones_sign_flag x t1: the next three bits give up to three signs (entries past t1 are overwritten below)
            const uint32_t s3 = br.peek(3);
            level[0] = 1 - (int)((s3 >> 1) & 2); level[1] = 1 - (int)(s3 & 2); level[2] = 1 - (int)((s3 << 1) & 2);
            br.skip(t1); i = t1;
        }
        for (; i < total; i++) {
            uint32_t w = br.peek(32);
            if (w == 0) return -1;
            int prefix = __builtin_clz(w);
            int code;
            if (prefix < 14) {                                // prefix, stop bit and suffix (at most 20 bits) are all inside w: one skip
                code = (prefix << suffix_len) + (int)((w >> (31 - prefix - suffix_len)) & ((1u << suffix_len) - 1));
                br.skip(prefix + 1 + suffix_len);
            } else {
                br.skip(prefix + 1);
                code = (prefix < 15 ? prefix : 15) << suffix_len;
                int size = (prefix == 14 && suffix_len == 0) ? 4 : (prefix >= 15 ? prefix - 3 : suffix_len);
                if (size > 0) code += (int)br.u(size);
                if (prefix >= 15 && suffix_len == 0) code += 15;
                if (prefix >= 16) code += (1 << (prefix - 3)) - 4096;
            }
            if (i == t1 && t1 < 3) code += 2;
            int lv = (code & 1) ? (-code - 1) >> 1 : (code + 2) >> 1;
            level[i] = lv;
            if (suffix_len == 0) suffix_len = 1;
            int a = lv < 0 ? -lv : lv;
            if (a > (3 << (suffix_len - 1)) && suffix_len < 6) suffix_len++;
        }
        int zeros_left = 0;
        if (total < max_num) {
            if (max_num == 4) { uint8_t z = g_ctz[total - 1][br.peek(3)]; if (!(z & 15)) return -1; br.skip(z & 15); zeros_left = z >> 4; }
            else { uint16_t z = g_tz[total - 1][br.peek(9)]; if (!(z & 0xff)) return -1; br.skip(z & 0xff); zeros_left = z >> 8; }
            if (zeros_left + total > max_num) return -1;
        }
        int pos = zeros_left + total - 1;                 // scan position of the first (highest) level
        for (i = 0; i < total; i++) {
            dst[map[pos + first]] = (int16_t)level[i];
            if (i == total - 1) break;
            int run = 0;
            if (zeros_left > 0) {
                if (zeros_left <= 6) { uint8_t r = g_run[zeros_left - 1][br.peek(3)]; br.skip(r & 15); run = r >> 4; }
                else {
                    uint32_t w3 = br.peek(3);
                    if (w3) { run = 7 - (int)w3; br.skip(3); }
                    else { uint32_t w = br.peek(16); if (!w) return -1; int lz = __builtin_clz(w) - 16; run = lz + 4; br.skip(lz + 1); }
                }
                if (run > zeros_left) return -1;
                zeros_left -= run;
            }
            pos -= run + 1;
        }
        return total;
    }

    // ---- motion vector prediction (8.4.1.3) ---------------------------------------------
    struct Nb { bool avail; int ref; int mvx, mvy; };
    Nb nb(int bx, int by, int l = 0) const {
        Nb n{false, -1, 0, 0};
        int m, rx, ry;
        if (by < 0) { ry = 3; if (bx < 0) { m = nD; rx = 3; } else if (bx > 3) { m = nC; rx = bx - 4; } else { m = nB; rx = bx; } }
        else if (bx < 0) { m = nA; rx = 3; ry = by; }
        else if (bx > 3) return n;
        else { if (!((decoded_mask >> (by * 4 + bx)) & 1)) return n; m = addr; rx = bx; ry = by; }
        if (m < 0) return n;
        n.avail = true;
        n.ref = (l ? cx.refidx1 : cx.refidx)[(size_t)m * 4 + (ry >> 1) * 2 + (rx >> 1)];
        if (n.ref >= 0) { const int16_t *q = &(l ? cx.mv1 : cx.mv)[(size_t)m * 32 + (ry * 4 + rx) * 2]; n.mvx = q[0]; n.mvy = q[1]; }
        return n;
    }
    static int med(int a, int b, int c) { int mx = a > b ? a : b, mn = a < b ? a : b; return c > mx ? mx : (c < mn ? mn : c); }
    void predict(int bx, int by, int bw, int refi, int shape, int part, int &px, int &py, int l = 0) const {
        Nb A = nb(bx - 1, by, l), B = nb(bx, by - 1, l), C = nb(bx + bw, by - 1, l);
        if (!C.avail) C = nb(bx - 1, by - 1, l);
        if (shape == 1) { if (part == 0) { if (B.ref == refi) { px = B.mvx; py = B.mvy; return; } } else if (A.ref == refi) { px = A.mvx; py = A.mvy; return; }
            }
        else if (shape == 2) { if (part == 0) { if (A.ref == refi) { px = A.mvx; py = A.mvy; return; } } else if (C.ref == refi) { px = C.mvx; py = C.mvy;
            return; } }
        if (!B.avail && !C.avail && A.avail) { B = A; C = A; }
        int ma = A.ref == refi, mb = B.ref == refi, mc = C.ref == refi;
        if (ma + mb + mc == 1) { const Nb &n = ma ? A : (mb ? B : C); px = n.mvx; py = n.mvy; }
        else { px = med(A.mvx, B.mvx, C.mvx); py = med(A.mvy, B.mvy, C.mvy); }
    }
    void set_mv(int bx, int by, int bw, int bh, int x, int y, int l = 0) {
        if (bw == 4 && bh == 4) {                              // whole macroblock: 16 identical pairs
            const uint32_t v = (uint32_t)(uint16_t)x | ((uint32_t)(uint16_t)y << 16);
            uint32_t *d = (uint32_t *)mvl[l];
            for (int i = 0; i < 16; i++) d[i] = v;
            decoded_mask = 0xffff;
            return;
        }
        for (int j = by; j < by + bh; j++) for (int i = bx; i < bx + bw; i++) { mvl[l][(j * 4 + i) * 2] = (int16_t)x; mvl[l][(j * 4 + i) * 2 + 1] = (int16_t)y;
            decoded_mask |= 1u << (j * 4 + i); }
    }

    // ---- macroblock start / finish -------------------------------------------------------
    MbRec *begin_mb() {
        MbRec *r = &out.mbs[addr];
        memset(r, 0, sizeof *r);
        r->flags = MBF_DECODED; r->slice = (uint8_t)slice_num; r->coef_off = out.coef_count;
        r->ref[0] = r->ref[1] = r->ref[2] = r->ref[3] = -1;
        cx.slice_of[addr] = (int16_t)slice_num;
        memset(tc, 0, 24); ref[0] = ref[1] = ref[2] = ref[3] = -1; memset(mv, 0, 64); memset(i4m, 2, 16);
        cx.info[addr] = 0; decoded_mask = 0;
        if (sh.type == SL_B || rf.track_uid) { refl[1][0] = refl[1][1] = refl[1][2] = refl[1][3] = -1; memset(mvl[1], 0, 64); memset(mvdl[1], 0, 32);
            cx.direct8[addr] = 0; }
        if (cb) { cx.cbp[addr] = 0; cx.cmode[addr] = 0; cx.cbf[addr] = 0; memset(mvd, 0, 32); }
        if (canon) { memset(canon, 0, sizeof *canon); canon->addr = (uint32_t)addr; canon->ref[0] = canon->ref[1] = canon->ref[2] = canon->ref[3] = -1;
            canon->ref1[0] = canon->ref1[1] = canon->ref1[2] = canon->ref1[3] = -1; }
        return r;
    }
    void finish_mb(const MbRec *r) {
        if (!dg) return;
        canon->kind = r->kind; canon->qp = r->qp;
        if ((r->kind & 15) == MB_INTER) { for (int i = 0; i < 4; i++) canon->ref[i] = ref[i]; memcpy(canon->mv, mv, 64);
            if (sh.type == SL_B) { for (int i = 0; i < 4; i++) canon->ref1[i] = refl[1][i]; memcpy(canon->mv1, mvl[1], 64); } }
        const uint8_t *p = (const uint8_t *)canon;
        uint64_t h = dg->h;
        for (size_t i = 0; i < sizeof(Canon); i++) { h ^= p[i]; h *= 1099511628211ull; }
        dg->h = h; dg->mbs++;
    }
    int16_t *alloc_coef(int n) {
        if (out.coef_count + (uint32_t)n > out.coef_cap) { err = "coefficient buffer overflow"; return nullptr; }
        int16_t *p = out.coef + out.coef_count; out.coef_count += n;
        memset(p, 0, sizeof(int16_t) * n);
        return p;
    }
    void write_motion(MbRec *r, bool sub8) {
        for (int i = 0; i < 4; i++) r->ref[i] = ref[i] >= 0 ? rf.slot[0][ref[i]] : (int8_t)-1;
        {   // downward reach of this macroblock's vectors (one vector per 8x8 unless sub-8x8 partitions / a second list exist)
            int m = out.max_mvy, mx = out.max_mvx;
            if (!sub8 && !rf.bipred_rec) { for (int i = 0; i < 4; i++) { int b = (i >> 1) * 8 + (i & 1) * 2; m = mv[b * 2 + 1] > m ? mv[b * 2 + 1] : m;
                mx = mv[b * 2] > mx ? mv[b * 2] : mx; } }
            else {
                // 32 (x, y) pairs per list: lane-wise maxima over the sixteen-byte pieces, then over the x lanes / the y lanes (every inter macroblock of a
                // B slice comes through here: the scalar loop was 8 % of a High + B picture's parse)
                __m128i a = _mm_loadu_si128((const __m128i *)mvl[0]);
                for (int i = 1; i < 4; i++) a = _mm_max_epi16(a, _mm_loadu_si128((const __m128i *)mvl[0] + i));
                if (rf.bipred_rec) for (int i = 0; i < 4; i++) a = _mm_max_epi16(a, _mm_loadu_si128((const __m128i *)mvl[1] + i));
                a = _mm_max_epi16(a, _mm_shuffle_epi32(a, 0x4e)); a = _mm_max_epi16(a, _mm_shuffle_epi32(a, 0xb1));      // lanes 0 / 1 = max x / max y
                const int ax = (int16_t)_mm_extract_epi16(a, 0), ay = (int16_t)_mm_extract_epi16(a, 1);
                mx = ax > mx ? ax : mx; m = ay > m ? ay : m;
            }
            out.max_mvy = m; out.max_mvx = mx;
        }
        if (rf.track_uid) for (int l = 0; l < 2; l++) { int32_t *u = &(l ? cx.uid1 : cx.uid0)[(size_t)addr * 4];
            for (int i = 0; i < 4; i++) u[i] = refl[l][i] >= 0 ? rf.uid[l][refl[l][i]] : -1; }
        if (rf.bipred_rec) {                                   // B slice / weighted prediction: full motion record (jobs.h MBM_BIPRED)
            if (out.mv_ext_count + kBiRecInt16 / 2 > out.mv_ext_cap) { err = "mv_ext overflow"; return; }
            r->modes |= MBM_BIPRED; r->u.mv_ext = out.mv_ext_count;
            int16_t *d = out.mv_ext + (size_t)out.mv_ext_count * 2;
            memcpy(d, mvl[0], 64); memcpy(d + 32, mvl[1], 64);
            int8_t *t = (int8_t *)(d + 64);
            for (int i = 0; i < 4; i++) { t[i] = refl[1][i] >= 0 ? rf.slot[1][refl[1][i]] : (int8_t)-1; t[4 + i] = ref[i]; t[8 + i] = refl[1][i];
                t[12 + i] = 0; }
            out.mv_ext_count += kBiRecInt16 / 2;
            return;
        }
        if (!sub8) { for (int i = 0; i < 4; i++) { int b = (i >> 1) * 8 + (i & 1) * 2; r->u.mv[i][0] = mv[b * 2]; r->u.mv[i][1] = mv[b * 2 + 1]; } }
        else {
            if (out.mv_ext_count + 16 > out.mv_ext_cap) { err = "mv_ext overflow"; return; }
            r->flags |= MBF_MV_EXT; r->u.mv_ext = out.mv_ext_count;
            memcpy(out.mv_ext + (size_t)out.mv_ext_count * 2, mv, 64); out.mv_ext_count += 16;
        }
    }

    // ---- direct prediction (8.4.1.2) of the 8x8 quadrants in mask -------------------------------------
    static int min_pos(int a, int b) { return (a >= 0 && b >= 0) ? (a < b ? a : b) : (a > b ? a : b); }
    // motion that goes with block r of the current macroblock in the colocated picture (8.4.1.2.1).  The vertical vector is returned as stored; the
    // identity of the referenced picture in the CURRENT picture's terms (Frm_To_Fld: the field of the current parity of that frame; Fld_To_Frm: the frame
    // that holds that field)
    void col_block(int r, int &refc, int &mx, int &my, int32_t &uid) const {
        const MotionField &c = *rf.col;
        size_t a = (size_t)addr;
        if (rf.col_mode == 1) {                                    // mbAddrCol6, yM = (2 * yCol) % 16
            const int yCol = (r >> 2) * 4, xCol = (r & 3) * 4;
            a = (size_t)(2 * mb_y + yCol / 8) * mb_w + mb_x; r = (((2 * yCol) % 16) >> 2) * 4 + (xCol >> 2);
        } else if (rf.col_mode == 2) {                             // mbAddrCol7, yM = 8 * (mb_y % 2) + 4 * (yCol / 8)
            const int yCol = (r >> 2) * 4, xCol = (r & 3) * 4, yM = 8 * (mb_y % 2) + 4 * (yCol / 8);
            a = (size_t)(mb_y / 2) * mb_w + mb_x; r = (yM >> 2) * 4 + (xCol >> 2);
        }
        refc = -1; mx = my = 0; uid = -1;
        if (a >= c.intra.size() || c.intra[a]) return;
        int b8 = (r >> 3) * 2 + ((r & 3) >> 1);
        int l = c.ref[0][a * 4 + b8] >= 0 ? 0 : 1;
        refc = c.ref[l][a * 4 + b8];
        if (refc < 0) return;
        mx = c.mv[l][a * 32 + r * 2]; my = c.mv[l][a * 32 + r * 2 + 1]; uid = c.uid[l][a * 4 + b8];
        if (rf.col_mode == 1) uid = (uid & ~1) | rf.cur_parity; else if (rf.col_mode == 2) uid &= ~1;
    }
    bool direct_pred(int mask) {
        if (!rf.col || rf.slot[1][0] < 0) { err = "direct prediction without RefPicList1[0]"; return false; }
        const bool inf8 = sps.direct_8x8_inference;
        if (sh.direct_spatial_mv_pred) {
            int rr[2], mvp[2][2] = {{0, 0}, {0, 0}};
            uint32_t saved = decoded_mask; decoded_mask = 0;          // only neighbouring macroblocks take part
            for (int l = 0; l < 2; l++) {
                Nb A = nb(-1, 0, l), B = nb(0, -1, l), C = nb(4, -1, l);
                if (!C.avail) C = nb(-1, -1, l);
                rr[l] = min_pos(A.ref, min_pos(B.ref, C.ref));
            }
            bool zero = rr[0] < 0 && rr[1] < 0;
            if (zero) rr[0] = rr[1] = 0;
            else for (int l = 0; l < 2; l++) if (rr[l] >= 0) predict(0, 0, 4, rr[l], 0, 0, mvp[l][0], mvp[l][1], l);
            decoded_mask = saved;
            const bool col_short = !rf.is_long[1][0];
            for (int q = 0; q < 4; q++) {
                if (!(mask & (1 << q))) continue;
                refl[0][q] = (int8_t)rr[0]; refl[1][q] = (int8_t)rr[1];
                bool col_zero = false;
                if (inf8) {
                    // direct_8x8_inference: the four blocks of the quadrant share the corner block of the colocated macroblock and therefore one vector
                    // per list -- two rows of two (x, y) pairs, stored as two 64-bit words
                    int refc, cx_, cy_; int32_t uid;
                    col_block((q >> 1) * 12 + (q & 1) * 3, refc, cx_, cy_, uid);
                    col_zero = col_short && refc == 0 && cx_ >= -1 && cx_ <= 1 && cy_ >= -1 && cy_ <= 1;
                    const int r0 = (q >> 1) * 8 + (q & 1) * 2;
                    for (int l = 0; l < 2; l++) {
                        const bool z = zero || rr[l] < 0 || (rr[l] == 0 && col_zero);
                        const uint32_t v = z ? 0u : ((uint32_t)(uint16_t)(int16_t)mvp[l][0] | (uint32_t)(uint16_t)(int16_t)mvp[l][1] << 16);
                        const uint64_t two = (uint64_t)v | (uint64_t)v << 32;
                        memcpy(mvl[l] + r0 * 2, &two, 8); memcpy(mvl[l] + (r0 + 4) * 2, &two, 8);
                    }
                    continue;
                }
                for (int k = 0; k < 4; k++) {
                    int r = ((q >> 1) * 2 + (k >> 1)) * 4 + (q & 1) * 2 + (k & 1);
                    int refc, cx_, cy_; int32_t uid;
                    col_block(r, refc, cx_, cy_, uid);
                    col_zero = col_short && refc == 0 && cx_ >= -1 && cx_ <= 1 && cy_ >= -1 && cy_ <= 1;
                    for (int l = 0; l < 2; l++) {
                        bool z = zero || rr[l] < 0 || (rr[l] == 0 && col_zero);
                        mvl[l][r * 2] = (int16_t)(z ? 0 : mvp[l][0]); mvl[l][r * 2 + 1] = (int16_t)(z ? 0 : mvp[l][1]);
                    }
                }
            }
        } else {
            for (int q = 0; q < 4; q++) {
                if (!(mask & (1 << q))) continue;
                int m0x = 0, m0y = 0, m1x = 0, m1y = 0, r0 = 0;
                for (int k = 0; k < 4; k++) {
                    int r = ((q >> 1) * 2 + (k >> 1)) * 4 + (q & 1) * 2 + (k & 1);
                    if (k > 0 && inf8) {                               // the quadrant's one colocated block: the vectors of block 0 again
                        mvl[0][r * 2] = (int16_t)m0x; mvl[0][r * 2 + 1] = (int16_t)m0y; mvl[1][r * 2] = (int16_t)m1x; mvl[1][r * 2 + 1] = (int16_t)m1y;
                        continue;
                    }
                    int rc = inf8 ? (q >> 1) * 12 + (q & 1) * 3 : r, refc, cmx, cmy; int32_t uid;
                    r0 = 0;
                    col_block(rc, refc, cmx, cmy, uid);
                    if (rf.col_mode == 1) cmy = cmy / 2; else if (rf.col_mode == 2) cmy *= 2;      // 8.4.1.2.3: Frm_To_Fld ("/" towards zero), Fld_To_Frm
                    if (refc >= 0) {
                        r0 = -1;
                        for (int i = 0; i < sh.num_ref_idx[0]; i++) if (rf.slot[0][i] >= 0 && rf.uid[0][i] == uid) { r0 = i; break; }
                        if (r0 < 0) { err = "temporal direct: colocated reference is not in RefPicList0"; return false; }
                    }
                    if (rf.slot[0][r0] < 0) { err = "temporal direct without its list-0 reference"; return false; }
                    int tb = rf.cur_poc - rf.poc[0][r0], td = rf.poc[1][0] - rf.poc[0][r0];
                    tb = tb < -128 ? -128 : (tb > 127 ? 127 : tb); td = td < -128 ? -128 : (td > 127 ? 127 : td);
                    if (rf.is_long[0][r0] || td == 0) { m0x = cmx; m0y = cmy; m1x = m1y = 0; }
                    else {
                        int tx = (16384 + (td < 0 ? -td : td) / 2) / td, dsf = (tb * tx + 32) >> 6;
                        dsf = dsf < -1024 ? -1024 : (dsf > 1023 ? 1023 : dsf);
                        m0x = (dsf * cmx + 128) >> 8; m0y = (dsf * cmy + 128) >> 8; m1x = m0x - cmx; m1y = m0y - cmy;
                    }
                    refl[0][q] = (int8_t)r0; refl[1][q] = 0;
                    mvl[0][r * 2] = (int16_t)m0x; mvl[0][r * 2 + 1] = (int16_t)m0y; mvl[1][r * 2] = (int16_t)m1x; mvl[1][r * 2 + 1] = (int16_t)m1y;
                }
            }
        }
        cx.direct8[addr] |= (uint8_t)mask;
        return true;
    }
    void mark8(int q) { int bx = (q & 1) * 2, by = (q >> 1) * 2;
        decoded_mask |= (1u << (by * 4 + bx)) | (1u << (by * 4 + bx + 1)) | (1u << (by * 4 + bx + 4)) | (1u << (by * 4 + bx + 5)); }

    bool skip_mb() {
        MbRec *r = begin_mb();
        r->kind = MB_INTER; r->qp = (uint8_t)qp;
        cx.info[addr] = 4; last_dqp = false;
        if (sh.type == SL_B) {                                  // B_Skip: direct prediction, no residual
            cx.direct8[addr] = 16;                               // bit 4: B_Skip / B_Direct_16x16 (mb_type context); direct_pred adds the quadrant bits
            if (!direct_pred(15)) return false;
            write_motion(r, true);
            finish_mb(r);
            return err == nullptr;
        }
        int px = 0, py = 0;
        if (nA >= 0 && nB >= 0) {
            Nb A = nb(-1, 0), B = nb(0, -1);
            if (!((A.ref == 0 && !A.mvx && !A.mvy) || (B.ref == 0 && !B.mvx && !B.mvy))) predict(0, 0, 4, 0, 0, 0, px, py);
        }
        ref[0] = ref[1] = ref[2] = ref[3] = 0;
        set_mv(0, 0, 4, 4, px, py);
        write_motion(r, false);
        finish_mb(r);
        return err == nullptr;
    }


    // ---- fast path (P slices with CAVLC, one motion record format, no digest): P_Skip and P_L0_16x16 -- 96 % of the macroblocks of a typical
    // P picture -- without the general partition / list machinery.  Same arrays, same records: any other macroblock type of the slice goes
    // through macroblock() and the two mix freely.  (tests/test_host_parser.py compares the job lists of both paths.)
    // motion vector and reference of the 4x4 block r (8x8 quadrant b8) of neighbouring macroblock m; a macroblock that is not there or is intra has
    // ref -1 and vector 0 (begin_mb clears the vectors of intra macroblocks)
    struct Nv { int ref, x, y; };
    inline Nv nbv(int m, int r, int b8) const {
        if (m < 0) return Nv{-2, 0, 0};                             // -2: not available (8.4.1.3.2), never equal to a reference index
        const int16_t *q = &cx.mv[(size_t)m * 32 + r * 2];
        return Nv{cx.refidx[(size_t)m * 4 + b8], q[0], q[1]};
    }
    // 8.4.1.3 for a 16x16 partition: neighbours A (left, block 3), B (above, block 12), C (above right, block 12; else D above left, block 15)
    inline void mvp16(int refi, const Nv &A, const Nv &B, int &px, int &py) const {
        const Nv C = nC >= 0 ? nbv(nC, 12, 2) : nbv(nD, 15, 3);
        if (B.ref == -2 && C.ref == -2 && A.ref != -2) { px = A.x; py = A.y; return; }
        const int ma = A.ref == refi, mb = B.ref == refi, mc = C.ref == refi;
        if (ma + mb + mc == 1) { const Nv &n = ma ? A : (mb ? B : C); px = n.x; py = n.y; }
        else { px = med(A.x, B.x, C.x); py = med(A.y, B.y, C.y); }
    }
    // the macroblock's record and neighbour state for one vector / one reference index over the whole macroblock
    // (the record is put together in four 64-bit words and stored once: byte-wise assembly followed by a 32-byte copy stalls on store forwarding)
    inline void put_inter16(MbRec *r, int refi, int x, int y, uint32_t coef_off, uint32_t flags, uint32_t cbp_blk, uint32_t cbp_cac) {
        const uint32_t v = (uint32_t)(uint16_t)x | ((uint32_t)(uint16_t)y << 16);
        const uint64_t vv = (uint64_t)v << 32 | v;
        uint64_t *d = (uint64_t *)mv;
        for (int i = 0; i < 8; i++) d[i] = vv;
        const uint32_t rr = (uint8_t)refi * 0x01010101u;
        memcpy(ref, &rr, 4);
        cx.slice_of[addr] = (int16_t)slice_num;
        // the stream may hold B pictures: this picture's motion may serve direct prediction later (list 1 unused here)
        if (rf.track_uid) {
            int32_t *u0 = &cx.uid0[(size_t)addr * 4], *u1 = &cx.uid1[(size_t)addr * 4];
            const int32_t u = rf.uid[0][refi];
            u0[0] = u0[1] = u0[2] = u0[3] = u; u1[0] = u1[1] = u1[2] = u1[3] = -1;
            refl[1][0] = refl[1][1] = refl[1][2] = refl[1][3] = -1;
            cx.direct8[addr] = 0;
        }
        const int my = (int16_t)y, mx = (int16_t)x;
        if (my > out.max_mvy) out.max_mvy = my;
        if (mx > out.max_mvx) out.max_mvx = mx;
        static_assert(MB_INTER == 0 && offsetof(MbRec, cbp_blk) == 4 && offsetof(MbRec, coef_off) == 8 && offsetof(MbRec, ref) == 12 && offsetof(MbRec,
            u) == 16, "record layout");
        uint64_t *q = (uint64_t *)r;
        q[0] = (uint64_t)(uint8_t)qp << 8 | (uint64_t)(flags | MBF_DECODED) << 24 | (uint64_t)cbp_blk << 32 | (uint64_t)cbp_cac << 48 |
            (uint64_t)(uint8_t)slice_num << 56;
        q[1] = (uint64_t)coef_off | (uint64_t)((uint8_t)rf.slot[0][refi] * 0x01010101u) << 32;
        q[2] = vv; q[3] = vv;
    }
    __attribute__((always_inline)) inline void skip_mb_fast() {
        int px = 0, py = 0;
        if (nA >= 0 && nB >= 0) {
            const Nv A = nbv(nA, 3, 1), B = nbv(nB, 12, 2);
            if (!((A.ref == 0 && !A.x && !A.y) || (B.ref == 0 && !B.x && !B.y))) mvp16(0, A, B, px, py);
        }
        memset(tc, 0, 24);
        cx.info[addr] = 4;
        put_inter16(&out.mbs[addr], 0, px, py, out.coef_count, 0, 0, 0);
    }
    // total_coeff of one block (9.2.1) given the sum s of its left and upper neighbours' counts in the window (64 = not available): both there ->
    // rounded mean, one -> that one, none -> 0
    static inline int nc_of(int s) { return s < 64 ? (s + 1) >> 1 : s & 31; }
    // coeff_token: total_coeff << 2 | trailing_ones, or -1; sum: see kTokClassOfSum
    __attribute__((always_inline)) static inline int token_fast(BitReader &br, int sum) {
        const int cls = kTokClassOfSum.t[sum];
        const uint32_t v = br.peek(16);
        if (cls == 0 && (v & 0x8000)) { br.skip(1); return 0; }     // nC < 2 and the codeword "1": an empty block (most blocks of a coded 8x8 are)
        const TokTable &T = g_tok[cls];
        uint32_t e = T.t[v >> 8];
        if ((e & 0xff) == 0xff) e = T.t[256 * (e >> 8) + (v & 0xff)];
        const int len = e & 0xff;
        if (len == 0) return -1;
        br.skip(len);
        return (int)(e >> 8);
    }
    // residual() of a CAVLC macroblock without the 8x8 transform: coded_block_pattern cbp, i16 = Intra16x16 (DC block + 15-level AC blocks)
    __attribute__((always_inline)) inline bool residual_fast(BitReader &br, int cbp, const bool i16, uint32_t &flags, uint32_t &bits, uint32_t &cbm) {
        static const uint8_t ident[16] = {0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15};
        // block (decoding order) -> window index (by + 1) * 8 + bx + 1
        static const uint8_t kWin[16] = {9, 10, 17, 18, 11, 12, 19, 20, 25, 26, 33, 34, 27, 28, 35, 36};
        static const uint8_t kRas[16] = {0, 1, 4, 5, 2, 3, 6, 7, 8, 9, 12, 13, 10, 11, 14, 15};                 // ... -> raster index by * 4 + bx
        // window of total_coeff: row 0 / column 0 = the macroblocks above / to the left
        alignas(8) uint8_t w[5 * 8];
        const uint8_t *ta = nA >= 0 ? &cx.tc[(size_t)nA * 24] : nullptr, *tb = nB >= 0 ? &cx.tc[(size_t)nB * 24] : nullptr;
        flags = bits = cbm = 0;
        memset(tc, 0, 24);
        // one room check per macroblock: the most this coded_block_pattern can store (16 slots per 4x4 block, 4 per chroma DC block; the Intra16x16 DC
        // block is allocated -- and checked -- by alloc_coef)
        const uint32_t need = (i16 ? 16u : 0u) + 64u * (uint32_t)__builtin_popcount(cbp & 15) + ((cbp & 0x30) ? 8u : 0u) + ((cbp & 0x20) ? 128u : 0u);
        if (out.coef_count + need > out.coef_cap) { err = "coefficient buffer overflow"; return false; }
        if ((cbp & 15) || i16) {
            memset(w, 0, sizeof w);
            if (tb) memcpy(w + 1, tb + 12, 4); else memset(w + 1, 64, 4);
            for (int j = 0; j < 4; j++) w[8 * (j + 1)] = ta ? ta[4 * j + 3] : 64;
            if (i16) {                                        // Intra16x16 DC levels: always 16 slots in the stream
                int16_t *d = alloc_coef(16); if (!d) return false;
                const int tk = token_fast(br, w[8] + w[1]);
                if (tk < 0 || (tk && levels_of(br, tk >> 2, tk & 3, 16, 0, d, zz4) < 0)) { err = "entropy error (Intra16x16 DC)"; return false; }
            }
            const int max_num = i16 ? 15 : 16, first = i16 ? 1 : 0;
            for (int b8 = 0; b8 < 4; b8++) {
                if (!(cbp & (1 << b8))) continue;
                for (int k = 0; k < 4; k++) {
                    const int blk = b8 * 4 + k, wi = kWin[blk];
                    const int tk = token_fast(br, w[wi - 1] + w[wi - 8]);
                    if (tk < 0) { err = "entropy error (luma block)"; return false; }
                    if (!tk) continue;
                    const int total = tk >> 2;
                    int16_t *d = out.coef + out.coef_count;
                    if (total > max_num || levels_of(br, total, tk & 3, max_num, first, d, zz4) < 0) { err = "entropy error (luma block)"; return false; }
                    w[wi] = (uint8_t)total; tc[kRas[blk]] = (uint8_t)total; bits |= 1u << blk; out.coef_count += 16;
                }
            }
        }
        if (cbp & 0x30) {
            for (int pl = 0; pl < 2; pl++) {
                int16_t *d = out.coef + out.coef_count;
                const uint32_t e = g_cdc[br.peek(8)];
                if (!(e & 0xff)) { err = "entropy error (chroma DC)"; return false; }
                br.skip((int)(e & 0xff));
                if (!(e >> 10)) continue;
                if ((int)(e >> 10) > 4 || levels_of(br, (int)(e >> 10), (int)(e >> 8) & 3, 4, 0, d, ident) < 0) { err = "entropy error (chroma DC)";
                    return false; }
                flags |= pl ? MBF_CR_DC : MBF_CB_DC; out.coef_count += 4;
            }
        }
        if (cbp & 0x20) {
            for (int pl = 0; pl < 2; pl++) {
                const int o = 16 + 4 * pl;
                // 3x3 window of the plane: [0] unused, [1] [2] above, [3] [6] left
                uint8_t c[9];
                c[1] = tb ? tb[o + 2] : 64; c[2] = tb ? tb[o + 3] : 64; c[3] = ta ? ta[o + 1] : 64; c[6] = ta ? ta[o + 3] : 64;
                for (int k = 0; k < 4; k++) {
                    const int ci = 4 + (k >> 1) * 3 + (k & 1);
                    const int tk = token_fast(br, c[ci - 1] + c[ci - 3]);
                    if (tk < 0 || (tk >> 2) > 15) { err = "entropy error (chroma AC)"; return false; }
                    c[ci] = (uint8_t)(tk >> 2);
                    if (!tk) continue;
                    int16_t *d = out.coef + out.coef_count;
                    if (levels_of(br, tk >> 2, tk & 3, 15, 1, d, zz4) < 0) { err = "entropy error (chroma AC)"; return false; }
                    tc[o + k] = (uint8_t)(tk >> 2); cbm |= 1u << (4 * pl + k); out.coef_count += 16;
                }
            }
        }
        return true;
    }
    __attribute__((always_inline)) inline bool p16x16_fast(BitReader &br) {
        int refi = 0;
        const int nref = sh.num_ref_idx[0];
        if (nref > 1) { refi = br.te(nref - 1); if (refi >= nref) { err = "ref_idx out of range"; return false; } }
        int px, py;
        mvp16(refi, nbv(nA, 3, 1), nbv(nB, 12, 2), px, py);
        const int dx = br.se(), dy = br.se();
        const uint32_t code = br.ue();
        if (code > 47) { err = "bad coded_block_pattern"; return false; }
        const int cbp = kCbpInter[code];
        MbRec *r = &out.mbs[addr];
        cx.info[addr] = 0;
        if (cbp) {
            const int dqp = br.se();
            if (dqp < -26 || dqp > 25) { err = "mb_qp_delta out of range"; return false; }
            qp += dqp; if (qp < 0) qp += 52; else if (qp > 51) qp -= 52;
            const uint32_t coef_off = out.coef_count;
            uint32_t flags, bits, cbm;
            if (!residual_fast(br, cbp, false, flags, bits, cbm)) return false;
            put_inter16(r, refi, px + dx, py + dy, coef_off, flags, bits, cbm);
        } else {
            memset(tc, 0, 24);
            put_inter16(r, refi, px + dx, py + dy, out.coef_count, 0, 0, 0);
        }
        if (br.overrun()) { err = "macroblock data truncated"; return false; }
        return true;
    }

    // The same two macroblock types in CABAC slices: the syntax elements go through the general ae_* routines (their contexts come from the neighbour
    // arrays, which are kept exactly as macroblock() keeps them); what is saved is the partition machinery around them.
    inline void skip_mb_fast_ae() {
        skip_mb_fast();
        cx.cbp[addr] = 0; cx.cmode[addr] = 0; cx.cbf[addr] = 0; memset(mvd, 0, 32); last_dqp = false;
    }
    bool p16x16_fast_ae() {
        MbRec *r = &out.mbs[addr];
        cx.info[addr] = 0; cx.cbp[addr] = 0; cx.cmode[addr] = 0; cx.cbf[addr] = 0;
        memset(tc, 0, 24);
        int refi = 0;
        const int nref = sh.num_ref_idx[0];
        if (nref > 1) { refi = ae_ref_idx(0, 0, 0); if (err || refi >= nref) { if (!err) err = "ref_idx out of range"; return false; } }
        int px, py, dx, dy;
        mvp16(refi, nbv(nA, 3, 1), nbv(nB, 12, 2), px, py);
        read_mvd(0, 0, 4, 4, dx, dy, 0);
        if (err) return false;
        const int cbp = ae_cbp();
        cx.cbp[addr] = (uint8_t)cbp;
        bool t8 = false;
        if ((cbp & 15) && pps.transform8x8) { t8 = ae_t8x8() != 0; if (t8) cx.info[addr] |= 8; }
        if (cbp) {
            const int dqp = ae_qp_delta();
            if (err || dqp < -26 || dqp > 25) { if (!err) err = "mb_qp_delta out of range"; return false; }
            qp += dqp; if (qp < 0) qp += 52; else if (qp > 51) qp -= 52;
            last_dqp = dqp != 0;
        } else last_dqp = false;
        put_inter16(r, refi, px + dx, py + dy, out.coef_count, 0, 0, 0);
        if (t8) r->modes |= MBM_T8X8;
        if (cbp && !residual(r, cbp, false, t8, false)) return false;
        if (cb->overrun) { err = "macroblock data truncated"; return false; }
        return true;
    }

    // ---- CABAC syntax elements (9.3.2 binarisation, 9.3.3.1 ctxIdxInc) ------------------------
    // macroblock holding the 4x4 block left of / above block (bx,by) and that block's raster index; -1 = not available
    int nb4(int bx, int by, bool left, int &r) const {
        if (left) { if (bx > 0) { r = by * 4 + bx - 1; return addr; } r = by * 4 + 3; return nA; }
        if (by > 0) { r = (by - 1) * 4 + bx; return addr; }
        r = 12 + bx; return nB;
    }
    int ae_mb_skip() { return cb->decision((sh.type == SL_B ? 24 : 11) + (nA >= 0 && !(cx.info[nA] & 4)) + (nB >= 0 && !(cx.info[nB] & 4))); }
    int ae_intra_mb_type(int base, bool in_i) {
        if (in_i) {
            int inc = (nA >= 0 && !(cx.info[nA] & 2)) + (nB >= 0 && !(cx.info[nB] & 2));       // neighbour not I_NxN
            if (!cb->decision(base + inc)) return 0;
            base += 2;
        } else if (!cb->decision(base)) return 0;
        if (cb->terminate()) return 25;
        int k = in_i ? 1 : 0, t = 1;
        t += 12 * cb->decision(base + 1);
        if (cb->decision(base + 2)) t += 4 + 4 * cb->decision(base + 2 + k);
        t += 2 * cb->decision(base + 3 + k);
        t += cb->decision(base + 3 + 2 * k);
        return t;
    }
    int ae_mb_type() {
        if (sh.type == SL_I) return ae_intra_mb_type(3, true);
        if (sh.type == SL_P) {
            if (!cb->decision(14)) {
                if (!cb->decision(15)) return 3 * cb->decision(16);
                return 2 - cb->decision(17);
            }
            return 5 + ae_intra_mb_type(17, false);
        }
        // B slice (Table 9-37 b): neighbours coded as B_Skip / B_Direct_16x16 count 0
        int inc = (nA >= 0 && !(cx.direct8[nA] & 16)) + (nB >= 0 && !(cx.direct8[nB] & 16));
        if (!cb->decision(27 + inc)) return 0;
        if (!cb->decision(27 + 3)) return 1 + cb->decision(27 + 5);
        int bits = cb->decision(27 + 4) << 3;
        bits |= cb->decision(27 + 5) << 2; bits |= cb->decision(27 + 5) << 1; bits |= cb->decision(27 + 5);
        if (bits < 8) return bits + 3;
        if (bits == 13) return 23 + ae_intra_mb_type(32, false);
        if (bits == 14) return 11;
        if (bits == 15) return 22;
        bits = (bits << 1) | cb->decision(27 + 5);
        return bits - 4;
    }
    int ae_sub_mb_type() {
        if (sh.type == SL_P) {
            if (cb->decision(21)) return 0;
            if (!cb->decision(22)) return 1;
            return cb->decision(23) ? 2 : 3;
        }
        if (!cb->decision(36)) return 0;
        if (!cb->decision(37)) return 1 + cb->decision(39);
        int t = 3;
        if (cb->decision(38)) { if (cb->decision(39)) return 11 + cb->decision(39); t += 4; }
        t += 2 * cb->decision(39);
        t += cb->decision(39);
        return t;
    }
    int ae_t8x8() { return cb->decision(399 + (nA >= 0 && (cx.info[nA] & 8)) + (nB >= 0 && (cx.info[nB] & 8))); }
    int ae_intra_mode(int pred) {
        if (cb->decision(68)) return pred;
        CabacRegs r(*cb);
        int rem = r.decision(69); rem |= r.decision(69) << 1; rem |= r.decision(69) << 2;
        r.commit();
        return rem < pred ? rem : rem + 1;
    }
    int ae_chroma_mode() {
        int inc = (nA >= 0 && cx.cmode[nA] != 0) + (nB >= 0 && cx.cmode[nB] != 0);     // cmode stays 0 for inter / I_PCM macroblocks
        if (!cb->decision(64 + inc)) return 0;
        if (!cb->decision(64 + 3)) return 1;
        return 2 + cb->decision(64 + 3);
    }
    int ae_ref_idx(int bx, int by, int l = 0) {
        int inc = 0;
        for (int k = 0; k < 2; k++) {
            int r, m = nb4(bx, by, k == 0, r);
            if (m < 0 || (cx.info[m] & 1)) continue;
            int q = (r >> 3) * 2 + ((r & 3) >> 1);
            if (sh.type == SL_B && ((cx.direct8[m] >> q) & 1)) continue;      // direct-predicted: refIdx was not parsed
            if ((l ? cx.refidx1 : cx.refidx)[(size_t)m * 4 + q] > 0) inc += k == 0 ? 1 : 2;
        }
        int v = 0, ctx = 54 + inc;
        while (cb->decision(ctx)) { v++; ctx = 54 + (v == 1 ? 4 : 5); if (v > 32) { err = "ref_idx out of range"; return 0; } }
        return v;
    }
    int ae_mvd(int bx, int by, int comp, int l = 0) {
        int sum = 0;
        for (int k = 0; k < 2; k++) { int r, m = nb4(bx, by, k == 0, r); if (m >= 0) sum += (l ? cx.mvd1 : cx.mvd)[(size_t)m * 32 + r * 2 + comp]; }
        int base = comp ? 47 : 40;
        if (!cb->decision(base + (sum < 3 ? 0 : (sum > 32 ? 2 : 1)))) return 0;
        CabacRegs r(*cb);                                       // the rest of the prefix, the suffix and the sign: three bins at least
        int v = 1, ctx = 3;
        while (v < 9 && r.decision(base + ctx)) { v++; if (ctx < 6) ctx++; }
        if (v == 9) {
            int k = 3;
            while (r.bypass()) { v += 1 << k; k++; if (k > 24) { err = "mvd out of range"; r.commit(); return 0; } }
            while (k--) v += r.bypass() << k;
        }
        const int neg = r.bypass();
        r.commit();
        return neg ? -v : v;
    }
    void read_mvd(int bx, int by, int bw, int bh, int &dx, int &dy, int l = 0) {
        if (!cb) { dx = br.se(); dy = br.se(); return; }
        uint8_t *mvd = mvdl[l];
        dx = ae_mvd(bx, by, 0, l); dy = ae_mvd(bx, by, 1, l);
        int ax = dx < 0 ? -dx : dx, ay = dy < 0 ? -dy : dy;
        uint8_t cx8 = (uint8_t)(ax > 255 ? 255 : ax), cy8 = (uint8_t)(ay > 255 ? 255 : ay);     // only sums <3 / >32 matter
        for (int j = by; j < by + bh; j++) for (int i = bx; i < bx + bw; i++) { mvd[(j * 4 + i) * 2] = cx8; mvd[(j * 4 + i) * 2 + 1] = cy8; }
    }
    int ae_cbp() {
        int cbp = 0;
        int ca4 = nA >= 0 ? cx.cbp[nA] : -1, cb4 = nB >= 0 ? cx.cbp[nB] : -1;
        CabacRegs r(*cb);                                       // five to six bins, each context chosen by the bins before it
        for (int b8 = 0; b8 < 4; b8++) {
            int a = (b8 & 1) ? !((cbp >> (b8 - 1)) & 1) : (ca4 >= 0 ? !((ca4 >> (b8 + 1)) & 1) : 0);
            int b = (b8 & 2) ? !((cbp >> (b8 - 2)) & 1) : (cb4 >= 0 ? !((cb4 >> (b8 + 2)) & 1) : 0);
            cbp |= r.decision(73 + a + 2 * b) << b8;
        }
        int a = ca4 >= 0 && (ca4 >> 4) != 0, b = cb4 >= 0 && (cb4 >> 4) != 0;
        if (r.decision(77 + a + 2 * b)) {
            a = ca4 >= 0 && (ca4 >> 4) == 2; b = cb4 >= 0 && (cb4 >> 4) == 2;
            cbp |= (1 + r.decision(77 + 4 + a + 2 * b)) << 4;
        }
        r.commit();
        return cbp;
    }
    int ae_qp_delta() {
        int ctx = 60 + (last_dqp ? 1 : 0), k = 0;
        while (cb->decision(ctx)) { k++; ctx = 60 + (k == 1 ? 2 : 3); if (k > 104) { err = "mb_qp_delta out of range"; return 0; } }
        return (k & 1) ? (k + 1) >> 1 : -(k >> 1);
    }
    // residual_block_cabac (7.3.5.3.3): levels written at dst[map[scan_pos + first]]; returns number of non-zero levels.
    // cat = ctxBlockCat, bit = this block's bit in cbf (or -1 for cat 5), fa / fb = coded_block_flag of the neighbouring
    // blocks (-1: neighbour macroblock not available)
    int residual_block_ae(int cat, int bit, int fa, int fb, int maxnum, int first, int16_t *dst, const uint8_t *map, bool intra) {
        static const int cbf_off[5] = {0, 4, 8, 12, 16}, sig_off[5] = {0, 15, 29, 44, 47}, abs_off[5] = {0, 10, 20, 30, 39};
        // the arithmetic decoder's variables in registers for the whole block (h264_cabac.h, CabacRegs): a block is dozens of bins in a row
        CabacRegs r(*cb);
        if (cat != 5) {
            if (fa < 0) fa = intra; if (fb < 0) fb = intra;
            if (!r.decision(85 + cbf_off[cat] + fa + 2 * fb)) { r.commit(); return 0; }
            cx.cbf[addr] |= 1u << bit;
        }
        if (cat == 5 && fld_ctx) { err = "CABAC residual of a field-coded 8x8 block is not supported"; r.commit(); return -1; }
        const int sig_base = cat == 5 ? 402 : 105 + fld_ctx + sig_off[cat], last_base = cat == 5 ? 417 : 166 + fld_ctx + sig_off[cat];
        const int abs_base = cat == 5 ? 426 : 227 + abs_off[cat];
        uint8_t pos[64]; int n = 0, i;
        if (cat == 5) {
            for (i = 0; i < 63; i++) if (r.decision(402 + cabac_sig8_inc[i])) { pos[n++] = (uint8_t)i; if (r.decision(417 + cabac_last8_inc[i])) break; }
        } else if (cat == 3) {
            for (i = 0; i < maxnum - 1; i++) { const int si = i < 2 ? i : 2; if (r.decision(sig_base + si)) { pos[n++] = (uint8_t)i;
                if (r.decision(last_base + si)) break; } }
        } else {
            for (i = 0; i < maxnum - 1; i++) if (r.decision(sig_base + i)) { pos[n++] = (uint8_t)i; if (r.decision(last_base + i)) break; }
        }
        if (i == maxnum - 1) pos[n++] = (uint8_t)(maxnum - 1);
        int gt1 = 0, eq1 = 0;
        const int gt1_max = 4 - (cat == 3);
        for (int k = n - 1; k >= 0; k--) {
            int v = 0;
            if (r.decision(abs_base + (gt1 ? 0 : (1 + eq1 < 4 ? 1 + eq1 : 4)))) {
                v = 1;
                int ctx = abs_base + 5 + (gt1 < gt1_max ? gt1 : gt1_max);
                while (v < 14 && r.decision(ctx)) v++;
                if (v == 14) {
                    int e = 0;
                    // (the slice is abandoned: nothing to write back)
                    while (r.bypass()) { v += 1 << e; e++; if (e > 20) { err = "coefficient level out of range"; return -1; } }
                    while (e--) v += r.bypass() << e;
                }
            }
            int level = v + 1;
            if (level == 1) eq1++; else gt1++;
            if (level > 32767) level = 32767;
            dst[map[pos[k] + first]] = (int16_t)(r.bypass() ? -level : level);
        }
        r.commit();
        return n;
    }
    // coded_block_flag of the 4x4 luma block left of / above (bx,by): -1 when the neighbouring macroblock is not available
    int cbf_luma_nb(int bx, int by, bool left) const {
        int r, m = nb4(bx, by, left, r);
        if (m < 0) return -1;
        return (int)((cx.cbf[m] >> r) & 1);
    }

    // residual(): luma + chroma blocks into the coefficient stream
    bool residual(MbRec *r, int cbp, bool i16, bool t8, bool intra) {
        static const uint8_t ident[16] = {0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15};
        if (i16) {
            int16_t *d = alloc_coef(16); if (!d) return false;
            int n;
            if (cb) n = residual_block_ae(0, 16, nA >= 0 ? (int)((cx.cbf[nA] >> 16) & 1) : -1, nB >= 0 ? (int)((cx.cbf[nB] >> 16) & 1) : -1, 16, 0, d,
                zz4, intra);
            else n = residual_block(nc_luma(0, 0), 16, 0, d, zz4);
            if (n < 0) { if (!err) err = "entropy error (Intra16x16 DC)"; return false; }
            if (canon) memcpy(canon->i16dc, d, 32);
        }
        uint32_t bits = 0;
        if (t8) {
            for (int b8 = 0; b8 < 4; b8++) {
                if (!(cbp & (1 << b8))) continue;
                if (out.coef_count + 64 > out.coef_cap) { err = "coefficient buffer overflow"; return false; }
                int16_t *d = out.coef + out.coef_count;
                memset(d, 0, 128);
                int ox = (b8 & 1) * 2, oy = (b8 >> 1) * 2, total = 0;
                if (cb) {
                    total = residual_block_ae(5, -1, 0, 0, 64, 0, d, zz8, intra);
                    if (total < 0) return false;
                    // 7.4.5.3.3: coded_block_flag of an 8x8 luma block is inferred to be 1 (4:2:0)
                    for (int k = 0; k < 4; k++) cx.cbf[addr] |= 1u << ((oy + (k >> 1)) * 4 + ox + (k & 1));
                } else {
                    for (int k = 0; k < 4; k++) {                       // 7.3.5.3.2: block k carries every fourth level of the 8x8 scan
                        int bx = ox + (k & 1), by = oy + (k >> 1);
                        int16_t tmp[16]; memset(tmp, 0, sizeof tmp);
                        int n = residual_block(nc_luma(bx, by), 16, 0, tmp, ident);
                        if (n < 0) { err = "CAVLC error (luma 8x8 block)"; return false; }
                        tc[by * 4 + bx] = (uint8_t)n; total += n;
                        for (int i = 0; i < 16; i++) if (tmp[i]) d[zz8[4 * i + k]] = tmp[i];
                    }
                }
                if (total) { bits |= 15u << (4 * b8); out.coef_count += 64; if (canon) memcpy(&canon->luma[0][0] + 64 * b8, d, 128); }
            }
        } else {
            for (int blk = 0; blk < 16; blk++) {
                if (!(cbp & (1 << (blk >> 2)))) continue;
                int bx = (blk & 1) + 2 * ((blk >> 2) & 1), by = ((blk >> 1) & 1) + 2 * (blk >> 3);
                // parse into the stream tail; keep the 16 slots only if the block turns out non-empty
                if (out.coef_count + 16 > out.coef_cap) { err = "coefficient buffer overflow"; return false; }
                int16_t *d = out.coef + out.coef_count;
                int n;
                if (cb) memset(d, 0, 32);
                if (cb) n = residual_block_ae(i16 ? 1 : 2, by * 4 + bx, cbf_luma_nb(bx, by, true), cbf_luma_nb(bx, by, false), i16 ? 15 : 16, i16 ? 1 : 0, d,
                    zz4, intra);
                else n = i16 ? residual_block(nc_luma(bx, by), 15, 1, d, zz4) : residual_block(nc_luma(bx, by), 16, 0, d, zz4);
                if (n < 0) { if (!err) err = "entropy error (luma block)"; return false; }
                tc[by * 4 + bx] = (uint8_t)n;
                if (n) { bits |= 1u << blk; out.coef_count += 16; if (canon) memcpy(canon->luma[by * 4 + bx], d, 32); }
            }
        }
        r->cbp_blk = (uint16_t)bits;
        if (cbp & 0x30) {
            for (int pl = 0; pl < 2; pl++) {
                if (out.coef_count + 4 > out.coef_cap) { err = "coefficient buffer overflow"; return false; }
                int16_t *d = out.coef + out.coef_count;
                int n;
                if (cb) memset(d, 0, 8);
                if (cb) n = residual_block_ae(3, 17 + pl, nA >= 0 ? (int)((cx.cbf[nA] >> (17 + pl)) & 1) : -1,
                    nB >= 0 ? (int)((cx.cbf[nB] >> (17 + pl)) & 1) : -1, 4, 0, d, ident, intra);
                else n = residual_block(-1, 4, 0, d, ident);
                if (n < 0) { if (!err) err = "entropy error (chroma DC)"; return false; }
                if (n) { r->flags |= pl ? MBF_CR_DC : MBF_CB_DC; out.coef_count += 4; if (canon) memcpy(canon->cdc[pl], d, 8); }
            }
        }
        if (cbp & 0x20) {
            uint32_t cbm = 0;
            for (int pl = 0; pl < 2; pl++) for (int k = 0; k < 4; k++) {
                if (out.coef_count + 16 > out.coef_cap) { err = "coefficient buffer overflow"; return false; }
                int16_t *d = out.coef + out.coef_count;
                int n;
                if (cb) {
                    memset(d, 0, 32);
                    int bx = k & 1, by = k >> 1, b0 = 19 + pl * 4, fa, fb;
                    if (bx) fa = (int)((cx.cbf[addr] >> (b0 + by * 2)) & 1); else fa = nA >= 0 ? (int)((cx.cbf[nA] >> (b0 + by * 2 + 1)) & 1) : -1;
                    if (by) fb = (int)((cx.cbf[addr] >> (b0 + bx)) & 1); else fb = nB >= 0 ? (int)((cx.cbf[nB] >> (b0 + 2 + bx)) & 1) : -1;
                    n = residual_block_ae(4, b0 + k, fa, fb, 15, 1, d, zz4, intra);
                } else n = residual_block(nc_chroma(pl, k & 1, k >> 1), 15, 1, d, zz4);
                if (n < 0) { if (!err) err = "entropy error (chroma AC)"; return false; }
                tc[16 + 4 * pl + k] = (uint8_t)n;
                if (n) { cbm |= 1u << (4 * pl + k); out.coef_count += 16; if (canon) memcpy(canon->cac[pl][k], d, 32); }
            }
            r->cbp_cac = (uint8_t)cbm;
        }
        return true;
    }

    // 8.3.1.1 / 8.3.2.1: predicted Intra4x4 / Intra8x8 mode of the block whose top-left 4x4 is (bx,by)
    int pred_intra_mode(int bx, int by) const {
        int mA = bx > 0 ? addr : nA, mB = by > 0 ? addr : nB;
        if (mA < 0 || mB < 0) return 2;
        if (pps.constrained_intra && (!(cx.info[mA] & 1) || !(cx.info[mB] & 1))) return 2;
        int a = bx > 0 ? i4m[by * 4 + bx - 1] : ((cx.info[mA] & 2) ? cx.i4[(size_t)mA * 16 + by * 4 + 3] : 2);
        int b = by > 0 ? i4m[(by - 1) * 4 + bx] : ((cx.info[mB] & 2) ? cx.i4[(size_t)mB * 16 + 12 + bx] : 2);
        return a < b ? a : b;
    }

    bool macroblock(int &n_intra, int &n_i8x8, int type_read = -1) {            // type_read: mb_type when the caller has read it already (CAVLC)
        MbRec *r = begin_mb();
        uint32_t mb_type = type_read >= 0 ? (uint32_t)type_read : (cb ? (uint32_t)ae_mb_type() : br.ue());
        int itype = -1;
        if (sh.type == SL_I) itype = (int)mb_type;
        else if (sh.type == SL_P) { if (mb_type >= 5) itype = (int)mb_type - 5; }
        else if (mb_type >= 23) itype = (int)mb_type - 23;
        if (mb_type > 48 || itype > 25) { err = "bad mb_type"; return false; }

        if (itype == 25) {                                     // I_PCM
            r->kind = MB_PCM; r->qp = 0;
            int16_t *d = alloc_coef(192); if (!d) return false;
            if (cb) {
                // the 9 bits the arithmetic decoder has read ahead end exactly with the last bit of the encoder's flush (9.3.4.5), so only
                // pcm_alignment_zero_bits remain before the samples; the engine restarts behind them (9.3.1.2)
                const uint8_t *pcm = cb->start + ((cb->bits_consumed() + 7) >> 3);
                if (cb->overrun || pcm + 384 > cb->end) { err = "I_PCM runs past the slice"; return false; }
                memcpy(d, pcm, 384);
                cx.cbp[addr] = 0x2f; cx.cbf[addr] = 0x7FFFFFF; last_dqp = false;
                cb->init_engine(pcm + 384, cb->end);
            } else {
                br.align_zero();
                if (br.overrun() || (br.bitpos() >> 3) + 384 > br.size()) { err = "I_PCM runs past the slice"; return false; }
                memcpy(d, br.byte_ptr(), 384); br.skip_bytes(384);
            }
            memset(tc, 16, 24); cx.info[addr] = 1 | 16;
            finish_mb(r);
            return true;
        }
        int cbp = 0; bool i16 = false, t8 = false;
        if (itype >= 0) {
            cx.info[addr] = 1;
            if (intra_usable(nA)) r->flags |= MBF_AVAIL_A;
            if (intra_usable(nB)) r->flags |= MBF_AVAIL_B;
            if (intra_usable(nC)) r->flags |= MBF_AVAIL_C;
            if (intra_usable(nD)) r->flags |= MBF_AVAIL_D;
            n_intra++;
            if (itype == 0) {
                r->kind = MB_I4; cx.info[addr] = 3;
                if (pps.transform8x8) t8 = cb ? ae_t8x8() : br.u1();
                if (t8) {
                    cx.info[addr] |= 8; n_i8x8++;
                    for (int b8 = 0; b8 < 4; b8++) {
                        int bx = (b8 & 1) * 2, by = (b8 >> 1) * 2, pred = pred_intra_mode(bx, by), mode;
                        if (cb) mode = ae_intra_mode(pred);
                        else if (br.u1()) mode = pred; else { int rem = (int)br.u(3); mode = rem < pred ? rem : rem + 1; }
                        i4m[by * 4 + bx] = i4m[by * 4 + bx + 1] = i4m[by * 4 + bx + 4] = i4m[by * 4 + bx + 5] = (uint8_t)mode;
                    }
                    // Intra8x8PredMode of block b8 in nibble b8
                    r->u.i4[0] = (uint8_t)(i4m[0] | (i4m[2] << 4)); r->u.i4[1] = (uint8_t)(i4m[8] | (i4m[10] << 4));
                } else {
                    if (cb) {
                        for (int blk = 0; blk < 16; blk++) {
                            int bx = (blk & 1) + 2 * ((blk >> 2) & 1), by = ((blk >> 1) & 1) + 2 * (blk >> 3);
                            i4m[by * 4 + bx] = (uint8_t)ae_intra_mode(pred_intra_mode(bx, by));
                        }
                    } else {
                        BitReader b = br;                          // (a local copy lives in registers, see parse_slice_data)
                        for (int blk = 0; blk < 16; blk++) {
                            int bx = (blk & 1) + 2 * ((blk >> 2) & 1), by = ((blk >> 1) & 1) + 2 * (blk >> 3);
                            int pred = pred_intra_mode(bx, by), mode;
                            const uint32_t v = b.peek(4);              // prev_intra4x4_pred_mode_flag, then rem_intra4x4_pred_mode (3 bits) when it is 0
                            if (v & 8) { mode = pred; b.skip(1); } else { const int rem = (int)(v & 7); mode = rem < pred ? rem : rem + 1; b.skip(4); }
                            i4m[by * 4 + bx] = (uint8_t)mode;
                        }
                        br = b;
                    }
                    for (int k = 0; k < 8; k++) r->u.i4[k] = (uint8_t)(i4m[2 * k] | (i4m[2 * k + 1] << 4));
                }
                if (canon) memcpy(canon->i4, i4m, 16);
            } else {
                r->kind = MB_I16; i16 = true;
                int k = itype - 1;
                r->modes = (uint8_t)((k & 3) << 2);
                cbp = (((k >> 2) % 3) << 4) | (k >= 12 ? 15 : 0);
                if (canon) canon->i16mode = (uint8_t)(k & 3);
            }
            uint32_t cm = cb ? (uint32_t)ae_chroma_mode() : br.ue();
            if (cm > 3) { err = "bad intra_chroma_pred_mode"; return false; }
            r->modes |= (uint8_t)cm;
            if (cb) cx.cmode[addr] = (uint8_t)cm;
            if (canon) canon->cmode = (uint8_t)cm;
        } else {
            r->kind = MB_INTER;
            static const uint8_t b_pair[9][2] = {{0, 0}, {1, 1}, {0, 1}, {1, 0}, {0, 2}, {1, 2}, {2, 0}, {2, 1}, {2, 2}};   // Table 7-14: 0 L0, 1 L1, 2 Bi
            static const uint8_t b_sub_pred[13] = {3, 0, 1, 2, 0, 0, 1, 1, 2, 2, 0, 1, 2};                                   // Table 7-18: 3 = direct
            static const uint8_t b_sub_shape[13] = {0, 0, 0, 0, 1, 2, 1, 2, 1, 2, 3, 3, 3};                                  // 0 8x8, 1 8x4, 2 4x8, 3 4x4
            const bool is_b = sh.type == SL_B;
            const int nl = is_b ? 2 : 1;
            bool sub8 = false;                              // a partition smaller than 8x8 (or direct without direct_8x8_inference): no 8x8 transform
            int shape, pred[4] = {0, 0, 0, 0};
            if (!is_b) shape = mb_type > 3 ? 3 : (int)mb_type;
            else if (mb_type == 0) shape = 4;
            else if (mb_type <= 3) { shape = 0; pred[0] = (int)mb_type - 1; }
            else if (mb_type == 22) shape = 3;
            else { shape = (mb_type & 1) ? 2 : 1; pred[0] = b_pair[(mb_type - 4) >> 1][0]; pred[1] = b_pair[(mb_type - 4) >> 1][1]; }
            if (shape == 4) {                               // B_Direct_16x16
                cx.direct8[addr] = 16;
                if (!direct_pred(15)) return false;
                if (!sps.direct_8x8_inference) sub8 = true;
            } else if (shape <= 2) {
                int np = shape == 0 ? 1 : 2, rf_[2][2] = {{-1, -1}, {-1, -1}};
                for (int l = 0; l < nl; l++) {
                    int nref = sh.num_ref_idx[l];
                    for (int p = 0; p < np; p++) {
                        if (!(pred[p] == 2 || pred[p] == l)) continue;
                        int bx = shape == 2 ? p * 2 : 0, by = shape == 1 ? p * 2 : 0, bw = shape == 2 ? 2 : 4, bh = shape == 1 ? 2 : 4;
                        rf_[l][p] = 0;
                        if (nref > 1) { rf_[l][p] = cb ? ae_ref_idx(bx, by, l) : br.te(nref - 1); if (rf_[l][p] >= nref) { err = "ref_idx out of range";
                            return false; } }
                        for (int j = by; j < by + bh; j += 2) for (int i = bx; i < bx + bw; i += 2) refl[l][(j >> 1) * 2 + (i >> 1)] = (int8_t)rf_[l][p];
                    }
                }
                for (int l = 0; l < nl; l++) {
                    decoded_mask = 0;
                    for (int p = 0; p < np; p++) {
                        int bx = shape == 2 ? p * 2 : 0, by = shape == 1 ? p * 2 : 0, bw = shape == 2 ? 2 : 4, bh = shape == 1 ? 2 : 4;
                        if (rf_[l][p] < 0) { for (int j = by; j < by + bh; j++) for (int i = bx; i < bx + bw; i++) decoded_mask |= 1u << (j * 4 + i); continue;
                            }
                        int px, py, dx, dy; predict(bx, by, bw, rf_[l][p], shape, p, px, py, l);
                        read_mvd(bx, by, bw, bh, dx, dy, l);
                        set_mv(bx, by, bw, bh, px + dx, py + dy, l);
                    }
                }
            } else {
                int sub[4], sshape[4], rf_[2][4] = {{-1, -1, -1, -1}, {-1, -1, -1, -1}}, dmask = 0;
                for (int i = 0; i < 4; i++) {
                    sub[i] = cb ? ae_sub_mb_type() : (int)br.ue();
                    if (sub[i] > (is_b ? 12 : 3)) { err = "bad sub_mb_type"; return false; }
                    if (is_b) { pred[i] = b_sub_pred[sub[i]]; sshape[i] = b_sub_shape[sub[i]]; if (pred[i] == 3) dmask |= 1 << i; }
                    else { pred[i] = 0; sshape[i] = sub[i]; }
                    if (sshape[i] != 0 || (pred[i] == 3 && !sps.direct_8x8_inference)) sub8 = true;
                }
                if (dmask && !direct_pred(dmask)) return false;     // from neighbouring macroblocks / the colocated picture only
                for (int l = 0; l < nl; l++) {
                    int nref = sh.num_ref_idx[l];
                    for (int i = 0; i < 4; i++) {
                        if (!(pred[i] == 2 || pred[i] == l)) continue;
                        rf_[l][i] = 0;
                        if (nref > 1 && !(!is_b && mb_type == 4)) { rf_[l][i] = cb ? ae_ref_idx((i & 1) * 2, (i >> 1) * 2, l) : br.te(nref - 1);
                            if (rf_[l][i] >= nref) { err = "ref_idx out of range"; return false; } }
                        refl[l][i] = (int8_t)rf_[l][i];
                    }
                }
                for (int l = 0; l < nl; l++) {
                    decoded_mask = 0;
                    for (int i = 0; i < 4; i++) {
                        int ox = (i & 1) * 2, oy = (i >> 1) * 2, st = sshape[i];
                        if (rf_[l][i] < 0) { mark8(i); continue; }       // direct, or this list unused
                        int nsp = st == 0 ? 1 : (st == 3 ? 4 : 2), bw = (st == 0 || st == 1) ? 2 : 1, bh = (st == 0 || st == 2) ? 2 : 1;
                        for (int p = 0; p < nsp; p++) {
                            int bx = ox + (st == 1 ? 0 : (st == 2 ? p : (p & 1))), by = oy + (st == 1 ? p : (st == 2 ? 0 : (p >> 1)));
                            int px, py, dx, dy; predict(bx, by, bw, rf_[l][i], 0, 0, px, py, l);
                            read_mvd(bx, by, bw, bh, dx, dy, l);
                            set_mv(bx, by, bw, bh, px + dx, py + dy, l);
                        }
                    }
                }
            }
            if (err) return false;
            write_motion(r, sub8);
            if (err) return false;
            if (!i16) {
                if (cb) cbp = ae_cbp();
                else { uint32_t code = br.ue(); if (code > 47) { err = "bad coded_block_pattern"; return false; } cbp = kCbpInter[code]; }
                if ((cbp & 15) && pps.transform8x8 && !sub8) { t8 = cb ? ae_t8x8() : br.u1(); if (t8) cx.info[addr] |= 8; }
            }
        }
        if (itype >= 0 && !i16) {
            if (cb) cbp = ae_cbp();
            else { uint32_t code = br.ue(); if (code > 47) { err = "bad coded_block_pattern"; return false; } cbp = kCbpIntra[code]; }
        }
        if (cb) cx.cbp[addr] = (uint8_t)cbp;
        if (t8) r->modes |= MBM_T8X8;
        if (cbp > 0 || i16) {
            int dqp = cb ? ae_qp_delta() : br.se();
            if (dqp < -26 || dqp > 25) { err = "mb_qp_delta out of range"; return false; }
            qp = (qp + dqp + 52) % 52;
            last_dqp = dqp != 0;
        } else last_dqp = false;
        r->qp = (uint8_t)qp;
        if ((cbp > 0 || i16) && !cb && !t8 && !canon) {
            BitReader b = br;                                  // (a local copy lives in registers, see parse_slice_data)
            uint32_t fl, bits, cbm;
            const bool ok = i16 ? residual_fast(b, cbp, true, fl, bits, cbm) : residual_fast(b, cbp, false, fl, bits, cbm);
            br = b;
            if (!ok) return false;
            r->flags |= (uint8_t)fl; r->cbp_blk = (uint16_t)bits; r->cbp_cac = (uint8_t)cbm;
        } else if (cbp > 0 || i16) { if (!residual(r, cbp, i16, t8, itype >= 0)) return false; }
        if (err) return false;
        if (cb ? cb->overrun : br.overrun()) { err = "macroblock data truncated"; return false; }
        if (canon && t8) r->kind |= 16;                        // digest only: restored below
        finish_mb(r);
        r->kind &= 15;
        return true;
    }
};

}  // namespace

SliceParseResult parse_slice_data(const SeqParams &sps, const PicParamSet &pps, const SliceHeader &sh,
                                  BitReader &br, int slice_num, const SliceRefs &refs,
                                  ParseScratch &cx, JobWriter &out, SyntaxDigest *digest, bool allow_fast) {
    cavlc_init_tables();
    SliceParseResult res;
    Canon canon;
    P p{sps, pps, sh, br, cx, out, digest, slice_num, refs, sps.mb_w};
    p.qp = sh.qp;
    p.canon = digest ? &canon : nullptr;
    if (sh.field_pic) { p.zz4 = kFieldScan.s4; p.zz8 = kFieldScan.s8; p.fld_ctx = 277 - 105; allow_fast = false; }
    int n_mbs = sps.mb_w * sps.mb_h, addr = sh.first_mb;
    if (pps.cabac) {
        static thread_local Cabac cabac;
        // 7.3.4: cabac_alignment_one_bit up to the byte boundary, then the arithmetic code (9.3.1.2)
        while (!br.aligned()) if (!br.u1()) { res.error = "cabac_alignment_one_bit is not 1"; return res; }
        if (sh.cabac_init_idc > 2) { res.error = "bad cabac_init_idc"; return res; }
        cabac.init_contexts(sh.type == SL_I ? 0 : 1 + sh.cabac_init_idc, sh.qp);
        cabac.init_engine(br.byte_ptr(), br.base() + br.size());
        p.cb = &cabac;
        const bool fast_ae = allow_fast && sh.type == SL_P && !digest && !refs.bipred_rec;
        for (;;) {
            if (addr >= n_mbs) { res.error = "slice runs past the end of the picture"; return res; }
            bool ok;
            if (fast_ae) {                                    // P_Skip / P_L0_16x16 without the partition machinery (see skip_mb_fast_ae)
                p.locate_next(addr, sh.first_mb);
                if (p.ae_mb_skip()) { p.skip_mb_fast_ae(); ok = true; }
                else {
                    const int mb_type = p.ae_mb_type();
                    ok = mb_type == 0 ? p.p16x16_fast_ae() : p.macroblock(res.n_intra, res.n_i8x8, mb_type);
                }
            } else {
                p.locate(addr);
                if (sh.type != SL_I && p.ae_mb_skip()) ok = p.skip_mb();
                else ok = p.macroblock(res.n_intra, res.n_i8x8);
            }
            if (!ok) { res.error = p.err ? p.err : "macroblock error"; return res; }
            addr++; res.mbs_decoded++;
            if (cabac.overrun || cabac.bits_consumed() > (size_t)(cabac.end - cabac.start) * 8 + 16) { res.error = "slice data truncated"; return res; }
            if (cabac.terminate()) break;                     // end_of_slice_flag
        }
        return res;
    }
    if (allow_fast && sh.type == SL_P && !digest && !pps.transform8x8 && !refs.bipred_rec) {
        // P_Skip / P_L0_16x16 through the fast path.  The reader is copied into a local whose address never leaves this loop (everything that
        // takes it is inlined), so its state lives in registers instead of being reloaded after every byte store into the neighbour arrays.
        BitReader b = br;
        bool more = true;
        while (more) {
            const uint32_t run = b.ue();
            if (b.overrun() || run > (uint32_t)(n_mbs - addr)) { res.error = "bad mb_skip_run"; break; }
            for (uint32_t i = 0; i < run; i++) { p.locate_next(addr, sh.first_mb); p.skip_mb_fast(); addr++; }
            res.mbs_decoded += (int)run;
            if (run > 0) more = b.more_rbsp_data();
            if (!more) break;
            if (addr >= n_mbs) { res.error = "slice runs past the end of the picture"; break; }
            p.locate_next(addr, sh.first_mb);
            const uint32_t mb_type = b.ue();
            bool ok;
            if (mb_type == 0) ok = p.p16x16_fast(b);
            else { br = b; ok = p.macroblock(res.n_intra, res.n_i8x8, (int)(mb_type > 255 ? 255 : mb_type)); b = br; }
            if (!ok) { res.error = p.err ? p.err : "macroblock error"; break; }
            addr++; res.mbs_decoded++;
            more = b.more_rbsp_data();
        }
        br = b;
        return res;
    }
    bool more = true;
    while (more) {
        if (sh.type != SL_I) {
            uint32_t run = br.ue();
            if (br.overrun() || run > (uint32_t)(n_mbs - addr)) { res.error = "bad mb_skip_run"; return res; }
            for (uint32_t i = 0; i < run; i++) { p.locate(addr); if (!p.skip_mb()) { res.error = p.err; return res; } addr++; res.mbs_decoded++; }
            if (run > 0) more = br.more_rbsp_data();
            if (!more) break;
        }
        if (addr >= n_mbs) { res.error = "slice runs past the end of the picture"; return res; }
        p.locate(addr);
        if (!p.macroblock(res.n_intra, res.n_i8x8)) { res.error = p.err ? p.err : "macroblock error"; return res; }
        addr++; res.mbs_decoded++;
        more = br.more_rbsp_data();
    }
    return res;
}

}  // namespace jmamd
