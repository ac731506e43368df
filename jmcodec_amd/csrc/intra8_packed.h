// jmcodec_amd/csrc/intra8_packed.h -- Intra8x8 prediction (H.264 8.3.2.2) of one 8x8 block on packed bytes, in registers.
//
// Part of the replacement for cuvidDecodePicture (/root/reference/nv_dec/nv_dec.cpp:33-41): k_intra_band / k_chain_i (intra_device.h).  Rounds 1-4 built the
// 25-sample reference path of a block in LDS -- two conditional byte reads per lane, a write, three reads for the [1 2 1] filter of 8.3.2.2.1, a write, then
// three byte reads per predicted sample: five dependent LDS round trips per block, four blocks per macroblock, and an all-intra High picture took 2.1x the time
// of a Baseline one (profiles/r05_intra8_probe.txt).  Here every lane reads the block's surroundings with thirteen unconditional dword loads (ONE wait), keeps
// the whole path in seven registers, filters it with two v_lerp_u8 per four samples and takes the taps of its four samples out with v_perm_b32 whose selectors
// come from a table by (mode, lane) -- the construction Intra4x4 got in round 4.  Host-testable like mc_packed.h: tests/native/intra8_packed_check.cpp
// compares every mode, position and availability pattern with a literal restatement of 8.3.2.2.
//
// Path index k: 0..7 = p[-1, 7] .. p[-1, 0] (left column, bottom to top), 8 = p[-1, -1], 9..24 = p[0, -1] .. p[15, -1]; register F[j] holds k = 4j .. 4j + 3.
// Samples of a neighbour that is not available count as 128 (as in k_recon_intra and the oracle; a conforming stream never selects a mode that reads them).
#pragma once
#include "mc_packed.h"

namespace jmamd {
namespace pk {

// (c, kind) of Intra8x8 mode `mode` for sample (x, y) on the FILTERED path: kind 0 copy P[c], 1 two-tap (P[c] + P[c+1] + 1) >> 1,
// 2 three-tap (P[c-1] + 2 P[c] + P[c+1] + 2) >> 2 (indices clamped to 0..24: 8.3.2.2.4 / 8.3.2.2.10 end cases), 3 DC.  entry = c | kind << 5
JM_HD int i8_table_entry(int mode, int x, int y) {
    int c = 0, kind = 0;
    switch (mode) {
    case 0: c = 9 + x; break;
    case 1: c = 7 - y; break;
    case 2: kind = 3; break;
    case 3: c = 10 + x + y; kind = 2; break;
    case 4: c = 8 + x - y; kind = 2; break;
    case 5: { int z = 2 * x - y, i = x - (y >> 1);
        if (z >= 0) { c = 8 + i; kind = (z & 1) ? 2 : 1; }
        else if (z == -1) { c = 8; kind = 2; }
        else { c = 9 - y + 2 * x; kind = 2; }
        break; }
    case 6: { int z = 2 * y - x, i = y - (x >> 1);
        if (z >= 0) { if (z & 1) { c = 8 - i; kind = 2; } else { c = 7 - i; kind = 1; } }
        else if (z == -1) { c = 8; kind = 2; }
        else { c = 7 + x - 2 * y; kind = 2; }
        break; }
    case 7: { int i = x + (y >> 1); if (y & 1) { c = 10 + i; kind = 2; } else { c = 9 + i; kind = 1; } break; }
    default: { int z = x + 2 * y, i = y + (x >> 1);
        if (z > 13) { c = 0; kind = 0; }
        else if (z == 13) { c = 0; kind = 2; }
        else if (z & 1) { c = 6 - i; kind = 2; }
        else { c = 6 - i; kind = 1; }
        break; }
    }
    return c | (kind << 5);
}

// The table entry of (mode, lane): lane = 2 * y + (x >> 2) predicts samples (x .. x + 3, y).  s[t][p]: v_perm_b32 selector that takes tap t (P[c-1], P[c],
// P[c+1]) of the lane's four samples out of the register pair F[2p+1] : F[2p] (a selector byte 0x0c yields zero: the OR over the four pairs has sample j in
// byte j).  m1 / m2 / m3: bytes 0xff where the sample is two-tap / three-tap / DC.
struct I8Sel { uint32_t s[3][4]; uint32_t m1, m2, m3, pad; };
static_assert(sizeof(I8Sel) == 64, "I8Sel layout");
JM_HD I8Sel i8_sel_entry(int mode, int lane) {
    // (every array index below is a constant once the loops are unrolled: the entry is built in registers -- indexing s[t][p] with a computed p put it in
    // scratch memory on the device)
    I8Sel e;
    const int y = lane >> 1, x0 = (lane & 1) * 4;
    int pos[4][3]; uint32_t kinds = 0;
    for (int j = 0; j < 4; j++) {
        const int ent = i8_table_entry(mode, x0 + j, y), c = ent & 31;
        pos[j][0] = c > 0 ? c - 1 : 0; pos[j][1] = c; pos[j][2] = c < 24 ? c + 1 : 24;
        kinds |= (uint32_t)(ent >> 5) << (8 * j);
    }
    for (int t = 0; t < 3; t++) for (int p = 0; p < 4; p++) {
        uint32_t v = 0;
        for (int j = 0; j < 4; j++) v |= (uint32_t)((pos[j][t] >> 3) == p ? (pos[j][t] & 7) : 0x0c) << (8 * j);
        e.s[t][p] = v;
    }
    // kinds holds 0..3 per byte: bit 0 alone = two-tap, bit 1 alone = three-tap, both = DC
    const uint32_t b0 = kinds & kOnes, b1 = (kinds >> 1) & kOnes;
    e.m1 = (b0 & ~b1) * 0xffu; e.m2 = (b1 & ~b0) * 0xffu; e.m3 = (b0 & b1) * 0xffu; e.pad = 0;
    return e;
}

// What a lane reads of the block's surroundings, as aligned dwords of the work tile (intra_device.h: kTS / kTO): byte 3 of tl = p[-1, -1]; t0 t1 = p[0..7, -1];
// r0 r1 = p[8..15, -1]; byte 3 of l[i] = p[-1, i]
struct I8Edge { uint32_t tl, t0, t1, r0, r1, l[8]; };

// 8.3.2.2.1: the reference path, substituted for what is not available and filtered.  a / b / c / d: left / above / above right / above left available.
JM_HD void i8_filtered_path(const I8Edge &e, bool a, bool b, bool c, bool d, uint32_t *F) {
    const uint32_t k128 = 0x80808080u;
    // the raw path
    uint32_t R[7];
    R[0] = perm(e.l[7], e.l[6], 0x0c0c0307u) | perm(e.l[5], e.l[4], 0x03070c0cu);      // p[-1, 7] p[-1, 6] p[-1, 5] p[-1, 4]
    R[1] = perm(e.l[3], e.l[2], 0x0c0c0307u) | perm(e.l[1], e.l[0], 0x03070c0cu);
    if (!a) { R[0] = k128; R[1] = k128; }
    const uint32_t tl = d ? e.tl : k128, t0 = b ? e.t0 : k128, t1 = b ? e.t1 : k128;
    const uint32_t rep = perm(0u, t1, 0x03030303u);                                    // no samples above right: p[7, -1] repeated (8.3.2.2: substitution)
    const uint32_t r0 = b ? (c ? e.r0 : rep) : k128, r1 = b ? (c ? e.r1 : rep) : k128;
    R[2] = alignbyte(t0, tl, 3); R[3] = alignbyte(t1, t0, 3); R[4] = alignbyte(r0, t1, 3); R[5] = alignbyte(r1, r0, 3); R[6] = r1 >> 24;
    // neighbours along the path: LO[k] = raw[k - 1], HI[k] = raw[k + 1], with the end cases of 8.3.2.2.1 (the first and last sample, and what the corner and
    // its two neighbours use when one of them is not available)
    uint32_t LO[7], HI[7];
    LO[0] = perm(R[0], R[0], 0x02010000u);
    for (int j = 1; j < 7; j++) LO[j] = alignbyte(R[j], R[j - 1], 3);
    for (int j = 0; j < 6; j++) HI[j] = alignbyte(R[j + 1], R[j], 1);
    HI[6] = R[6];
    if (!a) LO[2] = (LO[2] & 0xffffff00u) | (R[2] & 0x000000ffu);                      // p'[-1, -1] without a left neighbour: (3 p[-1, -1] + p[0, -1] + 2) >> 2
    if (!d) LO[2] = (LO[2] & 0xffff00ffu) | (R[2] & 0x0000ff00u);                      // p'[0, -1] without a corner
    if (!d) HI[1] = (HI[1] & 0x00ffffffu) | (R[1] & 0xff000000u);                      // p'[-1, 0] without a corner
    if (!b) HI[2] = (HI[2] & 0xffffff00u) | (R[2] & 0x000000ffu);                      // p'[-1, -1] without samples above
    // (lo + 2 c + hi + 2) >> 2 == (((lo + hi) >> 1) + c + 1) >> 1 for bytes: two v_lerp_u8
    for (int j = 0; j < 7; j++) F[j] = lerp(lerp(LO[j], HI[j], 0u), R[j], kOnes);
    if (!d) F[2] = (F[2] & 0xffffff00u) | 0x80u;
}

// 8.3.2.2.4 Intra_8x8_DC on the filtered path
JM_HD int i8_dc(const uint32_t *F, bool a, bool b) {
    const int sl = (int)udot4(F[0], kOnes, udot4(F[1], kOnes, 0u));
    const int st = (int)udot4(F[2] & 0xffffff00u, kOnes, udot4(F[3], kOnes, udot4(F[4] & 0xffu, kOnes, 0u)));
    return (a && b) ? (st + sl + 8) >> 4 : (a ? (sl + 4) >> 3 : (b ? (st + 4) >> 3 : 128));
}

// the lane's four predicted samples, one byte each
JM_HD uint32_t i8_predict4(const uint32_t *F, const I8Sel &s, int dc) {
    uint32_t v[3];
    for (int t = 0; t < 3; t++) v[t] = perm(F[1], F[0], s.s[t][0]) | perm(F[3], F[2], s.s[t][1]) | perm(F[5], F[4], s.s[t][2]) | perm(0u, F[6], s.s[t][3]);
    const uint32_t two = lerp(v[1], v[2], kOnes), three = lerp(lerp(v[0], v[2], 0u), v[1], kOnes);
    uint32_t p = v[1];
    p = (s.m1 & two) | (~s.m1 & p);
    p = (s.m2 & three) | (~s.m2 & p);
    p = (s.m3 & ((uint32_t)dc * 0x01010101u)) | (~s.m3 & p);
    return p;
}

}  // namespace pk
}  // namespace jmamd
