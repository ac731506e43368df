// jmcodec_amd/csrc/h264_cabac.h -- CABAC arithmetic decoding engine and context variables (H.264 9.3.1, 9.3.3.2).
//
// Host half of the entropy stage the reference leaves to the NVDEC ASIC behind cuvidDecodePicture
// (/root/reference/nv_dec/nv_dec.cpp:33-41; entropy_coding_mode_flag travels in CUVIDH264PICPARAMS,
// nv_sdk/inc/dynlink_cuviddec.h:243-298).  Syntax-element binarisations and ctxIdxInc derivations live with the
// macroblock layer in h264_cavlc.cpp; this file is only the engine.
#pragma once
#include "bitreader.h"
#include "cabac_tables.h"

namespace jmamd {

struct Cabac {
    BitReader *br = nullptr;
    uint32_t range = 510, offset = 0;
    uint8_t state[CABAC_N_CTX];              // pStateIdx << 1 | valMPS

    // 9.3.1.1: context variables from (m, n) and SliceQPY; table 0 = I slices, 1 + cabac_init_idc otherwise
    void init_contexts(int table, int slice_qp) {
        int qp = slice_qp < 0 ? 0 : (slice_qp > 51 ? 51 : slice_qp);
        for (int i = 0; i < CABAC_N_CTX; i++) {
            int pre = ((cabac_init_mn[table][i][0] * qp) >> 4) + cabac_init_mn[table][i][1];
            pre = pre < 1 ? 1 : (pre > 126 ? 126 : pre);
            state[i] = pre <= 63 ? (uint8_t)((63 - pre) << 1) : (uint8_t)(((pre - 64) << 1) | 1);
        }
    }
    // 9.3.1.2: the reader must be byte aligned at the first byte of the arithmetic code
    void init_engine(BitReader *b) { br = b; range = 510; offset = br->u(9); }

    inline int decision(int ctx) {
        uint32_t s = state[ctx], st = s >> 1, mps = s & 1;
        uint32_t lps = cabac_range_lps[st][(range >> 6) & 3];
        range -= lps;
        if (offset < range) {                                  // most probable symbol
            state[ctx] = (uint8_t)(((st < 62 ? st + 1 : st) << 1) | mps);
            if (range < 256) { range <<= 1; offset = (offset << 1) | br->u1(); }
            return (int)mps;
        }
        offset -= range; range = lps;
        state[ctx] = (uint8_t)((cabac_trans_lps[st] << 1) | (st == 0 ? mps ^ 1 : mps));
        int sh = __builtin_clz(range) - 23;                    // range in [6, 240] -> shift to bring it into [256, 511]
        range <<= sh; offset = (offset << sh) | br->u(sh);
        return (int)(mps ^ 1);
    }
    inline int bypass() {
        offset = (offset << 1) | br->u1();
        if (offset >= range) { offset -= range; return 1; }
        return 0;
    }
    inline int terminate() {
        range -= 2;
        if (offset >= range) return 1;
        if (range < 256) { range <<= 1; offset = (offset << 1) | br->u1(); }
        return 0;
    }
};

}  // namespace jmamd
