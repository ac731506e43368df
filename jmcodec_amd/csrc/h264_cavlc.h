// jmcodec_amd/csrc/h264_cavlc.h -- host entropy stage: slice_data() (CAVLC 9.2 or CABAC 9.3) -> macroblock job list.
//
// This is the "entropy decode stays on the host and feeds a per-macroblock job
// list" half of the replacement for cuvidDecodePicture
// (/root/reference/nv_dec/nv_dec.cpp:33-41).  No pixel is touched here.
#pragma once
#include "h264_syntax.h"
#include "jobs.h"
#include <condition_variable>
#include <mutex>
#include <vector>

namespace jmamd {

// Destination of one picture's job list (pointers into a pinned JobBuffer).
struct JobWriter {
    MbRec   *mbs = nullptr;
    int16_t *mv_ext = nullptr;  uint32_t mv_ext_count = 0, mv_ext_cap = 0;   // in MVs (int16 pairs)
    int16_t *coef = nullptr;    uint32_t coef_count = 0, coef_cap = 0;       // in int16
    int max_mvx = 0;            // largest rightward horizontal vector component (quarter samples, >= 0), for the same purpose
    int max_mvy = 0;            // largest downward vertical vector component of the picture (quarter samples, >= 0): how far below a macroblock
                                // its reference windows can reach -- the engine spaces the pictures of a chain launch accordingly (engine.cpp)
};

// Per-worker scratch: neighbour context of the picture being parsed.
struct ParseScratch {
    int mb_w = 0, mb_h = 0;
    std::vector<uint8_t> tc;       // [mb][24] total_coeff per 4x4 (16 luma raster, 4 Cb, 4 Cr)
    std::vector<int16_t> mv;       // [mb][16][2]
    std::vector<int8_t>  refidx;   // [mb][4]
    std::vector<uint8_t> i4;       // [mb][16] Intra4x4PredMode (2 when not I4x4)
    std::vector<uint8_t> info;     // [mb] bit0 intra, bit1 I_NxN (Intra4x4 / Intra8x8), bit2 skipped, bit3 transform_size_8x8_flag, bit4 I_PCM
    std::vector<int16_t> slice_of; // [mb] slice number or -1
    // CABAC neighbour context (9.3.3.1.1): coded_block_pattern (luma bits 0-3, chroma << 4), intra_chroma_pred_mode,
    // coded_block_flag bits (0-15 luma raster, 16 Intra16x16 DC, 17/18 Cb/Cr DC, 19-22 Cb AC, 23-26 Cr AC), |mvd| per 4x4
    std::vector<uint8_t> cbp, cmode; std::vector<uint32_t> cbf; std::vector<uint8_t> mvd;
    // list 1 (B slices), direct-predicted 8x8 quadrants (bit b8; bit 4: B_Skip / B_Direct_16x16), referenced picture ids
    std::vector<int16_t> mv1; std::vector<int8_t> refidx1; std::vector<uint8_t> mvd1, direct8; std::vector<int32_t> uid0, uid1;
    void resize(int w, int h);
    void begin_picture();
};

// Motion field of a decoded picture kept for direct prediction in later B pictures (8.4.1.2): written once by the
// worker that parsed the picture, read by workers parsing B pictures whose RefPicList1[0] it is.
struct MotionField {
    std::vector<int16_t> mv[2];    // [mb][16][2]
    std::vector<int8_t>  ref[2];   // [mb][4] reference index (-1 unused)
    std::vector<int32_t> uid[2];   // [mb][4] unique id of the referenced picture
    std::vector<uint8_t> intra;    // [mb]
    std::mutex m; std::condition_variable cv; bool done = false;
    void wait() { std::unique_lock<std::mutex> lk(m); cv.wait(lk, [&] { return done; }); }
    void publish() { { std::lock_guard<std::mutex> lk(m); done = true; } cv.notify_all(); }
};

// What a slice knows about its reference picture lists (built by the decoder front end, 8.2.4)
struct SliceRefs {
    int8_t  slot[2][32];           // DPB surface of RefPicListX[i], -1 = missing
    int32_t uid[2][32];            // unique picture id
    int32_t poc[2][32];
    uint8_t is_long[2][32];
    int32_t cur_poc = 0;
    const MotionField *col = nullptr;   // motion of RefPicList1[0] (B slices)
    // 8.4.1.2.1, Table 8-8 (no MBAFF): 0 = colocated picture coded like the current one (One_To_One); 1 = the current picture is a FIELD, `col` is the motion
    // of the FRAME picture that holds RefPicList1[0] (Frm_To_Fld); 2 = the current picture is a FRAME, `col` is the motion of the field of the
    // complementary field pair RefPicList1[0] that is nearer in order count (Fld_To_Frm).  uids: 2 x decode index of the store (+ parity for a field)
    int col_mode = 0, cur_parity = 0;
    bool track_uid = false;        // the stream may contain B pictures: remember which picture every block refers to
    bool bipred_rec = false;       // every inter macroblock of this slice uses the MBM_BIPRED motion record (B slice / weighted prediction)
};

// Optional syntax digest (tests): FNV-1a over a canonical serialisation of every macroblock,
// comparable with the CPU oracle's digest of the same stream (tests/test_host_parser.py).
struct SyntaxDigest { uint64_t h = 1469598103934665603ull; uint64_t mbs = 0; };

struct SliceParseResult {
    int mbs_decoded = 0;
    int n_intra = 0;            // I4x4 + I8x8 + I16x16 macroblocks (need the wavefront kernel)
    int n_i8x8 = 0;             // of which Intra8x8
    const char *error = nullptr;
};

// Parses slice_data() of one slice.  br must be positioned at sh.data_bit_offset with
// set_end_from_trailing() already called.  allow_fast = false: every macroblock through the general path (tests compare the two).
SliceParseResult parse_slice_data(const SeqParams &sps, const PicParamSet &pps, const SliceHeader &sh,
                                  BitReader &br, int slice_num, const SliceRefs &refs,
                                  ParseScratch &cx, JobWriter &out, SyntaxDigest *digest, bool allow_fast = true);

void cavlc_init_tables();   // idempotent, thread-safe

}  // namespace jmamd
