// jmcodec_amd/csrc/hevc_slice.h -- HEVC slice segment data: CABAC syntax decoding (H.265 7.3.8, 9.3) into the picture's job lists.
//
// The host half of what the reference delegates to cuvidDecodePicture for codec_type 1 (/root/reference/nv_dec/nv_dec.cpp:33-41):
// entropy decoding, motion vector prediction (8.5.3.2: merge, AMVP, temporal candidates), intra mode derivation (8.4.2), QP
// derivation (8.6.1), scaling of the coefficient levels (8.6.4.1) and the deblocking edge / strength decisions (8.7.2.2-8.7.2.4).
// Sample reconstruction is the device's job (hevc_kernels.hip).
#pragma once
#include "hevc_jobs.h"
#include "hevc_syntax.h"
#include "h264_cabac.h"
#include <condition_variable>
#include <cstdio>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

namespace jmamd {

struct HevcMotion { int16_t mv[2][2]; int8_t ref[2]; uint8_t pf, pad; };      // pf bit 0 = list 0 used, bit 1 = list 1 used

// motion of a decoded picture at 16x16 granularity (8.5.3.2.8); produced by the parse of that picture, consumed by later ones
// Rows are published as the parse of the picture advances, so a picture that uses it as collocated picture can be parsed a CTB row behind it
// instead of after it (the temporal candidate of a block never lies below the block's own CTB row, 8.5.3.2.8).
struct HevcColMotion {
    std::mutex m; std::condition_variable cv; bool ready = false; int rows_ready = 0;      // rows_ready: 16-sample rows whose entries are final
    int w16 = 0, h16 = 0, poc = 0;
    std::vector<HevcMotion> mot; std::vector<int32_t> ref_poc; std::vector<uint8_t> lt, intra;
    void publish() { { std::lock_guard<std::mutex> lk(m); ready = true; rows_ready = h16; } cv.notify_all(); }
    void publish_rows(int n) { { std::lock_guard<std::mutex> lk(m); if (n > rows_ready) rows_ready = n; } cv.notify_all(); }
    void wait() { std::unique_lock<std::mutex> lk(m); cv.wait(lk, [&] { return ready; }); }
    void wait_rows(int n) { std::unique_lock<std::mutex> lk(m); cv.wait(lk, [&] { return ready || rows_ready >= n; }); }
};

struct HevcSliceRefs {            // RefPicList0 / RefPicList1 of one slice (8.3.4)
    int cur_poc = 0;
    int poc[2][16]; uint8_t is_lt[2][16]; int8_t slot[2][16];
    std::shared_ptr<HevcColMotion> col;       // motion of the collocated picture (slice_temporal_mvp_enabled_flag)
};

// A growing array of plain records for the lists the parser appends to ten thousand times per picture: push_back is a compare and a store (no
// exception paths, no value-initialisation), and `tail(n)` hands out room for n more records that `take(k)` then keeps the first k of -- a transform
// block's coefficients are written where they stay.  The storage lives as long as the object (thread-local in the parse workers): no allocation after
// the first pictures.
template <class T> struct PodList {
    T *p = nullptr; size_t n = 0, cap = 0;
    PodList() = default; PodList(const PodList &) = delete; PodList &operator=(const PodList &) = delete;
    ~PodList() { free(p); }
    size_t size() const { return n; }
    T *data() { return p; } const T *data() const { return p; }
    T &operator[](size_t i) { return p[i]; } const T &operator[](size_t i) const { return p[i]; }
    T &back() { return p[n - 1]; }
    void clear() { n = 0; }
    T *tail(size_t more) {
        if (n + more > cap) { size_t c = cap ? cap : 4096; while (c < n + more) c *= 2; T *q = (T *)realloc(p, c * sizeof(T)); if (!q) throw std::bad_alloc();
            p = q; cap = c; }
        return p + n;
    }
    void take(size_t k) { n += k; }
    void push_back(const T &v) { *tail(1) = v; n++; }
};

struct HevcPicJobs {
    std::vector<HevcCtb> ctbs; std::vector<uint8_t> qp8;
    PodList<HevcPu> pus; PodList<HevcTb> tbs; PodList<HevcIntraTb> itbs; PodList<uint32_t> coefs; std::vector<HevcWp> wps;
    bool any_sao = false, any_deblock = false; int n_intra_cu = 0;
    void clear() { ctbs.clear(); qp8.clear(); pus.clear(); tbs.clear(); itbs.clear(); coefs.clear(); wps.clear();
        any_sao = any_deblock = false; n_intra_cu = 0; }
};

struct HevcDigest { bool on = false; uint64_t h = 0xcbf29ce484222325ULL; uint64_t n_cu = 0; FILE *trace = nullptr; };

class HevcPicParser {
public:
    // one picture: begin, every slice segment in decoding order, finish
    void begin_picture(const HevcSps &sps, const HevcPps &pps, int poc, HevcPicJobs *jobs, HevcDigest *dg, HevcColMotion *col_out);
    std::string parse_slice(const HevcSliceHeader &sh, const HevcSliceRefs &refs, const uint8_t *rbsp, size_t len);
    void finish_picture();                                // boundary strengths, QP map, the rest of the motion for temporal prediction

private:
    struct SliceInfo { int addr; bool deblock_disabled, lf_across; int8_t beta_off, tc_off; int8_t slot[2][16]; int poc[2][16]; uint8_t is_lt[2][16]; };
    // ---- syntax ----
    void parse_sao(int rs);
    void note_intra_bottom(int rs);
    bool coding_quadtree(int x0, int y0, int log2, int depth);
    bool coding_unit(int x0, int y0, int log2);
    bool prediction_unit(int xcb, int ycb, int ncb, int x0, int y0, int w, int h, int part_idx);
    bool transform_tree(int x0, int y0, int xb, int yb, int log2, int depth, int blk, int pcb, int pcr);
    bool transform_unit(int x0, int y0, int xb, int yb, int log2, int depth, int blk, int cbf_y, int cbf_cb, int cbf_cr);
    bool residual_coding(int x0, int y0, int log2, int c, int xp, int yp, bool intra_tb);
    void emit_intra_tb(int xp, int yp, int log2, int c, int mode, bool with_coefs);
    // ---- derivations ----
    bool avail_zs(int xc, int yc, int xn, int yn) const;
    int ctb_neighbours();
    bool avail_pb(int xcb, int ycb, int ncb, int xp, int yp, int w, int h, int part, int xn, int yn) const;
    void derive_qp(int xcb, int ycb);
    int merge_candidates(int xcb, int ycb, int ncb, int xp, int yp, int w, int h, int part, int want, HevcMotion *list);
    void amvp(int xcb, int ycb, int ncb, int xp, int yp, int w, int h, int part, int X, int ridx, int flag, int16_t out[2]);
    bool temporal(int xp, int yp, int w, int h, int X, int ridx, int16_t mv[2]);
    void init_contexts();
    inline void dg(int v) { if (dg_->on && dg_->trace) fprintf(dg_->trace, "D %d\n", v); if (dg_->on) for (int i = 0; i < 4; i++) {
        dg_->h ^= (uint8_t)((uint32_t)v >> (8 * i)); dg_->h *= 0x100000001b3ULL; } }
    inline int i4(int x, int y) const { return (y >> 2) * w4_ + (x >> 2); }

    const HevcSps *sps_ = nullptr; const HevcPps *pps_ = nullptr; const HevcSliceHeader *sh_ = nullptr; const HevcSliceRefs *refs_ = nullptr;
    HevcPicJobs *jobs_ = nullptr; HevcDigest *dg_ = nullptr; HevcColMotion *col_out_ = nullptr;
    void export_motion_rows(int r0, int r1);               // rows [r0, r1) of the 16x16 motion field -> col_out_
    int exported_rows_ = 0;
    Cabac cb_;
    Cabac::State wpp_state_[CABAC_N_CTX], dep_state_[CABAC_N_CTX]; bool wpp_valid_ = false, dep_valid_ = false;
    int poc_ = 0, w_ = 0, h_ = 0, w4_ = 0, h4_ = 0, ctb_w_ = 0, ctb_h_ = 0, ctb_size_ = 0, tb_w_ = 0;
    std::vector<int> rs2ts_, ts2rs_, tile_id_, ctb_slice_;
    std::vector<uint32_t> zs_;                 // MinTbAddrZs
    std::vector<uint8_t> pm_, skip_, depth_, ipm_, nofilter_; std::vector<int8_t> qp_; std::vector<HevcMotion> mot_;
    std::vector<uint16_t> ctb_sidx_;                      // index into slices_ by coding tree block (raster scan)
    std::vector<SliceInfo> slices_;
    int slice_idx_ = 0, ctb_rs_ = 0, ctb_ts_ = 0;
    int nbf_rs_ = -1, nbf_ = 0;                                      // ctb_neighbours(): the CTB the flags were looked up for
    // CU state
    int qp_y_ = 0, qp_prev_ = 0, last_cu_qp_ = 0, dqp_ = 0; bool dqp_coded_ = false, first_qg_ = true, cu_since_reset_ = false;
    bool cu_intra_ = false, cu_skip_ = false, tq_bypass_ = false, intra_split_ = false, last_merge_ = false;
    int cu_x_ = 0, cu_y_ = 0, part_mode_ = 0, ipm_c_ = 0, max_tr_depth_ = 0;
    uint16_t wp_index_ = 0;
    int16_t lev_[32 * 32]; uint16_t nz_pos_[32 * 32]; int nz_n_ = 0;
    bool err_ = false, layout_bad_ = false;
    uint64_t layout_key_ = 0;
};

}  // namespace jmamd
