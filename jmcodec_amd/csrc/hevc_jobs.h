// jmcodec_amd/csrc/hevc_jobs.h -- the host->device contract of one HEVC picture.
//
// Replaces the CUVIDHEVCPICPARAMS hand-off of the reference (/root/reference/nv_dec/nv_dec.cpp:33-41 -> cuvidDecodePicture with
// nv_sdk/inc/dynlink_cuviddec.h:428-530): there the slice bytes go to the NVDEC ASIC; here CABAC decoding, motion vector
// prediction (merge / AMVP / temporal), de-quantisation and the boundary-strength decisions stay on the host and the device gets
//   HevcCtb[ctbs]        SAO parameters, deblocking offsets, the CTB's intra blocks
//   qp8[]                QpY per 8x8 (bit 7: samples exempt from the loop filters: pcm_loop_filter_disabled / transquant bypass)
//   (round 5) the boundary strengths bs_v[] / bs_h[] (per 4-sample edge segment of the 8x8 grid, 0 = not filtered) are no longer computed by the host:
//   k_hevc_bs_raster / k_hevc_bs derive them on the device from the lists below (which already say where the prediction blocks, the intra blocks and the
//   coded transform blocks lie) and a few per-CTB flags (HevcCtb.db_flags) -- 9 % of the host's parse time per picture
//   HevcPu[]             motion-compensated blocks of at most 16x16 luma samples
//   HevcTb[]             transform blocks of inter coding units that carry coefficients
//   HevcIntraTb[]        intra-predicted transform blocks in decoding order, grouped per CTB
//   coefs[]              scaled (de-quantised) coefficients, sparse: position | value << 16
//   HevcWp[]             explicit weighted-prediction tables per slice
// in one pinned buffer, uploaded with a single copy.
#pragma once
#include <stdint.h>
#include "jobs.h"

namespace jmamd {

struct HevcCtb {                  // 32 bytes
    uint8_t  sao_type[3];         // 0 off, 1 band offset, 2 edge offset
    uint8_t  sao_pos[3];          // band position / edge class
    int8_t   sao_off[3][4];
    int8_t   beta_off, tc_off;    // slice_beta_offset_div2 / slice_tc_offset_div2 of the CTB's slice
    uint8_t  nb_mask;             // bit k set: SAO edge offset may use samples of neighbouring CTB k (L, R, T, B, TL, TR, BL, BR)
    uint8_t  intra_edge;          // bit 0: some intra block of the CTB reaches its bottom row -- the CTBs below (left, straight, right) read what k_hevc_intra
                                  // reconstructs there; bit 1: ... its right column -- the CTB to the right does
    uint8_t  db_flags;            // HDB_*: what the deblocking needs to know of the CTB's slice / tile (8.7.2.3: filterEdgeFlag)
    uint8_t  pad;
    uint32_t intra_first, intra_count;
};
static_assert(sizeof(HevcCtb) == 32, "HevcCtb layout");

struct HevcPu {                   // 20 bytes
    uint16_t x, y;                // luma position
    uint8_t  w, h;                // luma size (4..16)
    int8_t   slot0, slot1;        // DPB surface of list 0 / list 1, -1 = list unused
    int16_t  mv0[2], mv1[2];      // quarter-sample units
    uint8_t  ridx0, ridx1;        // reference indices (select the weights)
    uint16_t wp;                  // 0 = default weighted prediction, else 1 + index into HevcWp[]
};
static_assert(sizeof(HevcPu) == 20, "HevcPu layout");

enum : uint8_t { HTB_TSKIP = 1, HTB_BYPASS = 2, HTB_DST = 4, HTB_CORNER = 8 };
// HevcCtb.db_flags
enum : uint8_t { HDB_DISABLED = 1,     // slice_deblocking_filter_disabled_flag of the CTB's slice: no edge whose q side lies in the CTB is filtered
                 HDB_CONCEALED = 2,    // no slice delivered the CTB (concealment): its own edges are not filtered, its samples count as intra towards neighbours
                 HDB_NO_LEFT = 4,      // the CTB's left boundary is a slice / tile boundary that must not be filtered across
                 HDB_NO_TOP = 8 };     // ... its top boundary
constexpr uint8_t kHevcModePcm = 255;

struct HevcTb {                   // 16 bytes
    uint16_t x, y;                // position in samples of its plane
    uint8_t  log2, plane, flags, pad;
    uint32_t coef_off, coef_n;    // a LUMA block with cbf_luma = 1 is listed even when no level survives the scaling (coef_n == 0): its edges still get bS 1
};
struct HevcIntraTb {              // 20 bytes
    uint16_t x, y;
    uint8_t  log2, plane, mode, flags;   // mode: 0..34, kHevcModePcm = samples are the "coefficients"; flags: HTB_*
    uint32_t avail;               // bit i (0..15): left neighbour unit i (top to bottom, 4 luma rows each) available; bit 16 + i: top unit i (left to right)
    uint32_t coef_off, coef_n;    // coef_n == 0: prediction only
};
static_assert(sizeof(HevcTb) == 16 && sizeof(HevcIntraTb) == 20, "transform block record layout");

struct HevcWp {                   // explicit weighted prediction of one slice (8.5.3.3.4.3)
    int16_t log2wd[2];            // luma, chroma: denominator + 14 - bitDepth
    int16_t w[2][16][3], o[2][16][3];
};

enum : int { HPS_MC = 1, HPS_RESID = 2, HPS_INTRA = 4, HPS_DEBLOCK = 8, HPS_SAO = 16 };

struct HevcPicParams {
    int w, h;                     // coded luma size
    int pitch, chroma_offset;     // NV12 surfaces, as for H.264
    int ctb_log2, ctb_w, ctb_h;
    int w8;                       // qp8 row stride = w / 8
    int cb_qp_off, cr_qp_off;     // pps_cb_qp_offset / pps_cr_qp_offset (chroma deblocking)
    int strong_intra;
    int stages;                   // HPS_*
    int cur, work;                // surface the finished picture lands in / (diagnostic) index of the work set
    uint8_t *surf[kMaxSurfaces];
    uint8_t *work_surf;           // surface reconstruction and deblocking run in: surf[cur] without SAO, else one of the handle's pre-SAO work surfaces
    const HevcCtb *ctbs;
    const uint8_t *qp8;
    // device scratch of the picture's work set (decoder.cpp): boundary strengths as the deblocking kernel reads them, and what they are derived from --
    // pu_map[cell]: index into pus[] of the block that covers the 4x4 cell; cell_flags: four byte planes of w4 * h4 cells each:
    // [0] cell lies in an intra block, [1] in a luma transform block with cbf_luma = 1, [2] / [3] its left / top edge is an edge of such a block
    uint8_t *bs_v, *bs_h;
    uint32_t *pu_map; uint8_t *cell_flags;
    const HevcPu *pus; int n_pus;
    const HevcTb *tbs; int n_tbs;
    const HevcIntraTb *itbs; int n_itbs;
    const uint32_t *coefs;
    const HevcWp *wps;
    int16_t *resid;               // per-handle scratch: residual of the intra blocks, planar (Y w x h, Cb, Cr), written by k_hevc_iresid
};

}  // namespace jmamd
