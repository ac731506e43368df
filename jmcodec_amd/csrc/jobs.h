// jmcodec_amd/csrc/jobs.h -- the host->device contract: one picture's macroblock job list.
//
// Replaces the CUVIDPICPARAMS hand-off of the reference
// (/root/reference/nv_dec/nv_dec.cpp:33-41 -> cuvidDecodePicture;
//  struct at nv_sdk/inc/dynlink_cuviddec.h:635-663).  There the slice bytes go
// to a fixed-function ASIC; here entropy decoding stays on the host and the
// device receives, per picture:
//   PicParams   (kernel argument, by value)
//   MbRec[n_mbs]            fixed 32-byte record per macroblock
//   SliceRec[n_slices]      deblocking parameters per slice
//   int16 mv_ext[]          16 MVs for macroblocks with sub-8x8 partitions
//   int16 coef[]            raw (un-scaled) levels, 16 per CODED 4x4 block
// packed into one pinned buffer so that a single hipMemcpyAsync uploads it.
#pragma once
#include <stdint.h>

namespace jmamd {

enum : uint8_t { MB_INTER = 0, MB_I4 = 1, MB_I16 = 2, MB_PCM = 3 };

// MbRec.flags
enum : uint8_t {
    MBF_AVAIL_A = 1,      // left  MB usable for intra prediction
    MBF_AVAIL_B = 2,      // top
    MBF_AVAIL_C = 4,      // top-right
    MBF_AVAIL_D = 8,      // top-left
    MBF_CB_DC   = 16,     // chroma DC block present for Cb
    MBF_CR_DC   = 32,     // ... Cr
    MBF_MV_EXT  = 64,     // sub-8x8 partitions: u.mv_ext indexes 16 MVs in mv_ext[]
    MBF_DECODED = 128,    // macroblock was present in the bitstream (else concealment)
};

// MbRec.modes bit 4: transform_size_8x8_flag -- luma residual is four 8x8 blocks; with MB_I4 the prediction is Intra8x8
// MbRec.modes bit 5: the motion of this macroblock (B slices, slices with weighted prediction) is a 72-int16 record at
// mv_ext[u.mv_ext] (index in int16 pairs):
//   int16 mv[2][16][2]   list 0 then list 1, one vector per 4x4 block (raster)
//   int8  slot1[4]       DPB surface per 8x8 for list 1, -1 = list not used (MbRec.ref[] holds list 0 the same way)
//   int8  idx0[4], idx1[4]  reference indices per 8x8 (they select the weights in SliceWp)
//   int8  pad[4]
enum : uint8_t { MBM_T8X8 = 16, MBM_BIPRED = 32 };
constexpr int kBiRecInt16 = 72;

struct MbRec {            // 32 bytes
    uint8_t  kind;        // MB_*
    uint8_t  qp;          // QP_Y of the macroblock (0 for I_PCM: 8.7.2.2)
    uint8_t  modes;       // bits 0-1 intra_chroma_pred_mode, bits 2-3 Intra16x16PredMode, bit 4 MBM_T8X8
    uint8_t  flags;       // MBF_*
    uint16_t cbp_blk;     // bit b set: luma 4x4 block b (blkIdx, i.e. coding order) has coded levels
    uint8_t  cbp_cac;     // bits 0-3 Cb AC blocks, 4-7 Cr AC blocks
    uint8_t  slice;       // index into SliceRec[]
    uint32_t coef_off;    // first int16 of this macroblock in coef[]
    int8_t   ref[4];      // DPB surface slot per 8x8 (inter), -1 otherwise; in a field picture bit 5 = parity of the reference FIELD (bits 0-4 the slot)
    union {
        int16_t  mv[4][2];    // one MV per 8x8 (quarter-sample units)
        uint8_t  i4[8];       // Intra4x4PredMode, two per byte (low nibble = even raster index);
                              // MBM_T8X8: Intra8x8PredMode of 8x8 block b in nibble b of i4[0..1]
        uint32_t mv_ext;      // MBF_MV_EXT: index (in MVs) of 16 per-4x4 MVs in mv_ext[]
    } u;
};
static_assert(sizeof(MbRec) == 32, "MbRec must stay 32 bytes");

// Coefficient stream layout of one macroblock, starting at coef_off (int16 units):
//   MB_I16 : 16 DC levels (4x4 matrix, raster)              always
//   luma   : 16 levels (raster) for every set bit of cbp_blk, ascending bit order;
//            MBM_T8X8: a coded 8x8 block b sets the four bits 4b..4b+3 and stores 64 levels (8x8 raster) in their place
//   chroma : 4 Cb DC levels if MBF_CB_DC, then 4 Cr DC levels if MBF_CR_DC
//   chroma : 16 levels (raster, [0] unused) per set bit of cbp_cac, ascending
//   MB_PCM : 384 raw bytes (256 Y, 64 Cb, 64 Cr) = 192 int16 slots, nothing else

struct SliceRec {         // 4 bytes
    int8_t  alpha_off;    // FilterOffsetA
    int8_t  beta_off;     // FilterOffsetB
    uint8_t disable;      // disable_deblocking_filter_idc
    uint8_t pad;
};

// Weighted prediction tables of one slice (8.4.2.3); present (PicParams.wp != nullptr) only when a slice of the picture needs them.
struct SliceWp {
    uint8_t mode;             // 0 default, 1 explicit, 2 implicit (bi-predicted blocks only)
    uint8_t logwd_y, logwd_c, pad;
    int8_t  w[2][16][3];      // explicit weights [list][ref_idx][Y, Cb, Cr]
    int8_t  o[2][16][3];      // explicit offsets
    uint8_t imp_w1[16][16];   // implicit: 64 + w1 of the pair (ref_idx_l0, ref_idx_l1), w1 in [-64, 128]; w0 = 64 - w1
};
static_assert(sizeof(SliceWp) == 4 + 96 + 96 + 256, "SliceWp layout");

constexpr int kMaxSurfaces = 20;

struct PicParams {
    int mb_w, mb_h;               // of the picture: a field picture has half the frame's rows
    uint32_t mb_w_magic;          // ceil(2^32 / mb_w): macroblock address / mb_w == mulhi(address, magic) for every address of a picture (address * mb_w < 2^32);
                                  // 0 when mb_w == 1 (2^32 does not fit): the row is the address
    uint32_t surf_stride;         // surf[i] == surf_base + i * surf_stride (one allocation, decoder.cpp): a kernel that knows the slot needs no pointer load
    int field;                    // 0 frame picture; 1 / 2: top / bottom field picture -- the lines of that parity of surf[cur], pitch = 2 x the surface's
    int pitch;                    // bytes per luma row == bytes per interleaved chroma row
    int chroma_offset;            // byte offset of the UV plane inside a surface = pitch * coded_h
    int cb_qp_off, cr_qp_off;     // chroma_qp_index_offset, second_chroma_qp_index_offset
    int n_slices;
    int cur;                      // surface slot being written
    uint8_t *surf[kMaxSurfaces];  // device pointers of the DPB surfaces
    uint8_t *surf_base;           // the one allocation that holds them all (every surf[i] lies within 2^31 bytes above it)
    const MbRec *mbs;
    const SliceRec *slices;
    const int16_t *mv_ext;
    const int16_t *coef;
    int16_t *resid;               // per-handle scratch: residual of intra macroblocks, 384 int16 per MB (Y 16x16, Cb 8x8, Cr 8x8)
    void *dbrec;                  // per-handle scratch: 96 B per macroblock written by k_deblock_prep
    int want_intra_resid;         // 1: k_recon_inter also writes the residual of Intra4x4/16x16 macroblocks to resid
    int stages;                   // PS_* : which kernels of a batched launch act on this picture
    const SliceWp *wp;            // [n_slices] or nullptr
    // scaling matrices (8.5.9) in RASTER order: 4x4 lists Intra Y/Cb/Cr, Inter Y/Cb/Cr; 8x8 lists Intra Y, Inter Y
    int flat_scaling;             // 1: every weight is 16 (Flat_4x4_16 / Flat_8x8_16): kernels skip the table reads
    uint8_t wscale4[6][16];
    uint8_t wscale8[2][64];
    // chain launches (chain_common.h): this picture's block in the batch's chain buffer, and for every surface slot the chain index of
    // the picture of THIS launch that decodes into it (an earlier picture of the same stream), -1 = complete before the launch
    int chain_idx;
    int n_deps;                   // number of entries of dep_pic that are >= 0 and referenced by this picture
    int8_t dep_pic[kMaxSurfaces];
};

// One launch works on a BATCH of pictures (one per stream): kernels take an array of PicParams in device memory
// and use blockIdx.y as the picture index.
enum : int { PS_RECON = 1, PS_INTRA_LDS = 2, PS_INTRA_V1 = 4, PS_DEBLOCK_LDS = 8, PS_DEBLOCK_V1 = 16,
             PS_CHAIN = 32,       // PS_CHAIN: reconstruction + deblocking run inside k_chain (only k_deblock_prep of the stage kernels acts on it)
             PS_CHAIN_INTRA = 64 };   // with PS_CHAIN: a picture with (mostly) intra macroblocks -- its intra wavefront runs inside k_chain too

struct PackJob {                  // one display frame to pack out (k_packout, blockIdx.y = job)
    const uint8_t *src; uint8_t *dst;
    int pitch, chroma_offset, width, height, out_fmt;
    int lone_field;               // 0; 1 / 2: only the top / bottom field of the frame was decoded -- its lines are shown twice
};

}  // namespace jmamd
