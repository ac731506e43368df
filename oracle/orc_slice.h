/* oracle/orc_slice.h -- CPU ORACLE internals (test infrastructure only): state shared by the
 * macroblock layer (orc_slice.c) and the CABAC syntax-element decoders (orc_cabac.c). */
#ifndef ORC_SLICE_H
#define ORC_SLICE_H
#include "orc_internal.h"
#include "orc_cabac_tables.h"

/* 9.3.1 / 9.3.3.2 arithmetic decoding engine + context variables */
typedef struct {
    uint8_t state[ORC_CABAC_N_CTX];   /* pStateIdx (bits 0-5) | valMPS << 6 */
    unsigned range, offset;
    Bits *b;
} Cabac;

typedef struct Sl {
    OrcDec *d; Bits *b; Picture *pic; const Sps *sps; const Pps *pps; const SliceHdr *sh;
    int mb_x, mb_y, mb_addr;
    int qp;                          /* running QP_Y                                  */
    MbInfo *mb;
    /* parsed residual of the current MB */
    int16_t luma[16][16];            /* per 4x4 raster block index, coefficient raster */
    int16_t luma8[4][64];            /* per 8x8 block, raster                          */
    int16_t i16dc[16];               /* raster 4x4 matrix c                            */
    int16_t cdc[2][4];
    int16_t cac[2][4][16];
    int i16_pred_mode, chroma_pred_mode;
    int decoded_mask;                /* 4x4 blocks (raster bit) whose MVs are decoded  */
    /* CABAC */
    int cabac_on; Cabac c;
    int last_dqp_nonzero;            /* mb_qp_delta of the previous MB in decoding order != 0 */
} Sl;

MbInfo *orc_sl_mb_at(Sl *s, int mx, int my);      /* neighbour MB if available (same slice), else NULL */

/* orc_cabac.c */
void orc_cabac_init_contexts(Cabac *c, int slice_is_i, int cabac_init_idc, int slice_qp);
int  orc_cabac_init_engine(Cabac *c, Bits *b);     /* b must be byte aligned */
int  orc_cabac_terminate(Cabac *c);
int  orc_cabac_mb_skip_flag(Sl *s);
int  orc_cabac_mb_type(Sl *s);                     /* slice-type numbering of Tables 7-11/7-13/7-14 (intra = 5+ / 23+) */
int  orc_cabac_sub_mb_type(Sl *s);
int  orc_cabac_transform8x8_flag(Sl *s);
int  orc_cabac_intra_pred_mode(Sl *s);             /* -1: use predicted mode, else rem_intra_pred_mode */
int  orc_cabac_chroma_pred_mode(Sl *s);
int  orc_cabac_ref_idx(Sl *s, int list, int bx, int by);
int  orc_cabac_mvd(Sl *s, int list, int bx, int by, int comp);
int  orc_cabac_cbp(Sl *s);
int  orc_cabac_qp_delta(Sl *s);
/* cat: ctxBlockCat 0..5; idx: block index inside its class (luma 4x4 raster / chroma plane<<2|k / 8x8 index);
 * coef[] receives maxnum levels in scan order; returns number of non-zero levels or -1 */
int  orc_cabac_residual_block(Sl *s, int cat, int idx, int16_t *coef, int maxnum);

#endif
