/*
 * oracle/h264_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement ("oracle") of the H.264 encode hot path that the reference
 * delegates to ISVCEncoder::EncodeFrame
 * (/root/reference/video_codec/VideoEncoderOpenH264.cpp:344), plus a test-side
 * decoder used for encode->decode round trips.
 *
 * PARITY UNPINNED vs OpenH264: the reference ships no codec source, no
 * libopenh264.so, no tests and no golden vectors.
 * What pins this oracle instead: (i) known answers from ITU-T H.264 held in
 * tests/, (ii) encoder reconstruction == decoder output byte for byte, which
 * ties every normative stage to the standard.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library.  The product (media_amd/) never does.
 */
#ifndef ORACLE_H264_ORACLE_H
#define ORACLE_H264_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Layout of one macroblock's quantised levels (int16), shared with the GPU
 * path's debug dump so the two can be compared array-for-array. */
enum {
    H264O_LV_LUMA_DC = 0,     /* 16: Intra16x16 DC levels, zig-zag order            */
    H264O_LV_LUMA = 16,       /* 16 blocks (blkIdx order) x 16 levels, zig-zag      */
    H264O_LV_CHROMA_DC = 272, /* Cb 4, Cr 4 (raster 2x2 order = chroma DC scan)     */
    H264O_LV_CHROMA_AC = 280, /* Cb blk0..3, Cr blk0..3, x16 zig-zag (idx0 unused)  */
    H264O_LV_STRIDE = 416     /* int16 per macroblock (832 B)                       */
};

enum { H264O_MB_I16 = 0, H264O_MB_P16 = 1, H264O_MB_PSKIP = 2, H264O_MB_IPCM = 3, H264O_MB_I4 = 4,
       H264O_MB_P16X8 = 5, H264O_MB_P8X16 = 6, H264O_MB_P8X8 = 7 };   /* 5..7: two 16x8, two 8x16, four 8x8 partitions (one reference) */
#define H264O_MB_IS_INTRA(t) ((t) == H264O_MB_I16 || (t) == H264O_MB_IPCM || (t) == H264O_MB_I4)

typedef struct {
    int32_t width, height;  /* display size, even, 16..4096                          */
    int32_t fps;            /* 30 or 60 (level selection only)                       */
    int32_t qp;             /* fixed frame QP, 10..51                                */
    int32_t gop;            /* IDR period in frames (uiIntraPeriod, ref :242)        */
    int32_t profile_idc;    /* 66 baseline, 77 main (CAVLC), 100 high (no 8x8)       */
    int32_t disable_deblock;/* 0: in-loop filter on (ref :295 iLoopFilterDisableIdc) */
    int32_t slices;         /* 0/1: one slice per picture (the reference preset, SM_SINGLE_SLICE :247); n > 1: n bands of
                             * ceil(rows / n) macroblock rows, one slice NAL each, no loop filtering across them      */
    int32_t band_index;     /* band_count > 1: this instance codes only its share of the slices (slice bands of one   */
    int32_t band_count;     /* picture on several instances / GPUs, mirrors mi355x_h264_config)                       */
    int32_t refs;           /* 0/1: one reference frame (the reference preset, iNumRefFrame = 1 :290); 2, 3: the motion
                             * search runs on the last `refs` pictures (BASELINE.json configs[4]), ref_idx_l0 is coded      */
    int32_t search;         /* integer motion search (mirrors mi355x_h264_config.search): 0 = exhaustive +-16 around the co-located
                             * macroblock; 1 = seeded - when the macroblock's previous-picture vector, rounded to integer samples,
                             * is a strict local minimum of the search cost among its eight integer neighbours, it is taken as
                             * the integer winner and the exhaustive pass is skipped (it runs whenever the test fails)         */
} h264o_config;

typedef struct h264o_enc h264o_enc;
typedef struct h264o_dec h264o_dec;

/* per-macroblock side information exposed for stage-by-stage parity checks */
typedef struct {
    int16_t mvx, mvy;      /* quarter-pel motion vector (0 for intra); with partitions: that of the first one */
    uint8_t type;          /* H264O_MB_*                                            */
    uint8_t i16_mode;      /* Intra16x16PredMode 0..3; inter: transform_size_8x8_flag */
    uint8_t chroma_mode;   /* intra_chroma_pred_mode 0..3; inter: ref_idx_l0          */
    uint8_t cbp;           /* coded_block_pattern (luma | chroma<<4)                */
    uint8_t tc[24];        /* TotalCoeff per 4x4: 16 luma (blkIdx), 4 Cb, 4 Cr      */
} h264o_mbinfo;            /* 32 bytes                                              */

h264o_enc *h264o_enc_create(const h264o_config *cfg);
void h264o_enc_destroy(h264o_enc *e);
/* Encode one I420 picture.  Returns bytes written (Annex B, 4-byte start codes;
 * SPS+PPS precede every IDR) or <0 on error.  *is_idr receives 1 for IDR. */
int64_t h264o_enc_encode(h264o_enc *e, const uint8_t *y, int ys, const uint8_t *u, int us,
                         const uint8_t *v, int vs, int force_idr, uint8_t *out, size_t out_cap,
                         int *is_idr);
/* picture QP for the pictures that follow (mirrors mi355x_h264_set_qp) */
int h264o_enc_set_qp(h264o_enc *e, int qp);
/* idr_pic_id of the next IDR and its increment per IDR (mod 256): lets closed GOPs of one
 * stream be encoded by different instances and still concatenate to the serial stream */
int h264o_enc_set_idr_id(h264o_enc *e, int next, int step);
/* band mode: reference rows next to the band, exchanged with the neighbours (mirrors mi355x_h264_band_halo_*) */
size_t h264o_enc_halo_bytes(const h264o_enc *e);
void h264o_enc_halo_export(h264o_enc *e, int edge, uint8_t *dst);
void h264o_enc_halo_import(h264o_enc *e, int edge, const uint8_t *src);
/* Accessors valid until the next encode call.  Planes are coded size
 * (multiples of 16), pitch == coded width (chroma: half). */
int h264o_enc_coded_width(const h264o_enc *e);
int h264o_enc_coded_height(const h264o_enc *e);
const uint8_t *h264o_enc_recon(const h264o_enc *e, int plane);       /* deblocked */
const uint8_t *h264o_enc_recon_pre(const h264o_enc *e, int plane);   /* before loop filter */
const h264o_mbinfo *h264o_enc_mbinfo(const h264o_enc *e);
/* 8 int16 per macroblock beside mbinfo: the vectors (x, y) of the four 8x8 quadrants of an inter macroblock (a 16x16
 * macroblock carries its vector four times, a 16x8 one twice twice ...) */
const int16_t *h264o_enc_mvq(const h264o_enc *e);
/* 16 bytes per macroblock beside mbinfo: Intra4x4PredMode of the 16 blocks (blkIdx order) for H264O_MB_I4 */
const uint8_t *h264o_enc_mbaux(const h264o_enc *e);
/* Intra4x4 prediction (8.3.1.2) of one block: rec points at the block inside the picture under reconstruction;
 * avail bit0 left, bit1 top, bit2 top-left, bit3 top-right.  Returns 0, or -1 when the mode needs unavailable samples */
int h264o_pred4x4(const uint8_t *rec, int stride, int mode, int avail, uint8_t pred[16]);
const int16_t *h264o_enc_levels(const h264o_enc *e);
/* bits of slice_data() of the last slice, before trailing bits (for tests) */
int64_t h264o_enc_last_slice_bits(const h264o_enc *e);
/* scene-change statistic (mirrors mi355x_h264_last_me_cost) */
uint32_t h264o_enc_last_me_cost(const h264o_enc *e);

/* One picture of RANDOM conforming syntax (decoder-peer tests; nothing is reconstructed - oracle/h264_dec.c says what the
 * stream decodes to).  features: 1 mb_qp_delta + slice QPs, 2 chroma_qp_index_offsets, 4 filter offsets, 8 I_PCM macroblocks,
 * 16 a random disable_deblocking_filter_idc, 32 inter macroblocks written from random draws as such: sub_mb_types down to 4x4, a
 * ref_idx_l0 per partition, random mvd_l0 (the vectors are then whatever the decoder adds up: mvq / mbinfo vectors are not filled),
 * 64 (with 32) slices cut at random macroblocks instead of bands of rows, 128 ref_pic_list_modification commands in P slices,
 * 256 parameter sets and slice headers laid out the way OpenH264 writes them (15-bit frame_num, POC type 0, VUI, a list modification
 * in every P slice; the same for every picture of a stream), 512 levels of 128 .. 427 at QP_Y <= 14 (use with 1: QPs are then drawn from 4 .. 48),
 * 1024 constrained_intra_pred_flag = 1.  mbqp_out (one byte per macroblock, may be NULL) receives QP_Y of every
 * macroblock (0 for I_PCM, the value the loop filter uses).  Side information: h264o_enc_mbinfo / _mvq / _mbaux / _levels. */
int64_t h264o_enc_random_picture(h264o_enc *e, uint32_t seed, int force_idr, int features, uint8_t *out, size_t out_cap,
                                 int *is_idr, uint8_t *mbqp_out);

/* ---- stand-alone stage functions (kernel-level parity, known-answer tests) ---- */
void h264o_fdct4x4(const int16_t in[16], int16_t out[16]);
void h264o_idct4x4_add(const int16_t coef[16], uint8_t *dst, int stride);
/* quantise one 4x4 of forward-transform output (raster); intra selects the
 * rounding offset; returns levels in raster order */
void h264o_quant4x4(const int16_t w[16], int qp, int intra, int16_t lv[16]);
void h264o_dequant4x4(const int16_t lv[16], int qp, int16_t out[16]);
/* High profile 8x8 transform: forward (non-normative), quantiser, 8.5.13 scaling and inverse transform */
void h264o_fdct8x8(const int16_t in[64], int32_t out[64]);
void h264o_quant8x8(const int32_t w[64], int qp, int intra, int16_t lv[64]);
void h264o_dequant8x8(const int16_t lv[64], int qp, int32_t out[64]);
void h264o_idct8x8_add(const int32_t coef[64], uint8_t *dst, int stride);
/* luma quarter-pel and chroma eighth-pel motion compensation (8.4.2.2) on a
 * w x h plane with edge clamping */
void h264o_mc_luma(const uint8_t *ref, int stride, int w, int h, int x, int y, int mvx, int mvy,
                   int bw, int bh, uint8_t *dst, int dstride);
/* spec-literal one-sample evaluation of 8.4.2.2.1 (cross-check of the block form) */
int h264o_luma_sample_ref(const uint8_t *ref, int stride, int w, int h, int x, int y, int mvx, int mvy);
void h264o_mc_chroma(const uint8_t *ref, int stride, int w, int h, int x, int y, int mvx,
                     int mvy, int bw, int bh, uint8_t *dst, int dstride);
int h264o_sad16x16(const uint8_t *a, int as, const uint8_t *b, int bs);
int h264o_satd16x16(const uint8_t *a, int as, const uint8_t *b, int bs);
int h264o_satd8x8(const uint8_t *a, int as, const uint8_t *b, int bs);
/* (sum over the 4x4 blocks of a w x h rectangle of |Hadamard|) >> 1; w, h multiples of 4 */
int h264o_satd_rect(const uint8_t *a, int as, const uint8_t *b, int bs, int w, int h);
/* intra predictors; avail bit0 = left, bit1 = top, bit2 = top-left */
void h264o_pred16x16(const uint8_t *rec, int stride, int mode, int avail, uint8_t pred[256]);
void h264o_pred_chroma8x8(const uint8_t *rec, int stride, int mode, int avail, uint8_t pred[64]);
/* deblock a whole picture in place given per-MB info (8.7); slice_of (slice index per macroblock) non-NULL =
 * disable_deblocking_filter_idc 2, edges between different slices are left alone; macroblock rows row0..row1-1 */
void h264o_deblock_picture(uint8_t *y, uint8_t *u, uint8_t *v, int cw, int ch,
                           const h264o_mbinfo *mbs, const int16_t *mvq, int qp, const int16_t *slice_of, int row0, int row1);
/* Exp-Golomb / CAVLC helpers for known-answer tests */
/* RGBA ingest (h264_rgba.c): rgba = R, G, B, A bytes per sample, stride in bytes; i420 = w*h*3/2 bytes */
void h264o_rgba_to_i420(const uint8_t *rgba, int stride, int w, int h, uint8_t *i420);
int h264o_ue_bits(uint32_t v, uint32_t *code); /* returns length, *code = bit pattern */
int h264o_se_bits(int32_t v, uint32_t *code);
/* writes one residual block with CAVLC; returns number of bits appended to buf
 * (MSB first, buf must be zeroed, cap >= 64 bytes) */
int h264o_cavlc_block(const int16_t *lv, int max_coeff, int nC, uint8_t *buf);
size_t h264o_nal_escape(const uint8_t *rbsp, size_t n, uint8_t *out);

/* ---- decoder ---- */
h264o_dec *h264o_dec_create(void);
void h264o_dec_destroy(h264o_dec *d);
/* Feed one access unit (any number of Annex B NALs).  Returns 1 when a picture
 * was completed, 0 for parameter sets only, <0 on a syntax/feature error. */
int h264o_dec_decode(h264o_dec *d, const uint8_t *data, size_t len);
int h264o_dec_width(const h264o_dec *d);         /* cropped */
int h264o_dec_height(const h264o_dec *d);
int h264o_dec_coded_width(const h264o_dec *d);
int h264o_dec_coded_height(const h264o_dec *d);
const uint8_t *h264o_dec_plane(const h264o_dec *d, int plane); /* coded size, pitch = coded w */
const char *h264o_dec_error(const h264o_dec *d);
int h264o_dec_last_slice_type(const h264o_dec *d); /* 0 P, 2 I */
int h264o_dec_last_nal_type(const h264o_dec *d);

#ifdef __cplusplus
}
#endif
#endif
