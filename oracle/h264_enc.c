/*
 * oracle/h264_enc.c -- TEST INFRASTRUCTURE ONLY (see h264_oracle.h).
 *
 * Scalar CPU H.264 encoder: the checker the GPU path is compared against.
 * It restates, stage by stage, what happens inside the reference's single
 * external call ISVCEncoder::EncodeFrame
 * (/root/reference/video_codec/VideoEncoderOpenH264.cpp:344; preset at
 * :228-296: one spatial layer, one slice per picture (:247), one reference
 * frame (:290), loop filter on (:295), IDR every uiIntraPeriod (:242)).
 *
 * PARITY UNPINNED vs OpenH264 -- the non-normative choices below are this
 * build's own and are written so that every macroblock decision depends only
 * on data that is final before the stage starts (so the GPU can run each stage
 * over all macroblocks at once):
 *   - intra macroblocks: Intra16x16 (4 modes by SATD on the reconstruction) or Intra4x4, + chroma (4 modes by SATD).  Intra4x4
 *     or 16x16, and the nine-way mode of every 4x4 block, are decided from the SOURCE picture's own samples (every block's
 *     neighbours are then known up front, so all 16 x 9 candidates are costed side by side); the coding itself predicts
 *     each block from the true reconstruction, in blkIdx order.  The top-right block of a macroblock never takes the two
 *     modes that read the macroblock above-right (Diagonal_Down_Left, Vertical_Left): rows stay one macroblock apart
 *   - any macroblock whose CAVLC size could exceed the 3200 bits of A.3.1 is coded as I_PCM.  "Could": decided from an
 *     upper bound on the bits (mb_bits_bound below: sum over the blocks of simple statistics of the levels), not from the
 *     bits themselves, so the decision is made when the macroblock is coded and nothing coded later depends on a later
 *     stage.  A picture that holds an I_PCM macroblock is not loop-filtered (disable_deblocking_filter_idc 1).
 *   - P pictures, macroblocks that went through the motion search and whose motion cost is INTRA_TEST_MIN or more: an
 *     Intra16x16 cost is estimated from the SOURCE picture's neighbouring samples (vertical / horizontal / DC, SATD); if
 *     it is lower, the macroblock is coded Intra16x16 after all inter macroblocks, in raster order, predicting from the
 *     true reconstruction (constrained_intra_pred_flag = 0)
 *   - P pictures: per MB a zero-motion "all levels quantise to zero" test, then
 *     the same test at the macroblock's previous-picture vector rounded to
 *     integer samples, when non-zero (scrolling content), else full search dx,dy in [-16,15] on SAD + lambda*bits(mv - pmv), then half-
 *     and quarter-pel refinement on SATD + lambda*bits(mv - pmv), pmv = this
 *     macroblock's vector in the previous picture; P_L0_16x16 only;
 *     P_Skip iff mv == skip predictor and no coefficient survives
 *   - quantiser: reference-model multipliers, offsets 1/3 (intra), 1/6 (inter)
 *   - CAVLC, fixed picture QP, deblocking per 8.7 with offsets 0
 * Normative stages (dequant, inverse transform, interpolation, intra
 * prediction, deblocking, syntax) follow ITU-T H.264 and are pinned by
 * tests/ known answers plus the encode->decode round trip.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "h264_oracle.h"
#include "h264_tables.h"

static inline int clip3(int lo, int hi, int v) { return v < lo ? lo : (v > hi ? hi : v); }
static inline uint8_t clip1(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

/* ------------------------------------------------------------ bit writer */
typedef struct {
    uint8_t *buf;
    size_t cap;
    uint64_t bits; /* total bits written */
} bitw;

static void bw_put(bitw *b, int n, uint32_t v)
{
    for (int i = n - 1; i >= 0; i--) {
        size_t byte = b->bits >> 3;
        if (byte >= b->cap) { b->bits++; continue; }
        if ((v >> i) & 1) b->buf[byte] |= (uint8_t)(0x80 >> (b->bits & 7));
        b->bits++;
    }
}
static void bw_ue(bitw *b, uint32_t v)
{
    uint32_t code;
    int n = h264o_ue_bits(v, &code);
    if (n > 32) { bw_put(b, n - 32, 0); bw_put(b, 32, code); }
    else bw_put(b, n, code);
}
static void bw_se(bitw *b, int32_t v)
{
    uint32_t code;
    int n = h264o_se_bits(v, &code);
    bw_put(b, n, code);
}
static void bw_trailing(bitw *b)
{
    bw_put(b, 1, 1);
    while (b->bits & 7) bw_put(b, 1, 0);
}
static int se_len(int v)
{
    uint32_t c;
    return h264o_se_bits(v, &c);
}

/* ------------------------------------------------------------ CAVLC 9.2 */
static void cavlc_block(bitw *b, const int16_t *lv, int max_coeff, int nC)
{
    int level[16], idx[16], tc = 0, t1 = 0;
    for (int i = max_coeff - 1; i >= 0; i--)
        if (lv[i]) { level[tc] = lv[i]; idx[tc] = i; tc++; }
    while (t1 < tc && t1 < 3 && abs(level[t1]) == 1) t1++;
    if (nC == -1) {
        bw_put(b, o_chroma_dc_token_len[4 * tc + t1], o_chroma_dc_token_bits[4 * tc + t1]);
    } else {
        int tab = nC < 2 ? 0 : nC < 4 ? 1 : nC < 8 ? 2 : 3;
        bw_put(b, o_coeff_token_len[tab][4 * tc + t1], o_coeff_token_bits[tab][4 * tc + t1]);
    }
    if (!tc) return;
    for (int k = 0; k < t1; k++) bw_put(b, 1, level[k] < 0);
    int suffix_len = (tc > 10 && t1 < 3) ? 1 : 0;
    for (int k = t1; k < tc; k++) {
        int lvl = level[k];
        int code = lvl > 0 ? 2 * lvl - 2 : -2 * lvl - 1;
        if (k == t1 && t1 < 3) code -= 2;
        if (suffix_len == 0) {
            if (code < 14) {
                bw_put(b, code + 1, 1);
            } else if (code < 30) {
                bw_put(b, 15, 1);
                bw_put(b, 4, (uint32_t)(code - 14));
            } else {
                int c = code - 30, prefix = 15;
                while (c >= (1 << (prefix - 3))) { c -= 1 << (prefix - 3); prefix++; }
                /* prefix>15 handled per 9.2.2.1: levelCode += (1<<(prefix-3)) - 4096 */
                if (prefix > 15) c = code - 30 - ((1 << (prefix - 3)) - 4096);
                bw_put(b, prefix + 1, 1);
                bw_put(b, prefix - 3, (uint32_t)c);
            }
        } else {
            if (code < (15 << suffix_len)) {
                bw_put(b, (code >> suffix_len) + 1, 1);
                bw_put(b, suffix_len, (uint32_t)(code & ((1 << suffix_len) - 1)));
            } else {
                int c = code - (15 << suffix_len), prefix = 15;
                while (c >= (1 << (prefix - 3))) { c -= 1 << (prefix - 3); prefix++; }
                if (prefix > 15) c = code - (15 << suffix_len) - ((1 << (prefix - 3)) - 4096);
                bw_put(b, prefix + 1, 1);
                bw_put(b, prefix - 3, (uint32_t)c);
            }
        }
        if (suffix_len == 0) suffix_len = 1;
        if (abs(lvl) > (3 << (suffix_len - 1)) && suffix_len < 6) suffix_len++;
    }
    if (tc < max_coeff) {
        int tz = idx[0] + 1 - tc;
        if (nC == -1) bw_put(b, o_cdc_total_zeros_len[tc - 1][tz], o_cdc_total_zeros_bits[tc - 1][tz]);
        else bw_put(b, o_total_zeros_len[tc - 1][tz], o_total_zeros_bits[tc - 1][tz]);
        int zl = tz;
        for (int k = 0; k < tc - 1 && zl > 0; k++) {
            int run = idx[k] - idx[k + 1] - 1;
            int t = (zl > 7 ? 7 : zl) - 1;
            bw_put(b, o_run_len[t][run], o_run_bits[t][run]);
            zl -= run;
        }
    }
}

int h264o_cavlc_block(const int16_t *lv, int max_coeff, int nC, uint8_t *buf)
{
    bitw b = {buf, 64, 0};
    cavlc_block(&b, lv, max_coeff, nC);
    return (int)b.bits;
}

/* ------------------------------------------------------------ encoder state */
/* syntax elements only the random-stream generator (end of this file) sets; the encoder writes 0 / its fixed choice for each */
struct randsyn {
    int cqo[2];          /* chroma_qp_index_offset, second_chroma_qp_index_offset */
    int idc, oa, ob;     /* disable_deblocking_filter_idc, slice_alpha_c0_offset_div2, slice_beta_offset_div2 */
    int slice_qp[256];   /* SliceQP_Y per slice */
    int8_t *qpd;         /* mb_qp_delta per macroblock (written where the syntax carries one) */
    int32_t *slice_first;/* non-NULL: slices of any shape - per macroblock, the address of the first macroblock of its slice */
    int cur_slice_qp;    /* SliceQP_Y of the slice whose header is being written (slice_first mode) */
    int constrained;     /* constrained_intra_pred_flag = 1 (feature 1024): inter macroblocks are not available for intra prediction */
    int ohstyle;         /* headers laid out the way OpenH264 writes them (feature 256): 15-bit frame_num, pic_order_cnt_type 0 with
                          * pic_order_cnt_lsb in every slice header, VUI with bitstream restrictions, and every P slice carries
                          * ref_pic_list_modification (its one command names the previous picture) */
    int reorder;         /* P slices carry ref_pic_list_modification commands: nreorder of them, target pictures reorder_age[]
                          * (1 = the previous picture ...), the same in every slice of the picture */
    int nreorder, reorder_age[4];
    int direct;          /* inter macroblocks are written from random draws at writing time: sub_mb_types down to 4x4, a
                          * ref_idx_l0 per partition, mvd_l0 values as such (whatever they add up to IS the vector) */
    uint32_t rng;
};
static uint32_t rs_next(uint32_t *s)
{
    uint32_t x = *s;
    x ^= x << 13; x ^= x >> 17; x ^= x << 5;
    return *s = x ? x : 0x9E3779B9u;
}
static int rs_below(uint32_t *s, int n) { return (int)(rs_next(s) % (uint32_t)n); }   /* 0 .. n - 1 */
struct h264o_enc {
    h264o_config cfg;
    int mbw, mbh, cw, ch, level_idc;
    int slice_rows;   /* macroblock rows per slice (mbh for one slice); a slice is a band of whole rows */
    int band_row0, band_row1;   /* rows this instance codes (band_count > 1: its share of the slices), else 0..mbh */
    uint8_t *src[3], *rec[3], *cur[3], *ref[3]; /* coded size; pitch cw / cw/2; ref = the newest reference picture */
    uint8_t *older[2][3];   /* refs > 1: the reference pictures before it (older[0] = ref_idx 1, older[1] = ref_idx 2) */
    int nrefs, avail_refs;  /* configured reference frames; those available for the picture being coded */
    h264o_mbinfo *mb;
    int16_t *levels;
    int16_t *slice_of;   /* slice index of every macroblock */
    int frame_in_gop, frame_num, idr_id, idr_step;
    long frames;
    uint8_t *rbsp;
    size_t rbsp_cap;
    int64_t last_slice_bits;
    uint32_t me_cost; /* scene-change statistic of the last picture */
    int any_pcm;      /* the picture being coded holds an I_PCM macroblock: it is not loop-filtered */
    uint8_t *aux;          /* 16 bytes per macroblock: Intra4x4PredMode of the 16 blocks (blkIdx order) */
    int16_t *mvq;          /* 8 int16 per macroblock: vectors of the four 8x8 quadrants of an inter macroblock */
    uint8_t *pshape;       /* P pictures: partition shape the motion search chose (0 16x16, 1 16x8, 2 8x16, 3 8x8) */
    int rs_cqo[2], rs_constrained;   /* what the last parameter sets of h264o_enc_random_picture said (they travel with IDR pictures only) */
    struct randsyn *rs;   /* h264o_enc_random_picture (decoder-peer tests): syntax the encoder itself never uses; NULL while encoding */
    uint8_t *want_intra;   /* P pictures: 1 = the motion search handed the macroblock to the intra pass; 2 = one of the "nothing
                            * left to code" tests hit: the prediction is the reconstruction, no transform is run (the tests use
                            * the 4x4 transform whatever transform the profile codes with) */
};

static int pick_level(int mbs, int fps)
{
    for (unsigned i = 0; i < sizeof(o_levels) / sizeof(o_levels[0]); i++)
        if ((uint32_t)mbs <= o_levels[i].fs && (uint32_t)(mbs * fps) <= o_levels[i].mbps)
            return o_levels[i].idc;
    return 52;
}

h264o_enc *h264o_enc_create(const h264o_config *cfg)
{
    if (!cfg || cfg->width < 16 || cfg->height < 16 || cfg->width > 4096 || cfg->height > 4096) return NULL;
    if ((cfg->width | cfg->height) & 1) return NULL;
    if (cfg->qp < 10 || cfg->qp > 51) return NULL;
    h264o_enc *e = (h264o_enc *)calloc(1, sizeof(*e));
    e->cfg = *cfg;
    e->idr_step = 1;
    if (e->cfg.gop < 1) e->cfg.gop = 1;
    e->mbw = (cfg->width + 15) / 16;
    e->mbh = (cfg->height + 15) / 16;
    e->cw = e->mbw * 16;
    e->ch = e->mbh * 16;
    {   /* slices: bands of ceil(mbh / slices) macroblock rows (the last one may be shorter), at least two rows each */
        int most = e->mbh / 2 > 1 ? e->mbh / 2 : 1;
        int n = cfg->slices < 1 ? 1 : cfg->slices > most ? most : cfg->slices;
        e->slice_rows = (e->mbh + n - 1) / n;
        int nsl = (e->mbh + e->slice_rows - 1) / e->slice_rows;
        e->band_row0 = 0;
        e->band_row1 = e->mbh;
        if (cfg->band_count > 1 && cfg->band_count <= nsl && cfg->band_index >= 0 && cfg->band_index < cfg->band_count) {
            /* slice bands of one picture on several instances: slices [i*nsl/c, (i+1)*nsl/c) (mirrors mi355x_h264_create) */
            int s0 = (int)((long)cfg->band_index * nsl / cfg->band_count), s1 = (int)((long)(cfg->band_index + 1) * nsl / cfg->band_count);
            e->band_row0 = s0 * e->slice_rows;
            e->band_row1 = s1 * e->slice_rows < e->mbh ? s1 * e->slice_rows : e->mbh;
        }
    }
    int lvl = pick_level(e->mbw * e->mbh, cfg->fps > 0 ? cfg->fps : 30);
    e->level_idc = lvl < 32 ? 32 : lvl; /* the reference asks for LEVEL_3_2 (ref :255) */
    size_t ysz = (size_t)e->cw * e->ch, csz = ysz / 4;
    for (int p = 0; p < 3; p++) {
        size_t sz = p ? csz : ysz;
        e->src[p] = (uint8_t *)calloc(sz, 1);
        e->rec[p] = (uint8_t *)calloc(sz, 1);
        e->cur[p] = (uint8_t *)calloc(sz, 1);
        e->ref[p] = (uint8_t *)calloc(sz, 1);
        for (int k = 0; k < 2; k++) e->older[k][p] = (uint8_t *)calloc(sz, 1);
    }
    e->nrefs = cfg->refs < 1 ? 1 : cfg->refs > 3 ? 3 : cfg->refs;
    e->mb = (h264o_mbinfo *)calloc((size_t)e->mbw * e->mbh, sizeof(h264o_mbinfo));
    e->levels = (int16_t *)calloc((size_t)e->mbw * e->mbh * H264O_LV_STRIDE, sizeof(int16_t));
    e->slice_of = (int16_t *)calloc((size_t)e->mbw * e->mbh, sizeof(int16_t));
    e->want_intra = (uint8_t *)calloc((size_t)e->mbw * e->mbh, 1);
    e->aux = (uint8_t *)calloc((size_t)e->mbw * e->mbh, 16);
    e->mvq = (int16_t *)calloc((size_t)e->mbw * e->mbh, 8 * sizeof(int16_t));
    e->pshape = (uint8_t *)calloc((size_t)e->mbw * e->mbh, 1);
    for (int i = 0; i < e->mbw * e->mbh; i++) e->slice_of[i] = (int16_t)(i / e->mbw / e->slice_rows);
    e->rbsp_cap = ysz * 4 + 65536;
    e->rbsp = (uint8_t *)malloc(e->rbsp_cap);
    return e;
}

void h264o_enc_destroy(h264o_enc *e)
{
    if (!e) return;
    for (int p = 0; p < 3; p++) { free(e->src[p]); free(e->rec[p]); free(e->cur[p]); free(e->ref[p]); free(e->older[0][p]); free(e->older[1][p]); }
    free(e->mb);
    free(e->levels);
    free(e->slice_of);
    free(e->want_intra);
    free(e->aux);
    free(e->mvq);
    free(e->pshape);
    free(e->rbsp);
    free(e);
}

int h264o_enc_set_qp(h264o_enc *e, int qp)
{
    if (!e || qp < 10 || qp > 51) return -1;
    e->cfg.qp = qp;
    return 0;
}
int h264o_enc_set_idr_id(h264o_enc *e, int next, int step)
{
    if (!e) return -1;
    e->idr_id = next & 0xFF;
    e->idr_step = step;
    return 0;
}
/* slice bands on several instances: rows next to the band in the reference picture come from the neighbours
 * (same block layout as mi355x_h264_band_halo_*: 2 macroblock rows of Y, then U, then V) */
enum { HALO_MB_ROWS = 2 };
size_t h264o_enc_halo_bytes(const h264o_enc *e) { return (size_t)HALO_MB_ROWS * 16 * e->cw * 3 / 2; }
static void halo_copy(h264o_enc *e, int r0, int r1, uint8_t *blk, int to_block)
{
    for (int p = 0; p < 3; p++) {
        size_t pitch = p ? e->cw / 2 : e->cw, rpm = p ? 8 : 16;
        uint8_t *pl = e->ref[p] + (size_t)r0 * rpm * pitch;
        size_t n = (size_t)(r1 - r0) * rpm * pitch;
        if (to_block) memcpy(blk, pl, n); else memcpy(pl, blk, n);
        blk += (size_t)HALO_MB_ROWS * rpm * pitch;
    }
}
void h264o_enc_halo_export(h264o_enc *e, int edge, uint8_t *dst)
{
    int rows = e->band_row1 - e->band_row0, n = rows < HALO_MB_ROWS ? rows : HALO_MB_ROWS;
    int r0 = edge == 0 ? e->band_row0 : e->band_row1 - n;
    halo_copy(e, r0, r0 + n, dst, 1);
}
void h264o_enc_halo_import(h264o_enc *e, int edge, const uint8_t *src)
{
    int r0, r1;
    if (edge == 0) { r1 = e->band_row0; r0 = r1 - HALO_MB_ROWS < 0 ? 0 : r1 - HALO_MB_ROWS; }
    else { r0 = e->band_row1; r1 = r0 + HALO_MB_ROWS > e->mbh ? e->mbh : r0 + HALO_MB_ROWS; }
    if (r1 > r0) halo_copy(e, r0, r1, (uint8_t *)src, 0);
}
int h264o_enc_coded_width(const h264o_enc *e) { return e->cw; }
int h264o_enc_coded_height(const h264o_enc *e) { return e->ch; }
const uint8_t *h264o_enc_recon(const h264o_enc *e, int p) { return e->ref[p]; }
const uint8_t *h264o_enc_recon_pre(const h264o_enc *e, int p) { return e->rec[p]; }
const h264o_mbinfo *h264o_enc_mbinfo(const h264o_enc *e) { return e->mb; }
const uint8_t *h264o_enc_mbaux(const h264o_enc *e) { return e->aux; }
const int16_t *h264o_enc_mvq(const h264o_enc *e) { return e->mvq; }
const int16_t *h264o_enc_levels(const h264o_enc *e) { return e->levels; }
int64_t h264o_enc_last_slice_bits(const h264o_enc *e) { return e->last_slice_bits; }
uint32_t h264o_enc_last_me_cost(const h264o_enc *e) { return e->me_cost; }

/* ------------------------------------------------------------ headers 7.3.2 */
static void write_sps(h264o_enc *e, bitw *b)
{
    int prof = e->cfg.profile_idc;
    bw_put(b, 8, (uint32_t)prof);
    /* constraint_set0..5 + 2 reserved bits */
    bw_put(b, 8, prof == 66 ? 0xC0 : prof == 77 ? 0x40 : 0x00);
    bw_put(b, 8, (uint32_t)e->level_idc);
    bw_ue(b, 0); /* seq_parameter_set_id */
    if (prof == 100) {
        bw_ue(b, 1);     /* chroma_format_idc 4:2:0 */
        bw_ue(b, 0);     /* bit_depth_luma_minus8 */
        bw_ue(b, 0);     /* bit_depth_chroma_minus8 */
        bw_put(b, 1, 0); /* qpprime_y_zero_transform_bypass_flag */
        bw_put(b, 1, 0); /* seq_scaling_matrix_present_flag */
    }
    if (e->rs && e->rs->ohstyle) {
        bw_ue(b, 11);    /* log2_max_frame_num_minus4 -> 15 bits */
        bw_ue(b, 0);     /* pic_order_cnt_type 0 */
        bw_ue(b, 12);    /* log2_max_pic_order_cnt_lsb_minus4 -> 16 bits */
    } else {
    bw_ue(b, 4);     /* log2_max_frame_num_minus4 -> MaxFrameNum 256 */
    bw_ue(b, 2);     /* pic_order_cnt_type 2: output order == decode order */
    }
    bw_ue(b, (uint32_t)e->nrefs); /* max_num_ref_frames (ref :290 iNumRefFrame = 1; configs[4]: 3) */
    bw_put(b, 1, 0); /* gaps_in_frame_num_value_allowed_flag */
    bw_ue(b, (uint32_t)e->mbw - 1);
    bw_ue(b, (uint32_t)e->mbh - 1);
    bw_put(b, 1, 1); /* frame_mbs_only_flag */
    bw_put(b, 1, 1); /* direct_8x8_inference_flag */
    int cr = (e->cw - e->cfg.width) / 2, cb = (e->ch - e->cfg.height) / 2;
    if (cr || cb) {
        bw_put(b, 1, 1);
        bw_ue(b, 0);
        bw_ue(b, (uint32_t)cr);
        bw_ue(b, 0);
        bw_ue(b, (uint32_t)cb);
    } else {
        bw_put(b, 1, 0);
    }
    if (e->rs && e->rs->ohstyle) {   /* E.1.1 vui_parameters(): nothing but the bitstream restrictions */
        bw_put(b, 1, 1); /* vui_parameters_present_flag */
        bw_put(b, 1, 0); /* aspect_ratio_info_present_flag */
        bw_put(b, 1, 0); /* overscan_info_present_flag */
        bw_put(b, 1, 0); /* video_signal_type_present_flag */
        bw_put(b, 1, 0); /* chroma_loc_info_present_flag */
        bw_put(b, 1, 0); /* timing_info_present_flag */
        bw_put(b, 1, 0); /* nal_hrd_parameters_present_flag */
        bw_put(b, 1, 0); /* vcl_hrd_parameters_present_flag */
        bw_put(b, 1, 0); /* pic_struct_present_flag */
        bw_put(b, 1, 1); /* bitstream_restriction_flag */
        bw_put(b, 1, 1); /* motion_vectors_over_pic_boundaries_flag */
        bw_ue(b, 0);     /* max_bytes_per_pic_denom */
        bw_ue(b, 0);     /* max_bits_per_mb_denom */
        bw_ue(b, 16);    /* log2_max_mv_length_horizontal */
        bw_ue(b, 16);    /* log2_max_mv_length_vertical */
        bw_ue(b, 0);     /* max_num_reorder_frames */
        bw_ue(b, (uint32_t)e->nrefs); /* max_dec_frame_buffering */
    } else
    bw_put(b, 1, 0); /* vui_parameters_present_flag */
    bw_trailing(b);
}

static void write_pps(h264o_enc *e, bitw *b)
{
    bw_ue(b, 0);     /* pic_parameter_set_id */
    bw_ue(b, 0);     /* seq_parameter_set_id */
    bw_put(b, 1, 0); /* entropy_coding_mode_flag: CAVLC */
    bw_put(b, 1, 0); /* bottom_field_pic_order_in_frame_present_flag */
    bw_ue(b, 0);     /* num_slice_groups_minus1 */
    bw_ue(b, (uint32_t)e->nrefs - 1); /* num_ref_idx_l0_default_active_minus1 */
    bw_ue(b, 0);     /* num_ref_idx_l1_default_active_minus1 */
    bw_put(b, 1, 0); /* weighted_pred_flag */
    bw_put(b, 2, 0); /* weighted_bipred_idc */
    bw_se(b, 0);     /* pic_init_qp_minus26 */
    bw_se(b, 0);     /* pic_init_qs_minus26 */
    bw_se(b, e->rs ? e->rs->cqo[0] : 0);     /* chroma_qp_index_offset */
    bw_put(b, 1, 1); /* deblocking_filter_control_present_flag */
    bw_put(b, 1, e->rs && e->rs->constrained ? 1 : 0); /* constrained_intra_pred_flag */
    bw_put(b, 1, 0); /* redundant_pic_cnt_present_flag */
    if (e->cfg.profile_idc == 100) {
        bw_put(b, 1, 1); /* transform_8x8_mode_flag: inter macroblocks use the 8x8 transform */
        bw_put(b, 1, 0); /* pic_scaling_matrix_present_flag */
        bw_se(b, e->rs ? e->rs->cqo[1] : 0);     /* second_chroma_qp_index_offset */
    }
    bw_trailing(b);
}

static void write_slice_header(h264o_enc *e, bitw *b, int idr, int first_mb)
{
    bw_ue(b, (uint32_t)first_mb); /* first_mb_in_slice */
    bw_ue(b, idr ? 7 : 5);    /* slice_type: all slices of the picture I / P */
    bw_ue(b, 0);              /* pic_parameter_set_id */
    if (e->rs && e->rs->ohstyle) bw_put(b, 15, (uint32_t)e->frame_num);
    else
    bw_put(b, 8, (uint32_t)e->frame_num);
    if (idr) bw_ue(b, (uint32_t)e->idr_id);
    if (e->rs && e->rs->ohstyle) bw_put(b, 16, (uint32_t)(2 * e->frame_in_gop) & 0xFFFFu);   /* pic_order_cnt_lsb */
    if (!idr) {
        /* the first pictures after an IDR have fewer reference pictures than the PPS default announces */
        if (e->avail_refs != e->nrefs) { bw_put(b, 1, 1); bw_ue(b, (uint32_t)e->avail_refs - 1); }
        else bw_put(b, 1, 0); /* num_ref_idx_active_override_flag */
        if (e->rs && e->rs->reorder) {   /* 7.3.3.1: short-term commands by differences of picture numbers (frame_num never wraps here) */
            int pred = e->frame_num;
            bw_put(b, 1, 1);
            for (int k = 0; k < e->rs->nreorder; k++) {
                int target = e->frame_num - e->rs->reorder_age[k];
                if (target < pred) { bw_ue(b, 0); bw_ue(b, (uint32_t)(pred - target - 1)); }
                else { bw_ue(b, 1); bw_ue(b, (uint32_t)(target - pred - 1)); }
                pred = target;
            }
            bw_ue(b, 3);
        } else
        bw_put(b, 1, 0); /* ref_pic_list_modification_flag_l0 */
    }
    if (idr) {
        bw_put(b, 1, 0); /* no_output_of_prior_pics_flag */
        bw_put(b, 1, 0); /* long_term_reference_flag */
    } else {
        bw_put(b, 1, 0); /* adaptive_ref_pic_marking_mode_flag */
    }
    if (e->rs) {   /* random-stream generator: its own QP and filter parameters */
        bw_se(b, (e->rs->slice_first ? e->rs->cur_slice_qp : e->rs->slice_qp[(first_mb / e->mbw / e->slice_rows) & 255]) - 26);
        bw_ue(b, (uint32_t)e->rs->idc);
        if (e->rs->idc != 1) { bw_se(b, e->rs->oa); bw_se(b, e->rs->ob); }
        return;
    }
    bw_se(b, e->cfg.qp - 26); /* slice_qp_delta */
    /* several slices: 2 = no filtering across slice edges, so that the bands stay independent of one another */
    const int no_filter = e->cfg.disable_deblock || e->any_pcm;
    bw_ue(b, no_filter ? 1 : e->slice_rows < e->mbh ? 2 : 0);
    if (!no_filter) {
        bw_se(b, 0); /* slice_alpha_c0_offset_div2 */
        bw_se(b, 0); /* slice_beta_offset_div2 */
    }
}

static size_t emit_nal(uint8_t *out, size_t cap, size_t pos, int ref_idc, int type, const uint8_t *rbsp, size_t n)
{
    if (pos + 5 + n + n / 2 + 8 > cap) return (size_t)-1;
    out[pos++] = 0; out[pos++] = 0; out[pos++] = 0; out[pos++] = 1;
    out[pos++] = (uint8_t)((ref_idc << 5) | type);
    pos += h264o_nal_escape(rbsp, n, out + pos);
    return pos;
}

/* ------------------------------------------------------------ residual coding */
static const uint8_t xy2blk[16] = {0, 1, 4, 5, 2, 3, 6, 7, 8, 9, 12, 13, 10, 11, 14, 15};

/* ---- I_PCM fallback: an upper bound on the CAVLC bits of one residual block from simple statistics of its levels ----
 *   coeff_token <= 16 bits; total_zeros + run_before <= min(2 tc + 22, 73 - 4 tc) (Tables 9-7..9-10: total_zeros <= 9 bits,
 *   a run_before <= 3 bits below 7 zeros left and run - 3 above, at most min(tc - 1, 16 - tc) of them matter);
 *   levels (9.2.2.1): a level of magnitude a coded at suffixLength s >= 1 costs ((2a - 1) >> s) + 1 + s bits, which is at
 *   most max(a + 1, smax + 2) for every s in 1..smax, and never more than 28 (the escape) -> max(min(a, 27), smax + 1) + 1;
 *   smax(largest magnitude m) = 1 for m <= 3, then one more per doubling (the 3 << (s - 1) rule), at most 6 - taken
 *   from the bitwise OR of the magnitudes (>= m): bits(OR) <= 2 -> 1, else min(bits(OR), 6);
 *   the first level (s = 0: up to 2a, 19 or 28 bits) and an escape at s = 1 (28 bits for 15 <= a <= 26) can exceed that by
 *   at most 12, and only one of the two can happen in a block -> + 12.
 * The macroblock bound adds 96 for mb_type, vectors / prediction modes, coded_block_pattern and mb_qp_delta. */
static int blk_bits_bound(const int16_t *lv, int n)
{
    int tc = 0, orv = 0;
    for (int i = 0; i < n; i++) {
        int a = lv[i] < 0 ? -lv[i] : lv[i];
        tc += a != 0;
        orv |= a;
    }
    if (!tc) return 6;
    int b = 0;
    while ((orv >> b) != 0) b++;
    int smax = b <= 2 ? 1 : (b < 6 ? b : 6), h = smax + 1, sum = 0;
    for (int i = 0; i < n; i++) {
        int a = lv[i] < 0 ? -lv[i] : lv[i];
        if (a) sum += ((a < 27 ? a : 27) > h ? (a < 27 ? a : 27) : h) + 1;
    }
    int z = 2 * tc + 22 < 73 - 4 * tc ? 2 * tc + 22 : 73 - 4 * tc;
    return sum + 12 + 16 + z;
}
enum { MB_BITS_LIMIT = 3200, MB_HEADER_BOUND = 96, INTRA_TEST_MIN = 2000 };   /* header: I_NxN carries up to 16 x 4 mode bits */
static int mb_bits_bound(const int16_t *lv, int intra16)
{
    int b = MB_HEADER_BOUND;
    if (intra16) b += blk_bits_bound(lv + H264O_LV_LUMA_DC, 16);
    for (int k = 0; k < 16; k++) b += blk_bits_bound(lv + H264O_LV_LUMA + k * 16, 16);   /* (Intra16x16: level 0 of an AC list is 0) */
    b += blk_bits_bound(lv + H264O_LV_CHROMA_DC, 4) + blk_bits_bound(lv + H264O_LV_CHROMA_DC + 4, 4);
    for (int k = 0; k < 8; k++) b += blk_bits_bound(lv + H264O_LV_CHROMA_AC + k * 16, 16);
    return b;
}

/* forward transform + quant of one 4x4 of (src - pred); returns nnz over
 * zig-zag positions [first..15]; writes zig-zag levels and raster dequantised
 * coefficients (DC left to caller when first==1) */
static int tq_block(const uint8_t *s, int ss, const uint8_t *p, int ps, int qp, int intra, int first,
                    int16_t *lvz, int16_t deq[16], int16_t *dc_out)
{
    int16_t d[16], w[16], lv[16];
    for (int y = 0; y < 4; y++)
        for (int x = 0; x < 4; x++) d[4 * y + x] = (int16_t)(s[y * ss + x] - p[y * ps + x]);
    h264o_fdct4x4(d, w);
    if (dc_out) *dc_out = w[0];
    h264o_quant4x4(w, qp, intra, lv);
    if (first) lv[0] = 0;
    h264o_dequant4x4(lv, qp, deq);
    int nnz = 0;
    for (int i = 0; i < 16; i++) {
        lvz[i] = lv[o_zigzag4x4[i]];
        nnz += lvz[i] != 0;
    }
    return nnz;
}

/* chroma of one MB (both planes): pred in predc[2][64]; writes levels, tc, recon.
 * returns cbp_chroma 0..2 */
static int code_chroma(h264o_enc *e, int mx, int my, uint8_t predc[2][64], int intra, int16_t *lv,
                       uint8_t *tc, int write_recon)
{
    int qpc = o_chroma_qp[e->cfg.qp], cs = e->cw / 2;
    int qbits = 15 + qpc / 6, f = (1 << qbits) / (intra ? 3 : 6);
    int any_dc = 0, any_ac = 0;
    int16_t deq[2][4][16];
    for (int pl = 0; pl < 2; pl++) {
        const uint8_t *s = e->src[1 + pl] + (8 * my) * cs + 8 * mx;
        int16_t dc[4];
        for (int b = 0; b < 4; b++) {
            int bx = (b & 1) * 4, by = (b >> 1) * 4;
            int nnz = tq_block(s + by * cs + bx, cs, predc[pl] + by * 8 + bx, 8, qpc, intra, 1,
                               lv + H264O_LV_CHROMA_AC + (pl * 4 + b) * 16, deq[pl][b], &dc[b]);
            tc[16 + pl * 4 + b] = (uint8_t)nnz;
            any_ac |= nnz;
        }
        int fd[4] = {dc[0] + dc[1] + dc[2] + dc[3], dc[0] - dc[1] + dc[2] - dc[3],
                     dc[0] + dc[1] - dc[2] - dc[3], dc[0] - dc[1] - dc[2] + dc[3]};
        int16_t *ldc = lv + H264O_LV_CHROMA_DC + pl * 4;
        for (int i = 0; i < 4; i++) {
            int a = abs(fd[i]);
            int l = (int)(((int64_t)a * o_quant_mf[qpc % 6][0] + 2 * (int64_t)f) >> (qbits + 1));
            ldc[i] = (int16_t)(fd[i] < 0 ? -l : l);
            any_dc |= l;
        }
        /* 8.5.11.2 chroma DC scaling */
        int c0 = ldc[0], c1 = ldc[1], c2 = ldc[2], c3 = ldc[3];
        int fi[4] = {c0 + c1 + c2 + c3, c0 - c1 + c2 - c3, c0 + c1 - c2 - c3, c0 - c1 - c2 + c3};
        for (int b = 0; b < 4; b++)
            deq[pl][b][0] = (int16_t)(((fi[b] * 16 * o_dequant_v[qpc % 6][0]) << (qpc / 6)) >> 5);
    }
    int cbp = any_ac ? 2 : any_dc ? 1 : 0;
    if (cbp < 2)
        for (int i = 16; i < 24; i++) tc[i] = 0;
    if (write_recon)
        for (int pl = 0; pl < 2; pl++) {
            uint8_t *r = e->rec[1 + pl] + (8 * my) * cs + 8 * mx;
            for (int y = 0; y < 8; y++) memcpy(r + y * cs, predc[pl] + 8 * y, 8);
            for (int b = 0; b < 4; b++) h264o_idct4x4_add(deq[pl][b], r + (b >> 1) * 4 * cs + (b & 1) * 4, cs);
        }
    return cbp;
}

/* 6.4.4: a neighbour in another slice is not available; slices are bands of whole rows here, so only
 * the neighbours above are affected */
static int top_in_slice(const h264o_enc *e, int my) { return my % e->slice_rows != 0; }

/* the macroblock becomes I_PCM (7.3.5, 8.3.5): reconstruction = source samples, TotalCoeff 16 everywhere (9.2.1) */
static void make_pcm(h264o_enc *e, int mx, int my)
{
    int cw = e->cw, cs = cw / 2;
    h264o_mbinfo *mb = &e->mb[my * e->mbw + mx];
    memset(mb, 0, sizeof(*mb));
    mb->type = H264O_MB_IPCM;
    mb->cbp = 0x2F;
    memset(mb->tc, 16, 24);
    for (int y = 0; y < 16; y++) memcpy(e->rec[0] + (16 * my + y) * cw + 16 * mx, e->src[0] + (16 * my + y) * cw + 16 * mx, 16);
    for (int pl = 1; pl < 3; pl++)
        for (int y = 0; y < 8; y++) memcpy(e->rec[pl] + (8 * my + y) * cs + 8 * mx, e->src[pl] + (8 * my + y) * cs + 8 * mx, 8);
    e->any_pcm = 1;
}

/* ------------------------------------------------------------ intra picture */
static void encode_intra_mb(h264o_enc *e, int mx, int my)
{
    int qp = e->cfg.qp, cw = e->cw, cs = cw / 2;
    h264o_mbinfo *mb = &e->mb[my * e->mbw + mx];
    int16_t *lv = e->levels + (size_t)(my * e->mbw + mx) * H264O_LV_STRIDE;
    memset(lv, 0, H264O_LV_STRIDE * sizeof(int16_t));
    memset(mb, 0, sizeof(*mb));
    mb->type = H264O_MB_I16;
    int top = top_in_slice(e, my);
    int avail = (mx > 0 ? 1 : 0) | (top ? 2 : 0) | ((mx > 0 && top) ? 4 : 0);
    const uint8_t *s = e->src[0] + (16 * my) * cw + 16 * mx;
    uint8_t *r = e->rec[0] + (16 * my) * cw + 16 * mx;
    uint8_t pred[256], best_pred[256];
    int best = -1, best_cost = 0;
    /* ---- Intra4x4 or Intra16x16: both costed on the SOURCE picture (its samples stand in for the neighbours'
     * reconstruction), lambda * 48 / lambda * 8 for the modes' bits ---- */
    uint8_t i4mode[16];
    int i4avail[16], use_i4;
    {
        int lambda = o_lambda[qp], est16 = -1, sum4 = 0;
        for (int mode = 0; mode < 4; mode++) {
            if (mode == 0 && !(avail & 2)) continue;
            if (mode == 1 && !(avail & 1)) continue;
            if (mode == 3 && avail != 7) continue;
            h264o_pred16x16(s, cw, mode, avail, pred);
            int c = h264o_satd16x16(s, cw, pred, 16);
            if (est16 < 0 || c < est16) est16 = c;
        }
        for (int b = 0; b < 16; b++) {
            int x = o_blk_x[b], y = o_blk_y[b];
            int left = x > 0 || mx > 0, tp = y > 0 || top;
            int tl = (x > 0 && y > 0) ? 1 : x > 0 ? top : y > 0 ? (mx > 0) : (mx > 0 && top);
            int tr;
            if (y == 0) tr = x < 3 ? top : (top && mx + 1 < e->mbw);
            else if (x == 3) tr = 0;
            else tr = xy2blk[4 * (y - 1) + x + 1] < b;   /* inside the macroblock: available when coded before this block */
            i4avail[b] = left | (tp << 1) | (tl << 2) | (tr << 3);
            int bc = -1, bm = 2;
            for (int m = 0; m < 9; m++) {
                uint8_t p4[16];
                if (x == 3 && y == 0 && (m == 3 || m == 7)) continue;   /* would read the macroblock above-right */
                if (h264o_pred4x4(s + 4 * y * cw + 4 * x, cw, m, i4avail[b], p4)) continue;
                int16_t d[16];
                for (int j = 0; j < 4; j++)
                    for (int i = 0; i < 4; i++) d[4 * j + i] = (int16_t)(s[(4 * y + j) * cw + 4 * x + i] - p4[4 * j + i]);
                int t[16], c = 0;
                for (int i = 0; i < 4; i++) {
                    int s0 = d[4 * i] + d[4 * i + 3], s1 = d[4 * i + 1] + d[4 * i + 2], d0 = d[4 * i] - d[4 * i + 3], d1 = d[4 * i + 1] - d[4 * i + 2];
                    t[4 * i] = s0 + s1; t[4 * i + 1] = d0 + d1; t[4 * i + 2] = s0 - s1; t[4 * i + 3] = d0 - d1;
                }
                for (int j = 0; j < 4; j++) {
                    int s0 = t[j] + t[12 + j], s1 = t[4 + j] + t[8 + j], d0 = t[j] - t[12 + j], d1 = t[4 + j] - t[8 + j];
                    c += abs(s0 + s1) + abs(d0 + d1) + abs(s0 - s1) + abs(d0 - d1);
                }
                if (bc < 0 || c < bc) { bc = c; bm = m; }
            }
            i4mode[b] = (uint8_t)bm;
            sum4 += bc;
        }
        use_i4 = (sum4 >> 1) + 48 * lambda < est16 + 8 * lambda;
    }
    if (use_i4) {
        mb->type = H264O_MB_I4;
        int cbpl = 0;
        for (int b = 0; b < 16; b++) {
            int bx = o_blk_x[b] * 4, by = o_blk_y[b] * 4;
            uint8_t p4[16];
            int16_t deq[16];
            h264o_pred4x4(r + by * cw + bx, cw, i4mode[b], i4avail[b], p4);
            int nnz = tq_block(s + by * cw + bx, cw, p4, 4, qp, 1, 0, lv + H264O_LV_LUMA + b * 16, deq, NULL);
            mb->tc[b] = (uint8_t)nnz;
            for (int j = 0; j < 4; j++) memcpy(r + (by + j) * cw + bx, p4 + 4 * j, 4);
            if (nnz) { cbpl |= 1 << (b >> 2); h264o_idct4x4_add(deq, r + by * cw + bx, cw); }
            e->aux[(size_t)(my * e->mbw + mx) * 16 + b] = i4mode[b];
        }
        /* chroma exactly as in an Intra16x16 macroblock */
        uint8_t predc4[2][64], bestc4[2][64];
        int cb4 = -1, cc4 = 0;
        for (int mode = 0; mode < 4; mode++) {
            if (mode == 1 && !(avail & 1)) continue;
            if (mode == 2 && !(avail & 2)) continue;
            if (mode == 3 && avail != 7) continue;
            int cost = 0;
            for (int pl = 0; pl < 2; pl++) {
                h264o_pred_chroma8x8(e->rec[1 + pl] + (8 * my) * cs + 8 * mx, cs, mode, avail, predc4[pl]);
                cost += h264o_satd8x8(e->src[1 + pl] + (8 * my) * cs + 8 * mx, cs, predc4[pl], 8);
            }
            if (cb4 < 0 || cost < cc4) { cb4 = mode; cc4 = cost; memcpy(bestc4, predc4, sizeof(predc4)); }
        }
        mb->chroma_mode = (uint8_t)cb4;
        int cbpc4 = code_chroma(e, mx, my, bestc4, 1, lv, mb->tc, 1);
        mb->cbp = (uint8_t)(cbpl | (cbpc4 << 4));
        if (mb_bits_bound(lv, 0) > MB_BITS_LIMIT) make_pcm(e, mx, my);
        return;
    }
    for (int mode = 0; mode < 4; mode++) {
        if (mode == 0 && !(avail & 2)) continue;
        if (mode == 1 && !(avail & 1)) continue;
        if (mode == 3 && avail != 7) continue;
        h264o_pred16x16(r, cw, mode, avail, pred);
        int cost = h264o_satd16x16(s, cw, pred, 16);
        if (best < 0 || cost < best_cost) { best = mode; best_cost = cost; memcpy(best_pred, pred, 256); }
    }
    mb->i16_mode = (uint8_t)best;
    /* luma: 16 AC blocks + DC Hadamard */
    int16_t deq[16][16], dcw[16];
    int any_ac = 0;
    for (int b = 0; b < 16; b++) {
        int bx = o_blk_x[b] * 4, by = o_blk_y[b] * 4;
        int nnz = tq_block(s + by * cw + bx, cw, best_pred + by * 16 + bx, 16, qp, 1, 1,
                           lv + H264O_LV_LUMA + b * 16, deq[b], &dcw[o_blk_y[b] * 4 + o_blk_x[b]]);
        mb->tc[b] = (uint8_t)nnz;
        any_ac |= nnz;
    }
    /* forward 4x4 Hadamard of the DC terms (raster by block position), quant at
     * (qbits+2) with offset 4f: the 1/2 of the reference model folded in */
    int t[16], yd[16];
    for (int i = 0; i < 4; i++) {
        int a = dcw[4 * i], b = dcw[4 * i + 1], c = dcw[4 * i + 2], d = dcw[4 * i + 3];
        t[4 * i] = a + b + c + d; t[4 * i + 1] = a + b - c - d;
        t[4 * i + 2] = a - b - c + d; t[4 * i + 3] = a - b + c - d;
    }
    for (int j = 0; j < 4; j++) {
        int a = t[j], b = t[4 + j], c = t[8 + j], d = t[12 + j];
        yd[j] = a + b + c + d; yd[4 + j] = a + b - c - d;
        yd[8 + j] = a - b - c + d; yd[12 + j] = a - b + c - d;
    }
    int qbits = 15 + qp / 6, f = (1 << qbits) / 3;
    int16_t ldc[16];
    for (int i = 0; i < 16; i++) {
        int64_t a = yd[i] < 0 ? -yd[i] : yd[i];
        int l = (int)((a * o_quant_mf[qp % 6][0] + 4 * (int64_t)f) >> (qbits + 2));
        ldc[i] = (int16_t)(yd[i] < 0 ? -l : l);
    }
    for (int i = 0; i < 16; i++) lv[H264O_LV_LUMA_DC + i] = ldc[o_zigzag4x4[i]];
    /* 8.5.10: inverse Hadamard then scaling */
    int fi[16];
    for (int i = 0; i < 4; i++) {
        int a = ldc[4 * i], b = ldc[4 * i + 1], c = ldc[4 * i + 2], d = ldc[4 * i + 3];
        t[4 * i] = a + b + c + d; t[4 * i + 1] = a + b - c - d;
        t[4 * i + 2] = a - b - c + d; t[4 * i + 3] = a - b + c - d;
    }
    for (int j = 0; j < 4; j++) {
        int a = t[j], b = t[4 + j], c = t[8 + j], d = t[12 + j];
        fi[j] = a + b + c + d; fi[4 + j] = a + b - c - d;
        fi[8 + j] = a - b - c + d; fi[12 + j] = a - b + c - d;
    }
    int ls = 16 * o_dequant_v[qp % 6][0];
    for (int i = 0; i < 16; i++) {
        int dc = qp >= 36 ? (fi[i] * ls) << (qp / 6 - 6) : (fi[i] * ls + (1 << (5 - qp / 6))) >> (6 - qp / 6);
        deq[xy2blk[i]][0] = (int16_t)dc;
    }
    if (!any_ac)
        for (int b = 0; b < 16; b++) mb->tc[b] = 0;
    for (int y = 0; y < 16; y++) memcpy(r + y * cw, best_pred + 16 * y, 16);
    for (int b = 0; b < 16; b++) h264o_idct4x4_add(deq[b], r + o_blk_y[b] * 4 * cw + o_blk_x[b] * 4, cw);
    /* chroma mode by SATD over both planes */
    uint8_t predc[2][64], bestc[2][64];
    int cbest = -1, ccost = 0;
    for (int mode = 0; mode < 4; mode++) {
        if (mode == 1 && !(avail & 1)) continue;
        if (mode == 2 && !(avail & 2)) continue;
        if (mode == 3 && avail != 7) continue;
        int cost = 0;
        for (int pl = 0; pl < 2; pl++) {
            h264o_pred_chroma8x8(e->rec[1 + pl] + (8 * my) * cs + 8 * mx, cs, mode, avail, predc[pl]);
            cost += h264o_satd8x8(e->src[1 + pl] + (8 * my) * cs + 8 * mx, cs, predc[pl], 8);
        }
        if (cbest < 0 || cost < ccost) { cbest = mode; ccost = cost; memcpy(bestc, predc, sizeof(predc)); }
    }
    mb->chroma_mode = (uint8_t)cbest;
    int cbpc = code_chroma(e, mx, my, bestc, 1, lv, mb->tc, 1);
    mb->cbp = (uint8_t)((any_ac ? 15 : 0) | (cbpc << 4));
    if (mb_bits_bound(lv, 1) > MB_BITS_LIMIT) make_pcm(e, mx, my);
}

/* ------------------------------------------------------------ motion search */
typedef struct { int16_t x, y; } mv_t;

static inline int refpx(const uint8_t *ref, int stride, int w, int h, int x, int y)
{
    return ref[clip3(0, h - 1, y) * stride + clip3(0, w - 1, x)];
}

static uint8_t *const *ref_planes(h264o_enc *e, int ref_idx) { return ref_idx == 0 ? e->ref : e->older[ref_idx - 1]; }
/* bits of ref_idx_l0, te(v) (9.1): none with one active picture, one bit with two, ue(v) above */
static int ref_idx_bits(int ref_idx, int active) { return active <= 1 ? 0 : active == 2 ? 1 : (ref_idx == 0 ? 1 : 3); }

/* does the residual of this MB against the prediction at (mvx, mvy) quantise to nothing? */
static int mv_all_zero(h264o_enc *e, int mx, int my, int mvx, int mvy)
{
    int qp = e->cfg.qp, cw = e->cw, ch = e->ch, cs = cw / 2, qpc = o_chroma_qp[qp];
    int16_t lvz[16], deq[16];
    uint8_t py[256], pc[64];
    h264o_mc_luma(e->ref[0], cw, cw, ch, 16 * mx, 16 * my, mvx, mvy, 16, 16, py, 16);
    for (int b = 0; b < 16; b++) {
        int off = (16 * my + o_blk_y[b] * 4) * cw + 16 * mx + o_blk_x[b] * 4;
        if (tq_block(e->src[0] + off, cw, py + o_blk_y[b] * 4 * 16 + o_blk_x[b] * 4, 16, qp, 0, 0, lvz, deq, NULL)) return 0;
    }
    int qbits = 15 + qpc / 6, f = (1 << qbits) / 6;
    for (int pl = 0; pl < 2; pl++) {
        int16_t dc[4];
        h264o_mc_chroma(e->ref[1 + pl], cs, cs, ch / 2, 8 * mx, 8 * my, mvx, mvy, 8, 8, pc, 8);
        for (int b = 0; b < 4; b++) {
            int off = (8 * my + (b >> 1) * 4) * cs + 8 * mx + (b & 1) * 4;
            if (tq_block(e->src[1 + pl] + off, cs, pc + (b >> 1) * 4 * 8 + (b & 1) * 4, 8, qpc, 0, 1, lvz, deq, &dc[b])) return 0;
        }
        int fd[4] = {dc[0] + dc[1] + dc[2] + dc[3], dc[0] - dc[1] + dc[2] - dc[3],
                     dc[0] + dc[1] - dc[2] - dc[3], dc[0] - dc[1] - dc[2] + dc[3]};
        for (int i = 0; i < 4; i++)
            if ((((int64_t)abs(fd[i]) * o_quant_mf[qpc % 6][0] + 2 * (int64_t)f) >> (qbits + 1)) != 0) return 0;
    }
    return 1;
}

/* half-sample planes (8.4.2.2.1: G integer, b, h, j) on an 18x18 grid whose origin is one sample up-left of the integer
 * search's winner (ix, iy): every quarter-sample position within +-3 quarter samples of it is two taps away */
enum { ME_GS = 18 };
typedef struct { uint8_t G[ME_GS][ME_GS], B[ME_GS][ME_GS], H[ME_GS][ME_GS], J[ME_GS][ME_GS]; int ix, iy; } hp_planes;

/* prediction of the w x h rectangle at (x0, y0) of the macroblock for the vector (qx, qy) (quarter samples), into pred (pitch 16) */
static void hp_pred(const hp_planes *P, int qx, int qy, int x0, int y0, int w, int h, uint8_t *pred)
{
    int ox = qx - 4 * P->ix, oy = qy - 4 * P->iy; /* -3..3 relative to the integer winner */
    int gx = 1 + (ox >> 2), gy = 1 + (oy >> 2), fx = ox & 3, fy = oy & 3;
    for (int y = y0; y < y0 + h; y++)
        for (int x = x0; x < x0 + w; x++) {
            int X = gx + x, Y = gy + y, v;
            if (fy == 0) v = fx == 0 ? P->G[Y][X] : fx == 2 ? P->B[Y][X] : fx == 1 ? (P->G[Y][X] + P->B[Y][X] + 1) >> 1 : (P->G[Y][X + 1] + P->B[Y][X] + 1) >> 1;
            else if (fx == 0) v = fy == 2 ? P->H[Y][X] : fy == 1 ? (P->G[Y][X] + P->H[Y][X] + 1) >> 1 : (P->G[Y + 1][X] + P->H[Y][X] + 1) >> 1;
            else if (fx == 2 && fy == 2) v = P->J[Y][X];
            else if (fx == 2) v = ((fy == 1 ? P->B[Y][X] : P->B[Y + 1][X]) + P->J[Y][X] + 1) >> 1;
            else if (fy == 2) v = ((fx == 1 ? P->H[Y][X] : P->H[Y][X + 1]) + P->J[Y][X] + 1) >> 1;
            else v = ((fy == 1 ? P->B[Y][X] : P->B[Y + 1][X]) + (fx == 1 ? P->H[Y][X] : P->H[Y][X + 1]) + 1) >> 1;
            pred[16 * y + x] = (uint8_t)v;
        }
}

/* half- then quarter-sample refinement of one partition (rectangle x0, y0, w, h of the macroblock) starting at the integer
 * winner: SATD + lambda * bits(mv - pmv); the centre wins a tie, then the neighbours in the order below */
static int refine_part(const hp_planes *P, const uint8_t *s, int cw, int x0, int y0, int w, int h, mv_t pmv, int lambda, mv_t *out)
{
    static const int8_t nb[8][2] = {{-1, -1}, {0, -1}, {1, -1}, {-1, 0}, {1, 0}, {-1, 1}, {0, 1}, {1, 1}};
    int cx = 4 * P->ix, cy = 4 * P->iy, best_cost = 0;
    uint8_t pred[256];
    for (int pass = 0; pass < 2; pass++) {
        int step = pass == 0 ? 2 : 1, bcx = cx, bcy = cy;
        for (int k = (pass == 0 ? -1 : 0); k < 8; k++) {
            int qx = k < 0 ? cx : cx + step * nb[k][0], qy = k < 0 ? cy : cy + step * nb[k][1];
            hp_pred(P, qx, qy, x0, y0, w, h, pred);
            int cost = h264o_satd_rect(s + y0 * cw + x0, cw, pred + 16 * y0 + x0, 16, w, h) + lambda * (se_len(qx - pmv.x) + se_len(qy - pmv.y));
            if (k < 0 || cost < best_cost) { best_cost = cost; bcx = qx; bcy = qy; }
        }
        cx = bcx;
        cy = bcy;
    }
    out->x = (int16_t)cx;
    out->y = (int16_t)cy;
    return best_cost;
}

/* pmv: the vector this macroblock had in the previous picture (0 after an IDR) - the stand-in for the motion
 * vector predictor in the rate term: it is final before the picture starts, so every macroblock stays independent.
 * Returns the motion cost; qmv[4] receives the vectors of the four 8x8 quadrants, *shape the partitioning (0 16x16, 1 16x8,
 * 2 8x16, 3 8x8).  A macroblock whose 16x16 cost reaches PART_TEST_MIN is also costed as two 16x8, two 8x16 and four 8x8
 * partitions, each refined on its own from the same integer winner; a split pays lambda * (its extra mb_type / sub_mb_type
 * bits + one more ref_idx_l0 per further partition) and must be strictly cheaper than the larger partitions before it. */
enum { PART_TEST_MIN = 2000 };
static int motion_search(h264o_enc *e, int mx, int my, mv_t pmv, const uint8_t *refy, int refbits, mv_t qmv[4], int *shape)
{
    int cw = e->cw, ch = e->ch, lambda = o_lambda[e->cfg.qp];
    const uint8_t *s = e->src[0] + (16 * my) * cw + 16 * mx;
    int bx = 16 * mx, by = 16 * my;
    /* clamped 54x54 window: integer range [-16,15] + 16 + 3-tap apron each side */
    enum { R = 16, AP = 4, WS = 16 + 2 * R + 2 * AP, GS = ME_GS };
    static __thread uint8_t win[WS * WS];
    for (int y = 0; y < WS; y++)
        for (int x = 0; x < WS; x++) win[y * WS + x] = (uint8_t)refpx(refy, cw, cw, ch, bx - R - AP + x, by - R - AP + y);
    uint32_t best_key = 0xFFFFFFFFu;
    int seeded = 0;
    if (e->cfg.search == 1) {
        /* seeded search: the previous picture's vector of this macroblock, rounded to integer samples (the same rounding as
         * the "nothing left to code" test), against its eight integer neighbours, on the exhaustive pass's own key (SAD +
         * lambda * bits(mv - pmv), then the candidate index).  A strict local minimum inside the search range is taken as the
         * integer winner; anything else falls through to the exhaustive pass, whose result is then what it always was. */
        const int sx = (pmv.x + 2) >> 2, sy = (pmv.y + 2) >> 2;
        if (sx >= -R && sx < R && sy >= -R && sy < R) {
            uint32_t ckey = 0, nmin = 0xFFFFFFFFu;
            for (int dy = sy - 1; dy <= sy + 1; dy++)
                for (int dx = sx - 1; dx <= sx + 1; dx++) {
                    if (dx < -R || dx >= R || dy < -R || dy >= R) continue;
                    int sad = h264o_sad16x16(s, cw, win + (dy + R + AP) * WS + dx + R + AP, WS);
                    uint32_t cost = (uint32_t)(sad + lambda * (se_len(4 * dx - pmv.x) + se_len(4 * dy - pmv.y)));
                    uint32_t key = (cost << 10) | (uint32_t)(((dy + R) << 5) | (dx + R));
                    if (dx == sx && dy == sy) ckey = key;
                    else if (key < nmin) nmin = key;
                }
            if (ckey < nmin) { best_key = ckey; seeded = 1; }
        }
    }
    for (int dy = -R; dy < R && !seeded; dy++)
        for (int dx = -R; dx < R; dx++) {
            int sad = h264o_sad16x16(s, cw, win + (dy + R + AP) * WS + dx + R + AP, WS);
            uint32_t cost = (uint32_t)(sad + lambda * (se_len(4 * dx - pmv.x) + se_len(4 * dy - pmv.y)));
            uint32_t key = (cost << 10) | (uint32_t)(((dy + R) << 5) | (dx + R));
            if (key < best_key) best_key = key;
        }
    int idx = best_key & 1023, ix = (idx & 31) - R, iy = (idx >> 5) - R;
    static __thread hp_planes P;
    int b1[GS + 5][GS];
    const uint8_t *o = win + (iy + R + AP - 1) * WS + ix + R + AP - 1; /* grid (0,0) */
    P.ix = ix;
    P.iy = iy;
#define WP(x, y) o[(y) * WS + (x)]
    for (int y = -2; y < GS + 3; y++)
        for (int x = 0; x < GS; x++)
            b1[y + 2][x] = WP(x - 2, y) - 5 * WP(x - 1, y) + 20 * WP(x, y) + 20 * WP(x + 1, y) - 5 * WP(x + 2, y) + WP(x + 3, y);
    for (int y = 0; y < GS; y++)
        for (int x = 0; x < GS; x++) {
            P.G[y][x] = WP(x, y);
            P.B[y][x] = clip1((b1[y + 2][x] + 16) >> 5);
            P.H[y][x] = clip1((WP(x, y - 2) - 5 * WP(x, y - 1) + 20 * WP(x, y) + 20 * WP(x, y + 1) - 5 * WP(x, y + 2) + WP(x, y + 3) + 16) >> 5);
            P.J[y][x] = clip1((b1[y][x] - 5 * b1[y + 1][x] + 20 * b1[y + 2][x] + 20 * b1[y + 3][x] - 5 * b1[y + 4][x] + b1[y + 5][x] + 512) >> 10);
        }
#undef WP
    mv_t m16;
    int best = refine_part(&P, s, cw, 0, 0, 16, 16, pmv, lambda, &m16);
    *shape = 0;
    for (int q = 0; q < 4; q++) qmv[q] = m16;
    if (best >= PART_TEST_MIN) {
        mv_t t[4];
        int c = refine_part(&P, s, cw, 0, 0, 16, 8, pmv, lambda, &t[0]) + refine_part(&P, s, cw, 0, 8, 16, 8, pmv, lambda, &t[2]) + lambda * (2 + refbits);
        if (c < best) { best = c; *shape = 1; qmv[0] = qmv[1] = t[0]; qmv[2] = qmv[3] = t[2]; }
        c = refine_part(&P, s, cw, 0, 0, 8, 16, pmv, lambda, &t[0]) + refine_part(&P, s, cw, 8, 0, 8, 16, pmv, lambda, &t[1]) + lambda * (2 + refbits);
        if (c < best) { best = c; *shape = 2; qmv[0] = qmv[2] = t[0]; qmv[1] = qmv[3] = t[1]; }
        c = lambda * (8 + 3 * refbits);
        for (int q = 0; q < 4; q++) c += refine_part(&P, s, cw, (q & 1) * 8, (q >> 1) * 8, 8, 8, pmv, lambda, &t[q]);
        if (c < best) { best = c; *shape = 3; for (int q = 0; q < 4; q++) qmv[q] = t[q]; }
    }
    return best;
}

/* Intra16x16 cost estimate of a P macroblock from the SOURCE picture's own neighbouring samples (final before the picture
 * starts): SATD of the vertical / horizontal / DC prediction built from the source row above and column to the left, the
 * cheapest of those available (6.4.4 availability, as the real prediction), plus lambda * 8 for the mode bits */
static int intra_estimate(const h264o_enc *e, int mx, int my)
{
    int cw = e->cw, lambda = o_lambda[e->cfg.qp];
    const uint8_t *s = e->src[0] + (16 * my) * cw + 16 * mx;
    int top = top_in_slice(e, my), left = mx > 0;
    uint8_t pred[256];
    int best = -1;
    for (int mode = 0; mode < 3; mode++) {
        if (mode == 0 && !top) continue;
        if (mode == 1 && !left) continue;
        if (mode == 0) for (int y = 0; y < 16; y++) memcpy(pred + 16 * y, s - cw, 16);
        else if (mode == 1) for (int y = 0; y < 16; y++) memset(pred + 16 * y, s[y * cw - 1], 16);
        else {
            int sum = 0, dc;
            if (top) for (int x = 0; x < 16; x++) sum += s[x - cw];
            if (left) for (int y = 0; y < 16; y++) sum += s[y * cw - 1];
            dc = (top && left) ? (sum + 16) >> 5 : (top || left) ? (sum + 8) >> 4 : 128;
            memset(pred, dc, 256);
        }
        int c = h264o_satd16x16(s, cw, pred, 16);
        if (best < 0 || c < best) best = c;
    }
    return best + 8 * lambda;
}

/* 8.4.1.3 motion vector prediction.  Vectors are kept per 8x8 quadrant (mvq), which is the granularity of the smallest
 * partition this encoder produces: a neighbouring partition is looked up as the quadrant (qx, qy) of its macroblock. */
static void neighbour_q(const h264o_enc *e, int mx, int my, int qx, int qy, int cur_my, int *avail, int *ref, mv_t *mv)
{
    *avail = mx >= 0 && mx < e->mbw && my >= cur_my - cur_my % e->slice_rows;   /* inside the picture and the current slice */
    *ref = -1;
    mv->x = mv->y = 0;
    if (!*avail) return;
    const h264o_mbinfo *m = &e->mb[my * e->mbw + mx];
    if (!H264O_MB_IS_INTRA(m->type)) {   /* (ref_idx_l0 rides in chroma_mode: one reference per macroblock) */
        const int16_t *v = e->mvq + (size_t)(my * e->mbw + mx) * 8 + 2 * (2 * qy + qx);
        *ref = m->chroma_mode;
        mv->x = v[0];
        mv->y = v[1];
    }
}
static int med3(int a, int b, int c) { return a > b ? (b > c ? b : (a > c ? c : a)) : (a > c ? a : (b > c ? c : b)); }

/* predictor of the partition that covers quadrants x0 .. x0 + w - 1, y0 .. y0 + h - 1 (w, h in 1, 2) of macroblock (mx, my)
 * with reference index `ref`: neighbours A (left of its top-left sample), B (above it), C (above-right of its top-right
 * sample, or D above-left when C is not available: outside, or later in decoding order) - 6.4.11.7; directional rules of
 * 8.4.1.3 for 16x8 and 8x16; skip_mv (16x16 geometry only) receives the P_Skip vector (8.4.1.1) */
static mv_t predict_mv_part(const h264o_enc *e, int mx, int my, int x0, int y0, int w, int h, int ref, mv_t *skip_mv)
{
    int aA, aB, aC, rA, rB, rC;
    mv_t A, B, C;
    if (x0 == 0) neighbour_q(e, mx - 1, my, 1, y0, my, &aA, &rA, &A);
    else neighbour_q(e, mx, my, 0, y0, my, &aA, &rA, &A);
    if (y0 == 0) neighbour_q(e, mx, my - 1, x0, 1, my, &aB, &rB, &B);
    else neighbour_q(e, mx, my, x0, 0, my, &aB, &rB, &B);
    if (y0 == 0) {
        if (x0 + w <= 1) neighbour_q(e, mx, my - 1, x0 + w, 1, my, &aC, &rC, &C);
        else neighbour_q(e, mx + 1, my - 1, 0, 1, my, &aC, &rC, &C);
    } else if (x0 + w <= 1) neighbour_q(e, mx, my, x0 + w, 0, my, &aC, &rC, &C);   /* the quadrant above-right, coded before this one */
    else { aC = 0; rC = -1; C.x = C.y = 0; }                                        /* in the macroblock to the right: not yet coded */
    if (!aC) {
        if (x0 == 0 && y0 == 0) neighbour_q(e, mx - 1, my - 1, 1, 1, my, &aC, &rC, &C);
        else if (y0 == 0) neighbour_q(e, mx, my - 1, 0, 1, my, &aC, &rC, &C);
        else if (x0 == 0) neighbour_q(e, mx - 1, my, 1, 0, my, &aC, &rC, &C);
        else neighbour_q(e, mx, my, 0, 0, my, &aC, &rC, &C);
    }
    int zero_skip = !aA || !aB || (rA == 0 && A.x == 0 && A.y == 0) || (rB == 0 && B.x == 0 && B.y == 0);
    if (w == 2 && h == 1) {   /* 16x8: the upper partition takes B, the lower one A, when that neighbour uses the same picture */
        if (y0 == 0 && rB == ref) return B;
        if (y0 == 1 && rA == ref) return A;
    } else if (w == 1 && h == 2) {   /* 8x16: left A, right C */
        if (x0 == 0 && rA == ref) return A;
        if (x0 == 1 && rC == ref) return C;
    }
    if (!aB && !aC && aA) { B = A; C = A; rB = rA; rC = rA; }
    mv_t p;
    int n = (rA == ref) + (rB == ref) + (rC == ref);
    if (n == 1) p = rA == ref ? A : rB == ref ? B : C;
    else { p.x = (int16_t)med3(A.x, B.x, C.x); p.y = (int16_t)med3(A.y, B.y, C.y); }
    if (skip_mv) {
        if (zero_skip) skip_mv->x = skip_mv->y = 0;
        else if (ref == 0) *skip_mv = p;
        else {
            int n0 = (rA == 0) + (rB == 0) + (rC == 0);
            if (n0 == 1) *skip_mv = rA == 0 ? A : rB == 0 ? B : C;
            else { skip_mv->x = (int16_t)med3(A.x, B.x, C.x); skip_mv->y = (int16_t)med3(A.y, B.y, C.y); }
        }
    }
    return p;
}
static mv_t predict_mv(const h264o_enc *e, int mx, int my, mv_t *skip_mv)
{
    return predict_mv_part(e, mx, my, 0, 0, 2, 2, e->mb[my * e->mbw + mx].chroma_mode, skip_mv);   /* 16x16, the macroblock's own ref_idx_l0 */
}
/* partition k of a macroblock of shape 1 (16x8), 2 (8x16), 3 (8x8): its rectangle in quadrant units */
static void part_rect(int shape, int k, int *x0, int *y0, int *w, int *h)
{
    if (shape == 1) { *x0 = 0; *y0 = k; *w = 2; *h = 1; }
    else if (shape == 2) { *x0 = k; *y0 = 0; *w = 1; *h = 2; }
    else { *x0 = k & 1; *y0 = k >> 1; *w = 1; *h = 1; }
}

static void encode_inter_mb(h264o_enc *e, int mx, int my)
{
    int qp = e->cfg.qp, cw = e->cw, cs = cw / 2, ch = e->ch;
    h264o_mbinfo *mb = &e->mb[my * e->mbw + mx];
    int16_t *lv = e->levels + (size_t)(my * e->mbw + mx) * H264O_LV_STRIDE;
    memset(lv, 0, H264O_LV_STRIDE * sizeof(int16_t));
    uint8_t pred[256], predc[2][64];
    const int refi = mb->chroma_mode;   /* ref_idx_l0, set by the motion search */
    uint8_t *const *rp = ref_planes(e, refi);
    /* motion compensation quadrant by quadrant (a partition larger than 8x8 carries its vector in each of its quadrants:
     * prediction is a per-sample function of the vector, so this equals predicting the partition in one piece) */
    const int16_t *qv = e->mvq + (size_t)(my * e->mbw + mx) * 8;
    const int shape = e->pshape[my * e->mbw + mx];
    for (int q = 0; q < 4; q++) {
        int qx = (q & 1) * 8, qy = (q >> 1) * 8;
        h264o_mc_luma(rp[0], cw, cw, ch, 16 * mx + qx, 16 * my + qy, qv[2 * q], qv[2 * q + 1], 8, 8, pred + 16 * qy + qx, 16);
        for (int pl = 0; pl < 2; pl++)
            h264o_mc_chroma(rp[1 + pl], cs, cs, ch / 2, 8 * mx + qx / 2, 8 * my + qy / 2, qv[2 * q], qv[2 * q + 1], 4, 4, predc[pl] + 8 * (qy / 2) + qx / 2, 8);
    }
    const uint8_t *s = e->src[0] + (16 * my) * cw + 16 * mx;
    uint8_t *r = e->rec[0] + (16 * my) * cw + 16 * mx;
    int cbp = 0;
    for (int y = 0; y < 16; y++) memcpy(r + y * cw, pred + 16 * y, 16);
    if (e->want_intra[my * e->mbw + mx] == 2) {   /* nothing left to code: prediction = reconstruction */
        for (int pl = 0; pl < 2; pl++)
            for (int y = 0; y < 8; y++) memcpy(e->rec[1 + pl] + (8 * my + y) * cs + 8 * mx, predc[pl] + 8 * y, 8);
        mb->cbp = 0;
        mv_t skip0;
        predict_mv(e, mx, my, &skip0);
        mb->type = (refi == 0 && skip0.x == mb->mvx && skip0.y == mb->mvy) ? H264O_MB_PSKIP : H264O_MB_P16;
        mb->i16_mode = 0;
        return;
    }
    if (e->cfg.profile_idc == 100) {
        /* High profile: the luma residual of an inter macroblock goes through the 8x8 transform.  CAVLC carries an 8x8 block
         * as four interleaved 4x4 lists (7.3.5.3.2): level i of list k = level 4 i + k of the 8x8 zig-zag scan, stored here
         * where the 4x4 lists of the quadrant live, so that TotalCoeff / nC and the entropy coder need no special case */
        for (int b8 = 0; b8 < 4; b8++) {
            int bx = (b8 & 1) * 8, by = (b8 >> 1) * 8;
            int16_t d[64], l8[64];
            int32_t w8[64], dq8[64];
            for (int y = 0; y < 8; y++)
                for (int x = 0; x < 8; x++) d[8 * y + x] = (int16_t)(s[(by + y) * cw + bx + x] - pred[(by + y) * 16 + bx + x]);
            h264o_fdct8x8(d, w8);
            h264o_quant8x8(w8, qp, 0, l8);
            int any = 0;
            for (int k = 0; k < 4; k++) {
                int nnz = 0;
                for (int i = 0; i < 16; i++) {
                    int16_t v = l8[o_zigzag8x8[4 * i + k]];
                    lv[H264O_LV_LUMA + (4 * b8 + k) * 16 + i] = v;
                    nnz += v != 0;
                }
                mb->tc[4 * b8 + k] = (uint8_t)nnz;
                any |= nnz;
            }
            if (any) {
                cbp |= 1 << b8;
                h264o_dequant8x8(l8, qp, dq8);
                h264o_idct8x8_add(dq8, r + by * cw + bx, cw);
            }
        }
    } else
    for (int b = 0; b < 16; b++) {
        int bx = o_blk_x[b] * 4, by = o_blk_y[b] * 4;
        int16_t deq[16];
        int nnz = tq_block(s + by * cw + bx, cw, pred + by * 16 + bx, 16, qp, 0, 0, lv + H264O_LV_LUMA + b * 16, deq, NULL);
        mb->tc[b] = (uint8_t)nnz;
        if (nnz) {
            cbp |= 1 << (b >> 2);
            h264o_idct4x4_add(deq, r + by * cw + bx, cw);
        }
    }
    int cbpc = code_chroma(e, mx, my, predc, 0, lv, mb->tc, 1);
    mb->cbp = (uint8_t)(cbp | (cbpc << 4));
    if (mb_bits_bound(lv, 0) > MB_BITS_LIMIT) { make_pcm(e, mx, my); return; }
    if (shape) mb->type = (uint8_t)(H264O_MB_P16X8 + shape - 1);
    else {
        mv_t skip;
        predict_mv(e, mx, my, &skip);
        mb->type = (mb->cbp == 0 && refi == 0 && skip.x == mb->mvx && skip.y == mb->mvy) ? H264O_MB_PSKIP : H264O_MB_P16;
    }
    mb->i16_mode = (e->cfg.profile_idc == 100 && (mb->cbp & 15)) ? 1 : 0;   /* transform_size_8x8_flag (sent only with luma coefficients) */
}
/* the part of encode_inter_mb that needs every macroblock's FINAL type (an I_PCM / intra neighbour is not a vector): run
 * after all macroblocks of the picture are coded */
static void finish_inter_mb(h264o_enc *e, int mx, int my)
{
    h264o_mbinfo *mb = &e->mb[my * e->mbw + mx];
    if (mb->type != H264O_MB_P16 && mb->type != H264O_MB_PSKIP) return;
    mv_t skip;
    predict_mv(e, mx, my, &skip);
    mb->type = (mb->cbp == 0 && mb->chroma_mode == 0 && skip.x == mb->mvx && skip.y == mb->mvy) ? H264O_MB_PSKIP : H264O_MB_P16;
}

/* ------------------------------------------------------------ slice data 7.3.4/7.3.5 */
/* 6.4.9: macroblock (nx, ny), a neighbour of (mx, my) to the left or in the row above, is available - inside the picture, in
 * the same slice, earlier in decoding order.  The encoder's slices are bands of whole rows; the random-stream generator may
 * cut slices anywhere (slice_first) */
static int mb_avail(const h264o_enc *e, int mx, int my, int nx, int ny)
{
    if (nx < 0 || nx >= e->mbw || ny < 0) return 0;
    if (e->rs && e->rs->slice_first) return ny * e->mbw + nx >= e->rs->slice_first[my * e->mbw + mx];
    return ny == my ? 1 : top_in_slice(e, my);
}
/* ... and usable for intra prediction and for the derivation of Intra4x4PredMode (8.3.1.1): with constrained_intra_pred_flag
 * (the generator only) a macroblock coded in Inter prediction mode is not */
static int intra_avail(const h264o_enc *e, int mx, int my, int nx, int ny)
{
    if (!mb_avail(e, mx, my, nx, ny)) return 0;
    return !(e->rs && e->rs->constrained) || H264O_MB_IS_INTRA(e->mb[ny * e->mbw + nx].type);
}
static int nc_luma(const h264o_enc *e, int mx, int my, int b)
{
    int x = o_blk_x[b], y = o_blk_y[b], nA = -1, nB = -1;
    const h264o_mbinfo *m = &e->mb[my * e->mbw + mx];
    if (x > 0) nA = m->tc[xy2blk[4 * y + x - 1]];
    else if (mb_avail(e, mx, my, mx - 1, my)) nA = (m - 1)->tc[xy2blk[4 * y + 3]];
    if (y > 0) nB = m->tc[xy2blk[4 * (y - 1) + x]];
    else if (mb_avail(e, mx, my, mx, my - 1)) nB = (m - e->mbw)->tc[xy2blk[12 + x]];
    if (nA >= 0 && nB >= 0) return (nA + nB + 1) >> 1;
    return nA >= 0 ? nA : nB >= 0 ? nB : 0;
}
static int nc_chroma(const h264o_enc *e, int mx, int my, int pl, int b)
{
    int x = b & 1, y = b >> 1, nA = -1, nB = -1, base = 16 + pl * 4;
    const h264o_mbinfo *m = &e->mb[my * e->mbw + mx];
    if (x > 0) nA = m->tc[base + 2 * y];
    else if (mb_avail(e, mx, my, mx - 1, my)) nA = (m - 1)->tc[base + 2 * y + 1];
    if (y > 0) nB = m->tc[base + x];
    else if (mb_avail(e, mx, my, mx, my - 1)) nB = (m - e->mbw)->tc[base + 2 + x];
    if (nA >= 0 && nB >= 0) return (nA + nB + 1) >> 1;
    return nA >= 0 ? nA : nB >= 0 ? nB : 0;
}

static void write_mb(h264o_enc *e, bitw *b, int mx, int my, int p_slice)
{
    const h264o_mbinfo *mb = &e->mb[my * e->mbw + mx];
    const int16_t *lv = e->levels + (size_t)(my * e->mbw + mx) * H264O_LV_STRIDE;
    int cbpl = mb->cbp & 15, cbpc = mb->cbp >> 4;
    const int qpd = e->rs ? e->rs->qpd[my * e->mbw + mx] : 0;   /* mb_qp_delta: 0 for every macroblock the encoder codes */
    if (mb->type == H264O_MB_IPCM) {   /* 7.3.5: mb_type I_PCM, alignment, 256 + 2 x 64 samples */
        int cw = e->cw, cs = cw / 2;
        bw_ue(b, (uint32_t)(p_slice ? 5 + 25 : 25));
        while (b->bits & 7) bw_put(b, 1, 0);
        for (int y = 0; y < 16; y++)
            for (int x = 0; x < 16; x++) bw_put(b, 8, e->src[0][(16 * my + y) * cw + 16 * mx + x]);
        for (int pl = 1; pl < 3; pl++)
            for (int y = 0; y < 8; y++)
                for (int x = 0; x < 8; x++) bw_put(b, 8, e->src[pl][(8 * my + y) * cs + 8 * mx + x]);
        return;
    }
    if (mb->type == H264O_MB_I4) {   /* I_NxN: 7.3.5.1 mb_pred with the sixteen Intra4x4 modes, then the residual by coded_block_pattern */
        const uint8_t *am = e->aux + (size_t)(my * e->mbw + mx) * 16;
        bw_ue(b, (uint32_t)(p_slice ? 5 : 0));
        if (e->cfg.profile_idc == 100) bw_put(b, 1, 0);   /* transform_size_8x8_flag: Intra4x4, not Intra8x8 */
        for (int k = 0; k < 16; k++) {   /* 8.3.1.1: predicted mode = the smaller of the left and upper blocks' modes */
            int x = o_blk_x[k], y = o_blk_y[k], mA, mB, dc_only = 0;
            if (x > 0) mA = am[xy2blk[4 * y + x - 1]];
            else if (!intra_avail(e, mx, my, mx - 1, my)) { dc_only = 1; mA = 2; }
            else mA = (mb - 1)->type == H264O_MB_I4 ? (am - 16)[xy2blk[4 * y + 3]] : 2;
            if (y > 0) mB = am[xy2blk[4 * (y - 1) + x]];
            else if (!intra_avail(e, mx, my, mx, my - 1)) { dc_only = 1; mB = 2; }
            else mB = (mb - e->mbw)->type == H264O_MB_I4 ? (am - 16 * e->mbw)[xy2blk[12 + x]] : 2;
            int pm = dc_only ? 2 : (mA < mB ? mA : mB), m = am[k];
            if (m == pm) bw_put(b, 1, 1);
            else { bw_put(b, 1, 0); bw_put(b, 3, (uint32_t)(m < pm ? m : m - 1)); }
        }
        bw_ue(b, mb->chroma_mode);
        int code = 0;
        while (o_cbp_code2intra[code] != mb->cbp) code++;
        bw_ue(b, (uint32_t)code);
        if (mb->cbp) bw_se(b, qpd); /* mb_qp_delta */
    } else
    if (mb->type == H264O_MB_I16) {
        int t = 1 + mb->i16_mode + 4 * cbpc + (cbpl ? 12 : 0);
        bw_ue(b, (uint32_t)(p_slice ? 5 + t : t));
        bw_ue(b, mb->chroma_mode);
        bw_se(b, qpd); /* mb_qp_delta */
        cavlc_block(b, lv + H264O_LV_LUMA_DC, 16, nc_luma(e, mx, my, 0));
    } else {
        /* 7.3.5.1 / 7.3.5.2: mb_type (P_L0_16x16, P_L0_L0_16x8, P_L0_L0_8x16, P_8x8 with four sub_mb_type P_L0_8x8), then
         * every partition's ref_idx_l0, then every partition's mvd_l0 */
        const int shape = mb->type >= H264O_MB_P16X8 ? mb->type - H264O_MB_P16X8 + 1 : 0, nparts = shape == 0 ? 1 : shape == 3 ? 4 : 2;
        const int16_t *qv = e->mvq + (size_t)(my * e->mbw + mx) * 8;
        bw_ue(b, (uint32_t)shape);
        if (e->rs && e->rs->direct) {   /* random-stream generator, every inter macroblock of the picture */
            uint32_t *rng = &e->rs->rng;
            int sub[4] = {0, 0, 0, 0}, all8x8 = 1;
            if (shape == 3)
                for (int k = 0; k < 4; k++) {
                    sub[k] = rs_below(rng, 3) ? rs_below(rng, 4) : 0;   /* sub_mb_type: P_L0_8x8, 8x4, 4x8, 4x4 */
                    if (sub[k]) all8x8 = 0;
                    bw_ue(b, (uint32_t)sub[k]);
                }
            for (int k = 0; k < nparts; k++) {
                int r = rs_below(rng, e->avail_refs);
                if (e->avail_refs == 2) bw_put(b, 1, r ? 0 : 1);
                else if (e->avail_refs > 2) bw_ue(b, (uint32_t)r);
            }
            for (int k = 0; k < nparts; k++) {
                int nsub = shape == 3 ? (sub[k] == 0 ? 1 : sub[k] == 3 ? 4 : 2) : 1;
                for (int j = 0; j < nsub; j++)
                    for (int c = 0; c < 2; c++)
                        bw_se(b, rs_below(rng, 8) ? rs_below(rng, 13) - 6 : rs_below(rng, 10) ? rs_below(rng, 161) - 80 : rs_below(rng, 1201) - 600);
            }
            int code = 0;
            while (o_cbp_code2inter[code] != mb->cbp) code++;
            bw_ue(b, (uint32_t)code);
            if (e->cfg.profile_idc == 100 && (mb->cbp & 15) && all8x8) bw_put(b, 1, mb->i16_mode);
            if (mb->cbp) bw_se(b, qpd);
            goto residual;
        }
        if (shape == 3) for (int k = 0; k < 4; k++) bw_ue(b, 0);
        for (int k = 0; k < nparts; k++) {
            if (e->avail_refs == 2) bw_put(b, 1, mb->chroma_mode ? 0 : 1);        /* ref_idx_l0, te(v) with cMax 1: the inverted bit */
            else if (e->avail_refs > 2) bw_ue(b, mb->chroma_mode);
        }
        for (int k = 0; k < nparts; k++) {
            int x0 = 0, y0 = 0, w = 2, h = 2;
            if (shape) part_rect(shape, k, &x0, &y0, &w, &h);
            mv_t p = predict_mv_part(e, mx, my, x0, y0, w, h, mb->chroma_mode, NULL);
            bw_se(b, qv[2 * (2 * y0 + x0)] - p.x);
            bw_se(b, qv[2 * (2 * y0 + x0) + 1] - p.y);
        }
        int code = 0;
        while (o_cbp_code2inter[code] != mb->cbp) code++;
        bw_ue(b, (uint32_t)code);
        if (e->cfg.profile_idc == 100 && (mb->cbp & 15)) bw_put(b, 1, mb->i16_mode); /* transform_size_8x8_flag */
        if (mb->cbp) bw_se(b, qpd); /* mb_qp_delta */
    }
residual:
    for (int b8 = 0; b8 < 4; b8++)
        if (cbpl & (1 << b8))
            for (int k = 0; k < 4; k++) {
                int blk = 4 * b8 + k;
                if (mb->type == H264O_MB_I16) cavlc_block(b, lv + H264O_LV_LUMA + blk * 16 + 1, 15, nc_luma(e, mx, my, blk));
                else cavlc_block(b, lv + H264O_LV_LUMA + blk * 16, 16, nc_luma(e, mx, my, blk));
            }
    if (cbpc) {
        cavlc_block(b, lv + H264O_LV_CHROMA_DC, 4, -1);
        cavlc_block(b, lv + H264O_LV_CHROMA_DC + 4, 4, -1);
    }
    if (cbpc == 2)
        for (int pl = 0; pl < 2; pl++)
            for (int k = 0; k < 4; k++)
                cavlc_block(b, lv + H264O_LV_CHROMA_AC + (pl * 4 + k) * 16 + 1, 15, nc_chroma(e, mx, my, pl, k));
}

/* ------------------------------------------------------------ picture driver */
static void load_source(h264o_enc *e, const uint8_t *y, int ys, const uint8_t *u, int us, const uint8_t *v, int vs)
{
    const uint8_t *in[3] = {y, u, v};
    int st[3] = {ys, us, vs};
    for (int p = 0; p < 3; p++) {
        int w = p ? e->cfg.width / 2 : e->cfg.width, h = p ? e->cfg.height / 2 : e->cfg.height;
        int cw = p ? e->cw / 2 : e->cw, ch = p ? e->ch / 2 : e->ch;
        for (int r = 0; r < ch; r++) {
            const uint8_t *srow = in[p] + (size_t)(r < h ? r : h - 1) * st[p];
            uint8_t *d = e->src[p] + (size_t)r * cw;
            memcpy(d, srow, (size_t)w);
            for (int c = w; c < cw; c++) d[c] = srow[w - 1];
        }
    }
}

int64_t h264o_enc_encode(h264o_enc *e, const uint8_t *y, int ys, const uint8_t *u, int us,
                         const uint8_t *v, int vs, int force_idr, uint8_t *out, size_t out_cap, int *is_idr)
{
    if (!e || !y || !u || !v || !out) return -1;
    load_source(e, y, ys, u, us, v, vs);
    int idr = force_idr || e->frames == 0 || e->frame_in_gop >= e->cfg.gop;
    if (idr) { e->frame_in_gop = 0; e->frame_num = 0; }
    if (is_idr) *is_idr = idr;
    e->avail_refs = idr ? 0 : (e->frame_in_gop < e->nrefs ? e->frame_in_gop : e->nrefs);
    e->me_cost = 0;
    size_t pos = 0;
    bitw b;
    if (idr && e->band_row0 == 0) {   /* parameter sets go with the first band */
        memset(e->rbsp, 0, 256);
        b = (bitw){e->rbsp, e->rbsp_cap, 0};
        write_sps(e, &b);
        pos = emit_nal(out, out_cap, pos, 3, 7, e->rbsp, (size_t)(b.bits >> 3));
        if (pos == (size_t)-1) return -2;
        memset(e->rbsp, 0, 256);
        b = (bitw){e->rbsp, e->rbsp_cap, 0};
        write_pps(e, &b);
        pos = emit_nal(out, out_cap, pos, 3, 8, e->rbsp, (size_t)(b.bits >> 3));
        if (pos == (size_t)-1) return -2;
    }
    /* stage 1: decisions + reconstruction (pre-deblock) */
    e->any_pcm = 0;
    if (idr) {
        for (int my = e->band_row0; my < e->band_row1; my++)
            for (int mx = 0; mx < e->mbw; mx++) encode_intra_mb(e, mx, my);
    } else {
        for (int my = e->band_row0; my < e->band_row1; my++)
            for (int mx = 0; mx < e->mbw; mx++) {
                h264o_mbinfo *mb = &e->mb[my * e->mbw + mx];
                const mv_t pmv = {mb->mvx, mb->mvy};   /* previous picture's vector here (intra macroblocks carry 0) */
                const mv_t rmv = {(int16_t)(((pmv.x + 2) >> 2) * 4), (int16_t)(((pmv.y + 2) >> 2) * 4)};   /* nearest integer-sample vector */
                int16_t *qv = e->mvq + (size_t)(my * e->mbw + mx) * 8;
                memset(mb, 0, sizeof(*mb));
                memset(qv, 0, 8 * sizeof(int16_t));
                mb->type = H264O_MB_P16;
                e->want_intra[my * e->mbw + mx] = 0;
                e->pshape[my * e->mbw + mx] = 0;
                if (mv_all_zero(e, mx, my, 0, 0)) {
                    /* static: vector 0, no search */
                    e->want_intra[my * e->mbw + mx] = 2;
                } else if ((rmv.x | rmv.y) != 0 && mv_all_zero(e, mx, my, rmv.x, rmv.y)) {
                    /* scrolling: the previous vector, rounded to integer samples, predicts the macroblock completely */
                    mb->mvx = rmv.x;
                    mb->mvy = rmv.y;
                    for (int q = 0; q < 4; q++) { qv[2 * q] = rmv.x; qv[2 * q + 1] = rmv.y; }
                    e->want_intra[my * e->mbw + mx] = 2;
                } else {
                    /* every available reference picture is searched; the cheapest (motion cost + lambda * bits(ref_idx)) wins,
                     * the lower index on a tie */
                    int cost = 0;
                    for (int r = 0; r < e->avail_refs; r++) {
                        mv_t qm[4];
                        int shape = 0, rb = ref_idx_bits(r, e->avail_refs);
                        int c = motion_search(e, mx, my, pmv, ref_planes(e, r)[0], rb, qm, &shape) + o_lambda[e->cfg.qp] * rb;
                        if (r == 0 || c < cost) {
                            cost = c;
                            mb->chroma_mode = (uint8_t)r;
                            e->pshape[my * e->mbw + mx] = (uint8_t)shape;
                            for (int q = 0; q < 4; q++) { qv[2 * q] = qm[q].x; qv[2 * q + 1] = qm[q].y; }
                        }
                    }
                    e->me_cost += (uint32_t)(cost < 16383 ? cost : 16383);
                    mb->mvx = qv[0];
                    mb->mvy = qv[1];
                    /* intra or inter: decided from the source picture and the motion cost alone */
                    if (cost >= INTRA_TEST_MIN && intra_estimate(e, mx, my) < cost) {
                        e->want_intra[my * e->mbw + mx] = 1;
                        e->pshape[my * e->mbw + mx] = 0;
                        memset(qv, 0, 8 * sizeof(int16_t));
                        mb->mvx = mb->mvy = 0;
                        mb->chroma_mode = 0;
                        mb->type = H264O_MB_I16;
                    }
                }
            }
        for (int my = e->band_row0; my < e->band_row1; my++)
            for (int mx = 0; mx < e->mbw; mx++)
                if (e->want_intra[my * e->mbw + mx] != 1) encode_inter_mb(e, mx, my);
        /* intra macroblocks of the P picture: raster order, from the true reconstruction of their neighbours */
        for (int my = e->band_row0; my < e->band_row1; my++)
            for (int mx = 0; mx < e->mbw; mx++)
                if (e->want_intra[my * e->mbw + mx] == 1) encode_intra_mb(e, mx, my);
        for (int my = e->band_row0; my < e->band_row1; my++)
            for (int mx = 0; mx < e->mbw; mx++) finish_inter_mb(e, mx, my);
    }
    /* stage 2: entropy coding, one NAL unit per slice */
    e->last_slice_bits = 0;
    memset(e->rbsp, 0, e->rbsp_cap);
    for (int row0 = e->band_row0; row0 < e->band_row1; row0 += e->slice_rows) {
        int row1 = row0 + e->slice_rows < e->mbh ? row0 + e->slice_rows : e->mbh;
        b = (bitw){e->rbsp, e->rbsp_cap, 0};
        write_slice_header(e, &b, idr, row0 * e->mbw);
        uint64_t hdr_bits = b.bits;
        int skip_run = 0;
        for (int my = row0; my < row1; my++)
            for (int mx = 0; mx < e->mbw; mx++) {
                const h264o_mbinfo *mb = &e->mb[my * e->mbw + mx];
                if (!idr) {
                    if (mb->type == H264O_MB_PSKIP) { skip_run++; continue; }
                    bw_ue(&b, (uint32_t)skip_run);
                    skip_run = 0;
                }
                write_mb(e, &b, mx, my, !idr);
            }
        if (skip_run) bw_ue(&b, (uint32_t)skip_run);
        e->last_slice_bits += (int64_t)(b.bits - hdr_bits);
        bw_trailing(&b);
        if ((b.bits >> 3) > e->rbsp_cap) return -3;
        pos = emit_nal(out, out_cap, pos, idr ? 3 : 2, idr ? 5 : 1, e->rbsp, (size_t)(b.bits >> 3));
        if (pos == (size_t)-1) return -2;
        memset(e->rbsp, 0, (size_t)(b.bits >> 3) + 8);   /* the bit writer ORs into zeroed bytes */
    }
    /* stage 3: in-loop filter into the next reference */
    size_t ysz = (size_t)e->cw * e->ch;
    memcpy(e->cur[0], e->rec[0], ysz);
    memcpy(e->cur[1], e->rec[1], ysz / 4);
    memcpy(e->cur[2], e->rec[2], ysz / 4);
    if (!e->cfg.disable_deblock && !e->any_pcm)
        h264o_deblock_picture(e->cur[0], e->cur[1], e->cur[2], e->cw, e->ch, e->mb, e->mvq, e->cfg.qp, e->slice_rows < e->mbh ? e->slice_of : NULL,
                              e->band_row0, e->band_row1);
    for (int p = 0; p < 3; p++) {   /* sliding window (8.2.5.3): the new picture becomes ref_idx 0, the oldest buffer is reused */
        uint8_t *oldest = e->nrefs >= 3 ? e->older[1][p] : e->nrefs == 2 ? e->older[0][p] : e->ref[p];
        if (e->nrefs >= 3) e->older[1][p] = e->older[0][p];
        if (e->nrefs >= 2) e->older[0][p] = e->ref[p];
        e->ref[p] = e->cur[p];
        e->cur[p] = oldest;
    }
    if (idr) e->idr_id = (e->idr_id + e->idr_step) & 0xFF;
    e->frame_num = (e->frame_num + 1) & 255;
    e->frame_in_gop++;
    e->frames++;
    return (int64_t)pos;
}

/* ------------------------------------------------------------ random conforming streams (decoder-peer tests)
 * One picture of RANDOM syntax written with the writers above: macroblock types, prediction modes, vectors, reference
 * indices, coded_block_patterns, levels, I_PCM samples, and - what the encoder never produces - mb_qp_delta, slice QPs,
 * chroma_qp_index_offsets, filter offsets and every disable_deblocking_filter_idc.  Nothing is reconstructed here: the
 * stream is the test vector, oracle/h264_dec.c says what it decodes to, and the decoder peer (media_amd/csrc/h264_parse.h +
 * k_dec.h) has to produce the same samples.  Stays inside what that peer accepts (its header lists the limits): I / P
 * slices in bands of whole rows, one reference index per macroblock, sub_mb_type P_L0_8x8, no Intra8x8.
 * Conformance: a prediction mode is only chosen where its neighbours are available (8.3.1.2, 8.3.3, 8.3.4), QP_Y stays in
 * 0..51, levels are small enough for every intermediate of 8.5 to fit 16 bits at the QPs drawn (bounded below). */
/* n levels (zig-zag order) of one residual block: `density` in 1/16 of the positions non-zero, magnitudes 1 .. mag */
static int rs_levels(uint32_t *s, int16_t *lv, int first, int n, int density, int mag)
{
    int tc = 0;
    for (int i = first; i < n; i++) {
        lv[i] = 0;
        if (rs_below(s, 16) < density) {
            int a = 1 + (rs_below(s, 4) == 0 ? rs_below(s, mag > 1000 ? 9 : mag) : 0);
            if (mag > 1000 && rs_below(s, 8) == 0) a = 128 + rs_below(s, mag - 1000);   /* a level that needs more than a signed byte */
            lv[i] = (int16_t)(rs_below(s, 2) ? -a : a);
            tc++;
        }
    }
    return tc;
}

int64_t h264o_enc_random_picture(h264o_enc *e, uint32_t seed, int force_idr, int features, uint8_t *out, size_t out_cap,
                                 int *is_idr, uint8_t *mbqp_out)
{
    if (!e || !out) return -1;
    uint32_t rng = seed * 2654435761u + 12345u;
    rs_next(&rng);
    const int nmb = e->mbw * e->mbh, high = e->cfg.profile_idc == 100;
    struct randsyn rs;
    memset(&rs, 0, sizeof(rs));
    int *cqo_sticky = e->rs_cqo;   /* the PPS travels with IDR pictures only: its offsets hold until the next one */
    int idr = force_idr || e->frames == 0 || e->frame_in_gop >= e->cfg.gop;
    if (idr) { e->frame_in_gop = 0; e->frame_num = 0; }
    if (is_idr) *is_idr = idr;
    e->avail_refs = idr ? 0 : (e->frame_in_gop < e->nrefs ? e->frame_in_gop : e->nrefs);
    rs.qpd = (int8_t *)calloc((size_t)nmb, 1);
    if (idr) {
        cqo_sticky[0] = (features & 2) ? rs_below(&rng, 25) - 12 : 0;
        cqo_sticky[1] = (features & 2) && high ? rs_below(&rng, 25) - 12 : cqo_sticky[0];
    }
    rs.cqo[0] = cqo_sticky[0]; rs.cqo[1] = cqo_sticky[1];
    rs.idc = (features & 16) ? rs_below(&rng, 3) : (e->cfg.disable_deblock ? 1 : e->slice_rows < e->mbh ? 2 : 0);
    rs.direct = (features & 32) != 0;
    /* feature 128: reference list modification - up to avail_refs commands naming distinct reference pictures, never the same
     * picture as the command before (a difference of 0 cannot be written) */
    rs.ohstyle = (features & 256) != 0;
    if (idr) e->rs_constrained = (features & 1024) != 0;   /* (the PPS travels with IDR pictures only) */
    rs.constrained = e->rs_constrained;
    rs.reorder = 0; rs.nreorder = 0;
    if (rs.ohstyle && !idr) { rs.reorder = 1; rs.nreorder = 1; rs.reorder_age[0] = 1; }
    if ((features & 128) && !idr && e->frame_num >= e->avail_refs) {
        rs.reorder = 1;
        rs.nreorder = rs_below(&rng, e->avail_refs + 1);
        int used = 0;
        for (int k = 0; k < rs.nreorder; k++) {
            int age;
            do age = 1 + rs_below(&rng, e->avail_refs); while (used & (1 << age));
            used |= 1 << age;
            rs.reorder_age[k] = age;
        }
    }
    /* feature 64 (with 32: vectors are then not predicted here): slices cut at random macroblocks, up to 6 per picture */
    rs.slice_first = NULL;
    if ((features & 64) && rs.direct) {
        rs.slice_first = (int32_t *)calloc((size_t)nmb, sizeof(int32_t));
        int ncut = rs_below(&rng, 6), first = 0;
        for (int i = 0; i < nmb; i++) {
            if (i > 0 && ncut > 0 && rs_below(&rng, nmb) < ncut) first = i;
            rs.slice_first[i] = first;
        }
    }
    rs.oa = (features & 4) ? rs_below(&rng, 13) - 6 : 0;
    rs.ob = (features & 4) ? rs_below(&rng, 13) - 6 : 0;
    const int qlo = 4, qhi = 48;   /* QP_Y range drawn from: wide enough for every row of Tables 8-15 / 8-16 that filters */
    for (int i = 0; i < 256; i++) rs.slice_qp[i] = (features & 1) ? qlo + rs_below(&rng, qhi - qlo + 1) : e->cfg.qp;
    /* samples of the I_PCM macroblocks */
    for (int p = 0; p < 3; p++) {
        size_t n = (size_t)e->cw * e->ch / (p ? 4 : 1);
        for (size_t i = 0; i < n; i++) e->src[p][i] = (uint8_t)(rs_next(&rng) >> 11);
    }
    memset(e->levels, 0, (size_t)nmb * H264O_LV_STRIDE * sizeof(int16_t));
    memset(e->aux, 0, (size_t)nmb * 16);
    memset(e->mvq, 0, (size_t)nmb * 8 * sizeof(int16_t));
    int qp = 26, nslice = 0;
    e->rs = &rs;   /* (mb_avail reads the slice layout) */
    for (int my = 0; my < e->mbh; my++)
        for (int mx = 0; mx < e->mbw; mx++) {
            const int mbi = my * e->mbw + mx;
            h264o_mbinfo *mb = &e->mb[mbi];
            int16_t *lv = e->levels + (size_t)mbi * H264O_LV_STRIDE, *qv = e->mvq + (size_t)mbi * 8;
            uint8_t *am = e->aux + (size_t)mbi * 16;
            memset(mb, 0, sizeof(*mb));
            if (rs.slice_first ? rs.slice_first[mbi] == mbi : (mx == 0 && my % e->slice_rows == 0)) {   /* QP_Y,PRED at a slice start */
                nslice++;
                qp = rs.slice_qp[(rs.slice_first ? nslice - 1 : my / e->slice_rows) & 255];
            }
            const int left = intra_avail(e, mx, my, mx - 1, my), top = intra_avail(e, mx, my, mx, my - 1), topleft = intra_avail(e, mx, my, mx - 1, my - 1);
            int kind = rs_below(&rng, 100);
            /* I picture: I4 45 %, I16 45 %, I_PCM 10 %; P picture: skip 22, 16x16 22, 16x8 10, 8x16 10, 8x8 12, I16 9, I4 9, PCM 6 */
            int type;
            if (idr) type = kind < 45 ? H264O_MB_I4 : kind < 90 ? H264O_MB_I16 : H264O_MB_IPCM;
            else type = kind < 22 ? H264O_MB_PSKIP : kind < 44 ? H264O_MB_P16 : kind < 54 ? H264O_MB_P16X8 : kind < 64 ? H264O_MB_P8X16
                      : kind < 76 ? H264O_MB_P8X8 : kind < 85 ? H264O_MB_I16 : kind < 94 ? H264O_MB_I4 : H264O_MB_IPCM;
            if (type == H264O_MB_IPCM && !(features & 8)) type = H264O_MB_I16;
            mb->type = (uint8_t)type;
            /* residual density / magnitude of this macroblock (kept small at high QP: 16-bit intermediates of 8.5) */
            /* feature 512: at QP_Y <= 14 an eighth of the levels is 128 .. 427 (scaled coefficients stay below 2^15) */
            const int dens = 1 + rs_below(&rng, 6), mag = ((features & 512) && qp <= 14) ? 1300 : qp > 40 ? 2 : qp > 30 ? 4 : 9;
            if (type == H264O_MB_IPCM) {
                mb->cbp = 0x2F;
                memset(mb->tc, 16, 24);
                if (mbqp_out) mbqp_out[mbi] = 0;   /* 8.7.2.2: qP of an I_PCM macroblock */
                continue;
            }
            if (type == H264O_MB_PSKIP && rs.direct) {   /* (its vector is the decoder's business) */
                if (mbqp_out) mbqp_out[mbi] = (uint8_t)qp;
                continue;
            }
            if (type == H264O_MB_PSKIP) {
                mv_t skip;
                predict_mv_part(e, mx, my, 0, 0, 2, 2, 0, &skip);
                mb->mvx = skip.x; mb->mvy = skip.y;
                for (int q = 0; q < 4; q++) { qv[2 * q] = skip.x; qv[2 * q + 1] = skip.y; }
                if (mbqp_out) mbqp_out[mbi] = (uint8_t)qp;
                continue;
            }
            int carries_delta = 0;
            if (type == H264O_MB_I16 || type == H264O_MB_I4) {
                /* chroma: 0 DC, 1 horizontal (left), 2 vertical (top), 3 plane (left, top, top-left) */
                int cm;
                do cm = rs_below(&rng, 4); while ((cm == 1 && !left) || (cm == 2 && !top) || (cm == 3 && !(left && top && topleft)));
                mb->chroma_mode = (uint8_t)cm;
            }
            if (type == H264O_MB_I16) {
                int m;
                do m = rs_below(&rng, 4); while ((m == 0 && !top) || (m == 1 && !left) || (m == 3 && !(left && top && topleft)));
                mb->i16_mode = (uint8_t)m;
                const int cbpl = rs_below(&rng, 2) ? 15 : 0, cbpc = rs_below(&rng, 3);
                mb->cbp = (uint8_t)(cbpl | (cbpc << 4));
                rs_levels(&rng, lv + H264O_LV_LUMA_DC, 0, 16, dens + 2, mag);
                if (cbpl)
                    for (int b = 0; b < 16; b++) mb->tc[b] = (uint8_t)rs_levels(&rng, lv + H264O_LV_LUMA + b * 16, 1, 16, dens, mag);
                carries_delta = 1;
            } else if (type == H264O_MB_I4) {
                for (int k = 0; k < 16; k++) {
                    const int x = o_blk_x[k], y = o_blk_y[k];
                    const int aL = x > 0 || left, aT = y > 0 || top;
                    const int aTL = (x > 0 && y > 0) || (x > 0 && y == 0 && top) || (x == 0 && y > 0 && left) || (x == 0 && y == 0 && topleft);
                    int m;
                    for (;;) {
                        m = rs_below(&rng, 9);
                        const int need_t = m == 0 || m == 3 || m == 7, need_l = m == 1 || m == 8, need_all = m == 4 || m == 5 || m == 6;
                        if ((need_t && !aT) || (need_l && !aL) || (need_all && !(aL && aT && aTL))) continue;
                        break;
                    }
                    am[k] = (uint8_t)m;
                }
                mb->cbp = (uint8_t)(rs_below(&rng, 16) | (rs_below(&rng, 3) << 4));
            } else {
                const int shape = type == H264O_MB_P16 ? 0 : type - H264O_MB_P16X8 + 1, nparts = shape == 0 ? 1 : shape == 3 ? 4 : 2;
                mb->chroma_mode = (uint8_t)rs_below(&rng, e->avail_refs);   /* ref_idx_l0 of all its partitions */
                for (int k = 0; k < nparts && !rs.direct; k++) {
                    int x0 = 0, y0 = 0, w = 2, h = 2;
                    if (shape) part_rect(shape, k, &x0, &y0, &w, &h);
                    /* mostly near the predictor (short mvd codes), sometimes anywhere within +-40 samples, rarely far outside */
                    mv_t p = predict_mv_part(e, mx, my, x0, y0, w, h, mb->chroma_mode, NULL);
                    int vx, vy, r = rs_below(&rng, 10);
                    if (r < 5) { vx = p.x + rs_below(&rng, 9) - 4; vy = p.y + rs_below(&rng, 9) - 4; }
                    else if (r < 9) { vx = rs_below(&rng, 321) - 160; vy = rs_below(&rng, 321) - 160; }
                    else { vx = rs_below(&rng, 1601) - 800; vy = rs_below(&rng, 1601) - 800; }
                    vx = clip3(-2000, 2000, vx); vy = clip3(-2000, 2000, vy);
                    for (int qy = y0; qy < y0 + h; qy++)
                        for (int qx = x0; qx < x0 + w; qx++) { qv[2 * (2 * qy + qx)] = (int16_t)vx; qv[2 * (2 * qy + qx) + 1] = (int16_t)vy; }
                }
                mb->mvx = qv[0]; mb->mvy = qv[1];
                mb->cbp = (uint8_t)(rs_below(&rng, 3) == 0 ? 0 : (rs_below(&rng, 16) | (rs_below(&rng, 3) << 4)));
                mb->i16_mode = (uint8_t)(high && (mb->cbp & 15) ? rs_below(&rng, 2) : 0);   /* transform_size_8x8_flag */
            }
            if (type != H264O_MB_I16) {
                for (int b = 0; b < 16; b++)
                    if (mb->cbp & (1 << (b >> 2))) mb->tc[b] = (uint8_t)rs_levels(&rng, lv + H264O_LV_LUMA + b * 16, 0, 16, dens, mag);
                carries_delta = mb->cbp != 0;
            }
            if (mb->cbp >> 4) {
                rs_levels(&rng, lv + H264O_LV_CHROMA_DC, 0, 8, dens + 3, mag);
                if ((mb->cbp >> 4) == 2)
                    for (int b = 0; b < 8; b++) mb->tc[16 + b] = (uint8_t)rs_levels(&rng, lv + H264O_LV_CHROMA_AC + b * 16, 1, 16, dens, mag);
            }
            if (carries_delta && (features & 1) && rs_below(&rng, 2)) {
                int d = rs_below(&rng, 4) ? rs_below(&rng, 9) - 4 : rs_below(&rng, 52) - 26;
                const int nq = clip3(qlo, qhi, qp + d);
                d = nq - qp;
                if (d < -26) d += 52;
                if (d > 25) d -= 52;
                rs.qpd[mbi] = (int8_t)d;
                qp = nq;
            }
            if (mbqp_out) mbqp_out[mbi] = (uint8_t)qp;
        }
    /* headers and slices, as h264o_enc_encode writes them */
    size_t pos = 0;
    bitw b;
    e->rs = &rs;
    rs.rng = rng;
    if (idr) {
        memset(e->rbsp, 0, 256);
        b = (bitw){e->rbsp, e->rbsp_cap, 0};
        write_sps(e, &b);
        pos = emit_nal(out, out_cap, pos, 3, 7, e->rbsp, (size_t)(b.bits >> 3));
        memset(e->rbsp, 0, 256);
        b = (bitw){e->rbsp, e->rbsp_cap, 0};
        write_pps(e, &b);
        if (pos != (size_t)-1) pos = emit_nal(out, out_cap, pos, 3, 8, e->rbsp, (size_t)(b.bits >> 3));
    }
    memset(e->rbsp, 0, e->rbsp_cap);
    int slice_no = 0;
    for (int a0 = 0; a0 < nmb && pos != (size_t)-1; slice_no++) {
        int a1;   /* the slice is macroblocks a0 .. a1 - 1 */
        if (rs.slice_first) { a1 = a0 + 1; while (a1 < nmb && rs.slice_first[a1] == a0) a1++; }
        else { a1 = a0 + e->slice_rows * e->mbw; if (a1 > nmb) a1 = nmb; }
        b = (bitw){e->rbsp, e->rbsp_cap, 0};
        rs.cur_slice_qp = rs.slice_qp[slice_no & 255];
        write_slice_header(e, &b, idr, a0);
        int skip_run = 0;
        for (int a = a0; a < a1; a++) {
                const int mx = a % e->mbw, my = a / e->mbw;
                const h264o_mbinfo *mb = &e->mb[a];
                if (!idr) {
                    if (mb->type == H264O_MB_PSKIP) { skip_run++; continue; }
                    bw_ue(&b, (uint32_t)skip_run);
                    skip_run = 0;
                }
                write_mb(e, &b, mx, my, !idr);
            }
        a0 = a1;
        if (skip_run) bw_ue(&b, (uint32_t)skip_run);
        bw_trailing(&b);
        if ((b.bits >> 3) > e->rbsp_cap) { pos = (size_t)-1; break; }
        pos = emit_nal(out, out_cap, pos, idr ? 3 : 2, idr ? 5 : 1, e->rbsp, (size_t)(b.bits >> 3));
        memset(e->rbsp, 0, (size_t)(b.bits >> 3) + 8);
    }
    e->rs = NULL;
    free(rs.qpd);
    rs.qpd = NULL;
    free(rs.slice_first);
    rs.slice_first = NULL;
    if (pos == (size_t)-1) return -2;
    if (idr) e->idr_id = (e->idr_id + e->idr_step) & 0xFF;
    e->frame_num = (e->frame_num + 1) & 255;
    e->frame_in_gop++;
    e->frames++;
    return (int64_t)pos;
}
