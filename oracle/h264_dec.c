/*
 * oracle/h264_dec.c -- TEST INFRASTRUCTURE ONLY (see h264_oracle.h).
 *
 * Test-side H.264 decoder (Baseline/Main CAVLC subset: I and P slices,
 * Intra16x16, P_L0_16x16, P_Skip, one reference frame, frame macroblocks)
 * written from ITU-T H.264 clauses 7.3 (syntax), 8.4.1 (motion vector
 * prediction), 9.1/9.2 (Exp-Golomb, CAVLC).  Its job is the round trip
 *     decode(encode(yuv)) == encoder reconstruction, byte for byte,
 * which is what stands in for the golden bitstreams the reference does not
 * have.  Annex-B start-code scanning follows the
 * in-tree statement of it at
 * /root/reference/video_decoder/VideoDecoderNetint.cpp:794-860.
 * Syntax parsing, CAVLC table search and motion-vector prediction are written
 * independently of the encoder; sample reconstruction shares h264_common.c.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "h264_oracle.h"
#include "h264_tables.h"

typedef struct {
    const uint8_t *p;
    size_t nbits, pos;
    int err;
} bitr;

static uint32_t br_peek(bitr *b, int n) /* n <= 24; zero padded past the end */
{
    uint32_t v = 0;
    for (int i = 0; i < n; i++) {
        size_t q = b->pos + (size_t)i;
        int bit = q < b->nbits ? (b->p[q >> 3] >> (7 - (q & 7))) & 1 : 0;
        v = (v << 1) | (uint32_t)bit;
    }
    return v;
}
static uint32_t br_get(bitr *b, int n)
{
    uint32_t v = 0;
    while (n > 0) {
        int k = n > 16 ? 16 : n;
        v = (v << k) | br_peek(b, k);
        b->pos += (size_t)k;
        n -= k;
    }
    if (b->pos > b->nbits) b->err = 1;
    return v;
}
static uint32_t br_ue(bitr *b)
{
    int z = 0;
    while (br_get(b, 1) == 0) {
        if (++z > 32 || b->err) { b->err = 1; return 0; }
    }
    return z ? ((1u << z) - 1 + br_get(b, z)) : 0;
}
static int32_t br_se(bitr *b)
{
    uint32_t k = br_ue(b);
    return (k & 1) ? (int32_t)((k + 1) >> 1) : -(int32_t)(k >> 1);
}

struct h264o_dec {
    /* SPS */
    int have_sps, profile, level, log2_max_frame_num, poc_type, log2_max_poc_lsb;
    int mbw, mbh, crop_r, crop_b;
    /* PPS */
    int have_pps, init_qp, chroma_qp_offset, dbf_ctrl, num_ref_default;
    /* picture */
    int cw, ch;
    uint8_t *cur[3], *ref[3];
    h264o_mbinfo *mb;
    int8_t *mbqp;
    int16_t *slice_of;
    int slice_type, nal_type, slice_count;
    char err[160];
};

h264o_dec *h264o_dec_create(void) { return (h264o_dec *)calloc(1, sizeof(h264o_dec)); }
void h264o_dec_destroy(h264o_dec *d)
{
    if (!d) return;
    for (int p = 0; p < 3; p++) { free(d->cur[p]); free(d->ref[p]); }
    free(d->mb); free(d->mbqp); free(d->slice_of);
    free(d);
}
int h264o_dec_width(const h264o_dec *d) { return d->cw - 2 * d->crop_r; }
int h264o_dec_height(const h264o_dec *d) { return d->ch - 2 * d->crop_b; }
int h264o_dec_coded_width(const h264o_dec *d) { return d->cw; }
int h264o_dec_coded_height(const h264o_dec *d) { return d->ch; }
const uint8_t *h264o_dec_plane(const h264o_dec *d, int p) { return d->ref[p]; }
const char *h264o_dec_error(const h264o_dec *d) { return d->err; }
int h264o_dec_last_slice_type(const h264o_dec *d) { return d->slice_type; }
int h264o_dec_last_nal_type(const h264o_dec *d) { return d->nal_type; }

static int fail(h264o_dec *d, const char *msg)
{
    snprintf(d->err, sizeof(d->err), "%s", msg);
    return -1;
}

static int parse_sps(h264o_dec *d, bitr *b)
{
    d->profile = (int)br_get(b, 8);
    br_get(b, 8);
    d->level = (int)br_get(b, 8);
    if (br_ue(b) != 0) return fail(d, "sps id != 0");
    if (d->profile == 100 || d->profile == 110 || d->profile == 122 || d->profile == 244) {
        if (br_ue(b) != 1) return fail(d, "chroma_format_idc != 1");
        if (br_ue(b) || br_ue(b)) return fail(d, "bit depth != 8");
        br_get(b, 1);
        if (br_get(b, 1)) return fail(d, "scaling matrices unsupported");
    }
    d->log2_max_frame_num = (int)br_ue(b) + 4;
    d->poc_type = (int)br_ue(b);
    if (d->poc_type == 0) d->log2_max_poc_lsb = (int)br_ue(b) + 4;
    else if (d->poc_type == 1) return fail(d, "poc type 1 unsupported");
    br_ue(b); /* max_num_ref_frames */
    br_get(b, 1);
    int mbw = (int)br_ue(b) + 1, mbh = (int)br_ue(b) + 1;
    if (!br_get(b, 1)) return fail(d, "interlace unsupported");
    br_get(b, 1);
    d->crop_r = d->crop_b = 0;
    if (br_get(b, 1)) {
        if (br_ue(b)) return fail(d, "left crop unsupported");
        d->crop_r = (int)br_ue(b);
        if (br_ue(b)) return fail(d, "top crop unsupported");
        d->crop_b = (int)br_ue(b);
    }
    if (b->err) return fail(d, "sps truncated");
    if (mbw != d->mbw || mbh != d->mbh || !d->mb) {
        for (int p = 0; p < 3; p++) { free(d->cur[p]); free(d->ref[p]); }
        free(d->mb); free(d->mbqp); free(d->slice_of);
        d->mbw = mbw; d->mbh = mbh; d->cw = 16 * mbw; d->ch = 16 * mbh;
        size_t ysz = (size_t)d->cw * d->ch;
        for (int p = 0; p < 3; p++) {
            d->cur[p] = (uint8_t *)calloc(p ? ysz / 4 : ysz, 1);
            d->ref[p] = (uint8_t *)calloc(p ? ysz / 4 : ysz, 1);
        }
        d->mb = (h264o_mbinfo *)calloc((size_t)mbw * mbh, sizeof(h264o_mbinfo));
        d->mbqp = (int8_t *)calloc((size_t)mbw * mbh, 1);
        d->slice_of = (int16_t *)calloc((size_t)mbw * mbh, sizeof(int16_t));
    }
    d->have_sps = 1;
    return 0;
}

static int parse_pps(h264o_dec *d, bitr *b)
{
    if (br_ue(b) || br_ue(b)) return fail(d, "pps/sps id != 0");
    if (br_get(b, 1)) return fail(d, "CABAC unsupported by the test decoder");
    br_get(b, 1);
    if (br_ue(b)) return fail(d, "slice groups unsupported");
    d->num_ref_default = (int)br_ue(b) + 1;
    br_ue(b);
    if (br_get(b, 1)) return fail(d, "weighted prediction unsupported");
    br_get(b, 2);
    d->init_qp = 26 + br_se(b);
    br_se(b);
    d->chroma_qp_offset = br_se(b);
    d->dbf_ctrl = (int)br_get(b, 1);
    if (br_get(b, 1)) return fail(d, "constrained intra pred unsupported");
    if (br_get(b, 1)) return fail(d, "redundant pics unsupported");
    if (b->err) return fail(d, "pps truncated");
    if (d->chroma_qp_offset) return fail(d, "chroma qp offset unsupported");
    d->have_pps = 1;
    return 0;
}

/* ---- CAVLC residual block, 9.2 ---- */
static int read_vlc(bitr *b, const uint8_t *len, const uint8_t *bits, int n)
{
    uint32_t pk = br_peek(b, 16);
    for (int i = 0; i < n; i++)
        if (len[i] && (pk >> (16 - len[i])) == bits[i]) { b->pos += len[i]; return i; }
    b->err = 1;
    return 0;
}

/* returns TotalCoeff, fills coef[0..max-1] in scan order */
static int read_block(bitr *b, int nC, int max_coeff, int16_t *coef)
{
    memset(coef, 0, sizeof(int16_t) * (size_t)max_coeff);
    int tok;
    if (nC == -1) tok = read_vlc(b, o_chroma_dc_token_len, o_chroma_dc_token_bits, 20);
    else {
        int tab = nC < 2 ? 0 : nC < 4 ? 1 : nC < 8 ? 2 : 3;
        tok = read_vlc(b, o_coeff_token_len[tab], o_coeff_token_bits[tab], 68);
    }
    int tc = tok >> 2, t1 = tok & 3;
    if (b->err || tc > max_coeff) { b->err = 1; return 0; }
    if (!tc) return 0;
    int level[16];
    int suffix_len = (tc > 10 && t1 < 3) ? 1 : 0;
    for (int i = 0; i < tc; i++) {
        if (i < t1) { level[i] = br_get(b, 1) ? -1 : 1; continue; }
        int prefix = 0;
        while (br_get(b, 1) == 0) { if (++prefix > 28 || b->err) { b->err = 1; return 0; } }
        int code = (prefix < 15 ? prefix : 15) << suffix_len;
        int ssize = (prefix == 14 && suffix_len == 0) ? 4 : prefix >= 15 ? prefix - 3 : suffix_len;
        if (ssize) code += (int)br_get(b, ssize);
        if (prefix >= 15 && suffix_len == 0) code += 15;
        if (prefix >= 16) code += (1 << (prefix - 3)) - 4096;
        if (i == t1 && t1 < 3) code += 2;
        level[i] = (code & 1) ? (-code - 1) >> 1 : (code + 2) >> 1;
        if (suffix_len == 0) suffix_len = 1;
        if (abs(level[i]) > (3 << (suffix_len - 1)) && suffix_len < 6) suffix_len++;
    }
    int zeros = 0;
    if (tc < max_coeff) {
        if (nC == -1) zeros = read_vlc(b, o_cdc_total_zeros_len[tc - 1], o_cdc_total_zeros_bits[tc - 1], 4);
        else zeros = read_vlc(b, o_total_zeros_len[tc - 1], o_total_zeros_bits[tc - 1], 16);
    }
    int pos = tc + zeros - 1; /* scan position of the highest-frequency coefficient */
    for (int i = 0; i < tc; i++) {
        if (pos < 0 || pos >= max_coeff) { b->err = 1; return 0; }
        coef[pos] = (int16_t)level[i];
        int run = 0;
        if (i < tc - 1 && zeros > 0) {
            int t = (zeros > 7 ? 7 : zeros) - 1;
            run = read_vlc(b, o_run_len[t], o_run_bits[t], t == 6 ? 15 : t + 2);
            zeros -= run;
        }
        pos -= 1 + run;
    }
    return tc;
}

/* ---- neighbour helpers (own statement of 6.4.11, 8.4.1.3) ---- */
static const uint8_t d_xy2blk[16] = {0, 1, 4, 5, 2, 3, 6, 7, 8, 9, 12, 13, 10, 11, 14, 15};

static int mb_avail(const h264o_dec *d, int mx, int my, int cur_slice)
{
    if (mx < 0 || my < 0 || mx >= d->mbw || my >= d->mbh) return 0;
    return d->slice_of[my * d->mbw + mx] == cur_slice;
}
static int pred_nc(const h264o_dec *d, int mx, int my, int sl, int comp, int x, int y)
{
    /* comp 0: luma 4x4 grid (x,y in 0..3); 1/2: chroma 2x2 grid */
    int nA = -1, nB = -1, lim = comp ? 1 : 3;
    const h264o_mbinfo *m = &d->mb[my * d->mbw + mx];
#define TC(mbp, xx, yy) (comp ? (mbp)->tc[16 + (comp - 1) * 4 + 2 * (yy) + (xx)] : (mbp)->tc[d_xy2blk[4 * (yy) + (xx)]])
    if (x > 0) nA = TC(m, x - 1, y);
    else if (mb_avail(d, mx - 1, my, sl)) nA = TC(m - 1, lim, y);
    if (y > 0) nB = TC(m, x, y - 1);
    else if (mb_avail(d, mx, my - 1, sl)) nB = TC(m - d->mbw, x, lim);
#undef TC
    if (nA >= 0 && nB >= 0) return (nA + nB + 1) >> 1;
    return nA >= 0 ? nA : nB >= 0 ? nB : 0;
}

typedef struct { int avail, ref, x, y; } nbmv;
static nbmv get_nb(const h264o_dec *d, int mx, int my, int sl)
{
    nbmv n = {0, -1, 0, 0};
    if (!mb_avail(d, mx, my, sl)) return n;
    n.avail = 1;
    const h264o_mbinfo *m = &d->mb[my * d->mbw + mx];
    if (m->type != H264O_MB_I16) { n.ref = 0; n.x = m->mvx; n.y = m->mvy; }
    return n;
}
static int median3(int a, int b, int c)
{
    int mn = a < b ? a : b, mx = a < b ? b : a;
    mn = mn < c ? mn : c; mx = mx > c ? mx : c;
    return a + b + c - mn - mx;
}
static void pred_mv16(const h264o_dec *d, int mx, int my, int sl, int *px, int *py, int skip)
{
    nbmv A = get_nb(d, mx - 1, my, sl), B = get_nb(d, mx, my - 1, sl), C = get_nb(d, mx + 1, my - 1, sl);
    if (!C.avail) C = get_nb(d, mx - 1, my - 1, sl);
    if (skip && (!A.avail || !B.avail || (A.ref == 0 && !A.x && !A.y) || (B.ref == 0 && !B.x && !B.y))) {
        *px = *py = 0;
        return;
    }
    if (!B.avail && !C.avail && A.avail) { B = A; C = A; }
    int cnt = (A.ref == 0) + (B.ref == 0) + (C.ref == 0);
    if (cnt == 1) {
        nbmv s = A.ref == 0 ? A : B.ref == 0 ? B : C;
        *px = s.x; *py = s.y;
    } else {
        *px = median3(A.x, B.x, C.x);
        *py = median3(A.y, B.y, C.y);
    }
}

/* ---- macroblock reconstruction ---- */
static void scan_to_raster(const int16_t *scan, int first, int16_t raster[16])
{
    memset(raster, 0, 32);
    for (int i = first; i < 16; i++) raster[o_zigzag4x4[i]] = scan[i - first];
}

static int decode_chroma(h264o_dec *d, bitr *b, int mx, int my, int sl, int cbpc, int qp, uint8_t predc[2][64])
{
    int qpc = o_chroma_qp[qp < 0 ? 0 : qp > 51 ? 51 : qp], cs = d->cw / 2;
    h264o_mbinfo *m = &d->mb[my * d->mbw + mx];
    int16_t dc[2][4] = {{0}}, ac[2][4][15];
    memset(ac, 0, sizeof(ac));
    if (cbpc)
        for (int pl = 0; pl < 2; pl++) read_block(b, -1, 4, dc[pl]);
    for (int pl = 0; pl < 2; pl++)
        for (int k = 0; k < 4; k++) {
            int tc = 0;
            if (cbpc == 2) tc = read_block(b, pred_nc(d, mx, my, sl, 1 + pl, k & 1, k >> 1), 15, ac[pl][k]);
            m->tc[16 + pl * 4 + k] = (uint8_t)tc;
        }
    if (b->err) return -1;
    for (int pl = 0; pl < 2; pl++) {
        uint8_t *r = d->cur[1 + pl] + (8 * my) * cs + 8 * mx;
        for (int y = 0; y < 8; y++) memcpy(r + y * cs, predc[pl] + 8 * y, 8);
        int c0 = dc[pl][0], c1 = dc[pl][1], c2 = dc[pl][2], c3 = dc[pl][3];
        int f[4] = {c0 + c1 + c2 + c3, c0 - c1 + c2 - c3, c0 + c1 - c2 - c3, c0 - c1 - c2 + c3};
        for (int k = 0; k < 4; k++) {
            int16_t lv[16], dq[16];
            scan_to_raster(ac[pl][k], 1, lv);
            h264o_dequant4x4(lv, qpc, dq);
            dq[0] = (int16_t)(((f[k] * 16 * o_dequant_v[qpc % 6][0]) << (qpc / 6)) >> 5);
            h264o_idct4x4_add(dq, r + (k >> 1) * 4 * cs + (k & 1) * 4, cs);
        }
    }
    return 0;
}

static int decode_mb_intra16(h264o_dec *d, bitr *b, int mx, int my, int sl, int t, int *qp)
{
    /* t = mb_type - 1 within the I16x16 range 0..23 */
    int mode = t & 3, cbpc = (t >> 2) % 3, cbpl = t >= 12 ? 15 : 0, cw = d->cw, cs = cw / 2;
    h264o_mbinfo *m = &d->mb[my * d->mbw + mx];
    memset(m, 0, sizeof(*m));
    m->type = H264O_MB_I16;
    m->i16_mode = (uint8_t)mode;
    m->cbp = (uint8_t)(cbpl | (cbpc << 4));
    int cmode = (int)br_ue(b);
    if (cmode > 3) return fail(d, "intra_chroma_pred_mode > 3");
    m->chroma_mode = (uint8_t)cmode;
    *qp += br_se(b);
    *qp = (*qp + 52) % 52;
    int avail = (mb_avail(d, mx - 1, my, sl) ? 1 : 0) | (mb_avail(d, mx, my - 1, sl) ? 2 : 0) |
                (mb_avail(d, mx - 1, my - 1, sl) ? 4 : 0);
    if ((mode == 0 && !(avail & 2)) || (mode == 1 && !(avail & 1)) || (mode == 3 && avail != 7))
        return fail(d, "intra16x16 mode uses unavailable neighbours");
    if ((cmode == 1 && !(avail & 1)) || (cmode == 2 && !(avail & 2)) || (cmode == 3 && avail != 7))
        return fail(d, "chroma pred mode uses unavailable neighbours");
    int16_t dcs[16], acs[16][15];
    memset(acs, 0, sizeof(acs));
    read_block(b, pred_nc(d, mx, my, sl, 0, 0, 0), 16, dcs);
    for (int blk = 0; blk < 16; blk++) {
        int tc = 0;
        if (cbpl) tc = read_block(b, pred_nc(d, mx, my, sl, 0, o_blk_x[blk], o_blk_y[blk]), 15, acs[blk]);
        m->tc[blk] = (uint8_t)tc;
    }
    if (b->err) return fail(d, "residual parse error (I16x16)");
    uint8_t *r = d->cur[0] + (16 * my) * cw + 16 * mx;
    uint8_t pred[256];
    h264o_pred16x16(r, cw, mode, avail, pred);
    /* 8.5.10 luma DC */
    int16_t c[16];
    scan_to_raster(dcs, 0, c);
    int t4[16], f[16];
    for (int i = 0; i < 4; i++) {
        int a = c[4 * i], bb = c[4 * i + 1], cc = c[4 * i + 2], dd = c[4 * i + 3];
        t4[4 * i] = a + bb + cc + dd; t4[4 * i + 1] = a + bb - cc - dd;
        t4[4 * i + 2] = a - bb - cc + dd; t4[4 * i + 3] = a - bb + cc - dd;
    }
    for (int j = 0; j < 4; j++) {
        int a = t4[j], bb = t4[4 + j], cc = t4[8 + j], dd = t4[12 + j];
        f[j] = a + bb + cc + dd; f[4 + j] = a + bb - cc - dd;
        f[8 + j] = a - bb - cc + dd; f[12 + j] = a - bb + cc - dd;
    }
    int q = *qp, ls = 16 * o_dequant_v[q % 6][0];
    for (int y = 0; y < 16; y++) memcpy(r + y * cw, pred + 16 * y, 16);
    for (int blk = 0; blk < 16; blk++) {
        int16_t lv[16], dq[16];
        scan_to_raster(acs[blk], 1, lv);
        h264o_dequant4x4(lv, q, dq);
        int fi = f[o_blk_y[blk] * 4 + o_blk_x[blk]];
        dq[0] = (int16_t)(q >= 36 ? (fi * ls) << (q / 6 - 6) : (fi * ls + (1 << (5 - q / 6))) >> (6 - q / 6));
        h264o_idct4x4_add(dq, r + o_blk_y[blk] * 4 * cw + o_blk_x[blk] * 4, cw);
    }
    uint8_t predc[2][64];
    for (int pl = 0; pl < 2; pl++)
        h264o_pred_chroma8x8(d->cur[1 + pl] + (8 * my) * cs + 8 * mx, cs, cmode, avail, predc[pl]);
    if (decode_chroma(d, b, mx, my, sl, cbpc, q, predc)) return fail(d, "residual parse error (chroma)");
    return 0;
}

static int decode_mb_inter16(h264o_dec *d, bitr *b, int mx, int my, int sl, int skip, int *qp)
{
    int cw = d->cw, cs = cw / 2;
    h264o_mbinfo *m = &d->mb[my * d->mbw + mx];
    memset(m, 0, sizeof(*m));
    m->type = skip ? H264O_MB_PSKIP : H264O_MB_P16;
    int px, py;
    pred_mv16(d, mx, my, sl, &px, &py, skip);
    int cbp = 0;
    if (!skip) {
        px += br_se(b);
        py += br_se(b);
        uint32_t code = br_ue(b);
        if (code > 47) return fail(d, "coded_block_pattern out of range");
        cbp = o_cbp_code2inter[code];
        if (cbp) { *qp += br_se(b); *qp = (*qp + 52) % 52; }
    }
    m->mvx = (int16_t)px;
    m->mvy = (int16_t)py;
    m->cbp = (uint8_t)cbp;
    uint8_t pred[256], predc[2][64];
    h264o_mc_luma(d->ref[0], cw, cw, d->ch, 16 * mx, 16 * my, px, py, 16, 16, pred, 16);
    for (int pl = 0; pl < 2; pl++)
        h264o_mc_chroma(d->ref[1 + pl], cs, cs, d->ch / 2, 8 * mx, 8 * my, px, py, 8, 8, predc[pl], 8);
    uint8_t *r = d->cur[0] + (16 * my) * cw + 16 * mx;
    for (int y = 0; y < 16; y++) memcpy(r + y * cw, pred + 16 * y, 16);
    for (int blk = 0; blk < 16; blk++) {
        if (!(cbp & (1 << (blk >> 2)))) continue;
        int16_t sc[16], lv[16], dq[16];
        int tc = read_block(b, pred_nc(d, mx, my, sl, 0, o_blk_x[blk], o_blk_y[blk]), 16, sc);
        if (b->err) return fail(d, "residual parse error (inter luma)");
        m->tc[blk] = (uint8_t)tc;
        scan_to_raster(sc, 0, lv);
        h264o_dequant4x4(lv, *qp, dq);
        h264o_idct4x4_add(dq, r + o_blk_y[blk] * 4 * cw + o_blk_x[blk] * 4, cw);
    }
    if (decode_chroma(d, b, mx, my, sl, cbp >> 4, *qp, predc)) return fail(d, "residual parse error (chroma)");
    return 0;
}

static int decode_slice(h264o_dec *d, bitr *b, int nal_type, int nal_ref_idc)
{
    if (!d->have_sps || !d->have_pps) return fail(d, "slice before parameter sets");
    int first_mb = (int)br_ue(b);
    int st = (int)br_ue(b) % 5;
    if (st != 0 && st != 2) return fail(d, "only I and P slices supported");
    br_ue(b); /* pps id */
    br_get(b, d->log2_max_frame_num);
    if (nal_type == 5) br_ue(b);
    if (d->poc_type == 0) br_get(b, d->log2_max_poc_lsb);
    if (st == 0) {
        if (br_get(b, 1)) { if (br_ue(b) != 0) return fail(d, "num_ref_idx_active > 1 unsupported"); }
        else if (d->num_ref_default != 1) return fail(d, "num_ref_idx_active > 1 unsupported");
        if (br_get(b, 1)) return fail(d, "ref pic list modification unsupported");
    }
    if (nal_ref_idc) {
        if (nal_type == 5) br_get(b, 2);
        else if (br_get(b, 1)) return fail(d, "adaptive ref pic marking unsupported");
    }
    int qp = d->init_qp + br_se(b);
    int disable_dbf = 0;
    if (d->dbf_ctrl) {
        disable_dbf = (int)br_ue(b);
        if (disable_dbf != 1) {
            if (br_se(b) || br_se(b)) return fail(d, "deblock offsets unsupported");
        }
    }
    if (b->err) return fail(d, "slice header truncated");
    if (first_mb == 0) {
        d->slice_count = 0;
        memset(d->slice_of, 0xff, sizeof(int16_t) * (size_t)d->mbw * d->mbh);
    }
    int sl = d->slice_count++;
    d->slice_type = st;
    d->nal_type = nal_type;
    int nmb = d->mbw * d->mbh, addr = first_mb, slice_qp = qp;
    int more = 1;
    while (more && addr < nmb) {
        if (st == 0) {
            int run = (int)br_ue(b);
            if (b->err) return fail(d, "mb_skip_run parse error");
            for (; run > 0 && addr < nmb; run--, addr++) {
                d->slice_of[addr] = (int16_t)sl;
                d->mbqp[addr] = (int8_t)qp;
                if (decode_mb_inter16(d, b, addr % d->mbw, addr / d->mbw, sl, 1, &qp)) return -1;
            }
            if (run > 0) return fail(d, "mb_skip_run past end of picture");
            /* more_rbsp_data(): anything but the stop bit + zero padding left? */
            size_t last = b->nbits;
            while (last > 0 && !((b->p[(last - 1) >> 3] >> (7 - ((last - 1) & 7))) & 1)) last--;
            if (b->pos >= last - 1) break;
        }
        int mb_type = (int)br_ue(b);
        d->slice_of[addr] = (int16_t)sl;
        int rc;
        if (st == 0 && mb_type == 0) rc = decode_mb_inter16(d, b, addr % d->mbw, addr / d->mbw, sl, 0, &qp);
        else {
            int it = st == 0 ? mb_type - 5 : mb_type;
            if (it < 1 || it > 24) return fail(d, "unsupported mb_type");
            rc = decode_mb_intra16(d, b, addr % d->mbw, addr / d->mbw, sl, it - 1, &qp);
        }
        if (rc) return rc;
        if (qp != slice_qp) return fail(d, "per-MB QP change unsupported by the test decoder");
        d->mbqp[addr] = (int8_t)qp;
        addr++;
        size_t last = b->nbits;
        while (last > 0 && !((b->p[(last - 1) >> 3] >> (7 - ((last - 1) & 7))) & 1)) last--;
        more = b->pos < last - 1;
    }
    if (b->err) return fail(d, "slice data overrun");
    if (addr < nmb) return 0; /* more slices follow */
    /* (all slices of a picture carry the same filter idc and QP in the streams this test decoder reads) */
    if (disable_dbf != 1)
        h264o_deblock_picture(d->cur[0], d->cur[1], d->cur[2], d->cw, d->ch, d->mb, slice_qp, disable_dbf == 2 ? d->slice_of : NULL, 0, d->mbh);
    for (int p = 0; p < 3; p++) { uint8_t *t = d->ref[p]; d->ref[p] = d->cur[p]; d->cur[p] = t; }
    return 1;
}

int h264o_dec_decode(h264o_dec *d, const uint8_t *data, size_t len)
{
    int got_pic = 0;
    size_t i = 0;
    d->err[0] = 0;
    uint8_t *rbsp = (uint8_t *)malloc(len + 8);
    while (i + 3 <= len) {
        /* find 00 00 01 */
        if (!(data[i] == 0 && data[i + 1] == 0 && data[i + 2] == 1)) { i++; continue; }
        size_t s = i + 3, e = s;
        while (e + 3 <= len && !(data[e] == 0 && data[e + 1] == 0 && (data[e + 2] == 1 || data[e + 2] == 0))) e++;
        if (e + 3 > len) e = len;
        if (e <= s) { i = e; continue; }
        int hdr = data[s], type = hdr & 31, ref_idc = (hdr >> 5) & 3;
        if (hdr & 0x80) { free(rbsp); return fail(d, "forbidden_zero_bit set"); }
        size_t n = 0;
        int zeros = 0;
        for (size_t k = s + 1; k < e; k++) {
            if (zeros >= 2 && data[k] == 3) { zeros = 0; continue; }
            if (zeros >= 2 && data[k] < 3) { free(rbsp); return fail(d, "start code emulation inside NAL"); }
            rbsp[n++] = data[k];
            zeros = data[k] == 0 ? zeros + 1 : 0;
        }
        bitr b = {rbsp, n * 8, 0, 0};
        int rc = 0;
        if (type == 7) rc = parse_sps(d, &b);
        else if (type == 8) rc = parse_pps(d, &b);
        else if (type == 1 || type == 5) { rc = decode_slice(d, &b, type, ref_idc); if (rc == 1) got_pic = 1; }
        if (rc < 0) { free(rbsp); return rc; }
        i = e;
    }
    free(rbsp);
    return got_pic;
}
