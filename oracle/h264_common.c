/*
 * oracle/h264_common.c -- TEST INFRASTRUCTURE ONLY (see h264_oracle.h).
 *
 * Sample-level stages of the H.264 encode hot path, restated from ITU-T H.264:
 *   8.5.12  inverse 4x4 transform        8.5.10/11 DC Hadamards
 *   8.5.9   scaling (dequantisation)     8.4.2.2  fractional sample interpolation
 *   8.3.3/4 Intra16x16 / chroma predict  8.7      deblocking filter
 * and the non-normative forward transform / quantiser / SAD / SATD an encoder
 * needs around them.  These are the interior of ISVCEncoder::EncodeFrame, which
 * the reference reaches at /root/reference/video_codec/VideoEncoderOpenH264.cpp:344
 * (SURVEY.md 8a rows a6.1-a6.4).  PARITY UNPINNED vs OpenH264.
 */
#include <stdlib.h>
#include <string.h>
#include "h264_oracle.h"
#include "h264_tables.h"
#if defined(__SSE2__)
#include <emmintrin.h>
#endif

static inline int clip3(int lo, int hi, int v) { return v < lo ? lo : (v > hi ? hi : v); }
static inline uint8_t clip1(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

/* ---------------------------------------------------------------- transform */
/* forward core transform W = Cf X Cf^T, Cf rows (1,1,1,1)(2,1,-1,-2)(1,-1,-1,1)(1,-2,2,-1) */
void h264o_fdct4x4(const int16_t in[16], int16_t out[16])
{
    int t[16];
    for (int i = 0; i < 4; i++) { /* rows */
        int a = in[4 * i], b = in[4 * i + 1], c = in[4 * i + 2], d = in[4 * i + 3];
        int s0 = a + d, s1 = b + c, d0 = a - d, d1 = b - c;
        t[4 * i] = s0 + s1;
        t[4 * i + 1] = 2 * d0 + d1;
        t[4 * i + 2] = s0 - s1;
        t[4 * i + 3] = d0 - 2 * d1;
    }
    for (int j = 0; j < 4; j++) { /* columns */
        int a = t[j], b = t[4 + j], c = t[8 + j], d = t[12 + j];
        int s0 = a + d, s1 = b + c, d0 = a - d, d1 = b - c;
        out[j] = (int16_t)(s0 + s1);
        out[4 + j] = (int16_t)(2 * d0 + d1);
        out[8 + j] = (int16_t)(s0 - s1);
        out[12 + j] = (int16_t)(d0 - 2 * d1);
    }
}

/* 8.5.12.2: inverse transform of scaled coefficients d (raster), rounding (x+32)>>6 */
static void idct4x4_res(const int16_t d[16], int r[16])
{
    int f[16];
    for (int i = 0; i < 4; i++) {
        int d0 = d[4 * i], d1 = d[4 * i + 1], d2 = d[4 * i + 2], d3 = d[4 * i + 3];
        int e0 = d0 + d2, e1 = d0 - d2, e2 = (d1 >> 1) - d3, e3 = d1 + (d3 >> 1);
        f[4 * i] = e0 + e3;
        f[4 * i + 1] = e1 + e2;
        f[4 * i + 2] = e1 - e2;
        f[4 * i + 3] = e0 - e3;
    }
    for (int j = 0; j < 4; j++) {
        int f0 = f[j], f1 = f[4 + j], f2 = f[8 + j], f3 = f[12 + j];
        int g0 = f0 + f2, g1 = f0 - f2, g2 = (f1 >> 1) - f3, g3 = f1 + (f3 >> 1);
        r[j] = (g0 + g3 + 32) >> 6;
        r[4 + j] = (g1 + g2 + 32) >> 6;
        r[8 + j] = (g1 - g2 + 32) >> 6;
        r[12 + j] = (g0 - g3 + 32) >> 6;
    }
}

void h264o_idct4x4_add(const int16_t coef[16], uint8_t *dst, int stride)
{
    int r[16];
    idct4x4_res(coef, r);
    for (int y = 0; y < 4; y++)
        for (int x = 0; x < 4; x++) dst[y * stride + x] = clip1(dst[y * stride + x] + r[4 * y + x]);
}

void h264o_quant4x4(const int16_t w[16], int qp, int intra, int16_t lv[16])
{
    int qbits = 15 + qp / 6;
    int f = (1 << qbits) / (intra ? 3 : 6);
    for (int i = 0; i < 16; i++) {
        int mf = o_quant_mf[qp % 6][o_pos_class(i)];
        int a = w[i] < 0 ? -w[i] : w[i];
        int l = (a * mf + f) >> qbits;
        lv[i] = (int16_t)(w[i] < 0 ? -l : l);
    }
}

/* 8.5.12.1 with flat weight 16: d = (c * 16 v) << (qp/6) >> 4  ==  (c*v) << (qp/6) */
void h264o_dequant4x4(const int16_t lv[16], int qp, int16_t out[16])
{
    for (int i = 0; i < 16; i++) out[i] = (int16_t)((lv[i] * o_dequant_v[qp % 6][o_pos_class(i)]) << (qp / 6));
}

/* ---------------------------------------------------------------- 8x8 transform (High profile) */
/* forward 8x8 integer transform (the reference model's butterfly, the transpose of 8.5.13's inverse up to scaling) */
static void fdct8_1d(const int in[8], int out[8])
{
    const int s07 = in[0] + in[7], s16 = in[1] + in[6], s25 = in[2] + in[5], s34 = in[3] + in[4];
    const int a0 = s07 + s34, a1 = s16 + s25, a2 = s07 - s34, a3 = s16 - s25;
    const int d07 = in[0] - in[7], d16 = in[1] - in[6], d25 = in[2] - in[5], d34 = in[3] - in[4];
    const int a4 = d16 + d25 + (d07 + (d07 >> 1)), a5 = d07 - d34 - (d25 + (d25 >> 1));
    const int a6 = d07 + d34 - (d16 + (d16 >> 1)), a7 = d16 - d25 + (d34 + (d34 >> 1));
    out[0] = a0 + a1; out[1] = a4 + (a7 >> 2); out[2] = a2 + (a3 >> 1); out[3] = a5 + (a6 >> 2);
    out[4] = a0 - a1; out[5] = a6 - (a5 >> 2); out[6] = (a2 >> 1) - a3; out[7] = (a4 >> 2) - a7;
}
void h264o_fdct8x8(const int16_t in[64], int32_t out[64])
{
    int t[64];
    for (int i = 0; i < 8; i++) {
        int r[8], o[8];
        for (int j = 0; j < 8; j++) r[j] = in[8 * i + j];
        fdct8_1d(r, o);
        for (int j = 0; j < 8; j++) t[8 * i + j] = o[j];
    }
    for (int j = 0; j < 8; j++) {
        int c[8], o[8];
        for (int i = 0; i < 8; i++) c[i] = t[8 * i + j];
        fdct8_1d(c, o);
        for (int i = 0; i < 8; i++) out[8 * i + j] = o[i];
    }
}
void h264o_quant8x8(const int32_t w[64], int qp, int intra, int16_t lv[64])
{
    int qbits = 16 + qp / 6;
    int f = (1 << qbits) / (intra ? 3 : 6);
    for (int i = 0; i < 64; i++) {
        int64_t a = w[i] < 0 ? -w[i] : w[i];
        int l = (int)((a * o_quant8_mf[qp % 6][o_pos_class8(i)] + f) >> qbits);
        lv[i] = (int16_t)(w[i] < 0 ? -l : l);
    }
}
/* 8.5.13 scaling with flat weights 16 */
void h264o_dequant8x8(const int16_t lv[64], int qp, int32_t out[64])
{
    for (int i = 0; i < 64; i++) {
        int ls = 16 * o_dequant8_v[qp % 6][o_pos_class8(i)];
        out[i] = qp >= 36 ? (lv[i] * ls) << (qp / 6 - 6) : (lv[i] * ls + (1 << (5 - qp / 6))) >> (6 - qp / 6);
    }
}
static void idct8_1d(const int in[8], int out[8])
{
    const int a0 = in[0] + in[4], a2 = in[0] - in[4], a4 = (in[2] >> 1) - in[6], a6 = in[2] + (in[6] >> 1);
    const int b0 = a0 + a6, b2 = a2 + a4, b4 = a2 - a4, b6 = a0 - a6;
    const int a1 = -in[3] + in[5] - in[7] - (in[7] >> 1), a3 = in[1] + in[7] - in[3] - (in[3] >> 1);
    const int a5 = -in[1] + in[7] + in[5] + (in[5] >> 1), a7 = in[3] + in[5] + in[1] + (in[1] >> 1);
    const int b1 = a1 + (a7 >> 2), b3 = a3 + (a5 >> 2), b5 = (a3 >> 2) - a5, b7 = a7 - (a1 >> 2);
    out[0] = b0 + b7; out[1] = b2 + b5; out[2] = b4 + b3; out[3] = b6 + b1;
    out[4] = b6 - b1; out[5] = b4 - b3; out[6] = b2 - b5; out[7] = b0 - b7;
}
void h264o_idct8x8_add(const int32_t coef[64], uint8_t *dst, int stride)
{
    int t[64];
    for (int i = 0; i < 8; i++) {
        int r[8], o[8];
        for (int j = 0; j < 8; j++) r[j] = coef[8 * i + j];
        idct8_1d(r, o);
        for (int j = 0; j < 8; j++) t[8 * i + j] = o[j];
    }
    for (int j = 0; j < 8; j++) {
        int c[8], o[8];
        for (int i = 0; i < 8; i++) c[i] = t[8 * i + j];
        idct8_1d(c, o);
        for (int i = 0; i < 8; i++) dst[i * stride + j] = clip1(dst[i * stride + j] + ((o[i] + 32) >> 6));
    }
}

/* ---------------------------------------------------------------- SAD / SATD */
int h264o_sad16x16(const uint8_t *a, int as, const uint8_t *b, int bs)
{
#if defined(__SSE2__)
    __m128i acc = _mm_setzero_si128();
    for (int y = 0; y < 16; y++) {
        __m128i va = _mm_loadu_si128((const __m128i *)(a + y * as));
        __m128i vb = _mm_loadu_si128((const __m128i *)(b + y * bs));
        acc = _mm_add_epi64(acc, _mm_sad_epu8(va, vb));
    }
    return _mm_cvtsi128_si32(acc) + _mm_cvtsi128_si32(_mm_srli_si128(acc, 8));
#else
    int s = 0;
    for (int y = 0; y < 16; y++)
        for (int x = 0; x < 16; x++) s += abs(a[y * as + x] - b[y * bs + x]);
    return s;
#endif
}

/* sum of |H4 D H4^T| over one 4x4 difference block (no normalisation) */
static int hadamard4x4_abs(const uint8_t *a, int as, const uint8_t *b, int bs)
{
    int d[16], t[16], s = 0;
    for (int y = 0; y < 4; y++)
        for (int x = 0; x < 4; x++) d[4 * y + x] = a[y * as + x] - b[y * bs + x];
    for (int i = 0; i < 4; i++) {
        int s0 = d[4 * i] + d[4 * i + 3], s1 = d[4 * i + 1] + d[4 * i + 2];
        int d0 = d[4 * i] - d[4 * i + 3], d1 = d[4 * i + 1] - d[4 * i + 2];
        t[4 * i] = s0 + s1;
        t[4 * i + 1] = d0 + d1;
        t[4 * i + 2] = s0 - s1;
        t[4 * i + 3] = d0 - d1;
    }
    for (int j = 0; j < 4; j++) {
        int s0 = t[j] + t[12 + j], s1 = t[4 + j] + t[8 + j];
        int d0 = t[j] - t[12 + j], d1 = t[4 + j] - t[8 + j];
        s += abs(s0 + s1) + abs(d0 + d1) + abs(s0 - s1) + abs(d0 - d1);
    }
    return s;
}

/* SATD = (sum over 4x4 blocks of sum |Hadamard|) >> 1 */
int h264o_satd16x16(const uint8_t *a, int as, const uint8_t *b, int bs)
{
    int s = 0;
    for (int y = 0; y < 16; y += 4)
        for (int x = 0; x < 16; x += 4) s += hadamard4x4_abs(a + y * as + x, as, b + y * bs + x, bs);
    return s >> 1;
}

int h264o_satd_rect(const uint8_t *a, int as, const uint8_t *b, int bs, int w, int h)
{
    int s = 0;
    for (int y = 0; y < h; y += 4)
        for (int x = 0; x < w; x += 4) s += hadamard4x4_abs(a + y * as + x, as, b + y * bs + x, bs);
    return s >> 1;
}
int h264o_satd8x8(const uint8_t *a, int as, const uint8_t *b, int bs)
{
    int s = 0;
    for (int y = 0; y < 8; y += 4)
        for (int x = 0; x < 8; x += 4) s += hadamard4x4_abs(a + y * as + x, as, b + y * bs + x, bs);
    return s >> 1;
}

/* ------------------------------------------------- 8.4.2.2 interpolation */
static inline int refpx(const uint8_t *ref, int stride, int w, int h, int x, int y)
{
    return ref[clip3(0, h - 1, y) * stride + clip3(0, w - 1, x)];
}
static inline int tap6(int a, int b, int c, int d, int e, int f) { return a - 5 * b + 20 * c + 20 * d - 5 * e + f; }

/* one luma sample at full-pel (xi,yi) plus fractional (fx,fy) in quarter units */
static int luma_sample(const uint8_t *ref, int stride, int w, int h, int xi, int yi, int fx, int fy)
{
#define P(dx, dy) refpx(ref, stride, w, h, xi + (dx), yi + (dy))
#define HB1(dx, dy) tap6(P((dx) - 2, dy), P((dx) - 1, dy), P(dx, dy), P((dx) + 1, dy), P((dx) + 2, dy), P((dx) + 3, dy))
#define VH1(dx, dy) tap6(P(dx, (dy) - 2), P(dx, (dy) - 1), P(dx, dy), P(dx, (dy) + 1), P(dx, (dy) + 2), P(dx, (dy) + 3))
    int G = P(0, 0);
    if (fx == 0 && fy == 0) return G;
    int b = clip1((HB1(0, 0) + 16) >> 5);  /* half, horizontal, at (x+1/2, y)   */
    int hh = clip1((VH1(0, 0) + 16) >> 5); /* half, vertical, at (x, y+1/2)     */
    if (fy == 0) {
        if (fx == 2) return b;
        if (fx == 1) return (G + b + 1) >> 1;
        return (P(1, 0) + b + 1) >> 1;
    }
    if (fx == 0) {
        if (fy == 2) return hh;
        if (fy == 1) return (G + hh + 1) >> 1;
        return (P(0, 1) + hh + 1) >> 1;
    }
    /* centre sample j from unclipped horizontal intermediates */
    int j1 = tap6(HB1(0, -2), HB1(0, -1), HB1(0, 0), HB1(0, 1), HB1(0, 2), HB1(0, 3));
    int j = clip1((j1 + 512) >> 10);
    int s = clip1((HB1(0, 1) + 16) >> 5); /* b one row below   */
    int m = clip1((VH1(1, 0) + 16) >> 5); /* h one column right */
    if (fx == 2 && fy == 2) return j;
    if (fx == 2) return fy == 1 ? (b + j + 1) >> 1 : (s + j + 1) >> 1; /* f, q */
    if (fy == 2) return fx == 1 ? (hh + j + 1) >> 1 : (m + j + 1) >> 1; /* i, k */
    if (fx == 1 && fy == 1) return (b + hh + 1) >> 1; /* e */
    if (fx == 3 && fy == 1) return (b + m + 1) >> 1;  /* g */
    if (fx == 1 && fy == 3) return (hh + s + 1) >> 1; /* p */
    return (m + s + 1) >> 1;                          /* r */
#undef P
#undef HB1
#undef VH1
}

/* spec-literal single-sample form, kept as the cross-check of the block form */
int h264o_luma_sample_ref(const uint8_t *ref, int stride, int w, int h, int x, int y, int mvx, int mvy)
{
    return luma_sample(ref, stride, w, h, x + (mvx >> 2), y + (mvy >> 2), mvx & 3, mvy & 3);
}

/* block form: fetch a clamped (bw+6)x(bh+6) window once, filter separably */
void h264o_mc_luma(const uint8_t *ref, int stride, int w, int h, int x, int y, int mvx, int mvy,
                   int bw, int bh, uint8_t *dst, int dstride)
{
    enum { MAXB = 16, WS = MAXB + 6 };
    int xi = x + (mvx >> 2), yi = y + (mvy >> 2), fx = mvx & 3, fy = mvy & 3;
    uint8_t W[WS][WS];
    int b1[WS][MAXB + 1]; /* unclipped horizontal half sums, rows -2..bh+3 -> index +2 */
    for (int r = 0; r < bh + 6; r++) {
        const uint8_t *row = ref + clip3(0, h - 1, yi + r - 2) * stride;
        for (int c = 0; c < bw + 6; c++) W[r][c] = row[clip3(0, w - 1, xi + c - 2)];
    }
#define G(i, j) W[(j) + 2][(i) + 2]
    if (fx == 0 && fy == 0) {
        for (int j = 0; j < bh; j++)
            for (int i = 0; i < bw; i++) dst[j * dstride + i] = G(i, j);
        return;
    }
    if (fx != 0 || fy != 0)
        for (int r = 0; r < bh + 6; r++)
            for (int i = 0; i < bw; i++)
                b1[r][i] = tap6(W[r][i], W[r][i + 1], W[r][i + 2], W[r][i + 3], W[r][i + 4], W[r][i + 5]);
#define Bh(i, j) clip1((b1[(j) + 2][i] + 16) >> 5)
#define Hv(i, j) clip1((tap6(G(i, (j) - 2), G(i, (j) - 1), G(i, j), G(i, (j) + 1), G(i, (j) + 2), G(i, (j) + 3)) + 16) >> 5)
#define Jc(i, j) clip1((tap6(b1[j][i], b1[(j) + 1][i], b1[(j) + 2][i], b1[(j) + 3][i], b1[(j) + 4][i], b1[(j) + 5][i]) + 512) >> 10)
    for (int j = 0; j < bh; j++)
        for (int i = 0; i < bw; i++) {
            int v;
            if (fy == 0) {
                int b = Bh(i, j);
                v = fx == 2 ? b : fx == 1 ? (G(i, j) + b + 1) >> 1 : (G(i + 1, j) + b + 1) >> 1;
            } else if (fx == 0) {
                int hh = Hv(i, j);
                v = fy == 2 ? hh : fy == 1 ? (G(i, j) + hh + 1) >> 1 : (G(i, j + 1) + hh + 1) >> 1;
            } else if (fx == 2 && fy == 2) {
                v = Jc(i, j);
            } else if (fx == 2) {
                v = ((fy == 1 ? Bh(i, j) : Bh(i, j + 1)) + Jc(i, j) + 1) >> 1;
            } else if (fy == 2) {
                v = ((fx == 1 ? Hv(i, j) : Hv(i + 1, j)) + Jc(i, j) + 1) >> 1;
            } else {
                int bb = fy == 1 ? Bh(i, j) : Bh(i, j + 1);
                int hv = fx == 1 ? Hv(i, j) : Hv(i + 1, j);
                v = (bb + hv + 1) >> 1;
            }
            dst[j * dstride + i] = (uint8_t)v;
        }
#undef G
#undef Bh
#undef Hv
#undef Jc
}

/* chroma: (x,y) in chroma samples, mv in eighth chroma samples (= luma quarter mv) */
void h264o_mc_chroma(const uint8_t *ref, int stride, int w, int h, int x, int y, int mvx,
                     int mvy, int bw, int bh, uint8_t *dst, int dstride)
{
    int xi = x + (mvx >> 3), yi = y + (mvy >> 3), dx = mvx & 7, dy = mvy & 7;
    for (int j = 0; j < bh; j++)
        for (int i = 0; i < bw; i++) {
            int A = refpx(ref, stride, w, h, xi + i, yi + j), B = refpx(ref, stride, w, h, xi + i + 1, yi + j);
            int C = refpx(ref, stride, w, h, xi + i, yi + j + 1), D = refpx(ref, stride, w, h, xi + i + 1, yi + j + 1);
            dst[j * dstride + i] =
                (uint8_t)(((8 - dx) * (8 - dy) * A + dx * (8 - dy) * B + (8 - dx) * dy * C + dx * dy * D + 32) >> 6);
        }
}

/* ------------------------------------------------- 8.3.3 Intra16x16 pred */
void h264o_pred16x16(const uint8_t *rec, int stride, int mode, int avail, uint8_t pred[256])
{
    int left = avail & 1, top = (avail >> 1) & 1;
    const uint8_t *t = rec - stride;
    if (mode == 0) { /* vertical */
        for (int y = 0; y < 16; y++) memcpy(pred + 16 * y, t, 16);
    } else if (mode == 1) { /* horizontal */
        for (int y = 0; y < 16; y++) memset(pred + 16 * y, rec[y * stride - 1], 16);
    } else if (mode == 2) { /* DC */
        int s = 0, dc;
        if (top) for (int x = 0; x < 16; x++) s += t[x];
        if (left) for (int y = 0; y < 16; y++) s += rec[y * stride - 1];
        if (top && left) dc = (s + 16) >> 5;
        else if (top || left) dc = (s + 8) >> 4;
        else dc = 128;
        memset(pred, dc, 256);
    } else { /* plane */
        int H = 0, V = 0;
        for (int i = 0; i < 8; i++) {
            H += (i + 1) * (t[8 + i] - t[6 - i]); /* t[-1] is the top-left sample */
            V += (i + 1) * (rec[(8 + i) * stride - 1] - rec[(6 - i) * stride - 1]);
        }
        int a = 16 * (rec[15 * stride - 1] + t[15]);
        int b = (5 * H + 32) >> 6, c = (5 * V + 32) >> 6;
        for (int y = 0; y < 16; y++)
            for (int x = 0; x < 16; x++) pred[16 * y + x] = clip1((a + b * (x - 7) + c * (y - 7) + 16) >> 5);
    }
}

/* ------------------------------------------------- 8.3.1.2 Intra4x4 pred */
/* edge samples in one array: E[0..3] = left column bottom to top (p[-1,3] .. p[-1,0]), E[4] = p[-1,-1], E[5..12] = p[0..7,-1] */
int h264o_pred4x4(const uint8_t *rec, int stride, int mode, int avail, uint8_t pred[16])
{
    int left = avail & 1, top = (avail >> 1) & 1, tl = (avail >> 2) & 1, tr = (avail >> 3) & 1;
    int E[13];
    for (int y = 0; y < 4; y++) E[3 - y] = left ? rec[y * stride - 1] : 0;
    E[4] = tl ? rec[-stride - 1] : 0;
    for (int x = 0; x < 4; x++) E[5 + x] = top ? rec[-stride + x] : 0;
    for (int x = 4; x < 8; x++) E[5 + x] = top ? (tr ? rec[-stride + x] : rec[-stride + 3]) : 0;
#define T(x) E[5 + (x)]
#define L(y) E[3 - (y)]
    if ((mode == 0 || mode == 3 || mode == 7) && !top) return -1;
    if ((mode == 1 || mode == 8) && !left) return -1;
    if ((mode == 4 || mode == 5 || mode == 6) && !(top && left && tl)) return -1;
    for (int y = 0; y < 4; y++)
        for (int x = 0; x < 4; x++) {
            int v;
            switch (mode) {
                case 0: v = T(x); break;
                case 1: v = L(y); break;
                case 2:
                    if (top && left) v = (T(0) + T(1) + T(2) + T(3) + L(0) + L(1) + L(2) + L(3) + 4) >> 3;
                    else if (left) v = (L(0) + L(1) + L(2) + L(3) + 2) >> 2;
                    else if (top) v = (T(0) + T(1) + T(2) + T(3) + 2) >> 2;
                    else v = 128;
                    break;
                case 3: v = (x == 3 && y == 3) ? (T(6) + 3 * T(7) + 2) >> 2 : (T(x + y) + 2 * T(x + y + 1) + T(x + y + 2) + 2) >> 2; break;
                case 4: { int i = 4 + x - y; v = (E[i - 1] + 2 * E[i] + E[i + 1] + 2) >> 2; break; }   /* down-right: one filter along the edge array */
                case 5: {
                    int z = 2 * x - y, i = 4 + x - (y >> 1);
                    if (z >= 0 && !(z & 1)) v = (E[i] + E[i + 1] + 1) >> 1;
                    else if (z >= 0) v = (E[i - 1] + 2 * E[i] + E[i + 1] + 2) >> 2;
                    else if (z == -1) v = (L(0) + 2 * E[4] + T(0) + 2) >> 2;
                    else v = (L(y - 1) + 2 * L(y - 2) + L(y - 3) + 2) >> 2;
                    break;
                }
                case 6: {
                    int z = 2 * y - x, i = 4 - y + (x >> 1);
                    if (z >= 0 && !(z & 1)) v = (E[i - 1] + E[i] + 1) >> 1;
                    else if (z >= 0) v = (E[i - 1] + 2 * E[i] + E[i + 1] + 2) >> 2;
                    else if (z == -1) v = (L(0) + 2 * E[4] + T(0) + 2) >> 2;
                    else v = (T(x - 1) + 2 * T(x - 2) + T(x - 3) + 2) >> 2;
                    break;
                }
                case 7: {
                    int i = x + (y >> 1);
                    v = !(y & 1) ? (T(i) + T(i + 1) + 1) >> 1 : (T(i) + 2 * T(i + 1) + T(i + 2) + 2) >> 2;
                    break;
                }
                default: {
                    int z = x + 2 * y, i = y + (x >> 1);
                    if (z > 5) v = L(3);
                    else if (z == 5) v = (L(2) + 3 * L(3) + 2) >> 2;
                    else if (!(z & 1)) v = (L(i) + L(i + 1) + 1) >> 1;
                    else v = (L(i) + 2 * L(i + 1) + L(i + 2) + 2) >> 2;
                    break;
                }
            }
            pred[4 * y + x] = (uint8_t)v;
        }
#undef T
#undef L
    return 0;
}

/* 8.3.4 chroma 8x8 (4:2:0): 0 DC, 1 horizontal, 2 vertical, 3 plane */
void h264o_pred_chroma8x8(const uint8_t *rec, int stride, int mode, int avail, uint8_t pred[64])
{
    int left = avail & 1, top = (avail >> 1) & 1;
    const uint8_t *t = rec - stride;
    if (mode == 0) {
        for (int by = 0; by < 2; by++)
            for (int bx = 0; bx < 2; bx++) {
                int st = 0, sl = 0, dc;
                if (top) for (int x = 0; x < 4; x++) st += t[4 * bx + x];
                if (left) for (int y = 0; y < 4; y++) sl += rec[(4 * by + y) * stride - 1];
                if ((bx == 0 && by == 0) || (bx == 1 && by == 1)) {
                    if (top && left) dc = (st + sl + 4) >> 3;
                    else if (top) dc = (st + 2) >> 2;
                    else if (left) dc = (sl + 2) >> 2;
                    else dc = 128;
                } else if (bx == 1 && by == 0) { /* prefers top */
                    if (top) dc = (st + 2) >> 2;
                    else if (left) dc = (sl + 2) >> 2;
                    else dc = 128;
                } else { /* bx==0, by==1: prefers left */
                    if (left) dc = (sl + 2) >> 2;
                    else if (top) dc = (st + 2) >> 2;
                    else dc = 128;
                }
                for (int y = 0; y < 4; y++) memset(pred + 8 * (4 * by + y) + 4 * bx, dc, 4);
            }
    } else if (mode == 1) {
        for (int y = 0; y < 8; y++) memset(pred + 8 * y, rec[y * stride - 1], 8);
    } else if (mode == 2) {
        for (int y = 0; y < 8; y++) memcpy(pred + 8 * y, t, 8);
    } else {
        int H = 0, V = 0;
        for (int i = 0; i < 4; i++) {
            H += (i + 1) * (t[4 + i] - t[2 - i]);
            V += (i + 1) * (rec[(4 + i) * stride - 1] - rec[(2 - i) * stride - 1]);
        }
        int a = 16 * (rec[7 * stride - 1] + t[7]);
        int b = (34 * H + 32) >> 6, c = (34 * V + 32) >> 6;
        for (int y = 0; y < 8; y++)
            for (int x = 0; x < 8; x++) pred[8 * y + x] = clip1((a + b * (x - 3) + c * (y - 3) + 16) >> 5);
    }
}

/* ------------------------------------------------- 8.7 deblocking filter */
/* filter one line of samples across an edge; pix points at q0, xs = step across edge */
static void filter_line(uint8_t *pix, int xs, int bS, int alpha, int beta, int tc0, int chroma)
{
    int p0 = pix[-xs], p1 = pix[-2 * xs], q0 = pix[0], q1 = pix[xs];
    if (abs(p0 - q0) >= alpha || abs(p1 - p0) >= beta || abs(q1 - q0) >= beta) return;
    if (chroma) {
        if (bS < 4) {
            int tc = tc0 + 1;
            int d = clip3(-tc, tc, (((q0 - p0) << 2) + (p1 - q1) + 4) >> 3);
            pix[-xs] = clip1(p0 + d);
            pix[0] = clip1(q0 - d);
        } else {
            pix[-xs] = (uint8_t)((2 * p1 + p0 + q1 + 2) >> 2);
            pix[0] = (uint8_t)((2 * q1 + q0 + p1 + 2) >> 2);
        }
        return;
    }
    int p2 = pix[-3 * xs], q2 = pix[2 * xs];
    int ap = abs(p2 - p0), aq = abs(q2 - q0);
    if (bS < 4) {
        int tc = tc0 + (ap < beta) + (aq < beta);
        int d = clip3(-tc, tc, (((q0 - p0) << 2) + (p1 - q1) + 4) >> 3);
        pix[-xs] = clip1(p0 + d);
        pix[0] = clip1(q0 - d);
        if (ap < beta) pix[-2 * xs] = (uint8_t)(p1 + clip3(-tc0, tc0, (p2 + ((p0 + q0 + 1) >> 1) - (p1 << 1)) >> 1));
        if (aq < beta) pix[xs] = (uint8_t)(q1 + clip3(-tc0, tc0, (q2 + ((p0 + q0 + 1) >> 1) - (q1 << 1)) >> 1));
    } else {
        int p3 = pix[-4 * xs], q3 = pix[3 * xs];
        int strong = abs(p0 - q0) < ((alpha >> 2) + 2);
        if (ap < beta && strong) {
            pix[-xs] = (uint8_t)((p2 + 2 * p1 + 2 * p0 + 2 * q0 + q1 + 4) >> 3);
            pix[-2 * xs] = (uint8_t)((p2 + p1 + p0 + q0 + 2) >> 2);
            pix[-3 * xs] = (uint8_t)((2 * p3 + 3 * p2 + p1 + p0 + q0 + 4) >> 3);
        } else {
            pix[-xs] = (uint8_t)((2 * p1 + p0 + q1 + 2) >> 2);
        }
        if (aq < beta && strong) {
            pix[0] = (uint8_t)((p1 + 2 * p0 + 2 * q0 + 2 * q1 + q2 + 4) >> 3);
            pix[xs] = (uint8_t)((p0 + q0 + q1 + q2 + 2) >> 2);
            pix[2 * xs] = (uint8_t)((2 * q3 + 3 * q2 + q1 + q0 + p0 + 4) >> 3);
        } else {
            pix[0] = (uint8_t)((2 * q1 + q0 + p1 + 2) >> 2);
        }
    }
}

/* boundary strength between 4x4 block bq of macroblock q and bp of macroblock p
 * (8.7.2.1, frame pictures, single reference, one motion vector per MB) */
/* transform_size_8x8_flag of an inter macroblock rides in i16_mode (unused there); its residual blocks are 8x8: "contains
 * non-zero coefficients" (8.7.2.1) then refers to the 8x8 block, i.e. to the four interleaved 4x4 lists of the quadrant */
static int mb_t8x8(const h264o_mbinfo *m) { return (m->type == H264O_MB_P16 || m->type >= H264O_MB_P16X8) && m->i16_mode == 1; }
static int blk_nonzero(const h264o_mbinfo *m, int blk)
{
    if (!mb_t8x8(m)) return m->tc[blk] != 0;
    int q = blk & ~3;
    return (m->tc[q] | m->tc[q + 1] | m->tc[q + 2] | m->tc[q + 3]) != 0;
}
/* pv / qv: the four quadrant vectors of the two macroblocks (8 int16 each); block b lies in quadrant b >> 2 */
static int edge_bs(const h264o_mbinfo *p, const int16_t *pv, int bp, const h264o_mbinfo *q, const int16_t *qv, int bq, int mb_edge)
{
    if (H264O_MB_IS_INTRA(p->type) || H264O_MB_IS_INTRA(q->type)) return mb_edge ? 4 : 3;
    if (blk_nonzero(p, bp) || blk_nonzero(q, bq)) return 2;
    if (p->chroma_mode != q->chroma_mode) return 1;   /* different reference pictures (ref_idx_l0 rides in chroma_mode; one list, never reordered) */
    if (abs(pv[2 * (bp >> 2)] - qv[2 * (bq >> 2)]) >= 4 || abs(pv[2 * (bp >> 2) + 1] - qv[2 * (bq >> 2) + 1]) >= 4) return 1;
    return 0;
}

/* raster (x4,y4) -> blkIdx */
static const uint8_t xy2blk[16] = {0, 1, 4, 5, 2, 3, 6, 7, 8, 9, 12, 13, 10, 11, 14, 15};

void h264o_deblock_picture(uint8_t *Y, uint8_t *U, uint8_t *V, int cw, int ch,
                           const h264o_mbinfo *mbs, const int16_t *mvq, int qp, const int16_t *slice_of, int row0, int row1)
{
    int mbw = cw / 16, mbh = ch / 16;
    int qpc = o_chroma_qp[clip3(0, 51, qp)];
    /* all macroblocks share one QP in this encoder; indexA = indexB = qPav (offsets 0) */
    int aY = o_alpha[qp], bY = o_beta[qp], aC = o_alpha[qpc], bC = o_beta[qpc];
    if (row1 > mbh) row1 = mbh;
    for (int my = row0; my < row1; my++)
        for (int mx = 0; mx < mbw; mx++) {
            const h264o_mbinfo *q = &mbs[my * mbw + mx];
            /* vertical edges, left to right */
            for (int e = 0; e < 4; e++) {
                if (e == 0 && mx == 0) continue;
                if ((e & 1) && mb_t8x8(q)) continue;   /* 8x8 transform: no 4x4-internal edges */
                /* disable_deblocking_filter_idc 2: edges between slices stay unfiltered (slice_of given) */
                if (e == 0 && slice_of && slice_of[my * mbw + mx] != slice_of[my * mbw + mx - 1]) continue;
                const h264o_mbinfo *p = e == 0 ? q - 1 : q;
                const int16_t *qv = mvq + (size_t)(my * mbw + mx) * 8, *pv = e == 0 ? qv - 8 : qv;
                for (int r = 0; r < 4; r++) { /* four rows of 4x4 blocks */
                    int bq = xy2blk[4 * r + e], bp = e == 0 ? xy2blk[4 * r + 3] : xy2blk[4 * r + e - 1];
                    int bS = edge_bs(p, pv, bp, q, qv, bq, e == 0);
                    if (!bS) continue;
                    for (int k = 0; k < 4; k++)
                        filter_line(Y + (16 * my + 4 * r + k) * cw + 16 * mx + 4 * e, 1, bS, aY, bY,
                                    bS < 4 ? o_tc0[qp][bS - 1] : 0, 0);
                    if (!(e & 1))
                        for (int k = 0; k < 2; k++) {
                            int off = (8 * my + 2 * r + k) * (cw / 2) + 8 * mx + 2 * e;
                            filter_line(U + off, 1, bS, aC, bC, bS < 4 ? o_tc0[qpc][bS - 1] : 0, 1);
                            filter_line(V + off, 1, bS, aC, bC, bS < 4 ? o_tc0[qpc][bS - 1] : 0, 1);
                        }
                }
            }
            /* horizontal edges, top to bottom */
            for (int e = 0; e < 4; e++) {
                if (e == 0 && my == 0) continue;
                if ((e & 1) && mb_t8x8(q)) continue;
                if (e == 0 && slice_of && slice_of[my * mbw + mx] != slice_of[(my - 1) * mbw + mx]) continue;
                const h264o_mbinfo *p = e == 0 ? q - mbw : q;
                const int16_t *qv = mvq + (size_t)(my * mbw + mx) * 8, *pv = e == 0 ? qv - 8 * mbw : qv;
                for (int c = 0; c < 4; c++) {
                    int bq = xy2blk[4 * e + c], bp = e == 0 ? xy2blk[12 + c] : xy2blk[4 * (e - 1) + c];
                    int bS = edge_bs(p, pv, bp, q, qv, bq, e == 0);
                    if (!bS) continue;
                    for (int k = 0; k < 4; k++)
                        filter_line(Y + (16 * my + 4 * e) * cw + 16 * mx + 4 * c + k, cw, bS, aY, bY,
                                    bS < 4 ? o_tc0[qp][bS - 1] : 0, 0);
                    if (!(e & 1))
                        for (int k = 0; k < 2; k++) {
                            int off = (8 * my + 2 * e) * (cw / 2) + 8 * mx + 2 * c + k;
                            filter_line(U + off, cw / 2, bS, aC, bC, bS < 4 ? o_tc0[qpc][bS - 1] : 0, 1);
                            filter_line(V + off, cw / 2, bS, aC, bC, bS < 4 ? o_tc0[qpc][bS - 1] : 0, 1);
                        }
                }
            }
        }
}

/* ------------------------------------------------- Exp-Golomb helpers */
int h264o_ue_bits(uint32_t v, uint32_t *code)
{
    uint32_t x = v + 1;
    int n = 0;
    while ((x >> n) > 1) n++;
    *code = x; /* n leading zeros then the (n+1)-bit value x */
    return 2 * n + 1;
}
int h264o_se_bits(int32_t v, uint32_t *code)
{
    uint32_t k = v > 0 ? (uint32_t)(2 * v - 1) : (uint32_t)(-2 * v);
    return h264o_ue_bits(k, code);
}

/* 7.4.1.1 emulation prevention: returns escaped length */
size_t h264o_nal_escape(const uint8_t *rbsp, size_t n, uint8_t *out)
{
    size_t o = 0;
    int zeros = 0;
    for (size_t i = 0; i < n; i++) {
        if (zeros == 2 && rbsp[i] <= 3) {
            out[o++] = 3;
            zeros = 0;
        }
        out[o++] = rbsp[i];
        zeros = rbsp[i] == 0 ? zeros + 1 : 0;
    }
    return o;
}
