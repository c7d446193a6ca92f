// media_amd/csrc/k_dec.h -- reconstruction kernels of the decoder peer (row f4 of SURVEY.md 8; the interface it serves is
// /root/reference/video_decoder/include/VideoDecoder.h:83).  The host parser (h264_parse.h) has filled MbInfo, the quadrant
// vectors, the Intra4x4 modes and the level lists - the very arrays the encoder's kernels exchange - so that decoding is the
// encoder's own reconstruction path run from given decisions:
//   k_dec_inter   motion-compensated prediction of every inter macroblock (8.4.2.2: luma by the 6-tap / bilinear quarter-
//                 sample rules straight from the reference plane, chroma by the 1/8-sample bilinear rule), written into the
//                 picture; lane = (row, four samples), one wave per macroblock; a vector per 4x4 block and a reference
//                 picture per 8x8 quadrant (the parser's mv4 / refq: partitions down to 4x4, 7.3.5.2)
//   k_dec_widen   the levels arrive as one byte each (half the upload): widened into the int16 level lists the kernels read;
//                 k_dec_patch puts the few values that did not fit a byte in place
//   k_dec_bs      boundary strengths (8.7.2.1) from those arrays - the encoder's k_bs knows vectors per quadrant only
//   k_dec_resid   scaling + inverse transform (4x4: 8.5.12, 8x8: 8.5.13, chroma DC: 8.5.11) of the inter macroblocks' levels,
//                 added to the prediction in place; lane = one 4x4 block, four macroblocks per wave
//   k_pintra_rows<true> (k_intra.h)  the intra macroblocks in row-wavefront order, intra_mb_core<true>
//   k_deblock_rows (encoder)  the loop filter (PERMB = true when QPs or offsets differ inside the picture)
#pragma once
#include "dev_common.h"
#include "mc_filters.h"
#include "k_me.h"   // chroma_pred4
#include "k_tq.h"   // idct8_line, pos_class8, recon4

namespace h264 {

// four luma samples (x .. x + 3, y) of the prediction with quarter-sample vector (vx, vy) from plane R (pitch cw, ch rows);
// samples outside the picture repeat the edge (8.4.2.2.1, Table 8-12 spelled out)
__device__ __forceinline__ uint32_t mc_luma4(const uint8_t* R, int cw, int ch, int x, int y, int vx, int vy)
{
    const int ix = x + (vx >> 2), iy = y + (vy >> 2), fx = vx & 3, fy = vy & 3;
    int G[6][9];   // rows iy - 2 .. iy + 3, columns ix - 2 .. ix + 6
    const bool inside = ix - 2 >= 0 && ((ix - 2) & ~3) + 12 <= cw;
#pragma unroll
    for (int r = 0; r < 6; r++) {
        const uint8_t* row = R + (size_t)clip3(0, ch - 1, iy - 2 + r) * cw;
        if (inside) {
            const int xa = (ix - 2) & ~3, sh = (ix - 2) & 3;
            const uint32_t a = *(const uint32_t*)(row + xa), b = *(const uint32_t*)(row + xa + 4), c = *(const uint32_t*)(row + xa + 8);
            const uint32_t w0 = __builtin_amdgcn_alignbyte(b, a, sh), w1 = __builtin_amdgcn_alignbyte(c, b, sh), w2 = c >> (8 * sh);
#pragma unroll
            for (int k = 0; k < 4; k++) { G[r][k] = (int)byte_of(w0, k); G[r][4 + k] = (int)byte_of(w1, k); }
            G[r][8] = (int)(w2 & 255u);
        } else {
#pragma unroll
            for (int k = 0; k < 9; k++) G[r][k] = row[clip3(0, cw - 1, ix - 2 + k)];
        }
    }
    uint32_t out = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        auto b1 = [&](int r) { return G[r][i] - 5 * G[r][i + 1] + 20 * G[r][i + 2] + 20 * G[r][i + 3] - 5 * G[r][i + 4] + G[r][i + 5]; };
        auto h1 = [&](int c) { return G[0][c] - 5 * G[1][c] + 20 * G[2][c] + 20 * G[3][c] - 5 * G[4][c] + G[5][c]; };
        const int g0 = G[2][i + 2];
        int v;
        if ((fx | fy) == 0) v = g0;
        else if (fy == 0) {
            const int b = clip255((b1(2) + 16) >> 5);
            v = fx == 2 ? b : ((fx == 1 ? g0 : G[2][i + 3]) + b + 1) >> 1;
        } else if (fx == 0) {
            const int h = clip255((h1(i + 2) + 16) >> 5);
            v = fy == 2 ? h : ((fy == 1 ? g0 : G[3][i + 2]) + h + 1) >> 1;
        } else if (fx != 2 && fy != 2) {   // e, g, p, r: a horizontal and a vertical half sample
            // (both candidates are evaluated and one is selected: an index that depends on the vector would put G into scratch memory)
            const int b = clip255(((fy == 1 ? b1(2) : b1(3)) + 16) >> 5), h = clip255(((fx == 1 ? h1(i + 2) : h1(i + 3)) + 16) >> 5);
            v = (b + h + 1) >> 1;
        } else {
            const int j = clip255((b1(0) - 5 * b1(1) + 20 * b1(2) + 20 * b1(3) - 5 * b1(4) + b1(5) + 512) >> 10);
            if (fx == 2 && fy == 2) v = j;
            else if (fx == 2) v = (clip255(((fy == 1 ? b1(2) : b1(3)) + 16) >> 5) + j + 1) >> 1;     // f, q
            else v = (clip255(((fx == 1 ? h1(i + 2) : h1(i + 3)) + 16) >> 5) + j + 1) >> 1;           // i, k
        }
        out |= (uint32_t)v << (8 * i);
    }
    return out;
}

// the plane of reference picture ref_idx (0 .. 2) and component pl; the pointers pass through an opaque no-op so that the
// selection is a select between registers, not an indexed load of the parameter block (dev_common.h rec_chroma)
__device__ __forceinline__ const uint8_t* ref_plane(const FrameParams& P, int ref, int pl)
{
    const uint8_t *a = P.refs[0][0], *b = P.refs[1][0], *c = P.refs[2][0];
    const uint8_t *au = P.refs[0][1], *bu = P.refs[1][1], *cu = P.refs[2][1];
    const uint8_t *av = P.refs[0][2], *bv = P.refs[1][2], *cv = P.refs[2][2];
    asm volatile("" : "+s"(a), "+s"(b), "+s"(c), "+s"(au), "+s"(bu), "+s"(cu), "+s"(av), "+s"(bv), "+s"(cv));
    const uint8_t* y = ref == 0 ? a : (ref == 1 ? b : c);
    const uint8_t* u = ref == 0 ? au : (ref == 1 ? bu : cu);
    const uint8_t* v = ref == 0 ? av : (ref == 1 ? bv : cv);
    return pl == 0 ? y : (pl == 1 ? u : v);
}

__global__ __launch_bounds__(64) void k_dec_inter(FrameParams P0)
{
    const FrameParams P = batch_view(P0, blockIdx.y);
    const int lane = threadIdx.x;
    const int mbi = P.band.row0 * P.mbw + (int)blockIdx.x, my = P.mbdiv.row(mbi), mx = mbi - my * P.mbw;
    const uint32_t w1 = *(const uint32_t*)((const uint8_t*)(P.mb + mbi) + 4);
    if (mb_is_intra((int)(w1 & 255u))) return;
    const uint32_t* mv = (const uint32_t*)(P.mv4 + (size_t)mbi * 32);   // [4 * by + bx] = x | y << 16
    const uint32_t refs = *(const uint32_t*)(P.refq + (size_t)mbi * 4);
    {
        const int y = lane >> 2, seg = (lane & 3) * 4;
        const uint32_t v = mv[4 * (y >> 2) + (seg >> 2)];
        const int ref = (int)((refs >> (8 * (2 * (y >> 3) + (seg >> 3)))) & 255u);
        *(uint32_t*)(P.rec[0] + (size_t)(16 * my + y) * P.cw + 16 * mx + seg) =
            mc_luma4(ref_plane(P, ref, 0), P.cw, P.ch, 16 * mx + seg, 16 * my + y, (int)(int16_t)(v & 0xFFFFu), (int)(int16_t)(v >> 16));
    }
    if (lane < 32) {
        // four chroma samples of one row = the chroma of two 4x4 luma blocks (2 x 2 samples each), which may move differently
        const int pl = lane >> 4, cyy = (lane >> 1) & 7, cxx = (lane & 1) * 4;
        const int b = 4 * (cyy >> 1) + (cxx >> 1);
        const uint32_t va = mv[b], vb = mv[b + 1];
        const int ref = (int)((refs >> (8 * (2 * (cyy >> 2) + (cxx >> 2)))) & 255u);
        const uint8_t* R = ref_plane(P, ref, 1 + pl);
        auto pred = [&](uint32_t v) {
            const int vx = (int)(int16_t)(v & 0xFFFFu), vy = (int)(int16_t)(v >> 16);
            return chroma_pred4(R, P.cw / 2, P.ch / 2, 8 * mx + cxx + (vx >> 3), 8 * my + cyy + (vy >> 3), vx & 7, vy & 7);
        };
        uint32_t o = pred(va);
        if (vb != va) o = (o & 0xFFFFu) | (pred(vb) & 0xFFFF0000u);
        *(uint32_t*)(rec_chroma(P, pl) + (size_t)(8 * my + cyy) * (P.cw / 2) + 8 * mx + cxx) = o;
    }
}

// lane = four consecutive levels of one macroblock (LV_STRIDE = 416 = 104 words per macroblock); an I_PCM macroblock's area holds
// its 384 samples as bytes, which keep their place at the start of the int16 area (intra_mb_core<DEC> reads them there)
__global__ __launch_bounds__(256) void k_dec_widen(const uint32_t* lv8, const MbInfo* mb, int16_t* lv16, int nmb)
{
    const int i = (int)(blockIdx.x * 256 + threadIdx.x);   // word index
    if (i >= nmb * (LV_STRIDE / 4)) return;
    const int mbi = i / (LV_STRIDE / 4), k = i - mbi * (LV_STRIDE / 4);
    const uint32_t w = lv8[i];
    if (mb[mbi].type == MB_IPCM) {
        if (k < 96) ((uint32_t*)(lv16 + (size_t)mbi * LV_STRIDE))[k] = w;
        return;
    }
    const int a = (int)(int8_t)(w & 255u), b = (int)(int8_t)((w >> 8) & 255u), c = (int)(int8_t)((w >> 16) & 255u), d = (int)(int8_t)(w >> 24);
    uint2 o;
    o.x = ((uint32_t)a & 0xFFFFu) | ((uint32_t)b << 16);
    o.y = ((uint32_t)c & 0xFFFFu) | ((uint32_t)d << 16);
    ((uint2*)(lv16 + (size_t)mbi * LV_STRIDE))[k] = o;
}
struct DecBigLevel { uint32_t idx; int32_t val; };
__global__ __launch_bounds__(256) void k_dec_patch(const DecBigLevel* big, int n, int16_t* lv16)
{
    const int i = (int)(blockIdx.x * 256 + threadIdx.x);
    if (i < n) lv16[big[i].idx] = (int16_t)big[i].val;
}

// 8.7.2.1 for the decoder: lane = (macroblock of the pair, direction, edge, segment); vectors per 4x4 block, references per
// quadrant (every slice of the picture has the same list of distinct pictures: equal indices = the same picture).  Writes the layout k_bs writes.
struct DecBsParams {
    const MbInfo* mb;
    const int16_t* mv4;
    const uint8_t* refq;
    uint8_t* bs;
    int mbw, nmb;
    const uint8_t* mbavail;   // the parser's availability bits (bit 0 left, 1 above: the same slice)
    int across;        // disable_deblocking_filter_idc 0: edges between slices are filtered too
    MbDiv mbdiv;
};
__global__ __launch_bounds__(64) void k_dec_bs(DecBsParams C, unsigned* anybs, unsigned serial)
{
    const int lane = threadIdx.x, l = lane & 31, mbi = 2 * (int)blockIdx.x + (lane >> 5);
    int bs = 0;
    if (mbi < C.nmb) {
        const int my = C.mbdiv.row(mbi), mx = mbi - my * C.mbw;
        const int dir = l >> 4, e = (l >> 2) & 3, k = l & 3;
        const int av = C.mbavail[mbi];
        const bool edge_ok = e != 0 || (dir == 0 ? (C.across ? mx > 0 : (av & 1) != 0) : (C.across ? my > 0 : (av & 2) != 0));
        if (edge_ok) {
            const int pi = e == 0 ? (dir == 0 ? mbi - 1 : mbi - C.mbw) : mbi;
            const MbInfo *q = C.mb + mbi, *p = C.mb + pi;
            const int qx = dir == 0 ? e : k, qy = dir == 0 ? k : e;                                  // 4x4 raster position of q's block
            const int px = dir == 0 ? (e == 0 ? 3 : e - 1) : k, py = dir == 0 ? k : (e == 0 ? 3 : e - 1);
            const int bq = xy2blk(qx, qy), bp = xy2blk(px, py);
            if (mb_is_intra(p->type) || mb_is_intra(q->type)) bs = e == 0 ? 4 : 3;
            else {
                const bool q8 = (q->type == MB_P16 || q->type >= MB_P16X8) && q->i16_mode == 1, p8 = (p->type == MB_P16 || p->type >= MB_P16X8) && p->i16_mode == 1;
                if (!(q8 && (e & 1))) {
                    const bool nzq = q8 ? (*(const uint32_t*)(q->tc + (bq & ~3)) != 0) : q->tc[bq] != 0;
                    const bool nzp = p8 ? (*(const uint32_t*)(p->tc + (bp & ~3)) != 0) : p->tc[bp] != 0;
                    if (nzp || nzq) bs = 2;
                    else if (C.refq[(size_t)pi * 4 + 2 * (py >> 1) + (px >> 1)] != C.refq[(size_t)mbi * 4 + 2 * (qy >> 1) + (qx >> 1)]) bs = 1;
                    else {
                        const uint32_t vp = *(const uint32_t*)(C.mv4 + (size_t)pi * 32 + 2 * (4 * py + px)), vq = *(const uint32_t*)(C.mv4 + (size_t)mbi * 32 + 2 * (4 * qy + qx));
                        bs = (iabs((int)(int16_t)(vp & 0xFFFFu) - (int)(int16_t)(vq & 0xFFFFu)) >= 4 || iabs((int)(int16_t)(vp >> 16) - (int)(int16_t)(vq >> 16)) >= 4) ? 1 : 0;
                    }
                }
            }
        }
        C.bs[(size_t)mbi * 32 + l] = (uint8_t)bs;
    }
    if (__ballot(bs != 0) != 0ull && lane == 0) anybs[blockIdx.y] = serial;
}

__global__ __launch_bounds__(64) void k_dec_resid(FrameParams P0)
{
    const FrameParams P = batch_view(P0, blockIdx.y);
    const int lane = threadIdx.x, blk = lane & 15;
    const int first = P.band.row0 * P.mbw, end = first + P.band.rows * P.mbw;
    const int mbi = first + 4 * (int)blockIdx.x + (lane >> 4);
    if (mbi >= end) return;
    const uint32_t w1 = *(const uint32_t*)((const uint8_t*)(P.mb + mbi) + 4);
    const int type = (int)(w1 & 255u), cbp = (int)(w1 >> 24);
    if (mb_is_intra(type) || type == MB_PSKIP || cbp == 0) return;
    const bool t8 = ((w1 >> 8) & 1u) != 0;
    const int my = P.mbdiv.row(mbi), mx = mbi - my * P.mbw, cs = P.cw / 2;
    const int16_t* lv = P.levels + (size_t)mbi * LV_STRIDE;
    const int qp = (int)P.mbqp[mbi];   // the macroblock's own QP_Y (lanes of four macroblocks: a per-lane value)
    if (cbp & (1 << (blk >> 2))) {
        if (!t8) {
            int dq[3];
            dec_dq(qp, dq);
            int d[16];
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const int k = pos_class(i);
                d[i] = (int)lv[LV_LUMA + blk * 16 + c_zigzag_inv[i]] * (k == 0 ? dq[0] : (k == 1 ? dq[1] : dq[2]));
            }
            idct4x4(d);
            uint8_t* dst = P.rec[0] + (size_t)(16 * my + 4 * blk_y(blk)) * P.cw + 16 * mx + 4 * blk_x(blk);
#pragma unroll
            for (int r = 0; r < 4; r++) {
                uint32_t* q = (uint32_t*)(dst + (size_t)r * P.cw);
                *q = recon4(*q, d[4 * r], d[4 * r + 1], d[4 * r + 2], d[4 * r + 3]);
            }
        } else if ((blk & 3) == 0) {
            // 8x8 block blk >> 2: coefficient k of the 8x8 zig-zag scan is element k >> 2 of the quadrant's list k & 3 (7.3.5.3.2)
            constexpr int ZZ8[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6,  7,  14, 21, 28,
                                     35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
            const int b8 = blk >> 2, qp6 = qp / 6, qm = qp % 6;
            int ls[6];
#pragma unroll
            for (int c = 0; c < 6; c++) ls[c] = 16 * (int)c_dequant8_v[qm][c];
            int d[64];
#pragma unroll
            for (int k = 0; k < 64; k++) {
                const int pos = ZZ8[k];
                const int t = (int)lv[LV_LUMA + (4 * b8 + (k & 3)) * 16 + (k >> 2)] * ls[pos_class8(pos)];
                d[pos] = qp6 >= 6 ? t << (qp6 - 6) : (t + (1 << (5 - qp6))) >> (6 - qp6);
            }
#pragma unroll
            for (int r = 0; r < 8; r++) idct8_line<1>(d + 8 * r);
#pragma unroll
            for (int c = 0; c < 8; c++) idct8_line<8>(d + c);
            uint8_t* dst = P.rec[0] + (size_t)(16 * my + 8 * (b8 >> 1)) * P.cw + 16 * mx + 8 * (b8 & 1);
#pragma unroll
            for (int r = 0; r < 8; r++) {
                uint32_t* q = (uint32_t*)(dst + (size_t)r * P.cw);
                int* e = d + 8 * r;
                q[0] = recon4(q[0], (e[0] + 32) >> 6, (e[1] + 32) >> 6, (e[2] + 32) >> 6, (e[3] + 32) >> 6);
                q[1] = recon4(q[1], (e[4] + 32) >> 6, (e[5] + 32) >> 6, (e[6] + 32) >> 6, (e[7] + 32) >> 6);
            }
        }
    }
    if ((cbp >> 4) && blk < 8) {   // chroma: lane = (plane, block)
        const int pl = blk >> 2, cb = blk & 3;
        int dc[4];
#pragma unroll
        for (int i = 0; i < 4; i++) dc[i] = (int)lv[LV_CHROMA_DC + 4 * pl + i];
        const int fi[4] = {dc[0] + dc[1] + dc[2] + dc[3], dc[0] - dc[1] + dc[2] - dc[3], dc[0] + dc[1] - dc[2] - dc[3], dc[0] - dc[1] - dc[2] + dc[3]};
        int dq[3];
        dec_dq(dec_qpc(P, qp, pl), dq);
        int d[16];
#pragma unroll
        for (int i = 1; i < 16; i++) {
            const int k = pos_class(i);
            d[i] = (int)lv[LV_CHROMA_AC + (4 * pl + cb) * 16 + c_zigzag_inv[i]] * (k == 0 ? dq[0] : (k == 1 ? dq[1] : dq[2]));
        }
        d[0] = ((cb == 0 ? fi[0] : (cb == 1 ? fi[1] : (cb == 2 ? fi[2] : fi[3]))) * 16 * dq[0]) >> 5;
        idct4x4(d);
        uint8_t* dst = rec_chroma(P, pl) + (size_t)(8 * my + 4 * (cb >> 1)) * cs + 8 * mx + 4 * (cb & 1);
#pragma unroll
        for (int r = 0; r < 4; r++) {
            uint32_t* q = (uint32_t*)(dst + (size_t)r * cs);
            *q = recon4(*q, d[4 * r], d[4 * r + 1], d[4 * r + 2], d[4 * r + 3]);
        }
    }
}

}  // namespace h264
