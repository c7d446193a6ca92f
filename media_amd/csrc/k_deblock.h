// media_amd/csrc/k_deblock.h -- in-loop deblocking filter (H.264 8.7), one
// wavefront per macroblock, launched one wavefront step (mx + 2*my == s) at a
// time: macroblock (mx,my) filters against samples already finished by its left,
// top and top-right neighbours, which 8.7 orders strictly (vertical edges of a
// macroblock, then its horizontal edges, macroblocks in raster order).
//
// SURVEY.md 8a row a6.4 (iLoopFilterDisableIdc = 0 at
// /root/reference/video_codec/VideoEncoderOpenH264.cpp:295; alpha/beta offsets 0).
//
// Each step's macroblocks touch disjoint samples: own 16x16 (+8x8 chroma), the
// left neighbour's last 3 luma / 1 chroma columns, the top neighbour's last 3
// luma / 1 chroma rows.
#pragma once
#include "dev_common.h"

namespace h264 {

__constant__ const uint8_t c_alpha[52] = {0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,   0,   0,   0,   4,   4,
                                          5,  6,  7,  8,  9,  10, 12, 13, 15, 17, 20, 22, 25,  28,  32,  36,  40,  45,
                                          50, 56, 63, 71, 80, 90, 101, 113, 127, 144, 162, 182, 203, 226, 255, 255};
__constant__ const uint8_t c_beta[52] = {0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  2,  2,
                                         2,  3,  3,  3,  3,  4,  4,  4,  6,  6,  7,  7,  8,  8,  9,  9,  10, 10,
                                         11, 11, 12, 12, 13, 13, 14, 14, 15, 15, 16, 16, 17, 17, 18, 18};
__constant__ const uint8_t c_tc0[52][3] = {
    {0, 0, 0},   {0, 0, 0},   {0, 0, 0},    {0, 0, 0},    {0, 0, 0},    {0, 0, 0},   {0, 0, 0},   {0, 0, 0},  {0, 0, 0},
    {0, 0, 0},   {0, 0, 0},   {0, 0, 0},    {0, 0, 0},    {0, 0, 0},    {0, 0, 0},   {0, 0, 0},   {0, 0, 0},  {0, 0, 1},
    {0, 0, 1},   {0, 0, 1},   {0, 0, 1},    {0, 1, 1},    {0, 1, 1},    {1, 1, 1},   {1, 1, 1},   {1, 1, 1},  {1, 1, 1},
    {1, 1, 2},   {1, 1, 2},   {1, 1, 2},    {1, 1, 2},    {1, 2, 3},    {1, 2, 3},   {2, 2, 3},   {2, 2, 4},  {2, 3, 4},
    {2, 3, 4},   {3, 3, 5},   {3, 4, 6},    {3, 4, 6},    {4, 5, 7},    {4, 5, 8},   {4, 6, 9},   {5, 7, 10}, {6, 8, 11},
    {6, 8, 13},  {7, 10, 14}, {8, 11, 16},  {9, 12, 18},  {10, 13, 20}, {11, 15, 23}, {13, 17, 25}};

struct DbParams {
    uint8_t* pl[3];      // picture being filtered in place
    const MbInfo* mb;
    int cw, ch, mbw, mbh;
    int alpha_y, beta_y, alpha_c, beta_c;
    int tc0_y[3], tc0_c[3];  // by bS-1
};

// filter one line across an edge; p points at q0 inside LDS, xs = distance between samples across the edge
__device__ __forceinline__ void filter_line(uint8_t* pix, int xs, int bS, int alpha, int beta, int tc0, bool chroma)
{
    const int p0 = pix[-xs], p1 = pix[-2 * xs], q0 = pix[0], q1 = pix[xs];
    if (iabs(p0 - q0) >= alpha || iabs(p1 - p0) >= beta || iabs(q1 - q0) >= beta) return;
    if (chroma) {
        if (bS < 4) {
            const int tc = tc0 + 1;
            const int d = clip3(-tc, tc, (((q0 - p0) << 2) + (p1 - q1) + 4) >> 3);
            pix[-xs] = (uint8_t)clip255(p0 + d);
            pix[0] = (uint8_t)clip255(q0 - d);
        } else {
            pix[-xs] = (uint8_t)((2 * p1 + p0 + q1 + 2) >> 2);
            pix[0] = (uint8_t)((2 * q1 + q0 + p1 + 2) >> 2);
        }
        return;
    }
    const int p2 = pix[-3 * xs], q2 = pix[2 * xs];
    const int ap = iabs(p2 - p0), aq = iabs(q2 - q0);
    if (bS < 4) {
        const int tc = tc0 + (ap < beta) + (aq < beta);
        const int d = clip3(-tc, tc, (((q0 - p0) << 2) + (p1 - q1) + 4) >> 3);
        pix[-xs] = (uint8_t)clip255(p0 + d);
        pix[0] = (uint8_t)clip255(q0 - d);
        if (ap < beta) pix[-2 * xs] = (uint8_t)(p1 + clip3(-tc0, tc0, (p2 + ((p0 + q0 + 1) >> 1) - (p1 << 1)) >> 1));
        if (aq < beta) pix[xs] = (uint8_t)(q1 + clip3(-tc0, tc0, (q2 + ((p0 + q0 + 1) >> 1) - (q1 << 1)) >> 1));
    } else {
        const int p3 = pix[-4 * xs], q3 = pix[3 * xs];
        const bool strong = iabs(p0 - q0) < ((alpha >> 2) + 2);
        if (ap < beta && strong) {
            pix[-xs] = (uint8_t)((p2 + 2 * p1 + 2 * p0 + 2 * q0 + q1 + 4) >> 3);
            pix[-2 * xs] = (uint8_t)((p2 + p1 + p0 + q0 + 2) >> 2);
            pix[-3 * xs] = (uint8_t)((2 * p3 + 3 * p2 + p1 + p0 + q0 + 4) >> 3);
        } else pix[-xs] = (uint8_t)((2 * p1 + p0 + q1 + 2) >> 2);
        if (aq < beta && strong) {
            pix[0] = (uint8_t)((p1 + 2 * p0 + 2 * q0 + 2 * q1 + q2 + 4) >> 3);
            pix[xs] = (uint8_t)((p0 + q0 + q1 + q2 + 2) >> 2);
            pix[2 * xs] = (uint8_t)((2 * q3 + 3 * q2 + q1 + q0 + p0 + 4) >> 3);
        } else pix[0] = (uint8_t)((2 * q1 + q0 + p1 + 2) >> 2);
    }
}

// 8.7.2.1 boundary strength (frame macroblocks, one reference, one vector per MB)
__device__ __forceinline__ int edge_bs(const MbInfo* p, int bp, const MbInfo* q, int bq, bool mb_edge)
{
    if (p->type == MB_I16 || q->type == MB_I16) return mb_edge ? 4 : 3;
    if (p->tc[bp] || q->tc[bq]) return 2;
    if (iabs(p->mvx - q->mvx) >= 4 || iabs(p->mvy - q->mvy) >= 4) return 1;
    return 0;
}

enum { DB_LP = 24, DB_CP = 12 };  // LDS pitches: luma 20 wide, chroma 10 wide (4 / 2 apron)

__global__ __launch_bounds__(64) void k_deblock_diag(DbParams D, int s)
{
    const int lane = threadIdx.x;
    // macroblocks with mx + 2*my == s
    const int ymin = max(0, (s - (D.mbw - 1) + 1) >> 1);
    const int my = ymin + blockIdx.x, mx = s - 2 * my;
    if (my >= D.mbh || mx < 0 || mx >= D.mbw) return;
    const int cs = D.cw / 2;
    const MbInfo* q = D.mb + (size_t)my * D.mbw + mx;

    __shared__ __attribute__((aligned(16))) uint8_t s_y[20 * DB_LP];      // rows/cols -4..15
    __shared__ __attribute__((aligned(16))) uint8_t s_c[2][10 * DB_CP];   // rows/cols -2..7
    __shared__ uint8_t s_bs[2][16];  // [dir][edge*4 + segment]

    // load 20x20 luma (as 5 dwords per row) and 10x10 chroma; out-of-picture apron left as is (never filtered)
    for (int i = lane; i < 20 * 5; i += 64) {
        const int r = i / 5, c = i - r * 5;
        const int gy = 16 * my - 4 + r, gx = 16 * mx - 4 + c * 4;
        uint32_t v = 0;
        if (gy >= 0 && gx >= 0) v = *(const uint32_t*)(D.pl[0] + (size_t)gy * D.cw + gx);
        *(uint32_t*)(s_y + r * DB_LP + c * 4) = v;
    }
    for (int i = lane; i < 2 * 10 * 5; i += 64) {
        const int pl = i / 50, k = i - pl * 50, r = k / 5, c = k - r * 5;
        const int gy = 8 * my - 2 + r, gx = 8 * mx - 2 + c * 2;
        uint16_t v = 0;
        if (gy >= 0 && gx >= 0) v = *(const uint16_t*)(D.pl[1 + pl] + (size_t)gy * cs + gx);
        *(uint16_t*)(s_c[pl] + r * DB_CP + c * 2) = v;
    }
    if (lane < 32) {
        const int dir = lane >> 4, e = (lane >> 2) & 3, k = lane & 3;  // k: segment along the edge
        int bS = 0;
        if (dir == 0) {
            if (!(e == 0 && mx == 0)) {
                const MbInfo* p = e == 0 ? q - 1 : q;
                bS = edge_bs(p, e == 0 ? xy2blk(3, k) : xy2blk(e - 1, k), q, xy2blk(e, k), e == 0);
            }
        } else {
            if (!(e == 0 && my == 0)) {
                const MbInfo* p = e == 0 ? q - D.mbw : q;
                bS = edge_bs(p, e == 0 ? xy2blk(k, 3) : xy2blk(k, e - 1), q, xy2blk(k, e), e == 0);
            }
        }
        s_bs[dir][e * 4 + k] = (uint8_t)bS;
    }
    __syncthreads();

    // lanes 0..15: luma lines; 16..23: Cb lines; 24..31: Cr lines
    const bool isY = lane < 16, isC = lane >= 16 && lane < 32;
    const int cpl = (lane >> 3) & 1, cl = lane & 7;
#pragma unroll 1
    for (int dir = 0; dir < 2; dir++) {
#pragma unroll 1
        for (int e = 0; e < 4; e++) {
            if (isY) {
                const int bS = s_bs[dir][e * 4 + (lane >> 2)];
                if (bS) {
                    uint8_t* p = dir == 0 ? s_y + (4 + lane) * DB_LP + 4 + 4 * e : s_y + (4 + 4 * e) * DB_LP + 4 + lane;
                    filter_line(p, dir == 0 ? 1 : DB_LP, bS, D.alpha_y, D.beta_y, bS < 4 ? D.tc0_y[bS - 1] : 0, false);
                }
            } else if (isC && !(e & 1)) {
                const int bS = s_bs[dir][e * 4 + (cl >> 1)];
                if (bS) {
                    uint8_t* p = dir == 0 ? s_c[cpl] + (2 + cl) * DB_CP + 2 + 2 * e : s_c[cpl] + (2 + 2 * e) * DB_CP + 2 + cl;
                    filter_line(p, dir == 0 ? 1 : DB_CP, bS, D.alpha_c, D.beta_c, bS < 4 ? D.tc0_c[bS - 1] : 0, true);
                }
            }
            __syncthreads();
        }
    }
    // write back: own block plus the 3 (luma) / 1 (chroma) neighbour columns and rows that may have changed.
    // 20x20 region minus the untouched first row/column of the apron; corner apron samples are never
    // modified by this macroblock and belong to other macroblocks, so they are skipped.
    for (int i = lane; i < 20 * 5; i += 64) {
        const int r = i / 5, c = i - r * 5;
        const int gy = 16 * my - 4 + r, gx = 16 * mx - 4 + c * 4;
        if (gy < 0 || gx < 0) continue;
        if (r < 4 && c == 0) continue;  // top-left corner block: not ours
        *(uint32_t*)(D.pl[0] + (size_t)gy * D.cw + gx) = *(const uint32_t*)(s_y + r * DB_LP + c * 4);
    }
    for (int i = lane; i < 2 * 10 * 5; i += 64) {
        const int pl = i / 50, k = i - pl * 50, r = k / 5, c = k - r * 5;
        const int gy = 8 * my - 2 + r, gx = 8 * mx - 2 + c * 2;
        if (gy < 0 || gx < 0) continue;
        if (r < 2 && c == 0) continue;
        *(uint16_t*)(D.pl[1 + pl] + (size_t)gy * cs + gx) = *(const uint16_t*)(s_c[pl] + r * DB_CP + c * 2);
    }
}

}  // namespace h264
