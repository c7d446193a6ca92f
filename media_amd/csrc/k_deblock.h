// media_amd/csrc/k_deblock.h -- in-loop deblocking filter (H.264 8.7).  Macroblock (mx,my)
// filters against samples already finished by its left, top and top-right neighbours, which
// 8.7 orders strictly (vertical edges of a macroblock, then its horizontal edges, macroblocks
// in raster order).  Two forms, same results:
//   k_deblock_rows  (in use) one launch, one persistent wave per macroblock row (further down)
//   k_deblock_diag  (first form; MI355X_H264_DIAG=1) one wavefront per macroblock, one launch
//                   per wavefront step mx + 2*my == s
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
    SliceRows sl;        // several slices: disable_deblocking_filter_idc 2, the edge between two slices is left alone
    const uint8_t* bs;   // k_bs's boundary strengths, 32 per macroblock: [dir][edge][segment] (the diagonal form reads them here)
    // k_deblock_rows<.., PERMB = true> (decoder peer): thresholds per edge from the two macroblocks' own QPs (8.7.2.2)
    const uint8_t* mbqp; // QP_Y per macroblock (0 for I_PCM)
    int oa, ob;          // FilterOffsetA, FilterOffsetB
    int cqo_cb, cqo_cr;  // chroma_qp_index_offset, second_chroma_qp_index_offset
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

enum { DB_LP = 24, DB_CP = 12 };  // LDS pitches: luma 20 wide, chroma 10 wide (4 / 2 apron)

__global__ __launch_bounds__(64) void k_deblock_diag(DbParams D, int s)
{   // (single-item debug form: no batch dimension)
    const int lane = threadIdx.x;
    // macroblocks with mx + 2*my == s
    const int ymin = max(0, (s - (D.mbw - 1) + 1) >> 1);
    const int my = ymin + blockIdx.x, mx = s - 2 * my;
    if (my >= D.mbh || mx < 0 || mx >= D.mbw) return;
    const int cs = D.cw / 2;
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
        const int bS = D.bs[((size_t)my * D.mbw + mx) * 32 + lane];   // k_bs (k_cavlc.h mb_edge_strength), same (dir, edge, segment) order
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

// ===========================================================================
// Persistent form: ONE launch, one wavefront per macroblock row.  Row r walks its
// macroblocks left to right; macroblock mx needs the bottom four luma / two chroma
// sample rows of macroblock (mx, r-1), which row r-1 PUBLISHES once it has
// finished macroblock mx+1 (whose left-edge filter is the last thing that touches
// them).  That is exactly the 8.7 raster-order dependency (left, top, top-right).
//
// Hand-off (cdna_hip_programming.md Guideline 16, form R2 "the data is the flag"):
// 24 granules per macroblock, each ONE aligned 8-byte agent-scope relaxed atomic
// store of {tag = picture serial, 4 samples}; the consumer re-reads its granules
// with agent-scope atomic loads (L1-bypassing) until every tag matches.  No fence,
// no separate flag, no drain.  Each sample of the picture has exactly one writer:
// a row stores rows 0..11 of its macroblocks (0..5 chroma), the row below stores
// rows 12..15 (6..7) after its own top-edge filter; the last row stores all 16.
// Everything a macroblock needs from HBM (its samples, MbInfo, granules) is
// requested one macroblock ahead, so an iteration is LDS/VALU work only.
// Every spin is bounded; a timeout sets *err and the grid still drains.
// ===========================================================================
typedef unsigned long long u64;
#define AT_LOAD(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define AT_STORE(p, v) __hip_atomic_store((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)

struct DbRowParams {
    DbParams d;
    u64* handoff;        // [mbh][mbw][24] granules: 16 luma dwords (rows 12..15), 8 chroma dwords (rows 6..7 of Cb, Cr)
    const uint32_t* bs;  // [nmb][8]: boundary strengths (k_cavlc count pass), word dir*4+edge, byte = segment along the edge
    unsigned* err;
    unsigned serial;     // changes every picture, never 0
    const unsigned* anybs;   // [item] == serial when the picture has any non-zero boundary strength (k_bs)
    const unsigned* anypcm;  // [item] == pic_serial: the picture holds an I_PCM macroblock and is not filtered (slice header idc 1)
    const unsigned* anyintra;   // [item] == pic_serial: a P picture with intra macroblocks (bS 3 / 4 edges)
    unsigned pic_serial;
    int npic;                // pictures of the step (gridDim.y of them at a time)
    int need_intra;          // P pictures are launched in both forms: < 0 this form runs when the picture has no intra
                             // macroblock, > 0 when it has, 0 = unconditional (IDR)
    // lockstep batch strides (gridDim.y items)
    size_t st_y, st_c, st_handoff;   // bytes, bytes, u64 words
    int st_mb;
    int row0;            // first macroblock row of this instance's band (blockIdx.x counts from it)
    // indirect launches (IND = true, dev_common.h item_ref): position -> item, its ring slot and its QP; d.pl[] = plane bases
    const uint32_t* itemtab;
    size_t st_ring_y, st_ring_c;
};

// One edge, one line of samples held in registers, branch-free so that luma and
// chroma lines share one instruction stream (8.7.2.3 / 8.7.2.4).  chroma lanes
// never touch p1/q1 and use tc = tc0 + 1.  BS4 compiles the intra (bS = 4) form in.
template <bool BS4>
__device__ __forceinline__ void filt_uni(const int p3, int& p2, int& p1, int& p0, int& q0, int& q1, int& q2, const int q3,
                                         int bS, bool chroma, int alpha, int beta, int t1, int t2, int t3)
{
    // Every decision is an all-ones / all-zero MASK applied with AND, not a select: hipcc turns chains of selects over
    // this much arithmetic back into exec-masked regions with branches, and on a wave that runs alone on its SIMD
    // every s_and_saveexec / s_cbranch pair is dead time in the dependency chain.
    const int tc0 = bS == 1 ? t1 : (bS == 2 ? t2 : t3);
    const int fm = -(int)(bS != 0 && iabs(p0 - q0) < alpha && iabs(p1 - p0) < beta && iabs(q1 - q0) < beta);
    const int lum = chroma ? 0 : -1;
    const int apm = lum & -(int)(iabs(p2 - p0) < beta), aqm = lum & -(int)(iabs(q2 - q0) < beta);
    const int tc = tc0 + (chroma ? 1 : 0) - apm - aqm;          // + 1 per true luma condition
    const int d = clip3(-tc, tc, (((q0 - p0) << 2) + (p1 - q1) + 4) >> 3);
    const int avg = (p0 + q0 + 1) >> 1;
    int dp0 = clip255(p0 + d) - p0, dq0 = clip255(q0 - d) - q0;                 // changes of the normal filter
    int dp1 = apm & clip3(-tc0, tc0, (p2 + avg - (p1 << 1)) >> 1);
    int dq1 = aqm & clip3(-tc0, tc0, (q2 + avg - (q1 << 1)) >> 1);
    int dp2 = 0, dq2 = 0;
    if (BS4) {
        const int s4 = -(int)(bS == 4);
        const int strong = -(int)(iabs(p0 - q0) < ((alpha >> 2) + 2));
        const int sp = apm & strong, sq = aqm & strong;
        // strong (8.7.2.4) and weak intra forms as changes; sp / sq pick per side
        const int wp0 = (sp & (((p2 + 2 * p1 + 2 * p0 + 2 * q0 + q1 + 4) >> 3) - p0)) | (~sp & (((2 * p1 + p0 + q1 + 2) >> 2) - p0));
        const int wq0 = (sq & (((p1 + 2 * p0 + 2 * q0 + 2 * q1 + q2 + 4) >> 3) - q0)) | (~sq & (((2 * q1 + q0 + p1 + 2) >> 2) - q0));
        const int wp1 = sp & (((p2 + p1 + p0 + q0 + 2) >> 2) - p1), wq1 = sq & (((p0 + q0 + q1 + q2 + 2) >> 2) - q1);
        const int wp2 = sp & (((2 * p3 + 3 * p2 + p1 + p0 + q0 + 4) >> 3) - p2), wq2 = sq & (((2 * q3 + 3 * q2 + q1 + q0 + p0 + 4) >> 3) - q2);
        dp0 = (s4 & wp0) | (~s4 & dp0); dq0 = (s4 & wq0) | (~s4 & dq0);
        dp1 = (s4 & wp1) | (~s4 & dp1); dq1 = (s4 & wq1) | (~s4 & dq1);
        dp2 = s4 & wp2; dq2 = s4 & wq2;
    }
    p0 += fm & dp0; q0 += fm & dq0; p1 += fm & dp1; q1 += fm & dq1;
    if (BS4) { p2 += fm & dp2; q2 += fm & dq2; }
}

enum { DR_LP = 40, DR_CP = 24 };  // LDS pitches; luma tile cols -16..15 (+4 pad), rows -4..15; chroma cols -8..7, rows -4..15
// One tile holds luma and both chroma planes at offsets chosen for the LDS banks (64 x 4 B; ds_read / write_b32 serve 32 lanes at
// a time by (address / 4) mod 32).  The vertical-edge phase has 16 luma and 16 chroma lanes each read and write dword w of ITS
// line: luma lines are 10 dwords apart and take the 16 banks of one parity; chroma lines 6 dwords apart take 8 banks of the other
// parity when the plane starts at an odd dword (DR_CB), and the second plane the remaining 8 when it lies 16 mod 32 dwords
// after the first (DR_CPS) - 32 lanes, 32 banks.  (Separate arrays had put all 32 lines on the 16 odd banks, the two chroma
// planes partly on each other's: 2- and 3-way conflicts on every access of the phase)
#ifdef MI355X_AB_OLD_DB_TILE   // A/B builds only (media_amd/csrc/Makefile target `ab`): the layout before round 3, chroma straight behind luma
enum { DR_CB = 4 * 200, DR_CPS = 4 * 120, DR_TILE = DR_CB + 2 * DR_CPS + 12 };
#else
enum { DR_CB = 4 * 201, DR_CPS = 4 * 144, DR_TILE = DR_CB + 2 * DR_CPS + 12 };
#endif

// PERMB (the decoder peer, streams of other encoders): qPp / qPq are the two macroblocks' own QPs (mb_qp_delta, I_PCM = 0),
// the chroma QPs go through chroma_qp_index_offset, indexA / indexB through the slice's filter offsets - three threshold
// sets per macroblock (left edge, top edge, inner edges), looked up in LDS copies of Tables 8-15 / 8-16.  PERMB = false is
// the encoder's form: one set per picture, prepared on the host.
template <bool BS4, bool PERMB, bool IND>
__device__ __forceinline__ void deblock_rows_picture(const DbRowParams& R, const int pos)   // pos: the picture's position in the step (blockIdx.y, or the one this workgroup walks to)
{
    const int bitem = batch_item<IND>(R.itemtab, pos);
    if (R.anybs[bitem] != R.serial) return;   // no edge of this picture is filtered: nothing to do, nobody waits
    if (R.anypcm[bitem] == R.pic_serial) return;
    if (R.need_intra != 0 && (R.anyintra[bitem] == R.pic_serial) != (R.need_intra > 0)) return;
    DbParams D = R.d;
    {
        const size_t g = (size_t)bitem;
        D.pl[0] += g * R.st_y; D.pl[1] += g * R.st_c; D.pl[2] += g * R.st_c; D.mb += g * R.st_mb;
        if constexpr (IND) {   // the item's own ring slot and its own QP's thresholds (Tables 8-16 / 8-17, chroma through Table 8-15)
            const ItemRef it = item_ref(R.itemtab, pos);
            D.pl[0] += (size_t)it.cur * R.st_ring_y; D.pl[1] += (size_t)it.cur * R.st_ring_c; D.pl[2] += (size_t)it.cur * R.st_ring_c;
            const int qpc = c_chroma_qp[it.qp];
            D.alpha_y = c_alpha[it.qp]; D.beta_y = c_beta[it.qp]; D.alpha_c = c_alpha[qpc]; D.beta_c = c_beta[qpc];
#pragma unroll
            for (int i = 0; i < 3; i++) { D.tc0_y[i] = c_tc0[it.qp][i]; D.tc0_c[i] = c_tc0[qpc][i]; }
        }
        if (PERMB) D.mbqp += g * R.st_mb;
    }
    u64* const handoff = R.handoff + (size_t)bitem * R.st_handoff;
    const uint32_t* const bsw = R.bs + (size_t)bitem * R.st_mb * 8;
    const int lane = threadIdx.x, my = R.row0 + blockIdx.x, cs = D.cw / 2;   // (row0: first row of this instance's band)
    // a slice's rows form a wavefront of their own: its first row waits for nobody (no edge to the slice above is
    // filtered), its last row stores all sixteen sample rows itself
    const bool first_row = !D.sl.has_top(my);
    const bool last_row = my == D.mbh - 1 || !D.sl.has_top(my + 1);
    __shared__ __attribute__((aligned(16))) uint8_t s_y[DR_TILE];        // luma [row+4][col+16], then the chroma planes [row+4][col+8] at DR_CB + plane * DR_CPS
#define SY(r, c) s_y[((r) + 4) * DR_LP + (c) + 16]
#define SC(pl, r, c) s_y[DR_CB + (pl) * DR_CPS + ((r) + 4) * DR_CP + (c) + 8]
    bool timed_out = false;
    const int yr = lane >> 2, yc4 = (lane & 3) * 4;                                   // luma dword owned by this lane
    const int cpl_l = (lane >> 4) & 1, cr_l = (lane >> 1) & 7, cc4 = (lane & 1) * 4;  // chroma dword (lanes < 32)
    // granule k (lane < 24): 0..15 luma row 12 + k/4, dword k%4; 16..23 chroma plane (k-16)/4, row 6 + ((k-16)/2)%2, dword (k-16)%2
    const int gk = lane;
    // filter lanes: 0..15 luma lines, 16..31 chroma lines (plane, line)
    const bool isC = lane >= 16;
    const int fpl = (lane >> 3) & 1, fln = isC ? (lane & 7) : lane;
    const int seg8 = 8 * (isC ? (fln >> 1) : (fln >> 2));
    // thresholds of the inner edges, of the left edge (L) and of the top edge (T); one set unless PERMB
    int al = isC ? D.alpha_c : D.alpha_y, be = isC ? D.beta_c : D.beta_y;
    int t1 = isC ? D.tc0_c[0] : D.tc0_y[0], t2 = isC ? D.tc0_c[1] : D.tc0_y[1], t3 = isC ? D.tc0_c[2] : D.tc0_y[2];
    int alL = al, beL = be, t1L = t1, t2L = t2, t3L = t3, alT = al, beT = be, t1T = t1, t2T = t2, t3T = t3;
    __shared__ uint32_t s_tA[PERMB ? 52 : 1];   // by indexA: alpha << 24 | tc0(bS 3) << 16 | tc0(bS 2) << 8 | tc0(bS 1)
    __shared__ uint8_t s_tB[PERMB ? 52 : 1], s_cq[PERMB ? 52 : 1];   // beta by indexB; QP_C by the offset luma QP
    if (PERMB) {
        if (lane < 52) {
            s_tA[lane] = ((uint32_t)c_alpha[lane] << 24) | ((uint32_t)c_tc0[lane][2] << 16) | ((uint32_t)c_tc0[lane][1] << 8) | (uint32_t)c_tc0[lane][0];
            s_tB[lane] = c_beta[lane];
            s_cq[lane] = c_chroma_qp[lane];
        }
        wave_sync();
    }
    const int cqo = fpl ? D.cqo_cr : D.cqo_cb;
    auto thresholds = [&](const int qa, const int qb, int& A, int& B, int& T1, int& T2, int& T3) {
        const int ca = (int)s_cq[clip3(0, 51, qa + cqo)], cb = (int)s_cq[clip3(0, 51, qb + cqo)];
        const int av = ((isC ? ca : qa) + (isC ? cb : qb) + 1) >> 1;           // qPav (8-461)
        const uint32_t w = s_tA[clip3(0, 51, av + D.oa)];
        A = (int)(w >> 24); T1 = (int)(w & 255u); T2 = (int)((w >> 8) & 255u); T3 = (int)((w >> 16) & 255u);
        B = (int)s_tB[clip3(0, 51, av + D.ob)];
    };
    uint32_t pf_q = 0;   // QP_Y of the macroblock | of the one above << 8
    int qp_left = 0;
    uint8_t* vrow = isC ? &SC(fpl, fln, -4) : &SY(fln, -4);                 // V phase: this lane's line, 4-byte aligned
    uint8_t* hcol = isC ? &SC(fpl, -4, fln) : &SY(-4, fln);                 // H phase: top of this lane's column
    const int hstride = isC ? DR_CP : DR_LP;

    // Software pipeline: the samples, boundary strengths and hand-off granules of macroblock mx+1 are
    // requested (straight-line, unconditional loads) after macroblock mx has been put into LDS, travel
    // while mx is filtered, and are waited for BEFORE this iteration's stores are issued, so no iteration
    // ever waits for its own stores (vmcnt counts loads and stores together, in order).
    uint32_t pf_y = 0, pf_c = 0;
    uint4 pf_b0 = {0, 0, 0, 0}, pf_b1 = pf_b0;
    u64 pf_g = 0;
    const int grow = first_row ? my : my - 1;      // a first row reads (and ignores) its own slots
    const int glane = lane < 24 ? lane : lane - 24 < 24 ? lane - 24 : lane - 48;
    auto prefetch = [&](int mx) {
        pf_y = *(const uint32_t*)(D.pl[0] + (size_t)(16 * my + yr) * D.cw + 16 * mx + yc4);
        pf_c = *(const uint32_t*)((cpl_l ? D.pl[2] : D.pl[1]) + (size_t)(8 * my + cr_l) * cs + 8 * mx + cc4);   // lanes >= 32 mirror lanes < 32
        const uint4* b = (const uint4*)(bsw + ((size_t)my * D.mbw + mx) * 8);
        pf_b0 = b[0]; pf_b1 = b[1];
        pf_g = AT_LOAD(handoff + ((size_t)grow * D.mbw + mx) * 24 + glane);
        if (PERMB) pf_q = (uint32_t)D.mbqp[(size_t)my * D.mbw + mx] | ((uint32_t)D.mbqp[(size_t)grow * D.mbw + mx] << 8);
    };
    auto consume = [&]() {   // forces the waits for the prefetched registers to sit here
        asm volatile("" : "+v"(pf_y), "+v"(pf_c), "+v"(pf_g));
        asm volatile("" : "+v"(pf_b0.x), "+v"(pf_b0.y), "+v"(pf_b0.z), "+v"(pf_b0.w));
        asm volatile("" : "+v"(pf_b1.x), "+v"(pf_b1.y), "+v"(pf_b1.z), "+v"(pf_b1.w));
        if (PERMB) asm volatile("" : "+v"(pf_q));
    };
    prefetch(0);
    consume();
    uint32_t cur_y = pf_y, cur_c = pf_c, cur_q = pf_q;
    uint4 b0 = pf_b0, b1 = pf_b1;
    u64 g = pf_g;

    // 7. the previous macroblock is final with respect to this row once the current one's left edge has been filtered
    // (the horizontal edges of the current macroblock do not touch it): store / publish it.  Issued right after the
    // vertical-edge phase, so that the stores' acknowledgements (the agent-scope granule stores go all the way to the
    // memory side; vmcnt retires in order) have the rest of the iteration to come back before the next wait for loads.
    auto store_prev = [&](const int mx, const bool have_cur) {
        if (mx > 0) {
            const int pmx = mx - 1;
            const int co = have_cur ? -16 : 0, cco = have_cur ? -8 : 0;  // after the last macroblock there was no shift
            const int nrow = last_row ? 16 : 12, ncrow = last_row ? 8 : 6;
            if (yr < nrow) *(uint32_t*)(D.pl[0] + (size_t)(16 * my + yr) * D.cw + 16 * pmx + yc4) = *(const uint32_t*)&SY(yr, co + yc4);
            if (lane < 32 && cr_l < ncrow)
                *(uint32_t*)((cpl_l ? D.pl[2] : D.pl[1]) + (size_t)(8 * my + cr_l) * cs + 8 * pmx + cc4) = *(const uint32_t*)&SC(cpl_l, cr_l, cco + cc4);
            if (!last_row && lane < 24) {
                uint32_t v;
                if (lane < 16) v = *(const uint32_t*)&SY(12 + (gk >> 2), co + (gk & 3) * 4);
                else v = *(const uint32_t*)&SC((gk - 16) >> 2, 6 + (((gk - 16) >> 1) & 1), cco + ((gk - 16) & 1) * 4);
                AT_STORE(handoff + ((size_t)my * D.mbw + pmx) * 24 + gk, ((u64)R.serial << 32) | v);
            }
        }
    };
    for (int mx = 0; mx <= D.mbw; mx++) {
        const bool have_cur = mx < D.mbw;
        if (have_cur) {
            // 1. previous macroblock moves to the left half of the tile
            if (mx > 0) {
                *(uint32_t*)&SY(yr, yc4 - 16) = *(const uint32_t*)&SY(yr, yc4);
                if (lane < 32) *(uint32_t*)&SC(cpl_l, cr_l, cc4 - 8) = *(const uint32_t*)&SC(cpl_l, cr_l, cc4);
            }
            wave_sync();
            // 2. current macroblock into LDS
            *(uint32_t*)&SY(yr, yc4) = cur_y;
            if (lane < 32) *(uint32_t*)&SC(cpl_l, cr_l, cc4) = cur_c;
            // 3. top apron: wait until every granule carries this picture's tag
            if (!first_row) {
                unsigned spins = 0;
                while (!timed_out) {
                    const bool bad = lane < 24 && (unsigned)(g >> 32) != R.serial;
                    if (__ballot(bad) == 0ull) break;
                    if (++spins > (1u << 20)) { timed_out = true; break; }
                    __builtin_amdgcn_s_sleep(1);
                    g = AT_LOAD(handoff + ((size_t)grow * D.mbw + mx) * 24 + glane);
                }
                if (lane < 16) *(uint32_t*)&SY(-4 + (gk >> 2), (gk & 3) * 4) = (uint32_t)g;
                else if (lane < 24) *(uint32_t*)&SC((gk - 16) >> 2, -2 + (((gk - 16) >> 1) & 1), ((gk - 16) & 1) * 4) = (uint32_t)g;
            }
            // requests for the next macroblock leave now and are collected after the filter
            if (mx + 1 < D.mbw) prefetch(mx + 1);
            if (PERMB) {
                const int qc = (int)(cur_q & 255u), qt = (int)(cur_q >> 8);
                thresholds(qc, qc, al, be, t1, t2, t3);
                thresholds(mx > 0 ? qp_left : qc, qc, alL, beL, t1L, t2L, t3L);
                thresholds(qt, qc, alT, beT, t1T, t2T, t3T);
                qp_left = qc;
            }
            wave_sync();
            const bool any_v = (b0.x | b0.y | b0.z | b0.w) != 0, any_h = (b1.x | b1.y | b1.z | b1.w) != 0;
            // 4. vertical edges: lane = one line of samples, four edges in registers.
            //    chroma lines take part in steps 0 and 1 (chroma edges 0 and 4 <-> luma edges 0 and 8)
            if (any_v && lane < 32) {
                int px[20];
#pragma unroll
                for (int w = 0; w < 5; w++) {
                    const uint32_t v = *(const uint32_t*)(vrow + 4 * w);
#pragma unroll
                    for (int b = 0; b < 4; b++) px[4 * w + b] = (int)((v >> (8 * b)) & 255);
                }
                const int e0 = (int)(((isC ? b0.x : b0.x) >> seg8) & 255), e1 = (int)(((isC ? b0.z : b0.y) >> seg8) & 255);
                const int e2 = isC ? 0 : (int)((b0.z >> seg8) & 255), e3 = isC ? 0 : (int)((b0.w >> seg8) & 255);
                // an edge no line of which is filtered (bS 0 in every segment, luma and chroma: three of four inner edges on
                // ordinary P pictures) is skipped by ONE wave-uniform branch - not a branch inside the filter, see filt_uni
                if (__ballot(e0 != 0) != 0ull) filt_uni<BS4>(px[0], px[1], px[2], px[3], px[4], px[5], px[6], px[7], e0, isC, alL, beL, t1L, t2L, t3L);
                if (__ballot(e1 != 0) != 0ull) filt_uni<BS4>(px[4], px[5], px[6], px[7], px[8], px[9], px[10], px[11], e1, isC, al, be, t1, t2, t3);
                if (__ballot(e2 != 0) != 0ull) filt_uni<BS4>(px[8], px[9], px[10], px[11], px[12], px[13], px[14], px[15], e2, isC, al, be, t1, t2, t3);
                if (__ballot(e3 != 0) != 0ull) filt_uni<BS4>(px[12], px[13], px[14], px[15], px[16], px[17], px[18], px[19], e3, isC, al, be, t1, t2, t3);
#pragma unroll
                for (int w = 0; w < 5; w++)
                    if (w < 3 || !isC)
                        *(uint32_t*)(vrow + 4 * w) = (uint32_t)px[4 * w] | ((uint32_t)px[4 * w + 1] << 8) | ((uint32_t)px[4 * w + 2] << 16) | ((uint32_t)px[4 * w + 3] << 24);
            }
            wave_sync();
            store_prev(mx, true);
            // 5. horizontal edges: lane = one column of samples (rows -4..15; chroma uses rows -4..7 of its tile)
            if (any_h && lane < 32) {
                int px[20];
#pragma unroll
                for (int r = 0; r < 20; r++) px[r] = hcol[r * hstride];
                const int e0 = (int)((b1.x >> seg8) & 255), e1 = (int)(((isC ? b1.z : b1.y) >> seg8) & 255);
                const int e2 = isC ? 0 : (int)((b1.z >> seg8) & 255), e3 = isC ? 0 : (int)((b1.w >> seg8) & 255);
                if (__ballot(e0 != 0) != 0ull) filt_uni<BS4>(px[0], px[1], px[2], px[3], px[4], px[5], px[6], px[7], e0, isC, alT, beT, t1T, t2T, t3T);
                if (__ballot(e1 != 0) != 0ull) filt_uni<BS4>(px[4], px[5], px[6], px[7], px[8], px[9], px[10], px[11], e1, isC, al, be, t1, t2, t3);
                if (__ballot(e2 != 0) != 0ull) filt_uni<BS4>(px[8], px[9], px[10], px[11], px[12], px[13], px[14], px[15], e2, isC, al, be, t1, t2, t3);
                if (__ballot(e3 != 0) != 0ull) filt_uni<BS4>(px[12], px[13], px[14], px[15], px[16], px[17], px[18], px[19], e3, isC, al, be, t1, t2, t3);
                // samples an edge can change: p2..q2 around rows 0,4,8,12 -> rows -3..14; chroma: p0,q0 around rows 0,4
#pragma unroll
                for (int r = 1; r < 19; r++) {
                    const bool both = r == 3 || r == 4 || r == 7 || r == 8;
                    if (both || !isC) hcol[r * hstride] = (uint8_t)px[r];
                }
            }
            wave_sync();
            consume();   // the next macroblock's data has arrived; nothing below waits for memory any more
            // 6. the top apron (rows 12..15 / 6..7 of the macroblock above) is final: store it
            if (!first_row) {
                if (lane < 16) *(uint32_t*)(D.pl[0] + (size_t)(16 * my - 4 + (gk >> 2)) * D.cw + 16 * mx + (gk & 3) * 4) = *(const uint32_t*)&SY(-4 + (gk >> 2), (gk & 3) * 4);
                else if (lane < 24) {
                    const int k = gk - 16, pl = k >> 2, r = (k >> 1) & 1, c4 = (k & 1) * 4;
                    *(uint32_t*)((pl ? D.pl[2] : D.pl[1]) + (size_t)(8 * my - 2 + r) * cs + 8 * mx + c4) = *(const uint32_t*)&SC(pl, -2 + r, c4);
                }
            }
        }
        if (!have_cur) store_prev(mx, false);   // after the last macroblock of the row
        cur_y = pf_y; cur_c = pf_c; b0 = pf_b0; b1 = pf_b1; g = pf_g; cur_q = pf_q;
        wave_sync();
    }
    if (timed_out && lane == 0) *R.err = 1u;  // pinned host word, read after the picture's event
#undef SY
#undef SC
}
// The kernels: workgroup (row, y) filters its row of pictures y, y + gridDim.y, ... of the step's R.npic pictures.  A launch whose
// pictures nearly all return at once - the bS 4 form on P steps of content without intra macroblocks - is made with gridDim.y = 1: placing
// a thousand 120-register waves beside the other instance's kernels for nothing is what costs, not the walk (see k_pintra_rows).
template <bool BS4, bool PERMB = false, bool IND = false>
__global__ __launch_bounds__(64) void k_deblock_rows(DbRowParams R)
{
    // dependency-bound: when a throughput kernel of another stream shares the SIMD, this wave issues first
    __builtin_amdgcn_s_setprio(3);
    for (int pos = blockIdx.y; pos < R.npic; pos += gridDim.y) deblock_rows_picture<BS4, PERMB, IND>(R, pos);
}


// ===========================================================================
// Pair form of the persistent loop filter: ONE wave filters TWO macroblock rows - lanes 0..31 row 2p
// (macroblock t in iteration t), lanes 32..63 row 2p + 1 two macroblocks behind (macroblock t - 2), which is exactly the lag
// 8.7 demands (left, top, top-right).  Both rows run the SAME instruction stream: every phase of the row form used lanes
// 0..31 only for the filter arithmetic, so a lockstep batch needs half the waves and half the filter instructions per
// macroblock, and the upper row hands its bottom sample rows to the lower one through LDS (a two-slot ring: written in
// iteration t - 1, read in iteration t) instead of 24 global granules and their latency.  Between pairs the hand-off is
// the row form's ({tag, 4 samples} granules in global memory), and so is everything else: tile layout, the branch-free
// filter, prefetch one macroblock ahead, one writer per picture sample, bounded spins.
// MEASURED (bench workload, 32 pictures per launch): at first the launch got 7 % shorter and the pipeline 1 % slower (each step of
// the critical path carries the data movement of two macroblocks with 32 lanes per row instead of 64, and the waves the row form
// had too many of were asleep in their hand-off spins, not contending for issue).  With the edge skip below (an edge no line of
// which is filtered costs one scalar branch) the instruction count decides: 152 VALU per macroblock here, 181 in the row form,
// half the waves - 25.0 k fps against 24.0 k on one box.  So this form is the default from a lockstep batch of 8 pictures on
// (pictures of one slice; MI355X_H264_PAIR_FILTER=N moves the threshold, 0 turns it off); smaller batches, the latency mode and
// the decoder use the row form.  tests/test_gpu_parity.py runs both forms at batch 8 and this one forced on single pictures.
// ===========================================================================
template <bool BS4, bool IND>
__device__ __forceinline__ void deblock_pairs_picture(const DbRowParams& R, const int pos)
{
    const int bitem = batch_item<IND>(R.itemtab, pos);
    if (R.anybs[bitem] != R.serial) return;
    if (R.anypcm[bitem] == R.pic_serial) return;
    if (R.need_intra != 0 && (R.anyintra[bitem] == R.pic_serial) != (R.need_intra > 0)) return;
    DbParams D = R.d;
    {
        const size_t g = (size_t)bitem;
        D.pl[0] += g * R.st_y; D.pl[1] += g * R.st_c; D.pl[2] += g * R.st_c; D.mb += g * R.st_mb;
        if constexpr (IND) {   // the item's own ring slot and its own QP's thresholds (Tables 8-16 / 8-17, chroma through Table 8-15)
            const ItemRef it = item_ref(R.itemtab, pos);
            D.pl[0] += (size_t)it.cur * R.st_ring_y; D.pl[1] += (size_t)it.cur * R.st_ring_c; D.pl[2] += (size_t)it.cur * R.st_ring_c;
            const int qpc = c_chroma_qp[it.qp];
            D.alpha_y = c_alpha[it.qp]; D.beta_y = c_beta[it.qp]; D.alpha_c = c_alpha[qpc]; D.beta_c = c_beta[qpc];
#pragma unroll
            for (int i = 0; i < 3; i++) { D.tc0_y[i] = c_tc0[it.qp][i]; D.tc0_c[i] = c_tc0[qpc][i]; }
        }
    }
    u64* const handoff = R.handoff + (size_t)bitem * R.st_handoff;
    const uint32_t* const bsw = R.bs + (size_t)bitem * R.st_mb * 8;
    const int lane = threadIdx.x, half = lane >> 5, hl = lane & 31, cs = D.cw / 2;
    const int rowA = R.row0 + 2 * (int)blockIdx.x, rowB = rowA + 1;
    const bool hasB = rowB < D.mbh;                 // (an odd number of rows: the last wave has an upper row only)
    const int my = half ? (hasB ? rowB : rowA) : rowA;
    const bool live = half == 0 || hasB;            // lanes 32..63 of a wave without a lower row idle along on row A's addresses and store nothing
    const bool first_row = half == 0 && rowA == 0;  // (one slice: only picture row 0 has nothing above)
    const bool last_row = my == D.mbh - 1;
    const bool from_lds = half == 1;                // the lower row's top apron comes from the upper row through LDS
    __shared__ __attribute__((aligned(16))) uint8_t s_yy[2][(DR_TILE + 127) / 128 * 128];   // one tile per row of the pair (a multiple of 32 dwords apart: the same banks)
    __shared__ __attribute__((aligned(16))) uint32_t s_ring[2][24];   // bottom rows of the upper row's macroblock (parity of its index): 16 luma + 8 chroma dwords
    uint8_t* const s_y = s_yy[half];
#define SY(r, c) s_y[((r) + 4) * DR_LP + (c) + 16]
#define SC(pl, r, c) s_y[DR_CB + (pl) * DR_CPS + ((r) + 4) * DR_CP + (c) + 8]
    bool timed_out = false;
    // luma dwords owned by this lane: two of the 64 of a macroblock (rows yr0 and yr0 + 8); chroma dword as in the row form
    const int yr0 = hl >> 2, yc4 = (hl & 3) * 4;
    const int cpl_l = (hl >> 4) & 1, cr_l = (hl >> 1) & 7, cc4 = (hl & 1) * 4;
    const int gk = hl;   // granule index (hl < 24)
    const bool isC = hl >= 16;
    const int fpl = (hl >> 3) & 1, fln = isC ? (hl & 7) : hl;
    const int seg8 = 8 * (isC ? (fln >> 1) : (fln >> 2));
    const int al = isC ? D.alpha_c : D.alpha_y, be = isC ? D.beta_c : D.beta_y;
    const int t1 = isC ? D.tc0_c[0] : D.tc0_y[0], t2 = isC ? D.tc0_c[1] : D.tc0_y[1], t3 = isC ? D.tc0_c[2] : D.tc0_y[2];
    uint8_t* vrow = isC ? &SC(fpl, fln, -4) : &SY(fln, -4);
    uint8_t* hcol = isC ? &SC(fpl, -4, fln) : &SY(-4, fln);
    const int hstride = isC ? DR_CP : DR_LP;

    uint32_t pf_y0 = 0, pf_y1 = 0, pf_c = 0;
    uint4 pf_b0 = {0, 0, 0, 0}, pf_b1 = pf_b0;
    u64 pf_g = 0;
    // only the upper row reads global granules (those of the row above it); a first row and the lanes of the lower row read
    // and ignore row A's own slots - never row -1
    const int grow = (first_row || half == 1) ? rowA : rowA - 1;
    const int glane = hl < 24 ? hl : hl - 24;
    const int last_mx = D.mbw - 1;
    auto prefetch = [&](int mx) {   // mx clamped by the caller into 0 .. mbw - 1
        pf_y0 = *(const uint32_t*)(D.pl[0] + (size_t)(16 * my + yr0) * D.cw + 16 * mx + yc4);
        pf_y1 = *(const uint32_t*)(D.pl[0] + (size_t)(16 * my + yr0 + 8) * D.cw + 16 * mx + yc4);
        pf_c = *(const uint32_t*)((cpl_l ? D.pl[2] : D.pl[1]) + (size_t)(8 * my + cr_l) * cs + 8 * mx + cc4);
        const uint4* b = (const uint4*)(bsw + ((size_t)my * D.mbw + mx) * 8);
        pf_b0 = b[0]; pf_b1 = b[1];
        pf_g = AT_LOAD(handoff + ((size_t)grow * D.mbw + mx) * 24 + glane);
    };
    auto consume = [&]() {
        asm volatile("" : "+v"(pf_y0), "+v"(pf_y1), "+v"(pf_c), "+v"(pf_g));
        asm volatile("" : "+v"(pf_b0.x), "+v"(pf_b0.y), "+v"(pf_b0.z), "+v"(pf_b0.w));
        asm volatile("" : "+v"(pf_b1.x), "+v"(pf_b1.y), "+v"(pf_b1.z), "+v"(pf_b1.w));
    };
    prefetch(0);
    consume();
    uint32_t cur_y0 = pf_y0, cur_y1 = pf_y1, cur_c = pf_c;
    uint4 b0 = pf_b0, b1 = pf_b1;
    u64 g = pf_g;

    // the previous macroblock of this lane's row is final for the row: rows 0..11 (all 16 of a last row) to the picture, rows
    // 12..15 handed down - the upper row into the LDS ring, the lower row as global granules for the next pair
    auto store_prev = [&](const int mx, const bool have_cur) {
        if (mx > 0 && live) {
            const int pmx = mx - 1;
            const int co = have_cur ? -16 : 0, cco = have_cur ? -8 : 0;
            const int nrow = last_row ? 16 : 12, ncrow = last_row ? 8 : 6;
            *(uint32_t*)(D.pl[0] + (size_t)(16 * my + yr0) * D.cw + 16 * pmx + yc4) = *(const uint32_t*)&SY(yr0, co + yc4);
            if (yr0 + 8 < nrow) *(uint32_t*)(D.pl[0] + (size_t)(16 * my + yr0 + 8) * D.cw + 16 * pmx + yc4) = *(const uint32_t*)&SY(yr0 + 8, co + yc4);
            if (cr_l < ncrow)
                *(uint32_t*)((cpl_l ? D.pl[2] : D.pl[1]) + (size_t)(8 * my + cr_l) * cs + 8 * pmx + cc4) = *(const uint32_t*)&SC(cpl_l, cr_l, cco + cc4);
            if (!last_row && hl < 24) {
                uint32_t v;
                if (hl < 16) v = *(const uint32_t*)&SY(12 + (gk >> 2), co + (gk & 3) * 4);
                else v = *(const uint32_t*)&SC((gk - 16) >> 2, 6 + (((gk - 16) >> 1) & 1), cco + ((gk - 16) & 1) * 4);
                if (half == 0) s_ring[pmx & 1][gk] = v;
                else AT_STORE(handoff + ((size_t)my * D.mbw + pmx) * 24 + gk, ((u64)R.serial << 32) | v);
            }
        }
    };
    for (int t = 0; t <= D.mbw + 2; t++) {
        const int mx = t - 2 * half;                              // this lane's macroblock
        const bool in_range = mx >= 0 && mx <= D.mbw && live;     // (mx == mbw: the closing store of the row)
        const bool have_cur = in_range && mx < D.mbw;
        if (have_cur) {
            if (mx > 0) {
                *(uint32_t*)&SY(yr0, yc4 - 16) = *(const uint32_t*)&SY(yr0, yc4);
                *(uint32_t*)&SY(yr0 + 8, yc4 - 16) = *(const uint32_t*)&SY(yr0 + 8, yc4);
                *(uint32_t*)&SC(cpl_l, cr_l, cc4 - 8) = *(const uint32_t*)&SC(cpl_l, cr_l, cc4);
            }
        }
        wave_sync();
        if (have_cur) {
            *(uint32_t*)&SY(yr0, yc4) = cur_y0;
            *(uint32_t*)&SY(yr0 + 8, yc4) = cur_y1;
            *(uint32_t*)&SC(cpl_l, cr_l, cc4) = cur_c;
        }
        // top apron: the upper row waits for the previous pair's granules, the lower row takes the ring slot of its macroblock
        {
            const bool need_g = have_cur && !from_lds && !first_row;   // lanes of the upper half
            unsigned spins = 0;
            while (!timed_out) {
                const bool bad = need_g && hl < 24 && (unsigned)(g >> 32) != R.serial;
                if (__ballot(bad) == 0ull) break;
                if (++spins > (1u << 20)) { timed_out = true; break; }
                __builtin_amdgcn_s_sleep(1);
                if (need_g) g = AT_LOAD(handoff + ((size_t)grow * D.mbw + mx) * 24 + glane);
            }
            if (have_cur && !first_row && hl < 24) {
                const uint32_t v = from_lds ? s_ring[mx & 1][gk] : (uint32_t)g;
                if (hl < 16) *(uint32_t*)&SY(-4 + (gk >> 2), (gk & 3) * 4) = v;
                else *(uint32_t*)&SC((gk - 16) >> 2, -2 + (((gk - 16) >> 1) & 1), ((gk - 16) & 1) * 4) = v;
            }
        }
        {   // requests for this lane's next macroblock (clamped: the loads of a finished or not yet started row are harmless)
            const int nx = mx + 1 < 0 ? 0 : (mx + 1 > last_mx ? last_mx : mx + 1);
            prefetch(nx);
        }
        wave_sync();
        const bool any_v = have_cur && (b0.x | b0.y | b0.z | b0.w) != 0, any_h = have_cur && (b1.x | b1.y | b1.z | b1.w) != 0;
        if (any_v) {
            int px[20];
#pragma unroll
            for (int w = 0; w < 5; w++) {
                const uint32_t v = *(const uint32_t*)(vrow + 4 * w);
#pragma unroll
                for (int b = 0; b < 4; b++) px[4 * w + b] = (int)((v >> (8 * b)) & 255);
            }
            const int e0 = (int)((b0.x >> seg8) & 255), e1 = (int)(((isC ? b0.z : b0.y) >> seg8) & 255);
            const int e2 = isC ? 0 : (int)((b0.z >> seg8) & 255), e3 = isC ? 0 : (int)((b0.w >> seg8) & 255);
            if (__ballot(e0 != 0) != 0ull) filt_uni<BS4>(px[0], px[1], px[2], px[3], px[4], px[5], px[6], px[7], e0, isC, al, be, t1, t2, t3);
            if (__ballot(e1 != 0) != 0ull) filt_uni<BS4>(px[4], px[5], px[6], px[7], px[8], px[9], px[10], px[11], e1, isC, al, be, t1, t2, t3);
            if (__ballot(e2 != 0) != 0ull) filt_uni<BS4>(px[8], px[9], px[10], px[11], px[12], px[13], px[14], px[15], e2, isC, al, be, t1, t2, t3);
            if (__ballot(e3 != 0) != 0ull) filt_uni<BS4>(px[12], px[13], px[14], px[15], px[16], px[17], px[18], px[19], e3, isC, al, be, t1, t2, t3);
#pragma unroll
            for (int w = 0; w < 5; w++)
                if (w < 3 || !isC)
                    *(uint32_t*)(vrow + 4 * w) = (uint32_t)px[4 * w] | ((uint32_t)px[4 * w + 1] << 8) | ((uint32_t)px[4 * w + 2] << 16) | ((uint32_t)px[4 * w + 3] << 24);
        }
        wave_sync();
        if (have_cur) store_prev(mx, true);
        if (any_h) {
            int px[20];
#pragma unroll
            for (int r = 0; r < 20; r++) px[r] = hcol[r * hstride];
            const int e0 = (int)((b1.x >> seg8) & 255), e1 = (int)(((isC ? b1.z : b1.y) >> seg8) & 255);
            const int e2 = isC ? 0 : (int)((b1.z >> seg8) & 255), e3 = isC ? 0 : (int)((b1.w >> seg8) & 255);
            if (__ballot(e0 != 0) != 0ull) filt_uni<BS4>(px[0], px[1], px[2], px[3], px[4], px[5], px[6], px[7], e0, isC, al, be, t1, t2, t3);
            if (__ballot(e1 != 0) != 0ull) filt_uni<BS4>(px[4], px[5], px[6], px[7], px[8], px[9], px[10], px[11], e1, isC, al, be, t1, t2, t3);
            if (__ballot(e2 != 0) != 0ull) filt_uni<BS4>(px[8], px[9], px[10], px[11], px[12], px[13], px[14], px[15], e2, isC, al, be, t1, t2, t3);
            if (__ballot(e3 != 0) != 0ull) filt_uni<BS4>(px[12], px[13], px[14], px[15], px[16], px[17], px[18], px[19], e3, isC, al, be, t1, t2, t3);
#pragma unroll
            for (int r = 1; r < 19; r++) {
                const bool both = r == 3 || r == 4 || r == 7 || r == 8;
                if (both || !isC) hcol[r * hstride] = (uint8_t)px[r];
            }
        }
        wave_sync();
        consume();
        // the top apron (rows 12..15 / 6..7 of the macroblock above) is final: store it
        if (have_cur && !first_row) {
            if (hl < 16) *(uint32_t*)(D.pl[0] + (size_t)(16 * my - 4 + (gk >> 2)) * D.cw + 16 * mx + (gk & 3) * 4) = *(const uint32_t*)&SY(-4 + (gk >> 2), (gk & 3) * 4);
            else if (hl < 24) {
                const int k = gk - 16, pl = k >> 2, r = (k >> 1) & 1, c4 = (k & 1) * 4;
                *(uint32_t*)((pl ? D.pl[2] : D.pl[1]) + (size_t)(8 * my - 2 + r) * cs + 8 * mx + c4) = *(const uint32_t*)&SC(pl, -2 + r, c4);
            }
        }
        if (in_range && !have_cur) store_prev(mx, false);   // after the last macroblock of this lane's row
        if (mx + 1 >= 0 && mx + 1 <= last_mx) { cur_y0 = pf_y0; cur_y1 = pf_y1; cur_c = pf_c; b0 = pf_b0; b1 = pf_b1; g = pf_g; }
        wave_sync();
    }
    if (timed_out && lane == 0) *R.err = 1u;
#undef SY
#undef SC
}
template <bool BS4, bool IND = false>
__global__ __launch_bounds__(64) void k_deblock_pairs(DbRowParams R)
{
    __builtin_amdgcn_s_setprio(3);
    for (int pos = blockIdx.y; pos < R.npic; pos += gridDim.y) deblock_pairs_picture<BS4, IND>(R, pos);
}

}  // namespace h264
