// media_amd/csrc/k_intra.h -- IDR pictures: Intra16x16 + chroma intra.  Prediction needs the
// left / top / top-left reconstruction (8.3.3, 8.3.4), so macroblocks run as a wavefront:
//   k_intra_rows  (the form in use) one persistent wave per macroblock row, row r one
//                 macroblock behind row r-1, bottom sample rows handed down as {tag, data}
//                 granules (same hand-off as the loop filter, k_deblock.h)
//   k_intra_diag  (first, simpler form; MI355X_H264_DIAG=1) one launch per anti-diagonal
//                 mx + my == s, kernel boundaries carry the dependency
//
// SURVEY.md 8a row a6 / a6.2 (inside ISVCEncoder::EncodeFrame,
// /root/reference/video_codec/VideoEncoderOpenH264.cpp:344).
#pragma once
#include "dev_common.h"
#include "mc_filters.h"
#include "k_intra4.h"

namespace h264 {

// "shift, clamp to 0..255, pack bytes" written plainly makes hipcc (ROCm 7.2, gfx950) select v_ashr_pk_u8_i32 and OR
// further bytes into its result as if its upper half were zero (it is not: see k_pmb2.h); the clamped plane samples
// therefore pass through this no-op before they are packed.
__device__ __forceinline__ int plane_px(int v)
{
    int c = clip255(v >> 5);
    asm volatile("" : "+v"(c));
    return c;
}

// Intra16x16 predictor sample (8.3.3); top[0..16] holds p[-1..15,-1], left[y] = p[-1,y]
struct I16Params { int dc, a, b, c; };
__device__ __forceinline__ I16Params i16_params(const uint8_t* top_, const uint8_t* left_, int avail)
{
    // the 17 + 16 neighbour samples as nine dword reads (both arrays are 4-byte aligned in IntraLds)
    int top[20], left[16];
#pragma unroll
    for (int w = 0; w < 5; w++) {
        const uint32_t v = ((const uint32_t*)top_)[w];
#pragma unroll
        for (int k = 0; k < 4; k++) top[4 * w + k] = (int)((v >> (8 * k)) & 255);
    }
#pragma unroll
    for (int w = 0; w < 4; w++) {
        const uint32_t v = ((const uint32_t*)left_)[w];
#pragma unroll
        for (int k = 0; k < 4; k++) left[4 * w + k] = (int)((v >> (8 * k)) & 255);
    }
    I16Params q;
    int st = 0, sl = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) { st += top[1 + i]; sl += left[i]; }
    const bool t = avail & 2, l = avail & 1;
    q.dc = (t && l) ? (st + sl + 16) >> 5 : t ? (st + 8) >> 4 : l ? (sl + 8) >> 4 : 128;
    int H = 0, V = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        H += (i + 1) * (top[1 + 8 + i] - top[1 + 6 - i]);
        V += (i + 1) * (left[8 + i] - (i == 7 ? top[0] : left[6 - i]));
    }
    q.a = 16 * (left[15] + top[16]);
    q.b = (5 * H + 32) >> 6;
    q.c = (5 * V + 32) >> 6;
    return q;
}
__device__ __forceinline__ int i16_px(int mode, int x, int y, const uint8_t* top, const uint8_t* left, const I16Params& q)
{
    return mode == 0 ? top[1 + x] : mode == 1 ? left[y] : mode == 2 ? q.dc : clip255((q.a + q.b * (x - 7) + q.c * (y - 7) + 16) >> 5);
}

// chroma 8x8 predictor (8.3.4): 0 DC, 1 horizontal, 2 vertical, 3 plane
struct C8Params { int dc[4], a, b, c; };
__device__ __forceinline__ C8Params c8_params(const uint8_t* top_, const uint8_t* left_, int avail)
{
    int top[12], left[8];      // 9 + 8 neighbour samples as five dword reads (both arrays 4-byte aligned in IntraLds)
#pragma unroll
    for (int w = 0; w < 3; w++) {
        const uint32_t v = ((const uint32_t*)top_)[w];
#pragma unroll
        for (int k = 0; k < 4; k++) top[4 * w + k] = (int)((v >> (8 * k)) & 255);
    }
#pragma unroll
    for (int w = 0; w < 2; w++) {
        const uint32_t v = ((const uint32_t*)left_)[w];
#pragma unroll
        for (int k = 0; k < 4; k++) left[4 * w + k] = (int)((v >> (8 * k)) & 255);
    }
    C8Params q;
    const bool t = avail & 2, l = avail & 1;
    int st[2] = {0, 0}, sl[2] = {0, 0};
#pragma unroll
    for (int i = 0; i < 4; i++) { st[0] += top[1 + i]; st[1] += top[5 + i]; sl[0] += left[i]; sl[1] += left[4 + i]; }
#pragma unroll
    for (int b = 0; b < 4; b++) {
        const int bx = b & 1, byy = b >> 1;
        int dc;
        if (bx == byy) dc = (t && l) ? (st[bx] + sl[byy] + 4) >> 3 : t ? (st[bx] + 2) >> 2 : l ? (sl[byy] + 2) >> 2 : 128;
        else if (bx == 1) dc = t ? (st[1] + 2) >> 2 : l ? (sl[0] + 2) >> 2 : 128;
        else dc = l ? (sl[1] + 2) >> 2 : t ? (st[0] + 2) >> 2 : 128;
        q.dc[b] = dc;
    }
    int H = 0, V = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        H += (i + 1) * (top[1 + 4 + i] - top[1 + 2 - i]);
        V += (i + 1) * (left[4 + i] - (i == 3 ? top[0] : left[2 - i]));
    }
    q.a = 16 * (left[7] + top[8]);
    q.b = (34 * H + 32) >> 6;
    q.c = (34 * V + 32) >> 6;
    return q;
}
// element k of four values held in registers (an index into the array would move it to scratch memory)
__device__ __forceinline__ int pick4(const int v[4], int k) { return k == 0 ? v[0] : (k == 1 ? v[1] : (k == 2 ? v[2] : v[3])); }
__device__ __forceinline__ int c8_px(int mode, int x, int y, const uint8_t* top, const uint8_t* left, const C8Params& q)
{
    return mode == 0 ? q.dc[(y >> 2) * 2 + (x >> 2)] : mode == 1 ? left[y] : mode == 2 ? top[1 + x]
                                                                                       : clip255((q.a + q.b * (x - 3) + q.c * (y - 3) + 16) >> 5);
}

// LDS working set of one intra macroblock
struct IntraLds {
    __attribute__((aligned(16))) uint8_t src[256];
    __attribute__((aligned(16))) uint8_t srcc[128];
    __attribute__((aligned(16))) uint8_t py[256];
    __attribute__((aligned(16))) uint8_t pc[128];
    __attribute__((aligned(16))) uint8_t rec_y[256];   // reconstruction of this macroblock (pitch 16)
    __attribute__((aligned(16))) uint8_t rec_c[128];   // Cb 8x8, Cr 8x8 (pitch 8)
    __attribute__((aligned(16))) int16_t lv[LV_STRIDE];
    int dc[16];
    uint8_t top[24], left[16];        // luma neighbours; top[0] = top-left, top[1..16] above, top[17..20] above-right (decoder only)
    uint8_t ctop[2][12], cleft[2][8];
    I4Lds i4;                         // Intra4x4 macroblocks (k_intra4.h)
    int xchg[2][2][4];                // k_intra_rows' two waves, [macroblock & 1][luma, chroma]: {bit bound of its blocks, cbp, mode}
};

// the two waves of k_intra_rows meet: LDS traffic drained, not the global loads and stores in flight (__syncthreads() would wait
// for the reconstruction stores just issued and the source requested ahead)
__device__ __forceinline__ void pair_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// Mode decision, transform, quantisation and reconstruction of one intra macroblock (Intra16x16, or Intra4x4 with the
// modes k_i4_decide chose) whose source and
// neighbour samples are already in S (and visible to the whole wave).  Writes recon (global + S.rec_*),
// levels, MbInfo, mvd.
// DEC (the decoder peer, k_dec.h): types, modes and levels are given (MbInfo, aux, levels); prediction, scaling, inverse
// transforms and reconstruction are the code the encoder runs, nothing is decided, transformed forward or written back.
// PART: 0 the whole macroblock in this wave; 1 its luma, 2 its chroma - the two waves of one workgroup of k_intra_rows, which
// share S, work on the same macroblock side by side and meet once (pair_barrier) to add up the bit bound and to exchange what the
// macroblock's header needs.  Luma and chroma predict from their own planes only, so nothing else passes between them.
template <bool DEC = false, int PART = 0>
__device__ __forceinline__ void intra_mb_core(const FrameParams& P, int mx, int my, IntraLds& S, int lane, bool use_i4, uint32_t auxw)
{
    static_assert(!DEC || PART == 0, "the decoder runs the macroblock in one wave");
    constexpr bool LUMA = PART != 2, CHROMA = PART != 1;
    const int mbi = my * P.mbw + mx, bx = 16 * mx, by = 16 * my, cs = P.cw / 2;
    // neighbouring macroblocks available for prediction (6.4.9: in the picture, in this slice, decoded before).  Encoder: slices are
    // bands of whole rows.  DEC: the parser's bits per macroblock (16 left, 32 above, 64 above-right, 128 above-left) - slices of any shape, constrained_intra_pred_flag
    bool top = P.sl.has_top(my), left = mx > 0, topleft = mx > 0 && top, topright = top && mx + 1 < P.mbw;
    if (DEC) {
        const int a = __builtin_amdgcn_readfirstlane((int)P.mbavail[mbi]);
        left = (a & 16) != 0; top = (a & 32) != 0; topright = (a & 64) != 0; topleft = (a & 128) != 0;   // (bits 4..7: usable for intra prediction)
    }
    const int avail = (left ? 1 : 0) | (top ? 2 : 0) | (topleft ? 4 : 0);
    MbInfo* const mbp = P.mb + mbi;
    // DEC: the macroblock's own QP (mb_qp_delta) - of Quant only qp and dq[] are read on the decoding paths; chroma per component
    Quant dqy = {};
    int dqc[2][3] = {{0, 0, 0}, {0, 0, 0}};
    if (DEC) {
        dqy.qp = __builtin_amdgcn_readfirstlane((int)P.mbqp[mbi]);
        dec_dq(dqy.qp, dqy.dq);
        dec_dq(dec_qpc(P, dqy.qp, 0), dqc[0]);
        dec_dq(dec_qpc(P, dqy.qp, 1), dqc[1]);
    }
    const Quant& QY = DEC ? dqy : P.qy;
    if (DEC) {
        if (lane < LV_STRIDE * 2 / 16) ((uint4*)S.lv)[lane] = ((const uint4*)(P.levels + (size_t)mbi * LV_STRIDE))[lane];
        wave_sync();
        if (mbp->type == MB_IPCM) {   // the 384 samples travel as bytes at the start of the macroblock's level area
            const uint8_t* raw = (const uint8_t*)S.lv;
            {
                const int row = lane >> 2, xs = (lane & 3) * 4;
                const uint32_t v = *(const uint32_t*)(raw + row * 16 + xs);
                *(uint32_t*)(P.rec[0] + (size_t)(by + row) * P.cw + bx + xs) = v;
                *(uint32_t*)(S.rec_y + row * 16 + xs) = v;
            }
            if (lane < 32) {
                const int pl = lane >> 4, row = (lane >> 1) & 7, xs = (lane & 1) * 4;
                const uint32_t v = *(const uint32_t*)(raw + 256 + pl * 64 + row * 8 + xs);
                *(uint32_t*)(rec_chroma(P, pl) + (size_t)(8 * my + row) * cs + 8 * mx + xs) = v;
                *(uint32_t*)(S.rec_c + pl * 64 + row * 8 + xs) = v;
            }
            wave_sync();
            return;
        }
    } else {   // (luma DC + luma: 34 x 16 bytes, then the chroma lists)
        const int zi = (LUMA ? 0 : 2 * LV_CHROMA_DC / 16) + lane;
        if (zi < (CHROMA ? LV_STRIDE * 2 / 16 : 2 * LV_CHROMA_DC / 16)) ((uint4*)S.lv)[zi] = make_uint4(0u, 0u, 0u, 0u);
    }

    // ---- Intra4x4 (type and modes chosen by k_i4_decide; use_i4 is wave-uniform, auxw = lanes 0..3: the sixteen modes) ----
    int cbp_luma_i4 = 0, tc_i4 = 0;   // tc_i4: TotalCoeff of luma block blkIdx = lane
    if (LUMA && use_i4) {
        cbp_luma_i4 = i4_code_luma<DEC>(QY, S.i4, S.top, S.left, S.src, auxw, S.lv, tc_i4, DEC ? (int)left : mx, top, topright, lane);
        const uint32_t o = *(const uint32_t*)(S.i4.rb + (1 + (lane >> 2)) * 32 + 4 + (lane & 3) * 4);
        *(uint32_t*)(S.rec_y + (lane >> 2) * 16 + (lane & 3) * 4) = o;
        *(uint32_t*)(P.rec[0] + (size_t)(by + (lane >> 2)) * P.cw + bx + (lane & 3) * 4) = o;
    }

    // ---- luma mode decision: lane = (mode, 4x4 block), SATD per mode ----
    I16Params ip = {0, 0, 0, 0};
    if (LUMA && !use_i4) ip = i16_params(S.top, S.left, avail);
    int best_mode = DEC ? (int)mbp->i16_mode : 0;
    if (LUMA && !DEC && !use_i4) {
        const int mode = lane >> 4, blk = lane & 15, x0 = (blk & 3) * 4, y0 = (blk >> 2) * 4;
        // this lane's block: four source dwords, the four top and left neighbours, the plane value of its corner -
        // fetched once instead of per sample
        int tp[4], lf[4];
#pragma unroll
        for (int k = 0; k < 4; k++) { tp[k] = S.top[1 + x0 + k]; lf[k] = S.left[y0 + k]; }
        const int pl0 = ip.a + ip.b * (x0 - 7) + ip.c * (y0 - 7) + 16;
        int d[16];
#pragma unroll
        for (int y = 0; y < 4; y++) {
            const uint32_t sw = *(const uint32_t*)(S.src + (y0 + y) * 16 + x0);
#pragma unroll
            for (int x = 0; x < 4; x++) {
                const int pr = mode == 0 ? tp[x] : mode == 1 ? lf[y] : mode == 2 ? ip.dc : clip255((pl0 + ip.b * x + ip.c * y) >> 5);
                d[4 * y + x] = (int)((sw >> (8 * x)) & 255) - pr;
            }
        }
        int sum = row_sum16_dpp(hadamard_abs(d)) >> 1;
        const bool ok = mode == 0 ? (avail & 2) : mode == 1 ? (avail & 1) : mode == 2 ? true : avail == 7;
        unsigned key = ok ? (((unsigned)sum << 2) | (unsigned)mode) : 0xFFFFFFFFu;
        best_mode = (int)(wave_min_u32_dpp(key) & 3);
    }
    if (LUMA && !use_i4) {
        const int y = lane >> 2, xs = (lane & 3) * 4;
        uint32_t o = 0;
        if (best_mode == 0) {          // vertical: the four samples above (best_mode is wave-uniform)
#pragma unroll
            for (int k = 0; k < 4; k++) o |= (uint32_t)S.top[1 + xs + k] << (8 * k);
        } else if (best_mode == 1) o = 0x01010101u * (uint32_t)S.left[y];
        else if (best_mode == 2) o = 0x01010101u * (uint32_t)ip.dc;
        else {
            const int pl0 = ip.a + ip.b * (xs - 7) + ip.c * (y - 7) + 16;
#pragma unroll
            for (int k = 0; k < 4; k++) o |= (uint32_t)plane_px(pl0 + ip.b * k) << (8 * k);
        }
        *(uint32_t*)(S.py + y * 16 + xs) = o;
    }
    // ---- chroma mode decision: lane<32 = (mode, plane, block) ----
    int best_cmode = DEC ? (int)mbp->chroma_mode : 0;
    C8Params cp[2] = {};
    if (CHROMA) { cp[0] = c8_params(S.ctop[0], S.cleft[0], avail); cp[1] = c8_params(S.ctop[1], S.cleft[1], avail); }
    if (CHROMA && !DEC) {
        const int mode = (lane >> 3) & 3, pl = (lane >> 2) & 1, blk = lane & 3, x0 = (blk & 1) * 4, y0 = (blk >> 1) * 4;
        int tp[4], lf[4];
#pragma unroll
        for (int k = 0; k < 4; k++) { tp[k] = S.ctop[pl][1 + x0 + k]; lf[k] = S.cleft[pl][y0 + k]; }
        const int ca = pl ? cp[1].a : cp[0].a, cbb = pl ? cp[1].b : cp[0].b, cc = pl ? cp[1].c : cp[0].c;
        const int cdc = pl ? pick4(cp[1].dc, blk) : pick4(cp[0].dc, blk);       // blk = (y0 >> 2) * 2 + (x0 >> 2)
        const int pl0 = ca + cbb * (x0 - 3) + cc * (y0 - 3) + 16;
        int d[16];
#pragma unroll
        for (int y = 0; y < 4; y++) {
            const uint32_t sw = *(const uint32_t*)(S.srcc + pl * 64 + (y0 + y) * 8 + x0);
#pragma unroll
            for (int x = 0; x < 4; x++) {
                const int pr = mode == 0 ? cdc : mode == 1 ? lf[y] : mode == 2 ? tp[x] : clip255((pl0 + cbb * x + cc * y) >> 5);
                d[4 * y + x] = (int)((sw >> (8 * x)) & 255) - pr;
            }
        }
        // per-plane SATD is (sum over 4 blocks) >> 1; cost = Cb + Cr
        int sp = hadamard_abs(d);
        sp += __builtin_amdgcn_mov_dpp(sp, 0xB1, 0xf, 0xf, false);     // the four blocks of the plane (a quad)
        sp += __builtin_amdgcn_mov_dpp(sp, 0x4E, 0xf, 0xf, false);
        sp >>= 1;
        int sum = sp + __builtin_amdgcn_mov_dpp(sp, 0x141, 0xf, 0xf, false);   // + the other plane: lanes l ^ 4 = half-row mirror partner's quad (same sum in all four)
        const bool ok = mode == 0 ? true : mode == 1 ? (avail & 1) : mode == 2 ? (avail & 2) : avail == 7;
        unsigned key = (ok && lane < 32) ? (((unsigned)sum << 2) | (unsigned)mode) : 0xFFFFFFFFu;
        best_cmode = (int)(wave_min_u32_dpp(key) & 3);
    }
    if (CHROMA && lane < 32) {
        const int pl = lane >> 4, y = (lane >> 1) & 7, xs = (lane & 1) * 4;
        uint32_t o = 0;
        if (best_cmode == 0) o = 0x01010101u * (uint32_t)(pl ? pick4(cp[1].dc, (y >> 2) * 2 + (xs >> 2)) : pick4(cp[0].dc, (y >> 2) * 2 + (xs >> 2)));
        else if (best_cmode == 1) o = 0x01010101u * (uint32_t)S.cleft[pl][y];
        else if (best_cmode == 2) {
#pragma unroll
            for (int k = 0; k < 4; k++) o |= (uint32_t)S.ctop[pl][1 + xs + k] << (8 * k);
        } else {
            const int ca = pl ? cp[1].a : cp[0].a, cbb = pl ? cp[1].b : cp[0].b, cc = pl ? cp[1].c : cp[0].c;
            const int pl0 = ca + cbb * (xs - 3) + cc * (y - 3) + 16;
#pragma unroll
            for (int k = 0; k < 4; k++) o |= (uint32_t)plane_px(pl0 + cbb * k) << (8 * k);
        }
        *(uint32_t*)(S.pc + pl * 64 + y * 8 + xs) = o;
    }
    wave_sync();

    // ---- transform / quant: lanes 0..15 luma AC (+DC via Hadamard), 16..23 chroma ----
    int nnz = 0, dcw = 0;
    int d[16];
    const bool is_luma = LUMA && lane < 16 && !use_i4, is_chroma = CHROMA && lane >= 16 && lane < 24;
    const int cpl = (lane - 16) >> 2, cb = lane & 3;
    if (is_luma) {
        const int x = blk_x(lane) * 4, y = blk_y(lane) * 4;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const uint32_t sw = *(const uint32_t*)(S.src + (y + r) * 16 + x), pw = *(const uint32_t*)(S.py + (y + r) * 16 + x);
#pragma unroll
            for (int c = 0; c < 4; c++) d[4 * r + c] = (int)((sw >> (8 * c)) & 255) - (int)((pw >> (8 * c)) & 255);
        }
        if (DEC) {   // scaled coefficients from the given levels (the DC position comes through the Hadamard path below)
#pragma unroll
            for (int i = 1; i < 16; i++) d[i] = (int)S.lv[LV_LUMA + lane * 16 + c_zigzag_inv[i]] * QY.dq[pos_class(i)];
            d[0] = 0;
        } else {
            nnz = tq4x4(d, P.qy, P.qy.f_intra, 1, S.lv + LV_LUMA + lane * 16, &dcw, 0, false);
            S.dc[blk_y(lane) * 4 + blk_x(lane)] = dcw;
        }
    } else if (is_chroma) {
        const int x = (cb & 1) * 4, y = (cb >> 1) * 4;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const uint32_t sw = *(const uint32_t*)(S.srcc + cpl * 64 + (y + r) * 8 + x), pw = *(const uint32_t*)(S.pc + cpl * 64 + (y + r) * 8 + x);
#pragma unroll
            for (int c = 0; c < 4; c++) d[4 * r + c] = (int)((sw >> (8 * c)) & 255) - (int)((pw >> (8 * c)) & 255);
        }
        if (DEC) {
            const int ca = cpl ? dqc[1][0] : dqc[0][0], cbq = cpl ? dqc[1][1] : dqc[0][1], cc = cpl ? dqc[1][2] : dqc[0][2];
#pragma unroll
            for (int i = 1; i < 16; i++) {
                const int k = pos_class(i);
                d[i] = (int)S.lv[LV_CHROMA_AC + (cpl * 4 + cb) * 16 + c_zigzag_inv[i]] * (k == 0 ? ca : (k == 1 ? cbq : cc));
            }
            d[0] = 0;
        } else
            nnz = tq4x4(d, P.qc, P.qc.f_intra, 1, S.lv + LV_CHROMA_AC + (cpl * 4 + cb) * 16, &dcw, 0, false);
    }
    wave_sync();
    if (LUMA && !use_i4) {
        // luma DC (8.5.10): 4x4 Hadamard of the sixteen transformed DCs, quantised at qbits + 2, inverse Hadamard, scaling.
        // Lane = raster position of the block; each transform is four exchange stages (lane ^ 1, ^ 2: DPP quad permutes,
        // ^ 4, ^ 8: DPP row moves).  After the forward pass lane (i, j) holds coefficient (sg(i), sg(j)), sg = [0 3 1 2] (the
        // exchange network yields natural Hadamard order, the standard's matrix is that with its rows permuted); fed back in that
        // arrangement the same network returns the inverse in raster order, the matrix being symmetric.
        const int i = lane & 3, j = (lane >> 2) & 3;
        const int s1 = (i & 1) ? -1 : 1, s2 = (i & 2) ? -1 : 1, s4 = (j & 1) ? -1 : 1, s8 = (j & 2) ? -1 : 1;
        auto wht = [&](int v) {
            v = __builtin_amdgcn_mov_dpp(v, 0xB1, 0xf, 0xf, false) + s1 * v;
            v = __builtin_amdgcn_mov_dpp(v, 0x4E, 0xf, 0xf, false) + s2 * v;
            int q = __builtin_amdgcn_update_dpp(0, v, 0x104, 0xf, 0x5, false);   // row_shl:4 into quads 0, 2: from lane + 4
            q = __builtin_amdgcn_update_dpp(q, v, 0x114, 0xf, 0xa, false);       // row_shr:4 into quads 1, 3: from lane - 4
            v = q + s4 * v;
            return __builtin_amdgcn_mov_dpp(v, 0x128, 0xf, 0xf, false) + s8 * v; // row_ror:8 = lane ^ 8
        };
        const int cu = (0x2130 >> (4 * i)) & 3, cv = (0x2130 >> (4 * j)) & 3;   // sg: 0 3 1 2
        int ldc;
        if (DEC) ldc = lane < 16 ? (int)S.lv[LV_LUMA_DC + ((c_zz_row[cv] >> (4 * cu)) & 15)] : 0;
        else {
            const int hw = wht(lane < 16 ? S.dc[lane] : 0);
            const int qb = P.qy.qbits;
            const unsigned a = (unsigned)iabs(hw);
            const int lq = (int)((a * (unsigned)P.qy.mf[0] + 4u * (unsigned)P.qy.f_intra) >> (qb + 2));
            ldc = hw < 0 ? -lq : lq;
            if (lane < 16) S.lv[LV_LUMA_DC + ((c_zz_row[cv] >> (4 * cu)) & 15)] = (int16_t)ldc;
        }
        const int fi_r = wht(ldc);
        wave_sync();   // every lane has taken its S.dc value
        if (lane < 16) S.dc[lane] = fi_r;
        wave_sync();
        if (is_luma) {
            const int fi = S.dc[blk_y(lane) * 4 + blk_x(lane)];
            const int qp = QY.qp, ls = 16 * (QY.dq[0] >> (qp / 6));
            d[0] = qp >= 36 ? (fi * ls) << (qp / 6 - 6) : (fi * ls + (1 << (5 - qp / 6))) >> (6 - qp / 6);
        }
    }
    if (CHROMA) {   // chroma DC
        const int base = 16 + ((lane - 16) & 4);
        const int w4[4] = {__shfl(dcw, base), __shfl(dcw, base + 1), __shfl(dcw, base + 2), __shfl(dcw, base + 3)};
        int lv[4], deq[4];
        if (DEC) {   // inverse 2x2 Hadamard + scaling (8.5.11) of the given chroma DC levels
            const int pl = is_chroma ? cpl : 0;
#pragma unroll
            for (int i = 0; i < 4; i++) lv[i] = (int)S.lv[LV_CHROMA_DC + pl * 4 + i];
            const int fi[4] = {lv[0] + lv[1] + lv[2] + lv[3], lv[0] - lv[1] + lv[2] - lv[3], lv[0] + lv[1] - lv[2] - lv[3], lv[0] - lv[1] - lv[2] + lv[3]};
#pragma unroll
            for (int i = 0; i < 4; i++) deq[i] = (fi[i] * 16 * (pl ? dqc[1][0] : dqc[0][0])) >> 5;
            if (is_chroma) d[0] = pick4(deq, cb);
        } else {
            chroma_dc(w4, P.qc, P.qc.f_intra, lv, deq);
            if (is_chroma) {
                d[0] = deq[cb];
                if (cb == 0)
#pragma unroll
                    for (int i = 0; i < 4; i++) S.lv[LV_CHROMA_DC + cpl * 4 + i] = (int16_t)lv[i];
                dcw = (lv[0] | lv[1] | lv[2] | lv[3]) != 0;
            }
        }
    }
    const unsigned long long nzmask = __ballot(nnz != 0);
    const unsigned long long dcmask = __ballot(is_chroma && dcw);
    const int cbp_luma = use_i4 ? cbp_luma_i4 : ((nzmask & 0xFFFF) ? 15 : 0);
    const int cbp_chroma = ((nzmask >> 16) & 255) ? 2 : (dcmask ? 1 : 0);
    if (is_luma || is_chroma) {
        idct4x4(d);
        uint8_t *dst, *ldst;
        const uint8_t* pp;
        int dp, ppitch;
        if (is_luma) {
            const int x = blk_x(lane) * 4, y = blk_y(lane) * 4;
            dst = P.rec[0] + (size_t)(by + y) * P.cw + bx + x; dp = P.cw;
            pp = S.py + y * 16 + x; ppitch = 16; ldst = S.rec_y + y * 16 + x;
        } else {
            const int x = (cb & 1) * 4, y = (cb >> 1) * 4;
            dst = rec_chroma(P, cpl) + (size_t)(8 * my + y) * cs + 8 * mx + x; dp = cs;
            pp = S.pc + cpl * 64 + y * 8 + x; ppitch = 8; ldst = S.rec_c + cpl * 64 + y * 8 + x;
        }
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const uint32_t p = *(const uint32_t*)(pp + r * ppitch);
            uint32_t o = 0;
#pragma unroll
            for (int c = 0; c < 4; c++) o |= (uint32_t)clip255((int)((p >> (8 * c)) & 255) + d[4 * r + c]) << (8 * c);
            *(uint32_t*)(dst + (size_t)r * dp) = o;
            *(uint32_t*)(ldst + r * ppitch) = o;
        }
    }
    wave_sync();
    if (DEC) return;
    MbInfo* m = P.mb + mbi;
    int cbp_l = cbp_luma, cbp_c = cbp_chroma, mode_l = best_mode, mode_c = best_cmode;
    {   // I_PCM fallback (dev_common.h): bit bound of the 27 blocks, one per lane; above the limit of A.3.1 the macroblock is
        // re-written as I_PCM: reconstruction = source, for the picture and for this row's next prediction alike
        // one block per lane, every lane the same code: 0..15 luma, 16..23 chroma AC, 24 luma DC (Intra16x16 only), 25, 26 chroma DC
        // (four levels: the other twelve of the sixteen count as zero)
        const int off = lane < 16 ? 2 * LV_LUMA + 32 * lane : (lane < 24 ? 2 * LV_CHROMA_AC + 32 * (lane - 16) : (lane == 24 ? 2 * LV_LUMA_DC : 2 * LV_CHROMA_DC));
        const uint4 qa = *(const uint4*)((const uint8_t*)S.lv + off), qb = *(const uint4*)((const uint8_t*)S.lv + off + 16);
        const bool cdc = lane >= 25;
        uint32_t lvp[8] = {cdc ? (lane == 26 ? qa.z : qa.x) : qa.x, cdc ? (lane == 26 ? qa.w : qa.y) : qa.y, cdc ? 0u : qa.z, cdc ? 0u : qa.w,
                           cdc ? 0u : qb.x, cdc ? 0u : qb.y, cdc ? 0u : qb.z, cdc ? 0u : qb.w};
        const int bbv = blk_bits_bound_packed(lvp, count_nz16_packed(lvp));
        const bool luma_blk = lane < 16 || lane == 24;
        const bool mine = lane < 27 && !(lane == 24 && use_i4) && (PART == 0 || (PART == 1) == luma_blk);
        const int s16 = row_sum16_dpp(mine ? bbv : 0);
        int tot = MB_HEADER_BOUND + __builtin_amdgcn_readlane(s16, 0) + __builtin_amdgcn_readlane(s16, 16);
        if (PART != 0) {   // the other wave's blocks, and what the header needs of it
            int* const out = S.xchg[mx & 1][PART - 1];
            const int* const in = S.xchg[mx & 1][2 - PART];
            if (lane == 0) { out[0] = tot - MB_HEADER_BOUND; out[1] = LUMA ? cbp_luma : cbp_chroma; out[2] = LUMA ? best_mode : best_cmode; }
            pair_barrier();
            tot += __builtin_amdgcn_readfirstlane(in[0]);
            if (LUMA) { cbp_c = in[1]; mode_c = in[2]; }
            else { cbp_l = in[1]; mode_l = in[2]; }
        }
        if (tot > MB_BITS_LIMIT) {   // wave-uniform
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (LUMA) {
                const int row = lane >> 2, xs = (lane & 3) * 4;
                const uint32_t v = *(const uint32_t*)(S.src + row * 16 + xs);
                *(uint32_t*)(P.rec[0] + (size_t)(by + row) * P.cw + bx + xs) = v;
                *(uint32_t*)(S.rec_y + row * 16 + xs) = v;
            }
            if (CHROMA && lane < 32) {
                const int pl = lane >> 4, row = (lane >> 1) & 7, xs = (lane & 1) * 4;
                const uint32_t v = *(const uint32_t*)(S.srcc + pl * 64 + row * 8 + xs);
                *(uint32_t*)(rec_chroma(P, pl) + (size_t)(8 * my + row) * cs + 8 * mx + xs) = v;
                *(uint32_t*)(S.rec_c + pl * 64 + row * 8 + xs) = v;
            }
            if (LUMA && lane < 6) ((uint32_t*)m)[2 + lane] = 0x10101010u;
            if (LUMA && lane == 6) {
                *(uint2*)m = make_uint2(0u, (uint32_t)MB_IPCM | (0x2Fu << 24));
                *(uint32_t*)(P.mvd + 8 * (size_t)mbi) = 0u;
                *P.anypcm = P.pic_serial;
            }
            wave_sync();
            return;
        }
    }
    const int tcv = lane < 16 ? (use_i4 ? tc_i4 : (cbp_l != 0 ? nnz : 0)) : (cbp_c == 2 ? nnz : 0);
    if (lane < 24 && (PART == 0 || (PART == 1) == (lane < 16))) m->tc[lane] = (uint8_t)tcv;
    if (LUMA && lane == 0) {
        m->mvx = 0; m->mvy = 0; m->type = use_i4 ? MB_I4 : MB_I16;
        m->i16_mode = (uint8_t)mode_l; m->chroma_mode = (uint8_t)mode_c;
        m->cbp = (uint8_t)(cbp_l | (cbp_c << 4));
        *(uint32_t*)(P.mvd + 8 * (size_t)mbi) = 0u;
    }
    {   // the level lists: luma DC + luma, then chroma
        uint4* g = (uint4*)(P.levels + (size_t)mbi * LV_STRIDE);
        const uint4* sl = (const uint4*)S.lv;
        const int li = (LUMA ? 0 : 2 * LV_CHROMA_DC / 16) + lane;
        if (li < (CHROMA ? LV_STRIDE * 2 / 16 : 2 * LV_CHROMA_DC / 16)) g[li] = sl[li];
    }
}

// one anti-diagonal per launch (kept as the simple reference form; MI355X_H264_DIAG=1 selects it)
__global__ __launch_bounds__(64) void k_intra_diag(FrameParams P0, int s)
{
    const FrameParams P = batch_view(P0, blockIdx.y);
    const int lane = threadIdx.x;
    const int ymin = max(0, s - P.mbw + 1);
    const int my = ymin + blockIdx.x, mx = s - my;
    if (my >= P.mbh || mx < 0 || mx >= P.mbw) return;
    const int bx = 16 * mx, by = 16 * my, cs = P.cw / 2;
    __shared__ IntraLds S;
    load_src_mb(P, mx, my, S.src, S.srcc, lane);
    {
        const uint8_t* R = P.rec[0];
        const bool top = P.sl.has_top(my);
        if (lane < 17) S.top[lane] = (top && (lane > 0 || mx > 0)) ? R[(size_t)(by - 1) * P.cw + bx - 1 + lane] : 0;
        else if (lane < 33) S.left[lane - 17] = mx > 0 ? R[(size_t)(by + lane - 17) * P.cw + bx - 1] : 0;
        else if (lane < 33 + 18) {
            const int k = lane - 33, pl = k / 9, i = k % 9;
            S.ctop[pl][i] = (P.sl.has_top(my) && (i > 0 || mx > 0)) ? rec_chroma(P, pl)[(size_t)(8 * my - 1) * cs + 8 * mx - 1 + i] : 0;
        }
        if (lane < 16) {
            const int pl = lane >> 3, i = lane & 7;
            S.cleft[pl][i] = mx > 0 ? rec_chroma(P, pl)[(size_t)(8 * my + i) * cs + 8 * mx - 1] : 0;
        }
    }
    i4_lds_init(S.i4, lane);
    wave_sync();
    const int mbi = my * P.mbw + mx;
    const bool use_i4 = ((const uint8_t*)(P.mb + mbi))[4] == MB_I4;
    const uint32_t auxw = lane < 4 ? *(const uint32_t*)(P.aux + (size_t)mbi * 16 + 4 * lane) : 0u;
    intra_mb_core(P, mx, my, S, lane, __builtin_amdgcn_readfirstlane((int)use_i4) != 0, auxw);
}

// ===========================================================================
// Persistent form: ONE launch, one wavefront per macroblock row.  Intra16x16 / chroma
// prediction of macroblock (mx, r) needs the bottom sample row of (mx, r-1) (and of
// (mx-1, r-1) for the corner), the left column comes from this row's previous
// macroblock (kept in LDS).  Row r-1 publishes the bottom row of each macroblock as
// 8 granules {tag = picture serial, 4 samples} (16 luma + 8 Cb + 8 Cr samples), the
// same fence-free hand-off as the deblocking wavefront (k_deblock.h).  The source of
// the next macroblock is requested one iteration ahead.
// ===========================================================================
struct IntraRowParams {
    FrameParams p;
    unsigned long long* handoff;  // [batch][mbh][mbw][8]
    size_t st_handoff;            // u64 words between batch items
    unsigned* err;                // pinned host word
    unsigned serial;
    int npic;                     // pictures of the step (gridDim.y of them at a time)
};

// Two waves per row (one workgroup): wave 0 codes the luma of the row's macroblocks, wave 1 their chroma, the same macroblock at the
// same time (intra_mb_core PART 1 / 2; one pair_barrier per macroblock).  Each waits for its own granules of the row above
// (0..3 luma, 4..7 chroma) and publishes its own.  What an IDR picture waits for is the chain of dependent macroblocks along the
// wavefront, 120 + 67 of them at 1080p, and a lone wave issues one instruction every five to eight cycles whatever the other
// SIMDs do: halving the instructions in the chain is worth more than the second wave costs.
template <bool IND = false>
__global__ __launch_bounds__(128) void k_intra_rows(IntraRowParams R)
{
    __builtin_amdgcn_s_setprio(3);   // dependency-bound row wavefront: issue ahead of co-resident throughput kernels
    const int lane0 = threadIdx.x & 63;
    const bool chroma_wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) != 0;
    __shared__ IntraLds S;
    bool timed_out = false;
    if (!chroma_wave) i4_lds_init(S.i4, lane0);
    // The launch holds gridDim.y pictures at a time, not all of them: workgroup (row, y) takes pictures y, y + gridDim.y, ...  A row
    // wavefront is bound by its dependencies, not by the machine, and every resident wave holds registers the other instance's
    // kernels could run in: with all 32 pictures of a lockstep batch resident (4 352 waves of ~150 registers, three to a SIMD) neither
    // k_me nor the loop filter of the other instance found room beside them, and what the faster IDR step gained they lost.
    // (Row r of a picture waits for row r - 1 of the same picture, which the workgroup dispatched just before handles: whatever part of the
    // grid is resident, the row being waited for is in it.)
    for (int pic = blockIdx.y; pic < R.npic; pic += gridDim.y) {
    // (the lane number is made opaque per picture: otherwise every per-lane constant of BOTH waves' macroblock code is computed once
    // in front of this loop and kept alive across it - 240 VGPRs instead of 146, two waves to a SIMD instead of three)
    int lane = lane0;
    asm volatile("" : "+v"(lane));
    const FrameParams P = batch_view<IND>(R.p, pic);
    unsigned long long* const handoff = R.handoff + (size_t)batch_item<IND>(R.p.itemtab, pic) * R.st_handoff;
    const int my = P.band.row0 + blockIdx.x;
    const bool top = P.sl.has_top(my);   // first row of a slice: nothing above to wait for, the slices' wavefronts run side by side
    // this wave's four granules of macroblock mx of the row above / of this row
    const int gl = (chroma_wave ? 4 : 0) + (lane & 3);
    // A row that is not due yet polls at priority 0 and ever more rarely (all rows of all pictures are resident from the start, most
    // of them waiting: at the row wavefront's priority their polling took issue slots from the other instance's kernels); once
    // the row above is a macroblock ahead the granules requested one macroblock earlier are there and nothing is polled.
    auto wait_above = [&](unsigned long long g, int mx) {
        unsigned spins = 0;
        bool low = false;
        while (!timed_out) {
            const bool bad = lane < 4 && (unsigned)(g >> 32) != R.serial;
            if (__ballot(bad) == 0ull) break;
            if (++spins > (1u << 19)) { timed_out = true; break; }
            if (!low) { __builtin_amdgcn_s_setprio(0); low = true; }
            if (spins < 4) __builtin_amdgcn_s_sleep(1);
            else if (spins < 64) __builtin_amdgcn_s_sleep(8);
            else __builtin_amdgcn_s_sleep(32);
            if (lane < 4) g = __hip_atomic_load(handoff + ((size_t)(my - 1) * P.mbw + mx) * 8 + gl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (low) __builtin_amdgcn_s_setprio(3);
        return g;
    };
    if (!chroma_wave) {
        const uint8_t* Y = P.src;
        uint32_t pf_y = 0, pf_aux = 0;
        int pf_type = 0;
        unsigned long long pf_g = 0;
        auto prefetch = [&](int mx) {   // source, k_i4_decide's verdict (type and, lanes 0..3, the sixteen Intra4x4 modes), the row above
            const int mbi = my * P.mbw + mx;
            pf_type = ((const uint8_t*)(P.mb + mbi))[4];
            if (lane < 4) pf_aux = *(const uint32_t*)(P.aux + (size_t)mbi * 16 + 4 * lane);
            const int row = lane >> 2, xs = (lane & 3) * 4;
            const int gy = 16 * my + row, gx = 16 * mx + xs;
            const uint8_t* p = Y + (size_t)(gy < P.h ? gy : P.h - 1) * P.w + gx;
            if (gx + 3 < P.w && (((uintptr_t)p) & 3) == 0) pf_y = *(const uint32_t*)p;
            else {
                pf_y = 0;
#pragma unroll
                for (int k = 0; k < 4; k++) pf_y |= (uint32_t)src_px(Y, P.w, P.h, gx + k, gy) << (8 * k);
            }
            if (top && lane < 4) pf_g = __hip_atomic_load(handoff + ((size_t)(my - 1) * P.mbw + mx) * 8 + gl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        };
        prefetch(0);
        for (int mx = 0; mx < P.mbw; mx++) {
            const uint32_t cur_y = pf_y, cur_aux = pf_aux;
            const bool cur_i4 = __builtin_amdgcn_readfirstlane(pf_type) == MB_I4;
            unsigned long long g = pf_g;
            if (mx + 1 < P.mbw) prefetch(mx + 1);
            // left neighbours = last column of the previous reconstruction; corner = last sample of the previous top row
            if (mx > 0) {
                if (lane < 16) S.left[lane] = S.rec_y[lane * 16 + 15];
                else if (lane == 32) S.top[0] = S.top[16];
            }
            *(uint32_t*)(S.src + (lane >> 2) * 16 + (lane & 3) * 4) = cur_y;
            wave_sync();   // corner moved before the top row is overwritten
            if (top) {
                g = wait_above(g, mx);
                if (lane < 4) {   // granules 0..3: luma samples 0..15 -> top[1..16]
                    const uint32_t v = (uint32_t)g;
#pragma unroll
                    for (int k = 0; k < 4; k++) S.top[1 + 4 * lane + k] = (uint8_t)(v >> (8 * k));
                }
            }
            wave_sync();
            intra_mb_core<false, 1>(P, mx, my, S, lane, cur_i4, cur_aux);
            if (my + 1 < P.mbh && lane < 4)   // publish this macroblock's bottom sample row for the row below
                __hip_atomic_store(handoff + ((size_t)my * P.mbw + mx) * 8 + gl, ((unsigned long long)R.serial << 32) | *(const uint32_t*)(S.rec_y + 15 * 16 + 4 * lane),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            wave_sync();
        }
    } else {
        uint32_t pf_c = 0;
        unsigned long long pf_g = 0;
        auto prefetch = [&](int mx) {
            if (lane < 32) {
                const int pl = lane >> 4, row = (lane >> 1) & 7, xs = (lane & 1) * 4;
                pf_c = src_chroma4(P, pl, 8 * mx + xs, 8 * my + row);
            }
            if (top && lane < 4) pf_g = __hip_atomic_load(handoff + ((size_t)(my - 1) * P.mbw + mx) * 8 + gl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        };
        prefetch(0);
        for (int mx = 0; mx < P.mbw; mx++) {
            const uint32_t cur_c = pf_c;
            unsigned long long g = pf_g;
            if (mx + 1 < P.mbw) prefetch(mx + 1);
            if (mx > 0) {
                if (lane < 16) S.cleft[lane >> 3][lane & 7] = S.rec_c[(lane >> 3) * 64 + (lane & 7) * 8 + 7];
                else if (lane == 32) S.ctop[0][0] = S.ctop[0][8];
                else if (lane == 33) S.ctop[1][0] = S.ctop[1][8];
            }
            if (lane < 32) *(uint32_t*)(S.srcc + (lane >> 4) * 64 + ((lane >> 1) & 7) * 8 + (lane & 1) * 4) = cur_c;
            wave_sync();
            if (top) {
                g = wait_above(g, mx);
                if (lane < 4) {   // granules 4, 5: Cb samples 0..7 -> ctop[0][1..8]; 6, 7: Cr -> ctop[1][1..8]
                    const uint32_t v = (uint32_t)g;
                    uint8_t* dst = S.ctop[lane >> 1] + 1 + 4 * (lane & 1);
#pragma unroll
                    for (int k = 0; k < 4; k++) dst[k] = (uint8_t)(v >> (8 * k));
                }
            }
            wave_sync();
            intra_mb_core<false, 2>(P, mx, my, S, lane, false, 0u);
            if (my + 1 < P.mbh && lane < 4)
                __hip_atomic_store(handoff + ((size_t)my * P.mbw + mx) * 8 + gl, ((unsigned long long)R.serial << 32) | *(const uint32_t*)(S.rec_c + (lane >> 1) * 64 + 7 * 8 + 4 * (lane & 1)),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            wave_sync();
        }
    }
    }
    if (timed_out && lane0 == 0) *R.err = 2u;
}

// ===========================================================================
// Intra macroblocks inside P pictures.  k_me marks them (MbInfo.type == MB_I16, no inter prediction written), k_tq has
// reconstructed every inter macroblock of the picture in an earlier launch; this launch codes the marked macroblocks in
// raster-order dependency: one persistent wave per macroblock row walks its marked macroblocks left to right.  A
// neighbour that is an inter macroblock is read from the reconstruction planes (final since the earlier launch); a
// neighbour that is itself intra is taken from this wave's LDS copy (left) or from the granules the row above
// publishes (above, above-left) - the same {tag, 4 samples} hand-off as k_intra_rows.  A row waits only for intra
// macroblocks of the row above, which are coded left to right: no cycle.  Pictures without marked macroblocks return at once.
// ===========================================================================
// DEC (the decoder peer): the intra macroblocks are those of intra type in the MbInfo the host parser filled (every macroblock
// of an I picture); they are reconstructed from the given modes and levels.
template <bool DEC, bool IND = false>
__global__ __launch_bounds__(64) void k_pintra_rows(IntraRowParams R)
{
    __builtin_amdgcn_s_setprio(3);
    const int lane = threadIdx.x;
    __shared__ IntraLds S;
    bool tab_ready = false;   // the Intra4x4 address table is built by the first macroblock that needs this wave: most rows of most P pictures have none
    bool timed_out = false;
    // gridDim.y pictures at a time, workgroup (row, y) taking pictures y, y + gridDim.y, ... (as k_intra_rows): this kernel is on every P
    // step's chain, nearly all of its waves find nothing to do, and each needs ~200 registers to be placed - 2 176 of them per
    // step beside the other instance's motion search (80 registers a wave, six to a SIMD) cost 7 % of the throughput, 272 cost nothing.
    for (int pic = blockIdx.y; pic < R.npic; pic += gridDim.y) {
    const FrameParams P = batch_view<IND>(R.p, pic);
    if (!DEC && *P.anyintra != P.pic_serial) continue;
    unsigned long long* const handoff = R.handoff + (size_t)batch_item<IND>(R.p.itemtab, pic) * R.st_handoff;
    const int my = P.band.row0 + blockIdx.x, cs = P.cw / 2;
    const bool top = P.sl.has_top(my);
    int last_done = -2;   // the macroblock whose reconstruction S.rec_* holds
    // bit 15 of me_cost = "handed to this pass by k_me": unlike MbInfo.type (an I_PCM conversion changes it, here or in k_tq)
    // it does not change during the launch, so every row sees the same set of macroblocks to wait for
    auto marked = [&](int x, int y) {
        if (DEC) return mb_is_intra(P.mb[(size_t)y * P.mbw + x].type);
        return (P.me_cost[(size_t)y * P.mbw + x] & 0x8000u) != 0;
    };
    auto wait_granules = [&](int gx, int nlanes, unsigned long long& g) {   // lanes < nlanes: granule `lane` of macroblock (gx, my - 1)
        const unsigned long long* src = handoff + ((size_t)(my - 1) * P.mbw + gx) * 8;
        unsigned spins = 0;
        g = lane < nlanes ? __hip_atomic_load(src + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
        while (!timed_out) {
            const bool bad = lane < nlanes && (unsigned)(g >> 32) != R.serial;
            if (__ballot(bad) == 0ull) break;
            if (++spins > (1u << 20)) { timed_out = true; break; }
            __builtin_amdgcn_s_sleep(1);
            if (lane < nlanes) g = __hip_atomic_load(src + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    };
    for (int base = 0; base < P.mbw; base += 64) {
        const int xm = base + lane;
        unsigned long long todo = __ballot(xm < P.mbw && marked(xm < P.mbw ? xm : 0, my));
        while (todo) {
            const int mx = base + __ffsll((long long)todo) - 1;
            todo &= todo - 1;
            if (!DEC) load_src_mb(P, mx, my, S.src, S.srcc, lane);
            // left column + corner sources
            if (mx > 0) {
                if (last_done == mx - 1) {
                    if (lane < 16) S.left[lane] = S.rec_y[lane * 16 + 15];
                    else if (lane < 32) S.cleft[(lane >> 3) & 1][lane & 7] = S.rec_c[((lane >> 3) & 1) * 64 + (lane & 7) * 8 + 7];
                } else {
                    if (lane < 16) S.left[lane] = P.rec[0][(size_t)(16 * my + lane) * P.cw + 16 * mx - 1];
                    else if (lane < 32) S.cleft[(lane >> 3) & 1][lane & 7] = rec_chroma(P, lane & 8)[(size_t)(8 * my + (lane & 7)) * cs + 8 * mx - 1];
                }
            }
            wave_sync();
            if (top) {
                const bool aI = marked(mx, my - 1), dI = mx > 0 && marked(mx - 1, my - 1);   // wave-uniform
                if (aI) {
                    unsigned long long g;
                    wait_granules(mx, 8, g);
                    if (lane < 8) {
                        const uint32_t v = (uint32_t)g;
                        uint8_t* dst = lane < 4 ? S.top + 1 + 4 * lane : S.ctop[(lane - 4) >> 1] + 1 + 4 * (lane & 1);
#pragma unroll
                        for (int k = 0; k < 4; k++) dst[k] = (uint8_t)(v >> (8 * k));
                    }
                } else {
                    if (lane < 16) S.top[1 + lane] = P.rec[0][(size_t)(16 * my - 1) * P.cw + 16 * mx + lane];
                    else if (lane < 32) S.ctop[(lane >> 3) & 1][1 + (lane & 7)] = rec_chroma(P, lane & 8)[(size_t)(8 * my - 1) * cs + 8 * mx + (lane & 7)];
                }
                if (DEC && mx + 1 < P.mbw) {   // the decoder also needs the four samples above-right (Intra4x4 modes 3 / 7 of block (3, 0))
                    if (marked(mx + 1, my - 1)) {
                        unsigned long long g;
                        wait_granules(mx + 1, 1, g);
                        if (lane == 0) {
                            const uint32_t v = (uint32_t)g;
#pragma unroll
                            for (int k = 0; k < 4; k++) S.top[17 + k] = (uint8_t)(v >> (8 * k));
                        }
                    } else if (lane < 4) S.top[17 + lane] = P.rec[0][(size_t)(16 * my - 1) * P.cw + 16 * (mx + 1) + lane];
                }
                if (mx > 0) {
                    if (dI) {   // last samples of the bottom rows of the macroblock above-left: granules 3 (luma), 5 (Cb), 7 (Cr)
                        unsigned long long g;
                        wait_granules(mx - 1, 8, g);
                        if (lane == 3) S.top[0] = (uint8_t)((uint32_t)g >> 24);
                        else if (lane == 5) S.ctop[0][0] = (uint8_t)((uint32_t)g >> 24);
                        else if (lane == 7) S.ctop[1][0] = (uint8_t)((uint32_t)g >> 24);
                    } else {
                        if (lane == 0) S.top[0] = P.rec[0][(size_t)(16 * my - 1) * P.cw + 16 * mx - 1];
                        else if (lane == 1) S.ctop[0][0] = P.rec[1][(size_t)(8 * my - 1) * cs + 8 * mx - 1];
                        else if (lane == 2) S.ctop[1][0] = P.rec[2][(size_t)(8 * my - 1) * cs + 8 * mx - 1];
                    }
                }
            }
            wave_sync();
            {
                const int mbi = my * P.mbw + mx;
                const bool use_i4 = ((const uint8_t*)(P.mb + mbi))[4] == MB_I4;
                const uint32_t auxw = lane < 4 ? *(const uint32_t*)(P.aux + (size_t)mbi * 16 + 4 * lane) : 0u;
                if (!tab_ready) { i4_lds_init(S.i4, lane); tab_ready = true; wave_sync(); }
                intra_mb_core<DEC>(P, mx, my, S, lane, __builtin_amdgcn_readfirstlane((int)use_i4) != 0, auxw);
            }
            last_done = mx;
            // publish the bottom sample row for the row below (it asks only where this macroblock is its neighbour)
            if (my + 1 < P.mbh && lane < 8) {
                uint32_t v;
                if (lane < 4) v = *(const uint32_t*)(S.rec_y + 15 * 16 + 4 * lane);
                else v = *(const uint32_t*)(S.rec_c + ((lane - 4) >> 1) * 64 + 7 * 8 + 4 * (lane & 1));
                __hip_atomic_store(handoff + ((size_t)my * P.mbw + mx) * 8 + lane, ((unsigned long long)R.serial << 32) | v,
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            wave_sync();
        }
    }
    }
    if (timed_out && lane == 0) *R.err = 3u;
}

}  // namespace h264
