// media_amd/csrc/k_pmb.h -- inter macroblock coding, one wavefront per macroblock:
// motion compensation (luma quarter-pel 6-tap, chroma eighth-pel bilinear,
// 8.4.2.2) -> residual -> 4x4 forward transform -> quantisation -> scaling ->
// inverse transform (8.5.12) -> reconstruction, plus the P_Skip decision.
//
// SURVEY.md 8a rows a6.2 + a6.3 (inside ISVCEncoder::EncodeFrame,
// /root/reference/video_codec/VideoEncoderOpenH264.cpp:344).
//
// HBM traffic per macroblock (algorithmic): source 384 B + reference 384 B in,
// reconstruction 384 B + levels 768 B + side info 32 B out.
#pragma once
#include "dev_common.h"

namespace h264 {

// Transform / quantise / reconstruct one 4x4 block held by one lane.
// d: residual in, reconstructed residual out.  lv: zig-zag levels out.
// first = 1 skips the DC position (coded separately).  Returns count of non-zero
// levels over positions [first..15]; *dc_w receives the forward-transform DC.
__device__ __forceinline__ int tq4x4(int d[16], const Quant& q, int f, int first, int16_t* lvz, int* dc_w,
                                     int dc_deq, bool have_dc_deq)
{
    fdct4x4(d);
    if (dc_w) *dc_w = d[0];
    int nnz = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const int cls = pos_class(i);
        int l = (i == 0 && first) ? 0 : quant1(d[i], q.mf[cls], f, q.qbits);
        lvz[c_zigzag_inv[i]] = (int16_t)l;
        nnz += l != 0;
        d[i] = l * q.dq[cls];
    }
    if (have_dc_deq) d[0] = dc_deq;
    return nnz;
}

// Chroma DC of one plane: fwd 2x2 Hadamard of the four block DCs, quantise,
// inverse Hadamard + scaling (8.5.11).  dcw[4] in; lv[4] and deq[4] out.
__device__ __forceinline__ void chroma_dc(const int dcw[4], const Quant& q, int f, int lv[4], int deq[4])
{
    const int fd[4] = {dcw[0] + dcw[1] + dcw[2] + dcw[3], dcw[0] - dcw[1] + dcw[2] - dcw[3],
                       dcw[0] + dcw[1] - dcw[2] - dcw[3], dcw[0] - dcw[1] - dcw[2] + dcw[3]};
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int a = iabs(fd[i]);
        const int l = (int)(((unsigned)a * (unsigned)q.mf[0] + 2u * (unsigned)f) >> (q.qbits + 1));
        lv[i] = fd[i] < 0 ? -l : l;
    }
    const int fi[4] = {lv[0] + lv[1] + lv[2] + lv[3], lv[0] - lv[1] + lv[2] - lv[3],
                       lv[0] + lv[1] - lv[2] - lv[3], lv[0] - lv[1] - lv[2] + lv[3]};
    // ((f * 16 v0) << (qp/6)) >> 5  with q.dq[0] = v0 << (qp/6)
#pragma unroll
    for (int i = 0; i < 4; i++) deq[i] = (fi[i] * 16 * q.dq[0]) >> 5;
}

__global__ __launch_bounds__(64) void k_pmb(FrameParams P0)
{
    const FrameParams P = batch_view(P0, blockIdx.y);
    const int lane = threadIdx.x;
    const int mbi = P.band.row0 * P.mbw + xcd_mb_index(blockIdx.x, P.mbw * P.band.rows), my = P.mbdiv.row(mbi), mx = mbi - my * P.mbw;
    const int bx = 16 * mx, by = 16 * my, cs = P.cw / 2;

    __shared__ __attribute__((aligned(16))) uint8_t s_src[256];
    __shared__ __attribute__((aligned(16))) uint8_t s_srcc[128];
    __shared__ __attribute__((aligned(16))) uint8_t s_w[21 * 28 + 4];   // luma window, pitch 28 (7 aligned dwords per row)
    __shared__ __attribute__((aligned(16))) uint8_t s_cw[2][9 * 12 + 4];  // chroma windows, pitch 12 (3 aligned dwords per row)
    __shared__ int16_t s_b1[21 * 16];
    __shared__ __attribute__((aligned(16))) uint8_t s_py[256];
    __shared__ __attribute__((aligned(16))) uint8_t s_pc[128];
    __shared__ __attribute__((aligned(16))) int16_t s_lv[LV_STRIDE];

    MbInfo* m = P.mb + mbi;
    const int mvx = m->mvx, mvy = m->mvy;
    Mv skip;
    const Mv pred = predict_mv(P, mx, my, skip);

    load_src_mb(P, mx, my, s_src, s_srcc, lane);
    for (int i = lane; i < LV_STRIDE / 2; i += 64) ((uint32_t*)s_lv)[i] = 0;
    // luma window: 21x21 samples from (bx + ix - 2, by + iy - 2), clamped at the picture edge.
    // Interior macroblocks fetch it as aligned dwords (7 per row) with all requests in flight together;
    // wxo is the column of the first wanted sample inside the 28-byte LDS row.
    const int x0 = bx + (mvx >> 2) - 2, y0 = by + (mvy >> 2) - 2;
    const int xa = x0 & ~3, wxo = x0 - xa;
    const int cx0 = 8 * mx + (mvx >> 3), cy0 = 8 * my + (mvy >> 3);
    const int cxa = cx0 & ~3, cxo = cx0 - cxa;
    {
        const bool interior = xa >= 0 && xa + 28 <= P.cw && y0 >= 0 && y0 + 21 <= P.ch;
        if (interior) {
            uint32_t v[3];
#pragma unroll
            for (int t = 0; t < 3; t++) {
                const int i = lane + 64 * t, r = i / 7, c = i - r * 7;
                v[t] = i < 147 ? *(const uint32_t*)(P.ref[0] + (size_t)(y0 + r) * P.cw + xa + 4 * c) : 0;
            }
#pragma unroll
            for (int t = 0; t < 3; t++) {
                const int i = lane + 64 * t, r = i / 7, c = i - r * 7;
                if (i < 147) *(uint32_t*)(s_w + r * 28 + 4 * c) = v[t];
            }
        } else {
            for (int i = lane; i < 21 * 21; i += 64) {
                const int r = i / 21, c = i - r * 21;
                s_w[r * 28 + wxo + c] = P.ref[0][(size_t)clip3(0, P.ch - 1, y0 + r) * P.cw + clip3(0, P.cw - 1, x0 + c)];
            }
        }
        const bool cinterior = cxa >= 0 && cxa + 12 <= cs && cy0 >= 0 && cy0 + 9 <= P.ch / 2;
        if (cinterior) {
            if (lane < 54) {   // 2 planes x 9 rows x 3 dwords
                const int pl = lane / 27, k = lane - pl * 27, r = k / 3, c = k - r * 3;
                *(uint32_t*)(s_cw[pl] + r * 12 + 4 * c) = *(const uint32_t*)((pl ? P.ref[2] : P.ref[1]) + (size_t)(cy0 + r) * cs + cxa + 4 * c);
            }
        } else {
            for (int i = lane; i < 2 * 81; i += 64) {
                const int pl = i / 81, k = i - pl * 81, r = k / 9, c = k - r * 9;
                s_cw[pl][r * 12 + cxo + c] =
                    (pl ? P.ref[2] : P.ref[1])[(size_t)clip3(0, P.ch / 2 - 1, cy0 + r) * cs + clip3(0, cs - 1, cx0 + c)];
            }
        }
    }
    __syncthreads();
    const int fx = mvx & 3, fy = mvy & 3;
    if (fx) {
        for (int i = lane; i < 21 * 16; i += 64) {
            const int r = i >> 4, c = i & 15;
            const uint8_t* p = s_w + r * 28 + wxo + c;  // p[2] is sample (c, r-2)
            s_b1[i] = (int16_t)(p[0] - 5 * p[1] + 20 * p[2] + 20 * p[3] - 5 * p[4] + p[5]);
        }
        __syncthreads();
    }
    {   // luma prediction: lane -> 4 samples of one row
        const int y = lane >> 2, xs = (lane & 3) * 4;
        uint32_t out = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int x = xs + k;
            const uint8_t* g = s_w + (y + 2) * 28 + wxo + x + 2;  // integer sample G(x,y)
#define HV(px) clip255(((px)[-56] - 5 * (px)[-28] + 20 * (px)[0] + 20 * (px)[28] - 5 * (px)[56] + (px)[84] + 16) >> 5)
#define BH(xx, yy) clip255((s_b1[((yy) + 2) * 16 + (xx)] + 16) >> 5)
#define JC(xx, yy) clip255((s_b1[(yy) * 16 + (xx)] - 5 * s_b1[((yy) + 1) * 16 + (xx)] + 20 * s_b1[((yy) + 2) * 16 + (xx)] + \
                            20 * s_b1[((yy) + 3) * 16 + (xx)] - 5 * s_b1[((yy) + 4) * 16 + (xx)] + s_b1[((yy) + 5) * 16 + (xx)] + 512) >> 10)
            int v;
            if (fy == 0) {
                if (fx == 0) v = g[0];
                else {
                    const int b = BH(x, y);
                    v = fx == 2 ? b : (fx == 1 ? (g[0] + b + 1) >> 1 : (g[1] + b + 1) >> 1);
                }
            } else if (fx == 0) {
                const int h = HV(g);
                v = fy == 2 ? h : (fy == 1 ? (g[0] + h + 1) >> 1 : (g[28] + h + 1) >> 1);
            } else if (fx == 2 && fy == 2) {
                v = JC(x, y);
            } else if (fx == 2) {
                v = ((fy == 1 ? BH(x, y) : BH(x, y + 1)) + JC(x, y) + 1) >> 1;
            } else if (fy == 2) {
                const int h = fx == 1 ? HV(g) : HV(g + 1);
                v = (h + JC(x, y) + 1) >> 1;
            } else {
                const int b = fy == 1 ? BH(x, y) : BH(x, y + 1);
                const int h = fx == 1 ? HV(g) : HV(g + 1);
                v = (b + h + 1) >> 1;
            }
#undef HV
#undef BH
#undef JC
            out |= (uint32_t)v << (8 * k);
        }
        *(uint32_t*)(s_py + y * 16 + xs) = out;
    }
    if (lane < 32) {  // chroma prediction
        const int pl = lane >> 4, y = (lane >> 1) & 7, xs = (lane & 1) * 4;
        const int dx = mvx & 7, dy = mvy & 7;
        uint32_t out = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint8_t* p = s_cw[pl] + y * 12 + cxo + xs + k;
            const int v = ((8 - dx) * (8 - dy) * p[0] + dx * (8 - dy) * p[1] + (8 - dx) * dy * p[12] + dx * dy * p[13] + 32) >> 6;
            out |= (uint32_t)v << (8 * k);
        }
        *(uint32_t*)(s_pc + pl * 64 + y * 8 + xs) = out;
    }
    __syncthreads();

    // ---- transform / quant / reconstruct: lanes 0..15 luma, 16..19 Cb, 20..23 Cr ----
    int nnz = 0, dcw = 0;
    int d[16];
    const bool is_luma = lane < 16, is_chroma = lane >= 16 && lane < 24;
    const int cpl = (lane - 16) >> 2, cb = lane & 3;
    if (is_luma) {
        const int x = blk_x(lane) * 4, y = blk_y(lane) * 4;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const uint32_t s = *(const uint32_t*)(s_src + (y + r) * 16 + x), p = *(const uint32_t*)(s_py + (y + r) * 16 + x);
#pragma unroll
            for (int c = 0; c < 4; c++) d[4 * r + c] = (int)((s >> (8 * c)) & 255) - (int)((p >> (8 * c)) & 255);
        }
        nnz = tq4x4(d, P.qy, P.qy.f_inter, 0, s_lv + LV_LUMA + lane * 16, nullptr, 0, false);
    } else if (is_chroma) {
        const int x = (cb & 1) * 4, y = (cb >> 1) * 4;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const uint32_t s = *(const uint32_t*)(s_srcc + cpl * 64 + (y + r) * 8 + x), p = *(const uint32_t*)(s_pc + cpl * 64 + (y + r) * 8 + x);
#pragma unroll
            for (int c = 0; c < 4; c++) d[4 * r + c] = (int)((s >> (8 * c)) & 255) - (int)((p >> (8 * c)) & 255);
        }
        nnz = tq4x4(d, P.qc, P.qc.f_inter, 1, s_lv + LV_CHROMA_AC + (cpl * 4 + cb) * 16, &dcw, 0, false);
    }
    // chroma DC across the four lanes of each plane
    {
        const int base = 16 + ((lane - 16) & 4);
        const int w4[4] = {__shfl(dcw, base), __shfl(dcw, base + 1), __shfl(dcw, base + 2), __shfl(dcw, base + 3)};
        int lv[4], deq[4];
        chroma_dc(w4, P.qc, P.qc.f_inter, lv, deq);
        if (is_chroma) {
            d[0] = deq[cb];
            if (cb == 0)
#pragma unroll
                for (int i = 0; i < 4; i++) s_lv[LV_CHROMA_DC + cpl * 4 + i] = (int16_t)lv[i];
            dcw = (lv[0] | lv[1] | lv[2] | lv[3]) != 0;  // any DC level in this plane
        }
    }
    const unsigned long long nzmask = __ballot(nnz != 0);
    const unsigned long long dcmask = __ballot(is_chroma && dcw);
    const int cbp_luma = (((nzmask >> 0) & 15) ? 1 : 0) | (((nzmask >> 4) & 15) ? 2 : 0) | (((nzmask >> 8) & 15) ? 4 : 0) |
                         (((nzmask >> 12) & 15) ? 8 : 0);
    const int cbp_chroma = ((nzmask >> 16) & 255) ? 2 : (dcmask ? 1 : 0);
    const int cbp = cbp_luma | (cbp_chroma << 4);
    // reconstruction
    if (is_luma || is_chroma) {
        idct4x4(d);
        uint8_t* dst;
        const uint8_t* pp;
        int dp, ppitch;
        if (is_luma) {
            const int x = blk_x(lane) * 4, y = blk_y(lane) * 4;
            dst = P.rec[0] + (size_t)(by + y) * P.cw + bx + x; dp = P.cw;
            pp = s_py + y * 16 + x; ppitch = 16;
        } else {
            const int x = (cb & 1) * 4, y = (cb >> 1) * 4;
            dst = (cpl ? P.rec[2] : P.rec[1]) + (size_t)(8 * my + y) * cs + 8 * mx + x; dp = cs;
            pp = s_pc + cpl * 64 + y * 8 + x; ppitch = 8;
        }
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const uint32_t p = *(const uint32_t*)(pp + r * ppitch);
            uint32_t o = 0;
#pragma unroll
            for (int c = 0; c < 4; c++) o |= (uint32_t)clip255((int)((p >> (8 * c)) & 255) + d[4 * r + c]) << (8 * c);
            *(uint32_t*)(dst + (size_t)r * dp) = o;
        }
    }
    __syncthreads();
    // side info + levels
    const int tcv = (lane < 16 || cbp_chroma == 2) ? nnz : 0;
    if (lane < 24) m->tc[lane] = (uint8_t)tcv;
    if (lane == 0) {
        m->type = (cbp == 0 && skip.x == mvx && skip.y == mvy) ? MB_PSKIP : MB_P16;
        m->i16_mode = 0; m->chroma_mode = 0; m->cbp = (uint8_t)cbp;
        P.mvd[2 * mbi] = (int16_t)(mvx - pred.x);
        P.mvd[2 * mbi + 1] = (int16_t)(mvy - pred.y);
    }
    {
        uint4* g = (uint4*)(P.levels + (size_t)mbi * LV_STRIDE);
        const uint4* s = (const uint4*)s_lv;
        if (lane < LV_STRIDE * 2 / 16) g[lane] = s[lane];
    }
}

}  // namespace h264
