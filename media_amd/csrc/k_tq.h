// media_amd/csrc/k_tq.h -- inter macroblock coding after the motion search: residual -> 4x4 forward transform ->
// quantisation -> scaling -> inverse transform (8.5.12) -> reconstruction, as ONE streaming pass over the picture, plus
// the small kernel that turns final vectors into motion vector differences and P_Skip decisions (8.4.1.1, 8.4.1.3).
//
// SURVEY.md 8a row a6.2 (inside ISVCEncoder::EncodeFrame, /root/reference/video_codec/VideoEncoderOpenH264.cpp:344).
// This is the kernel the bench's `roofline` block prices against HBM.  Algorithmic HBM bytes per macroblock:
// source 384 + prediction 384 in; reconstruction 384 + levels 768 + side info 32 out = 1952.
//
// What changed against round 1's k_pmb2 (one wave per macroblock pair, lane = (4x4 block, row), column passes over
// DPP quads, motion compensation in front): the prediction now arrives from k_me, which has every sample of it in LDS
// anyway and writes it straight into the reconstruction planes; this kernel reads source + prediction, and overwrites
// the prediction with the reconstruction in place.  The lane mapping follows what tools/ubench_issue.hip measured on
// MI355X (profiles/r02_ubench_issue.jsonl): a plain 32-bit VOP2 add / sub / and / or / xor / right shift with VGPR
// operands issues in ~2.3 cycles per wave, every VOP3, packed-16, DPP, SDWA or SGPR-operand form in ~4.15.  So:
//   - lane = ONE 4x4 block, all 16 samples in registers: both passes of the transforms are in-lane adds and
//     subtractions (no DPP, no LDS), the zig-zag is register renaming;
//   - a wave codes EIGHT macroblocks: two luma passes of 4 macroblocks x 16 blocks, one chroma pass of
//     8 macroblocks x 2 planes x 4 blocks - every pass uses all 64 lanes;
//   - the quantiser works on signed values, l = (w * mf + (w < 0 ? 2^q - 1 - f : f)) >> q (equal to the
//     sign / magnitude form of the oracle for every w), with its wave-uniform constants pinned in VGPRs;
//   - TotalCoeff is counted on the packed int16 level pairs (6 fast operations per pair).
// Macroblocks whose prediction already quantises to nothing (k_me's tests) are not touched at all: k_me has written
// their side info, and prediction = reconstruction; macroblocks k_me handed to the intra pass are skipped too.
// I_PCM fallback: every block's levels also give an upper bound on its CAVLC bits (dev_common.h, derivation in
// oracle/h264_enc.c); a macroblock whose bound passes the 3200 bits of A.3.1 is re-written as I_PCM on the spot
// (reconstruction = source), so no picture is ever refused for its content and nothing downstream has to change its mind.
#pragma once
#include "dev_common.h"

namespace h264 {

// keep a wave-uniform value in a VGPR: VOP2 with an SGPR operand issues at half the rate of the all-VGPR form
__device__ __forceinline__ int vreg(int x)
{
    asm volatile("" : "+v"(x));
    return x;
}

struct TqConst { int mf[3], dq[3], f, c, q; };   // c = 2^q - 1 - 2 f (bias of negative values minus f)
__device__ __forceinline__ TqConst tq_consts(const Quant& qn)
{
    TqConst k;
#pragma unroll
    for (int i = 0; i < 3; i++) { k.mf[i] = vreg(qn.mf[i]); k.dq[i] = vreg(qn.dq[i]); }
    k.f = vreg(qn.f_inter);
    k.c = vreg((1 << qn.qbits) - 1 - 2 * qn.f_inter);
    k.q = vreg(qn.qbits);
    return k;
}
// sign(w) * ((|w| * mf + f) >> q) without the absolute value: for w < 0, -floor((|w| mf + f) / 2^q) = floor((w mf + 2^q - 1 - f) / 2^q)
__device__ __forceinline__ int quant_signed(int w, int mf, int f, int c, int q)
{
    const int s = w >> 31;
    return (__mul24(w, mf) + (f + (s & c))) >> q;
}

// forward transform, quantisation, scaling and inverse transform of the 4x4 block in d[] (raster; residual in, decoded
// residual out).  CHROMA: position 0 is left out of the levels (it goes through the 2x2 Hadamard), *dcw receives the
// transformed DC and dc_deq() supplies the scaled DC before the inverse transform.
// lvp[k] = zig-zag levels 2k (low half) and 2k + 1 (high half) as int16; returns TotalCoeff.
template <bool CHROMA, class DcFn>
__device__ __forceinline__ int tq_block(int d[16], const TqConst& K, uint32_t lvp[8], DcFn dc_deq)
{
    fdct4x4(d);
    int l[16];
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const int cls = ((i & 1) & ((i >> 2) & 1)) ? 1 : (((i & 1) | ((i >> 2) & 1)) ? 2 : 0);
        l[i] = (CHROMA && i == 0) ? 0 : quant_signed(d[i], K.mf[cls], K.f, K.c, K.q);
    }
    constexpr int ZZ[16] = {0, 1, 4, 8, 5, 2, 3, 6, 9, 12, 13, 10, 7, 11, 14, 15};   // scan index -> raster position
    uint32_t cnt2 = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const uint32_t p = ((uint32_t)l[ZZ[2 * k]] & 0xFFFFu) | ((uint32_t)l[ZZ[2 * k + 1]] << 16);
        lvp[k] = p;
        const uint32_t t = p | ((p & 0x7FFF7FFFu) + 0x7FFF7FFFu);   // bit 15 / 31: that half is non-zero
        cnt2 += (t >> 15) & 0x00010001u;
    }
    const int dc = CHROMA ? dc_deq(d[0]) : 0;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const int cls = ((i & 1) & ((i >> 2) & 1)) ? 1 : (((i & 1) | ((i >> 2) & 1)) ? 2 : 0);
        d[i] = __mul24(l[i], K.dq[cls]);
    }
    if (CHROMA) d[0] = dc;
    idct4x4(d);
    return (int)((cnt2 & 0xFFFFu) + (cnt2 >> 16));
}

// prediction row (4 samples in one word) + decoded residual -> reconstructed samples, clipped
__device__ __forceinline__ uint32_t recon4(uint32_t pred, int r0, int r1, int r2, int r3)
{
    const int a = clip255((int)(pred & 255u) + r0), b = clip255((int)((pred >> 8) & 255u) + r1);
    const int c = clip255((int)((pred >> 16) & 255u) + r2), e = clip255((int)(pred >> 24) + r3);
    return (uint32_t)a | ((uint32_t)b << 8) | ((uint32_t)c << 16) | ((uint32_t)e << 24);
}

__global__ __launch_bounds__(64) void k_tq(FrameParams P0)
{
    __builtin_amdgcn_s_setprio(2);   // short and on the way to the loop filter: ahead of another stream's motion search
    const FrameParams P = batch_view(P0, blockIdx.y);
    const int lane = threadIdx.x;
    const int nmb = P.mbw * P.band.rows, mb0 = P.band.row0 * P.mbw, end = mb0 + nmb;
    const int first = mb0 + 8 * xcd_mb_index(blockIdx.x, (nmb + 7) >> 3);
    const int cs = P.cw >> 1;
    const bool src_al = ((P.w | (int)(uintptr_t)P.src) & 3) == 0;   // source rows are dword aligned
    const int cwv = vreg(P.cw), wv = vreg(P.w), csv = vreg(cs);
    unsigned long long ymask[2];
    int ybound[2];   // bit bound of the lane's macroblock, luma part (in all 16 lanes of the macroblock)

    // ---- luma: two passes of 4 macroblocks, lane = (macroblock, blkIdx) ----
    {
        const TqConst K = tq_consts(P.qy);
#pragma unroll
        for (int p = 0; p < 2; p++) {
            const int blk = lane & 15, mbi = first + 4 * p + (lane >> 4);
            bool act = mbi < end;
            // MbInfo.type == MB_P16 and i16_mode == 0 (k_me sets i16_mode when nothing is left to code, type MB_I16 for the intra pass)
            if (act) act = *(const uint16_t*)((const uint8_t*)P.mb + (uint32_t)(mbi * 32 + 4)) == (uint16_t)MB_P16;
            int nz = 0, bb = 0;
            if (act) {
                const int my = P.mbdiv.row(mbi), mx = mbi - my * P.mbw;
                const int x = 16 * mx + 4 * blk_x(blk), y = 16 * my + 4 * blk_y(blk);
                uint32_t s4[4], p4[4];
                // one 32-bit offset per access from a wave-uniform base (global_load saddr form: no 64-bit address arithmetic)
                uint32_t po[4];
                po[0] = (uint32_t)__mul24(y, P.cw) + (uint32_t)x;
#pragma unroll
                for (int r = 1; r < 4; r++) po[r] = po[r - 1] + (uint32_t)cwv;
#pragma unroll
                for (int r = 0; r < 4; r++) p4[r] = *(const uint32_t*)(P.rec[0] + po[r]);
                if (src_al && x + 3 < P.w) {
                    const uint32_t omax = (uint32_t)__mul24(P.h - 1, P.w) + (uint32_t)x;   // rows below the picture repeat its last row
                    uint32_t o = (uint32_t)__mul24(y, P.w) + (uint32_t)x;
#pragma unroll
                    for (int r = 0; r < 4; r++) { s4[r] = *(const uint32_t*)(P.src + min(o, omax)); o += (uint32_t)wv; }
                } else {
#pragma unroll
                    for (int r = 0; r < 4; r++)
                        s4[r] = (uint32_t)src_px(P.src, P.w, P.h, x, y + r) | ((uint32_t)src_px(P.src, P.w, P.h, x + 1, y + r) << 8) |
                                ((uint32_t)src_px(P.src, P.w, P.h, x + 2, y + r) << 16) | ((uint32_t)src_px(P.src, P.w, P.h, x + 3, y + r) << 24);
                }
                int d[16];
#pragma unroll
                for (int r = 0; r < 4; r++)
#pragma unroll
                    for (int c = 0; c < 4; c++) d[4 * r + c] = (int)((s4[r] >> (8 * c)) & 255u) - (int)((p4[r] >> (8 * c)) & 255u);
                uint32_t lvp[8];
                nz = tq_block<false>(d, K, lvp, [](int) { return 0; });
                bb = blk_bits_bound_packed(lvp, nz);
                const uint32_t lo = (uint32_t)__mul24(mbi, LV_STRIDE * 2) + (uint32_t)((LV_LUMA + blk * 16) * 2);
                *(uint4*)((uint8_t*)P.levels + lo) = make_uint4(lvp[0], lvp[1], lvp[2], lvp[3]);
                *(uint4*)((uint8_t*)P.levels + lo + 16u) = make_uint4(lvp[4], lvp[5], lvp[6], lvp[7]);
#pragma unroll
                for (int r = 0; r < 4; r++) *(uint32_t*)(P.rec[0] + po[r]) = recon4(p4[r], d[4 * r], d[4 * r + 1], d[4 * r + 2], d[4 * r + 3]);
                *((uint8_t*)P.mb + (uint32_t)(mbi * 32 + 8 + blk)) = (uint8_t)nz;   // MbInfo.tc[blk]
            }
            ymask[p] = __ballot(nz != 0);
            ybound[p] = row_sum16_dpp(bb);
        }
    }

    // ---- chroma of the 8 macroblocks in one pass, lane = (macroblock, plane, block); the four blocks of a plane are a DPP quad ----
    {
        const TqConst K = tq_consts(P.qc);
        const int m8 = lane >> 3, pl = (lane >> 2) & 1, cb = lane & 3, mbi = first + m8;
        bool act = mbi < end;
        if (act) act = *(const uint16_t*)((const uint8_t*)P.mb + (uint32_t)(mbi * 32 + 4)) == (uint16_t)MB_P16;
        int cnz = 0, ldc = 0, bb = 0;
        if (act) {
            const int my = P.mbdiv.row(mbi), mx = mbi - my * P.mbw;
            const int x = 8 * mx + 4 * (cb & 1), y = 8 * my + 4 * (cb >> 1);
            uint8_t* const cplane = pl ? P.rec[2] : P.rec[1];
            uint32_t s4[4], p4[4], po[4];
            po[0] = (uint32_t)__mul24(y, cs) + (uint32_t)x;
#pragma unroll
            for (int r = 1; r < 4; r++) po[r] = po[r - 1] + (uint32_t)csv;
#pragma unroll
            for (int r = 0; r < 4; r++) { p4[r] = *(const uint32_t*)(cplane + po[r]); s4[r] = src_chroma4(P, pl, x, y + r); }
            int d[16];
#pragma unroll
            for (int r = 0; r < 4; r++)
#pragma unroll
                for (int c = 0; c < 4; c++) d[4 * r + c] = (int)((s4[r] >> (8 * c)) & 255u) - (int)((p4[r] >> (8 * c)) & 255u);
            uint32_t lvp[8];
            // 8.5.11: the four DC terms of the plane (one per lane of the quad) through the 2x2 Hadamard, quantised at q + 1 with
            // offset 2f, transformed back and scaled; lane cb keeps output cb of both transforms
            const int sg1 = (cb & 1) ? -1 : 1, sg2 = (cb & 2) ? -1 : 1;
            auto had2x2 = [&](int v) {
                const int a = __builtin_amdgcn_mov_dpp(v, 0x00, 0xf, 0xf, false), b = __builtin_amdgcn_mov_dpp(v, 0x55, 0xf, 0xf, false);
                const int c = __builtin_amdgcn_mov_dpp(v, 0xAA, 0xf, 0xf, false), e = __builtin_amdgcn_mov_dpp(v, 0xFF, 0xf, 0xf, false);
                return a + sg1 * b + sg2 * (c + sg1 * e);
            };
            cnz = tq_block<true>(d, K, lvp, [&](int w0) {
                const int fd = had2x2(w0);
                ldc = quant_signed(fd, K.mf[0], 2 * K.f, 2 * K.c + 1, K.q + 1);   // 2^(q+1) - 1 - 2 (2f) = 2 c + 1
                return (had2x2(ldc) * 16 * K.dq[0]) >> 5;
            });
            bb = blk_bits_bound_packed(lvp, cnz);
            const uint32_t lb = (uint32_t)__mul24(mbi, LV_STRIDE * 2), lo = lb + (uint32_t)((LV_CHROMA_AC + (pl * 4 + cb) * 16) * 2);
            *(uint4*)((uint8_t*)P.levels + lo) = make_uint4(lvp[0], lvp[1], lvp[2], lvp[3]);
            *(uint4*)((uint8_t*)P.levels + lo + 16u) = make_uint4(lvp[4], lvp[5], lvp[6], lvp[7]);
            *(int16_t*)((uint8_t*)P.levels + lb + (uint32_t)((LV_CHROMA_DC + pl * 4 + cb) * 2)) = (int16_t)ldc;
#pragma unroll
            for (int r = 0; r < 4; r++) *(uint32_t*)(cplane + po[r]) = recon4(p4[r], d[4 * r], d[4 * r + 1], d[4 * r + 2], d[4 * r + 3]);
            *((uint8_t*)P.mb + (uint32_t)(mbi * 32 + 24 + pl * 4 + cb)) = (uint8_t)cnz;   // MbInfo.tc[16 + plane * 4 + block]
        }
        const unsigned long long acm = __ballot(cnz != 0), dcm = __ballot(ldc != 0);
        // bit bound of the macroblock: its 8 chroma AC blocks + the two DC blocks (four lanes of a quad each) + the luma part
        unsigned long long pcm;
        {
            auto quad_red = [](int v, bool orop) {
                int t = __builtin_amdgcn_mov_dpp(v, 0xB1, 0xf, 0xf, false);
                v = orop ? (v | t) : (v + t);
                t = __builtin_amdgcn_mov_dpp(v, 0x4E, 0xf, 0xf, false);
                return orop ? (v | t) : (v + t);
            };
            const int adc = iabs(ldc), tcdc = quad_red(ldc != 0, false), h = pcm_smax((unsigned)quad_red(adc, true) | 1u) + 1;
            const int dcb = tcdc ? quad_red(adc ? max(min(adc, 27), h) + 1 : 0, false) + pcm_blk_tail(tcdc) : 6;
            const int csum = group_sum8_dpp(bb + (cb == 0 ? dcb : 0));
            const int y0 = __shfl(ybound[0], 16 * (m8 & 3)), y1 = __shfl(ybound[1], 16 * (m8 & 3));
            pcm = __ballot(act && (lane & 7) == 0 && MB_HEADER_BOUND + csum + (m8 < 4 ? y0 : y1) > MB_BITS_LIMIT);
        }
        if (act && (lane & 7) == 0) {
            const unsigned m16 = (unsigned)(ymask[m8 >> 2] >> (16 * (m8 & 3))) & 0xFFFFu;
            const int cbpl = ((m16 & 0x000Fu) ? 1 : 0) | ((m16 & 0x00F0u) ? 2 : 0) | ((m16 & 0x0F00u) ? 4 : 0) | ((m16 & 0xF000u) ? 8 : 0);
            const int cbpc = ((acm >> (8 * m8)) & 255ull) ? 2 : (((dcm >> (8 * m8)) & 255ull) ? 1 : 0);
            *((uint8_t*)P.mb + (uint32_t)(mbi * 32 + 7)) = (uint8_t)(cbpl | (cbpc << 4));   // MbInfo.cbp
        }
        if (pcm) {   // wave-uniform and rare: re-write those macroblocks as I_PCM, the whole wave per macroblock
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's earlier stores to the same samples / bytes have landed
            while (pcm) {
                const int pm = first + ((__ffsll((long long)pcm) - 1) >> 3);
                pcm &= pcm - 1;
                const int py = P.mbdiv.row(pm), px = pm - py * P.mbw;
                {
                    const int row = lane >> 2, xs = (lane & 3) * 4, gx = 16 * px + xs, gy = 16 * py + row;
                    *(uint32_t*)(P.rec[0] + (size_t)gy * P.cw + gx) = (uint32_t)src_px(P.src, P.w, P.h, gx, gy) | ((uint32_t)src_px(P.src, P.w, P.h, gx + 1, gy) << 8) |
                                                                     ((uint32_t)src_px(P.src, P.w, P.h, gx + 2, gy) << 16) | ((uint32_t)src_px(P.src, P.w, P.h, gx + 3, gy) << 24);
                }
                if (lane < 32) {
                    const int cp = lane >> 4, row = (lane >> 1) & 7, xs = (lane & 1) * 4;
                    *(uint32_t*)((cp ? P.rec[2] : P.rec[1]) + (size_t)(8 * py + row) * cs + 8 * px + xs) = src_chroma4(P, cp, 8 * px + xs, 8 * py + row);
                }
                if (lane < 6) ((uint32_t*)(P.mb + pm))[2 + lane] = 0x10101010u;                      // TotalCoeff 16 everywhere (9.2.1)
                if (lane == 6) *(uint2*)(P.mb + pm) = make_uint2(0u, (uint32_t)MB_IPCM | (0x2Fu << 24));   // no vector, type, coded_block_pattern 47
            }
            if (lane == 0) *P.anypcm = P.pic_serial;
        }
    }
}

// Motion vector differences and P_Skip: lane = macroblock.  Needs every macroblock's final vector (k_me) and
// coded_block_pattern (k_tq); writes mvd and MbInfo.type / i16_mode (k_me's "nothing to code" mark is cleared).
__global__ __launch_bounds__(64) void k_mvpred(FrameParams P0)
{
    __builtin_amdgcn_s_setprio(1);
    const FrameParams P = batch_view(P0, blockIdx.y);
    const int nmb = P.mbw * P.band.rows, mb0 = P.band.row0 * P.mbw;
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= nmb) return;
    const int mbi = mb0 + i, my = P.mbdiv.row(mbi), mx = mbi - my * P.mbw;
    const bool top = P.sl.has_top(my);
    const bool avA = mx > 0, avB = top, avC0 = top && mx + 1 < P.mbw, avD = mx > 0 && top;
    const MbInfo* base = P.mb + mbi;
    const uint2 self = *(const uint2*)base;
    if (mb_is_intra((int)(self.y & 255u))) return;   // (intra macroblocks carry no vector)
    const uint2 wA = *(const uint2*)(avA ? base - 1 : base);
    const uint2 wB = *(const uint2*)(avB ? base - P.mbw : base);
    const uint2 wC = *(const uint2*)(avC0 ? base - P.mbw + 1 : base);
    const uint2 wD = *(const uint2*)(avD ? base - P.mbw - 1 : base);
    auto unpack = [](const uint2 w, bool av, int& ref, Mv& mv) {
        ref = -1; mv.x = 0; mv.y = 0;
        if (av && !mb_is_intra((int)(w.y & 255u))) { ref = 0; mv.x = (int)(int16_t)(w.x & 0xFFFFu); mv.y = (int)(int16_t)(w.x >> 16); }
    };
    int rA, rB, rC;
    Mv A, B, C;
    unpack(wA, avA, rA, A);
    unpack(wB, avB, rB, B);
    bool aC = avC0;
    if (avC0) unpack(wC, true, rC, C);
    else { unpack(wD, avD, rC, C); aC = avD; }
    const bool zero_skip = !avA || !avB || (rA == 0 && A.x == 0 && A.y == 0) || (rB == 0 && B.x == 0 && B.y == 0);
    if (!avB && !aC && avA) { B = A; C = A; rB = rA; rC = rA; }
    Mv p;
    const int n = (rA == 0) + (rB == 0) + (rC == 0);
    if (n == 1) p = rA == 0 ? A : (rB == 0 ? B : C);
    else { p.x = med3(A.x, B.x, C.x); p.y = med3(A.y, B.y, C.y); }
    const Mv skip = zero_skip ? Mv{0, 0} : p;
    const int mvx = (int)(int16_t)(self.x & 0xFFFFu), mvy = (int)(int16_t)(self.x >> 16), cbp = (int)(self.y >> 24);
    const int type = (cbp == 0 && skip.x == mvx && skip.y == mvy) ? MB_PSKIP : MB_P16;
    *(uint16_t*)((uint8_t*)(P.mb + mbi) + 4) = (uint16_t)type;   // type, i16_mode = 0
    *(uint32_t*)(P.mvd + 2 * mbi) = (uint32_t)((mvx - p.x) & 0xFFFF) | ((uint32_t)(mvy - p.y) << 16);
}

}  // namespace h264
