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

// {type, i16_mode} of a macroblock k_tq / k_tq8 must code: an inter macroblock (16x16 or partitioned) whose i16_mode byte is
// still 0 (k_me sets 0x80 there when nothing is left to code, and type MB_I16 for the intra pass)
__device__ __forceinline__ bool mb_to_code(unsigned type_mode) { return type_mode == (unsigned)MB_P16 || (type_mode >= (unsigned)MB_P16X8 && type_mode <= (unsigned)MB_P8X8); }

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

// Chroma of 8 macroblocks first8 .. first8 + 7 in one pass, lane = (macroblock, plane, block); the four blocks of a plane are
// a DPP quad.  cbpl / ybnd: coded_block_pattern bits and bit bound of the luma of THIS lane's macroblock (from the caller's
// luma passes); t8: the luma went through the 8x8 transform (MbInfo.i16_mode of an inter macroblock = transform_size_8x8_flag).
// Finishes the macroblock: coded_block_pattern, and the I_PCM fallback when the bit bound passes the limit of A.3.1.
__device__ __forceinline__ void tq_chroma8(const FrameParams& P, const int first, const int end, const int lane, const int cs, const int csv,
                                           const int cbpl, const int ybnd, const bool t8)
{

        const TqConst K = tq_consts(P.qc);
        const int m8 = lane >> 3, pl = (lane >> 2) & 1, cb = lane & 3, mbi = first + m8;
        bool act = mbi < end;
        if (act) act = mb_to_code(*(const uint16_t*)((const uint8_t*)P.mb + (uint32_t)(mbi * 32 + 4)));
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
            pcm = __ballot(act && (lane & 7) == 0 && MB_HEADER_BOUND + csum + ybnd > MB_BITS_LIMIT);
        }
        if (act && (lane & 7) == 0) {
            const int cbpc = ((acm >> (8 * m8)) & 255ull) ? 2 : (((dcm >> (8 * m8)) & 255ull) ? 1 : 0);
            *((uint8_t*)P.mb + (uint32_t)(mbi * 32 + 7)) = (uint8_t)(cbpl | (cbpc << 4));   // MbInfo.cbp
            if (t8 && cbpl) *((uint8_t*)P.mb + (uint32_t)(mbi * 32 + 5)) = 1;                  // transform_size_8x8_flag
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

template <bool IND = false>
__global__ __launch_bounds__(64) void k_tq(FrameParams P0)
{
    __builtin_amdgcn_s_setprio(2);   // short and on the way to the loop filter: ahead of another stream's motion search
    const FrameParams P = batch_view<IND>(P0, blockIdx.y);
    const int lane = threadIdx.x;
    const int nmb = P.mbw * P.band.rows, mb0 = P.band.row0 * P.mbw, end = mb0 + nmb;
    const int first = mb0 + 8 * xcd_mb_index(blockIdx.x, (nmb + 7) >> 3);
    const int cs = P.cw >> 1;
    const bool src_al = ((P.w | (int)(uintptr_t)P.src) & 3) == 0;   // source rows are dword aligned
    const int cwv = vreg(P.cw), wv = vreg(P.w), csv = vreg(cs);
#ifdef MI355X_AB_TQ_4MB   // (the layout of rounds 1-3: 4 macroblocks x 16 blocks per pass, four 64-byte row segments per instruction)
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
            if (act) act = mb_to_code(*(const uint16_t*)((const uint8_t*)P.mb + (uint32_t)(mbi * 32 + 4)));
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

    // ---- chroma of the 8 macroblocks + coded_block_pattern + I_PCM fallback ----
    {
        const int m8 = lane >> 3;
        const unsigned m16 = (unsigned)(ymask[m8 >> 2] >> (16 * (m8 & 3))) & 0xFFFFu;
        const int cbpl = ((m16 & 0x000Fu) ? 1 : 0) | ((m16 & 0x00F0u) ? 2 : 0) | ((m16 & 0x0F00u) ? 4 : 0) | ((m16 & 0xF000u) ? 8 : 0);
        const int y0 = __shfl(ybound[0], 16 * (m8 & 3)), y1 = __shfl(ybound[1], 16 * (m8 & 3));
        tq_chroma8(P, first, end, lane, cs, csv, cbpl, m8 < 4 ? y0 : y1, false);
    }
#else
    // ---- luma: two passes over the wave's 8 macroblocks, lane = (macroblock, block row of the pass, block column).  Pass p takes
    // block rows 2 p and 2 p + 1: a load or store instruction then touches TWO picture rows of 128 contiguous bytes (8 macroblocks
    // x 16 samples) instead of four rows of 64 - whole cache lines; the kernel is bound by the shape of its accesses ----
    const int m8 = lane >> 3, bxl = lane & 3, by2 = (lane >> 2) & 1, mbi = first + m8;
    bool act = mbi < end;
    // MbInfo.type == MB_P16 and i16_mode == 0 (k_me sets i16_mode when nothing is left to code, type MB_I16 for the intra pass)
    if (act) act = mb_to_code(*(const uint16_t*)((const uint8_t*)P.mb + (uint32_t)(mbi * 32 + 4)));
    unsigned long long ymask[2];
    int ybnd = 0;   // bit bound of the lane's macroblock, luma part (in all 8 lanes of the macroblock)
    {
        const TqConst K = tq_consts(P.qy);
        const int my = P.mbdiv.row(act ? mbi : mb0), mx = (act ? mbi : mb0) - my * P.mbw;
#pragma unroll
        for (int p = 0; p < 2; p++) {
            const int byl = 2 * p + by2, blk = xy2blk(bxl, byl);
            int nz = 0, bb = 0;
            if (act) {
                const int x = 16 * mx + 4 * bxl, y = 16 * my + 4 * byl;
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
            ybnd += group_sum8_dpp(bb);
        }
    }

    // ---- chroma of the 8 macroblocks + coded_block_pattern + I_PCM fallback (the same lanes hold the same macroblock) ----
    {
        // lanes of the macroblock in a pass: bit = block row of the pass * 4 + block column; an 8x8 quadrant = columns 0,1 or 2,3 of both rows
        const unsigned t8 = (unsigned)(ymask[0] >> (8 * m8)) & 0xFFu, b8 = (unsigned)(ymask[1] >> (8 * m8)) & 0xFFu;
        const int cbpl = ((t8 & 0x33u) ? 1 : 0) | ((t8 & 0xCCu) ? 2 : 0) | ((b8 & 0x33u) ? 4 : 0) | ((b8 & 0xCCu) ? 8 : 0);
        tq_chroma8(P, first, end, lane, cs, csv, cbpl, ybnd, false);
    }
#endif
}


// ===========================================================================
// High profile: the luma residual of inter macroblocks through the 8x8 transform (8.5.13; transform_8x8_mode_flag = 1).
// lane = ONE 8x8 block with all 64 samples in registers (both passes of the transforms in-lane, as in k_tq); a wave codes
// SIXTEEN macroblocks: one luma pass (16 macroblocks x 4 blocks) and two chroma passes of 8 macroblocks (4x4 transform,
// tq_chroma8).  The 64 levels leave as the four interleaved 4x4 lists CAVLC wants (7.3.5.3.2: level i of list k = level
// 4 i + k of the 8x8 zig-zag scan), stored where the quadrant's 4x4 lists live: TotalCoeff, nC, the bit bound and the entropy
// coder need no special case.  Oracle: oracle/h264_enc.c encode_inter_mb (profile_idc 100).
// ===========================================================================
template <int S>
__device__ __forceinline__ void fdct8_line(int* d)   // elements d[0], d[S], .. d[7 S], in place (reference-model forward butterfly)
{
    const int s07 = d[0] + d[7 * S], s16 = d[S] + d[6 * S], s25 = d[2 * S] + d[5 * S], s34 = d[3 * S] + d[4 * S];
    const int a0 = s07 + s34, a1 = s16 + s25, a2 = s07 - s34, a3 = s16 - s25;
    const int d07 = d[0] - d[7 * S], d16 = d[S] - d[6 * S], d25 = d[2 * S] - d[5 * S], d34 = d[3 * S] - d[4 * S];
    const int a4 = d16 + d25 + (d07 + (d07 >> 1)), a5 = d07 - d34 - (d25 + (d25 >> 1));
    const int a6 = d07 + d34 - (d16 + (d16 >> 1)), a7 = d16 - d25 + (d34 + (d34 >> 1));
    d[0] = a0 + a1; d[S] = a4 + (a7 >> 2); d[2 * S] = a2 + (a3 >> 1); d[3 * S] = a5 + (a6 >> 2);
    d[4 * S] = a0 - a1; d[5 * S] = a6 - (a5 >> 2); d[6 * S] = (a2 >> 1) - a3; d[7 * S] = (a4 >> 2) - a7;
}
template <int S>
__device__ __forceinline__ void idct8_line(int* d)   // 8.5.13, one dimension, in place
{
    const int a0 = d[0] + d[4 * S], a2 = d[0] - d[4 * S], a4 = (d[2 * S] >> 1) - d[6 * S], a6 = d[2 * S] + (d[6 * S] >> 1);
    const int b0 = a0 + a6, b2 = a2 + a4, b4 = a2 - a4, b6 = a0 - a6;
    const int a1 = -d[3 * S] + d[5 * S] - d[7 * S] - (d[7 * S] >> 1), a3 = d[S] + d[7 * S] - d[3 * S] - (d[3 * S] >> 1);
    const int a5 = -d[S] + d[7 * S] + d[5 * S] + (d[5 * S] >> 1), a7 = d[3 * S] + d[5 * S] + d[S] + (d[S] >> 1);
    const int b1 = a1 + (a7 >> 2), b3 = a3 + (a5 >> 2), b5 = (a3 >> 2) - a5, b7 = a7 - (a1 >> 2);
    d[0] = b0 + b7; d[S] = b2 + b5; d[2 * S] = b4 + b3; d[3 * S] = b6 + b1;
    d[4 * S] = b6 - b1; d[5 * S] = b4 - b3; d[6 * S] = b2 - b5; d[7 * S] = b0 - b7;
}
__device__ constexpr int pos_class8(int pos)
{
    const int i = pos >> 3, j = pos & 7;
    return (i % 4 == 0 && j % 4 == 0) ? 0 : ((i & 1) && (j & 1)) ? 1 : (i % 4 == 2 && j % 4 == 2) ? 2
         : ((i % 4 == 0 && (j & 1)) || ((i & 1) && j % 4 == 0)) ? 3 : ((i % 4 == 0 && j % 4 == 2) || (i % 4 == 2 && j % 4 == 0)) ? 4 : 5;
}

template <bool IND = false>
__global__ __launch_bounds__(64) void k_tq8(FrameParams P0)
{
    __builtin_amdgcn_s_setprio(2);
    const FrameParams P = batch_view<IND>(P0, blockIdx.y);
    const int lane = threadIdx.x;
    const int nmb = P.mbw * P.band.rows, mb0 = P.band.row0 * P.mbw, end = mb0 + nmb;
    const int first = mb0 + 16 * xcd_mb_index(blockIdx.x, (nmb + 15) >> 4);
    const int cs = P.cw >> 1;
    const bool src_al = ((P.w | (int)(uintptr_t)P.src) & 3) == 0;
    const int cwv = vreg(P.cw), wv = vreg(P.w), csv = vreg(cs);
    unsigned long long ymask;
    int ybound;
    {
        constexpr int ZZ8[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6,  7,  14, 21, 28,
                                 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
        int mf[6], ls[6];
#pragma unroll
        for (int c = 0; c < 6; c++) { mf[c] = vreg(P.qy.mf8[c]); ls[c] = vreg(P.qy.ls8[c]); }
        const int q8 = P.qy.qbits + 1;                                   // 16 + qp / 6
        const int f8 = vreg((1 << q8) / 6), c8 = vreg((1 << q8) - 1 - 2 * ((1 << q8) / 6)), q8v = vreg(q8);
        const int qp6 = P.qy.qp / 6;
        const int b8 = lane & 3, mbi = first + (lane >> 2);
        bool act = mbi < end;
        if (act) act = mb_to_code(*(const uint16_t*)((const uint8_t*)P.mb + (uint32_t)(mbi * 32 + 4)));
        int nzany = 0, bb = 0;
        if (act) {
            const int my = P.mbdiv.row(mbi), mx = mbi - my * P.mbw;
            const int x = 16 * mx + 8 * (b8 & 1), y = 16 * my + 8 * (b8 >> 1);
            uint32_t po[8], p8[16], s8[16];
            po[0] = (uint32_t)__mul24(y, P.cw) + (uint32_t)x;
#pragma unroll
            for (int r = 1; r < 8; r++) po[r] = po[r - 1] + (uint32_t)cwv;
#pragma unroll
            for (int r = 0; r < 8; r++) { p8[2 * r] = *(const uint32_t*)(P.rec[0] + po[r]); p8[2 * r + 1] = *(const uint32_t*)(P.rec[0] + po[r] + 4u); }
            if (src_al && x + 7 < P.w) {
                const uint32_t omax = (uint32_t)__mul24(P.h - 1, P.w) + (uint32_t)x;
                uint32_t o = (uint32_t)__mul24(y, P.w) + (uint32_t)x;
#pragma unroll
                for (int r = 0; r < 8; r++) {
                    const uint32_t oo = min(o, omax);
                    s8[2 * r] = *(const uint32_t*)(P.src + oo); s8[2 * r + 1] = *(const uint32_t*)(P.src + oo + 4u);
                    o += (uint32_t)wv;
                }
            } else {
#pragma unroll
                for (int r = 0; r < 8; r++)
#pragma unroll
                    for (int hh = 0; hh < 2; hh++)
                        s8[2 * r + hh] = (uint32_t)src_px(P.src, P.w, P.h, x + 4 * hh, y + r) | ((uint32_t)src_px(P.src, P.w, P.h, x + 4 * hh + 1, y + r) << 8) |
                                         ((uint32_t)src_px(P.src, P.w, P.h, x + 4 * hh + 2, y + r) << 16) | ((uint32_t)src_px(P.src, P.w, P.h, x + 4 * hh + 3, y + r) << 24);
            }
            int d[64];
#pragma unroll
            for (int i = 0; i < 64; i++) d[i] = (int)((s8[i >> 2] >> (8 * (i & 3))) & 255u) - (int)((p8[i >> 2] >> (8 * (i & 3))) & 255u);
#pragma unroll
            for (int r = 0; r < 8; r++) fdct8_line<1>(d + 8 * r);
#pragma unroll
            for (int c = 0; c < 8; c++) fdct8_line<8>(d + c);
            // quantise (signed form, see quant_signed) and scale (8.5.13) in place; l[] keeps the levels for the lists
            int l[64];
#pragma unroll
            for (int i = 0; i < 64; i++) {
                constexpr int dummy = 0; (void)dummy;
                const int cls = pos_class8(i);
                const int w = d[i], sg = w >> 31;
                l[i] = (w * mf[cls] + (f8 + (sg & c8))) >> q8v;
                const int t = l[i] * ls[cls];
                d[i] = qp6 >= 6 ? t << (qp6 - 6) : (t + (1 << (5 - qp6))) >> (6 - qp6);
            }
            uint32_t lvp[32];
            int tcs[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                uint32_t cnt2 = 0;
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const uint32_t pk = ((uint32_t)l[ZZ8[4 * (2 * j) + k]] & 0xFFFFu) | ((uint32_t)l[ZZ8[4 * (2 * j + 1) + k]] << 16);
                    lvp[8 * k + j] = pk;
                    const uint32_t t = pk | ((pk & 0x7FFF7FFFu) + 0x7FFF7FFFu);
                    cnt2 += (t >> 15) & 0x00010001u;
                }
                tcs[k] = (int)((cnt2 & 0xFFFFu) + (cnt2 >> 16));
                bb += blk_bits_bound_packed(lvp + 8 * k, tcs[k]);
                nzany |= tcs[k];
            }
            const uint32_t lo = (uint32_t)__mul24(mbi, LV_STRIDE * 2) + (uint32_t)((LV_LUMA + b8 * 64) * 2);
#pragma unroll
            for (int j = 0; j < 8; j++) *(uint4*)((uint8_t*)P.levels + lo + 16u * j) = make_uint4(lvp[4 * j], lvp[4 * j + 1], lvp[4 * j + 2], lvp[4 * j + 3]);
            *(uint32_t*)((uint8_t*)P.mb + (uint32_t)(mbi * 32 + 8 + 4 * b8)) = (uint32_t)tcs[0] | ((uint32_t)tcs[1] << 8) | ((uint32_t)tcs[2] << 16) | ((uint32_t)tcs[3] << 24);   // MbInfo.tc[4 b8 ..]
            if (nzany) {   // (an 8x8 block without levels reconstructs to the prediction: nothing to store)
#pragma unroll
                for (int r = 0; r < 8; r++) idct8_line<1>(d + 8 * r);
#pragma unroll
                for (int c = 0; c < 8; c++) idct8_line<8>(d + c);
#pragma unroll
                for (int r = 0; r < 8; r++) {
                    int* e = d + 8 * r;
                    *(uint32_t*)(P.rec[0] + po[r]) = recon4(p8[2 * r], (e[0] + 32) >> 6, (e[1] + 32) >> 6, (e[2] + 32) >> 6, (e[3] + 32) >> 6);
                    *(uint32_t*)(P.rec[0] + po[r] + 4u) = recon4(p8[2 * r + 1], (e[4] + 32) >> 6, (e[5] + 32) >> 6, (e[6] + 32) >> 6, (e[7] + 32) >> 6);
                }
            }
        }
        ymask = __ballot(nzany != 0);
        int t = bb;   // bit bound of the macroblock's luma = sum over its quad
        t += __builtin_amdgcn_mov_dpp(t, 0xB1, 0xf, 0xf, false);
        t += __builtin_amdgcn_mov_dpp(t, 0x4E, 0xf, 0xf, false);
        ybound = t;
    }
#pragma unroll 1
    for (int c = 0; c < 2; c++) {
        const int m = 8 * c + (lane >> 3);
        const int cbpl = (int)((ymask >> (4 * m)) & 15ull);
        const int yb = __shfl(ybound, 4 * m);
        tq_chroma8(P, first + 8 * c, end, lane, cs, csv, cbpl, yb, true);
    }
}

// Motion vector differences and P_Skip: lane = macroblock.  Needs every macroblock's final vectors (k_me) and
// coded_block_pattern (k_tq); writes mvd and MbInfo.type / i16_mode (k_me's "nothing to code" mark is cleared).
// 8.4.1.3 with vectors kept per 8x8 quadrant (mvq), the granularity of the smallest partition: a neighbouring partition
// is the quadrant (qx, qy) of its macroblock; A left of the partition's top-left sample, B above it, C above-right of its
// top-right sample or, when that is outside or later in decoding order, D above-left (6.4.11.7).
template <bool IND = false>
__global__ __launch_bounds__(64) void k_mvpred(FrameParams P0)
{
    __builtin_amdgcn_s_setprio(AB_PRIO_EC);
    const FrameParams P = batch_view<IND>(P0, blockIdx.y);
    const int nmb = P.mbw * P.band.rows, mb0 = P.band.row0 * P.mbw;
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= nmb) return;
    const int mbi = mb0 + i, my = P.mbdiv.row(mbi), mx = mbi - my * P.mbw;
    const bool top = P.sl.has_top(my);
    const bool avL = mx > 0, avT = top, avTR = top && mx + 1 < P.mbw, avTL = mx > 0 && top;
    const MbInfo* base = P.mb + mbi;
    const uint2 self = *(const uint2*)base;
    const int stype = (int)(self.y & 255u);
    if (mb_is_intra(stype)) return;   // (intra macroblocks carry no vector)
    const int cref = (int)((self.y >> 16) & 255u);
    // {type, ref} of the four neighbouring macroblocks and the quadrant vectors that can be asked for
    struct Nb { int ref; uint32_t v[4]; };   // ref -1: not available or intra; v[q] = vector of quadrant q (x | y << 16)
    auto load = [&](bool av, int d, Nb& n) {
        n.ref = -1; n.v[0] = n.v[1] = n.v[2] = n.v[3] = 0u;
        if (av) {
            const uint32_t w = *(const uint32_t*)((const uint8_t*)(base + d) + 4);
            if (!mb_is_intra((int)(w & 255u))) {
                n.ref = (int)((w >> 16) & 255u);
                const uint4 q = *(const uint4*)(P.mvq + (size_t)(mbi + d) * 8);
                n.v[0] = q.x; n.v[1] = q.y; n.v[2] = q.z; n.v[3] = q.w;
            }
        }
    };
    Nb L, T, TR, TL, S;
    load(avL, -1, L);
    load(avT, -P.mbw, T);
    load(avTR, -P.mbw + 1, TR);
    load(avTL, -P.mbw - 1, TL);
    load(true, 0, S);
    struct Cand { bool av; int ref; Mv mv; };
    auto from = [](const Nb& n, bool av, int q) {
        Cand c;
        c.av = av; c.ref = av ? n.ref : -1;
        const uint32_t v = q == 0 ? n.v[0] : (q == 1 ? n.v[1] : (q == 2 ? n.v[2] : n.v[3]));
        c.mv.x = c.ref >= 0 ? (int)(int16_t)(v & 0xFFFFu) : 0; c.mv.y = c.ref >= 0 ? (int)(int16_t)(v >> 16) : 0;
        return c;
    };
    // predictor of the partition covering quadrants x0 .. x0 + w - 1, y0 .. y0 + h - 1; skip (16x16 geometry) = the P_Skip vector
    auto predict = [&](int x0, int y0, int w, int h, Mv* skip) {
        const Cand A = x0 == 0 ? from(L, avL, 2 * y0 + 1) : from(S, true, 2 * y0);
        const Cand B = y0 == 0 ? from(T, avT, 2 + x0) : from(S, true, x0);
        Cand C;
        if (y0 == 0) C = x0 + w <= 1 ? from(T, avT, 2 + x0 + w) : from(TR, avTR, 2);
        else if (x0 + w <= 1) C = from(S, true, x0 + w);        // the quadrant above-right, coded before this one
        else { C.av = false; C.ref = -1; C.mv = Mv{0, 0}; }    // in the macroblock to the right: not yet coded
        if (!C.av) {
            if (x0 == 0 && y0 == 0) C = from(TL, avTL, 3);
            else if (y0 == 0) C = from(T, avT, 2);
            else if (x0 == 0) C = from(L, avL, 1);
            else C = from(S, true, 0);
        }
        Cand a = A, b = B, c = C;
        const bool zero_skip = !a.av || !b.av || (a.ref == 0 && a.mv.x == 0 && a.mv.y == 0) || (b.ref == 0 && b.mv.x == 0 && b.mv.y == 0);
        if (w == 2 && h == 1) {          // 16x8: upper partition B, lower partition A, when that neighbour uses the same picture
            if (y0 == 0 && b.ref == cref) return b.mv;
            if (y0 == 1 && a.ref == cref) return a.mv;
        } else if (w == 1 && h == 2) {   // 8x16: left A, right C
            if (x0 == 0 && a.ref == cref) return a.mv;
            if (x0 == 1 && c.ref == cref) return c.mv;
        }
        if (!b.av && !c.av && a.av) { b = a; c = a; }
        auto pred_for = [&](int ref) {   // 8.4.1.3.1
            Mv p;
            const int n = (a.ref == ref) + (b.ref == ref) + (c.ref == ref);
            if (n == 1) p = a.ref == ref ? a.mv : (b.ref == ref ? b.mv : c.mv);
            else { p.x = med3(a.mv.x, b.mv.x, c.mv.x); p.y = med3(a.mv.y, b.mv.y, c.mv.y); }
            return p;
        };
        const Mv p = pred_for(cref);
        if (skip) *skip = zero_skip ? Mv{0, 0} : (cref == 0 ? p : pred_for(0));   // 8.4.1.1: P_Skip predicts for ref_idx 0
        return p;
    };
    uint32_t d[4] = {0u, 0u, 0u, 0u};
    auto diff = [](uint32_t v, Mv p) { return (uint32_t)(((int)(int16_t)(v & 0xFFFFu) - p.x) & 0xFFFF) | ((uint32_t)((int)(int16_t)(v >> 16) - p.y) << 16); };
    int type = stype;
    if (stype == MB_P16X8) { d[0] = diff(S.v[0], predict(0, 0, 2, 1, nullptr)); d[1] = diff(S.v[2], predict(0, 1, 2, 1, nullptr)); }
    else if (stype == MB_P8X16) { d[0] = diff(S.v[0], predict(0, 0, 1, 2, nullptr)); d[1] = diff(S.v[1], predict(1, 0, 1, 2, nullptr)); }
    else if (stype == MB_P8X8) {
        d[0] = diff(S.v[0], predict(0, 0, 1, 1, nullptr)); d[1] = diff(S.v[1], predict(1, 0, 1, 1, nullptr));
        d[2] = diff(S.v[2], predict(0, 1, 1, 1, nullptr)); d[3] = diff(S.v[3], predict(1, 1, 1, 1, nullptr));
    } else {
        Mv skip;
        const Mv p = predict(0, 0, 2, 2, &skip);
        const int mvx = (int)(int16_t)(self.x & 0xFFFFu), mvy = (int)(int16_t)(self.x >> 16), cbp = (int)(self.y >> 24);
        type = (cbp == 0 && cref == 0 && skip.x == mvx && skip.y == mvy) ? MB_PSKIP : MB_P16;
        d[0] = diff(self.x, p);
    }
    // type; i16_mode keeps transform_size_8x8_flag (bit 0, k_tq8), k_me's "nothing to code" mark (0x80) goes
    *(uint16_t*)((uint8_t*)(P.mb + mbi) + 4) = (uint16_t)((unsigned)type | (type != MB_PSKIP ? ((self.y >> 8) & 1u) << 8 : 0u));
    *(uint4*)(P.mvd + 8 * (size_t)mbi) = make_uint4(d[0], d[1], d[2], d[3]);
}

}  // namespace h264
