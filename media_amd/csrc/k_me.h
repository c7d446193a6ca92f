// media_amd/csrc/k_me.h -- motion search, one wavefront per macroblock.
//
// SURVEY.md 8a row a6.1 (the SAD/SATD block-matching part of
// ISVCEncoder::EncodeFrame, /root/reference/video_codec/VideoEncoderOpenH264.cpp:344).
//
// Per macroblock, entirely out of LDS after one coalesced window load:
//   1. zero-motion test: does (src - ref) quantise to nothing?  -> mv = 0, done
//   (rate term of 2 and 3: lambda * bits(mv - pmv), pmv = this macroblock's vector in the previous picture)
//   2. full search dx,dy in [-16,15]: lane = (dx, half of dy range); each lane
//      keeps 16 SAD accumulators (v_sad_u8, four pixels per instruction) and
//      walks 31 window rows, every row feeding up to 16 candidates; the source
//      macroblock sits in SGPRs
//   3. half- then quarter-pel refinement on SATD over half-sample planes built
//      once in LDS (18x18 grid of G/b/h/j, 8.4.2.2.1)
//   4. the winning prediction (luma from those planes, chroma by 8.4.2.2.2) is written into the
//      reconstruction planes; k_tq.h codes the residual against it in place
// Decisions use only the previous picture and this macroblock's source, so the
// kernel is one launch over all macroblocks.
#pragma once
#include "dev_common.h"
#include "mc_filters.h"   // lds_ld4 / avg4 and the packed half-sample filter helpers

namespace h264 {

enum { ME_R = 16, ME_AP = 4, ME_WS = 16 + 2 * ME_R + 2 * ME_AP, ME_WDW = ME_WS / 4, ME_GS = 18, ME_GP = 20, ME_PLS = ME_GS * ME_GP };

// two-tap description of every quarter-sample position: pred = (T0 + T1 + 1) >> 1
// with Tk read from plane pk at grid offset (dxk, dyk); planes 0 G, 1 b, 2 h, 3 j
__device__ __forceinline__ void qpel_taps(int fx, int fy, int& o0, int& o1)
{
    const int PL = ME_PLS;
    const int G = 0, B = PL, H = 2 * PL, J = 3 * PL, R = 1, D = ME_GP;
    switch (fy * 4 + fx) {
        case 0: o0 = G; o1 = G; break;
        case 1: o0 = G; o1 = B; break;
        case 2: o0 = B; o1 = B; break;
        case 3: o0 = G + R; o1 = B; break;
        case 4: o0 = G; o1 = H; break;
        case 5: o0 = B; o1 = H; break;
        case 6: o0 = B; o1 = J; break;
        case 7: o0 = B; o1 = H + R; break;
        case 8: o0 = H; o1 = H; break;
        case 9: o0 = H; o1 = J; break;
        case 10: o0 = J; o1 = J; break;
        case 11: o0 = H + R; o1 = J; break;
        case 12: o0 = G + D; o1 = H; break;
        case 13: o0 = B + D; o1 = H; break;
        case 14: o0 = B + D; o1 = J; break;
        default: o0 = B + D; o1 = H + R; break;
    }
}


// four chroma prediction samples (8.4.2.2.2) of plane cp (pitch cs2, chh rows): integer position (x0, y0), eighth-sample
// fraction (fx, fy); samples clamped at the picture edge exactly as motion compensation does.  Inside the picture the two
// rows are read as aligned dwords and realigned (v_alignbyte), at the edge sample by sample.
__device__ __forceinline__ uint32_t chroma_pred4(const uint8_t* cp, int cs2, int chh, int x0, int y0, int fx, int fy)
{
    const uint8_t* r0 = cp + (size_t)clip3(0, chh - 1, y0) * cs2;
    const uint8_t* r1 = cp + (size_t)clip3(0, chh - 1, y0 + 1) * cs2;
    uint32_t A, B, C, D;
    if (x0 >= 0 && (x0 & ~3) + 8 <= cs2) {
        const int xa = x0 & ~3, sh = x0 & 3;
        const uint32_t a0 = *(const uint32_t*)(r0 + xa), a1 = *(const uint32_t*)(r0 + xa + 4);
        const uint32_t c0 = *(const uint32_t*)(r1 + xa), c1 = *(const uint32_t*)(r1 + xa + 4);
        A = __builtin_amdgcn_alignbyte(a1, a0, sh); B = sh == 3 ? a1 : __builtin_amdgcn_alignbyte(a1, a0, sh + 1);
        C = __builtin_amdgcn_alignbyte(c1, c0, sh); D = sh == 3 ? c1 : __builtin_amdgcn_alignbyte(c1, c0, sh + 1);
    } else {
        A = B = C = D = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int xa = clip3(0, cs2 - 1, x0 + k), xb = clip3(0, cs2 - 1, x0 + k + 1);
            A |= (uint32_t)r0[xa] << (8 * k); B |= (uint32_t)r0[xb] << (8 * k);
            C |= (uint32_t)r1[xa] << (8 * k); D |= (uint32_t)r1[xb] << (8 * k);
        }
    }
    if ((fx | fy) == 0) return A;
    const int w00 = (8 - fx) * (8 - fy), w10 = fx * (8 - fy), w01 = (8 - fx) * fy, w11 = fx * fy;
    uint32_t o = 0;
#pragma unroll
    for (int k = 0; k < 4; k++)
        o |= (uint32_t)((w00 * byte_of(A, k) + w10 * byte_of(B, k) + w01 * byte_of(C, k) + w11 * byte_of(D, k) + 32) >> 6) << (8 * k);
    return o;
}

#ifndef ME_WAVES_MIN
#define ME_WAVES_MIN 6
#endif
template <bool IND = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(ME_WAVES_MIN, 8))) void k_me(FrameParams P0)
{
    if (AB_PRIO_ME) __builtin_amdgcn_s_setprio(AB_PRIO_ME);
    const FrameParams P = batch_view<IND>(P0, blockIdx.y);
    const int lane = threadIdx.x;
    const int mbi = P.band.row0 * P.mbw + xcd_mb_index(blockIdx.x, P.mbw * P.band.rows), my = P.mbdiv.row(mbi), mx = mbi - my * P.mbw;
    const int bx = 16 * mx, by = 16 * my;

    __shared__ __attribute__((aligned(16))) uint32_t s_win[ME_WS * ME_WDW];  // 56 x 56 bytes
    __shared__ __attribute__((aligned(16))) uint8_t s_src[256];
    __shared__ __attribute__((aligned(16))) uint8_t s_srcc[128];
    __shared__ __attribute__((aligned(16))) uint32_t s_ytab[32];
    __shared__ __attribute__((aligned(16))) uint8_t s_refc[128];
    // one region, two lives: first the survivor list of the integer search (up to ME_TASKCAP keys), then the
    // half-sample planes (s_b1: unclipped horizontal sums, pitch 20 int16; s_pl: planes G,b,h,j, 18 rows, pitch 20)
    enum { ME_B1_BYTES = (ME_GS + 5) * ME_GP * 2, ME_PL_BYTES = 4 * ME_PLS + 16 };
    enum { ME_TASKCAP = 760 };   // 3 040 B: with the rest of the LDS, 24 workgroups (6 waves per SIMD) fit a CU
    __shared__ __attribute__((aligned(16))) uint8_t s_scr[4 * ME_TASKCAP > ME_B1_BYTES + ME_PL_BYTES ? 4 * ME_TASKCAP : ME_B1_BYTES + ME_PL_BYTES];
    __shared__ unsigned s_ntask;
    uint32_t* const s_task = (uint32_t*)s_scr;
    int16_t* const s_b1 = (int16_t*)s_scr;
    uint8_t* const s_pl = s_scr + ME_B1_BYTES;

    // the vector this macroblock had in the previous picture (0 after an IDR): stand-in for the motion vector predictor
    // in the rate term of the search; final before the launch, so every macroblock stays independent
    // (with several reference pictures the first launch parks it in P.pmv: MbInfo is rewritten by then)
    const int pmw = __builtin_amdgcn_readfirstlane(P.rf == 0 ? *(const int*)(P.mb + mbi) : P.pmv[mbi]);
    if (P.rf == 0 && P.rf_last > 0 && lane == 0) P.pmv[mbi] = pmw;
    const int pmx = (int)(int16_t)(pmw & 0xFFFF), pmy = pmw >> 16;
    load_src_mb(P, mx, my, s_src, s_srcc, lane);
    // Several reference pictures (config.refs, BASELINE.json configs[4]: 3): ONE LAUNCH PER REFERENCE PICTURE, P.ref = the
    // planes of ref_idx_l0 = P.rf.  A launch after the first leaves a macroblock alone unless its motion cost + lambda *
    // bits(ref_idx_l0) beats what the earlier launches left in me_total (the lower index keeps a tie); the last launch makes
    // the intra decision on the overall best.  (A loop over the pictures inside one launch spilled 370 registers.)
    const int rf = P.rf;
    unsigned prev_total = 0xFFFFFFFFu;
    if (rf > 0) {
        prev_total = (unsigned)__builtin_amdgcn_readfirstlane((int)P.me_total[mbi]);
        if (prev_total == 0u) return;   // a "nothing left to code" test hit on ref_idx 0: settled
    }
    const uint8_t* const RY = P.ref[0];
    const uint8_t* const RU = P.ref[1];
    const uint8_t* const RV = P.ref[2];
    // reference window, clamped at the picture edge (unrestricted motion vectors); all requests of a lane are
    // issued before the first is consumed (one memory latency).  Macroblocks whose window (widened to 64 B
    // rows) lies inside the picture use a fixed pattern: lane = (dword column 0..15, row group 0..3), 14 rows
    // each, one address increment per request.
    {
        const int wx0 = bx - ME_R - ME_AP, wy0 = by - ME_R - ME_AP;
        const bool inside = wx0 >= 0 && wx0 + 64 <= P.cw && wy0 >= 0 && wy0 + ME_WS <= P.ch;
        if (inside) {
            const int c = lane & 15, rg = lane >> 4;
            const uint8_t* rp = RY + (size_t)(wy0 + rg) * P.cw + wx0 + 4 * c;
            const size_t step = (size_t)4 * P.cw;
            uint32_t v[14];
#pragma unroll
            for (int t = 0; t < 14; t++) v[t] = *(const uint32_t*)(rp + t * step);
            if (c < ME_WDW)
#pragma unroll
                for (int t = 0; t < 14; t++) s_win[(rg + 4 * t) * ME_WDW + c] = v[t];
        } else {
            const bool interior = wx0 >= 0 && bx + 16 + ME_R + ME_AP <= P.cw;
            uint32_t v[13];
#pragma unroll
            for (int t = 0; t < 13; t++) {
                const int i = lane + 64 * t;
                const int row = i / ME_WDW, dw = i - row * ME_WDW;
                const int gy = clip3(0, P.ch - 1, wy0 + row);
                const int gx = wx0 + dw * 4;
                const uint8_t* rp = RY + (size_t)gy * P.cw;
                v[t] = 0;
                if (i < ME_WS * ME_WDW) {
                    if (interior) v[t] = *(const uint32_t*)(rp + gx);
                    else {
#pragma unroll
                        for (int k = 0; k < 4; k++) v[t] |= (uint32_t)rp[clip3(0, P.cw - 1, gx + k)] << (8 * k);
                    }
                }
            }
#pragma unroll
            for (int t = 0; t < 13; t++)
                if (lane + 64 * t < ME_WS * ME_WDW) s_win[lane + 64 * t] = v[t];
        }
    }
    if (rf == 0 && lane < 32) {  // co-located chroma (the zero tests)
        const int pl = lane >> 4, row = (lane >> 1) & 7, xs = (lane & 1) * 4;
        *(uint32_t*)(s_refc + pl * 64 + row * 8 + xs) =
            *(const uint32_t*)((pl ? P.ref[2] : P.ref[1]) + (size_t)(8 * my + row) * (P.cw / 2) + 8 * mx + xs);
    }
    __syncthreads();
    const uint8_t* winb = (const uint8_t*)s_win;

    // ---- 1. "nothing left to code" tests: does the residual against a given prediction quantise to nothing?
    // lane = (4x4 block, row); luma rides in the low and chroma (lanes 0..31) in the high 16 bits of every register,
    // so one packed forward transform (row pass in the lane, column pass over the DPP quad) serves both.  Tried at
    // the zero vector (static content) and, when that fails, at the macroblock's previous-picture vector rounded to
    // integer samples, if non-zero (scrolling content); a hit fixes the vector and ends the search. ----
    if (rf == 0) {
        typedef unsigned short pk16 __attribute__((ext_vector_type(2)));
        const int r = lane & 3, b4 = lane >> 2;
        const int cplz = lane >> 4, cyz = ((lane >> 3) & 1) * 4 + r, cxz = ((lane >> 2) & 1) * 4;   // chroma lane (lanes < 32)
        const uint32_t sy = *(const uint32_t*)(s_src + ((b4 >> 2) * 4 + r) * 16 + (b4 & 3) * 4);
        const uint32_t sc = lane < 32 ? *(const uint32_t*)(s_srcc + cplz * 64 + cyz * 8 + cxz) : 0u;
        const uint32_t one = 0x00010001u;
        const pk16 sA = __builtin_bit_cast(pk16, (r & 1) ? 0xFFFFFFFFu : one);                          // +-1
        const pk16 mA = __builtin_bit_cast(pk16, r == 1 ? 2u * one : one);
        const pk16 mB = __builtin_bit_cast(pk16, r < 2 ? one : (r == 2 ? 0xFFFFFFFFu : 0xFFFEFFFEu));    // 1 1 -1 -2
        // thresholds - 1 per position class (0 even/even, 1 odd/odd, 2 mixed), luma | chroma << 16
        const uint32_t t0 = (uint32_t)(P.qy.thr_inter[0] - 1) | ((uint32_t)(P.qc.thr_inter[0] - 1) << 16);
        const uint32_t t1 = (uint32_t)(P.qy.thr_inter[1] - 1) | ((uint32_t)(P.qc.thr_inter[1] - 1) << 16);
        const uint32_t t2 = (uint32_t)(P.qy.thr_inter[2] - 1) | ((uint32_t)(P.qc.thr_inter[2] - 1) << 16);
        const pk16 th_even = __builtin_bit_cast(pk16, (r & 1) ? t2 : t0), th_odd = __builtin_bit_cast(pk16, (r & 1) ? t1 : t2);
        // ry / rc: this lane's four predicted luma / chroma samples (rc = 0 in lanes >= 32)
        auto quantises_to_nothing = [&](uint32_t ry, uint32_t rc) -> bool {
            pk16 d[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t sel = 0x0c040c00u + 0x00010001u * k;   // byte k of the low operand -> bits 0..7, of the high -> 16..23
                d[k] = __builtin_bit_cast(pk16, __builtin_amdgcn_perm(sc, sy, sel)) - __builtin_bit_cast(pk16, __builtin_amdgcn_perm(rc, ry, sel));
            }
            {
                const pk16 s0 = d[0] + d[3], s1 = d[1] + d[2], d0 = d[0] - d[3], d1 = d[1] - d[2];
                d[0] = s0 + s1; d[1] = d0 + d0 + d1; d[2] = s0 - s1; d[3] = d0 - d1 - d1;
            }
            uint32_t over = 0;
            int dc = 0;
#pragma unroll
            for (int c = 0; c < 4; c++) {
                // a residual that does not vanish shows in the first coefficient column (the DC column) nearly always:
                // ask once after it and spare the other three
                if (c == 1 && __ballot(over != 0) != 0ull) return false;
                const pk16 q0 = __builtin_bit_cast(pk16, quad_bcast<0>(__builtin_bit_cast(int, d[c])));
                const pk16 q1 = __builtin_bit_cast(pk16, quad_bcast<1>(__builtin_bit_cast(int, d[c])));
                const pk16 q2 = __builtin_bit_cast(pk16, quad_bcast<2>(__builtin_bit_cast(int, d[c])));
                const pk16 q3 = __builtin_bit_cast(pk16, quad_bcast<3>(__builtin_bit_cast(int, d[c])));
                const pk16 A = q0 + sA * q3, B = q1 + sA * q2;
                const pk16 w = mA * A + mB * B;                        // row r of the 4x4 core transform, column c
                if (c == 0) dc = (int)(short)w.y;                     // chroma DC of the block where r == 0
                typedef short spk16 __attribute__((ext_vector_type(2)));
                const spk16 ws = __builtin_bit_cast(spk16, w);
                const pk16 aw = __builtin_bit_cast(pk16, __builtin_elementwise_max(ws, -ws));
                uint32_t ov = __builtin_bit_cast(uint32_t, __builtin_elementwise_sub_sat(aw, (c & 1) ? th_odd : th_even));   // != 0 <=> |w| >= thr
                if (c == 0 && r == 0) ov &= 0xFFFFu;                   // the chroma DC goes through the 2x2 Hadamard instead
                over |= ov;
            }
            // chroma DC: block DCs sit in lanes plane * 16 + block * 4
            bool dcnz = false;
#pragma unroll
            for (int pl = 0; pl < 2; pl++) {
                const int d0 = __builtin_amdgcn_readlane(dc, pl * 16), d1 = __builtin_amdgcn_readlane(dc, pl * 16 + 4);
                const int d2 = __builtin_amdgcn_readlane(dc, pl * 16 + 8), d3 = __builtin_amdgcn_readlane(dc, pl * 16 + 12);
                const int t = P.qc.thr_dc_inter;
                dcnz |= iabs(d0 + d1 + d2 + d3) >= t || iabs(d0 - d1 + d2 - d3) >= t || iabs(d0 + d1 - d2 - d3) >= t || iabs(d0 - d1 - d2 + d3) >= t;
            }
            return !dcnz && __ballot(over != 0) == 0ull;
        };
        // Exact shortcut: if every luma coefficient stays below its threshold t_ij, then (the core transform's rows are
        // orthogonal with squared norms 4, 10, 4, 10) the residual energy of a 4x4 block is below E = sum t_ij^2 / (n_i n_j),
        // and by Cauchy-Schwarz the macroblock's SAD below 64 sqrt(E) = P.sad_nz.  A SAD at or above that (a moving
        // macroblock tested at the zero vector) needs no transform to be turned down.
        auto luma_sad = [&](uint32_t ry) -> unsigned {
            const int s = row_sum16_dpp((int)__builtin_amdgcn_sad_u8(sy, ry, 0u));
            return (unsigned)(__builtin_amdgcn_readlane(s, 0) + __builtin_amdgcn_readlane(s, 16) + __builtin_amdgcn_readlane(s, 32) + __builtin_amdgcn_readlane(s, 48));
        };
        // the vector is final and nothing is left to code: the prediction IS the reconstruction.  Every lane stores its
        // four luma (lanes < 32: and chroma) samples; lane 0 writes the whole side info (k_tq skips this macroblock)
        auto settle = [&](int vx, int vy, uint32_t ry, uint32_t rc) {
            *(uint32_t*)(P.rec[0] + (size_t)(by + (b4 >> 2) * 4 + r) * P.cw + bx + (b4 & 3) * 4) = ry;
            if (lane < 32) *(uint32_t*)((cplz ? P.rec[2] : P.rec[1]) + (size_t)(8 * my + cyz) * (P.cw / 2) + 8 * mx + cxz) = rc;
            if (lane == 0) {
                uint4* m = (uint4*)(P.mb + mbi);
                m[0] = make_uint4(((uint32_t)vx & 0xFFFFu) | ((uint32_t)vy << 16), (uint32_t)MB_P16 | (0x80u << 8), 0u, 0u);   // i16_mode = 0x80: mark for k_tq / k_mvpred
                m[1] = make_uint4(0u, 0u, 0u, 0u);
                const uint32_t v = ((uint32_t)vx & 0xFFFFu) | ((uint32_t)vy << 16);
                *(uint4*)(P.mvq + (size_t)mbi * 8) = make_uint4(v, v, v, v);
                P.me_cost[mbi] = 0;
                P.me_total[mbi] = 0u;   // settled: later reference pictures are not searched
            }
        };
        {   // the zero vector: co-located samples
            const uint32_t ry = s_win[(ME_R + ME_AP + (b4 >> 2) * 4 + r) * ME_WDW + (ME_R + ME_AP) / 4 + (b4 & 3)];
            const uint32_t rc = lane < 32 ? *(const uint32_t*)(s_refc + cplz * 64 + cyz * 8 + cxz) : 0u;
            if (luma_sad(ry) < (unsigned)P.sad_nz && quantises_to_nothing(ry, rc)) { settle(0, 0, ry, rc); return; }
        }
        const int rvx = ((pmx + 2) >> 2) * 4, rvy = ((pmy + 2) >> 2) * 4;   // the previous vector rounded to integer samples
        if ((rvx | rvy) != 0) {   // wave-uniform
            // luma: integer displacement inside the window; chroma: 1/8-sample bilinear (8.4.2.2.2) at fractions 0 or 4,
            // samples clamped at the picture edge exactly as motion compensation does
            const uint32_t ry = lds_ld4(winb, (ME_R + ME_AP + (b4 >> 2) * 4 + r + (rvy >> 2)) * ME_WS + ME_R + ME_AP + (b4 & 3) * 4 + (rvx >> 2));
            uint32_t rc = 0;
            if (lane < 32) rc = chroma_pred4(cplz ? P.ref[2] : P.ref[1], P.cw / 2, P.ch / 2, 8 * mx + cxz + (rvx >> 3), 8 * my + cyz + (rvy >> 3), rvx & 7, rvy & 7);
            if (luma_sad(ry) < (unsigned)P.sad_nz && quantises_to_nothing(ry, rc)) { settle(rvx, rvy, ry, rc); return; }
        }
    }

    // ---- 1b. seeded search (config.search = 1): the macroblock's previous-picture vector, rounded to integer samples, against
    // its eight integer neighbours on the exhaustive pass's own key.  A strict local minimum inside the search range is the
    // integer winner and the exhaustive pass is skipped; else it runs and the result is what it always was (oracle/h264_enc.c
    // motion_search).  lane = (dy - sy + 1, source row): three SADs (dx = sx - 1, sx, sx + 1) of one row each, summed over the
    // sixteen rows by DPP; the nine keys are compared on the scalar unit. ----
    unsigned best = 0xFFFFFFFFu;
    bool seeded = false;
    if (P.search == 1) {
        const int sx = (pmx + 2) >> 2, sy = (pmy + 2) >> 2;
        if (sx >= -ME_R && sx < ME_R && sy >= -ME_R && sy < ME_R) {   // wave-uniform
            const int g = lane >> 4, j = lane & 15;
            const int dyc = sy - 1 + (g < 3 ? g : 0);
            const uint4 sv = *(const uint4*)(s_src + 16 * j);
            const int ob = (ME_R + ME_AP + dyc + j) * ME_WS + ME_R + ME_AP + sx - 1;
            int sad3[3];
#pragma unroll
            for (int k = 0; k < 3; k++) {
                uint32_t a = __builtin_amdgcn_sad_u8(lds_ld4(winb, ob + k), sv.x, 0u);
                a = __builtin_amdgcn_sad_u8(lds_ld4(winb, ob + k + 4), sv.y, a);
                a = __builtin_amdgcn_sad_u8(lds_ld4(winb, ob + k + 8), sv.z, a);
                a = __builtin_amdgcn_sad_u8(lds_ld4(winb, ob + k + 12), sv.w, a);
                sad3[k] = row_sum16_dpp((int)a);
            }
            unsigned ckey = 0, nmin = 0xFFFFFFFFu;
#pragma unroll
            for (int gy = 0; gy < 3; gy++)
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    const int dx = sx - 1 + k, dy = sy - 1 + gy;
                    const unsigned sad = (unsigned)__builtin_amdgcn_readlane(sad3[k], 16 * gy);
                    const unsigned key = ((sad + (unsigned)(P.lambda * (se_len(4 * dx - pmx) + se_len(4 * dy - pmy)))) << 10) | (unsigned)(((dy + ME_R) << 5) | (dx + ME_R));
                    const bool inr = dx >= -ME_R && dx < ME_R && dy >= -ME_R && dy < ME_R;
                    if (gy == 1 && k == 1) ckey = key;
                    else if (inr && key < nmin) nmin = key;
                }
            if (ckey < nmin) { best = ckey; seeded = true; }
        }
    }

    // ---- 2. integer full search: lane = (dx, half of the dy range); 16 SAD accumulators per lane, one pass over
    // 31 window rows, every row feeding up to 16 candidates (v_sad_u8, four samples per instruction) ----
    if (!seeded) {
        const int dxi = lane & 31, half = lane >> 5;
        const int col = dxi + ME_AP, cdw = col >> 2, sh = col & 3;
        // motion-vector cost and candidate index of every dy, shifted into key position: one table per macroblock
        if (lane < 32) s_ytab[lane] = ((uint32_t)__mul24(P.lambda, se_len(4 * (lane - ME_R) - pmy)) << 10) | ((uint32_t)lane << 5);
        // the source macroblock is wave-uniform: keep its 64 dwords in SGPRs (v_sad_u8 takes one scalar operand).
        // Macroblocks inside the picture read it with scalar loads straight from the source picture.
        uint32_t srow[16][4];
        const uint8_t* sp0 = P.src + (size_t)by * P.w + bx;
        if (bx + 16 <= P.w && by + 16 <= P.h && ((P.w | (int)(uintptr_t)P.src) & 3) == 0) {
#pragma unroll
            for (int j = 0; j < 16; j++) {
                // constant address space: the source picture is read-only for this kernel, and a uniform load from
                // it is a scalar load whatever stores the compiler sees elsewhere in the function
                typedef const __attribute__((address_space(4))) uint32_t* const_u32p;
                const const_u32p q = (const_u32p)(uintptr_t)(sp0 + (size_t)j * P.w);
                srow[j][0] = q[0]; srow[j][1] = q[1]; srow[j][2] = q[2]; srow[j][3] = q[3];
            }
        } else {
#pragma unroll
            for (int j = 0; j < 16; j++) {
                const uint4 v = *(const uint4*)(s_src + 16 * j);
                srow[j][0] = __builtin_amdgcn_readfirstlane(v.x); srow[j][1] = __builtin_amdgcn_readfirstlane(v.y);
                srow[j][2] = __builtin_amdgcn_readfirstlane(v.z); srow[j][3] = __builtin_amdgcn_readfirstlane(v.w);
            }
        }
        const uint32_t kbase = ((uint32_t)__mul24(P.lambda, se_len(4 * (dxi - ME_R) - pmx)) << 10) + (uint32_t)dxi;
        if (lane == 0) s_ntask = 0;
        // two row bases (rows 0..15, 16..30) keep every row's dword offset inside the 8-bit ds_read2 offset fields
        lds_u32p wb0 = (lds_u32p)(s_win + (ME_AP + half * 16) * ME_WDW + cdw), wb1 = wb0 + 16 * ME_WDW;
        asm("" : "+v"(wb0));
        asm("" : "+v"(wb1));
        // Pass 1 - a LOWER BOUND of every candidate's SAD: columns 0..3 and 8..11 of all 16 rows (half the samples,
        // half the v_sad_u8, half the byte alignments).  Window rows stream through two register sets: row r+1 is
        // requested before row r is consumed; the scheduling barrier stops the compiler from hoisting every row's
        // LDS read to the top.
        uint32_t acc[16];
#pragma unroll
        for (int kk = 0; kk < 16; kk++) acc[kk] = 0;
        uint32_t w[2][4];
#pragma unroll
        for (int c = 0; c < 4; c++) w[0][c] = wb0[c];
#pragma unroll
        for (int r = 0; r < 31; r++) {
            if (r + 1 < 31) {
#pragma unroll
                for (int c = 0; c < 4; c++) w[(r + 1) & 1][c] = r + 1 < 16 ? wb0[(r + 1) * ME_WDW + c] : wb1[(r + 1 - 16) * ME_WDW + c];
            }
            const uint32_t* wr = w[r & 1];
            const uint32_t a0 = __builtin_amdgcn_alignbyte(wr[1], wr[0], sh), a2 = __builtin_amdgcn_alignbyte(wr[3], wr[2], sh);
#pragma unroll
            for (int kk = 0; kk < 16; kk++) {
                const int j = r - kk;       // source row met by candidate kk on window row r
                if (j >= 0 && j < 16) {
                    acc[kk] = __builtin_amdgcn_sad_u8(a0, srow[j][0], acc[kk]);
                    acc[kk] = __builtin_amdgcn_sad_u8(a2, srow[j][2], acc[kk]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // bound key = (partial SAD + lambda * bits(mv)) << 10 | dy index << 5 | dx index <= the candidate's true key
        uint32_t lbk[16];
        uint32_t bestlb = 0xFFFFFFFFu;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint4 t = *(const uint4*)(s_ytab + half * 16 + 4 * q);
            lbk[4 * q] = (acc[4 * q] << 10) + (t.x + kbase); lbk[4 * q + 1] = (acc[4 * q + 1] << 10) + (t.y + kbase);
            lbk[4 * q + 2] = (acc[4 * q + 2] << 10) + (t.z + kbase); lbk[4 * q + 3] = (acc[4 * q + 3] << 10) + (t.w + kbase);
            bestlb = min(min(bestlb, lbk[4 * q]), min(lbk[4 * q + 1], min(lbk[4 * q + 2], lbk[4 * q + 3])));
        }
        bestlb = wave_min_u32_dpp(bestlb);
        // the other half of a candidate's SAD (columns 4..7, 12..15), candidate given by its key's index bits
        auto rest_of_sad = [&](uint32_t key) -> uint32_t {
            const int ccol = (int)(key & 31) + ME_AP, csh = ccol & 3;
            lds_u32p p = (lds_u32p)(s_win + (ME_AP + (int)((key >> 5) & 31)) * ME_WDW + (ccol >> 2) + 1);
            asm("" : "+v"(p));
            uint32_t sum = 0;
            uint32_t v[2][4];
#pragma unroll
            for (int c = 0; c < 4; c++) v[0][c] = p[c];
#pragma unroll
            for (int j = 0; j < 16; j++) {
                if (j + 1 < 16) {
#pragma unroll
                    for (int c = 0; c < 4; c++) v[(j + 1) & 1][c] = p[(j + 1) * ME_WDW + c];
                }
                const uint32_t* vr = v[j & 1];
                sum = __builtin_amdgcn_sad_u8(__builtin_amdgcn_alignbyte(vr[1], vr[0], csh), srow[j][1], sum);
                sum = __builtin_amdgcn_sad_u8(__builtin_amdgcn_alignbyte(vr[3], vr[2], csh), srow[j][3], sum);
                // pin the order "row j consumed, then row j+2 requested": the optimiser would otherwise sink all the
                // (pure) SAD arithmetic below the last read and keep 16 rows in registers
                asm volatile("" : "+v"(sum) : : "memory");
                __builtin_amdgcn_sched_barrier(0);
            }
            return sum;
        };
        // the candidate with the smallest bound, completed: its true key bounds the optimum from above.  One candidate
        // for the whole wave: lane = (source row, column group) takes one dword of it, the wave adds up
        uint32_t bound;
        {
            const int j = lane & 15, g = (lane >> 4) & 1;                // columns 4..7 (g = 0) or 12..15 (g = 1)
            const int ccol = (int)(bestlb & 31) + ME_AP + 4 + 8 * g, crow = ME_AP + (int)((bestlb >> 5) & 31) + j;
            const uint32_t rv = lds_ld4(winb, crow * ME_WS + ccol), sv = *(const uint32_t*)(s_src + 16 * j + 4 + 8 * g);
            const int part = row_sum16_dpp(lane < 32 ? (int)__builtin_amdgcn_sad_u8(rv, sv, 0u) : 0);
            bound = bestlb + ((uint32_t)(__builtin_amdgcn_readlane(part, 0) + __builtin_amdgcn_readlane(part, 16)) << 10);
        }
        // Pass 2 - only candidates whose bound does not exceed that key can still win (exact: true key >= bound key);
        // they are compacted into one list and completed 64 at a time
        {
            unsigned n = 0;
#pragma unroll
            for (int kk = 0; kk < 16; kk++) n += lbk[kk] <= bound;
            unsigned at = n ? atomicAdd(&s_ntask, n) : 0u;
#pragma unroll
            for (int kk = 0; kk < 16; kk++)
                if (lbk[kk] <= bound) {
                    if (at < (unsigned)ME_TASKCAP) s_task[at] = lbk[kk];
                    at++;
                }
        }
        __syncthreads();
        const unsigned ntask = s_ntask;
        best = 0xFFFFFFFFu;
        if (ntask <= (unsigned)ME_TASKCAP) {
            for (unsigned t0 = 0; t0 < ntask; t0 += 64) {
                const bool on = t0 + lane < ntask;
                const uint32_t tk = s_task[on ? t0 + lane : 0];
                const uint32_t key = tk + (rest_of_sad(tk) << 10);
                best = on && key < best ? key : best;
            }
        } else {
            // noise-like content: nearly every candidate survives the bound; complete them where they are,
            // one candidate of every lane per round
#pragma unroll 1
            for (int kk = 0; kk < 16; kk++) {
                uint32_t tk = lbk[0];
#pragma unroll
                for (int i = 1; i < 16; i++) tk = kk == i ? lbk[i] : tk;
                const uint32_t key = tk + (rest_of_sad(tk) << 10);
                best = tk <= bound && key < best ? key : best;
            }
        }
        best = wave_min_u32_dpp(best);
        __syncthreads();   // the list's memory becomes the half-sample planes
    }
    const int ix = (int)(best & 31) - ME_R, iy = (int)((best >> 5) & 31) - ME_R;

    // ---- 3. half-sample planes on an 18x18 grid, origin (ix-1, iy-1); four samples per lane-task ----
    const int oo = (iy + ME_R + ME_AP - 1) * ME_WS + ix + ME_R + ME_AP - 1;   // window byte offset of grid (0,0)
    for (int i = lane; i < (ME_GS + 5) * 5; i += 64) {
        const int rr = i / 5, seg = (i - rr * 5) * 4;                          // b1 row rr <-> grid row rr - 2
        *(uint2*)(s_b1 + rr * ME_GP + seg) = htap4_pk(winb, oo + (rr - 2) * ME_WS + seg - 2);
    }
    __syncthreads();
    for (int i = lane; i < ME_GS * 5; i += 64) {
        const int y = i / 5, seg = (i - y * 5) * 4;
        const uint32_t Gv = lds_ld4(winb, oo + y * ME_WS + seg);
        uint2 rw[6];
#pragma unroll
        for (int k = 0; k < 6; k++) rw[k] = *(const uint2*)(s_b1 + (y + k) * ME_GP + seg);
        const uint32_t Bv = round5_pk(rw[2]);                                   // b: the horizontal sums of this row, rounded
        uint32_t Hv;                                                            // h: vertical 6-tap on the integer samples
        {
            uint32_t c[6];
            const int o = oo + (y - 2) * ME_WS + seg;
            lds_u32p pc = (lds_u32p)(winb + (o & ~3));   // window rows are ME_WS = 56 bytes apart
            asm("" : "+v"(pc));
#pragma unroll
            for (int k = 0; k < 6; k++) c[k] = __builtin_amdgcn_alignbyte(pc[k * ME_WDW + 1], pc[k * ME_WDW], o & 3);
            Hv = round5_pk(vtap4_pk(c));
        }
        const uint32_t Jv = jtap4(rw);                                          // j: vertical 6-tap on the unclipped horizontal sums
        *(uint32_t*)(s_pl + y * ME_GP + seg) = Gv;
        *(uint32_t*)(s_pl + ME_PLS + y * ME_GP + seg) = Bv;
        *(uint32_t*)(s_pl + 2 * ME_PLS + y * ME_GP + seg) = Hv;
        *(uint32_t*)(s_pl + 3 * ME_PLS + y * ME_GP + seg) = Jv;
    }
    __syncthreads();

    // ---- 4. sub-pel refinement: 8 candidates per round; lane = (candidate, pair of 4x4 blocks).  The two
    // blocks of a lane (b and b+8) ride in the low / high 16 bits of every register: differences, the 4x4
    // Hadamard (v_pk_add/sub_u16) and the absolute sum (v_sad_u16 against the bias that sample 0 carries
    // through the transform: every Hadamard output contains +d0) handle both at once. ----
    const int cand = lane >> 3, bp = lane & 7, b4x = (bp & 3) * 4, b4y = (bp >> 2) * 4;
    typedef unsigned short pk16 __attribute__((ext_vector_type(2)));
    const uint32_t sel0 = 0x0c040c00u;                    // byte 0 <- low operand byte 0, byte 2 <- high operand byte 0
    pk16 S[16];
#pragma unroll
    for (int y = 0; y < 4; y++) {
        const uint32_t lo = *(const uint32_t*)(s_src + (b4y + y) * 16 + b4x), hi = *(const uint32_t*)(s_src + (b4y + 8 + y) * 16 + b4x);
#pragma unroll
        for (int x = 0; x < 4; x++) S[4 * y + x] = __builtin_bit_cast(pk16, __builtin_amdgcn_perm(hi, lo, sel0 + 0x00010001u * x));
    }
    S[0] = __builtin_bit_cast(pk16, __builtin_bit_cast(uint32_t, S[0]) ^ 0x80008000u);
    int cx = 4 * ix, cy = 4 * iy;
    // The centre (the integer position itself) is the one candidate whose prediction is plain window samples and that
    // has no seven companions to share a round with: it is costed apart, quad-mapped - lane = (4x4 block, row), the
    // Hadamard's rows in the lane, its columns as two butterflies over the DPP quad - for a third of a round's price.
    unsigned centre_key;
    {
        const int r = lane & 3, b4 = lane >> 2;
        const uint32_t sy = *(const uint32_t*)(s_src + ((b4 >> 2) * 4 + r) * 16 + (b4 & 3) * 4);
        const uint32_t ry = lds_ld4(winb, (ME_R + ME_AP + (b4 >> 2) * 4 + r + iy) * ME_WS + ME_R + ME_AP + (b4 & 3) * 4 + ix);
        const int d0 = (int)(sy & 255u) - (int)(ry & 255u), d1 = (int)((sy >> 8) & 255u) - (int)((ry >> 8) & 255u);
        const int d2 = (int)((sy >> 16) & 255u) - (int)((ry >> 16) & 255u), d3 = (int)(sy >> 24) - (int)(ry >> 24);
        const int s0 = d0 + d3, s1 = d1 + d2, u0 = d0 - d3, u1 = d1 - d2;
        const int hr[4] = {s0 + s1, u0 + u1, s0 - s1, u0 - u1};
        const int sg1 = (lane & 1) ? -1 : 1, sg2 = (lane & 2) ? -1 : 1;
        int acc = 0;
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const int y = __builtin_amdgcn_mov_dpp(hr[c], 0xB1, 0xf, 0xf, false) + sg1 * hr[c];   // rows r, r^1: sum / difference
            const int z = __builtin_amdgcn_mov_dpp(y, 0x4E, 0xf, 0xf, false) + sg2 * y;           // rows r, r^2
            acc += z < 0 ? -z : z;
        }
        const int s16 = row_sum16_dpp(acc);
        const int tot = __builtin_amdgcn_readlane(s16, 0) + __builtin_amdgcn_readlane(s16, 16) + __builtin_amdgcn_readlane(s16, 32) + __builtin_amdgcn_readlane(s16, 48);
        centre_key = ((unsigned)(tot >> 1) + (unsigned)(P.lambda * (se_len(cx - pmx) + se_len(cy - pmy)))) << 4;   // order 0
    }
    unsigned best_cost_r = 0, bestk = 0xFFFFFFFFu;
#pragma unroll 1
    for (int round = 0; round < 2; round++) {
        // round 0: the 8 half-sample neighbours (then the centre joins the comparison); round 1: the 8 quarter-sample neighbours
        const int step = round == 1 ? 1 : 2;
        const int ord = cand + 1;                         // 0 = centre
        // neighbour order (-1,-1)(0,-1)(1,-1)(-1,0)(1,0)(-1,1)(0,1)(1,1)
        const int nn = cand >= 4 ? cand + 1 : cand;
        const int ddx = (nn % 3) - 1, ddy = (nn / 3) - 1;
        const int qx = cx + step * ddx, qy = cy + step * ddy;
        const int ox = qx - 4 * ix, oy = qy - 4 * iy;
        const int gx = 1 + (ox >> 2), gy = 1 + (oy >> 2);
        int t0, t1;
        qpel_taps(ox & 3, oy & 3, t0, t1);
        const int gb = (gy + b4y) * ME_GP + gx + b4x;
        // plane rows are ME_GP = 20 bytes apart: one dword base pointer and byte shift per tap, rows by constant index
        lds_u32p pa = (lds_u32p)(s_pl + ((t0 + gb) & ~3));
        lds_u32p pb = (lds_u32p)(s_pl + ((t1 + gb) & ~3));
        asm("" : "+v"(pa));   // keep the array's own LDS offset in the register, so that the row offsets fit the ds_read2 fields
        asm("" : "+v"(pb));
        const int sa = (t0 + gb) & 3, sb = (t1 + gb) & 3;
        pk16 d[16];
#pragma unroll
        for (int y = 0; y < 4; y++) {
            const int rl = (ME_GP / 4) * y, rh = (ME_GP / 4) * (y + 8);
            const uint32_t pl = avg4(__builtin_amdgcn_alignbyte(pa[rl + 1], pa[rl], sa), __builtin_amdgcn_alignbyte(pb[rl + 1], pb[rl], sb));
            const uint32_t ph = avg4(__builtin_amdgcn_alignbyte(pa[rh + 1], pa[rh], sa), __builtin_amdgcn_alignbyte(pb[rh + 1], pb[rh], sb));
#pragma unroll
            for (int x = 0; x < 4; x++)
                d[4 * y + x] = S[4 * y + x] - __builtin_bit_cast(pk16, __builtin_amdgcn_perm(ph, pl, sel0 + 0x00010001u * x));
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const pk16 s0 = d[4 * i] + d[4 * i + 3], s1 = d[4 * i + 1] + d[4 * i + 2];
            const pk16 d0 = d[4 * i] - d[4 * i + 3], d1 = d[4 * i + 1] - d[4 * i + 2];
            d[4 * i] = s0 + s1; d[4 * i + 1] = d0 + d1; d[4 * i + 2] = s0 - s1; d[4 * i + 3] = d0 - d1;
        }
        uint32_t sum = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const pk16 s0 = d[j] + d[12 + j], s1 = d[4 + j] + d[8 + j];
            const pk16 d0 = d[j] - d[12 + j], d1 = d[4 + j] - d[8 + j];
            sum = __builtin_amdgcn_sad_u16(__builtin_bit_cast(uint32_t, s0 + s1), 0x80008000u, sum);
            sum = __builtin_amdgcn_sad_u16(__builtin_bit_cast(uint32_t, d0 + d1), 0x80008000u, sum);
            sum = __builtin_amdgcn_sad_u16(__builtin_bit_cast(uint32_t, s0 - s1), 0x80008000u, sum);
            sum = __builtin_amdgcn_sad_u16(__builtin_bit_cast(uint32_t, d0 - d1), 0x80008000u, sum);
        }
        const int s = group_sum8_dpp((int)sum);
        const unsigned cost = (unsigned)(s >> 1) + (unsigned)(P.lambda * (se_len(qx - pmx) + se_len(qy - pmy)));
        const unsigned key = (cost << 4) | (unsigned)ord;
        bestk = key < bestk ? key : bestk;
        if (round == 0) bestk = centre_key < bestk ? centre_key : bestk;
        bestk = wave_min_u32_dpp(bestk);
        const int w = (int)(bestk & 15);
        best_cost_r = bestk >> 4;
        if (w) {
            const int n = w - 1, wn = n >= 4 ? n + 1 : n;
            cx += step * ((wn % 3) - 1);
            cy += step * ((wn / 3) - 1);
        }
        bestk = best_cost_r << 4;                        // the next pass starts from "stay" (order 0)
    }
    const int rbits = P.nref <= 1 ? 0 : (P.nref == 2 ? 1 : (rf == 0 ? 1 : 3));   // te(v) of ref_idx_l0 (9.1)
    // ---- 4b. partitions (oracle/h264_enc.c motion_search): a macroblock whose 16x16 cost reaches PART_TEST_MIN is also costed
    // as two 16x8, two 8x16 and four 8x8 partitions, each refined on its own (half-, then quarter-sample neighbours, SATD +
    // lambda * bits(mv - pmv)) from the integer winner - all within the half-sample planes already in LDS.  Same lane layout
    // as above (candidate, pair of 4x4 blocks); the low half of every register belongs to a block of the upper two
    // quadrants, the high half to one of the lower two, and each half follows its own partition's centre.
    int shape = 0;
    int qvx0 = cx, qvy0 = cy, qvx1 = cx, qvy1 = cy, qvx2 = cx, qvy2 = cy, qvx3 = cx, qvy3 = cy;   // vectors of the four quadrants
    if (best_cost_r >= (unsigned)PART_TEST_MIN) {   // wave-uniform
        // sums of |Hadamard| of the lane's two blocks (low | high << 16) with the low block predicted at (qlx, qly), the high at (qhx, qhy)
        auto eval = [&](int qlx, int qly, int qhx, int qhy) -> uint32_t {
            const int oxl = qlx - 4 * ix, oyl = qly - 4 * iy, oxh = qhx - 4 * ix, oyh = qhy - 4 * iy;
            int t0, t1, u0, u1;
            qpel_taps(oxl & 3, oyl & 3, t0, t1);
            qpel_taps(oxh & 3, oyh & 3, u0, u1);
            const int gbl = (1 + (oyl >> 2) + b4y) * ME_GP + 1 + (oxl >> 2) + b4x;
            const int gbh = (1 + (oyh >> 2) + b4y + 8) * ME_GP + 1 + (oxh >> 2) + b4x;
            lds_u32p pal = (lds_u32p)(s_pl + ((t0 + gbl) & ~3)), pbl = (lds_u32p)(s_pl + ((t1 + gbl) & ~3));
            lds_u32p pah = (lds_u32p)(s_pl + ((u0 + gbh) & ~3)), pbh = (lds_u32p)(s_pl + ((u1 + gbh) & ~3));
            const int sal = (t0 + gbl) & 3, sbl = (t1 + gbl) & 3, sah = (u0 + gbh) & 3, sbh = (u1 + gbh) & 3;
            pk16 d[16];
#pragma unroll
            for (int y = 0; y < 4; y++) {
                const int r = (ME_GP / 4) * y;
                const uint32_t pl = avg4(__builtin_amdgcn_alignbyte(pal[r + 1], pal[r], sal), __builtin_amdgcn_alignbyte(pbl[r + 1], pbl[r], sbl));
                const uint32_t ph = avg4(__builtin_amdgcn_alignbyte(pah[r + 1], pah[r], sah), __builtin_amdgcn_alignbyte(pbh[r + 1], pbh[r], sbh));
#pragma unroll
                for (int x = 0; x < 4; x++)
                    d[4 * y + x] = S[4 * y + x] - __builtin_bit_cast(pk16, __builtin_amdgcn_perm(ph, pl, sel0 + 0x00010001u * x));
            }
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const pk16 s0 = d[4 * i] + d[4 * i + 3], s1 = d[4 * i + 1] + d[4 * i + 2];
                const pk16 d0 = d[4 * i] - d[4 * i + 3], d1 = d[4 * i + 1] - d[4 * i + 2];
                d[4 * i] = s0 + s1; d[4 * i + 1] = d0 + d1; d[4 * i + 2] = s0 - s1; d[4 * i + 3] = d0 - d1;
            }
            // every output carries the bias 0x8000 of source sample 0 (above); |x - bias| per half, summed per half (<= 16 320)
            typedef short spk16 __attribute__((ext_vector_type(2)));
            pk16 acc = {0, 0};
            auto add_abs = [&](pk16 v) {
                const spk16 t = __builtin_bit_cast(spk16, __builtin_bit_cast(uint32_t, v) ^ 0x80008000u);
                acc += __builtin_bit_cast(pk16, __builtin_elementwise_max(t, -t));
            };
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const pk16 s0 = d[j] + d[12 + j], s1 = d[4 + j] + d[8 + j];
                const pk16 d0 = d[j] - d[12 + j], d1 = d[4 + j] - d[8 + j];
                add_abs(s0 + s1); add_abs(d0 + d1); add_abs(s0 - s1); add_abs(d0 - d1);
            }
            return __builtin_bit_cast(uint32_t, acc);
        };
        auto xor4 = [](int v) {   // the value of lane ^ 4
            int q = __builtin_amdgcn_update_dpp(0, v, 0x104, 0xf, 0x5, false);   // row_shl:4 into quads 0, 2: from lane + 4
            return __builtin_amdgcn_update_dpp(q, v, 0x114, 0xf, 0xa, false);    // row_shr:4 into quads 1, 3: from lane - 4
        };
        // a partition's sum from the blocks' sums, for the partition of the lane's low block (slo) and of its high block (shi)
        auto part_sums = [&](int sh, uint32_t acc, int& slo, int& shi) {
            const int alo = (int)(acc & 0xFFFFu), ahi = (int)(acc >> 16);
            if (sh == 1) { slo = group_sum8_dpp(alo); shi = group_sum8_dpp(ahi); }   // 16x8: all upper / all lower blocks
            else if (sh == 2) {   // 8x16: the blocks of the left (b4x < 8) or right half, both register halves
                int v = alo + ahi;
                v += __builtin_amdgcn_mov_dpp(v, 0xB1, 0xf, 0xf, false);
                v += xor4(v);
                slo = v; shi = v;
            } else {              // 8x8: left / right half, register halves apart
                int v = alo, u = ahi;
                v += __builtin_amdgcn_mov_dpp(v, 0xB1, 0xf, 0xf, false);
                u += __builtin_amdgcn_mov_dpp(u, 0xB1, 0xf, 0xf, false);
                slo = v + xor4(v); shi = u + xor4(u);
            }
        };
        auto min_over_candidates = [](unsigned k) {   // lanes 8 apart hold the same blocks for the eight candidates
            k = min(k, (unsigned)__shfl_xor((int)k, 8));
            k = min(k, (unsigned)__shfl_xor((int)k, 16));
            return min(k, (unsigned)__shfl_xor((int)k, 32));
        };
        const uint32_t acc0 = eval(4 * ix, 4 * iy, 4 * ix, 4 * iy);   // the common starting point of every partition
        const int nn = cand >= 4 ? cand + 1 : cand, ddx = (nn % 3) - 1, ddy = (nn / 3) - 1;
        unsigned best_all = best_cost_r;
#pragma unroll 1
        for (int sh = 1; sh <= 3; sh++) {
            int clx = 4 * ix, cly = 4 * iy, chx = 4 * ix, chy = 4 * iy;   // centre of the low / high block's partition
            unsigned klo, khi;
            {
                int slo, shi;
                part_sums(sh, acc0, slo, shi);
                const unsigned mvb = (unsigned)(P.lambda * (se_len(4 * ix - pmx) + se_len(4 * iy - pmy)));
                klo = ((unsigned)(slo >> 1) + mvb) << 4; khi = ((unsigned)(shi >> 1) + mvb) << 4;   // order 0: stay
            }
#pragma unroll 1
            for (int round = 0; round < 2; round++) {
                const int step = round == 1 ? 1 : 2;
                const int qlx = clx + step * ddx, qly = cly + step * ddy, qhx = chx + step * ddx, qhy = chy + step * ddy;
                int slo, shi;
                part_sums(sh, eval(qlx, qly, qhx, qhy), slo, shi);
                const unsigned cl = (unsigned)(slo >> 1) + (unsigned)(P.lambda * (se_len(qlx - pmx) + se_len(qly - pmy)));
                const unsigned chh = (unsigned)(shi >> 1) + (unsigned)(P.lambda * (se_len(qhx - pmx) + se_len(qhy - pmy)));
                klo = min_over_candidates(min(klo, (cl << 4) | (unsigned)(cand + 1)));
                khi = min_over_candidates(min(khi, (chh << 4) | (unsigned)(cand + 1)));
                const int wl = (int)(klo & 15u), wh = (int)(khi & 15u);
                if (wl) { const int n = wl - 1, wn = n >= 4 ? n + 1 : n; clx += step * ((wn % 3) - 1); cly += step * ((wn / 3) - 1); }
                if (wh) { const int n = wh - 1, wn = n >= 4 ? n + 1 : n; chx += step * ((wn % 3) - 1); chy += step * ((wn / 3) - 1); }
                klo &= ~15u; khi &= ~15u;   // the next round starts from "stay"
            }
            // lane 0 holds quadrants 0 (low) and 2 (high), lane 2 quadrants 1 and 3
            const unsigned l0 = (unsigned)__builtin_amdgcn_readlane((int)klo, 0) >> 4, l2 = (unsigned)__builtin_amdgcn_readlane((int)klo, 2) >> 4;
            const unsigned h0 = (unsigned)__builtin_amdgcn_readlane((int)khi, 0) >> 4, h2 = (unsigned)__builtin_amdgcn_readlane((int)khi, 2) >> 4;
            const unsigned tot = (sh == 1 ? l0 + h0 : (sh == 2 ? l0 + l2 : l0 + l2 + h0 + h2)) + (unsigned)(P.lambda * (sh == 3 ? 8 + 3 * rbits : 2 + rbits));
            if (tot < best_all) {
                best_all = tot; shape = sh;
                qvx0 = __builtin_amdgcn_readlane(clx, 0); qvy0 = __builtin_amdgcn_readlane(cly, 0);
                qvx1 = __builtin_amdgcn_readlane(clx, 2); qvy1 = __builtin_amdgcn_readlane(cly, 2);
                qvx2 = __builtin_amdgcn_readlane(chx, 0); qvy2 = __builtin_amdgcn_readlane(chy, 0);
                qvx3 = __builtin_amdgcn_readlane(chx, 2); qvy3 = __builtin_amdgcn_readlane(chy, 2);
            }
        }
        best_cost_r = best_all;
    }
    // this reference picture against the best so far
    const unsigned total = best_cost_r + (unsigned)(P.lambda * rbits);
    const bool better = total < prev_total;   // wave-uniform; rf == 0: always
    if (better) {
        // the prediction goes to the reconstruction planes (k_tq turns it into the reconstruction in place): luma from the
        // half-sample planes still in LDS, lane = (row, 4-sample segment); chroma by 8.4.2.2.2; every sample by the vector of
        // its quadrant
        const int y = lane >> 2, seg = (lane & 3) * 4;
        const int pl = lane >> 4, cyy = (lane >> 1) & 7, cxx = (lane & 1) * 4;   // chroma: lanes < 32
        if (shape == 0) {   // one vector: the tap selection stays on the scalar unit
            const int ox = cx - 4 * ix, oy = cy - 4 * iy;
            int t0, t1;
            qpel_taps(ox & 3, oy & 3, t0, t1);
            const int gb = (1 + (oy >> 2) + y) * ME_GP + 1 + (ox >> 2) + seg;
            *(uint32_t*)(P.rec[0] + (size_t)(by + y) * P.cw + bx + seg) = avg4(lds_ld4(s_pl, t0 + gb), lds_ld4(s_pl, t1 + gb));
            if (lane < 32)
                *(uint32_t*)((pl ? P.rec[2] : P.rec[1]) + (size_t)(8 * my + cyy) * (P.cw / 2) + 8 * mx + cxx) =
                    chroma_pred4(pl ? RV : RU, P.cw / 2, P.ch / 2, 8 * mx + cxx + (cx >> 3), 8 * my + cyy + (cy >> 3), cx & 7, cy & 7);
        } else {
            {
                const bool lowq = y < 8, leftq = seg < 8;
                const int vx = lowq ? (leftq ? qvx0 : qvx1) : (leftq ? qvx2 : qvx3), vy = lowq ? (leftq ? qvy0 : qvy1) : (leftq ? qvy2 : qvy3);
                const int ox = vx - 4 * ix, oy = vy - 4 * iy;
                int t0, t1;
                qpel_taps(ox & 3, oy & 3, t0, t1);
                const int gb = (1 + (oy >> 2) + y) * ME_GP + 1 + (ox >> 2) + seg;
                *(uint32_t*)(P.rec[0] + (size_t)(by + y) * P.cw + bx + seg) = avg4(lds_ld4(s_pl, t0 + gb), lds_ld4(s_pl, t1 + gb));
            }
            if (lane < 32) {
                const bool lowq = cyy < 4, leftq = cxx < 4;
                const int vx = lowq ? (leftq ? qvx0 : qvx1) : (leftq ? qvx2 : qvx3), vy = lowq ? (leftq ? qvy0 : qvy1) : (leftq ? qvy2 : qvy3);
                *(uint32_t*)((pl ? P.rec[2] : P.rec[1]) + (size_t)(8 * my + cyy) * (P.cw / 2) + 8 * mx + cxx) =
                    chroma_pred4(pl ? RV : RU, P.cw / 2, P.ch / 2, 8 * mx + cxx + (vx >> 3), 8 * my + cyy + (vy >> 3), vx & 7, vy & 7);
            }
        }
        if (lane == 0) {
            uint4* m = (uint4*)(P.mb + mbi);
            const uint32_t type = shape ? (uint32_t)(MB_P16X8 + shape - 1) : (uint32_t)MB_P16;
            auto pk = [](int x, int yv) { return ((uint32_t)x & 0xFFFFu) | ((uint32_t)yv << 16); };
            m[0] = make_uint4(pk(qvx0, qvy0), type | ((uint32_t)rf << 16), 0u, 0u);   // ref_idx_l0 rides in chroma_mode
            m[1] = make_uint4(0u, 0u, 0u, 0u);
            *(uint4*)(P.mvq + (size_t)mbi * 8) = make_uint4(pk(qvx0, qvy0), pk(qvx1, qvy1), pk(qvx2, qvy2), pk(qvx3, qvy3));
            P.me_total[mbi] = total;
        }
    }
    const unsigned best_cost = better ? total : prev_total;
    if (lane == 0) P.me_cost[mbi] = (uint16_t)(best_cost < 16383u ? best_cost : 16383u);   // scene-change statistic, summed by k_bit_scan
    if (rf != P.rf_last) return;
    // ---- 5. intra or inter: from the motion cost and the SOURCE picture alone (both final before the launch).  A macroblock
    // whose motion cost is INTRA_TEST_MIN or more is also costed as Intra16x16 with the vertical / horizontal / DC prediction
    // built from the source samples above and to the left (SATD, lane = (mode, 4x4 block)) + 8 lambda; if that is lower it is
    // marked for the intra pass (k_pintra_rows predicts from the true reconstruction) and gets no inter prediction ----
    if (best_cost >= (unsigned)INTRA_TEST_MIN) {   // wave-uniform
        const bool topav = P.sl.has_top(my), leftav = mx > 0;
        uint8_t* const s_nb = (uint8_t*)s_ytab;     // the search's table is no longer needed: 16 samples above, 16 to the left
        int nb = 0;
        if (lane < 16) nb = topav ? src_px(P.src, P.w, P.h, bx + lane, by - 1) : 0;
        else if (lane < 32) nb = leftav ? src_px(P.src, P.w, P.h, bx - 1, by + lane - 16) : 0;
        if (lane < 32) s_nb[lane] = (uint8_t)nb;
        const int s16 = row_sum16_dpp(lane < 32 ? nb : 0);
        const int st = __builtin_amdgcn_readlane(s16, 0), sl = __builtin_amdgcn_readlane(s16, 16);
        const int dc = (topav && leftav) ? (st + sl + 16) >> 5 : topav ? (st + 8) >> 4 : leftav ? (sl + 8) >> 4 : 128;
        __syncthreads();
        const int mode = lane >> 4, blk = lane & 15, x0 = (blk & 3) * 4, y0 = (blk >> 2) * 4;
        int d[16];
#pragma unroll
        for (int y = 0; y < 4; y++) {
            const uint32_t sw = *(const uint32_t*)(s_src + (y0 + y) * 16 + x0);
#pragma unroll
            for (int x = 0; x < 4; x++) {
                const int pr = mode == 0 ? s_nb[x0 + x] : mode == 1 ? s_nb[16 + y0 + y] : dc;
                d[4 * y + x] = (int)((sw >> (8 * x)) & 255u) - pr;
            }
        }
        const int satd = row_sum16_dpp(lane < 48 ? hadamard_abs(d) : 0) >> 1;
        const bool ok = mode == 0 ? topav : mode == 1 ? leftav : mode == 2;
        const unsigned est = wave_min_u32_dpp(ok ? (unsigned)satd : 0xFFFFFFFFu) + 8u * (unsigned)P.lambda;
        if (est < best_cost) {
            if (lane == 0) {
                uint4* m = (uint4*)(P.mb + mbi);
                m[0] = make_uint4(0u, (uint32_t)MB_I16, 0u, 0u);
                m[1] = make_uint4(0u, 0u, 0u, 0u);
                P.me_cost[mbi] = (uint16_t)(0x8000u | (best_cost < 16383u ? best_cost : 16383u));   // bit 15: for k_pintra_rows
                *P.anyintra = P.pic_serial;
            }
            return;
        }
    }
}

}  // namespace h264
