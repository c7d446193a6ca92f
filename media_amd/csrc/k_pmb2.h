// media_amd/csrc/k_pmb2.h -- inter macroblock coding, second form of k_pmb.h (same results, bit for bit).
//
// One wavefront per PAIR of macroblocks with lane = (4x4 block, row): every lane owns FOUR consecutive samples
// of one row from motion compensation to reconstruction, so nothing per-sample goes through LDS:
//   - the reference window is fetched as aligned dwords; a lane reads its samples as dwords and realigns
//     them with v_alignbyte (no byte-wide LDS traffic)
//   - the 4x4 transforms run as an in-lane row pass plus a column pass over the four lanes of a quad
//     (DPP quad_perm broadcasts), forward and inverse
//   - prediction stays in a register until it is added back for the reconstruction
// Luma uses all 64 lanes (16 blocks x 4 rows) per macroblock; chroma needs 32 (2 planes x 4 blocks x 4 rows),
// so the two macroblocks of the pair share one chroma pass.
//
// SURVEY.md 8a rows a6.2 + a6.3 (inside ISVCEncoder::EncodeFrame,
// /root/reference/video_codec/VideoEncoderOpenH264.cpp:344).  Algorithmic HBM bytes per macroblock:
// source 384 + reference 384 in; reconstruction 384 + levels 768 + side info 32 out = 1952.
#pragma once
#include "dev_common.h"
#include "k_pmb.h"

namespace h264 {

template <int K>
__device__ __forceinline__ int quad_bcast(int v)
{
    return __builtin_amdgcn_mov_dpp(v, K * 0x55, 0xf, 0xf, false);  // quad_perm:[K,K,K,K]
}
// four bytes starting at byte offset o of an LDS byte array (any alignment)
__device__ __forceinline__ uint32_t lds_ld4(const uint8_t* base, int o)
{
    const uint32_t* p = (const uint32_t*)(base + (o & ~3));
    return __builtin_amdgcn_alignbyte(p[1], p[0], o & 3);
}
__device__ __forceinline__ int byte_of(uint32_t v, int k) { return (int)((v >> (8 * k)) & 255); }
__device__ __forceinline__ uint32_t pack4(int a, int b, int c, int d)
{
    return (uint32_t)a | ((uint32_t)b << 8) | ((uint32_t)c << 16) | ((uint32_t)d << 24);
}
__device__ __forceinline__ uint32_t avg4(uint32_t a, uint32_t b)  // per-byte (a + b + 1) >> 1
{
    return __builtin_amdgcn_lerp(a, b, 0x01010101u);   // v_lerp_u8: rounding bit = bit 0 of the third operand's bytes
}

// Note on rounding: "shift, clamp to 0..255, pack two bytes" written in C makes hipcc (ROCm 7.2, gfx950) select
// v_ashr_pk_u8_i32 and then OR further bytes into its result assuming the upper 16 bits are zero; the instruction
// leaves them unchanged (observed: bytes 2,3 corrupted).  The filters below round with v_sat_pk_u8_i16 instead.
typedef const __attribute__((address_space(3))) uint32_t* lds_u32p;   // dword pointer into LDS

// ---- packed 16-bit helpers for the half-sample filters (8.4.2.2.1) ----
typedef unsigned short me_pk16 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ me_pk16 me_pk(uint32_t v) { return __builtin_bit_cast(me_pk16, v); }
__device__ __forceinline__ uint32_t me_u32(me_pk16 v) { return __builtin_bit_cast(uint32_t, v); }
// bytes j, j+1 (0 <= j <= 6) of the 8-byte window {hi:lo}, zero-extended into the two 16-bit halves
__device__ __forceinline__ me_pk16 byte_pair(uint32_t hi, uint32_t lo, int j)
{
    return me_pk(__builtin_amdgcn_perm(hi, lo, 0x0c000c00u + (uint32_t)j + ((uint32_t)(j + 1) << 16)));
}
// 6-tap (1,-5,20,20,-5,1) of six packed operands; every partial sum of 8-bit samples fits 16 bits
__device__ __forceinline__ me_pk16 tap6_pk(me_pk16 a, me_pk16 b, me_pk16 c, me_pk16 d, me_pk16 e, me_pk16 f)
{
    const me_pk16 m5 = me_pk(0xFFFBFFFBu), p20 = me_pk(0x00140014u);
    return (a + f) + m5 * (b + e) + p20 * (c + d);
}
// two packed unclipped sums -> (x + 16) >> 5 clamped to 0..255, as two bytes in bits 0..15
__device__ __forceinline__ uint32_t round5_sat(me_pk16 v)
{
    typedef short spk __attribute__((ext_vector_type(2)));
    const spk r = __builtin_bit_cast(spk, v + me_pk(0x00100010u)) >> 5;
    uint32_t o;
    asm("v_sat_pk_u8_i16 %0, %1" : "=v"(o) : "v"(__builtin_bit_cast(uint32_t, r)));
    return o;
}
__device__ __forceinline__ uint32_t bytes4(uint32_t lo2, uint32_t hi2) { return __builtin_amdgcn_perm(hi2, lo2, 0x05040100u); }
// acc + (int16 half of x) * (int16 c): v_mad_i32_i16, the half chosen by op_sel
__device__ __forceinline__ int mad16_lo(uint32_t x, int c, int acc)
{
    int o;
    asm("v_mad_i32_i16 %0, %1, %2, %3" : "=v"(o) : "v"(x), "v"(c), "v"(acc));
    return o;
}
__device__ __forceinline__ int mad16_hi(uint32_t x, int c, int acc)
{
    int o;
    asm("v_mad_i32_i16 %0, %1, %2, %3 op_sel:[1,0,0,0]" : "=v"(o) : "v"(x), "v"(c), "v"(acc));
    return o;
}

// four horizontal 6-tap sums (unclipped) as two packed words; `o` = LDS byte offset of the sample 2 left of output 0
__device__ __forceinline__ uint2 htap4_pk(const uint8_t* base, int o)
{
    const uint32_t* p = (const uint32_t*)(base + (o & ~3));
    const int sh = o & 3;
    const uint32_t d0 = p[0], d1 = p[1], d2 = p[2], d3 = p[3];
    const uint32_t a0 = __builtin_amdgcn_alignbyte(d1, d0, sh), a1 = __builtin_amdgcn_alignbyte(d2, d1, sh), a2 = __builtin_amdgcn_alignbyte(d3, d2, sh);
    // V[j] = samples (j, j+1) of the nine the four outputs need
    const me_pk16 V0 = byte_pair(a1, a0, 0), V1 = byte_pair(a1, a0, 1), V2 = byte_pair(a1, a0, 2), V3 = byte_pair(a1, a0, 3);
    const me_pk16 V4 = byte_pair(a2, a1, 0), V5 = byte_pair(a2, a1, 1), V6 = byte_pair(a2, a1, 2), V7 = byte_pair(a2, a1, 3);
    return make_uint2(me_u32(tap6_pk(V0, V1, V2, V3, V4, V5)), me_u32(tap6_pk(V2, V3, V4, V5, V6, V7)));
}
// four vertical 6-tap sums from six rows of four samples each
__device__ __forceinline__ uint2 vtap4_pk(const uint32_t c[6])
{
    me_pk16 lo[6], hi[6];
#pragma unroll
    for (int k = 0; k < 6; k++) { lo[k] = byte_pair(0, c[k], 0); hi[k] = byte_pair(0, c[k], 2); }
    return make_uint2(me_u32(tap6_pk(lo[0], lo[1], lo[2], lo[3], lo[4], lo[5])), me_u32(tap6_pk(hi[0], hi[1], hi[2], hi[3], hi[4], hi[5])));
}
__device__ __forceinline__ uint32_t round5_pk(uint2 sums) { return bytes4(round5_sat(me_pk(sums.x)), round5_sat(me_pk(sums.y))); }
// centre sample j: vertical 6-tap over six rows of packed unclipped horizontal sums (32-bit), (x + 512) >> 10, clamped
__device__ __forceinline__ uint32_t jtap4(const uint2 rw[6])
{
    int t[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        int acc = 512;
#pragma unroll
        for (int m = 0; m < 6; m++) {
            const uint32_t x = (k & 2) ? rw[m].y : rw[m].x;
            const int cf = (m == 0 || m == 5) ? 1 : ((m == 1 || m == 4) ? -5 : 20);
            acc = (k & 1) ? mad16_hi(x, cf, acc) : mad16_lo(x, cf, acc);
        }
        t[k] = acc >> 10;
    }
    uint32_t j01, j23;
    asm("v_sat_pk_u8_i16 %0, %1" : "=v"(j01) : "v"((uint32_t)(t[0] & 0xFFFF) | ((uint32_t)t[1] << 16)));
    asm("v_sat_pk_u8_i16 %0, %1" : "=v"(j23) : "v"((uint32_t)(t[2] & 0xFFFF) | ((uint32_t)t[3] << 16)));
    return bytes4(j01, j23);
}

// forward 4x4 core transform of the quad's block: in d[4] = this lane's residual row, out d[4] = row `r` of W
__device__ __forceinline__ void fdct_quad(int d[4], int r)
{
    {
        const int s0 = d[0] + d[3], s1 = d[1] + d[2], d0 = d[0] - d[3], d1 = d[1] - d[2];
        d[0] = s0 + s1; d[1] = 2 * d0 + d1; d[2] = s0 - s1; d[3] = d0 - 2 * d1;
    }
    const bool odd = r & 1;
    const int mA = r == 1 ? 2 : 1, mB = r == 0 ? 1 : (r == 1 ? 1 : (r == 2 ? -1 : -2));
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const int t0 = quad_bcast<0>(d[c]), t1 = quad_bcast<1>(d[c]), t2 = quad_bcast<2>(d[c]), t3 = quad_bcast<3>(d[c]);
        const int A = odd ? t0 - t3 : t0 + t3, B = odd ? t1 - t2 : t1 + t2;
        d[c] = mA * A + mB * B;  // r0: s0+s1, r1: 2d0+d1, r2: s0-s1, r3: d0-2d1
    }
}
// inverse (8.5.12.2): in d[4] = row r of the scaled coefficients, out d[4] = row r of the residual (rounded)
__device__ __forceinline__ void idct_quad(int d[4], int r)
{
    {
        const int e0 = d[0] + d[2], e1 = d[0] - d[2], e2 = (d[1] >> 1) - d[3], e3 = d[1] + (d[3] >> 1);
        d[0] = e0 + e3; d[1] = e1 + e2; d[2] = e1 - e2; d[3] = e0 - e3;
    }
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const int f0 = quad_bcast<0>(d[c]), f1 = quad_bcast<1>(d[c]), f2 = quad_bcast<2>(d[c]), f3 = quad_bcast<3>(d[c]);
        const int g0 = f0 + f2, g1 = f0 - f2, g2 = (f1 >> 1) - f3, g3 = f1 + (f3 >> 1);
        const int v = r == 0 ? g0 + g3 : (r == 1 ? g1 + g2 : (r == 2 ? g1 - g2 : g0 - g3));
        d[c] = (v + 32) >> 6;
    }
}

__constant__ const uint16_t c_zz_row[4] = {0x6510, 0xC742, 0xDB83, 0xFEA9};  // zig-zag index of raster (r, c), nibble c

// One wavefront per PAIR of raster-consecutive macroblocks (2q, 2q+1): the luma chain runs once per macroblock
// on all 64 lanes (lane = 4x4 block, row); the chroma chain needs 32 lanes per macroblock, so both macroblocks'
// chroma share ONE pass (lanes 0..31 first, 32..63 second macroblock) instead of two half-empty ones.
__global__ __launch_bounds__(64) void k_pmb2(FrameParams P0)
{
    __builtin_amdgcn_s_setprio(2);   // short and on the way to the loop filter: ahead of another stream's motion search
    const FrameParams P = batch_view(P0, blockIdx.y);
    const int lane = threadIdx.x;
    const int nmb = P.mbw * P.band.rows, mb0 = P.band.row0 * P.mbw;   // (pairs are formed inside this instance's band)
    const int q = xcd_mb_index(blockIdx.x, (nmb + 1) >> 1);
    const int cs = P.cw / 2;
    const bool two = 2 * q + 1 < nmb;   // the last pair of an odd macroblock count has one member

    __shared__ __attribute__((aligned(16))) uint8_t s_w[2][21 * 28 + 12];     // luma windows, pitch 28
    __shared__ __attribute__((aligned(16))) uint8_t s_cw[2][2][9 * 12 + 12];  // chroma windows, pitch 12
    __shared__ __attribute__((aligned(16))) int16_t s_b1[2][21 * 16];         // unclipped horizontal sums (centre positions)
    __shared__ __attribute__((aligned(16))) int16_t s_lv[2][LV_STRIDE];

    // lane geometry: luma (blk, r) -> samples (lx..lx+3, ly); chroma: macroblock hh, plane, block, row
    const int blk = lane >> 2, r = lane & 3, lx = blk_x(blk) * 4, ly = blk_y(blk) * 4 + r;
    const int hh = lane >> 5, cpl = (lane >> 4) & 1, cblk = (lane >> 2) & 3, cx = (cblk & 1) * 4, cy = (cblk >> 1) * 4 + r;

    // ---- wave-uniform context of both macroblocks (scalar unit) ----
    int mbi[2], mx[2], my[2], mvx[2], mvy[2], wxo[2], cxo[2];
    bool zres[2];   // k_me found that the zero-motion residual quantises to nothing: prediction = reconstruction
    Mv pred[2], skip[2];
    MbInfo* m[2];
#pragma unroll
    for (int h = 0; h < 2; h++) {
        mbi[h] = mb0 + 2 * q + ((h && two) ? 1 : 0);
        my[h] = P.mbdiv.row(mbi[h]); mx[h] = mbi[h] - my[h] * P.mbw;
        m[h] = P.mb + mbi[h];
        const int mvw = __builtin_amdgcn_readfirstlane(*(const int*)m[h]);   // mvx | mvy << 16
        mvx[h] = (int)(int16_t)(mvw & 0xFFFF); mvy[h] = mvw >> 16;
        zres[h] = ((__builtin_amdgcn_readfirstlane(((const int*)m[h])[1]) >> 8) & 255) != 0;   // MbInfo.i16_mode, set by k_me
        pred[h] = predict_mv(P, mx[h], my[h], skip[h]);
    }

    // ---- all global requests first: reference windows, source samples ----
    uint32_t wv[2][3], cwv[2], src4[2], csrc4 = 0;
    bool interior[2], cinterior[2];
    int x0[2], y0[2], cx0[2], cy0[2];
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const int bx = 16 * mx[h], by = 16 * my[h];
        x0[h] = bx + (mvx[h] >> 2) - 2; y0[h] = by + (mvy[h] >> 2) - 2;
        const int xa = x0[h] & ~3;
        wxo[h] = x0[h] - xa;
        cx0[h] = 8 * mx[h] + (mvx[h] >> 3); cy0[h] = 8 * my[h] + (mvy[h] >> 3);
        const int cxa = cx0[h] & ~3;
        cxo[h] = cx0[h] - cxa;
        interior[h] = xa >= 0 && xa + 28 <= P.cw && y0[h] >= 0 && y0[h] + 21 <= P.ch;
        cinterior[h] = cxa >= 0 && cxa + 12 <= cs && cy0[h] >= 0 && cy0[h] + 9 <= P.ch / 2;
        wv[h][0] = wv[h][1] = wv[h][2] = 0; cwv[h] = 0;
        if (interior[h]) {
#pragma unroll
            for (int t = 0; t < 3; t++) {
                const int i = lane + 64 * t, rr = (int)(((unsigned)i * 9363u) >> 16), c = i - rr * 7;   // i / 7 for i < 192
                if (i < 147) wv[h][t] = *(const uint32_t*)(P.ref[0] + (size_t)(y0[h] + rr) * P.cw + xa + 4 * c);
            }
        }
        if (cinterior[h] && lane < 54) {
            const int pl = lane >= 27, k = lane - pl * 27, rr = (int)(((unsigned)k * 21846u) >> 16), c = k - rr * 3;   // k / 3 for k < 27
            cwv[h] = *(const uint32_t*)((pl ? P.ref[2] : P.ref[1]) + (size_t)(cy0[h] + rr) * cs + cxa + 4 * c);
        }
        {
            const uint8_t* Y = P.src;
            const int gy = by + ly, gx = bx + lx;
            const uint8_t* p = Y + (size_t)(gy < P.h ? gy : P.h - 1) * P.w + gx;
            if (gx + 3 < P.w && (((uintptr_t)p) & 3) == 0) src4[h] = *(const uint32_t*)p;
            else src4[h] = pack4(src_px(Y, P.w, P.h, gx, gy), src_px(Y, P.w, P.h, gx + 1, gy), src_px(Y, P.w, P.h, gx + 2, gy), src_px(Y, P.w, P.h, gx + 3, gy));
        }
    }
    {   // chroma source samples of this lane's macroblock
        const int cmx = hh ? mx[1] : mx[0], cmy = hh ? my[1] : my[0];
        csrc4 = src_chroma4(P, cpl, 8 * cmx + cx, 8 * cmy + cy);
    }
    for (int i = lane; i < 2 * (LV_STRIDE * 2 / 16); i += 64) ((uint4*)&s_lv[0][0])[i] = make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int h = 0; h < 2; h++) {
        if (interior[h]) {
#pragma unroll
            for (int t = 0; t < 3; t++) {
                const int i = lane + 64 * t, rr = (int)(((unsigned)i * 9363u) >> 16), c = i - rr * 7;
                if (i < 147) *(uint32_t*)(s_w[h] + rr * 28 + 4 * c) = wv[h][t];
            }
        } else {
            for (int i = lane; i < 21 * 21; i += 64) {
                const int rr = i / 21, c = i - rr * 21;
                s_w[h][rr * 28 + wxo[h] + c] = P.ref[0][(size_t)clip3(0, P.ch - 1, y0[h] + rr) * P.cw + clip3(0, P.cw - 1, x0[h] + c)];
            }
        }
        if (cinterior[h]) {
            if (lane < 54) {
                const int pl = lane >= 27, k = lane - pl * 27, rr = (int)(((unsigned)k * 21846u) >> 16), c = k - rr * 3;
                *(uint32_t*)(s_cw[h][pl] + rr * 12 + 4 * c) = cwv[h];
            }
        } else {
            for (int i = lane; i < 2 * 81; i += 64) {
                const int pl = i / 81, k = i - pl * 81, rr = k / 9, c = k - rr * 9;
                s_cw[h][pl][rr * 12 + cxo[h] + c] = (pl ? P.ref[2] : P.ref[1])[(size_t)clip3(0, P.ch / 2 - 1, cy0[h] + rr) * cs + clip3(0, cs - 1, cx0[h] + c)];
            }
        }
    }
    __syncthreads();

    // ---- luma, one macroblock at a time: prediction (8.4.2.2.1, wave-uniform branches) -> residual -> fdct -> quant
    // -> dequant -> idct -> reconstruction ----
    int cbp_luma[2];
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const int fx = mvx[h] & 3, fy = mvy[h] & 3;
        const bool need_j = (fx == 2 && fy != 0) || (fy == 2 && fx != 0);
        if (need_j) {  // stage the unclipped horizontal sums of all 21 window rows (84 row segments)
            for (int i = lane; i < 84; i += 64) {
                const int rr = i >> 2, seg = (i & 3) * 4;
                *(uint2*)(s_b1[h] + rr * 16 + seg) = htap4_pk(s_w[h], rr * 28 + wxo[h] + seg);
            }
            __syncthreads();
        }
        uint32_t pred4;
        {
            const int g = (ly + 2) * 28 + wxo[h] + lx + 2;  // LDS offset of integer sample G(lx, ly)
            uint32_t Gv = 0, Bv = 0, Hv = 0, Jv = 0;
            const bool use_b = fx != 0 && fy != 2, use_h = fy != 0 && fx != 2;
            if (fx == 0 || fy == 0) Gv = lds_ld4(s_w[h], g + (fx == 3 ? 1 : 0) + (fy == 3 ? 28 : 0));
            if (use_b) Bv = round5_pk(htap4_pk(s_w[h], g - 2 + (fy == 3 ? 28 : 0)));
            if (use_h) {
                uint32_t c[6];
                const int o = g - 56 + (fx == 3 ? 1 : 0);
#pragma unroll
                for (int i = 0; i < 6; i++) c[i] = lds_ld4(s_w[h], o + i * 28);
                Hv = round5_pk(vtap4_pk(c));
            }
            if (need_j) {
                const int16_t* qq = s_b1[h] + ly * 16 + lx;  // row (ly-2)+2
                uint2 rw[6];
#pragma unroll
                for (int i = 0; i < 6; i++) rw[i] = *(const uint2*)(qq + i * 16);
                Jv = jtap4(rw);
            }
            if (fx == 0 && fy == 0) pred4 = Gv;
            else if (fy == 0) pred4 = fx == 2 ? Bv : avg4(Gv, Bv);
            else if (fx == 0) pred4 = fy == 2 ? Hv : avg4(Gv, Hv);
            else if (fx == 2 && fy == 2) pred4 = Jv;
            else if (fx == 2) pred4 = avg4(Bv, Jv);
            else if (fy == 2) pred4 = avg4(Hv, Jv);
            else pred4 = avg4(Bv, Hv);
        }
        int d[4] = {0, 0, 0, 0};
        int nz = 0;
        if (!zres[h]) {
#pragma unroll
        for (int k = 0; k < 4; k++) d[k] = byte_of(src4[h], k) - byte_of(pred4, k);
        fdct_quad(d, r);
        const int zz = c_zz_row[r];
        {
            const Quant& qn = P.qy;
            const int mf0 = (r & 1) ? qn.mf[2] : qn.mf[0], mf1 = (r & 1) ? qn.mf[1] : qn.mf[2];
            const int dq0 = (r & 1) ? qn.dq[2] : qn.dq[0], dq1 = (r & 1) ? qn.dq[1] : qn.dq[2];
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const int l = quant1(d[c], (c & 1) ? mf1 : mf0, qn.f_inter, qn.qbits);
                s_lv[h][LV_LUMA + blk * 16 + ((zz >> (4 * c)) & 15)] = (int16_t)l;
                nz += l != 0;
                d[c] = l * ((c & 1) ? dq1 : dq0);
            }
        }
        nz += __builtin_amdgcn_mov_dpp(nz, 0xB1, 0xf, 0xf, false);   // quad_perm [1,0,3,2]
        nz += __builtin_amdgcn_mov_dpp(nz, 0x4E, 0xf, 0xf, false);   // quad_perm [2,3,0,1]  -> TotalCoeff of the block in all 4 lanes
        idct_quad(d, r);
        }
        const unsigned long long ymask = __ballot(nz != 0);
        cbp_luma[h] = ((ymask & 0xFFFFull) ? 1 : 0) | (((ymask >> 16) & 0xFFFFull) ? 2 : 0) | (((ymask >> 32) & 0xFFFFull) ? 4 : 0) |
                      (((ymask >> 48) & 0xFFFFull) ? 8 : 0);
        if (h == 0 || two) {
            *(uint32_t*)(P.rec[0] + (size_t)(16 * my[h] + ly) * P.cw + 16 * mx[h] + lx) =
                pack4(clip255(byte_of(pred4, 0) + d[0]), clip255(byte_of(pred4, 1) + d[1]), clip255(byte_of(pred4, 2) + d[2]), clip255(byte_of(pred4, 3) + d[3]));
            if (r == 0) m[h]->tc[blk] = (uint8_t)nz;
        }
    }

    // ---- chroma of both macroblocks in one pass: bilinear MC, same chain with the DC terms through the 2x2 Hadamard ----
    const bool cact = hh == 0 || two;
    const int cmx = hh ? mx[1] : mx[0], cmy = hh ? my[1] : my[0];
    int cnz = 0, cdc = 0;
    uint32_t cpred4;
    int cd[4];
    {
        const int cmvx = hh ? mvx[1] : mvx[0], cmvy = hh ? mvy[1] : mvy[0];
        const int dx = cmvx & 7, dy = cmvy & 7;
        const uint8_t* cwp = s_cw[hh][cpl];
        const int o = cy * 12 + (hh ? cxo[1] : cxo[0]) + cx;
        const uint32_t A = lds_ld4(cwp, o), B = lds_ld4(cwp, o + 1), Cc = lds_ld4(cwp, o + 12), D = lds_ld4(cwp, o + 13);
        const int w00 = (8 - dx) * (8 - dy), w10 = dx * (8 - dy), w01 = (8 - dx) * dy, w11 = dx * dy;
        int pv[4];
#pragma unroll
        for (int k = 0; k < 4; k++) pv[k] = (w00 * byte_of(A, k) + w10 * byte_of(B, k) + w01 * byte_of(Cc, k) + w11 * byte_of(D, k) + 32) >> 6;
        cpred4 = pack4(pv[0], pv[1], pv[2], pv[3]);
#pragma unroll
        for (int k = 0; k < 4; k++) cd[k] = byte_of(csrc4, k) - pv[k];
    }
    const bool czero = zres[0] && (zres[1] || !two);   // both macroblocks: nothing to code, prediction = reconstruction
    int any_dc[2] = {0, 0};
    if (!czero) {
    fdct_quad(cd, r);
    cdc = cd[0];        // valid in lanes with r == 0
    const int zz = c_zz_row[r];
    {
        const Quant& qn = P.qc;
        const int mf0 = (r & 1) ? qn.mf[2] : qn.mf[0], mf1 = (r & 1) ? qn.mf[1] : qn.mf[2];
        const int dq0 = (r & 1) ? qn.dq[2] : qn.dq[0], dq1 = (r & 1) ? qn.dq[1] : qn.dq[2];
        int16_t* lvp = s_lv[hh] + LV_CHROMA_AC + (cpl * 4 + cblk) * 16;
#pragma unroll
        for (int c = 0; c < 4; c++) {
            int l = quant1(cd[c], (c & 1) ? mf1 : mf0, qn.f_inter, qn.qbits);
            if (r == 0 && c == 0) l = 0;  // DC goes through the 2x2 Hadamard
            lvp[(zz >> (4 * c)) & 15] = (int16_t)l;
            cnz += l != 0;
            cd[c] = l * ((c & 1) ? dq1 : dq0);
        }
    }
    cnz += __builtin_amdgcn_mov_dpp(cnz, 0xB1, 0xf, 0xf, false);
    cnz += __builtin_amdgcn_mov_dpp(cnz, 0x4E, 0xf, 0xf, false);
    {   // chroma DC: block DCs sit in lanes h*32 + pl*16 + blk*4 (r == 0); the 2x2 transforms run on the scalar unit
        int mydeq = 0;
#pragma unroll
        for (int h = 0; h < 2; h++)
#pragma unroll
            for (int pl = 0; pl < 2; pl++) {
                int dcw[4], lv[4], deq[4];
#pragma unroll
                for (int b = 0; b < 4; b++) dcw[b] = __builtin_amdgcn_readlane(cdc, h * 32 + pl * 16 + b * 4);
                chroma_dc(dcw, P.qc, P.qc.f_inter, lv, deq);
                any_dc[h] |= lv[0] | lv[1] | lv[2] | lv[3];
                if (lane == h * 32 + pl)
#pragma unroll
                    for (int i = 0; i < 4; i++) s_lv[h][LV_CHROMA_DC + pl * 4 + i] = (int16_t)lv[i];
                if (hh == h && cpl == pl) mydeq = cblk == 0 ? deq[0] : (cblk == 1 ? deq[1] : (cblk == 2 ? deq[2] : deq[3]));
            }
        if (r == 0) cd[0] = mydeq;
    }
    idct_quad(cd, r);
    } else {
#pragma unroll
        for (int k = 0; k < 4; k++) cd[k] = 0;
    }
    if (cact)
        *(uint32_t*)((cpl ? P.rec[2] : P.rec[1]) + (size_t)(8 * cmy + cy) * cs + 8 * cmx + cx) =
            pack4(clip255(byte_of(cpred4, 0) + cd[0]), clip255(byte_of(cpred4, 1) + cd[1]), clip255(byte_of(cpred4, 2) + cd[2]), clip255(byte_of(cpred4, 3) + cd[3]));
    const unsigned long long cmask = __ballot(cnz != 0);
    int cbp[2];
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const unsigned half = (unsigned)(cmask >> (32 * h));
        const int cbp_chroma = half ? 2 : (any_dc[h] ? 1 : 0);
        cbp[h] = cbp_luma[h] | (cbp_chroma << 4);
    }
    {
        const int mycbp = hh ? cbp[1] : cbp[0];
        if (cact && r == 0) (hh ? m[1] : m[0])->tc[16 + cpl * 4 + cblk] = (uint8_t)((mycbp >> 4) == 2 ? cnz : 0);
    }
#pragma unroll
    for (int h = 0; h < 2; h++) {
        if (lane == 32 * h && (h == 0 || two)) {
            MbInfo* mm = m[h];
            mm->type = (cbp[h] == 0 && skip[h].x == mvx[h] && skip[h].y == mvy[h]) ? MB_PSKIP : MB_P16;
            mm->i16_mode = 0; mm->chroma_mode = 0; mm->cbp = (uint8_t)cbp[h];
            P.mvd[2 * mbi[h]] = (int16_t)(mvx[h] - pred[h].x);
            P.mvd[2 * mbi[h] + 1] = (int16_t)(mvy[h] - pred[h].y);
        }
    }
    __syncthreads();
    // the two level blocks are consecutive in HBM (macroblocks 2q, 2q+1)
    const int nv = (two ? 2 : 1) * (LV_STRIDE * 2 / 16);
    for (int i = lane; i < nv; i += 64) ((uint4*)(P.levels + (size_t)(mb0 + 2 * q) * LV_STRIDE))[i] = ((const uint4*)&s_lv[0][0])[i];
}

}  // namespace h264
