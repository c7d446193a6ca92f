// media_amd/csrc/mc_filters.h -- register-level helpers shared by the motion search (k_me.h) and the intra kernel
// (k_intra.h): unaligned LDS reads, the packed 16-bit half-sample filters of 8.4.2.2.1, the quad-mapped 4x4 transforms
// and the chroma DC path of 8.5.11.
//
// SURVEY.md 8a rows a6.2 + a6.3 (inside ISVCEncoder::EncodeFrame,
// /root/reference/video_codec/VideoEncoderOpenH264.cpp:344).
#pragma once
#include "dev_common.h"

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

}  // namespace h264
