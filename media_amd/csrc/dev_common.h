// media_amd/csrc/dev_common.h -- shared device-side definitions of the MI355X
// (gfx950) H.264 encode path.  Wave = 64 lanes everywhere; every kernel in this
// directory launches 64-thread workgroups (one wavefront per macroblock) unless
// it says otherwise.
//
// What is restated here is the interior of ISVCEncoder::EncodeFrame, which the
// reference reaches at /root/reference/video_codec/VideoEncoderOpenH264.cpp:344
// (SURVEY.md 8a rows a6.1-a6.5).  Tables are ITU-T H.264 constants.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

// wave priorities of the entropy-coding kernels and of the motion search (row wavefronts run at 3, the transform at 2).
// A/B builds (media_amd/csrc/Makefile target `ab`) may swap them: -DAB_PRIO_EC=0 -DAB_PRIO_ME=1
#ifndef AB_PRIO_EC
#define AB_PRIO_EC 1
#endif
#ifndef AB_PRIO_ME
#define AB_PRIO_ME 0
#endif

namespace h264 {

enum { MB_I16 = 0, MB_P16 = 1, MB_PSKIP = 2, MB_IPCM = 3, MB_I4 = 4, MB_P16X8 = 5, MB_P8X16 = 6, MB_P8X8 = 7 };   // 5..7: two 16x8, two 8x16, four 8x8 partitions
__device__ __forceinline__ bool mb_is_intra(int type) { return type == MB_I16 || type == MB_IPCM || type == MB_I4; }
// A.3.1: macroblock_layer() of a CAVLC macroblock may not exceed 128 + 3072 bits; a macroblock whose BOUND (below) does
// is coded as I_PCM.  INTRA_TEST_MIN: motion cost from which a P macroblock is also costed as Intra16x16.
enum { MB_BITS_LIMIT = 3200, MB_HEADER_BOUND = 96, INTRA_TEST_MIN = 2000, PART_TEST_MIN = 2000 };   // PART_TEST_MIN: 16x16 motion cost from which the partitions are tried
enum { LV_LUMA_DC = 0, LV_LUMA = 16, LV_CHROMA_DC = 272, LV_CHROMA_AC = 280, LV_STRIDE = 416 };

// 32 bytes; identical to the debug layout documented in include/mi355x_h264.h
struct MbInfo {
    int16_t mvx, mvy;
    uint8_t type, i16_mode, chroma_mode, cbp;
    uint8_t tc[24];  // TotalCoeff: 16 luma (blkIdx order), 4 Cb, 4 Cr
};
static_assert(sizeof(MbInfo) == 32, "MbInfo layout");

// forward quantiser constants for one QP (host-prepared)
struct Quant {
    int qbits;          // 15 + qp/6
    int f_intra, f_inter;
    int mf[3];          // multiplier per position class
    int dq[3];          // dequant v[class] << (qp/6)
    int thr_inter[3];   // smallest |w| whose inter-rounded level is non-zero
    int thr_dc_inter;   // same for the chroma DC path (2x2 Hadamard output)
    int qp;
    int mf8[6], ls8[6]; // 8x8 transform (High profile): forward multipliers and 16 * normAdjust8x8 by position class (8.5.9)
};

// everything a picture's QP fixes, prepared once per encoder for every QP (the host's fill_quant): indirect launches read it
struct QpEntry { Quant qy, qc; int lambda, sad_nz; };

// Slices are bands of `rows` whole macroblock rows (the last band may be shorter); with one slice rows = mbh.
// row_in_slice() is my % rows through a 32-bit reciprocal (inv = floor(2^32 / rows) + 1, exact for my < 65536):
// two scalar multiplies where a runtime modulo would be a float division sequence.
struct SliceRows {
    int rows;
    unsigned inv;
    __device__ __forceinline__ int row_in_slice(int my) const { return my - (int)__umulhi((unsigned)my, inv) * rows; }
    // 6.4.4: the macroblocks above belong to another slice (or lie outside the picture) on a slice's first row
    __device__ __forceinline__ bool has_top(int my) const { return row_in_slice(my) != 0; }
};

// macroblock index -> (mx, my) without a runtime division (a float reciprocal sequence of ~20 instructions, per lane in the
// entropy kernels): q = mulhi(i, inv) with inv = floor(2^32 / mbw) + 1 is exact for i * mbw < 2^32 (i < 65 536, mbw <= 256);
// mbw = 1 has no 32-bit inv and is flagged by inv = 0.
struct MbDiv {
    unsigned inv;
    __device__ __forceinline__ int row(int i) const { return inv ? (int)__umulhi((unsigned)i, inv) : i; }
};

// Band of the picture this encoder instance works on (slice bands of one picture on several GPUs):
// macroblock rows row0 .. row0 + rows - 1, always whole slices.  One instance alone: row0 = 0, rows = mbh.
struct Band { int row0, rows; };

struct FrameParams {
    const uint8_t* src;  // tight picture in HBM: Y (w*h), then U, V planes (I420) or one interleaved UV plane (NV12)
    int src_nv12;        // 1: chroma is read straight from the interleaved plane (no conversion pass)
    int w, h;            // display size
    int cw, ch, mbw, mbh;
    uint8_t* rec[3];        // current picture reconstruction (pitch cw, cw/2, cw/2)
    const uint8_t* ref[3];  // previous deblocked picture (= refs[0])
    const uint8_t* refs[3][3];   // the reference pictures, ref_idx_l0 order (newest first); entries >= nref are not read
    int nref;               // reference pictures available to this picture (1 .. config.refs)
    int rf, rf_last;        // k_me runs once per reference picture: ref_idx_l0 of this launch (ref = its planes), and of the last launch
    uint32_t* me_total;     // per macroblock: best motion cost + lambda * bits(ref_idx_l0) so far, 0 = settled without a search
    int* pmv;               // per macroblock: the previous picture's vector (rate predictor), parked by the first launch
    MbInfo* mb;
    int16_t* levels;     // LV_STRIDE int16 per macroblock
    int16_t* mvd;        // 8 int16 per macroblock: (mv - predictor) of its partitions, in coding order
    int16_t* mvq;        // 8 int16 per macroblock: the vectors (x, y) of its four 8x8 quadrants (a 16x16 macroblock carries its vector four times)
    uint8_t* aux;        // 16 bytes per macroblock: Intra4x4PredMode of the 16 blocks (blkIdx order) of an MB_I4 macroblock
    uint16_t* me_cost;   // per macroblock: min(final motion cost, 16383), 0 where the zero-motion test hit (summed by k_bit_scan)
    Quant qy, qc;        // luma / chroma quantisers
    int lambda;
    int sad_nz;          // a luma SAD of this much or more cannot quantise to nothing (fill in submit(): exact bound)
    int search;          // config.search: 0 exhaustive integer search, 1 seeded by the previous picture's vector (k_me section 1b)
    // lockstep batch (gridDim.y = number of independent closed GOPs / streams encoded together):
    // element strides between consecutive batch items
    size_t st_src;       // bytes between the source pictures of two batch items
    size_t st_y, st_c;   // bytes between reconstruction planes (luma, chroma)
    int st_mb;           // macroblocks per batch item (MbInfo / levels / mvd arrays)
    // per batch item, value = pic_serial when raised by a kernel of this picture (never cleared: the serial changes):
    unsigned* anypcm;    //   the picture holds an I_PCM macroblock (it is then not loop-filtered: slice header idc 1)
    unsigned* anyintra;  //   P picture: the motion search handed macroblocks to the intra pass (k_pintra_rows, bS 3 / 4 edges)
    unsigned pic_serial;
    SliceRows sl;        // slices of the picture: bands of sl.rows macroblock rows (sl.rows = mbh: one slice)
    Band band;           // the rows this instance encodes; grids cover the band, coordinates stay those of the picture
    MbDiv mbdiv;         // macroblock index / mbw
    // decoder peer only (the encoder's pictures have one QP: qy, qc above; it leaves these 0):
    const uint8_t* mbqp; // QP_Y of every macroblock (7.4.5: slice_qp_delta, mb_qp_delta); 0 for I_PCM
    int cqo_cb, cqo_cr;  // chroma_qp_index_offset, second_chroma_qp_index_offset
    const int16_t* mv4;  // 32 int16 per macroblock: the vector (x, y) of every 4x4 block, raster order (sub-macroblock partitions)
    const uint8_t* refq; // 4 per macroblock: ref_idx_l0 of the four 8x8 quadrants
    const uint8_t* mbavail;   // per macroblock: neighbours (left, above, above-right, above-left) in this slice: bits 0..3; usable for intra prediction: bits 4..7
    // INDIRECT launches (kernels instantiated with IND = true; the stream hub of mi355x_h264.hip: pictures of DIFFERENT streams in
    // one lockstep step).  gridDim.y counts POSITIONS; itemtab[position] names the batch item, the ring slot its picture is
    // reconstructed into and its QP.  rec[] then holds the BASE of the reconstruction planes, which lie [item][ring slot]
    // (st_y / st_c bytes between items, st_ring_y / st_ring_c between the nbuf slots of one item); qtab is the table of
    // quantiser constants by QP.  Direct launches leave itemtab null and never read these.
    const uint32_t* itemtab;
    const QpEntry* qtab;
    size_t st_ring_y, st_ring_c;
    int nbuf;
};

// itemtab word: bits 0..7 batch item, 8..9 ring slot of the picture being coded, 16..21 QP
struct ItemRef { int item, cur, qp; };
__device__ __forceinline__ ItemRef item_ref(const uint32_t* itemtab, int pos)
{
    const uint32_t w = (uint32_t)__builtin_amdgcn_readfirstlane((int)itemtab[pos]);   // uniform: a scalar load
    return ItemRef{(int)(w & 0xFFu), (int)((w >> 8) & 3u), (int)((w >> 16) & 63u)};
}
// the parameter block of batch item g (pointers advanced by g strides)
__device__ __forceinline__ FrameParams batch_view(FrameParams P, int g)
{
    P.src += (size_t)g * P.st_src;
    P.rec[0] += (size_t)g * P.st_y; P.rec[1] += (size_t)g * P.st_c; P.rec[2] += (size_t)g * P.st_c;
    P.ref[0] += (size_t)g * P.st_y; P.ref[1] += (size_t)g * P.st_c; P.ref[2] += (size_t)g * P.st_c;
#pragma unroll
    for (int r = 0; r < 3; r++) { P.refs[r][0] += (size_t)g * P.st_y; P.refs[r][1] += (size_t)g * P.st_c; P.refs[r][2] += (size_t)g * P.st_c; }
    P.mb += (size_t)g * P.st_mb;
    P.levels += (size_t)g * P.st_mb * LV_STRIDE;
    P.mvd += (size_t)g * P.st_mb * 8;
    P.mvq += (size_t)g * P.st_mb * 8;
    P.aux += (size_t)g * P.st_mb * 16;
    P.me_cost += (size_t)g * P.st_mb;
    P.anypcm += g; P.anyintra += g;
    P.me_total += (size_t)g * P.st_mb; P.pmv += (size_t)g * P.st_mb;
    return P;
}
// IND = true: the view of POSITION pos of an indirect launch - the item's arrays, its own ring slots and its own QP's constants
template <bool IND>
__device__ __forceinline__ FrameParams batch_view(FrameParams P, int pos)
{
    if constexpr (!IND) return batch_view(P, pos);
    else {
        const ItemRef it = item_ref(P.itemtab, pos);
        uint8_t* const by = P.rec[0] + (size_t)it.item * P.st_y;
        uint8_t* const bu = P.rec[1] + (size_t)it.item * P.st_c;
        uint8_t* const bv = P.rec[2] + (size_t)it.item * P.st_c;
        P.rec[0] = by + (size_t)it.cur * P.st_ring_y; P.rec[1] = bu + (size_t)it.cur * P.st_ring_c; P.rec[2] = bv + (size_t)it.cur * P.st_ring_c;
        int rs = it.cur;
#pragma unroll
        for (int r = 0; r < 3; r++) {   // ref_idx_l0 r: the slot written r + 1 pictures ago
            rs = rs == 0 ? P.nbuf - 1 : rs - 1;
            P.refs[r][0] = by + (size_t)rs * P.st_ring_y; P.refs[r][1] = bu + (size_t)rs * P.st_ring_c; P.refs[r][2] = bv + (size_t)rs * P.st_ring_c;
        }
        P.ref[0] = P.refs[0][0]; P.ref[1] = P.refs[0][1]; P.ref[2] = P.refs[0][2];
        const int g = it.item;
        P.src += (size_t)g * P.st_src;
        P.mb += (size_t)g * P.st_mb;
        P.levels += (size_t)g * P.st_mb * LV_STRIDE;
        P.mvd += (size_t)g * P.st_mb * 8;
        P.mvq += (size_t)g * P.st_mb * 8;
        P.aux += (size_t)g * P.st_mb * 16;
        P.me_cost += (size_t)g * P.st_mb;
        P.anypcm += g; P.anyintra += g;
        P.me_total += (size_t)g * P.st_mb; P.pmv += (size_t)g * P.st_mb;
        const QpEntry* q = P.qtab + it.qp;
        P.qy = q->qy; P.qc = q->qc; P.lambda = q->lambda; P.sad_nz = q->sad_nz;
        return P;
    }
}
// the batch item a position of the grid stands for (per-item scratch outside FrameParams: hand-off granules, flags)
template <bool IND>
__device__ __forceinline__ int batch_item(const uint32_t* itemtab, int pos)
{
    if constexpr (!IND) return pos;
    else return item_ref(itemtab, pos).item;
}

// Synchronisation inside a ONE-WAVE workgroup.  LDS instructions of one wave execute in program order,
// so a ds_write is visible to a later ds_read of any lane of the same wave; all that is needed is to stop
// the compiler from reordering them.  Unlike __syncthreads() this emits no s_waitcnt vmcnt(0), so global
// loads requested ahead of time (prefetch of the next macroblock) stay in flight across it.
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// compile-time loop: f(std::integral_constant<int, I>) for I = A .. B - 1
template <int A, int B, class F>
__device__ __forceinline__ void static_for(F&& f)
{
    if constexpr (A < B) {
        f(std::integral_constant<int, A>{});
        static_for<A + 1, B>(f);
    }
}

__device__ __forceinline__ int clip3(int lo, int hi, int v) { return v < lo ? lo : (v > hi ? hi : v); }
__device__ __forceinline__ int clip255(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }
__device__ __forceinline__ int iabs(int v) { return v < 0 ? -v : v; }

// position class of raster index i in a 4x4 block: 0 (even,even) 1 (odd,odd) 2 otherwise
__device__ __forceinline__ int pos_class(int i)
{
    const int x = i & 1, y = (i >> 2) & 1;
    return (x & y) ? 1 : ((x | y) ? 2 : 0);
}

// 6.4.3: blkIdx <-> 4x4 raster position
__device__ __forceinline__ int blk_x(int b) { return (b & 1) | ((b >> 1) & 2); }
__device__ __forceinline__ int blk_y(int b) { return ((b >> 1) & 1) | ((b >> 2) & 2); }
__device__ __forceinline__ constexpr int xy2blk(int x, int y) { return (x & 1) | ((y & 1) << 1) | ((x & 2) << 1) | ((y & 2) << 2); }

__constant__ const uint8_t c_zigzag[16] = {0, 1, 4, 8, 5, 2, 3, 6, 9, 12, 13, 10, 7, 11, 14, 15};
// raster position -> zig-zag index
__constant__ const uint8_t c_zigzag_inv[16] = {0, 1, 5, 6, 2, 4, 7, 12, 3, 8, 11, 13, 9, 10, 14, 15};

// length in bits of se(v)
__device__ __forceinline__ int se_len(int v)
{
    const unsigned k1 = v > 0 ? 2u * (unsigned)v : 1u - 2u * (unsigned)v;  // codeNum + 1 (>= 1, so no branch for v == 0)
    return 63 - 2 * __builtin_clz(k1);
}

// forward 4x4 core transform, in place on 16 ints (raster)
__device__ __forceinline__ void fdct4x4(int d[16])
{
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int a = d[4 * i], b = d[4 * i + 1], c = d[4 * i + 2], e = d[4 * i + 3];
        const int s0 = a + e, s1 = b + c, d0 = a - e, d1 = b - c;
        d[4 * i] = s0 + s1; d[4 * i + 1] = 2 * d0 + d1; d[4 * i + 2] = s0 - s1; d[4 * i + 3] = d0 - 2 * d1;
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int a = d[j], b = d[4 + j], c = d[8 + j], e = d[12 + j];
        const int s0 = a + e, s1 = b + c, d0 = a - e, d1 = b - c;
        d[j] = s0 + s1; d[4 + j] = 2 * d0 + d1; d[8 + j] = s0 - s1; d[12 + j] = d0 - 2 * d1;
    }
}

// 8.5.12.2 inverse transform; in: scaled coefficients, out: residual (rounded >>6)
__device__ __forceinline__ void idct4x4(int d[16])
{
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int d0 = d[4 * i], d1 = d[4 * i + 1], d2 = d[4 * i + 2], d3 = d[4 * i + 3];
        const int e0 = d0 + d2, e1 = d0 - d2, e2 = (d1 >> 1) - d3, e3 = d1 + (d3 >> 1);
        d[4 * i] = e0 + e3; d[4 * i + 1] = e1 + e2; d[4 * i + 2] = e1 - e2; d[4 * i + 3] = e0 - e3;
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int f0 = d[j], f1 = d[4 + j], f2 = d[8 + j], f3 = d[12 + j];
        const int g0 = f0 + f2, g1 = f0 - f2, g2 = (f1 >> 1) - f3, g3 = f1 + (f3 >> 1);
        d[j] = (g0 + g3 + 32) >> 6; d[4 + j] = (g1 + g2 + 32) >> 6;
        d[8 + j] = (g1 - g2 + 32) >> 6; d[12 + j] = (g0 - g3 + 32) >> 6;
    }
}

// sum |H4 D H4^T| of 16 differences (no normalisation)
__device__ __forceinline__ int hadamard_abs(int d[16])
{
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int s0 = d[4 * i] + d[4 * i + 3], s1 = d[4 * i + 1] + d[4 * i + 2];
        const int d0 = d[4 * i] - d[4 * i + 3], d1 = d[4 * i + 1] - d[4 * i + 2];
        d[4 * i] = s0 + s1; d[4 * i + 1] = d0 + d1; d[4 * i + 2] = s0 - s1; d[4 * i + 3] = d0 - d1;
    }
    int s = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int s0 = d[j] + d[12 + j], s1 = d[4 + j] + d[8 + j];
        const int d0 = d[j] - d[12 + j], d1 = d[4 + j] - d[8 + j];
        s += iabs(s0 + s1) + iabs(d0 + d1) + iabs(s0 - s1) + iabs(d0 - d1);
    }
    return s;
}

// symmetric 4x4 Hadamard (rows 1111 / 11-1-1 / 1-1-11 / 1-11-1), in place
__device__ __forceinline__ void hadamard4x4(int d[16])
{
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int a = d[4 * i], b = d[4 * i + 1], c = d[4 * i + 2], e = d[4 * i + 3];
        d[4 * i] = a + b + c + e; d[4 * i + 1] = a + b - c - e; d[4 * i + 2] = a - b - c + e; d[4 * i + 3] = a - b + c - e;
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int a = d[j], b = d[4 + j], c = d[8 + j], e = d[12 + j];
        d[j] = a + b + c + e; d[4 + j] = a + b - c - e; d[8 + j] = a - b - c + e; d[12 + j] = a - b + c - e;
    }
}

__device__ __forceinline__ int quant1(int w, int mf, int f, int qbits)
{
    const int a = iabs(w);
    const int l = (int)(((unsigned)a * (unsigned)mf + (unsigned)f) >> qbits);
    return w < 0 ? -l : l;
}
// keep a wave-uniform value in a VGPR: VOP2 with an SGPR operand issues at half the rate of the all-VGPR form
__device__ __forceinline__ int vreg(int x)
{
    asm volatile("" : "+v"(x));
    return x;
}
// sign(w) * ((|w| * mf + f) >> q) without the absolute value: for w < 0, -floor((|w| mf + f) / 2^q) = floor((w mf + 2^q - 1 - f) / 2^q)
__device__ __forceinline__ int quant_signed(int w, int mf, int f, int c, int q)
{
    const int s = w >> 31;
    return (__mul24(w, mf) + (f + (s & c))) >> q;
}

// ---- I_PCM fallback: upper bound on the CAVLC bits of one residual block (oracle/h264_enc.c blk_bits_bound states the
// derivation): tc levels, `sum16` = sum over all 16 positions of max(min(|level|, 27), h) with h = smax + 1, smax from the
// bitwise OR of the magnitudes ----
__device__ __forceinline__ int pcm_smax(unsigned orv)
{
    const int b = 32 - __clz((int)orv);   // orv > 0
    return b <= 2 ? 1 : (b < 6 ? b : 6);
}
__device__ __forceinline__ int pcm_blk_tail(int tc) { return 12 + 16 + min(2 * tc + 22, 73 - 4 * tc); }
// plain form over n int16 levels (the intra kernel, one block per lane; not a hot path)
__device__ __forceinline__ int blk_bits_bound(const int16_t* lv, int n)
{
    int tc = 0;
    unsigned orv = 0;
    for (int i = 0; i < n; i++) { const int a = iabs((int)lv[i]); tc += a != 0; orv |= (unsigned)a; }
    if (!tc) return 6;
    const int h = pcm_smax(orv) + 1;
    int sum = 0;
    for (int i = 0; i < n; i++) { const int a = iabs((int)lv[i]); if (a) sum += max(min(a, 27), h) + 1; }
    return sum + pcm_blk_tail(tc);
}
// packed form: lvp[8] = 16 int16 levels (two per word), tc known
__device__ __forceinline__ int blk_bits_bound_packed(const uint32_t lvp[8], int tc)
{
    typedef short s2 __attribute__((ext_vector_type(2)));
    typedef unsigned short u2 __attribute__((ext_vector_type(2)));
    uint32_t a[8], orv = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const s2 p = __builtin_bit_cast(s2, lvp[k]);
        a[k] = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(p, -p));
        orv |= a[k];
    }
    orv = (orv | (orv >> 16)) & 0xFFFFu;
    const int h = pcm_smax(orv | 1u) + 1;
    const u2 hh = {(unsigned short)h, (unsigned short)h}, c27 = {27, 27};
    uint32_t sum = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const u2 m = __builtin_elementwise_max(__builtin_elementwise_min(__builtin_bit_cast(u2, a[k]), c27), hh);
        sum = __builtin_amdgcn_sad_u16(__builtin_bit_cast(uint32_t, m), 0u, sum);
    }
    // the 16 - tc zero positions were counted as h each, the non-zero ones lack their + 1
    return tc ? (int)sum - (16 - tc) * h + tc + pcm_blk_tail(tc) : 6;
}

// number of non-zero levels among 16 int16 (two per word)
__device__ __forceinline__ int count_nz16_packed(const uint32_t lvp[8])
{
    typedef short s2 __attribute__((ext_vector_type(2)));
    typedef unsigned short u2 __attribute__((ext_vector_type(2)));
    const u2 one = {1, 1};
    uint32_t n = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const s2 p = __builtin_bit_cast(s2, lvp[k]);
        const u2 a = __builtin_bit_cast(u2, __builtin_elementwise_max(p, -p));
        n = __builtin_amdgcn_sad_u16(__builtin_bit_cast(uint32_t, __builtin_elementwise_min(a, one)), 0u, n);
    }
    return (int)n;
}

// wave-wide reductions over 64 lanes
__device__ __forceinline__ unsigned wave_min_u32(unsigned v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned t = (unsigned)__shfl_xor((int)v, o);
        v = t < v ? t : v;
    }
    return v;
}
// sum across aligned groups of `width` lanes (power of two <= 64)
template <int WIDTH>
__device__ __forceinline__ int group_sum(int v)
{
#pragma unroll
    for (int o = WIDTH / 2; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// The same reductions with DPP moves instead of ds_bpermute round trips: ~8 cycles instead of ~120 per step for a
// wave that runs alone on its SIMD (the row-wavefront kernels), and no address arithmetic for the VALU-bound ones.
// Mirrors are as good as xor exchanges for a commutative reduction.
__device__ __forceinline__ int row_sum16_dpp(int v)   // sum over aligned groups of 16 lanes, in every lane
{
    v += __builtin_amdgcn_mov_dpp(v, 0xB1, 0xf, 0xf, false);    // quad_perm [1,0,3,2]
    v += __builtin_amdgcn_mov_dpp(v, 0x4E, 0xf, 0xf, false);    // quad_perm [2,3,0,1]
    v += __builtin_amdgcn_mov_dpp(v, 0x141, 0xf, 0xf, false);   // row_half_mirror
    v += __builtin_amdgcn_mov_dpp(v, 0x140, 0xf, 0xf, false);   // row_mirror
    return v;
}
__device__ __forceinline__ int group_sum8_dpp(int v)   // sum over aligned groups of 8 lanes, in every lane
{
    v += __builtin_amdgcn_mov_dpp(v, 0xB1, 0xf, 0xf, false);    // quad_perm [1,0,3,2]
    v += __builtin_amdgcn_mov_dpp(v, 0x4E, 0xf, 0xf, false);    // quad_perm [2,3,0,1]
    v += __builtin_amdgcn_mov_dpp(v, 0x141, 0xf, 0xf, false);   // row_half_mirror
    return v;
}
__device__ __forceinline__ int wave_sum_dpp(int v)   // sum over the 64 lanes, wave-uniform result
{
    v = row_sum16_dpp(v);
    return __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) + __builtin_amdgcn_readlane(v, 32) + __builtin_amdgcn_readlane(v, 48);
}
__device__ __forceinline__ unsigned wave_min_u32_dpp(unsigned v)   // minimum over the 64 lanes, wave-uniform result
{
    auto step = [](unsigned x, unsigned t) { return t < x ? t : x; };
    v = step(v, (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xf, 0xf, false));
    v = step(v, (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xf, 0xf, false));
    v = step(v, (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0x141, 0xf, 0xf, false));
    v = step(v, (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0x140, 0xf, 0xf, false));
    const unsigned a = (unsigned)__builtin_amdgcn_readlane((int)v, 0), b = (unsigned)__builtin_amdgcn_readlane((int)v, 16);
    const unsigned c = (unsigned)__builtin_amdgcn_readlane((int)v, 32), d = (unsigned)__builtin_amdgcn_readlane((int)v, 48);
    return step(step(a, b), step(c, d));
}

// clamped fetch of one source sample (the coded picture extends the display
// picture by replicating its last column / row)
__device__ __forceinline__ int src_px(const uint8_t* plane, int pw, int ph, int x, int y)
{
    return plane[(size_t)(y < ph ? y : ph - 1) * pw + (x < pw ? x : pw - 1)];
}

// The chroma reconstruction plane of component pl (0 Cb, 1 Cr), pl differing between lanes.  Written as a select between
// P.rec[1] and P.rec[2] it compiles to an indexed load of P.rec[], and that moves the whole parameter block into scratch memory
// (every later use of P then costs a scratch load); the two pointers therefore pass through an opaque no-op first.
__device__ __forceinline__ uint8_t* rec_chroma(const FrameParams& P, int pl)
{
    uint8_t *a = P.rec[1], *b = P.rec[2];
    asm volatile("" : "+s"(a), "+s"(b));
    return pl ? b : a;
}

// four source chroma samples (plane pl, columns gx..gx+3 of row gy) packed into one word, clamped like src_px.
// NV12: one 8-byte read of the interleaved row, even (Cb) or odd (Cr) bytes picked by v_perm_b32.
__device__ __forceinline__ uint32_t src_chroma4(const FrameParams& P, int pl, int gx, int gy)
{
    const int pw = P.w / 2, ph = P.h / 2;
    const int yy = gy < ph ? gy : ph - 1;
    const uint8_t* base = P.src + (size_t)P.w * P.h;
    if (P.src_nv12) {
        const uint8_t* row = base + (size_t)yy * (2 * pw);
        const uint8_t* p = row + 2 * gx;
        if (gx + 3 < pw && (((uintptr_t)p) & 7) == 0) {
            const uint2 v = *(const uint2*)p;
            return __builtin_amdgcn_perm(v.y, v.x, pl ? 0x07050301u : 0x06040200u);
        }
        uint32_t v = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) v |= (uint32_t)row[2 * (gx + k < pw ? gx + k : pw - 1) + pl] << (8 * k);
        return v;
    }
    const uint8_t* C = base + (pl ? (size_t)pw * ph : 0);
    const uint8_t* p = C + (size_t)yy * pw + gx;
    if (gx + 3 < pw && (((uintptr_t)p) & 3) == 0) return *(const uint32_t*)p;
    uint32_t v = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) v |= (uint32_t)src_px(C, pw, ph, gx + k, gy) << (8 * k);
    return v;
}

// ---- decoder peer: the scaling constants of a macroblock's own QP (the encoder has them host-prepared in Quant) ----
__constant__ const uint8_t c_dequant_v[6][3] = {{10, 16, 13}, {11, 18, 14}, {13, 20, 16}, {14, 23, 18}, {16, 25, 20}, {18, 29, 23}};   // Table in 8.5.9 by position class
__constant__ const uint8_t c_dequant8_v[6][6] = {{20, 18, 32, 19, 25, 24}, {22, 19, 35, 21, 28, 26}, {26, 23, 42, 24, 33, 31},
                                                 {28, 25, 45, 26, 35, 33}, {32, 28, 51, 30, 40, 38}, {36, 32, 58, 34, 46, 43}};
__constant__ const uint8_t c_chroma_qp[52] = {0,  1,  2,  3,  4,  5,  6,  7,  8,  9,  10, 11, 12, 13, 14, 15, 16, 17,
                                              18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 29, 30, 31, 32, 32, 33,
                                              34, 34, 35, 35, 36, 36, 37, 37, 37, 38, 38, 38, 39, 39, 39, 39};   // Table 8-15
// dq[class] = v << (qp / 6), the form Quant::dq has
__device__ __forceinline__ void dec_dq(int qp, int dq[3])
{
    const int m = qp % 6, s = qp / 6;
#pragma unroll
    for (int c = 0; c < 3; c++) dq[c] = (int)c_dequant_v[m][c] << s;
}
// QP_C of chroma component pl (0 Cb, 1 Cr) of a macroblock with luma QP qp (8.5.8 -> Table 8-15 at the offset index)
__device__ __forceinline__ int dec_qpc(const FrameParams& P, int qp, int pl)
{
    return (int)c_chroma_qp[clip3(0, 51, qp + (pl ? P.cqo_cr : P.cqo_cb))];
}

// Load the source macroblock (mx,my) into LDS: y[256] (pitch 16), c[128] (Cb 8x8 then Cr 8x8).
__device__ __forceinline__ void load_src_mb(const FrameParams& P, int mx, int my, uint8_t* sy, uint8_t* sc, int lane)
{
    const uint8_t* Y = P.src;
    {
        const int row = lane >> 2, xs = (lane & 3) * 4;
        const int gy = 16 * my + row, gx = 16 * mx + xs;
        const int yy = gy < P.h ? gy : P.h - 1;
        const uint8_t* p = Y + (size_t)yy * P.w + gx;
        uint32_t v;
        if (gx + 3 < P.w && (((uintptr_t)p) & 3) == 0) v = *(const uint32_t*)p;
        else {
            v = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) v |= (uint32_t)src_px(Y, P.w, P.h, gx + k, gy) << (8 * k);
        }
        *(uint32_t*)(sy + row * 16 + xs) = v;
    }
    if (lane < 32) {
        const int pl = lane >> 4, row = (lane >> 1) & 7, xs = (lane & 1) * 4;
        const uint32_t v = src_chroma4(P, pl, 8 * mx + xs, 8 * my + row);
        *(uint32_t*)(sc + pl * 64 + row * 8 + xs) = v;
    }
}

// XCD-aware block -> macroblock map.  Workgroups are dealt round-robin over the 8 XCDs (blocks b and
// b+8 share an L2), so block b works on macroblock (b % 8) * (n/8) + b/8: each XCD's L2 then sees one
// contiguous band of macroblock rows and overlapping reference windows hit in L2 instead of being
// fetched once per XCD.  Placement only affects speed, never results.
__device__ __forceinline__ int xcd_mb_index(int b, int n)
{
    const int n8 = n >> 3;
    return b < 8 * n8 ? (b & 7) * n8 + (b >> 3) : b;
}

// 8.4.1.3 motion vector prediction for a 16x16 partition with one reference
// frame.  All macroblocks of the picture already carry their final vectors, so
// a neighbour is "available" when it lies inside the picture and in the same slice (P.sl).
struct Mv { int x, y; };
__device__ __forceinline__ int med3(int a, int b, int c)
{
    const int mn = min(a, min(b, c)), mxv = max(a, max(b, c));
    return a + b + c - mn - mxv;
}
// returns predictor; skip receives the P_Skip vector.  The four candidate neighbours (A left, B top,
// C top-right, D top-left) are fetched with independent 8-byte loads issued together (one memory round
// trip), availability is applied afterwards.  Neighbour `type` fields of a P picture are never MB_I16 in
// this build, but the general rule is kept.
__device__ __forceinline__ Mv predict_mv(const FrameParams& P, int mx, int my, Mv& skip)
{
    const bool top = P.sl.has_top(my);   // neighbours above exist and lie in the same slice
    const bool avA = mx > 0, avB = top, avC0 = top && mx + 1 < P.mbw, avD = mx > 0 && top;
    const MbInfo* base = P.mb + (size_t)my * P.mbw + mx;
    // every lane loads the same words: hand them to the scalar unit, which then does the whole prediction
    auto uni = [](const uint2 v) { return make_uint2((uint32_t)__builtin_amdgcn_readfirstlane((int)v.x), (uint32_t)__builtin_amdgcn_readfirstlane((int)v.y)); };
    const uint2 wA = uni(*(const uint2*)(avA ? base - 1 : base));
    const uint2 wB = uni(*(const uint2*)(avB ? base - P.mbw : base));
    const uint2 wC = uni(*(const uint2*)(avC0 ? base - P.mbw + 1 : base));
    const uint2 wD = uni(*(const uint2*)(avD ? base - P.mbw - 1 : base));
    auto unpack = [](const uint2 w, bool av, int& ref, Mv& mv) {
        const int type = (int)(w.y & 255);
        ref = -1; mv.x = 0; mv.y = 0;
        if (av && !mb_is_intra(type)) { ref = 0; mv.x = (int)(int16_t)(w.x & 0xFFFF); mv.y = (int)(int16_t)(w.x >> 16); }
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
    if (zero_skip) { skip.x = 0; skip.y = 0; } else skip = p;
    return p;
}

}  // namespace h264
