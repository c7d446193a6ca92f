// media_amd/csrc/k_cavlc.h -- CAVLC entropy coding on the device (H.264 7.3.5, 9.2).
//
// SURVEY.md 8a row a6.5 (inside ISVCEncoder::EncodeFrame,
// /root/reference/video_codec/VideoEncoderOpenH264.cpp:344).
//
// Every syntax element of a macroblock depends only on data that is final once
// the reconstruction kernels have run (levels, TotalCoeff of neighbours, motion
// vector differences, skip flags), so the slice is coded in three launches:
//   k_cavlc<false>  one lane per (macroblock, block slot): codes the slot once, keeps its
//                   bit LENGTH and - when they fit 64 - the bits themselves
//   k_bit_scan      exclusive prefix sum of macroblock lengths -> bit offsets,
//                   slice header, trailing mb_skip_run and rbsp stop bit
//   k_cavlc<true>   places every stored slot at its final bit position (slots longer
//                   than 64 bits are coded again, straight into the buffer)
//   k_pack          device bit buffer -> pinned host access unit, counting the
//                   byte patterns that need emulation prevention (7.4.1)
#pragma once
#include "dev_common.h"
#include "h264_vlc_tables.h"

namespace h264 {

// ---- Table 9-5 .. 9-10 (ITU-T H.264): initialisers in h264_vlc_tables.h ----
__constant__ const uint8_t c_ct_len[4][68] = H264_TAB_CT_LEN;
__constant__ const uint8_t c_ct_bits[4][68] = H264_TAB_CT_BITS;
__constant__ const uint8_t c_cdc_len[20] = H264_TAB_CDC_LEN;
__constant__ const uint8_t c_cdc_bits[20] = H264_TAB_CDC_BITS;
__constant__ const uint8_t c_tz_len[15][16] = H264_TAB_TZ_LEN;
__constant__ const uint8_t c_tz_bits[15][16] = H264_TAB_TZ_BITS;
__constant__ const uint8_t c_ctz_len[3][4] = H264_TAB_CTZ_LEN;
__constant__ const uint8_t c_ctz_bits[3][4] = H264_TAB_CTZ_BITS;
__constant__ const uint8_t c_run_len[7][16] = H264_TAB_RUN_LEN;
__constant__ const uint8_t c_run_bits[7][16] = H264_TAB_RUN_BITS;
// Table 9-4, inverted for the encoder: coded_block_pattern -> codeNum (inter)
__constant__ const uint8_t c_cbp2code_inter[48] = H264_TAB_CBP2CODE_INTER;

// Table 9-4, inverted: coded_block_pattern -> codeNum (Intra4x4)
__constant__ const uint8_t c_cbp2code_intra[48] = H264_TAB_CBP2CODE_INTRA;

// ---- bit sinks ----
struct BitCount {
    unsigned n;
    __device__ __forceinline__ void init(unsigned) { n = 0; }
    __device__ __forceinline__ void put(int len, unsigned) { n += (unsigned)len; }
    __device__ __forceinline__ void flush() {}
};
// counts like BitCount and keeps the first 64 bits, left aligned: slots that fit (nearly all of them at usual
// QPs) are coded ONCE, in the count pass; the write pass only places the stored word
struct BitPack {
    unsigned n;
    unsigned long long acc;
    __device__ __forceinline__ void init(unsigned) { n = 0; acc = 0; }
    __device__ __forceinline__ void put(int len, unsigned v)
    {
        const int sh = 64 - (int)n - len;
        if (len > 0 && sh >= 0) acc |= (unsigned long long)v << sh;
        n += (unsigned)(len > 0 ? len : 0);
    }
    __device__ __forceinline__ void flush() {}
};
struct BitWrite {
    uint32_t* buf;
    unsigned word, nb;
    unsigned long long acc;  // pending bits, left aligned
    __device__ __forceinline__ void init(unsigned pos) { word = pos >> 5; nb = pos & 31; acc = 0; }
    __device__ __forceinline__ void put(int len, unsigned v)
    {
        if (len <= 0) return;
        acc |= (unsigned long long)v << (64 - nb - len);
        nb += (unsigned)len;
        if (nb >= 32) {
            const unsigned w = (unsigned)(acc >> 32);
            if (w) atomicOr(buf + word, __builtin_bswap32(w));
            acc <<= 32; nb -= 32; word++;
        }
    }
    __device__ __forceinline__ void flush()
    {
        const unsigned w = (unsigned)(acc >> 32);
        if (nb && w) atomicOr(buf + word, __builtin_bswap32(w));
        nb = 0; acc = 0;
    }
};

template <class S>
__device__ __forceinline__ void put_ue(S& s, unsigned v)
{
    const unsigned x = v + 1;
    const int n = 31 - __clz((int)x);  // floor(log2 x), x < 2^31
    if (2 * n + 1 > 32) { s.put(2 * n + 1 - 32, 0); s.put(32, x); }
    else s.put(2 * n + 1, x);
}
template <class S>
__device__ __forceinline__ void put_se(S& s, int v)
{
    put_ue(s, v > 0 ? (unsigned)(2 * v - 1) : (unsigned)(-2 * v));
}

// 9.2.2.1 level_prefix / level_suffix for one levelCode.  Prefix (zeros, then the one) and suffix leave as ONE
// put of at most 28 bits: the sink's 64-bit shift-and-or is the expensive part of a level.
template <class S>
__device__ __forceinline__ void put_level(S& s, int code, int suffix_len)
{
    if (suffix_len == 0) {
        if (code < 14) { s.put(code + 1, 1); return; }
        if (code < 30) { s.put(19, 16u | (unsigned)(code - 14)); return; }
        int c = code - 30, prefix = 15;
        while (c >= (1 << (prefix - 3))) { c -= 1 << (prefix - 3); prefix++; }
        s.put(2 * prefix - 2, (1u << (prefix - 3)) | (unsigned)c);
        return;
    }
    if (code < (15 << suffix_len)) {
        s.put((code >> suffix_len) + 1 + suffix_len, (1u << suffix_len) | (unsigned)(code & ((1 << suffix_len) - 1)));
        return;
    }
    int c = code - (15 << suffix_len), prefix = 15;
    while (c >= (1 << (prefix - 3))) { c -= 1 << (prefix - 3); prefix++; }
    s.put(2 * prefix - 2, (1u << (prefix - 3)) | (unsigned)c);
}

// residual_block_cavlc(): lv points at scan position 0 of this block's list
template <class S>
__device__ __forceinline__ void cavlc_block(S& s, const int16_t* lv, int maxc, int nC)
{
    unsigned nzm = 0, one = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const int v = i < maxc ? (int)lv[i] : 0;
        nzm |= (unsigned)(v != 0) << i;
        one |= (unsigned)(v == 1 || v == -1) << i;
    }
    const int tc = __popc(nzm);
    int t1 = 0;
    {
        unsigned m = nzm;
        while (m && t1 < 3) {
            const int i = 31 - __clz((int)m);
            if (!((one >> i) & 1)) break;
            t1++; m &= ~(1u << i);
        }
    }
    if (nC == -1) s.put(c_cdc_len[4 * tc + t1], c_cdc_bits[4 * tc + t1]);
    else {
        const int tab = nC < 2 ? 0 : nC < 4 ? 1 : nC < 8 ? 2 : 3;
        s.put(c_ct_len[tab][4 * tc + t1], c_ct_bits[tab][4 * tc + t1]);
    }
    if (!tc) return;
    unsigned m = nzm;
    {   // trailing_ones_sign_flag of the (up to three) trailing ones, highest frequency first: one put
        unsigned signs = 0;
        for (int k = 0; k < t1; k++) {
            const int i = 31 - __clz((int)m);
            signs = (signs << 1) | (lv[i] < 0 ? 1u : 0u);
            m &= ~(1u << i);
        }
        s.put(t1, signs);
    }
    int suffix_len = (tc > 10 && t1 < 3) ? 1 : 0;
    bool first = true;
    while (m) {
        const int i = 31 - __clz((int)m);
        m &= ~(1u << i);
        const int lvl = lv[i];
        int code = lvl > 0 ? 2 * lvl - 2 : -2 * lvl - 1;
        if (first && t1 < 3) code -= 2;
        first = false;
        put_level(s, code, suffix_len);
        if (suffix_len == 0) suffix_len = 1;
        if (iabs(lvl) > (3 << (suffix_len - 1)) && suffix_len < 6) suffix_len++;
    }
    const int top = 31 - __clz((int)nzm);
    if (tc < maxc) {
        const int tz = top + 1 - tc;
        if (nC == -1) s.put(c_ctz_len[tc - 1][tz], c_ctz_bits[tc - 1][tz]);
        else s.put(c_tz_len[tc - 1][tz], c_tz_bits[tc - 1][tz]);
        int zl = tz;
        m = nzm;
        int hi = top;
        m &= ~(1u << hi);
        while (m && zl > 0) {
            const int lo = 31 - __clz((int)m);
            m &= ~(1u << lo);
            const int run = hi - lo - 1;
            const int t = (zl > 7 ? 7 : zl) - 1;
            s.put(c_run_len[t][run], c_run_bits[t][run]);
            zl -= run;
            hi = lo;
        }
    }
}

// nC of a luma 4x4 block (bx4,by4 in 0..3) / chroma block (2x2 grid), 9.2.1
// (top: the macroblock above is in this slice)
__device__ __forceinline__ int nc_luma(const MbInfo* m, int mx, bool top, int mbw, int x, int y)
{
    int nA = -1, nB = -1;
    if (x > 0) nA = m->tc[xy2blk(x - 1, y)];
    else if (mx > 0) nA = (m - 1)->tc[xy2blk(3, y)];
    if (y > 0) nB = m->tc[xy2blk(x, y - 1)];
    else if (top) nB = (m - mbw)->tc[xy2blk(x, 3)];
    if (nA >= 0 && nB >= 0) return (nA + nB + 1) >> 1;
    return nA >= 0 ? nA : (nB >= 0 ? nB : 0);
}
__device__ __forceinline__ int nc_chroma(const MbInfo* m, int mx, bool top, int mbw, int pl, int x, int y)
{
    const int base = 16 + pl * 4;
    int nA = -1, nB = -1;
    if (x > 0) nA = m->tc[base + 2 * y];
    else if (mx > 0) nA = (m - 1)->tc[base + 2 * y + 1];
    if (y > 0) nB = m->tc[base + x];
    else if (top) nB = (m - mbw)->tc[base + 2 + x];
    if (nA >= 0 && nB >= 0) return (nA + nB + 1) >> 1;
    return nA >= 0 ? nA : (nB >= 0 ? nB : 0);
}

struct CavlcParams {
    const MbInfo* mb;
    const int16_t* levels;
    const int16_t* mvd;   // 8 per macroblock: mvd_l0 of its partitions, coding order
    const int16_t* mvq;   // 8 per macroblock: vectors of its four 8x8 quadrants
    int mbw, nmb, p_slice;
    int t8x8;             // PPS transform_8x8_mode_flag (High profile)
    int nref;             // num_ref_idx_l0_active of the slice (ref_idx_l0 is coded when > 1)
    SliceRows sl;         // slices of the picture (bands of sl.rows macroblock rows)
    int mb_first, mb_end; // the macroblocks this instance codes (its band of whole slices; 0 .. nmb alone)
    MbDiv mbdiv;          // macroblock index / mbw
    unsigned slice_cap;   // bytes of payload buffer per slice: slice s owns bytes [s * slice_cap, (s + 1) * slice_cap)
    uint16_t* slotbits;   // 32 per macroblock
    unsigned long long* slotcode;   // 32 per macroblock: the slot's bits, left aligned, when slotbits <= 64
    uint32_t* mbbits;     // per macroblock, then (after the scan) bit offsets
    uint32_t* bitbuf;     // zeroed slice payload buffer
    int32_t* prevcoded;   // [nmb + 1]: index of the last macroblock before i that is not P_Skip (-1: none); k_skip_scan
    uint8_t* bs;          // boundary strengths for the loop filter, 32 B per macroblock (written by k_bs)
    int st_mb;            // lockstep batch: macroblocks per batch item (all per-MB arrays)
    size_t st_bitbuf;     // 32-bit words between the payload buffers of two batch items
    // the source picture, for the samples of I_PCM macroblocks (the reconstruction is being loop-filtered meanwhile)
    const uint8_t* src;
    int w, h, src_nv12;
    size_t st_src;
    const uint8_t* aux;   // 16 bytes per macroblock: Intra4x4PredMode of the blocks of MB_I4 macroblocks (blkIdx order)
    const uint32_t* itemtab;   // indirect launches (IND = true): position -> batch item (dev_common.h item_ref); null otherwise
};
__device__ __forceinline__ CavlcParams batch_view(CavlcParams C, int g)
{
    C.mb += (size_t)g * C.st_mb; C.levels += (size_t)g * C.st_mb * LV_STRIDE; C.mvd += (size_t)g * C.st_mb * 8; C.mvq += (size_t)g * C.st_mb * 8;
    C.prevcoded += (size_t)g * (C.st_mb + 1);
    C.slotbits += (size_t)g * C.st_mb * 32; C.slotcode += (size_t)g * C.st_mb * 32; C.mbbits += (size_t)g * C.st_mb; C.bitbuf += (size_t)g * C.st_bitbuf;
    if (C.bs) C.bs += (size_t)g * C.st_mb * 32;
    C.src += (size_t)g * C.st_src;
    C.aux += (size_t)g * C.st_mb * 16;
    return C;
}
template <bool IND>
__device__ __forceinline__ CavlcParams batch_view(CavlcParams C, int pos) { return batch_view(C, batch_item<IND>(C.itemtab, pos)); }
enum { MAX_BATCH = 64 };
// One limit for both halves of the overflow guard: a slice is accepted by k_bit_scan only when header + data +
// tail end at least SLICE_GUARD_BITS before the end of its share of the payload buffer, and k_cavlc<true> drops a slot
// only when it would end past that same line - so an accepted slice never has a slot missing.
enum { SLICE_GUARD_BITS = 1024, SLICE_TAIL_BITS = 64 };   // tail: the final mb_skip_run (<= 33 bits) + rbsp_stop_one_bit
struct HdrBatch { unsigned long long bits[MAX_BATCH]; unsigned char len[MAX_BATCH]; };  // slice header of every batch item, from slice_type on

// 8.7.2.1 boundary strength of one 4-sample edge segment; l = (dir, edge, segment) within the macroblock.
// top: the macroblock above is in this slice; with several slices the stream says disable_deblocking_filter_idc 2
// and the edge between two slices is not filtered.
// qv: the quadrant vectors of q's macroblock (8 int16; the neighbours' lie 8 * (macroblock distance) before it)
__device__ __forceinline__ int mb_edge_strength(const MbInfo* q, const int16_t* qv, int mx, bool top, int mbw, int l)
{
    const int dir = l >> 4, e = (l >> 2) & 3, k = l & 3;
    if (e == 0 && (dir == 0 ? mx == 0 : !top)) return 0;
    const MbInfo* p = e == 0 ? (dir == 0 ? q - 1 : q - mbw) : q;
    const int16_t* pv = e == 0 ? (dir == 0 ? qv - 8 : qv - 8 * mbw) : qv;
    const int bq = dir == 0 ? xy2blk(e, k) : xy2blk(k, e);
    const int bp = dir == 0 ? (e == 0 ? xy2blk(3, k) : xy2blk(e - 1, k)) : (e == 0 ? xy2blk(k, 3) : xy2blk(k, e - 1));
    if (mb_is_intra(p->type) || mb_is_intra(q->type)) return e == 0 ? 4 : 3;
    // transform_size_8x8_flag (i16_mode of an inter macroblock): no 4x4-internal edges, and "contains non-zero coefficients"
    // (8.7.2.1) refers to the 8x8 block = the four interleaved lists of the quadrant
    const bool q8 = (q->type == MB_P16 || q->type >= MB_P16X8) && q->i16_mode == 1, p8 = (p->type == MB_P16 || p->type >= MB_P16X8) && p->i16_mode == 1;
    if (q8 && (e & 1)) return 0;
    const bool nzq = q8 ? (*(const uint32_t*)(q->tc + (bq & ~3)) != 0) : q->tc[bq] != 0;
    const bool nzp = p8 ? (*(const uint32_t*)(p->tc + (bp & ~3)) != 0) : p->tc[bp] != 0;
    if (nzp || nzq) return 2;
    if (p->chroma_mode != q->chroma_mode) return 1;   // different reference pictures (ref_idx_l0 rides in chroma_mode; one list, never reordered)
    // the vectors of the two blocks' partitions (block b lies in quadrant b >> 2)
    const uint32_t vp = *(const uint32_t*)(pv + 2 * (bp >> 2)), vq = *(const uint32_t*)(qv + 2 * (bq >> 2));
    if (iabs((int)(int16_t)(vp & 0xFFFFu) - (int)(int16_t)(vq & 0xFFFFu)) >= 4 || iabs((int)(int16_t)(vp >> 16) - (int)(int16_t)(vq >> 16)) >= 4) return 1;
    return 0;
}

// one source sample of component comp (0 Y, 1 Cb, 2 Cr), clamped like every other source read
__device__ __forceinline__ unsigned pcm_sample(const CavlcParams& C, int comp, int x, int y)
{
    if (comp == 0) return (unsigned)src_px(C.src, C.w, C.h, x, y);
    const int pw = C.w / 2, ph = C.h / 2, xx = x < pw ? x : pw - 1, yy = y < ph ? y : ph - 1;
    const uint8_t* base = C.src + (size_t)C.w * C.h;
    if (C.src_nv12) return base[(size_t)yy * (2 * pw) + 2 * xx + (comp - 1)];
    return base[(comp == 2 ? (size_t)pw * ph : 0) + (size_t)yy * pw + xx];
}

// slot: 0 header, 1 Intra16x16 DC, 2..17 luma blkIdx 0..15, 18/19 chroma DC, 20..27 chroma AC
template <class S>
__device__ __forceinline__ void code_slot(S& s, const CavlcParams& C, int mbi, int slot)
{
    const MbInfo* m = C.mb + mbi;
    if (m->type == MB_PSKIP) return;
    const int my = C.mbdiv.row(mbi), mx = mbi - my * C.mbw;
    const int srow = C.sl.row_in_slice(my);
    const bool top = srow != 0;
    const int16_t* lv = C.levels + (size_t)mbi * LV_STRIDE;
    const int cbpl = m->cbp & 15, cbpc = m->cbp >> 4;
    const bool i16 = m->type == MB_I16;
    if (m->type == MB_IPCM) {
        // 7.3.5: mb_type I_PCM (slot 0; the alignment zero bits after it are left to the zeroed buffer, k_bit_scan and the
        // write pass place the samples on the next byte boundary), then 384 samples, 16 per slot: slots 1..16 the luma rows,
        // 17..20 Cb (two rows each), 21..24 Cr
        if (slot == 0) {
            if (C.p_slice) put_ue(s, (unsigned)(mbi - 1 - max(C.prevcoded[mbi], (my - srow) * C.mbw - 1)));
            put_ue(s, C.p_slice ? 30u : 25u);
        } else if (slot <= 24) {
            const int comp = slot <= 16 ? 0 : (slot <= 20 ? 1 : 2);
#pragma unroll 1
            for (int j = 0; j < 4; j++) {
                unsigned v = 0;
                for (int k = 0; k < 4; k++) {
                    const int i = 4 * j + k;   // sample index inside the slot
                    const unsigned smp = comp == 0 ? pcm_sample(C, 0, 16 * mx + i, 16 * my + slot - 1)
                                                   : pcm_sample(C, comp, 8 * mx + (i & 7), 8 * my + 2 * ((slot - 17) & 3) + (i >> 3));
                    v = (v << 8) | smp;
                }
                s.put(32, v);
            }
        }
        return;
    }
    if (slot == 0) {
        if (C.p_slice) {
            // mb_skip_run: the P_Skip macroblocks right before this one, counted from the slice's first macroblock
            put_ue(s, (unsigned)(mbi - 1 - max(C.prevcoded[mbi], (my - srow) * C.mbw - 1)));
        }
        if (m->type == MB_I4) {
            // I_NxN (7.3.5.1): the sixteen Intra4x4 modes against their predictions (8.3.1.1: the smaller of the left and upper
            // blocks' modes; DC when such a neighbour is not an Intra4x4 macroblock, or missing), then chroma mode and coded_block_pattern
            const uint8_t* am = C.aux + (size_t)mbi * 16;
            put_ue(s, C.p_slice ? 5u : 0u);
            if (C.t8x8) s.put(1, 0u);   // transform_size_8x8_flag: Intra4x4, not Intra8x8
#pragma unroll 1
            for (int k = 0; k < 16; k++) {
                const int x = blk_x(k), y = blk_y(k);
                int mA, mB;
                bool dc_only = false;
                if (x > 0) mA = am[xy2blk(x - 1, y)];
                else if (mx == 0) { dc_only = true; mA = 2; }
                else mA = (m - 1)->type == MB_I4 ? (am - 16)[xy2blk(3, y)] : 2;
                if (y > 0) mB = am[xy2blk(x, y - 1)];
                else if (!top) { dc_only = true; mB = 2; }
                else mB = (m - C.mbw)->type == MB_I4 ? (am - 16 * C.mbw)[xy2blk(x, 3)] : 2;
                const int pm = dc_only ? 2 : min(mA, mB), mode = am[k];
                if (mode == pm) s.put(1, 1u);
                else s.put(4, (unsigned)(mode < pm ? mode : mode - 1));   // prev_intra4x4_pred_mode_flag = 0, rem_intra4x4_pred_mode
            }
            put_ue(s, m->chroma_mode);
            put_ue(s, c_cbp2code_intra[m->cbp]);
            if (m->cbp) put_se(s, 0);
        } else if (i16) {
            const unsigned t = 1u + m->i16_mode + 4u * (unsigned)cbpc + (cbpl ? 12u : 0u);
            put_ue(s, C.p_slice ? 5u + t : t);
            put_ue(s, m->chroma_mode);
            put_se(s, 0);
        } else {
            // 7.3.5.1 / 7.3.5.2: mb_type (0 P_L0_16x16, 1 P_L0_L0_16x8, 2 P_L0_L0_8x16, 3 P_8x8 with four sub_mb_type P_L0_8x8),
            // every partition's ref_idx_l0, then every partition's mvd_l0 (k_mvpred left them in coding order)
            const int shape = m->type >= MB_P16X8 ? m->type - MB_P16X8 + 1 : 0, nparts = shape == 0 ? 1 : (shape == 3 ? 4 : 2);
            put_ue(s, (unsigned)shape);
            if (shape == 3) s.put(4, 15u);   // four times ue(0)
            for (int k = 0; k < nparts; k++) {
                if (C.nref == 2) s.put(1, m->chroma_mode ? 0u : 1u);   // ref_idx_l0, te(v): with two pictures the inverted bit (9.1)
                else if (C.nref > 2) put_ue(s, m->chroma_mode);
            }
            for (int k = 0; k < nparts; k++) {
                put_se(s, C.mvd[8 * mbi + 2 * k]);
                put_se(s, C.mvd[8 * mbi + 2 * k + 1]);
            }
            put_ue(s, c_cbp2code_inter[m->cbp]);
            if (C.t8x8 && cbpl) s.put(1, m->i16_mode & 1u);   // transform_size_8x8_flag (High profile, luma coefficients present)
            if (m->cbp) put_se(s, 0);
        }
    } else if (slot == 1) {
        if (i16) cavlc_block(s, lv + LV_LUMA_DC, 16, nc_luma(m, mx, top, C.mbw, 0, 0));
    } else if (slot < 18) {
        const int b = slot - 2;
        if (cbpl & (1 << (b >> 2))) {
            const int nC = nc_luma(m, mx, top, C.mbw, blk_x(b), blk_y(b));
            if (i16) cavlc_block(s, lv + LV_LUMA + b * 16 + 1, 15, nC);
            else cavlc_block(s, lv + LV_LUMA + b * 16, 16, nC);
        }
    } else if (slot < 20) {
        if (cbpc) cavlc_block(s, lv + LV_CHROMA_DC + (slot - 18) * 4, 4, -1);
    } else if (slot < 28) {
        const int k = slot - 20, pl = k >> 2, b = k & 3;
        if (cbpc == 2) cavlc_block(s, lv + LV_CHROMA_AC + k * 16 + 1, 15, nc_chroma(m, mx, top, C.mbw, pl, b & 1, b >> 1));
    }
}

// boundary strengths of every macroblock for the loop filter (lane = macroblock, edge segment).  Its own small
// launch on the reconstruction stream: the filter then never waits for the entropy-coding stream.
// anybs[item] is set to the picture's serial as soon as any strength of the picture is non-zero: a picture without
// any (a static screen) needs no loop filter pass at all, and k_deblock_rows returns at once.
template <bool IND = false>
__global__ __launch_bounds__(64) void k_bs(CavlcParams C0, unsigned* anybs, unsigned serial)
{
    __builtin_amdgcn_s_setprio(2);
    const CavlcParams C = batch_view<IND>(C0, blockIdx.y);
    const int lane = threadIdx.x, slot = lane & 31;
    // a wave walks its share of the macroblock pairs (the host launches at most BS_WAVES waves per picture): 130 560 tiny waves
    // per launch of 32 pictures cost more to dispatch beside the other instance's motion search than their loads take
    int any = 0;
    for (int pair = blockIdx.x; 2 * pair < C.mb_end - C.mb_first; pair += gridDim.x) {
        const int mbi = C.mb_first + pair * 2 + (lane >> 5);
        if (mbi < C.mb_end) {
            const int my = C.mbdiv.row(mbi);
            const int bs = mb_edge_strength(C.mb + mbi, C.mvq + (size_t)mbi * 8, mbi - my * C.mbw, C.sl.has_top(my), C.mbw, slot);
            C.bs[(size_t)mbi * 32 + slot] = (uint8_t)bs;
            any |= bs;
        }
    }
    if (__ballot(any != 0) != 0ull && lane == 0) anybs[batch_item<IND>(C0.itemtab, blockIdx.y)] = serial;   // same value from every writer: a plain store
}
enum { BS_WAVES = 1020 };

// P slices: prevcoded[i] = index of the last macroblock before i that is not P_Skip (a prefix maximum over the
// picture, one 256-thread workgroup per picture), so that mb_skip_run costs no walk over the skipped macroblocks
// (a static screen is one long run).
template <bool IND = false>
__global__ __launch_bounds__(256) void k_skip_scan(CavlcParams C0)
{
    const CavlcParams C = batch_view<IND>(C0, blockIdx.x);
    __shared__ int s_last[256];
    const int t = threadIdx.x;
    const int per = (C.mb_end - C.mb_first + 255) / 256;
    const int b0 = min(C.mb_end, C.mb_first + t * per), b1 = min(C.mb_end, b0 + per);
    int last = -1;
    for (int i = b0; i < b1; i++)
        if (C.mb[i].type != MB_PSKIP) last = i;
    s_last[t] = last;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {       // inclusive prefix maximum
        const int v = t >= o ? s_last[t - o] : -1;
        __syncthreads();
        s_last[t] = max(s_last[t], v);
        __syncthreads();
    }
    int run = t ? s_last[t - 1] : -1;         // last coded macroblock before this thread's range
    for (int i = b0; i < b1; i++) {
        C.prevcoded[i] = run;
        if (C.mb[i].type != MB_PSKIP) run = i;
    }
    if (t == 255) C.prevcoded[C.mb_end] = s_last[255];
}

template <bool WRITE, bool IND = false>
__global__ __launch_bounds__(64) void k_cavlc(CavlcParams C0)
{
    __builtin_amdgcn_s_setprio(AB_PRIO_EC);
    const CavlcParams C = batch_view<IND>(C0, blockIdx.y);
    const int lane = threadIdx.x, slot = lane & 31;
    const int mbi = C.mb_first + blockIdx.x * 2 + (lane >> 5);
    const bool live = mbi < C.mb_end;
    if (!WRITE) {
        BitPack s;
        s.init(0);
        if (live) code_slot(s, C, mbi, slot);
        if (live) C.slotbits[(size_t)mbi * 32 + slot] = (uint16_t)s.n;
        if (live && s.n && s.n <= 64u) C.slotcode[(size_t)mbi * 32 + slot] = s.acc;
        const int tot = group_sum<32>((int)s.n);
        if (live && slot == 0) C.mbbits[mbi] = (uint32_t)tot | (C.mb[mbi].type == MB_IPCM ? 0x80000000u : 0u);   // bit 31: k_bit_scan aligns the samples
    } else {
        // exclusive prefix over the 32 slots of this macroblock
        unsigned n = live ? C.slotbits[(size_t)mbi * 32 + slot] : 0, incl = n;
#pragma unroll
        for (int o = 1; o < 32; o <<= 1) {
            const unsigned t = (unsigned)__shfl_up((int)incl, o, 32);
            if (slot >= o) incl += t;
        }
        // A slice that outgrows its share of the payload buffer (twice its luma bytes: only noise at the lowest QPs
        // codes to that) is reported by k_bit_scan and refused by the host; nothing may be written past the share.
        unsigned pos = live ? C.mbbits[mbi] + incl - n : 0u;
        {   // I_PCM: pcm_alignment_zero_bits between the header (slot 0) and the samples
            const unsigned hdr = (unsigned)__shfl((int)n, lane & 32);
            if (live && slot > 0 && C.mb[mbi].type == MB_IPCM) pos += (0u - (C.mbbits[mbi] + hdr)) & 7u;
        }
        const unsigned lim = (__umulhi((unsigned)C.mbdiv.row(mbi), C.sl.inv) + 1u) * C.slice_cap * 8u - (unsigned)SLICE_GUARD_BITS;
        if (live && n && pos + n <= lim) {
            if (n <= 64u) {   // coded by the count pass: OR the stored word in at its final bit position
                const unsigned long long code = C.slotcode[(size_t)mbi * 32 + slot];
                const unsigned off = pos & 31;
                const unsigned long long hi = code >> off;
                const uint32_t w0 = (uint32_t)(hi >> 32), w1 = (uint32_t)hi, w2 = off ? (uint32_t)((code << (64 - off)) >> 32) : 0u;
                uint32_t* dst = C.bitbuf + (pos >> 5);
                if (w0) atomicOr(dst, __builtin_bswap32(w0));
                if (w1) atomicOr(dst + 1, __builtin_bswap32(w1));
                if (w2) atomicOr(dst + 2, __builtin_bswap32(w2));
            } else {
                BitWrite s;
                s.buf = C.bitbuf;
                s.init(pos);
                code_slot(s, C, mbi, slot);
                s.flush();
            }
        }
    }
}

struct SliceInfo {       // lives in pinned host memory, written by the device
    uint32_t total_bits; // slice_data + header + trailing bits
    uint32_t total_bytes;
    uint32_t epb_count;  // positions needing an emulation prevention byte
    uint32_t error;
    uint32_t me_cost;    // scene-change statistic gathered by k_me (0 for IDR pictures)
    uint32_t searched;   // P slices: macroblocks k_me searched (its "nothing left to code" tests settled the others: me_cost 0)
    uint32_t tq_coded;   // of those, the ones k_tq / k_tq8 coded (not handed to the intra pass: me_cost bit 15)
    uint32_t pad[1];
};

// one workgroup of SCAN_NT threads per slice (item = picture * nsl + slice): exclusive scan of mbbits over the
// slice's macroblocks (in place -> bit offsets inside the picture's payload buffer, where slice s starts at byte
// s * slice_cap), slice header, tail.
// Kept small (4 waves): a workgroup is dispatched only when one CU has room for all of its waves, and beside
// another instance's motion search a 16-wave workgroup waits long for that.
enum { SCAN_NT = 256 };
// H: the picture's slice header; Hpcm: the same with disable_deblocking_filter_idc 1, taken when the picture holds an I_PCM
// macroblock (anypcm[picture] == pic_serial): such a picture is not loop-filtered.
// IND: blockIdx.x / nsl is a POSITION of the step; H / Hpcm are laid out by position, everything else by batch item
template <bool IND = false>
__global__ __launch_bounds__(SCAN_NT) void k_bit_scan(CavlcParams C0, HdrBatch H, HdrBatch Hpcm, const unsigned* anypcm, unsigned pic_serial, SliceInfo* info0,
                                                   const uint16_t* me_cost0, int nsl, int sl0, unsigned slice_cap)
{   // nsl slices of this instance's band per picture, the first of them is slice sl0 of the picture
    __builtin_amdgcn_s_setprio(AB_PRIO_EC);
    const int pos = blockIdx.x / nsl, sl = sl0 + blockIdx.x - pos * nsl;
    const int pic = batch_item<IND>(C0.itemtab, pos), item = pic * nsl + (sl - sl0);
    const CavlcParams C = batch_view(C0, pic);
    const int mb0 = sl * C.sl.rows * C.mbw, mb1 = min(C.nmb, mb0 + C.sl.rows * C.mbw), cnt = mb1 - mb0;
    // slice_header(): first_mb_in_slice is written here, the rest (the same for every slice of the picture) comes from the host
    const bool pcm_pic = anypcm[pic] == pic_serial;
    const unsigned long long hdr_bits = pcm_pic ? Hpcm.bits[pos] : H.bits[pos];
    const int hdr_rest = pcm_pic ? Hpcm.len[pos] : H.len[pos];
    BitCount fl;
    fl.init(0);
    put_ue(fl, (unsigned)mb0);
    const unsigned base = (unsigned)sl * slice_cap * 8u, hdr_len = fl.n + (unsigned)hdr_rest;
    SliceInfo* info = info0 + item;
    __shared__ unsigned s_part[SCAN_NT];
    __shared__ unsigned s_cost, s_searched, s_intra;
    const int t = threadIdx.x;
    if (t == 0) { s_cost = 0; s_searched = 0; s_intra = 0; }
    const int per = (cnt + SCAN_NT - 1) / SCAN_NT;
    const int b0 = min(mb1, mb0 + t * per), b1 = min(mb1, b0 + per);
    unsigned sum = 0, flag = 0;
    for (int i = b0; i < b1; i++) { const unsigned n = C.mbbits[i]; sum += n & 0x7FFFFFFFu; flag |= n >> 31; }
    s_part[t] = sum;
    // I_PCM macroblocks (bit 31) need their samples byte aligned, which makes a macroblock's position depend on the exact
    // position of everything before it: such a slice (rare) is laid out by one thread, in order
    const bool seq = __syncthreads_or((int)flag) != 0;
    __shared__ unsigned s_total;
    if (C.p_slice) {   // scene-change statistic: sum of the per-macroblock motion costs k_me left
        const uint16_t* mc = me_cost0 + (size_t)pic * C.st_mb;
        // (bit 15 marks macroblocks of the intra pass; a macroblock settled by k_me's tests has cost 0, a searched one at least
        // 2 lambda: the two counts say what k_me and k_tq really worked on - the bench's roofline blocks are priced on them)
        unsigned cs = 0, ns = 0, ni = 0;
        for (int i = b0; i < b1; i++) { const unsigned v = mc[i]; cs += v & 0x7FFFu; ns += v != 0u; ni += v >> 15; }
        if (cs) atomicAdd(&s_cost, cs);
        if (ns) atomicAdd(&s_searched, ns);
        if (ni) atomicAdd(&s_intra, ni);
    }
    for (int o = 1; o < SCAN_NT; o <<= 1) {
        const unsigned v = t >= o ? s_part[t - o] : 0;
        __syncthreads();
        s_part[t] += v;
        __syncthreads();
    }
    if (!seq) {
        unsigned run = base + hdr_len + s_part[t] - sum;
        for (int i = b0; i < b1; i++) {
            const unsigned n = C.mbbits[i];
            C.mbbits[i] = run;
            run += n;
        }
        if (t == SCAN_NT - 1) s_total = hdr_len + s_part[SCAN_NT - 1];
    } else if (t == 0) {
        unsigned pos = base + hdr_len;   // base is a multiple of 8: positions in the buffer and in the RBSP align alike
        for (int i = mb0; i < mb1; i++) {
            const unsigned n = C.mbbits[i];
            C.mbbits[i] = pos;
            if (n >> 31) pos = ((pos + ((n & 0x7FFFFFFFu) - 3072u) + 7u) & ~7u) + 3072u;
            else pos += n;
        }
        s_total = pos - base;
    }
    __syncthreads();
    if (t == 0) {
        BitWrite s;
        s.buf = C.bitbuf;
        s.init(base);
        put_ue(s, (unsigned)mb0);
        if (hdr_rest > 32) { s.put(hdr_rest - 32, (unsigned)(hdr_bits >> 32)); s.put(32, (unsigned)hdr_bits); }
        else s.put(hdr_rest, (unsigned)hdr_bits);
        s.flush();
        unsigned total = s_total;
        const bool fits = total + (unsigned)SLICE_TAIL_BITS + (unsigned)SLICE_GUARD_BITS <= slice_cap * 8u;
        s.init(base + (fits ? total : hdr_len));
        BitCount c;
        c.init(0);
        if (C.p_slice) {
            // P_Skip macroblocks that end the slice
            const unsigned skips = (unsigned)(mb1 - 1 - max(C.prevcoded[mb1], mb0 - 1));
            if (skips) { if (fits) put_ue(s, skips); put_ue(c, skips); }
        }
        if (fits) { s.put(1, 1); s.flush(); }  // rbsp_stop_one_bit
        total += c.n + 1;
        info->total_bits = total;
        info->total_bytes = (total + 7) >> 3;
        info->epb_count = 0;
        info->error = fits ? 0u : 1u;   // 1: the slice outgrew its share of the payload buffer (k_cavlc<true> writes nothing past it)
        info->me_cost = s_cost;   // complete: every thread passed the scan's barriers after its atomicAdd
        info->searched = s_searched;
        info->tq_coded = s_searched - s_intra;
    }
}

// Copy the payload of one slice to the pinned access unit buffer, count emulation-prevention sites, publish
// SliceInfo to pinned host memory and leave the device bit buffer zeroed for its next use (one workgroup per slice).
template <bool IND = false>
__global__ __launch_bounds__(SCAN_NT) void k_pack(uint8_t* bitbuf0, size_t st_bitbuf_bytes, uint8_t* dst0, size_t st_dst, const SliceInfo* info0,
                                               SliceInfo* host_info0, int nsl, int sl0, unsigned slice_cap, const uint32_t* itemtab)
{
    const int pos = blockIdx.x / nsl, sl = sl0 + blockIdx.x - pos * nsl;
    const int pic = batch_item<IND>(itemtab, pos), item = pic * nsl + (sl - sl0);
    uint8_t* bitbuf = bitbuf0 + (size_t)pic * st_bitbuf_bytes + (size_t)sl * slice_cap;
    uint8_t* dst = dst0 + (size_t)pic * st_dst + (size_t)sl * slice_cap;
    const SliceInfo* info = info0 + item;
    SliceInfo* host_info = host_info0 + item;
    __shared__ unsigned s_cnt;
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    const unsigned nbytes = min(info->total_bytes, slice_cap - 32u);   // (an overflowing slice is reported through info->error)
    unsigned cnt = 0;
    for (unsigned i = threadIdx.x * 16u; i < nbytes; i += blockDim.x * 16u) {
        const uint4 v = *(const uint4*)(bitbuf + i);  // buffer is padded and zeroed past the end
        *(uint4*)(dst + i) = v;
        const uint32_t nxt = *(const uint32_t*)(bitbuf + i + 16);
        const uint32_t w[5] = {v.x, v.y, v.z, v.w, nxt};
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const unsigned b0 = (w[k >> 2] >> (8 * (k & 3))) & 255, b1 = (w[(k + 1) >> 2] >> (8 * ((k + 1) & 3))) & 255,
                           b2 = (w[(k + 2) >> 2] >> (8 * ((k + 2) & 3))) & 255;
            if (i + k + 2 < nbytes && b0 == 0 && b1 == 0 && b2 <= 3) cnt++;
        }
    }
    if (cnt) atomicAdd(&s_cnt, cnt);
    __syncthreads();   // every thread has read its own chunk and its look-ahead word
    for (unsigned i = threadIdx.x * 16u; i < nbytes + 16u; i += blockDim.x * 16u) *(uint4*)(bitbuf + i) = make_uint4(0, 0, 0, 0);
    if (threadIdx.x == 0) {
        SliceInfo o = *info;
        o.epb_count = s_cnt;
        *host_info = o;
    }
}

}  // namespace h264
