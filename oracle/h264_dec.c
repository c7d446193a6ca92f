/*
 * oracle/h264_dec.c -- TEST INFRASTRUCTURE ONLY (see h264_oracle.h).
 *
 * Test-side H.264 decoder, written clause by clause from ITU-T H.264 and sharing NOTHING with the
 * encoder side of this directory: no function of h264_common.c / h264_enc.c and no table of
 * h264_tables.h is used here (every table the decoder needs is typed again below, in the layout the
 * standard prints it in, e.g. the VLC tables as bit strings).  Its job is the round trip
 *     decode(encode(yuv)) == encoder reconstruction, byte for byte,
 * which stands in for the golden bitstreams the reference does not have:
 * a wrong filter tap, rounding rule, table entry or boundary-strength rule on the encoder side
 * (oracle or GPU) cannot agree with this file by construction, only by both being right.
 * tests/test_oracle_kat.py additionally compares this file's tables with h264_tables.h entry by entry.
 *
 * Covered (frame macroblocks, 4:2:0, 8 bit, CAVLC): I and P slices; I_NxN (Intra4x4), Intra16x16, I_PCM;
 * P_L0_16x16 / 16x8 / 8x16 / P_8x8 (all sub-macroblock types) / P_8x8ref0 / P_Skip; intra macroblocks in P slices;
 * several reference frames (sliding window marking; reference list modification by short-term picture numbers); mb_qp_delta;
 * chroma_qp_index_offset; constrained_intra_pred_flag; transform_size_8x8_flag for inter macroblocks (8x8 residual blocks,
 * 8.5.13) -- Intra8x8 is refused; deblocking 8.7 with slice alpha/beta offsets and
 * disable_deblocking_filter_idc 0/1/2; several slices per picture.
 * Not covered (refused with an error): CABAC, B slices, interlace, weighted prediction, FMO/ASO,
 * long-term reference pictures, memory management control operations, scaling matrices.
 *
 * Clause map: 7.3 syntax -> parse_*; 7.4.1.1 NAL -> h264o_dec_decode; 9.1 -> rd_ue/rd_se/rd_te; 9.2 -> residual_block;
 * 8.3.1 -> intra4x4_*; 8.3.3 -> intra16x16_pred; 8.3.4 -> intra_chroma_pred; 8.3.5 -> I_PCM; 8.4.1 -> mv prediction;
 * 8.4.2.2.1 -> luma_sample_interp; 8.4.2.2.2 -> chroma_sample_interp; 8.5.6/8.5.7 inverse scans; 8.5.9-8.5.13 scaling
 * and transforms; 8.7 -> deblock_picture; 8.2.1 (POC type 2 only needs frame_num order), 8.2.4.2.1 list order, 8.2.5.3 sliding window.
 * Annex-B start-code scanning follows the in-tree statement of it at
 * /root/reference/video_decoder/VideoDecoderNetint.cpp:794-860.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "h264_oracle.h"

/* ------------------------------------------------------------------ tables, typed from the standard */
/* Table 9-5, printed layout: rows (TrailingOnes, TotalCoeff) in the standard's order, columns 0<=nC<2, 2<=nC<4,
 * 4<=nC<8, 8<=nC, nC==-1 (ChromaDCLevel, 4:2:0).  "-" = no entry. */
typedef struct { int t1, tc; const char *c[5]; } ct_row;
static const ct_row D_COEFF_TOKEN[] = {
    {0, 0, {"1", "11", "1111", "000011", "01"}},
    {0, 1, {"000101", "001011", "001111", "000000", "000111"}},
    {1, 1, {"01", "10", "1110", "000001", "1"}},
    {0, 2, {"00000111", "000111", "001011", "000100", "000100"}},
    {1, 2, {"000100", "00111", "01111", "000101", "000110"}},
    {2, 2, {"001", "011", "1101", "000110", "001"}},
    {0, 3, {"000000111", "0000111", "001000", "001000", "000011"}},
    {1, 3, {"00000110", "001010", "01100", "001001", "0000011"}},
    {2, 3, {"0000101", "001001", "01110", "001010", "0000010"}},
    {3, 3, {"00011", "0101", "1100", "001011", "000101"}},
    {0, 4, {"0000000111", "00000111", "0001111", "001100", "000010"}},
    {1, 4, {"000000110", "000110", "01010", "001101", "00000011"}},
    {2, 4, {"00000101", "000101", "01011", "001110", "00000010"}},
    {3, 4, {"000011", "0100", "1011", "001111", "0000000"}},
    {0, 5, {"00000000111", "00000100", "0001011", "010000", "-"}},
    {1, 5, {"0000000110", "0000110", "01000", "010001", "-"}},
    {2, 5, {"000000101", "0000101", "01001", "010010", "-"}},
    {3, 5, {"0000100", "00110", "1010", "010011", "-"}},
    {0, 6, {"0000000001111", "000000111", "0001001", "010100", "-"}},
    {1, 6, {"00000000110", "00000110", "001110", "010101", "-"}},
    {2, 6, {"0000000101", "00000101", "001101", "010110", "-"}},
    {3, 6, {"00000100", "001000", "1001", "010111", "-"}},
    {0, 7, {"0000000001011", "00000001111", "0001000", "011000", "-"}},
    {1, 7, {"0000000001110", "000000110", "001010", "011001", "-"}},
    {2, 7, {"00000000101", "000000101", "001001", "011010", "-"}},
    {3, 7, {"000000100", "000100", "1000", "011011", "-"}},
    {0, 8, {"0000000001000", "00000001011", "00001111", "011100", "-"}},
    {1, 8, {"0000000001010", "00000001110", "0001110", "011101", "-"}},
    {2, 8, {"0000000001101", "00000001101", "0001101", "011110", "-"}},
    {3, 8, {"0000000100", "0000100", "01101", "011111", "-"}},
    {0, 9, {"00000000001111", "000000001111", "00001011", "100000", "-"}},
    {1, 9, {"00000000001110", "00000001010", "00001110", "100001", "-"}},
    {2, 9, {"0000000001001", "00000001001", "0001010", "100010", "-"}},
    {3, 9, {"00000000100", "000000100", "001100", "100011", "-"}},
    {0, 10, {"00000000001011", "000000001011", "000001111", "100100", "-"}},
    {1, 10, {"00000000001010", "000000001110", "00001010", "100101", "-"}},
    {2, 10, {"00000000001101", "000000001101", "00001101", "100110", "-"}},
    {3, 10, {"0000000001100", "00000001100", "0001100", "100111", "-"}},
    {0, 11, {"000000000001111", "000000001000", "000001011", "101000", "-"}},
    {1, 11, {"000000000001110", "000000001010", "000001110", "101001", "-"}},
    {2, 11, {"00000000001001", "000000001001", "00001001", "101010", "-"}},
    {3, 11, {"00000000001100", "00000001000", "00001100", "101011", "-"}},
    {0, 12, {"000000000001011", "0000000001111", "000001000", "101100", "-"}},
    {1, 12, {"000000000001010", "0000000001110", "000001010", "101101", "-"}},
    {2, 12, {"000000000001101", "0000000001101", "000001101", "101110", "-"}},
    {3, 12, {"00000000001000", "000000001100", "00001000", "101111", "-"}},
    {0, 13, {"0000000000001111", "0000000001011", "0000001101", "110000", "-"}},
    {1, 13, {"000000000000001", "0000000001010", "000000111", "110001", "-"}},
    {2, 13, {"000000000001001", "0000000001001", "000001001", "110010", "-"}},
    {3, 13, {"000000000001100", "0000000001100", "000001100", "110011", "-"}},
    {0, 14, {"0000000000001011", "0000000000111", "0000001001", "110100", "-"}},
    {1, 14, {"0000000000001110", "00000000001011", "0000001100", "110101", "-"}},
    {2, 14, {"0000000000001101", "0000000000110", "0000001011", "110110", "-"}},
    {3, 14, {"000000000001000", "0000000001000", "0000001010", "110111", "-"}},
    {0, 15, {"0000000000000111", "00000000001001", "0000000101", "111000", "-"}},
    {1, 15, {"0000000000001010", "00000000001000", "0000001000", "111001", "-"}},
    {2, 15, {"0000000000001001", "00000000001010", "0000000111", "111010", "-"}},
    {3, 15, {"0000000000001100", "0000000000001", "0000000110", "111011", "-"}},
    {0, 16, {"0000000000000100", "00000000000111", "0000000001", "111100", "-"}},
    {1, 16, {"0000000000000110", "00000000000110", "0000000100", "111101", "-"}},
    {2, 16, {"0000000000000101", "00000000000101", "0000000011", "111110", "-"}},
    {3, 16, {"0000000000001000", "00000000000100", "0000000010", "111111", "-"}},
};
enum { D_NTOKEN = sizeof(D_COEFF_TOKEN) / sizeof(D_COEFF_TOKEN[0]) };

/* Tables 9-7 and 9-8: total_zeros for 4x4 blocks, column tzVlcIndex 1..15, row total_zeros 0..15 */
static const char *const D_TOTAL_ZEROS[15][16] = {
    {"1", "011", "010", "0011", "0010", "00011", "00010", "000011", "000010", "0000011", "0000010", "00000011", "00000010", "000000011", "000000010", "000000001"},
    {"111", "110", "101", "100", "011", "0101", "0100", "0011", "0010", "00011", "00010", "000011", "000010", "000001", "000000", 0},
    {"0101", "111", "110", "101", "0100", "0011", "100", "011", "0010", "00011", "00010", "000001", "00001", "000000", 0, 0},
    {"00011", "111", "0101", "0100", "110", "101", "100", "0011", "011", "0010", "00010", "00001", "00000", 0, 0, 0},
    {"0101", "0100", "0011", "111", "110", "101", "100", "011", "0010", "00001", "0001", "00000", 0, 0, 0, 0},
    {"000001", "00001", "111", "110", "101", "100", "011", "010", "0001", "001", "000000", 0, 0, 0, 0, 0},
    {"000001", "00001", "101", "100", "011", "11", "010", "0001", "001", "000000", 0, 0, 0, 0, 0, 0},
    {"000001", "0001", "00001", "011", "11", "10", "010", "001", "000000", 0, 0, 0, 0, 0, 0, 0},
    {"000001", "000000", "0001", "11", "10", "001", "01", "00001", 0, 0, 0, 0, 0, 0, 0, 0},
    {"00001", "00000", "001", "11", "10", "01", "0001", 0, 0, 0, 0, 0, 0, 0, 0, 0},
    {"0000", "0001", "001", "010", "1", "011", 0, 0, 0, 0, 0, 0, 0, 0, 0, 0},
    {"0000", "0001", "01", "1", "001", 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0},
    {"000", "001", "1", "01", 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0},
    {"00", "01", "1", 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0},
    {"0", "1", 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0},
};
/* Table 9-9 (a): total_zeros for chroma DC 2x2, tzVlcIndex 1..3 */
static const char *const D_TOTAL_ZEROS_CDC[3][4] = {{"1", "01", "001", "000"}, {"1", "01", "00", 0}, {"1", "0", 0, 0}};
/* Table 9-10: run_before, column zerosLeft 1..6 and >6, row run_before 0..14 */
static const char *const D_RUN_BEFORE[7][15] = {
    {"1", "0", 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0},
    {"1", "01", "00", 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0},
    {"11", "10", "01", "00", 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0},
    {"11", "10", "01", "001", "000", 0, 0, 0, 0, 0, 0, 0, 0, 0, 0},
    {"11", "10", "011", "010", "001", "000", 0, 0, 0, 0, 0, 0, 0, 0, 0},
    {"11", "000", "001", "011", "010", "101", "100", 0, 0, 0, 0, 0, 0, 0, 0},
    {"111", "110", "101", "100", "011", "010", "001", "0001", "00001", "000001", "0000001", "00000001", "000000001", "0000000001", "00000000001"},
};
/* Table 9-4, ChromaArrayType 1: rows codeNum 0..47, {Intra_4x4 / Intra_8x8, Inter} */
static const uint8_t D_CBP[48][2] = {
    {47, 0},  {31, 16}, {15, 1},  {0, 2},   {23, 4},  {27, 8},  {29, 32}, {30, 3},  {7, 5},   {11, 10}, {13, 12}, {14, 15},
    {39, 47}, {43, 7},  {45, 11}, {46, 13}, {16, 14}, {3, 6},   {5, 9},   {10, 31}, {12, 35}, {19, 37}, {21, 42}, {26, 44},
    {28, 33}, {35, 34}, {37, 36}, {42, 40}, {44, 39}, {1, 43},  {2, 45},  {4, 46},  {8, 17},  {17, 18}, {18, 20}, {20, 24},
    {24, 19}, {6, 21},  {9, 26},  {22, 28}, {25, 23}, {32, 27}, {33, 29}, {34, 30}, {36, 22}, {40, 25}, {38, 38}, {41, 41}};
/* Figure 8-8 (a), 4x4 zig-zag: idx -> (x, y) */
static const uint8_t D_ZZ4[16][2] = {{0, 0}, {1, 0}, {0, 1}, {0, 2}, {1, 1}, {2, 0}, {3, 0}, {2, 1},
                                     {1, 2}, {0, 3}, {1, 3}, {2, 2}, {3, 1}, {3, 2}, {2, 3}, {3, 3}};
/* Figure 8-9 (a), 8x8 zig-zag: idx -> (x, y) */
static const uint8_t D_ZZ8[64][2] = {
    {0, 0}, {1, 0}, {0, 1}, {0, 2}, {1, 1}, {2, 0}, {3, 0}, {2, 1}, {1, 2}, {0, 3}, {0, 4}, {1, 3}, {2, 2}, {3, 1}, {4, 0}, {5, 0},
    {4, 1}, {3, 2}, {2, 3}, {1, 4}, {0, 5}, {0, 6}, {1, 5}, {2, 4}, {3, 3}, {4, 2}, {5, 1}, {6, 0}, {7, 0}, {6, 1}, {5, 2}, {4, 3},
    {3, 4}, {2, 5}, {1, 6}, {0, 7}, {1, 7}, {2, 6}, {3, 5}, {4, 4}, {5, 3}, {6, 2}, {7, 1}, {7, 2}, {6, 3}, {5, 4}, {4, 5}, {3, 6},
    {2, 7}, {3, 7}, {4, 6}, {5, 5}, {6, 4}, {7, 3}, {7, 4}, {6, 5}, {5, 6}, {4, 7}, {5, 7}, {6, 6}, {7, 5}, {7, 6}, {6, 7}, {7, 7}};
/* 8.5.9: normAdjust4x4 v(m, 0..2) (positions (0,0) class / (1,1) class / other) */
static const uint8_t D_V4[6][3] = {{10, 16, 13}, {11, 18, 14}, {13, 20, 16}, {14, 23, 18}, {16, 25, 20}, {18, 29, 23}};
/* 8.5.9: normAdjust8x8 v(m, 0..5) */
static const uint8_t D_V8[6][6] = {{20, 18, 32, 19, 25, 24}, {22, 19, 35, 21, 28, 26}, {26, 23, 42, 24, 33, 31},
                                   {28, 25, 45, 26, 35, 33}, {32, 28, 51, 30, 40, 38}, {36, 32, 58, 34, 46, 43}};
/* Table 8-16: alpha' and beta' as functions of indexA / indexB */
static const uint8_t D_ALPHA[52] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 4, 4, 5, 6, 7, 8, 9, 10, 12, 13,
                                    15, 17, 20, 22, 25, 28, 32, 36, 40, 45, 50, 56, 63, 71, 80, 90, 101, 113, 127, 144, 162, 182, 203, 226, 255, 255};
static const uint8_t D_BETA[52] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4,
                                   6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13, 14, 14, 15, 15, 16, 16, 17, 17, 18, 18};
/* Table 8-17: t'C0 by bS (row) and indexA (column) */
static const uint8_t D_TC0[3][52] = {
    {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 4, 4, 4, 5, 6, 6, 7, 8, 9, 10, 11, 13},
    {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 4, 4, 5, 5, 6, 7, 8, 8, 10, 11, 12, 13, 15, 17},
    {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 4, 4, 4, 5, 6, 6, 7, 8, 9, 10, 11, 13, 14, 16, 18, 20, 23, 25}};
/* Table 8-15: QPc for qPI >= 30 (below 30 QPc = qPI) */
static const uint8_t D_QPC_HIGH[22] = {29, 30, 31, 32, 32, 33, 34, 34, 35, 35, 36, 36, 37, 37, 37, 38, 38, 38, 39, 39, 39, 39};

/* ------------------------------------------------------------------ bit reader, 7.2 / 9.1 */
typedef struct { const uint8_t *p; size_t nbits, pos; int err; } bitr;

static unsigned rd_bit(bitr *b)
{
    if (b->pos >= b->nbits) { b->err = 1; b->pos++; return 0; }
    unsigned v = (b->p[b->pos >> 3] >> (7 - (b->pos & 7))) & 1u;
    b->pos++;
    return v;
}
static uint32_t rd_u(bitr *b, int n)
{
    uint32_t v = 0;
    while (n-- > 0) v = (v << 1) | rd_bit(b);
    return v;
}
static uint32_t rd_ue(bitr *b) /* 9.1: leadingZeroBits, codeNum = 2^lz - 1 + read_bits(lz) */
{
    int lz = 0;
    while (!rd_bit(b)) { if (++lz > 31 || b->err) { b->err = 1; return 0; } }
    return (uint32_t)(((uint64_t)1 << lz) - 1 + rd_u(b, lz));
}
static int32_t rd_se(bitr *b) /* 9.1.1: (-1)^(k+1) Ceil(k / 2) */
{
    uint32_t k = rd_ue(b);
    int32_t m = (int32_t)((k + 1) / 2);
    return (k & 1) ? m : -m;
}
static uint32_t rd_te(bitr *b, int range) /* 9.1: te(v) with cMax = range */
{
    if (range > 1) return rd_ue(b);
    return rd_bit(b) ? 0u : 1u;
}
/* 7.2 more_rbsp_data(): is there more than the rbsp_trailing_bits left? */
static int more_rbsp_data(const bitr *b)
{
    size_t last = b->nbits;
    while (last > 0 && !((b->p[(last - 1) >> 3] >> (7 - ((last - 1) & 7))) & 1)) last--;
    /* `last` is one past the stop bit */
    return last > 0 && b->pos < last - 1;
}
/* match one of n bit strings at the read position; returns its index or -1 */
static int rd_vlc(bitr *b, const char *const *codes, int n)
{
    for (int i = 0; i < n; i++) {
        const char *s = codes[i];
        if (!s || s[0] == '-') continue;
        size_t k = 0;
        for (; s[k]; k++) {
            size_t q = b->pos + k;
            int bit = q < b->nbits ? (b->p[q >> 3] >> (7 - (q & 7))) & 1 : 2;
            if (bit != s[k] - '0') break;
        }
        if (!s[k]) { b->pos += k; return i; }
    }
    b->err = 1;
    return -1;
}

/* ------------------------------------------------------------------ decoder state */
enum { DMB_NONE = 0, DMB_I4, DMB_I16, DMB_IPCM, DMB_INTER, DMB_SKIP };
enum { D_MAXREF = 16 };

typedef struct {
    uint8_t kind;          /* DMB_* */
    uint8_t t8x8;          /* transform_size_8x8_flag */
    uint8_t i16_mode, chroma_mode, cbp;
    int8_t qp;             /* QP_Y of this macroblock */
    int16_t slice;         /* slice index inside the picture */
    uint8_t i4mode[16];    /* Intra4x4PredMode by raster 4x4 position y*4+x */
    uint8_t tcl[16];       /* total_coeff of the luma 4x4 blocks by raster position (9.2.1 nA/nB) */
    uint8_t tcc[2][4];     /* chroma AC total_coeff, [plane][2*y + x] */
    uint8_t nzl[16];       /* 8.7.2.1: "the 4x4 (8x8 with transform_size_8x8_flag) luma block contains non-zero coefficients" */
    int16_t mv[16][2];     /* per 4x4 block, raster */
    int8_t refidx[4];      /* per 8x8 quadrant, -1 = not inter */
    int32_t refpic[4];     /* identity of the reference picture per quadrant (8.7.2.1 compares pictures, not indices) */
    uint16_t bits;         /* bits of macroblock_layer() (A.3.1 limit 3200) */
} dmb;

typedef struct { uint8_t *pl[3]; int frame_num; int32_t id; } dpic;

struct h264o_dec {
    /* SPS */
    int have_sps, profile, level, log2_max_frame_num, poc_type, log2_max_poc_lsb, max_refs;
    int mbw, mbh, crop_r, crop_b;
    /* PPS */
    int have_pps, init_qp, cqp_off[2], dbf_ctrl, num_ref_default, t8x8_mode, constrained_intra;
    /* picture */
    int cw, ch;
    dpic cur, out;             /* `out`: last finished picture (returned by h264o_dec_plane) */
    dpic refs[D_MAXREF];       /* short-term reference frames, most recent first */
    int nrefs;
    int list0[D_MAXREF + 1];   /* RefPicList0 of the current slice: indices into refs[], -1 = "no reference picture" */
    int32_t next_id;
    dmb *mb;
    int slice_type, nal_type, slice_count;
    int pic_open;              /* macroblocks of the current picture decoded so far */
    int dbf_idc_pic, dbf_a_pic, dbf_b_pic;   /* filter parameters (the streams read here use one set per picture) */
    int8_t *slice_dbf;         /* per slice: idc | (alpha offset + 6) << 2 | ... kept simple: arrays below */
    int slice_idc[256], slice_oa[256], slice_ob[256];
    int cur_is_ref, cur_frame_num, cur_is_idr;
    int max_mb_bits, max_level_prefix;
    char err[200];
};

h264o_dec *h264o_dec_create(void) { return (h264o_dec *)calloc(1, sizeof(h264o_dec)); }
static void free_pic(dpic *p) { for (int i = 0; i < 3; i++) { free(p->pl[i]); p->pl[i] = NULL; } }
void h264o_dec_destroy(h264o_dec *d)
{
    if (!d) return;
    free_pic(&d->cur); free_pic(&d->out);
    for (int i = 0; i < D_MAXREF; i++) free_pic(&d->refs[i]);
    free(d->mb);
    free(d);
}
int h264o_dec_width(const h264o_dec *d) { return d->cw - 2 * d->crop_r; }
int h264o_dec_height(const h264o_dec *d) { return d->ch - 2 * d->crop_b; }
int h264o_dec_coded_width(const h264o_dec *d) { return d->cw; }
int h264o_dec_coded_height(const h264o_dec *d) { return d->ch; }
const uint8_t *h264o_dec_plane(const h264o_dec *d, int p) { return d->out.pl[p]; }
const char *h264o_dec_error(const h264o_dec *d) { return d->err; }
int h264o_dec_last_slice_type(const h264o_dec *d) { return d->slice_type; }
int h264o_dec_last_nal_type(const h264o_dec *d) { return d->nal_type; }
/* statistics of the last decoded picture, for tests: kind of macroblock `addr` (DMB_*), the largest
 * macroblock_layer() in bits (A.3.1: <= 3200), the largest level_prefix met (A.2: <= 15 outside the High profiles) */
int h264o_dec_mb_kind(const h264o_dec *d, int addr) { return addr >= 0 && addr < d->mbw * d->mbh ? d->mb[addr].kind : -1; }
int h264o_dec_mb_qp(const h264o_dec *d, int addr) { return addr >= 0 && addr < d->mbw * d->mbh ? d->mb[addr].qp : -1; }   /* QP_Y (0 for I_PCM) */
int h264o_dec_max_mb_bits(const h264o_dec *d) { return d->max_mb_bits; }
int h264o_dec_max_level_prefix(const h264o_dec *d) { return d->max_level_prefix; }
int h264o_dec_mb_mv(const h264o_dec *d, int addr, int blk4, int *mvx, int *mvy, int *ref)
{
    if (addr < 0 || addr >= d->mbw * d->mbh || blk4 < 0 || blk4 > 15) return -1;
    const dmb *m = &d->mb[addr];
    *mvx = m->mv[blk4][0]; *mvy = m->mv[blk4][1];
    *ref = m->refidx[(blk4 >> 3) * 2 + ((blk4 >> 1) & 1)];
    return 0;
}

static int fail(h264o_dec *d, const char *msg)
{
    snprintf(d->err, sizeof(d->err), "%s", msg);
    return -1;
}
static int clip3i(int lo, int hi, int v) { return v < lo ? lo : v > hi ? hi : v; }
static int clip1y(int v) { return v < 0 ? 0 : v > 255 ? 255 : v; }

static int alloc_pic(const h264o_dec *d, dpic *p)
{
    size_t ysz = (size_t)d->cw * d->ch;
    for (int i = 0; i < 3; i++) {
        free(p->pl[i]);
        p->pl[i] = (uint8_t *)calloc(i ? ysz / 4 : ysz, 1);
        if (!p->pl[i]) return -1;
    }
    return 0;
}

/* ------------------------------------------------------------------ 7.3.2.1 / 7.3.2.2 parameter sets */
static int parse_sps(h264o_dec *d, bitr *b)
{
    d->profile = (int)rd_u(b, 8);
    rd_u(b, 8); /* constraint flags + reserved */
    d->level = (int)rd_u(b, 8);
    if (rd_ue(b) != 0) return fail(d, "seq_parameter_set_id != 0");
    if (d->profile == 100 || d->profile == 110 || d->profile == 122 || d->profile == 244 || d->profile == 44 ||
        d->profile == 83 || d->profile == 86 || d->profile == 118 || d->profile == 128) {
        if (rd_ue(b) != 1) return fail(d, "chroma_format_idc != 1");
        if (rd_ue(b) || rd_ue(b)) return fail(d, "bit depth != 8");
        rd_bit(b); /* qpprime_y_zero_transform_bypass_flag */
        if (rd_bit(b)) return fail(d, "seq_scaling_matrix_present_flag unsupported");
    }
    d->log2_max_frame_num = (int)rd_ue(b) + 4;
    d->poc_type = (int)rd_ue(b);
    if (d->poc_type == 0) d->log2_max_poc_lsb = (int)rd_ue(b) + 4;
    else if (d->poc_type == 1) return fail(d, "pic_order_cnt_type 1 unsupported");
    d->max_refs = (int)rd_ue(b);
    if (d->max_refs > D_MAXREF) return fail(d, "max_num_ref_frames > 16");
    rd_bit(b); /* gaps_in_frame_num_value_allowed_flag */
    int mbw = (int)rd_ue(b) + 1, mbh = (int)rd_ue(b) + 1;
    if (!rd_bit(b)) return fail(d, "frame_mbs_only_flag = 0 unsupported");
    rd_bit(b); /* direct_8x8_inference_flag */
    d->crop_r = d->crop_b = 0;
    if (rd_bit(b)) {
        if (rd_ue(b)) return fail(d, "frame_crop_left_offset unsupported");
        d->crop_r = (int)rd_ue(b);
        if (rd_ue(b)) return fail(d, "frame_crop_top_offset unsupported");
        d->crop_b = (int)rd_ue(b);
    }
    if (b->err) return fail(d, "sps truncated");
    if (mbw != d->mbw || mbh != d->mbh || !d->mb) {
        d->mbw = mbw; d->mbh = mbh; d->cw = 16 * mbw; d->ch = 16 * mbh;
        free(d->mb);
        d->mb = (dmb *)calloc((size_t)mbw * mbh, sizeof(dmb));
        if (!d->mb || alloc_pic(d, &d->cur) || alloc_pic(d, &d->out)) return fail(d, "out of memory");
        for (int i = 0; i < D_MAXREF; i++) free_pic(&d->refs[i]);
        d->nrefs = 0;
    }
    d->have_sps = 1;
    return 0;
}

static int parse_pps(h264o_dec *d, bitr *b)
{
    if (rd_ue(b) || rd_ue(b)) return fail(d, "pic/seq_parameter_set_id != 0");
    if (rd_bit(b)) return fail(d, "entropy_coding_mode_flag = 1 (CABAC) unsupported by the test decoder");
    rd_bit(b); /* bottom_field_pic_order_in_frame_present_flag */
    if (rd_ue(b)) return fail(d, "slice groups unsupported");
    d->num_ref_default = (int)rd_ue(b) + 1;
    rd_ue(b); /* num_ref_idx_l1_default_active_minus1 */
    if (rd_bit(b)) return fail(d, "weighted_pred_flag unsupported");
    rd_u(b, 2);
    d->init_qp = 26 + rd_se(b);
    rd_se(b); /* pic_init_qs */
    d->cqp_off[0] = d->cqp_off[1] = rd_se(b);
    d->dbf_ctrl = (int)rd_bit(b);
    d->constrained_intra = (int)rd_bit(b);   /* constrained_intra_pred_flag */
    if (rd_bit(b)) return fail(d, "redundant_pic_cnt_present_flag unsupported");
    d->t8x8_mode = 0;
    if (b->err) return fail(d, "pps truncated");
    if (more_rbsp_data(b)) {
        d->t8x8_mode = (int)rd_bit(b);
        if (rd_bit(b)) return fail(d, "pic_scaling_matrix_present_flag unsupported");
        d->cqp_off[1] = rd_se(b);
        if (b->err) return fail(d, "pps truncated");
    }
    d->have_pps = 1;
    return 0;
}

/* ------------------------------------------------------------------ 9.2 CAVLC residual block */
/* returns TotalCoeff; coef[0..max-1] = coeffLevel in scan order (9.2.4) */
static int residual_block(h264o_dec *d, bitr *b, int nC, int max_coeff, int *coef)
{
    for (int i = 0; i < max_coeff; i++) coef[i] = 0;
    const int col = nC < 0 ? 4 : nC < 2 ? 0 : nC < 4 ? 1 : nC < 8 ? 2 : 3;
    const char *codes[D_NTOKEN];
    for (int i = 0; i < D_NTOKEN; i++) codes[i] = D_COEFF_TOKEN[i].c[col];
    int k = rd_vlc(b, codes, D_NTOKEN);
    if (k < 0) return 0;
    const int total = D_COEFF_TOKEN[k].tc, t1s = D_COEFF_TOKEN[k].t1;
    if (total > max_coeff) { b->err = 1; return 0; }
    if (!total) return 0;
    int level[16];
    int suffixLength = (total > 10 && t1s < 3) ? 1 : 0;   /* 9.2.2 */
    for (int i = 0; i < total; i++) {
        if (i < t1s) { level[i] = 1 - 2 * (int)rd_bit(b); continue; }
        int level_prefix = 0;                              /* 9.2.2.1 */
        while (!rd_bit(b)) { if (++level_prefix > 25 || b->err) { b->err = 1; return 0; } }
        if (level_prefix > d->max_level_prefix) d->max_level_prefix = level_prefix;
        int levelCode = (level_prefix < 15 ? level_prefix : 15) << suffixLength;
        int levelSuffixSize = (level_prefix == 14 && suffixLength == 0) ? 4 : level_prefix >= 15 ? level_prefix - 3 : suffixLength;
        if (levelSuffixSize > 0) levelCode += (int)rd_u(b, levelSuffixSize);
        if (level_prefix >= 15 && suffixLength == 0) levelCode += 15;
        if (level_prefix >= 16) levelCode += (1 << (level_prefix - 3)) - 4096;
        if (i == t1s && t1s < 3) levelCode += 2;
        level[i] = (levelCode % 2 == 0) ? (levelCode + 2) >> 1 : (-levelCode - 1) >> 1;
        if (suffixLength == 0) suffixLength = 1;
        if (abs(level[i]) > (3 << (suffixLength - 1)) && suffixLength < 6) suffixLength++;
    }
    int zerosLeft = 0;                                       /* 9.2.3 */
    if (total < max_coeff) {
        int tz;
        if (nC < 0) tz = rd_vlc(b, D_TOTAL_ZEROS_CDC[total - 1], 4);
        else tz = rd_vlc(b, D_TOTAL_ZEROS[total - 1], 16);
        if (tz < 0) return 0;
        zerosLeft = tz;
    }
    int run[16];
    for (int i = 0; i < total - 1; i++) {
        if (zerosLeft > 0) {
            int r = rd_vlc(b, D_RUN_BEFORE[(zerosLeft > 6 ? 7 : zerosLeft) - 1], 15);
            if (r < 0) return 0;
            run[i] = r;
        } else run[i] = 0;
        zerosLeft -= run[i];
        if (zerosLeft < 0) { b->err = 1; return 0; }
    }
    run[total - 1] = zerosLeft;
    int coeffNum = -1;                                        /* 9.2.4 */
    for (int i = total - 1; i >= 0; i--) {
        coeffNum += run[i] + 1;
        if (coeffNum >= max_coeff) { b->err = 1; return 0; }
        coef[coeffNum] = level[i];
    }
    return total;
}

/* ------------------------------------------------------------------ neighbour availability 6.4.x */
static const dmb *mb_at(const h264o_dec *d, int mx, int my, int cur_slice, int cur_addr)
{
    if (mx < 0 || my < 0 || mx >= d->mbw || my >= d->mbh) return NULL;
    const int a = my * d->mbw + mx;
    if (a >= cur_addr) return NULL;                 /* not yet decoded (6.4.8: addresses above the current are not available) */
    const dmb *m = &d->mb[a];
    return (m->kind != DMB_NONE && m->slice == cur_slice) ? m : NULL;
}
/* 9.2.1: nC of luma 4x4 block at raster (x, y) / chroma block of plane pl */
static int total_coeff_of(const dmb *m, int comp, int x, int y)
{
    if (m->kind == DMB_IPCM) return 16;
    if (m->kind == DMB_SKIP) return 0;
    return comp == 0 ? m->tcl[4 * y + x] : m->tcc[comp - 1][2 * y + x];
}
static int derive_nC(const h264o_dec *d, const dmb *cur, int mx, int my, int sl, int addr, int comp, int x, int y)
{
    const int lim = comp ? 1 : 3;
    int availA = 0, availB = 0, nA = 0, nB = 0;
    if (x > 0) { availA = 1; nA = total_coeff_of(cur, comp, x - 1, y); }
    else { const dmb *a = mb_at(d, mx - 1, my, sl, addr); if (a) { availA = 1; nA = total_coeff_of(a, comp, lim, y); } }
    if (y > 0) { availB = 1; nB = total_coeff_of(cur, comp, x, y - 1); }
    else { const dmb *bm = mb_at(d, mx, my - 1, sl, addr); if (bm) { availB = 1; nB = total_coeff_of(bm, comp, x, lim); } }
    if (availA && availB) return (nA + nB + 1) >> 1;
    if (availA) return nA;
    if (availB) return nB;
    return 0;
}

/* ------------------------------------------------------------------ scaling and transforms 8.5 */
static int qpc_of(const h264o_dec *d, int qpy, int pl) /* 8.5.8 with Table 8-15 */
{
    const int qpi = clip3i(0, 51, qpy + d->cqp_off[pl]);
    return qpi < 30 ? qpi : D_QPC_HIGH[qpi - 30];
}
static int level_scale4(int m, int i, int j) /* LevelScale4x4 with Flat_4x4_16: 16 * normAdjust4x4(m, i, j) */
{
    const int v = (i % 2 == 0 && j % 2 == 0) ? D_V4[m][0] : (i % 2 == 1 && j % 2 == 1) ? D_V4[m][1] : D_V4[m][2];
    return 16 * v;
}
static int level_scale8(int m, int i, int j) /* LevelScale8x8 with Flat_8x8_16 */
{
    int k;
    if (i % 4 == 0 && j % 4 == 0) k = 0;
    else if (i % 2 == 1 && j % 2 == 1) k = 1;
    else if (i % 4 == 2 && j % 4 == 2) k = 2;
    else if ((i % 4 == 0 && j % 2 == 1) || (i % 2 == 1 && j % 4 == 0)) k = 3;
    else if ((i % 4 == 0 && j % 4 == 2) || (i % 4 == 2 && j % 4 == 0)) k = 4;
    else k = 5;
    return 16 * D_V8[m][k];
}
/* 8.5.12.1 scaling of a 4x4 block c[i][j] (i = row... the standard writes c_ij with i horizontal? it writes d_ij with i the
 * row index of the matrix; normAdjust is symmetric in (i, j), so the distinction does not matter) ; dc_done: d00 is given */
static void scale4x4(const int c[4][4], int qp, int keep_dc, int dcval, int dd[4][4])
{
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            if (keep_dc && i == 0 && j == 0) { dd[0][0] = dcval; continue; }
            const int ls = level_scale4(qp % 6, i, j);
            if (qp >= 24) dd[i][j] = (c[i][j] * ls) << (qp / 6 - 4);
            else dd[i][j] = (c[i][j] * ls + (1 << (3 - qp / 6))) >> (4 - qp / 6);
        }
}
/* 8.5.12.2 */
static void inverse4x4(const int dd[4][4], int r[4][4])
{
    int f[4][4], h[4][4];
    for (int i = 0; i < 4; i++) {
        const int e0 = dd[i][0] + dd[i][2], e1 = dd[i][0] - dd[i][2];
        const int e2 = (dd[i][1] >> 1) - dd[i][3], e3 = dd[i][1] + (dd[i][3] >> 1);
        f[i][0] = e0 + e3; f[i][1] = e1 + e2; f[i][2] = e1 - e2; f[i][3] = e0 - e3;
    }
    for (int j = 0; j < 4; j++) {
        const int g0 = f[0][j] + f[2][j], g1 = f[0][j] - f[2][j];
        const int g2 = (f[1][j] >> 1) - f[3][j], g3 = f[1][j] + (f[3][j] >> 1);
        h[0][j] = g0 + g3; h[1][j] = g1 + g2; h[2][j] = g1 - g2; h[3][j] = g0 - g3;
    }
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) r[i][j] = (h[i][j] + 32) >> 6;
}
/* 8.5.13: 8x8 scaling + inverse transform */
static void one_d8(const int in[8], int out[8])
{
    const int a0 = in[0] + in[4], a1 = -in[3] + in[5] - in[7] - (in[7] >> 1);
    const int a2 = in[0] - in[4], a3 = in[1] + in[7] - in[3] - (in[3] >> 1);
    const int a4 = (in[2] >> 1) - in[6], a5 = -in[1] + in[7] + in[5] + (in[5] >> 1);
    const int a6 = in[2] + (in[6] >> 1), a7 = in[3] + in[5] + in[1] + (in[1] >> 1);
    const int b0 = a0 + a6, b1 = a1 + (a7 >> 2), b2 = a2 + a4, b3 = a3 + (a5 >> 2);
    const int b4 = a2 - a4, b5 = (a3 >> 2) - a5, b6 = a0 - a6, b7 = a7 - (a1 >> 2);
    out[0] = b0 + b7; out[1] = b2 + b5; out[2] = b4 + b3; out[3] = b6 + b1;
    out[4] = b6 - b1; out[5] = b4 - b3; out[6] = b2 - b5; out[7] = b0 - b7;
}
static void scale_inverse8x8(const int c[8][8], int qp, int r[8][8])
{
    int dd[8][8], g[8][8], m[8][8];
    for (int i = 0; i < 8; i++)
        for (int j = 0; j < 8; j++) {
            const int ls = level_scale8(qp % 6, i, j);
            if (qp >= 36) dd[i][j] = (c[i][j] * ls) << (qp / 6 - 6);
            else dd[i][j] = (c[i][j] * ls + (1 << (5 - qp / 6))) >> (6 - qp / 6);
        }
    for (int i = 0; i < 8; i++) one_d8(dd[i], g[i]);            /* each row */
    for (int j = 0; j < 8; j++) {                                /* each column */
        int col[8], o[8];
        for (int i = 0; i < 8; i++) col[i] = g[i][j];
        one_d8(col, o);
        for (int i = 0; i < 8; i++) m[i][j] = o[i];
    }
    for (int i = 0; i < 8; i++)
        for (int j = 0; j < 8; j++) r[i][j] = (m[i][j] + 32) >> 6;
}
static void add_residual(uint8_t *dst, int stride, int n, const int *r /* n x n row-major */)
{
    for (int y = 0; y < n; y++)
        for (int x = 0; x < n; x++) dst[y * stride + x] = (uint8_t)clip1y(dst[y * stride + x] + r[y * n + x]);
}

/* ------------------------------------------------------------------ 8.3 intra prediction */
/* 8.3.1.2: one 4x4 block; p(x, -1) x = -1..7 in top[0..8] (top[0] = corner), p(-1, y) in left[0..3] */
static int intra4x4_pred(int mode, int have_top, int have_left, int have_tl, int have_tr, const uint8_t *rec, int stride, uint8_t out[16])
{
    int T[9], L[4];   /* T[1 + x] = p[x, -1], T[0] = p[-1, -1] */
    for (int x = 0; x < 4; x++) T[1 + x] = have_top ? rec[-stride + x] : 0;
    for (int x = 4; x < 8; x++) T[1 + x] = have_top ? (have_tr ? rec[-stride + x] : rec[-stride + 3]) : 0;
    T[0] = have_tl ? rec[-stride - 1] : 0;
    for (int y = 0; y < 4; y++) L[y] = have_left ? rec[y * stride - 1] : 0;
#define PT(x) T[1 + (x)]
#define PL(y) ((y) < 0 ? T[0] : L[y])
    for (int y = 0; y < 4; y++)
        for (int x = 0; x < 4; x++) {
            int v = 0;
            switch (mode) {
                case 0: if (!have_top) return -1; v = PT(x); break;
                case 1: if (!have_left) return -1; v = PL(y); break;
                case 2:
                    if (have_top && have_left) v = (PT(0) + PT(1) + PT(2) + PT(3) + L[0] + L[1] + L[2] + L[3] + 4) >> 3;
                    else if (have_left) v = (L[0] + L[1] + L[2] + L[3] + 2) >> 2;
                    else if (have_top) v = (PT(0) + PT(1) + PT(2) + PT(3) + 2) >> 2;
                    else v = 128;
                    break;
                case 3: /* Diagonal_Down_Left */
                    if (!have_top) return -1;
                    if (x == 3 && y == 3) v = (PT(6) + 3 * PT(7) + 2) >> 2;
                    else v = (PT(x + y) + 2 * PT(x + y + 1) + PT(x + y + 2) + 2) >> 2;
                    break;
                case 4: /* Diagonal_Down_Right */
                    if (!have_top || !have_left || !have_tl) return -1;
                    if (x > y) v = (PT(x - y - 2) + 2 * PT(x - y - 1) + PT(x - y) + 2) >> 2;
                    else if (x < y) v = (PL(y - x - 2) + 2 * PL(y - x - 1) + PL(y - x) + 2) >> 2;
                    else v = (PT(0) + 2 * PT(-1) + PL(0) + 2) >> 2;
                    break;
                case 5: { /* Vertical_Right */
                    if (!have_top || !have_left || !have_tl) return -1;
                    const int z = 2 * x - y;
                    if (z >= 0 && (z & 1) == 0) v = (PT(x - (y >> 1) - 1) + PT(x - (y >> 1)) + 1) >> 1;
                    else if (z >= 0) v = (PT(x - (y >> 1) - 2) + 2 * PT(x - (y >> 1) - 1) + PT(x - (y >> 1)) + 2) >> 2;
                    else if (z == -1) v = (PL(0) + 2 * PT(-1) + PT(0) + 2) >> 2;
                    else v = (PL(y - 1) + 2 * PL(y - 2) + PL(y - 3) + 2) >> 2;
                    break;
                }
                case 6: { /* Horizontal_Down */
                    if (!have_top || !have_left || !have_tl) return -1;
                    const int z = 2 * y - x;
                    if (z >= 0 && (z & 1) == 0) v = (PL(y - (x >> 1) - 1) + PL(y - (x >> 1)) + 1) >> 1;
                    else if (z >= 0) v = (PL(y - (x >> 1) - 2) + 2 * PL(y - (x >> 1) - 1) + PL(y - (x >> 1)) + 2) >> 2;
                    else if (z == -1) v = (PL(0) + 2 * PT(-1) + PT(0) + 2) >> 2;
                    else v = (PT(x - 1) + 2 * PT(x - 2) + PT(x - 3) + 2) >> 2;
                    break;
                }
                case 7: /* Vertical_Left */
                    if (!have_top) return -1;
                    if ((y & 1) == 0) v = (PT(x + (y >> 1)) + PT(x + (y >> 1) + 1) + 1) >> 1;
                    else v = (PT(x + (y >> 1)) + 2 * PT(x + (y >> 1) + 1) + PT(x + (y >> 1) + 2) + 2) >> 2;
                    break;
                case 8: { /* Horizontal_Up */
                    if (!have_left) return -1;
                    const int z = x + 2 * y;
                    if (z > 5) v = L[3];
                    else if (z == 5) v = (L[2] + 3 * L[3] + 2) >> 2;
                    else if ((z & 1) == 0) v = (L[y + (x >> 1)] + L[y + (x >> 1) + 1] + 1) >> 1;
                    else v = (L[y + (x >> 1)] + 2 * L[y + (x >> 1) + 1] + L[y + (x >> 1) + 2] + 2) >> 2;
                    break;
                }
                default: return -1;
            }
            out[4 * y + x] = (uint8_t)v;
        }
#undef PT
#undef PL
    return 0;
}

/* 8.3.3: Intra16x16; rec points at the macroblock's first sample inside the picture under reconstruction */
static int intra16x16_pred(int mode, int have_top, int have_left, int have_tl, const uint8_t *rec, int stride, uint8_t out[256])
{
#define P(x, y) ((int)rec[(y) * stride + (x)])
    if (mode == 0) {
        if (!have_top) return -1;
        for (int y = 0; y < 16; y++) for (int x = 0; x < 16; x++) out[16 * y + x] = (uint8_t)P(x, -1);
    } else if (mode == 1) {
        if (!have_left) return -1;
        for (int y = 0; y < 16; y++) for (int x = 0; x < 16; x++) out[16 * y + x] = (uint8_t)P(-1, y);
    } else if (mode == 2) {
        int st = 0, sl = 0, v;
        for (int i = 0; i < 16; i++) { if (have_top) st += P(i, -1); if (have_left) sl += P(-1, i); }
        if (have_top && have_left) v = (st + sl + 16) >> 5;
        else if (have_left) v = (sl + 8) >> 4;
        else if (have_top) v = (st + 8) >> 4;
        else v = 128;
        memset(out, v, 256);
    } else if (mode == 3) {
        if (!have_top || !have_left || !have_tl) return -1;
        int H = 0, V = 0;
        for (int k = 0; k <= 7; k++) { H += (k + 1) * (P(8 + k, -1) - P(6 - k, -1)); V += (k + 1) * (P(-1, 8 + k) - P(-1, 6 - k)); }
        const int a = 16 * (P(-1, 15) + P(15, -1)), bb = (5 * H + 32) >> 6, c = (5 * V + 32) >> 6;
        for (int y = 0; y < 16; y++) for (int x = 0; x < 16; x++) out[16 * y + x] = (uint8_t)clip1y((a + bb * (x - 7) + c * (y - 7) + 16) >> 5);
    } else return -1;
    return 0;
}
/* 8.3.4: chroma, 4:2:0 (8x8) */
static int intra_chroma_pred(int mode, int have_top, int have_left, int have_tl, const uint8_t *rec, int stride, uint8_t out[64])
{
    if (mode == 0) {
        for (int by = 0; by < 2; by++)
            for (int bx = 0; bx < 2; bx++) {
                int st = 0, sl = 0, v;
                for (int k = 0; k < 4; k++) { if (have_top) st += P(4 * bx + k, -1); if (have_left) sl += P(-1, 4 * by + k); }
                if ((bx == 0 && by == 0) || (bx == 1 && by == 1)) {
                    if (have_top && have_left) v = (st + sl + 4) >> 3;
                    else if (have_left) v = (sl + 2) >> 2;
                    else if (have_top) v = (st + 2) >> 2;
                    else v = 128;
                } else if (bx == 1 && by == 0) {
                    if (have_top) v = (st + 2) >> 2;
                    else if (have_left) v = (sl + 2) >> 2;
                    else v = 128;
                } else {
                    if (have_left) v = (sl + 2) >> 2;
                    else if (have_top) v = (st + 2) >> 2;
                    else v = 128;
                }
                for (int y = 0; y < 4; y++) for (int x = 0; x < 4; x++) out[8 * (4 * by + y) + 4 * bx + x] = (uint8_t)v;
            }
    } else if (mode == 1) {
        if (!have_left) return -1;
        for (int y = 0; y < 8; y++) for (int x = 0; x < 8; x++) out[8 * y + x] = (uint8_t)P(-1, y);
    } else if (mode == 2) {
        if (!have_top) return -1;
        for (int y = 0; y < 8; y++) for (int x = 0; x < 8; x++) out[8 * y + x] = (uint8_t)P(x, -1);
    } else if (mode == 3) {
        if (!have_top || !have_left || !have_tl) return -1;
        int H = 0, V = 0;
        for (int k = 0; k <= 3; k++) { H += (k + 1) * (P(4 + k, -1) - P(2 - k, -1)); V += (k + 1) * (P(-1, 4 + k) - P(-1, 2 - k)); }
        const int a = 16 * (P(-1, 7) + P(7, -1)), bb = (34 * H + 32) >> 6, c = (34 * V + 32) >> 6;
        for (int y = 0; y < 8; y++) for (int x = 0; x < 8; x++) out[8 * y + x] = (uint8_t)clip1y((a + bb * (x - 3) + c * (y - 3) + 16) >> 5);
    } else return -1;
    return 0;
#undef P
}

/* ------------------------------------------------------------------ 8.4.2.2 fractional sample interpolation */
static int ref_luma(const h264o_dec *d, const dpic *r, int x, int y) /* 8-239, 8-240: clipped sample coordinates */
{
    return r->pl[0][(size_t)clip3i(0, d->ch - 1, y) * d->cw + clip3i(0, d->cw - 1, x)];
}
static int tap(int e, int f, int g, int h, int i, int j) { return e - 5 * f + 20 * g + 20 * h - 5 * i + j; }
/* intermediate b1 at full-sample (x, y): horizontal filter; h1: vertical */
static int b1_at(const h264o_dec *d, const dpic *r, int x, int y)
{
    return tap(ref_luma(d, r, x - 2, y), ref_luma(d, r, x - 1, y), ref_luma(d, r, x, y), ref_luma(d, r, x + 1, y), ref_luma(d, r, x + 2, y), ref_luma(d, r, x + 3, y));
}
static int h1_at(const h264o_dec *d, const dpic *r, int x, int y)
{
    return tap(ref_luma(d, r, x, y - 2), ref_luma(d, r, x, y - 1), ref_luma(d, r, x, y), ref_luma(d, r, x, y + 1), ref_luma(d, r, x, y + 2), ref_luma(d, r, x, y + 3));
}
/* 8.4.2.2.1, Table 8-12: predPartL[xL, yL] for the sample whose integer part is G = (x, y), fraction (xF, yF) */
static int luma_sample_interp(const h264o_dec *d, const dpic *r, int x, int y, int xF, int yF)
{
    const int G = ref_luma(d, r, x, y);
    if (xF == 0 && yF == 0) return G;
    const int b = clip1y((b1_at(d, r, x, y) + 16) >> 5);       /* between G and H (right neighbour) */
    const int h = clip1y((h1_at(d, r, x, y) + 16) >> 5);       /* between G and M (below)            */
    const int j1 = tap(b1_at(d, r, x, y - 2), b1_at(d, r, x, y - 1), b1_at(d, r, x, y), b1_at(d, r, x, y + 1), b1_at(d, r, x, y + 2), b1_at(d, r, x, y + 3));
    const int j = clip1y((j1 + 512) >> 10);
    const int s = clip1y((b1_at(d, r, x, y + 1) + 16) >> 5);   /* b of the row below        */
    const int m = clip1y((h1_at(d, r, x + 1, y) + 16) >> 5);   /* h of the column to the right */
    const int H = ref_luma(d, r, x + 1, y), M = ref_luma(d, r, x, y + 1);
    switch (4 * yF + xF) {
        case 1: return (G + b + 1) >> 1;   /* a */
        case 2: return b;
        case 3: return (b + H + 1) >> 1;   /* c */
        case 4: return (G + h + 1) >> 1;   /* d */
        case 5: return (b + h + 1) >> 1;   /* e */
        case 6: return (b + j + 1) >> 1;   /* f */
        case 7: return (b + m + 1) >> 1;   /* g */
        case 8: return h;
        case 9: return (h + j + 1) >> 1;   /* i */
        case 10: return j;
        case 11: return (j + m + 1) >> 1;  /* k */
        case 12: return (h + M + 1) >> 1;  /* n */
        case 13: return (h + s + 1) >> 1;  /* p */
        case 14: return (j + s + 1) >> 1;  /* q */
        default: return (m + s + 1) >> 1;  /* r */
    }
}
/* 8.4.2.2.2 */
static int chroma_sample_interp(const h264o_dec *d, const dpic *r, int pl, int x, int y, int xF, int yF)
{
    const int w = d->cw / 2, hh = d->ch / 2;
    const uint8_t *c = r->pl[pl];
    const int xA = clip3i(0, w - 1, x), xB = clip3i(0, w - 1, x + 1), yA = clip3i(0, hh - 1, y), yC = clip3i(0, hh - 1, y + 1);
    const int A = c[(size_t)yA * w + xA], B = c[(size_t)yA * w + xB], C = c[(size_t)yC * w + xA], D = c[(size_t)yC * w + xB];
    return ((8 - xF) * (8 - yF) * A + xF * (8 - yF) * B + (8 - xF) * yF * C + xF * yF * D + 32) >> 6;
}
/* inter prediction of one partition (x0, y0, w, h in luma samples relative to the picture) */
static void inter_pred_part(h264o_dec *d, const dpic *r, int x0, int y0, int w, int h, int mvx, int mvy)
{
    const int cw = d->cw, cs = cw / 2;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const int xq = 4 * (x0 + x) + mvx, yq = 4 * (y0 + y) + mvy;   /* 8-225 .. 8-228 */
            d->cur.pl[0][(size_t)(y0 + y) * cw + x0 + x] = (uint8_t)luma_sample_interp(d, r, xq >> 2, yq >> 2, xq & 3, yq & 3);
        }
    for (int pl = 1; pl < 3; pl++)
        for (int y = 0; y < h / 2; y++)
            for (int x = 0; x < w / 2; x++) {
                const int xo = 8 * (x0 / 2 + x) + mvx, yo = 8 * (y0 / 2 + y) + mvy;   /* chroma vector = luma vector in 1/8 units */
                d->cur.pl[pl][(size_t)(y0 / 2 + y) * cs + x0 / 2 + x] = (uint8_t)chroma_sample_interp(d, r, pl, xo >> 3, yo >> 3, xo & 7, yo & 7);
            }
}

/* ------------------------------------------------------------------ 8.4.1 motion vector prediction */
typedef struct { int avail, ref, mvx, mvy; } nbr;
/* neighbouring 4x4 block at luma offset (x, y) relative to the current macroblock's origin (6.4.12); `done` marks the 4x4
 * blocks of the current macroblock whose vectors are already derived (a partition later in decoding order is "not available") */
static nbr neighbour_blk(const h264o_dec *d, const dmb *cur, const uint8_t done[16], int mx, int my, int sl, int addr, int x, int y)
{
    nbr n = {0, -1, 0, 0};
    const dmb *m;
    if (x >= 0 && x < 16 && y >= 0 && y < 16) {
        const int bi = 4 * (y >> 2) + (x >> 2);
        if (!done[bi]) return n;
        m = cur;
    } else {
        if (y >= 16) return n;
        if (y >= 0 && x >= 16) return n;                  /* the macroblock to the right is never available */
        const int nmx = mx + (x < 0 ? -1 : x >= 16 ? 1 : 0), nmy = my + (y < 0 ? -1 : 0);
        m = mb_at(d, nmx, nmy, sl, addr);
        if (!m) return n;
        x &= 15; y &= 15;
    }
    n.avail = 1;
    if (m->kind == DMB_INTER || m->kind == DMB_SKIP) {
        const int bi = 4 * (y >> 2) + (x >> 2);
        n.ref = m->refidx[(y >> 3) * 2 + (x >> 3)];
        n.mvx = m->mv[bi][0]; n.mvy = m->mv[bi][1];
    }
    return n;
}
static int median3(int a, int b, int c)
{
    const int lo = a < b ? a : b, hi = a < b ? b : a;
    return c < lo ? lo : c > hi ? hi : c;
}
/* 8.4.1.3 for the partition at (x, y) of size (w, h) inside the macroblock with reference index `ref`;
 * shape: 0 other, 1 = 16x8, 2 = 8x16 (directional rules), part = mbPartIdx */
static void predict_mv(const h264o_dec *d, const dmb *cur, const uint8_t done[16], int mx, int my, int sl, int addr,
                       int x, int y, int w, int h, int ref, int shape, int part, int *px, int *py)
{
    (void)h;
    nbr A = neighbour_blk(d, cur, done, mx, my, sl, addr, x - 1, y);
    nbr B = neighbour_blk(d, cur, done, mx, my, sl, addr, x, y - 1);
    nbr C = neighbour_blk(d, cur, done, mx, my, sl, addr, x + w, y - 1);
    if (!C.avail) C = neighbour_blk(d, cur, done, mx, my, sl, addr, x - 1, y - 1);   /* 8.4.1.3.2: D replaces C */
    if (shape == 1) {
        if (part == 0 && B.ref == ref) { *px = B.mvx; *py = B.mvy; return; }
        if (part == 1 && A.ref == ref) { *px = A.mvx; *py = A.mvy; return; }
    } else if (shape == 2) {
        if (part == 0 && A.ref == ref) { *px = A.mvx; *py = A.mvy; return; }
        if (part == 1 && C.ref == ref) { *px = C.mvx; *py = C.mvy; return; }
    }
    /* 8.4.1.3.1 median prediction */
    if (!B.avail && !C.avail && A.avail) { B = A; C = A; }
    const int eq = (A.ref == ref) + (B.ref == ref) + (C.ref == ref);
    if (eq == 1) {
        const nbr *s = A.ref == ref ? &A : B.ref == ref ? &B : &C;
        *px = s->mvx; *py = s->mvy;
    } else {
        *px = median3(A.mvx, B.mvx, C.mvx);
        *py = median3(A.mvy, B.mvy, C.mvy);
    }
}

/* ------------------------------------------------------------------ macroblock layer 7.3.5 */
static const dpic *ref_of(h264o_dec *d, int idx, int nactive)
{
    if (idx < 0 || idx >= nactive || d->list0[idx] < 0) return NULL;
    return &d->refs[d->list0[idx]];
}

/* residual_luma + reconstruction for non-Intra16x16 macroblocks: blocks in blkIdx order; for intra4x4 the prediction of a
 * block is made right before its residual is added (pred_cb) */
typedef struct { h264o_dec *d; bitr *b; dmb *m; int mx, my, sl, addr; } mbctx;

static int blk_x4(int blkIdx) { return (blkIdx & 1) + 2 * ((blkIdx >> 2) & 1); }   /* 6.4.3, in units of 4 samples */
static int blk_y4(int blkIdx) { return ((blkIdx >> 1) & 1) + 2 * (blkIdx >> 3); }

static int chroma_residual(mbctx *c, int qpy)
{
    h264o_dec *d = c->d;
    dmb *m = c->m;
    const int cbpc = m->cbp >> 4, cs = d->cw / 2;
    int dc[2][4] = {{0}}, ac[2][4][16];
    memset(ac, 0, sizeof(ac));
    if (cbpc & 3)
        for (int pl = 0; pl < 2; pl++) residual_block(d, c->b, -1, 4, dc[pl]);
    for (int pl = 0; pl < 2; pl++)
        for (int k = 0; k < 4; k++) {
            int tc = 0;
            if (cbpc & 2) tc = residual_block(d, c->b, derive_nC(d, m, c->mx, c->my, c->sl, c->addr, 1 + pl, k & 1, k >> 1), 15, ac[pl][k] + 1);
            m->tcc[pl][k] = (uint8_t)tc;
        }
    if (c->b->err) return -1;
    for (int pl = 0; pl < 2; pl++) {
        const int qpc = qpc_of(d, qpy, pl);
        /* 8.5.11.1: c = [[c0 c1][c2 c3]] (raster 2x2 = chroma DC scan), f = A c A with A = [[1 1][1 -1]] */
        const int c0 = dc[pl][0], c1 = dc[pl][1], c2 = dc[pl][2], c3 = dc[pl][3];
        const int f[4] = {c0 + c1 + c2 + c3, c0 - c1 + c2 - c3, c0 + c1 - c2 - c3, c0 - c1 - c2 + c3};
        for (int k = 0; k < 4; k++) {
            const int dcC = ((f[k] * level_scale4(qpc % 6, 0, 0)) << (qpc / 6)) >> 5;   /* 8.5.11.2 */
            int cc[4][4], dd[4][4], r[4][4];
            memset(cc, 0, sizeof(cc));
            for (int i = 1; i < 16; i++) cc[D_ZZ4[i][1]][D_ZZ4[i][0]] = ac[pl][k][i];
            scale4x4(cc, qpc, 1, dcC, dd);
            inverse4x4(dd, r);
            add_residual(d->cur.pl[1 + pl] + (size_t)(8 * c->my + 4 * (k >> 1)) * cs + 8 * c->mx + 4 * (k & 1), cs, 4, &r[0][0]);
        }
    }
    return 0;
}

static int decode_intra_mb(mbctx *c, int mbt /* I-slice mb_type */, int *qp)
{
    h264o_dec *d = c->d;
    bitr *b = c->b;
    dmb *m = c->m;
    const int cw = d->cw, cs = cw / 2, mx = c->mx, my = c->my;
    uint8_t *Y = d->cur.pl[0] + (size_t)16 * my * cw + 16 * mx;
    const dmb *mA = mb_at(d, mx - 1, my, c->sl, c->addr), *mB = mb_at(d, mx, my - 1, c->sl, c->addr);
    const dmb *mC = mb_at(d, mx + 1, my - 1, c->sl, c->addr), *mD = mb_at(d, mx - 1, my - 1, c->sl, c->addr);
    if (d->constrained_intra) {
        /* 8.3.1.1 / 8.3.1.2 / 8.3.3 / 8.3.4: with constrained_intra_pred_flag a macroblock coded in Inter prediction mode is "not
         * available" for intra prediction (nor for the derivation of Intra4x4PredMode) */
        if (mA && (mA->kind == DMB_INTER || mA->kind == DMB_SKIP)) mA = NULL;
        if (mB && (mB->kind == DMB_INTER || mB->kind == DMB_SKIP)) mB = NULL;
        if (mC && (mC->kind == DMB_INTER || mC->kind == DMB_SKIP)) mC = NULL;
        if (mD && (mD->kind == DMB_INTER || mD->kind == DMB_SKIP)) mD = NULL;
    }
    for (int i = 0; i < 4; i++) { m->refidx[i] = -1; m->refpic[i] = -1; }
    if (mbt == 25) { /* I_PCM, 7.3.5 + 8.3.5 */
        m->kind = DMB_IPCM;
        while (b->pos & 7) if (rd_bit(b)) return fail(d, "pcm_alignment_zero_bit != 0");
        for (int y = 0; y < 16; y++) for (int x = 0; x < 16; x++) Y[y * cw + x] = (uint8_t)rd_u(b, 8);
        for (int pl = 1; pl < 3; pl++)
            for (int y = 0; y < 8; y++) for (int x = 0; x < 8; x++) d->cur.pl[pl][(size_t)(8 * my + y) * cs + 8 * mx + x] = (uint8_t)rd_u(b, 8);
        memset(m->tcl, 16, 16); memset(m->tcc, 16, 8); memset(m->nzl, 1, 16);
        m->qp = 0;          /* 8.7.2.2: qPp of an I_PCM macroblock is 0; QP_Y,PRED of the next macroblock is unchanged */
        m->cbp = 0x2F;
        return b->err ? fail(d, "I_PCM truncated") : 0;
    }
    if (mbt == 0) { /* I_NxN */
        m->kind = DMB_I4;
        if (d->t8x8_mode && rd_bit(b)) return fail(d, "Intra8x8 (transform_size_8x8_flag in I_NxN) unsupported by the test decoder");
        int prev[16], rem[16];
        for (int k = 0; k < 16; k++) { prev[k] = (int)rd_bit(b); rem[k] = prev[k] ? 0 : (int)rd_u(b, 3); }
        for (int k = 0; k < 16; k++) {   /* 8.3.1.1 */
            const int x = blk_x4(k), y = blk_y4(k);
            int modeA, modeB, dcOnly = 0;
            if (x > 0) modeA = m->i4mode[4 * y + x - 1];
            else if (!mA) { dcOnly = 1; modeA = 2; }
            else modeA = mA->kind == DMB_I4 ? mA->i4mode[4 * y + 3] : 2;
            if (y > 0) modeB = m->i4mode[4 * (y - 1) + x];
            else if (!mB) { dcOnly = 1; modeB = 2; }
            else modeB = mB->kind == DMB_I4 ? mB->i4mode[12 + x] : 2;
            if (dcOnly) modeA = modeB = 2;
            const int pred = modeA < modeB ? modeA : modeB;
            m->i4mode[4 * y + x] = (uint8_t)(prev[k] ? pred : (rem[k] < pred ? rem[k] : rem[k] + 1));
        }
    } else {
        m->kind = DMB_I16;
        const int t = mbt - 1;
        m->i16_mode = (uint8_t)(t & 3);
        m->cbp = (uint8_t)((t >= 12 ? 15 : 0) | (((t >> 2) % 3) << 4));
    }
    const uint32_t cmode = rd_ue(b);
    if (cmode > 3) return fail(d, "intra_chroma_pred_mode > 3");
    m->chroma_mode = (uint8_t)cmode;
    if (m->kind == DMB_I4) {
        const uint32_t code = rd_ue(b);
        if (code > 47) return fail(d, "coded_block_pattern out of range");
        m->cbp = D_CBP[code][0];
    }
    if (m->kind == DMB_I16 || m->cbp) {
        const int dq = rd_se(b);
        if (dq < -26 || dq > 25) return fail(d, "mb_qp_delta out of range");
        *qp = (*qp + dq + 52) % 52;
    }
    m->qp = (int8_t)*qp;
    const int q = *qp;
    if (m->kind == DMB_I16) {
        int dcs[16], acs[16][16];
        memset(acs, 0, sizeof(acs));
        residual_block(d, b, derive_nC(d, m, mx, my, c->sl, c->addr, 0, 0, 0), 16, dcs);
        for (int k = 0; k < 16; k++) {
            int tc = 0;
            if (m->cbp & 15) tc = residual_block(d, b, derive_nC(d, m, mx, my, c->sl, c->addr, 0, blk_x4(k), blk_y4(k)), 15, acs[k] + 1);
            m->tcl[4 * blk_y4(k) + blk_x4(k)] = (uint8_t)tc;
        }
        if (b->err) return fail(d, "residual parse error (Intra16x16)");
        uint8_t pred[256];
        if (intra16x16_pred(m->i16_mode, mB != NULL, mA != NULL, mD != NULL, Y, cw, pred)) return fail(d, "Intra16x16 mode uses unavailable neighbours");
        for (int y = 0; y < 16; y++) memcpy(Y + y * cw, pred + 16 * y, 16);
        /* 8.5.10: c = 4x4 of DC levels (inverse zig-zag), f = A c A, A rows 1111 / 11-1-1 / 1-1-11 / 1-11-1 */
        int cm[4][4], t[4][4], f[4][4];
        for (int i = 0; i < 16; i++) cm[D_ZZ4[i][1]][D_ZZ4[i][0]] = dcs[i];
        static const int A4[4][4] = {{1, 1, 1, 1}, {1, 1, -1, -1}, {1, -1, -1, 1}, {1, -1, 1, -1}};
        for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) { t[i][j] = 0; for (int k = 0; k < 4; k++) t[i][j] += A4[i][k] * cm[k][j]; }
        for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) { f[i][j] = 0; for (int k = 0; k < 4; k++) f[i][j] += t[i][k] * A4[k][j]; }
        const int ls = level_scale4(q % 6, 0, 0);
        for (int k = 0; k < 16; k++) {
            const int bx = blk_x4(k), by = blk_y4(k);
            const int fi = f[by][bx];
            const int dcY = q >= 36 ? (fi * ls) << (q / 6 - 6) : (fi * ls + (1 << (5 - q / 6))) >> (6 - q / 6);
            int cc[4][4], dd[4][4], r[4][4];
            memset(cc, 0, sizeof(cc));
            for (int i = 1; i < 16; i++) cc[D_ZZ4[i][1]][D_ZZ4[i][0]] = acs[k][i];
            scale4x4(cc, q, 1, dcY, dd);
            inverse4x4(dd, r);
            add_residual(Y + 4 * by * cw + 4 * bx, cw, 4, &r[0][0]);
        }
        memset(m->nzl, 1, 16);
    } else {
        for (int k = 0; k < 16; k++) {
            const int bx = blk_x4(k), by = blk_y4(k);
            int lv[16];
            int tc = 0;
            if (m->cbp & (1 << (k >> 2))) tc = residual_block(d, b, derive_nC(d, m, mx, my, c->sl, c->addr, 0, bx, by), 16, lv);
            else memset(lv, 0, sizeof(lv));
            if (b->err) return fail(d, "residual parse error (Intra4x4)");
            m->tcl[4 * by + bx] = (uint8_t)tc;
            m->nzl[4 * by + bx] = tc != 0;
            /* availability of the block's neighbours (6.4.11.4): inside the macroblock a block is available when it
             * precedes this one in blkIdx order */
            const int have_left = bx > 0 || mA != NULL, have_top = by > 0 || mB != NULL;
            const int have_tl = (bx > 0 && by > 0) ? 1 : (bx > 0 ? mB != NULL : (by > 0 ? mA != NULL : mD != NULL));
            int have_tr;
            if (by == 0) have_tr = bx < 3 ? mB != NULL : mC != NULL;
            else if (bx == 3) have_tr = 0;
            else {   /* block (bx + 1, by - 1) of this macroblock: available iff its blkIdx < k */
                int kk = -1;
                for (int q2 = 0; q2 < 16; q2++) if (blk_x4(q2) == bx + 1 && blk_y4(q2) == by - 1) kk = q2;
                have_tr = kk < k;
            }
            uint8_t pred[16];
            uint8_t *p = Y + 4 * by * cw + 4 * bx;
            if (intra4x4_pred(m->i4mode[4 * by + bx], have_top, have_left, have_tl, have_tr, p, cw, pred)) return fail(d, "Intra4x4 mode uses unavailable neighbours");
            for (int y = 0; y < 4; y++) memcpy(p + y * cw, pred + 4 * y, 4);
            if (tc) {
                int cc[4][4], dd[4][4], r[4][4];
                for (int i = 0; i < 16; i++) cc[D_ZZ4[i][1]][D_ZZ4[i][0]] = lv[i];
                scale4x4(cc, q, 0, 0, dd);
                inverse4x4(dd, r);
                add_residual(p, cw, 4, &r[0][0]);
            }
        }
    }
    for (int pl = 0; pl < 2; pl++) {
        uint8_t pc[64];
        uint8_t *C = d->cur.pl[1 + pl] + (size_t)8 * my * cs + 8 * mx;
        if (intra_chroma_pred(m->chroma_mode, mB != NULL, mA != NULL, mD != NULL, C, cs, pc)) return fail(d, "intra chroma mode uses unavailable neighbours");
        for (int y = 0; y < 8; y++) memcpy(C + y * cs, pc + 8 * y, 8);
    }
    if (chroma_residual(c, q)) return fail(d, "residual parse error (chroma)");
    return 0;
}

static int decode_inter_mb(mbctx *c, int mbt /* P mb_type 0..4, -1 = P_Skip */, int nactive, int *qp)
{
    h264o_dec *d = c->d;
    bitr *b = c->b;
    dmb *m = c->m;
    const int cw = d->cw, mx = c->mx, my = c->my;
    uint8_t done[16];
    memset(done, 0, sizeof(done));
    m->kind = mbt < 0 ? DMB_SKIP : DMB_INTER;
    /* partition geometry: list of (x, y, w, h, quadrant-for-ref, shape, partidx) */
    struct part { int x, y, w, h, q, shape, idx; } parts[16];
    int np = 0, sub[4] = {0, 0, 0, 0};
    if (mbt < 0) {   /* 8.4.1.1 */
        nbr A = neighbour_blk(d, m, done, mx, my, c->sl, c->addr, -1, 0), B = neighbour_blk(d, m, done, mx, my, c->sl, c->addr, 0, -1);
        int px = 0, py = 0;
        if (A.avail && B.avail && !(A.ref == 0 && A.mvx == 0 && A.mvy == 0) && !(B.ref == 0 && B.mvx == 0 && B.mvy == 0))
            predict_mv(d, m, done, mx, my, c->sl, c->addr, 0, 0, 16, 16, 0, 0, 0, &px, &py);
        for (int i = 0; i < 16; i++) { m->mv[i][0] = (int16_t)px; m->mv[i][1] = (int16_t)py; }
        for (int i = 0; i < 4; i++) m->refidx[i] = 0;
        m->qp = (int8_t)*qp;
    } else {
        if (mbt == 0) parts[np++] = (struct part){0, 0, 16, 16, 0, 0, 0};
        else if (mbt == 1) { parts[np++] = (struct part){0, 0, 16, 8, 0, 1, 0}; parts[np++] = (struct part){0, 8, 16, 8, 2, 1, 1}; }
        else if (mbt == 2) { parts[np++] = (struct part){0, 0, 8, 16, 0, 2, 0}; parts[np++] = (struct part){8, 0, 8, 16, 1, 2, 1}; }
        else {
            for (int s = 0; s < 4; s++) { sub[s] = (int)rd_ue(b); if (sub[s] > 3) return fail(d, "sub_mb_type out of range"); }
            for (int s = 0; s < 4; s++) {
                const int ox = 8 * (s & 1), oy = 8 * (s >> 1);
                if (sub[s] == 0) parts[np++] = (struct part){ox, oy, 8, 8, s, 0, 0};
                else if (sub[s] == 1) { parts[np++] = (struct part){ox, oy, 8, 4, s, 0, 0}; parts[np++] = (struct part){ox, oy + 4, 8, 4, s, 0, 1}; }
                else if (sub[s] == 2) { parts[np++] = (struct part){ox, oy, 4, 8, s, 0, 0}; parts[np++] = (struct part){ox + 4, oy, 4, 8, s, 0, 1}; }
                else for (int k = 0; k < 4; k++) parts[np++] = (struct part){ox + 4 * (k & 1), oy + 4 * (k >> 1), 4, 4, s, 0, k};
            }
        }
        /* ref_idx_l0 for every macroblock partition first, then the vector differences (7.3.5.1 / 7.3.5.2) */
        int refq[4] = {0, 0, 0, 0};
        if (mbt != 4 && nactive > 1) {
            if (mbt == 0) { const int r = (int)rd_te(b, nactive - 1); refq[0] = refq[1] = refq[2] = refq[3] = r; }
            else if (mbt == 1) { const int r0 = (int)rd_te(b, nactive - 1), r1 = (int)rd_te(b, nactive - 1); refq[0] = refq[1] = r0; refq[2] = refq[3] = r1; }
            else if (mbt == 2) { const int r0 = (int)rd_te(b, nactive - 1), r1 = (int)rd_te(b, nactive - 1); refq[0] = refq[2] = r0; refq[1] = refq[3] = r1; }
            else for (int s = 0; s < 4; s++) refq[s] = (int)rd_te(b, nactive - 1);
        }
        for (int s = 0; s < 4; s++) {
            if (refq[s] >= nactive) return fail(d, "ref_idx_l0 out of range");
            m->refidx[s] = (int8_t)refq[s];
        }
        for (int i = 0; i < np; i++) {
            const struct part *p = &parts[i];
            int px, py;
            predict_mv(d, m, done, mx, my, c->sl, c->addr, p->x, p->y, p->w, p->h, refq[p->q], p->shape, p->idx, &px, &py);
            const int vx = px + rd_se(b), vy = py + rd_se(b);
            for (int y = p->y; y < p->y + p->h; y += 4)
                for (int x = p->x; x < p->x + p->w; x += 4) {
                    const int bi = 4 * (y >> 2) + (x >> 2);
                    m->mv[bi][0] = (int16_t)vx; m->mv[bi][1] = (int16_t)vy; done[bi] = 1;
                }
        }
        const uint32_t code = rd_ue(b);
        if (code > 47) return fail(d, "coded_block_pattern out of range");
        m->cbp = D_CBP[code][1];
        if ((m->cbp & 15) && d->t8x8_mode) {
            /* 7.3.5: transform_size_8x8_flag for inter macroblocks without sub-8x8 partitions (direct_8x8_inference irrelevant for P) */
            int ok = 1;
            if (mbt >= 3) for (int s = 0; s < 4; s++) if (sub[s] != 0) ok = 0;
            if (ok) m->t8x8 = (uint8_t)rd_bit(b);
        }
        if (m->cbp) {
            const int dq = rd_se(b);
            if (dq < -26 || dq > 25) return fail(d, "mb_qp_delta out of range");
            *qp = (*qp + dq + 52) % 52;
        }
        m->qp = (int8_t)*qp;
        if (b->err) return fail(d, "macroblock header parse error");
    }
    /* prediction: per quadrant / partition with its own reference picture */
    if (mbt < 0) {
        const dpic *r = ref_of(d, 0, nactive);
        if (!r) return fail(d, "P_Skip without a reference picture");
        m->refpic[0] = m->refpic[1] = m->refpic[2] = m->refpic[3] = r->id;
        inter_pred_part(d, r, 16 * mx, 16 * my, 16, 16, m->mv[0][0], m->mv[0][1]);
        return 0;
    }
    for (int s = 0; s < 4; s++) {
        const dpic *r = ref_of(d, m->refidx[s], nactive);
        if (!r) return fail(d, "reference index without a picture");
        m->refpic[s] = r->id;
    }
    for (int i = 0; i < np; i++) {
        const struct part *p = &parts[i];
        const int bi = 4 * (p->y >> 2) + (p->x >> 2);
        inter_pred_part(d, ref_of(d, m->refidx[p->q], nactive), 16 * mx + p->x, 16 * my + p->y, p->w, p->h, m->mv[bi][0], m->mv[bi][1]);
    }
    /* residual */
    const int q = *qp;
    uint8_t *Y = d->cur.pl[0] + (size_t)16 * my * cw + 16 * mx;
    if (m->t8x8) {
        for (int b8 = 0; b8 < 4; b8++) {
            int c8[8][8], nz8 = 0;
            memset(c8, 0, sizeof(c8));
            for (int k4 = 0; k4 < 4; k4++) {   /* 7.3.5.3.2: 4 interleaved 4x4 "blocks" of 16 levels each: level i of block k4 = level 4i + k4 of the 8x8 scan */
                const int k = 4 * b8 + k4, bx = blk_x4(k), by = blk_y4(k);
                int lv[16], tc = 0;
                if (m->cbp & (1 << b8)) tc = residual_block(d, b, derive_nC(d, m, mx, my, c->sl, c->addr, 0, bx, by), 16, lv);
                else memset(lv, 0, sizeof(lv));
                if (b->err) return fail(d, "residual parse error (inter luma 8x8)");
                m->tcl[4 * by + bx] = (uint8_t)tc;
                nz8 |= tc;
                for (int i = 0; i < 16; i++) { const int zi = 4 * i + k4; c8[D_ZZ8[zi][1]][D_ZZ8[zi][0]] = lv[i]; }
            }
            for (int k4 = 0; k4 < 4; k4++) { const int k = 4 * b8 + k4; m->nzl[4 * blk_y4(k) + blk_x4(k)] = nz8 != 0; }
            if (nz8) {
                int r[8][8];
                scale_inverse8x8(c8, q, r);
                add_residual(Y + 8 * (b8 >> 1) * cw + 8 * (b8 & 1), cw, 8, &r[0][0]);
            }
        }
    } else {
        for (int k = 0; k < 16; k++) {
            const int bx = blk_x4(k), by = blk_y4(k);
            if (!(m->cbp & (1 << (k >> 2)))) continue;
            int lv[16];
            const int tc = residual_block(d, b, derive_nC(d, m, mx, my, c->sl, c->addr, 0, bx, by), 16, lv);
            if (b->err) return fail(d, "residual parse error (inter luma)");
            m->tcl[4 * by + bx] = (uint8_t)tc;
            m->nzl[4 * by + bx] = tc != 0;
            if (tc) {
                int cc[4][4], dd[4][4], r[4][4];
                for (int i = 0; i < 16; i++) cc[D_ZZ4[i][1]][D_ZZ4[i][0]] = lv[i];
                scale4x4(cc, q, 0, 0, dd);
                inverse4x4(dd, r);
                add_residual(Y + 4 * by * cw + 4 * bx, cw, 4, &r[0][0]);
            }
        }
    }
    if (chroma_residual(c, q)) return fail(d, "residual parse error (chroma)");
    return 0;
}

/* ------------------------------------------------------------------ 8.7 deblocking filter */
/* 8.7.2.3 / 8.7.2.4: one line of samples; p[i] = pix[-(i+1) * step], q[i] = pix[i * step] */
static void filter_samples(uint8_t *pix, int step, int bS, int chroma, int chromaEdge, int indexA, int alpha, int beta)
{
    (void)chromaEdge;
    const int p0 = pix[-step], p1 = pix[-2 * step], q0 = pix[0], q1 = pix[step];
    if (!(bS != 0 && abs(p0 - q0) < alpha && abs(p1 - p0) < beta && abs(q1 - q0) < beta)) return;   /* filterSamplesFlag 8-468 */
    if (bS < 4) {
        const int tc0 = D_TC0[bS - 1][indexA];
        int tc, ap = 0, aq = 0, p2 = 0, q2 = 0;
        if (!chroma) {
            p2 = pix[-3 * step]; q2 = pix[2 * step];
            ap = abs(p2 - p0); aq = abs(q2 - q0);
            tc = tc0 + (ap < beta ? 1 : 0) + (aq < beta ? 1 : 0);
        } else tc = tc0 + 1;
        const int delta = clip3i(-tc, tc, ((((q0 - p0) << 2) + (p1 - q1) + 4) >> 3));
        pix[-step] = (uint8_t)clip1y(p0 + delta);
        pix[0] = (uint8_t)clip1y(q0 - delta);
        if (!chroma) {
            if (ap < beta) pix[-2 * step] = (uint8_t)(p1 + clip3i(-tc0, tc0, (p2 + ((p0 + q0 + 1) >> 1) - (p1 << 1)) >> 1));
            if (aq < beta) pix[step] = (uint8_t)(q1 + clip3i(-tc0, tc0, (q2 + ((p0 + q0 + 1) >> 1) - (q1 << 1)) >> 1));
        }
    } else {
        if (!chroma) {
            const int p2 = pix[-3 * step], q2 = pix[2 * step], p3 = pix[-4 * step], q3 = pix[3 * step];
            const int ap = abs(p2 - p0), aq = abs(q2 - q0);
            const int small = abs(p0 - q0) < ((alpha >> 2) + 2);
            if (ap < beta && small) {
                pix[-step] = (uint8_t)((p2 + 2 * p1 + 2 * p0 + 2 * q0 + q1 + 4) >> 3);
                pix[-2 * step] = (uint8_t)((p2 + p1 + p0 + q0 + 2) >> 2);
                pix[-3 * step] = (uint8_t)((2 * p3 + 3 * p2 + p1 + p0 + q0 + 4) >> 3);
            } else pix[-step] = (uint8_t)((2 * p1 + p0 + q1 + 2) >> 2);
            if (aq < beta && small) {
                pix[0] = (uint8_t)((p1 + 2 * p0 + 2 * q0 + 2 * q1 + q2 + 4) >> 3);
                pix[step] = (uint8_t)((p0 + q0 + q1 + q2 + 2) >> 2);
                pix[2 * step] = (uint8_t)((2 * q3 + 3 * q2 + q1 + q0 + p0 + 4) >> 3);
            } else pix[0] = (uint8_t)((2 * q1 + q0 + p1 + 2) >> 2);
        } else {
            pix[-step] = (uint8_t)((2 * p1 + p0 + q1 + 2) >> 2);
            pix[0] = (uint8_t)((2 * q1 + q0 + p1 + 2) >> 2);
        }
    }
}
static int is_intra(const dmb *m) { return m->kind == DMB_I4 || m->kind == DMB_I16 || m->kind == DMB_IPCM; }
/* 8.7.2.1 for frame macroblocks of P / I slices: p-side 4x4 block bp of mp, q-side bq of mq (raster indices) */
static int derive_bS(const dmb *mp, int bp, const dmb *mq, int bq, int mb_edge)
{
    if (is_intra(mp) || is_intra(mq)) return mb_edge ? 4 : 3;
    if (mp->nzl[bp] || mq->nzl[bq]) return 2;
    const int qp_ = (bp >> 3) * 2 + ((bp >> 1) & 1), qq_ = (bq >> 3) * 2 + ((bq >> 1) & 1);
    if (mp->refpic[qp_] != mq->refpic[qq_]) return 1;           /* different reference pictures */
    if (abs(mp->mv[bp][0] - mq->mv[bq][0]) >= 4 || abs(mp->mv[bp][1] - mq->mv[bq][1]) >= 4) return 1;
    return 0;
}
static void deblock_picture(h264o_dec *d)
{
    const int cw = d->cw, cs = cw / 2, mbw = d->mbw, mbh = d->mbh;
    uint8_t *Y = d->cur.pl[0];
    for (int my = 0; my < mbh; my++)
        for (int mx = 0; mx < mbw; mx++) {
            const dmb *mq = &d->mb[my * mbw + mx];
            const int idc = d->slice_idc[mq->slice & 255], oa = d->slice_oa[mq->slice & 255], ob = d->slice_ob[mq->slice & 255];
            if (idc == 1) continue;
            /* 8.7: filterLeftMbEdgeFlag / filterTopMbEdgeFlag */
            int fleft = mx > 0, ftop = my > 0;
            if (idc == 2) {
                if (fleft && d->mb[my * mbw + mx - 1].slice != mq->slice) fleft = 0;
                if (ftop && d->mb[(my - 1) * mbw + mx].slice != mq->slice) ftop = 0;
            }
            for (int dir = 0; dir < 2; dir++)        /* vertical edges (left to right) first, then horizontal (top to bottom) */
                for (int e = 0; e < 4; e++) {
                    if (e == 0 && !(dir == 0 ? fleft : ftop)) continue;
                    if (mq->t8x8 && (e & 1)) continue;   /* transform_size_8x8_flag: no 4x4-internal luma edges */
                    const dmb *mp = e ? mq : (dir == 0 ? mq - 1 : mq - mbw);
                    for (int k = 0; k < 16; k++) {       /* sample k along the edge */
                        const int blk = k >> 2;
                        int bq, bp;
                        if (dir == 0) { bq = 4 * blk + e; bp = e ? bq - 1 : 4 * blk + 3; }
                        else { bq = 4 * e + blk; bp = e ? bq - 4 : 12 + blk; }
                        const int bS = derive_bS(mp, bp, mq, bq, e == 0);
                        if (!bS) continue;
                        {   /* luma */
                            const int qpav = (mp->qp + mq->qp + 1) >> 1;
                            const int indexA = clip3i(0, 51, qpav + oa), indexB = clip3i(0, 51, qpav + ob);
                            uint8_t *pix = dir == 0 ? Y + (size_t)(16 * my + k) * cw + 16 * mx + 4 * e : Y + (size_t)(16 * my + 4 * e) * cw + 16 * mx + k;
                            filter_samples(pix, dir == 0 ? 1 : cw, bS, 0, 0, indexA, D_ALPHA[indexA], D_BETA[indexB]);
                        }
                        /* chroma (4:2:0): edges 0 and 2 of the luma grid are chroma edges 0 and 4; chroma sample k/2 takes the
                         * bS of the luma sample position 2 * (k/2) (8.7.2: "bS of the corresponding luma edge") */
                        if (!(e & 1) && !(k & 1)) {
                            const int kc = k >> 1, ec = 2 * e;   /* chroma sample offset of the edge: 0 or 4 */
                            for (int pl = 0; pl < 2; pl++) {
                                const int qpp = mp->kind == DMB_IPCM ? qpc_of(d, 0, pl) : qpc_of(d, mp->qp, pl);
                                const int qpq = mq->kind == DMB_IPCM ? qpc_of(d, 0, pl) : qpc_of(d, mq->qp, pl);
                                const int qpav = (qpp + qpq + 1) >> 1;
                                const int indexA = clip3i(0, 51, qpav + oa), indexB = clip3i(0, 51, qpav + ob);
                                uint8_t *C = d->cur.pl[1 + pl];
                                uint8_t *pix = dir == 0 ? C + (size_t)(8 * my + kc) * cs + 8 * mx + ec : C + (size_t)(8 * my + ec) * cs + 8 * mx + kc;
                                filter_samples(pix, dir == 0 ? 1 : cs, bS, 1, 1, indexA, D_ALPHA[indexA], D_BETA[indexB]);
                            }
                        }
                    }
                }
        }
}

/* ------------------------------------------------------------------ 7.3.3 slice header + 7.3.4 slice data */
static int finish_picture(h264o_dec *d)
{
    deblock_picture(d);
    /* the finished picture becomes the output and, when it is a reference picture, enters the list (8.2.5.3 sliding window) */
    {
        dpic t = d->out; d->out = d->cur; d->cur = t;   /* out now holds the finished picture; cur gets a scratch buffer */
    }
    if (d->cur_is_ref) {
        if (d->cur_is_idr) { for (int i = 0; i < d->nrefs; i++) { /* all reference pictures become unused */ } d->nrefs = 0; }
        const int cap = d->max_refs > 0 ? d->max_refs : 1;
        /* make room: drop the oldest */
        if (d->nrefs >= cap) d->nrefs = cap - 1;
        /* shift and insert a COPY of the output picture at the front */
        dpic last = d->refs[D_MAXREF - 1];
        for (int i = D_MAXREF - 1; i > 0; i--) d->refs[i] = d->refs[i - 1];
        d->refs[0] = last;
        if (!d->refs[0].pl[0] && alloc_pic(d, &d->refs[0])) return fail(d, "out of memory");
        const size_t ysz = (size_t)d->cw * d->ch;
        memcpy(d->refs[0].pl[0], d->out.pl[0], ysz);
        memcpy(d->refs[0].pl[1], d->out.pl[1], ysz / 4);
        memcpy(d->refs[0].pl[2], d->out.pl[2], ysz / 4);
        d->refs[0].frame_num = d->cur_frame_num;
        d->refs[0].id = ++d->next_id;
        d->nrefs++;
    }
    d->pic_open = 0;
    return 1;
}

/* RefPicList0 of a P slice of a frame: 8.2.4.1 picture numbers (FrameNumWrap, 8-27; PicNum = FrameNumWrap, 8-28), 8.2.4.2.1
 * initial order (descending PicNum), 8.2.4.3 / 8.2.4.3.1 modification by the slice header's commands (short-term only) */
static int build_list0(h264o_dec *d, bitr *b, int frame_num, int nactive, int modify)
{
    const int max_fn = 1 << d->log2_max_frame_num;
    int picnum[D_MAXREF];
    for (int i = 0; i < d->nrefs; i++) picnum[i] = d->refs[i].frame_num > frame_num ? d->refs[i].frame_num - max_fn : d->refs[i].frame_num;
    int n = 0;
    for (int i = 0; i < d->nrefs; i++) {   /* insertion sort, descending PicNum */
        int k = n++;
        while (k > 0 && picnum[d->list0[k - 1]] < picnum[i]) { d->list0[k] = d->list0[k - 1]; k--; }
        d->list0[k] = i;
    }
    for (int i = (n < nactive ? n : nactive); i <= D_MAXREF; i++) d->list0[i] = -1;   /* longer than num_ref_idx_l0_active: truncated (8.2.4.2) */
    if (!modify) return 0;
    int pred = frame_num, idx = 0;   /* picNumL0Pred = CurrPicNum */
    for (int guard = 0; guard < 64; guard++) {
        const uint32_t idc = rd_ue(b);
        if (b->err) return fail(d, "ref_pic_list_modification truncated");
        if (idc == 3) return 0;
        if (idc == 2) return fail(d, "long-term reference pictures unsupported");
        if (idc > 3) return fail(d, "modification_of_pic_nums_idc out of range");
        const int diff = (int)rd_ue(b) + 1;
        if (idx >= nactive || nactive > D_MAXREF) return fail(d, "more list modifications than list entries");
        int nowrap;
        if (idc == 0) { nowrap = pred - diff; if (nowrap < 0) nowrap += max_fn; }              /* (8-34) */
        else { nowrap = pred + diff; if (nowrap >= max_fn) nowrap -= max_fn; }                 /* (8-35) */
        pred = nowrap;
        const int pn = nowrap > frame_num ? nowrap - max_fn : nowrap;                          /* (8-36) */
        int k = -1;
        for (int i = 0; i < d->nrefs; i++) if (picnum[i] == pn) k = i;
        if (k < 0) return fail(d, "list modification names a picture that is not a reference");
        /* (8-37): the entries from idx on move up by one, the picture is put at idx, its other occurrence is dropped */
        for (int c = nactive; c > idx; c--) d->list0[c] = d->list0[c - 1];
        d->list0[idx++] = k;
        int w = idx;
        for (int c = idx; c <= nactive; c++) if (d->list0[c] != k) d->list0[w++] = d->list0[c];
        for (; w <= nactive; w++) d->list0[w] = -1;
    }
    return fail(d, "too many list modifications");
}

static int decode_slice(h264o_dec *d, bitr *b, int nal_type, int nal_ref_idc)
{
    if (!d->have_sps || !d->have_pps) return fail(d, "slice before parameter sets");
    const int first_mb = (int)rd_ue(b);
    const int st = (int)rd_ue(b) % 5;
    if (st != 0 && st != 2) return fail(d, "only I and P slices supported");
    if (rd_ue(b)) return fail(d, "pic_parameter_set_id != 0");
    const int frame_num = (int)rd_u(b, d->log2_max_frame_num);
    if (nal_type == 5) rd_ue(b); /* idr_pic_id */
    if (d->poc_type == 0) rd_u(b, d->log2_max_poc_lsb);
    int nactive = d->num_ref_default;
    if (st == 0) {
        if (rd_bit(b)) nactive = (int)rd_ue(b) + 1;       /* num_ref_idx_active_override_flag */
        if (nactive > D_MAXREF) return fail(d, "num_ref_idx_l0_active > 16");
        if (nal_type == 5) return fail(d, "P slice in an IDR picture");
        if (build_list0(d, b, frame_num, nactive, rd_bit(b))) return -1;   /* ref_pic_list_modification_flag_l0 */
    }
    if (nal_ref_idc) {
        if (nal_type == 5) { rd_bit(b); if (rd_bit(b)) return fail(d, "long_term_reference_flag unsupported"); }
        else if (rd_bit(b)) return fail(d, "adaptive_ref_pic_marking_mode_flag unsupported");
    }
    int qp = d->init_qp + rd_se(b);
    int idc = 0, oa = 0, ob = 0;
    if (d->dbf_ctrl) {
        idc = (int)rd_ue(b);
        if (idc > 2) return fail(d, "disable_deblocking_filter_idc out of range");
        if (idc != 1) { oa = 2 * rd_se(b); ob = 2 * rd_se(b); }
    }
    if (b->err) return fail(d, "slice header truncated");
    if (qp < 0 || qp > 51) return fail(d, "SliceQPY out of range");
    const int nmb = d->mbw * d->mbh;
    if (first_mb == 0) {
        if (d->pic_open) return fail(d, "new picture before the previous one was complete");
        d->slice_count = 0;
        for (int i = 0; i < nmb; i++) d->mb[i].kind = DMB_NONE;
        d->max_mb_bits = 0; d->max_level_prefix = 0;
        d->cur_is_ref = nal_ref_idc != 0; d->cur_frame_num = frame_num; d->cur_is_idr = nal_type == 5;
        if (nal_type == 5) d->nrefs = 0;   /* 8.2.1: an IDR picture marks all reference pictures unused before it is decoded */
    } else if (first_mb != d->pic_open) return fail(d, "first_mb_in_slice does not continue the picture (ASO unsupported)");
    if (d->slice_count >= 256) return fail(d, "more than 256 slices");
    const int sl = d->slice_count++;
    d->slice_idc[sl] = idc; d->slice_oa[sl] = oa; d->slice_ob[sl] = ob;
    d->slice_type = st;
    d->nal_type = nal_type;
    if (st == 0 && d->nrefs < 1) return fail(d, "P slice without a reference picture");
    int addr = first_mb, more = 1, prev_skipped = 0;
    (void)prev_skipped;
    while (more && addr < nmb) {
        if (st == 0) {
            uint32_t run = rd_ue(b);
            if (b->err) return fail(d, "mb_skip_run parse error");
            for (; run > 0 && addr < nmb; run--, addr++) {
                dmb *m = &d->mb[addr];
                memset(m, 0, sizeof(*m));
                m->slice = (int16_t)sl;
                mbctx c = {d, b, m, addr % d->mbw, addr / d->mbw, sl, addr};
                if (decode_inter_mb(&c, -1, nactive, &qp)) return -1;
            }
            if (run > 0) return fail(d, "mb_skip_run past the end of the picture");
            more = more_rbsp_data(b);
            if (!more) break;
            if (addr >= nmb) return fail(d, "slice data continues past the last macroblock");
        }
        const size_t mb_start = b->pos;
        const uint32_t mb_type = rd_ue(b);
        dmb *m = &d->mb[addr];
        memset(m, 0, sizeof(*m));
        m->slice = (int16_t)sl;
        mbctx c = {d, b, m, addr % d->mbw, addr / d->mbw, sl, addr};
        int rc;
        if (st == 0 && mb_type < 5) rc = decode_inter_mb(&c, (int)mb_type, nactive, &qp);
        else {
            const int it = st == 0 ? (int)mb_type - 5 : (int)mb_type;
            if (it < 0 || it > 25) return fail(d, "mb_type out of range");
            rc = decode_intra_mb(&c, it, &qp);
        }
        if (rc) return rc;
        if (b->err) return fail(d, "slice data overrun");
        m->bits = (uint16_t)(b->pos - mb_start > 65535 ? 65535 : b->pos - mb_start);
        if ((int)m->bits > d->max_mb_bits) d->max_mb_bits = (int)m->bits;
        addr++;
        more = more_rbsp_data(b);
    }
    if (b->err) return fail(d, "slice data overrun");
    d->pic_open = addr;
    if (addr < nmb) return 0; /* more slices follow */
    return finish_picture(d);
}

/* ------------------------------------------------------------------ Annex B + 7.4.1.1 */
int h264o_dec_decode(h264o_dec *d, const uint8_t *data, size_t len)
{
    int got_pic = 0;
    size_t i = 0;
    d->err[0] = 0;
    uint8_t *rbsp = (uint8_t *)malloc(len + 8);
    if (!rbsp) return fail(d, "out of memory");
    while (i + 3 <= len) {
        if (!(data[i] == 0 && data[i + 1] == 0 && data[i + 2] == 1)) { i++; continue; }
        size_t s = i + 3, e = s;
        /* the NAL unit ends before the next 00 00 00 / 00 00 01 or at the end of the buffer */
        while (e + 3 <= len && !(data[e] == 0 && data[e + 1] == 0 && data[e + 2] <= 1)) e++;
        if (e + 3 > len) e = len;
        if (e <= s) { i = e; continue; }
        const int hdr = data[s], type = hdr & 31, ref_idc = (hdr >> 5) & 3;
        if (hdr & 0x80) { free(rbsp); return fail(d, "forbidden_zero_bit set"); }
        size_t n = 0;
        int zeros = 0;
        for (size_t k = s + 1; k < e; k++) {
            if (zeros >= 2 && data[k] == 3) {   /* emulation_prevention_three_byte */
                if (k + 1 < e && data[k + 1] > 3) { free(rbsp); return fail(d, "emulation prevention byte followed by a byte > 3"); }
                zeros = 0;
                continue;
            }
            if (zeros >= 2 && data[k] < 3) { free(rbsp); return fail(d, "start code emulation inside a NAL unit"); }
            rbsp[n++] = data[k];
            zeros = data[k] == 0 ? zeros + 1 : 0;
        }
        bitr b = {rbsp, n * 8, 0, 0};
        int rc = 0;
        if (type == 7) rc = parse_sps(d, &b);
        else if (type == 8) rc = parse_pps(d, &b);
        else if (type == 1 || type == 5) { rc = decode_slice(d, &b, type, ref_idc); if (rc == 1) got_pic = 1; }
        if (rc < 0) { free(rbsp); return rc; }
        i = e;
    }
    free(rbsp);
    return got_pic;
}

/* ------------------------------------------------------------------ table export for tests/test_oracle_kat.py:
 * the decoder's VLC tables in (length, bits) form, so that they can be compared with the encoder's h264_tables.h */
int h264o_dec_table_coeff_token(int col, int total_coeff, int trailing_ones, int *len, unsigned *bits)
{
    for (int i = 0; i < D_NTOKEN; i++)
        if (D_COEFF_TOKEN[i].tc == total_coeff && D_COEFF_TOKEN[i].t1 == trailing_ones && col >= 0 && col < 5) {
            const char *s = D_COEFF_TOKEN[i].c[col];
            if (s[0] == '-') return -1;
            unsigned v = 0; int n = 0;
            for (; s[n]; n++) v = (v << 1) | (unsigned)(s[n] - '0');
            *len = n; *bits = v;
            return 0;
        }
    return -1;
}
static int str_code(const char *s, int *len, unsigned *bits)
{
    if (!s) return -1;
    unsigned v = 0; int n = 0;
    for (; s[n]; n++) v = (v << 1) | (unsigned)(s[n] - '0');
    *len = n; *bits = v;
    return 0;
}
int h264o_dec_table_total_zeros(int chroma_dc, int tz_vlc_index, int total_zeros, int *len, unsigned *bits)
{
    if (chroma_dc) return (tz_vlc_index < 1 || tz_vlc_index > 3 || total_zeros < 0 || total_zeros > 3) ? -1 : str_code(D_TOTAL_ZEROS_CDC[tz_vlc_index - 1][total_zeros], len, bits);
    return (tz_vlc_index < 1 || tz_vlc_index > 15 || total_zeros < 0 || total_zeros > 15) ? -1 : str_code(D_TOTAL_ZEROS[tz_vlc_index - 1][total_zeros], len, bits);
}
int h264o_dec_table_run_before(int zeros_left, int run, int *len, unsigned *bits)
{
    if (zeros_left < 1 || run < 0 || run > 14) return -1;
    return str_code(D_RUN_BEFORE[(zeros_left > 6 ? 7 : zeros_left) - 1][run], len, bits);
}
int h264o_dec_table_misc(int which, int i, int j)
{
    switch (which) {
        case 0: return i >= 0 && i < 48 && j >= 0 && j < 2 ? D_CBP[i][j] : -1;
        case 1: return i >= 0 && i < 16 ? D_ZZ4[i][0] + 4 * D_ZZ4[i][1] : -1;
        case 2: return i >= 0 && i < 52 ? D_ALPHA[i] : -1;
        case 3: return i >= 0 && i < 52 ? D_BETA[i] : -1;
        case 4: return i >= 0 && i < 52 && j >= 0 && j < 3 ? D_TC0[j][i] : -1;
        case 5: return i >= 0 && i < 52 ? (i < 30 ? i : D_QPC_HIGH[i - 30]) : -1;
        case 6: return i >= 0 && i < 6 && j >= 0 && j < 3 ? D_V4[i][j] : -1;
        case 7: return i >= 0 && i < 64 ? D_ZZ8[i][0] + 8 * D_ZZ8[i][1] : -1;
        case 8: return i >= 0 && i < 6 && j >= 0 && j < 6 ? D_V8[i][j] : -1;
        default: return -1;
    }
}
