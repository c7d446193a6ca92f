// media_amd/csrc/h264_parse.h -- host side of the decoder peer (row f4 of SURVEY.md 8: a VideoDecoder next to
// /root/reference/video_decoder/VideoDecoderNetint.cpp, interface video_decoder/include/VideoDecoder.h:83).
//
// CAVLC slice data is a serial code: bit i cannot be placed before bit i - 1 has been understood.  The decoder therefore
// splits like every software-assisted GPU decoder does: THIS file walks the NAL units of an access unit, parses parameter
// sets, slice headers and slice data on the host and fills, per macroblock, exactly the side arrays the encoder's kernels
// exchange (MbInfo, quadrant vectors, Intra4x4 modes, level lists); the reconstruction - motion compensation, inverse
// transforms, intra prediction in row-wavefront order, the loop filter - then runs on the GPU (k_dec.h and the encoder's own
// k_deblock_rows).  Nothing here touches samples except I_PCM's raw bytes.
//
// Supported streams = a superset of what this repository's encoder produces (which is what the reference preset asks of
// OpenH264 minus CABAC): baseline / main / high with CAVLC, frame macroblocks, I and P slices; Intra16x16, Intra4x4, I_PCM
// (also in loop-filtered pictures); P_L0_16x16, 16x8, 8x16, P_8x8 / P_8x8ref0 with every sub_mb_type (8x8, 8x4, 4x8, 4x4),
// P_Skip; up to 3 reference pictures by sliding window, ref_pic_list_modification by short-term picture numbers, a reference
// index per partition; 4x4 transform, 8x8 transform on inter macroblocks; QP per macroblock (slice_qp_delta per slice, mb_qp_delta), chroma_qp_index_offset and
// second_chroma_qp_index_offset, slice_alpha_c0_offset_div2 / slice_beta_offset_div2 (one pair per picture);
// slices of any shape in raster order (bands of whole rows decode as independent wavefronts); disable_deblocking_filter_idc
// 0 / 1 / 2 (one value per picture).
// Anything else is refused with a message naming the syntax element (never decoded wrongly).
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <new>
#include <string>
#include <vector>
#include "h264_vlc_tables.h"

namespace h264dec {

enum { T_I16 = 0, T_P16 = 1, T_PSKIP = 2, T_IPCM = 3, T_I4 = 4, T_P16X8 = 5, T_P8X16 = 6, T_P8X8 = 7 };   // = dev_common.h MB_*
enum { L_LUMA_DC = 0, L_LUMA = 16, L_CHROMA_DC = 272, L_CHROMA_AC = 280, L_STRIDE = 416 };                // = dev_common.h LV_*

struct MbRec {   // = dev_common.h MbInfo (32 bytes)
    int16_t mvx, mvy;
    uint8_t type, i16_mode, chroma_mode, cbp;
    uint8_t tc[24];
};
static_assert(sizeof(MbRec) == 32, "MbRec layout");

struct Sps {
    bool valid = false;
    int profile_idc = 0, level_idc = 0, log2_max_frame_num = 4, poc_type = 0, log2_max_poc_lsb = 4, max_refs = 1;
    int mbw = 0, mbh = 0, crop_l = 0, crop_r = 0, crop_t = 0, crop_b = 0;
    bool delta_pic_order_always_zero = false;
};
struct Pps {
    bool valid = false;
    int sps_id = 0, num_ref_default = 1, pic_init_qp = 26;
    int cqo[2] = {0, 0};   // chroma_qp_index_offset, second_chroma_qp_index_offset (Cb, Cr)
    bool deblock_control = false, t8x8 = false, bottom_field_pic_order = false, redundant_pic_cnt = false;
    bool constrained_intra = false;   // constrained_intra_pred_flag: inter macroblocks are not available for intra prediction
};

class BitReader {
public:
    BitReader(const uint8_t* p, size_t n) : p_(p), n_(n)
    {
        // rbsp_trailing_bits: the last 1 bit of the payload is the stop bit; more_rbsp_data() = something before it
        size_t k = n;
        while (k > 0 && p[k - 1] == 0) k--;
        stop_ = 0;
        if (k > 0) {
            int b = 0;
            while (!((p[k - 1] >> b) & 1)) b++;
            stop_ = (k - 1) * 8 + (size_t)(7 - b);
        }
    }
    bool bad() const { return bad_; }
    size_t pos() const { return pos_; }
    bool more_data() const { return pos_ < stop_; }
    bool aligned() const { return (pos_ & 7) == 0; }
    // The payload is followed by PAD zero bytes (the caller's buffer): eight bytes can be loaded at every position up to the
    // end, and bits past the end read as 0.
    enum { PAD = 8 };
    uint32_t peek(int k)   // 1 <= k <= 32
    {
        const size_t byte = pos_ >> 3 < n_ ? pos_ >> 3 : n_;
        uint64_t v;
        memcpy(&v, p_ + byte, 8);
        v = __builtin_bswap64(v) << (pos_ & 7);
        return (uint32_t)(v >> (64 - k));
    }
    void skip(int k) { pos_ += (size_t)k; if (pos_ > n_ * 8) bad_ = true; }
    uint32_t u(int k)
    {
        uint32_t v = 0;
        while (k > 24) { v = (v << 24) | peek(24); skip(24); k -= 24; }
        if (k > 0) { v = (v << k) | peek(k); skip(k); }
        return v;
    }
    uint32_t ue()
    {
        const uint32_t w = peek(32);
        if (w >> 16) {   // up to 15 leading zeros: the whole code word (2 z + 1 bits) is inside w
            const int z = __builtin_clz(w);
            skip(2 * z + 1);
            return (w >> (31 - 2 * z)) - 1u;
        }
        int z = 0;
        while (z < 32 && peek(1) == 0 && !bad_) { skip(1); z++; }
        if (z >= 32) { bad_ = true; return 0; }
        skip(1);
        return z ? ((1u << z) - 1u + u(z)) : 0u;
    }
    // number of zero bits before the next 1 bit (level_prefix, 9.2.2.1), consumed with that bit; -1: none within 32 bits
    int unary()
    {
        const uint32_t w = peek(32);
        if (!w) return -1;
        const int z = __builtin_clz(w);
        skip(z + 1);
        return z;
    }
    int32_t se() { const uint32_t k = ue(); return (k & 1) ? (int32_t)((k + 1) >> 1) : -(int32_t)(k >> 1); }
    const uint8_t* byte_ptr() const { return p_ + (pos_ >> 3); }
private:
    const uint8_t* p_;
    size_t n_, pos_ = 0, stop_ = 0;
    bool bad_ = false;
};

// ---- VLC look-ups built once from the (length, bits) tables ----
struct VlcLut {
    // entry: len << 8 | symbol; 0 = no code
    std::vector<uint16_t> ct[4], cdc, tz[15], ctz[3], run[7];
    uint16_t ct8[4][256];   // coeff_token codes of up to 8 bits (nearly all that occur) by the next 8 bits: 2 kB instead of the 512 kB of ct[]
    uint8_t code2cbp_inter[48], code2cbp_intra[48];
    VlcLut()
    {
        static const uint8_t ct_len[4][68] = H264_TAB_CT_LEN, ct_bits[4][68] = H264_TAB_CT_BITS;
        static const uint8_t cdc_len[20] = H264_TAB_CDC_LEN, cdc_bits[20] = H264_TAB_CDC_BITS;
        static const uint8_t tz_len[15][16] = H264_TAB_TZ_LEN, tz_bits[15][16] = H264_TAB_TZ_BITS;
        static const uint8_t ctz_len[3][4] = H264_TAB_CTZ_LEN, ctz_bits[3][4] = H264_TAB_CTZ_BITS;
        static const uint8_t run_len[7][16] = H264_TAB_RUN_LEN, run_bits[7][16] = H264_TAB_RUN_BITS;
        static const uint8_t c2i[48] = H264_TAB_CBP2CODE_INTER, c2a[48] = H264_TAB_CBP2CODE_INTRA;
        for (int c = 0; c < 4; c++) {
            ct[c].assign(1u << 16, 0);
            for (int s = 0; s < 68; s++) fill(ct[c], 16, ct_len[c][s], ct_bits[c][s], s, (s & 3) <= (s >> 2));
            for (int i = 0; i < 256; i++) { const uint16_t e = ct[c][(size_t)i << 8]; ct8[c][i] = (e >> 8) <= 8 ? e : 0; }
        }
        cdc.assign(1u << 8, 0);
        for (int s = 0; s < 20; s++) fill(cdc, 8, cdc_len[s], cdc_bits[s], s, (s & 3) <= (s >> 2));
        for (int t = 0; t < 15; t++) {
            tz[t].assign(1u << 9, 0);
            for (int s = 0; s < 16 - t; s++) fill(tz[t], 9, tz_len[t][s], tz_bits[t][s], s, true);
        }
        for (int t = 0; t < 3; t++) {
            ctz[t].assign(1u << 3, 0);
            for (int s = 0; s < 4 - t; s++) fill(ctz[t], 3, ctz_len[t][s], ctz_bits[t][s], s, true);
        }
        for (int t = 0; t < 7; t++) {
            run[t].assign(1u << 11, 0);
            for (int s = 0; s < (t < 6 ? t + 2 : 15); s++) fill(run[t], 11, run_len[t][s], run_bits[t][s], s, true);
        }
        for (int v = 0; v < 48; v++) { code2cbp_inter[c2i[v]] = (uint8_t)v; code2cbp_intra[c2a[v]] = (uint8_t)v; }
    }
private:
    static void fill(std::vector<uint16_t>& lut, int width, int len, unsigned bits, int sym, bool legal)
    {
        if (!len || !legal) return;
        const unsigned base = bits << (width - len);
        for (unsigned i = 0; i < (1u << (width - len)); i++) lut[base + i] = (uint16_t)((len << 8) | sym);
    }
};
inline const VlcLut& vlc() { static const VlcLut L; return L; }

// The per-macroblock arrays are what the decoder uploads every picture: it switches the allocation to pinned host memory
// (HostMem::use), so that the copies run asynchronously while the next access unit is parsed.  Every block remembers the
// function that releases it, so blocks made before and after a switch can be freed in any order.
struct HostMem {
    typedef void* (*AllocFn)(size_t);
    typedef void (*FreeFn)(void*);
    static inline AllocFn alloc_fn = nullptr;   // nullptr: malloc / free
    static inline FreeFn free_fn = nullptr;
    static void use(AllocFn a, FreeFn f) { alloc_fn = a; free_fn = f; }
    static void* get(size_t n)
    {
        const AllocFn a = alloc_fn;
        const FreeFn f = a ? free_fn : nullptr;
        char* raw = (char*)(a ? a(n + 64) : malloc(n + 64));
        if (!raw) throw std::bad_alloc();
        *(FreeFn*)raw = f;
        return raw + 64;
    }
    static void put(void* p)
    {
        char* raw = (char*)p - 64;
        const FreeFn f = *(FreeFn*)raw;
        if (f) f(raw); else free(raw);
    }
};
template <class T>
struct HostAlloc {
    typedef T value_type;
    HostAlloc() = default;
    template <class U> HostAlloc(const HostAlloc<U>&) {}
    T* allocate(size_t n) { return (T*)HostMem::get(n * sizeof(T)); }
    void deallocate(T* p, size_t) { HostMem::put(p); }
    template <class U> bool operator==(const HostAlloc<U>&) const { return true; }
    template <class U> bool operator!=(const HostAlloc<U>&) const { return false; }
};
template <class T> using HostVec = std::vector<T, HostAlloc<T>>;

struct Picture {
    int mbw = 0, mbh = 0, width = 0, height = 0;   // macroblocks; cropped size
    bool idr = false, is_ref = true;
    int qp = 26, slice_rows = 0, deblock_idc = 0, num_ref_active = 0, t8x8_mode = 0, profile_idc = 66;
    bool has_pcm = false, has_intra = false, has_inter = false;
    int cqo[2] = {0, 0};             // chroma QP index offsets (Cb, Cr) of the picture parameter set
    int filter_oa = 0, filter_ob = 0;   // FilterOffsetA / FilterOffsetB (2 * slice_alpha_c0_offset_div2, 2 * slice_beta_offset_div2)
    bool one_qp = true;              // every macroblock has QP_Y = qp, no chroma / filter offset, no I_PCM macroblock: the picture
                                     // reconstructs with the per-picture constants the encoder's kernels use
    HostVec<uint8_t> mbqp;       // QP_Y of every macroblock (7.4.5); 0 for I_PCM, the value its edges filter with (8.7.2.2)
    HostVec<MbRec> mb;
    HostVec<int16_t> mvq;      // 8 per macroblock: the vectors of the four 8x8 quadrants' first blocks (the encoder's layout)
    HostVec<int16_t> mv4;      // 32 per macroblock: the vector (x, y) of every 4x4 block, raster order (sub-macroblock partitions)
    HostVec<uint8_t> mbavail;  // per macroblock: neighbours available for prediction (6.4.9: in the picture, in the same slice, decoded
                               // before): bit 0 left, 1 above, 2 above-right, 3 above-left - slices may start and end at any macroblock
    int ref_age[3] = {0, 1, 2};   // RefPicList0: entry r is the reference picture decoded ref_age[r] + 1 reference pictures ago (the default
                               // order is 0, 1, 2; ref_pic_list_modification permutes it)
    int slices = 0;            // slices of the picture; slice_rows > 0: they are bands of that many whole rows, -1: of any other shape
    HostVec<uint8_t> refq;     // 4 per macroblock: ref_idx_l0 of the four quadrants (0xFF: intra)
    HostVec<uint8_t> aux;      // 16 per macroblock
    // The levels travel to the GPU as ONE BYTE each (L_STRIDE per macroblock, the layout of dev_common.h LV_*; an I_PCM macroblock's
    // 384 samples as bytes at the start of its area): half the upload of an int16 array.  The few that do not fit a signed byte
    // are 0 there and listed in `big` (index into levels8, value); the decoder widens the array on the GPU and patches those in.
    HostVec<int8_t> levels8;
    struct Big { uint32_t idx; int32_t val; };
    HostVec<Big> big;
};

class Parser {
public:
    const std::string& error() const { return err_; }
    void set_error(const std::string& e) { err_ = e; }
    const Picture& picture() const { return *picp_; }
    // two picture buffers: the decoder parses access unit n + 1 into one while the arrays of picture n are still being uploaded
    // from the other (select() before parse_access_unit)
    void select(int k) { picp_ = &pics_[k & 1]; }
    const Sps& sps() const { return sps_[active_sps_]; }

    // one access unit (Annex B).  Returns 1: a picture is ready in picture(); 0: no slice in it (parameter sets only); -1: error().
    // The picture enters the parser's list of reference pictures (dpb_fn_, the mirror of the decoder's reconstruction ring) through
    // commit(): at once with auto_commit (the parser used alone), else when the decoder has accepted and launched it - a picture
    // the decoder still refuses must not shift the list.  An access unit that is refused - here or by the decoder - may have been
    // a reference picture: lose_refs() empties the list, and P pictures are refused until the next IDR picture.
    int parse_access_unit(const uint8_t* au, size_t len, bool auto_commit = true)
    {
        const int rc = parse_access_unit_(au, len);
        if (rc < 0) lose_refs();
        else if (rc > 0 && auto_commit) commit();
        return rc;
    }
    void commit()
    {
        if (picp_->idr) dpb_fn_.clear();
        if (picp_->is_ref) {
            dpb_fn_.insert(dpb_fn_.begin(), cur_frame_num_);
            const size_t cap = (size_t)(sps().max_refs > 0 ? sps().max_refs : 1);
            if (dpb_fn_.size() > cap) dpb_fn_.resize(cap);
            prev_ref_fn_ = cur_frame_num_;
        }
    }
    void lose_refs() { dpb_fn_.clear(); prev_ref_fn_ = -1; }

private:
    int parse_access_unit_(const uint8_t* au, size_t len)
    {
        err_.clear();
        bool have_pic = false;
        int next_mb = 0;
        size_t i = 0;
        while (i + 3 < len) {
            // next start code
            if (!(au[i] == 0 && au[i + 1] == 0 && (au[i + 2] == 1 || (au[i + 2] == 0 && i + 3 < len && au[i + 3] == 1)))) { i++; continue; }
            const size_t s = i + (au[i + 2] == 1 ? 3 : 4);
            size_t e = s;
            while (e + 2 < len && !(au[e] == 0 && au[e + 1] == 0 && (au[e + 2] == 1 || (au[e + 2] == 0 && e + 3 < len && au[e + 3] == 1)))) e++;
            if (e + 2 >= len) e = len;
            if (s < e && !nal(au + s, e - s, have_pic, next_mb)) return -1;
            i = e;
        }
        if (!have_pic) return 0;
        if (next_mb != picp_->mbw * picp_->mbh) return fail("picture incomplete: slices cover %d of %d macroblocks", next_mb, picp_->mbw * picp_->mbh);
        // bands of slice_rows rows only if every band start was seen (a last slice longer than the others is "any shape")
        if (picp_->slice_rows > 0 && picp_->slices != (picp_->mbh + picp_->slice_rows - 1) / picp_->slice_rows) picp_->slice_rows = -1;
        return 1;
    }

    Sps sps_[32];
    Pps pps_[256];
    int active_sps_ = 0;
    Picture pics_[2];
    Picture* picp_ = &pics_[0];
    std::string err_;
    std::vector<uint8_t> rbsp_;
    // state of the slice being parsed
    int slice_first_ = 0, slice_type_ = 0, slice_qp_ = 26, num_ref_ = 1;
    int qp_ = 26;   // QP_Y of the previous macroblock of the slice in decoding order
    bool constrained_ = false;   // constrained_intra_pred_flag of the slice's picture parameter set
    // frame_num of the short-term reference pictures, the one decoded last first (sliding window, 8.2.5.3)
    std::vector<int> dpb_fn_;
    int cur_frame_num_ = 0;
    int prev_ref_fn_ = -1;   // frame_num of the previous reference picture (PrevRefFrameNum, 7.4.3); -1: none held

    int fail(const char* fmt, int a = 0, int b = 0)
    {
        char buf[256];
        snprintf(buf, sizeof(buf), fmt, a, b);
        err_ = buf;
        return -1;
    }
    bool nal(const uint8_t* p, size_t n, bool& have_pic, int& next_mb)
    {
        const int hdr = p[0], ref_idc = (hdr >> 5) & 3, type = hdr & 31;
        if (hdr & 0x80) { fail("forbidden_zero_bit set"); return false; }
        // emulation prevention (7.4.1.1): 00 00 03 -> 00 00
        rbsp_.resize(n + BitReader::PAD);
        size_t m = 0;
        int zeros = 0;
        for (size_t i = 1; i < n; i++) {
            if (zeros >= 2 && p[i] == 3) { zeros = 0; continue; }
            rbsp_[m++] = p[i];
            zeros = p[i] == 0 ? zeros + 1 : 0;
        }
        memset(rbsp_.data() + m, 0, BitReader::PAD);
        BitReader br(rbsp_.data(), m);
        if (type == 7) return parse_sps(br);
        if (type == 8) return parse_pps(br);
        if (type == 1 || type == 5) return parse_slice(br, type == 5, ref_idc, have_pic, next_mb);
        if (type == 2 || type == 3 || type == 4) { fail("data partitioning (nal_unit_type %d) is not supported", type); return false; }
        return true;   // SEI, AUD, end of sequence, filler ...: nothing to decode
    }

    bool parse_sps(BitReader& br)
    {
        Sps s;
        s.profile_idc = (int)br.u(8);
        br.u(8);   // constraint flags + reserved
        s.level_idc = (int)br.u(8);
        const unsigned id = br.ue();
        if (id > 31) { fail("seq_parameter_set_id %d", (int)id); return false; }
        if (s.profile_idc == 100 || s.profile_idc == 110 || s.profile_idc == 122 || s.profile_idc == 244 || s.profile_idc == 44 ||
            s.profile_idc == 83 || s.profile_idc == 86 || s.profile_idc == 118 || s.profile_idc == 128) {
            const unsigned cf = br.ue();
            if (cf != 1) { fail("chroma_format_idc %d (only 4:2:0)", (int)cf); return false; }
            if (br.ue() != 0 || br.ue() != 0) { fail("bit depth above 8"); return false; }
            br.u(1);   // qpprime_y_zero_transform_bypass_flag
            if (br.u(1)) { fail("seq_scaling_matrix_present_flag"); return false; }
        } else if (s.profile_idc != 66 && s.profile_idc != 77 && s.profile_idc != 88) { fail("profile_idc %d", s.profile_idc); return false; }
        const unsigned l2f = br.ue();
        if (l2f > 12) { fail("log2_max_frame_num_minus4 %d", (int)l2f); return false; }
        s.log2_max_frame_num = (int)l2f + 4;
        const unsigned pt = br.ue();
        if (pt > 2) { fail("pic_order_cnt_type %d", (int)pt); return false; }
        s.poc_type = (int)pt;
        if (s.poc_type == 0) {
            const unsigned l2p = br.ue();
            if (l2p > 12) { fail("log2_max_pic_order_cnt_lsb_minus4 %d", (int)l2p); return false; }
            s.log2_max_poc_lsb = (int)l2p + 4;
        }
        else if (s.poc_type == 1) {
            s.delta_pic_order_always_zero = br.u(1) != 0;
            br.se(); br.se();
            const unsigned n = br.ue();
            if (n > 255) { fail("num_ref_frames_in_pic_order_cnt_cycle %d", (int)n); return false; }
            for (unsigned k = 0; k < n && !br.bad(); k++) br.se();
        }
        const unsigned mr = br.ue();
        if (mr > 16) { fail("max_num_ref_frames %d", (int)mr); return false; }
        // The decoder holds the three most recent reference pictures.  A stream that announces more is decoded as long as it only
        // USES those three (the default list order is the same); a slice with more active references, or a list modification that
        // names an older picture, is refused where it occurs.
        s.max_refs = mr > 3 ? 3 : (int)mr;
        br.u(1);   // gaps_in_frame_num_value_allowed_flag
        const unsigned wmb = br.ue(), hmb = br.ue();
        if (wmb > 255 || hmb > 255) { fail("picture of %d x %d macroblocks (up to 256 x 256)", (int)wmb + 1, (int)hmb + 1); return false; }
        s.mbw = (int)wmb + 1;
        s.mbh = (int)hmb + 1;
        if (!br.u(1)) { fail("frame_mbs_only_flag = 0 (interlaced coding)"); return false; }
        br.u(1);   // direct_8x8_inference_flag
        if (br.u(1)) {
            const unsigned cl = br.ue(), cr = br.ue(), ct = br.ue(), cb = br.ue();
            if (cl + cr >= 8u * (unsigned)s.mbw || ct + cb >= 8u * (unsigned)s.mbh || cl > 4096 || cr > 4096 || ct > 4096 || cb > 4096) { fail("frame cropping larger than the picture"); return false; }
            s.crop_l = (int)cl; s.crop_r = (int)cr; s.crop_t = (int)ct; s.crop_b = (int)cb;
        }
        // (vui_parameters are not needed to decode samples)
        if (br.bad()) { fail("sequence parameter set damaged"); return false; }
        s.valid = true;
        sps_[id] = s;
        return true;
    }
    bool parse_pps(BitReader& br)
    {
        Pps p;
        const unsigned id = br.ue();
        const unsigned sid = br.ue();
        if (id > 255 || sid > 31) { fail("pic_parameter_set_id %d", (int)id); return false; }
        p.sps_id = (int)sid;
        if (br.u(1)) { fail("entropy_coding_mode_flag = 1 (CABAC)"); return false; }
        p.bottom_field_pic_order = br.u(1) != 0;
        if (br.ue() != 0) { fail("num_slice_groups_minus1 > 0 (FMO)"); return false; }
        const unsigned nrd = br.ue();
        if (nrd > 31) { fail("num_ref_idx_l0_default_active_minus1 %d", (int)nrd); return false; }
        p.num_ref_default = (int)nrd + 1;
        br.ue();   // num_ref_idx_l1_default_active_minus1
        if (br.u(1) || br.u(2)) { fail("weighted prediction"); return false; }
        const int iq = br.se();
        if (iq < -26 || iq > 25) { fail("pic_init_qp_minus26 %d", iq); return false; }
        p.pic_init_qp = 26 + iq;
        br.se();   // pic_init_qs_minus26
        const int cq = br.se();
        if (cq < -12 || cq > 12) { fail("chroma_qp_index_offset %d", cq); return false; }
        p.cqo[0] = p.cqo[1] = cq;
        p.deblock_control = br.u(1) != 0;
        p.constrained_intra = br.u(1) != 0;
        p.redundant_pic_cnt = br.u(1) != 0;
        if (br.more_data()) {
            p.t8x8 = br.u(1) != 0;
            if (br.u(1)) { fail("pic_scaling_matrix_present_flag"); return false; }
            const int cq2 = br.se();
            if (cq2 < -12 || cq2 > 12) { fail("second_chroma_qp_index_offset %d", cq2); return false; }
            p.cqo[1] = cq2;
        }
        if (br.bad()) { fail("picture parameter set damaged"); return false; }
        p.valid = true;
        pps_[id] = p;
        return true;
    }

    // ---- neighbourhood helpers (6.4.9 .. 6.4.11): an address is available inside the picture and not before the slice's start
    bool avail(int mx, int my) const { return mx >= 0 && mx < picp_->mbw && my >= 0 && my * picp_->mbw + mx >= slice_first_; }
    MbRec& M(int mx, int my) { return picp_->mb[(size_t)my * picp_->mbw + mx]; }
    static bool intra(int t) { return t == T_I16 || t == T_IPCM || t == T_I4; }
    static int blk_idx(int x, int y) { return (x & 1) | ((y & 1) << 1) | ((x & 2) << 1) | ((y & 2) << 2); }   // 4x4 raster -> blkIdx

    int nc_luma(int mx, int my, int bx, int by)   // 9.2.1: nC of the 4x4 luma block at raster (bx, by)
    {
        int nA = -1, nB = -1;
        if (bx > 0) nA = M(mx, my).tc[blk_idx(bx - 1, by)];
        else if (avail(mx - 1, my)) nA = M(mx - 1, my).tc[blk_idx(3, by)];
        if (by > 0) nB = M(mx, my).tc[blk_idx(bx, by - 1)];
        else if (avail(mx, my - 1)) nB = M(mx, my - 1).tc[blk_idx(bx, 3)];
        if (nA >= 0 && nB >= 0) return (nA + nB + 1) >> 1;
        return nA >= 0 ? nA : (nB >= 0 ? nB : 0);
    }
    int nc_chroma(int mx, int my, int pl, int bx, int by)
    {
        const int base = 16 + 4 * pl;
        int nA = -1, nB = -1;
        if (bx > 0) nA = M(mx, my).tc[base + 2 * by];
        else if (avail(mx - 1, my)) nA = M(mx - 1, my).tc[base + 2 * by + 1];
        if (by > 0) nB = M(mx, my).tc[base + bx];
        else if (avail(mx, my - 1)) nB = M(mx, my - 1).tc[base + 2 + bx];
        if (nA >= 0 && nB >= 0) return (nA + nB + 1) >> 1;
        return nA >= 0 ? nA : (nB >= 0 ? nB : 0);
    }

    // 9.2: one residual block; out[0 .. maxc - 1] in scan order (out is zeroed by the caller); returns TotalCoeff or -1
    void put_level(int8_t* p, int v)   // (|v| <= 32767: residual_block refuses anything larger)
    {
        if (v >= -128 && v <= 127) *p = (int8_t)v;
        else { *p = 0; picp_->big.push_back(Picture::Big{(uint32_t)(p - picp_->levels8.data()), (int32_t)v}); }
    }
    int residual_block(BitReader& br, int8_t* out, int maxc, int nC)
    {
        const VlcLut& L = vlc();
        int tc, t1;
        if (nC < 0) {
            const uint16_t e = L.cdc[br.peek(8)];
            if (!e) return -1;
            br.skip(e >> 8); tc = (e & 255) >> 2; t1 = e & 3;
        } else {
            const int cls = nC < 2 ? 0 : (nC < 4 ? 1 : (nC < 8 ? 2 : 3));
            const uint32_t w = br.peek(16);
            uint16_t e = L.ct8[cls][w >> 8];
            if (!e) e = L.ct[cls][w];
            if (!e) return -1;
            br.skip(e >> 8); tc = (e & 255) >> 2; t1 = e & 3;
        }
        if (tc == 0) return 0;
        if (tc > maxc) return -1;
        int level[16];
        int suffix_len = (tc > 10 && t1 < 3) ? 1 : 0;
        for (int i = 0; i < tc; i++) {
            if (i < t1) { level[i] = br.u(1) ? -1 : 1; continue; }
            const int prefix = br.unary();
            // level_prefix is at most 15 in the Baseline / Main / Extended profiles and 16 for the 8-bit High profile (9.2.2.1 with
            // bit depth 8): a longer one would make a level outside the int16 lists the reconstruction kernels read - refused
            if (prefix < 0 || prefix > 16) return -1;
            int code = (prefix < 15 ? prefix : 15) << suffix_len;
            if (suffix_len > 0 || prefix >= 14) {
                const int size = (prefix == 14 && suffix_len == 0) ? 4 : (prefix >= 15 ? prefix - 3 : suffix_len);
                if (size > 0) code += (int)br.u(size);
            }
            if (prefix >= 15 && suffix_len == 0) code += 15;
            if (prefix >= 16) code += (1 << (prefix - 3)) - 4096;
            if (i == t1 && t1 < 3) code += 2;
            level[i] = (code & 1) ? (-code - 1) >> 1 : (code + 2) >> 1;
            if (level[i] < -32767 || level[i] > 32767) return -1;
            if (suffix_len == 0) suffix_len = 1;
            const int a = level[i] < 0 ? -level[i] : level[i];
            if (a > (3 << (suffix_len - 1)) && suffix_len < 6) suffix_len++;
        }
        int zeros_left = 0;
        if (tc < maxc) {
            if (maxc == 4) {
                const uint16_t e = L.ctz[tc - 1][br.peek(3)];
                if (!e) return -1;
                br.skip(e >> 8); zeros_left = e & 255;
            } else {
                const uint16_t e = L.tz[tc - 1][br.peek(9)];
                if (!e) return -1;
                br.skip(e >> 8); zeros_left = e & 255;
                if (zeros_left + tc > maxc) return -1;   // (a 15-coefficient block shares the 16-coefficient tables)
            }
        }
        int pos = tc + zeros_left - 1;   // scan position of the first (highest-frequency) coefficient
        for (int i = 0; i < tc; i++) {
            if (pos < 0) return -1;
            put_level(out + pos, level[i]);
            int run = 0;
            if (i < tc - 1 && zeros_left > 0) {
                const uint16_t e = L.run[(zeros_left < 7 ? zeros_left : 7) - 1][br.peek(11)];
                if (!e) return -1;
                br.skip(e >> 8); run = e & 255;
                if (run > zeros_left) return -1;
                zeros_left -= run;
            }
            pos -= 1 + run;
        }
        return br.bad() ? -1 : tc;
    }

    // ---- 8.4.1.3 on the 4x4 grid (sub-macroblock partitions go down to 4x4): neighbours A (left of the partition's first block),
    // B (above it), C (above-right of its top-right block; D above-left when C is not available - outside, or later in decoding
    // order, 6.4.11.7) ----
    struct Cand { bool av; int ref, x, y; };
    Cand block(int mx, int my, int bx, int by, bool self = false)
    {
        Cand c{false, -1, 0, 0};
        if (!self && !avail(mx, my)) return c;
        c.av = true;
        const size_t i = (size_t)my * picp_->mbw + mx;
        if (!intra(picp_->mb[i].type)) {
            c.ref = picp_->refq[i * 4 + 2 * (by >> 1) + (bx >> 1)];
            c.x = picp_->mv4[i * 32 + 2 * (4 * by + bx)];
            c.y = picp_->mv4[i * 32 + 2 * (4 * by + bx) + 1];
        }
        return c;
    }
    static int med3(int a, int b, int c) { return a > b ? (b > c ? b : (a > c ? c : a)) : (a > c ? a : (b > c ? c : b)); }
    // partition of w4 x h4 blocks at (x4, y4); mbpart: a macroblock partition (the directional rules of 16x8 / 8x16 apply)
    void predict(int mx, int my, int x4, int y4, int w4, int h4, int ref, int& px, int& py, bool mbpart, bool skip_rule = false)
    {
        const Cand A = x4 == 0 ? block(mx - 1, my, 3, y4) : block(mx, my, x4 - 1, y4, true);
        const Cand B = y4 == 0 ? block(mx, my - 1, x4, 3) : block(mx, my, x4, y4 - 1, true);
        Cand C{false, -1, 0, 0};
        if (y4 == 0) C = x4 + w4 < 4 ? block(mx, my - 1, x4 + w4, 3) : block(mx + 1, my - 1, 0, 3);
        else if (x4 + w4 < 4 && blk_idx(x4 + w4, y4 - 1) < blk_idx(x4, y4)) C = block(mx, my, x4 + w4, y4 - 1, true);   // decoded before this partition
        if (!C.av) {
            if (x4 == 0 && y4 == 0) C = block(mx - 1, my - 1, 3, 3);
            else if (y4 == 0) C = block(mx, my - 1, x4 - 1, 3);
            else if (x4 == 0) C = block(mx - 1, my, 3, y4 - 1);
            else C = block(mx, my, x4 - 1, y4 - 1, true);
        }
        if (skip_rule && (!A.av || !B.av || (A.ref == 0 && A.x == 0 && A.y == 0) || (B.ref == 0 && B.x == 0 && B.y == 0))) { px = py = 0; return; }
        if (mbpart && w4 == 4 && h4 == 2) {
            if (y4 == 0 && B.ref == ref) { px = B.x; py = B.y; return; }
            if (y4 == 2 && A.ref == ref) { px = A.x; py = A.y; return; }
        } else if (mbpart && w4 == 2 && h4 == 4) {
            if (x4 == 0 && A.ref == ref) { px = A.x; py = A.y; return; }
            if (x4 == 2 && C.ref == ref) { px = C.x; py = C.y; return; }
        }
        Cand a = A, b = B, c = C;
        if (!b.av && !c.av && a.av) { b = a; c = a; }
        const int n = (a.ref == ref) + (b.ref == ref) + (c.ref == ref);
        if (n == 1) { const Cand& o = a.ref == ref ? a : (b.ref == ref ? b : c); px = o.x; py = o.y; }
        else { px = med3(a.x, b.x, c.x); py = med3(a.y, b.y, c.y); }
    }

    // RefPicList0 of a P slice (frames, short-term pictures only): 8.2.4.1 PicNum = FrameNumWrap, 8.2.4.2.1 the initial order
    // (descending PicNum), 8.2.4.3.1 the slice header's modification commands.  list[r] = age of entry r: position in dpb_fn_
    // (0 = the reference picture decoded last), which is how the decoder's reconstruction ring is indexed.
    bool ref_list(BitReader& br, const Sps& sps, int frame_num, int num_ref, bool modify, int list[4])
    {
        const int max_fn = 1 << sps.log2_max_frame_num, n = (int)dpb_fn_.size();
        int picnum[4], order[5] = {-1, -1, -1, -1, -1}, cnt = 0;
        for (int i = 0; i < n && i < 4; i++) picnum[i] = dpb_fn_[i] > frame_num ? dpb_fn_[i] - max_fn : dpb_fn_[i];
        for (int i = 0; i < n && i < 4; i++) {
            int k = cnt++;
            while (k > 0 && picnum[order[k - 1]] < picnum[i]) { order[k] = order[k - 1]; k--; }
            order[k] = i;
        }
        for (int i = num_ref < 4 ? num_ref : 4; i < 5; i++) order[i] = -1;   // the initial list is cut to num_ref_idx_l0_active entries
        if (modify) {
            if (num_ref > 4) { fail("ref_pic_list_modification with more than 4 active references"); return false; }
            int pred = frame_num, idx = 0;
            for (int guard = 0;; guard++) {
                const unsigned idc = br.ue();
                if (br.bad() || guard > 32) { fail("ref_pic_list_modification damaged"); return false; }
                if (idc == 3) break;
                if (idc == 2) { fail("long-term reference pictures (modification_of_pic_nums_idc 2)"); return false; }
                if (idc > 3) { fail("modification_of_pic_nums_idc %d", (int)idc); return false; }
                const unsigned du = br.ue();
                if (du >= (unsigned)max_fn || idx >= num_ref) { fail("ref_pic_list_modification damaged"); return false; }
                const int diff = (int)du + 1;
                int nowrap;
                if (idc == 0) { nowrap = pred - diff; if (nowrap < 0) nowrap += max_fn; }
                else { nowrap = pred + diff; if (nowrap >= max_fn) nowrap -= max_fn; }
                pred = nowrap;
                const int pn = nowrap > frame_num ? nowrap - max_fn : nowrap;
                int k = -1;
                for (int i = 0; i < n && i < 4; i++) if (picnum[i] == pn) k = i;
                if (k < 0) { fail("ref_pic_list_modification names picture number %d, which is not a reference picture", pn); return false; }
                for (int c = num_ref; c > idx; c--) order[c] = order[c - 1];
                order[idx++] = k;
                int w = idx;
                for (int c = idx; c <= num_ref; c++) if (order[c] != k) order[w++] = order[c];
                for (; w <= num_ref; w++) order[w] = -1;
            }
        }
        for (int r = 0; r < 4; r++) list[r] = order[r] >= 0 ? order[r] : r;   // (entries without a picture are refused by the decoder)
        return true;
    }

    bool parse_slice(BitReader& br, bool idr, int ref_idc, bool& have_pic, int& next_mb)
    {
        const unsigned fm = br.ue(), stu = br.ue();
        if (fm > 65535u || stu > 9u) { fail("slice header damaged (first_mb_in_slice / slice_type)"); return false; }
        const int first_mb = (int)fm;
        int st = (int)stu;
        if (st > 4) st -= 5;
        if (st != 0 && st != 2) { fail("slice_type %d (only I and P)", st); return false; }
        const unsigned pps_id = br.ue();
        if (pps_id > 255 || !pps_[pps_id].valid || !sps_[pps_[pps_id].sps_id].valid) { fail("slice refers to a missing parameter set"); return false; }
        const Pps& pps = pps_[pps_id];
        active_sps_ = pps.sps_id;
        const Sps& sps = sps_[active_sps_];
        const int frame_num = (int)br.u(sps.log2_max_frame_num);
        if (idr) br.ue();               // idr_pic_id
        if (sps.poc_type == 0) {
            br.u(sps.log2_max_poc_lsb);
            if (pps.bottom_field_pic_order) br.se();
        } else if (sps.poc_type == 1 && !sps.delta_pic_order_always_zero) {
            br.se();
            if (pps.bottom_field_pic_order) br.se();
        }
        if (pps.redundant_pic_cnt) br.ue();
        int num_ref = pps.num_ref_default;
        int list[4] = {0, 1, 2, 3};
        if (st == 0) {
            if (br.u(1)) { const unsigned nr = br.ue(); num_ref = nr > 31 ? 99 : (int)nr + 1; }
            if (idr) { fail("P slice in an IDR picture"); return false; }
            if (dpb_fn_.empty()) { fail("P slice without a reference picture (none decoded yet, or lost with a refused access unit)"); return false; }
            if (br.bad() || num_ref < 1 || num_ref > 3) { fail("num_ref_idx_l0_active %d (the decoder holds three reference pictures)", num_ref); return false; }
            if (!ref_list(br, sps, frame_num, num_ref, br.u(1) != 0, list)) return false;
        }
        if (ref_idc != 0) {
            if (idr) { br.u(1); if (br.u(1)) { fail("long_term_reference_flag"); return false; } }
            else if (br.u(1)) { fail("adaptive_ref_pic_marking_mode_flag"); return false; }
        }
        const int qd = br.se();
        const int qp = (qd < -64 || qd > 64) ? -1 : pps.pic_init_qp + qd;
        int idc = 0, oa = 0, ob = 0;
        if (pps.deblock_control) {
            const unsigned di = br.ue();
            idc = di > 2 ? 3 : (int)di;
            if (idc != 1) {
                oa = br.se(); ob = br.se();
                if (oa < -6 || oa > 6 || ob < -6 || ob > 6) { fail("slice_alpha_c0_offset_div2 / slice_beta_offset_div2 out of range"); return false; }
                oa *= 2; ob *= 2;
            }
        }
        if (br.bad() || qp < 0 || qp > 51 || idc > 2 || num_ref < 1 || num_ref > 3) { fail("slice header damaged"); return false; }

        if (!have_pic) {   // first slice of the picture
            if (first_mb != 0) { fail("first slice of the access unit starts at macroblock %d", first_mb); return false; }
            // 7.4.3 without gaps_in_frame_num: an IDR picture has frame_num 0, every other picture PrevRefFrameNum + 1.  Anything
            // else means a reference picture went missing between the two: predicting from the ring would use the wrong pictures
            if (idr && frame_num != 0) { fail("frame_num %d in an IDR picture", frame_num); return false; }
            if (!idr && prev_ref_fn_ >= 0 && frame_num != ((prev_ref_fn_ + 1) & ((1 << sps.log2_max_frame_num) - 1))) {
                fail("frame_num %d does not follow the previous reference picture's %d: a picture is missing", frame_num, prev_ref_fn_);
                return false;
            }
            picp_->mbw = sps.mbw; picp_->mbh = sps.mbh;
            picp_->width = 16 * sps.mbw - 2 * (sps.crop_l + sps.crop_r); picp_->height = 16 * sps.mbh - 2 * (sps.crop_t + sps.crop_b);
            picp_->idr = idr; picp_->is_ref = ref_idc != 0; picp_->qp = qp; picp_->deblock_idc = idc; picp_->slice_rows = 0;
            for (int r = 0; r < 3; r++) picp_->ref_age[r] = list[r];
            cur_frame_num_ = frame_num;
            picp_->num_ref_active = st == 0 ? num_ref : 0; picp_->t8x8_mode = pps.t8x8 ? 1 : 0; picp_->profile_idc = sps.profile_idc;
            picp_->has_pcm = picp_->has_intra = picp_->has_inter = false;
            picp_->cqo[0] = pps.cqo[0]; picp_->cqo[1] = pps.cqo[1]; picp_->filter_oa = oa; picp_->filter_ob = ob;
            picp_->one_qp = pps.cqo[0] == 0 && pps.cqo[1] == 0 && oa == 0 && ob == 0;
            const size_t n = (size_t)sps.mbw * sps.mbh;
            picp_->mbqp.assign(n, (uint8_t)qp);
            picp_->mb.assign(n, MbRec{});
            picp_->mvq.assign(n * 8, 0);
            picp_->mv4.assign(n * 32, 0);
            picp_->refq.assign(n * 4, 0xFF);
            picp_->mbavail.assign(n, 0);
            picp_->slices = 1;
            picp_->aux.assign(n * 16, 0);
            picp_->levels8.assign(n * L_STRIDE, 0);
            picp_->big.clear();
            have_pic = true;
        } else {
            if (first_mb != next_mb) { fail("slices out of order (first_mb_in_slice %d, expected %d)", first_mb, next_mb); return false; }
            // bands of equal height (the last may be shorter): slice k starts at row k * slice_rows; anything else is "any shape"
            if (picp_->slices == 1 && first_mb % picp_->mbw == 0) picp_->slice_rows = first_mb / picp_->mbw;
            if (picp_->slice_rows <= 0 || first_mb != picp_->slices * picp_->slice_rows * picp_->mbw) picp_->slice_rows = -1;
            picp_->slices++;
            if (qp != picp_->qp) picp_->one_qp = false;
            if (idc != picp_->deblock_idc) { fail("disable_deblocking_filter_idc differs between slices"); return false; }
            if (oa != picp_->filter_oa || ob != picp_->filter_ob) { fail("deblocking filter offsets differ between slices"); return false; }
            if (pps.cqo[0] != picp_->cqo[0] || pps.cqo[1] != picp_->cqo[1]) { fail("chroma QP offsets differ between slices"); return false; }
            if (st == 0 && picp_->num_ref_active && num_ref != picp_->num_ref_active) { fail("num_ref_idx_active differs between slices"); return false; }
            if (st == 0 && !picp_->num_ref_active) { picp_->num_ref_active = num_ref; for (int r = 0; r < 3; r++) picp_->ref_age[r] = list[r]; }
            else if (st == 0 && (list[0] != picp_->ref_age[0] || list[1] != picp_->ref_age[1] || list[2] != picp_->ref_age[2])) { fail("reference picture lists differ between slices"); return false; }
        }
        slice_first_ = first_mb; slice_type_ = st; slice_qp_ = qp; num_ref_ = num_ref;
        qp_ = qp;   // QP_Y,PRED of the slice's first macroblock (7.4.5)
        constrained_ = pps.constrained_intra;

        // ---- slice_data (7.3.4) ----
        int addr = first_mb;
        const int nmb = picp_->mbw * picp_->mbh;
        bool more = true;
        while (more) {
            if (st == 0) {
                unsigned run = br.ue();
                // (compared as unsigned: a code word with 31 leading zeros carries a value of 2^31 or more, which a cast to int
                // would turn negative and let through)
                if (br.bad() || addr > nmb || run > (unsigned)(nmb - addr)) { fail("mb_skip_run past the picture"); return false; }
                while (run--) { skip_mb(addr % picp_->mbw, addr / picp_->mbw); addr++; }
                more = br.more_data();
                if (!more) break;
            }
            if (addr >= nmb) { fail("slice data past the picture"); return false; }
            if (!macroblock(br, addr % picp_->mbw, addr / picp_->mbw, pps)) return false;
            addr++;
            more = br.more_data();
        }
        next_mb = addr;
        return true;
    }

    // the vector of the w4 x h4 blocks at (x4, y4); a quadrant's entry of mvq is its first block's
    void set_vectors(int mx, int my, int x4, int y4, int w4, int h4, int x, int y)
    {
        const size_t i = (size_t)my * picp_->mbw + mx;
        for (int by = y4; by < y4 + h4; by++)
            for (int bx = x4; bx < x4 + w4; bx++) {
                int16_t* v = &picp_->mv4[i * 32 + 2 * (4 * by + bx)];
                v[0] = (int16_t)x; v[1] = (int16_t)y;
                if (!(bx & 1) && !(by & 1)) { int16_t* q = &picp_->mvq[i * 8 + 2 * (2 * (by >> 1) + (bx >> 1))]; q[0] = (int16_t)x; q[1] = (int16_t)y; }
            }
    }
    void set_refs(int mx, int my, int r0, int r1, int r2, int r3)
    {
        uint8_t* r = &picp_->refq[((size_t)my * picp_->mbw + mx) * 4];
        r[0] = (uint8_t)r0; r[1] = (uint8_t)r1; r[2] = (uint8_t)r2; r[3] = (uint8_t)r3;
    }
    // bits 0..3: the neighbour (left, above, above-right, above-left) is in the picture, in this slice, decoded before (6.4.9);
    // bits 4..7: and may be used for intra prediction - with constrained_intra_pred_flag a macroblock coded in Inter
    // prediction mode may not (8.3.1.2, 8.3.3, 8.3.4)
    bool intra_avail(int mx, int my) { return avail(mx, my) && (!constrained_ || intra(M(mx, my).type)); }
    void set_avail(int mx, int my)
    {
        picp_->mbavail[(size_t)my * picp_->mbw + mx] =
            (uint8_t)((avail(mx - 1, my) ? 1 : 0) | (avail(mx, my - 1) ? 2 : 0) | (avail(mx + 1, my - 1) ? 4 : 0) | (avail(mx - 1, my - 1) ? 8 : 0) |
                      (intra_avail(mx - 1, my) ? 16 : 0) | (intra_avail(mx, my - 1) ? 32 : 0) | (intra_avail(mx + 1, my - 1) ? 64 : 0) | (intra_avail(mx - 1, my - 1) ? 128 : 0));
    }
    void set_qp(int mx, int my, int qp)
    {
        picp_->mbqp[(size_t)my * picp_->mbw + mx] = (uint8_t)qp;
        if (qp != picp_->qp) picp_->one_qp = false;
    }
    void skip_mb(int mx, int my)
    {
        MbRec& m = M(mx, my);
        m = MbRec{};
        m.type = T_PSKIP;
        m.chroma_mode = 0;   // ref_idx_l0 = 0
        set_avail(mx, my);
        set_qp(mx, my, qp_);
        int px, py;
        set_refs(mx, my, 0, 0, 0, 0);
        predict(mx, my, 0, 0, 4, 4, 0, px, py, true, true);
        m.mvx = (int16_t)px; m.mvy = (int16_t)py;
        set_vectors(mx, my, 0, 0, 4, 4, px, py);
        picp_->has_inter = true;
    }

    bool macroblock(BitReader& br, int mx, int my, const Pps& pps)
    {
        MbRec& m = M(mx, my);
        m = MbRec{};
        int8_t* lv = &picp_->levels8[((size_t)my * picp_->mbw + mx) * L_STRIDE];
        uint8_t* am = &picp_->aux[((size_t)my * picp_->mbw + mx) * 16];
        unsigned t = br.ue();
        bool is_intra = slice_type_ == 2;
        if (slice_type_ == 0 && t >= 5) { is_intra = true; t -= 5; }
        if (br.bad() || (is_intra && t > 25) || (!is_intra && t > 4)) { fail("mb_type %d", (int)t); return false; }
        int cbp = 0;
        bool i16 = false, t8flag = false;
        set_avail(mx, my);
        if (is_intra) set_refs(mx, my, 0xFF, 0xFF, 0xFF, 0xFF);
        bool all8x8 = true;   // NoSubMbPartSizeLessThan8x8Flag
        if (is_intra && t == 25) {   // I_PCM
            while (!br.aligned()) br.u(1);
            uint8_t* raw = (uint8_t*)lv;
            for (int i = 0; i < 384; i++) raw[i] = (uint8_t)br.u(8);
            if (br.bad()) { fail("I_PCM samples past the slice"); return false; }
            m.type = T_IPCM; m.cbp = 0x2F;
            memset(m.tc, 16, 24);
            picp_->has_pcm = picp_->has_intra = true;
            set_qp(mx, my, 0);        // its edges filter with qP 0; QP_Y,PRED of the next macroblock stays
            picp_->one_qp = false;
            return true;
        }
        if (is_intra) {
            picp_->has_intra = true;
            if (t == 0) {   // I_NxN
                m.type = T_I4;
                if (pps.t8x8 && br.u(1)) { fail("Intra8x8 (transform_size_8x8_flag in an I_NxN macroblock)"); return false; }
                for (int k = 0; k < 16; k++) {   // 8.3.1.1
                    const int x = (k & 1) | ((k >> 1) & 2), y = ((k >> 1) & 1) | ((k >> 2) & 2);
                    int mA, mB;
                    bool dc = false;
                    if (x > 0) mA = am[blk_idx(x - 1, y)];
                    else if (!intra_avail(mx - 1, my)) { dc = true; mA = 2; }
                    else mA = M(mx - 1, my).type == T_I4 ? (am - 16)[blk_idx(3, y)] : 2;
                    if (y > 0) mB = am[blk_idx(x, y - 1)];
                    else if (!intra_avail(mx, my - 1)) { dc = true; mB = 2; }
                    else mB = M(mx, my - 1).type == T_I4 ? (am - 16 * picp_->mbw)[blk_idx(x, 3)] : 2;
                    const int pm = dc ? 2 : (mA < mB ? mA : mB);
                    int mode = pm;
                    if (!br.u(1)) { const int rem = (int)br.u(3); mode = rem < pm ? rem : rem + 1; }
                    am[k] = (uint8_t)mode;
                }
            } else {
                i16 = true;
                m.type = T_I16;
                m.i16_mode = (uint8_t)((t - 1) & 3);
                cbp = (int)(((t - 1) >> 2) % 3) << 4 | ((t - 1) >= 12 ? 15 : 0);
            }
            const unsigned cm = br.ue();
            if (cm > 3) { fail("intra_chroma_pred_mode %d", (int)cm); return false; }
            m.chroma_mode = (uint8_t)cm;
        } else {
            picp_->has_inter = true;
            const int shape = t == 4 ? 3 : (int)t;
            m.type = (uint8_t)(shape == 0 ? T_P16 : T_P16X8 + shape - 1);
            const int nparts = shape == 0 ? 1 : (shape == 3 ? 4 : 2);
            int sub[4] = {0, 0, 0, 0};   // sub_mb_type: 0 8x8, 1 8x4, 2 4x8, 3 4x4
            if (shape == 3)
                for (int k = 0; k < 4; k++) {
                    const unsigned su = br.ue();
                    if (su > 3) { fail("sub_mb_type %d", (int)su); return false; }
                    sub[k] = (int)su;
                    if (su) all8x8 = false;
                }
            int ref[4] = {0, 0, 0, 0};
            if (num_ref_ > 1 && t != 4)
                for (int k = 0; k < nparts; k++) {
                    const unsigned ru = num_ref_ == 2 ? 1u - br.u(1) : br.ue();
                    if (ru >= (unsigned)num_ref_) { fail("ref_idx_l0 %d", (int)ru); return false; }
                    ref[k] = (int)ru;
                }
            // refs of the four quadrants, known before the first vector is predicted
            if (shape == 0) set_refs(mx, my, ref[0], ref[0], ref[0], ref[0]);
            else if (shape == 1) set_refs(mx, my, ref[0], ref[0], ref[1], ref[1]);
            else if (shape == 2) set_refs(mx, my, ref[0], ref[1], ref[0], ref[1]);
            else set_refs(mx, my, ref[0], ref[1], ref[2], ref[3]);
            m.chroma_mode = (uint8_t)ref[0];
            auto one_vector = [&](int x4, int y4, int w4, int h4, int r, bool mbpart, bool first) -> bool {
                int px, py;
                predict(mx, my, x4, y4, w4, h4, r, px, py, mbpart);
                const int dx = br.se(), dy = br.se();
                if (dx < -8192 || dx > 8191 || dy < -8192 || dy > 8191) { fail("mvd_l0 out of range"); return false; }
                const int vx = px + dx, vy = py + dy;
                if (vx < -16384 || vx > 16383 || vy < -16384 || vy > 16383) { fail("motion vector out of range"); return false; }
                set_vectors(mx, my, x4, y4, w4, h4, vx, vy);
                if (first) { m.mvx = (int16_t)vx; m.mvy = (int16_t)vy; }
                return true;
            };
            for (int k = 0; k < nparts; k++) {
                if (shape == 0) { if (!one_vector(0, 0, 4, 4, ref[0], true, true)) return false; }
                else if (shape == 1) { if (!one_vector(0, 2 * k, 4, 2, ref[k], true, k == 0)) return false; }
                else if (shape == 2) { if (!one_vector(2 * k, 0, 2, 4, ref[k], true, k == 0)) return false; }
                else {
                    const int qx = 2 * (k & 1), qy = 2 * (k >> 1);
                    const int nsub = sub[k] == 0 ? 1 : (sub[k] == 3 ? 4 : 2);
                    for (int j = 0; j < nsub; j++) {
                        int x4 = qx, y4 = qy, w4 = 2, h4 = 2;
                        if (sub[k] == 1) { y4 += j; h4 = 1; }
                        else if (sub[k] == 2) { x4 += j; w4 = 1; }
                        else if (sub[k] == 3) { x4 += j & 1; y4 += j >> 1; w4 = h4 = 1; }
                        if (!one_vector(x4, y4, w4, h4, ref[k], false, k == 0 && j == 0)) return false;
                    }
                }
            }
        }
        if (!i16) {
            const unsigned code = br.ue();
            if (code > 47) { fail("coded_block_pattern code %d", (int)code); return false; }
            cbp = is_intra ? vlc().code2cbp_intra[code] : vlc().code2cbp_inter[code];
            if ((cbp & 15) && pps.t8x8 && !is_intra && all8x8) t8flag = br.u(1) != 0;
        }
        m.cbp = (uint8_t)cbp;
        if (!is_intra) m.i16_mode = t8flag ? 1 : 0;
        if (cbp > 0 || i16) {
            const int dq = br.se();
            if (dq < -26 || dq > 25) { fail("mb_qp_delta %d", dq); return false; }
            qp_ = (qp_ + dq + 52) % 52;
            // residual (7.3.5.3)
            if (i16) {
                if (residual_block(br, lv + L_LUMA_DC, 16, nc_luma(mx, my, 0, 0)) < 0) { fail("Intra16x16 DC levels"); return false; }
            }
            for (int b8 = 0; b8 < 4; b8++)
                for (int k = 0; k < 4; k++) {
                    const int b = 4 * b8 + k, bx = (b & 1) | ((b >> 1) & 2), by = ((b >> 1) & 1) | ((b >> 2) & 2);
                    if (!(cbp & (1 << b8))) continue;
                    const int n = residual_block(br, lv + L_LUMA + b * 16 + (i16 ? 1 : 0), i16 ? 15 : 16, nc_luma(mx, my, bx, by));
                    if (n < 0) { fail("luma levels of macroblock %d", my * picp_->mbw + mx); return false; }
                    m.tc[b] = (uint8_t)n;
                }
            if (cbp >> 4) {
                for (int pl = 0; pl < 2; pl++)
                    if (residual_block(br, lv + L_CHROMA_DC + 4 * pl, 4, -1) < 0) { fail("chroma DC levels"); return false; }
                if ((cbp >> 4) == 2)
                    for (int pl = 0; pl < 2; pl++)
                        for (int k = 0; k < 4; k++) {
                            const int n = residual_block(br, lv + L_CHROMA_AC + (4 * pl + k) * 16 + 1, 15, nc_chroma(mx, my, pl, k & 1, k >> 1));
                            if (n < 0) { fail("chroma AC levels"); return false; }
                            m.tc[16 + 4 * pl + k] = (uint8_t)n;
                        }
            }
        }
        if (br.bad()) { fail("macroblock %d runs past the slice", my * picp_->mbw + mx); return false; }
        set_qp(mx, my, qp_);
        return true;
    }
};

}  // namespace h264dec
