// media_amd/csrc/mi355x_h264.hip -- C ABI of include/mi355x_h264.h: device
// memory, stream, launches and the host-side framing (SPS/PPS/slice header,
// NAL wrapping, emulation prevention) around the HIP kernels in this directory.
//
// This file stands where the reference's adapter calls into libopenh264.so
// (/root/reference/video_codec/VideoEncoderOpenH264.cpp:142, :257, :344, :382,
// :408).  There is no CPU encode path here: without a HIP device create() fails.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <atomic>
#include <cstdarg>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "../../include/mi355x_h264.h"
#include "dev_common.h"
#include "k_cavlc.h"
#include "k_deblock.h"
#include "k_intra.h"
#include "k_me.h"
#include "k_tq.h"
#include "k_dec.h"
#include "h264_parse.h"
#include "../../include/mi355x_h264_dec.h"

using namespace h264;

namespace {

// ---- host tables (ITU-T H.264 Table 8-15, A-1; quantiser of the reference model) ----
const uint8_t h_chroma_qp[52] = {0,  1,  2,  3,  4,  5,  6,  7,  8,  9,  10, 11, 12, 13, 14, 15, 16, 17,
                                 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 29, 30, 31, 32, 32, 33,
                                 34, 34, 35, 35, 36, 36, 37, 37, 37, 38, 38, 38, 39, 39, 39, 39};
const uint8_t h_dequant_v[6][3] = {{10, 16, 13}, {11, 18, 14}, {13, 20, 16}, {14, 23, 18}, {16, 25, 20}, {18, 29, 23}};
const uint16_t h_quant_mf[6][3] = {{13107, 5243, 8066}, {11916, 4660, 7490}, {10082, 4194, 6554},
                                   {9362, 3647, 5825},  {8192, 3355, 5243},  {7282, 2893, 4559}};
const uint8_t h_lambda[52] = {1,  1,  1,  1,  1,  1,  1,  1,  1,  1,  1,  1,  1,  1,  1,  1,  2,  2,
                              2,  2,  3,  3,  3,  4,  4,  4,  5,  6,  6,  7,  8,  9,  10, 11, 13, 14,
                              16, 18, 20, 23, 25, 29, 32, 36, 40, 45, 51, 57, 64, 72, 81, 91};
const struct { uint8_t idc; uint32_t mbps, fs; } h_levels[] = {
    {10, 1485, 99},     {11, 3000, 396},     {12, 6000, 396},     {13, 11880, 396},   {20, 11880, 396},  {21, 19800, 792},
    {22, 20250, 1620},  {30, 40500, 1620},   {31, 108000, 3600},  {32, 216000, 5120}, {40, 245760, 8192}, {41, 245760, 8192},
    {42, 522240, 8704}, {50, 589824, 22080}, {51, 983040, 36864}, {52, 2073600, 36864}};
const uint8_t h_alpha[52] = {0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,   0,   0,   0,   4,   4,
                             5,  6,  7,  8,  9,  10, 12, 13, 15, 17, 20, 22, 25,  28,  32,  36,  40,  45,
                             50, 56, 63, 71, 80, 90, 101, 113, 127, 144, 162, 182, 203, 226, 255, 255};
const uint8_t h_beta[52] = {0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  2,  2,
                            2,  3,  3,  3,  3,  4,  4,  4,  6,  6,  7,  7,  8,  8,  9,  9,  10, 10,
                            11, 11, 12, 12, 13, 13, 14, 14, 15, 15, 16, 16, 17, 17, 18, 18};
const uint8_t h_tc0[52][3] = {
    {0, 0, 0},   {0, 0, 0},   {0, 0, 0},    {0, 0, 0},    {0, 0, 0},    {0, 0, 0},   {0, 0, 0},   {0, 0, 0},  {0, 0, 0},
    {0, 0, 0},   {0, 0, 0},   {0, 0, 0},    {0, 0, 0},    {0, 0, 0},    {0, 0, 0},   {0, 0, 0},   {0, 0, 0},  {0, 0, 1},
    {0, 0, 1},   {0, 0, 1},   {0, 0, 1},    {0, 1, 1},    {0, 1, 1},    {1, 1, 1},   {1, 1, 1},   {1, 1, 1},  {1, 1, 1},
    {1, 1, 2},   {1, 1, 2},   {1, 1, 2},    {1, 1, 2},    {1, 2, 3},    {1, 2, 3},   {2, 2, 3},   {2, 2, 4},  {2, 3, 4},
    {2, 3, 4},   {3, 3, 5},   {3, 4, 6},    {3, 4, 6},    {4, 5, 7},    {4, 5, 8},   {4, 6, 9},   {5, 7, 10}, {6, 8, 11},
    {6, 8, 13},  {7, 10, 14}, {8, 11, 16},  {9, 12, 18},  {10, 13, 20}, {11, 15, 23}, {13, 17, 25}};

// ---- host bit writer for parameter sets and slice headers ----
struct HostBits {
    std::vector<uint8_t> bytes;
    uint64_t nbits = 0;
    void put(int n, uint32_t v)
    {
        for (int i = n - 1; i >= 0; i--) {
            if ((nbits >> 3) >= bytes.size()) bytes.push_back(0);
            if ((v >> i) & 1) bytes[nbits >> 3] |= (uint8_t)(0x80 >> (nbits & 7));
            nbits++;
        }
    }
    void ue(uint32_t v)
    {
        uint32_t x = v + 1;
        int n = 0;
        while ((x >> n) > 1) n++;
        put(n, 0);
        put(n + 1, x);
    }
    void se(int32_t v) { ue(v > 0 ? (uint32_t)(2 * v - 1) : (uint32_t)(-2 * v)); }
    void trailing()
    {
        put(1, 1);
        while (nbits & 7) put(1, 0);
    }
};

size_t nal_escape(const uint8_t* rbsp, size_t n, uint8_t* out)
{
    size_t o = 0;
    int zeros = 0;
    for (size_t i = 0; i < n; i++) {
        if (zeros == 2 && rbsp[i] <= 3) { out[o++] = 3; zeros = 0; }
        out[o++] = rbsp[i];
        zeros = rbsp[i] == 0 ? zeros + 1 : 0;
    }
    return o;
}

void append_nal(std::vector<uint8_t>& au, int ref_idc, int type, const HostBits& b)
{
    const uint8_t sc[5] = {0, 0, 0, 1, (uint8_t)((ref_idc << 5) | type)};
    au.insert(au.end(), sc, sc + 5);
    std::vector<uint8_t> esc(b.bytes.size() * 3 / 2 + 4);
    const size_t n = nal_escape(b.bytes.data(), b.bytes.size(), esc.data());
    au.insert(au.end(), esc.begin(), esc.begin() + n);
}

void fill_quant(Quant& q, int qp)
{
    q.qp = qp;
    q.qbits = 15 + qp / 6;
    q.f_intra = (1 << q.qbits) / 3;
    q.f_inter = (1 << q.qbits) / 6;
    for (int c = 0; c < 3; c++) {
        q.mf[c] = h_quant_mf[qp % 6][c];
        q.dq[c] = h_dequant_v[qp % 6][c] << (qp / 6);
        q.thr_inter[c] = (int)((((int64_t)1 << q.qbits) - q.f_inter + q.mf[c] - 1) / q.mf[c]);
    }
    q.thr_dc_inter = (int)((((int64_t)1 << (q.qbits + 1)) - 2 * (int64_t)q.f_inter + q.mf[0] - 1) / q.mf[0]);
    static const uint8_t v8[6][6] = {{20, 18, 32, 19, 25, 24}, {22, 19, 35, 21, 28, 26}, {26, 23, 42, 24, 33, 31},
                                     {28, 25, 45, 26, 35, 33}, {32, 28, 51, 30, 40, 38}, {36, 32, 58, 34, 46, 43}};
    static const uint16_t m8[6][6] = {{13107, 11428, 20972, 12222, 16777, 15481}, {11916, 10826, 19174, 11058, 14980, 14290},
                                      {10082, 8943, 15978, 9675, 12710, 11985},   {9362, 8228, 14913, 8931, 11984, 11259},
                                      {8192, 7346, 13159, 7740, 10486, 9777},     {7282, 6428, 11570, 6830, 9118, 8640}};
    for (int c = 0; c < 6; c++) { q.mf8[c] = m8[qp % 6][c]; q.ls8[c] = 16 * v8[qp % 6][c]; }
}

// everything one picture QP fixes for the kernels (FrameParams qy / qc / lambda / sad_nz; QpEntry of the indirect launches)
void fill_qp(Quant& qy, Quant& qc, int& lambda, int& sad_nz, int qp)
{
    fill_quant(qy, qp);
    fill_quant(qc, h_chroma_qp[qp]);
    lambda = h_lambda[qp];
    // k_me's shortcut for the "quantises to nothing" test: 64 sqrt(sum over the 16 positions of t^2 / (n_i n_j)), rounded up
    const double t0 = qy.thr_inter[0], t1 = qy.thr_inter[1], t2 = qy.thr_inter[2];
    sad_nz = (int)std::ceil(64.0 * std::sqrt(4 * t0 * t0 / 16.0 + 4 * t1 * t1 / 100.0 + 8 * t2 * t2 / 40.0)) + 1;
}

constexpr int NSLOT = 3;          // access-unit slots in flight

struct Slot {
    uint32_t* d_bitbuf = nullptr;   // device slice payload (zeroed before use)
    SliceInfo* d_info = nullptr;
    SliceInfo* h_info = nullptr;    // pinned
    unsigned* h_err = nullptr;      // pinned copy of the wavefront kernels' timeout flag
    uint8_t* h_au = nullptr;        // pinned access unit buffer
    size_t payload_off = 0;         // offset of the slice payload inside h_au
    size_t au_start = 0;            // offset of the first byte of the access unit
    int nal_hdr = 0;
    bool idr = false;
    bool busy = false;
    hipEvent_t done = nullptr;
    hipEvent_t recon_ready = nullptr, entropy_done = nullptr;   // fork / join of the entropy-coding stream
    // stats events of this frame: pairs (start, stop, kernel id, launches, mbs)
    struct Ev { hipEvent_t a, b; int k; uint32_t launches, mbs; };
    std::vector<Ev> evs;
};

}  // namespace

struct mi355x_h264_encoder {
    mi355x_h264_config cfg{};
    int mbw = 0, mbh = 0, cw = 0, ch = 0, nmb = 0, level_idc = 0;
    int device = 0;
    int G = 1;                               // lockstep batch: closed GOPs / streams encoded together
    int nsl = 1;                             // slices per picture: bands of sl.rows macroblock rows
    SliceRows sl{};
    size_t slice_cap = 0;                    // bytes of payload buffer per slice (multiple of 16)
    // slice bands over several GPUs: this instance codes slices b_sl0 .. b_sl0 + b_nsl - 1 = rows b_row0 .. b_row0 + b_rows - 1
    int b_sl0 = 0, b_nsl = 1, b_row0 = 0, b_rows = 0, b_nmb = 0;
    size_t st_y = 0, st_c = 0, st_bitbuf_bytes = 0, st_au = 0, st_handoff = 0;  // per-item strides
    hipStream_t stream = nullptr;
    hipStream_t stream_ec = nullptr;         // entropy coding runs here, beside the deblocking wavefront (= stream when the process holds many engines)
    std::atomic<int>* counted_live = nullptr;
    enum { MAX_REFS = 3 };
    int nrefs = 1, nbuf = 2;                 // reference frames searched (config.refs) and reconstruction buffers (nrefs + 1)
    uint8_t* d_planes[MAX_REFS + 1][3] = {{nullptr}};  // ring: [index][plane]; `cur` is written, cur - 1 - r (mod nbuf) is ref_idx_l0 r
    // the planes lie [batch item][ring slot]: d_planes[b][p] = d_plane_base[p] + b * st_ring, st_y / st_c (the item strides) = nbuf
    // ring strides - so that an indirect launch (stream hub) can address every item's OWN ring slot from one base pointer
    uint8_t* d_plane_base[3] = {nullptr, nullptr, nullptr};
    size_t st_ring_y = 0, st_ring_c = 0;
    QpEntry* d_qtab = nullptr;               // [52] quantiser constants by QP (indirect launches)
    int nslots = NSLOT;                      // access-unit slots allocated (the hub's engine needs one)
    uint8_t* d_pre[3] = {nullptr};           // copy of the reconstruction before the loop filter (debug)
    int cur = 0;                             // index written by the picture being encoded
    bool pair_filter = true;                 // two macroblock rows per wave in the loop filter for lockstep batches of pair_min_batch pictures or more
    int pair_min_batch = 8;                  // (MI355X_H264_PAIR_FILTER=N sets it, 0 turns the pair form off)
    MbInfo* d_mb = nullptr;
    int16_t* d_levels = nullptr;
    int16_t* d_mvd = nullptr;
    uint8_t* d_aux = nullptr;                // [G][nmb][16] Intra4x4 modes
    int16_t* d_mvq = nullptr;                // [G][nmb][8] vectors of the four 8x8 quadrants of inter macroblocks
    uint32_t* d_me_total = nullptr;          // [G][nmb] best motion cost so far over the reference pictures (k_me, one launch each)
    int* d_pmv = nullptr;                    // [G][nmb] the previous picture's vectors, parked for the later launches
    uint16_t* d_slotbits = nullptr;
    unsigned long long* d_slotcode = nullptr;
    uint32_t* d_mbbits = nullptr;
    unsigned* d_anybs = nullptr;             // [G] picture serial when any boundary strength is non-zero
    unsigned* d_anypcm = nullptr;            // [G] == pic_serial: the picture holds an I_PCM macroblock (not loop-filtered)
    unsigned* d_anyintra = nullptr;          // [G] == pic_serial: P picture with macroblocks for the intra pass
    unsigned pic_serial = 0;                 // changes every picture, never 0
    int32_t* d_prevcoded = nullptr;          // [G][nmb + 1] skip-run helper (k_skip_scan)
    unsigned long long* d_handoff = nullptr; // row-to-row hand-off of the wavefront kernels
    uint32_t* d_bs = nullptr;                // boundary strengths, 32 B per macroblock
    uint16_t* d_me_cost = nullptr;           // [G][nmb] per-macroblock motion cost (scene-change statistic)
    std::vector<uint32_t> last_me_cost;      // of the last finished picture, per batch item
    unsigned serial = 0;
    bool diag_mode = false;                  // debug: one launch per wavefront step instead
    uint8_t* d_stage = nullptr;              // device copy of a host-supplied picture
    uint8_t* h_stage = nullptr;              // pinned staging for strided host input
    uint8_t* d_rgba = nullptr, *h_rgba = nullptr;   // RGBA pictures on their way to the conversion kernel (allocated with the first)
    size_t frame_bytes = 0, bitbuf_cap = 0, au_cap = 0;
    Slot slots[NSLOT];
    int next_slot = 0;
    std::vector<uint8_t> sps_pps;            // Annex-B SPS + PPS NALs
    std::vector<std::vector<uint8_t>> esc_buf;  // slow path: escaped access unit, per batch item
    long frames = 0;
    int frame_in_gop = 0, frame_num = 0, idr_id = 0, idr_step = 1, force_idr = 0;
    int qp = 26;
    bool keep_pre = false, stats_on = false;
    std::vector<hipEvent_t> ev_pool;
    int me_turn = 0;                         // this engine's id at the GPU's motion-search lock (0: takes no part)
    uint32_t p_intra_x16 = 0;                // intra macroblocks per P picture, recent pictures (x 16, a running mean): sizes k_pintra_rows' grid
    mi355x_h264_stats stats{};
    char err[256] = {0};
};

// ---------------------------------------------------------------------------------------------------------------
// One motion search of a lockstep batch at a time per GPU.
// Two instances beside each other are worth more than one because the dependency-bound kernels of one (loop filter, entropy
// coding, row wavefronts) run in the issue slots the other's motion search leaves.  Left to themselves the instances settle in
// whatever phase their first steps put them - search beside filter (good), or search beside search and filter beside filter (2 - 5 %
// less, run by run: section 7 of DESIGN.md).  A lock word in device memory per GPU keeps the searches apart: a one-wave kernel in
// front of a search takes it (compare-and-swap, sleeping between tries), a one-thread kernel behind the search gives it back.
// Whoever comes first goes first - no order is imposed, so an engine in its IDR step, or gone, holds nobody up (an ORDER between the
// engines' searches, by events or by counters, follows the order in which the host threads happened to queue them and left one
// instance idle for 1.4 ms of every step) - and the holder's search is already queued behind its acquire, so the lock is always given
// back; the wait gives up after TURN_TIMEOUT_US all the same.
// ---------------------------------------------------------------------------------------------------------------
__global__ void k_turn_acquire(unsigned* lock, unsigned id, int timeout_us)
{
    if (threadIdx.x) return;
    const long long t0 = wall_clock64();   // 100 MHz
    for (;;) {
        unsigned expect = 0u;
        if (__hip_atomic_compare_exchange_strong(lock, &expect, id, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
        if (wall_clock64() - t0 > (long long)timeout_us * 100) { __hip_atomic_store(lock, id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return; }
        __builtin_amdgcn_s_sleep(16);
    }
}
__global__ void k_turn_release(unsigned* lock, unsigned id)
{
    unsigned expect = id;
    (void)__hip_atomic_compare_exchange_strong(lock, &expect, 0u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

namespace {
enum { TURN_MIN_BATCH = 16, TURN_DEVICES = 16, TURN_TIMEOUT_US = 3000 };
struct MeTurns {
    std::mutex mu;
    unsigned* d_lock[TURN_DEVICES] = {};   // allocated with the first engine of the device, kept for the life of the process
    unsigned next_id = 1;
};
MeTurns g_turns;
const bool g_turns_on = !(getenv("MI355X_H264_ME_TURNS") && atoi(getenv("MI355X_H264_ME_TURNS")) == 0);

enum { PINTRA_SPARSE_MBS = 8 };   // intra macroblocks per P picture up to which k_pintra_rows takes the step's pictures one after the other
}  // namespace

namespace {

int fail(mi355x_h264_encoder* e, int code, const char* fmt, ...)
{
    if (e) {
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(e->err, sizeof(e->err), fmt, ap);
        va_end(ap);
    }
    return code;
}

#define HIPCHK(e, call)                                                                              \
    do {                                                                                             \
        hipError_t _r = (call);                                                                      \
        if (_r != hipSuccess) return fail((e), MI355X_H264_E_HIP, "%s: %s", #call, hipGetErrorString(_r)); \
    } while (0)

void build_parameter_sets(mi355x_h264_encoder* e)
{
    const int prof = e->cfg.profile_idc;
    HostBits s;
    s.put(8, (uint32_t)prof);
    s.put(8, prof == 66 ? 0xC0 : prof == 77 ? 0x40 : 0x00);
    s.put(8, (uint32_t)e->level_idc);
    s.ue(0);
    if (prof == 100) { s.ue(1); s.ue(0); s.ue(0); s.put(1, 0); s.put(1, 0); }
    s.ue(4);      // log2_max_frame_num_minus4
    s.ue(2);      // pic_order_cnt_type
    s.ue((uint32_t)e->nrefs);   // max_num_ref_frames (ref :290: 1; config.refs)
    s.put(1, 0);  // gaps_in_frame_num_value_allowed_flag
    s.ue((uint32_t)e->mbw - 1);
    s.ue((uint32_t)e->mbh - 1);
    s.put(1, 1);  // frame_mbs_only_flag
    s.put(1, 1);  // direct_8x8_inference_flag
    const int cr = (e->cw - e->cfg.width) / 2, cb = (e->ch - e->cfg.height) / 2;
    if (cr || cb) { s.put(1, 1); s.ue(0); s.ue((uint32_t)cr); s.ue(0); s.ue((uint32_t)cb); }
    else s.put(1, 0);
    s.put(1, 0);  // vui_parameters_present_flag
    s.trailing();
    HostBits p;
    p.ue(0); p.ue(0);
    p.put(1, 0);  // CAVLC
    p.put(1, 0);
    p.ue(0); p.ue((uint32_t)e->nrefs - 1); p.ue(0);   // slice groups, num_ref_idx_l0 / l1_default_active_minus1
    p.put(1, 0); p.put(2, 0);
    p.se(0); p.se(0); p.se(0);
    p.put(1, 1);  // deblocking_filter_control_present_flag
    p.put(1, 0); p.put(1, 0);
    if (prof == 100) { p.put(1, 1); p.put(1, 0); p.se(0); }   // transform_8x8_mode_flag = 1: inter macroblocks use the 8x8 transform (k_tq8)
    p.trailing();
    e->sps_pps.clear();
    append_nal(e->sps_pps, 3, 7, s);
    append_nal(e->sps_pps, 3, 8, p);
}

// slice_header() of 7.3.3 for this build's fixed choices, from slice_type on (first_mb_in_slice differs per slice and
// is written by k_bit_scan); returns bit count (< 64)
int avail_refs(const mi355x_h264_encoder* e, bool idr) { return idr ? 0 : std::min(e->nrefs, e->frame_in_gop); }

// frame_num, qp, nact (num_ref_idx_l0_active of a P slice): the picture's own - one per batch item in the stream hub's steps
int build_slice_header(const mi355x_h264_encoder* e, bool idr, int idr_id, bool no_filter, int frame_num, int qp, int nact, uint64_t* bits)
{
    HostBits h;
    h.ue(idr ? 7 : 5);
    h.ue(0);
    h.put(8, (uint32_t)frame_num);
    if (idr) h.ue((uint32_t)idr_id);
    if (!idr) {   // num_ref_idx_active_override_flag: the first pictures after an IDR have fewer reference pictures than the PPS announces
        if (nact != e->nrefs) { h.put(1, 1); h.ue((uint32_t)nact - 1); } else h.put(1, 0);
        h.put(1, 0);   // ref_pic_list_modification_flag_l0
    }
    if (idr) { h.put(1, 0); h.put(1, 0); } else h.put(1, 0);
    h.se(qp - 26);
    no_filter = no_filter || e->cfg.disable_deblock;            // (a picture with an I_PCM macroblock is not filtered)
    h.ue(no_filter ? 1 : e->nsl > 1 ? 2 : 0);   // several slices: no filtering across slice edges, the bands stay independent
    if (!no_filter) { h.se(0); h.se(0); }
    uint64_t v = 0;
    for (uint64_t i = 0; i < h.nbits; i++) v = (v << 1) | ((h.bytes[i >> 3] >> (7 - (i & 7))) & 1);
    *bits = v;
    return (int)h.nbits;
}

hipEvent_t get_event(mi355x_h264_encoder* e)
{
    if (!e->ev_pool.empty()) { hipEvent_t ev = e->ev_pool.back(); e->ev_pool.pop_back(); return ev; }
    hipEvent_t ev = nullptr;
    if (hipEventCreate(&ev) != hipSuccess) return nullptr;
    return ev;
}

struct StatScope {
    mi355x_h264_encoder* e; Slot* s; int k; uint32_t launches, mbs; hipStream_t st; hipEvent_t a = nullptr, b = nullptr;
    StatScope(mi355x_h264_encoder* e_, Slot* s_, int k_, uint32_t l, uint32_t m, hipStream_t st_ = nullptr)
        : e(e_), s(s_), k(k_), launches(l), mbs(m), st(st_ ? st_ : e_->stream)
    {
        if (e->stats_on) { a = get_event(e); b = get_event(e); if (a) (void)hipEventRecord(a, st); }
    }
    ~StatScope()
    {
        if (e->stats_on && a && b) { (void)hipEventRecord(b, st); s->evs.push_back({a, b, k, launches, mbs}); }
    }
};

// ---- one lockstep step: which pictures, where from, on which streams ----
// Direct (items == nullptr): the n = e->G batch items of the encoder, one QP, one ring position, consecutive idr_pic_ids - the
// closed-GOP batch of mi355x_h264_encode_gops_device and the single-picture calls.  Indirect (the stream hub below): position k
// of the grid is picture items[k] - its own batch item, ring slot, QP, frame_num and idr_pic_id; the kernels are the IND = true
// instantiations and read d_itemtab.  A step holds pictures of ONE type (IDR or P): the two run different kernels.
struct ItemPic { int item, cur, qp, frame_num, idr_id; };
struct Step {
    const uint8_t* d_src = nullptr; size_t src_item_stride = 0; bool nv12 = false; bool idr = false;
    int n = 1;
    const ItemPic* items = nullptr;
    const uint32_t* d_itemtab = nullptr;
    hipStream_t st = nullptr, ec = nullptr;
    hipEvent_t recon_ready = nullptr, entropy_done = nullptr, done = nullptr;
    unsigned* h_err = nullptr;
    Slot* slot = nullptr;   // payload / access-unit buffers (laid out by batch item) and, with stats on, the event list
    // out: where the access units lie in slot->h_au
    size_t au_start = 0, payload_off = 0;
    int nal_hdr = 0;
};

#define LAUNCH2(ind, KT, KF, grid, block, stream, ...)                                   \
    do {                                                                                 \
        if (ind) hipLaunchKernelGGL(KT, grid, block, 0, stream, __VA_ARGS__);            \
        else hipLaunchKernelGGL(KF, grid, block, 0, stream, __VA_ARGS__);                \
    } while (0)

int submit_step(mi355x_h264_encoder* e, Step& T)
{
    Slot& S = *T.slot;
    const bool idr = T.idr, ind = T.items != nullptr;
    const int cur = e->cur;
    FrameParams P{};
    P.src = T.d_src; P.src_nv12 = T.nv12 ? 1 : 0; P.w = e->cfg.width; P.h = e->cfg.height;
    P.cw = e->cw; P.ch = e->ch; P.mbw = e->mbw; P.mbh = e->mbh;
    P.nref = ind ? 1 : std::max(1, avail_refs(e, idr));
    for (int p = 0; p < 3; p++) {
        P.rec[p] = ind ? e->d_plane_base[p] : e->d_planes[cur][p];
        for (int r = 0; r < mi355x_h264_encoder::MAX_REFS; r++) P.refs[r][p] = e->d_planes[(cur + e->nbuf - 1 - std::min(r, e->nrefs - 1)) % e->nbuf][p];
        P.ref[p] = P.refs[0][p];
    }
    P.itemtab = T.d_itemtab; P.qtab = e->d_qtab; P.st_ring_y = e->st_ring_y; P.st_ring_c = e->st_ring_c; P.nbuf = e->nbuf;
    P.mb = e->d_mb; P.levels = e->d_levels; P.mvd = e->d_mvd; P.mvq = e->d_mvq; P.aux = e->d_aux; P.me_cost = e->d_me_cost; P.me_total = e->d_me_total; P.pmv = e->d_pmv;
    P.st_src = T.src_item_stride; P.st_y = e->st_y; P.st_c = e->st_c; P.st_mb = e->nmb; P.sl = e->sl;
    P.band.row0 = e->b_row0; P.band.rows = e->b_rows;
    P.mbdiv.inv = e->mbw > 1 ? (unsigned)(0x100000000ull / (unsigned)e->mbw) + 1u : 0u;
    e->pic_serial = e->pic_serial == 0xFFFFFFFFu ? 1u : e->pic_serial + 1u;
    P.anypcm = e->d_anypcm; P.anyintra = e->d_anyintra; P.pic_serial = e->pic_serial;
    const unsigned pic_serial = e->pic_serial;
    const unsigned G = (unsigned)T.n;
    fill_qp(P.qy, P.qc, P.lambda, P.sad_nz, e->qp);   // (indirect launches take these from qtab by the item's own QP)
    P.search = e->cfg.search;
    hipStream_t st = T.st;
    auto next_serial = [&]() { e->serial = e->serial == 0xFFFFFFFFu ? 1 : e->serial + 1; return e->serial; };

    // (the payload buffers of the items were left zeroed by the k_pack of their previous use)

    if (idr) {
        StatScope sc(e, &S, MI355X_H264_K_INTRA, (uint32_t)(e->diag_mode ? e->mbw + e->mbh - 1 : 1), (uint32_t)(e->b_nmb * T.n), st);
        LAUNCH2(ind, k_i4_decide<true>, k_i4_decide<false>, dim3((e->b_nmb + 3) / 4, G), dim3(64), st, P, 0);   // Intra4x4 or Intra16x16, and the block modes: from the source alone
        if (e->diag_mode && !ind) {
            for (int s = 0; s < e->mbw + e->mbh - 1; s++) {
                const int ymin = std::max(0, s - e->mbw + 1), ymax = std::min(e->mbh - 1, s);
                hipLaunchKernelGGL(k_intra_diag, dim3(ymax - ymin + 1, G), dim3(64), 0, st, P, s);
            }
        } else {
            IntraRowParams R{};
            R.p = P; R.handoff = e->d_handoff; R.st_handoff = e->st_handoff; R.err = T.h_err;
            R.serial = next_serial();
            {   // MI355X_H264_INTRA_SLOTS: pictures the row wavefront holds at a time (k_intra_rows)
                static const int slots = getenv("MI355X_H264_INTRA_SLOTS") ? std::max(1, atoi(getenv("MI355X_H264_INTRA_SLOTS"))) : 24;
                R.npic = (int)G;
                LAUNCH2(ind, k_intra_rows<true>, k_intra_rows<false>, dim3(e->b_rows, std::min(G, (unsigned)slots)), dim3(128), st, R);
            }
        }
    } else {
        { const bool turns = g_turns_on && !ind && e->me_turn > 0 && T.n >= TURN_MIN_BATCH;
          if (turns) hipLaunchKernelGGL(k_turn_acquire, dim3(1), dim3(64), 0, st, g_turns.d_lock[e->device], (unsigned)e->me_turn, (int)TURN_TIMEOUT_US);
          { StatScope sc(e, &S, MI355X_H264_K_ME, (uint32_t)P.nref, (uint32_t)(e->b_nmb * T.n), st);
            FrameParams Q = P;   // one launch per reference picture (config.refs): Q.ref = the planes of ref_idx_l0 = Q.rf
            Q.rf_last = P.nref - 1;
            for (int r = 0; r < P.nref; r++) {
                Q.rf = r;
                for (int p = 0; p < 3; p++) Q.ref[p] = P.refs[r][p];
                LAUNCH2(ind, k_me<true>, k_me<false>, dim3(e->b_nmb, G), dim3(64), st, Q);
            } }
          if (turns) hipLaunchKernelGGL(k_turn_release, dim3(1), dim3(1), 0, st, g_turns.d_lock[e->device], (unsigned)e->me_turn);
        }
        { StatScope sc(e, &S, MI355X_H264_K_PMB, 1, (uint32_t)(e->b_nmb * T.n), st);
          if (e->cfg.profile_idc == 100) LAUNCH2(ind, k_tq8<true>, k_tq8<false>, dim3((e->b_nmb + 15) / 16, G), dim3(64), st, P);   // High: 8x8 transform, sixteen macroblocks per wave
          else LAUNCH2(ind, k_tq<true>, k_tq<false>, dim3((e->b_nmb + 7) / 8, G), dim3(64), st, P); }   // one wave per eight macroblocks
        {   // macroblocks the motion search handed to the intra pass (returns at once when there are none)
            IntraRowParams R{};
            R.p = P; R.handoff = e->d_handoff; R.st_handoff = e->st_handoff; R.err = T.h_err;
            R.serial = next_serial();
            // Their grids hold ONE picture at a time (the workgroups walk the step's pictures) while the recent P pictures had next to
            // no intra macroblocks, all of them once they have: see k_pintra_rows.  MI355X_H264_PINTRA_SLOTS fixes the number.
            static const int pslots_env = getenv("MI355X_H264_PINTRA_SLOTS") ? std::max(1, atoi(getenv("MI355X_H264_PINTRA_SLOTS"))) : 0;
            const unsigned pslots = std::min(G, pslots_env ? (unsigned)pslots_env : (e->p_intra_x16 > 16u * PINTRA_SPARSE_MBS ? G : 1u));
            R.npic = (int)G;
            LAUNCH2(ind, k_i4_decide<true>, k_i4_decide<false>, dim3(std::min((e->b_nmb + 3) / 4, (int)I4_MARKED_WAVES), pslots), dim3(64), st, P, (int)G);
            LAUNCH2(ind, (k_pintra_rows<false, true>), (k_pintra_rows<false, false>), dim3(e->b_rows, pslots), dim3(64), st, R);
        }
    }
    // entropy coding: slice headers per position
    HdrBatch H{}, Hpcm{};
    for (int g = 0; g < T.n; g++) {
        uint64_t hdr = 0;
        const int fn = ind ? T.items[g].frame_num : e->frame_num, qp = ind ? T.items[g].qp : e->qp;
        const int id = ind ? T.items[g].idr_id : ((e->idr_id + g * e->idr_step) & 0xFF);
        const int nact = ind ? (idr ? 0 : 1) : avail_refs(e, idr);
        H.len[g] = (unsigned char)build_slice_header(e, idr, id, false, fn, qp, nact, &hdr);
        H.bits[g] = hdr;
        Hpcm.len[g] = (unsigned char)build_slice_header(e, idr, id, true, fn, qp, nact, &hdr);
        Hpcm.bits[g] = hdr;
    }
    // entropy coding needs only levels / MbInfo, the loop filter the reconstruction and the boundary strengths
    // (a small launch of its own on this stream): the two run side by side and the filter never waits for the coder
    hipStream_t ec = T.ec;
    CavlcParams C{};
    C.mb = e->d_mb; C.levels = e->d_levels; C.mvd = e->d_mvd; C.mbw = e->mbw; C.nmb = e->nmb; C.p_slice = idr ? 0 : 1; C.t8x8 = e->cfg.profile_idc == 100 ? 1 : 0;
    C.nref = ind ? (idr ? 0 : 1) : avail_refs(e, idr); C.sl = e->sl;
    C.mb_first = e->b_row0 * e->mbw; C.mb_end = C.mb_first + e->b_nmb;
    C.slice_cap = (unsigned)e->slice_cap;
    C.mbdiv = P.mbdiv;
    C.slotbits = e->d_slotbits; C.slotcode = e->d_slotcode; C.mbbits = e->d_mbbits; C.bitbuf = S.d_bitbuf;
    C.bs = (uint8_t*)e->d_bs; C.prevcoded = e->d_prevcoded;
    C.st_mb = e->nmb; C.st_bitbuf = e->st_bitbuf_bytes / 4;
    C.aux = e->d_aux; C.mvq = e->d_mvq;
    C.src = T.d_src; C.w = e->cfg.width; C.h = e->cfg.height; C.src_nv12 = T.nv12 ? 1 : 0; C.st_src = T.src_item_stride;
    C.itemtab = T.d_itemtab;
    const int cavlc_grid = (e->b_nmb + 1) / 2;
    unsigned db_serial = 0;
    if (!e->cfg.disable_deblock) {   // (the diagonal debug form of the filter reads the strengths too)
        db_serial = next_serial();   // the serial the loop filter of this picture will run under
        LAUNCH2(ind, k_bs<true>, k_bs<false>, dim3(std::min(cavlc_grid, (int)BS_WAVES), G), dim3(64), st, C, e->d_anybs, db_serial);
    }
    const bool fork = ec != st;
    if (fork) {
        HIPCHK(e, hipEventRecord(T.recon_ready, st));
        HIPCHK(e, hipStreamWaitEvent(ec, T.recon_ready, 0));
    }
    {
        StatScope sc(e, &S, MI355X_H264_K_CAVLC, 4, (uint32_t)(e->b_nmb * T.n), ec);
        const int grid = cavlc_grid;
        if (!idr) {
            LAUNCH2(ind, k_mvpred<true>, k_mvpred<false>, dim3((e->b_nmb + 63) / 64, G), dim3(64), ec, P);   // vectors + coded_block_pattern are final: mvd, P_Skip
            LAUNCH2(ind, k_skip_scan<true>, k_skip_scan<false>, dim3(G), dim3(256), ec, C);
        }
        LAUNCH2(ind, (k_cavlc<false, true>), (k_cavlc<false, false>), dim3(grid, G), dim3(64), ec, C);
        LAUNCH2(ind, k_bit_scan<true>, k_bit_scan<false>, dim3(G * (unsigned)e->b_nsl), dim3(SCAN_NT), ec, C, H, Hpcm, (const unsigned*)e->d_anypcm, pic_serial, S.d_info, e->d_me_cost,
                e->b_nsl, e->b_sl0, (unsigned)e->slice_cap);
        LAUNCH2(ind, (k_cavlc<true, true>), (k_cavlc<true, false>), dim3(grid, G), dim3(64), ec, C);
        // access unit layout in the pinned buffer: [pad][SPS PPS (IDR only)][00 00 00 01 hdr][payload...];
        // with several slices: the payload of slice s at s * slice_cap, the access unit is put together by finish_item
        const size_t pre = (idr ? e->sps_pps.size() : 0) + 5;
        const size_t pad = (16 - (pre & 15)) & 15;
        T.au_start = pad;
        T.payload_off = e->nsl > 1 ? 0 : pad + pre;
        T.nal_hdr = idr ? ((3 << 5) | 5) : ((2 << 5) | 1);
        LAUNCH2(ind, k_pack<true>, k_pack<false>, dim3(G * (unsigned)e->b_nsl), dim3(SCAN_NT), ec, (uint8_t*)S.d_bitbuf, e->st_bitbuf_bytes, S.h_au + T.payload_off, e->st_au,
                (const SliceInfo*)S.d_info, S.h_info, e->b_nsl, e->b_sl0, (unsigned)e->slice_cap, T.d_itemtab);
    }
    if (fork) HIPCHK(e, hipEventRecord(T.entropy_done, ec));
    if (e->keep_pre && !ind)
        for (int p = 0; p < 3; p++) {   // (the items' planes lie nbuf ring slots apart: one row of the 2-D copy per item)
            const size_t ring = p ? e->st_ring_c : e->st_ring_y;
            HIPCHK(e, hipMemcpy2DAsync(e->d_pre[p], ring, e->d_planes[cur][p], p ? e->st_c : e->st_y, ring, (size_t)e->G, hipMemcpyDeviceToDevice, st));
        }
    if (!e->cfg.disable_deblock) {
        const int steps = e->mbw + 2 * (e->mbh - 1);
        StatScope sc(e, &S, MI355X_H264_K_DEBLOCK, (uint32_t)(e->diag_mode ? steps : 1), (uint32_t)(e->b_nmb * T.n), st);
        DbParams D{};
        for (int p = 0; p < 3; p++) D.pl[p] = ind ? e->d_plane_base[p] : e->d_planes[cur][p];
        D.mb = e->d_mb; D.cw = e->cw; D.ch = e->ch; D.mbw = e->mbw; D.mbh = e->mbh; D.sl = e->sl; D.bs = (const uint8_t*)e->d_bs;
        const int qp = e->qp, qpc = h_chroma_qp[qp];   // (indirect launches look the item's own QP up on the device)
        D.alpha_y = h_alpha[qp]; D.beta_y = h_beta[qp]; D.alpha_c = h_alpha[qpc]; D.beta_c = h_beta[qpc];
        for (int i = 0; i < 3; i++) { D.tc0_y[i] = h_tc0[qp][i]; D.tc0_c[i] = h_tc0[qpc][i]; }
        if (e->diag_mode && !ind) {
            for (int s = 0; s < steps; s++) {
                const int ymin = std::max(0, (s - (e->mbw - 1) + 1) >> 1), ymax = std::min(e->mbh - 1, s >> 1);
                if (ymax < ymin) continue;
                hipLaunchKernelGGL(k_deblock_diag, dim3(ymax - ymin + 1), dim3(64), 0, st, D, s);
            }
        } else {
            DbRowParams R{};
            R.d = D; R.handoff = e->d_handoff; R.err = T.h_err;
            R.st_y = e->st_y; R.st_c = e->st_c; R.st_handoff = e->st_handoff; R.st_mb = e->nmb;
            R.serial = db_serial; R.row0 = e->b_row0;
            R.bs = e->d_bs; R.anybs = e->d_anybs;
            R.anypcm = e->d_anypcm; R.anyintra = e->d_anyintra; R.pic_serial = pic_serial;
            R.itemtab = T.d_itemtab; R.st_ring_y = e->st_ring_y; R.st_ring_c = e->st_ring_c;
            // two macroblock rows per wave (k_deblock_pairs) for lockstep batches of pictures of one slice; else one row per wave
            const bool pairs = e->pair_filter && T.n >= e->pair_min_batch && e->nsl == 1 && e->b_rows == e->mbh;
            R.npic = (int)G;
            const unsigned grid_x = pairs ? (unsigned)((e->b_rows + 1) / 2) : (unsigned)e->b_rows;
            auto filter = [&](bool bs4, unsigned at_a_time) {
                const dim3 grid(grid_x, std::min(G, at_a_time));
                if (pairs) { if (bs4) LAUNCH2(ind, (k_deblock_pairs<true, true>), (k_deblock_pairs<true, false>), grid, dim3(64), st, R);
                             else LAUNCH2(ind, (k_deblock_pairs<false, true>), (k_deblock_pairs<false, false>), grid, dim3(64), st, R); }
                else { if (bs4) LAUNCH2(ind, (k_deblock_rows<true, false, true>), (k_deblock_rows<true, false, false>), grid, dim3(64), st, R);
                       else LAUNCH2(ind, (k_deblock_rows<false, false, true>), (k_deblock_rows<false, false, false>), grid, dim3(64), st, R); }
            };
            if (idr) { R.need_intra = 0; filter(true, G); }
            else {   // P pictures: the form without the bS 4 filter, or - when the picture has intra macroblocks - the one with it
                // (while the recent P pictures had next to none, the second launch holds one picture at a time: see k_deblock_rows)
                R.need_intra = -1; filter(false, G);
                R.need_intra = 1; filter(true, e->p_intra_x16 > 16u * PINTRA_SPARSE_MBS ? G : 1u);
            }
        }
    }
    if (fork) HIPCHK(e, hipStreamWaitEvent(st, T.entropy_done, 0));   // join: the next picture rewrites MbInfo / levels
    HIPCHK(e, hipEventRecord(T.done, st));
    HIPCHK(e, hipGetLastError());
    return MI355X_H264_OK;
}

// enqueue everything for one picture (every batch item's) whose I420 samples are at d_src: the direct form
int submit(mi355x_h264_encoder* e, const uint8_t* d_src, size_t src_item_stride, int slot_idx, bool nv12)
{
    Slot& S = e->slots[slot_idx];
    const bool idr = e->force_idr || e->frames == 0 || e->frame_in_gop >= e->cfg.gop;
    if (idr) { e->frame_in_gop = 0; e->frame_num = 0; }
    e->force_idr = 0;
    Step T;
    T.d_src = d_src; T.src_item_stride = src_item_stride; T.nv12 = nv12; T.idr = idr; T.n = e->G;
    T.st = e->stream; T.ec = e->stream_ec; T.recon_ready = S.recon_ready; T.entropy_done = S.entropy_done; T.done = S.done; T.h_err = S.h_err;
    T.slot = &S;
    const int rc = submit_step(e, T);
    if (rc) return rc;
    S.au_start = T.au_start; S.payload_off = T.payload_off; S.idr = idr; S.nal_hdr = T.nal_hdr;
    S.busy = true;
    // bookkeeping for the next picture
    e->cur = (e->cur + 1) % e->nbuf;
    if (idr) e->idr_id = (e->idr_id + e->idr_step * e->G) & 0xFF;
    e->frame_num = (e->frame_num + 1) & 255;
    e->frame_in_gop++;
    e->frames++;
    return MI355X_H264_OK;
}

// wait for a slot (all batch items of one lockstep picture); stats are folded in once
int wait_slot(mi355x_h264_encoder* e, int slot_idx)
{
    Slot& S = e->slots[slot_idx];
    if (!S.busy) return fail(e, MI355X_H264_E_INTERNAL, "collect on an idle slot");
    HIPCHK(e, hipEventSynchronize(S.done));
    S.busy = false;
    for (auto& ev : S.evs) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, ev.a, ev.b) == hipSuccess) {
            e->stats.ms[ev.k] += ms; e->stats.launches[ev.k] += ev.launches; e->stats.mbs[ev.k] += ev.mbs;
        }
        e->ev_pool.push_back(ev.a); e->ev_pool.push_back(ev.b);
    }
    S.evs.clear();
    e->stats.frames += (uint64_t)e->G;
    if (*S.h_err) {
        // the picture's reconstruction is not to be trusted: it must not become a reference, and the flag is per report
        const unsigned flag = *S.h_err;
        *S.h_err = 0;
        e->force_idr = 1;
        return fail(e, MI355X_H264_E_INTERNAL, "wavefront kernel hand-off timed out (flag %u)", flag);
    }
    return MI355X_H264_OK;
}

// finish the access unit of batch item g on the host: S = the buffers it was written to, L = where and of which type
struct AuLayout { size_t au_start, payload_off; bool idr; int nal_hdr; };
int finish_item(mi355x_h264_encoder* e, Slot& S, const AuLayout& L, int g, uint8_t** out, uint32_t* out_len, int* frame_type);
int finish_item(mi355x_h264_encoder* e, int slot_idx, int g, uint8_t** out, uint32_t* out_len, int* frame_type)
{
    Slot& S = e->slots[slot_idx];
    return finish_item(e, S, AuLayout{S.au_start, S.payload_off, S.idr, S.nal_hdr}, g, out, out_len, frame_type);
}
int finish_item(mi355x_h264_encoder* e, Slot& S, const AuLayout& L, int g, uint8_t** out, uint32_t* out_len, int* frame_type)
{
    uint8_t* base = S.h_au + (size_t)g * e->st_au;
    if (e->nsl > 1) {
        // several slices: one NAL unit each, put together here (the payloads lie slice_cap apart in the pinned buffer)
        std::vector<uint8_t>& eb = e->esc_buf[g];
        size_t need = e->sps_pps.size() + 16;
        uint32_t cost = 0, p_intra = 0;
        for (int sl = 0; sl < e->b_nsl; sl++) {
            const SliceInfo& si = S.h_info[(size_t)g * e->b_nsl + sl];
            if (si.error) {
                e->force_idr = 1;   // the refused picture is missing from the stream: the next one must not refer to it
                return fail(e, si.error == 1 ? MI355X_H264_E_OVERFLOW : MI355X_H264_E_INTERNAL, "device reported error %u (slice %d)", si.error, sl);
            }
            need += 5 + (size_t)si.total_bytes * 3 / 2 + 16;
            cost += si.me_cost;
            if (!L.idr) { e->stats.me_searched_mbs += si.searched; e->stats.tq_coded_mbs += si.tq_coded; p_intra += si.searched - si.tq_coded; }
        }
        e->last_me_cost[g] = cost;
        if (!L.idr) { e->stats.p_mbs += (uint64_t)e->b_nmb; e->p_intra_x16 = (3 * e->p_intra_x16 + 16 * p_intra) / 4; }
        eb.resize(need);
        size_t pos = 0;
        if (L.idr && e->b_sl0 == 0) { memcpy(eb.data(), e->sps_pps.data(), e->sps_pps.size()); pos = e->sps_pps.size(); }   // parameter sets go with the first band
        for (int sl = 0; sl < e->b_nsl; sl++) {
            const SliceInfo& si = S.h_info[(size_t)g * e->b_nsl + sl];
            const uint8_t* pay = base + (size_t)(e->b_sl0 + sl) * e->slice_cap;
            uint8_t* o = eb.data() + pos;
            o[0] = 0; o[1] = 0; o[2] = 0; o[3] = 1; o[4] = (uint8_t)L.nal_hdr;
            pos += 5;
            if (si.epb_count == 0) { memcpy(eb.data() + pos, pay, si.total_bytes); pos += si.total_bytes; }
            else pos += nal_escape(pay, si.total_bytes, eb.data() + pos);
        }
        *out = eb.data();
        *out_len = (uint32_t)pos;
        if (frame_type) *frame_type = L.idr ? MI355X_H264_FRAME_IDR : MI355X_H264_FRAME_P;
        return MI355X_H264_OK;
    }
    const SliceInfo info = S.h_info[g];
    e->last_me_cost[g] = info.me_cost;
    if (!L.idr) {
        e->stats.p_mbs += (uint64_t)e->b_nmb; e->stats.me_searched_mbs += info.searched; e->stats.tq_coded_mbs += info.tq_coded;
        e->p_intra_x16 = (3 * e->p_intra_x16 + 16 * (info.searched - info.tq_coded)) / 4;
    }
    if (info.error) {
        e->force_idr = 1;   // the refused picture is missing from the stream: the next one must not refer to it
        return fail(e, info.error == 1 ? MI355X_H264_E_OVERFLOW : MI355X_H264_E_INTERNAL, "device reported error %u", info.error);
    }
    uint8_t* au = base + L.au_start;
    size_t pos = 0;
    if (L.idr) { memcpy(au, e->sps_pps.data(), e->sps_pps.size()); pos = e->sps_pps.size(); }
    au[pos++] = 0; au[pos++] = 0; au[pos++] = 0; au[pos++] = 1; au[pos++] = (uint8_t)L.nal_hdr;
    if (info.epb_count == 0) {
        *out = au;
        *out_len = (uint32_t)(pos + info.total_bytes);
    } else {  // rare: some 00 00 0x pattern needs an emulation prevention byte
        std::vector<uint8_t>& eb = e->esc_buf[g];
        eb.resize(pos + (size_t)info.total_bytes * 3 / 2 + 16);
        memcpy(eb.data(), au, pos);
        const size_t n = nal_escape(base + L.payload_off, info.total_bytes, eb.data() + pos);
        *out = eb.data();
        *out_len = (uint32_t)(pos + n);
    }
    if (frame_type) *frame_type = L.idr ? MI355X_H264_FRAME_IDR : MI355X_H264_FRAME_P;
    return MI355X_H264_OK;
}

int collect(mi355x_h264_encoder* e, int slot_idx, uint8_t** out, uint32_t* out_len, int* frame_type)
{
    int rc = wait_slot(e, slot_idx);
    if (rc) return rc;
    return finish_item(e, slot_idx, 0, out, out_len, frame_type);
}

int pick_level(int mbs, int fps)
{
    for (const auto& l : h_levels)
        if ((uint32_t)mbs <= l.fs && (uint32_t)(mbs * fps) <= l.mbps) return l.idc;
    return 52;
}

}  // namespace

extern "C" {

int mi355x_h264_abi_version(void) { return MI355X_H264_ABI_VERSION; }

void mi355x_h264_default_config(mi355x_h264_config* c)
{
    if (!c) return;
    memset(c, 0, sizeof(*c));
    c->struct_size = sizeof(*c);
    c->width = 720; c->height = 1280;  // reference defaults, VideoEncoderOpenH264.h:13-24
    c->fps = 30; c->bitrate = 5000000; c->gop = 30; c->profile_idc = 66;
    c->rc_mode = MI355X_H264_RC_FIXED_QP; c->qp = 26; c->device = 0; c->disable_deblock = 0;
    c->search = MI355X_H264_SEARCH_SEEDED;
}

// hub_engine: the engine of a stream hub (below) - one set of output buffers instead of NSLOT, never more than one HIP stream
// pair of its own (the hub's step contexts bring theirs)
static int create_engine(const mi355x_h264_config* cfg, mi355x_h264_encoder** out, bool hub_engine);
int mi355x_h264_create(const mi355x_h264_config* cfg, mi355x_h264_encoder** out) { return create_engine(cfg, out, false); }
static int create_engine(const mi355x_h264_config* cfg, mi355x_h264_encoder** out, bool hub_engine)
{
    if (!cfg || !out || cfg->struct_size != sizeof(mi355x_h264_config)) return MI355X_H264_E_ARG;
    {   // HIP maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4): an engine has two streams, a stream hub seven, and
        // a host process runs several.  Ask for more before the runtime comes up - unless the host has chosen (measured: 64
        // plugin streams 7.4 k fps on 4 queues, 10.4 k on 32).  Has no effect once another HIP user has initialised the runtime.
        static std::once_flag once;
        std::call_once(once, [] { setenv("GPU_MAX_HW_QUEUES", "16", 0); });
    }
    *out = nullptr;
    if (cfg->width < 16 || cfg->height < 16 || cfg->width > 4096 || cfg->height > 4096 || ((cfg->width | cfg->height) & 1))
        return MI355X_H264_E_ARG;
    if (cfg->qp < 10 || cfg->qp > 51 || cfg->gop < 1) return MI355X_H264_E_ARG;
    if (cfg->profile_idc != 66 && cfg->profile_idc != 77 && cfg->profile_idc != 100) return MI355X_H264_E_ARG;
    if (cfg->input_format != MI355X_H264_INPUT_I420 && cfg->input_format != MI355X_H264_INPUT_NV12) return MI355X_H264_E_ARG;
    if (cfg->slices < 0 || cfg->slices > 64) return MI355X_H264_E_ARG;
    if (cfg->refs < 0 || cfg->refs > mi355x_h264_encoder::MAX_REFS) return MI355X_H264_E_ARG;
    if (cfg->search != MI355X_H264_SEARCH_EXHAUSTIVE && cfg->search != MI355X_H264_SEARCH_SEEDED) return MI355X_H264_E_ARG;
    if (cfg->band_count < 0 || cfg->band_index < 0 || (cfg->band_count > 1 && (cfg->band_index >= cfg->band_count || cfg->batch > 1))) return MI355X_H264_E_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || cfg->device < 0 || cfg->device >= ndev) return MI355X_H264_E_NODEVICE;
    mi355x_h264_encoder* e = new (std::nothrow) mi355x_h264_encoder();
    if (!e) return MI355X_H264_E_NOMEM;
    e->cfg = *cfg;
    e->device = cfg->device;
    e->qp = cfg->qp;
    e->mbw = (cfg->width + 15) / 16; e->mbh = (cfg->height + 15) / 16;
    e->cw = e->mbw * 16; e->ch = e->mbh * 16; e->nmb = e->mbw * e->mbh;
    e->level_idc = std::max(32, pick_level(e->nmb, cfg->fps > 0 ? cfg->fps : 30));
    {   // slices: bands of ceil(rows / slices) macroblock rows, at least two rows each
        const int n = std::min(std::max(cfg->slices, 1), std::max(1, e->mbh / 2));
        e->sl.rows = (e->mbh + n - 1) / n;
        e->sl.inv = e->sl.rows > 1 ? (unsigned)(0x100000000ull / (unsigned)e->sl.rows) + 1u : 0u;   // (one row: my is always 0)
        e->nsl = (e->mbh + e->sl.rows - 1) / e->sl.rows;
    }
    e->b_sl0 = 0; e->b_nsl = e->nsl;
    if (cfg->band_count > 1) {   // this instance codes its share of the slices; the others belong to the neighbours
        if (cfg->band_count > e->nsl) { delete e; return MI355X_H264_E_ARG; }
        e->b_sl0 = (int)((long)cfg->band_index * e->nsl / cfg->band_count);
        e->b_nsl = (int)((long)(cfg->band_index + 1) * e->nsl / cfg->band_count) - e->b_sl0;
    }
    e->b_row0 = e->b_sl0 * e->sl.rows;
    e->b_rows = std::min(e->mbh, (e->b_sl0 + e->b_nsl) * e->sl.rows) - e->b_row0;
    e->b_nmb = e->b_rows * e->mbw;
    e->nrefs = cfg->refs > 1 ? cfg->refs : 1;
    e->nbuf = e->nrefs + 1;
    e->G = cfg->batch > 1 ? cfg->batch : 1;
    if (e->G > MAX_BATCH) { delete e; return MI355X_H264_E_ARG; }
    e->esc_buf.resize((size_t)e->G);
    build_parameter_sets(e);
#define CK(call)                                                              \
    do {                                                                      \
        hipError_t _r = (call);                                               \
        if (_r != hipSuccess) {                                               \
            fprintf(stderr, "mi355x_h264_create: %s: %s\n", #call, hipGetErrorString(_r)); \
            mi355x_h264_destroy(e);                                           \
            return _r == hipErrorOutOfMemory ? MI355X_H264_E_NOMEM : MI355X_H264_E_HIP; \
        }                                                                     \
    } while (0)
    CK(hipSetDevice(e->device));
    CK(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
    // Entropy coding normally runs on a stream of its own beside the loop filter (shorter picture latency).  A process that
    // holds many engines (the plugin surface with many streams: one engine per VideoEncoder object) would then ask for more
    // hardware queues than the device has, and the runtime's multiplexing costs more than the overlap gains (16 plugin streams: 3.8 k -> 6.0 k fps; 4 streams: p99 8 -> 4 ms): from the third
    // live engine on (or with MI355X_H264_ONE_STREAM=1) an engine uses its one stream for everything.
    {
        static std::atomic<int> live{0};
        const char* one = getenv("MI355X_H264_ONE_STREAM");
        const int n = live.fetch_add(1) + 1;
        e->counted_live = &live;
        if ((one && one[0] == '1') || (n > 2 && !(one && one[0] == '0'))) e->stream_ec = e->stream;
        else CK(hipStreamCreateWithFlags(&e->stream_ec, hipStreamNonBlocking));
    }
    const size_t ysz = (size_t)e->cw * e->ch;
    const size_t Gn = (size_t)e->G;
    e->st_ring_y = ysz + 256; e->st_ring_c = ysz / 4 + 256;
    e->st_y = e->st_ring_y * e->nbuf; e->st_c = e->st_ring_c * e->nbuf;
    for (int p = 0; p < 3; p++) {
        CK(hipMalloc((void**)&e->d_plane_base[p], (p ? e->st_c : e->st_y) * Gn));
        CK(hipMemset(e->d_plane_base[p], 0, (p ? e->st_c : e->st_y) * Gn));
        for (int b = 0; b < e->nbuf; b++) e->d_planes[b][p] = e->d_plane_base[p] + (size_t)b * (p ? e->st_ring_c : e->st_ring_y);
    }
    for (int p = 0; p < 3; p++) CK(hipMalloc((void**)&e->d_pre[p], (p ? e->st_ring_c : e->st_ring_y) * Gn));
    {
        std::vector<QpEntry> qt(52);
        for (int q = 0; q < 52; q++) fill_qp(qt[q].qy, qt[q].qc, qt[q].lambda, qt[q].sad_nz, q);
        CK(hipMalloc((void**)&e->d_qtab, 52 * sizeof(QpEntry)));
        CK(hipMemcpy(e->d_qtab, qt.data(), 52 * sizeof(QpEntry), hipMemcpyHostToDevice));
    }
    CK(hipMalloc((void**)&e->d_mb, Gn * e->nmb * sizeof(MbInfo)));
    CK(hipMemset(e->d_mb, 0, Gn * e->nmb * sizeof(MbInfo)));
    CK(hipMalloc((void**)&e->d_levels, Gn * e->nmb * LV_STRIDE * sizeof(int16_t)));
    CK(hipMalloc((void**)&e->d_mvd, Gn * e->nmb * 8 * sizeof(int16_t)));
    CK(hipMalloc((void**)&e->d_mvq, Gn * e->nmb * 8 * sizeof(int16_t)));
    CK(hipMemset(e->d_mvq, 0, Gn * e->nmb * 8 * sizeof(int16_t)));
    CK(hipMalloc((void**)&e->d_me_total, Gn * e->nmb * sizeof(uint32_t)));
    CK(hipMalloc((void**)&e->d_pmv, Gn * e->nmb * sizeof(int)));
    CK(hipMalloc((void**)&e->d_aux, Gn * e->nmb * 16));
    CK(hipMemset(e->d_aux, 0, Gn * e->nmb * 16));
    CK(hipMalloc((void**)&e->d_slotbits, Gn * e->nmb * 32 * sizeof(uint16_t)));
    CK(hipMalloc((void**)&e->d_slotcode, Gn * e->nmb * 32 * sizeof(unsigned long long)));
    CK(hipMalloc((void**)&e->d_mbbits, Gn * e->nmb * sizeof(uint32_t)));
    CK(hipMalloc((void**)&e->d_anybs, Gn * sizeof(unsigned)));
    CK(hipMemset(e->d_anybs, 0, Gn * sizeof(unsigned)));
    CK(hipMalloc((void**)&e->d_anypcm, Gn * sizeof(unsigned)));
    CK(hipMemset(e->d_anypcm, 0, Gn * sizeof(unsigned)));
    CK(hipMalloc((void**)&e->d_anyintra, Gn * sizeof(unsigned)));
    CK(hipMemset(e->d_anyintra, 0, Gn * sizeof(unsigned)));
    CK(hipMalloc((void**)&e->d_prevcoded, Gn * (e->nmb + 1) * sizeof(int32_t)));
    e->st_handoff = (size_t)e->nmb * 24;
    CK(hipMalloc((void**)&e->d_handoff, Gn * e->st_handoff * sizeof(unsigned long long)));
    CK(hipMemset(e->d_handoff, 0, Gn * e->st_handoff * sizeof(unsigned long long)));
    CK(hipMalloc((void**)&e->d_bs, Gn * e->nmb * 32));
    CK(hipMalloc((void**)&e->d_me_cost, Gn * e->nmb * sizeof(uint16_t)));
    CK(hipMemset(e->d_me_cost, 0, Gn * e->nmb * sizeof(uint16_t)));
    e->last_me_cost.assign(Gn, 0);
    e->diag_mode = getenv("MI355X_H264_DIAG") != nullptr && e->G == 1 && e->b_nsl == e->nsl;
    {
        // The loop filter takes two macroblock rows per wave (k_deblock_pairs) from a lockstep batch of 8 pictures on (pictures of one
        // slice): measured on the bench workload with the filter's edge skip in place, same box, row form / pair form: batch 4
        // 8 837 / 8 851 fps, batch 8 15.0 / 15.3 k, batch 16 21.2 / 22.1 k, batch 32 24.0 / 25.0 k; one GOP in flight 1 230 / 1 209 fps -
        // so small batches and the latency mode keep one row per wave.  MI355X_H264_PAIR_FILTER=N moves the threshold, 0 turns the
        // pair form off.
        const char* pf = getenv("MI355X_H264_PAIR_FILTER");
        e->pair_filter = !(pf && pf[0] == '0');
        e->pair_min_batch = (pf && pf[0] >= '1' && pf[0] <= '9') ? atoi(pf) : 8;
    }
    e->frame_bytes = (size_t)cfg->width * cfg->height * 3 / 2;
    CK(hipMalloc((void**)&e->d_stage, e->frame_bytes + 256));
    CK(hipHostMalloc((void**)&e->h_stage, e->frame_bytes + 256, hipHostMallocDefault));
    e->bitbuf_cap = ysz * 2 + (1 << 16);
    e->slice_cap = e->bitbuf_cap;
    if (e->nsl > 1) {   // every slice gets room for twice its luma bytes (CAVLC's worst case is about 1.6 times)
        e->slice_cap = ((size_t)e->sl.rows * 256 * e->mbw * 2 + 4096 + 15) & ~(size_t)15;
        e->bitbuf_cap = e->slice_cap * e->nsl;
    }
    e->st_bitbuf_bytes = (e->bitbuf_cap + 256 + 255) & ~(size_t)255;
    e->au_cap = e->bitbuf_cap + e->sps_pps.size() + 64;
    e->st_au = (e->au_cap + 256 + 255) & ~(size_t)255;
    e->nslots = hub_engine ? 1 : NSLOT;
    if (!hub_engine && e->G >= TURN_MIN_BATCH && e->device >= 0 && e->device < TURN_DEVICES) {   // takes part in the turn-taking of the motion searches
        std::lock_guard<std::mutex> tl(g_turns.mu);
        if (!g_turns.d_lock[e->device]) {
            CK(hipMalloc((void**)&g_turns.d_lock[e->device], sizeof(unsigned)));
            CK(hipMemset(g_turns.d_lock[e->device], 0, sizeof(unsigned)));
        }
        e->me_turn = (int)g_turns.next_id++;
    }
    for (int si = 0; si < e->nslots; si++) {
        Slot& S = e->slots[si];
        CK(hipMalloc((void**)&S.d_bitbuf, e->st_bitbuf_bytes * Gn));
        CK(hipMemset(S.d_bitbuf, 0, e->st_bitbuf_bytes * Gn));
        CK(hipMalloc((void**)&S.d_info, sizeof(SliceInfo) * Gn * e->nsl));
        CK(hipHostMalloc((void**)&S.h_info, sizeof(SliceInfo) * Gn * e->nsl, hipHostMallocDefault));
        CK(hipHostMalloc((void**)&S.h_err, sizeof(unsigned), hipHostMallocDefault));
        *S.h_err = 0;
        CK(hipHostMalloc((void**)&S.h_au, e->st_au * Gn, hipHostMallocDefault));
        CK(hipEventCreateWithFlags(&S.done, hipEventDisableTiming));
        CK(hipEventCreateWithFlags(&S.recon_ready, hipEventDisableTiming));
        CK(hipEventCreateWithFlags(&S.entropy_done, hipEventDisableTiming));
    }
    CK(hipDeviceSynchronize());
#undef CK
    *out = e;
    return MI355X_H264_OK;
}

void mi355x_h264_destroy(mi355x_h264_encoder* e)
{
    if (!e) return;
    (void)hipSetDevice(e->device);
    if (e->stream) (void)hipStreamSynchronize(e->stream);
    for (int p = 0; p < 3; p++) { (void)hipFree(e->d_plane_base[p]); (void)hipFree(e->d_pre[p]); }
    (void)hipFree(e->d_qtab);
    (void)hipFree(e->d_mb); (void)hipFree(e->d_levels); (void)hipFree(e->d_mvd); (void)hipFree(e->d_mvq); (void)hipFree(e->d_aux); (void)hipFree(e->d_me_total); (void)hipFree(e->d_pmv);
    (void)hipFree(e->d_slotbits); (void)hipFree(e->d_slotcode); (void)hipFree(e->d_mbbits); (void)hipFree(e->d_prevcoded); (void)hipFree(e->d_anybs); (void)hipFree(e->d_anypcm); (void)hipFree(e->d_anyintra); (void)hipFree(e->d_stage);
    (void)hipFree(e->d_handoff); (void)hipFree(e->d_bs); (void)hipFree(e->d_me_cost);
    if (e->h_stage) (void)hipHostFree(e->h_stage);
    (void)hipFree(e->d_rgba);
    if (e->h_rgba) (void)hipHostFree(e->h_rgba);
    for (auto& S : e->slots) {
        (void)hipFree(S.d_bitbuf); (void)hipFree(S.d_info);
        if (S.h_info) (void)hipHostFree(S.h_info);
        if (S.h_err) (void)hipHostFree(S.h_err);
        if (S.h_au) (void)hipHostFree(S.h_au);
        if (S.done) (void)hipEventDestroy(S.done);
        if (S.recon_ready) (void)hipEventDestroy(S.recon_ready);
        if (S.entropy_done) (void)hipEventDestroy(S.entropy_done);
        for (auto& ev : S.evs) { (void)hipEventDestroy(ev.a); (void)hipEventDestroy(ev.b); }
    }
    for (auto ev : e->ev_pool) (void)hipEventDestroy(ev);
    if (e->stream_ec && e->stream_ec != e->stream) { (void)hipStreamSynchronize(e->stream_ec); (void)hipStreamDestroy(e->stream_ec); }
    if (e->counted_live) e->counted_live->fetch_sub(1);
    if (e->stream) (void)hipStreamDestroy(e->stream);
    delete e;
}

// one picture already in device memory, in the given layout
static int encode_one_device(mi355x_h264_encoder* e, const void* d_pic, bool nv12, uint8_t** out, uint32_t* out_len, int* frame_type)
{
    if (!e || !d_pic || !out || !out_len) return fail(e, MI355X_H264_E_ARG, "null argument");
    HIPCHK(e, hipSetDevice(e->device));
    const int slot = e->next_slot;
    e->next_slot = (e->next_slot + 1) % NSLOT;
    if (e->G != 1) return fail(e, MI355X_H264_E_ARG, "single-picture calls need a batch-1 encoder");
    int rc = submit(e, (const uint8_t*)d_pic, 0, slot, nv12);
    if (rc) return rc;
    return collect(e, slot, out, out_len, frame_type);
}

int mi355x_h264_encode_device(mi355x_h264_encoder* e, const void* d_pic, uint8_t** out, uint32_t* out_len, int* frame_type)
{
    return encode_one_device(e, d_pic, e && e->cfg.input_format == MI355X_H264_INPUT_NV12, out, out_len, frame_type);
}

int mi355x_h264_encode(mi355x_h264_encoder* e, const uint8_t* y, int ys, const uint8_t* u, int us, const uint8_t* v, int vs,
                       uint8_t** out, uint32_t* out_len, int* frame_type)
{
    if (!e || !y || !u || !v || !out || !out_len) return fail(e, MI355X_H264_E_ARG, "null argument");
    const int w = e->cfg.width, h = e->cfg.height;
    if (ys < w || us < w / 2 || vs < w / 2) return fail(e, MI355X_H264_E_ARG, "stride smaller than width");
    HIPCHK(e, hipSetDevice(e->device));
    // the previous picture's use of the staging buffers has completed (encode is synchronous)
    const size_t ysz = (size_t)w * h;
    if (ys == w && us == w / 2 && vs == w / 2 && u == y + ysz && v == u + ysz / 4) {
        // the reference's own layout (InitSrcPic, ref :354-365: one tight I420 buffer): no per-row work.  The picture goes to
        // pinned memory and on to the device in four pieces, the copy of piece k+1 overlapping the transfer of piece k
        const size_t n = e->frame_bytes, piece = ((n / 4) + 255) & ~(size_t)255;
        for (size_t o = 0; o < n; o += piece) {
            const size_t len = std::min(piece, n - o);
            memcpy(e->h_stage + o, y + o, len);
            HIPCHK(e, hipMemcpyAsync(e->d_stage + o, e->h_stage + o, len, hipMemcpyHostToDevice, e->stream));
        }
    } else {
        uint8_t* d = e->h_stage;
        for (int r = 0; r < h; r++) memcpy(d + (size_t)r * w, y + (size_t)r * ys, (size_t)w);
        d += ysz;
        for (int r = 0; r < h / 2; r++) memcpy(d + (size_t)r * (w / 2), u + (size_t)r * us, (size_t)(w / 2));
        d += ysz / 4;
        for (int r = 0; r < h / 2; r++) memcpy(d + (size_t)r * (w / 2), v + (size_t)r * vs, (size_t)(w / 2));
        HIPCHK(e, hipMemcpyAsync(e->d_stage, e->h_stage, e->frame_bytes, hipMemcpyHostToDevice, e->stream));
    }
    return encode_one_device(e, e->d_stage, false, out, out_len, frame_type);
}

int mi355x_h264_encode_nv12_device(mi355x_h264_encoder* e, const void* d_nv12, uint8_t** out, uint32_t* out_len, int* frame_type)
{
    return encode_one_device(e, d_nv12, true, out, out_len, frame_type);   // the kernels read the interleaved chroma themselves
}

int mi355x_h264_encode_nv12(mi355x_h264_encoder* e, const uint8_t* y, int ys, const uint8_t* uv, int uvs, uint8_t** out,
                            uint32_t* out_len, int* frame_type)
{
    if (!e || !y || !uv || !out || !out_len) return fail(e, MI355X_H264_E_ARG, "null argument");
    const int w = e->cfg.width, h = e->cfg.height;
    if (ys < w || uvs < w) return fail(e, MI355X_H264_E_ARG, "stride smaller than width");
    HIPCHK(e, hipSetDevice(e->device));
    uint8_t* d = e->h_stage;
    for (int r = 0; r < h; r++) memcpy(d + (size_t)r * w, y + (size_t)r * ys, (size_t)w);
    d += (size_t)w * h;
    for (int r = 0; r < h / 2; r++) memcpy(d + (size_t)r * w, uv + (size_t)r * uvs, (size_t)w);
    HIPCHK(e, hipMemcpyAsync(e->d_stage, e->h_stage, e->frame_bytes, hipMemcpyHostToDevice, e->stream));
    return encode_one_device(e, e->d_stage, true, out, out_len, frame_type);
}

// RGBA ingest: one conversion pass into the I420 staging picture (include/mi355x_h264.h states the arithmetic;
// oracle/h264_rgba.c is its CPU restatement).  Thread = one 2x2 block: two 8-byte loads, two 2-byte luma stores, one Cb, one Cr.
__global__ __launch_bounds__(256) void k_rgba_to_i420(const uint8_t* __restrict__ rgba, size_t stride, uint8_t* __restrict__ i420, int w, int h)
{
    const int bx = blockIdx.x * blockDim.x + threadIdx.x, by = blockIdx.y;
    if (bx >= w / 2) return;
    uint8_t* const Y = i420;
    uint8_t* const U = i420 + (size_t)w * h;
    uint8_t* const V = U + (size_t)(w / 2) * (h / 2);
    int sr = 0, sg = 0, sb = 0;
#pragma unroll
    for (int r = 0; r < 2; r++) {
        const uint2 p = *(const uint2*)(rgba + (size_t)(2 * by + r) * stride + 8 * (size_t)bx);   // (rows start on 8 bytes: stride % 8 == 0 checked by the host)
        const int r0 = p.x & 255, g0 = (p.x >> 8) & 255, b0 = (p.x >> 16) & 255;
        const int r1 = p.y & 255, g1 = (p.y >> 8) & 255, b1 = (p.y >> 16) & 255;
        const int y0 = ((66 * r0 + 129 * g0 + 25 * b0 + 128) >> 8) + 16, y1 = ((66 * r1 + 129 * g1 + 25 * b1 + 128) >> 8) + 16;
        *(uint16_t*)(Y + (size_t)(2 * by + r) * w + 2 * bx) = (uint16_t)(y0 | (y1 << 8));
        sr += r0 + r1; sg += g0 + g1; sb += b0 + b1;
    }
    const int r = (sr + 2) >> 2, g = (sg + 2) >> 2, b = (sb + 2) >> 2;
    U[(size_t)by * (w / 2) + bx] = (uint8_t)(((-38 * r - 74 * g + 112 * b + 128) >> 8) + 128);
    V[(size_t)by * (w / 2) + bx] = (uint8_t)(((112 * r - 94 * g - 18 * b + 128) >> 8) + 128);
}

static int encode_rgba_from_device(mi355x_h264_encoder* e, const uint8_t* d_rgba, size_t stride, uint8_t** out, uint32_t* out_len, int* frame_type)
{
    const int w = e->cfg.width, h = e->cfg.height;
    hipLaunchKernelGGL(k_rgba_to_i420, dim3((unsigned)((w / 2 + 255) / 256), (unsigned)(h / 2)), dim3(256), 0, e->stream, d_rgba, stride, e->d_stage, w, h);
    HIPCHK(e, hipGetLastError());
    return encode_one_device(e, e->d_stage, false, out, out_len, frame_type);
}

int mi355x_h264_encode_rgba_device(mi355x_h264_encoder* e, const void* d_rgba, uint8_t** out, uint32_t* out_len, int* frame_type)
{
    if (!e || !d_rgba || !out || !out_len) return fail(e, MI355X_H264_E_ARG, "null argument");
    if (e->G != 1) return fail(e, MI355X_H264_E_ARG, "single-picture calls need a batch-1 encoder");
    if (((uintptr_t)d_rgba & 7) != 0) return fail(e, MI355X_H264_E_ARG, "RGBA picture not aligned to 8 bytes");
    HIPCHK(e, hipSetDevice(e->device));
    return encode_rgba_from_device(e, (const uint8_t*)d_rgba, (size_t)e->cfg.width * 4, out, out_len, frame_type);
}

int mi355x_h264_encode_rgba(mi355x_h264_encoder* e, const uint8_t* rgba, int stride, uint8_t** out, uint32_t* out_len, int* frame_type)
{
    if (!e || !rgba || !out || !out_len) return fail(e, MI355X_H264_E_ARG, "null argument");
    if (e->G != 1) return fail(e, MI355X_H264_E_ARG, "single-picture calls need a batch-1 encoder");
    const int w = e->cfg.width, h = e->cfg.height;
    if (stride < 4 * w) return fail(e, MI355X_H264_E_ARG, "stride smaller than 4 * width");
    HIPCHK(e, hipSetDevice(e->device));
    const size_t row = (size_t)w * 4, n = row * h;
    if (!e->d_rgba) {   // staging for RGBA pictures: allocated with the first one
        HIPCHK(e, hipMalloc((void**)&e->d_rgba, n + 256));
        HIPCHK(e, hipHostMalloc((void**)&e->h_rgba, n + 256, hipHostMallocDefault));
    }
    // (the previous picture's use of the staging buffers has completed: encode is synchronous)
    if ((size_t)stride == row) memcpy(e->h_rgba, rgba, n);
    else for (int r = 0; r < h; r++) memcpy(e->h_rgba + (size_t)r * row, rgba + (size_t)r * stride, row);
    HIPCHK(e, hipMemcpyAsync(e->d_rgba, e->h_rgba, n, hipMemcpyHostToDevice, e->stream));
    return encode_rgba_from_device(e, e->d_rgba, row, out, out_len, frame_type);
}

int mi355x_h264_encode_batch_device(mi355x_h264_encoder* e, const void* d_frames, size_t stride, int count, uint8_t* host_out,
                                    size_t out_cap, uint32_t* sizes, size_t* total_len)
{
    if (!e || !d_frames || !host_out || !sizes || count < 0) return fail(e, MI355X_H264_E_ARG, "null argument");
    if (e->G != 1) return fail(e, MI355X_H264_E_ARG, "use mi355x_h264_encode_gops_device with a batched encoder");
    HIPCHK(e, hipSetDevice(e->device));
    size_t pos = 0;
    int pending[NSLOT], npend = 0, head = 0;
    auto drain_one = [&]() -> int {
        uint8_t* p = nullptr; uint32_t n = 0;
        const int slot = pending[head % NSLOT];
        int rc = collect(e, slot, &p, &n, nullptr);
        if (rc) return rc;
        const int idx = head;
        head++; npend--;
        if (pos + n > out_cap) return fail(e, MI355X_H264_E_OVERFLOW, "batch output buffer too small");
        memcpy(host_out + pos, p, n);
        sizes[idx] = n;
        pos += n;
        return 0;
    };
    // on an error the pictures still in flight are waited for (and dropped) before returning: no slot stays busy
    auto bail = [&](int rc) -> int {
        char first[sizeof(e->err)];
        memcpy(first, e->err, sizeof(first));
        for (; npend > 0; head++, npend--) (void)wait_slot(e, pending[head % NSLOT]);
        memcpy(e->err, first, sizeof(first));
        return rc;
    };
    for (int i = 0; i < count; i++) {
        if (npend == NSLOT - 1) { int rc = drain_one(); if (rc) return bail(rc); }
        const int slot = e->next_slot;
        e->next_slot = (e->next_slot + 1) % NSLOT;
        pending[i % NSLOT] = slot;
        int rc = submit(e, (const uint8_t*)d_frames + (size_t)i * stride, 0, slot, e->cfg.input_format == MI355X_H264_INPUT_NV12);
        if (rc) return bail(rc);
        npend++;
    }
    while (npend) { int rc = drain_one(); if (rc) return bail(rc); }
    if (total_len) *total_len = pos;
    return MI355X_H264_OK;
}

int mi355x_h264_encode_gops_device(mi355x_h264_encoder* e, const void* d_frames, size_t frame_stride, size_t gop_stride, int frames_per_gop,
                                   uint8_t* host_out, size_t out_cap_per_gop, uint32_t* sizes, size_t* gop_bytes)
{
    if (!e || !d_frames || !host_out || !sizes || !gop_bytes || frames_per_gop < 1) return fail(e, MI355X_H264_E_ARG, "null argument");
    HIPCHK(e, hipSetDevice(e->device));
    const int G = e->G;
    for (int g = 0; g < G; g++) gop_bytes[g] = 0;
    e->force_idr = 1;                     // every call starts closed GOPs
    int pending[NSLOT], npend = 0, head = 0;
    auto drain_one = [&]() -> int {
        const int slot = pending[head % NSLOT];
        int rc = wait_slot(e, slot);
        if (rc) return rc;
        for (int g = 0; g < G; g++) {
            uint8_t* p = nullptr; uint32_t n = 0;
            rc = finish_item(e, slot, g, &p, &n, nullptr);
            if (rc) return rc;
            if (gop_bytes[g] + n > out_cap_per_gop) return fail(e, MI355X_H264_E_OVERFLOW, "gop output buffer too small");
            memcpy(host_out + (size_t)g * out_cap_per_gop + gop_bytes[g], p, n);
            sizes[(size_t)g * frames_per_gop + head] = n;
            gop_bytes[g] += n;
        }
        head++; npend--;
        return 0;
    };
    auto bail = [&](int rc) -> int {   // as in mi355x_h264_encode_batch_device
        char first[sizeof(e->err)];
        memcpy(first, e->err, sizeof(first));
        for (; npend > 0; head++, npend--) (void)wait_slot(e, pending[head % NSLOT]);
        memcpy(e->err, first, sizeof(first));
        return rc;
    };
    for (int i = 0; i < frames_per_gop; i++) {
        if (npend == NSLOT - 1) { int rc = drain_one(); if (rc) return bail(rc); }
        const int slot = e->next_slot;
        e->next_slot = (e->next_slot + 1) % NSLOT;
        pending[i % NSLOT] = slot;
        int rc = submit(e, (const uint8_t*)d_frames + (size_t)i * frame_stride, gop_stride, slot, e->cfg.input_format == MI355X_H264_INPUT_NV12);
        if (rc) return bail(rc);
        npend++;
    }
    while (npend) { int rc = drain_one(); if (rc) return bail(rc); }
    return MI355X_H264_OK;
}

int mi355x_h264_force_idr(mi355x_h264_encoder* e)
{
    if (!e) return MI355X_H264_E_ARG;
    e->force_idr = 1;
    return MI355X_H264_OK;
}

int mi355x_h264_last_me_cost(const mi355x_h264_encoder* e, uint32_t* cost)
{
    if (!e || !cost) return MI355X_H264_E_ARG;
    for (int g = 0; g < e->G; g++) cost[g] = e->last_me_cost[g];
    return MI355X_H264_OK;
}

int mi355x_h264_set_qp(mi355x_h264_encoder* e, int qp)
{
    if (!e || qp < 10 || qp > 51) return MI355X_H264_E_ARG;
    e->qp = qp;
    return MI355X_H264_OK;
}

int mi355x_h264_set_idr_pic_id(mi355x_h264_encoder* e, int next, int step)
{
    if (!e) return MI355X_H264_E_ARG;
    e->idr_id = next & 0xFF;
    e->idr_step = step;
    return MI355X_H264_OK;
}

// ---- slice bands over several GPUs: the rows next to a band in the reference picture come from the neighbours ----
namespace {
enum { HALO_MB_ROWS = 2 };   // 32 luma rows: the search reaches 16 rows + 0.75 + the 6-tap filter's 3, chroma half of that
// rows [r0, r1) of the newest reconstruction <-> a packed block (Y rows, then U rows, then V rows)
int halo_copy(mi355x_h264_encoder* e, int r0, int r1, void* d_blk, bool to_block)
{
    const int last = (e->cur + e->nbuf - 1) % e->nbuf;
    uint8_t* blk = (uint8_t*)d_blk;
    for (int p = 0; p < 3; p++) {
        const size_t pitch = p ? e->cw / 2 : e->cw, rows_per_mb = p ? 8 : 16;
        const size_t off = (size_t)r0 * rows_per_mb * pitch, n = (size_t)(r1 - r0) * rows_per_mb * pitch;
        uint8_t* pl = e->d_planes[last][p] + off;
        if (n) HIPCHK(e, hipMemcpyAsync(to_block ? (void*)blk : (void*)pl, to_block ? (const void*)pl : (const void*)blk, n, hipMemcpyDefault, e->stream));   // the block may be device or host memory
        blk += (size_t)HALO_MB_ROWS * rows_per_mb * pitch;   // fixed layout, whatever the number of rows present
    }
    HIPCHK(e, hipStreamSynchronize(e->stream));
    return MI355X_H264_OK;
}
}  // namespace

int mi355x_h264_band_info(const mi355x_h264_encoder* e, int* first_row, int* rows, int* first_slice, int* slices, size_t* halo_bytes)
{
    if (!e) return MI355X_H264_E_ARG;
    if (first_row) *first_row = e->b_row0;
    if (rows) *rows = e->b_rows;
    if (first_slice) *first_slice = e->b_sl0;
    if (slices) *slices = e->b_nsl;
    if (halo_bytes) *halo_bytes = (size_t)HALO_MB_ROWS * 16 * e->cw * 3 / 2;
    return MI355X_H264_OK;
}

int mi355x_h264_band_halo_export(mi355x_h264_encoder* e, int edge, void* d_dst)
{
    if (!e || !d_dst || (edge != 0 && edge != 1)) return fail(e, MI355X_H264_E_ARG, "bad argument");
    HIPCHK(e, hipSetDevice(e->device));
    const int n = std::min((int)HALO_MB_ROWS, e->b_rows);
    const int r0 = edge == 0 ? e->b_row0 : e->b_row0 + e->b_rows - n;
    return halo_copy(e, r0, r0 + n, d_dst, true);
}

int mi355x_h264_band_halo_import(mi355x_h264_encoder* e, int edge, const void* d_src)
{
    if (!e || !d_src || (edge != 0 && edge != 1)) return fail(e, MI355X_H264_E_ARG, "bad argument");
    HIPCHK(e, hipSetDevice(e->device));
    // above: the neighbour's LAST rows end right above this band; below: its FIRST rows start right below
    int r0, r1;
    if (edge == 0) { r1 = e->b_row0; r0 = std::max(0, r1 - (int)HALO_MB_ROWS); if (r1 - r0 < (int)HALO_MB_ROWS && r1 > 0) return fail(e, MI355X_H264_E_INTERNAL, "band above is shorter than the halo"); }
    else { r0 = e->b_row0 + e->b_rows; r1 = std::min(e->mbh, r0 + (int)HALO_MB_ROWS); }
    if (r1 <= r0) return MI355X_H264_OK;   // picture edge: nothing beyond
    return halo_copy(e, r0, r1, const_cast<void*>(d_src), false);
}

const char* mi355x_h264_last_error(const mi355x_h264_encoder* e) { return e ? e->err : "null encoder"; }
int mi355x_h264_coded_width(const mi355x_h264_encoder* e) { return e ? e->cw : 0; }
int mi355x_h264_coded_height(const mi355x_h264_encoder* e) { return e ? e->ch : 0; }

int mi355x_h264_debug_keep_pre(mi355x_h264_encoder* e, int on)
{
    if (!e) return MI355X_H264_E_ARG;
    e->keep_pre = on != 0;
    return MI355X_H264_OK;
}

int64_t mi355x_h264_debug_read(mi355x_h264_encoder* e, int what, void* dst, size_t cap)
{
    if (!e || !dst) return MI355X_H264_E_ARG;
    if (hipSetDevice(e->device) != hipSuccess) return MI355X_H264_E_HIP;
    const void* src = nullptr;
    size_t n = 0;
    const size_t ysz = (size_t)e->cw * e->ch;
    const int last = (e->cur + e->nbuf - 1) % e->nbuf;  // picture finished by the last encode
    switch (what) {
        case MI355X_H264_DBG_RECON_Y: case MI355X_H264_DBG_RECON_U: case MI355X_H264_DBG_RECON_V:
            src = e->d_planes[last][what]; n = what ? ysz / 4 : ysz; break;
        case MI355X_H264_DBG_PRE_Y: case MI355X_H264_DBG_PRE_U: case MI355X_H264_DBG_PRE_V:
            src = e->d_pre[what - MI355X_H264_DBG_PRE_Y]; n = what != MI355X_H264_DBG_PRE_Y ? ysz / 4 : ysz; break;
        case MI355X_H264_DBG_MBINFO: src = e->d_mb; n = (size_t)e->nmb * sizeof(MbInfo); break;
        case MI355X_H264_DBG_LEVELS: src = e->d_levels; n = (size_t)e->nmb * LV_STRIDE * 2; break;
        case MI355X_H264_DBG_MBAUX: src = e->d_aux; n = (size_t)e->nmb * 16; break;
        case MI355X_H264_DBG_MVQ: src = e->d_mvq; n = (size_t)e->nmb * 16; break;
        default: return MI355X_H264_E_ARG;
    }
    if (cap < n) return MI355X_H264_E_ARG;
    if (hipStreamSynchronize(e->stream) != hipSuccess) return MI355X_H264_E_HIP;
    if (hipMemcpy(dst, src, n, hipMemcpyDeviceToHost) != hipSuccess) return MI355X_H264_E_HIP;
    return (int64_t)n;
}

int mi355x_h264_stats_enable(mi355x_h264_encoder* e, int on)
{
    if (!e) return MI355X_H264_E_ARG;
    e->stats_on = on != 0;
    return MI355X_H264_OK;
}

int mi355x_h264_stats_read(mi355x_h264_encoder* e, mi355x_h264_stats* out, int reset)
{
    if (!e || !out) return MI355X_H264_E_ARG;
    *out = e->stats;
    if (reset) memset(&e->stats, 0, sizeof(e->stats));
    return MI355X_H264_OK;
}

}  // extern "C"

// ===========================================================================
// Stream hub (include/mi355x_h264.h, "streams"): the reference's operating mode - many encoder objects in one process, each
// handed ONE picture per call by its own thread (VideoEncoderOpenH264.cpp:304-352) - without one engine and ~15 single-picture
// launches per object.  Streams of one geometry share an engine whose batch items are the streams; the pictures that calls
// deliver while the engine is busy leave together as ONE lockstep step (the IND = true kernels: every position of the grid has
// its own item, ring slot, QP, frame_num, idr_pic_id), split only by picture type.  No thread is created: the caller that finds
// a free step context becomes the step's leader (gathers what is queued, launches, waits, finishes every picture of the step),
// the others sleep until their picture is done.  Two step contexts per hub: one step's loop filter overlaps the next one's
// motion search, as two instances do in the closed-GOP mode.
// ===========================================================================
namespace {

struct HubItem {
    bool open = false;
    // coding state of the stream (what mi355x_h264_encoder keeps for its one stream)
    int cur = 0, frame_in_gop = 0, frame_num = 0, idr_id = 0, force_idr = 0, qp = 26, gop = 30;
    long frames = 0;
    int last_cur = 0;                // ring slot of the last finished picture
    hipEvent_t copied = nullptr;     // the picture's upload has finished
    // the request in flight
    bool pending = false, done = false;
    int rc = 0, frame_type = 0;
    uint8_t* out = nullptr;
    uint32_t out_len = 0;
    char err[256] = {0};
};

struct HubCtx {
    hipStream_t st = nullptr, ec = nullptr;
    hipEvent_t recon_ready = nullptr, entropy_done = nullptr, done = nullptr;
    unsigned* h_err = nullptr;       // pinned: hand-off time-out flag of the wavefront kernels
    uint32_t* h_itemtab = nullptr;   // pinned, MAX_BATCH words
    uint32_t* d_itemtab = nullptr;
    bool busy = false;
};

struct Hub {
    std::mutex mu;                   // queue + item states
    std::condition_variable cv;
    std::mutex launch_mu;            // one leader at a time touches the engine's host state (serials, statistics)
    mi355x_h264_encoder* e = nullptr;
    mi355x_h264_config cfg{};
    int cap = 0, nopen = 0, uploading = 0;
    HubItem items[MAX_BATCH];
    // P pictures and IDR pictures never share a step: an IDR picture's row wavefront (k_intra_rows) runs for milliseconds, and
    // the P pictures of other streams must not wait for it.  Contexts 0 and 1 take the P steps (one's loop filter overlaps the
    // other's motion search), context 2 the IDR steps.
    enum { MAX_CTX = 9 };
    int nctx_p = 2;                  // contexts for P steps: ctx[0 .. nctx_p - 1]; ctx[nctx_p] takes the IDR steps
    std::vector<int> queue[2];       // [0] P pictures, [1] IDR pictures waiting for a step
    bool collecting = false;         // a leader is gathering a P step
    HubCtx ctx[MAX_CTX];
    bool any_busy() const { for (int i = 0; i <= nctx_p; i++) if (ctx[i].busy) return true; return false; }
    uint8_t* d_stage = nullptr;      // [cap] pictures as the callers hand them over (tight I420)
    uint8_t* h_stage = nullptr;      // pinned
    size_t st_stage = 0;
    // uploads: item k on copy stream k % NCOPY.  Two streams fill most of the link (tools/ubench_h2d.hip: 1 stream 32 GB/s, 2: 46-51,
    // 4+: 52-57); HIP streams are a scarce resource on this runtime - beyond about a dozen live streams in the process every launch
    // gets slower (measured: 8 copy streams per hub halved the throughput at 64 streams)
    enum { NCOPY = 2 };
    hipStream_t copy_st[NCOPY] = {nullptr};
    int window_us = 200;
    uint64_t steps = 0, pictures = 0, max_batch = 0;
    // where a picture's time goes (microseconds, summed; MI355X_H264_HUB_VERBOSE=1 prints them when the hub is freed)
    std::atomic<uint64_t> us_upload{0}, us_queue{0}, us_launch{0}, us_gpu{0}, us_finish{0}, us_total{0};
    bool verbose = false;
};
inline uint64_t now_us() { return (uint64_t)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

std::mutex g_hubs_mu;
std::vector<Hub*> g_hubs;
std::atomic<int> g_streams_open{0};   // over all hubs of the process

bool same_geometry(const mi355x_h264_config& a, const mi355x_h264_config& b)
{
    return a.width == b.width && a.height == b.height && a.fps == b.fps && a.profile_idc == b.profile_idc && a.device == b.device &&
           a.disable_deblock == b.disable_deblock && a.slices == b.slices && a.search == b.search;
}

void hub_free(Hub* h)
{
    if (!h) return;
    (void)hipSetDevice(h->cfg.device);
    for (auto& c : h->ctx) {
        if (!c.st) continue;
        if (c.st) (void)hipStreamSynchronize(c.st);
        if (c.ec && c.ec != c.st) { (void)hipStreamSynchronize(c.ec); (void)hipStreamDestroy(c.ec); }
        if (c.st) (void)hipStreamDestroy(c.st);
        if (c.recon_ready) (void)hipEventDestroy(c.recon_ready);
        if (c.entropy_done) (void)hipEventDestroy(c.entropy_done);
        if (c.done) (void)hipEventDestroy(c.done);
        if (c.h_err) (void)hipHostFree(c.h_err);
        if (c.h_itemtab) (void)hipHostFree(c.h_itemtab);
        (void)hipFree(c.d_itemtab);
    }
    for (auto& cs : h->copy_st) if (cs) { (void)hipStreamSynchronize(cs); (void)hipStreamDestroy(cs); }
    for (auto& it : h->items) if (it.copied) (void)hipEventDestroy(it.copied);
    (void)hipFree(h->d_stage);
    if (h->h_stage) (void)hipHostFree(h->h_stage);
    if (h->verbose && h->pictures)
        fprintf(stderr, "mi355x_h264 hub %dx%d: %llu pictures in %llu steps (%.2f per step, largest %llu); per picture: upload %.0f us, queued %.0f us, "
                        "whole call %.0f us; per step: launch %.0f us, GPU wait %.0f us, finish %.0f us\n", h->cfg.width, h->cfg.height,
                (unsigned long long)h->pictures, (unsigned long long)h->steps, (double)h->pictures / h->steps, (unsigned long long)h->max_batch,
                (double)h->us_upload / h->pictures, (double)h->us_queue / h->pictures, (double)h->us_total / h->pictures,
                (double)h->us_launch / h->steps, (double)h->us_gpu / h->steps, (double)h->us_finish / h->steps);
    if (h->e) mi355x_h264_destroy(h->e);
    delete h;
}

int hub_create(const mi355x_h264_config& cfg, Hub** out)
{
    Hub* h = new (std::nothrow) Hub();
    if (!h) return MI355X_H264_E_NOMEM;
    h->cfg = cfg;
    const char* ci = getenv("MI355X_H264_HUB_ITEMS");
    h->cap = std::min((int)MAX_BATCH, std::max(1, ci ? atoi(ci) : 32));
    const char* wu = getenv("MI355X_H264_HUB_WINDOW_US");
    if (wu) h->window_us = std::max(0, atoi(wu));
    h->verbose = getenv("MI355X_H264_HUB_VERBOSE") != nullptr;
    // MI355X_H264_HUB_CTX = contexts for P steps (default 2, 1..8): one step's loop filter overlaps the other's motion search.  More
    // contexts mean more HIP streams, and those cost more than they bring (measured: 4 contexts -5 %, 6 contexts -50 %)
    const char* nc = getenv("MI355X_H264_HUB_CTX");
    h->nctx_p = std::min((int)Hub::MAX_CTX - 1, std::max(1, nc ? atoi(nc) : 2));
    mi355x_h264_config ec = cfg;
    ec.batch = h->cap; ec.refs = 1; ec.band_index = 0; ec.band_count = 0; ec.input_format = MI355X_H264_INPUT_I420;
    int rc = create_engine(&ec, &h->e, true);
    if (rc != MI355X_H264_OK) { h->e = nullptr; hub_free(h); return rc; }
    const size_t fb = (size_t)cfg.width * cfg.height * 3 / 2;
    h->st_stage = (fb + 255) & ~(size_t)255;
#define HK(call) do { if ((call) != hipSuccess) { hub_free(h); return MI355X_H264_E_HIP; } } while (0)
    HK(hipSetDevice(cfg.device));
    HK(hipMalloc((void**)&h->d_stage, h->st_stage * h->cap));
    HK(hipHostMalloc((void**)&h->h_stage, h->st_stage * h->cap, hipHostMallocDefault));
    for (auto& cs : h->copy_st) HK(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
    for (int ci = 0; ci <= h->nctx_p; ci++) {
        HubCtx& c = h->ctx[ci];
        HK(hipStreamCreateWithFlags(&c.st, hipStreamNonBlocking));
        const char* one = getenv("MI355X_H264_ONE_STREAM");
        if ((one && one[0] == '1') || ci == h->nctx_p) c.ec = c.st;   // (the IDR context: its row wavefront dominates, nothing to overlap)
        else HK(hipStreamCreateWithFlags(&c.ec, hipStreamNonBlocking));
        HK(hipEventCreateWithFlags(&c.recon_ready, hipEventDisableTiming));
        HK(hipEventCreateWithFlags(&c.entropy_done, hipEventDisableTiming));
        HK(hipEventCreateWithFlags(&c.done, hipEventDisableTiming));
        HK(hipHostMalloc((void**)&c.h_err, sizeof(unsigned), hipHostMallocDefault));
        *c.h_err = 0;
        HK(hipHostMalloc((void**)&c.h_itemtab, MAX_BATCH * sizeof(uint32_t), hipHostMallocDefault));
        HK(hipMalloc((void**)&c.d_itemtab, MAX_BATCH * sizeof(uint32_t)));
    }
    for (int i = 0; i < h->cap; i++) HK(hipEventCreateWithFlags(&h->items[i].copied, hipEventDisableTiming));
#undef HK
    *out = h;
    return MI355X_H264_OK;
}

// will the stream's next picture be an IDR picture?
bool hub_next_is_idr(const HubItem& it) { return it.force_idr || it.frames == 0 || it.frame_in_gop >= it.gop; }

// one lockstep step for the queued pictures `batch` (all of one type) on context c: launch, wait, finish
void hub_run_step(Hub* h, HubCtx& c, const std::vector<int>& batch, bool idr)
{
    mi355x_h264_encoder* e = h->e;
    (void)hipSetDevice(h->cfg.device);
    ItemPic pics[MAX_BATCH];
    const int n = (int)batch.size();
    for (int k = 0; k < n; k++) {
        HubItem& it = h->items[batch[k]];
        if (idr) { it.frame_in_gop = 0; it.frame_num = 0; }
        it.force_idr = 0;
        pics[k] = ItemPic{batch[k], it.cur, it.qp, it.frame_num, it.idr_id};
    }
    Step T;
    int rc = MI355X_H264_OK;
    char errtxt[256] = {0};
    const uint64_t t0 = now_us();
    {
        std::lock_guard<std::mutex> lk(h->launch_mu);
        for (int k = 0; k < n; k++) {
            c.h_itemtab[k] = (uint32_t)pics[k].item | ((uint32_t)pics[k].cur << 8) | ((uint32_t)pics[k].qp << 16);
            if (hipStreamWaitEvent(c.st, h->items[pics[k].item].copied, 0) != hipSuccess) rc = MI355X_H264_E_HIP;
        }
        if (hipMemcpyAsync(c.d_itemtab, c.h_itemtab, (size_t)n * sizeof(uint32_t), hipMemcpyHostToDevice, c.st) != hipSuccess) rc = MI355X_H264_E_HIP;
        if (rc == MI355X_H264_OK) {
            T.d_src = h->d_stage; T.src_item_stride = h->st_stage; T.nv12 = false; T.idr = idr; T.n = n;
            T.items = pics; T.d_itemtab = c.d_itemtab;
            // entropy coding beside the loop filter shortens a picture's latency; with many streams open the second HIP stream
            // costs more than the overlap brings (64 streams: 10.9 k -> 12.1 k fps on one stream per step)
            T.st = c.st; T.ec = g_streams_open.load(std::memory_order_relaxed) > 40 ? c.st : c.ec; T.recon_ready = c.recon_ready; T.entropy_done = c.entropy_done; T.done = c.done; T.h_err = c.h_err;
            T.slot = &e->slots[0];
            rc = submit_step(e, T);
        }
        if (rc != MI355X_H264_OK) snprintf(errtxt, sizeof(errtxt), "%s", e->err);
    }
    const uint64_t t1 = now_us();
    if (rc == MI355X_H264_OK) {
        if (hipEventSynchronize(c.done) != hipSuccess) { rc = MI355X_H264_E_HIP; snprintf(errtxt, sizeof(errtxt), "hipEventSynchronize failed"); }
    } else (void)hipStreamSynchronize(c.st);
    if (rc == MI355X_H264_OK && *c.h_err) {
        snprintf(errtxt, sizeof(errtxt), "wavefront kernel hand-off timed out (flag %u)", *c.h_err);
        *c.h_err = 0;
        rc = MI355X_H264_E_INTERNAL;
    }
    const uint64_t t2 = now_us();
    std::lock_guard<std::mutex> lk(h->launch_mu);   // (finish_item touches the engine's statistics and error text)
    struct Acc { Hub* h; uint64_t a, b, c; ~Acc() { h->us_launch += b - a; h->us_gpu += c - b; h->us_finish += now_us() - c; } } acc{h, t0, t1, t2};
    for (int k = 0; k < n; k++) {
        HubItem& it = h->items[pics[k].item];
        it.rc = rc;
        if (rc == MI355X_H264_OK) {
            const AuLayout L{T.au_start, T.payload_off, idr, T.nal_hdr};
            it.rc = finish_item(e, e->slots[0], L, pics[k].item, &it.out, &it.out_len, &it.frame_type);
            if (it.rc != MI355X_H264_OK) snprintf(it.err, sizeof(it.err), "%s", e->err);
        } else snprintf(it.err, sizeof(it.err), "%s", errtxt);
        if (it.rc == MI355X_H264_OK) {
            it.last_cur = it.cur;
            it.cur = (it.cur + 1) % e->nbuf;
            if (idr) it.idr_id = (it.idr_id + 1) & 0xFF;
            it.frame_num = (it.frame_num + 1) & 255;
            it.frame_in_gop++;
            it.frames++;
        } else it.force_idr = 1;   // the picture is missing from the stream (or not to be trusted): the next one must not refer to it
    }
}

}  // namespace

struct mi355x_h264_stream { Hub* hub; int item; };

extern "C" {

int mi355x_h264_stream_open(const mi355x_h264_config* cfg, mi355x_h264_stream** out)
{
    if (!cfg || !out || cfg->struct_size != sizeof(mi355x_h264_config)) return MI355X_H264_E_ARG;
    *out = nullptr;
    if (cfg->refs > 1 || cfg->band_count > 1 || cfg->batch > 1 || cfg->input_format != MI355X_H264_INPUT_I420) return MI355X_H264_E_ARG;
    if (cfg->qp < 10 || cfg->qp > 51 || cfg->gop < 1) return MI355X_H264_E_ARG;
    mi355x_h264_stream* s = new (std::nothrow) mi355x_h264_stream();
    if (!s) return MI355X_H264_E_NOMEM;
    std::lock_guard<std::mutex> gl(g_hubs_mu);
    Hub* h = nullptr;
    for (Hub* c : g_hubs) {
        std::lock_guard<std::mutex> lk(c->mu);
        if (same_geometry(c->cfg, *cfg) && c->nopen < c->cap) { h = c; break; }
    }
    if (!h) {
        const int rc = hub_create(*cfg, &h);
        if (rc != MI355X_H264_OK) { delete s; return rc; }
        g_hubs.push_back(h);
    }
    std::lock_guard<std::mutex> lk(h->mu);
    int idx = 0;
    while (h->items[idx].open) idx++;
    HubItem& it = h->items[idx];
    hipEvent_t ev = it.copied;
    it = HubItem();
    it.copied = ev;
    it.open = true; it.qp = cfg->qp; it.gop = cfg->gop;
    h->nopen++;
    g_streams_open.fetch_add(1);
    s->hub = h; s->item = idx;
    *out = s;
    return MI355X_H264_OK;
}

void mi355x_h264_stream_close(mi355x_h264_stream* s)
{
    if (!s) return;
    std::lock_guard<std::mutex> gl(g_hubs_mu);
    Hub* h = s->hub;
    bool last;
    {
        std::unique_lock<std::mutex> lk(h->mu);
        h->items[s->item].open = false;
        g_streams_open.fetch_sub(1);
        last = --h->nopen == 0;
        if (last) h->cv.wait(lk, [&] { return !h->any_busy(); });
    }
    if (last) {
        g_hubs.erase(std::find(g_hubs.begin(), g_hubs.end(), h));
        hub_free(h);
    }
    delete s;
}

int mi355x_h264_stream_encode(mi355x_h264_stream* s, const uint8_t* y, int ys, const uint8_t* u, int us, const uint8_t* v, int vs,
                              uint8_t** out, uint32_t* out_len, int* frame_type)
{
    if (!s || !y || !u || !v || !out || !out_len) return MI355X_H264_E_ARG;
    Hub* h = s->hub;
    HubItem& it = h->items[s->item];
    const int w = h->cfg.width, hh = h->cfg.height;
    if (ys < w || us < w / 2 || vs < w / 2) { snprintf(it.err, sizeof(it.err), "stride smaller than width"); return MI355X_H264_E_ARG; }
    if (hipSetDevice(h->cfg.device) != hipSuccess) { snprintf(it.err, sizeof(it.err), "hipSetDevice"); return MI355X_H264_E_HIP; }
    const uint64_t t_in = now_us();
    int nopen;
    {
        std::lock_guard<std::mutex> lk(h->mu);
        h->uploading++;   // a step that is being gathered waits (briefly) for this picture
        nopen = h->nopen;
    }
    // 1. the picture into the stream's staging slot: pinned copy, then the transfer - in pieces, so that the copy of piece k + 1
    // runs while piece k is on the bus (the reference's tight layout, InitSrcPic ref :354-365; other layouts row by row)
    uint8_t* hs = h->h_stage + (size_t)s->item * h->st_stage;
    uint8_t* ds = h->d_stage + (size_t)s->item * h->st_stage;
    hipStream_t cs = h->copy_st[s->item % Hub::NCOPY];
    const size_t ysz = (size_t)w * hh, fb = ysz * 3 / 2;
    bool ok = true;
    if (ys == w && us == w / 2 && vs == w / 2 && u == y + ysz && v == u + ysz / 4) {
        // few streams: four pieces, so that the copy of piece k + 1 runs while piece k is on the bus (latency); many streams: one
        // transfer per picture (every queued command costs, and other streams' transfers fill the bus anyway: 16 / 32 / 64 streams
        // went from 7.3 / 8.0 / 8.6 k to 8.8 / 11.5 / 10.9 k fps with this alone, profiles/r03_hub_sweep_*.log)
        const size_t piece = nopen > 4 ? fb : (((fb / 4) + 255) & ~(size_t)255);
        for (size_t o = 0; o < fb && ok; o += piece) {
            const size_t len = std::min(piece, fb - o);
            memcpy(hs + o, y + o, len);
            ok = hipMemcpyAsync(ds + o, hs + o, len, hipMemcpyHostToDevice, cs) == hipSuccess;
        }
    } else {
        uint8_t* d = hs;
        for (int r = 0; r < hh; r++) memcpy(d + (size_t)r * w, y + (size_t)r * ys, (size_t)w);
        d += ysz;
        for (int r = 0; r < hh / 2; r++) memcpy(d + (size_t)r * (w / 2), u + (size_t)r * us, (size_t)(w / 2));
        d += ysz / 4;
        for (int r = 0; r < hh / 2; r++) memcpy(d + (size_t)r * (w / 2), v + (size_t)r * vs, (size_t)(w / 2));
        ok = hipMemcpyAsync(ds, hs, fb, hipMemcpyHostToDevice, cs) == hipSuccess;
    }
    ok = ok && hipEventRecord(it.copied, cs) == hipSuccess;
    // 2. queue the picture; lead a step or wait for the one that takes it
    std::unique_lock<std::mutex> lk(h->mu);
    h->uploading--;
    if (!ok) { h->cv.notify_all(); snprintf(it.err, sizeof(it.err), "upload of the picture failed"); return MI355X_H264_E_HIP; }
    it.pending = true; it.done = false;
    const uint64_t t_q = now_us();
    h->us_upload += t_q - t_in;
    h->queue[hub_next_is_idr(it) ? 1 : 0].push_back(s->item);
    h->cv.notify_all();   // (a leader that is gathering counts the uploads still on their way)
    while (!it.done) {
        // lead a step if one can start: an IDR step when IDR pictures wait and the IDR context is free, else a P step
        HubCtx* c = nullptr;
        bool idr = false;
        if (!h->queue[1].empty() && !h->ctx[h->nctx_p].busy) { c = &h->ctx[h->nctx_p]; idr = true; }
        else if (!h->queue[0].empty() && !h->collecting)
            for (int ci = 0; ci < h->nctx_p && !c; ci++) if (!h->ctx[ci].busy) c = &h->ctx[ci];
        if (!c) { h->cv.wait(lk); continue; }
        c->busy = true;
        if (!idr && h->uploading > 0 && h->window_us > 0) {   // pictures on their way in join this step if they make it within the window
            h->collecting = true;
            const auto deadline = std::chrono::steady_clock::now() + std::chrono::microseconds(h->window_us);
            h->cv.wait_until(lk, deadline, [&] { return h->uploading == 0; });
            h->collecting = false;
        }
        // A P step takes at most its share of the open streams: with nctx_p steps in flight and one share uploading, a context
        // that frees finds pictures already uploaded instead of waiting for the streams it has just released to come back
        std::vector<int> batch;
        std::vector<int>& q = h->queue[idr ? 1 : 0];
        const size_t share = idr ? q.size() : std::max<size_t>(1, ((size_t)h->nopen + h->nctx_p) / (h->nctx_p + 1));
        if (q.size() <= share) batch.swap(q);
        else { batch.assign(q.begin(), q.begin() + share); q.erase(q.begin(), q.begin() + share); }
        h->steps++; h->pictures += batch.size(); h->max_batch = std::max<uint64_t>(h->max_batch, batch.size());
        h->us_queue += (now_us() - t_q);   // (the leader's own wait; the followers' is within a step of it)
        if (!q.empty()) h->cv.notify_all();   // what is left can start on another free context at once
        lk.unlock();
        hub_run_step(h, *c, batch, idr);
        lk.lock();
        for (int idx : batch) { h->items[idx].done = true; h->items[idx].pending = false; }
        c->busy = false;
        h->cv.notify_all();
    }
    h->us_total += now_us() - t_in;
    *out = it.out; *out_len = it.out_len;
    if (frame_type) *frame_type = it.frame_type;
    return it.rc;
}

int mi355x_h264_stream_set_qp(mi355x_h264_stream* s, int qp)
{
    if (!s || qp < 10 || qp > 51) return MI355X_H264_E_ARG;
    s->hub->items[s->item].qp = qp;
    return MI355X_H264_OK;
}

int mi355x_h264_stream_force_idr(mi355x_h264_stream* s)
{
    if (!s) return MI355X_H264_E_ARG;
    s->hub->items[s->item].force_idr = 1;
    return MI355X_H264_OK;
}

int mi355x_h264_stream_set_idr_pic_id(mi355x_h264_stream* s, int next)
{
    if (!s) return MI355X_H264_E_ARG;
    s->hub->items[s->item].idr_id = next & 0xFF;
    return MI355X_H264_OK;
}

int mi355x_h264_stream_last_me_cost(const mi355x_h264_stream* s, uint32_t* cost)
{
    if (!s || !cost) return MI355X_H264_E_ARG;
    *cost = s->hub->e->last_me_cost[s->item];
    return MI355X_H264_OK;
}

const char* mi355x_h264_stream_last_error(const mi355x_h264_stream* s) { return s ? s->hub->items[s->item].err : "null stream"; }
int mi355x_h264_stream_coded_width(const mi355x_h264_stream* s) { return s ? s->hub->e->cw : 0; }
int mi355x_h264_stream_coded_height(const mi355x_h264_stream* s) { return s ? s->hub->e->ch : 0; }

// reconstruction planes of the stream's last picture (MI355X_H264_DBG_RECON_Y / _U / _V); the stream's calls are synchronous, so
// the picture is complete
int64_t mi355x_h264_stream_debug_read(mi355x_h264_stream* s, int what, void* dst, size_t cap)
{
    if (!s || !dst || what < MI355X_H264_DBG_RECON_Y || what > MI355X_H264_DBG_RECON_V) return MI355X_H264_E_ARG;
    Hub* h = s->hub;
    const mi355x_h264_encoder* e = h->e;
    if (hipSetDevice(h->cfg.device) != hipSuccess) return MI355X_H264_E_HIP;
    const size_t ysz = (size_t)e->cw * e->ch, n = what ? ysz / 4 : ysz;
    if (cap < n) return MI355X_H264_E_ARG;
    const HubItem& it = h->items[s->item];
    const uint8_t* src = e->d_plane_base[what] + (size_t)s->item * (what ? e->st_c : e->st_y) + (size_t)it.last_cur * (what ? e->st_ring_c : e->st_ring_y);
    if (hipMemcpy(dst, src, n, hipMemcpyDeviceToHost) != hipSuccess) return MI355X_H264_E_HIP;
    return (int64_t)n;
}

// how the hub of this stream has been batching: steps launched, pictures coded, the largest step
int mi355x_h264_stream_hub_stats(const mi355x_h264_stream* s, uint64_t* steps, uint64_t* pictures, uint64_t* max_batch, int* open_streams)
{
    if (!s) return MI355X_H264_E_ARG;
    Hub* h = s->hub;
    std::lock_guard<std::mutex> lk(h->mu);
    if (steps) *steps = h->steps;
    if (pictures) *pictures = h->pictures;
    if (max_batch) *max_batch = h->max_batch;
    if (open_streams) *open_streams = h->nopen;
    return MI355X_H264_OK;
}

}  // extern "C"

// ===========================================================================
// Decoder peer (include/mi355x_h264_dec.h): host parser (h264_parse.h) + the reconstruction kernels (k_dec.h, k_intra.h,
// k_deblock.h).  The decoder owns an engine instance for its device buffers (reconstruction ring, per-macroblock
// arrays, hand-off granules, streams): decoding is the encoder's reconstruction path run from parsed decisions.
// ===========================================================================
struct mi355x_h264_decoder {
    h264dec::Parser parser;
    mi355x_h264_encoder* eng = nullptr;
    int device = 0;
    int mbw = 0, mbh = 0;
    int have_refs = 0;   // reference pictures in the ring (sliding window)
    int max_refs = 1;
    int last = -1;       // ring index of the last decoded picture
    int width = 0, height = 0, crop_x = 0, crop_y = 0;
    uint64_t pictures = 0;
    double parse_ms = 0, gpu_ms = 0;
    uint8_t* d_mbqp = nullptr;   // QP_Y per macroblock of the picture being reconstructed
    int16_t* d_mv4 = nullptr;    // its vectors per 4x4 block (32 int16 per macroblock)
    uint8_t* d_refq = nullptr;   // and reference indices per quadrant (4 per macroblock)
    uint8_t* d_mbavail = nullptr;   // neighbour availability bits per macroblock
    uint32_t* d_lv8 = nullptr;      // the levels as they arrive: one byte each (k_dec_widen fills the engine's int16 lists)
    DecBigLevel* d_big = nullptr;   // levels that did not fit a byte
    size_t big_cap = 0;
    // One picture of look-ahead: decode() returns once picture n is LAUNCHED; the parse of access unit n + 1 then runs on the
    // host while the GPU reconstructs n.  The parser fills two picture buffers in turn (pinned memory: the uploads are
    // asynchronous); up_done[k] = the uploads out of buffer k have finished, so it may be parsed into again.
    int buf = 0;
    hipEvent_t up_done[2] = {nullptr, nullptr};
    bool up_pending[2] = {false, false};
    bool busy = false;           // a picture is in flight on the engine's stream
    char err[256] = {0};
};

namespace {

int dfail(mi355x_h264_decoder* d, int code, const char* fmt, ...)
{
    if (d) {
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(d->err, sizeof(d->err), fmt, ap);
        va_end(ap);
    }
    return code;
}
#define DHIP(d, call)                                                                                \
    do {                                                                                             \
        hipError_t _r = (call);                                                                      \
        if (_r != hipSuccess) return dfail((d), MI355X_H264_E_HIP, "%s: %s", #call, hipGetErrorString(_r)); \
    } while (0)

double now_ms()
{
    timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return t.tv_sec * 1e3 + t.tv_nsec * 1e-6;
}

// the picture in flight has finished (and its wavefront kernels did not time out)
int dec_wait(mi355x_h264_decoder* d)
{
    if (!d->busy) return MI355X_H264_OK;
    d->busy = false;
    mi355x_h264_encoder* e = d->eng;
    DHIP(d, hipStreamSynchronize(e->stream));
    Slot& S = e->slots[0];
    if (*S.h_err) {
        const unsigned flag = *S.h_err;
        *S.h_err = 0;
        d->have_refs = 0;   // that picture is not a usable reference: P pictures are refused until the next IDR
        return dfail(d, MI355X_H264_E_INTERNAL, "wavefront kernel hand-off timed out (flag %u)", flag);
    }
    return MI355X_H264_OK;
}

// launch the reconstruction of the parsed picture (parser buffer d->buf) into ring slot e->cur; does not wait for it
int dec_submit(mi355x_h264_decoder* d, const h264dec::Picture& pic)
{
    mi355x_h264_encoder* e = d->eng;
    const size_t nmb = (size_t)e->nmb;
    hipStream_t st = e->stream;
    DHIP(d, hipMemcpyAsync(e->d_mb, pic.mb.data(), nmb * sizeof(MbInfo), hipMemcpyHostToDevice, st));
    DHIP(d, hipMemcpyAsync(e->d_mvq, pic.mvq.data(), nmb * 16, hipMemcpyHostToDevice, st));
    DHIP(d, hipMemcpyAsync(e->d_aux, pic.aux.data(), nmb * 16, hipMemcpyHostToDevice, st));
    DHIP(d, hipMemcpyAsync(d->d_lv8, pic.levels8.data(), nmb * LV_STRIDE, hipMemcpyHostToDevice, st));
    {
        const int words = (int)(nmb * (LV_STRIDE / 4));
        hipLaunchKernelGGL(k_dec_widen, dim3((words + 255) / 256), dim3(256), 0, st, (const uint32_t*)d->d_lv8, (const MbInfo*)e->d_mb, e->d_levels, (int)nmb);
        if (!pic.big.empty()) {
            static_assert(sizeof(h264dec::Picture::Big) == sizeof(DecBigLevel), "layout of the list of large levels");
            if (pic.big.size() > d->big_cap) {
                DHIP(d, hipStreamSynchronize(st));
                if (d->d_big) (void)hipFree(d->d_big);
                d->d_big = nullptr;
                d->big_cap = pic.big.size() * 2 + 1024;
                DHIP(d, hipMalloc((void**)&d->d_big, d->big_cap * sizeof(DecBigLevel)));
            }
            DHIP(d, hipMemcpyAsync(d->d_big, pic.big.data(), pic.big.size() * sizeof(DecBigLevel), hipMemcpyHostToDevice, st));
            hipLaunchKernelGGL(k_dec_patch, dim3(((int)pic.big.size() + 255) / 256), dim3(256), 0, st, (const DecBigLevel*)d->d_big, (int)pic.big.size(), e->d_levels);
        }
    }
    DHIP(d, hipMemcpyAsync(d->d_mbqp, pic.mbqp.data(), nmb, hipMemcpyHostToDevice, st));
    DHIP(d, hipMemcpyAsync(d->d_mbavail, pic.mbavail.data(), nmb, hipMemcpyHostToDevice, st));
    if (pic.has_inter) {
        DHIP(d, hipMemcpyAsync(d->d_mv4, pic.mv4.data(), nmb * 64, hipMemcpyHostToDevice, st));
        DHIP(d, hipMemcpyAsync(d->d_refq, pic.refq.data(), nmb * 4, hipMemcpyHostToDevice, st));
    }
    const int cur = e->cur;
    FrameParams P{};
    P.w = e->cw; P.h = e->ch; P.cw = e->cw; P.ch = e->ch; P.mbw = e->mbw; P.mbh = e->mbh;
    P.nref = std::max(1, d->have_refs);
    for (int p = 0; p < 3; p++) {
        P.rec[p] = e->d_planes[cur][p];
        for (int r = 0; r < mi355x_h264_encoder::MAX_REFS; r++) {
            // RefPicList0 entry r = the reference picture decoded ref_age[r] + 1 reference pictures ago (ring slot cur - 1 - age)
            const int age = std::min(r < pic.num_ref_active ? pic.ref_age[r] : r, std::max(0, d->have_refs - 1));
            P.refs[r][p] = e->d_planes[(cur + e->nbuf - 1 - age) % e->nbuf][p];
        }
        P.ref[p] = P.refs[0][p];
    }
    P.mb = e->d_mb; P.levels = e->d_levels; P.mvd = e->d_mvd; P.mvq = e->d_mvq; P.aux = e->d_aux; P.me_cost = e->d_me_cost; P.me_total = e->d_me_total; P.pmv = e->d_pmv;
    P.st_y = e->st_y; P.st_c = e->st_c; P.st_mb = e->nmb;
    // slices that are bands of whole rows run as independent wavefronts; any other shape: one wavefront over the picture (what may
    // be used for prediction is in mbavail either way)
    P.sl.rows = pic.slice_rows > 0 ? pic.slice_rows : e->mbh;
    P.sl.inv = P.sl.rows > 1 ? (unsigned)(0x100000000ull / (unsigned)P.sl.rows) + 1u : 0u;
    P.band.row0 = 0; P.band.rows = e->mbh;
    P.mbdiv.inv = e->mbw > 1 ? (unsigned)(0x100000000ull / (unsigned)e->mbw) + 1u : 0u;
    e->pic_serial = e->pic_serial == 0xFFFFFFFFu ? 1u : e->pic_serial + 1u;
    P.anypcm = e->d_anypcm; P.anyintra = e->d_anyintra; P.pic_serial = e->pic_serial;
    fill_quant(P.qy, pic.qp);                 // (the reconstruction kernels scale with the macroblock's own QP: mbqp)
    fill_quant(P.qc, h_chroma_qp[pic.qp]);
    P.mbqp = d->d_mbqp; P.cqo_cb = pic.cqo[0]; P.cqo_cr = pic.cqo[1]; P.mv4 = d->d_mv4; P.refq = d->d_refq; P.mbavail = d->d_mbavail;
    {   // the flags the loop filter launches look at: intra macroblocks present (bS 3 / 4 form); I_PCM never switches the filter off here
        const unsigned flags[2] = {0u, pic.has_intra ? e->pic_serial : 0u};
        DHIP(d, hipMemcpyAsync(e->d_anypcm, &flags[0], sizeof(unsigned), hipMemcpyHostToDevice, st));
        DHIP(d, hipMemcpyAsync(e->d_anyintra, &flags[1], sizeof(unsigned), hipMemcpyHostToDevice, st));
    }
    DHIP(d, hipEventRecord(d->up_done[d->buf], st));   // every copy out of the parser's buffer has been queued
    d->up_pending[d->buf] = true;
    Slot& S = e->slots[0];
    if (pic.has_inter) {
        hipLaunchKernelGGL(k_dec_inter, dim3(e->nmb, 1), dim3(64), 0, st, P);
        hipLaunchKernelGGL(k_dec_resid, dim3((e->nmb + 3) / 4, 1), dim3(64), 0, st, P);
    }
    if (pic.has_intra) {
        IntraRowParams R{};
        R.p = P; R.handoff = e->d_handoff; R.st_handoff = e->st_handoff; R.err = S.h_err;
        e->serial = e->serial == 0xFFFFFFFFu ? 1 : e->serial + 1;
        R.serial = e->serial;
        R.npic = 1;
        hipLaunchKernelGGL(k_pintra_rows<true>, dim3(e->mbh, 1), dim3(64), 0, st, R);
    }
    if (pic.deblock_idc != 1) {
        // disable_deblocking_filter_idc 0 filters the edges between slices too: the filter then sees one slice
        SliceRows dsl = P.sl;
        // (slices of any other shape than bands: k_dec_bs has zeroed the strengths of the edges between them where idc 2 says so)
        if (pic.deblock_idc == 0 || pic.slice_rows < 0) { dsl.rows = e->mbh; dsl.inv = e->mbh > 1 ? (unsigned)(0x100000000ull / (unsigned)e->mbh) + 1u : 0u; }
        e->serial = e->serial == 0xFFFFFFFFu ? 1 : e->serial + 1;
        const unsigned db_serial = e->serial;
        {   // vectors per 4x4 block, references per quadrant, slice edges from the availability bits
            DecBsParams B{};
            B.mb = e->d_mb; B.mv4 = d->d_mv4; B.refq = d->d_refq; B.bs = (uint8_t*)e->d_bs; B.mbw = e->mbw; B.nmb = e->nmb; B.mbdiv = P.mbdiv;
            B.mbavail = d->d_mbavail; B.across = pic.deblock_idc == 0;
            hipLaunchKernelGGL(k_dec_bs, dim3((e->nmb + 1) / 2, 1), dim3(64), 0, st, B, e->d_anybs, db_serial);
        }
        DbParams D{};
        for (int p = 0; p < 3; p++) D.pl[p] = e->d_planes[cur][p];
        D.mb = e->d_mb; D.cw = e->cw; D.ch = e->ch; D.mbw = e->mbw; D.mbh = e->mbh; D.sl = dsl; D.bs = (const uint8_t*)e->d_bs;
        D.mbqp = d->d_mbqp; D.oa = pic.filter_oa; D.ob = pic.filter_ob; D.cqo_cb = pic.cqo[0]; D.cqo_cr = pic.cqo[1];
        const int qp = pic.qp, qpc = h_chroma_qp[qp];
        D.alpha_y = h_alpha[qp]; D.beta_y = h_beta[qp]; D.alpha_c = h_alpha[qpc]; D.beta_c = h_beta[qpc];
        for (int i = 0; i < 3; i++) { D.tc0_y[i] = h_tc0[qp][i]; D.tc0_c[i] = h_tc0[qpc][i]; }
        DbRowParams R{};
        R.npic = 1;
        R.d = D; R.handoff = e->d_handoff; R.err = S.h_err;
        R.st_y = e->st_y; R.st_c = e->st_c; R.st_handoff = e->st_handoff; R.st_mb = e->nmb;
        R.serial = db_serial; R.row0 = 0;
        R.bs = e->d_bs; R.anybs = e->d_anybs;
        R.anypcm = e->d_anypcm; R.anyintra = e->d_anyintra; R.pic_serial = e->pic_serial;
        // one_qp: the per-picture thresholds above are every edge's (the encoder's own streams); else per edge from mbqp
        if (!pic.one_qp) {
            R.need_intra = 0;
            if (pic.has_intra) hipLaunchKernelGGL((k_deblock_rows<true, true>), dim3(e->mbh, 1), dim3(64), 0, st, R);
            else hipLaunchKernelGGL((k_deblock_rows<false, true>), dim3(e->mbh, 1), dim3(64), 0, st, R);
        } else if (!pic.has_inter) { R.need_intra = 0; hipLaunchKernelGGL(k_deblock_rows<true>, dim3(e->mbh, 1), dim3(64), 0, st, R); }
        else {
            R.need_intra = -1; hipLaunchKernelGGL(k_deblock_rows<false>, dim3(e->mbh, 1), dim3(64), 0, st, R);
            R.need_intra = 1; hipLaunchKernelGGL(k_deblock_rows<true>, dim3(e->mbh, 1), dim3(64), 0, st, R);
        }
    }
    DHIP(d, hipGetLastError());
    d->busy = true;
    return MI355X_H264_OK;
}

void* pinned_alloc(size_t n)
{
    void* p = nullptr;
    return hipHostMalloc(&p, n, hipHostMallocPortable) == hipSuccess ? p : nullptr;
}
void pinned_free(void* p) { (void)hipHostFree(p); }

}  // namespace

extern "C" {

int mi355x_h264_dec_create(int device, mi355x_h264_decoder** out)
{
    if (!out) return MI355X_H264_E_ARG;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return MI355X_H264_E_NODEVICE;
    mi355x_h264_decoder* d = new (std::nothrow) mi355x_h264_decoder();
    if (!d) return MI355X_H264_E_NOMEM;
    d->device = device;
    if (hipSetDevice(device) != hipSuccess || hipEventCreateWithFlags(&d->up_done[0], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&d->up_done[1], hipEventDisableTiming) != hipSuccess) {
        delete d;
        return MI355X_H264_E_HIP;
    }
    // the parsers' per-macroblock arrays from now on: pinned (asynchronous uploads); MI355X_H264_DEC_PAGEABLE=1 keeps malloc (measurements)
    if (!getenv("MI355X_H264_DEC_PAGEABLE")) h264dec::HostMem::use(pinned_alloc, pinned_free);
    *out = d;
    return MI355X_H264_OK;
}

void mi355x_h264_dec_destroy(mi355x_h264_decoder* d)
{
    if (!d) return;
    (void)hipSetDevice(d->device);
    if (d->eng) { (void)dec_wait(d); }
    for (int k = 0; k < 2; k++)
        if (d->up_done[k]) { if (d->up_pending[k]) (void)hipEventSynchronize(d->up_done[k]); (void)hipEventDestroy(d->up_done[k]); }
    if (d->eng) mi355x_h264_destroy(d->eng);
    if (d->d_mbqp) (void)hipFree(d->d_mbqp);
    if (d->d_mv4) (void)hipFree(d->d_mv4);
    if (d->d_refq) (void)hipFree(d->d_refq);
    if (d->d_mbavail) (void)hipFree(d->d_mbavail);
    if (d->d_lv8) (void)hipFree(d->d_lv8);
    if (d->d_big) (void)hipFree(d->d_big);
    delete d;
}

const char* mi355x_h264_dec_last_error(const mi355x_h264_decoder* d) { return d ? d->err : "no decoder"; }

static int dec_decode_unit(mi355x_h264_decoder* d, const uint8_t* au, size_t len, int* got_picture)
{
    if (!d || !au) return MI355X_H264_E_ARG;
    if (got_picture) *got_picture = 0;
    d->err[0] = 0;
    if (hipSetDevice(d->device) != hipSuccess) return dfail(d, MI355X_H264_E_HIP, "hipSetDevice");
    // parse into the buffer the picture in flight does NOT come from (its uploads, two pictures back, have long finished)
    const int k = d->buf ^ 1;
    if (d->up_pending[k]) { DHIP(d, hipEventSynchronize(d->up_done[k])); d->up_pending[k] = false; }
    d->parser.select(k);
    const double t0 = now_ms();
    const int rc = d->parser.parse_access_unit(au, len, false);   // (the picture enters the parser's reference list below, once launched)
    const double t1 = now_ms();
    d->parse_ms += t1 - t0;
    if (rc <= 0) d->parser.select(d->buf);   // nothing to launch: picture() stays the last good one
    if (rc < 0) return dfail(d, MI355X_H264_E_STREAM, "%s", d->parser.error().c_str());
    if (rc == 0) return MI355X_H264_OK;
    {   // the picture in flight must be out of the way before this one is launched (one picture of look-ahead, and its
        // time-out flag is checked here)
        const int wrc = dec_wait(d);
        if (wrc != MI355X_H264_OK) return wrc;
    }
    d->buf = k;
    const h264dec::Picture& pic = d->parser.picture();
    const h264dec::Sps& sps = d->parser.sps();
    if (!d->eng || d->mbw != pic.mbw || d->mbh != pic.mbh) {
        if (!pic.idr) return dfail(d, MI355X_H264_E_STREAM, "the stream must start with an IDR picture");
        if (d->eng) { mi355x_h264_destroy(d->eng); d->eng = nullptr; }
        mi355x_h264_config cfg;
        mi355x_h264_default_config(&cfg);
        cfg.width = 16 * pic.mbw; cfg.height = 16 * pic.mbh; cfg.refs = 3; cfg.device = d->device; cfg.batch = 1;
        const int crc = mi355x_h264_create(&cfg, &d->eng);
        if (crc != MI355X_H264_OK) return dfail(d, crc, "engine for %dx%d macroblocks could not be created", pic.mbw, pic.mbh);
        d->mbw = pic.mbw; d->mbh = pic.mbh; d->have_refs = 0; d->last = -1;
        if (d->d_mbqp) { (void)hipFree(d->d_mbqp); d->d_mbqp = nullptr; }
        if (d->d_mv4) { (void)hipFree(d->d_mv4); d->d_mv4 = nullptr; }
        if (d->d_refq) { (void)hipFree(d->d_refq); d->d_refq = nullptr; }
        if (d->d_mbavail) { (void)hipFree(d->d_mbavail); d->d_mbavail = nullptr; }
        if (d->d_lv8) { (void)hipFree(d->d_lv8); d->d_lv8 = nullptr; }
        const size_t n = (size_t)pic.mbw * pic.mbh;
        if (hipMalloc((void**)&d->d_mbqp, n) != hipSuccess || hipMalloc((void**)&d->d_mv4, n * 64) != hipSuccess || hipMalloc((void**)&d->d_refq, n * 4) != hipSuccess ||
            hipMalloc((void**)&d->d_mbavail, n) != hipSuccess || hipMalloc((void**)&d->d_lv8, n * LV_STRIDE) != hipSuccess)
            return dfail(d, MI355X_H264_E_NOMEM, "hipMalloc (per-macroblock decoder arrays)");
    }
    d->width = pic.width; d->height = pic.height; d->crop_x = 2 * sps.crop_l; d->crop_y = 2 * sps.crop_t;
    d->max_refs = std::max(1, sps.max_refs);
    if (pic.idr) d->have_refs = 0;
    if (pic.has_inter && (d->have_refs < 1 || pic.num_ref_active > d->have_refs))
        return dfail(d, MI355X_H264_E_STREAM, "a P picture refers to %d reference pictures, %d are held", pic.num_ref_active, d->have_refs);
    for (int r = 0; pic.has_inter && r < pic.num_ref_active && r < 3; r++)
        if (pic.ref_age[r] < 0 || pic.ref_age[r] >= d->have_refs) return dfail(d, MI355X_H264_E_STREAM, "reference list entry %d is not a held picture", r);
    int src = dec_submit(d, pic);
    static const bool no_lookahead = getenv("MI355X_H264_DEC_SYNC") != nullptr;   // (measurements: wait for every picture before returning)
    if (src == MI355X_H264_OK && no_lookahead) src = dec_wait(d);
    d->gpu_ms += now_ms() - t1;
    if (src != MI355X_H264_OK) return src;
    d->last = d->eng->cur;
    d->parser.commit();   // parser and ring take the picture in together
    if (pic.is_ref) {   // sliding window (8.2.5.3)
        d->eng->cur = (d->eng->cur + 1) % d->eng->nbuf;
        d->have_refs = std::min(d->have_refs + 1, std::min(d->max_refs, d->eng->nrefs));
    }
    d->pictures++;
    if (got_picture) *got_picture = 1;
    return MI355X_H264_OK;
}


// The C entry point: no exception leaves it (the parser's arrays are std::vectors over pinned memory: an allocation failure
// arrives as std::bad_alloc), and an access unit that is refused at ANY stage - parser, stream checks, allocation, launch, the
// time-out of the picture in flight - may have been a reference picture: parser and ring then drop their reference pictures
// together, so that every P picture is refused until the next IDR picture instead of being predicted from the wrong slot.
int mi355x_h264_dec_decode(mi355x_h264_decoder* d, const uint8_t* au, size_t len, int* got_picture)
{
    if (!d || !au) return MI355X_H264_E_ARG;
    int rc;
    try {
        rc = dec_decode_unit(d, au, len, got_picture);
    } catch (const std::bad_alloc&) {
        rc = dfail(d, MI355X_H264_E_NOMEM, "out of host memory while parsing the access unit");
    } catch (const std::exception& ex) {
        rc = dfail(d, MI355X_H264_E_NOMEM, "access unit refused: %s", ex.what());
    }
    if (rc != MI355X_H264_OK) {
        d->have_refs = 0;
        d->parser.lose_refs();
        if (got_picture) *got_picture = 0;
    }
    return rc;
}

int mi355x_h264_dec_sync(mi355x_h264_decoder* d)
{
    if (!d) return MI355X_H264_E_ARG;
    if (!d->eng) return MI355X_H264_OK;
    if (hipSetDevice(d->device) != hipSuccess) return dfail(d, MI355X_H264_E_HIP, "hipSetDevice");
    return dec_wait(d);
}

int mi355x_h264_dec_picture_info(const mi355x_h264_decoder* d, int* width, int* height, int* coded_width, int* coded_height)
{
    if (!d || d->last < 0) return MI355X_H264_E_ARG;
    if (width) *width = d->width;
    if (height) *height = d->height;
    if (coded_width) *coded_width = 16 * d->mbw;
    if (coded_height) *coded_height = 16 * d->mbh;
    return MI355X_H264_OK;
}

// the last decoded picture, cropped, as tight I420 (Y, U, V); to_device: dst is device memory
static int64_t dec_read(mi355x_h264_decoder* d, void* dst, size_t cap, bool to_device)
{
    if (!d || !dst || d->last < 0) return MI355X_H264_E_ARG;
    const size_t w = (size_t)d->width, h = (size_t)d->height, need = w * h * 3 / 2;
    if (cap < need) return MI355X_H264_E_ARG;
    if (hipSetDevice(d->device) != hipSuccess) return dfail(d, MI355X_H264_E_HIP, "hipSetDevice");
    {
        const int wrc = dec_wait(d);   // the picture asked for may still be in flight
        if (wrc != MI355X_H264_OK) return wrc;
    }
    const mi355x_h264_encoder* e = d->eng;
    uint8_t* o = (uint8_t*)dst;
    const hipMemcpyKind kind = to_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
    for (int p = 0; p < 3; p++) {
        const size_t pw = p ? w / 2 : w, ph = p ? h / 2 : h, pitch = p ? (size_t)e->cw / 2 : (size_t)e->cw;
        const uint8_t* s = e->d_planes[d->last][p] + (size_t)(p ? d->crop_y / 2 : d->crop_y) * pitch + (size_t)(p ? d->crop_x / 2 : d->crop_x);
        if (hipMemcpy2D(o, pw, s, pitch, pw, ph, kind) != hipSuccess) return dfail(d, MI355X_H264_E_HIP, "hipMemcpy2D");
        o += pw * ph;
    }
    return (int64_t)need;
}
int64_t mi355x_h264_dec_read_i420(mi355x_h264_decoder* d, uint8_t* dst, size_t cap) { return dec_read(d, dst, cap, false); }
int64_t mi355x_h264_dec_read_i420_device(mi355x_h264_decoder* d, void* d_dst, size_t cap) { return dec_read(d, d_dst, cap, true); }

// coded-size planes of the last picture (test hook: compared with the oracle decoder's planes)
int64_t mi355x_h264_dec_debug_plane(mi355x_h264_decoder* d, int plane, void* dst, size_t cap)
{
    if (!d || !dst || d->last < 0 || plane < 0 || plane > 2) return MI355X_H264_E_ARG;
    const mi355x_h264_encoder* e = d->eng;
    const size_t n = (size_t)e->cw * e->ch / (plane ? 4 : 1);
    if (cap < n) return MI355X_H264_E_ARG;
    if (hipSetDevice(d->device) != hipSuccess) return MI355X_H264_E_HIP;
    {
        const int wrc = dec_wait(d);
        if (wrc != MI355X_H264_OK) return wrc;
    }
    if (hipMemcpy(dst, e->d_planes[d->last][plane], n, hipMemcpyDeviceToHost) != hipSuccess) return MI355X_H264_E_HIP;
    return (int64_t)n;
}

int mi355x_h264_dec_timing(const mi355x_h264_decoder* d, uint64_t* pictures, double* parse_ms, double* gpu_ms)
{
    if (!d) return MI355X_H264_E_ARG;
    if (pictures) *pictures = d->pictures;
    if (parse_ms) *parse_ms = d->parse_ms;
    if (gpu_ms) *gpu_ms = d->gpu_ms;
    return MI355X_H264_OK;
}

// ---- the host parser alone (no GPU): what it recovered from the last access unit, for the CPU tests ----
struct mi355x_h264_parser { h264dec::Parser p; };
mi355x_h264_parser* mi355x_h264_parser_create(void) { return new (std::nothrow) mi355x_h264_parser(); }
void mi355x_h264_parser_destroy(mi355x_h264_parser* p) { delete p; }
int mi355x_h264_parser_parse(mi355x_h264_parser* p, const uint8_t* au, size_t len)
{
    if (!p || !au) return -1;
    try {
        return p->p.parse_access_unit(au, len);
    } catch (const std::exception& ex) {   // (allocation failure of a per-macroblock array: reported, never thrown through the C ABI)
        p->p.lose_refs();
        p->p.set_error(std::string("out of memory: ") + ex.what());
        return -1;
    }
}
const char* mi355x_h264_parser_error(const mi355x_h264_parser* p) { return p ? p->p.error().c_str() : "no parser"; }
int mi355x_h264_parser_info(const mi355x_h264_parser* p, int32_t* out, int n)
{
    if (!p || !out || n < 12) return -1;
    const h264dec::Picture& c = p->p.picture();
    const int32_t v[20] = {c.mbw, c.mbh, c.width, c.height, c.idr, c.qp, c.slice_rows, c.deblock_idc, c.num_ref_active, c.t8x8_mode, c.has_pcm, c.has_intra | (c.has_inter << 1),
                           c.cqo[0], c.cqo[1], c.filter_oa, c.filter_ob, c.one_qp, c.ref_age[0], c.ref_age[1], c.ref_age[2]};
    const int m = n < 17 ? 12 : (n < 20 ? 17 : 20);   // (a caller with an earlier layout's slots gets those)
    memcpy(out, v, (size_t)m * sizeof(int32_t));
    return m;
}
// what: 0 MbInfo (32 B / macroblock), 1 quadrant vectors (16 B), 2 Intra4x4 modes (16 B), 3 levels (832 B)
int64_t mi355x_h264_parser_read(const mi355x_h264_parser* p, int what, void* dst, size_t cap)
{
    if (!p || !dst) return -1;
    const h264dec::Picture& c = p->p.picture();
    const void* src = nullptr;
    size_t n = 0;
    switch (what) {
        case 0: src = c.mb.data(); n = c.mb.size() * sizeof(h264dec::MbRec); break;
        case 1: src = c.mvq.data(); n = c.mvq.size() * sizeof(int16_t); break;
        case 2: src = c.aux.data(); n = c.aux.size(); break;
        case 3: {   // the level lists as int16 (what k_dec_widen + k_dec_patch make of levels8 + big on the GPU)
            n = c.levels8.size() * sizeof(int16_t);
            if (cap < n) return -1;
            int16_t* o = (int16_t*)dst;
            for (size_t m = 0; m < c.mb.size(); m++) {
                const int8_t* s8 = c.levels8.data() + m * h264dec::L_STRIDE;
                int16_t* d16 = o + m * h264dec::L_STRIDE;
                if (c.mb[m].type == h264dec::T_IPCM) { memset(d16, 0, h264dec::L_STRIDE * sizeof(int16_t)); memcpy(d16, s8, 384); }
                else for (int k = 0; k < h264dec::L_STRIDE; k++) d16[k] = s8[k];
            }
            for (const auto& b : c.big) o[b.idx] = (int16_t)b.val;
            return (int64_t)n;
        }
        case 4: src = c.mbqp.data(); n = c.mbqp.size(); break;
        case 5: src = c.mv4.data(); n = c.mv4.size() * sizeof(int16_t); break;
        case 6: src = c.refq.data(); n = c.refq.size(); break;
        case 7: src = c.mbavail.data(); n = c.mbavail.size(); break;
        default: return -1;
    }
    if (cap < n) return -1;
    memcpy(dst, src, n);
    return (int64_t)n;
}

}  // extern "C"
