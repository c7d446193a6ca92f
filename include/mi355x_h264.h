/*
 * include/mi355x_h264.h -- the drop-in boundary: a plain C ABI over the
 * hand-written HIP (gfx950) H.264 encode path.  No C++ or torch types cross it.
 *
 * Each entry point replaces one use of the OpenH264 vtable that the
 * reference's adapter makes (paths relative to /root/reference):
 *
 *   mi355x_h264_create      <- WelsCreateSVCEncoder + ISVCEncoder::InitializeExt
 *                              + SetOption(ENCODER_OPTION_DATAFORMAT)
 *                              video_codec/VideoEncoderOpenH264.cpp:142, :257, :262
 *                              (decl vendor/openh264/codec_api.h:286, :545)
 *   mi355x_h264_encode      <- ISVCEncoder::EncodeFrame
 *                              video_codec/VideoEncoderOpenH264.cpp:344
 *                              (decl vendor/openh264/codec_api.h:309); the
 *                              SSourcePicture plane/stride triple of :354-365
 *                              becomes the y/u/v + stride arguments, and the
 *                              SFrameBSInfo consumption of :349-350 becomes
 *                              (*out, *out_len)
 *   mi355x_h264_force_idr   <- ISVCEncoder::ForceIntraFrame(true)
 *                              video_codec/VideoEncoderOpenH264.cpp:408
 *                              (decl vendor/openh264/codec_api.h:323)
 *   mi355x_h264_destroy     <- ISVCEncoder::Uninitialize + WelsDestroySVCEncoder
 *                              video_codec/VideoEncoderOpenH264.cpp:382-383
 *
 * The *_device / *_batch entry points are this build's additions for callers
 * whose frames are already resident in HBM (bench.py, multi-stream sharding);
 * the mi355x_h264_debug_* ones expose intermediate device buffers so tests can
 * compare every stage against the CPU oracle.
 *
 * Ownership mirrors the reference: input is caller-owned and
 * only read during the call; the output bitstream lives in encoder-owned
 * (pinned host) memory and stays valid until the next encode / destroy on the
 * same handle.  All functions return 0 on success or a negative MI355X_H264_E_*.
 * Nothing here falls back to a CPU encoder: if no HIP device is usable,
 * mi355x_h264_create fails with MI355X_H264_E_NODEVICE.
 */
#ifndef MI355X_H264_H
#define MI355X_H264_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI355X_H264_ABI_VERSION 3

enum {
    MI355X_H264_OK = 0,
    MI355X_H264_E_ARG = -1,      /* bad argument / unsupported configuration        */
    MI355X_H264_E_NODEVICE = -2, /* no usable HIP device (never a silent CPU path)  */
    MI355X_H264_E_HIP = -3,      /* a HIP runtime call failed (see last_error)      */
    MI355X_H264_E_NOMEM = -4,
    MI355X_H264_E_OVERFLOW = -5, /* a slice coded to more than its payload buffer (twice its luma bytes; only synthetic
                                  * noise at the lowest QPs does): the picture is refused, nothing is written past the
                                  * buffer, and the next picture is coded as an IDR (the refused one is missing from
                                  * the stream and must not be referred to)                                            */
    MI355X_H264_E_INTERNAL = -6
};

enum { MI355X_H264_FRAME_IDR = 1, MI355X_H264_FRAME_P = 3 }; /* EVideoFrameType values, codec_def.h:70,72 */

enum { MI355X_H264_RC_FIXED_QP = 0, MI355X_H264_RC_BITRATE = 1 };
enum { MI355X_H264_INPUT_I420 = 0, MI355X_H264_INPUT_NV12 = 1 };
enum { MI355X_H264_SEARCH_EXHAUSTIVE = 0, MI355X_H264_SEARCH_SEEDED = 1 };

typedef struct mi355x_h264_config {
    uint32_t struct_size;    /* sizeof(mi355x_h264_config), for ABI growth          */
    int32_t width, height;   /* iPicWidth/iPicHeight (ref :235-236); even, 16..4096 */
    int32_t fps;             /* fMaxFrameRate (ref :241): 30 or 60                  */
    int32_t bitrate;         /* iTargetBitrate (ref :239), used when rc_mode == 1   */
    int32_t gop;             /* uiIntraPeriod (ref :242)                            */
    int32_t profile_idc;     /* uiProfileIdc (ref :248-254): 66, 77 or 100          */
    int32_t rc_mode;         /* MI355X_H264_RC_*; the reference preset is BITRATE   */
    int32_t qp;              /* picture QP for FIXED_QP (10..51); initial QP else   */
    int32_t device;          /* HIP device ordinal                                  */
    int32_t disable_deblock; /* iLoopFilterDisableIdc (ref :295), 0 = filter on     */
    int32_t batch;           /* closed GOPs (or independent streams) encoded in lockstep by one instance,
                              * 1..64; > 1 is driven through mi355x_h264_encode_gops_device only      */
    int32_t input_format;    /* layout of pictures handed over in DEVICE memory (encode_device, encode_batch_device,
                              * encode_gops_device): MI355X_H264_INPUT_I420 (default) or MI355X_H264_INPUT_NV12     */
    int32_t slices;          /* 0 / 1: one slice per picture (the reference preset, SM_SINGLE_SLICE, ref :247).  n > 1:
                              * n bands of ceil(rows / n) macroblock rows (at least two rows each), one slice NAL
                              * unit per band, disable_deblocking_filter_idc = 2: the bands of a picture
                              * do not depend on one another, their row wavefronts run side by side                  */
    int32_t band_index;      /* slice bands of ONE picture over several GPUs (BASELINE.json configs[4]): */
    int32_t band_count;      /* with band_count > 1 this instance codes only its share of the `slices` slices (a contiguous
                              * run of whole slices, index band_index of band_count); see mi355x_h264_band_*              */
    int32_t refs;            /* iNumRefFrame (ref :290: 1).  0 / 1: one reference frame; 2, 3: the motion search covers the last
                              * `refs` pictures and ref_idx_l0 is coded (BASELINE.json configs[4] asks for 3)                  */
    int32_t search;          /* integer motion search (ABI 3).  MI355X_H264_SEARCH_EXHAUSTIVE: every position of +-16 samples around the
                              * co-located macroblock.  MI355X_H264_SEARCH_SEEDED (what mi355x_h264_default_config sets): the
                              * macroblock's vector in the previous picture, rounded to integer samples, is taken as the integer
                              * winner when it is a strict local minimum of the search cost among its eight integer neighbours -
                              * what a predictor-seeded search does - and the exhaustive pass runs whenever that
                              * test fails.  Costs 0.0 .. 0.2 % bytes at equal PSNR on five of six test contents, 1.3 % at worst
                              * (DESIGN.md section 3); both forms are bit-exact against the oracle's.                          */
} mi355x_h264_config;

typedef struct mi355x_h264_encoder mi355x_h264_encoder;

int mi355x_h264_abi_version(void);
void mi355x_h264_default_config(mi355x_h264_config *cfg);
int mi355x_h264_create(const mi355x_h264_config *cfg, mi355x_h264_encoder **out);
void mi355x_h264_destroy(mi355x_h264_encoder *enc);

/* Encode one I420 picture given as host pointers.  *out receives an Annex-B
 * access unit (SPS+PPS+IDR slice, or one P slice) in encoder-owned memory. */
int mi355x_h264_encode(mi355x_h264_encoder *enc, const uint8_t *y, int y_stride, const uint8_t *u,
                       int u_stride, const uint8_t *v, int v_stride, uint8_t **out, uint32_t *out_len,
                       int *frame_type);

/* Same, with the picture already in device memory as one tightly packed I420
 * buffer (Y, then U, then V; width*height*3/2 bytes). */
int mi355x_h264_encode_device(mi355x_h264_encoder *enc, const void *d_i420, uint8_t **out,
                              uint32_t *out_len, int *frame_type);

/* NV12 ingest (BASELINE.json configs[2]): Y plane followed by one interleaved
 * UV plane.  OpenH264 itself only takes I420 (ref :256,:262); here the kernels read the interleaved
 * chroma rows directly (one 8-byte load + byte permute per four samples): no conversion pass, no extra
 * HBM traffic, and the host never touches the samples.  uv_stride in bytes. */
int mi355x_h264_encode_nv12(mi355x_h264_encoder *enc, const uint8_t *y, int y_stride, const uint8_t *uv,
                            int uv_stride, uint8_t **out, uint32_t *out_len, int *frame_type);
/* tightly packed NV12 in device memory: Y (width*height) then UV (width*height/2) */
int mi355x_h264_encode_nv12_device(mi355x_h264_encoder *enc, const void *d_nv12, uint8_t **out,
                                   uint32_t *out_len, int *frame_type);

/* RGBA ingest (the reference's own interface names the layout only on its decoder side,
 * video_decoder/include/VideoDecoder.h:40-47 PIXEL_FORMAT_RGBA_8888 - no reference ENCODER path takes it, so the conversion
 * is this library's definition, stated in oracle/h264_rgba.c): 8-bit R, G, B, A bytes per sample, alpha ignored.  One
 * conversion kernel writes the I420 picture the encoder then reads: BT.601 studio swing in the usual integer form,
 *   Y = ((66 R + 129 G + 25 B + 128) >> 8) + 16,
 *   Cb = ((-38 r - 74 g + 112 b + 128) >> 8) + 128,  Cr = ((112 r - 94 g - 18 b + 128) >> 8) + 128
 * with r, g, b the rounded mean of the four samples of a 2x2 block.  stride in bytes (>= 4 * width).  Batch-1 encoders. */
int mi355x_h264_encode_rgba(mi355x_h264_encoder *enc, const uint8_t *rgba, int stride, uint8_t **out,
                            uint32_t *out_len, int *frame_type);
/* tightly packed RGBA in device memory (4 * width bytes per row) */
int mi355x_h264_encode_rgba_device(mi355x_h264_encoder *enc, const void *d_rgba, uint8_t **out,
                                   uint32_t *out_len, int *frame_type);

/* Encode `count` device-resident pictures back to back; picture i starts at
 * d_frames + i*frame_stride_bytes.  Access units are appended to host_out
 * (capacity out_cap); sizes[i] receives the byte length of picture i.  The
 * host-side finishing of picture i overlaps the GPU work of picture i+1. */
int mi355x_h264_encode_batch_device(mi355x_h264_encoder *enc, const void *d_frames,
                                    size_t frame_stride_bytes, int count, uint8_t *host_out,
                                    size_t out_cap, uint32_t *sizes, size_t *total_len);

/* Lockstep encode of `batch` closed GOPs (config.batch): picture t of GOP g is read from
 * d_frames + g*gop_stride + t*frame_stride (device memory, tight I420).  Every kernel runs once per
 * picture index over all GOPs (grid.y = batch), so the launches are `batch` times larger instead of
 * `batch` times more numerous.  The first picture of every GOP is an IDR; idr_pic_id of GOP g is
 * next + g*step (set_idr_pic_id), so consecutive GOPs concatenate to the serial stream.  GOP g's access
 * units are appended at host_out + g*out_cap_per_gop; sizes[g*frames_per_gop + t], gop_bytes[g]. */
int mi355x_h264_encode_gops_device(mi355x_h264_encoder *enc, const void *d_frames, size_t frame_stride,
                                   size_t gop_stride, int frames_per_gop, uint8_t *host_out,
                                   size_t out_cap_per_gop, uint32_t *sizes, size_t *gop_bytes);

/* ---- slice bands of one picture on several encoder instances / GPUs (config.band_count > 1) ----
 * Every instance is created with the full picture geometry and the same `slices`, and is handed the full source picture
 * (only its band's rows are read).  mi355x_h264_encode* then returns this band's slice NAL units only (SPS/PPS with
 * band 0); the access unit is the concatenation over band_index.  Motion search and compensation of the next picture
 * reach up to 19 sample rows beyond the band, so after every picture the neighbours swap two macroblock rows of
 * reconstruction (the host class does it with one RCCL send/recv pair per neighbour over xGMI):
 *     export(enc, 0, buf) -> send to band_index-1;  export(enc, 1, buf) -> send to band_index+1
 *     import(enc, 0, buf) <- received from band_index-1 (its bottom rows);  import(enc, 1, buf) <- from band_index+1
 * buf: halo_bytes (band_info) of device memory - or of host memory, when the transport between the ranks is a host one.
 * The result equals the stream ONE instance with the same `slices` makes. */
int mi355x_h264_band_info(const mi355x_h264_encoder *enc, int *first_row, int *rows, int *first_slice, int *slices,
                          size_t *halo_bytes);
int mi355x_h264_band_halo_export(mi355x_h264_encoder *enc, int edge, void *d_dst);
int mi355x_h264_band_halo_import(mi355x_h264_encoder *enc, int edge, const void *d_src);

int mi355x_h264_force_idr(mi355x_h264_encoder *enc);
/* picture QP (10..51) for the pictures that follow; the hook the rate controller
 * of the host class drives (RC_BITRATE_MODE, ref :274) */
int mi355x_h264_set_qp(mi355x_h264_encoder *enc, int qp);
/* idr_pic_id of the next IDR picture and its increment per IDR (mod 256).  With closed GOPs
 * sharded over several encoder instances / GPUs (instance i of G: next = i, step = G) the
 * concatenated output equals the serial stream byte for byte. */
int mi355x_h264_set_idr_pic_id(mi355x_h264_encoder *enc, int next, int step);
/* Scene-change statistic of the last finished picture, one value per batch item: the sum over the
 * macroblocks that went through motion search of min(final SATD-based motion cost, 16383); 0 for IDR
 * pictures.  The plugin class compares it with a threshold and re-codes the picture as IDR
 * (bEnableSceneChangeDetect = 1 in the reference preset, VideoEncoderOpenH264.cpp:283). */
int mi355x_h264_last_me_cost(const mi355x_h264_encoder *enc, uint32_t *cost);
const char *mi355x_h264_last_error(const mi355x_h264_encoder *enc);

/* coded picture geometry (multiples of 16) */
int mi355x_h264_coded_width(const mi355x_h264_encoder *enc);
int mi355x_h264_coded_height(const mi355x_h264_encoder *enc);

/* ---- test / measurement hooks ---- */
enum {
    MI355X_H264_DBG_RECON_Y = 0,   /* deblocked reconstruction = next reference, coded size */
    MI355X_H264_DBG_RECON_U = 1,
    MI355X_H264_DBG_RECON_V = 2,
    MI355X_H264_DBG_MBINFO = 3,    /* 32 B per macroblock, layout of h264 mbinfo below      */
    MI355X_H264_DBG_LEVELS = 4,    /* 416 int16 per macroblock                              */
    MI355X_H264_DBG_PRE_Y = 5,     /* reconstruction before the loop filter (needs          */
    MI355X_H264_DBG_PRE_U = 6,     /*  mi355x_h264_debug_keep_pre(enc, 1))                  */
    MI355X_H264_DBG_PRE_V = 7,
    MI355X_H264_DBG_MBAUX = 8,     /* 16 B per macroblock: Intra4x4PredMode of the blocks of type-4 macroblocks   */
    MI355X_H264_DBG_MVQ = 9        /* 8 int16 per macroblock: (x, y) vectors of the four 8x8 quadrants of an inter macroblock */
};
int mi355x_h264_debug_keep_pre(mi355x_h264_encoder *enc, int on);
/* copies the named device buffer of the last encoded picture (batch item 0) to dst; returns bytes or <0 */
int64_t mi355x_h264_debug_read(mi355x_h264_encoder *enc, int what, void *dst, size_t cap);

/* ---- streams (ABI 3): many encoders of one geometry, one picture per call each, sharing one engine ----
 * The reference's operating mode is one VideoEncoder object per cloud-phone stream, each driven by its own thread with one
 * EncodeOneFrame per tick (/root/reference/video_codec/VideoEncoderOpenH264.cpp:304-352).  A stream is such an encoder whose
 * pictures are coded together with the pictures other streams of the same geometry (size, fps, profile, slices, filter switch,
 * device) deliver at about the same time: ONE lockstep launch sequence per step instead of one per stream, every picture with
 * its own QP, picture type, frame_num and reference pictures.  mi355x_h264_stream_encode is synchronous and may be called from
 * one thread per stream concurrently; the output is bit-for-bit what mi355x_h264_create / mi355x_h264_encode with the same
 * config and the same QP sequence produce (tests/test_gpu_streams.py).  One reference picture (refs <= 1), host I420 input.
 * *out stays valid until the stream's next encode / close.  MI355X_H264_HUB_ITEMS (default 32) streams share an engine;
 * MI355X_H264_HUB_WINDOW_US (default 200): how long a step waits for pictures that are already being uploaded. */
typedef struct mi355x_h264_stream mi355x_h264_stream;
int mi355x_h264_stream_open(const mi355x_h264_config *cfg, mi355x_h264_stream **out);
void mi355x_h264_stream_close(mi355x_h264_stream *s);
int mi355x_h264_stream_encode(mi355x_h264_stream *s, const uint8_t *y, int y_stride, const uint8_t *u, int u_stride,
                              const uint8_t *v, int v_stride, uint8_t **out, uint32_t *out_len, int *frame_type);
int mi355x_h264_stream_set_qp(mi355x_h264_stream *s, int qp);            /* as mi355x_h264_set_qp            */
int mi355x_h264_stream_force_idr(mi355x_h264_stream *s);                 /* as mi355x_h264_force_idr         */
int mi355x_h264_stream_set_idr_pic_id(mi355x_h264_stream *s, int next);  /* idr_pic_id of the next IDR       */
int mi355x_h264_stream_last_me_cost(const mi355x_h264_stream *s, uint32_t *cost);   /* as mi355x_h264_last_me_cost */
const char *mi355x_h264_stream_last_error(const mi355x_h264_stream *s);
int mi355x_h264_stream_coded_width(const mi355x_h264_stream *s);
int mi355x_h264_stream_coded_height(const mi355x_h264_stream *s);
/* reconstruction plane (MI355X_H264_DBG_RECON_Y / _U / _V) of the stream's last picture; returns bytes or < 0 */
int64_t mi355x_h264_stream_debug_read(mi355x_h264_stream *s, int what, void *dst, size_t cap);
/* how the stream's engine has been batching: steps launched, pictures coded, largest step, streams open on it */
int mi355x_h264_stream_hub_stats(const mi355x_h264_stream *s, uint64_t *steps, uint64_t *pictures, uint64_t *max_batch,
                                 int *open_streams);

/* per-kernel device time accumulated since the last reset, measured with HIP
 * events on the encoder's own stream */
enum {
    MI355X_H264_K_ME = 0,       /* motion search (SAD integer + SATD sub-pel)      */
    MI355X_H264_K_PMB = 1,      /* residual + fDCT + quant + dequant + iDCT + recon (k_tq / k_tq8; the MC moved into the search) */
    MI355X_H264_K_INTRA = 2,    /* Intra16x16 wavefront                            */
    MI355X_H264_K_CAVLC = 3,    /* entropy coding + packing                        */
    MI355X_H264_K_DEBLOCK = 4,  /* loop filter wavefront                           */
    MI355X_H264_K_COUNT = 5
};
typedef struct mi355x_h264_stats {
    double ms[MI355X_H264_K_COUNT];       /* summed device milliseconds           */
    uint64_t launches[MI355X_H264_K_COUNT];
    uint64_t mbs[MI355X_H264_K_COUNT];    /* macroblocks processed by that kernel */
    uint64_t frames;
    /* what the P pictures' macroblocks became (counted on the device by the entropy stage, always on; ABI 3):
     * p_mbs      macroblocks of P pictures;
     * me_searched_mbs  of those, the ones k_me's "nothing left to code" tests did NOT settle - the search, the
     *            sub-sample refinement and the prediction write ran for them;
     * tq_coded_mbs     of those, the ones k_tq / k_tq8 coded (searched, and not handed to the intra pass): the
     *            macroblocks the transform kernel loads and stores samples and levels for.  bench.py prices the
     *            roofline of the two kernels on these counts, not on the launch geometry. */
    uint64_t p_mbs, me_searched_mbs, tq_coded_mbs;
} mi355x_h264_stats;
int mi355x_h264_stats_enable(mi355x_h264_encoder *enc, int on);
int mi355x_h264_stats_read(mi355x_h264_encoder *enc, mi355x_h264_stats *out, int reset);

#ifdef __cplusplus
}
#endif
#endif
