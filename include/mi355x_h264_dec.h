/*
 * include/mi355x_h264_dec.h -- C ABI of the MI355X H.264 decoder peer.
 *
 * It serves a VideoDecoder peer of the reference's NETINT adapter (interface /root/reference/video_decoder/include/
 * VideoDecoder.h:83, call sites VideoDecoderNetint.cpp: SendStreamData :568, RetrieveFrameData :640): one access unit in,
 * one picture out, output order = decoding order (the streams the reference's encoder side produces have no B pictures).
 *
 * Division of labour: CAVLC slice data is a serial code and is parsed on the host
 * (media_amd/csrc/h264_parse.h); motion compensation, inverse transforms, intra prediction and the loop filter run on the
 * GPU with the encoder's own reconstruction kernels.  No device -> MI355X_H264_E_NODEVICE; there is no CPU reconstruction.
 *
 * Supported: baseline / main / high streams with CAVLC, frame macroblocks, I and P slices, Intra16x16 / Intra4x4 / I_PCM,
 * 16x16 .. 4x4 partitions (every sub_mb_type), up to 3 reference pictures (sliding window, list modification by short-term
 * picture numbers, one index per partition), 4x4 and
 * (inter) 8x8 transform, QP per macroblock
 * (slice_qp_delta, mb_qp_delta), chroma QP index offsets, deblocking filter offsets and idc 0 / 1 / 2 (one set per picture),
 * slices of any shape in raster order (no FMO / ASO).  A stream outside that is refused with MI355X_H264_E_STREAM and a message naming the
 * syntax element; nothing is ever decoded approximately.
 */
#ifndef MI355X_H264_DEC_H
#define MI355X_H264_DEC_H

#include <stddef.h>
#include <stdint.h>
#include "mi355x_h264.h"   /* error codes */

#ifdef __cplusplus
extern "C" {
#endif

#define MI355X_H264_E_STREAM (-7)   /* the access unit is damaged or uses a feature outside the supported set (see last_error) */

typedef struct mi355x_h264_decoder mi355x_h264_decoder;

/* replaces ni_device_session_open / ni_logan_decoder_init_default_params (VideoDecoderNetint.cpp:208-262) */
int mi355x_h264_dec_create(int device, mi355x_h264_decoder **out);
void mi355x_h264_dec_destroy(mi355x_h264_decoder *dec);
const char *mi355x_h264_dec_last_error(const mi355x_h264_decoder *dec);

/* one access unit, Annex B with start codes (SendStreamData, VideoDecoderNetint.cpp:568).  *got_picture = 1 when a picture
 * was decoded (0: the unit held parameter sets / SEI only).  The unit is parsed and the picture LAUNCHED on return, not
 * necessarily finished: the next call parses its access unit while the GPU reconstructs this one (one picture of look-ahead).
 * Every call that looks at the picture (read_i420, debug_plane, sync) waits for it first; a failure of the reconstruction
 * itself (MI355X_H264_E_INTERNAL) is reported by that call or by the next decode. */
int mi355x_h264_dec_decode(mi355x_h264_decoder *dec, const uint8_t *au, size_t len, int *got_picture);
/* wait until the picture launched by the last decode call is complete */
int mi355x_h264_dec_sync(mi355x_h264_decoder *dec);

/* cropped and coded size of the last decoded picture (INDEX_PIC_INFO, VideoDecoder.h:57) */
int mi355x_h264_dec_picture_info(const mi355x_h264_decoder *dec, int *width, int *height, int *coded_width, int *coded_height);

/* the last decoded picture as tight I420 (Y, U, V planes of the cropped size) into host / device memory; returns bytes or < 0
 * (RetrieveFrameData, VideoDecoderNetint.cpp:640, PIXEL_FORMAT_YUV_420P) */
int64_t mi355x_h264_dec_read_i420(mi355x_h264_decoder *dec, uint8_t *dst, size_t cap);
int64_t mi355x_h264_dec_read_i420_device(mi355x_h264_decoder *dec, void *d_dst, size_t cap);

/* ---- test / measurement hooks ---- */
int64_t mi355x_h264_dec_debug_plane(mi355x_h264_decoder *dec, int plane, void *dst, size_t cap);   /* coded-size plane 0..2 */
int mi355x_h264_dec_timing(const mi355x_h264_decoder *dec, uint64_t *pictures, double *parse_ms, double *gpu_ms);

/* the host parser alone (needs no GPU): parse one access unit and read back what it recovered */
typedef struct mi355x_h264_parser mi355x_h264_parser;
mi355x_h264_parser *mi355x_h264_parser_create(void);
void mi355x_h264_parser_destroy(mi355x_h264_parser *p);
int mi355x_h264_parser_parse(mi355x_h264_parser *p, const uint8_t *au, size_t len);   /* 1 picture, 0 none, -1 error */
const char *mi355x_h264_parser_error(const mi355x_h264_parser *p);
/* out[12]: mbw, mbh, width, height, idr, qp (of the first slice), slice_rows (0 = one slice, n > 0 = bands of n rows, -1 = slices
 * of any other shape), deblocking idc, num_ref_idx_active,
 * transform_8x8_mode, has I_PCM, bit 0 has intra | bit 1 has inter; with n >= 17 also: chroma_qp_index_offset,
 * second_chroma_qp_index_offset, FilterOffsetA, FilterOffsetB, 1 = every macroblock has that one QP and no offset applies;
 * with n >= 20 also RefPicList0 entries 0..2 as "reference pictures ago" (0 = the one decoded last; default 0, 1, 2).
 * Returns the number of values written */
int mi355x_h264_parser_info(const mi355x_h264_parser *p, int32_t *out, int n);
/* what: 0 MbInfo (32 B / macroblock, layout of mi355x_h264.h), 1 quadrant vectors (8 int16), 2 Intra4x4 modes (16 B), 3 levels (416 int16),
 * 4 QP_Y (1 B / macroblock; 0 for I_PCM), 5 vectors per 4x4 block (32 int16, raster order), 6 ref_idx_l0 per 8x8 quadrant (4 B; 255 intra),
 * 7 neighbour availability (1 B: bits 0..3 the left, above, above-right, above-left macroblock lies in this slice and was decoded
 *   before; bits 4..7 it may also be used for intra prediction: constrained_intra_pred_flag) */
int64_t mi355x_h264_parser_read(const mi355x_h264_parser *p, int what, void *dst, size_t cap);

#ifdef __cplusplus
}
#endif
#endif
