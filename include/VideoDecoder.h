/*
 * include/VideoDecoder.h -- public plugin surface of libVideoDecoder, source compatible with the reference's
 * /root/reference/video_decoder/include/VideoDecoder.h (enums :10-61, parameter structs :63-81, abstract class
 * VideoDecoder :83-189, the two extern "C" factory functions :191-195).  Same names, same values, same virtual order, so a
 * caller built against the reference header links against this library unchanged (Itanium C++ ABI).
 */
#ifndef VIDEO_DECODER_H
#define VIDEO_DECODER_H

#include <cstdint>
#include <functional>

enum DecoderRetCode : uint32_t {
    VIDEO_DECODER_SUCCESS,
    VIDEO_DECODER_CREATE_FAIL,
    VIDEO_DECODER_INIT_FAIL,
    VIDEO_DECODER_START_FAIL,
    VIDEO_DECODER_DECODE_FAIL,
    VIDEO_DECODER_STOP_FAIL,
    VIDEO_DECODER_DESTROY_FAIL,
    VIDEO_DECODER_RESET_FAIL,
    VIDEO_DECODER_GET_DECODE_PARAMS_FAIL,
    VIDEO_DECODER_SET_DECODE_PARAMS_FAIL,
    VIDEO_DECODER_SET_FUNC_FAIL,
    VIDEO_DECODER_WRITE_OVERFLOW,   /* the input side is full: retrieve a picture first */
    VIDEO_DECODER_READ_UNDERFLOW,   /* no decoded picture is waiting                     */
    VIDEO_DECODER_BAD_PIC_SIZE,     /* the decoded size differs from the configured one  */
    VIDEO_DECODER_EOS
};

enum DecoderPort : uint32_t { IN_PORT, OUT_PORT };

enum MediaStreamFormat : uint32_t { STREAM_FORMAT_AVC, STREAM_FORMAT_HEVC, STREAM_FORMAT_NONE };

enum MediaPixelFormat : uint32_t {
    PIXEL_FORMAT_RGBA_8888,
    PIXEL_FORMAT_YUV_420P,
    PIXEL_FORMAT_FLEX_YUV_420P,
    PIXEL_FORMAT_NV12,
    PIXEL_FORMAT_NV21,
    PIXEL_FORMAT_NONE
};

enum DecodeEventIndex : uint32_t { INDEX_PIC_INFO_CHANGE, INDEX_EVENT_NONE };

enum DecodeParamsIndex : uint32_t { INDEX_PIC_INFO, INDEX_PORT_FORMAT_INFO, INDEX_ALIGN_INFO, INDEX_PARAM_NONE };

struct AlignInfoParams {
    uint32_t widthAlign = 0;
    uint32_t heightAlign = 0;
};

struct PicInfoParams {
    uint32_t width = 0;
    uint32_t height = 0;
    int32_t stride = 0;
    uint32_t scanLines = 0;
    uint32_t cropWidth = 0;
    uint32_t cropHeight = 0;
};

struct PortFormatParams {
    DecoderPort port {};
    int32_t format = 0;
};

class VideoDecoder {
public:
    VideoDecoder() = default;
    virtual ~VideoDecoder() = default;

    /* decType: the kind of stream that will be sent */
    virtual DecoderRetCode CreateDecoder(MediaStreamFormat decType) = 0;
    virtual DecoderRetCode InitDecoder() = 0;
    /* decParams points at the struct that goes with index (PicInfoParams, PortFormatParams, AlignInfoParams) */
    virtual DecoderRetCode SetDecodeParams(DecodeParamsIndex index, void *decParams) = 0;
    virtual DecoderRetCode GetDecodeParams(DecodeParamsIndex index, void *decParams) = 0;
    /* eventCallBack(event, data1, data2): how the decoder reports events (a changed picture size) to its owner */
    virtual DecoderRetCode SetCallbacks(std::function<void(DecodeEventIndex, uint32_t, void *)> eventCallBack) = 0;
    /* copyFrame(src, dst, picture info, capacity of dst) -> bytes written: moves one decoded picture into the owner's buffer */
    virtual DecoderRetCode SetCopyFrameFunc(
        std::function<uint32_t(uint8_t*, uint8_t*, const PicInfoParams &, uint32_t)> copyFrame) = 0;
    /* one chunk of the stream (an access unit); WRITE_OVERFLOW: come back after RetrieveFrameData */
    virtual DecoderRetCode SendStreamData(uint8_t *buffer, uint32_t filledLen) = 0;
    /* one decoded picture through the copy hook; READ_UNDERFLOW when none is waiting */
    virtual DecoderRetCode RetrieveFrameData(uint8_t *buffer, uint32_t maxLen, uint32_t *filledLen) = 0;
    virtual DecoderRetCode Flush() = 0;
    virtual DecoderRetCode StartDecoder() = 0;
    virtual DecoderRetCode StopDecoder() = 0;
    virtual void DestroyDecoder() = 0;
};

extern "C" {
DecoderRetCode CreateVideoDecoder(VideoDecoder** decoder);

DecoderRetCode DestroyVideoDecoder(VideoDecoder* decoder);
}

#endif
