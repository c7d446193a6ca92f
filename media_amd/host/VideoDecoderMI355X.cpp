/*
 * media_amd/host/VideoDecoderMI355X.cpp -- see VideoDecoderMI355X.h.  Reference behaviour cited as
 * VideoDecoderNetint.cpp:line (/root/reference/video_decoder/).
 */
#define LOG_TAG "VideoDecoderMI355X"
#include "VideoDecoderMI355X.h"
#include <cstring>
#include "MediaLog.h"

VideoDecoderMI355X::~VideoDecoderMI355X()
{
    DestroyDecoder();
    INFO("decoder object gone");
}

DecoderRetCode VideoDecoderMI355X::CreateDecoder(MediaStreamFormat decType)
{
    // the reference adapter takes AVC and HEVC (ref :209-222); this engine decodes H.264 only
    if (decType != STREAM_FORMAT_AVC) {
        ERR("create decoder: stream format %u is not supported (H.264 only)", static_cast<unsigned>(decType));
        return VIDEO_DECODER_CREATE_FAIL;
    }
    m_created = true;
    INFO("MI355X decoder constructed, h.264");
    return VIDEO_DECODER_SUCCESS;
}

DecoderRetCode VideoDecoderMI355X::InitDecoder()
{
    INFO("init decoder");   // (ref :225-229: nothing happens before StartDecoder)
    return VIDEO_DECODER_SUCCESS;
}

DecoderRetCode VideoDecoderMI355X::SetDecodeParams(DecodeParamsIndex index, void *decParams)
{
    if (index == INDEX_PIC_INFO && decParams != nullptr) {   // ref :268-277
        const auto *p = static_cast<const PicInfoParams *>(decParams);
        INFO("set decode params: width %u height %u stride %d", p->width, p->height, p->stride);
        m_writeWidth = p->width;
        m_writeHeight = p->height;
        m_stride = p->stride;
    }
    return VIDEO_DECODER_SUCCESS;
}

DecoderRetCode VideoDecoderMI355X::GetDecodeParams(DecodeParamsIndex index, void *decParams)
{
    if (decParams == nullptr) {
        return VIDEO_DECODER_GET_DECODE_PARAMS_FAIL;
    }
    switch (index) {
        case INDEX_PIC_INFO: {   // ref :287-293
            auto *p = static_cast<PicInfoParams *>(decParams);
            p->width = m_writeWidth;
            p->stride = static_cast<int32_t>(m_writeWidth);
            p->height = p->scanLines = m_writeHeight;
            break;
        }
        case INDEX_PORT_FORMAT_INFO: {   // ref :294-304; the output is tight planar 4:2:0
            auto *p = static_cast<PortFormatParams *>(decParams);
            if (p->port == OUT_PORT) {
                p->format = PIXEL_FORMAT_YUV_420P;
            } else if (p->port == IN_PORT) {
                p->format = STREAM_FORMAT_AVC;
            } else {
                return VIDEO_DECODER_GET_DECODE_PARAMS_FAIL;
            }
            break;
        }
        case INDEX_ALIGN_INFO: {   // ref :305-316: no padding here, the picture comes out cropped; 4:2:0 needs even sizes
            auto *p = static_cast<AlignInfoParams *>(decParams);
            p->widthAlign = 2;
            p->heightAlign = 2;
            break;
        }
        default:
            break;
    }
    return VIDEO_DECODER_SUCCESS;
}

DecoderRetCode VideoDecoderMI355X::SetCallbacks(std::function<void(DecodeEventIndex, uint32_t, void *)> eventCallBack)
{
    m_eventCallBack = eventCallBack;
    return VIDEO_DECODER_SUCCESS;
}

DecoderRetCode VideoDecoderMI355X::SetCopyFrameFunc(
    std::function<uint32_t(uint8_t*, uint8_t*, const PicInfoParams &, uint32_t)> copyFrame)
{
    m_copyFrame = copyFrame;
    return VIDEO_DECODER_SUCCESS;
}

DecoderRetCode VideoDecoderMI355X::StartDecoder()
{
    if (!m_created) {
        ERR("start decoder before CreateDecoder");
        return VIDEO_DECODER_START_FAIL;
    }
    if (m_engine == nullptr) {
        const int rc = mi355x_h264_dec_create(0, &m_engine);
        if (rc != MI355X_H264_OK) {   // (ref :339-348: a missing library / device session fails the start)
            ERR("mi355x_h264_dec_create returned %d (no HIP device?)", rc);
            m_engine = nullptr;
            return VIDEO_DECODER_START_FAIL;
        }
    }
    m_stop = false;
    m_pending = false;
    INFO("start decoder success");
    return VIDEO_DECODER_SUCCESS;
}

DecoderRetCode VideoDecoderMI355X::SendStreamData(uint8_t *buffer, uint32_t filledLen)
{
    if (m_stop) {   // ref :233-236
        ERR("send stream data, stop status");
        return VIDEO_DECODER_DECODE_FAIL;
    }
    if (buffer == nullptr || filledLen == 0) {
        ERR("send stream data: empty input");
        return VIDEO_DECODER_DECODE_FAIL;
    }
    if (m_pending) {
        return VIDEO_DECODER_WRITE_OVERFLOW;   // one picture waits at the output: retrieve it first (ref :595-598)
    }
    int got = 0;
    const int rc = mi355x_h264_dec_decode(m_engine, buffer, filledLen, &got);
    if (rc != MI355X_H264_OK) {
        ERR("decode failed (%d): %s", rc, mi355x_h264_dec_last_error(m_engine));
        return VIDEO_DECODER_DECODE_FAIL;
    }
    if (got) {
        int w = 0, h = 0;
        (void) mi355x_h264_dec_picture_info(m_engine, &w, &h, nullptr, nullptr);
        m_frameWidth = static_cast<uint32_t>(w);
        m_frameHeight = static_cast<uint32_t>(h);
        m_frame.resize(static_cast<size_t>(w) * h * 3 / 2);
        if (mi355x_h264_dec_read_i420(m_engine, m_frame.data(), m_frame.size()) != static_cast<int64_t>(m_frame.size())) {
            ERR("reading the decoded picture failed: %s", mi355x_h264_dec_last_error(m_engine));
            return VIDEO_DECODER_DECODE_FAIL;
        }
        m_pending = true;
        m_pictures++;
    }
    return VIDEO_DECODER_SUCCESS;
}

DecoderRetCode VideoDecoderMI355X::RetrieveFrameData(uint8_t *buffer, uint32_t maxLen, uint32_t *filledLen)
{
    if (m_stop) {   // ref :243-246
        ERR("retrieve frame data, stop status");
        return VIDEO_DECODER_DECODE_FAIL;
    }
    if (buffer == nullptr || filledLen == nullptr) {
        return VIDEO_DECODER_DECODE_FAIL;
    }
    if (!m_pending) {
        return VIDEO_DECODER_READ_UNDERFLOW;   // ref :658
    }
    if (m_frameWidth != m_writeWidth || m_frameHeight != m_writeHeight) {
        // the stream's size is not the configured one: tell the owner and keep the picture (ref :673-685)
        PicInfoParams info;
        info.width = m_frameWidth;
        info.height = m_frameHeight;
        info.stride = static_cast<int32_t>(m_frameWidth);
        info.scanLines = m_frameHeight;
        info.cropWidth = m_frameWidth;
        info.cropHeight = m_frameHeight;
        INFO("decoded size %ux%u differs from the configured %ux%u", m_frameWidth, m_frameHeight, m_writeWidth, m_writeHeight);
        if (m_eventCallBack) {
            m_eventCallBack(INDEX_PIC_INFO_CHANGE, 0, &info);
        }
        return VIDEO_DECODER_BAD_PIC_SIZE;
    }
    PicInfoParams params;
    params.width = m_writeWidth;
    params.height = m_writeHeight;
    params.stride = m_stride;
    params.scanLines = m_writeHeight;
    if (m_copyFrame) {   // ref :687-689
        *filledLen = m_copyFrame(m_frame.data(), buffer, params, maxLen);
    } else {
        if (m_frame.size() > maxLen) {
            ERR("output buffer of %u bytes is too small for %zu", maxLen, m_frame.size());
            return VIDEO_DECODER_DECODE_FAIL;
        }
        std::memcpy(buffer, m_frame.data(), m_frame.size());
        *filledLen = static_cast<uint32_t>(m_frame.size());
    }
    m_pending = false;
    return VIDEO_DECODER_SUCCESS;
}

DecoderRetCode VideoDecoderMI355X::Flush()
{
    INFO("decoder flush");   // ref :323-334: what is in flight is dropped; decoding resumes at the next IDR picture
    m_pending = false;
    if (m_engine != nullptr) {
        mi355x_h264_dec_destroy(m_engine);
        m_engine = nullptr;
        if (!m_stop && mi355x_h264_dec_create(0, &m_engine) != MI355X_H264_OK) {
            m_engine = nullptr;
            return VIDEO_DECODER_RESET_FAIL;
        }
    }
    return VIDEO_DECODER_SUCCESS;
}

DecoderRetCode VideoDecoderMI355X::StopDecoder()
{
    if (m_stop) {   // ref :357-360
        INFO("stop decoder, stop already");
        return VIDEO_DECODER_SUCCESS;
    }
    if (m_engine != nullptr) {
        mi355x_h264_dec_destroy(m_engine);
        m_engine = nullptr;
    }
    m_pending = false;
    m_stop = true;
    return VIDEO_DECODER_SUCCESS;
}

void VideoDecoderMI355X::DestroyDecoder()
{
    (void) StopDecoder();
    m_created = false;
}
