/*
 * media_amd/host/VideoDecoderApi.cpp -- factory of libVideoDecoder (reference: /root/reference/video_decoder/
 * VideoDecoderApi.cpp:11-37, which always builds the NETINT adapter).  Here it builds the MI355X backend.
 */
#define LOG_TAG "VideoDecoderApi"
#include <new>
#include "MediaLog.h"
#include "VideoDecoderMI355X.h"

DecoderRetCode CreateVideoDecoder(VideoDecoder **decoder)
{
    if (decoder == nullptr) {
        ERR("create video decoder failed: null output pointer");
        return VIDEO_DECODER_CREATE_FAIL;
    }
    *decoder = new (std::nothrow) VideoDecoderMI355X();
    if (*decoder == nullptr) {
        ERR("create video decoder failed");
        return VIDEO_DECODER_CREATE_FAIL;
    }
    return VIDEO_DECODER_SUCCESS;
}

DecoderRetCode DestroyVideoDecoder(VideoDecoder *decoder)
{
    if (decoder == nullptr) {   // ref :26-29
        WARN("input decoder is null");
        return VIDEO_DECODER_SUCCESS;
    }
    delete decoder;
    return VIDEO_DECODER_SUCCESS;
}
