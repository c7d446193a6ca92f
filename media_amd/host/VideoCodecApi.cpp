// media_amd/host/VideoCodecApi.cpp -- the factory of the plugin surface
// (mirrors /root/reference/video_codec/VideoCodecApi.cpp:14-55).  The reference
// dispatches 0 OpenH264 / 1 NETINT H.264 / 2 NETINT H.265; this build adds
// 3 = MI355X.  The CPU and ASIC backends are not part of this library (their
// engines are third-party binaries absent here), so 0..2 fail to create; in the
// reference tree the maintainer adds only the `case 3` arm.
#define LOG_TAG "VideoCodecApi"
#include "VideoCodecApi.h"
#include <new>
#include "MediaLog.h"
#include "Property.h"
#include "VideoEncoderMI355X.h"

namespace {
enum EncoderType : uint32_t {
    ENCODER_TYPE_OPENH264 = 0,
    ENCODER_TYPE_NETINTH264 = 1,
    ENCODER_TYPE_NETINTH265 = 2,
    ENCODER_TYPE_MI355X = 3   // hand-written HIP path on AMD Instinct MI355X
};
}

EncoderRetCode CreateVideoEncoder(VideoEncoder **encoder)
{
    if (encoder == nullptr) {
        ERR("create video encoder failed: null output pointer");
        return VIDEO_ENCODER_CREATE_FAIL;
    }
    uint32_t encType = static_cast<uint32_t>(GetIntEncParam("ro.vmi.demo.video.encode.format"));
    INFO("create video encoder: encoder type %u", encType);
    switch (encType) {
        case ENCODER_TYPE_MI355X:
            *encoder = new (std::nothrow) VideoEncoderMI355X();
            break;
        case ENCODER_TYPE_OPENH264:
        case ENCODER_TYPE_NETINTH264:
        case ENCODER_TYPE_NETINTH265:
            ERR("create video encoder failed: encoder type %u is not built into this library", encType);
            return VIDEO_ENCODER_CREATE_FAIL;
        default:
            ERR("create video encoder failed: unknown encoder type %u", encType);
            return VIDEO_ENCODER_CREATE_FAIL;
    }
    if (*encoder == nullptr) {
        ERR("create video encoder failed: encoder type %u", encType);
        return VIDEO_ENCODER_CREATE_FAIL;
    }
    return VIDEO_ENCODER_SUCCESS;
}

EncoderRetCode DestroyVideoEncoder(VideoEncoder *encoder)
{
    if (encoder == nullptr) {
        WARN("input encoder is null");
        return VIDEO_ENCODER_SUCCESS;
    }
    delete encoder;
    encoder = nullptr;
    return VIDEO_ENCODER_SUCCESS;
}
