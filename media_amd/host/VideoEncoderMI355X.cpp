// media_amd/host/VideoEncoderMI355X.cpp -- see VideoEncoderMI355X.h.
// Behaviour follows /root/reference/video_codec/VideoEncoderOpenH264.cpp line by
// line where the wrapper is concerned:
//   :62-122  property reads (video / instruction mode), validation, write-back of
//            last good bitrate/gop/profile to the persist.vmi.video.encode.* keys
//   :131-157 InitEncoder        :304-352 EncodeOneFrame (size guard, param_adjusting
//            poll, reset, keyframe poll, encode)   :388-404 ResetEncoder
//   :406-429 ForceKeyFrame / SetEncodeParams       :379-386 Release (idempotent)
// What differs is only the engine behind it: the HIP path via include/mi355x_h264.h.
#define LOG_TAG "VideoEncoderMI355X"
#include "VideoEncoderMI355X.h"
#include <algorithm>
#include "MediaLog.h"
#include "Property.h"

namespace {
    constexpr uint32_t COMPRESS_RATIO = 2;
    constexpr uint32_t PRIMARY_COLOURS = 3;
    const std::string ENCODE_PROFILE_BASELINE = "baseline";
    const std::string ENCODE_PROFILE_MAIN = "main";
    const std::string ENCODE_PROFILE_HIGH = "high";
}

VideoEncoderMI355X::VideoEncoderMI355X()
{
    INFO("VideoEncoderMI355X constructor");
}

VideoEncoderMI355X::~VideoEncoderMI355X()
{
    Release();
    INFO("VideoEncoderMI355X destructor");
}

bool VideoEncoderMI355X::GetRoEncParam()
{
    int32_t width = 0;
    int32_t height = 0;
    int32_t framerate = 0;
    std::string phoneMode = GetStrEncParam("ro.sys.vmi.cloudphone");
    if (phoneMode == "video") {
        width = GetIntEncParam("ro.hardware.width");
        height = GetIntEncParam("ro.hardware.height");
        framerate = GetIntEncParam("ro.hardware.fps");
    } else if (phoneMode == "instruction") {
        width = GetIntEncParam("persist.vmi.demo.video.encode.width");
        height = GetIntEncParam("persist.vmi.demo.video.encode.height");
        framerate = GetIntEncParam("persist.vmi.demo.video.encode.framerate");
    } else {
        ERR("Invalid property value[%s] for property[ro.sys.vmi.cloudphone], get property failed!", phoneMode.c_str());
        return false;
    }
    if (!VerifyEncodeRoParams(width, height, framerate)) {
        ERR("encoder params is not supported");
        return false;
    }
    m_tmpEncParams.width = static_cast<uint32_t>(width);
    m_tmpEncParams.height = static_cast<uint32_t>(height);
    m_tmpEncParams.framerate = static_cast<uint32_t>(framerate);
    return true;
}

bool VideoEncoderMI355X::GetPersistEncParam()
{
    std::string bitrate;
    std::string gopsize;
    std::string profile;
    std::string phoneMode = GetStrEncParam("ro.sys.vmi.cloudphone");
    if (phoneMode == "video") {
        bitrate = GetStrEncParam("persist.vmi.video.encode.bitrate");
        gopsize = GetStrEncParam("persist.vmi.video.encode.gopsize");
        profile = GetStrEncParam("persist.vmi.video.encode.profile");
    } else if (phoneMode == "instruction") {
        bitrate = GetStrEncParam("persist.vmi.demo.video.encode.bitrate");
        gopsize = GetStrEncParam("persist.vmi.demo.video.encode.gopsize");
        profile = GetStrEncParam("persist.vmi.demo.video.encode.profile");
    } else {
        ERR("Invalid property value[%s] for property[ro.sys.vmi.cloudphone], get property failed!", phoneMode.c_str());
        return false;
    }
    if (!VerifyEncodeParams(bitrate, gopsize, profile)) {
        // not an error: the last good values are written back and init proceeds (ref :111-115)
        SetEncParam("persist.vmi.video.encode.bitrate", std::to_string(m_encParams.bitrate).c_str());
        SetEncParam("persist.vmi.video.encode.gopsize", std::to_string(m_encParams.gopsize).c_str());
        SetEncParam("persist.vmi.video.encode.profile", m_encParams.profile.c_str());
    } else {
        m_tmpEncParams.bitrate = static_cast<uint32_t>(StrToInt(bitrate));
        m_tmpEncParams.gopsize = static_cast<uint32_t>(StrToInt(gopsize));
        m_tmpEncParams.profile = profile;
    }
    return true;
}

bool VideoEncoderMI355X::EncodeParamsChange()
{
    return (m_tmpEncParams.bitrate != m_encParams.bitrate) || (m_tmpEncParams.gopsize != m_encParams.gopsize) ||
           (m_tmpEncParams.profile != m_encParams.profile) || (m_tmpEncParams.width != m_encParams.width) ||
           (m_tmpEncParams.height != m_encParams.height) || (m_tmpEncParams.framerate != m_encParams.framerate);
}

EncoderRetCode VideoEncoderMI355X::InitEncoder()
{
    if ((!GetRoEncParam()) || (!GetPersistEncParam())) {
        ERR("init encoder failed: GetEncParam failed");
        return VIDEO_ENCODER_INIT_FAIL;
    }
    m_encParams = m_tmpEncParams;
    m_frameSize = m_encParams.width * m_encParams.height * PRIMARY_COLOURS / COMPRESS_RATIO;
    m_yLength = m_encParams.width * m_encParams.height;
    if (!InitParams()) {
        ERR("init encoder failed: init params failed");
        return VIDEO_ENCODER_INIT_FAIL;
    }
    INFO("init encoder success");
    return VIDEO_ENCODER_SUCCESS;
}

bool VideoEncoderMI355X::VerifyEncodeRoParams(int32_t width, int32_t height, int32_t framerate)
{
    bool isEncodeParamsTrue = true;
    if (width > static_cast<int32_t>(MI355X::WH_MAX) || height > static_cast<int32_t>(MI355X::WH_MAX) ||
        width < static_cast<int32_t>(MI355X::WH_MIN) || height < static_cast<int32_t>(MI355X::WH_MIN)) {
        ERR("Invalid property value[%dx%d] for property[width,height], get property failed!", width, height);
        isEncodeParamsTrue = false;
    }
    if (framerate != static_cast<int32_t>(MI355X::FRAMERATE_MIN) && framerate != static_cast<int32_t>(MI355X::FRAMERATE_MAX)) {
        ERR("Invalid property value[%d] for property[framerate], get property failed!", framerate);
        isEncodeParamsTrue = false;
    }
    return isEncodeParamsTrue;
}

bool VideoEncoderMI355X::VerifyEncodeParams(std::string &bitrate, std::string &gopsize, std::string &profile)
{
    bool isEncodeParamsTrue = true;
    if ((StrToInt(bitrate) < static_cast<int32_t>(MI355X::BITRATE_MIN)) || (StrToInt(bitrate) > static_cast<int32_t>(MI355X::BITRATE_MAX))) {
        WARN("Invalid property value[%s] for property[bitrate], use last correct encode bitrate[%u]", bitrate.c_str(),
             m_encParams.bitrate);
        isEncodeParamsTrue = false;
    }
    if ((StrToInt(gopsize) < static_cast<int32_t>(MI355X::GOPSIZE_MIN)) || (StrToInt(gopsize) > static_cast<int32_t>(MI355X::GOPSIZE_MAX))) {
        WARN("Invalid property value[%s] for property[gopsize], use last correct encode gopsize[%u]", gopsize.c_str(),
             m_encParams.gopsize);
        isEncodeParamsTrue = false;
    }
    if (profile != ENCODE_PROFILE_BASELINE && profile != ENCODE_PROFILE_MAIN && profile != ENCODE_PROFILE_HIGH) {
        WARN("Invalid property value[%s] for property[profile], use last correct encode profile[%s]", profile.c_str(),
             m_encParams.profile.c_str());
        isEncodeParamsTrue = false;
    }
    return isEncodeParamsTrue;
}

bool VideoEncoderMI355X::InitParams()
{
    // the preset of VideoEncoderOpenH264::InitParams / InitParamExt (ref :228-296), expressed in
    // the C ABI's config: one layer, single slice, one reference, loop filter on, CAVLC
    mi355x_h264_config cfg;
    mi355x_h264_default_config(&cfg);
    cfg.width = static_cast<int32_t>(m_encParams.width);
    cfg.height = static_cast<int32_t>(m_encParams.height);
    cfg.fps = static_cast<int32_t>(m_encParams.framerate);
    cfg.bitrate = static_cast<int32_t>(m_encParams.bitrate);
    cfg.gop = static_cast<int32_t>(m_encParams.gopsize);
    cfg.profile_idc = m_encParams.profile == ENCODE_PROFILE_HIGH ? 100 : (m_encParams.profile == ENCODE_PROFILE_MAIN ? 77 : 66);
    cfg.disable_deblock = 0;
    const int32_t dev = GetIntEncParam("persist.vmi.video.encode.device");
    cfg.device = dev >= 0 ? dev : 0;
    // extension knob: a valid QP here selects fixed-QP coding,
    // otherwise the reference's bitrate mode is used
    const int32_t qp = GetIntEncParam("persist.vmi.video.encode.qp");
    if (qp >= 10 && qp <= 51) {
        m_fixedQp = qp;
        cfg.rc_mode = MI355X_H264_RC_FIXED_QP;
        cfg.qp = qp;
    } else {
        m_fixedQp = -1;
        cfg.rc_mode = MI355X_H264_RC_BITRATE;
        cfg.qp = MI355X::QP_START;
    }
    m_qp = cfg.qp;
    m_bufferBits = 0;
    // bEnableSceneChangeDetect = 1 in the reference preset (ref :283); "0" in this extension key turns it off
    m_sceneDetect = GetStrEncParam("persist.vmi.video.encode.scenedetect") != "0";
    const int rc = mi355x_h264_create(&cfg, &m_encoder);
    if (rc != MI355X_H264_OK) {
        ERR("mi355x_h264_create failed, rc = %d", rc);
        m_encoder = nullptr;
        return false;
    }
    return true;
}

EncoderRetCode VideoEncoderMI355X::StartEncoder()
{
    INFO("start encoder success");
    return VIDEO_ENCODER_SUCCESS;
}

// Frame-level rate control for RC_BITRATE_MODE.  Integer arithmetic only, so a test can
// replay the QP sequence.  Target per picture = bitrate / fps; IDR pictures are budgeted
// four pictures' worth.  PARITY UNPINNED: OpenH264's own RC model is not available.
void VideoEncoderMI355X::RateControlUpdate(uint32_t frameBytes, bool isIdr)
{
    if (m_fixedQp >= 0) return;
    const int64_t target = static_cast<int64_t>(m_encParams.bitrate) / std::max<uint32_t>(1, m_encParams.framerate);
    const int64_t bits = static_cast<int64_t>(frameBytes) * 8;
    m_bufferBits += bits - target;
    m_bufferBits = std::max<int64_t>(m_bufferBits, -static_cast<int64_t>(m_encParams.bitrate));
    const int64_t budget = isIdr ? 4 * target : target;
    int32_t step = 0;
    if (bits * 2 > budget * 3) step = 2;            // > 1.5x
    else if (bits * 10 > budget * 11) step = 1;     // > 1.1x
    else if (bits * 3 < budget * 2) step = -2;      // < 0.67x
    else if (bits * 10 < budget * 9) step = -1;     // < 0.9x
    // virtual buffer: more than half a second of debt / credit biases the step
    if (m_bufferBits * 2 > static_cast<int64_t>(m_encParams.bitrate)) step += 1;
    if (m_bufferBits * 2 < -static_cast<int64_t>(m_encParams.bitrate)) step -= 1;
    m_qp = std::min(MI355X::QP_MAX, std::max(MI355X::QP_MIN, m_qp + step));
}

EncoderRetCode VideoEncoderMI355X::EncodeOneFrame(const uint8_t *inputData, uint32_t inputSize, uint8_t **outputData,
                                                  uint32_t *outputSize)
{
    if (inputSize < static_cast<size_t>(m_frameSize)) {
        ERR("input size error: input size(%u) < frame size(%u)", inputSize, m_frameSize);
        return VIDEO_ENCODER_ENCODE_FAIL;
    }

    std::string isParamChange = GetStrEncParam("persist.vmi.video.encode.param_adjusting");
    if (isParamChange == "1") {
        if (!GetPersistEncParam()) {
            ERR("init encoder failed: GetEncParam failed");
            return VIDEO_ENCODER_INIT_FAIL;  // quirk kept from the reference (:314-317)
        }
        SetEncodeParams();
        SetEncParam("persist.vmi.video.encode.param_adjusting", "0");
    } else if (isParamChange != "0") {
        WARN("Invalid property value[%s] for encode param adjusting", isParamChange.c_str());
        SetEncParam("persist.vmi.video.encode.param_adjusting", "0");
    }

    if (m_resetFlag) {
        if (ResetEncoder() != VIDEO_ENCODER_SUCCESS) {
            ERR("reset encoder failed while encoding");
            return VIDEO_ENCODER_ENCODE_FAIL;
        }
        m_resetFlag = false;
    }

    std::string isKeyframeChange = GetStrEncParam("persist.vmi.video.encode.keyframe");
    if (isKeyframeChange == "1") {
        INFO("Encoder set key frame");
        ForceKeyFrame();
        SetEncParam("persist.vmi.video.encode.keyframe", "0");
    } else if (isKeyframeChange != "0") {
        WARN("Invalid property value[%s] for property[keyFrame], set to [0]", isKeyframeChange.c_str());
        SetEncParam("persist.vmi.video.encode.keyframe", "0");
    }

    if (m_encoder == nullptr) {
        ERR("encode before init");
        return VIDEO_ENCODER_ENCODE_FAIL;
    }
    // plane pointers exactly as InitSrcPic computes them (ref :354-365): tight I420
    const uint8_t *y = inputData;
    const uint8_t *u = y + m_yLength;
    const uint8_t *v = u + (m_yLength >> COMPRESS_RATIO);
    const int stride = static_cast<int>(m_encParams.width);
    (void) mi355x_h264_set_qp(m_encoder, m_qp);
    m_lastQp = m_qp;
    int frameType = 0;
    const int rc = mi355x_h264_encode(m_encoder, y, stride, u, stride / 2, v, stride / 2, outputData, outputSize, &frameType);
    if (rc != MI355X_H264_OK) {
        ERR("encoder encode frame failed, rc = %d (%s)", rc, mi355x_h264_last_error(m_encoder));
        return VIDEO_ENCODER_ENCODE_FAIL;
    }
    if (m_sceneDetect && frameType == MI355X_H264_FRAME_P) {
        // scene change: the motion search found no good match anywhere -> code this picture as IDR instead
        uint32_t cost = 0;
        const uint32_t mbs = ((m_encParams.width + 15) / 16) * ((m_encParams.height + 15) / 16);
        if (mi355x_h264_last_me_cost(m_encoder, &cost) == MI355X_H264_OK &&
            static_cast<uint64_t>(cost) > static_cast<uint64_t>(MI355X::SCENE_CUT_COST_PER_MB) * mbs) {
            INFO("scene change detected (motion cost %u over %u macroblocks), re-coding as IDR", cost, mbs);
            (void) mi355x_h264_force_idr(m_encoder);
            const int rc2 = mi355x_h264_encode(m_encoder, y, stride, u, stride / 2, v, stride / 2, outputData, outputSize, &frameType);
            if (rc2 != MI355X_H264_OK) {
                ERR("encoder encode frame failed, rc = %d (%s)", rc2, mi355x_h264_last_error(m_encoder));
                return VIDEO_ENCODER_ENCODE_FAIL;
            }
            m_sceneCuts++;
        }
    }
    RateControlUpdate(*outputSize, frameType == MI355X_H264_FRAME_IDR);
    return VIDEO_ENCODER_SUCCESS;
}

EncoderRetCode VideoEncoderMI355X::StopEncoder()
{
    INFO("stop encoder success");
    return VIDEO_ENCODER_SUCCESS;
}

void VideoEncoderMI355X::DestroyEncoder()
{
    Release();
    INFO("destroy encoder success");
}

void VideoEncoderMI355X::Release()
{
    if (m_encoder != nullptr) {
        mi355x_h264_destroy(m_encoder);
        m_encoder = nullptr;
    }
}

EncoderRetCode VideoEncoderMI355X::ResetEncoder()
{
    INFO("resetting encoder");
    DestroyEncoder();
    EncoderRetCode ret = InitEncoder();
    if (ret != VIDEO_ENCODER_SUCCESS) {
        ERR("init encoder failed %#x while resetting", ret);
        return VIDEO_ENCODER_RESET_FAIL;
    }
    ret = StartEncoder();
    if (ret != VIDEO_ENCODER_SUCCESS) {
        ERR("start encoder failed %#x while resetting", ret);
        return VIDEO_ENCODER_RESET_FAIL;
    }
    INFO("reset encoder success");
    return VIDEO_ENCODER_SUCCESS;
}

EncoderRetCode VideoEncoderMI355X::ForceKeyFrame()
{
    if (m_encoder == nullptr || mi355x_h264_force_idr(m_encoder) != MI355X_H264_OK) {
        ERR("encoder force intra frame failed");
        return VIDEO_ENCODER_FORCE_KEY_FRAME_FAIL;
    }
    INFO("force key frame success");
    return VIDEO_ENCODER_SUCCESS;
}

EncoderRetCode VideoEncoderMI355X::SetEncodeParams()
{
    if (EncodeParamsChange()) {
        m_encParams = m_tmpEncParams;
        m_resetFlag = true;
        INFO("Handle encoder config change: [bitrate, gopsize, profile] = [%u,%u,%s]", m_encParams.bitrate,
             m_encParams.gopsize, m_encParams.profile.c_str());
    } else {
        INFO("Using encoder config: [bitrate, gopsize, profile] = [%u,%u,%s]", m_encParams.bitrate,
             m_encParams.gopsize, m_encParams.profile.c_str());
    }
    return VIDEO_ENCODER_SUCCESS;
}
