// media_amd/host/PropertyDrivenEncoder.cpp -- see PropertyDrivenEncoder.h.
//
// Wrapper behaviour reproduced from /root/reference/video_codec/VideoEncoderOpenH264.cpp:
//   * two property modes, "video" and "instruction", select where width/height/fps and bitrate/gop/profile are
//     read from (:62-122); anything else fails initialisation
//   * width/height must lie in 16..4096 and fps be 30 or 60, else InitEncoder fails (:159-171)
//   * bitrate 1..10 Mbps, gop 30..3000, profile baseline|main|high; a bad value is NOT an error: the last good
//     values are written back to the persist.vmi.video.encode.* keys and used (:107-120, :173-195)
//   * EncodeOneFrame: size guard (:307), then param_adjusting poll -> reset -> keyframe poll -> encode (:312-351);
//     a failing property read during the poll returns INIT_FAIL from EncodeOneFrame (:314-317)
//   * any parameter change = destroy + init + start; the next output starts with SPS/PPS + IDR (:388-404)
//   * Start/Stop only log (:298-302, :367-371); Destroy is idempotent (:379-386)
//   * the output buffer belongs to the encoder and stays valid until the next call (:349-350)
#define LOG_TAG "VideoEncoder"
#include "PropertyDrivenEncoder.h"
#include "MediaLog.h"
#include "Property.h"

namespace {

// property names per phone mode
struct KeySet {
    const char *width, *height, *fps, *bitrate, *gop, *profile;
};
constexpr KeySet kVideoKeys = {"ro.hardware.width", "ro.hardware.height", "ro.hardware.fps",
                               "persist.vmi.video.encode.bitrate", "persist.vmi.video.encode.gopsize",
                               "persist.vmi.video.encode.profile"};
constexpr KeySet kInstructionKeys = {"persist.vmi.demo.video.encode.width", "persist.vmi.demo.video.encode.height",
                                     "persist.vmi.demo.video.encode.framerate", "persist.vmi.demo.video.encode.bitrate",
                                     "persist.vmi.demo.video.encode.gopsize", "persist.vmi.demo.video.encode.profile"};
constexpr const char *kAdjustKey = "persist.vmi.video.encode.param_adjusting";
constexpr const char *kKeyframeKey = "persist.vmi.video.encode.keyframe";

bool Within(int32_t v, int32_t lo, int32_t hi) { return v >= lo && v <= hi; }

bool KnownProfile(const std::string &name) { return name == "baseline" || name == "main" || name == "high"; }

}  // namespace

constexpr int32_t PropertyDrivenEncoder::Limits::kFps[2];

PropertyDrivenEncoder::PhoneMode PropertyDrivenEncoder::ReadPhoneMode()
{
    const std::string mode = GetStrEncParam("ro.sys.vmi.cloudphone");
    if (mode == "video") {
        return PhoneMode::Video;
    }
    if (mode == "instruction") {
        return PhoneMode::Instruction;
    }
    ERR("property ro.sys.vmi.cloudphone = [%s] is neither video nor instruction", mode.c_str());
    return PhoneMode::Invalid;
}

bool PropertyDrivenEncoder::ReadGeometry(Settings &into) const
{
    const PhoneMode mode = ReadPhoneMode();
    if (mode == PhoneMode::Invalid) {
        return false;
    }
    const KeySet &k = mode == PhoneMode::Video ? kVideoKeys : kInstructionKeys;
    const int32_t w = GetIntEncParam(k.width), h = GetIntEncParam(k.height), fps = GetIntEncParam(k.fps);
    bool ok = true;
    if (!Within(w, Limits::kSideMin, Limits::kSideMax) || !Within(h, Limits::kSideMin, Limits::kSideMax)) {
        ERR("picture size %dx%d is outside %d..%d", w, h, Limits::kSideMin, Limits::kSideMax);
        ok = false;
    }
    if (fps != Limits::kFps[0] && fps != Limits::kFps[1]) {
        ERR("frame rate %d is not %d or %d", fps, Limits::kFps[0], Limits::kFps[1]);
        ok = false;
    }
    if (!ok) {
        return false;
    }
    into.width = static_cast<uint32_t>(w);
    into.height = static_cast<uint32_t>(h);
    into.fps = static_cast<uint32_t>(fps);
    return true;
}

bool PropertyDrivenEncoder::ReadTunables(Settings &into)
{
    const PhoneMode mode = ReadPhoneMode();
    if (mode == PhoneMode::Invalid) {
        return false;
    }
    const KeySet &k = mode == PhoneMode::Video ? kVideoKeys : kInstructionKeys;
    const std::string bitrate = GetStrEncParam(k.bitrate), gop = GetStrEncParam(k.gop), profile = GetStrEncParam(k.profile);
    bool ok = true;
    if (!Within(StrToInt(bitrate), Limits::kBitrateMin, Limits::kBitrateMax)) {
        WARN("bitrate [%s] rejected, keeping %u", bitrate.c_str(), m_active.bitrate);
        ok = false;
    }
    if (!Within(StrToInt(gop), Limits::kGopMin, Limits::kGopMax)) {
        WARN("gop size [%s] rejected, keeping %u", gop.c_str(), m_active.gop);
        ok = false;
    }
    if (!KnownProfile(profile)) {
        WARN("profile [%s] rejected, keeping %s", profile.c_str(), m_active.profile.c_str());
        ok = false;
    }
    if (ok) {
        into.bitrate = static_cast<uint32_t>(StrToInt(bitrate));
        into.gop = static_cast<uint32_t>(StrToInt(gop));
        into.profile = profile;
    } else {
        // the reference publishes the values it keeps using, always under the video-mode keys (:111-115)
        SetEncParam(kVideoKeys.bitrate, std::to_string(m_active.bitrate).c_str());
        SetEncParam(kVideoKeys.gop, std::to_string(m_active.gop).c_str());
        SetEncParam(kVideoKeys.profile, m_active.profile.c_str());
    }
    return true;
}

bool PropertyDrivenEncoder::EncodeParamsChange() { return !m_pending.SameAs(m_active); }

EncoderRetCode PropertyDrivenEncoder::InitEncoder()
{
    if (!ReadGeometry(m_pending) || !ReadTunables(m_pending)) {
        ERR("InitEncoder: configuration could not be read");
        return VIDEO_ENCODER_INIT_FAIL;
    }
    m_active = m_pending;
    m_lumaBytes = m_active.width * m_active.height;
    m_frameBytes = m_lumaBytes * 3 / 2;
    if (!EngineOpen(m_active)) {
        ERR("InitEncoder: the %s engine could not be opened", BackendName());
        return VIDEO_ENCODER_INIT_FAIL;
    }
    INFO("InitEncoder (%s): %ux%u @%u, %u bps, gop %u, %s", BackendName(), m_active.width, m_active.height, m_active.fps,
         m_active.bitrate, m_active.gop, m_active.profile.c_str());
    return VIDEO_ENCODER_SUCCESS;
}

EncoderRetCode PropertyDrivenEncoder::StartEncoder()
{
    INFO("StartEncoder");
    return VIDEO_ENCODER_SUCCESS;
}

EncoderRetCode PropertyDrivenEncoder::StopEncoder()
{
    INFO("StopEncoder");
    return VIDEO_ENCODER_SUCCESS;
}

bool PropertyDrivenEncoder::PollParamAdjust()
{
    const std::string flag = GetStrEncParam(kAdjustKey);
    if (flag == "1") {
        if (!ReadTunables(m_pending)) {
            return false;
        }
        (void) SetEncodeParams();
    } else if (flag == "0") {
        return true;
    } else {
        WARN("%s = [%s] is neither 0 nor 1", kAdjustKey, flag.c_str());
    }
    SetEncParam(kAdjustKey, "0");
    return true;
}

void PropertyDrivenEncoder::PollKeyframeRequest()
{
    const std::string flag = GetStrEncParam(kKeyframeKey);
    if (flag == "0") {
        return;
    }
    if (flag == "1") {
        (void) ForceKeyFrame();
    } else {
        WARN("%s = [%s] is neither 0 nor 1", kKeyframeKey, flag.c_str());
    }
    SetEncParam(kKeyframeKey, "0");
}

EncoderRetCode PropertyDrivenEncoder::EncodeOneFrame(const uint8_t *inputData, uint32_t inputSize, uint8_t **outputData,
                                                      uint32_t *outputSize)
{
    if (inputSize < m_frameBytes) {
        ERR("EncodeOneFrame: %u input bytes, a picture needs %u", inputSize, m_frameBytes);
        return VIDEO_ENCODER_ENCODE_FAIL;
    }
    if (!PollParamAdjust()) {
        ERR("EncodeOneFrame: configuration could not be re-read");
        return VIDEO_ENCODER_INIT_FAIL;  // quirk kept from the reference
    }
    if (m_needReset) {
        if (ResetEncoder() != VIDEO_ENCODER_SUCCESS) {
            ERR("EncodeOneFrame: reset after a parameter change failed");
            return VIDEO_ENCODER_ENCODE_FAIL;
        }
        m_needReset = false;
    }
    PollKeyframeRequest();
    if (!EngineReady()) {
        ERR("EncodeOneFrame: encoder is not initialised");
        return VIDEO_ENCODER_ENCODE_FAIL;
    }
    if (!EngineEncode(inputData, outputData, outputSize)) {
        return VIDEO_ENCODER_ENCODE_FAIL;
    }
    return VIDEO_ENCODER_SUCCESS;
}

void PropertyDrivenEncoder::DestroyEncoder()
{
    EngineClose();
    INFO("DestroyEncoder");
}

EncoderRetCode PropertyDrivenEncoder::ResetEncoder()
{
    INFO("ResetEncoder");
    DestroyEncoder();
    if (InitEncoder() != VIDEO_ENCODER_SUCCESS || StartEncoder() != VIDEO_ENCODER_SUCCESS) {
        ERR("ResetEncoder: could not bring the encoder back up");
        return VIDEO_ENCODER_RESET_FAIL;
    }
    return VIDEO_ENCODER_SUCCESS;
}

EncoderRetCode PropertyDrivenEncoder::ForceKeyFrame()
{
    if (!EngineReady() || !EngineForceIdr()) {
        ERR("ForceKeyFrame: engine refused");
        return VIDEO_ENCODER_FORCE_KEY_FRAME_FAIL;
    }
    INFO("ForceKeyFrame: next picture is an IDR");
    return VIDEO_ENCODER_SUCCESS;
}

EncoderRetCode PropertyDrivenEncoder::SetEncodeParams()
{
    if (EncodeParamsChange()) {
        m_active = m_pending;
        m_needReset = true;
        INFO("parameters changed: %u bps, gop %u, %s (encoder restarts on the next picture)", m_active.bitrate, m_active.gop,
             m_active.profile.c_str());
    } else {
        INFO("parameters unchanged: %u bps, gop %u, %s", m_active.bitrate, m_active.gop, m_active.profile.c_str());
    }
    return VIDEO_ENCODER_SUCCESS;
}
