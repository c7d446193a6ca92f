// media_amd/host/VideoEncoderMI355X.cpp -- see VideoEncoderMI355X.h.
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
#define LOG_TAG "VideoEncoderMI355X"
#include "VideoEncoderMI355X.h"
#include <algorithm>
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

int ProfileIdc(const std::string &name) { return name == "high" ? 100 : (name == "main" ? 77 : 66); }

bool KnownProfile(const std::string &name) { return name == "baseline" || name == "main" || name == "high"; }

}  // namespace

constexpr int32_t VideoEncoderMI355X::Limits::kFps[2];

VideoEncoderMI355X::VideoEncoderMI355X() { INFO("MI355X encoder object created"); }

VideoEncoderMI355X::~VideoEncoderMI355X()
{
    CloseEngine();
    INFO("MI355X encoder object gone");
}

VideoEncoderMI355X::PhoneMode VideoEncoderMI355X::ReadPhoneMode()
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

bool VideoEncoderMI355X::ReadGeometry(Settings &into) const
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

bool VideoEncoderMI355X::ReadTunables(Settings &into)
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

bool VideoEncoderMI355X::EncodeParamsChange() { return !m_pending.SameAs(m_active); }

EncoderRetCode VideoEncoderMI355X::InitEncoder()
{
    if (!ReadGeometry(m_pending) || !ReadTunables(m_pending)) {
        ERR("InitEncoder: configuration could not be read");
        return VIDEO_ENCODER_INIT_FAIL;
    }
    m_active = m_pending;
    m_lumaBytes = m_active.width * m_active.height;
    m_frameBytes = m_lumaBytes * 3 / 2;
    if (!OpenEngine()) {
        ERR("InitEncoder: the HIP encode engine could not be opened");
        return VIDEO_ENCODER_INIT_FAIL;
    }
    INFO("InitEncoder: %ux%u @%u, %u bps, gop %u, %s", m_active.width, m_active.height, m_active.fps, m_active.bitrate,
         m_active.gop, m_active.profile.c_str());
    return VIDEO_ENCODER_SUCCESS;
}

bool VideoEncoderMI355X::OpenEngine()
{
    // the preset of InitParams / InitParamExt (ref :228-296) in the C ABI's terms: one layer, one slice per
    // picture, one reference frame, loop filter on, CAVLC, IDR every gop pictures
    mi355x_h264_config cfg;
    mi355x_h264_default_config(&cfg);
    cfg.width = static_cast<int32_t>(m_active.width);
    cfg.height = static_cast<int32_t>(m_active.height);
    cfg.fps = static_cast<int32_t>(m_active.fps);
    cfg.bitrate = static_cast<int32_t>(m_active.bitrate);
    cfg.gop = static_cast<int32_t>(m_active.gop);
    cfg.profile_idc = ProfileIdc(m_active.profile);
    cfg.disable_deblock = 0;
    cfg.batch = 1;
    cfg.device = std::max(0, GetIntEncParam("persist.vmi.video.encode.device"));
    // extension keys: a valid QP selects fixed-QP coding instead of the preset's
    // bitrate mode; "0" switches the scene-change IDR off
    const int32_t qp = GetIntEncParam("persist.vmi.video.encode.qp");
    m_fixedQp = Within(qp, 10, 51) ? qp : -1;
    cfg.rc_mode = m_fixedQp >= 0 ? MI355X_H264_RC_FIXED_QP : MI355X_H264_RC_BITRATE;
    cfg.qp = m_fixedQp >= 0 ? m_fixedQp : Limits::kQpStart;
    m_qp = cfg.qp;
    m_bufferBits = 0;
    m_sceneDetect = GetStrEncParam("persist.vmi.video.encode.scenedetect") != "0";
    const int rc = mi355x_h264_create(&cfg, &m_engine);
    if (rc != MI355X_H264_OK) {
        ERR("mi355x_h264_create returned %d", rc);
        m_engine = nullptr;
        return false;
    }
    return true;
}

EncoderRetCode VideoEncoderMI355X::StartEncoder()
{
    INFO("StartEncoder");
    return VIDEO_ENCODER_SUCCESS;
}

EncoderRetCode VideoEncoderMI355X::StopEncoder()
{
    INFO("StopEncoder");
    return VIDEO_ENCODER_SUCCESS;
}

// Frame-level rate control for the bitrate mode.  Integer arithmetic only, so a test can replay the QP
// sequence on the oracle.  Target per picture = bitrate / fps; an IDR picture is budgeted four pictures' worth.
// PARITY UNPINNED: OpenH264's own rate-control model is not available.
void VideoEncoderMI355X::RateControlUpdate(uint32_t frameBytes, bool isIdr)
{
    if (m_fixedQp >= 0) {
        return;
    }
    const int64_t rate = static_cast<int64_t>(m_active.bitrate);
    const int64_t target = rate / std::max<uint32_t>(1, m_active.fps);
    const int64_t bits = static_cast<int64_t>(frameBytes) * 8;
    m_bufferBits = std::max<int64_t>(m_bufferBits + bits - target, -rate);
    const int64_t budget = isIdr ? 4 * target : target;
    int32_t step = 0;
    if (bits * 2 > budget * 3) {
        step = 2;   // more than 1.5x the budget
    } else if (bits * 10 > budget * 11) {
        step = 1;   // more than 1.1x
    } else if (bits * 3 < budget * 2) {
        step = -2;  // less than 2/3
    } else if (bits * 10 < budget * 9) {
        step = -1;  // less than 0.9x
    }
    // half a second of debt (credit) in the virtual buffer pushes one step further
    if (m_bufferBits * 2 > rate) {
        step += 1;
    }
    if (m_bufferBits * 2 < -rate) {
        step -= 1;
    }
    m_qp = std::min(Limits::kQpMax, std::max(Limits::kQpMin, m_qp + step));
}

bool VideoEncoderMI355X::PollParamAdjust()
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

void VideoEncoderMI355X::PollKeyframeRequest()
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

int VideoEncoderMI355X::EncodePicture(const uint8_t *i420, uint8_t **out, uint32_t *outLen, int *frameType)
{
    // tight I420 exactly as the reference's InitSrcPic lays the planes out (ref :354-365)
    const int pitch = static_cast<int>(m_active.width);
    const uint8_t *u = i420 + m_lumaBytes;
    const uint8_t *v = u + m_lumaBytes / 4;
    return mi355x_h264_encode(m_engine, i420, pitch, u, pitch / 2, v, pitch / 2, out, outLen, frameType);
}

EncoderRetCode VideoEncoderMI355X::EncodeOneFrame(const uint8_t *inputData, uint32_t inputSize, uint8_t **outputData,
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
    if (m_engine == nullptr) {
        ERR("EncodeOneFrame: encoder is not initialised");
        return VIDEO_ENCODER_ENCODE_FAIL;
    }
    (void) mi355x_h264_set_qp(m_engine, m_qp);
    m_lastQp = m_qp;
    int frameType = 0;
    int rc = EncodePicture(inputData, outputData, outputSize, &frameType);
    if (rc == MI355X_H264_OK && m_sceneDetect && frameType == MI355X_H264_FRAME_P) {
        // scene change: the motion search found no good match anywhere -> code this picture as IDR instead
        uint32_t cost = 0;
        const uint64_t mbs = static_cast<uint64_t>((m_active.width + 15) / 16) * ((m_active.height + 15) / 16);
        if (mi355x_h264_last_me_cost(m_engine, &cost) == MI355X_H264_OK && cost > Limits::kSceneCutCostPerMb * mbs) {
            INFO("scene change (motion cost %u over %llu macroblocks): picture re-coded as IDR", cost,
                 static_cast<unsigned long long>(mbs));
            (void) mi355x_h264_force_idr(m_engine);
            rc = EncodePicture(inputData, outputData, outputSize, &frameType);
            m_sceneCuts++;
        }
    }
    if (rc != MI355X_H264_OK) {
        ERR("EncodeOneFrame: engine returned %d (%s)", rc, mi355x_h264_last_error(m_engine));
        return VIDEO_ENCODER_ENCODE_FAIL;
    }
    RateControlUpdate(*outputSize, frameType == MI355X_H264_FRAME_IDR);
    return VIDEO_ENCODER_SUCCESS;
}

void VideoEncoderMI355X::CloseEngine()
{
    if (m_engine != nullptr) {
        mi355x_h264_destroy(m_engine);
        m_engine = nullptr;
    }
}

void VideoEncoderMI355X::DestroyEncoder()
{
    CloseEngine();
    INFO("DestroyEncoder");
}

EncoderRetCode VideoEncoderMI355X::ResetEncoder()
{
    INFO("ResetEncoder");
    DestroyEncoder();
    if (InitEncoder() != VIDEO_ENCODER_SUCCESS || StartEncoder() != VIDEO_ENCODER_SUCCESS) {
        ERR("ResetEncoder: could not bring the encoder back up");
        return VIDEO_ENCODER_RESET_FAIL;
    }
    return VIDEO_ENCODER_SUCCESS;
}

EncoderRetCode VideoEncoderMI355X::ForceKeyFrame()
{
    if (m_engine == nullptr || mi355x_h264_force_idr(m_engine) != MI355X_H264_OK) {
        ERR("ForceKeyFrame: engine refused");
        return VIDEO_ENCODER_FORCE_KEY_FRAME_FAIL;
    }
    INFO("ForceKeyFrame: next picture is an IDR");
    return VIDEO_ENCODER_SUCCESS;
}

EncoderRetCode VideoEncoderMI355X::SetEncodeParams()
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
