// media_amd/host/VideoEncoderMI355X.cpp -- see VideoEncoderMI355X.h.
#define LOG_TAG "VideoEncoderMI355X"
#include "VideoEncoderMI355X.h"
#include <algorithm>
#include <cstdlib>
#include "MediaLog.h"
#include "Property.h"

namespace {
bool Within(int32_t v, int32_t lo, int32_t hi) { return v >= lo && v <= hi; }
int ProfileIdc(const std::string &name) { return name == "high" ? 100 : (name == "main" ? 77 : 66); }
}  // namespace

VideoEncoderMI355X::VideoEncoderMI355X() { INFO("MI355X encoder object created"); }

VideoEncoderMI355X::~VideoEncoderMI355X()
{
    EngineClose();
    INFO("MI355X encoder object gone");
}

bool VideoEncoderMI355X::EngineOpen(const Settings &s)
{
    // the preset of InitParams / InitParamExt (ref :228-296) in the C ABI's terms: one layer, one slice per
    // picture, one reference frame, loop filter on, CAVLC, IDR every gop pictures
    if (((s.width | s.height) & 1u) != 0) {   // 4:2:0: the chroma planes are (width / 2) x (height / 2)
        ERR("picture size %ux%u: width and height must be even", s.width, s.height);
        return false;
    }
    mi355x_h264_config cfg;
    mi355x_h264_default_config(&cfg);
    cfg.width = static_cast<int32_t>(s.width);
    cfg.height = static_cast<int32_t>(s.height);
    cfg.fps = static_cast<int32_t>(s.fps);
    cfg.bitrate = static_cast<int32_t>(s.bitrate);
    cfg.gop = static_cast<int32_t>(s.gop);
    cfg.profile_idc = ProfileIdc(s.profile);
    if (cfg.profile_idc != 66) {
        // the reference preset asks for CABAC (iEntropyCodingModeFlag = 1, ref :291); this engine codes CAVLC in every profile
        INFO("profile_idc %d is coded with CAVLC (entropy_coding_mode_flag = 0): CABAC is not built", cfg.profile_idc);
    }
    cfg.disable_deblock = 0;
    cfg.batch = 1;
    cfg.device = std::max(0, GetIntEncParam("persist.vmi.video.encode.device"));
    // extension keys: a valid QP selects fixed-QP coding instead of the preset's
    // bitrate mode; "0" switches the scene-change IDR off
    const int32_t qp = GetIntEncParam("persist.vmi.video.encode.qp");
    m_fixedQp = Within(qp, 10, 51) ? qp : -1;
    cfg.rc_mode = m_fixedQp >= 0 ? MI355X_H264_RC_FIXED_QP : MI355X_H264_RC_BITRATE;
    cfg.qp = m_fixedQp >= 0 ? m_fixedQp : StartQp(s.bitrate, s.fps, s.width, s.height);
    m_qp = cfg.qp;
    m_bufferBits = 0;
    m_gopLeft = 0;
    m_picsLeft = 0;
    m_meanP = 0;
    m_sceneDetect = GetStrEncParam("persist.vmi.video.encode.scenedetect") != "0";
    // extension: 2..64 slice bands per picture for a shorter per-picture latency; anything else keeps the preset's
    // single slice (SM_SINGLE_SLICE, ref :247)
    const int32_t slices = GetIntEncParam("persist.vmi.video.encode.slices");
    cfg.slices = Within(slices, 2, 64) ? slices : 0;
    // extension: "0" = exhaustive integer motion search instead of the default seeded one (include/mi355x_h264.h config.search)
    if (GetStrEncParam("persist.vmi.video.encode.search") == "0") {
        cfg.search = MI355X_H264_SEARCH_EXHAUSTIVE;
    }
    // One engine per object (mi355x_h264_create) costs every picture its own launch sequence.  The default is a STREAM of the
    // shared engine: the pictures that the encoder objects of one process hand over at about the same time are coded in one
    // lockstep step (include/mi355x_h264.h, "streams"; same bitstream).  persist.vmi.video.encode.shared = 0 (or the environment
    // variable MI355X_H264_HUB=0) keeps the engine of its own.
    const char *hubEnv = getenv("MI355X_H264_HUB");
    const bool shared = GetStrEncParam("persist.vmi.video.encode.shared") != "0" && !(hubEnv != nullptr && hubEnv[0] == '0');
    if (shared) {
        const int rc = mi355x_h264_stream_open(&cfg, &m_stream);
        if (rc != MI355X_H264_OK) {
            ERR("mi355x_h264_stream_open returned %d", rc);
            m_stream = nullptr;
            return false;
        }
        return true;
    }
    const int rc = mi355x_h264_create(&cfg, &m_engine);
    if (rc != MI355X_H264_OK) {
        ERR("mi355x_h264_create returned %d", rc);
        m_engine = nullptr;
        return false;
    }
    return true;
}

// Frame-level rate control for the bitrate mode: GOP budget + damped QP steps, the C++ statement of media_amd/ratecontrol.py
// (integer for integer; a test replays the QP sequence on the oracle).  A GOP has target * gop bits, less the debt the virtual
// buffer carries into it (within [1/2, 3/2] of that); the IDR picture is paid out of it and every P picture is budgeted (what is
// left) / (pictures left).  The QP moves by one step when the running mean of the P pictures' bits (3/4 old + 1/4 new) is 15 %
// over or 13 % under the next picture's budget, by two beyond 3/2 and 2/3.
// PARITY UNPINNED: OpenH264's own rate-control model is not available.
void VideoEncoderMI355X::RateControlUpdate(uint32_t frameBytes, bool isIdr)
{
    if (m_fixedQp >= 0) {
        return;
    }
    const int64_t rate = static_cast<int64_t>(Active().bitrate);
    const int64_t target = rate / std::max<uint32_t>(1, Active().fps);
    const int64_t gop = std::max<int64_t>(1, static_cast<int64_t>(Active().gop));
    const int64_t bits = static_cast<int64_t>(frameBytes) * 8;
    const int64_t debt = m_bufferBits;
    m_bufferBits = std::max<int64_t>(m_bufferBits + bits - target, -rate);
    int64_t est;
    if (isIdr || m_picsLeft <= 0) {
        const int64_t full = target * gop;
        const int64_t budget = std::min(std::max(full - debt, full / 2), full * 3 / 2);
        m_gopLeft = budget - bits;
        m_picsLeft = gop - 1;
        est = m_meanP != 0 ? m_meanP : bits / 9;   // (no P picture yet: an IDR picture costs about nine of them)
    } else {
        m_gopLeft -= bits;
        m_picsLeft -= 1;
        m_meanP = m_meanP != 0 ? (m_meanP * 3 + bits) / 4 : bits;
        est = m_meanP;
    }
    // (floor division as in the Python statement: m_gopLeft may be negative)
    const int64_t n = std::max<int64_t>(1, m_picsLeft);
    int64_t share = m_gopLeft / n;
    if (m_gopLeft < 0 && m_gopLeft % n != 0) {
        share -= 1;
    }
    const int64_t next = std::max(share, target / 4);
    int32_t step = 0;
    if (est * 2 > next * 3) {
        step = 2;
    } else if (est * 100 > next * 115) {
        step = 1;
    } else if (est * 3 < next * 2) {
        step = -2;
    } else if (est * 100 < next * 87) {
        step = -1;
    }
    m_qp = std::min(Rc::kQpMax, std::max(Rc::kQpMin, m_qp + step));
}

// QP of a stream's first picture from its bits per pixel (media_amd/ratecontrol.py start_qp)
int32_t VideoEncoderMI355X::StartQp(uint32_t bitrate, uint32_t fps, uint32_t width, uint32_t height)
{
    const int64_t mbpp = static_cast<int64_t>(bitrate) * 1000 / std::max<int64_t>(1, static_cast<int64_t>(fps) * width * height);
    static const struct { int64_t lim; int32_t qp; } table[] = {{200, 24}, {100, 27}, {50, 30}, {25, 34}, {12, 38}};
    for (const auto &t : table) {
        if (mbpp >= t.lim) {
            return t.qp;
        }
    }
    return 42;
}

int VideoEncoderMI355X::EncodePicture(const uint8_t *i420, uint8_t **out, uint32_t *outLen, int *frameType)
{
    // tight I420 exactly as the reference's InitSrcPic lays the planes out (ref :354-365)
    const int pitch = static_cast<int>(Active().width);
    const uint8_t *u = i420 + LumaBytes();
    const uint8_t *v = u + LumaBytes() / 4;
    if (m_stream != nullptr) {
        return mi355x_h264_stream_encode(m_stream, i420, pitch, u, pitch / 2, v, pitch / 2, out, outLen, frameType);
    }
    return mi355x_h264_encode(m_engine, i420, pitch, u, pitch / 2, v, pitch / 2, out, outLen, frameType);
}

bool VideoEncoderMI355X::EngineEncode(const uint8_t *i420, uint8_t **out, uint32_t *outLen)
{
    if (m_stream != nullptr) {
        (void) mi355x_h264_stream_set_qp(m_stream, m_qp);
    } else {
        (void) mi355x_h264_set_qp(m_engine, m_qp);
    }
    m_lastQp = m_qp;
    int frameType = 0;
    int rc = EncodePicture(i420, out, outLen, &frameType);
    if (rc == MI355X_H264_OK && m_sceneDetect && frameType == MI355X_H264_FRAME_P) {
        // scene change: the motion search found no good match anywhere -> code this picture as IDR instead
        uint32_t cost = 0;
        const uint64_t mbs = static_cast<uint64_t>((Active().width + 15) / 16) * ((Active().height + 15) / 16);
        const int crc = m_stream != nullptr ? mi355x_h264_stream_last_me_cost(m_stream, &cost) : mi355x_h264_last_me_cost(m_engine, &cost);
        if (crc == MI355X_H264_OK && cost > Rc::kSceneCutCostPerMb * mbs) {
            INFO("scene change (motion cost %u over %llu macroblocks): picture re-coded as IDR", cost,
                 static_cast<unsigned long long>(mbs));
            (void) EngineForceIdr();
            rc = EncodePicture(i420, out, outLen, &frameType);
            m_sceneCuts++;
        }
    }
    if (rc != MI355X_H264_OK) {
        ERR("EncodeOneFrame: engine returned %d (%s)", rc,
            m_stream != nullptr ? mi355x_h264_stream_last_error(m_stream) : mi355x_h264_last_error(m_engine));
        return false;
    }
    RateControlUpdate(*outLen, frameType == MI355X_H264_FRAME_IDR);
    return true;
}

void VideoEncoderMI355X::EngineClose()
{
    if (m_engine != nullptr) {
        mi355x_h264_destroy(m_engine);
        m_engine = nullptr;
    }
    if (m_stream != nullptr) {
        mi355x_h264_stream_close(m_stream);
        m_stream = nullptr;
    }
}

bool VideoEncoderMI355X::EngineForceIdr()
{
    if (m_stream != nullptr) {
        return mi355x_h264_stream_force_idr(m_stream) == MI355X_H264_OK;
    }
    return m_engine != nullptr && mi355x_h264_force_idr(m_engine) == MI355X_H264_OK;
}

int64_t VideoEncoderMI355X::ReadReconY(void *dst, size_t cap, int32_t *codedWidth, int32_t *codedHeight)
{
    if (m_stream != nullptr) {
        if (codedWidth != nullptr) *codedWidth = mi355x_h264_stream_coded_width(m_stream);
        if (codedHeight != nullptr) *codedHeight = mi355x_h264_stream_coded_height(m_stream);
        return mi355x_h264_stream_debug_read(m_stream, MI355X_H264_DBG_RECON_Y, dst, cap);
    }
    if (m_engine == nullptr) return -1;
    if (codedWidth != nullptr) *codedWidth = mi355x_h264_coded_width(m_engine);
    if (codedHeight != nullptr) *codedHeight = mi355x_h264_coded_height(m_engine);
    return mi355x_h264_debug_read(m_engine, MI355X_H264_DBG_RECON_Y, dst, cap);
}
