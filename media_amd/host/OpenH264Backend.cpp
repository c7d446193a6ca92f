// media_amd/host/OpenH264Backend.cpp -- factory type 0: the CPU path of the reference (Cisco OpenH264 behind the
// same plugin surface), kept so that a box which HAS libopenh264.so can run both backends side by side.
//
// Compiled only when the OpenH264 ABI headers the reference vendors are present at build time
// (/root/reference/vendor/openh264, used in place: media_amd/host/Makefile).  Binding and preset follow
// /root/reference/video_codec/VideoEncoderOpenH264.cpp: dlopen("libopenh264.so") + WelsCreateSVCEncoder /
// WelsDestroySVCEncoder (:197-226), the SEncParamExt preset of :228-296, EncodeFrame on a
// zero-copy SSourcePicture (:344, :354-365), first layer's pBsBuf + iFrameSizeInBytes as the access unit (:349-350).
// No such library exists in this image or on the GPU box: InitEncoder then fails exactly as the reference does
// when dlopen fails (:203-208).  This file never substitutes anything for the library.
#define LOG_TAG "OpenH264Backend"
#include <dlfcn.h>
#include <cstring>
#include "MediaLog.h"
#include "OpenH264Backend.h"
#include "codec_api.h"

namespace {
using CreateFn = int (*)(ISVCEncoder **);
using DestroyFn = void (*)(ISVCEncoder *);
void *g_lib = nullptr;
CreateFn g_create = nullptr;
DestroyFn g_destroy = nullptr;

bool BindLibrary()
{
    if (g_create != nullptr && g_destroy != nullptr) {
        return true;
    }
    g_lib = dlopen("libopenh264.so", RTLD_LAZY);
    if (g_lib == nullptr) {
        ERR("libopenh264.so cannot be loaded: %s", dlerror());
        return false;
    }
    g_create = reinterpret_cast<CreateFn>(dlsym(g_lib, "WelsCreateSVCEncoder"));
    g_destroy = reinterpret_cast<DestroyFn>(dlsym(g_lib, "WelsDestroySVCEncoder"));
    if (g_create == nullptr || g_destroy == nullptr) {
        ERR("libopenh264.so lacks the encoder entry points");
        dlclose(g_lib);
        g_lib = nullptr;
        g_create = nullptr;
        g_destroy = nullptr;
        return false;
    }
    return true;
}
}  // namespace

struct OpenH264Backend::State {
    ISVCEncoder *enc = nullptr;
    SFrameBSInfo info;
    SSourcePicture pic;
};

OpenH264Backend::OpenH264Backend() : m_state(new State) {}

OpenH264Backend::~OpenH264Backend()
{
    EngineClose();
    delete m_state;
}

bool OpenH264Backend::EngineReady() const { return m_state->enc != nullptr; }

bool OpenH264Backend::EngineOpen(const Settings &s)
{
    if (!BindLibrary()) {
        return false;
    }
    if (g_create(&m_state->enc) != 0 || m_state->enc == nullptr) {
        ERR("WelsCreateSVCEncoder failed");
        m_state->enc = nullptr;
        return false;
    }
    SEncParamExt p;
    m_state->enc->GetDefaultParams(&p);
    const int w = static_cast<int>(s.width), h = static_cast<int>(s.height), rate = static_cast<int>(s.bitrate);
    p.iUsageType = CAMERA_VIDEO_REAL_TIME;
    p.iRCMode = RC_BITRATE_MODE;
    p.iPicWidth = w;
    p.iPicHeight = h;
    p.iTargetBitrate = rate;
    p.iMaxBitrate = rate;
    p.fMaxFrameRate = static_cast<float>(s.fps);
    p.uiIntraPeriod = s.gop;
    p.iTemporalLayerNum = 1;
    p.iSpatialLayerNum = 1;
    SSpatialLayerConfig &l = p.sSpatialLayers[0];
    l.iVideoWidth = w;
    l.iVideoHeight = h;
    l.fFrameRate = static_cast<float>(s.fps);
    l.iSpatialBitrate = rate;
    l.iMaxSpatialBitrate = rate;
    l.sSliceArgument.uiSliceMode = SM_SINGLE_SLICE;
    l.uiProfileIdc = s.profile == "high" ? PRO_HIGH : (s.profile == "main" ? PRO_MAIN : PRO_BASELINE);
    l.uiLevelIdc = LEVEL_3_2;
    p.iComplexityMode = HIGH_COMPLEXITY;
    p.iNumRefFrame = 1;
    p.iEntropyCodingModeFlag = 1;
    p.iMultipleThreadIdc = 1;
    p.iLoopFilterDisableIdc = 0;
    p.eSpsPpsIdStrategy = CONSTANT_ID;
    p.bPrefixNalAddingCtrl = false;
    p.bSimulcastAVC = false;
    p.iPaddingFlag = 0;
    p.uiMaxNalSize = 0;
    p.bEnableDenoise = false;
    p.bEnableBackgroundDetection = true;
    p.bEnableSceneChangeDetect = true;
    p.bEnableAdaptiveQuant = false;
    p.bEnableFrameSkip = false;
    p.bEnableLongTermReference = false;
    p.iLTRRefNum = 0;
    p.iLtrMarkPeriod = 30;
    p.bIsLosslessLink = false;
    if (m_state->enc->InitializeExt(&p) != 0) {
        ERR("InitializeExt rejected the preset");
        EngineClose();
        return false;
    }
    int fmt = videoFormatI420;
    (void) m_state->enc->SetOption(ENCODER_OPTION_DATAFORMAT, &fmt);
    std::memset(&m_state->pic, 0, sizeof(m_state->pic));
    m_state->pic.iColorFormat = videoFormatI420;
    m_state->pic.iPicWidth = w;
    m_state->pic.iPicHeight = h;
    m_state->pic.iStride[0] = w;
    m_state->pic.iStride[1] = m_state->pic.iStride[2] = w / 2;
    return true;
}

void OpenH264Backend::EngineClose()
{
    if (m_state->enc != nullptr) {
        m_state->enc->Uninitialize();
        if (g_destroy != nullptr) {
            g_destroy(m_state->enc);
        }
        m_state->enc = nullptr;
    }
}

bool OpenH264Backend::EngineEncode(const uint8_t *i420, uint8_t **out, uint32_t *outLen)
{
    uint8_t *base = const_cast<uint8_t *>(i420);
    m_state->pic.pData[0] = base;
    m_state->pic.pData[1] = base + LumaBytes();
    m_state->pic.pData[2] = base + LumaBytes() + LumaBytes() / 4;
    std::memset(&m_state->info, 0, sizeof(m_state->info));
    const int rc = m_state->enc->EncodeFrame(&m_state->pic, &m_state->info);
    if (rc != 0) {
        ERR("EncodeFrame returned %d", rc);
        return false;
    }
    *out = m_state->info.sLayerInfo[0].pBsBuf;
    *outLen = static_cast<uint32_t>(m_state->info.iFrameSizeInBytes);
    return true;
}

bool OpenH264Backend::EngineForceIdr() { return m_state->enc != nullptr && m_state->enc->ForceIntraFrame(true) == 0; }
