/*
 * media_amd/host/VideoEncoderMI355X.h -- the MI355X backend of the VideoEncoder
 * plugin surface: a peer of the reference's VideoEncoderOpenH264
 * (/root/reference/video_codec/VideoEncoderOpenH264.h:27-196) with the same
 * operator API, property-driven configuration, live re-configuration handshake
 * and buffer ownership, calling the HIP encode path through the C ABI of
 * include/mi355x_h264.h instead of the ISVCEncoder vtable.
 */
#ifndef VIDEO_ENCODER_MI355X_H
#define VIDEO_ENCODER_MI355X_H

#include <atomic>
#include <string>
#include "VideoCodecApi.h"
#include "mi355x_h264.h"

namespace MI355X {
    // limits and defaults of the reference adapter (VideoEncoderOpenH264.h:12-25, .cpp:16-23)
    constexpr uint32_t DEFAULT_WIDTH = 720;
    constexpr uint32_t DEFAULT_HEIGHT = 1280;
    constexpr uint32_t WH_MIN = 16;
    constexpr uint32_t WH_MAX = 4096;
    constexpr uint32_t FRAMERATE_MIN = 30;
    constexpr uint32_t FRAMERATE_MAX = 60;
    constexpr uint32_t GOPSIZE_MIN = 30;
    constexpr uint32_t GOPSIZE_MAX = 3000;
    constexpr uint32_t BITRATE_MIN = 1000000;
    constexpr uint32_t BITRATE_MAX = 10000000;
    constexpr uint32_t BITRATE_DEFAULT_264 = 5000000;
    constexpr int32_t QP_MIN = 12;
    constexpr int32_t QP_MAX = 48;
    constexpr int32_t QP_START = 30;
    // scene change: mean motion cost per macroblock above this re-codes the picture as IDR
    constexpr uint32_t SCENE_CUT_COST_PER_MB = 3000;
}

class VideoEncoderMI355X : public VideoEncoder {
public:
    VideoEncoderMI355X();
    ~VideoEncoderMI355X() override;

    EncoderRetCode InitEncoder() override;
    EncoderRetCode StartEncoder() override;
    EncoderRetCode EncodeOneFrame(const uint8_t *inputData, uint32_t inputSize, uint8_t **outputData,
                                  uint32_t *outputSize) override;
    EncoderRetCode StopEncoder() override;
    void DestroyEncoder() override;
    EncoderRetCode ResetEncoder() override;

    EncoderRetCode ForceKeyFrame();
    EncoderRetCode SetEncodeParams();
    bool EncodeParamsChange();

    // picture QP used for the last encoded picture (test hook for the rate controller)
    int32_t LastFrameQp() const { return m_lastQp; }
    // number of pictures re-coded as IDR by the scene-change detector (test hook)
    uint32_t SceneCuts() const { return m_sceneCuts; }

private:
    struct EncodeParams {
        uint32_t framerate = 0;
        uint32_t bitrate = 0;
        uint32_t gopsize = 0;
        std::string profile = "";
        uint32_t width = 0;
        uint32_t height = 0;
    };

    bool GetRoEncParam();
    bool GetPersistEncParam();
    bool VerifyEncodeRoParams(int32_t width, int32_t height, int32_t framerate);
    bool VerifyEncodeParams(std::string &bitrate, std::string &gopsize, std::string &profile);
    bool InitParams();
    void Release();
    void RateControlUpdate(uint32_t frameBytes, bool isIdr);

    EncodeParams m_encParams = {MI355X::FRAMERATE_MIN, MI355X::BITRATE_DEFAULT_264, MI355X::GOPSIZE_MIN, "baseline",
                                MI355X::DEFAULT_WIDTH, MI355X::DEFAULT_HEIGHT};
    EncodeParams m_tmpEncParams = m_encParams;
    std::atomic<bool> m_resetFlag = { false };
    mi355x_h264_encoder *m_encoder = nullptr;
    uint32_t m_yLength = 0;
    uint32_t m_frameSize = 0;
    // rate control (RC_BITRATE_MODE of the reference preset, VideoEncoderOpenH264.cpp:274)
    int32_t m_fixedQp = -1;      // >= 0: extension property persist.vmi.video.encode.qp selects fixed QP
    int32_t m_qp = MI355X::QP_START;
    int32_t m_lastQp = 0;
    int64_t m_bufferBits = 0;    // virtual buffer fullness relative to the target rate
    bool m_sceneDetect = true;   // bEnableSceneChangeDetect = 1 in the reference preset (ref :283)
    uint32_t m_sceneCuts = 0;
};

#endif  // VIDEO_ENCODER_MI355X_H
