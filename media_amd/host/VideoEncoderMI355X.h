/*
 * media_amd/host/VideoEncoderMI355X.h -- the MI355X backend of the VideoEncoder plugin surface, a peer of the
 * reference's OpenH264 adapter (/root/reference/video_codec/VideoEncoderOpenH264.h:27-196): same operator
 * API, same property-driven configuration, same live re-configuration handshake and buffer ownership.  The
 * engine behind it is the HIP encode path reached through the C ABI of include/mi355x_h264.h.
 */
#ifndef VIDEO_ENCODER_MI355X_H
#define VIDEO_ENCODER_MI355X_H

#include <atomic>
#include <cstdint>
#include <string>
#include "VideoCodecApi.h"
#include "mi355x_h264.h"

class VideoEncoderMI355X : public VideoEncoder {
public:
    // limits of the reference adapter (VideoEncoderOpenH264.h:12-25, .cpp:16-23) and this backend's own
    struct Limits {
        static constexpr int32_t kSideMin = 16, kSideMax = 4096;            // picture width / height
        static constexpr int32_t kFps[2] = {30, 60};                        // the only accepted frame rates
        static constexpr int32_t kGopMin = 30, kGopMax = 3000;
        static constexpr int32_t kBitrateMin = 1000000, kBitrateMax = 10000000;
        static constexpr int32_t kQpMin = 12, kQpMax = 48, kQpStart = 30;  // rate-control range
        static constexpr uint32_t kSceneCutCostPerMb = 3000;               // mean motion cost that triggers an IDR
    };

    VideoEncoderMI355X();
    ~VideoEncoderMI355X() override;

    EncoderRetCode InitEncoder() override;
    EncoderRetCode StartEncoder() override;
    EncoderRetCode EncodeOneFrame(const uint8_t *inputData, uint32_t inputSize, uint8_t **outputData,
                                  uint32_t *outputSize) override;
    EncoderRetCode StopEncoder() override;
    void DestroyEncoder() override;
    EncoderRetCode ResetEncoder() override;

    // same extras as the reference adapter exposes (VideoEncoderOpenH264.h:84-101)
    EncoderRetCode ForceKeyFrame();
    EncoderRetCode SetEncodeParams();
    bool EncodeParamsChange();

    // test hooks
    int32_t LastFrameQp() const { return m_lastQp; }
    uint32_t SceneCuts() const { return m_sceneCuts; }

private:
    // what the properties configure; defaults = reference defaults (720x1280 portrait, 30 fps, 5 Mbps, GOP 30)
    struct Settings {
        uint32_t width = 720, height = 1280, fps = 30;
        uint32_t bitrate = 5000000, gop = 30;
        std::string profile = "baseline";
        bool SameAs(const Settings &o) const
        {
            return width == o.width && height == o.height && fps == o.fps && bitrate == o.bitrate && gop == o.gop &&
                   profile == o.profile;
        }
    };
    enum class PhoneMode { Video, Instruction, Invalid };

    static PhoneMode ReadPhoneMode();
    bool ReadGeometry(Settings &into) const;          // width / height / fps   (read-only properties)
    bool ReadTunables(Settings &into);                // bitrate / gop / profile (live-adjustable properties)
    bool OpenEngine();
    void CloseEngine();
    bool PollParamAdjust();                           // persist.vmi.video.encode.param_adjusting handshake
    void PollKeyframeRequest();                       // persist.vmi.video.encode.keyframe handshake
    int EncodePicture(const uint8_t *i420, uint8_t **out, uint32_t *outLen, int *frameType);
    void RateControlUpdate(uint32_t frameBytes, bool isIdr);

    Settings m_active;                 // what the engine was opened with
    Settings m_pending;                // last values read from the properties
    std::atomic<bool> m_needReset{false};
    mi355x_h264_encoder *m_engine = nullptr;
    uint32_t m_lumaBytes = 0, m_frameBytes = 0;
    // rate control (the reference preset runs RC_BITRATE_MODE, VideoEncoderOpenH264.cpp:274)
    int32_t m_fixedQp = -1;            // >= 0: extension property persist.vmi.video.encode.qp selects fixed QP
    int32_t m_qp = Limits::kQpStart, m_lastQp = 0;
    int64_t m_bufferBits = 0;          // virtual buffer fullness relative to the target rate
    bool m_sceneDetect = true;         // bEnableSceneChangeDetect = 1 in the reference preset (ref :283)
    uint32_t m_sceneCuts = 0;
};

#endif  // VIDEO_ENCODER_MI355X_H
