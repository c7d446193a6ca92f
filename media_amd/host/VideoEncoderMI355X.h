/*
 * media_amd/host/VideoEncoderMI355X.h -- the MI355X backend of the VideoEncoder plugin surface, a peer of the
 * reference's OpenH264 adapter (/root/reference/video_codec/VideoEncoderOpenH264.h:27-196).  The shared wrapper
 * behaviour lives in PropertyDrivenEncoder; the engine here is the HIP encode path behind the C ABI of
 * include/mi355x_h264.h, plus the two pieces of host logic OpenH264 keeps inside its library: frame-level rate
 * control for the bitrate mode and the scene-change IDR.
 */
#ifndef VIDEO_ENCODER_MI355X_H
#define VIDEO_ENCODER_MI355X_H

#include "PropertyDrivenEncoder.h"
#include "mi355x_h264.h"

class VideoEncoderMI355X : public PropertyDrivenEncoder {
public:
    struct Rc {
        static constexpr int32_t kQpMin = 12, kQpMax = 48, kQpStart = 30;  // rate-control range
        static constexpr uint32_t kSceneCutCostPerMb = 3000;               // mean motion cost that triggers an IDR
    };

    VideoEncoderMI355X();
    ~VideoEncoderMI355X() override;

    // test hooks
    int32_t LastFrameQp() const { return m_lastQp; }
    uint32_t SceneCuts() const { return m_sceneCuts; }
    // measurement hook (bench.py reads the reconstruction for PSNR): luma reconstruction of the last picture, coded size
    int64_t ReadReconY(void *dst, size_t cap, int32_t *codedWidth, int32_t *codedHeight);
    bool Shared() const { return m_stream != nullptr; }

protected:
    const char *BackendName() const override { return "MI355X HIP"; }
    bool EngineOpen(const Settings &s) override;
    void EngineClose() override;
    bool EngineReady() const override { return m_engine != nullptr || m_stream != nullptr; }
    bool EngineEncode(const uint8_t *i420, uint8_t **out, uint32_t *outLen) override;
    bool EngineForceIdr() override;

private:
    int EncodePicture(const uint8_t *i420, uint8_t **out, uint32_t *outLen, int *frameType);
    void RateControlUpdate(uint32_t frameBytes, bool isIdr);
    static int32_t StartQp(uint32_t bitrate, uint32_t fps, uint32_t width, uint32_t height);

    mi355x_h264_encoder *m_engine = nullptr;   // an engine of its own (persist.vmi.video.encode.shared = 0), or
    mi355x_h264_stream *m_stream = nullptr;    // a stream of the process-wide shared engine (the default)
    // rate control (the reference preset runs RC_BITRATE_MODE, VideoEncoderOpenH264.cpp:274)
    int32_t m_fixedQp = -1;            // >= 0: extension property persist.vmi.video.encode.qp selects fixed QP
    int32_t m_qp = Rc::kQpStart, m_lastQp = 0;
    int64_t m_bufferBits = 0;          // virtual buffer fullness relative to the target rate
    int64_t m_gopLeft = 0, m_picsLeft = 0, m_meanP = 0;   // GOP budget left, P pictures left in the GOP, running mean of the P pictures' bits
    bool m_sceneDetect = true;         // bEnableSceneChangeDetect = 1 in the reference preset (ref :283)
    uint32_t m_sceneCuts = 0;
};

#endif  // VIDEO_ENCODER_MI355X_H
