/*
 * media_amd/host/PropertyDrivenEncoder.h -- what every backend of the VideoEncoder plugin surface shares:
 * the property-driven configuration, the live re-configuration / key-frame handshake and the call contract of
 * the reference's OpenH264 adapter (/root/reference/video_codec/VideoEncoderOpenH264.h:27-196, .cpp:62-429;
 * SURVEY.md Appendix A, D).  A backend supplies the engine behind six small hooks.
 */
#ifndef PROPERTY_DRIVEN_ENCODER_H
#define PROPERTY_DRIVEN_ENCODER_H

#include <atomic>
#include <cstdint>
#include <string>
#include "VideoCodecApi.h"

class PropertyDrivenEncoder : public VideoEncoder {
public:
    // limits of the reference adapter (VideoEncoderOpenH264.h:12-25, .cpp:16-23)
    struct Limits {
        static constexpr int32_t kSideMin = 16, kSideMax = 4096;            // picture width / height
        static constexpr int32_t kFps[2] = {30, 60};                        // the only accepted frame rates
        static constexpr int32_t kGopMin = 30, kGopMax = 3000;
        static constexpr int32_t kBitrateMin = 1000000, kBitrateMax = 10000000;
    };
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

    ~PropertyDrivenEncoder() override = default;

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

protected:
    // ---- the engine behind the surface ----
    virtual const char *BackendName() const = 0;
    virtual bool EngineOpen(const Settings &s) = 0;          // create + configure; false -> InitEncoder fails
    virtual void EngineClose() = 0;                          // idempotent
    virtual bool EngineReady() const = 0;
    // one tight I420 picture (Y w*h, U, V) -> access unit in ENGINE-owned memory, valid until the next call
    virtual bool EngineEncode(const uint8_t *i420, uint8_t **out, uint32_t *outLen) = 0;
    virtual bool EngineForceIdr() = 0;

    const Settings &Active() const { return m_active; }
    uint32_t LumaBytes() const { return m_lumaBytes; }

private:
    enum class PhoneMode { Video, Instruction, Invalid };
    static PhoneMode ReadPhoneMode();
    bool ReadGeometry(Settings &into) const;          // width / height / fps   (read-only properties)
    bool ReadTunables(Settings &into);                // bitrate / gop / profile (live-adjustable properties)
    bool PollParamAdjust();                           // persist.vmi.video.encode.param_adjusting handshake
    void PollKeyframeRequest();                       // persist.vmi.video.encode.keyframe handshake

    Settings m_active;                 // what the engine was opened with
    Settings m_pending;                // last values read from the properties
    std::atomic<bool> m_needReset{false};
    uint32_t m_lumaBytes = 0, m_frameBytes = 0;
};

#endif  // PROPERTY_DRIVEN_ENCODER_H
