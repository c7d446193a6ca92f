/*
 * media_amd/host/VideoDecoderMI355X.h -- H.264 decoder backend for the VideoDecoder plugin surface, a peer of the reference's
 * NETINT adapter (/root/reference/video_decoder/VideoDecoderNetint.h:13-31).  Same operator API and the same observable
 * behaviour at the surface (stop state, picture-size change event, copy hook); the engine beneath is the C ABI of
 * include/mi355x_h264_dec.h: host CAVLC parser + reconstruction on the MI355X.
 */
#ifndef VIDEO_DECODER_MI355X_H
#define VIDEO_DECODER_MI355X_H

#include <vector>
#include "VideoDecoder.h"
#include "mi355x_h264_dec.h"

class VideoDecoderMI355X : public VideoDecoder {
public:
    VideoDecoderMI355X() = default;
    ~VideoDecoderMI355X() override;

    DecoderRetCode CreateDecoder(MediaStreamFormat decType) override;
    DecoderRetCode InitDecoder() override;
    DecoderRetCode SetDecodeParams(DecodeParamsIndex index, void *decParams) override;
    DecoderRetCode GetDecodeParams(DecodeParamsIndex index, void *decParams) override;
    DecoderRetCode SetCallbacks(std::function<void(DecodeEventIndex, uint32_t, void *)> eventCallBack) override;
    DecoderRetCode SetCopyFrameFunc(
        std::function<uint32_t(uint8_t*, uint8_t*, const PicInfoParams &, uint32_t)> copyFrame) override;
    DecoderRetCode SendStreamData(uint8_t *buffer, uint32_t filledLen) override;
    DecoderRetCode RetrieveFrameData(uint8_t *buffer, uint32_t maxLen, uint32_t *filledLen) override;
    DecoderRetCode Flush() override;
    DecoderRetCode StartDecoder() override;
    DecoderRetCode StopDecoder() override;
    void DestroyDecoder() override;

    // test hooks
    uint64_t PicturesDecoded() const { return m_pictures; }

private:
    static constexpr uint32_t kDefaultWidth = 1280, kDefaultHeight = 720;   // the reference adapter's defaults (VideoDecoderNetint.h:34-35)

    mi355x_h264_decoder *m_engine = nullptr;
    bool m_created = false, m_stop = true, m_pending = false;
    uint32_t m_writeWidth = kDefaultWidth, m_writeHeight = kDefaultHeight;
    int32_t m_stride = static_cast<int32_t>(kDefaultWidth);
    uint64_t m_pictures = 0;
    std::vector<uint8_t> m_frame;   // the waiting picture, tight I420
    uint32_t m_frameWidth = 0, m_frameHeight = 0;
    std::function<void(DecodeEventIndex, uint32_t, void *)> m_eventCallBack;
    std::function<uint32_t(uint8_t*, uint8_t*, const PicInfoParams &, uint32_t)> m_copyFrame;
};

#endif
