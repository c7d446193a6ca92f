/*
 * include/VideoCodecApi.h -- public plugin surface of libVideoCodec, source
 * compatible with the reference's /root/reference/video_codec/VideoCodecApi.h
 * (enum EncoderRetCode :8-20, abstract class VideoEncoder :22-78, the two
 * extern "C" factory functions :80-96).  Same names, same values, same virtual
 * order, so a caller built against the reference header links against this
 * library unchanged (Itanium C++ ABI).
 */
#ifndef VIDEO_CODEC_API_H
#define VIDEO_CODEC_API_H
#include <cstdint>

enum EncoderRetCode : uint32_t {
    VIDEO_ENCODER_SUCCESS = 0,
    VIDEO_ENCODER_CREATE_FAIL = 1,            /* could not create the encoder object   */
    VIDEO_ENCODER_INIT_FAIL = 2,              /* InitEncoder failed                     */
    VIDEO_ENCODER_START_FAIL = 3,
    VIDEO_ENCODER_ENCODE_FAIL = 4,            /* EncodeOneFrame failed                  */
    VIDEO_ENCODER_STOP_FAIL = 5,
    VIDEO_ENCODER_DESTROY_FAIL = 6,
    VIDEO_ENCODER_REGISTER_FAIL = 7,
    VIDEO_ENCODER_RESET_FAIL = 8,
    VIDEO_ENCODER_FORCE_KEY_FRAME_FAIL = 9,
    VIDEO_ENCODER_SET_ENCODE_PARAMS_FAIL = 10
};

class VideoEncoder {
public:
    VideoEncoder() = default;
    virtual ~VideoEncoder() = default;

    /* read and validate the configuration, create the engine */
    virtual EncoderRetCode InitEncoder() = 0;
    virtual EncoderRetCode StartEncoder() = 0;
    /* inputData: contiguous I420, at least width*height*3/2 bytes.
     * outputData: encoder-owned Annex-B access unit, valid until the next call. */
    virtual EncoderRetCode EncodeOneFrame(const uint8_t *inputData, uint32_t inputSize, uint8_t **outputData,
                                          uint32_t *outputSize) = 0;
    virtual EncoderRetCode StopEncoder() = 0;
    virtual void DestroyEncoder() = 0;
    virtual EncoderRetCode ResetEncoder() = 0;
};

extern "C" {
/* picks the backend from property ro.vmi.demo.video.encode.format
 * (reference values 0/1/2; 3 = the MI355X backend added by this build) */
EncoderRetCode CreateVideoEncoder(VideoEncoder **encoder);
/* deletes the object; a null pointer is accepted (SUCCESS with a warning) */
EncoderRetCode DestroyVideoEncoder(VideoEncoder *encoder);
}

#endif /* VIDEO_CODEC_API_H */
