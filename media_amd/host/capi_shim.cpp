// media_amd/host/capi_shim.cpp -- flat C entry points over the C++ plugin surface so
// that the Python tests can drive CreateVideoEncoder / VideoEncoder exactly as the
// (unseen) VMI caller would: Create -> Init -> Start -> Encode x N -> Stop -> Destroy.
#include <cstring>
#include "MediaLog.h"
#include "Property.h"
#include "VideoCodecApi.h"
#include "VideoEncoderMI355X.h"

extern "C" {
uint32_t vc_create(void **enc) { return CreateVideoEncoder(reinterpret_cast<VideoEncoder **>(enc)); }
uint32_t vc_delete(void *enc) { return DestroyVideoEncoder(static_cast<VideoEncoder *>(enc)); }
uint32_t vc_init(void *enc) { return static_cast<VideoEncoder *>(enc)->InitEncoder(); }
uint32_t vc_start(void *enc) { return static_cast<VideoEncoder *>(enc)->StartEncoder(); }
uint32_t vc_encode(void *enc, const uint8_t *in, uint32_t inSize, uint8_t **out, uint32_t *outSize)
{
    return static_cast<VideoEncoder *>(enc)->EncodeOneFrame(in, inSize, out, outSize);
}
uint32_t vc_stop(void *enc) { return static_cast<VideoEncoder *>(enc)->StopEncoder(); }
void vc_destroy(void *enc) { static_cast<VideoEncoder *>(enc)->DestroyEncoder(); }
uint32_t vc_reset(void *enc) { return static_cast<VideoEncoder *>(enc)->ResetEncoder(); }
int32_t vc_last_qp(void *enc)
{
    auto *m = dynamic_cast<VideoEncoderMI355X *>(static_cast<VideoEncoder *>(enc));
    return m != nullptr ? m->LastFrameQp() : -1;
}
uint32_t vc_scene_cuts(void *enc)
{
    auto *m = dynamic_cast<VideoEncoderMI355X *>(static_cast<VideoEncoder *>(enc));
    return m != nullptr ? m->SceneCuts() : 0;
}
// copies the luma reconstruction of the last picture (coded size) into dst; returns bytes or < 0 (measurement hook: PSNR)
int64_t vc_debug_recon_y(void *enc, void *dst, uint64_t cap, int32_t *codedWidth, int32_t *codedHeight)
{
    auto *m = dynamic_cast<VideoEncoderMI355X *>(static_cast<VideoEncoder *>(enc));
    if (m == nullptr) return -1;
    return m->ReadReconY(dst, static_cast<size_t>(cap), codedWidth, codedHeight);
}
void vc_prop_set(const char *key, const char *value) { SetEncParam(key, value); }
int32_t vc_prop_get_int(const char *key) { return GetIntEncParam(key); }
int32_t vc_prop_get_str(const char *key, char *buf, int32_t cap)
{
    const std::string v = GetStrEncParam(key);
    if (cap <= 0) return -1;
    std::strncpy(buf, v.c_str(), static_cast<size_t>(cap) - 1);
    buf[cap - 1] = 0;
    return static_cast<int32_t>(v.size());
}
}
