// media_amd/host/VideoCodecApi.cpp -- factory of the plugin surface.
//
// Behaviour contract (what /root/reference/video_codec/VideoCodecApi.cpp:21-55 does): the integer property
// ro.vmi.demo.video.encode.format names the backend; an unknown value or a failed allocation yields
// VIDEO_ENCODER_CREATE_FAIL; DestroyVideoEncoder deletes the object and accepts nullptr.  The reference knows
// 0 (OpenH264), 1 (NETINT H.264) and 2 (NETINT H.265); this build registers 3 = MI355X and keeps 0: the OpenH264
// backend binds libopenh264.so at InitEncoder like the reference and fails there when the library is absent (it is,
// on this pool).  1 and 2 need NETINT hardware and its closed library: known-but-unavailable.  In the reference
// tree the maintainer keeps its switch and adds one case.
#define LOG_TAG "VideoCodecApi"
#include "VideoCodecApi.h"
#include <array>
#include <new>
#include "MediaLog.h"
#include "Property.h"
#include "VideoEncoderMI355X.h"
#ifdef HAVE_OPENH264_HEADERS
#include "OpenH264Backend.h"
#endif

namespace {

using Maker = VideoEncoder *(*)();

struct Backend {
    int32_t id;          // value of ro.vmi.demo.video.encode.format
    const char *name;
    Maker make;          // nullptr: recognised by the reference, not built into this library
};

VideoEncoder *MakeMI355X() { return new (std::nothrow) VideoEncoderMI355X(); }
#ifdef HAVE_OPENH264_HEADERS
VideoEncoder *MakeOpenH264() { return new (std::nothrow) OpenH264Backend(); }   // binds libopenh264.so in InitEncoder
constexpr Maker kOpenH264Maker = &MakeOpenH264;
#else
constexpr Maker kOpenH264Maker = nullptr;   // built without the OpenH264 ABI headers
#endif

const std::array<Backend, 4> kBackends = {{
    {0, "OpenH264 (CPU, needs libopenh264.so at run time)", kOpenH264Maker},
    {1, "NETINT T408 H.264", nullptr},
    {2, "NETINT T408 H.265", nullptr},
    {3, "AMD Instinct MI355X (HIP)", &MakeMI355X},
}};

const Backend *FindBackend(int32_t id)
{
    for (const Backend &b : kBackends) {
        if (b.id == id) {
            return &b;
        }
    }
    return nullptr;
}

}  // namespace

EncoderRetCode CreateVideoEncoder(VideoEncoder **encoder)
{
    if (encoder == nullptr) {
        ERR("CreateVideoEncoder: no place to return the encoder");
        return VIDEO_ENCODER_CREATE_FAIL;
    }
    const int32_t wanted = GetIntEncParam("ro.vmi.demo.video.encode.format");
    const Backend *backend = FindBackend(wanted);
    if (backend == nullptr) {
        ERR("CreateVideoEncoder: encoder format %d is not known", wanted);
        return VIDEO_ENCODER_CREATE_FAIL;
    }
    if (backend->make == nullptr) {
        ERR("CreateVideoEncoder: backend %d (%s) is not part of this library", wanted, backend->name);
        return VIDEO_ENCODER_CREATE_FAIL;
    }
    VideoEncoder *made = backend->make();
    if (made == nullptr) {
        ERR("CreateVideoEncoder: out of memory creating backend %d (%s)", wanted, backend->name);
        return VIDEO_ENCODER_CREATE_FAIL;
    }
    INFO("CreateVideoEncoder: backend %d (%s)", wanted, backend->name);
    *encoder = made;
    return VIDEO_ENCODER_SUCCESS;
}

EncoderRetCode DestroyVideoEncoder(VideoEncoder *encoder)
{
    if (encoder != nullptr) {
        delete encoder;
    } else {
        WARN("DestroyVideoEncoder: nothing to destroy");
    }
    return VIDEO_ENCODER_SUCCESS;
}
