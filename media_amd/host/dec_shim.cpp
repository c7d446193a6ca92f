// media_amd/host/dec_shim.cpp -- flat C entry points over the C++ VideoDecoder surface so that the Python tests can drive
// CreateVideoDecoder / VideoDecoder exactly as an OMX component would: Create -> CreateDecoder -> Init -> SetCallbacks /
// SetCopyFrameFunc -> Start -> (SendStreamData, RetrieveFrameData) x N -> Stop -> Destroy.
#include <cstring>
#include "VideoDecoder.h"

extern "C" {
struct vd_events { uint32_t count, width, height, stride; };

uint32_t vd_create(void **dec) { return CreateVideoDecoder(reinterpret_cast<VideoDecoder **>(dec)); }
uint32_t vd_delete(void *dec) { return DestroyVideoDecoder(static_cast<VideoDecoder *>(dec)); }
uint32_t vd_create_decoder(void *dec, uint32_t type) { return static_cast<VideoDecoder *>(dec)->CreateDecoder(static_cast<MediaStreamFormat>(type)); }
uint32_t vd_init(void *dec) { return static_cast<VideoDecoder *>(dec)->InitDecoder(); }
uint32_t vd_start(void *dec) { return static_cast<VideoDecoder *>(dec)->StartDecoder(); }
uint32_t vd_stop(void *dec) { return static_cast<VideoDecoder *>(dec)->StopDecoder(); }
void vd_destroy(void *dec) { static_cast<VideoDecoder *>(dec)->DestroyDecoder(); }
uint32_t vd_flush(void *dec) { return static_cast<VideoDecoder *>(dec)->Flush(); }
uint32_t vd_send(void *dec, uint8_t *buf, uint32_t len) { return static_cast<VideoDecoder *>(dec)->SendStreamData(buf, len); }
uint32_t vd_retrieve(void *dec, uint8_t *buf, uint32_t cap, uint32_t *len) { return static_cast<VideoDecoder *>(dec)->RetrieveFrameData(buf, cap, len); }
uint32_t vd_set_pic_info(void *dec, uint32_t w, uint32_t h, int32_t stride)
{
    PicInfoParams p;
    p.width = w; p.height = h; p.stride = stride; p.scanLines = h;
    return static_cast<VideoDecoder *>(dec)->SetDecodeParams(INDEX_PIC_INFO, &p);
}
uint32_t vd_get_pic_info(void *dec, uint32_t *out4)
{
    PicInfoParams p;
    const uint32_t rc = static_cast<VideoDecoder *>(dec)->GetDecodeParams(INDEX_PIC_INFO, &p);
    out4[0] = p.width; out4[1] = p.height; out4[2] = static_cast<uint32_t>(p.stride); out4[3] = p.scanLines;
    return rc;
}
uint32_t vd_get_port_format(void *dec, uint32_t port, int32_t *format)
{
    PortFormatParams p;
    p.port = static_cast<DecoderPort>(port);
    const uint32_t rc = static_cast<VideoDecoder *>(dec)->GetDecodeParams(INDEX_PORT_FORMAT_INFO, &p);
    *format = p.format;
    return rc;
}
uint32_t vd_get_align(void *dec, uint32_t *out2)
{
    AlignInfoParams p;
    const uint32_t rc = static_cast<VideoDecoder *>(dec)->GetDecodeParams(INDEX_ALIGN_INFO, &p);
    out2[0] = p.widthAlign; out2[1] = p.heightAlign;
    return rc;
}
// installs an event callback that records picture-size changes into *ev, and a copy hook that copies the tight I420 picture
// row by row at the configured stride (what the OMX component's hook does for PIXEL_FORMAT_YUV_420P)
uint32_t vd_install_hooks(void *dec, vd_events *ev)
{
    auto *d = static_cast<VideoDecoder *>(dec);
    uint32_t rc = d->SetCallbacks([ev](DecodeEventIndex idx, uint32_t, void *data) {
        if (idx == INDEX_PIC_INFO_CHANGE && data != nullptr) {
            const auto *p = static_cast<const PicInfoParams *>(data);
            ev->count++; ev->width = p->width; ev->height = p->height; ev->stride = static_cast<uint32_t>(p->stride);
        }
    });
    if (rc != VIDEO_DECODER_SUCCESS) return rc;
    return d->SetCopyFrameFunc([](uint8_t *src, uint8_t *dst, const PicInfoParams &p, uint32_t cap) -> uint32_t {
        const uint32_t stride = static_cast<uint32_t>(p.stride), need = stride * p.scanLines * 3 / 2;
        if (need > cap) return 0;
        uint8_t *o = dst;
        for (int pl = 0; pl < 3; pl++) {
            const uint32_t w = pl ? p.width / 2 : p.width, h = pl ? p.height / 2 : p.height, st = pl ? stride / 2 : stride;
            for (uint32_t y = 0; y < h; y++) { std::memcpy(o, src, w); o += st; src += w; }
            o += st * ((pl ? p.scanLines / 2 : p.scanLines) - h);
        }
        return need;
    });
}
}
