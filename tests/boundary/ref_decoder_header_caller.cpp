// tests/boundary/ref_decoder_header_caller.cpp -- a caller of the decoder plugin surface compiled against the REFERENCE's own
// header, included where it lies (/root/reference/video_decoder/include/VideoDecoder.h:10-195; -I given by the recipe in
// oracle/Makefile, output oracle/_ref/ref_decoder_header_caller), and linked with this build's libVideoDecoder.so: same enum
// values, same parameter structs, same vtable order, same two extern "C" symbols.  Nothing of this build's include/ is visible here.
//
// It drives the sequence an OMX component makes (VideoDecoderNetint.cpp is the reference's implementation of the other
// side): CreateVideoDecoder -> CreateDecoder(AVC) -> InitDecoder -> SetCallbacks / SetCopyFrameFunc -> StartDecoder ->
// (SendStreamData, RetrieveFrameData) x N -> Flush -> StopDecoder -> DestroyVideoDecoder.  The configured picture size starts
// at the adapter's default, so the first RetrieveFrameData answers BAD_PIC_SIZE and raises INDEX_PIC_INFO_CHANGE; the caller
// then sets the size it was told and asks again - the reference's own protocol (VideoDecoderNetint.cpp:673-685).
//
// usage: ref_decoder_header_caller <in: [u32 length][access unit] ...> <out: tight I420 pictures>
// prints one JSON line with every return code.
#include <stdint.h>   // (the reference header uses uint32_t without including it; its own users include other headers first)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "VideoDecoder.h"

int main(int argc, char **argv)
{
    if (argc != 3) { fprintf(stderr, "usage: %s in.aus out.i420\n", argv[0]); return 2; }
    FILE *in = fopen(argv[1], "rb");
    FILE *out = fopen(argv[2], "wb");
    if (in == nullptr || out == nullptr) { fprintf(stderr, "cannot open files\n"); return 2; }
    VideoDecoder *dec = nullptr;
    const uint32_t rcCreate = CreateVideoDecoder(&dec);
    printf("{\"create\":%u", rcCreate);
    if (rcCreate != VIDEO_DECODER_SUCCESS || dec == nullptr) { printf("}\n"); return 0; }
    printf(",\"hevc\":%u", (uint32_t)dec->CreateDecoder(STREAM_FORMAT_HEVC));
    const uint32_t rcAvc = dec->CreateDecoder(STREAM_FORMAT_AVC);
    const uint32_t rcInit = dec->InitDecoder();
    printf(",\"avc\":%u,\"init\":%u", rcAvc, rcInit);
    PicInfoParams told;
    uint32_t events = 0;
    dec->SetCallbacks([&](DecodeEventIndex idx, uint32_t, void *data) {
        if (idx == INDEX_PIC_INFO_CHANGE && data != nullptr) { told = *static_cast<PicInfoParams *>(data); events++; }
    });
    dec->SetCopyFrameFunc([](uint8_t *src, uint8_t *dst, const PicInfoParams &p, uint32_t cap) -> uint32_t {
        const uint32_t n = p.width * p.height * 3 / 2;
        if (n > cap) return 0;
        memcpy(dst, src, n);
        return n;
    });
    PortFormatParams pf;
    pf.port = OUT_PORT;
    const uint32_t rcPort = dec->GetDecodeParams(INDEX_PORT_FORMAT_INFO, &pf);
    printf(",\"port\":%u,\"out_format\":%d", rcPort, pf.format);
    uint8_t probe[16] = {0};
    uint32_t n = 0;
    printf(",\"send_before_start\":%u", (uint32_t)dec->SendStreamData(probe, sizeof(probe)));
    const uint32_t rcStart = dec->StartDecoder();
    printf(",\"start\":%u", rcStart);
    if (rcStart == VIDEO_DECODER_SUCCESS) {
        std::vector<uint8_t> au, frame(4096 * 2304 * 3 / 2);
        printf(",\"underflow\":%u,\"steps\":[", (uint32_t)dec->RetrieveFrameData(frame.data(), (uint32_t)frame.size(), &n));
        uint32_t len = 0;
        bool first = true;
        while (fread(&len, 4, 1, in) == 1) {
            au.resize(len);
            if (fread(au.data(), 1, len, in) != len) break;
            const uint32_t s = dec->SendStreamData(au.data(), len);
            uint32_t r = dec->RetrieveFrameData(frame.data(), (uint32_t)frame.size(), &n);
            uint32_t r2 = 99;
            if (r == VIDEO_DECODER_BAD_PIC_SIZE) {   // told the real size: configure it and ask again
                dec->SetDecodeParams(INDEX_PIC_INFO, &told);
                r2 = dec->RetrieveFrameData(frame.data(), (uint32_t)frame.size(), &n);
            }
            if (r == VIDEO_DECODER_SUCCESS || r2 == VIDEO_DECODER_SUCCESS) fwrite(frame.data(), 1, n, out);
            printf("%s[%u,%u,%u]", first ? "" : ",", s, r, r2);
            first = false;
        }
        printf("],\"events\":%u,\"width\":%u,\"height\":%u", events, told.width, told.height);
        const uint32_t rcFlush = dec->Flush();
        const uint32_t rcStop = dec->StopDecoder();
        const uint32_t rcAfter = dec->SendStreamData(probe, sizeof(probe));
        printf(",\"flush\":%u,\"stop\":%u,\"send_after_stop\":%u", rcFlush, rcStop, rcAfter);
    }
    const uint32_t rcNull = DestroyVideoDecoder(nullptr);
    const uint32_t rcDestroy = DestroyVideoDecoder(dec);
    printf(",\"destroy_null\":%u,\"destroy\":%u}\n", rcNull, rcDestroy);
    fclose(in);
    fclose(out);
    return 0;
}
