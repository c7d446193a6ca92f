// tests/boundary/ref_header_caller.cpp -- a caller of the plugin surface that is compiled against the REFERENCE's own
// header, included where it lies (/root/reference/video_codec/VideoCodecApi.h:8-96; -I given by the recipe in
// oracle/Makefile, output oracle/_ref/ref_header_caller), and linked with this build's libVideoCodec.so.  It proves the
// drop-in claim of SURVEY.md 8(b) at the ABI level: same enum values, same vtable order, same two extern "C" symbols.
// Nothing of this build's include/ is visible to this translation unit.
//
// It drives the sequence the (unseen) VMI caller makes: CreateVideoEncoder -> InitEncoder -> StartEncoder ->
// EncodeOneFrame x N -> ResetEncoder -> EncodeOneFrame -> StopEncoder -> DestroyEncoder -> DestroyVideoEncoder.
// Configuration reaches the library the way the reference reads it: properties (Appendix A), which this build's
// Property.cpp seeds from environment variables (RO_HARDWARE_WIDTH ...), set by the test.
//
// usage: ref_header_caller <i420 file> <width> <height> <frames> <out file>
// prints one JSON line: {"create":..,"init":..,"start":..,"encode":[..],"sizes":[..],"short_input":..,"reset":..,"stop":..,"destroy_null":..,"destroy":..}
// out file: every access unit as [u32 little-endian length][bytes].
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "VideoCodecApi.h"

int main(int argc, char **argv)
{
    if (argc != 6) { fprintf(stderr, "usage: %s in.i420 w h frames out.bin\n", argv[0]); return 2; }
    const int w = atoi(argv[2]), h = atoi(argv[3]), frames = atoi(argv[4]);
    const size_t fsz = (size_t)w * h * 3 / 2;
    FILE *in = fopen(argv[1], "rb");
    FILE *out = fopen(argv[5], "wb");
    if (in == nullptr || out == nullptr) { fprintf(stderr, "cannot open files\n"); return 2; }
    std::vector<uint8_t> buf(fsz);

    VideoEncoder *enc = nullptr;
    const uint32_t rcCreate = CreateVideoEncoder(&enc);
    printf("{\"create\":%u", rcCreate);
    if (rcCreate != VIDEO_ENCODER_SUCCESS || enc == nullptr) { printf("}\n"); return 0; }
    const uint32_t rcInit = enc->InitEncoder();
    printf(",\"init\":%u", rcInit);
    if (rcInit == VIDEO_ENCODER_SUCCESS) {
        printf(",\"start\":%u,\"encode\":[", (uint32_t)enc->StartEncoder());
        std::vector<uint32_t> sizes;
        for (int i = 0; i < frames; i++) {
            if (fread(buf.data(), 1, fsz, in) != fsz) break;
            uint8_t *au = nullptr;
            uint32_t n = 0;
            const uint32_t rc = enc->EncodeOneFrame(buf.data(), (uint32_t)fsz, &au, &n);
            printf("%s%u", i ? "," : "", rc);
            if (rc == VIDEO_ENCODER_SUCCESS) {
                fwrite(&n, 4, 1, out);
                fwrite(au, 1, n, out);   // encoder-owned buffer, valid until the next call (ref VideoEncoderOpenH264.cpp:349)
                sizes.push_back(n);
            }
        }
        printf("],\"sizes\":[");
        for (size_t i = 0; i < sizes.size(); i++) printf("%s%u", i ? "," : "", sizes[i]);
        uint8_t *au = nullptr;
        uint32_t n = 0;
        // size guard: inputSize < w*h*3/2 -> ENCODE_FAIL (ref :307)
        printf("],\"short_input\":%u", (uint32_t)enc->EncodeOneFrame(buf.data(), (uint32_t)fsz - 1, &au, &n));
        printf(",\"reset\":%u", (uint32_t)enc->ResetEncoder());
        const uint32_t rc = enc->EncodeOneFrame(buf.data(), (uint32_t)fsz, &au, &n);
        printf(",\"after_reset\":%u,\"after_reset_nal\":%d", rc, rc == VIDEO_ENCODER_SUCCESS && n > 4 ? (au[4] & 31) : -1);
        printf(",\"stop\":%u", (uint32_t)enc->StopEncoder());
    }
    enc->DestroyEncoder();
    enc->DestroyEncoder();   // idempotent (ref :381)
    printf(",\"destroy_null\":%u", (uint32_t)DestroyVideoEncoder(nullptr));   // SUCCESS with a warning (ref VideoCodecApi.cpp:48-51)
    printf(",\"destroy\":%u}\n", (uint32_t)DestroyVideoEncoder(enc));
    fclose(in);
    fclose(out);
    return 0;
}
