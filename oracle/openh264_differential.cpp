// oracle/openh264_differential.cpp -- run-time differential against a REAL libopenh264.so, if the box has one.
// TEST INFRASTRUCTURE ONLY (SURVEY.md 8c item iv, BASELINE.md "oracle availability").
//
// The reference binds OpenH264 by dlopen("libopenh264.so") + dlsym of WelsCreateSVCEncoder / WelsDestroySVCEncoder
// (/root/reference/video_codec/VideoEncoderOpenH264.cpp:197-226) and configures it with the preset of
// InitParams / InitParamExt (:228-296).  This tool does the same through the three ABI headers the reference vendors
// (vendor/openh264/, used where they lie: see oracle/Makefile, target _ref) and encodes a raw I420 file with it:
//
//   openh264_differential <in.i420> <width> <height> <fps> <bitrate> <gop> <frames> [out.264]
//
// Output: one JSON line.  {"oracle": "absent"} when no libopenh264.so can be loaded - no OpenH264 number is then
// claimed anywhere (bench.py keeps its own CPU restatement as the baseline and says so).  With a library:
// {"oracle": "openh264", "frames": n, "bytes": total, "seconds": t, "fps": n / t, "first_frame_bytes": ...}; the
// Annex-B stream goes to out.264 for byte comparison with this build's output.
// Neither this container nor the GPU box ships the library, so only the "absent" branch has
// ever run; the other branch is written against the headers alone.
#include <dlfcn.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "codec_api.h"

namespace {
using CreateFn = int (*)(ISVCEncoder**);
using DestroyFn = void (*)(ISVCEncoder*);

int Absent(const char* why)
{
    std::printf("{\"oracle\": \"absent\", \"reason\": \"%s\"}\n", why);
    return 0;
}
}  // namespace

int main(int argc, char** argv)
{
    void* lib = dlopen("libopenh264.so", RTLD_LAZY);
    if (lib == nullptr) {
        return Absent("dlopen(libopenh264.so) failed");
    }
    auto create = reinterpret_cast<CreateFn>(dlsym(lib, "WelsCreateSVCEncoder"));
    auto destroy = reinterpret_cast<DestroyFn>(dlsym(lib, "WelsDestroySVCEncoder"));
    if (create == nullptr || destroy == nullptr) {
        return Absent("library lacks WelsCreateSVCEncoder / WelsDestroySVCEncoder");
    }
    if (argc < 8) {
        std::fprintf(stderr, "usage: %s in.i420 width height fps bitrate gop frames [out.264]\n", argv[0]);
        return 2;
    }
    const int w = std::atoi(argv[2]), h = std::atoi(argv[3]), fps = std::atoi(argv[4]), bitrate = std::atoi(argv[5]);
    const int gop = std::atoi(argv[6]), frames = std::atoi(argv[7]);
    FILE* in = std::fopen(argv[1], "rb");
    FILE* out = argc > 8 ? std::fopen(argv[8], "wb") : nullptr;
    if (in == nullptr) {
        std::fprintf(stderr, "cannot open %s\n", argv[1]);
        return 2;
    }
    ISVCEncoder* enc = nullptr;
    if (create(&enc) != 0 || enc == nullptr) {
        return Absent("WelsCreateSVCEncoder failed");
    }
    // the reference preset, field by field
    SEncParamExt p;
    enc->GetDefaultParams(&p);
    p.iUsageType = CAMERA_VIDEO_REAL_TIME;
    p.iRCMode = RC_BITRATE_MODE;
    p.iPicWidth = w;
    p.iPicHeight = h;
    p.iTargetBitrate = bitrate;
    p.iMaxBitrate = bitrate;
    p.fMaxFrameRate = static_cast<float>(fps);
    p.uiIntraPeriod = static_cast<unsigned>(gop);
    p.iTemporalLayerNum = 1;
    p.iSpatialLayerNum = 1;
    p.sSpatialLayers[0].iVideoWidth = w;
    p.sSpatialLayers[0].iVideoHeight = h;
    p.sSpatialLayers[0].fFrameRate = static_cast<float>(fps);
    p.sSpatialLayers[0].iSpatialBitrate = bitrate;
    p.sSpatialLayers[0].iMaxSpatialBitrate = bitrate;
    p.sSpatialLayers[0].sSliceArgument.uiSliceMode = SM_SINGLE_SLICE;
    p.sSpatialLayers[0].uiProfileIdc = PRO_BASELINE;
    p.sSpatialLayers[0].uiLevelIdc = LEVEL_3_2;
    p.iComplexityMode = HIGH_COMPLEXITY;
    p.iNumRefFrame = 1;
    p.iEntropyCodingModeFlag = 1;
    p.iMultipleThreadIdc = 1;
    p.iLoopFilterDisableIdc = 0;
    p.eSpsPpsIdStrategy = CONSTANT_ID;
    p.bPrefixNalAddingCtrl = false;
    p.bSimulcastAVC = false;
    p.iPaddingFlag = 0;
    p.uiMaxNalSize = 0;
    p.bEnableDenoise = false;
    p.bEnableBackgroundDetection = true;
    p.bEnableSceneChangeDetect = true;
    p.bEnableAdaptiveQuant = false;
    p.bEnableFrameSkip = false;
    p.bEnableLongTermReference = false;
    p.iLTRRefNum = 0;
    p.iLtrMarkPeriod = 30;
    p.bIsLosslessLink = false;
    if (enc->InitializeExt(&p) != 0) {
        destroy(enc);
        return Absent("InitializeExt rejected the reference preset");
    }
    int fmt = videoFormatI420;
    enc->SetOption(ENCODER_OPTION_DATAFORMAT, &fmt);

    const size_t ysz = static_cast<size_t>(w) * h, fsz = ysz * 3 / 2;
    std::vector<unsigned char> buf(fsz);
    SSourcePicture pic;
    std::memset(&pic, 0, sizeof(pic));
    pic.iColorFormat = videoFormatI420;
    pic.iPicWidth = w;
    pic.iPicHeight = h;
    pic.iStride[0] = w;
    pic.iStride[1] = pic.iStride[2] = w / 2;
    pic.pData[0] = buf.data();
    pic.pData[1] = buf.data() + ysz;
    pic.pData[2] = buf.data() + ysz + ysz / 4;
    SFrameBSInfo info;
    long long total = 0, first = 0;
    int done = 0;
    double seconds = 0;
    for (int i = 0; i < frames; i++) {
        if (std::fread(buf.data(), 1, fsz, in) != fsz) {
            break;
        }
        std::memset(&info, 0, sizeof(info));
        pic.uiTimeStamp = static_cast<long long>(i) * 1000 / (fps > 0 ? fps : 30);
        const auto t0 = std::chrono::steady_clock::now();
        const int rc = enc->EncodeFrame(&pic, &info);
        seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (rc != 0) {
            break;
        }
        // the reference hands out sLayerInfo[0].pBsBuf with iFrameSizeInBytes (:349-350): layers are contiguous
        if (out != nullptr && info.iFrameSizeInBytes > 0) {
            std::fwrite(info.sLayerInfo[0].pBsBuf, 1, static_cast<size_t>(info.iFrameSizeInBytes), out);
        }
        total += info.iFrameSizeInBytes;
        if (i == 0) {
            first = info.iFrameSizeInBytes;
        }
        done++;
    }
    std::printf("{\"oracle\": \"openh264\", \"frames\": %d, \"bytes\": %lld, \"seconds\": %.6f, \"fps\": %.3f, "
                "\"first_frame_bytes\": %lld, \"threads\": 1}\n",
                done, total, seconds, seconds > 0 ? done / seconds : 0.0, first);
    enc->Uninitialize();
    destroy(enc);
    if (out != nullptr) {
        std::fclose(out);
    }
    std::fclose(in);
    return 0;
}
