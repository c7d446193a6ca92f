// tools/plugin_bench.cpp -- native measurement of the drop-in boundary in the reference's operating mode: S host threads, each
// CreateVideoEncoder -> InitEncoder -> StartEncoder -> EncodeOneFrame x F on host I420 pictures (bitrate mode, scene detection
// on), compiled against include/VideoCodecApi.h and linked with libVideoCodec.so.  No Python in the timed region (bench.py's
// own plugin harness holds the interpreter lock between calls); bench.py --mode plugin runs this binary when it has been built
// (media_amd/lib/plugin_bench, recipe in media_amd/host/Makefile).  Configuration reaches the library through the property
// store, seeded from environment variables as in tests/boundary/ref_header_caller.cpp.
//
// usage: plugin_bench <i420 file with N pictures> <width> <height> <N> <frames per stream> <S1,S2,...>
// prints one JSON object per S on its own line.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
#include "VideoCodecApi.h"

int main(int argc, char **argv)
{
    if (argc != 7) { fprintf(stderr, "usage: %s in.i420 w h pictures frames_per_stream S1,S2,...\n", argv[0]); return 2; }
    const int w = atoi(argv[2]), h = atoi(argv[3]), npic = atoi(argv[4]), frames = atoi(argv[5]);
    const size_t fsz = (size_t)w * h * 3 / 2;
    std::vector<uint8_t> pics(fsz * npic);
    FILE *in = fopen(argv[1], "rb");
    if (in == nullptr || fread(pics.data(), 1, pics.size(), in) != pics.size()) { fprintf(stderr, "cannot read %s\n", argv[1]); return 2; }
    fclose(in);
    std::vector<int> counts;
    for (char *tok = strtok(argv[6], ","); tok != nullptr; tok = strtok(nullptr, ",")) counts.push_back(atoi(tok));
    for (int S : counts) {
        std::vector<VideoEncoder *> encs(S, nullptr);
        bool ok = true;
        for (int k = 0; k < S && ok; k++) {
            ok = CreateVideoEncoder(&encs[k]) == VIDEO_ENCODER_SUCCESS && encs[k] != nullptr && encs[k]->InitEncoder() == VIDEO_ENCODER_SUCCESS &&
                 encs[k]->StartEncoder() == VIDEO_ENCODER_SUCCESS;
            if (ok) {   // warm-up outside the clock: first IDR, allocations
                uint8_t *au = nullptr;
                uint32_t n = 0;
                ok = encs[k]->EncodeOneFrame(pics.data() + fsz * (size_t)(k % npic), (uint32_t)fsz, &au, &n) == VIDEO_ENCODER_SUCCESS;
            }
        }
        if (!ok) { printf("{\"streams\":%d,\"error\":\"an encoder could not be opened\"}\n", S); continue; }
        std::vector<std::vector<double>> lat(S);
        std::vector<uint64_t> bytes(S, 0);
        std::atomic<int> failures{0};
        auto work = [&](int k) {
            lat[k].reserve(frames);
            for (int i = 0; i < frames; i++) {
                const uint8_t *f = pics.data() + fsz * (size_t)((k + 1 + i) % npic);   // (the pool holds frames + S + 1 pictures: no wrap)
                uint8_t *au = nullptr;
                uint32_t n = 0;
                const auto t0 = std::chrono::steady_clock::now();
                const EncoderRetCode rc = encs[k]->EncodeOneFrame(f, (uint32_t)fsz, &au, &n);
                lat[k].push_back(std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
                if (rc != VIDEO_ENCODER_SUCCESS) failures++;
                else bytes[k] += n;
            }
        };
        std::vector<std::thread> ths;
        const auto t0 = std::chrono::steady_clock::now();
        for (int k = 0; k < S; k++) ths.emplace_back(work, k);
        for (auto &t : ths) t.join();
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        for (auto *e : encs) { e->StopEncoder(); e->DestroyEncoder(); DestroyVideoEncoder(e); }
        std::vector<double> all;
        uint64_t total = 0;
        for (int k = 0; k < S; k++) { all.insert(all.end(), lat[k].begin(), lat[k].end()); total += bytes[k]; }
        std::sort(all.begin(), all.end());
        const size_t n = all.size();
        printf("{\"streams\":%d,\"fps_aggregate\":%.1f,\"fps_per_stream\":%.1f,\"latency_ms_p50\":%.3f,\"latency_ms_p99\":%.3f,\"bytes_per_picture\":%.1f,"
               "\"bitrate_achieved\":%.0f,\"pictures\":%zu,\"encode_failures\":%d}\n",
               S, n / dt, n / dt / S, all[n / 2], all[std::min(n - 1, (size_t)(n * 0.99))], (double)total / n, (double)total * 8 * 30 / n, n, failures.load());
        fflush(stdout);
    }
    return 0;
}
