// media_amd/host/MediaLog.cpp -- see MediaLog.h
#include "MediaLog.h"
#include <atomic>
#include <cstdarg>
#include <cstdlib>
#include <cstdio>
#include <ctime>
#include <string>
#include <unistd.h>

namespace {
std::atomic<MediaLogCallbackFunc> g_cb{nullptr};
std::atomic<int> g_level{LOG_LEVEL_INFO};

void DefaultSink(int level, const char *tag, const char *msg)
{
    static const char *names[] = {"?", "?", "?", "D", "I", "W", "E", "F"};
    if (std::getenv("MEDIA_LOG_QUIET") != nullptr && level < LOG_LEVEL_ERROR) return;
    struct timespec ts {};
    clock_gettime(CLOCK_REALTIME, &ts);
    std::fprintf(stderr, "%ld.%03ld %d %s %s: %s\n", static_cast<long>(ts.tv_sec), ts.tv_nsec / 1000000, getpid(),
                 names[level & 7], tag, msg);
}
}

void MediaLogPrint(int level, const char *tag, const char *fmt, ...)
{
    if (level < g_level.load()) return;
    char msg[512];
    va_list ap;
    va_start(ap, fmt);
    std::vsnprintf(msg, sizeof(msg), fmt, ap);
    va_end(ap);
    const std::string fullTag = std::string("Media_") + (tag != nullptr ? tag : "");
    MediaLogCallbackFunc cb = g_cb.load();
    if (cb != nullptr) cb(level, fullTag.c_str(), msg);
    else DefaultSink(level, fullTag.c_str(), msg);
}

extern "C" void SetMediaLogCallback(MediaLogCallbackFunc logCallback)
{
    g_cb.store(logCallback);
    g_level.store(logCallback != nullptr ? LOG_LEVEL_DEBUG : LOG_LEVEL_INFO);
}
