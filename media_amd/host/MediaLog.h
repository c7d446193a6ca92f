/*
 * media_amd/host/MediaLog.h -- the logging call sites of the plugin surface
 * (DBG/INFO/WARN/ERR/FATAL macros + user callback), a small stand-in for the
 * reference's /root/reference/common/log (MediaLog.h:43-47, MediaLogDefs.h:15,
 * MediaLogManager.cpp:44-82): printf-style message, 512-byte buffer, tag
 * "Media_" + LOG_TAG, default level INFO, setting a callback lowers it to DEBUG.
 */
#ifndef MEDIA_AMD_MEDIALOG_H
#define MEDIA_AMD_MEDIALOG_H

enum MediaLogLevel { LOG_LEVEL_DEBUG = 3, LOG_LEVEL_INFO = 4, LOG_LEVEL_WARN = 5, LOG_LEVEL_ERROR = 6, LOG_LEVEL_FATAL = 7 };
using MediaLogCallbackFunc = void (*)(int level, const char *tag, const char *fmt);

void MediaLogPrint(int level, const char *tag, const char *fmt, ...) __attribute__((format(printf, 3, 4)));
extern "C" void SetMediaLogCallback(MediaLogCallbackFunc logCallback);

#ifndef LOG_TAG
#define LOG_TAG "Media"
#endif
#define DBG(fmt, ...) MediaLogPrint(LOG_LEVEL_DEBUG, LOG_TAG, fmt, ##__VA_ARGS__)
#define INFO(fmt, ...) MediaLogPrint(LOG_LEVEL_INFO, LOG_TAG, fmt, ##__VA_ARGS__)
#define WARN(fmt, ...) MediaLogPrint(LOG_LEVEL_WARN, LOG_TAG, fmt, ##__VA_ARGS__)
#define ERR(fmt, ...) MediaLogPrint(LOG_LEVEL_ERROR, LOG_TAG, fmt, ##__VA_ARGS__)
#define FATAL(fmt, ...) MediaLogPrint(LOG_LEVEL_FATAL, LOG_TAG, fmt, ##__VA_ARGS__)

#endif
