/*
 * media_amd/host/Property.h -- the configuration channel of the plugin surface.
 * Same four functions as the reference's /root/reference/common/prop/Property.h:6-9.
 * The reference reads Android system properties (Property.cpp:8-44, bionic only);
 * here the store is an in-process map, seeded on first use from environment
 * variables: key "ro.hardware.width" <- env "RO_HARDWARE_WIDTH" (dots become
 * underscores, upper case).  Junk parses to -1 exactly as in the reference.
 */
#ifndef MEDIA_AMD_PROPERTY_H
#define MEDIA_AMD_PROPERTY_H
#include <cstdint>
#include <string>

int32_t GetIntEncParam(const char *inputValue);
std::string GetStrEncParam(const char *inputValue);
void SetEncParam(const char *key, const char *value);
int32_t StrToInt(std::string inputValue);

#endif
