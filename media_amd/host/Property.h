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

// value of an integer-valued key; -1 when the key is unset or its text is not a number
int32_t GetIntEncParam(const char *key);
// text of a key; "" when unset
std::string GetStrEncParam(const char *key);
// create or overwrite
void SetEncParam(const char *key, const char *text);
// the reference's number parse: leading integer of the text, 0 for junk, -1 for an empty string
int32_t StrToInt(std::string text);

#endif
