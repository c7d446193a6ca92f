// media_amd/host/Property.cpp -- see Property.h (mirrors /root/reference/common/prop/Property.cpp:8-44)
#include "Property.h"
#include <cctype>
#include <cstdlib>
#include <map>
#include <mutex>
#include <sstream>

namespace {
std::mutex g_lock;
std::map<std::string, std::string> g_props;

std::string EnvName(const std::string &key)
{
    std::string e;
    for (char c : key) e.push_back(c == '.' ? '_' : static_cast<char>(std::toupper(static_cast<unsigned char>(c))));
    return e;
}

std::string Lookup(const char *key)
{
    if (key == nullptr) return "";
    std::lock_guard<std::mutex> g(g_lock);
    auto it = g_props.find(key);
    if (it != g_props.end()) return it->second;
    const char *env = std::getenv(EnvName(key).c_str());
    return env != nullptr ? std::string(env) : std::string();
}
}

// stream extraction semantics on purpose (the reference parses its property text this way, Property.cpp:8-14):
// "42abc" -> 42, "abc" -> 0 (failed extraction writes 0 since C++11), "" -> the initial -1
int32_t StrToInt(std::string text)
{
    int32_t number = -1;
    std::istringstream(text) >> number;
    return number;
}

int32_t GetIntEncParam(const char *key) { return StrToInt(Lookup(key)); }
std::string GetStrEncParam(const char *key) { return Lookup(key); }

void SetEncParam(const char *key, const char *text)
{
    if (key == nullptr || text == nullptr) return;
    std::lock_guard<std::mutex> g(g_lock);
    g_props[key] = text;
}
