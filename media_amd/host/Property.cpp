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

int32_t StrToInt(std::string inputValue)
{
    std::stringstream strStream;
    strStream << inputValue;
    int32_t result = -1;
    strStream >> result;
    return result;
}

int32_t GetIntEncParam(const char *inputValue) { return StrToInt(Lookup(inputValue)); }
std::string GetStrEncParam(const char *inputValue) { return Lookup(inputValue); }

void SetEncParam(const char *key, const char *value)
{
    if (key == nullptr || value == nullptr) return;
    std::lock_guard<std::mutex> g(g_lock);
    g_props[key] = value;
}
