/* media_amd/host/OpenH264Backend.h -- factory type 0 (see OpenH264Backend.cpp); no OpenH264 type appears here, so
 * the factory can name the class whether or not the backend was compiled in. */
#ifndef OPENH264_BACKEND_H
#define OPENH264_BACKEND_H

#include "PropertyDrivenEncoder.h"

class OpenH264Backend : public PropertyDrivenEncoder {
public:
    OpenH264Backend();
    ~OpenH264Backend() override;

protected:
    const char *BackendName() const override { return "OpenH264 (CPU, libopenh264.so)"; }
    bool EngineOpen(const Settings &s) override;
    void EngineClose() override;
    bool EngineReady() const override;
    bool EngineEncode(const uint8_t *i420, uint8_t **out, uint32_t *outLen) override;
    bool EngineForceIdr() override;

private:
    struct State;
    State *m_state;
};

#endif  // OPENH264_BACKEND_H
