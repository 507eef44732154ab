// Test-only: the harness's PNG reader (tools/png_loader.h) behind a C function for ctypes.
#include "../../tools/png_loader.h"
extern "C" int shim_load_png(const uint8_t *data, size_t n, uint8_t *out, size_t cap, uint32_t *w, uint32_t *h, char *err, size_t err_cap)
{
    std::vector<uint8_t> rgba;
    std::string e;
    if (!png_loader::load_rgba(data, n, rgba, *w, *h, e)) {
        snprintf(err, err_cap, "%s", e.c_str());
        return 0;
    }
    if (rgba.size() > cap) return 0;
    memcpy(out, rgba.data(), rgba.size());
    return 1;
}
