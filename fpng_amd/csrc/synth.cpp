// synth.cpp -- deterministic synthetic images for parity tests and bench.py (host code).
//
// Definitions follow SURVEY.md Appendix B.1 exactly, so that the known answers of B.2 (sizes and
// sha256 of the reference encoder's output) can be reproduced.  xorshift32 is advanced once per
// pixel in raster order for EVERY kind.
#include "fpng_amd.h"

extern "C" int fpng_amd_synth_image(int kind, uint32_t seed, uint32_t w, uint32_t h, uint32_t num_chans, uint8_t *out)
{
    if (!out || (num_chans != 3 && num_chans != 4) || !w || !h) return FPNG_AMD_ERR_INVALID_ARG;
    uint32_t s = seed;
    for (uint32_t y = 0; y < h; y++) {
        for (uint32_t x = 0; x < w; x++) {
            s ^= s << 13;
            s ^= s >> 17;
            s ^= s << 5;
            const uint32_t r = s;
            uint8_t p[4];
            switch (kind) {
            case FPNG_AMD_SYNTH_NOISE:
                p[0] = (uint8_t)r, p[1] = (uint8_t)(r >> 8), p[2] = (uint8_t)(r >> 16), p[3] = (uint8_t)(r >> 24);
                break;
            case FPNG_AMD_SYNTH_SOLID:
                p[0] = 0x40, p[1] = 0x80, p[2] = 0xC0, p[3] = 0xFF;
                break;
            case FPNG_AMD_SYNTH_GRAD:
                p[0] = (uint8_t)(x * 255u / w + (r & 3u));
                p[1] = (uint8_t)(y * 255u / h + ((r >> 2) & 3u));
                p[2] = (uint8_t)((x + y) * 255u / (w + h) + ((r >> 4) & 3u));
                p[3] = 0xFF;
                break;
            case FPNG_AMD_SYNTH_BLOCKS: {
                uint32_t t = ((x >> 6) * 73856093u) ^ ((y >> 6) * 19349663u);
                t ^= t >> 13;
                t *= 0x5bd1e995u;
                t ^= t >> 15;
                p[0] = (uint8_t)t, p[1] = (uint8_t)(t >> 8), p[2] = (uint8_t)(t >> 16), p[3] = (uint8_t)((t >> 24) | 0x80u);
                break;
            }
            default:
                return FPNG_AMD_ERR_INVALID_ARG;
            }
            for (uint32_t i = 0; i < num_chans; i++) *out++ = p[i];
        }
    }
    return FPNG_AMD_OK;
}
