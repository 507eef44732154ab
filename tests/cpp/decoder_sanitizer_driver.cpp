// decoder_sanitizer_driver.cpp -- TEST INFRASTRUCTURE: fpng::fpng_decode_memory (the drop-in CPU decoder, fpng_amd/csrc/fpng_decode.cpp) over a
// corpus file of length-prefixed PNGs, each decoded from an EXACT-SIZE copy to 3 and to 4 channels; built with -fsanitize=address,undefined by
// tests/test_dropin_decode.py: reads behind a file, writes behind a row buffer and undefined shifts are the sanitizers to see.
#include "fpng.h"
#include <cstdio>
#include <vector>
#include <cstring>
int main(int argc, char **argv)
{
    FILE *f = fopen(argv[1], "rb");
    std::vector<uint8_t> all;
    uint8_t buf[1 << 16];
    size_t k;
    while ((k = fread(buf, 1, sizeof buf, f)) > 0) all.insert(all.end(), buf, buf + k);
    fclose(f);
    size_t o = 0, n = 0, ok = 0;
    std::vector<uint8_t> out;
    while (o + 4 <= all.size()) {
        uint32_t len;
        memcpy(&len, &all[o], 4);
        o += 4;
        // an exact-size copy: reads behind the file are the sanitizer's to see
        std::vector<uint8_t> one(all.begin() + o, all.begin() + o + len);
        o += len;
        for (uint32_t d = 3; d <= 4; d++) {
            uint32_t w, h, c;
            ok += fpng::fpng_decode_memory(one.data(), (uint32_t)one.size(), out, w, h, c, d) == 0;
            n++;
        }
    }
    printf("%zu decodes, %zu succeeded\n", n, ok);
}
