// sync_stats.c -- how fast does a Huffman decoder started at a wrong bit of an fpng stream fall into step with the true token
// sequence?  (The numbers behind the decoder's design: DESIGN 4.4.)  For every subsequence boundary b = first + k * 512 it starts a
// decode V bits in front of b and reports whether the first token boundary at or behind b is the true one, for several V; and the
// distribution of tokens / literal groups per subsequence.  Uses only the host side of the library (fpng_amd_decode_plan): no GPU.
//   gcc -O2 -o /tmp/sync_stats tools/sync_stats.c -Iinclude -Lfpng_amd/lib -lfpng_amd -Wl,-rpath,$PWD/fpng_amd/lib
#include "fpng_amd.h"
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static const uint8_t *z;
static uint32_t lut[4096];
static uint64_t limit_bit;

static inline uint32_t window(uint64_t pos)
{
    uint64_t v;
    memcpy(&v, z + (pos >> 3), 8);
    return (uint32_t)(v >> (pos & 7));
}
// returns 0 ok, 1 eob, 2 invalid; advances *pos
static inline int token(uint64_t *pos, uint32_t *bytes)
{
    const uint32_t w = window(*pos), e = lut[w & 4095], len = (e >> 9) & 15, sym = e & 511;
    if (!len) return 2;
    if (sym < 256) {
        *pos += len, *bytes = 1;
        return 0;
    }
    if (sym == 256) return 1;
    const uint32_t xb = (e >> 13) & 7;
    *bytes = (e >> 16) + ((w >> len) & ((1u << xb) - 1));
    *pos += len + xb + 1;
    return 0;
}

int main(int argc, char **argv)
{
    for (int a = 1; a < argc; a++) {
        FILE *f = fopen(argv[a], "rb");
        if (!f) return 1;
        fseek(f, 0, SEEK_END);
        long n = ftell(f);
        fseek(f, 0, SEEK_SET);
        uint8_t *png = malloc(n + 64);
        memset(png + n, 0, 64);
        if (fread(png, 1, n, f) != (size_t)n) return 1;
        fclose(f);
        fpng_amd_decode_result res;
        uint32_t mode, ofs, len;
        uint64_t first, limit;
        if (fpng_amd_decode_plan(png, (uint32_t)n, &res, &mode, &ofs, &len, &first, &limit, lut) || res.status || mode) {
            printf("%s: not a dynamic fpng file\n", argv[a]);
            continue;
        }
        z = png + ofs + 8;
        limit_bit = limit;
        // true token boundaries
        uint8_t *is_start = calloc((limit >> 3) + 64, 1);
        uint64_t pos = first, ntok = 0, nlit = 0, end = 0;
        for (;;) {
            is_start[pos >> 3] |= 1u << (pos & 7);
            uint32_t b;
            const int r = token(&pos, &b);
            if (r) {
                end = pos;
                break;
            }
            ntok++, nlit += b == 1;
        }
        const uint64_t nsub = (end - first) / 512;
        printf("%s: %ux%ux%u, %llu token bits, %llu tokens (%.2f bits each), %.1f %% literals, %.1f tokens per 512 bits\n", argv[a], res.w, res.h, res.channels_in_file,
               (unsigned long long)(end - first), (unsigned long long)ntok, (double)(end - first) / ntok, 100.0 * nlit / ntok, 512.0 * ntok / (end - first));
        // code length histogram of the table
        int hist[16] = {0};
        for (int k = 0; k < 4096; k++) hist[(lut[k] >> 9) & 15]++;
        printf("  table slots by code length:");
        for (int k = 0; k < 13; k++) printf(" %d:%d", k, hist[k]);
        printf("\n");
        static const int Vs[] = {0, 32, 64, 96, 128, 192, 256, 384, 512};
        for (unsigned vi = 0; vi < sizeof Vs / sizeof *Vs; vi++) {
            const int V = Vs[vi];
            uint64_t bad = 0, longest = 0, cur = 0;
            for (uint64_t k = 1; k < nsub; k++) {
                const uint64_t b = first + k * 512;
                uint64_t p = b - V;
                if (p < first) p = first;
                int r = 0;
                while (p < b && !r) {
                    uint32_t by;
                    r = token(&p, &by);
                }
                const int ok = !r && ((is_start[p >> 3] >> (p & 7)) & 1);
                if (!ok) bad++, cur++; else cur = 0;
                if (cur > longest) longest = cur;
            }
            printf("  lead-in %3d bits: %8llu of %llu boundaries out of step (%.4f %%), longest streak %llu\n", V, (unsigned long long)bad, (unsigned long long)nsub,
                   100.0 * bad / (nsub ? nsub : 1), (unsigned long long)longest);
        }
        // lookups per subsequence with a multi-literal table: up to 3 literals whose codes fit 12 bits together per lookup
        uint64_t lookups = 0;
        pos = first;
        while (pos < end) {
            uint32_t b;
            uint64_t p = pos;
            int r = token(&p, &b);
            if (r) break;
            if (b == 1 && (lut[window(pos) & 4095] & 511) < 256) {
                uint32_t used = (uint32_t)(p - pos);
                int cnt = 1;
                while (cnt < 3) {
                    uint64_t q = p;
                    uint32_t b2;
                    const uint32_t e = lut[window(p) & 4095];
                    if ((e & 511) >= 256 || !((e >> 9) & 15)) break;
                    if (used + ((e >> 9) & 15) > 12) break;
                    token(&q, &b2);
                    used += (uint32_t)(q - p), p = q, cnt++;
                }
            }
            pos = p, lookups++;
        }
        printf("  multi-literal table (<= 3 literals in 12 bits): %llu lookups = %.2f tokens each, %.1f lookups per 512 bits\n", (unsigned long long)lookups,
               (double)ntok / lookups, 512.0 * lookups / (end - first));
        free(is_start);
        free(png);
    }
    return 0;
}
