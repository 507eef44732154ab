// format.h -- the fpng bitstream constants and the device-visible table layout.
//
// The only literal data here are the two trained 1-pass Deflate block prefixes, which ARE the
// file format (reference src/fpng.cpp:532-535, :548-551).  Everything else (per-symbol codes,
// RLE-chunk tokens, CRC constants) is derived from them or from RFC 1950/1951 at init time.
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace fpng_amd {

constexpr uint32_t kAdlerMod = 65521u;
constexpr uint32_t kPngHeaderBytes = 58;      // sig + IHDR + fdEC + IDAT len/type (reference src/fpng.cpp:1701)
constexpr uint32_t kPngTrailerBytes = 16;     // IDAT crc + IEND (reference src/fpng.cpp:1794)
constexpr uint32_t kMaxChunkPixels3 = 85;     // 255 / 3  (reference src/fpng.cpp:1212)
constexpr uint32_t kMaxChunkPixels4 = 63;     // 252 / 4  (reference src/fpng.cpp:1506)
constexpr uint32_t kStoredBlockMax = 65535;   // reference src/fpng.cpp:834

// Token table as the kernels consume it (lives in device global memory, staged into LDS).
//   lit[s]   : Huffman code of literal/length symbol s, bit-reversed for LSB-first emission, in bits
//              0..15; code length in bits 16..20.
//   chunk[q] : complete token of an RLE chunk of q pixels (match of q*c bytes at distance c):
//              length code, extra bits and the 1-bit distance code (always 0) in bits 0..23, total
//              bit count in bits 24..31.  (reference src/fpng.cpp:1221-1226, :1530-1531)
struct TokenTable {
    uint32_t lit[288];
    uint32_t chunk[96];
    uint32_t first_token_bit; // zlib-stream bit position where row tokens start (490 / 503 for 1-pass)
    uint32_t header_bits;     // == first_token_bit; header[] holds ceil(header_bits/8) bytes
    uint32_t pad[2];
    uint8_t header[400];      // 78 01 + BFINAL/BTYPE + code-length header, LSB-first, zero padded
};

inline uint32_t lit_code(uint32_t e) { return e & 0xFFFFu; }
inline uint32_t lit_len(uint32_t e) { return e >> 16; }

// Host-side construction ----------------------------------------------------------------------
// Derives the 1-pass tables for 3 and 4 channels by parsing the trained prefixes like an inflater
// would.  Returns false if a prefix does not parse to the expected layout (never, unless edited).
bool build_1pass_tables(TokenTable *t3, TokenTable *t4);

// RFC 1951 length-symbol mapping for match_len-3 in [0,255]
void deflate_length_symbol(uint32_t adj_len, uint32_t *sym, uint32_t *extra_bits);

// checksums on host memory (fpng_crc32 / fpng_adler32 calling convention)
uint32_t host_crc32(const void *data, size_t size, uint32_t prev);
uint32_t host_adler32(const void *data, size_t size, uint32_t prev);
// GF(2) helpers, reflected CRC-32 domain (bit 31 = x^0)
uint32_t gf2_mulmod(uint32_t a, uint32_t b);
uint32_t gf2_xpow8n(uint64_t nbytes); // x^(8*nbytes) mod P
uint32_t crc32_combine(uint32_t crc_x, uint32_t crc_y, uint64_t len_y);
uint32_t adler32_combine(uint32_t adler_x, uint32_t adler_y, uint64_t len_y);

uint32_t gf2_xpow(uint64_t e); // x^e mod P (any e; x has order dividing 2^32-1)

} // namespace fpng_amd
