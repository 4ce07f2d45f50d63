// Two-bit codec of the host side of a packed transfer: symbols min_sym .. min_sym + 3 as two bits each, four to a
// byte, least significant first.  Plain 64-bit arithmetic, eight symbols a step, written so that the compiler turns
// it into vector code (pack2.cc builds an AVX2 clone beside the baseline one); no GPU code here.
#pragma once

#include <stdint.h>
#include <string.h>

namespace bwtamd {
namespace pack2 {

constexpr uint64_t ONES = 0x0101010101010101ull;
constexpr uint64_t HIGH = 0x8080808080808080ull;

// x - m in every byte, modulo 256, no borrow between bytes
static inline uint64_t sub_bytes(uint64_t x, uint64_t m) { return ((x | HIGH) - (m & ~HIGH)) ^ ((x ^ ~m) & HIGH); }

// the low two bits of the four bytes of v, gathered into one byte (byte k -> bits 2k, 2k+1).  The product moves
// byte k to bits 24 + 2k; every other partial product lands below bit 24 or beyond bit 31, and no two overlap.
static inline uint32_t gather4(uint32_t v) { return ((v & 0x03030303u) * 0x01041040u) >> 24; }

// Packs src[0:count) into dst[0:(count+3)/4); returns false if a symbol lies outside [min_sym, min_sym + span]
// (span <= 3).  dst is written either way.
static inline bool pack_body(const uint8_t *src, int64_t count, uint8_t *dst, uint32_t min_sym, uint32_t span) {
  const uint64_t M = ONES * min_sym, slack = ONES * (3 - span);
  uint64_t bad = 0;
  int64_t i = 0;
  for (; i + 8 <= count; i += 8) {
    uint64_t x;
    memcpy(&x, src + i, 8);
    const uint64_t d = sub_bytes(x, M);
    bad |= d | (d + slack);  // a byte beyond span shows in bits 2..7 of d or of d + (3 - span)
    const uint16_t w = (uint16_t)(gather4((uint32_t)d) | (gather4((uint32_t)(d >> 32)) << 8));
    memcpy(dst + (i >> 2), &w, 2);
  }
  uint32_t tail_bad = 0;
  for (; i < count; i += 4) {
    uint32_t b = 0;
    for (int64_t j = i; j < i + 4 && j < count; j++) {
      const uint32_t c = (uint32_t)src[j] - min_sym;  // wraps far beyond span for a symbol below min_sym
      tail_bad |= c > span;
      b |= (c & 3u) << (2 * (j - i));
    }
    dst[i >> 2] = (uint8_t)b;
  }
  return (bad & (ONES * 0xFC)) == 0 && tail_bad == 0;
}

// Expands count symbols from src into dst: dst[i] = (min_sym + code i) mod 256
static inline void unpack_body(const uint8_t *src, int64_t count, uint8_t *dst, uint32_t min_sym) {
  const uint64_t M = ONES * min_sym;
  int64_t i = 0;
  for (; i + 8 <= count; i += 8) {
    uint16_t w;
    memcpy(&w, src + (i >> 2), 2);
    uint64_t x = w;
    x = (x | (x << 24)) & 0x000000FF000000FFull;
    x = (x | (x << 12)) & 0x000F000F000F000Full;
    x = (x | (x << 6)) & 0x0303030303030303ull;
    x = (x + (M & ~HIGH)) ^ (M & HIGH);  // byte-wise sum: 3 + 127 carries nowhere, the top bit is added by xor
    memcpy(dst + i, &x, 8);
  }
  for (; i < count; i++) dst[i] = (uint8_t)(min_sym + ((src[i >> 2] >> (2 * (i & 3))) & 3u));
}

// the two above, compiled for the CPU at hand (pack2.cc)
bool pack(const uint8_t *src, int64_t count, uint8_t *dst, uint32_t min_sym, uint32_t span);
void unpack(const uint8_t *src, int64_t count, uint8_t *dst, uint32_t min_sym);

}  // namespace pack2
}  // namespace bwtamd
