// Self-test and timing of the host side's two-bit codec (bwt_amd/csrc/pack2.h) against its byte-by-byte definition.
// Usage: pack2_selftest [symbols to time, default 0 = test only]
#include <stdio.h>
#include <stdlib.h>
#include <chrono>
#include <vector>

#include "../bwt_amd/csrc/pack2.h"

using namespace bwtamd;

static uint64_t rng_state = 88172645463325252ull;
static uint64_t rng() {
  rng_state ^= rng_state << 13;
  rng_state ^= rng_state >> 7;
  rng_state ^= rng_state << 17;
  return rng_state;
}

static int fails = 0;
#define EXPECT(c)                                         \
  do {                                                    \
    if (!(c)) {                                           \
      printf("FAIL line %d: %s\n", __LINE__, #c);         \
      fails++;                                            \
    }                                                     \
  } while (0)

int main(int argc, char **argv) {
  const int64_t lens[] = {0, 1, 2, 3, 4, 5, 7, 8, 9, 15, 16, 17, 31, 33, 64, 1000, 4099};
  for (int64_t len : lens)
    for (uint32_t min_sym : {1u, 2u, 65u, 97u, 250u, 252u})
      for (uint32_t span = 0; span <= 3; span++) {
        std::vector<uint8_t> src((size_t)len + 8), pk((size_t)len / 4 + 8, 0xAB), back((size_t)len + 8, 0xCD);
        for (int64_t i = 0; i < len; i++) src[(size_t)i] = (uint8_t)(min_sym + rng() % (span + 1));
        EXPECT(pack2::pack(src.data(), len, pk.data(), min_sym, span));
        for (int64_t i = 0; i < len; i++) EXPECT(((pk[(size_t)(i >> 2)] >> (2 * (i & 3))) & 3u) == (uint32_t)(src[(size_t)i] - min_sym));
        EXPECT(pk[(size_t)((len + 3) / 4)] == 0xAB);
        pack2::unpack(pk.data(), len, back.data(), min_sym);
        for (int64_t i = 0; i < len; i++) EXPECT(back[(size_t)i] == src[(size_t)i]);
        EXPECT(back[(size_t)len] == 0xCD);
        // one symbol outside the range, at every position in turn (short texts) or at a few (long ones)
        for (int64_t at = 0; at < len; at += (len > 64 ? 97 : 1))
          for (int v : {0, (int)min_sym - 1, (int)(min_sym + span + 1), 255, (int)min_sym + 4, (int)min_sym + 128}) {
            if (v < 0 || v > 255 || (v >= (int)min_sym && v <= (int)(min_sym + span))) continue;
            const uint8_t keep = src[(size_t)at];
            src[(size_t)at] = (uint8_t)v;
            EXPECT(!pack2::pack(src.data(), len, pk.data(), min_sym, span));
            src[(size_t)at] = keep;
          }
      }
  // every code wraps modulo 256 on the way back (the code that stands in for the terminator may exceed 255)
  {
    uint8_t pk[4] = {0xE4, 0x1B, 0xFF, 0x00}, out[16];
    pack2::unpack(pk, 16, out, 254);
    for (int i = 0; i < 16; i++) EXPECT(out[i] == (uint8_t)(254 + ((pk[i >> 2] >> (2 * (i & 3))) & 3)));
  }
  const int64_t n = argc > 1 ? atoll(argv[1]) : 0;
  if (n > 0) {
    std::vector<uint8_t> src((size_t)n), pk((size_t)n / 4 + 8), back((size_t)n);
    for (int64_t i = 0; i < n; i++) src[(size_t)i] = (uint8_t)(1 + (rng() & 3));
    for (int rep = 0; rep < 3; rep++) {
      auto t0 = std::chrono::steady_clock::now();
      const bool ok = pack2::pack(src.data(), n, pk.data(), 1, 3);
      auto t1 = std::chrono::steady_clock::now();
      pack2::unpack(pk.data(), n, back.data(), 1);
      auto t2 = std::chrono::steady_clock::now();
      printf("pack %.2f Gsym/s  unpack %.2f Gsym/s  (one thread, %ld symbols, ok=%d)\n",
             n / std::chrono::duration<double>(t1 - t0).count() / 1e9, n / std::chrono::duration<double>(t2 - t1).count() / 1e9, (long)n, (int)ok);
    }
    EXPECT(back == src);
  }
  printf(fails ? "FAILED\n" : "OK\n");
  return fails != 0;
}
