// The two-bit codec of pack2.h for the CPU at hand: 32 symbols a step with AVX2 where the CPU has it (checked once,
// when the library is loaded), the 64-bit arithmetic of pack2.h otherwise and for the ends.
#include "pack2.h"

#include <immintrin.h>

namespace bwtamd {
namespace pack2 {

// 32 symbols -> 8 bytes per step.  d = symbol - min_sym in every byte; pairs of bytes d0 + 4 d1 (16-bit lanes), pairs
// of those + 16 times the next (32-bit lanes): the packed byte is the low byte of every 32-bit lane.  A byte beyond
// span shows in the running maximum.
__attribute__((target("avx2"))) static bool pack_avx2(const uint8_t *src, int64_t count, uint8_t *dst, uint32_t min_sym, uint32_t span) {
  const __m256i m = _mm256_set1_epi8((char)min_sym);
  const __m256i mul1 = _mm256_set1_epi16(0x0401);      // bytes (1, 4): d0 + 4 d1
  const __m256i mul2 = _mm256_set1_epi32(0x00100001);  // words (1, 16)
  const __m256i pick = _mm256_setr_epi8(0, 4, 8, 12, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, 0, 4, 8, 12, -1, -1, -1, -1, -1, -1, -1,
                                        -1, -1, -1, -1, -1);
  __m256i worst = _mm256_setzero_si256();
  int64_t i = 0;
  for (; i + 32 <= count; i += 32) {
    const __m256i x = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(src + i));
    const __m256i d = _mm256_sub_epi8(x, m);
    worst = _mm256_max_epu8(worst, d);
    const __m256i d3 = _mm256_and_si256(d, _mm256_set1_epi8(3));
    const __m256i w = _mm256_maddubs_epi16(d3, mul1);
    const __m256i q = _mm256_madd_epi16(w, mul2);
    const __m256i b = _mm256_shuffle_epi8(q, pick);  // four packed bytes at the bottom of either half
    const uint32_t lo = (uint32_t)_mm256_cvtsi256_si32(b), hi = (uint32_t)_mm256_extract_epi32(b, 4);
    memcpy(dst + (i >> 2), &lo, 4);
    memcpy(dst + (i >> 2) + 4, &hi, 4);
  }
  alignas(32) uint8_t wb[32];
  _mm256_store_si256(reinterpret_cast<__m256i *>(wb), worst);
  uint32_t mx = 0;
  for (int k = 0; k < 32; k++) mx = wb[k] > mx ? wb[k] : mx;
  bool ok = mx <= span;
  if (i < count) ok = pack_body(src + i, count - i, dst + (i >> 2), min_sym, span) && ok;
  return ok;
}

// 8 bytes -> 32 symbols per step: every packed byte four times over (byte shuffle), shifted by 0, 2, 4 and 6 bits in
// 16-bit lanes (what a shift drags in from the neighbouring byte lands above the two bits kept), the shift that belongs
// to a position picked by masks.  Whole lines leave with non-temporal stores when dst allows it: the caller's array is
// not read again here, and a line written this way is not fetched first.
__attribute__((target("avx2"))) static void unpack_avx2(const uint8_t *src, int64_t count, uint8_t *dst, uint32_t min_sym) {
  const __m256i rep = _mm256_setr_epi8(0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 6, 6, 6, 6, 7, 7, 7, 7);
  const __m256i three = _mm256_set1_epi8(3), m = _mm256_set1_epi8((char)min_sym);
  const __m256i p0 = _mm256_set1_epi32(0x000000FF), p1 = _mm256_set1_epi32(0x0000FF00), p2 = _mm256_set1_epi32(0x00FF0000),
                p3 = _mm256_set1_epi32((int)0xFF000000u);
  int64_t i = 0;
  // (up to the first 32-byte boundary of dst with plain code, so that the stores below are aligned)
  const int64_t lead = (int64_t)((32 - (reinterpret_cast<uintptr_t>(dst) & 31)) & 31);
  const bool stream = lead % 4 == 0 && count >= 4096;
  if (stream && lead) {
    unpack_body(src, lead, dst, min_sym);
    i = lead;
  }
  for (; i + 32 <= count; i += 32) {
    uint64_t w;
    memcpy(&w, src + (i >> 2), 8);
    const __m256i b = _mm256_shuffle_epi8(_mm256_set1_epi64x((long long)w), rep);
    const __m256i s0 = _mm256_and_si256(b, p0);
    const __m256i s1 = _mm256_and_si256(_mm256_srli_epi16(b, 2), p1);
    const __m256i s2 = _mm256_and_si256(_mm256_srli_epi16(b, 4), p2);
    const __m256i s3 = _mm256_and_si256(_mm256_srli_epi16(b, 6), p3);
    __m256i x = _mm256_or_si256(_mm256_or_si256(s0, s1), _mm256_or_si256(s2, s3));
    x = _mm256_add_epi8(_mm256_and_si256(x, three), m);
    if (stream) _mm256_stream_si256(reinterpret_cast<__m256i *>(dst + i), x);
    else _mm256_storeu_si256(reinterpret_cast<__m256i *>(dst + i), x);
  }
  if (stream) _mm_sfence();
  if (i < count) unpack_body(src + (i >> 2), count - i, dst + i, min_sym);
}

static const bool have_avx2 = __builtin_cpu_supports("avx2");

bool pack(const uint8_t *src, int64_t count, uint8_t *dst, uint32_t min_sym, uint32_t span) {
  return have_avx2 ? pack_avx2(src, count, dst, min_sym, span) : pack_body(src, count, dst, min_sym, span);
}

void unpack(const uint8_t *src, int64_t count, uint8_t *dst, uint32_t min_sym) {
  if (have_avx2) unpack_avx2(src, count, dst, min_sym);
  else unpack_body(src, count, dst, min_sym);
}

}  // namespace pack2
}  // namespace bwtamd
