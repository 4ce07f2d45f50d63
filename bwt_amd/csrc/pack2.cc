// The two-bit codec of pack2.h as two clones each: one for CPUs with AVX2, picked when the library is loaded, and
// the baseline x86-64 one.
#include "pack2.h"

namespace bwtamd {
namespace pack2 {

__attribute__((target_clones("avx2", "default"))) bool pack(const uint8_t *src, int64_t count, uint8_t *dst, uint32_t min_sym, uint32_t span) {
  return pack_body(src, count, dst, min_sym, span);
}

__attribute__((target_clones("avx2", "default"))) void unpack(const uint8_t *src, int64_t count, uint8_t *dst, uint32_t min_sym) {
  unpack_body(src, count, dst, min_sym);
}

}  // namespace pack2
}  // namespace bwtamd
