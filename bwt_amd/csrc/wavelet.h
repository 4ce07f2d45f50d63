// Device view of the wavelet matrix and its rank functions.
//
// Layout ("rank block" = one 64-byte line = one HBM transaction per rank):
//   word 0      number of 1 bits of this level before the block (low 32 bits)
//   words 1..7  448 bits of the level's bit plane, bit i of the plane at
//               bit (i % 64) of word 1 + (i % 448) / 64 of block i / 448
// Each 64-bit plane word is one wavefront ballot of 64 consecutive symbols.
// Levels are stored one after another, `level_words` words apart.
//
// rank(c, i) needs one rank block per level: the walk of the range start
// (position 0) through the levels depends on c only and is tabulated
// (p_end[c]), so only the walk of i touches memory.
#pragma once

#include "device_utils.h"

namespace bwtamd {

constexpr u32 WM_BLOCK_BITS = 448;
constexpr u32 WM_BLOCK_WORDS = 8;

struct WmView {
  const u64 *planes;
  u64 level_words;
  int levels;
  u32 max_pos;  // last position that has a rank block (>= the block length)
  u32 n_zeros[8];
};

// number of 1 bits in plane[0:pos) of one level
__device__ __forceinline__ u32 wm_rank1(const u64 *__restrict__ lvl, u32 pos) {
  u32 blk = pos / WM_BLOCK_BITS;
  u32 off = pos - blk * WM_BLOCK_BITS;
  const ulonglong2 *p = reinterpret_cast<const ulonglong2 *>(lvl + (u64)blk * WM_BLOCK_WORDS);
  ulonglong2 v0 = p[0], v1 = p[1], v2 = p[2], v3 = p[3];
  u64 wd[8] = {v0.x, v0.y, v1.x, v1.y, v2.x, v2.y, v3.x, v3.y};
  u32 cnt = (u32)wd[0];
  u32 wi = off >> 6, bi = off & 63;
  u64 part = (1ull << bi) - 1;
#pragma unroll
  for (u32 k = 0; k < 7; k++) {
    u64 msk = (k < wi) ? ~0ull : (k == wi ? part : 0ull);
    cnt += __popcll(wd[1 + k] & msk);
  }
  return cnt;
}

// position reached by walking index i down all levels along the bits of c
__device__ __forceinline__ u32 wm_walk(const WmView &wm, u32 c, u32 i) {
  u32 q = i;
  for (int l = 0; l < wm.levels; l++) {
    q = q < wm.max_pos ? q : wm.max_pos;  // never read past the planes, whatever the caller passes
    u32 r1 = wm_rank1(wm.planes + (u64)l * wm.level_words, q);
    q = ((c >> (wm.levels - 1 - l)) & 1) ? wm.n_zeros[l] + r1 : q - r1;
  }
  return q;
}

}  // namespace bwtamd
