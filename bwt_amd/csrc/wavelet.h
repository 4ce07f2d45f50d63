// Device view of the wavelet matrix and its rank functions.
//
// The matrix is 4-ary: every level consumes two bits of the symbol code, so a
// rank query touches ceil(bits/2) rank blocks instead of `bits` -- one block for
// DNA, four for bytes.  Symbol codes are c - min_sym (min_sym = smallest symbol
// of the text besides the terminator), which makes a 4-letter alphabet exactly
// two bits wide.
//
// "Rank block" = one aligned 64-byte sector (what the memory side fetches for a
// random access; a query touches exactly one per level):
//   4 x u32   occurrences of digit 0,1,2,3 on this level before the block
//   3 x 16 B  chunk k = { low-bit plane word k, high-bit plane word k }: the two
//             bits of the digits of positions 64k .. 64k+63 of the block, each
//             plane word being one wavefront ballot
// so a block covers 192 positions (1/3 byte per position and level).  Levels are
// stored one after another, `level_words` 64-bit words apart.
//
// Level l+1 holds the sequence of level l stably partitioned by its digit
// (all 0s, then 1s, 2s, 3s).  rank(c, i) walks i down the levels:
//   q <- start[l][d] + (occurrences of d on level l before q),  d = l-th digit of c
// The walk of the range start (position 0) depends on c only and is tabulated
// (p_end[c]); occ(c, i) = q_end - p_end[c].
//
// One position of L (the rank of the block's first suffix) holds a symbol that
// does not belong to the block (reference: include/bwt.h:234-236 hides it as 0).
// Here it is stored with code 0 and taken out again by the caller: occ(c, i)
// drops by one when code(c) == 0 and the hidden position is below i.
#pragma once

#include "device_utils.h"

namespace bwtamd {

constexpr u32 WM_CHUNKS = 3;                       // 64-position chunks per rank block
constexpr u32 WM_BLOCK_POS = 64 * WM_CHUNKS;       // positions per rank block
constexpr u32 WM_BLOCK_WORDS = 2 + 2 * WM_CHUNKS;  // 64-bit words per rank block (64 bytes)
constexpr int WM_MAX_LEVELS = 4;

// Alphabets of more than four symbols take four code bits per level instead of two (16-ary levels): a rank
// block is then 128 bytes = 16 x u32 digit counts + four bit planes of 128 positions (plane b: one word for
// positions 0..63, one for 64..127).  The memory side serves a random 128-byte line at the same rate as a
// random 64-byte sector, so a byte alphabet costs two requests per LF step instead of four.
constexpr u32 WM16_BLOCK_POS = 128;
constexpr u32 WM16_BLOCK_WORDS = 16;

struct WmView {
  const u64 *planes;
  u64 level_words;
  int levels;
  int dbits;                 // code bits per level: 2 (64-byte blocks) or 4 (128-byte blocks)
  u32 min_sym;
  u32 max_pos;               // last position that has a rank block (>= the block length)
  u32 hidden;                // position of the foreign symbol
  u32 start[WM_MAX_LEVELS][16];  // first position of each digit's bucket on the NEXT level
};

// occurrences of digit d in level[0:pos)
__device__ __forceinline__ u32 wm_rank_digit(const u64 *__restrict__ lvl, u32 pos, u32 d) {
  u32 blk = pos / WM_BLOCK_POS;
  u32 off = pos - blk * WM_BLOCK_POS;
  const uint4 *p = reinterpret_cast<const uint4 *>(lvl + (u64)blk * WM_BLOCK_WORDS);
  u32 wi = off >> 6, bi = off & 63;
  uint4 hdr = p[0];
  uint4 ch[WM_CHUNKS];
#pragma unroll
  for (u32 k = 0; k < WM_CHUNKS; k++) ch[k] = p[1 + k];  // the whole block: one sector, no per-lane branches
  u32 cnt = d == 0 ? hdr.x : (d == 1 ? hdr.y : (d == 2 ? hdr.z : hdr.w));
  const u64 inv0 = (d & 1) ? 0ull : ~0ull, inv1 = (d & 2) ? 0ull : ~0ull;
  const u64 part = (1ull << bi) - 1;
#pragma unroll
  for (u32 k = 0; k < WM_CHUNKS; k++) {
    u64 lo = ((u64)ch[k].y << 32) | ch[k].x, hi = ((u64)ch[k].w << 32) | ch[k].z;
    u64 msk = (k < wi) ? ~0ull : (k == wi ? part : 0ull);
    cnt += __popcll((lo ^ inv0) & (hi ^ inv1) & msk);
  }
  return cnt;
}

// occurrences of the 4-bit digit d in level[0:pos), 16-ary levels
__device__ __forceinline__ u32 wm_rank_digit16(const u64 *__restrict__ lvl, u32 pos, u32 d) {
  const u32 blk = pos / WM16_BLOCK_POS, off = pos % WM16_BLOCK_POS;
  const uint4 *p = reinterpret_cast<const uint4 *>(lvl + (u64)blk * WM16_BLOCK_WORDS);
  const uint4 h = p[d >> 2];  // the four counts that hold d's
  uint4 pl[4];
#pragma unroll
  for (u32 b = 0; b < 4; b++) pl[b] = p[4 + b];
  u32 cnt = (d & 3) == 0 ? h.x : ((d & 3) == 1 ? h.y : ((d & 3) == 2 ? h.z : h.w));
  u64 m0 = ~0ull, m1 = ~0ull;  // positions 0..63 and 64..127 that hold digit d
#pragma unroll
  for (u32 b = 0; b < 4; b++) {
    const u64 inv = ((d >> b) & 1) ? 0ull : ~0ull;
    m0 &= (((u64)pl[b].y << 32) | pl[b].x) ^ inv;
    m1 &= (((u64)pl[b].w << 32) | pl[b].z) ^ inv;
  }
  const u64 k0 = off >= 64 ? ~0ull : ((1ull << off) - 1), k1 = off > 64 ? ((1ull << (off - 64)) - 1) : 0ull;
  return cnt + __popcll(m0 & k0) + __popcll(m1 & k1);
}

// position reached by walking index i down all levels along the digits of code
__device__ __forceinline__ u32 wm_walk(const WmView &wm, u32 code, u32 i) {
  u32 q = i;
  if (wm.dbits == 4) {
    for (int l = 0; l < wm.levels; l++) {
      q = q < wm.max_pos ? q : wm.max_pos;
      const u32 d = (code >> (4 * (wm.levels - 1 - l))) & 15;
      q = wm.start[l][d] + wm_rank_digit16(wm.planes + (u64)l * wm.level_words, q, d);
    }
    return q;
  }
  for (int l = 0; l < wm.levels; l++) {
    q = q < wm.max_pos ? q : wm.max_pos;  // never read past the planes, whatever the caller passes
    u32 d = (code >> (2 * (wm.levels - 1 - l))) & 3;
    q = wm.start[l][d] + wm_rank_digit(wm.planes + (u64)l * wm.level_words, q, d);
  }
  return q;
}

// LF-style count: lf_base[c] + walk, minus the hidden symbol when it was counted.
// lf_base[c] = C[c] - p_end[c]; the result is C[c] + occ(c, i).
__device__ __forceinline__ u32 wm_lf(const WmView &wm, const u32 *__restrict__ lf_base, u32 c, u32 i) {
  u32 code = c - wm.min_sym;
  u32 k = lf_base[c] + wm_walk(wm, code, i);
  return k - (u32)(code == 0 && wm.hidden < i);
}

struct SymNext {
  u32 sym;   // L[i]
  u32 next;  // LF_map_mod(L[i], i), relative to the block's first rank
};

// digit at `pos` of a 4-ary level and its occurrences before pos: one 64-byte block
__device__ __forceinline__ void wm_access_digit(const u64 *__restrict__ lvl, u32 pos, u32 *digit, u32 *before) {
  const u32 blk = pos / WM_BLOCK_POS, off = pos - blk * WM_BLOCK_POS;
  const uint4 *p = reinterpret_cast<const uint4 *>(lvl + (u64)blk * WM_BLOCK_WORDS);
  const u32 wi = off >> 6, bi = off & 63;
  const uint4 hdr = p[0];
  uint4 ch[WM_CHUNKS];
#pragma unroll
  for (u32 k = 0; k < WM_CHUNKS; k++) ch[k] = p[1 + k];
  u64 lo[WM_CHUNKS], hi[WM_CHUNKS];
#pragma unroll
  for (u32 k = 0; k < WM_CHUNKS; k++) {
    lo[k] = ((u64)ch[k].y << 32) | ch[k].x;
    hi[k] = ((u64)ch[k].w << 32) | ch[k].z;
  }
  const u64 mylo = wi == 0 ? lo[0] : (wi == 1 ? lo[1] : lo[2]), myhi = wi == 0 ? hi[0] : (wi == 1 ? hi[1] : hi[2]);
  const u32 d = (u32)((mylo >> bi) & 1) | ((u32)((myhi >> bi) & 1) << 1);
  u32 cnt = d == 0 ? hdr.x : (d == 1 ? hdr.y : (d == 2 ? hdr.z : hdr.w));
  const u64 inv0 = (d & 1) ? 0ull : ~0ull, inv1 = (d & 2) ? 0ull : ~0ull;
  const u64 part = (1ull << bi) - 1;
#pragma unroll
  for (u32 k = 0; k < WM_CHUNKS; k++) {
    const u64 msk = (k < wi) ? ~0ull : (k == wi ? part : 0ull);
    cnt += __popcll((lo[k] ^ inv0) & (hi[k] ^ inv1) & msk);
  }
  *digit = d;
  *before = cnt;
}

// the same for a 16-ary level: one 128-byte block
__device__ __forceinline__ void wm_access_digit16(const u64 *__restrict__ lvl, u32 pos, u32 *digit, u32 *before) {
  const u32 blk = pos / WM16_BLOCK_POS, off = pos % WM16_BLOCK_POS;
  const uint4 *p = reinterpret_cast<const uint4 *>(lvl + (u64)blk * WM16_BLOCK_WORDS);
  uint4 pl[4];
#pragma unroll
  for (u32 b = 0; b < 4; b++) pl[b] = p[4 + b];
  u32 d = 0;
#pragma unroll
  for (u32 b = 0; b < 4; b++) {
    const u64 word = off < 64 ? (((u64)pl[b].y << 32) | pl[b].x) : (((u64)pl[b].w << 32) | pl[b].z);
    d |= (u32)((word >> (off & 63)) & 1) << b;
  }
  const uint4 h = p[d >> 2];
  u32 cnt = (d & 3) == 0 ? h.x : ((d & 3) == 1 ? h.y : ((d & 3) == 2 ? h.z : h.w));
  u64 m0 = ~0ull, m1 = ~0ull;
#pragma unroll
  for (u32 b = 0; b < 4; b++) {
    const u64 inv = ((d >> b) & 1) ? 0ull : ~0ull;
    m0 &= (((u64)pl[b].y << 32) | pl[b].x) ^ inv;
    m1 &= (((u64)pl[b].w << 32) | pl[b].z) ^ inv;
  }
  const u64 k0 = off >= 64 ? ~0ull : ((1ull << off) - 1), k1 = off > 64 ? ((1ull << (off - 64)) - 1) : 0ull;
  *digit = d;
  *before = cnt + __popcll(m0 & k0) + __popcll(m1 & k1);
}

// L[i] and LF_map_mod(L[i], i) (include/bwt.h:253-259) for a rank i other than the hidden one
__device__ __forceinline__ SymNext wm_access_lf(const WmView &wm, const u32 *__restrict__ lf_base, u32 i, u32 e_char,
                                                u32 e_rank) {
  u32 q = i, code = 0;
  for (int l = 0; l < wm.levels; l++) {
    q = q < wm.max_pos ? q : wm.max_pos;
    u32 d, before;
    if (wm.dbits == 4) wm_access_digit16(wm.planes + (u64)l * wm.level_words, q, &d, &before);
    else wm_access_digit(wm.planes + (u64)l * wm.level_words, q, &d, &before);
    code = (code << wm.dbits) | d;
    q = wm.start[l][d] + before;
  }
  const u32 c = code + wm.min_sym;
  u32 k = lf_base[c] + q - (u32)(code == 0 && wm.hidden < i);
  k += (u32)(k >= e_rank && c == e_char);
  return SymNext{c, k};
}

}  // namespace bwtamd
