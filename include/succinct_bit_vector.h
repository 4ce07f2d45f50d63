/* succinct_bit_vector.h -- host-side rank directory over a slice of a bit array,
   with the public face of the reference's structure
   (include/succinct_bit_vector.h:25-300): init over the bits
   [begin, end) of a byte array (bit k of the array is bit k%8 of byte k/8,
   least significant first, :185-190), rank1 / rank0 of a prefix of the slice,
   n_zeros, rank_slow.

   Nothing on the MI355X path uses it: there the rank directory is part of the
   wavelet matrix's rank blocks on the device (bwt_amd/csrc/wavelet.h).  It is
   here for host code written against the reference.

   Layout: a cumulative count per superblock of 2^15 bits, a count relative to
   its superblock per block of 64 bits, and a popcount of at most 63 bits of the
   array for the rest, as in the reference (:264-283); the slice may start at any
   bit, so the 64 bits of a block are gathered from up to nine bytes. */
#pragma once

#include <stdint.h>
#include <string.h>

#include "typedefs.h"
#include "util.h"

namespace bwt {

  struct succinct_bit_vector {
    uint8_t * bits;
    idx_t begin; /* the slice is bits [begin, end) of the array */
    idx_t end;
    idx_t n_superblocks;
    idx_t n_blocks;
    idx_t * superblocks;  /* ones before the superblock */
    uint16_t * blocks;    /* ones between the superblock's start and the block */
    idx_t n_zeros;        /* zeros in the whole slice */

    succinct_bit_vector() : bits(0), begin(0), end(0), n_superblocks(0), n_blocks(0), superblocks(0), blocks(0), n_zeros(0) {}

    /* up to 64 bits of the slice from slice position p on, bit 0 = position p; positions at or beyond `end` read as 0 */
    uint64_t window(idx_t p, idx_t len) const {
      if (len <= 0) return 0;
      const idx_t first = begin + p;
      const idx_t byte0 = first >> 3, shift = first & 7;
      const idx_t last_byte = (begin + p + len - 1) >> 3;
      uint64_t lo = 0;
      unsigned char tmp[9] = { 0, 0, 0, 0, 0, 0, 0, 0, 0 };
      memcpy(tmp, bits + byte0, (size_t)(last_byte - byte0 + 1));
      memcpy(&lo, tmp, 8);
      uint64_t v = lo >> shift;
      if (shift) v |= (uint64_t)tmp[8] << (64 - shift);
      return len >= 64 ? v : (v & (((uint64_t)1 << len) - 1));
    }

    idx_t is_bit_set(idx_t idx) const {
      const idx_t k = idx + begin;
      return (bits[k >> 3] >> (k & 7)) & 1;
    }

    /* ones among the slice positions [b, e) */
    idx_t count_ones_in_range(idx_t b, idx_t e) const {
      idx_t s = 0;
      while (b < e) {
        const idx_t len = e - b < 64 ? e - b : 64;
        s += __builtin_popcountll(window(b, len));
        b += len;
      }
      return s;
    }

    void init(uint8_t * bits_, idx_t begin_, idx_t end_, mallocator & mem, idx_t prefix_sum_gran = 1000) {
      (void)prefix_sum_gran;
      assert(block_bits == 64 && superblock_bits % block_bits == 0);
      bits = bits_;
      begin = begin_;
      end = end_;
      const idx_t n = end - begin;
      n_superblocks = n / superblock_bits + 1;
      n_blocks = n / block_bits + 1;
      superblocks = mem.new_<idx_t>(n_superblocks, mem_reason_succinct_bitvec_superblock);
      blocks = mem.new_<uint16_t>(n_blocks, mem_reason_succinct_bitvec_block);
      const idx_t per = superblock_bits / block_bits;
      idx_t total = 0;
      for (idx_t s = 0; s < n_superblocks; s++) {
        superblocks[s] = total;
        idx_t inside = 0;
        for (idx_t k = s * per; k < (s + 1) * per && k < n_blocks; k++) {
          blocks[k] = (uint16_t)inside;
          const idx_t p = k * block_bits;
          inside += __builtin_popcountll(window(p, n - p < 64 ? n - p : 64));
        }
        total += inside;
      }
      n_zeros = n - total;
    }

    void fini(mallocator & mem) {
      if (superblocks) mem.delete_(superblocks, n_superblocks, mem_reason_succinct_bitvec_superblock);
      if (blocks) mem.delete_(blocks, n_blocks, mem_reason_succinct_bitvec_block);
      superblocks = 0;
      blocks = 0;
    }

    /* ones among the first n positions of the slice */
    idx_t rank1(idx_t n) {
      stat.start(ts_event_sbv_rank1);
      assert(0 <= n && n <= end - begin);
      const idx_t k = n / block_bits;
      const idx_t r = superblocks[n / superblock_bits] + blocks[k] + __builtin_popcountll(window(k * block_bits, n - k * block_bits));
      stat.end(ts_event_sbv_rank1);
      return r;
    }

    idx_t rank0(idx_t n) { return n - rank1(n); }

    /* bit by bit, for tests */
    idx_t rank_slow(idx_t b, idx_t e) const {
      idx_t s = 0;
      for (idx_t i = b; i < e; i++) s += is_bit_set(i);
      return s;
    }
  };

}
