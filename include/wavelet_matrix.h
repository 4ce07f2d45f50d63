/* wavelet_matrix.h -- host-side rank structure over a block's L, used by the
   optional checks on a finished BWT (init_extra / ibwt / sa / LF_map_mod on the
   struct that pmbwt returns).  The merges themselves use the device structure
   in bwt_amd/csrc/wavelet.h.

   Interface of the reference's wavelet_matrix (include/wavelet_matrix.h:198-290):
   init(s, n, w, mem, alpha0, alpha1, ...), rank(c, i) = occurrences of c in
   s[0:i), rank_slow, fini.  Per level: one bit plane in 64-bit words and a
   two-level rank directory of the reference's proportions, so that the
   mallocator's per-reason bounds (include/util.h) hold: one count per 2^15 bits
   and one 16-bit count per 64-bit word, relative to its 2^15-bit group. */
#pragma once

#include <stdint.h>
#include <string.h>

#include "typedefs.h"
#include "util.h"

namespace bwt {

  struct wavelet_matrix {
  private:
    alpha_t * s;
    idx_t n;
    int alpha_width;
    idx_t n_words;      /* per level */
    uint64_t * bits;    /* alpha_width * n_words */
    idx_t * super[8];   /* ones before each group of 512 words */
    uint16_t * blk[8];  /* ones before each word inside its group */
    idx_t n_super;
    idx_t n_zeros[8];
    alpha_t * scratch;  /* the caller's workspace w, n symbols */

  public:
    wavelet_matrix() : s(0), n(0), alpha_width(0), n_words(0), bits(0), n_super(0), scratch(0) {}

    void init(alpha_t * s_, idx_t n_, alpha_t * w, mallocator & mem,
              alpha_t alpha0_ = 0, alpha_t alpha1_ = 255,
              idx_t sum_interval = 10000, idx_t add_zero_count_gran = 10000,
              size_t memcpy_gran = 10000, idx_t transpose_gran = 10000, idx_t prefix_sum_gran = 1000) {
      (void)alpha0_; (void)sum_interval; (void)add_zero_count_gran; (void)memcpy_gran;
      (void)transpose_gran; (void)prefix_sum_gran;
      s = s_;
      n = n_;
      scratch = w;
      alpha_width = int_log2(alpha1_);
      assert(alpha_width > 0 && alpha_width <= 8);
      n_words = n / 64 + 1;
      n_super = n / superblock_bits + 1;
      bits = (uint64_t *)mem.new_<alpha_t>((size_t)alpha_width * n_words * 8, mem_reason_wavelet_matrix_bitmaps);
      memset(bits, 0, (size_t)alpha_width * n_words * 8);
      for (int l = 0; l < alpha_width; l++) {
        super[l] = mem.new_<idx_t>(n_super, mem_reason_succinct_bitvec_superblock);
        blk[l] = mem.new_<uint16_t>(n_words, mem_reason_succinct_bitvec_block);
      }
      alpha_t * cur = new alpha_t[n + 1];
      alpha_t * nxt = new alpha_t[n + 1];
      memcpy(cur, s, n);
      for (int l = 0; l < alpha_width; l++) {
        int shift = alpha_width - 1 - l;
        uint64_t * plane = bits + (size_t)l * n_words;
        idx_t z = 0;
        for (idx_t i = 0; i < n; i++) {
          if ((cur[i] >> shift) & 1) plane[i >> 6] |= (uint64_t)1 << (i & 63); else z++;
        }
        n_zeros[l] = z;
        const idx_t words_per_super = superblock_bits / 64;
        idx_t acc = 0, in_group = 0;
        for (idx_t k = 0; k < n_words; k++) {
          if (k % words_per_super == 0) { super[l][k / words_per_super] = acc; in_group = 0; }
          blk[l][k] = (uint16_t)in_group;
          idx_t c = __builtin_popcountll(plane[k]);
          acc += c;
          in_group += c;
        }
        idx_t p0 = 0, p1 = z; /* zeros first, ones after, order kept */
        for (idx_t i = 0; i < n; i++) {
          if ((cur[i] >> shift) & 1) nxt[p1++] = cur[i]; else nxt[p0++] = cur[i];
        }
        alpha_t * t = cur; cur = nxt; nxt = t;
      }
      delete[] cur;
      delete[] nxt;
    }

    void fini(mallocator & mem) {
      if (bits) {
        mem.delete_((alpha_t *)bits, (size_t)alpha_width * n_words * 8, mem_reason_wavelet_matrix_bitmaps);
        for (int l = 0; l < alpha_width; l++) {
          mem.delete_(super[l], n_super, mem_reason_succinct_bitvec_superblock);
          mem.delete_(blk[l], n_words, mem_reason_succinct_bitvec_block);
        }
      }
      bits = 0;
    }

    idx_t rank1(int l, idx_t i) const {
      const uint64_t * plane = bits + (size_t)l * n_words;
      uint64_t w = plane[i >> 6] & (((uint64_t)1 << (i & 63)) - 1);
      return super[l][i / superblock_bits] + blk[l][i >> 6] + __builtin_popcountll(w);
    }

    /* occurrences of c in s[0:i) */
    idx_t rank(alpha_t c, idx_t i) {
      stat.start(ts_event_wm_rank);
      idx_t p = 0, q = i;
      for (int l = 0; l < alpha_width; l++) {
        if ((c >> (alpha_width - 1 - l)) & 1) {
          p = n_zeros[l] + rank1(l, p);
          q = n_zeros[l] + rank1(l, q);
        } else {
          p -= rank1(l, p);
          q -= rank1(l, q);
        }
      }
      stat.end(ts_event_wm_rank);
      return q - p;
    }

    idx_t rank_slow(alpha_t c, idx_t i) {
      idx_t k = 0;
      for (idx_t j = 0; j < i; j++) k += (s[j] == c);
      return k;
    }
  };

}
