/* suffix_array.h -- suffix comparison and the leaf suffix array of the bwt::
   API (the reference's include/suffix_array.h:14-49), with sa_range computed on
   the GPU by radix sort + prefix doubling instead of a comparison sort. */
#pragma once

#include <assert.h>
#include <stdio.h>

#include "typedefs.h"
#include "util.h"
#include "device_session.h"

namespace bwt {

  /* T[a:] < T[b:] in the full text; distinct suffixes differ before the end
     because the terminating 0 occurs once */
  struct idx_less {
    const alpha_t * T;
    idx_t n;
    idx_less(const alpha_t * T_, idx_t n_) : T(T_), n(n_) {}
    bool operator() (const idx_t & a, const idx_t & b) const {
      idx_t i = a, j = b;
      while (T[i] == T[j]) { i++; j++; assert(i < n && j < n); }
      return T[i] < T[j];
    }
  };

  /* SA[a:b) = the positions a..b-1 ordered by their full-text suffixes */
  inline int sa_range(const alpha_t * T, idx_t n, idx_t a, idx_t b, idx_t * SA, mallocator & mem,
                      idx_t sort_rec_threshold = 30, idx_t merge_rec_threshold = 1000, idx_t sa_init_gran = 10000) {
    (void)sort_rec_threshold; (void)merge_rec_threshold; (void)sa_init_gran;
    bwt_hip_ctx * c = device_session::get().with_text(T, n, mem.opt.device);
    bwt_hip_ok(bwt_hip_leaf_sa(c, a, b, (int64_t *)(SA + a)));
    return 1;
  }

  /* debug listing of SA[a:b) (reference include/suffix_array.h:67-81): rank, position, the symbol in front of
     the suffix ('$' for the terminator) and the suffix itself, one line per rank */
  inline void sa_show(const alpha_t * T, idx_t n, idx_t a, idx_t b, idx_t * SA) {
    for (idx_t i = a; i < b; i++) {
      const idx_t o = SA[i];
      assert(o >= a && o < b);
      const alpha_t p = o ? T[o - 1] : T[n - 1];
      printf("%2lu %2lu %c %s\n", (unsigned long)i, (unsigned long)o, p ? p : '$', (const char *)(T + o));
    }
  }

}
