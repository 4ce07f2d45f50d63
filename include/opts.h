/* opts.h -- bwt::bwt_opt, field for field the reference's knob set
   (include/opts.h:36-202) with the same set_defaults arithmetic (:214-291).

   Most *_gran fields are CPU task grain sizes; the MI355X path keeps them so
   that code written against the reference compiles, and ignores them.  The
   fields that steer the GPU path are mem_budget (through bwt_rec_threshold and
   ssa_n_samples), alpha_min/alpha_max, bwt_rec_threshold, ssa_n_samples and
   build_gap_segment_sz.  New zero-defaulted fields sit at the end. */
#pragma once

#include <stdio.h>

#include "typedefs.h"

namespace bwt {

/* name, type, value given by set_defaults when the field is 0 */
#define BWT_OPT_GRAINS(X)                         \
  X(sort_rec_threshold, idx_t, 30)                \
  X(merge_rec_threshold, idx_t, 10000)            \
  X(build_gap_segment_sz, idx_t, 1024)            \
  X(gap_sum_gran, idx_t, 128)                     \
  X(sa_to_bwt_gran, idx_t, 10000)                 \
  X(wavelet_matrix_sum_interval, idx_t, 4096)     \
  X(count_alpha_gran, idx_t, 10000)               \
  X(merge_gran, idx_t, 1000)                      \
  X(memcpy_gran, idx_t, 10000)                    \
  X(gap_init_gran, idx_t, 10000)                  \
  X(ssa_init_gran, idx_t, 10000)                  \
  X(sa_init_gran, idx_t, 10000)                   \
  X(ssa_sample_gran, idx_t, 10000)                \
  X(wavelet_matrix_add_zero_count_gran, idx_t, 10000) \
  X(wavelet_matrix_transpose_gran, idx_t, 10000)  \
  X(prefix_sum_gran, idx_t, 1000)                 \
  X(n_approx_workers, int, 256)

  struct bwt_opt {
    double mem_budget;       /* bytes of working memory per input symbol; 0 = 4.0 */
    alpha_t alpha_min;       /* smallest symbol besides the terminator; 0,0 = 1..255 */
    alpha_t alpha_max;       /* largest symbol */
    idx_t ssa_n_samples;     /* samples kept per partial BWT (derived from mem_budget) */
    idx_t bwt_rec_threshold; /* ranges up to this length are leaves (derived from mem_budget) */
#define X(name, type, dflt) type name;
    BWT_OPT_GRAINS(X)
#undef X
    int assert_level;
    /* MI355X additions */
    int device; /* HIP device ordinal used by this process, default 0 */
    int n_gpus; /* pmbwt: GPUs to spread the build over, devices device .. device+n_gpus-1 (0 or 1: one GPU).  The
                   recursion tree is the one bwt_rec_threshold gives; as many GPUs are used as it has subtrees at
                   their depth (a power of two) */
    int fit_leaves_to_device; /* 1: pmbwt replaces bwt_rec_threshold by the largest leaf the GPU's free memory holds
                   (n / 2^k for the smallest k that fits): the fewest merges, which on this hardware is the fastest tree
                   (one leaf of 3.1e9 symbols builds 5 times faster than mem_budget 4's sixteen).  L does not depend on
                   the tree; the sample set does.  0: the reference's tree */

    bwt_opt() {
      mem_budget = 0;
      alpha_min = alpha_max = 0;
      ssa_n_samples = bwt_rec_threshold = 0;
#define X(name, type, dflt) name = 0;
      BWT_OPT_GRAINS(X)
#undef X
      assert_level = 0;
      device = 0;
      n_gpus = 0;
      fit_leaves_to_device = 0;
    }

    /* number of partial BWTs alive on the right spine of the recursion */
    idx_t max_num_ssa(idx_t n, idx_t th) {
      idx_t k = 0;
      for (; n > th; n -= n / 2) k++;
      return k;
    }

    void set_defaults() { set_defaults(0, 0); }

    /* Fills every unset (zero) field.  bwt_rec_threshold and ssa_n_samples are
       always recomputed from mem_budget, exactly as the reference does: of the
       mem_budget - 2 bytes per symbol left after L and the merge workspace, 80 %
       pay for a leaf's suffix array and its sort buffer (16 bytes per leaf
       symbol) and 20 % for the sample tables of the live partial BWTs. */
    void set_defaults(const alpha_t * T, idx_t n) {
      (void)T;
      if (alpha_min == 0 && alpha_max == 0) { alpha_min = 1; alpha_max = 255; }
#define X(name, type, dflt) if (name == 0) name = dflt;
      BWT_OPT_GRAINS(X)
#undef X
      if (mem_budget == 0) mem_budget = 4.0;
      if (mem_budget < 2.5) {
        printf("warning: mem_budget %f too small. defaults to 4.0\n", mem_budget);
        mem_budget = 4.0;
      }
      const size_t idx2 = sizeof(idx_t) * 2;
      if (mem_budget >= 2 + idx2) {
        bwt_rec_threshold = n;
        ssa_n_samples = 2;
      } else {
        const double leaf_share = 0.8;
        bwt_rec_threshold = ((mem_budget - 2) * leaf_share * n) / idx2;
        bwt_rec_threshold = max(1000, bwt_rec_threshold);
        idx_t table_bytes = (mem_budget - 2) * (1 - leaf_share) * n;
        idx_t per_table = table_bytes / (1 + max_num_ssa(n, bwt_rec_threshold));
        ssa_n_samples = max(1000, (idx_t)(per_table / idx2));
      }
    }
  };

}
