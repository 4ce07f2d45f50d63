/* conc.h -- what is left of the reference's CPU concurrency layer
   (include/conc.h:10-299) in a build whose parallel work runs on the GPU:
   the three parallel_model names, so that programs written against the
   reference still compile with -Dparallel_model=..., and serial forms of its
   small host helpers.  Nothing on the hot path uses them. */
#pragma once

#include <stdint.h>
#include <string.h>
#include <algorithm>

#include "typedefs.h"
#include "util.h"

#define parallel_model_none       0
#define parallel_model_native_tbb 1
#define parallel_model_task       2

#if !defined(parallel_model)
#define parallel_model parallel_model_none
#endif

#define decl_task_group int
#define tg_run(tg, X) X
#define tg_wait(tg)   (void)tg

#define pfor(type, var, init, fin, gran) (void)gran; for (type var = init; var < fin; var++)
#define end_pfor
#define pfor_step(type, var, init, fin, step) for (type var = init; var < fin; var += step)
#define end_pfor_step

namespace bwt {

  inline bool cmp_and_set_8(volatile uint8_t * p, uint8_t x, uint8_t y) { return __sync_bool_compare_and_swap(p, x, y); }
  inline bool cmp_and_set_idx(volatile idx_t * p, idx_t x, idx_t y) { return __sync_bool_compare_and_swap(p, x, y); }
  inline void atomic_inc(volatile idx_t * p) { __sync_fetch_and_add(p, 1); }

  inline void * pmemcpy(void * dest, const void * src, size_t n, size_t gran) {
    (void)gran;
    return memcpy(dest, src, n);
  }

  /* x[a:b) becomes its exclusive prefix sum; returns the total */
  inline idx_t prefix_sum(idx_t * x, idx_t a, idx_t b, idx_t gran, mallocator & mem, mem_reason_kind_t reason) {
    (void)gran; (void)mem; (void)reason;
    idx_t run = 0;
    for (idx_t i = a; i < b; i++) { idx_t v = x[i]; x[i] = run; run += v; }
    return run;
  }

  template<typename T, typename Compare>
  void psort(T * first, T * last, const Compare & lt, mallocator & mem, mem_reason_kind_t reason,
             idx_t sort_rec_threshold, idx_t merge_rec_threshold) {
    (void)mem; (void)reason; (void)sort_rec_threshold; (void)merge_rec_threshold;
    std::sort(first, last, lt);
  }

  /* every element strictly below its successor (include/task_parallel_sort.h:146-153) */
  template<typename T, typename Compare>
  int check_sorted(T * a, T * e, const Compare & lt) {
    for (idx_t i = 0; i + 1 < (idx_t)(e - a); i++)
      if (!check(lt(a[i], a[i + 1]))) return 0;
    return 1;
  }

}
