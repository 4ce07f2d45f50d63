/* gap_array.h -- host-side counters gap[a..b] with sparse prefix sums, with the
   public face of the reference's structure (include/gap_array.h:15-334):
   init / inc / get / set_prefix_sum / sum / fini and the same fields.

   Nothing on the MI355X path uses it: there a merge never materialises
   per-position counters -- the g values of the LF chains are bucketed and
   counted in LDS, and what comes out is the inclusive prefix sum S[i] directly
   (bwt_amd/csrc/gap.hip).  It is here for host code written against the
   reference.

   Storage as in the reference (:20-29): one byte per index for counts up to
   254; an index whose count goes beyond that has the byte 255 and an entry
   in a small open-addressing table that holds the count above 255; exclusive
   prefix sums at every sum_interval-th index (absolute indices: ps[j] covers the
   indices a .. j*sum_interval - 1).  inc() may be called from several threads
   at once (:206-260 there): every update is a compare-and-swap or an atomic add. */
#pragma once

#include <stdint.h>
#include <stdio.h>

#include "typedefs.h"
#include "util.h"

namespace bwt {

  struct overflow_entry {
    idx_t i; /* index, -1 = free */
    idx_t c; /* count above 255 */
  };

  struct gap_array {
    idx_t a;
    idx_t b;
    idx_t m; /* entries of the overflow table, a power of two */
    uint8_t * small;           /* small[a..b]: counts 0..254, 255 = see overflow */
    overflow_entry * overflow;
    idx_t * ps;                /* ps[ps_a..ps_b] */
    idx_t ps_a;
    idx_t ps_b;
    idx_t sum_interval;

    gap_array() : a(0), b(0), m(0), small(0), overflow(0), ps(0), ps_a(0), ps_b(0), sum_interval(0) {}

    /* a power of two of at least one entry per 128 indices (:93-99) */
    idx_t calc_overflow_array_sz(idx_t n) {
      idx_t sz = 16;
      while (sz * 128 < n) sz *= 2;
      return sz;
    }

    void init(idx_t a_, idx_t b_, mallocator & mem, idx_t sum_interval_ = 128, idx_t init_gran = 10000) {
      (void)init_gran;
      stat.start(ts_event_gap_array_init);
      a = a_;
      b = b_;
      sum_interval = sum_interval_;
      m = calc_overflow_array_sz(b - a + 1);
      ps_a = a / sum_interval;
      ps_b = b / sum_interval;
      small = mem.new_<uint8_t>(b - a + 1, mem_reason_gap_small) - a;
      overflow = mem.new_<overflow_entry>(m, mem_reason_gap_overflow);
      ps = mem.new_<idx_t>(ps_b - ps_a + 1, mem_reason_gap_prefix_sum) - ps_a;
      for (idx_t i = a; i <= b; i++) small[i] = 0;
      for (idx_t j = 0; j < m; j++) { overflow[j].i = -1; overflow[j].c = 0; }
      for (idx_t j = ps_a; j <= ps_b; j++) ps[j] = 0;
      stat.end(ts_event_gap_array_init);
    }

    void fini(mallocator & mem) {
      stat.start(ts_event_gap_array_fini);
      if (small) mem.delete_(small + a, b - a + 1, mem_reason_gap_small);
      if (overflow) mem.delete_(overflow, m, mem_reason_gap_overflow);
      if (ps) mem.delete_(ps + ps_a, ps_b - ps_a + 1, mem_reason_gap_prefix_sum);
      small = 0;
      overflow = 0;
      ps = 0;
      stat.end(ts_event_gap_array_fini);
    }

    /* debug listing (reference include/gap_array.h:70-90): the layout, then every non-zero byte counter and every
       overflow entry in use */
    void show() {
      printf("range: [%ld,%ld] %ld elements\n", (long)a, (long)b, (long)(b - a + 1));
      printf("overflow: %ld elements\n", (long)m);
      printf("prefix sum: [%ld,%ld] %ld elements\n", (long)ps_a, (long)ps_b, (long)(ps_b - ps_a + 1));
      printf("sum interval: %ld\n", (long)sum_interval);
      if (small) {
        printf("small:\n");
        for (idx_t i = a; i <= b; i++)
          if (small[i]) printf("%6ld : %d\n", (long)i, (int)small[i]);
      }
      if (overflow) {
        printf("overflow:\n");
        for (idx_t j = 0; j < m; j++)
          if (overflow[j].i != -1) printf("%6ld : %ld %ld\n", (long)j, (long)overflow[j].i, (long)overflow[j].c);
      }
    }

    /* Sets gap[i] = count for an index that is still 0 (a bulk fill from counts computed elsewhere; one thread). */
    void set_count(idx_t i, idx_t count) {
      assert(a <= i && i <= b && small[i] == 0);
      if (count < 255) { small[i] = (uint8_t)count; return; }
      overflow[overflow_slot(i, 1)].c = count - 255;
      small[i] = 255;
    }

    /* the table slot of index i; claims a free one if `insert` (several threads may ask for the same i) */
    idx_t overflow_slot(idx_t i, int insert) {
      for (idx_t h = i & (m - 1), probes = 0; probes < m; h = (h + 1) & (m - 1), probes++) {
        idx_t seen = __atomic_load_n(&overflow[h].i, __ATOMIC_ACQUIRE);
        if (seen == -1 && insert) {
          idx_t expected = -1;
          if (__atomic_compare_exchange_n(&overflow[h].i, &expected, i, 0, __ATOMIC_ACQ_REL, __ATOMIC_ACQUIRE)) return h;
          seen = expected;
        }
        if (seen == i) return h;
        if (seen == -1) return -1;
      }
      check(!insert); /* table full */
      return -1;
    }

    void inc(idx_t i) {
      assert(a <= i && i <= b);
      uint8_t * p = &small[i];
      for (;;) {
        uint8_t x = __atomic_load_n(p, __ATOMIC_ACQUIRE);
        if (x < 254) {
          if (__atomic_compare_exchange_n(p, &x, (uint8_t)(x + 1), 0, __ATOMIC_ACQ_REL, __ATOMIC_ACQUIRE)) return;
        } else if (x == 254) {
          /* the entry first, then the byte: whoever sees 255 finds the entry */
          overflow_slot(i, 1);
          if (__atomic_compare_exchange_n(p, &x, (uint8_t)255, 0, __ATOMIC_ACQ_REL, __ATOMIC_ACQUIRE)) return;
        } else {
          __atomic_fetch_add(&overflow[overflow_slot(i, 1)].c, (idx_t)1, __ATOMIC_ACQ_REL);
          return;
        }
      }
    }

    idx_t get(idx_t i) {
      assert(a <= i && i <= b);
      const uint8_t x = small[i];
      if (x < 255) return x;
      const idx_t h = overflow_slot(i, 0);
      check(h != -1);
      return 255 + overflow[h].c;
    }

    /* ps[j] = gap[a] + ... + gap[j * sum_interval - 1] */
    void set_prefix_sum(mallocator & mem, idx_t prefix_sum_gran = 1000) {
      (void)mem; (void)prefix_sum_gran;
      stat.start(ts_event_gap_array_set_prefix_sum);
      idx_t run = 0;
      for (idx_t j = ps_a; j <= ps_b; j++) {
        ps[j] = run;
        const idx_t lo = j * sum_interval < a ? a : j * sum_interval;
        const idx_t hi = (j + 1) * sum_interval - 1 > b ? b : (j + 1) * sum_interval - 1;
        for (idx_t k = lo; k <= hi; k++) run += get(k);
      }
      stat.end(ts_event_gap_array_set_prefix_sum);
    }

    /* gap[a] + ... + gap[rank], after set_prefix_sum (:307-334) */
    idx_t sum(idx_t rank) {
      assert(a <= rank && rank <= b);
      const idx_t j = rank / sum_interval;
      idx_t s = ps[j];
      for (idx_t k = (j * sum_interval < a ? a : j * sum_interval); k <= rank; k++) s += get(k);
      return s;
    }
  };

}
