/* sampled_suffix_array.h -- host-side table of (rank, pos) samples of a partial
   BWT, with the public face of the reference's structure
   (include/sampled_suffix_array.h:36-213): entries `a[0..sz)`, sz a power of
   two at least twice the request, unused entries hold -1, put/get by rank,
   sort_by_pos and the two position iterators.

   On the MI355X path the samples are produced on the device, already ordered by
   position (the sampled positions depend only on a, b and the sample count);
   load_sorted() brings them in.  Lookup by rank uses the same open addressing
   as the reference (key = rank mod sz, linear probing), so code that walks
   `a[]` directly keeps working. */
#pragma once

#include <algorithm>

#include "typedefs.h"
#include "util.h"

namespace bwt {

  struct ssa_entry {
    idx_t rank; /* key */
    idx_t pos;  /* value */
  };

  struct ssa_pos_lt {
    int operator() (const ssa_entry & x, const ssa_entry & y) const { return x.pos < y.pos; }
  };

  struct ssa_iterator {
    ssa_entry * e;
    ssa_entry * end;
    ssa_iterator(ssa_entry * e_, ssa_entry * end_) : e(e_), end(end_) {}
    void next() { e++; }
    int has_next() { return e < end; }
  };

  struct ssa_reverse_iterator {
    ssa_entry * c;
    ssa_entry * begin;
    ssa_reverse_iterator(ssa_entry * c_, ssa_entry * begin_) : c(c_), begin(begin_) {}
    void next() { c--; }
    int has_next() { return c >= begin; }
  };

  struct sampled_suffix_array {
    idx_t n;           /* entries in use */
    idx_t sz;          /* capacity, a power of two */
    idx_t log_sz;
    ssa_entry * a;
    ssa_entry * begin; /* set by sort_by_pos: first used entry */
    ssa_entry * end;

    sampled_suffix_array() : n(0), sz(0), log_sz(0), a(0), begin(0), end(0) {}

    void init(idx_t sz_req, mallocator & mem, idx_t init_gran = 10000) {
      (void)init_gran;
      n = 0;
      log_sz = calc_log_sample_sa_sz(sz_req);
      sz = (idx_t)1 << (log_sz + 1);
      a = mem.new_<ssa_entry>(sz, mem_reason_sample_sa);
      for (idx_t i = 0; i < sz; i++) a[i].rank = a[i].pos = -1;
      begin = 0;
      end = a + sz;
    }

    void fini(mallocator & mem) {
      if (a) mem.delete_(a, sz, mem_reason_sample_sa);
      a = 0;
    }

    idx_t hash(idx_t r) { return r & (sz - 1); }

    /* returns 1 if (rank,pos) was inserted, 0 if rank was already there */
    int put(idx_t rank, idx_t pos) {
      assert(begin == 0);
      stat.start(ts_event_ssa_put);
      idx_t h = hash(rank);
      for (idx_t k = 0; k < sz; k++) {
        ssa_entry & e = a[(h + k) & (sz - 1)];
        if (e.rank == rank) { stat.end(ts_event_ssa_put); return 0; }
        if (e.rank == -1) {
          e.rank = rank;
          e.pos = pos;
          n++;
          stat.end(ts_event_ssa_put);
          return 1;
        }
      }
      check(0); /* table full */
      return 0;
    }

    /* position of the suffix of this rank, or -1 if it is not sampled */
    idx_t get(idx_t rank) {
      assert(begin == 0);
      stat.start(ts_event_ssa_get);
      idx_t h = hash(rank), found = -1;
      for (idx_t k = 0; k < sz; k++) {
        ssa_entry & e = a[(h + k) & (sz - 1)];
        if (e.rank == rank) { found = e.pos; break; }
        if (e.rank == -1) break;
      }
      stat.end(ts_event_ssa_get);
      return found;
    }

    idx_t count() {
      idx_t c = 0;
      for (idx_t i = 0; i < sz; i++) c += (a[i].rank != -1);
      return c;
    }

    /* orders the whole table by pos (unused entries, pos -1, come first);
       lookups by rank are no longer possible afterwards */
    void sort_by_pos(mallocator & mem, idx_t sort_rec_threshold = 30, idx_t merge_rec_threshold = 1000) {
      (void)mem; (void)sort_rec_threshold; (void)merge_rec_threshold;
      stat.start(ts_event_ssa_sort_by_pos);
      std::sort(a, a + sz, ssa_pos_lt());
      ssa_entry * p = a;
      while (p < a + sz && p->pos == -1) p++;
      begin = p;
      end = a + sz;
      stat.end(ts_event_ssa_sort_by_pos);
    }

    /* last entry with pos <= j (after sort_by_pos) */
    ssa_reverse_iterator pos_reverse_begin_from(idx_t j) {
      assert(begin);
      ssa_entry * lo = begin, * hi = end; /* first entry with pos > j */
      while (lo < hi) {
        ssa_entry * mid = lo + (hi - lo) / 2;
        if (mid->pos <= j) lo = mid + 1; else hi = mid;
      }
      return ssa_reverse_iterator(lo - 1, begin);
    }

    ssa_iterator pos_begin(mallocator & mem) {
      if (begin == 0) sort_by_pos(mem);
      return ssa_iterator(begin, end);
    }

    /* fills the table from k samples (absolute ranks); any previous content is dropped */
    void load(idx_t sz_req, const idx_t * ranks, const idx_t * poss, idx_t k, mallocator & mem) {
      if (a) fini(mem);
      init(sz_req, mem);
      for (idx_t i = 0; i < k; i++) put(ranks[i], poss[i]);
    }
  };

}
