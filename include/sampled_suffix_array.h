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
#include <thread>
#include <vector>

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

    /* runs body(t, nt) on nt host threads (the calling one included) when there is enough work for them */
    template <class F>
    static void on_threads(idx_t work, F body) {
      unsigned hw = std::thread::hardware_concurrency();
      idx_t nt = work < ((idx_t)1 << 20) ? 1 : (hw > 16 ? 16 : (hw < 1 ? 1 : (idx_t)hw));
      std::vector<std::thread> th;
      for (idx_t t = 1; t < nt; t++) th.push_back(std::thread(body, t, nt));
      body((idx_t)0, nt);
      for (size_t t = 0; t < th.size(); t++) th[t].join();
    }

    struct fill_body {
      ssa_entry * a; idx_t sz;
      void operator() (idx_t t, idx_t nt) const {
        for (idx_t i = sz * t / nt; i < sz * (t + 1) / nt; i++) a[i].rank = a[i].pos = -1;
      }
    };

    void init(idx_t sz_req, mallocator & mem, idx_t init_gran = 10000) {
      (void)init_gran;
      n = 0;
      log_sz = calc_log_sample_sa_sz(sz_req);
      sz = (idx_t)1 << (log_sz + 1);
      a = mem.new_<ssa_entry>(sz, mem_reason_sample_sa);
      fill_body fb = { a, sz };
      on_threads(sz, fb); /* first touch of up to gigabytes of fresh pages */
      begin = 0;
      end = a + sz;
    }

    void fini(mallocator & mem) {
      if (a) mem.delete_(a, sz, mem_reason_sample_sa);
      a = 0;
    }

    idx_t hash(idx_t r) { return r & (sz - 1); }

    /* returns 1 if (rank,pos) was inserted, 0 if rank was already there.  May be called from several threads
       at once (include/sampled_suffix_array.h:183-211 there): the key slot is claimed with a compare-and-swap. */
    int put(idx_t rank, idx_t pos) {
      assert(begin == 0);
      idx_t h = hash(rank);
      for (idx_t k = 0; k < sz; k++) {
        ssa_entry & e = a[(h + k) & (sz - 1)];
        idx_t seen = __sync_val_compare_and_swap(&e.rank, (idx_t)-1, rank);
        if (seen == -1) {
          e.pos = pos;
          __sync_fetch_and_add(&n, (idx_t)1);
          return 1;
        }
        if (seen == rank) return 0;
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

    /* One slice of a bulk load: claims the key slot with a compare-and-swap, like the reference's concurrent put
       (include/sampled_suffix_array.h:183-211 there), and asks for the table line a few samples ahead: the
       table is gigabytes for a genome-sized text and every insertion is a cache miss. */
    struct load_body {
      sampled_suffix_array * s; const idx_t * ranks; const idx_t * poss; idx_t k; idx_t * inserted;
      void operator() (idx_t t, idx_t nt) const {
        const idx_t lo = k * t / nt, hi = k * (t + 1) / nt, mask = s->sz - 1;
        ssa_entry * a = s->a;
        idx_t mine = 0;
        for (idx_t i = lo; i < hi; i++) {
          if (i + 8 < hi) __builtin_prefetch(&a[ranks[i + 8] & mask], 1);
          const idx_t rank = ranks[i];
          for (idx_t h = rank & mask, probes = 0; ; h = (h + 1) & mask, probes++) {
            check(probes < s->sz); /* table full */
            idx_t seen = __sync_val_compare_and_swap(&a[h].rank, (idx_t)-1, rank);
            if (seen == -1) { a[h].pos = poss[i]; mine++; break; }
            if (seen == rank) break;
          }
        }
        __sync_fetch_and_add(inserted, mine);
      }
    };

    /* fills the table from k samples (absolute ranks); any previous content is dropped */
    void load(idx_t sz_req, const idx_t * ranks, const idx_t * poss, idx_t k, mallocator & mem) {
      if (a) fini(mem);
      init(sz_req, mem);
      stat.start(ts_event_ssa_put);
      idx_t inserted = 0;
      load_body lb = { this, ranks, poss, k, &inserted };
      on_threads(k, lb);
      n = inserted;
      stat.end(ts_event_ssa_put);
    }
  };

}
