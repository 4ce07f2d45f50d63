/* util.h -- checks, event statistics and memory accounting of the bwt:: API.

   Same names and observable behaviour as the reference's include/util.h
   (check_ :26-38, get_tsc :45, ts_stat :138-209, mem_reason_kind_t :215-233,
   int_log2 :244, calc_log_sample_sa_sz :257, mallocator :267-505), written for
   a library whose heavy lifting happens on the GPU: the event table is filled
   from device phase timers, and the allocator also records the device pool's
   high-water mark. */
#pragma once

#include <assert.h>
#include <stdio.h>
#include <stdlib.h>
#include <new>

#include "typedefs.h"
#include "opts.h"

namespace bwt {

  /* like assert, but never compiled out: prints and aborts when cond is false */
  inline int check_(int cond, const char * exp, const char * file, int line, const char * func) {
    if (!cond) {
      fprintf(stderr, "error:%s:%d: check faild (%s) in %s\n", file, line, exp, func);
      abort();
    }
    return 1;
  }
#define check(exp) check_(exp, #exp, __FILE__, __LINE__, __func__)

  typedef unsigned long long tsc_t;

  inline tsc_t get_tsc() {
    unsigned int lo, hi;
    asm volatile ("rdtsc" : "=a"(lo), "=d"(hi));
    return ((tsc_t)hi << 32) | lo;
  }

  /* cumulative time and call count of one named event */
  struct ts_record {
    long n;
    tsc_t start_time;
    tsc_t total_time;
    const char * name;
    ts_record() : n(0), start_time(0), total_time(0), name(0) {}
    void start() { assert(start_time == 0); start_time = get_tsc(); }
    void end() { assert(start_time); total_time += get_tsc() - start_time; start_time = 0; n++; }
    void add(tsc_t clocks, long calls) { total_time += clocks; n += calls; }
    void reset() { n = 0; start_time = 0; total_time = 0; }
    void print() {
      if (n) printf("%30s : %12llu / %9ld = %14.2f\n", name, total_time, n, total_time / (double)n);
    }
  };

  /* the reference's thirty events, in its order */
#define BWT_TS_EVENTS(X) \
  X(ssa_sort_by_pos) X(ssa_get) X(ssa_put) \
  X(gap_array_init) X(gap_array_fini) X(gap_array_insert_overflow) X(gap_array_wait_overflow) \
  X(gap_array_find_overflow) X(gap_array_inc) X(gap_array_get) X(gap_array_set_prefix_sum) X(gap_array_sum) \
  X(bwt_leaf) X(bwt_sample_sa) X(bwt_count_alphabets) X(bwt_build_wavelet_matrix) X(bwt_bin_search) \
  X(bwt_LF_map_mod) X(sbv_rank1) X(wm_rank) X(bwt_LF_map_mod2) X(bwt_sa) X(bwt_build_gap) \
  X(bwt_merge_loop) X(bwt_memcpy) X(bwt_resample_sa) X(bwt_merge_bwt) X(bwt_range_rec) X(bwt_fini) X(pmbwt)

  typedef enum {
#define X(name) ts_event_ ## name,
    BWT_TS_EVENTS(X)
#undef X
    n_ts_events
  } ts_event_kind_t;

  struct ts_stat {
    int level;
    ts_record events[n_ts_events];
    ts_stat() : level(0) {
#define X(ev) events[ts_event_ ## ev].name = #ev;
      BWT_TS_EVENTS(X)
#undef X
    }
    void start(ts_event_kind_t k) { if (level >= 1) events[k].start(); }
    void end(ts_event_kind_t k) { if (level >= 1) events[k].end(); }
    /* credit time measured elsewhere (device phase timers) to an event */
    void add(ts_event_kind_t k, tsc_t clocks, long calls) { if (level >= 1) events[k].add(clocks, calls); }
    void reset() { if (level >= 1) for (int k = 0; k < n_ts_events; k++) events[k].reset(); }
    void print() { if (level >= 1) for (int k = 0; k < n_ts_events; k++) events[k].print(); }
  };

  /* the one global event table (as in the reference: include from one translation unit) */
  inline ts_stat & stat_instance() { static ts_stat s; return s; }
  static ts_stat & stat = stat_instance();

  /* why a block of memory was allocated; the reference's sixteen, then ours */
#define BWT_MEM_REASONS(X) \
  X(leaf_sa) X(sort_leaf_sa) X(sample_sa) X(sort_sample_sa) X(count_alphabets) \
  X(succinct_bitvec_superblock) X(succinct_bitvec_block) X(succinct_bitvec_prefix_sum_temp) \
  X(wavelet_matrix_count_bits) X(wavelet_matrix_bitvec_array) X(wavelet_matrix_bitmaps) \
  X(gap_small) X(gap_overflow) X(gap_prefix_sum) X(gap_prefix_sum_temp) X(workspace_to_merge) \
  X(device_pool)

  typedef enum {
#define X(name) mem_reason_ ## name,
    BWT_MEM_REASONS(X)
#undef X
    n_mem_reasons,
  } mem_reason_kind_t;

  enum {
    superblock_bits = (1 << 15),
    block_bits      = (1 << 6),
  };

  /* bits needed to write x: l with 2^(l-1) <= x < 2^l */
  inline int int_log2(alpha_t x) {
    int l = 0;
    for (idx_t y = 1; y <= (idx_t)x; y += y) l++;
    return l;
  }

  /* ceil(log2(sz_req)) */
  inline idx_t calc_log_sample_sa_sz(idx_t sz_req) {
    idx_t log_sz = 0;
    for (idx_t sz = 1; sz < sz_req; sz *= 2) log_sz++;
    return log_sz;
  }

  /* Host allocator with per-reason accounting.
       level 0: plain new[] / delete[]
       level 1: also tracks bytes allocated, deallocated and the high-water mark
       level 2: also prints every event
     Constructed with N != 0 it asserts the reference's per-reason size bounds
     and live counts (at most one live block per reason, n_sample_sa sample
     tables, eight of each rank directory).  Device memory does not go through
     new_/delete_: the pool's high-water mark of a build is recorded once, under
     mem_reason_device_pool, by note_device(). */
  struct mallocator {
    tsc_t started;
    size_t allocated;
    size_t deallocated;
    size_t max_allocated;
    size_t device_peak; /* bytes, high-water mark of the GPU pool over all builds */
    idx_t N;
    int n_sample_sa;
    int level;
    const char * reasons[n_mem_reasons];
    int n_allocated[n_mem_reasons];
    bwt_opt opt;

    mallocator(bwt_opt & opt_) { construct(0, opt_); }
    mallocator(idx_t N_, bwt_opt & opt_) { construct(N_, opt_); }

    void reset(idx_t N_, bwt_opt & opt_) {
      started = 0;
      allocated = deallocated = max_allocated = 0;
      N = N_;
      opt = opt_;
      n_sample_sa = (int)calc_n_sample_sa(0, N, opt_) + 1;
    }

    idx_t calc_n_sample_sa(idx_t a, idx_t b, bwt_opt & o) {
      idx_t k = 1;
      while (b - a > o.bwt_rec_threshold) { a = (a + b) / 2; k++; }
      return k;
    }

    template<class T>
    T * new_(size_t n, mem_reason_kind_t why) {
      T * p = new T[n];
      if (N) check_bound(n, why);
      n_allocated[why]++;
      if (level >= 1) account(n * sizeof(T), (long)(n * sizeof(T)), why);
      return p;
    }

    template<class T>
    void delete_(T * p, size_t n, mem_reason_kind_t why) {
      assert(n_allocated[why] > 0);
      n_allocated[why]--;
      if (level >= 1) account(n * sizeof(T), -(long)(n * sizeof(T)), why);
      delete[] p;
    }

    /* one build's device high-water mark, booked as an allocation released at once */
    void note_device(size_t peak_bytes) {
      if (peak_bytes > device_peak) device_peak = peak_bytes;
      if (level >= 1) {
        account(peak_bytes, (long)peak_bytes, mem_reason_device_pool);
        account(peak_bytes, -(long)peak_bytes, mem_reason_device_pool);
      }
    }

    void update() {
      size_t live = allocated - deallocated;
      if (level >= 1 && live > max_allocated) max_allocated = live;
    }

    void print() {
      if (level >= 1) {
        printf("allocated     : %ld bytes\n", allocated);
        printf("deallocated   : %ld bytes\n", deallocated);
        printf("possible leak : %ld bytes\n", allocated - deallocated);
        printf("max allocated : %ld bytes\n", max_allocated);
        printf("device peak   : %ld bytes\n", device_peak);
      }
    }

  private:
    void construct(idx_t N_, bwt_opt & opt_) {
      level = 0;
      device_peak = 0;
      for (int k = 0; k < n_mem_reasons; k++) n_allocated[k] = 0;
#define X(name) reasons[mem_reason_ ## name] = #name;
      BWT_MEM_REASONS(X)
#undef X
      reset(N_, opt_);
    }

    void account(size_t bytes, long delta, mem_reason_kind_t why) {
      if (delta >= 0) allocated += bytes; else deallocated += bytes;
      tsc_t t = 0;
      if (started) t = get_tsc() - started; else started = get_tsc();
      if (level >= 2)
        printf("alloc/dealloc %9ld at %12llu -> %9ld bytes ( %s )\n",
               delta, t, allocated - deallocated, reasons[why]);
      update();
    }

    /* the size and live-count limits of each reason for an input of N symbols */
    void check_bound(size_t n, mem_reason_kind_t why) {
      size_t limit = 0;
      int live = 1;
      idx_t table = (idx_t)1 << (1 + calc_log_sample_sa_sz(min(N, opt.ssa_n_samples)));
      switch (why) {
      case mem_reason_leaf_sa:
      case mem_reason_sort_leaf_sa: limit = opt.bwt_rec_threshold; break;
      case mem_reason_count_alphabets: limit = opt.alpha_max + 1; assert(n == limit); break;
      case mem_reason_wavelet_matrix_count_bits:
        limit = ((N + opt.wavelet_matrix_sum_interval - 1) / opt.wavelet_matrix_sum_interval) * 2; break;
      case mem_reason_wavelet_matrix_bitvec_array: limit = int_log2(opt.alpha_max); break;
      case mem_reason_wavelet_matrix_bitmaps: limit = N + 64; break; /* planes padded to whole words */
      case mem_reason_gap_small:
      case mem_reason_workspace_to_merge: limit = N; break;
      case mem_reason_gap_overflow: limit = N / 64; break;
      case mem_reason_gap_prefix_sum: limit = N / opt.gap_sum_gran + 1; break;
      case mem_reason_succinct_bitvec_prefix_sum_temp:
        limit = max((4 * (N / superblock_bits + 1)) / opt.prefix_sum_gran - 1, 1); break;
      case mem_reason_gap_prefix_sum_temp:
        limit = max((4 * (N / opt.gap_sum_gran + 1)) / opt.prefix_sum_gran - 1, 1); break;
      case mem_reason_sort_sample_sa: limit = table; break;
      case mem_reason_sample_sa: limit = table; live = n_sample_sa; break;
      case mem_reason_succinct_bitvec_superblock: limit = N / superblock_bits + 1; live = 8; break;
      case mem_reason_succinct_bitvec_block: limit = N / block_bits + 1; live = 8; break;
      case mem_reason_device_pool: return;
      default: check(0);
      }
      (void)limit; (void)live;
      assert(n <= limit);
      assert(n_allocated[why] < live);
    }
  };

}
