/* bwt.h -- the bwt:: entry points of taura/bwt on MI355X.

   Drop-in for the reference's include/bwt.h: same namespace, same struct bwt
   fields and methods, same free functions
       bwt_leaf   (reference include/bwt.h:615-635)
       bwt_merge  (:791-847)
       bwt_rec    (:854-865)
       pmbwt      (:867-883)
       check_equal, random_init (:885-896)
   Host code is thin: it stages T, L and the samples and calls the extern "C"
   layer of libbwt_hip.so (bwt_hip.h), where the leaf suffix sorts, the wavelet
   matrices, the LF-mapping chains, the gap histogram/scan and the merge scatter
   run as gfx950 kernels.  Link with -lbwt_hip.

   What differs from the reference, on purpose:
     - a partial BWT keeps a device handle (`dev`) so that merges consume blocks
       that are already in HBM; L[a:b) and the sample table are also copied to
       the host after every public call, as the reference leaves them;
     - the workspace W of bwt_merge/bwt_rec is accepted and not used (merges
       write to a fresh device buffer); pmbwt does not allocate it;
     - build_gap and resample_sa exist for host code written against them
       (build_gap: the chains and the histogram on the device through
       bwt_hip_gap, the counters then filled into the caller's gap_array);
       bwt_merge itself does not go through them: a merge stays on the device;
     - failures print to stderr and abort(), as check() does in the reference.
   Like the reference, include this from one translation unit per program. */
#pragma once

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "typedefs.h"
#include "opts.h"
#include "util.h"
#include "conc.h"
#include "device_session.h"
#include "suffix_array.h"
#include "wavelet_matrix.h"
#include "sampled_suffix_array.h"
#include "succinct_bit_vector.h"
#include "gap_array.h"

namespace bwt {

  struct bwt {
    /* BWT of T[a:b) in L[a:b): a permutation of T[a-1:b-1] (T[-1] = T[n-1]) */
    const alpha_t * T;
    idx_t n;
    alpha_t * L;
    idx_t a;
    idx_t b;
    alpha_t e_char;            /* T[b-1], the symbol L lacks */
    sampled_suffix_array ssa;
    idx_t s_rank;              /* rank of T[a:]   (L[s_rank] = T[a-1] is foreign to the block) */
    idx_t e_rank;              /* rank of T[b-1:] */
    idx_t * C;                 /* C[c] = symbols < c in T[a:b); after init_extra */
    wavelet_matrix wm;         /* of L[a:b) with L[s_rank] hidden; after init_extra */
    bwt_hip_block * dev;       /* the same block in HBM, or 0 */

    bwt() : T(0), n(0), L(0), a(0), b(0), e_char(0), s_rank(-1), e_rank(-1), C(0), dev(0) {}

    void init(const alpha_t * T_, idx_t n_, alpha_t * L_, idx_t a_, idx_t b_) {
      T = T_; n = n_; L = L_; a = a_; b = b_;
      e_char = T[b - 1];
      s_rank = e_rank = -1;
      C = 0;
      dev = 0;
    }

    /* what a merge (or a check by ibwt / sa) needs on top of L: C[] and the rank structure */
    void init_extra(alpha_t * W, mallocator & mem, bwt_opt & opt) {
      count_alphabets(mem, opt);
      build_wavelet_matrix(W, mem, opt);
    }

    void fini(mallocator & mem, bwt_opt & opt) {
      stat.start(ts_event_bwt_fini);
      if (C) {
        wm.fini(mem);
        mem.delete_(C, opt.alpha_max + 1, mem_reason_count_alphabets);
        C = 0;
      }
      ssa.fini(mem);
      if (dev) {
        bwt_hip_block_free(device_session::get().ctx, dev);
        dev = 0;
      }
      stat.end(ts_event_bwt_fini);
    }

    /* symbol counts of the ORIGINAL substring: L's histogram, minus T[a-1], plus T[b-1] */
    void count_alphabets(mallocator & mem, bwt_opt & opt) {
      stat.start(ts_event_bwt_count_alphabets);
      assert(C == 0);
      const idx_t A = opt.alpha_max;
      C = mem.new_<idx_t>(A + 1, mem_reason_count_alphabets);
      for (idx_t c = 0; c <= A; c++) C[c] = 0;
      for (idx_t i = a; i < b; i++) { check(L[i] <= A); C[L[i]]++; }
      C[L[s_rank]]--;
      C[e_char]++;
      idx_t before = 0;
      for (idx_t c = 0; c <= A; c++) { idx_t k = C[c]; C[c] = before; before += k; }
      stat.end(ts_event_bwt_count_alphabets);
    }

    void build_wavelet_matrix(alpha_t * W, mallocator & mem, bwt_opt & opt) {
      stat.start(ts_event_bwt_build_wavelet_matrix);
      alpha_t foreign = L[s_rank];
      L[s_rank] = 0; /* T[a-1] must not be counted by rank() */
      wm.init(L + a, b - a, W ? W + a : 0, mem, opt.alpha_min, opt.alpha_max);
      L[s_rank] = foreign;
      stat.end(ts_event_bwt_build_wavelet_matrix);
    }

    /* rank of c.T[SA[i]:] among the block's suffixes, for a suffix of the block itself */
    idx_t LF_map_mod(alpha_t c, idx_t i) {
      stat.start(ts_event_bwt_LF_map_mod);
      idx_t k = C[c] + wm.rank(c, i - a) + a;
      if (c == e_char && k >= e_rank) k++; /* T[b-1:] is not seen through L */
      stat.end(ts_event_bwt_LF_map_mod);
      return k;
    }

    /* the same for a suffix T[j:] from outside the block: T[b-1:] counts only if it is smaller */
    idx_t LF_map_mod2(alpha_t c, idx_t i, idx_t j) {
      stat.start(ts_event_bwt_LF_map_mod2);
      idx_t k = C[c] + wm.rank(c, i - a) + a;
      if (c == e_char && k >= e_rank && idx_less(T, n)(b - 1, j)) k++;
      stat.end(ts_event_bwt_LF_map_mod2);
      return k;
    }

    /* Positions of T[a:b) that L does not decode to, plus samples that are not where the decoding passes:
       0 iff L and the sample table describe T[a:b).  Decoded on the device, one LF walk per sample, while the
       block is still in HBM (no init_extra needed); by ibwt() on the host otherwise (after init_extra). */
    idx_t round_trip_mismatches() {
      if (dev) {
        int64_t bad = -1;
        bwt_hip_ok(bwt_hip_block_ibwt(device_session::get().ctx, dev, 0, &bad));
        return (idx_t)bad;
      }
      alpha_t * I = new alpha_t[b - a];
      ibwt(I - a);
      idx_t bad = 0;
      for (idx_t j = a; j < b; j++) bad += (I[j - a] != T[j]);
      delete[] I;
      return bad;
    }

    /* inverse transform of the block into I[a:b) */
    void ibwt(alpha_t * I) {
      idx_t r = e_rank;
      alpha_t c = e_char;
      for (idx_t j = b - 1; j >= a; j--) {
        I[j] = c;
        c = L[r];
        r = LF_map_mod(c, r);
      }
    }

    /* SA[r]: walk backwards to the nearest sample */
    idx_t sa(idx_t r) {
      stat.start(ts_event_bwt_sa);
      idx_t steps = 0;
      idx_t p;
      while ((p = ssa.get(r)) == -1) {
        r = LF_map_mod(L[r], r);
        steps++;
      }
      stat.end(ts_event_bwt_sa);
      return p + steps;
    }

    /* i with T[sa(i-1):] < T[x:] < T[sa(i):]; a if T[x:] is below all, b if above all */
    idx_t bin_search(idx_t x, bwt_opt & opt) {
      (void)opt;
      stat.start(ts_event_bwt_bin_search);
      idx_less lt(T, n);
      idx_t res;
      if (lt(x, sa(a))) res = a;
      else if (lt(sa(b - 1), x)) res = b;
      else {
        idx_t lo = a, hi = b - 1; /* T[sa(lo):] < T[x:] < T[sa(hi):] */
        while (hi - lo > 1) {
          idx_t mid = (lo + hi) / 2;
          if (lt(x, sa(mid))) hi = mid; else lo = mid;
        }
        res = hi;
      }
      stat.end(ts_event_bwt_bin_search);
      return res;
    }

    /* host form of the leaf sampling, for callers that hold a suffix array
       (sa[r] for r in [a,b)): keeps (r, sa[r]) wherever the sample index
       floor((i-a)(ns-1)/(b-a-1)) advances; a and b-1 are always kept */
    int sample_sa(idx_t * sa_, idx_t ns, mallocator & mem, bwt_opt & opt) {
      stat.start(ts_event_bwt_sample_sa);
      ssa.init(ns, mem, opt.ssa_init_gran);
      for (idx_t r = a; r < b; r++) {
        idx_t i = sa_[r];
        if (i == a) s_rank = r;
        if (i + 1 == b) e_rank = r;
        if (b - a == 1) { ssa.put(r, i); continue; }
        idx_t x = floor((i - 1 - a) * (ns - 1) / (double)(b - a - 1));
        idx_t y = floor((i - a) * (ns - 1) / (double)(b - a - 1));
        if (x != y) ssa.put(r, i);
      }
      ssa.n = min(ns, b - a);
      stat.end(ts_event_bwt_sample_sa);
      return 1;
    }

    /* The samples of the merged block [a,b) from those of its halves l = [a,c) and r = [c,b)
       (reference include/bwt.h:366-453): the two sets in position order form one sequence of nn samples, of which
       sample i is kept while (kept so far) * (nn-1) <= i * (ns-1); a left sample's new rank is its rank plus the
       right suffixes in front of it, gap[a] + ... + gap[rank]; a right sample's rank has had its g added by
       build_gap and lies c too high.  Sets s_rank and e_rank of this block. */
    int resample_sa(sampled_suffix_array & l, sampled_suffix_array & r, idx_t c, idx_t ns, gap_array & gap,
                    mallocator & mem, bwt_opt & opt) {
      stat.start(ts_event_bwt_resample_sa);
      ssa.init(ns, mem, opt.ssa_init_gran);
      const idx_t nn = l.n + r.n;
      idx_t seen = 0, kept = 0;
      for (ssa_iterator it = l.pos_begin(mem); it.has_next(); it.next(), seen++) {
        const idx_t rank = it.e->rank, pos = it.e->pos;
        assert(a <= rank && rank < c);
        if (kept * (nn - 1) <= seen * (ns - 1)) {
          const idx_t nr = rank + gap.sum(rank);
          assert(a <= nr && nr < b);
          ssa.put(nr, pos);
          if (pos == a) s_rank = nr;
          kept++;
        }
      }
      for (ssa_iterator it = r.pos_begin(mem); it.has_next(); it.next(), seen++) {
        const idx_t nr = it.e->rank - c, pos = it.e->pos;
        assert(a <= nr && nr < b);
        if (kept * (nn - 1) <= seen * (ns - 1)) {
          ssa.put(nr, pos);
          if (pos == b - 1) e_rank = nr;
          kept++;
        }
      }
      check(s_rank != -1 && e_rank != -1);
      stat.end(ts_event_bwt_resample_sa);
      return 1;
    }
  };

  /* ---- staging between struct bwt and a device block ---------------------- */

  /* copies the device block's scalars, L[a:b) and samples into t (t.T, t.n, t.L set by init) */
  inline void fetch_block(bwt & t, bwt_hip_ctx * c, bwt_hip_block * h, mallocator & mem, bwt_opt & opt, bool want_L = true) {
    int64_t a, b, s, e, ns;
    int ec;
    bwt_hip_ok(bwt_hip_block_info(h, &a, &b, &s, &e, &ec, &ns));
    check(a == t.a && b == t.b);
    t.s_rank = s;
    t.e_rank = e;
    t.e_char = (alpha_t)ec;
    t.dev = h;
    if (want_L) bwt_hip_ok(bwt_hip_block_get_L(c, h, t.L + a));
    idx_t * rk = new idx_t[ns];
    idx_t * ps = new idx_t[ns];
    bwt_hip_ok(bwt_hip_block_get_samples(c, h, (int64_t *)rk, (int64_t *)ps, ns));
    t.ssa.load(min(b - a, opt.ssa_n_samples), rk, ps, ns, mem);
    delete[] rk;
    delete[] ps;
  }

  /* makes sure t has a device block, uploading L[a:b) and the samples if it has none */
  inline bwt_hip_block * device_block(bwt & t, bwt_hip_ctx * c, mallocator & mem) {
    if (t.dev) return t.dev;
    idx_t k = 0;
    idx_t * rk = new idx_t[t.ssa.sz];
    idx_t * ps = new idx_t[t.ssa.sz];
    for (ssa_iterator it = t.ssa.pos_begin(mem); it.has_next(); it.next()) {
      rk[k] = it.e->rank;
      ps[k] = it.e->pos;
      k++;
    }
    bwt_hip_ok(bwt_hip_block_upload(c, t.a, t.b, t.L + t.a, t.s_rank, t.e_rank,
                                    (const int64_t *)rk, (const int64_t *)ps, k, &t.dev));
    delete[] rk;
    delete[] ps;
    return t.dev;
  }

  /* ---- the pipeline --------------------------------------------------------- */

  /* BWT of T[a:b) into L[a:b), from its suffix array in full-text order */
  inline bwt bwt_leaf(const alpha_t * T, idx_t n, idx_t a, idx_t b, alpha_t * L, mallocator & mem, bwt_opt & opt) {
    stat.start(ts_event_bwt_leaf);
    assert(a < b);
    bwt_hip_ctx * c = device_session::get().with_text(T, n, opt.device, !device_session::get().text_is_stable);
    bwt_hip_block * h = 0;
    bwt_hip_ok(bwt_hip_leaf(c, a, b, opt.ssa_n_samples, &h));
    bwt res;
    res.init(T, n, L, a, b);
    fetch_block(res, c, h, mem, opt);
    stat.end(ts_event_bwt_leaf);
    return res;
  }

  /* The gap array of a merge (reference include/bwt.h:730-787): for every position j of r, g_j = l.a + the number
     of l's suffixes below T[j:]; gap[g_j] is counted, r's samples get g added to their rank (r.ssa is left ordered
     by position, as there), and the sparse prefix sums are set.  The LF chains and the histogram run on the device
     (bwt_hip_gap, which leaves both blocks as they are); the counters and the g values come back to fill the caller's
     gap_array.  `gap` has been initialised over [l.a, l.b]. */
  inline int build_gap(bwt & l, bwt & r, gap_array & gap, mallocator & mem, bwt_opt & opt) {
    stat.start(ts_event_bwt_build_gap);
    assert(l.T == r.T && l.n == r.n && l.L == r.L && l.b == r.a);
    check(gap.a == l.a && gap.b == l.b);
    bwt_hip_ctx * c = device_session::get().with_text(l.T, l.n, opt.device);
    bwt_hip_block * hl = device_block(l, c, mem);
    bwt_hip_block * hr = device_block(r, c, mem);
    const idx_t m = l.b - l.a, rn = r.b - r.a;
    int64_t * counts = new int64_t[m + 1];
    int64_t * g = new int64_t[rn];
    bwt_hip_ok(bwt_hip_gap(c, hl, hr, opt.build_gap_segment_sz, counts, g));
    for (idx_t i = 0; i <= m; i++)
      if (counts[i]) gap.set_count(l.a + i, counts[i]);
    r.ssa.sort_by_pos(mem);
    for (ssa_iterator it = r.ssa.pos_begin(mem); it.has_next(); it.next()) it.e->rank += g[it.e->pos - r.a];
    /* the device copy of r still holds the ranks as they were: it no longer describes r */
    bwt_hip_block_free(c, hr);
    r.dev = 0;
    delete[] counts;
    delete[] g;
    gap.set_prefix_sum(mem, opt.prefix_sum_gran);
    stat.end(ts_event_bwt_build_gap);
    return 1;
  }

  /* merges the BWTs of T[a:c) and T[c:b) into the BWT of T[a:b); l and r are finished */
  inline bwt bwt_merge(bwt & l, bwt & r, alpha_t * w, mallocator & mem, bwt_opt & opt) {
    (void)w;
    stat.start(ts_event_bwt_merge_bwt);
    assert(l.L == r.L);
    assert(l.b == r.a);
    bwt_hip_ctx * c = device_session::get().with_text(l.T, l.n, opt.device);
    bwt_hip_block * hl = device_block(l, c, mem);
    bwt_hip_block * hr = device_block(r, c, mem);
    bwt_hip_block * h = 0;
    bwt_hip_ok(bwt_hip_merge(c, hl, hr, opt.ssa_n_samples, opt.build_gap_segment_sz, &h));
    l.dev = r.dev = 0; /* consumed */
    bwt m;
    m.init(l.T, l.n, l.L, l.a, r.b);
    fetch_block(m, c, h, mem, opt);
    l.fini(mem, opt);
    r.fini(mem, opt);
    stat.end(ts_event_bwt_merge_bwt);
    return m;
  }

  /* BWT of T[a:b) by balanced recursion: ranges up to bwt_rec_threshold are
     leaves, longer ones split at (a+b)/2.  The whole recursion runs on the
     device; only the final block comes back. */
  inline bwt bwt_rec(const alpha_t * T, idx_t n, idx_t a, idx_t b, alpha_t * L, alpha_t * W,
                     mallocator & mem, bwt_opt & opt) {
    (void)W;
    stat.start(ts_event_bwt_range_rec);
    bwt_hip_ctx * c = device_session::get().with_text(T, n, opt.device, !device_session::get().text_is_stable);
    bwt_hip_block * h = 0;
    bwt_hip_ok(bwt_hip_rec(c, a, b, opt.bwt_rec_threshold, opt.ssa_n_samples, opt.build_gap_segment_sz, &h));
    bwt t;
    t.init(T, n, L, a, b);
    fetch_block(t, c, h, mem, opt);
    stat.end(ts_event_bwt_range_rec);
    return t;
  }

  /* credits the device's phase timers to the reference's event names; a phase the table does not know goes to
     the event of its family (leaf_*, gap_*, merge_*), so that no device time is dropped from the table */
  inline void credit_device_timers(bwt_hip_ctx * c, double clocks_per_ms) {
    struct map_t { const char * phase; ts_event_kind_t ev; };
    static const map_t table[] = {
      { "leaf_radix_sort", ts_event_bwt_leaf }, { "leaf_group", ts_event_bwt_leaf },
      { "leaf_text_round", ts_event_bwt_leaf }, { "leaf_doubling_round", ts_event_bwt_leaf },
      /* the passes that turn the sorted window into L, s_rank/e_rank and the samples (sa_to_bwt + sample_sa) */
      { "leaf_direct_emit", ts_event_bwt_sample_sa }, { "leaf_emit", ts_event_bwt_sample_sa },
      { "count_alphabets", ts_event_bwt_count_alphabets },
      { "wavelet_build", ts_event_bwt_build_wavelet_matrix },
      { "gap_chains", ts_event_bwt_build_gap }, { "gap_histogram", ts_event_gap_array_set_prefix_sum },
      { "merge_scatter", ts_event_bwt_merge_loop }, { "merge_resample", ts_event_bwt_resample_sa },
    };
    static const map_t family[] = {
      { "leaf_", ts_event_bwt_leaf }, { "gap_", ts_event_bwt_build_gap }, { "merge_", ts_event_bwt_merge_bwt },
    };
    int k = bwt_hip_timer_count(c);
    for (int i = 0; i < k; i++) {
      const char * name; double ms, work; int64_t calls;
      if (bwt_hip_timer_get(c, i, &name, &ms, &work, &calls) != 0 || calls == 0) continue;
      if (strchr(name, ':') || strchr(name, '=')) continue; /* per-kernel timers and counters lie inside a phase */
      bool done = false;
      for (size_t j = 0; j < sizeof(table) / sizeof(table[0]) && !done; j++)
        if (strcmp(table[j].phase, name) == 0) { stat.add(table[j].ev, (tsc_t)(ms * clocks_per_ms), (long)calls); done = true; }
      for (size_t j = 0; j < sizeof(family) / sizeof(family[0]) && !done; j++)
        if (strncmp(family[j].phase, name, strlen(family[j].phase)) == 0) { stat.add(family[j].ev, (tsc_t)(ms * clocks_per_ms), (long)calls); done = true; }
    }
  }

  /* BWT of the whole text T[0:n) (T[n-1] = 0 and no other 0) into L[0:n) */
  inline bwt pmbwt(const alpha_t * T, idx_t n, alpha_t * L, int stat_level, mallocator & mem, bwt_opt & opt) {
    stat.reset();
    stat.level = stat_level;
    stat.start(ts_event_pmbwt);
    struct timespec w0, w1;
    clock_gettime(CLOCK_MONOTONIC, &w0);
    tsc_t t0 = get_tsc();

    if (opt.n_gpus > 1) {
      /* One host thread and one context per GPU inside bwt_hip_pmbwt_multi; the blocks live and die there, so the
         result has no device handle (dev = 0) and the per-phase device timers are not collected. */
      int * devs = new int[opt.n_gpus];
      for (int k = 0; k < opt.n_gpus; k++) devs[k] = opt.device + k;
      if (const char * list = getenv("BWT_HIP_DEVICES")) { /* e.g. "0,0,0,0": ranks that share a GPU (tests) */
        int k = 0;
        for (const char * p = list; *p && k < opt.n_gpus; k++) { devs[k] = atoi(p); while (*p && *p != ',') p++; if (*p) p++; }
      }
      bwt t;
      t.init(T, n, L, 0, n);
      idx_t cap = min(n, opt.ssa_n_samples);
      idx_t * rk = new idx_t[cap];
      idx_t * ps = new idx_t[cap];
      int64_t s = -1, e = -1, ns = 0;
      int used = 0;
      bwt_hip_ok(bwt_hip_pmbwt_multi(devs, opt.n_gpus, T, n, opt.bwt_rec_threshold, opt.ssa_n_samples, opt.build_gap_segment_sz,
                                     L, &s, &e, (int64_t *)rk, (int64_t *)ps, cap, &ns, &used));
      t.s_rank = s;
      t.e_rank = e;
      t.ssa.load(cap, rk, ps, ns < cap ? ns : cap, mem);
      delete[] rk;
      delete[] ps;
      delete[] devs;
      tsc_t t1 = get_tsc();
      stat.end(ts_event_pmbwt);
      stat.print();
      printf("%llu clocks to build bwt for %ld chars\n", t1 - t0, n);
      return t;
    }
    /* T goes up in chunks while the first leaf is already being sorted; L comes back with the same call */
    bwt_hip_ctx * c = device_session::get().context(opt.device);
    if (opt.fit_leaves_to_device) {
      int64_t cap = 0;
      bwt_hip_ok(bwt_hip_leaf_capacity(c, n, &cap));
      if (cap > 0) {
        idx_t leaf = n; /* bwt_rec halves ranges: the first n / 2^k that fits */
        while (leaf > cap && leaf > 1) leaf = (leaf + 1) / 2;
        opt.bwt_rec_threshold = leaf;
      }
    }
    if (stat_level >= 1) { bwt_hip_ok(bwt_hip_set_timing(c, 1)); bwt_hip_ok(bwt_hip_reset_timers(c)); }
    bwt_hip_block * h = 0;
    bwt_hip_ok(bwt_hip_pmbwt_host(c, T, n, opt.bwt_rec_threshold, opt.ssa_n_samples, opt.build_gap_segment_sz, L, &h));
    device_session::get().note_text(T, n);
    bwt t;
    t.init(T, n, L, 0, n);
    fetch_block(t, c, h, mem, opt, false);
    int64_t live, peak, reserved;
    bwt_hip_ok(bwt_hip_mem_stats(c, &live, &peak, &reserved));
    mem.note_device((size_t)peak);

    tsc_t t1 = get_tsc();
    clock_gettime(CLOCK_MONOTONIC, &w1);
    stat.end(ts_event_pmbwt);
    if (stat_level >= 1) {
      double ms = (w1.tv_sec - w0.tv_sec) * 1e3 + (w1.tv_nsec - w0.tv_nsec) * 1e-6;
      credit_device_timers(c, ms > 0 ? (t1 - t0) / ms : 0);
      bwt_hip_ok(bwt_hip_set_timing(c, 0));
    }
    stat.print();
    printf("%llu clocks to build bwt for %ld chars\n", t1 - t0, n);
    return t;
  }

  inline int check_equal(alpha_t * T, alpha_t * I, idx_t a, idx_t b) {
    for (idx_t i = a; i < b; i++)
      if (!check(T[i] == I[i])) return 0;
    return 1;
  }

  inline void random_init(alpha_t * L, idx_t n, unsigned short rg[3]) {
    for (idx_t i = 0; i < n; i++) L[i] = (alpha_t)nrand48(rg);
  }

}
