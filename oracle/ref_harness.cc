/*
 * ref_harness.cc -- extern "C" taps onto the REAL reference, compiled from the
 * headers where they lie under /root/reference/include (never copied here).
 *
 * TEST INFRASTRUCTURE ONLY: builds into oracle/_ref/libbwt_ref.so, which the
 * golden-vector generator, the oracle's own validation tests and the bench's
 * cpu_baseline leg load.  The product never links it.
 *
 * Build: see oracle/Makefile (g++ -std=c++11 -O3 -DNDEBUG
 * -Dparallel_model=parallel_model_none; the TBB and MassiveThreads models need
 * headers this image does not have, so the reference runs serially here).
 */
#include <algorithm>
#include <vector>
#include <string.h>

#include "bwt.h" /* /root/reference/include/bwt.h */

namespace {

struct ref_handle {
  bwt::bwt t;
  bwt::bwt_opt opt;
  bwt::mallocator *mem;
  bool extra;
};

void fill_opt(bwt::bwt_opt &opt, double mem_budget, int alpha_min, int alpha_max,
              bwt::idx_t bwt_rec_threshold, bwt::idx_t ssa_n_samples, bwt::idx_t seg,
              const bwt::alpha_t *T, bwt::idx_t n) {
  opt.mem_budget = mem_budget;
  opt.alpha_min = (bwt::alpha_t)alpha_min;
  opt.alpha_max = (bwt::alpha_t)alpha_max;
  opt.build_gap_segment_sz = seg;
  opt.set_defaults(T, n);
  /* set_defaults always overwrites these two (include/opts.h:278-290), so a
     caller's explicit choice is applied afterwards, as src/test/test_rec.cc does */
  if (bwt_rec_threshold > 0) opt.bwt_rec_threshold = bwt_rec_threshold;
  if (ssa_n_samples > 0) opt.ssa_n_samples = ssa_n_samples;
}

ref_handle *new_handle(const bwt::bwt_opt &opt) {
  ref_handle *h = new ref_handle;
  h->opt = opt;
  h->mem = new bwt::mallocator(h->opt);
  h->extra = false;
  return h;
}

}  // namespace

extern "C" {

/* include/opts.h:214-291 */
void ref_set_defaults(double mem_budget, long n, long *threshold_out, long *ns_out) {
  bwt::bwt_opt opt;
  opt.mem_budget = mem_budget;
  opt.set_defaults(0, n);
  *threshold_out = opt.bwt_rec_threshold;
  *ns_out = opt.ssa_n_samples;
}

/* include/suffix_array.h:34-49 */
void ref_sa_range(const unsigned char *T, long n, long a, long b, long *SA_out) {
  bwt::bwt_opt opt;
  opt.set_defaults(T, n);
  bwt::mallocator mem(opt);
  std::vector<bwt::idx_t> sa((size_t)(b - a));
  bwt::sa_range(T, n, a, b, sa.data() - a, mem);
  for (long i = 0; i < b - a; i++) SA_out[i] = sa[(size_t)i];
}

/* include/bwt.h:615-635 */
void *ref_leaf(const unsigned char *T, long n, long a, long b, unsigned char *L, int alpha_min,
               int alpha_max, long ssa_n_samples, long seg) {
  bwt::bwt_opt opt;
  fill_opt(opt, 0, alpha_min, alpha_max, 0, ssa_n_samples, seg, T, n);
  ref_handle *h = new_handle(opt);
  h->t = bwt::bwt_leaf(T, n, a, b, L, *h->mem, h->opt);
  return h;
}

/* The gap array of include/bwt.h:795-799 for merging l and r, without
   consuming them: gap_out[i - l.a] = gap.get(i) for i in [l.a, l.b]. */
void ref_gap(void *l_, void *r_, long *gap_out) {
  ref_handle *l = (ref_handle *)l_, *r = (ref_handle *)r_;
  bwt::bwt &lt = l->t, &rt = r->t;
  std::vector<bwt::alpha_t> W((size_t)lt.n);
  std::vector<bwt::ssa_entry> keep(rt.ssa.a, rt.ssa.a + rt.ssa.sz);
  lt.init_extra(W.data(), *l->mem, l->opt);
  bwt::gap_array gap;
  gap.init(lt.a, lt.b, *l->mem, l->opt.gap_sum_gran, l->opt.gap_init_gran);
  bwt::build_gap(lt, rt, gap, *l->mem, l->opt);
  for (bwt::idx_t i = lt.a; i <= lt.b; i++) gap_out[i - lt.a] = gap.get(i);
  gap.fini(*l->mem);
  /* undo init_extra and the in-place sample updates of build_gap */
  lt.wm.fini(*l->mem);
  l->mem->delete_(lt.C, l->opt.alpha_max + 1, bwt::mem_reason_count_alphabets);
  lt.C = 0;
  memcpy(rt.ssa.a, keep.data(), sizeof(bwt::ssa_entry) * keep.size());
  rt.ssa.begin = 0;
  rt.ssa.end = rt.ssa.a + rt.ssa.sz;
}

/* include/bwt.h:791-847; consumes l and r */
void *ref_merge(void *l_, void *r_) {
  ref_handle *l = (ref_handle *)l_, *r = (ref_handle *)r_;
  ref_handle *m = new_handle(l->opt);
  std::vector<bwt::alpha_t> W((size_t)l->t.n);
  m->t = bwt::bwt_merge(l->t, r->t, W.data(), *m->mem, m->opt);
  delete l->mem;
  delete l;
  delete r->mem;
  delete r;
  return m;
}

/* include/bwt.h:867-883 */
void *ref_pmbwt(const unsigned char *T, long n, unsigned char *L, double mem_budget, int alpha_min,
                int alpha_max, long bwt_rec_threshold, long ssa_n_samples, long seg) {
  bwt::bwt_opt opt;
  fill_opt(opt, mem_budget, alpha_min, alpha_max, bwt_rec_threshold, ssa_n_samples, seg, T, n);
  ref_handle *h = new_handle(opt);
  h->t = bwt::pmbwt(T, n, L, 0, *h->mem, h->opt);
  return h;
}

long ref_threshold(void *h) { return ((ref_handle *)h)->opt.bwt_rec_threshold; }
long ref_ssa_n_samples(void *h) { return ((ref_handle *)h)->opt.ssa_n_samples; }
long ref_a(void *h) { return ((ref_handle *)h)->t.a; }
long ref_b(void *h) { return ((ref_handle *)h)->t.b; }
long ref_s_rank(void *h) { return ((ref_handle *)h)->t.s_rank; }
long ref_e_rank(void *h) { return ((ref_handle *)h)->t.e_rank; }
int ref_e_char(void *h) { return ((ref_handle *)h)->t.e_char; }
long ref_n_samples(void *h) { return ((ref_handle *)h)->t.ssa.n; }

/* samples ordered by pos; returns the number of filled entries in the table */
long ref_samples(void *h_, long *rank_out, long *pos_out, long cap) {
  ref_handle *h = (ref_handle *)h_;
  std::vector<bwt::ssa_entry> v;
  for (bwt::idx_t i = 0; i < h->t.ssa.sz; i++)
    if (h->t.ssa.a[i].pos != -1) v.push_back(h->t.ssa.a[i]);
  std::sort(v.begin(), v.end(), bwt::ssa_pos_lt());
  for (size_t i = 0; i < v.size() && (long)i < cap; i++) {
    rank_out[i] = v[i].rank;
    pos_out[i] = v[i].pos;
  }
  return (long)v.size();
}

/* include/bwt.h:169-172; W must hold n bytes */
void ref_init_extra(void *h_, unsigned char *W) {
  ref_handle *h = (ref_handle *)h_;
  h->t.init_extra(W, *h->mem, h->opt);
  h->extra = true;
}

void ref_C(void *h_, long *C_out) {
  ref_handle *h = (ref_handle *)h_;
  for (int c = 0; c <= h->opt.alpha_max; c++) C_out[c] = h->t.C[c];
}

long ref_rank(void *h, int c, long i) { /* include/wavelet_matrix.h:260-290 */
  return ((ref_handle *)h)->t.wm.rank((bwt::alpha_t)c, i);
}

long ref_LF(void *h, int c, long i) { /* include/bwt.h:253-259 */
  return ((ref_handle *)h)->t.LF_map_mod((bwt::alpha_t)c, i);
}

long ref_sa(void *h, long r) { return ((ref_handle *)h)->t.sa(r); } /* include/bwt.h:456-481 */

long ref_bin_search(void *h_, long x) { /* include/bwt.h:489-537 */
  ref_handle *h = (ref_handle *)h_;
  return h->t.bin_search(x, h->opt);
}

void ref_ibwt(void *h, unsigned char *I) { ((ref_handle *)h)->t.ibwt(I); } /* include/bwt.h:276-285 */

void ref_free(void *h_) {
  ref_handle *h = (ref_handle *)h_;
  h->t.fini(*h->mem, h->opt);
  delete h->mem;
  delete h;
}

}  // extern "C"
