// Exercises the bwt:: header API the way the reference's own programs do
// (src/simplest_example.cc, src/pmbwt.cc:64-81, src/test/test_leaf.cc,
// test_merge.cc, test_ssa.cc), on the GPU.  Prints one "OK name" line per
// check and exits non-zero on the first failure.
#include <stdint.h>

#include "bwt.h"

static bwt::alpha_t * gen(bwt::idx_t n, uint64_t seed, int sigma) {
  bwt::alpha_t * T = new bwt::alpha_t[n];
  uint64_t s = seed;
  for (bwt::idx_t i = 0; i + 1 < n; i++) {
    s += 0x9E3779B97F4A7C15ull;
    uint64_t z = s;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    T[i] = (bwt::alpha_t)(1 + z % sigma);
  }
  T[n - 1] = 0;
  return T;
}

static int fail(const char * what) {
  printf("NG %s\n", what);
  return 1;
}

int main() {
  {  // known answer of the reference's smallest example
    const bwt::alpha_t * T = (const bwt::alpha_t *)"mississippi";
    bwt::idx_t n = 12;
    bwt::alpha_t L[12];
    bwt::bwt_opt opt;
    opt.set_defaults(T, n);
    bwt::mallocator mem(n, opt);
    bwt::bwt t = bwt::pmbwt(T, n, L, 0, mem, opt);
    if (memcmp(L, "ipssm\0pissii", 12) != 0 || t.s_rank != 5 || t.e_rank != 0) return fail("mississippi");
    t.fini(mem, opt);
    printf("OK mississippi\n");
  }
  {  // whole pipeline with the default budget (16 leaves) and the round trip of src/pmbwt.cc:64-81
    bwt::idx_t n = 300000;
    bwt::alpha_t * T = gen(n, 42, 4), * L = new bwt::alpha_t[n], * I = new bwt::alpha_t[n];
    bwt::bwt_opt opt;
    opt.alpha_min = 1;
    opt.alpha_max = 4;
    opt.set_defaults(T, n);
    bwt::mallocator mem(n, opt);
    mem.level = 1;
    bwt::bwt t = bwt::pmbwt(T, n, L, 1, mem, opt);
    t.init_extra(0, mem, opt);
    t.ibwt(I);
    if (!bwt::check_equal(T, I, 0, n)) return fail("pmbwt round trip");
    if (t.sa(t.s_rank) != 0 || t.sa(t.e_rank) != n - 1) return fail("sa of special ranks");
    for (bwt::idx_t r = 0; r < n; r += 9973)
      if (L[r] != (t.sa(r) ? T[t.sa(r) - 1] : 0)) return fail("L[r] = T[sa(r)-1]");
    t.fini(mem, opt);
    if (mem.allocated != mem.deallocated) return fail("host allocations balanced");
    if (mem.device_peak == 0) return fail("device peak recorded");
    printf("OK pmbwt default budget\n");
    delete[] T; delete[] L; delete[] I;
  }
  {  // DNA as letters (65, 67, 71, 84: the device holds such a text renumbered, the host never sees the codes):
     // the pipeline's round trip, and a merge of blocks that only exist on the host
    bwt::idx_t n = 120000;
    bwt::alpha_t * T = gen(n, 11, 4), * L = new bwt::alpha_t[n], * I = new bwt::alpha_t[n];
    const bwt::alpha_t letters[5] = {0, 'A', 'C', 'G', 'T'};
    for (bwt::idx_t i = 0; i < n; i++) T[i] = letters[T[i]];
    bwt::bwt_opt opt;
    opt.set_defaults(T, n);
    bwt::mallocator mem(n, opt);
    bwt::bwt t = bwt::pmbwt(T, n, L, 0, mem, opt);
    for (bwt::idx_t r = 0; r < n; r++)
      if (L[r] != 0 && L[r] != 'A' && L[r] != 'C' && L[r] != 'G' && L[r] != 'T') return fail("L holds the text's own symbols");
    t.init_extra(0, mem, opt);
    t.ibwt(I);
    if (!bwt::check_equal(T, I, 0, n)) return fail("letters round trip");
    for (bwt::idx_t r = 0; r < n; r += 4999)
      if (L[r] != (t.sa(r) ? T[t.sa(r) - 1] : 0)) return fail("letters: L[r] = T[sa(r)-1]");
    t.fini(mem, opt);
    bwt::idx_t a = 0, c = 50000;
    bwt::bwt l = bwt::bwt_leaf(T, n, a, c, L, mem, opt);
    bwt::bwt r = bwt::bwt_leaf(T, n, c, n, L, mem, opt);
    bwt_hip_block_free(bwt::device_session::get().ctx, l.dev); l.dev = 0;
    bwt_hip_block_free(bwt::device_session::get().ctx, r.dev); r.dev = 0;
    bwt::bwt m = bwt::bwt_merge(l, r, 0, mem, opt);
    memset(I, 0, n);
    m.init_extra(0, mem, opt);
    m.ibwt(I);
    if (!bwt::check_equal(T, I, 0, n)) return fail("letters: merge of host blocks");
    m.fini(mem, opt);
    printf("OK letters\n");
    delete[] T; delete[] L; delete[] I;
  }
  {  // two leaves of a sub-range and their merge (src/test/test_merge.cc), blocks kept in HBM
    bwt::idx_t n = 50000, a = 1234, b = 43210, c = (a + b) / 2;
    bwt::alpha_t * T = gen(n, 7, 26), * M = new bwt::alpha_t[n], * I = new bwt::alpha_t[n];
    bwt::bwt_opt opt;
    opt.set_defaults();
    bwt::mallocator mem(opt);
    bwt::bwt l = bwt::bwt_leaf(T, n, a, c, M, mem, opt);
    bwt::bwt r = bwt::bwt_leaf(T, n, c, b, M, mem, opt);
    bwt::bwt m = bwt::bwt_merge(l, r, 0, mem, opt);
    m.init_extra(0, mem, opt);
    m.ibwt(I);
    if (!bwt::check_equal(T, I, a, b)) return fail("merge round trip");
    m.fini(mem, opt);
    printf("OK leaf+leaf+merge on device blocks\n");
    // the same with blocks that only exist on the host (device copies dropped): upload path
    bwt::bwt l2 = bwt::bwt_leaf(T, n, a, c, M, mem, opt);
    bwt::bwt r2 = bwt::bwt_leaf(T, n, c, b, M, mem, opt);
    bwt_hip_block_free(bwt::device_session::get().ctx, l2.dev); l2.dev = 0;
    bwt_hip_block_free(bwt::device_session::get().ctx, r2.dev); r2.dev = 0;
    bwt::bwt m2 = bwt::bwt_merge(l2, r2, 0, mem, opt);
    memset(I, 0, n);
    m2.init_extra(0, mem, opt);
    m2.ibwt(I);
    if (!bwt::check_equal(T, I, a, b)) return fail("merge of uploaded blocks");
    m2.fini(mem, opt);
    printf("OK merge of host blocks\n");
    delete[] T; delete[] M; delete[] I;
  }
  {  // sampled SA against the real SA (src/test/test_ssa.cc), recursion on a sub-range (test_rec.cc)
    bwt::idx_t n = 20000, a = 100, b = 19000;
    bwt::alpha_t * T = gen(n, 9, 4), * L = new bwt::alpha_t[n], * I = new bwt::alpha_t[n];
    bwt::idx_t * SA = new bwt::idx_t[n];
    bwt::bwt_opt opt;
    opt.set_defaults();
    opt.bwt_rec_threshold = 1000;
    opt.ssa_n_samples = 100;
    bwt::mallocator mem(opt);
    bwt::sa_range(T, n, a, b, SA, mem);
    bwt::idx_less lt(T, n);
    for (bwt::idx_t r = a; r + 1 < b; r++)
      if (!lt(SA[r], SA[r + 1])) return fail("sa_range sorted");
    bwt::bwt t = bwt::bwt_rec(T, n, a, b, L, 0, mem, opt);
    t.init_extra(0, mem, opt);
    for (bwt::idx_t r = a; r < b; r += 37)
      if (t.sa(r) != SA[r]) return fail("sa(r) == SA[r]");
    t.ibwt(I);
    if (!bwt::check_equal(T, I, a, b)) return fail("rec round trip");
    t.fini(mem, opt);
    printf("OK sa_range, bwt_rec, sampled sa\n");
    delete[] T; delete[] L; delete[] I; delete[] SA;
  }
  {  // the opt-in tree: leaves as long as the device's memory holds (here the whole text: no merge), same L
    bwt::idx_t n = 400000;
    bwt::alpha_t * T = gen(n, 5, 4), * L = new bwt::alpha_t[n], * L2 = new bwt::alpha_t[n];
    bwt::bwt_opt opt;
    opt.alpha_min = 1;
    opt.alpha_max = 4;
    opt.set_defaults(T, n);
    bwt::mallocator mem(n, opt);
    bwt::bwt t = bwt::pmbwt(T, n, L, 0, mem, opt);
    bwt::bwt_opt opt2;
    opt2.alpha_min = 1;
    opt2.alpha_max = 4;
    opt2.fit_leaves_to_device = 1;
    opt2.set_defaults(T, n);
    bwt::mallocator mem2(n, opt2);
    bwt::bwt t2 = bwt::pmbwt(T, n, L2, 0, mem2, opt2);
    if (opt2.bwt_rec_threshold != n || opt.bwt_rec_threshold >= n) return fail("fit_leaves_to_device sizes the leaf");
    if (memcmp(L, L2, n) != 0 || t.s_rank != t2.s_rank) return fail("fit_leaves_to_device: same L");
    if (t2.round_trip_mismatches() != 0) return fail("fit_leaves_to_device: round trip");
    t.fini(mem, opt);
    t2.fini(mem2, opt2);
    printf("OK fit_leaves_to_device\n");
    delete[] T; delete[] L; delete[] L2;
  }
  {  // build_gap and resample_sa called by hand (reference include/bwt.h:730-787, :366-453) against a direct count
     // and against the samples the device merge keeps
    bwt::idx_t n = 9000, a = 300, b = 8700, c = (a + b) / 2;
    bwt::alpha_t * T = gen(n, 21, 4), * M = new bwt::alpha_t[n];
    for (bwt::idx_t i = 4000; i < 4700; i++) T[i] = 3;  // a run across the boundary
    for (bwt::idx_t i = 5000; i < 8000; i += 7) T[i] = 9;  // a symbol the left block lacks: its suffixes share the last gap index (a count above 254)
    bwt::idx_t * SA = new bwt::idx_t[n];
    bwt::bwt_opt opt;
    opt.set_defaults();
    opt.ssa_n_samples = 150;
    opt.build_gap_segment_sz = 64;
    bwt::mallocator mem(opt);
    bwt::bwt l = bwt::bwt_leaf(T, n, a, c, M, mem, opt);
    bwt::bwt r = bwt::bwt_leaf(T, n, c, b, M, mem, opt);
    bwt::gap_array gap;
    gap.init(l.a, l.b, mem, opt.gap_sum_gran);
    if (!bwt::build_gap(l, r, gap, mem, opt)) return fail("build_gap");
    bwt::sa_range(T, n, a, c, SA, mem);
    bwt::idx_less lt(T, n);
    bwt::idx_t * want = new bwt::idx_t[c - a + 1];
    for (bwt::idx_t i = 0; i <= c - a; i++) want[i] = 0;
    for (bwt::idx_t j = c; j < b; j++) {
      bwt::idx_t lo = a, hi = c;  // first left rank whose suffix is above T[j:]
      while (lo < hi) {
        bwt::idx_t mid = (lo + hi) / 2;
        if (lt(SA[mid], j)) lo = mid + 1; else hi = mid;
      }
      want[lo - a]++;
    }
    bwt::idx_t run = 0, big = 0;
    for (bwt::idx_t i = a; i <= c; i++) {
      if (gap.get(i) != want[i - a]) return fail("gap counts");
      run += want[i - a];
      if (gap.sum(i) != run) return fail("gap prefix sums");
      if (want[i - a] > 254) big++;
    }
    if (run != b - c || big == 0) return fail("gap total / overflow path exercised");
    bwt::bwt mh;
    mh.init(T, n, M, a, b);
    if (!mh.resample_sa(l.ssa, r.ssa, c, bwt::min(opt.ssa_n_samples, b - a), gap, mem, opt)) return fail("resample_sa");
    gap.fini(mem);
    // the device merge of the same two leaves
    bwt::bwt l2 = bwt::bwt_leaf(T, n, a, c, M, mem, opt);
    bwt::bwt r2 = bwt::bwt_leaf(T, n, c, b, M, mem, opt);
    bwt::bwt md = bwt::bwt_merge(l2, r2, 0, mem, opt);
    if (mh.s_rank != md.s_rank || mh.e_rank != md.e_rank) return fail("resample_sa special ranks");
    if (mh.ssa.n != md.ssa.n) return fail("resample_sa sample count");
    for (bwt::idx_t k = 0; k < md.ssa.sz; k++)
      if (md.ssa.a[k].rank != -1 && mh.ssa.get(md.ssa.a[k].rank) != md.ssa.a[k].pos) return fail("resample_sa sample set");
    l.fini(mem, opt);
    r.fini(mem, opt);
    mh.fini(mem, opt);
    md.fini(mem, opt);
    printf("OK build_gap and resample_sa by hand\n");
    delete[] T; delete[] M; delete[] SA; delete[] want;
  }
  printf("ALL OK\n");
  return 0;
}
