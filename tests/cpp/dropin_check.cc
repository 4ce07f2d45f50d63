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
  printf("ALL OK\n");
  return 0;
}
