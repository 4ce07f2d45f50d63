// Gap array of a merge (reference: include/bwt.h:730-787 build_gap,
// include/gap_array.h): for every position j of the right block,
//   g_j = #{ i in [l.a, l.b) : T[i:] < T[j:] }            (relative to l.a)
// and S[i] = #{ j : g_j <= i } = gap[0] + ... + gap[i], i in [0, m].
//
// One thread per chain of `seg` consecutive right positions, walked backwards
// with the modified LF mapping (include/bwt.h:262-273): one 64-byte rank block
// per wavelet level and step.  A chain does not know the rank of its first
// suffix; it starts `warm` positions further right with the whole left block as
// the candidate interval [lo, hi] and LF-maps both ends: the interval shrinks
// to a point after about log_sigma(m) steps (a backward search for the text that
// follows).  If it has not collapsed when the chain's own positions begin, the
// chain restarts from four times as far.
//
// g values are written step-major ([step][chain]) so that a wavefront's 64
// chains store 64 consecutive words.  The histogram is atomic-free: radix sort
// of g, run ends scattered into S, inclusive max-scan.
#include "wavelet.h"

#include "internal.h"
#include "scan_tmpl.h"

namespace bwtamd {

struct ChainParams {
  const u8 *T;
  u64 n;
  u64 ra, rb;  // right block
  u64 lb;      // end of the left block (== ra)
  u32 m;       // left block length
  u32 seg, n_chains;
  u32 warm;
  u32 e_char, e_rank;  // of the left block; e_rank relative to l.a
  WmView wm;
  const u32 *lf_tab;  // lf_base[256], p_end[256]
  u32 *g;             // seg * n_chains
};

// T[x:] < T[y:] for x != y, both < n (the unique 0 at n-1 ends the loop)
__device__ __forceinline__ bool suffix_less(const u8 *__restrict__ T, u64 x, u64 y) {
  for (;; x++, y++) {
    u32 cx = T[x], cy = T[y];
    if (cx != cy) return cx < cy;
  }
}

constexpr int CHAIN_THREADS = 256;

__global__ void __launch_bounds__(CHAIN_THREADS) k_chains(ChainParams P) {
  __shared__ u32 base_s[256];
  base_s[threadIdx.x] = P.lf_tab[threadIdx.x];
  __syncthreads();
  const u32 k = blockIdx.x * CHAIN_THREADS + threadIdx.x;
  if (k >= P.n_chains) return;
  const u8 *__restrict__ T = P.T;
  const u64 begin = P.ra + (u64)k * P.seg;
  const u64 end = (begin + P.seg < P.rb) ? begin + P.seg : P.rb;
  const u64 j0 = end - 1;
  const u32 e_char = P.e_char, e_rank = P.e_rank;

  // interval phase: positions E-1 .. j0
  u32 lo, hi;
  u64 ell = P.warm;
  for (;;) {
    u64 E = j0 + 1 + ell;
    if (E > P.n) E = P.n;
    lo = 0;
    hi = P.m;
    for (u64 j = E; j-- > j0;) {
      u32 c = T[j];
      if (c == 0) {  // the sentinel suffix is smaller than every left suffix
        lo = hi = 0;
        continue;
      }
      u32 klo = wm_lf(P.wm, base_s, c, lo);
      u32 khi = (hi == lo) ? klo : wm_lf(P.wm, base_s, c, hi);
      if (c == e_char && khi >= e_rank) {
        if (suffix_less(T, P.lb - 1, j)) {
          klo += (klo >= e_rank);
          khi += 1;
        }
      }
      lo = klo;
      hi = khi;
    }
    if (lo == hi || E == P.n) break;
    ell *= 4;
  }
  u32 g = lo;
  u32 *__restrict__ gout = P.g + k;
  gout[0] = g;
  u32 step = 1;
  for (u64 j = j0; j-- > begin; step++) {
    u32 c = T[j];
    u32 kk = wm_lf(P.wm, base_s, c, g);
    if (c == e_char && kk >= e_rank) {
      if (suffix_less(T, P.lb - 1, j)) kk++;
    }
    g = kk;
    gout[(u64)step * P.n_chains] = g;
  }
  // the last chain may be short: fill its tail with a value above every valid g
  for (; step < P.seg; step++) gout[(u64)step * P.n_chains] = P.m + 1;
}

// g of the right block's samples, read before the sort below destroys the layout
__global__ void k_sample_g(const u32 *__restrict__ g, const u32 *__restrict__ spos, u32 ns, u64 ra, u64 rb,
                           u32 seg, u32 n_chains, u32 *__restrict__ out) {
  u32 k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= ns) return;
  u64 j = spos[k];
  u32 c = (u32)((j - ra) / seg);
  u64 end = ra + (u64)(c + 1) * seg;
  if (end > rb) end = rb;
  out[k] = g[(u64)(end - 1 - j) * n_chains + c];
}

// g in text order (test tap): out[j - ra] = g_j
__global__ void k_g_text_order(const u32 *__restrict__ g, u64 ra, u64 rb, u32 seg, u32 n_chains,
                               u32 *__restrict__ out) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rb - ra) return;
  u32 c = (u32)(i / seg);
  u64 end = (u64)(c + 1) * seg;
  if (end > rb - ra) end = rb - ra;
  out[i] = g[(end - 1 - i) * n_chains + c];
}

// sorted g -> S[v] = 1 + index of the last occurrence of v
__global__ void k_run_ends(const u32 *__restrict__ s, u32 r, u32 *__restrict__ S) {
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= r) return;
  u32 v = s[i];
  if (i + 1 == r || s[i + 1] != v) S[v] = i + 1;
}

static int bits_for(u64 x) {
  int b = 1;
  while (b < 64 && (x >> b)) b++;
  return b;
}

GapResult build_gap(Ctx &c, Block &l, Block &r, i64 seg_req, bool want_g) {
  if (l.b != r.a) fail(BWT_HIP_EINVAL, "blocks are not adjacent: [%ld,%ld) and [%ld,%ld)", (long)l.a, (long)l.b,
                       (long)r.a, (long)r.b);
  build_wavelet(c, l);
  const u64 m = (u64)l.len(), rn = (u64)r.len();
  if (m + rn >= (1ull << 32) - 2) fail(BWT_HIP_EINVAL, "merged block of %lu symbols exceeds 2^32-3", (unsigned long)(m + rn));
  GapResult G;
  u64 seg = seg_req > 0 ? (u64)seg_req : 1024;
  if (seg > rn) seg = rn;
  G.seg = (i64)seg;
  G.n_chains = (i64)((rn + seg - 1) / seg);
  const u64 total = seg * (u64)G.n_chains;
  DBuf<u32> g0(c, total), g1(c, total);
  ChainParams P;
  P.T = c.T;
  P.n = (u64)c.n;
  P.ra = (u64)r.a;
  P.rb = (u64)r.b;
  P.lb = (u64)l.b;
  P.m = (u32)m;
  P.seg = (u32)seg;
  P.n_chains = (u32)G.n_chains;
  // expected collapse after log_sigma(m) steps; sigma from the symbol width
  {
    double sig = (double)(c.max_sym < 2 ? 2 : c.max_sym);
    int need = 1;
    double x = 1;
    while (x < (double)m && need < 64) {
      x *= sig;
      need++;
    }
    P.warm = (u32)(need + 8);
  }
  P.e_char = l.e_char;
  P.e_rank = (u32)l.e_rank;
  P.wm = make_view(l);
  P.lf_tab = l.wm.lf_base.get();
  P.g = g0.get();
  {
    Phase ph(c, "gap_chains", (double)rn * (l.wm.levels + 1));  // random line transactions
    hipLaunchKernelGGL(k_chains, dim3(ceil_div(G.n_chains, CHAIN_THREADS)), dim3(CHAIN_THREADS), 0, c.stream, P);
    HIP_CHECK(hipGetLastError());
  }
  G.samp_g = DBuf<u32>(c, (size_t)(r.n_samp > 0 ? r.n_samp : 1));
  if (r.n_samp > 0) {
    hipLaunchKernelGGL(k_sample_g, dim3(ceil_div(r.n_samp, 256)), dim3(256), 0, c.stream, g0.get(), r.samp_pos.get(),
                       (u32)r.n_samp, P.ra, P.rb, P.seg, P.n_chains, G.samp_g.get());
    HIP_CHECK(hipGetLastError());
  }
  if (want_g) {
    G.g_text = DBuf<u32>(c, rn);
    hipLaunchKernelGGL(k_g_text_order, dim3(ceil_div(rn, 256)), dim3(256), 0, c.stream, g0.get(), P.ra, P.rb, P.seg,
                       P.n_chains, G.g_text.get());
    HIP_CHECK(hipGetLastError());
  }
  {
    int bits = bits_for(m + 1);
    Phase ph(c, "gap_histogram", (double)total * ((bits + 7) / 8) * 12 + (double)(m + 1) * 12);
    u32 *keys[2] = {g0.get(), g1.get()};
    u32 *vals[2] = {nullptr, nullptr};
    c.sort_tag = "gap:";
    int cur = radix_sort_u32(c, keys, vals, total, 0, bits, 2);
    c.sort_tag = "";
    G.S = DBuf<u32>(c, m + 1);
    HIP_CHECK(hipMemsetAsync(G.S.get(), 0, (m + 1) * 4, c.stream));
    hipLaunchKernelGGL(k_run_ends, dim3(ceil_div(rn, 256)), dim3(256), 0, c.stream, keys[cur], (u32)rn, G.S.get());
    HIP_CHECK(hipGetLastError());
    scan_inclusive_max_u32(c, G.S.get(), G.S.get(), m + 1);
  }
  return G;
}

}  // namespace bwtamd
