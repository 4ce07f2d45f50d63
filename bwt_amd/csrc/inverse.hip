// Inverse transform and suffix-array look-ups of a block on the device (reference: include/bwt.h:276-285
// ibwt, :456-481 sa; used by src/pmbwt.cc:64-81 check_result and the reference's round-trip tests).
//
// The reference decodes serially: one LF step per symbol from e_rank down.  Here the block's samples cut
// the walk into independent pieces: sample k says that the suffix at text position pos_k has rank rank_k,
// so the symbols T[pos_{k-1} .. pos_k) come out of the LF walk that starts at rank_k, one thread per
// sample.  A step reads ONE rank block per wavelet level: the block holds the bit planes of L itself, so
// the symbol L[i] is taken from the block that also answers rank(L[i], i).
#include "wavelet.h"

#include "internal.h"

namespace bwtamd {

struct InverseParams {
  WmView wm;
  const u32 *lf_tab;
  const u32 *samp_rank, *samp_pos;  // ordered by position
  u32 ns;
  u64 a, b;
  u32 e_char, e_rank, s_rank;
  const u8 *T;  // compared with when `bad` is not null
  u8 *out;      // out[j - a] = decoded symbol of position j; may be null
  unsigned long long *bad;
};

constexpr int INV_THREADS = 256;

// Thread k in [1, ns) decodes the positions [pos_{k-1}, pos_k) from rank_k and must arrive at rank_{k-1}.  Thread 0
// decodes what lies before the first sample, [a, pos_0), and must arrive at s_rank (the rank of the suffix at a);
// thread ns decodes what lies behind the last one, [pos_{ns-1}, b-1), starting from e_rank (the rank of the suffix at
// b-1), and owns position b-1 itself (= e_char).  Blocks built here have samples at both ends, so both extra walks
// are empty for them; a block brought in from outside need not.
__global__ void __launch_bounds__(INV_THREADS) k_inverse(InverseParams P) {
  __shared__ u32 base_s[256];
  base_s[threadIdx.x] = P.lf_tab[threadIdx.x];
  __syncthreads();
  const u32 k = blockIdx.x * INV_THREADS + threadIdx.x;
  if (k > P.ns) return;
  u32 wrong = 0;
  u32 r, arrive;
  u64 hi, lo;
  if (k == P.ns) {
    if (P.out) P.out[P.b - 1 - P.a] = (u8)P.e_char;
    if (P.bad) wrong += P.T[P.b - 1] != P.e_char;
    r = P.e_rank;
    hi = P.b - 1;
    lo = P.samp_pos[P.ns - 1];
    arrive = P.samp_rank[P.ns - 1];
  } else {
    r = P.samp_rank[k];
    hi = P.samp_pos[k];
    lo = k ? (u64)P.samp_pos[k - 1] : P.a;
    arrive = k ? P.samp_rank[k - 1] : P.s_rank;
  }
  if (hi < lo || hi >= P.b) {  // samples out of order or outside the block
    if (P.bad) atomicAdd(P.bad, 1ull);
    return;
  }
  for (u64 j = hi; j-- > lo;) {
    const SymNext s = wm_access_lf(P.wm, base_s, r, P.e_char, P.e_rank);
    if (P.out) P.out[j - P.a] = (u8)s.sym;
    if (P.bad) wrong += P.T[j] != s.sym;
    r = s.next;
  }
  // the walk must arrive at the previous sample's rank: catches a wrong sample even where the symbols agree
  if (P.bad) wrong += r != arrive;
  if (P.bad && wrong) atomicAdd(P.bad, (unsigned long long)wrong);
}

// Decodes the block: out_dev (b-a bytes, may be null) receives T[a:b) as L encodes it; *mismatches (may be
// null) the number of positions where that differs from the context's text, plus the samples whose rank
// the walk does not reach.
void inverse_block(Ctx &c, Block &blk, u8 *out_dev, i64 *mismatches) {
  if (blk.n_samp < 1) fail(BWT_HIP_EINVAL, "inverse: the block has no samples");
  build_wavelet(c, blk);
  c.wait_text((u64)c.n);
  DBuf<unsigned long long> bad(c, 1);
  HIP_CHECK(hipMemsetAsync(bad.get(), 0, sizeof(unsigned long long), c.stream));
  InverseParams P;
  P.wm = make_view(blk);
  P.lf_tab = blk.wm.lf_base.get();
  P.samp_rank = blk.samp_rank.get();
  P.samp_pos = blk.samp_pos.get();
  P.ns = (u32)blk.n_samp;
  P.a = (u64)blk.a;
  P.b = (u64)blk.b;
  P.e_char = blk.e_char;
  P.e_rank = (u32)blk.e_rank;
  P.s_rank = (u32)blk.s_rank;
  P.T = c.T;
  P.out = out_dev;
  P.bad = mismatches ? bad.get() : nullptr;
  {
    Phase ph(c, "inverse_chains", (double)blk.len() * blk.wm.levels);
    hipLaunchKernelGGL(k_inverse, dim3(ceil_div((u64)blk.n_samp + 1, INV_THREADS)), dim3(INV_THREADS), 0, c.stream, P);
    HIP_CHECK(hipGetLastError());
  }
  if (mismatches) {
    unsigned long long h = 0;
    HIP_CHECK(hipMemcpyAsync(&h, bad.get(), sizeof(h), hipMemcpyDeviceToHost, c.stream));
    c.sync();
    *mismatches = (i64)h;
  }
}

// ----------------------------------------------------------------------------
// SA[r] for a batch of ranks (include/bwt.h:456-481): walk backwards until a sampled rank is met.  The
// reference asks its hash table at every step; here a bitmap over the block's ranks answers "sampled?"
// and a binary search in the samples ordered by rank gives the position.
// ----------------------------------------------------------------------------
__global__ void k_mark_samples(const u32 *__restrict__ rank, u32 ns, u32 *__restrict__ bits) {
  const u32 k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < ns) atomicOr(&bits[rank[k] >> 5], 1u << (rank[k] & 31));
}

__global__ void __launch_bounds__(INV_THREADS)
    k_sa_batch(WmView wm, const u32 *__restrict__ lf_tab, const u32 *__restrict__ bits, const u32 *__restrict__ by_rank,
               const u32 *__restrict__ by_rank_pos, u32 ns, u32 m, u32 e_char, u32 e_rank, const u32 *__restrict__ q,
               u32 nq, u32 *__restrict__ out) {
  __shared__ u32 base_s[256];
  base_s[threadIdx.x] = lf_tab[threadIdx.x];
  __syncthreads();
  const u32 k = blockIdx.x * INV_THREADS + threadIdx.x;
  if (k >= nq) return;
  u32 r = q[k], steps = 0;
  // ends at the latest at the rank of the block's first suffix, which is sampled; the bounds only matter for
  // a block whose L and samples do not belong together
  while (r < m && steps < m && !((bits[r >> 5] >> (r & 31)) & 1)) {
    r = wm_access_lf(wm, base_s, r, e_char, e_rank).next;
    steps++;
  }
  u32 lo = 0, hi = ns;  // by_rank[lo] == r
  while (hi - lo > 1) {
    const u32 mid = (lo + hi) / 2;
    if (by_rank[mid] <= r) lo = mid;
    else hi = mid;
  }
  out[k] = by_rank_pos[lo] + steps;
}

// which ranks of the block are sampled (a bit per rank) and the samples ordered by rank
void build_rank_index(Ctx &c, Block &blk, RankIndex &ix) {
  if (blk.n_samp < 1) fail(BWT_HIP_EINVAL, "sa: the block has no samples");
  const size_t ns = (size_t)blk.n_samp, m = (size_t)blk.len();
  ix.bits = DBuf<u32>(c, m / 32 + 1);
  HIP_CHECK(hipMemsetAsync(ix.bits.get(), 0, (m / 32 + 1) * sizeof(u32), c.stream));
  hipLaunchKernelGGL(k_mark_samples, dim3(ceil_div(ns, 256)), dim3(256), 0, c.stream, blk.samp_rank.get(), (u32)ns, ix.bits.get());
  HIP_CHECK(hipGetLastError());
  DBuf<u32> k0(c, ns), k1(c, ns), v0(c, ns), v1(c, ns);
  HIP_CHECK(hipMemcpyAsync(k0.get(), blk.samp_rank.get(), ns * 4, hipMemcpyDeviceToDevice, c.stream));
  HIP_CHECK(hipMemcpyAsync(v0.get(), blk.samp_pos.get(), ns * 4, hipMemcpyDeviceToDevice, c.stream));
  u32 *kk[2] = {k0.get(), k1.get()}, *vv[2] = {v0.get(), v1.get()};
  int bits_needed = 1;
  while (bits_needed < 32 && (m >> bits_needed)) bits_needed++;
  const int cur = radix_sort_u32(c, kk, vv, ns, 0, bits_needed);
  ix.by_rank = std::move(cur ? k1 : k0);
  ix.by_rank_pos = std::move(cur ? v1 : v0);
  ix.ns = (u32)ns;
}

void sa_batch(Ctx &c, Block &blk, const u32 *d_rank, u32 *d_out, size_t nq) {
  if (nq == 0) return;
  build_wavelet(c, blk);
  RankIndex ix;
  build_rank_index(c, blk, ix);
  hipLaunchKernelGGL(k_sa_batch, dim3(ceil_div(nq, INV_THREADS)), dim3(INV_THREADS), 0, c.stream, make_view(blk),
                     blk.wm.lf_base.get(), ix.bits.get(), ix.by_rank.get(), ix.by_rank_pos.get(), ix.ns, (u32)blk.len(),
                     (u32)blk.e_char, (u32)blk.e_rank, d_rank, (u32)nq, d_out);
  HIP_CHECK(hipGetLastError());
  c.sync();  // the index dies with this scope
}

}  // namespace bwtamd
