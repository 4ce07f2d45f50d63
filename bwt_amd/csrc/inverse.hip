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

struct SymNext {
  u32 sym;   // L[i]
  u32 next;  // LF_map_mod(L[i], i), relative to the block's first rank
};

// digit at `pos` of a 4-ary level and its occurrences before pos: one 64-byte block
__device__ __forceinline__ void wm_access_digit(const u64 *__restrict__ lvl, u32 pos, u32 *digit, u32 *before) {
  const u32 blk = pos / WM_BLOCK_POS, off = pos - blk * WM_BLOCK_POS;
  const uint4 *p = reinterpret_cast<const uint4 *>(lvl + (u64)blk * WM_BLOCK_WORDS);
  const u32 wi = off >> 6, bi = off & 63;
  const uint4 hdr = p[0];
  uint4 ch[WM_CHUNKS];
#pragma unroll
  for (u32 k = 0; k < WM_CHUNKS; k++) ch[k] = p[1 + k];
  u64 lo[WM_CHUNKS], hi[WM_CHUNKS];
#pragma unroll
  for (u32 k = 0; k < WM_CHUNKS; k++) {
    lo[k] = ((u64)ch[k].y << 32) | ch[k].x;
    hi[k] = ((u64)ch[k].w << 32) | ch[k].z;
  }
  const u64 mylo = wi == 0 ? lo[0] : (wi == 1 ? lo[1] : lo[2]), myhi = wi == 0 ? hi[0] : (wi == 1 ? hi[1] : hi[2]);
  const u32 d = (u32)((mylo >> bi) & 1) | ((u32)((myhi >> bi) & 1) << 1);
  u32 cnt = d == 0 ? hdr.x : (d == 1 ? hdr.y : (d == 2 ? hdr.z : hdr.w));
  const u64 inv0 = (d & 1) ? 0ull : ~0ull, inv1 = (d & 2) ? 0ull : ~0ull;
  const u64 part = (1ull << bi) - 1;
#pragma unroll
  for (u32 k = 0; k < WM_CHUNKS; k++) {
    const u64 msk = (k < wi) ? ~0ull : (k == wi ? part : 0ull);
    cnt += __popcll((lo[k] ^ inv0) & (hi[k] ^ inv1) & msk);
  }
  *digit = d;
  *before = cnt;
}

// the same for a 16-ary level: one 128-byte block
__device__ __forceinline__ void wm_access_digit16(const u64 *__restrict__ lvl, u32 pos, u32 *digit, u32 *before) {
  const u32 blk = pos / WM16_BLOCK_POS, off = pos % WM16_BLOCK_POS;
  const uint4 *p = reinterpret_cast<const uint4 *>(lvl + (u64)blk * WM16_BLOCK_WORDS);
  uint4 pl[4];
#pragma unroll
  for (u32 b = 0; b < 4; b++) pl[b] = p[4 + b];
  u32 d = 0;
#pragma unroll
  for (u32 b = 0; b < 4; b++) {
    const u64 word = off < 64 ? (((u64)pl[b].y << 32) | pl[b].x) : (((u64)pl[b].w << 32) | pl[b].z);
    d |= (u32)((word >> (off & 63)) & 1) << b;
  }
  const uint4 h = p[d >> 2];
  u32 cnt = (d & 3) == 0 ? h.x : ((d & 3) == 1 ? h.y : ((d & 3) == 2 ? h.z : h.w));
  u64 m0 = ~0ull, m1 = ~0ull;
#pragma unroll
  for (u32 b = 0; b < 4; b++) {
    const u64 inv = ((d >> b) & 1) ? 0ull : ~0ull;
    m0 &= (((u64)pl[b].y << 32) | pl[b].x) ^ inv;
    m1 &= (((u64)pl[b].w << 32) | pl[b].z) ^ inv;
  }
  const u64 k0 = off >= 64 ? ~0ull : ((1ull << off) - 1), k1 = off > 64 ? ((1ull << (off - 64)) - 1) : 0ull;
  *digit = d;
  *before = cnt + __popcll(m0 & k0) + __popcll(m1 & k1);
}

// L[i] and LF_map_mod(L[i], i) (include/bwt.h:253-259) for a rank i other than the hidden one
__device__ __forceinline__ SymNext wm_access_lf(const WmView &wm, const u32 *__restrict__ lf_base, u32 i, u32 e_char,
                                                u32 e_rank) {
  u32 q = i, code = 0;
  for (int l = 0; l < wm.levels; l++) {
    q = q < wm.max_pos ? q : wm.max_pos;
    u32 d, before;
    if (wm.dbits == 4) wm_access_digit16(wm.planes + (u64)l * wm.level_words, q, &d, &before);
    else wm_access_digit(wm.planes + (u64)l * wm.level_words, q, &d, &before);
    code = (code << wm.dbits) | d;
    q = wm.start[l][d] + before;
  }
  const u32 c = code + wm.min_sym;
  u32 k = lf_base[c] + q - (u32)(code == 0 && wm.hidden < i);
  k += (u32)(k >= e_rank && c == e_char);
  return SymNext{c, k};
}

struct InverseParams {
  WmView wm;
  const u32 *lf_tab;
  const u32 *samp_rank, *samp_pos;  // ordered by position; the first is the block's first position
  u32 ns;
  u64 a, b;
  u32 e_char, e_rank;
  const u8 *T;  // compared with when `bad` is not null
  u8 *out;      // out[j - a] = decoded symbol of position j; may be null
  unsigned long long *bad;
};

constexpr int INV_THREADS = 256;

// thread k >= 1 decodes the positions [pos_{k-1}, pos_k) from rank_k; thread 0 owns position b-1 (= e_char)
__global__ void __launch_bounds__(INV_THREADS) k_inverse(InverseParams P) {
  __shared__ u32 base_s[256];
  base_s[threadIdx.x] = P.lf_tab[threadIdx.x];
  __syncthreads();
  const u32 k = blockIdx.x * INV_THREADS + threadIdx.x;
  if (k >= P.ns) return;
  u32 wrong = 0;
  if (k == 0) {
    if (P.out) P.out[P.b - 1 - P.a] = (u8)P.e_char;
    if (P.bad) wrong += P.T[P.b - 1] != P.e_char;
  } else {
    u32 r = P.samp_rank[k];
    const u64 hi = P.samp_pos[k], lo = P.samp_pos[k - 1];
    for (u64 j = hi; j-- > lo;) {
      const SymNext s = wm_access_lf(P.wm, base_s, r, P.e_char, P.e_rank);
      if (P.out) P.out[j - P.a] = (u8)s.sym;
      if (P.bad) wrong += P.T[j] != s.sym;
      r = s.next;
    }
    // the walk must arrive at the previous sample's rank: catches a wrong sample even where the symbols agree
    if (P.bad) wrong += r != P.samp_rank[k - 1];
  }
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
  P.T = c.T;
  P.out = out_dev;
  P.bad = mismatches ? bad.get() : nullptr;
  {
    Phase ph(c, "inverse_chains", (double)blk.len() * blk.wm.levels);
    hipLaunchKernelGGL(k_inverse, dim3(ceil_div((u64)blk.n_samp, INV_THREADS)), dim3(INV_THREADS), 0, c.stream, P);
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

void sa_batch(Ctx &c, Block &blk, const u32 *d_rank, u32 *d_out, size_t nq) {
  if (nq == 0) return;
  if (blk.n_samp < 1) fail(BWT_HIP_EINVAL, "sa: the block has no samples");
  build_wavelet(c, blk);
  const size_t ns = (size_t)blk.n_samp, m = (size_t)blk.len();
  DBuf<u32> bits(c, m / 32 + 1);
  HIP_CHECK(hipMemsetAsync(bits.get(), 0, (m / 32 + 1) * sizeof(u32), c.stream));
  hipLaunchKernelGGL(k_mark_samples, dim3(ceil_div(ns, 256)), dim3(256), 0, c.stream, blk.samp_rank.get(), (u32)ns, bits.get());
  HIP_CHECK(hipGetLastError());
  DBuf<u32> k0(c, ns), k1(c, ns), v0(c, ns), v1(c, ns);
  HIP_CHECK(hipMemcpyAsync(k0.get(), blk.samp_rank.get(), ns * 4, hipMemcpyDeviceToDevice, c.stream));
  HIP_CHECK(hipMemcpyAsync(v0.get(), blk.samp_pos.get(), ns * 4, hipMemcpyDeviceToDevice, c.stream));
  u32 *kk[2] = {k0.get(), k1.get()}, *vv[2] = {v0.get(), v1.get()};
  int bits_needed = 1;
  while (bits_needed < 32 && (m >> bits_needed)) bits_needed++;
  const int cur = radix_sort_u32(c, kk, vv, ns, 0, bits_needed);
  hipLaunchKernelGGL(k_sa_batch, dim3(ceil_div(nq, INV_THREADS)), dim3(INV_THREADS), 0, c.stream, make_view(blk),
                     blk.wm.lf_base.get(), bits.get(), kk[cur], vv[cur], (u32)ns, (u32)m, (u32)blk.e_char, (u32)blk.e_rank, d_rank,
                     (u32)nq, d_out);
  HIP_CHECK(hipGetLastError());
  c.sync();  // the sorted samples die with this scope
}

}  // namespace bwtamd
