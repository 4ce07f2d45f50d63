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
// chain restarts from four times as far, once.  On texts with long repeats that
// is not enough (the interval keeps every left suffix that shares the following
// hundreds or thousands of symbols): such chains are put on a list and a second
// launch finds their heads the reference's way (include/bwt.h:489-537 bin_search
// over :456-481 sa), narrowed to the interval the backward search has left --
// a binary search over its ranks, SA[rank] from the left block's samples, one
// suffix comparison each that starts behind the symbols the interval shares.
//
// g values are written step-major ([step][chain]) so that a wavefront's 64
// chains store 64 consecutive words.  The histogram uses no global atomics: g is
// radix-sorted on its high bits into buckets, each bucket is counted in LDS.
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
  // chains whose head interval has not collapsed after the second try (null: keep restarting from further right)
  u32 *defer_list, *defer_count;
  // second launch: SA[rank] of the left block (samples as inverse.hip's RankIndex holds them)
  const u32 *samp_bits, *by_rank, *by_rank_pos;
  u32 ns;
  // long stretches of a short period in T (runs of one symbol: period 1), by position: n_runs starts, ends (exclusive)
  // and periods; suffix_less steps over them
  const u32 *runs;
  u32 n_runs;
  // ranks of the text's difference-cover sample, where a leaf has made them (rank == null otherwise): comparisons then
  // read at most v symbols
  DcView dc;
};

// T[x:] < T[y:] for x != y, both < n (the unique 0 at n-1 ends the loop).  Eight symbols per (unaligned) load
// while both suffixes have eight left: on repetitive texts these comparisons run for thousands of symbols.
// Runs of one symbol (the N of an assembly's gaps, millions long) and other stretches of a short period would make
// them run for the length of the stretch, once per position of it: after every 256 equal symbols the comparison looks
// both positions up in the list of long periodic stretches and, where both lie inside one of the same period, steps to
// the end of the shorter remainder (the symbols behind both positions are equal and both stretches continue them by
// the same rule, so everything stepped over is equal).
__device__ __forceinline__ bool suffix_less(const u8 *__restrict__ T, u64 n, u64 x, u64 y, const u32 *__restrict__ runs,
                                            u32 n_runs) {
  u32 since = 0;
  while (x + 8 <= n && y + 8 <= n) {
    u64 wx, wy;
    __builtin_memcpy(&wx, T + x, 8);
    __builtin_memcpy(&wy, T + y, 8);
    if (wx != wy) {
      const int at = __builtin_ctzll(wx ^ wy) & ~7;  // the first symbol is the lowest byte
      return ((wx >> at) & 0xFFu) < ((wy >> at) & 0xFFu);
    }
    x += 8;
    y += 8;
    if (n_runs && ++since >= 32) {
      since = 0;
      u64 ex = 0, ey = 0;
      const u32 px = run_around(runs, n_runs, x, &ex);
      const u32 py = px ? run_around(runs, n_runs, y, &ey) : 0;
      if (px && px == py) {
        const u64 skip = ex - x < ey - y ? ex - x : ey - y;
        x += skip;
        y += skip;
      }
    }
  }
  for (;; x++, y++) {
    u32 cx = T[x], cy = T[y];
    if (cx != cy) return cx < cy;
  }
}

// The same with the ranks of a difference-cover sample: the one shift d < v that makes x + d and y + d sample positions
// comes from a table; the suffixes compare on their first d symbols and then on the two ranks.
__device__ __forceinline__ bool suffix_less_dc(const u8 *__restrict__ T, u64 n, u64 x, u64 y, const DcView &dc) {
  const u32 bx = (u32)x / dc.v, rx = (u32)x - bx * dc.v;
  const u32 by = (u32)y / dc.v, ry = (u32)y - by * dc.v;
  const u32 d = dc.pair_tab[rx * dc.v + ry] & 31u;
  u32 k = 0;
  while (k < d) {
    if (x + k + 8 <= n && y + k + 8 <= n) {
      u64 wx, wy;
      __builtin_memcpy(&wx, T + x + k, 8);
      __builtin_memcpy(&wy, T + y + k, 8);
      if (wx != wy) {
        const u32 at = (u32)__builtin_ctzll(wx ^ wy) >> 3;
        if (k + at >= d) break;  // the first difference lies behind the shift: the ranks decide
        return ((wx >> (8 * at)) & 0xFFu) < ((wy >> (8 * at)) & 0xFFu);
      }
      k += 8;
    } else {  // near the end of the text: the unique terminator ends this at the latest
      const u32 cx = T[x + k], cy = T[y + k];
      if (cx != cy) return cx < cy;
      k++;
    }
  }
  // d equal symbols, none of them the terminator: both shifted positions are below n
  u32 sx = rx + d, sy = ry + d, kx = bx, ky = by;
  if (sx >= dc.v) {
    sx -= dc.v;
    kx++;
  }
  if (sy >= dc.v) {
    sy -= dc.v;
    ky++;
  }
  return dc.rank[kx * dc.nd + dc.lo_tab[sx]] < dc.rank[ky * dc.nd + dc.lo_tab[sy]];
}

constexpr int CHAIN_THREADS = 256;

__device__ __forceinline__ bool chain_suffix_less(const ChainParams &P, u64 x, u64 y) {
  return P.dc.rank ? suffix_less_dc(P.T, P.n, x, y, P.dc) : suffix_less(P.T, P.n, x, y, P.runs, P.n_runs);
}

// Backward sequential reader of the text.  A chain's text positions are 1024 bytes away from its
// neighbours', so a byte load per step would touch a different cache line in every lane and find it
// evicted 16 steps later; instead each lane keeps the aligned 16-byte block around its position in
// registers.  (An aligned 16-byte block never straddles a page, so reading the whole block of a
// valid byte is safe at the ends of the text.)
struct TextBack {
  const u8 *T;
  uintptr_t held;
  uint4 blk;
  __device__ __forceinline__ TextBack() : T(nullptr), held(~(uintptr_t)0), blk(make_uint4(0, 0, 0, 0)) {}
  __device__ __forceinline__ explicit TextBack(const u8 *t) : T(t), held(~(uintptr_t)0), blk(make_uint4(0, 0, 0, 0)) {}
  __device__ __forceinline__ u32 operator()(u64 j) {
    const uintptr_t ad = reinterpret_cast<uintptr_t>(T + j);
    const uintptr_t b = ad & ~(uintptr_t)15;
    if (b != held) {
      blk = *reinterpret_cast<const uint4 *>(b);
      held = b;
    }
    const u32 k = (u32)(ad & 15), ws = k >> 2;
    const u32 wd = ws == 0 ? blk.x : (ws == 1 ? blk.y : (ws == 2 ? blk.z : blk.w));
    return (wd >> ((k & 3) * 8)) & 255u;
  }
};

// Backward search for the text behind the chain's last position j0: the interval [lo, hi] of left ranks after the
// positions E-1 .. j0, E = j0 + 1 + ell (cut at n).  Returns E - j0, the number of symbols the left suffixes of the
// ranks lo .. hi-1 share with T[j0:].
__device__ __forceinline__ u64 chain_interval(const ChainParams &P, const u32 *base_s, TextBack &text, u64 j0, u64 ell,
                                              u32 *lo_out, u32 *hi_out) {
  const u32 e_char = P.e_char, e_rank = P.e_rank;
  u64 E = j0 + 1 + ell;
  if (E > P.n) E = P.n;
  u32 lo = 0, hi = P.m;
  for (u64 j = E; j-- > j0;) {
    u32 c = text(j);
    if (c == 0) {  // the sentinel suffix is smaller than every left suffix
      lo = hi = 0;
      continue;
    }
    u32 klo = wm_lf(P.wm, base_s, c, lo);
    u32 khi = (hi == lo) ? klo : wm_lf(P.wm, base_s, c, hi);
    if (c == e_char && khi >= e_rank) {
      if (chain_suffix_less(P, P.lb - 1, j)) {
        klo += (klo >= e_rank);
        khi += 1;
      }
    }
    lo = klo;
    hi = khi;
  }
  *lo_out = lo;
  *hi_out = hi;
  return E - j0;
}

// rank of the suffix at the chain's last position j0 among the left suffixes.  False: the interval has not
// collapsed after the second try and the chain is to be deferred (only when P.defer_list is set).
__device__ __forceinline__ bool chain_head(const ChainParams &P, const u32 *base_s, TextBack &text, u64 j0, u32 *g) {
  u32 lo, hi;
  u64 ell = P.warm;
  for (int attempt = 0;; attempt++) {
    const u64 len = chain_interval(P, base_s, text, j0, ell, &lo, &hi);
    if (lo == hi || j0 + len == P.n) break;
    if (attempt >= 1 && P.defer_list) return false;
    ell *= 4;
  }
  *g = lo;
  return true;
}

// text position of the left suffix of rank r (include/bwt.h:456-481): LF steps to the nearest sampled rank
__device__ __forceinline__ u64 left_sa(const ChainParams &P, const u32 *base_s, u32 r) {
  u32 steps = 0;
  while (steps < P.m && !((P.samp_bits[r >> 5] >> (r & 31)) & 1)) {
    r = wm_access_lf(P.wm, base_s, r, P.e_char, P.e_rank).next;
    steps++;
  }
  u32 lo = 0, hi = P.ns;  // by_rank[lo] == r
  while (hi - lo > 1) {
    const u32 mid = (lo + hi) / 2;
    if (P.by_rank[mid] <= r) lo = mid;
    else hi = mid;
  }
  return (u64)P.by_rank_pos[lo] + steps;
}

// the chain's positions below its last one: LF steps from the head's rank g
__device__ __forceinline__ void chain_walk(const ChainParams &P, const u32 *base_s, TextBack &text, u32 chain, u64 begin,
                                           u64 j0, u32 g) {
  const u32 e_char = P.e_char, e_rank = P.e_rank;
  P.g[chain] = g;
  for (u32 step = 1; step < P.seg; step++) {
    const bool act = begin + step <= j0;  // false only in the tail of the short last chain
    const u32 sym = text(act ? j0 - step : begin);
    u32 kk = wm_lf(P.wm, base_s, sym ? sym : P.wm.min_sym, g);
    if (act) {
      // kk counts the left suffixes below T[j:] other than the block's last one, T[lb-1:], whose rank is e_rank: more
      // than e_rank of them means that one is below T[j:] too; exactly e_rank leaves it to a comparison
      if (sym == e_char && kk >= e_rank) {
        if (kk > e_rank || chain_suffix_less(P, P.lb - 1, j0 - step)) kk++;
      }
      g = kk;
    }
    // the last chain may be short: its tail gets a value above every valid g
    __builtin_nontemporal_store(act ? g : P.m + 1, &P.g[(u64)step * P.n_chains + chain]);
  }
}

// One chain per thread.  (Two chains per thread walked in lockstep -- twice the random reads in flight per wave --
// take the same 10.4 ms on a 5e8-position merge: with every wave slot of the chip occupied the memory side is
// saturated at about 5.1e10 random 64-byte reads per second.)
__global__ void __launch_bounds__(CHAIN_THREADS) k_chains(ChainParams P) {
  __shared__ u32 base_s[256];
  base_s[threadIdx.x] = P.lf_tab[threadIdx.x];
  __syncthreads();
  const u32 chain = blockIdx.x * CHAIN_THREADS + threadIdx.x;
  if (chain >= P.n_chains) return;
  const u64 begin = P.ra + (u64)chain * P.seg;
  const u64 end = (begin + P.seg < P.rb) ? begin + P.seg : P.rb;
  const u64 j0 = end - 1;
  TextBack text(P.T);
  u32 g;
  if (!chain_head(P, base_s, text, j0, &g)) {
    P.defer_list[atomicAdd(P.defer_count, 1u)] = chain;
    return;
  }
  chain_walk(P, base_s, text, chain, begin, j0, g);
}

// the deferred chains: head by binary search inside the interval a short backward search leaves
__global__ void __launch_bounds__(CHAIN_THREADS) k_chains_deferred(ChainParams P, u32 count) {
  __shared__ u32 base_s[256];
  base_s[threadIdx.x] = P.lf_tab[threadIdx.x];
  __syncthreads();
  const u32 t = blockIdx.x * CHAIN_THREADS + threadIdx.x;
  if (t >= count) return;
  const u32 chain = P.defer_list ? P.defer_list[t] : t;  // no list: every chain of the merge
  const u64 begin = P.ra + (u64)chain * P.seg;
  const u64 end = (begin + P.seg < P.rb) ? begin + P.seg : P.rb;
  const u64 j0 = end - 1;
  TextBack text(P.T);
  u32 lo, hi;
  const u64 len = chain_interval(P, base_s, text, j0, (u64)P.warm * 4, &lo, &hi);
  // the left suffixes of ranks lo .. hi-1 begin with T[j0 : j0+len); those below T[j0:] come first
  while (lo < hi) {
    const u32 mid = lo + (hi - lo) / 2;
    const u64 p = left_sa(P, base_s, mid);
    if (chain_suffix_less(P, p + len, j0 + len)) lo = mid + 1;
    else hi = mid;
  }
  chain_walk(P, base_s, text, chain, begin, j0, lo);
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

// g in text order: out[j - ra] = g_j.  The chains wrote g step-major ([step][chain]: a wavefront's 64 chains store 64
// consecutive words); text order is chain-major with the steps reversed, so this is a transpose, done through LDS in
// tiles of 64 chains x 64 steps: rows of the tile are read along the chains, columns written along the text.
constexpr int GT_TILE = 64;
__global__ void __launch_bounds__(256) k_g_text_order(const u32 *__restrict__ g, u64 rn, u32 seg, u32 n_chains, u32 tiles_c,
                                                      u32 *__restrict__ out) {
  __shared__ u32 tile[GT_TILE][GT_TILE + 1];
  const u32 c0 = (blockIdx.x % tiles_c) * GT_TILE, s0 = (blockIdx.x / tiles_c) * GT_TILE;
  const u32 tx = threadIdx.x & (GT_TILE - 1), ty = threadIdx.x / GT_TILE;  // 64 x 4
  for (u32 s = ty; s < GT_TILE; s += 4) {
    const u32 step = s0 + s, c = c0 + tx;
    if (step < seg && c < n_chains) tile[s][tx] = g[(u64)step * n_chains + c];
  }
  __syncthreads();
  for (u32 cc = ty; cc < GT_TILE; cc += 4) {
    const u32 c = c0 + cc;
    if (c >= n_chains) break;
    const u64 begin = (u64)c * seg;
    const u64 end = begin + seg < rn ? begin + seg : rn;  // the last chain may be short
    // position end - 1 - step holds the value of `step`; lanes run along the text: step = s0 + 63 - tx downwards
    const u32 step = s0 + (GT_TILE - 1 - tx);
    if (step < seg && begin + step < end) out[end - 1 - step] = tile[GT_TILE - 1 - tx][cc];
  }
}

// ----------------------------------------------------------------------------
// histogram + prefix sums of g without global atomics.
//
// g is radix-sorted on its HIGH bits only (two 8-bit passes for up to 2^29 left
// suffixes), which groups it into buckets of 2^low consecutive gap indices
// (low <= 13).  One workgroup per bucket finds its slice [lo, hi) of the
// partially sorted array, counts it into 2^low LDS counters private to the
// workgroup, scans them, and writes S[v] = lo + (inclusive count up to v): every
// g below the bucket lies before `lo` in the array, so no global scan is needed.
// ----------------------------------------------------------------------------
constexpr int GH_THREADS = 256;
constexpr int GH_LOW_MAX = 13;  // 8192 counters = 32 KB of LDS

// number of entries of keys[0:n) whose (key >> low) is < bucket; all 64 lanes cooperate
__device__ __forceinline__ u32 wave_bucket_start(const u32 *__restrict__ keys, u32 n, int low, u32 bucket) {
  u32 lo = 0, hi = n;
  const u32 lane = lane_id();
  for (;;) {
    u32 span = hi - lo;
    if (span <= 64) {
      bool t = lane < span && (keys[lo + lane] >> low) < bucket;
      return lo + (u32)__popcll(__ballot(t));
    }
    u32 step = span / 64;
    u32 i = lo + (lane + 1) * step - 1;
    bool t = (keys[i] >> low) < bucket;
    u32 cnt = (u32)__popcll(__ballot(t));
    u32 nlo = lo + cnt * step;
    if (cnt < 64) hi = lo + (cnt + 1) * step - 1;
    lo = nlo;
  }
}

// LDS counter v lives at word v + v/32: a thread's 32 consecutive counters then start 33 words apart
// from its neighbour's, so the per-thread sequential scan below is free of bank conflicts.
__device__ __forceinline__ u32 gh_slot(u32 v) { return v + (v >> 5); }

__global__ void __launch_bounds__(GH_THREADS)
    k_bucket_prefix(const u32 *__restrict__ keys, u32 total, int low, u32 m, u32 *__restrict__ S) {
  __shared__ u32 hist[(1 << GH_LOW_MAX) + (1 << (GH_LOW_MAX - 5))];
  __shared__ u32 bounds[2];
  __shared__ u32 scan_s[17];
  const u32 tid = threadIdx.x;
  const u32 width = 1u << low;
  const u32 bucket = blockIdx.x;
  for (u32 i = tid; i < width + (width >> 5); i += GH_THREADS) hist[i] = 0;
  if (tid < 128) {
    u32 v = wave_bucket_start(keys, total, low, bucket + (tid >> 6));
    if ((tid & 63) == 0) bounds[tid >> 6] = v;
  }
  __syncthreads();
  const u32 lo = bounds[0], hi = bounds[1];
  // eight independent loads in flight per thread before the first LDS update
  for (u32 i0 = lo + tid; i0 < hi; i0 += 8 * GH_THREADS) {
    u32 g[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
      u32 i = i0 + k * GH_THREADS;
      g[k] = i < hi ? keys[i] : 0xFFFFFFFFu;
    }
#pragma unroll
    for (int k = 0; k < 8; k++)
      if (g[k] <= m) atomicAdd(&hist[gh_slot(g[k] & (width - 1))], 1u);  // LDS counter; fillers (m+1) are skipped
  }
  __syncthreads();
  // inclusive scan of the counters: each thread owns width/256 consecutive ones
  const u32 per = (width + GH_THREADS - 1) / GH_THREADS;
  const u32 j0 = tid * per;
  u32 mine = 0;
  for (u32 j = j0; j < j0 + per && j < width; j++) mine += hist[gh_slot(j)];
  u32 all;
  u32 run = lo + block_scan_incl<OpSum>(mine, scan_s, &all) - mine;
  for (u32 j = j0; j < j0 + per && j < width; j++) {
    run += hist[gh_slot(j)];
    hist[gh_slot(j)] = run;
  }
  __syncthreads();
  const u64 base = (u64)bucket << low;
  for (u32 j = tid; j < width; j += GH_THREADS)
    if (base + j <= m) S[base + j] = hist[gh_slot(j)];
}

// Heavy buckets.  The suffixes of a long run of one symbol in the right block can all fall into one gap of the left
// block: a bucket then holds millions of values, which its one workgroup below would count for milliseconds.  The
// (at most GH_HEAVY_MAX) buckets of more than GH_HEAVY_MIN values are listed, counted by GH_HEAVY_PARTS workgroups each
// into counters in memory, and the bucket's own workgroup starts from those.
constexpr u32 GH_HEAVY_MIN = 1u << 17, GH_HEAVY_MAX = 64, GH_HEAVY_PARTS = 128;

// heavy[0] = how many (may exceed GH_HEAVY_MAX: the ones beyond are not listed), heavy[1 + i] = bucket
__global__ void k_gap_heavy_list(const u32 *__restrict__ start, u32 n_buckets, u32 *__restrict__ heavy) {
  const u32 b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= n_buckets || start[b + 1] - start[b] <= GH_HEAVY_MIN) return;
  const u32 i = atomicAdd(&heavy[0], 1u);
  if (i < GH_HEAVY_MAX) heavy[1 + i] = b;
}

// counts[i << low | v] += the values v of part blockIdx.x of heavy bucket i = blockIdx.y (counts: zeroed by the caller)
__global__ void __launch_bounds__(GH_THREADS)
    k_gap_heavy_count(const u16 *__restrict__ keys, const u32 *__restrict__ start, int low, u32 m, u32 slot,
                      const u32 *__restrict__ heavy, u32 *__restrict__ counts) {
  __shared__ u32 hist[(1 << GH_LOW_MAX) + (1 << (GH_LOW_MAX - 5))];
  const u32 n_heavy = heavy[0] < GH_HEAVY_MAX ? heavy[0] : GH_HEAVY_MAX;
  if (blockIdx.y >= n_heavy) return;
  const u32 tid = threadIdx.x, width = 1u << low, bucket = heavy[1 + blockIdx.y];
  for (u32 i = tid; i < width + (width >> 5); i += GH_THREADS) hist[i] = 0;
  __syncthreads();
  const u32 lo = start[bucket], cnt = start[bucket + 1] - lo;
  const u64 base = (u64)bucket << low;
  const u16 *words = keys + (slot ? (size_t)bucket * slot : (size_t)lo);
  const u32 per = (cnt + GH_HEAVY_PARTS - 1) / GH_HEAVY_PARTS;
  const u32 i_lo = blockIdx.x * per, i_hi = i_lo + per < cnt ? i_lo + per : cnt;
  for (u32 i = i_lo + tid; i < i_hi; i += GH_THREADS) {
    const u32 g = words[i];
    if (base + g <= m) atomicAdd(&hist[gh_slot(g)], 1u);
  }
  __syncthreads();
  for (u32 j = tid; j < width; j += GH_THREADS) {
    const u32 v = hist[gh_slot(j)];
    if (v) atomicAdd(&counts[((size_t)blockIdx.y << low) + j], v);
  }
}

// The same from the two-pass grouping that keeps neither digit (gap_sort_lean): a bucket's values are the 16-bit
// words [start[bucket], start[bucket + 1]), each the low bits of a g whose high bits are the bucket's number.
// heavy, heavy_counts: the buckets already counted (k_gap_heavy_count).
__global__ void __launch_bounds__(GH_THREADS)
    k_bucket_prefix16(const u16 *__restrict__ keys, const u32 *__restrict__ start, int low, u32 m, u32 *__restrict__ S, u32 slot,
                      const u32 *__restrict__ heavy, const u32 *__restrict__ heavy_counts) {
  __shared__ u32 hist[(1 << GH_LOW_MAX) + (1 << (GH_LOW_MAX - 5))];
  __shared__ u32 scan_s[17];
  __shared__ u32 heavy_at;
  const u32 tid = threadIdx.x;
  const u32 width = 1u << low;
  const u32 bucket = blockIdx.x;
  if (tid == 0) heavy_at = ~0u;
  for (u32 i = tid; i < width + (width >> 5); i += GH_THREADS) hist[i] = 0;
  __syncthreads();
  const u32 lo = start[bucket], cnt = start[bucket + 1] - lo;
  const u64 base = (u64)bucket << low;
  if (cnt > GH_HEAVY_MIN && heavy) {  // (the same for the whole workgroup)
    const u32 n_heavy = heavy[0] < GH_HEAVY_MAX ? heavy[0] : GH_HEAVY_MAX;
    if (tid < n_heavy && heavy[1 + tid] == bucket) heavy_at = tid;
    __syncthreads();
    if (heavy_at != ~0u)
      for (u32 j = tid; j < width; j += GH_THREADS) hist[gh_slot(j)] = heavy_counts[((size_t)heavy_at << low) + j];
  }
  // (slot: the bucket's words lie from bucket * slot on, not behind those of the buckets below it)
  const u16 *words = keys + (slot ? (size_t)bucket * slot : (size_t)lo);
  for (u32 i0 = tid; i0 < (heavy_at != ~0u ? 0u : cnt); i0 += 8 * GH_THREADS) {
    u32 g[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
      u32 i = i0 + k * GH_THREADS;
      g[k] = i < cnt ? (u32)words[i] : 0xFFFFFFFFu;
    }
#pragma unroll
    for (int k = 0; k < 8; k++)
      if (g[k] != 0xFFFFFFFFu && base + g[k] <= m) atomicAdd(&hist[gh_slot(g[k])], 1u);  // fillers (m+1) are skipped
  }
  __syncthreads();
  const u32 per = (width + GH_THREADS - 1) / GH_THREADS;
  const u32 j0 = tid * per;
  u32 mine = 0;
  for (u32 j = j0; j < j0 + per && j < width; j++) mine += hist[gh_slot(j)];
  u32 all;
  u32 run = lo + block_scan_incl<OpSum>(mine, scan_s, &all) - mine;
  for (u32 j = j0; j < j0 + per && j < width; j++) {
    run += hist[gh_slot(j)];
    hist[gh_slot(j)] = run;
  }
  __syncthreads();
  for (u32 j = tid; j < width; j += GH_THREADS)
    if (base + j <= m) S[base + j] = hist[gh_slot(j)];
}

// The same for a part of a merge that is spread over several GPUs: the values come grouped from `sources` senders,
// each with the grouping of its own values (gap_group16).  Sender s's words lie at words[word_off[s] ...), bucket b
// of them at [st[b] - st[0], st[b + 1] - st[0]) with st = starts + s * (nbk + 1); the part has nbk buckets, and the
// launch has one workgroup more than that when the part's last gap index opens a bucket of its own (then empty).
constexpr u32 GH_MAX_SOURCES = 256;
__global__ void __launch_bounds__(GH_THREADS)
    k_bucket_prefix16_multi(const u16 *__restrict__ words, const u32 *__restrict__ word_off, const u32 *__restrict__ starts,
                            u32 sources, u32 nbk, int low, u32 m, u32 *__restrict__ S) {
  __shared__ u32 hist[(1 << GH_LOW_MAX) + (1 << (GH_LOW_MAX - 5))];
  __shared__ u32 scan_s[17];
  // the bucket's words of all senders as one index space: sender s holds the indices [cum[s], cum[s + 1]), its first at src[s]
  __shared__ u32 cum[GH_MAX_SOURCES + 1], src[GH_MAX_SOURCES], below_s;
  const u32 tid = threadIdx.x;
  const u32 width = 1u << low;
  const u32 bucket = blockIdx.x, bk = bucket < nbk ? bucket : nbk;
  for (u32 i = tid; i < width + (width >> 5); i += GH_THREADS) hist[i] = 0;
  {
    // one thread per sender (GH_THREADS >= GH_MAX_SOURCES): its words in this bucket and in the buckets below
    u32 mine = 0, lower = 0;
    if (tid < sources) {
      const u32 *st = starts + (size_t)tid * (nbk + 1);
      const u32 first = st[0];
      lower = st[bk] - first;
      if (bucket < nbk) {
        mine = st[bucket + 1] - st[bucket];
        src[tid] = word_off[tid] + (st[bucket] - first);
      }
    }
    u32 all_mine, all_lower;
    const u32 incl = block_scan_incl<OpSum>(mine, scan_s, &all_mine);
    if (tid < sources) cum[tid + 1] = incl;
    if (tid == 0) cum[0] = 0;
    __syncthreads();
    block_scan_incl<OpSum>(lower, scan_s, &all_lower);
    if (tid == 0) below_s = all_lower;
    __syncthreads();
  }
  const u64 base = (u64)bucket << low;
  const u32 below = below_s, total = cum[sources];
  for (u32 x0 = tid; x0 < total; x0 += 8 * GH_THREADS) {
    u32 g[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const u32 x = x0 + k * GH_THREADS;
      g[k] = 0xFFFFFFFFu;
      if (x < total) {
        u32 lo = 0, hi = sources;  // the sender whose range holds x
        while (hi - lo > 1) {
          const u32 mid = (lo + hi) / 2;
          if (cum[mid] <= x) lo = mid;
          else hi = mid;
        }
        g[k] = (u32)words[src[lo] + (x - cum[lo])];
      }
    }
#pragma unroll
    for (int k = 0; k < 8; k++)
      if (g[k] != 0xFFFFFFFFu && base + g[k] <= m) atomicAdd(&hist[gh_slot(g[k])], 1u);  // fillers (above m) are skipped
  }
  __syncthreads();
  const u32 per = (width + GH_THREADS - 1) / GH_THREADS;
  const u32 j0 = tid * per;
  u32 mine = 0;
  for (u32 j = j0; j < j0 + per && j < width; j++) mine += hist[gh_slot(j)];
  u32 all;
  u32 run = below + block_scan_incl<OpSum>(mine, scan_s, &all) - mine;
  for (u32 j = j0; j < j0 + per && j < width; j++) {
    run += hist[gh_slot(j)];
    hist[gh_slot(j)] = run;
  }
  __syncthreads();
  for (u32 j = tid; j < width; j += GH_THREADS)
    if (base + j <= m) S[base + j] = hist[gh_slot(j)];
}

static int bits_for(u64 x) {
  int b = 1;
  while (b < 64 && (x >> b)) b++;
  return b;
}

// Left blocks whose gap indices have 9 to 18 bits above the 13 that a bucket spans take the two-pass grouping
// (gap_sort_lean): 2^21 to 2^31 - 1 gap indices.
bool gap_grouping_applies(const Ctx &c, u64 m) {
  const int high = bits_for(m + 1) - GH_LOW_MAX;
  return c.gap_lean && high > 8 && high <= 18;
}

size_t gap_group16_starts(u64 m) { return ((size_t)1 << (bits_for(m + 1) - GH_LOW_MAX)) + 1; }

void gap_group16(Ctx &c, u32 *g0, u32 *g1, u64 total, u64 m, u32 *start) {
  const int high = bits_for(m + 1) - GH_LOW_MAX;
  const int b2 = (high + 1) / 2, b1 = high - b2;
  u32 *keys[2] = {g0, g1};
  Phase ph(c, "gap_histogram", (double)total * (12 + 10));
  gap_sort_lean(c, keys, total, GH_LOW_MAX, b1, b2, start);
}

void gap_prefix_grouped(Ctx &c, const u16 *words, const u32 *word_off, const u32 *starts, u32 sources, u32 nbk, u64 m, u32 *S) {
  const u32 n_buckets = (u32)((m >> GH_LOW_MAX) + 1);
  if (sources < 1 || sources > GH_MAX_SOURCES) fail(BWT_HIP_EINVAL, "grouped prefix: 1..%u senders", GH_MAX_SOURCES);
  if (n_buckets != nbk && n_buckets != nbk + 1) fail(BWT_HIP_EINTERNAL, "grouped prefix: %u buckets for %lu gap indices", nbk, (unsigned long)(m + 1));
  KernelTimer kt(c, "gap:k_bucket_prefix", (double)(m + 1) * 4);
  hipLaunchKernelGGL(k_bucket_prefix16_multi, dim3(n_buckets), dim3(GH_THREADS), 0, c.stream, words, word_off, starts, sources, nbk,
                     GH_LOW_MAX, (u32)m, S);
  HIP_CHECK(hipGetLastError());
}

// ----------------------------------------------------------------------------
// The list of long periodic stretches: runs of one symbol (period 1) and repeats of a unit of up to RUN_PERIOD_MAX
// symbols that hold a whole aligned stretch of RUN_BLOCK symbols.  One pass over the text finds, for every aligned
// block, the smallest period p with T[i] = T[i+p] for every i of the block; the host joins neighbouring blocks of one
// period; a thread per end extends the stretch by the same rule, 3 * RUN_BLOCK + 8 steps at most (a stretch between
// scanned blocks ends within RUN_BLOCK + 8 of them; behind the last scanned block up to two blocks' worth of text is
// left).  The list only saves work: a stretch listed shorter than it is, or not at all, is read symbol by symbol.
// ----------------------------------------------------------------------------
constexpr u32 RUN_BLOCK = 4096;
constexpr u32 RUN_MAX = 1u << 16;  // listed stretches at most

// flags[block] = smallest period 1..RUN_PERIOD_MAX of the block, 0 if it has none; *found counts the blocks that have
// one.  A wavefront per block, 64 symbols a lane (plus the 8 behind them: the text holds 64 bytes more behind the
// last block), nothing through LDS.
__global__ void __launch_bounds__(256) k_periodic_blocks(const u8 *__restrict__ T, u32 n_blocks, u8 *__restrict__ flags,
                                                         u32 *__restrict__ found) {
  static_assert(RUN_BLOCK == 64 * 64 && RUN_PERIOD_MAX == 8, "64 lanes of 64 symbols; periods 1..8");
  const u32 blk = blockIdx.x * 4 + (threadIdx.x >> 6);  // the same for a whole wavefront
  if (blk >= n_blocks) return;
  const u32 lane = threadIdx.x & 63;
  const u8 *src = T + (u64)blk * RUN_BLOCK + lane * 64;  // T is 16-byte aligned
  u64 w[9];
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const uint4 v = *reinterpret_cast<const uint4 *>(src + q * 16);
    w[2 * q] = (u64)v.x | ((u64)v.y << 32);
    w[2 * q + 1] = (u64)v.z | ((u64)v.w << 32);
  }
  w[8] = *reinterpret_cast<const u64 *>(src + 64);
  u32 mine = 0;  // bit p-1: T[i] != T[i+p] for one of this lane's symbols
#pragma unroll
  for (int k = 0; k < 8; k++) {
#pragma unroll
    for (int p = 1; p < 8; p++) mine |= (u32)(((w[k] >> (8 * p)) | (w[k + 1] << (64 - 8 * p))) != w[k]) << (p - 1);
    mine |= (u32)(w[k + 1] != w[k]) << 7;
  }
  u32 broken = 0;
#pragma unroll
  for (int p = 0; p < 8; p++) broken |= (__ballot((mine >> p) & 1u) != 0ull) ? 1u << p : 0u;
  if (lane == 0) {
    const u32 ok = ~broken & 0xFFu;
    flags[blk] = ok ? (u8)__ffs(ok) : (u8)0;
    if (ok) atomicAdd(found, 1u);
  }
}

// runs[i], runs[count + i], runs[2 * count + i]: S, E, p with T[j] = T[j+p] for S <= j < E on entry; on exit the
// stretch [start, end) that rule extends to (by at most 3 * RUN_BLOCK + 8 positions each way), and its last symbol in
// runs[3 * count + i]
__global__ void k_extend_runs(const u8 *__restrict__ T, u64 n, u32 count, u32 *__restrict__ runs) {
  const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= 2 * count) return;
  const u32 i = t < count ? t : t - count;
  const u64 p = runs[2 * count + i];
  if (t < count) {
    u64 s = runs[i];
    for (u32 k = 0; k < 3 * RUN_BLOCK + 8 && s > 0 && T[s - 1] == T[s - 1 + p]; k++) s--;
    runs[i] = (u32)s;
  } else {
    u64 e = runs[count + i];
    for (u32 k = 0; k < 3 * RUN_BLOCK + 8 && e + p < n && T[e] == T[e + p]; k++) e++;
    runs[count + i] = (u32)(e + p);  // T[j] = T[j-p] for start + p <= j < e + p
    runs[3 * count + i] = T[runs[count + i] - 1];  // (of a run of one symbol: that symbol)
  }
}

void ensure_run_table(Ctx &c) {
  if (c.runs_valid) return;
  if (c.runs) c.release(c.runs);
  c.runs = nullptr;
  c.n_runs = 0;
  c.runs_host.clear();
  c.runs_valid = true;
  const u32 n_blocks = c.n > 64 ? (u32)(((u64)c.n - 64) / RUN_BLOCK) : 0;
  if (!c.run_table || n_blocks == 0) return;
  Phase ph(c, "gap:run_table", (double)c.n);  // (named like a kernel: once per text, not a call of a merge's phases)
  DBuf<u8> flags(c, n_blocks);
  u32 *d_found = c.d_scalars + 24;
  HIP_CHECK(hipMemsetAsync(d_found, 0, sizeof(u32), c.stream));
  hipLaunchKernelGGL(k_periodic_blocks, dim3(ceil_div(n_blocks, 4)), dim3(256), 0, c.stream, c.T, n_blocks, flags.get(), d_found);
  HIP_CHECK(hipGetLastError());
  HIP_CHECK(hipMemcpyAsync(c.h_scalars, d_found, sizeof(u32), hipMemcpyDeviceToHost, c.stream));
  c.sync();
  if (c.h_scalars[0] == 0) return;  // the usual case: nothing to list
  std::vector<u8> h(n_blocks);
  HIP_CHECK(hipMemcpyAsync(h.data(), flags.get(), n_blocks, hipMemcpyDeviceToHost, c.stream));
  c.sync();
  std::vector<u32> first, last, period;  // neighbouring blocks of one period, joined
  for (u32 i = 0; i < n_blocks;) {
    if (!h[i]) {
      i++;
      continue;
    }
    u32 j = i + 1;
    while (j < n_blocks && h[j] == h[i]) j++;
    if (first.size() < RUN_MAX) {
      first.push_back(i * RUN_BLOCK);
      last.push_back(j * RUN_BLOCK);  // (n <= 2^32 - 65536: no wrap)
      period.push_back(h[i]);
    }
    i = j;
  }
  const u32 count = (u32)first.size();
  if (count == 0) return;
  first.insert(first.end(), last.begin(), last.end());
  first.insert(first.end(), period.begin(), period.end());
  first.resize((size_t)count * 4, 0);
  c.runs = (u32 *)c.alloc((size_t)count * 4 * sizeof(u32));
  HIP_CHECK(hipMemcpyAsync(c.runs, first.data(), (size_t)count * 4 * sizeof(u32), hipMemcpyHostToDevice, c.stream));
  hipLaunchKernelGGL(k_extend_runs, dim3(ceil_div(2 * count, 64)), dim3(64), 0, c.stream, c.T, (u64)c.n, count, c.runs);
  HIP_CHECK(hipGetLastError());
  HIP_CHECK(hipMemcpyAsync(first.data(), c.runs, (size_t)count * 4 * sizeof(u32), hipMemcpyDeviceToHost, c.stream));
  c.sync();  // `first` is on the stack
  c.n_runs = count;
  c.runs_host = std::move(first);
}

// LF-mapping chains of the right positions [ra, rb) against the block l (ra >= l.b; the positions need
// not belong to one block: g depends on l and the text only).  g0 receives seg * n_chains values in
// step-major order, fillers (m + 1) in the tail of a short last chain.
ChainLayout run_chains(Ctx &c, Block &l, i64 ra, i64 rb, i64 seg_req, DBuf<u32> &g0) {
  if (ra < l.b || rb <= ra || rb > c.n)
    fail(BWT_HIP_EINVAL, "chains: positions [%ld,%ld) are not to the right of block [%ld,%ld)", (long)ra, (long)rb,
         (long)l.a, (long)l.b);
  build_wavelet(c, l);
  c.wait_text((u64)c.n);  // the chains' suffix comparisons may read to the end of the text
  ensure_run_table(c);
  const u64 m = (u64)l.len(), rn = (u64)(rb - ra);
  if (m + rn >= (1ull << 32) - 2) fail(BWT_HIP_EINVAL, "merged block of %lu symbols exceeds 2^32-3", (unsigned long)(m + rn));
  ChainLayout lay;
  u64 seg = seg_req > 0 ? (u64)seg_req : 1024;
  if (seg > rn) seg = rn;
  lay.ra = (u64)ra;
  lay.rb = (u64)rb;
  lay.seg = (u32)seg;
  lay.n_chains = (u32)((rn + seg - 1) / seg);
  const u64 total = seg * (u64)lay.n_chains;
  g0 = DBuf<u32>(c, total);
  ChainParams P;
  P.T = c.T;
  P.n = (u64)c.n;
  P.ra = (u64)ra;
  P.rb = (u64)rb;
  P.lb = (u64)l.b;
  P.m = (u32)m;
  P.seg = lay.seg;
  P.n_chains = lay.n_chains;
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
  // chains can be deferred when the left block has samples to look suffix array entries up with
  DBuf<u32> defer_list;
  u32 *d_count = c.d_scalars + 20;
  P.defer_list = nullptr;
  P.defer_count = d_count;
  P.samp_bits = P.by_rank = P.by_rank_pos = nullptr;
  P.ns = 0;
  P.runs = c.runs;
  P.n_runs = c.n_runs;
  P.dc = dc_view(c);
  // A text whose sample ranks exist has been found repetitive by its leaves: its chains' heads will not come out of a
  // short backward search, so all of them go the second way at once.
  const bool all_second = c.dc.valid && l.n_samp > 0 && c.defer_chains;
  if (l.n_samp > 0 && c.defer_chains && !all_second) {
    defer_list = DBuf<u32>(c, lay.n_chains);
    P.defer_list = defer_list.get();
    HIP_CHECK(hipMemsetAsync(d_count, 0, sizeof(u32), c.stream));
  }
  if (!all_second) {
    // 64-byte random reads: one rank block per level and step, one 16-byte text block every 16 steps
    Phase ph(c, "gap_chains", (double)rn * l.wm.levels + (double)rn / 16);
    hipLaunchKernelGGL(k_chains, dim3(ceil_div(lay.n_chains, CHAIN_THREADS)), dim3(CHAIN_THREADS), 0, c.stream, P);
    HIP_CHECK(hipGetLastError());
  }
  if (P.defer_list || all_second) {
    u32 deferred = lay.n_chains;
    if (!all_second) {
      HIP_CHECK(hipMemcpyAsync(c.h_scalars, d_count, sizeof(u32), hipMemcpyDeviceToHost, c.stream));
      c.sync();
      deferred = c.h_scalars[0];
    }
    if (deferred > 0) {
      RankIndex ix;
      build_rank_index(c, l, ix);
      P.samp_bits = ix.bits.get();
      P.by_rank = ix.by_rank.get();
      P.by_rank_pos = ix.by_rank_pos.get();
      P.ns = ix.ns;
      Phase ph(c, "gap_chains_deferred", (double)deferred * seg * l.wm.levels);
      hipLaunchKernelGGL(k_chains_deferred, dim3(ceil_div(deferred, CHAIN_THREADS)), dim3(CHAIN_THREADS), 0, c.stream, P, deferred);
      HIP_CHECK(hipGetLastError());
      c.sync();  // the index dies with this scope
    }
  }
  return lay;
}

void chains_text_order(Ctx &c, const ChainLayout &lay, const u32 *g0, u32 *out) {
  const u32 tiles_c = ceil_div(lay.n_chains, GT_TILE), tiles_s = ceil_div(lay.seg, GT_TILE);
  if ((u64)tiles_c * tiles_s >= (1ull << 31)) fail(BWT_HIP_EINVAL, "chains: %u chains of %u steps are too many tiles", lay.n_chains, lay.seg);
  hipLaunchKernelGGL(k_g_text_order, dim3(tiles_c * tiles_s), dim3(256), 0, c.stream, g0, lay.rb - lay.ra, lay.seg, lay.n_chains,
                     tiles_c, out);
  HIP_CHECK(hipGetLastError());
}

// S[i] = #{ k : g[k] <= i } for i in [0, m], from `total` values g[k] in [0, m] (values above m are
// ignored).  g0 holds the values and is destroyed; g1 is scratch of the same size.
void gap_prefix(Ctx &c, u32 *g0, u32 *g1, u64 total, u64 m, u32 *S) {
  const int bits = bits_for(m + 1);
  const int low = bits < GH_LOW_MAX ? bits : GH_LOW_MAX;
  const int passes = (bits - low + 7) / 8;
  u32 *keys[2] = {g0, g1};
  const u32 n_buckets = (u32)(((m + 1) + ((u64)1 << low) - 1) >> low);
  const int high = bits - low;
  if (total > 0 && high > 8 && high <= 18 && low == GH_LOW_MAX && c.gap_lean) {
    // two passes that keep neither digit; the second leaves 16-bit bucket-local values and the buckets' bounds
    const int b2 = (high + 1) / 2, b1 = high - b2;
    Phase ph(c, "gap_histogram", (double)total * (12 + 10 + 2) + (double)(m + 1) * 4);
    DBuf<u32> start(c, ((size_t)1 << high) + 1);
    u32 slot = 0;
    gap_sort_lean(c, keys, total, low, b1, b2, start.get(), &slot, n_buckets);
    KernelTimer kt(c, "gap:k_bucket_prefix", (double)total * 2 + (double)(m + 1) * 4);
    DBuf<u32> heavy, heavy_counts;
    if (!c.arena_on) {  // (not in the call that may not allocate: 2 MB of counters)
      heavy = DBuf<u32>(c, 1 + GH_HEAVY_MAX);
      heavy_counts = DBuf<u32>(c, (size_t)GH_HEAVY_MAX << low);
      HIP_CHECK(hipMemsetAsync(heavy.get(), 0, sizeof(u32), c.stream));
      HIP_CHECK(hipMemsetAsync(heavy_counts.get(), 0, ((size_t)GH_HEAVY_MAX << low) * sizeof(u32), c.stream));
      hipLaunchKernelGGL(k_gap_heavy_list, dim3(ceil_div(n_buckets, 256)), dim3(256), 0, c.stream, (const u32 *)start.get(), n_buckets, heavy.get());
      hipLaunchKernelGGL(k_gap_heavy_count, dim3(GH_HEAVY_PARTS, GH_HEAVY_MAX), dim3(GH_THREADS), 0, c.stream,
                         reinterpret_cast<const u16 *>(keys[0]), (const u32 *)start.get(), low, (u32)m, slot, (const u32 *)heavy.get(),
                         heavy_counts.get());
    }
    hipLaunchKernelGGL(k_bucket_prefix16, dim3(n_buckets), dim3(GH_THREADS), 0, c.stream, reinterpret_cast<const u16 *>(keys[0]),
                       (const u32 *)start.get(), low, (u32)m, S, slot, (const u32 *)heavy.get(), (const u32 *)heavy_counts.get());
    HIP_CHECK(hipGetLastError());
    return;
  }
  Phase ph(c, "gap_histogram", (double)total * (passes * 12 + 4) + (double)(m + 1) * 4);
  u32 *vals[2] = {nullptr, nullptr};
  int cur = 0;
  if (total > 0) {
    c.sort_tag = "gap:";
    cur = radix_sort_u32(c, keys, vals, total, low, bits, 2);
    c.sort_tag = "";
  }
  KernelTimer kt(c, "gap:k_bucket_prefix", (double)total * 4 + (double)(m + 1) * 4);
  hipLaunchKernelGGL(k_bucket_prefix, dim3(n_buckets), dim3(GH_THREADS), 0, c.stream, keys[cur], (u32)total, low, (u32)m,
                     S);
  HIP_CHECK(hipGetLastError());
}

GapResult build_gap(Ctx &c, Block &l, Block &r, i64 seg_req, bool want_g) {
  if (l.b != r.a) fail(BWT_HIP_EINVAL, "blocks are not adjacent: [%ld,%ld) and [%ld,%ld)", (long)l.a, (long)l.b,
                       (long)r.a, (long)r.b);
  const u64 m = (u64)l.len(), rn = (u64)r.len();
  GapResult G;
  DBuf<u32> g0;
  ChainLayout lay = run_chains(c, l, r.a, r.b, seg_req, g0);
  G.seg = lay.seg;
  G.n_chains = lay.n_chains;
  const u64 total = (u64)lay.seg * lay.n_chains;
  DBuf<u32> g1(c, total);
  G.samp_g = DBuf<u32>(c, (size_t)(r.n_samp > 0 ? r.n_samp : 1));
  if (r.n_samp > 0) {
    hipLaunchKernelGGL(k_sample_g, dim3(ceil_div(r.n_samp, 256)), dim3(256), 0, c.stream, g0.get(), r.samp_pos.get(),
                       (u32)r.n_samp, lay.ra, lay.rb, lay.seg, lay.n_chains, G.samp_g.get());
    HIP_CHECK(hipGetLastError());
  }
  if (want_g) {
    G.g_text = DBuf<u32>(c, rn);
    chains_text_order(c, lay, g0.get(), G.g_text.get());
  }
  G.S = DBuf<u32>(c, m + 1);
  gap_prefix(c, g0.get(), g1.get(), total, m, G.S.get());
  return G;
}

}  // namespace bwtamd
