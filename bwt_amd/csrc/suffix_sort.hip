// Leaf suffix array: positions a..b-1 of the text ordered by their FULL-TEXT
// suffixes (what the reference's idx_less comparison sort produces,
// include/suffix_array.h:14-49), by radix sort + prefix doubling.
//
//   round 0   key = codes of the first K0 symbols (plus the preceding symbol,
//             which is not sorted on and later yields the BWT symbol without a
//             read of the text); LSD radix sort over the window [a, e),
//             e = b + slack; the first pass computes the keys from the text on
//             the fly.  Two layouts (TextKeys in internal.h): (key, position)
//             pairs of 12 bytes, or -- small alphabets, windows below 2^29 --
//             one 8-byte element holding codes, preceding symbol and position
//   buckets   large windows of 8-byte elements stop the radix sort after the top 16
//             key bits: a bucket (equal top 16 bits) fits LDS, where one workgroup
//             orders it on the remaining bits and does the direct pass below
//             without the sorted window ever going back to memory (k_bucket_emit)
//   direct    one pass that ends the sort on texts without long repeats: a suffix
//             whose key is unique is final; one that shares its key counts the
//             members of its (small) group that are smaller by comparing the text
//             itself; either way it then writes its own BWT symbol, samples and
//             suffix array entry.  Given up as a whole when a group is large or
//             a comparison runs long; then:
//   tie list  suffixes whose key is shared form groups; only those stay active
//   text rnd  active suffixes are ordered inside their group by the next K1
//             symbols read from the text (no inverse array needed)
//   doubling  the inverse suffix array is built once, then every round orders
//             the active suffixes by (group, rank[pos + h]) and doubles h
//
// A group is named by the rank of its first member, which does not change when
// the group is refined, so ranks of inactive suffixes never need rewriting.
//
// The window is longer than the leaf because comparisons of leaf suffixes run
// past b.  Ranks of window positions p are exact to depth h while p + h <= e;
// a leaf position that would have to look past e raises `overflow` and the
// caller retries with a longer window (never needed when e == n).
#include <math.h>
#include <string.h>

#include <type_traits>

#include "bucket_rank.h"
#include "device_utils.h"
#include "internal.h"
#include "scan_tmpl.h"

namespace bwtamd {

// ----------------------------------------------------------------------------
// grouping scan: counts ties and carries the last group head
// ----------------------------------------------------------------------------
struct TieHead {
  u32 cnt;   // number of tied (active) elements
  u32 hmax;  // 1 + index of the last group head seen, 0 if none
};
struct TieHeadOp {
  __device__ __forceinline__ static TieHead identity() { return TieHead{0, 0}; }
  __device__ __forceinline__ static TieHead apply(TieHead x, TieHead y) {
    return TieHead{x.cnt + y.cnt, x.hmax > y.hmax ? x.hmax : y.hmax};
  }
};

// 64-bit keys held as one array (refinement rounds)
struct PlainKeys {
  const u64 *p;
  __device__ __forceinline__ u64 operator[](size_t i) const { return p[i]; }
};

// The sorted window after round 0, in either layout.  key(i): symbol codes in the top bits, the preceding
// symbol in the low byte.  pos(i): text position.  i0 of the *4 forms is a multiple of four.
struct PairsView {
  const u32 *lo, *hi;  // key words, both 16-byte aligned
  const u32 *sa;
  __device__ __forceinline__ u64 key(size_t i) const { return ((u64)hi[i] << 32) | lo[i]; }
  __device__ __forceinline__ u32 pos(size_t i) const { return sa[i]; }
  __device__ __forceinline__ void key4(size_t i0, u64 (&k)[4]) const {
    const uint4 l = *reinterpret_cast<const uint4 *>(lo + i0), h = *reinterpret_cast<const uint4 *>(hi + i0);
    k[0] = ((u64)h.x << 32) | l.x;
    k[1] = ((u64)h.y << 32) | l.y;
    k[2] = ((u64)h.z << 32) | l.z;
    k[3] = ((u64)h.w << 32) | l.w;
  }
  __device__ __forceinline__ void pos4(size_t i0, u32 (&p)[4]) const {
    const uint4 v = *reinterpret_cast<const uint4 *>(sa + i0);
    p[0] = v.x;
    p[1] = v.y;
    p[2] = v.z;
    p[3] = v.w;
  }
};
struct PackedView {
  const u64 *e;    // | codes, 32 bits | preceding symbol | position - a, pbits |
  const u32 *sa;   // null, or the positions once refinement rounds have taken them over
  u32 a, posmask;
  int pbits;
  static constexpr bool WIDE = false;
  // what the bucket kernel needs of an element (sd: its side byte, none in this layout)
  __device__ __forceinline__ const u64 *bucket_elems(u32, u32 first_rank) const { return e + first_rank; }
  __device__ __forceinline__ u32 bucket_key(u64 x, u32) const { return (u32)(x >> 32) & 0xFFFFu; }  // the 16 key bits below the bucket's
  __device__ __forceinline__ u32 elem_pos(u64 x) const { return a + ((u32)x & posmask); }
  __device__ __forceinline__ u32 elem_prev(u64 x, u32) const { return ((u32)x >> pbits) & 0xFFu; }
  __device__ __forceinline__ u64 cvt(u64 x) const { return (x & 0xFFFFFFFF00000000ull) | ((u32)x >> pbits); }
  __device__ __forceinline__ u64 key(size_t i) const { return cvt(e[i]); }
  __device__ __forceinline__ u32 pos(size_t i) const { return sa ? sa[i] : a + ((u32)e[i] & posmask); }
  __device__ __forceinline__ void raw4(size_t i0, u64 (&x)[4]) const {
    const uint4 v0 = *reinterpret_cast<const uint4 *>(e + i0), v1 = *reinterpret_cast<const uint4 *>(e + i0 + 2);
    x[0] = ((u64)v0.y << 32) | v0.x;
    x[1] = ((u64)v0.w << 32) | v0.z;
    x[2] = ((u64)v1.y << 32) | v1.x;
    x[3] = ((u64)v1.w << 32) | v1.z;
  }
  __device__ __forceinline__ void key4(size_t i0, u64 (&k)[4]) const {
    raw4(i0, k);
#pragma unroll
    for (int j = 0; j < 4; j++) k[j] = cvt(k[j]);
  }
  __device__ __forceinline__ void pos4(size_t i0, u32 (&p)[4]) const {
    if (sa) {
      const uint4 v = *reinterpret_cast<const uint4 *>(sa + i0);
      p[0] = v.x;
      p[1] = v.y;
      p[2] = v.z;
      p[3] = v.w;
    } else {
      u64 x[4];
      raw4(i0, x);
#pragma unroll
      for (int j = 0; j < 4; j++) p[j] = a + ((u32)x[j] & posmask);
    }
  }
};

// Wide elements (TextKeys::wide) sorted on their top bbits key bits; only the bucket kernel reads them.
struct WideView {
  const u64 *e;    // | key bits 63..32 | absolute position |
  const u8 *side;  // | key bits 31..28 | preceding symbol |
  int kshift;      // 20 - bbits: the 16 key bits below the bucket's start this far up in the 36-bit key
  static constexpr bool WIDE = true;
  __device__ __forceinline__ const u64 *bucket_elems(u32, u32 first_rank) const { return e + first_rank; }
  __device__ __forceinline__ u32 bucket_key(u64 x, u32 sd) const {
    return (u32)((((x >> 32) << 4) | (sd >> 4)) >> kshift) & 0xFFFFu;
  }
  __device__ __forceinline__ u32 elem_pos(u64 x) const { return (u32)x; }
  __device__ __forceinline__ u32 elem_prev(u64, u32 sd) const { return sd & 15u; }
};

// Lean elements (TextKeys::lean) after their two passes; only the bucket kernel reads them.
//   | d2, b2 bits | deep | zero padding | preceding symbol, vbits | position - a, pbits |
// The bucket kernel orders a bucket on the top 16 bits of `deep`; the xbits bits below those settle most of what
// that leaves tied without a look at the text.
struct LeanView {
  const u64 *e;
  u32 a, posmask, vmask;
  int pbits, b2, xbits;
  u32 slot;  // not 0: bucket k's elements lie from k * slot on, whatever its first rank
  static constexpr bool WIDE = false;
  __device__ __forceinline__ const u64 *bucket_elems(u32 k, u32 first_rank) const { return slot ? e + (size_t)k * slot : e + first_rank; }
  __device__ __forceinline__ u32 bucket_key(u64 x, u32) const { return (u32)((x << b2) >> 48); }
  __device__ __forceinline__ u32 elem_pos(u64 x) const { return a + ((u32)x & posmask); }
  __device__ __forceinline__ u32 elem_prev(u64 x, u32) const { return (u32)(x >> pbits) & vmask; }
  __device__ __forceinline__ u32 extra(u64 x) const { return xbits ? (u32)((x << (b2 + 16)) >> (64 - xbits)) : 0u; }
  // all key bits an element of a bucket shows below the bucket's own (what bucket_key and extra split)
  __device__ __forceinline__ u64 sort_bits(u64 x) const { return (x << b2) >> (48 - xbits); }
};

__device__ __host__ inline void set_positions(PairsView &, const u32 *) {}
__device__ __host__ inline void set_positions(PackedView &v, const u32 *sa) { v.sa = sa; }

// zero_low_unique (refinement rounds): a key whose low word is 0 is never tied.  That word is the
// secondary key "nothing is left of the suffix at this depth"; several such suffixes in one group are the
// ones that ran into the terminator inside their round-0 key (TextKeys, zero-based coding), they already
// stand in their final order and the stable sort has kept it.
template <class Keys>
struct GroupIn {
  Keys key;
  size_t n;
  u64 mask;  // the key bits that count
  bool zero_low_unique;
  __device__ __forceinline__ bool same(u64 x, u64 y) const { return x == y && !(zero_low_unique && (u32)x == 0); }
  __device__ __forceinline__ TieHead operator()(size_t i) const {
    u64 k = key[i] & mask;
    bool head = (i == 0) || !same(key[i - 1] & mask, k);
    bool next_head = (i + 1 == n) || !same(key[i + 1] & mask, k);
    return TieHead{(u32)!(head && next_head), head ? (u32)(i + 1) : 0u};
  }
};

// ties of the sorted window: four consecutive elements at a time
template <class View>
struct ViewTies {
  View v;
  size_t n;
  u64 mask;
  __device__ __forceinline__ TieHead operator()(size_t i) const {
    u64 k = v.key(i) & mask;
    bool head = (i == 0) || (v.key(i - 1) & mask) != k;
    bool next_head = (i + 1 == n) || (v.key(i + 1) & mask) != k;
    return TieHead{(u32)!(head && next_head), head ? (u32)(i + 1) : 0u};
  }
  __device__ __forceinline__ void vec4(size_t i0, TieHead *out) const {
    u64 k[6], c[4];
    v.key4(i0, c);
#pragma unroll
    for (int j = 0; j < 4; j++) k[j + 1] = c[j] & mask;
    k[0] = i0 ? (v.key(i0 - 1) & mask) : ~k[1];
    k[5] = (i0 + 4 < n) ? (v.key(i0 + 4) & mask) : ~k[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      bool head = k[j] != k[j + 1], next_head = k[j + 1] != k[j + 2];
      out[j] = TieHead{(u32)!(head && next_head), head ? (u32)(i0 + j + 1) : 0u};
    }
  }
};

// after round 0: emit the active list (slot = rank index, pos, group = head rank)
template <class View>
struct FirstListOut {
  View v;
  u32 *slot, *pos, *grp;
  u32 *total;
  size_t n;
  // not null: receives the size of the largest group above group_limit, if there is one (zeroed by the caller);
  // mask = the key bits groups are made on
  u32 *max_group = nullptr;
  u64 mask = 0;
  u32 group_limit = 0;
  __device__ __forceinline__ void operator()(size_t i, TieHead ex, TieHead in) const {
    if (in.cnt != ex.cnt) {
      u32 k = ex.cnt;
      slot[k] = (u32)i;
      pos[k] = v.pos(i);
      grp[k] = in.hmax - 1;
      // (a plain store by the few groups above the limit: one atomic per group on one address would serialise them)
      if (max_group && (u32)(i + 2 - in.hmax) > group_limit && (i + 1 == n || (v.key(i + 1) & mask) != (v.key(i) & mask)))
        atomicMax(max_group, (u32)(i + 2 - in.hmax));
    }
    if (i + 1 == n) *total = in.cnt;
  }
};

// after a refinement round: write the sorted positions back to their slots,
// rename refined groups, update the inverse array, emit the next active list
struct RefineOut {
  const u32 *old_slot;  // slots of the active list, ascending
  const u32 *spos;      // positions in refined order
  u32 *sa;
  u32 *isa;  // may be null
  u32 a;
  u32 *slot, *pos, *grp;  // next active list
  u32 *total;
  size_t n;
  __device__ __forceinline__ void operator()(size_t i, TieHead ex, TieHead in) const {
    u32 s = old_slot[i];
    u32 p = spos[i];
    u32 g = old_slot[in.hmax - 1];
    sa[s] = p;
    if (isa) isa[p - a] = g;
    if (in.cnt != ex.cnt) {
      u32 k = ex.cnt;
      slot[k] = s;
      pos[k] = p;
      grp[k] = g;
    }
    if (i + 1 == n) *total = in.cnt;
  }
};

// Group names of the active list as dense indices 0, 1, 2 ... in list order (a group's members are consecutive in the
// list and the names ascend along it): the composite keys below then need bits_for(groups) bits for the group instead
// of a rank's 29 to 32, which on the lists that reach the radix passes (few, long groups) is three passes fewer of eight.
struct GroupHeadIn {
  const u32 *grp;
  __device__ __forceinline__ u32 operator()(size_t i) const { return (i == 0 || grp[i] != grp[i - 1]) ? 1u : 0u; }
};
struct DenseGroupOut {
  u32 *dense;
  u32 *total;
  size_t n;
  __device__ __forceinline__ void operator()(size_t i, u32 ex, u32 in) const {
    (void)ex;
    dense[i] = in - 1;
    if (i + 1 == n) *total = in;
  }
};

// ----------------------------------------------------------------------------
// secondary keys
// ----------------------------------------------------------------------------

// text round: composite = group << 32 | next K1 symbols after depth h
__global__ void k_comp_text(const u8 *__restrict__ T, u64 n, const u32 *__restrict__ pos,
                            const u32 *__restrict__ grp, u32 M, u32 h, int w, int K1,
                            u64 *__restrict__ comp, u32 *__restrict__ pout) {
  u32 k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= M) return;
  u32 p = pos[k];
  u64 q = (u64)p + h;
  u32 sec = 0;
  for (int j = 0; j < K1; j++) {
    u32 c = (q + j < n) ? T[q + j] : 0u;
    sec = (sec << w) | c;
  }
  sec <<= (32 - K1 * w);
  comp[k] = ((u64)grp[k] << 32) | sec;
  pout[k] = p;
}

// doubling round at depth h: composite = group << 32 | (rank of suffix pos+h, plus one; 0 past the text)
//
// rank[q] at depth h is exact when the h symbols from q lie inside the window (q + h <= e), or when
// h <= h_exact, the depth the ranks reached by reading the text itself (round 0 and the text rounds).
// A position whose look-up is not exact cannot be placed by this window: a leaf position raises
// `overflow` (the caller retries with a longer window); a position of the slack beyond the leaf is
// retired with a key of its own, so that it leaves the active list.  Its rank is then arbitrary inside
// its group, which is harmless: by the rule above nothing exact ever reads it.
//
// Runs of one symbol (RunJump::state != 0).  A suffix c^L X with L >= h stays tied with every other such suffix until
// h exceeds L: log2 of the run's length rounds, each over all of the run's suffixes.  Where a window holds long runs
// the rounds take them out of that.
// state 1, a round of its own before the first doubling round, at that round's depth h_jump, which it leaves as it is:
// the group whose first h_jump symbols are c holds exactly the suffixes with L >= h_jump; it is cut into those with
// L < 2 h_jump whose X begins with a smaller symbol than c (key 1), the long ones (L >= 2 h_jump) by what the run
// decides of their order -- by L, shorter first where X begins with a smaller symbol than c, longer first where with a
// larger one, all of the first kind before all of the second: the key of the runs' closed form, "Runs in closed form" --
// and the short ones with a larger symbol behind the run (key 2^32 - 1): their true order.  Every other suffix keeps
// its group (key 1 for all).  No rank is looked up.  All groups still agree on h_jump symbols, the long ones' on L.
// state 2, the doubling rounds: a suffix with L >= max(h, 2 h_jump) steps over what is left of its run, secondary key
// = rank of X (whose first h symbols must lie inside the window like those of any look-up).  Its group is of one L,
// so all its members do the same, and what the round leaves tied agrees on L + h >= 2h symbols: the rounds' invariant.
// A suffix whose run is shorter than h is looked up at p + h like any other.
struct RunJump {
  int state;          // 0: off
  u64 h_jump;
  const u32 *ends;    // ends[N - 1 - i] = N - i_end for the last position i_end >= i of the run that holds window index i,
                      // 0 where the run reaches the window's last position (a maximum scan from the window's end)
  u64 N, tail;        // window length; how far the run at the window's end goes on behind it
  const u8 *T;
  // what is left of the run from text position p = a + i on, p included
  __device__ __forceinline__ u64 left(u64 i) const {
    const u32 v = ends[N - 1 - i];
    return v ? (N - v) - i + 1 : (N - i) + tail;
  }
};
// the look-up position of the suffix at p in a round of depth h (*q = p + h on entry), or true: *sec_out is the key and
// nothing is looked up
__device__ __forceinline__ bool run_jump_key(const RunJump &J, u64 a, u64 n, u64 h, u32 p, u64 *q, u32 *sec_out) {
  const u64 L = J.left((u64)p - a);
  if (J.state == 1) {
    u32 key = 1;
    if (L >= J.h_jump) {
      const u64 end = (u64)p + L;
      const u32 c = J.T[p], f = end < n ? J.T[end] : 0u;
      if (L >= 2 * J.h_jump) key = f > c ? 0x80000000u + (0x7FFFFFFEu - (u32)L) : 2u + (u32)L;
      else key = f > c ? 0xFFFFFFFFu : 1u;
    }
    *sec_out = key;
    return true;
  }
  if (L >= h && L >= 2 * J.h_jump) *q = (u64)p + L;
  return false;
}

__global__ void k_comp_isa(const u32 *__restrict__ isa, u64 a, u64 b, u64 e, u64 n, u64 h_exact,
                           const u32 *__restrict__ pos, const u32 *__restrict__ grp, u32 M, u64 h,
                           u64 *__restrict__ comp, u32 *__restrict__ pout, u32 *__restrict__ overflow, RunJump J) {
  u32 k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= M) return;
  u32 p = pos[k];
  u64 q = (u64)p + h;
  if (J.state) {
    u32 key;
    if (run_jump_key(J, a, n, h, p, &q, &key)) {
      comp[k] = ((u64)grp[k] << 32) | key;
      pout[k] = p;
      return;
    }
  }
  u64 need = h > h_exact ? q + h : q + 1;  // end of the stretch that must lie inside the window
  u32 sec;
  if (e < n && need > e) {
    if (p < b) *overflow = 1;
    sec = 0xFFFFFFFFu - (u32)(p - a);  // unique among the retired
  } else {
    sec = q < e ? isa[q - a] + 1 : 0u;  // q >= e only when e == n: the empty suffix is the smallest
  }
  comp[k] = ((u64)grp[k] << 32) | sec;
  pout[k] = p;
}

// ----------------------------------------------------------------------------
// A doubling round in one pass over the active list, for lists whose groups are short (repetitive texts:
// every suffix is tied with the same position of the other copies).  The list is in rank order and a group's
// members are consecutive in it, so ordering (group, rank[pos + h]) is a sort INSIDE each group: a workgroup
// takes the groups that start in its stretch of the list, reads the secondary keys (the round's one random
// access per suffix), ranks every member among its group in LDS, and writes positions, new group names and
// "still tied" flags.  Eight radix passes over the list per round become one.  The inverse array is updated by
// the compaction pass that follows, never here: a group must be ordered on ranks of ONE depth, and another
// workgroup may be reading them.  A group longer than a workgroup holds raises *fail; nothing that counts has
// been written then and the round is done by the radix passes.
// ----------------------------------------------------------------------------
// 512 threads and 50 KB of LDS per workgroup: three workgroups share a CU, so that one's gather of secondary keys
// runs under another's ordering in LDS (1024 threads with twice the entries, one workgroup per CU: 9.5 instead of 6.5
// ms per round of 2.3e8 entries).
constexpr int RT_THREADS = 512;
constexpr int RT_TILE = 1024;  // list entries whose groups a workgroup owns
constexpr int RT_CAP = 2048;   // entries it can hold: the groups that start in its stretch may run this far

struct RefineTileParams {
  const u32 *slot, *pos, *grp;  // the active list: rank index, position, group name (rank of the group's first member)
  u32 M;
  const u32 *isa;
  u32 *sa;
  u64 a, b, e, n, h_exact, h;
  u32 *overflow, *fail;
  u32 *npos, *ngrp;  // per list entry, in the new order: position, new group name (the rank indices stay as they are)
  u8 *ntied;         // bit 0: still tied; bit 1: the group name changed (the inverse array needs it)
  u32 *wg_tied;      // per workgroup (zeroed by the caller): its entries that are still tied
  RunJump J;         // runs of one symbol: see k_comp_isa
};

__global__ void __launch_bounds__(RT_THREADS) k_refine_tile(RefineTileParams P) {
  __shared__ u32 A[RT_CAP];   // group names, later the sorted positions
  __shared__ u32 Bp[RT_CAP];  // positions
  __shared__ u32 Cs[RT_CAP];  // secondary keys
  __shared__ u32 Ds[RT_CAP];  // rank indices
  __shared__ u32 Es[RT_CAP];  // sorted secondary keys
  __shared__ u16 gs_s[RT_CAP];  // first index of the entry's group
  __shared__ u16 ge_s[RT_CAP];  // at a group's first index: one past its last
  __shared__ u32 tmx[RT_THREADS];
  __shared__ u32 scan_s[17];
  __shared__ u32 bounds[2];
  static_assert(RT_CAP == 4 * RT_THREADS, "four entries per thread");
  const u32 tid = threadIdx.x;
  const u32 base = blockIdx.x * RT_TILE;
  const u32 cnt = P.M - base < (u32)RT_CAP ? P.M - base : (u32)RT_CAP;
  const u32 prev_grp = base ? P.grp[base - 1] : 0u;
  if (tid < 2) bounds[tid] = RT_CAP;
  for (u32 i = tid; i < cnt; i += RT_THREADS) A[i] = P.grp[base + i];
  __syncthreads();
  auto is_head = [&](u32 i) { return (base + i == 0) || A[i] != (i ? A[i - 1] : prev_grp); };
  for (u32 i = tid; i < cnt; i += RT_THREADS)
    if (is_head(i)) atomicMin(&bounds[i < (u32)RT_TILE ? 0 : 1], i);
  __syncthreads();
  const u32 lo = bounds[0];
  if (lo == (u32)RT_CAP) return;  // no group starts here: the stretch belongs to a group of an earlier workgroup
  u32 hi = bounds[1];
  if (hi == (u32)RT_CAP) {
    if (base + cnt == P.M) {
      hi = cnt;
    } else {
      if (tid == 0) *P.fail = 1;  // the last group runs past what this workgroup holds
      return;
    }
  }
  // positions, rank indices, secondary keys (k_comp_isa's rules) of the entries lo .. hi-1
  for (u32 i = lo + tid; i < hi; i += RT_THREADS) {
    const u32 p = P.pos[base + i];
    Bp[i] = p;
    Ds[i] = P.slot[base + i];
    u64 q = (u64)p + P.h;
    u32 sec;
    if (P.J.state && run_jump_key(P.J, P.a, P.n, P.h, p, &q, &sec)) {
      Cs[i] = sec;
      continue;
    }
    const u64 need = P.h > P.h_exact ? q + P.h : q + 1;
    if (P.e < P.n && need > P.e) {
      if (p < P.b) *P.overflow = 1;
      sec = 0xFFFFFFFFu - (u32)(p - P.a);
    } else {
      sec = q < P.e ? P.isa[q - P.a] + 1 : 0u;
    }
    Cs[i] = sec;
  }
  // first index of every entry's group: running maximum of the head indices; thread t owns entries 4t .. 4t+3
  u32 v[4], mx = 0;
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const u32 i = tid * 4 + j;
    v[j] = (i >= lo && i < hi && is_head(i)) ? i + 1 : 0u;
    mx = mx > v[j] ? mx : v[j];
  }
  u32 tot;
  tmx[tid] = block_scan_incl<OpMax>(mx, scan_s, &tot);
  __syncthreads();
  u32 run = tid ? tmx[tid - 1] : 0u;
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const u32 i = tid * 4 + j;
    run = run > v[j] ? run : v[j];
    if (i >= lo && i < hi) gs_s[i] = (u16)(run - 1);
  }
  __syncthreads();
  for (u32 i = lo + tid; i < hi; i += RT_THREADS) {
    if (i > lo && gs_s[i] == i) ge_s[gs_s[i - 1]] = (u16)i;
    if (i + 1 == hi) ge_s[gs_s[i]] = (u16)hi;
  }
  __syncthreads();
  // rank of every entry among its group: stable (equal keys keep the list's order)
  u32 ni[4];
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const u32 i = lo + tid + r * RT_THREADS;
    ni[r] = 0;
    if (i < hi) {
      const u32 s = gs_s[i], t = ge_s[s], my = Cs[i];
      u32 below = 0;
      for (u32 j = s; j < t; j++) {
        const u32 x = Cs[j];
        below += (x < my) || (x == my && j < i);
      }
      ni[r] = s + below;
    }
  }
  __syncthreads();  // the group names in A are dead (the heads live in gs_s)
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const u32 i = lo + tid + r * RT_THREADS;
    if (i < hi) {
      A[ni[r]] = Bp[i];
      Es[ni[r]] = Cs[i];
    }
  }
  __syncthreads();
  // new groups: a head where the group begins or the key changes; a key of 0 (nothing left of the suffix at this
  // depth) is never tied (GroupIn::zero_low_unique)
  auto new_head = [&](u32 i) { return gs_s[i] == i || Es[i] != Es[i - 1] || Es[i] == 0u; };
  mx = 0;
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const u32 i = tid * 4 + j;
    v[j] = (i >= lo && i < hi && new_head(i)) ? i + 1 : 0u;
    mx = mx > v[j] ? mx : v[j];
  }
  tmx[tid] = block_scan_incl<OpMax>(mx, scan_s, &tot);
  __syncthreads();
  run = tid ? tmx[tid - 1] : 0u;
  u32 n_tied = 0;
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const u32 i = tid * 4 + j;
    run = run > v[j] ? run : v[j];
    if (i >= lo && i < hi) {
      const bool head = v[j] != 0;
      const bool next_head = (i + 1 == ge_s[gs_s[i]]) || new_head(i + 1);
      const bool tied = !(head && next_head);
      const u32 p = A[i];
      if (!tied) P.sa[Ds[i]] = p;  // placed for good; a position that is still tied is written by a later round
      P.npos[base + i] = p;
      const u32 name = Ds[run - 1];  // the new group is named by the rank of its first member, as the old one was
      P.ngrp[base + i] = name;
      P.ntied[base + i] = (u8)((u32)tied | ((u32)(name != Ds[gs_s[i]]) << 1));
      n_tied += tied;
    }
  }
  u32 all_tied;
  block_scan_incl<OpSum>(n_tied, scan_s, &all_tied);
  if (tid == 0) P.wg_tied[blockIdx.x] = all_tied;
}

// *out = sum of v[0:n), one workgroup
__global__ void __launch_bounds__(1024) k_sum_u32(const u32 *__restrict__ v, u32 n, u32 *__restrict__ out) {
  __shared__ u32 scan_s[17];
  u32 mine = 0;
  for (u32 i = threadIdx.x; i < n; i += 1024) mine += v[i];
  u32 tot;
  block_scan_incl<OpSum>(mine, scan_s, &tot);
  if (threadIdx.x == 0) *out = tot;
}

// after a round that keeps its list: the inverse array takes the new names (four entries per thread)
__global__ void k_isa_update(const u8 *__restrict__ flags, const u32 *__restrict__ npos, const u32 *__restrict__ ngrp, u32 M,
                             u32 a, u32 *__restrict__ isa) {
  const u32 i0 = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i0 + 4 <= M) {
    const u32 f = *reinterpret_cast<const u32 *>(flags + i0);
    if (f & 0x02020202u) {
      const uint4 p4 = *reinterpret_cast<const uint4 *>(npos + i0), g4 = *reinterpret_cast<const uint4 *>(ngrp + i0);
      if (f & 0x00000002u) isa[p4.x - a] = g4.x;
      if (f & 0x00000200u) isa[p4.y - a] = g4.y;
      if (f & 0x00020000u) isa[p4.z - a] = g4.z;
      if (f & 0x02000000u) isa[p4.w - a] = g4.w;
    }
  } else {
    for (u32 i = i0; i < M; i++)
      if (flags[i] & 2u) isa[npos[i] - a] = ngrp[i];
  }
}

// after k_refine_tile: the inverse array takes the new names, the still-tied entries form the next list
struct TiedIn {
  const u8 *tied;
  __device__ __forceinline__ u32 operator()(size_t i) const { return tied[i] & 1u; }
  __device__ __forceinline__ void vec4(size_t i0, u32 *out) const {
    const u32 f = *reinterpret_cast<const u32 *>(tied + i0);
#pragma unroll
    for (int j = 0; j < 4; j++) out[j] = (f >> (8 * j)) & 1u;
  }
};
struct RefineCompactOut {
  const u8 *flags;
  const u32 *nslot, *npos, *ngrp;
  u32 *isa;
  u32 a;
  u32 *slot, *pos, *grp;
  u32 *total;
  size_t n;
  __device__ __forceinline__ void operator()(size_t i, u32 ex, u32 in) const {
    if (flags[i] & 2u) isa[npos[i] - a] = ngrp[i];  // the first part of a split group keeps its name
    if (in != ex) {
      slot[ex] = nslot[i];
      pos[ex] = npos[i];
      grp[ex] = ngrp[i];
    }
    if (i + 1 == n) *total = in;
  }
  // four consecutive entries: 16-byte loads, and only where something is to be done
  __device__ __forceinline__ void vec4(size_t i0, const u32 *ex, const u32 *in) const {
    const u32 f = *reinterpret_cast<const u32 *>(flags + i0);
    if (f & 0x03030303u) {
      const uint4 p4 = *reinterpret_cast<const uint4 *>(npos + i0), g4 = *reinterpret_cast<const uint4 *>(ngrp + i0);
      const u32 pp[4] = {p4.x, p4.y, p4.z, p4.w}, gg[4] = {g4.x, g4.y, g4.z, g4.w};
      uint4 s4 = make_uint4(0, 0, 0, 0);
      if (f & 0x01010101u) s4 = *reinterpret_cast<const uint4 *>(nslot + i0);
      const u32 ss[4] = {s4.x, s4.y, s4.z, s4.w};
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const u32 fj = f >> (8 * j);
        if (fj & 2u) isa[pp[j] - a] = gg[j];
        if (fj & 1u) {
          slot[ex[j]] = ss[j];
          pos[ex[j]] = pp[j];
          grp[ex[j]] = gg[j];
        }
      }
    }
    if (i0 + 4 == n) *total = in[3];
  }
};

// ----------------------------------------------------------------------------
// ties ordered by comparing the text itself, for sparse and shallow ties.  All or
// nothing: a group of more than DF_GROUP members or a comparison longer than
// DF_DEPTH symbols gives up and the regular rounds take over.  Comparing the
// text gives the full-text order whatever the window is.
// ----------------------------------------------------------------------------
constexpr u32 DF_GROUP = 32;
constexpr u32 DF_DEPTH = 512;

// is T[y:] < T[x:] ?  (x != y)  Positions at or past n read as the empty suffix, the smallest; of two
// exhausted suffixes the shorter is the smaller (TextKeys, zero-based coding).  *decided = false when the
// first DF_DEPTH symbols are equal.  runs (may be null): the text's long periodic stretches (Ctx::runs); where both
// suffixes have come to lie inside stretches of one period the comparison steps to the end of the shorter remainder,
// which does not count as depth (two suffixes that begin just before long runs of one symbol).
__device__ __forceinline__ bool text_less(const u8 *__restrict__ T, u64 n, u64 y, u64 x, bool *decided,
                                          const u32 *__restrict__ runs = nullptr, u32 n_runs = 0) {
  *decided = true;
  u32 d = 0;
  while (d < DF_DEPTH) {
    if (y + d + 8 <= n && x + d + 8 <= n) {  // eight symbols per (unaligned) load
      u64 wy, wx;
      __builtin_memcpy(&wy, T + y + d, 8);
      __builtin_memcpy(&wx, T + x + d, 8);
      if (wy != wx) {
        const int at = __builtin_ctzll(wy ^ wx) & ~7;  // the first symbol is the lowest byte
        return ((wy >> at) & 0xFFu) < ((wx >> at) & 0xFFu);
      }
      d += 8;
      if (n_runs && (d & 63u) == 0) {
        u64 ey = 0, ex = 0;
        const u32 py = run_around(runs, n_runs, y + d, &ey);
        const u32 px = py ? run_around(runs, n_runs, x + d, &ex) : 0u;
        if (py && py == px) {
          const u64 skip = ey - (y + d) < ex - (x + d) ? ey - (y + d) : ex - (x + d);
          y += skip;
          x += skip;
        }
      }
    } else {
      const bool ey = y + d >= n, ex = x + d >= n;
      if (ey || ex) return ey && (!ex || y > x);
      const u32 cy = T[y + d], cx = T[x + d];
      if (cy != cx) return cy < cx;
      d++;
    }
  }
  *decided = false;
  return false;
}

__global__ void k_unpack_pos(const u64 *__restrict__ e, u32 N, u32 a, u32 posmask, u32 *__restrict__ sa) {
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N) sa[i] = a + ((u32)e[i] & posmask);
}

__global__ void k_isa_init(const u32 *__restrict__ sa, u32 a, u32 N, u32 *__restrict__ isa) {
  u32 r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r < N) isa[sa[r] - a] = r;
}

// the same for the suffixes round 0 has placed (the tied ones get their group's name from k_isa_fix: one random
// write per window position instead of two)
template <class View>
__global__ void k_isa_init_untied(ViewTies<View> ties, const u32 *__restrict__ sa, u32 a, u32 N, u32 *__restrict__ isa) {
  u32 r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r < N && ties((size_t)r).cnt == 0) isa[sa[r] - a] = r;
}

__global__ void k_isa_fix(const u32 *__restrict__ pos, const u32 *__restrict__ grp, u32 M, u32 a,
                          u32 *__restrict__ isa) {
  u32 k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < M) isa[pos[k] - a] = grp[k];
}

// ----------------------------------------------------------------------------
// final pass: the leaf's BWT, special ranks and samples straight from the sorted
// window (reference: include/bwt.h:589-609 sa_to_bwt, :294-361 sample_sa)
// ----------------------------------------------------------------------------
struct EmitParams {
  const u8 *T;
  u64 n, a, m;
  i64 ns;            // samples to take (already min(m, ssa_n_samples))
  u8 *L;             // may be null: suffix array only
  const u32 *runs;   // the text's long periodic stretches for the comparisons to step over (may be null)
  u32 n_runs;
  u64 *samp;         // slot y of the position-ordered samples: position << 32 | rank (one store per sample); ~0 = empty
  u32 *special;
};

template <class View>
struct LeafEmit : EmitParams {
  u32 *sa_out;       // may be null
  View v;            // the sorted window
  u64 N;             // window length
  u64 sym_mask;      // symbol field of a key
  u64 h;             // direct_tied(): depth of the round-0 keys
  u32 leaf_end;      // direct_tied(): positions at or beyond it are window slack, not part of the leaf
  u32 *fail;         // direct_tied(): raised when a group is too large or a comparison too long
  // the bucket kernel's way out for a bucket with too many equal keys (null: it raises *fail): the bucket's number
  // goes on this list, *defer_count counts them, and k_bucket_general takes those buckets afterwards
  u32 *defer_list, *defer_count;

  double samp_ratio; // (ns-1)/(m-1), only used to skip positions that are certainly not sampled

  // special ranks and samples of the suffix at `pos`, which has leaf rank r
  // In two parts, because most positions are not sampled and the sampled ones of a wave's lanes rarely sit
  // in the same iteration: note_fast() does the cheap part and says whether note_sample() has to look; a
  // caller with several positions per lane collects those and looks at them afterwards, lanes together.
  __device__ __forceinline__ void note(u64 r, u32 pos) const {
    if (note_fast(r, pos)) note_sample(r, pos);
  }

  __device__ __forceinline__ bool note_fast(u64 r, u32 pos) const {
    const u32 rel32 = pos - (u32)a;  // all positions are below 2^32
    if (rel32 == 0 || rel32 + 1 == (u32)m) {  // the leaf's first or last position: one rare branch for both
      if (rel32 == 0) special[0] = (u32)r;           // s_rank
      if (rel32 + 1 == (u32)m) special[1] = (u32)r;  // e_rank
      // Both ends are always sampled (include/bwt.h:296-299): the first in slot 0, the last in slot ns - 1.  The
      // reference's floor expressions say the same while (b-a)(ns-1) < 2^53; beyond that the product's
      // conversion to double can round the last position's quotient below ns - 1 and lose that sample (an
      // assertion failure there, :347), so the ends are placed here and not by those expressions.
      if (rel32 == 0) samp[0] = ((u64)pos << 32) | (u32)r;
      if (rel32 + 1 == (u32)m && m > 1) samp[ns - 1] = ((u64)pos << 32) | (u32)r;
      return false;
    }
    // A sampled position i lands in slot y = floor((i-a)(ns-1)/(b-a-1)): slots are in position
    // order, so the sample arrays come out sorted by position without a sort.  The two floor()
    // expressions are the reference's own (include/bwt.h:339-340), evaluated in IEEE double as
    // there: 64-bit integer product, one conversion, one correctly rounded division.  Divisions are
    // slow and most positions are not sampled: position i is sampled iff the fractional part of
    // i*(ns-1)/(m-1) is below (ns-1)/(m-1), so one multiplication (error far below the 2^-16
    // margin for any ns < 2^36) rules out the rest.
    // The same product also settles nearly every sampled position: with the fractional part well inside
    // (0, ratio) the position is sampled whatever the roundings of the reference's two divisions do (they move
    // the quotient by far less than the margin) and its slot is the product's integer part; only a fractional part
    // within the margin of 0, of the ratio or of 1 needs the divisions themselves.
    const double q = (double)rel32 * samp_ratio;
    const double f = __builtin_amdgcn_fract(q);
    if (f > samp_ratio + 0x1p-16 && f < 1.0 - 0x1p-16) return false;
    if (f > 0x1p-16 && f < samp_ratio - 0x1p-16) {
      samp[(u32)q] = ((u64)pos << 32) | (u32)r;  // q < ns <= 2^32
      return false;
    }
    return true;
  }

  __device__ __forceinline__ void note_sample(u64 r, u32 pos) const {
    const i64 rel = (i64)(pos - (u32)a);
    i64 A = ns - 1;
    double D = (double)(i64)(m - 1);
    i64 x = (i64)floor((double)((rel - 1) * A) / D);
    i64 y = (i64)floor((double)(rel * A) / D);
    if (x != y) samp[y] = ((u64)pos << 32) | (u32)r;
  }

  __device__ __forceinline__ u8 sym_before(u32 pos) const { return pos ? T[pos - 1] : T[n - 1]; }

  // place() without the sample look-up; true if note_sample(r, pos) is still due
  __device__ __forceinline__ bool place_fast(u64 r, u32 pos, u32 sym) const {
    if (sa_out) sa_out[r] = pos;
    if (!L) return false;
    L[r] = (u8)sym;
    return note_fast(r, pos);
  }

  // the suffix at `pos` has leaf rank r and is preceded by `sym`
  __device__ __forceinline__ void place(u64 r, u32 pos, u32 sym) const {
    if (sa_out) sa_out[r] = pos;
    if (L) {
      L[r] = (u8)sym;
      note(r, pos);
    }
  }

  // The suffix at window rank i, whose key is shared, finds its final window rank by itself: the start of
  // its group of equal keys plus the number of members that are smaller on the text, compared from depth h
  // on.  Key and position belong to the same suffix, so the key's low byte is its BWT symbol wherever it
  // ends up.  slack_before: window-slack positions at ranks below i in the order the sort left.  Raises
  // *fail when it cannot decide: then nothing the pass has written counts.  w: key(i), pos(i) of the
  // window, from wherever they are at hand.
  // (One thread per group, which reads each member's text once instead of once per comparison, is slower:
  // the pass waits on the latency of these reads, not on their number.)
  // lds_L (with the leaf rank of its first byte): where L[r] goes instead of memory, for a caller that stores its
  // stretch of L itself afterwards
  // soft (may be null): where a group that is too large is reported instead of *fail (the caller has another way)
  template <class Window>
  __device__ __forceinline__ void direct_tied(const Window &w, u64 i, u32 slack_before, u8 *lds_L = nullptr, u64 lds_first = 0,
                                              u32 *soft = nullptr) const {
    if (*reinterpret_cast<volatile u32 *>(fail)) return;  // someone gave up already
    const u64 kfull = w.key(i);
    const u64 k = kfull & sym_mask;
    const u32 pos = w.pos(i);
    u64 gs = i, ge = i;
    while (gs > 0 && i - gs < DF_GROUP && (w.key(gs - 1) & sym_mask) == k) gs--;
    while (ge + 1 < N && ge - i < DF_GROUP && (w.key(ge + 1) & sym_mask) == k) ge++;
    if (i - gs >= DF_GROUP || ge - i >= DF_GROUP) {
      if (soft) *soft = 1;
      else *fail = 1;
      return;
    }
    u32 smaller = 0, slack_smaller = 0, slack_ahead = 0;
    const u32 xi = w.extra(i);  // key bits below the ones the window is ordered on (0 where there are none)
    for (u64 j = gs; j <= ge; j++) {
      if (j == i) continue;
      const u32 pj = w.pos(j);
      const bool sl = pj >= leaf_end;
      slack_ahead += (j < i && sl);
      bool decided = true, less;
      const u32 xj = w.extra(j);
      if (xj != xi) less = xj < xi;
      else less = text_less(T, n, (u64)pj + h, (u64)pos + h, &decided, runs, n_runs);
      if (!decided) {
        *fail = 1;
        return;
      }
      smaller += less;
      slack_smaller += (less && sl);
    }
    if (pos >= leaf_end) return;
    const u64 r = gs + smaller - (slack_before - slack_ahead + slack_smaller);
    if (lds_L) {
      if (sa_out) sa_out[r] = pos;
      if (L) {
        lds_L[r - lds_first] = (u8)kfull;
        note(r, pos);
      }
    } else {
      place(r, pos, (u32)kfull & 0xFFu);
    }
  }

  // window rank i holds leaf rank r
  __device__ __forceinline__ void emit(u64 r, u64 i) const {
    u32 pos = v.pos(i);
    if (sa_out) sa_out[r] = pos;
    if (!L) return;
    u64 k = v.key(i);
    u64 top = k & sym_mask;
    bool tied = (i > 0 && (v.key(i - 1) & sym_mask) == top) || (i + 1 < N && (v.key(i + 1) & sym_mask) == top);
    // a suffix that round 0 placed still sits next to its key; a refined one moved, so read the text
    L[r] = tied ? sym_before(pos) : (u8)((u32)k & 0xFFu);
    note(r, pos);
  }

  // BWT symbols of window ranks i0 .. i0+3 (i0 a multiple of four, all below N), from 16-byte loads
  __device__ __forceinline__ void symbols4(u64 i0, const u32 (&pos)[4], u32 (&ch)[4]) const {
    u64 k[6], c[4];
    v.key4(i0, c);
#pragma unroll
    for (int j = 0; j < 4; j++) k[j + 1] = c[j];
    k[0] = i0 ? v.key(i0 - 1) : ~c[0];
    k[5] = (i0 + 4 < N) ? v.key(i0 + 4) : ~c[3];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      u64 top = k[j + 1] & sym_mask;
      bool tied = (k[j] & sym_mask) == top || (k[j + 2] & sym_mask) == top;
      ch[j] = tied ? (u32)sym_before(pos[j]) : ((u32)k[j + 1] & 0xFFu);
    }
  }

  // four window ranks of a window that is the leaf itself (leaf rank == window rank): one 4-byte store of L
  __device__ __forceinline__ void emit4(u64 i0) const {
    u32 pos[4];
    v.pos4(i0, pos);
    if (sa_out) *reinterpret_cast<uint4 *>(sa_out + i0) = make_uint4(pos[0], pos[1], pos[2], pos[3]);
    if (!L) return;
    u32 ch[4];
    symbols4(i0, pos, ch);
#pragma unroll
    for (int j = 0; j < 4; j++) note(i0 + j, pos[j]);
    *reinterpret_cast<u32 *>(L + i0) = ch[0] | (ch[1] << 8) | (ch[2] << 16) | (ch[3] << 24);
  }

  // four window ranks of a longer window: rank[j] is the leaf rank of i0+j where keep[j] says it belongs to the leaf
  __device__ __forceinline__ void emit4_filtered(u64 i0, const u32 (&rank)[4], const bool (&keep)[4]) const {
    u32 pos[4];
    v.pos4(i0, pos);
    u32 ch[4] = {0, 0, 0, 0};
    if (L) symbols4(i0, pos, ch);
#pragma unroll
    for (int j = 0; j < 4; j++) {
      if (!keep[j]) continue;
      if (sa_out) sa_out[rank[j]] = pos[j];
      if (L) {
        L[rank[j]] = (u8)ch[j];
        note(rank[j], pos[j]);
      }
    }
  }
};

template <class View>
__global__ void k_emit_all(LeafEmit<View> E) {
  u64 i0 = ((u64)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i0 + 4 <= E.N) {
    E.emit4(i0);
  } else {
    for (u64 i = i0; i < E.N; i++) E.emit(i, i);
  }
}

// One pass from the sorted window to the leaf's BWT (LeafEmit::direct_tied).  A workgroup takes DE_TILE
// consecutive window ranks: suffixes with a unique key are emitted where they stand; the tied ones are
// gathered into a list in LDS and then taken one per lane, so the dependent text reads of the comparisons
// run in full wavefronts instead of behind the few tied lanes of every wavefront.
// SLACK: the window is longer than the leaf and leaf rank = window rank - slack positions before it;
// tile_slack[t] = slack positions at window ranks below tile t.
constexpr int DE_THREADS = 256;
constexpr int DE_TILE = DE_THREADS * 4;

template <class View>
__global__ void __launch_bounds__(DE_THREADS) k_tile_slack(View v, u64 N, u32 leaf_end, u32 *__restrict__ out) {
  __shared__ u32 scan_s[17];
  const u64 i0 = (u64)blockIdx.x * DE_TILE + threadIdx.x * 4;
  u32 cnt = 0;
  if (i0 + 4 <= N) {
    u32 p[4];
    v.pos4(i0, p);
#pragma unroll
    for (int j = 0; j < 4; j++) cnt += p[j] >= leaf_end;
  } else {
    for (u64 i = i0; i < N; i++) cnt += v.pos(i) >= leaf_end;
  }
  u32 tot;
  block_scan_incl<OpSum>(cnt, scan_s, &tot);
  if (threadIdx.x == 0) out[blockIdx.x] = tot;
}

// a tile's keys and positions in LDS, the rest of the window from memory
template <class View>
struct TileWindow {
  const View &v;
  const u64 *keys;
  const u32 *poss;
  u64 base;
  __device__ __forceinline__ u64 key(u64 i) const { return i - base < (u64)DE_TILE ? keys[i - base] : v.key(i); }
  __device__ __forceinline__ u32 pos(u64 i) const { return i - base < (u64)DE_TILE ? poss[i - base] : v.pos(i); }
  __device__ __forceinline__ u32 extra(u64) const { return 0u; }
};

template <class View, bool SLACK>
__global__ void __launch_bounds__(DE_THREADS) k_direct_emit(LeafEmit<View> E, const u32 *__restrict__ tile_slack) {
  __shared__ __align__(16) u64 tile_key[DE_TILE];
  __shared__ __align__(16) u32 tile_pos[DE_TILE];
  __shared__ u16 tied_list[DE_TILE];
  __shared__ u16 slack_at[SLACK ? DE_TILE : 1];  // slack positions of the tile before each of its tied ranks
  __shared__ u32 scan_s[17];
  const u32 tid = threadIdx.x;
  const u64 N = E.N;
  const u64 base = (u64)blockIdx.x * DE_TILE;
  const u64 i0 = base + tid * 4;
  const bool whole = i0 + 4 <= N;
  u64 full[4], k[6];
  u32 pos[4];
  if (whole) {
    E.v.key4(i0, full);
    E.v.pos4(i0, pos);
  } else {
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const bool ok = i0 + j < N;
      full[j] = ok ? E.v.key(i0 + j) : 0;
      pos[j] = ok ? E.v.pos(i0 + j) : 0;
    }
  }
#pragma unroll
  for (int j = 0; j < 4; j++) {  // ranks beyond the window are never looked at
    tile_key[tid * 4 + j] = full[j];
    tile_pos[tid * 4 + j] = pos[j];
  }
  __syncthreads();
  // k[j + 1]: the symbols of rank i0 + j; a rank outside the window differs from its neighbours
  k[1] = full[0] & E.sym_mask;
  k[0] = (i0 > 0 && i0 - 1 < N) ? ((tid > 0 ? tile_key[tid * 4 - 1] : E.v.key(i0 - 1)) & E.sym_mask) : ~k[1];
#pragma unroll
  for (int j = 0; j < 4; j++) k[j + 1] = (i0 + j < N) ? (full[j] & E.sym_mask) : ~k[j];
  k[5] = (i0 + 4 < N) ? ((tid + 1 < DE_THREADS ? tile_key[tid * 4 + 4] : E.v.key(i0 + 4)) & E.sym_mask) : ~k[4];
  bool tied[4], slk[4];
  u32 n_tied = 0, n_slack = 0;
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const bool ok = i0 + j < N;
    tied[j] = ok && (k[j] == k[j + 1] || k[j + 1] == k[j + 2]);
    slk[j] = SLACK && ok && pos[j] >= E.leaf_end;
    n_tied += tied[j];
    n_slack += slk[j];
  }
  const u32 mine = n_tied | (n_slack << 16);  // either count is at most DE_TILE
  u32 tot;
  const u32 excl = block_scan_incl<OpSum>(mine, scan_s, &tot) - mine;
  const u32 carry = SLACK ? tile_slack[blockIdx.x] : 0u;
  u32 look = 0;  // bit j: the sample look-up of rank i0 + j is still due
  u32 rank_of[4] = {0, 0, 0, 0};
  if (!SLACK && whole && n_tied == 0) {
    if (E.sa_out) *reinterpret_cast<uint4 *>(E.sa_out + i0) = make_uint4(pos[0], pos[1], pos[2], pos[3]);
    if (E.L) {
#pragma unroll
      for (int j = 0; j < 4; j++) {
        rank_of[j] = (u32)(i0 + j);
        if (E.note_fast(i0 + j, pos[j])) look |= 1u << j;
      }
      *reinterpret_cast<u32 *>(E.L + i0) = ((u32)full[0] & 0xFFu) | (((u32)full[1] & 0xFFu) << 8) |
                                           (((u32)full[2] & 0xFFu) << 16) | (((u32)full[3] & 0xFFu) << 24);
    }
  } else {
    u32 t_at = excl & 0xFFFFu, s_at = excl >> 16;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      if (tied[j]) {
        tied_list[t_at++] = (u16)(tid * 4 + j);
        if (SLACK) slack_at[tid * 4 + j] = (u16)s_at;
      } else if (i0 + j < N && !slk[j]) {
        rank_of[j] = (u32)(i0 + j - (carry + s_at));
        if (E.place_fast(rank_of[j], pos[j], (u32)full[j] & 0xFFu)) look |= 1u << j;
      }
      s_at += slk[j];
    }
  }
  while (look) {  // the lanes' look-ups together: usually one pass per wave
    const u32 j = __builtin_ctz(look);
    look &= look - 1;
    const u32 rk = j == 0 ? rank_of[0] : (j == 1 ? rank_of[1] : (j == 2 ? rank_of[2] : rank_of[3]));
    const u32 ps = j == 0 ? pos[0] : (j == 1 ? pos[1] : (j == 2 ? pos[2] : pos[3]));
    E.note_sample(rk, ps);
  }
  __syncthreads();
  const u32 all_tied = tot & 0xFFFFu;
  for (u32 t = tid; t < all_tied; t += DE_THREADS) {
    const u32 at = tied_list[t];
    E.direct_tied(TileWindow<View>{E.v, tile_key, tile_pos, base}, base + at, SLACK ? carry + slack_at[at] : 0u);
  }
}

// ----------------------------------------------------------------------------
// Bucket route (packed layout): the radix passes have stopped after the top 16 key bits.  One workgroup
// loads a bucket, orders it on the remaining key bits in LDS and, instead of writing the sorted
// bucket back for k_direct_emit to read it again, does that kernel's work straight from LDS: unique keys
// are emitted where they stand, tied ones by LeafEmit::direct_tied.  Groups of equal keys never straddle
// buckets.  slack_before[k] = window-slack positions in the buckets below k.
// ----------------------------------------------------------------------------
template <class V>
struct BucketWindow;
template <>
struct BucketWindow<PackedView> {  // the bucket from LDS, anything else (never a match) from memory
  const PackedView &v;
  const u64 *buf;
  const u8 *sbuf;
  u64 s;
  u32 cnt;
  __device__ __forceinline__ u64 key(u64 i) const { return i - s < (u64)cnt ? v.cvt(buf[i - s]) : v.key(i); }
  __device__ __forceinline__ u32 pos(u64 i) const {
    return i - s < (u64)cnt ? v.a + ((u32)buf[i - s] & v.posmask) : v.pos(i);
  }
  __device__ __forceinline__ u32 extra(u64) const { return 0u; }
};
template <>
struct BucketWindow<WideView> {  // key = | the bucket's 16 sorted key bits | preceding symbol, 8 bits |; outside the bucket: no match
  const WideView &v;
  const u64 *buf;
  const u8 *sbuf;
  u64 s;
  u32 cnt;
  __device__ __forceinline__ u64 key(u64 i) const {
    if (i - s >= (u64)cnt) return ~0ull;
    const u32 sd = sbuf[i - s];
    return ((u64)v.bucket_key(buf[i - s], sd) << 8) | (sd & 15u);
  }
  __device__ __forceinline__ u32 pos(u64 i) const { return i - s < (u64)cnt ? (u32)buf[i - s] : 0u; }
  __device__ __forceinline__ u32 extra(u64) const { return 0u; }
};
template <>
struct BucketWindow<LeanView> {  // key = | the bucket's 16 sorted key bits | preceding symbol, 8 bits |; outside the bucket: no match
  const LeanView &v;
  const u64 *buf;
  const u8 *sbuf;
  u64 s;
  u32 cnt;
  __device__ __forceinline__ u64 key(u64 i) const {
    if (i - s >= (u64)cnt) return ~0ull;
    const u64 x = buf[i - s];
    return ((u64)v.bucket_key(x, 0u) << 8) | v.elem_prev(x, 0u);
  }
  __device__ __forceinline__ u32 pos(u64 i) const { return i - s < (u64)cnt ? v.elem_pos(buf[i - s]) : 0u; }
  __device__ __forceinline__ u32 extra(u64 i) const { return v.extra(buf[i - s]); }  // of a member of the bucket
};

// index of the k-th set bit of m (k below its population count)
__device__ __forceinline__ u32 select_bit(u64 m, u32 k) {
  u32 word = (u32)m, base = 0;
  const u32 c = __popc(word);
  if (k >= c) {
    k -= c;
    word = (u32)(m >> 32);
    base = 32;
  }
#pragma unroll
  for (int h = 16; h > 0; h >>= 1) {
    const u32 low = word & ((1u << h) - 1);
    const u32 cl = __popc(low);
    if (k >= cl) {
      k -= cl;
      word >>= h;
      base += h;
    } else {
      word = low;
    }
  }
  return base;
}


// ITEMS: elements per thread; the caller takes the smallest variant that holds the largest bucket, because
// empty element slots still cost their instructions.
template <class V, bool SLACK, int ITEMS>
__global__ void __launch_bounds__(BK_THREADS, 8)  // two workgroups per CU
    k_bucket_emit(LeafEmit<V> E, const u32 *__restrict__ start, const u32 *__restrict__ slack_before) {
  constexpr bool WIDE = V::WIDE;
  __shared__ u8 sbuf[WIDE ? ITEMS * BK_THREADS : 1];  // the side bytes of the sorted bucket
  constexpr int BK_MASKS = ITEMS * BK_WAVES;  // one 64-bit flag word per wave and round
  constexpr int BUF_ELEMS = ITEMS * BK_THREADS > 6144 ? ITEMS * BK_THREADS : 6144;  // the counters need 48 KB
  __shared__ __align__(16) u64 buf[BUF_ELEMS];
  __shared__ u64 tied_mask[BK_MASKS], slack_mask[SLACK ? BK_MASKS : 1];
  __shared__ u32 tied_pre[BK_MASKS + 1], slack_pre[SLACK ? BK_MASKS : 1];
  __shared__ u32 scan_s[4];
  __shared__ u32 soft_s;  // a tied group too large for the direct comparisons: the bucket goes on the list of deferred ones
  // The bucket's stretch of L is collected here and leaves as aligned 16-byte stores (one byte per lane, and none
  // for the tied suffixes in between, left memory with partly written sectors: twice the bytes on the way out).
  // Byte j of the stretch lies at lsym[lead + j], lead = its address modulo 16, so that aligned groups of the array
  // are aligned groups of memory.  Not with window slack (ranks are then not index + constant) and not where the
  // array would cost the second workgroup per CU its LDS (nine elements per thread; eight plus side bytes).
  constexpr bool STAGE_L = !SLACK && ITEMS + (WIDE ? 1 : 0) <= 8;
  __shared__ __align__(16) u8 lsym[STAGE_L ? ITEMS * BK_THREADS + 16 : 16];
  static_assert(BK_MASKS <= 256, "the flag words are scanned by the first four waves");
  const u32 s = start[blockIdx.x], cnt = start[blockIdx.x + 1] - s;  // cnt <= ITEMS * BK_THREADS: the caller has checked
  if (cnt == 0) return;
  const u32 lead = STAGE_L ? (u32)(reinterpret_cast<uintptr_t>(E.L + s) & 15) : 0u;
  const u32 tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const u32 wbase = w * (64 * ITEMS);
  const u64 *mine = E.v.bucket_elems(blockIdx.x, s);
  const u8 *mine_side = nullptr;
  if constexpr (WIDE) mine_side = E.v.side + s;
  {
    // Order of the bucket by the 16 key bits below the bucket's own (zero padding included when the key is
    // shorter).  No order is needed among equal keys -- ties are decided on the text -- so one counting round
    // does: a 4-bit counter per key value (65536 of them, 8 per word; they use the buffer while the elements
    // are in registers), one atomic add per element whose old value numbers the element among its equals,
    // prefix sums of the counters per word, rank = keys below + own number.  A value that occurs more than
    // 15 times gives the direct pass up (the general path can order large groups).
    constexpr u32 NIB_WORDS = 8192;
    static_assert(NIB_WORDS == 8 * BK_THREADS && NIB_WORDS * 4 + NIB_WORDS * 2 <= sizeof(u64) * BUF_ELEMS, "counters fit the buffer");
    u32 *nib = reinterpret_cast<u32 *>(buf);
    u16 *wpre = reinterpret_cast<u16 *>(nib + NIB_WORDS);  // counters in the words below word j
    u64 key[ITEMS];
    u8 sd[WIDE ? ITEMS : 1];
#pragma unroll
    for (int r = 0; r < ITEMS; r++) {
      const u32 idx = wbase + r * 64 + lane;
      key[r] = idx < cnt ? mine[idx] : 0ull;
      if constexpr (WIDE) sd[r] = idx < cnt ? mine_side[idx] : (u8)0;
    }
    reinterpret_cast<uint4 *>(nib)[tid * 2] = make_uint4(0, 0, 0, 0);
    reinterpret_cast<uint4 *>(nib)[tid * 2 + 1] = make_uint4(0, 0, 0, 0);
    if (tid == 0) {
      scan_s[3] = 0;
      soft_s = 0;
    }
    if (tid < BK_MASKS) {
      tied_mask[tid] = 0;
      if (SLACK) slack_mask[tid] = 0;
    }
    lds_barrier();
    u64 ord = 0;  // 4 bits per element: its number among the elements of the same key
    bool full = false;
#pragma unroll
    for (int r = 0; r < ITEMS; r++) {
      if (wbase + r * 64 + lane < cnt) {
        const u32 k = E.v.bucket_key(key[r], WIDE ? sd[r] : 0u), sh = (k & 7) * 4;
        const u32 old = (atomicAdd(&nib[k >> 3], 1u << sh) >> sh) & 15u;
        full = full || old == 15u;
        ord |= (u64)old << (4 * r);
      }
    }
    if (full) scan_s[3] = 1;
    lds_barrier();
    if (scan_s[3]) {  // the same for every thread
      if (tid == 0) {
        if (!SLACK && E.defer_list) E.defer_list[atomicAdd(E.defer_count, 1u)] = blockIdx.x;
        else *E.fail = 1;
      }
      return;
    }
    {
      const uint4 x0 = reinterpret_cast<const uint4 *>(nib)[tid * 2], x1 = reinterpret_cast<const uint4 *>(nib)[tid * 2 + 1];
      const u32 xs[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
      u32 run[8], sum = 0;
#pragma unroll
      for (int j = 0; j < 8; j++) {
        run[j] = sum;
        const u32 y = (xs[j] & 0x0F0F0F0Fu) + ((xs[j] >> 4) & 0x0F0F0F0Fu);
        sum += (y * 0x01010101u) >> 24;
      }
      u32 incl = wave_scan_incl_sum(sum);
      __shared__ u32 wave_sum[BK_WAVES];
      if (lane == 63) wave_sum[w] = incl;
      lds_barrier();
      u32 base = incl - sum;
      for (u32 k = 0; k < w; k++) base += wave_sum[k];
      reinterpret_cast<uint4 *>(wpre)[tid] =
          make_uint4((base + run[0]) | ((base + run[1]) << 16), (base + run[2]) | ((base + run[3]) << 16),
                     (base + run[4]) | ((base + run[5]) << 16), (base + run[6]) | ((base + run[7]) << 16));
    }
    lds_barrier();
    // All elements of the bucket share the 16 key bits above these, and the bits below the symbol codes are
    // zero: two elements have the same key exactly when they share a counter.  So the counter also says
    // whether an element is tied, and the flag words -- bit i: the element at sorted index i is tied / is
    // window slack -- are set from here, by the few elements that are either.
    u32 pos[ITEMS], tied_in = 0;
#pragma unroll
    for (int r = 0; r < ITEMS; r++) {
      const u32 k = E.v.bucket_key(key[r], WIDE ? sd[r] : 0u), sh = (k & 7) * 4;
      const u32 word = nib[k >> 3];
      const u32 below = word & ((1u << sh) - 1u);
      const u32 y = (below & 0x0F0F0F0Fu) + ((below >> 4) & 0x0F0F0F0Fu);
      pos[r] = wpre[k >> 3] + ((y * 0x01010101u) >> 24) + ((u32)(ord >> (4 * r)) & 15u);
      tied_in |= (u32)(((word >> sh) & 15u) > 1u) << r;
    }
    lds_barrier();  // the counters are dead: the buffer takes the elements
    u32 *tied32 = reinterpret_cast<u32 *>(tied_mask), *slack32 = reinterpret_cast<u32 *>(slack_mask);
#pragma unroll
    for (int r = 0; r < ITEMS; r++) {
      if (wbase + r * 64 + lane < cnt) {
        buf[pos[r]] = key[r];
        if constexpr (WIDE) sbuf[pos[r]] = sd[r];
        if ((tied_in >> r) & 1u) atomicOr(&tied32[pos[r] >> 5], 1u << (pos[r] & 31u));
        if (SLACK && E.v.elem_pos(key[r]) >= E.leaf_end) atomicOr(&slack32[pos[r] >> 5], 1u << (pos[r] & 31u));
      }
    }
    lds_barrier();
  }
  // thread t emits the sorted indices r * 1024 + t; flag word r * 16 + w covers those of wave w in round r
  {
    const u32 tc = tid < BK_MASKS ? (u32)__popcll(tied_mask[tid]) : 0u;
    const u32 sc = (SLACK && tid < BK_MASKS) ? (u32)__popcll(slack_mask[tid]) : 0u;
    const u32 both = tc | (sc << 16);  // either total is at most ITEMS * BK_THREADS
    u32 incl = wave_scan_incl_sum(both);
    if (lane == 63 && w < 4) scan_s[w] = incl;
    lds_barrier();
    for (u32 k = 0; k < w && k < 4; k++) incl += scan_s[k];
    if (tid < BK_MASKS) {
      tied_pre[tid] = (incl - both) & 0xFFFFu;
      if (SLACK) slack_pre[tid] = (incl - both) >> 16;
    }
    if (tid == BK_MASKS - 1) tied_pre[BK_MASKS] = incl & 0xFFFFu;
  }
  lds_barrier();
  const u32 below = SLACK ? slack_before[blockIdx.x] : 0u;
  u32 look = 0;  // bit r: the sample look-up of round r is still due
  const u32 *flag32 = reinterpret_cast<const u32 *>(tied_mask), *sflag32 = reinterpret_cast<const u32 *>(slack_mask);
  const u32 my_bit = 1u << (tid & 31u);
#pragma unroll
  for (int r = 0; r < ITEMS; r++) {
    const u32 idx = r * BK_THREADS + tid;
    u32 flags = flag32[idx >> 5];
    if (SLACK) flags |= sflag32[idx >> 5];
    if (idx < cnt && !(flags & my_bit)) {
      const u64 me = buf[idx];
      u32 sb = below;
      if (SLACK) sb += slack_pre[r * BK_WAVES + w] + (u32)__popcll(slack_mask[r * BK_WAVES + w] & lanemask_lt());
      if constexpr (STAGE_L) {
        const u32 pos = E.v.elem_pos(me);
        if (E.sa_out) E.sa_out[(u64)s + idx] = pos;
        if (E.L) {
          lsym[lead + idx] = (u8)E.v.elem_prev(me, WIDE ? sbuf[idx] : 0u);
          if (E.note_fast((u64)s + idx, pos)) look |= 1u << r;
        }
      } else {
        if (E.place_fast((u64)s + idx - sb, E.v.elem_pos(me), E.v.elem_prev(me, WIDE ? sbuf[idx] : 0u))) look |= 1u << r;
      }
    }
  }
  while (look) {  // usually one pass per wave instead of one per round
    const u32 r = __builtin_ctz(look);
    look &= look - 1;
    const u32 idx = r * BK_THREADS + tid;
    const u64 me = buf[idx];
    u32 sb = below;
    if (SLACK) sb += slack_pre[r * BK_WAVES + w] + (u32)__popcll(slack_mask[r * BK_WAVES + w] & lanemask_lt());
    E.note_sample((u64)s + idx - sb, E.v.elem_pos(me));
  }
  const u32 all_tied = tied_pre[BK_MASKS];
  const BucketWindow<V> win{E.v, buf, sbuf, (u64)s, cnt};
  for (u32 t = tid; t < all_tied; t += BK_THREADS) {
    u32 lo = 0, hi = BK_MASKS;  // the last flag word whose prefix is <= t
    while (hi - lo > 1) {
      const u32 mid = (lo + hi) / 2;
      if (tied_pre[mid] <= t) lo = mid;
      else hi = mid;
    }
    // flag word lo = r * 16 + w covers the indices r * 1024 + w * 64 + lane
    const u32 idx = lo * 64 + select_bit(tied_mask[lo], t - tied_pre[lo]);
    u32 sb = below;
    if (SLACK) sb += slack_pre[idx >> 6] + (u32)__popcll(slack_mask[idx >> 6] & ((1ull << (idx & 63)) - 1ull));
    u32 *soft = (!SLACK && E.defer_list) ? &soft_s : nullptr;
    if constexpr (STAGE_L) E.direct_tied(win, (u64)s + idx, sb, lsym + lead, (u64)s, soft);
    else E.direct_tied(win, (u64)s + idx, sb, nullptr, 0, soft);
  }
  const bool may_defer = !SLACK && E.defer_list;
  if (may_defer || (STAGE_L && E.L)) lds_barrier();  // (one barrier for both: the tied suffixes' flag and their bytes of L)
  if (may_defer && soft_s) {  // (what the bucket has written so far is written again, the same or right, by k_bucket_general)
    if (tid == 0) E.defer_list[atomicAdd(E.defer_count, 1u)] = blockIdx.x;
    return;
  }
  if constexpr (STAGE_L) {
    if (!E.L) return;
    u8 *g0 = E.L + s - lead;  // 16-byte aligned
    const u32 end = lead + cnt;
    for (u32 t = tid; t * 16 < end; t += BK_THREADS) {
      const u32 o = t * 16;
      if (o >= lead && o + 16 <= end) {
        *reinterpret_cast<uint4 *>(g0 + o) = *reinterpret_cast<const uint4 *>(lsym + o);
      } else {
        const u32 lo = o < lead ? lead : o, hi = o + 16 < end ? o + 16 : end;
        for (u32 j = lo; j < hi; j++) g0[j] = lsym[j];
      }
    }
  }
}

// A bucket the kernel above has given up on (a key value more than 15 times, a tied group of more than DF_GROUP): all of
// it in LDS, ordered by a bitonic network whose comparison is the elements' key bits first and the text behind them
// second -- any number of equal keys, as long as the comparisons themselves stay short (*E.fail otherwise).  Slow per
// element and meant for the few buckets that need it.
constexpr int BG_THREADS = 1024;
constexpr int BG_CAP = 16384;  // the largest bucket (BK_CAP) rounded up to a power of two

__global__ void __launch_bounds__(BG_THREADS) k_bucket_general(LeafEmit<LeanView> E, const u32 *__restrict__ start, const u32 *__restrict__ list) {
  __shared__ u64 buf[BG_CAP];
  const u32 b = list[blockIdx.x];
  const u32 s = start[b], cnt = start[b + 1] - s;
  const u32 tid = threadIdx.x;
  u32 n2 = 2;
  while (n2 < cnt) n2 <<= 1;
  const u64 *mine = E.v.bucket_elems(b, s);
  for (u32 i = tid; i < n2; i += BG_THREADS) buf[i] = i < cnt ? mine[i] : ~0ull;  // (all ones: no element; they end up last)
  __syncthreads();
  auto less = [&](u64 x, u64 y) -> bool {
    if (x == ~0ull || y == ~0ull) return x != ~0ull && y == ~0ull;
    const u64 kx = E.v.sort_bits(x), ky = E.v.sort_bits(y);
    if (kx != ky) return kx < ky;
    bool decided;
    const bool lt = text_less(E.T, E.n, (u64)E.v.elem_pos(x) + E.h, (u64)E.v.elem_pos(y) + E.h, &decided, E.runs, E.n_runs);
    if (!decided) *E.fail = 1;
    return lt;
  };
  for (u32 k = 2; k <= n2; k <<= 1) {
    for (u32 j = k >> 1; j > 0; j >>= 1) {
      for (u32 i = tid; i < n2; i += BG_THREADS) {
        const u32 p = i ^ j;
        if (p > i) {
          const u64 x = buf[i], y = buf[p];
          const bool up = (i & k) == 0;
          if (up ? less(y, x) : less(x, y)) {
            buf[i] = y;
            buf[p] = x;
          }
        }
      }
      __syncthreads();
    }
  }
  for (u32 i = tid; i < cnt; i += BG_THREADS) {
    const u64 x = buf[i];
    E.place((u64)s + i, E.v.elem_pos(x), E.v.elem_prev(x, 0u));
  }
}

template <class View>
struct LeafFilterIn {
  View v;
  u32 b;
  __device__ __forceinline__ u32 operator()(size_t i) const { return v.pos(i) < b; }
  __device__ __forceinline__ void vec4(size_t i0, u32 *out) const {
    u32 p[4];
    v.pos4(i0, p);
#pragma unroll
    for (int j = 0; j < 4; j++) out[j] = p[j] < b;
  }
};
template <class View>
struct LeafFilterOut {
  LeafEmit<View> E;
  __device__ __forceinline__ void operator()(size_t i, u32 ex, u32 in) const {
    if (in != ex) E.emit(ex, i);
  }
  __device__ __forceinline__ void vec4(size_t i0, const u32 *ex, const u32 *in) const {
    const u32 rank[4] = {ex[0], ex[1], ex[2], ex[3]};
    const bool keep[4] = {in[0] != ex[0], in[1] != ex[1], in[2] != ex[2], in[3] != ex[3]};
    E.emit4_filtered(i0, rank, keep);
  }
};

static u32 read_u32(Ctx &c, const u32 *d) {
  HIP_CHECK(hipMemcpyAsync(c.h_scalars, d, sizeof(u32), hipMemcpyDeviceToHost, c.stream));
  c.sync();
  return c.h_scalars[0];
}

static int bits_for(u64 x) {  // bits needed to hold values 0..x
  int b = 1;
  while (b < 64 && (x >> b)) b++;
  return b;
}

// Does the window hold long repeats?  A bucket route or the direct pass would do their work and then give up on
// such a text, so large windows are probed first: RP_SAMPLES positions, one in every stretch of N / RP_SAMPLES,
// hashed over their next 64 symbols into a table; two positions with the same 64 symbols count as a collision.  A
// text without long repeats has none (64 symbols of even a two-letter alphabet), a text whose repeats would defeat
// the direct pass -- copies of long stretches -- has many: the copies' positions collide whenever both are sampled.
constexpr u32 RP_SAMPLES = 1u << 18;
constexpr u32 RP_TABLE = 1u << 20;

__global__ void k_repeat_probe(const u8 *__restrict__ T, u64 n, u64 a, u64 N, u64 stride, unsigned long long *__restrict__ table,
                               u32 *__restrict__ collisions, int skip_runs) {
  const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= RP_SAMPLES) return;
  u64 x = (t + 1) * 0x9E3779B97F4A7C15ull;  // the position's place inside its stretch
  x ^= x >> 29;
  const u64 p = a + t * stride + (x % stride);
  if (p + 64 > n || p >= a + N) return;
  u64 h = 0xCBF29CE484222325ull, first = 0;
  bool one_symbol = true;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    u64 w;
    __builtin_memcpy(&w, T + p + 8 * k, 8);
    if (k == 0) first = w;
    one_symbol = one_symbol && w == first && w == (w & 0xFFu) * 0x0101010101010101ull;
    h = (h ^ w) * 0x100000001B3ull;
    h ^= h >> 32;
  }
  if (skip_runs && one_symbol) return;  // inside a run of one symbol: those are dealt with apart (RunPlan)
  h |= 1ull;  // 0 marks an empty slot
  u32 slot = (u32)(h >> 40) & (RP_TABLE - 1);
  for (u32 probes = 0; probes < RP_TABLE; probes++) {
    const unsigned long long old = atomicCAS(&table[slot], 0ull, (unsigned long long)h);
    if (old == 0ull) return;
    if (old == h) {
      atomicAdd(collisions, 1u);
      return;
    }
    slot = (slot + 1) & (RP_TABLE - 1);
  }
}

static bool window_has_long_repeats(Ctx &c, i64 a, u64 N, bool skip_runs) {
  if (N < c.bucket_min_n || N < (u64)RP_SAMPLES * 64 || !c.repeat_probe) return false;
  KernelTimer kt(c, "leaf:k_repeat_probe", (double)RP_SAMPLES * 64);
  DBuf<unsigned long long> table(c, RP_TABLE);
  u32 *d_count = c.d_scalars + 13;
  HIP_CHECK(hipMemsetAsync(table.get(), 0, (size_t)RP_TABLE * 8, c.stream));
  HIP_CHECK(hipMemsetAsync(d_count, 0, sizeof(u32), c.stream));
  hipLaunchKernelGGL(k_repeat_probe, dim3(RP_SAMPLES / 256), dim3(256), 0, c.stream, c.T, (u64)c.n, (u64)a, N, N / RP_SAMPLES,
                     table.get(), d_count, skip_runs ? 1 : 0);
  HIP_CHECK(hipGetLastError());
  return read_u32(c, d_count) >= 8;
}

// Layout of the round-0 key (TextKeys): coding, number of symbols K0, pairs or packed elements.  Every 8
// sorted key bits cost one radix pass over the whole window; fewer symbols leave more suffixes tied for the
// refinement, which only touches the tied ones.
//   pairs:  the coding and the fewest passes for which a random text of this alphabet would leave under 5 %
//           tied (below that, ordering the ties by direct comparison is cheaper than another pass).
//   packed: 8-byte elements (codes in the high word, preceding symbol and window-relative position in the
//           low word) move a third less per pass and need at most 4 passes; taken when the position and the
//           preceding symbol fit the low word and the expected ties stay under 15 %.
struct KeyPlan {
  int w, K0, one_based, passes;
  int pbits;  // 0: pairs; otherwise the position bits of a packed element
};

static double expected_ties(double sigma, int K0, u64 N) {
  double distinct = 1;
  for (int j = 0; j < K0 && distinct < 1e30; j++) distinct *= sigma;
  return (double)N / distinct;
}

static KeyPlan plan_keys(int span, int raw_bits, u64 N) {  // span = max_sym - min_sym; raw_bits holds any symbol
  const double sigma = span + 1 < 2 ? 2.0 : (double)(span + 1);
  KeyPlan best;
  memset(&best, 0, sizeof(best));
  double best_tied = 2.0;
  for (int one_based = 1; one_based >= 0; one_based--) {
    const int w = bits_for((u64)(span + one_based));
    if (!one_based && w == bits_for((u64)(span + 1))) continue;  // no bit saved: keep the exact coding
    for (int passes = 1; passes <= 7; passes++) {
      const int K0 = 8 * passes / w;  // <= 56 / w: the low byte is the preceding symbol
      if (K0 < 1) continue;
      const double tied = expected_ties(sigma, K0, N);
      const bool good = tied <= 0.05;
      const bool better = best.K0 == 0 || (good && (best_tied > 0.05 || passes < best.passes)) ||
                          (!good && best_tied > 0.05 && tied < best_tied);
      if (better) {
        best = KeyPlan{w, K0, one_based, passes, 0};
        best_tied = tied;
      }
      if (good) break;  // more passes of this coding cannot be better
    }
  }
  // packed elements: codes <= 32 bits, low word = preceding symbol | position
  const int pbits = bits_for(N - 1);
  if (N >= 2 && pbits + raw_bits <= 32) {
    for (int one_based = 1; one_based >= 0; one_based--) {
      const int w = bits_for((u64)(span + one_based));
      if (!one_based && w == bits_for((u64)(span + 1))) continue;
      for (int passes = 1; passes <= 4; passes++) {
        const int K0 = 8 * passes / w;
        if (K0 < 1) continue;
        if (expected_ties(sigma, K0, N) <= 0.15) {
          if (best.pbits == 0 ? passes <= best.passes : passes < best.passes) best = KeyPlan{w, K0, one_based, passes, pbits};
          break;
        }
      }
    }
  }
  return best;
}

template <class V, bool SLACK>
static void launch_bucket_emit(Ctx &c, const LeafEmit<V> &em, u32 n_buckets, u32 max_bucket, const u32 *start, const u32 *slack_before) {
  const u32 items = ceil_div(max_bucket ? max_bucket : 1, BK_THREADS);
  constexpr u32 MAX_ITEMS = V::WIDE ? 8 : BK_ITEMS;  // a wide bucket also keeps its side bytes in LDS
  if (items > MAX_ITEMS) fail(BWT_HIP_EINTERNAL, "bucket of %u elements", max_bucket);
  const dim3 grid(n_buckets), block(BK_THREADS);
  if (c.timing_level >= 2) {  // a counter, not a timer: which instantiation the bucket sizes selected
    const u32 variant = items <= 4 ? 4 : items;
    char name[64];
    snprintf(name, sizeof(name), "leaf:bucket_emit_items=%u%s%s%s", variant, SLACK ? ",slack" : "", V::WIDE ? ",wide" : "",
             std::is_same_v<V, LeanView> ? ",lean" : "");
    c.timers[(size_t)c.timer_index(name)].launches++;
  }
  if (items <= 4) hipLaunchKernelGGL((k_bucket_emit<V, SLACK, 4>), grid, block, 0, c.stream, em, start, slack_before);
  else if (items == 5) hipLaunchKernelGGL((k_bucket_emit<V, SLACK, 5>), grid, block, 0, c.stream, em, start, slack_before);
  else if (items == 6) hipLaunchKernelGGL((k_bucket_emit<V, SLACK, 6>), grid, block, 0, c.stream, em, start, slack_before);
  else if (items == 7) hipLaunchKernelGGL((k_bucket_emit<V, SLACK, 7>), grid, block, 0, c.stream, em, start, slack_before);
  else if (items == 8) hipLaunchKernelGGL((k_bucket_emit<V, SLACK, 8>), grid, block, 0, c.stream, em, start, slack_before);
  else if constexpr (!V::WIDE) hipLaunchKernelGGL((k_bucket_emit<V, SLACK, BK_ITEMS>), grid, block, 0, c.stream, em, start, slack_before);
  HIP_CHECK(hipGetLastError());
}

// The refinement rounds' state: the active list (rank index, position, group name of every suffix that still shares
// its group; in rank order), the sorted positions they index, the inverse array once doubling has begun.  Positions
// and ranks live in the index space [a, e) of a window; n = where that space ends for good (look-ups past it read as
// the empty suffix), b = end of the part whose ranks must come out exact.
struct RefineRounds {
  u64 a = 0, b = 0, e = 0, n = 0, N = 0;
  u64 h = 0, h_exact = 0;  // depth reached; depth reached by reading the text (exact whatever the window)
  u32 *sa_w = nullptr, *isa = nullptr;
  u32 *wl_slot = nullptr, *wl_pos = nullptr, *wl_grp = nullptr;
  u32 M = 0;
  DBuf<u32> nl_slot, nl_pos, nl_grp;  // owners of the list after the first round
  int round = 0;
  // the step over long runs of one symbol (k_comp_isa): state 0 not in these rounds, 1 in the next round (the first
  // doubling round), 2 from then on
  RunJump J{0, 0, nullptr, 0, 0, nullptr};
  DBuf<u32> run_ends;  // owner of J.ends
};

// One round through the radix passes: the list ordered by (group, secondary key), groups split where the key
// changes.  T != null: a text round (secondary = the next K1 symbols at depth h); else a doubling round.  Returns
// false when a doubling round had to look beyond the window.
static bool radix_round(Ctx &c, RefineRounds &S, const u8 *T, int K1, bool keep_depth = false) {
  u32 *d_total = c.d_scalars + 8, *d_overflow = c.d_scalars + 9;
  const u32 M = S.M;
  const int gbits = bits_for(S.N);
  DBuf<u64> ck0(c, M), ck1(c, M);
  DBuf<u32> cp0(c, M), cp1(c, M);
  // the keys carry the groups as dense indices (the names themselves come back from the slots, RefineOut)
  DBuf<u32> dense(c, M);
  scan_generic<u32, SumU32>(c, M, GroupHeadIn{S.wl_grp}, DenseGroupOut{dense.get(), d_total, M});
  const u32 n_groups = read_u32(c, d_total);
  const int dbits = bits_for(n_groups ? n_groups - 1 : 0);
  if (dbits > gbits) fail(BWT_HIP_EINTERNAL, "refinement: %u groups in a window of %lu", n_groups, (unsigned long)S.N);
  if (T) {
    hipLaunchKernelGGL(k_comp_text, dim3(ceil_div(M, 256)), dim3(256), 0, c.stream, T, S.n, S.wl_pos, (const u32 *)dense.get(), M,
                       (u32)S.h, c.sym_bits, K1, ck0.get(), cp0.get());
  } else {
    KernelTimer kt(c, S.J.state == 1 ? "refine:k_comp_isa_run_lengths" : "refine:k_comp_isa", (double)M * 16);
    hipLaunchKernelGGL(k_comp_isa, dim3(ceil_div(M, 256)), dim3(256), 0, c.stream, (const u32 *)S.isa, S.a, S.b, S.e, S.n, S.h_exact,
                       S.wl_pos, (const u32 *)dense.get(), M, S.h, ck0.get(), cp0.get(), d_overflow, S.J);
  }
  HIP_CHECK(hipGetLastError());
  dense.reset();
  u64 *ck[2] = {ck0.get(), ck1.get()};
  u32 *cp[2] = {cp0.get(), cp1.get()};
  c.sort_tag = "refine:";
  const int cc = radix_sort_u64(c, ck, cp, M, 0, 32 + dbits);
  c.sort_tag = "";
  DBuf<u32> ns(c, M), np(c, M), ng(c, M);
  scan_generic<TieHead, TieHeadOp>(
      c, M, GroupIn<PlainKeys>{PlainKeys{ck[cc]}, M, ~0ull, true},
      RefineOut{S.wl_slot, cp[cc], S.sa_w, S.isa, (u32)S.a, ns.get(), np.get(), ng.get(), d_total, M});
  S.M = read_u32(c, d_total);
  S.nl_slot = std::move(ns);
  S.nl_pos = std::move(np);
  S.nl_grp = std::move(ng);
  S.wl_slot = S.nl_slot.get();
  S.wl_pos = S.nl_pos.get();
  S.wl_grp = S.nl_grp.get();
  if (keep_depth) return true;  // (the round that cuts the groups of long runs: no look-ups, the depth stays)
  S.h = T ? S.h + (u64)K1 : 2 * S.h;
  S.round++;
  if (!T && read_u32(c, d_overflow)) return false;
  if (S.round > 80) fail(BWT_HIP_EINTERNAL, "suffix sort did not converge");
  return true;
}

// Doubling rounds until no suffix shares its group: inside the groups in LDS (k_refine_tile) while the groups are
// short enough, through the radix passes otherwise.  False: a look-up went beyond the window (the caller retries with
// a longer one).
// *out = members of the longest group of the list (a group's members hold consecutive ranks from its name on, so its
// last member knows)
__global__ void k_max_group(const u32 *__restrict__ slot, const u32 *__restrict__ grp, u32 M, u32 *__restrict__ out) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  u32 len = 0;
  if (i < M) {
    const u32 g = grp[i];
    if (i + 1 == M || grp[i + 1] != g) len = slot[i] - g + 1u;
  }
  len = wave_reduce_max(len);
  // (an add per wavefront on one word would take longer than the pass: only a wavefront that beats what is there adds)
  if (lane_id() == 0 && len > __hip_atomic_load(out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(out, len);
}

static bool doubling_rounds(Ctx &c, RefineRounds &S, const char *phase) {
  u32 *d_total = c.d_scalars + 8, *d_overflow = c.d_scalars + 9, *d_fail = c.d_scalars + 10;
  u32 *d_longest = c.d_scalars + 39;
  bool groups_fit = false;  // every group fits a workgroup (groups only get shorter: once true, true for good)
  if (S.J.state == 1 && S.M > 0) {  // long runs of one symbol: their groups are cut first (RunJump)
    Phase ph(c, phase, (double)S.M * 8 * 32);
    S.J.h_jump = S.h;
    radix_round(c, S, nullptr, 0, true);
    S.J.state = 2;
  }
  while (S.M > 0) {
    const u32 M = S.M;
    Phase ph(c, phase, (double)M * 8 * 32);
    // Inside the groups, in LDS, when every group fits a workgroup (one that starts in a workgroup's RT_TILE entries
    // must end within its RT_CAP); through the radix passes otherwise.  A pass over the list says how long the
    // longest group is, until it fits (the kernel checks as well, and raises *fail where a group does not fit).
    if (!groups_fit) {
      HIP_CHECK(hipMemsetAsync(d_longest, 0, sizeof(u32), c.stream));
      hipLaunchKernelGGL(k_max_group, dim3(ceil_div(M, 256)), dim3(256), 0, c.stream, (const u32 *)S.wl_slot, (const u32 *)S.wl_grp, M, d_longest);
      HIP_CHECK(hipGetLastError());
      groups_fit = read_u32(c, d_longest) <= (u32)(RT_CAP - RT_TILE);
    }
    if (groups_fit) {
      DBuf<u32> tp(c, M), tg(c, M);
      DBuf<u8> tt(c, M);
      const u32 n_wg = ceil_div(M, RT_TILE);
      DBuf<u32> wg_tied(c, n_wg);
      HIP_CHECK(hipMemsetAsync(d_fail, 0, sizeof(u32), c.stream));
      HIP_CHECK(hipMemsetAsync(wg_tied.get(), 0, (size_t)n_wg * sizeof(u32), c.stream));
      RefineTileParams P{S.wl_slot, S.wl_pos, S.wl_grp, M, S.isa, S.sa_w, S.a, S.b, S.e, S.n, S.h_exact, S.h,
                         d_overflow, d_fail, tp.get(), tg.get(), tt.get(), wg_tied.get(), S.J};
      {
        KernelTimer kt(c, "refine:k_refine_tile", (double)M * 25);
        hipLaunchKernelGGL(k_refine_tile, dim3(n_wg), dim3(RT_THREADS), 0, c.stream, P);
        HIP_CHECK(hipGetLastError());
      }
      hipLaunchKernelGGL(k_sum_u32, dim3(1), dim3(1024), 0, c.stream, (const u32 *)wg_tied.get(), n_wg, d_total);
      HIP_CHECK(hipGetLastError());
      HIP_CHECK(hipMemcpyAsync(c.h_scalars + 8, c.d_scalars + 8, 3 * sizeof(u32), hipMemcpyDeviceToHost, c.stream));
      c.sync();
      if (c.h_scalars[10] == 0) {  // d_fail
        const bool overflow = c.h_scalars[9] != 0;
        const u64 still = c.h_scalars[8];  // d_total
        if (!overflow && still * 100 > (u64)M * 85) {
          // Nearly all entries are still tied: the list stays as it is (the placed entries are groups of one from now
          // on and cost the next round a look-up each), only the inverse array takes the new names.
          hipLaunchKernelGGL(k_isa_update, dim3(ceil_div(ceil_div(M, 4), 256)), dim3(256), 0, c.stream, (const u8 *)tt.get(),
                             (const u32 *)tp.get(), (const u32 *)tg.get(), M, (u32)S.a, S.isa);
          HIP_CHECK(hipGetLastError());
        } else if (!overflow) {
          DBuf<u32> ns(c, M), np(c, M), ng(c, M);
          scan_generic<u32, SumU32>(c, M, TiedIn{tt.get()},
                                    RefineCompactOut{tt.get(), S.wl_slot, tp.get(), tg.get(), S.isa, (u32)S.a, ns.get(), np.get(), ng.get(),
                                                     d_total, M});
          S.M = read_u32(c, d_total);
          S.nl_slot = std::move(ns);
          tp = std::move(np);
          tg = std::move(ng);
          S.wl_slot = S.nl_slot.get();
        }
        S.nl_pos = std::move(tp);
        S.nl_grp = std::move(tg);
        S.wl_pos = S.nl_pos.get();
        S.wl_grp = S.nl_grp.get();
        S.h = 2 * S.h;
        S.round++;
        if (overflow) return false;
        if (S.round > 80) fail(BWT_HIP_EINTERNAL, "suffix sort did not converge");
        continue;
      }
      groups_fit = false;  // (not expected: the pass above said they fit)
    }
    if (!radix_round(c, S, nullptr, 0)) return false;
  }
  return true;
}

// ----------------------------------------------------------------------------
// Ranks of a difference-cover sample of the whole text (what the reference's README.md:109 names as missing: "a
// suffix array of sampled characters from the whole input to guarantee a single string comparison does not take too
// long").  D is a perfect difference set modulo v: for any two positions i != j there is exactly one d < v with
// (i + d) mod v and (j + d) mod v both in D.  Once the suffixes at the sample positions {p : p mod v in D} are
// ranked among themselves, ANY two suffixes compare in at most v symbols plus one look-up of two ranks -- whatever
// the text repeats.
//
// Ranking the sample is prefix doubling on a text |D| / v as long: sample k (its compact index: block p / v times
// |D| plus the index of p mod v in D) shifted by a multiple of v symbols is sample k + that many |D|, so the rounds
// above run unchanged on the index space [0, n_s) with a first shift of |D| indices = v symbols, after a radix sort
// on the first K >= v symbols.
// ----------------------------------------------------------------------------
constexpr u32 DC_ND_MAX = 6;

struct DcGeom {
  u32 v, nd, n_s;
  u64 dpack;  // residue t of the cover in byte t
  __device__ __host__ __forceinline__ u32 pos(u32 k) const {
    const u32 blk = k / nd;
    return blk * v + (u32)((dpack >> (8 * (k - blk * nd))) & 0xFFu);
  }
};

// The sample's sort keys, in decreasing order of position (TextKeys: with zero-based codes the terminator looks like
// the smallest symbol, and of equal keys the suffixes that ran into it must come first, the shortest first; the
// stable sort then leaves them so): key = codes of K symbols, first symbol highest; value = the sample's index.
__global__ void k_dc_keys(const u8 *__restrict__ T, u64 n, DcGeom g, int w, int K, u32 min_sym, int one_based,
                          u64 *__restrict__ keys, u32 *__restrict__ vals) {
  const u32 e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= g.n_s) return;
  const u32 k = g.n_s - 1 - e;
  const u64 p = g.pos(k);
  u64 key = 0;
  for (int j = 0; j < K; j++) {
    const u32 ch = p + j < n ? T[p + j] : 0u;
    const u32 code = ch ? ch - min_sym + (u32)one_based : 0u;
    key = (key << w) | code;
  }
  keys[e] = key << (64 - K * w);
  vals[e] = k;
}

struct SampleView {
  const u64 *k;
  const u32 *p;
  __device__ __forceinline__ u64 key(size_t i) const { return k[i]; }
  __device__ __forceinline__ u32 pos(size_t i) const { return p[i]; }
  __device__ __forceinline__ void key4(size_t i0, u64 (&o)[4]) const {
#pragma unroll
    for (int j = 0; j < 4; j++) o[j] = k[i0 + j];
  }
  __device__ __forceinline__ void pos4(size_t i0, u32 (&o)[4]) const {
    const uint4 x = *reinterpret_cast<const uint4 *>(p + i0);
    o[0] = x.x;
    o[1] = x.y;
    o[2] = x.z;
    o[3] = x.w;
  }
};

// For a pair of residues (ri, rj) of two positions i, j: the smallest d with (ri + d) mod v and (rj + d) mod v both
// in the cover, and how many sample positions lie in [i, i + d) resp. [j, j + d) -- the slots of i + d and j + d in
// the windows of sample ranks that the two suffixes hold.  d in bits 0..4, slot of i in 5..8, slot of j in 9..12.
static u16 dc_pair_entry(u32 v, const bool *in_cover, u32 ri, u32 rj) {
  for (u32 d = 0; d < v; d++) {
    if (in_cover[(ri + d) % v] && in_cover[(rj + d) % v]) {
      u32 si = 0, sj = 0;
      for (u32 t = 0; t < d; t++) {
        si += in_cover[(ri + t) % v];
        sj += in_cover[(rj + t) % v];
      }
      return (u16)(d | (si << 5) | (sj << 9));
    }
  }
  fail(BWT_HIP_EINTERNAL, "difference cover modulo %u does not cover (%u, %u)", v, ri, rj);
}

void drop_dc_sample(Ctx &c) {
  if (c.dc.rank) c.release(c.dc.rank);
  if (c.dc.pair_tab) c.release(c.dc.pair_tab);
  c.dc = Ctx::DcSample();
}

void ensure_dc_sample(Ctx &c) {
  if (c.dc.valid) return;
  if (!c.T) fail(BWT_HIP_EINVAL, "no text set");
  c.wait_text((u64)c.n);
  Phase ph(c, "dc_sample", (double)c.n);
  drop_dc_sample(c);
  // symbol codes as the leaf sort's keys have them (TextKeys): zero-based when the alphabet has 2^k symbols
  const u32 min_sym = (u32)(c.min_sym > 0 ? c.min_sym : 1);
  const int span = c.max_sym > (int)min_sym ? c.max_sym - (int)min_sym : 0;
  const bool pow2 = ((u32)(span + 1) & (u32)span) == 0;
  const int one_based = pow2 ? 0 : 1;
  const int w = bits_for((u64)(span + one_based));
  const int K = 64 / w;  // symbols of a sort key: the cover's period must not exceed it
  // perfect difference sets: every non-zero residue is the difference of exactly one pair
  static const struct { u32 v, nd, D[DC_ND_MAX]; } covers[] = {
      {31, 6, {0, 1, 3, 8, 12, 18}}, {21, 5, {0, 1, 4, 14, 16}}, {13, 4, {0, 1, 3, 9}}, {7, 3, {0, 1, 3}}};
  int pick = 3;
  for (int t = 3; t >= 0; t--)
    if ((int)covers[t].v <= K) pick = t;
  Ctx::DcSample &dc = c.dc;
  dc.v = covers[pick].v;
  dc.nd = covers[pick].nd;
  DcGeom g;
  g.v = dc.v;
  g.nd = dc.nd;
  g.dpack = 0;
  bool in_cover[32] = {};
  for (u32 t = 0; t < dc.nd; t++) {
    dc.D[t] = covers[pick].D[t];
    g.dpack |= (u64)dc.D[t] << (8 * t);
    in_cover[dc.D[t]] = true;
  }
  const u64 n = (u64)c.n;
  const u64 last_blk = (n - 1) / dc.v;
  u64 n_s = last_blk * dc.nd;
  for (u32 t = 0; t < dc.nd; t++) n_s += (last_blk * dc.v + dc.D[t] < n);
  dc.n_s = (u32)n_s;
  g.n_s = dc.n_s;
  // the tables of the pairwise comparison
  {
    std::vector<u16> tab((size_t)dc.v * dc.v + (dc.v + 1) / 2 + 1);
    for (u32 ri = 0; ri < dc.v; ri++)
      for (u32 rj = 0; rj < dc.v; rj++) tab[(size_t)ri * dc.v + rj] = dc_pair_entry(dc.v, in_cover, ri, rj);
    u8 *lo = reinterpret_cast<u8 *>(tab.data() + (size_t)dc.v * dc.v);
    for (u32 r = 0; r < dc.v; r++) {
      u32 below = 0;
      for (u32 t = 0; t < dc.nd; t++) below += dc.D[t] < r;
      lo[r] = (u8)below;
    }
    dc.pair_tab = (u16 *)c.alloc(tab.size() * sizeof(u16));
    dc.lo_tab = reinterpret_cast<u8 *>(dc.pair_tab + (size_t)dc.v * dc.v);
    HIP_CHECK(hipMemcpyAsync(dc.pair_tab, tab.data(), tab.size() * sizeof(u16), hipMemcpyHostToDevice, c.stream));
    c.sync();  // `tab` is on the stack
  }
  const size_t N = dc.n_s;
  u32 *d_total = c.d_scalars + 8;
  DBuf<u32> isa(c, N + 8);
  if (N == 1) {
    HIP_CHECK(hipMemsetAsync(isa.get(), 0, sizeof(u32), c.stream));
  } else {
    DBuf<u64> kb0(c, N), kb1(c, N);
    DBuf<u32> vb0(c, N), vb1(c, N);
    {
      KernelTimer kt(c, "dc:k_dc_keys", (double)N * (K + 12));
      hipLaunchKernelGGL(k_dc_keys, dim3(ceil_div(N, 256)), dim3(256), 0, c.stream, c.T, n, g, w, K, min_sym, one_based, kb0.get(),
                         vb0.get());
      HIP_CHECK(hipGetLastError());
    }
    u64 *keys[2] = {kb0.get(), kb1.get()};
    u32 *vals[2] = {vb0.get(), vb1.get()};
    const int low_bit = 64 - K * w;
    c.sort_tag = "dc:";
    const int cur = radix_sort_u64(c, keys, vals, N, low_bit & ~7, 64);
    c.sort_tag = "";
    const u64 sym_mask = ~0ull << low_bit;
    SampleView sv{keys[cur], vals[cur]};
    DBuf<u32> wl_slot(c, N), wl_pos(c, N), wl_grp(c, N);
    scan_generic<TieHead, TieHeadOp>(c, N, ViewTies<SampleView>{sv, N, sym_mask},
                                     FirstListOut<SampleView>{sv, wl_slot.get(), wl_pos.get(), wl_grp.get(), d_total, N});
    RefineRounds R;
    R.a = 0;
    R.b = R.e = R.n = R.N = N;
    R.h = dc.nd;  // a shift of |D| indices is a shift of v <= K symbols: the depth the keys have reached
    R.h_exact = R.h;
    R.sa_w = vals[cur];
    R.isa = isa.get();
    R.wl_slot = wl_slot.get();
    R.wl_pos = wl_pos.get();
    R.wl_grp = wl_grp.get();
    R.M = read_u32(c, d_total);
    hipLaunchKernelGGL(k_isa_init_untied<SampleView>, dim3(ceil_div(N, 256)), dim3(256), 0, c.stream, ViewTies<SampleView>{sv, N, sym_mask},
                       (const u32 *)vals[cur], 0u, (u32)N, isa.get());
    if (R.M > 0) hipLaunchKernelGGL(k_isa_fix, dim3(ceil_div(R.M, 256)), dim3(256), 0, c.stream, R.wl_pos, R.wl_grp, R.M, 0u, isa.get());
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipMemsetAsync(c.d_scalars + 9, 0, sizeof(u32), c.stream));  // the rounds' overflow flag (an earlier sort may have left it raised)
    if (!doubling_rounds(c, R, "dc_doubling_round")) fail(BWT_HIP_EINTERNAL, "difference-cover sample: a look-up beyond the text");
    c.sync();  // the sort's buffers die here
  }
  dc.rank = isa.get();
  isa.p = nullptr;  // kept by the context until the text goes
  dc.valid = true;
}

// One pass from the window sorted on its first K0 symbols to the leaf's BWT, ties settled by the sample ranks.  The
// window is the leaf itself (no slack: the ranks are those of the whole text).  A workgroup takes the groups of equal
// keys that START in its DT_TILE window ranks (a group's members are consecutive), at most DT_CAP entries.  A suffix
// alone in its group is final where it stands.  A member of a longer group loads what any comparison with another
// member can need -- the ranks of the |D| sample positions in [p, p + v) and the symbols K0 .. v-2 behind p, packed --
// and counts the members below it: for residues (p mod v, q mod v) a table gives the one shift d < v that makes
// both p + d and q + d sample positions; the suffixes agree on their first K0 symbols, compare on the symbols K0 .. d-1
// from the packed words, and then on the two ranks.  A group that does not fit raises *fail (the caller has checked
// the largest group beforehand).
constexpr int DT_THREADS = 512;
constexpr int DT_TILE = 512;       // window ranks whose groups a workgroup owns: groups of up to DT_CAP - DT_TILE + 1 members fit
constexpr int DT_TILE_WIDE = 768;  // the same where no group has more than DT_CAP - DT_TILE_WIDE + 1 members
constexpr int DT_CAP = 1024;
constexpr int DT_EPT = DT_CAP / DT_THREADS;  // entries per thread

// XW: 8-byte words of text read per member (the symbols K0 .. K0 + 8 XW - 1 as they are in the text; first symbol =
// lowest byte; 0 where the keys already cover v - 1 symbols).  PK: the nx symbols that count are packed into one word
// as codes of xbits bits (symbol - min_sym + 1; 0 = the terminator and beyond), first symbol highest, above six bits
// that hold the position mod v and the "ends inside its key" flag: a pair then costs one 8-byte look-up for the other
// member's symbols and flags, one shift and one compare for their common prefix.  Needs nx * xbits <= 58.
template <class View, int XW, bool PK>
__global__ void __launch_bounds__(DT_THREADS) k_dc_emit(LeafEmit<View> E, DcView dc, u32 K0, u32 nx, u32 xbits, u32 min_sym, u32 tile) {
  constexpr int XS = PK ? 1 : XW;         // words staged per member
  __shared__ u32 Rw[DT_CAP * DC_ND_MAX];  // first the entries' keys (two words each), then the windows of sample ranks
  __shared__ u64 Xw[XS ? DT_CAP * XS : 1];  // first the head scan's partial results
  __shared__ u32 Bp[DT_CAP];              // positions
  __shared__ u16 gs_s[DT_CAP], ge_s[DT_CAP];
  __shared__ u16 meta[DT_CAP];  // bits 0..4 position mod v; bit 5: the suffix ends inside its key; bits 8..15: preceding symbol
  __shared__ u16 tab_s[31 * 31];
  __shared__ u32 scan_s[17];
  __shared__ u32 bounds[2];
  __shared__ u32 tmx_own[XS ? 1 : DT_THREADS];
  static_assert(DT_CAP == DT_EPT * DT_THREADS && DC_ND_MAX * 4 >= 8 && (!PK || XW > 0), "the keys fit the rank windows");
  const u32 tid = threadIdx.x;
  const u64 base = (u64)blockIdx.x * tile;
  const u32 cnt = E.N - base < (u64)DT_CAP ? (u32)(E.N - base) : (u32)DT_CAP;
  u64 *keyb = reinterpret_cast<u64 *>(Rw);
  u32 *tmx = XS ? reinterpret_cast<u32 *>(Xw) : tmx_own;
  const u64 prev_key = base ? (E.v.key(base - 1) & E.sym_mask) : 0ull;
  if (tid < 2) bounds[tid] = DT_CAP;
  for (u32 i = tid; i < dc.v * dc.v; i += DT_THREADS) tab_s[i] = dc.pair_tab[i];
  for (u32 i = tid; i < cnt; i += DT_THREADS) {
    const u64 k = E.v.key(base + i);
    keyb[i] = k & E.sym_mask;
    meta[i] = (u16)(((u32)k & 0xFFu) << 8);
    Bp[i] = E.v.pos(base + i);
  }
  __syncthreads();
  auto is_head = [&](u32 i) { return (base + i == 0) || keyb[i] != (i ? keyb[i - 1] : prev_key); };
  for (u32 i = tid; i < cnt; i += DT_THREADS)
    if (is_head(i)) atomicMin(&bounds[i < tile ? 0 : 1], i);
  __syncthreads();
  const u32 lo = bounds[0];
  if (lo == (u32)DT_CAP) return;  // no group starts here: the stretch belongs to a group of an earlier workgroup
  u32 hi = bounds[1];
  if (hi == (u32)DT_CAP) {
    if (base + cnt == E.N) {
      hi = cnt;
    } else {
      if (tid == 0) *E.fail = 1;  // the last group runs past what this workgroup holds
      return;
    }
  }
  // first index of every entry's group: running maximum of the head indices; thread t owns entries EPT t .. EPT t + EPT - 1
  u32 hv[DT_EPT], mx = 0;
#pragma unroll
  for (int j = 0; j < DT_EPT; j++) {
    const u32 i = tid * DT_EPT + j;
    hv[j] = (i >= lo && i < hi && is_head(i)) ? i + 1 : 0u;
    mx = mx > hv[j] ? mx : hv[j];
  }
  u32 tot;
  tmx[tid] = block_scan_incl<OpMax>(mx, scan_s, &tot);
  __syncthreads();
  u32 run = tid ? tmx[tid - 1] : 0u;
#pragma unroll
  for (int j = 0; j < DT_EPT; j++) {
    const u32 i = tid * DT_EPT + j;
    run = run > hv[j] ? run : hv[j];
    if (i >= lo && i < hi) gs_s[i] = (u16)(run - 1);
  }
  __syncthreads();
  for (u32 i = lo + tid; i < hi; i += DT_THREADS) {
    if (i > lo && gs_s[i] == i) ge_s[gs_s[i - 1]] = (u16)i;
    if (i + 1 == hi) ge_s[gs_s[i]] = (u16)hi;
  }
  __syncthreads();  // the keys and the scan's partial results are dead: their room takes the rank windows and the text
  // Suffixes alone in their group are final; the others load what the comparisons need.  All of a thread's loads are
  // issued before the first is used: they are random reads, and the pass lives on how many are in flight.
  const u8 *__restrict__ T = E.T;
  u32 rk[DT_EPT][DC_ND_MAX];
  u64 xw[DT_EPT][XW ? XW : 1];
  bool tied[DT_EPT];
#pragma unroll
  for (int r = 0; r < DT_EPT; r++) {
    const u32 i = lo + tid + r * DT_THREADS;
    tied[r] = false;
    if (i >= hi) continue;
    const u32 s = gs_s[i];
    tied[r] = ge_s[s] - s > 1;
    if (!tied[r]) continue;
    const u32 p = Bp[i];
    const u32 blk = p / dc.v, res = p - blk * dc.v;
    meta[i] = (u16)((meta[i] & 0xFF00u) | res | ((u64)p + K0 >= E.n ? 0x20u : 0u));
    const u32 k0 = blk * dc.nd + dc.lo_tab[res];
#pragma unroll
    for (u32 x = 0; x < DC_ND_MAX; x++) rk[r][x] = (x < dc.nd && k0 + x < dc.n_s) ? dc.rank[k0 + x] : 0u;
    if constexpr (XW > 0) {
      const u64 q = (u64)p + K0;
      if (q + 8 * XW <= E.n) {
#pragma unroll
        for (int x = 0; x < XW; x++) __builtin_memcpy(&xw[r][x], T + q + 8 * x, 8);
      } else {  // the end of the text: symbol by symbol, nothing beyond it
#pragma unroll
        for (int x = 0; x < XW; x++) {
          u64 word = 0;
          for (u32 y = 0; y < 8; y++)
            if (q + 8 * x + y < E.n) word |= (u64)T[q + 8 * x + y] << (8 * y);
          xw[r][x] = word;
        }
      }
    }
  }
  u64 pk[DT_EPT];  // PK: this thread's own packed words
#pragma unroll
  for (int r = 0; r < DT_EPT; r++) {
    const u32 i = lo + tid + r * DT_THREADS;
    pk[r] = 0;
    if (i >= hi) continue;
    if (!tied[r]) {
      E.place(base + i, Bp[i], (u32)meta[i] >> 8);
      continue;
    }
#pragma unroll
    for (u32 x = 0; x < DC_ND_MAX; x++) Rw[i * DC_ND_MAX + x] = rk[r][x];
    if constexpr (PK) {
      u64 word = meta[i] & 0x3Fu;
      u32 at = 64;
#pragma unroll
      for (int x = 0; x < 8 * XW; x++) {
        if ((u32)x < nx) {
          const u32 ch = (u32)(xw[r][x >> 3] >> (8 * (x & 7))) & 0xFFu;
          at -= xbits;
          word |= (u64)(ch ? ch - min_sym + 1u : 0u) << at;
        }
      }
      pk[r] = word;
      Xw[i] = word;
    } else if constexpr (XW > 0) {
#pragma unroll
      for (int x = 0; x < XW; x++) Xw[i * XW + x] = xw[r][x];
    }
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < DT_EPT; r++) {
    const u32 i = lo + tid + r * DT_THREADS;
    if (i >= hi || !tied[r]) continue;
    const u32 s = gs_s[i], t = ge_s[s];
    const u32 mi = meta[i], ri = mi & 31u, pi = Bp[i];
    const bool short_i = (mi & 0x20u) != 0;
    const u16 *row = tab_s + ri * dc.v;
    const u32 *mine = Rw + i * DC_ND_MAX;
    u32 below = 0;
    // No branch on the pair inside the loop (its iterations then overlap their LDS look-ups): the ranks decide unless the
    // staged symbols differ before the shift d.  Entry i against itself comes out "not below" by the same arithmetic.
#pragma unroll 4
    for (u32 j = s; j < t; j++) {
      u32 mj;
      u64 xj = 0;
      if constexpr (PK) {
        xj = Xw[j];
        mj = (u32)xj & 0x3Fu;
      } else {
        mj = meta[j];
      }
      const u32 e = row[mj & 31u];
      const u32 d = e & 31u;
      bool less = Rw[j * DC_ND_MAX + ((e >> 9) & 15u)] < mine[(e >> 5) & 15u];  // is the suffix of entry j below that of entry i?
      const u32 nb = d > K0 ? d - K0 : 0u;  // the symbols K0 .. d-1 have to agree before the ranks count
      if constexpr (PK) {
        const u32 sh = 64u - nb * xbits;  // at least 6
        const u64 a = nb ? pk[r] >> sh : 0ull, b = nb ? xj >> sh : 0ull;
        less = b < a || (b == a && less);
      } else if constexpr (XW > 0) {
        u32 fd = 8 * XW;     // first byte in which the two differ
        u64 da = 0, db = 0;  // the words that hold it
#pragma unroll
        for (int x = XW - 1; x >= 0; x--) {
          const u64 a = xw[r][x], b = Xw[j * XW + x];
          if (a != b) {
            fd = 8 * x + ((u32)__builtin_ctzll(a ^ b) >> 3);
            da = a;
            db = b;
          }
        }
        if (fd < nb) less = ((db >> (8 * (fd & 7u))) & 0xFFu) < ((da >> (8 * (fd & 7u))) & 0xFFu);
      }
      if (__builtin_expect(((mj | mi) & 0x20u) != 0, 0)) {
        // a suffix that ends inside its key is below every longer one with the same key; of two such the shorter
        less = (mj & 0x20u) != 0 && (!short_i || Bp[j] > pi);
      }
      below += less;
    }
    E.place(base + s + below, pi, mi >> 8);
  }
}

// What the pass above needs to know of the sorted window beforehand, without the list of tied suffixes that the general
// path builds: out[0] += suffixes that share their key with a neighbour; out[1] = 1 if a group has more than `limit`
// members (in sorted order that is: some key equals the key `limit` ranks further on), out[2] likewise for `limit2`.
constexpr int TS_PER_THREAD = 32;  // (one atomic per workgroup on one address: few workgroups)
template <class View>
__global__ void __launch_bounds__(256) k_tie_stats(View v, u64 N, u64 mask, u32 limit, u32 limit2, u32 *__restrict__ out) {
  __shared__ u32 scan_s[17];
  const u64 first = (u64)blockIdx.x * (256 * TS_PER_THREAD) + threadIdx.x;
  u32 tied = 0;
  for (int r = 0; r < TS_PER_THREAD; r++) {
    const u64 i = first + (u64)r * 256;
    if (i >= N) break;
    const u64 k = v.key(i) & mask;
    const bool t = (i > 0 && (v.key(i - 1) & mask) == k) || (i + 1 < N && (v.key(i + 1) & mask) == k);
    if (t && i + limit < N && (v.key(i + limit) & mask) == k) out[1] = 1;
    if (t && i + limit2 < N && (v.key(i + limit2) & mask) == k) out[2] = 1;
    tied += t;
  }
  u32 tot;
  block_scan_incl<OpSum>(tied, scan_s, &tot);
  if (threadIdx.x == 0 && tot) atomicAdd(&out[0], tot);
}

// Is the pass above the way to finish this window?  M tied suffixes of N, the largest group, K0 symbols in the keys.
static bool dc_pass_applies(Ctx &c, u64 N, u32 M, u32 max_group, int K0, bool repeats) {
  if (c.dc_mode <= 0 || M == 0) return false;
  if (max_group > (u32)(DT_CAP - DT_TILE + 1)) return false;  // a group that may not fit a workgroup
  if (c.dc_mode >= 2) return true;
  // dense ties (every round of the general path would go over most of the window) or a text known to repeat itself
  return c.dc.valid || repeats || (u64)M * 8 >= N;
}

// marks[j] = j + 1 where a run of one symbol ends at window index i = N - 1 - j (the symbol behind it differs, or the
// text ends), 0 elsewhere: a maximum scan over j then says, for every i, where its run ends (RunJump::ends)
__global__ void k_run_end_marks(const u8 *__restrict__ T, u64 a, u64 N, u64 n, u32 *__restrict__ marks) {
  const u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= N) return;
  const u64 p = a + (N - 1 - j);
  marks[j] = (p + 1 >= n || T[p] != T[p + 1]) ? (u32)(j + 1) : 0u;
}

// *ext = how many positions from x0 on hold the symbol sym (x0 < n; the terminator's position never does).  One
// workgroup: 2048 positions a step, and after every step that found nothing else a look at the list of long runs,
// which says where the one around the position ends at the earliest (the list only shortens the way).
__global__ void __launch_bounds__(256) k_run_tail(const u8 *__restrict__ T, u64 n, u64 x0, u32 sym, const u32 *__restrict__ runs,
                                                  u32 n_runs, u32 *__restrict__ ext) {
  __shared__ u32 first_s;
  u64 p = x0;
  for (;;) {
    if (threadIdx.x == 0) first_s = 2048;
    __syncthreads();
    const u64 mine = p + 8 * threadIdx.x;
    u32 first = 8;
    for (int j = 7; j >= 0; j--)
      if (!(mine + j + 1 < n && T[mine + j] == sym)) first = (u32)j;
    if (first < 8) atomicMin(&first_s, 8 * threadIdx.x + first);
    __syncthreads();
    const u32 found = first_s;
    __syncthreads();
    if (found < 2048) {
      if (threadIdx.x == 0) *ext = (u32)(p + found - x0);
      return;
    }
    p += 2048;
    u64 end = 0;
    if (n_runs && run_around(runs, n_runs, p, &end) == 1 && end > p + 16) p = end - 8;
  }
}

// Everything after the round-0 sort: tie list, refinement, emit.  sa_w: the writable positions (pairs
// layout) or null; e_w: the writable elements (packed layout) or null.  scratch: 2N free 32-bit words for
// the first active list; grp_buf: N more words (pairs: the unused value buffer).
// dc_state: 1 = the window is the leaf itself on purpose, ties are to be settled by the difference-cover sample ranks;
// 0 = undecided; -1 = not by them.  *redo (on a false return): REDO_DC = sort the leaf alone and come back with
// dc_state 1; REDO_NO_DC = the sample ranks cannot finish this window, come back with dc_state -1; 0 = the window was
// too short (the caller retries with a longer one).
constexpr int REDO_DC = 1, REDO_NO_DC = 2;
template <class View>
static bool finish_window(Ctx &c, i64 a, i64 b, i64 e, u64 N, int K0, u64 sym_mask, View v, u32 *sa_w, u64 *e_w,
                          u32 *scratch, u32 *grp_buf, const EmitParams *E, DBuf<u32> *sa_out, int dc_state, int *redo,
                          BucketState *bs = nullptr, const TextKeys *tk = nullptr, bool skip_direct = false) {
  const int K1 = 32 / c.sym_bits;  // the text rounds read the symbols as they are
  u32 *d_total = c.d_scalars + 8;
  u32 *d_overflow = c.d_scalars + 9;
  u32 *d_fail = c.d_scalars + 10;
  HIP_CHECK(hipMemsetAsync(d_overflow, 0, sizeof(u32), c.stream));
  DBuf<u32> res;
  if (sa_out) res = DBuf<u32>(c, (size_t)(b - a));
  if (!skip_direct) {
    // First try to finish in one pass: ties ordered by direct comparison and everything emitted at once.
    // It gives up as a whole on large tie groups or long common prefixes (repetitive texts); what it wrote
    // until then is overwritten by the general path below.
    Phase ph(c, "leaf_direct_emit", (double)N * (e_w ? 8 : 12) + (double)(b - a) * 2);
    LeafEmit<View> em;
    memset(&em, 0, sizeof(em));
    if (E) static_cast<EmitParams &>(em) = *E;
    em.sa_out = sa_out ? res.get() : nullptr;
    em.v = v;
    em.N = N;
    em.sym_mask = sym_mask;
    em.samp_ratio = em.m > 1 ? (double)(em.ns - 1) / (double)(i64)(em.m - 1) : 1.0;
    em.h = (u64)K0;
    em.leaf_end = (u32)b;
    em.fail = d_fail;
    HIP_CHECK(hipMemsetAsync(d_fail, 0, sizeof(u32), c.stream));
    em.T = c.T;
    em.n = (u64)c.n;
    const u32 tiles = (u32)ceil_div(N, DE_TILE);
    bool from_buckets = false;
    if constexpr (std::is_same_v<View, PackedView>) {
      if (bs && bs->pending) {
        from_buckets = true;
        KernelTimer kt(c, "leaf:k_bucket_emit", (double)N * 8 + (double)(b - a));
        if (e == b) {
          launch_bucket_emit<PackedView, false>(c, em, BK_BUCKETS, bs->max_bucket, bs->start.get(), nullptr);
        } else {
          DBuf<u32> sb(c, BK_BUCKETS);
          HIP_CHECK(hipMemsetAsync(sb.get(), 0, BK_BUCKETS * sizeof(u32), c.stream));
          bucket_histogram(c, *tk, (u64)b, (u64)(e - b), 16, sb.get());
          scan_generic<u32, SumU32>(c, BK_BUCKETS, ArrayIn<u32>{sb.get()}, ArrayOutExcl<u32>{sb.get(), nullptr, BK_BUCKETS});
          launch_bucket_emit<PackedView, true>(c, em, BK_BUCKETS, bs->max_bucket, bs->start.get(), sb.get());
        }
      }
    }
    if (from_buckets) {
    } else if (e == b) {
      hipLaunchKernelGGL((k_direct_emit<View, false>), dim3(tiles), dim3(DE_THREADS), 0, c.stream, em,
                         (const u32 *)nullptr);
      HIP_CHECK(hipGetLastError());
    } else {
      DBuf<u32> ts(c, tiles);
      hipLaunchKernelGGL(k_tile_slack<View>, dim3(tiles), dim3(DE_THREADS), 0, c.stream, v, N, (u32)b, ts.get());
      HIP_CHECK(hipGetLastError());
      scan_generic<u32, SumU32>(c, tiles, ArrayIn<u32>{ts.get()}, ArrayOutExcl<u32>{ts.get(), nullptr, tiles});
      hipLaunchKernelGGL((k_direct_emit<View, true>), dim3(tiles), dim3(DE_THREADS), 0, c.stream, em,
                         (const u32 *)ts.get());
      HIP_CHECK(hipGetLastError());
    }
    if (read_u32(c, d_fail) == 0) {
      if (sa_out) *sa_out = std::move(res);
      return true;
    }
    if (from_buckets) {  // the general path wants the whole window sorted in memory
      bucket_sort_in_place(c, e_w, N, *bs);
      bs->pending = false;
    }
  }
  u32 *wl_slot = scratch;
  u32 *wl_pos = scratch + N;
  u32 *wl_grp = grp_buf;
  u32 M, max_group;
  if (dc_state == 1 && c.dc_mode > 0) {
    // the window was sorted for the sample ranks: how many ties there are and whether every group fits is all that
    // pass needs to know
    Phase ph(c, "leaf_group", (double)N * 8);
    u32 *d_stats = c.d_scalars + 26;
    HIP_CHECK(hipMemsetAsync(d_stats, 0, 3 * sizeof(u32), c.stream));
    hipLaunchKernelGGL(k_tie_stats<View>, dim3(ceil_div(N, 256 * TS_PER_THREAD)), dim3(256), 0, c.stream, v, N, sym_mask,
                       (u32)(DT_CAP - DT_TILE + 1), (u32)(DT_CAP - DT_TILE_WIDE + 1), d_stats);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipMemcpyAsync(c.h_scalars + 26, d_stats, 3 * sizeof(u32), hipMemcpyDeviceToHost, c.stream));
    c.sync();
    M = c.h_scalars[26];
    max_group = c.h_scalars[27] ? (u32)(DT_CAP - DT_TILE + 2) : (c.h_scalars[28] ? (u32)(DT_CAP - DT_TILE_WIDE + 2) : 0u);
  } else {
    Phase ph(c, "leaf_group", (double)N * (e_w ? 16 : 16));
    u32 *d_maxg = c.d_scalars + 11;
    HIP_CHECK(hipMemsetAsync(d_maxg, 0, sizeof(u32), c.stream));
    scan_generic<TieHead, TieHeadOp>(c, N, ViewTies<View>{v, N, sym_mask},
                                     FirstListOut<View>{v, wl_slot, wl_pos, wl_grp, d_total, N, d_maxg, sym_mask, (u32)(DT_CAP - DT_TILE_WIDE + 1)});
    HIP_CHECK(hipMemcpyAsync(c.h_scalars + 8, d_total, sizeof(u32), hipMemcpyDeviceToHost, c.stream));
    HIP_CHECK(hipMemcpyAsync(c.h_scalars + 11, d_maxg, sizeof(u32), hipMemcpyDeviceToHost, c.stream));
    c.sync();
    M = c.h_scalars[8];
    max_group = c.h_scalars[11];
  }
  // Dense or deep ties: one pass that settles every group with the ranks of the text's difference-cover sample,
  // instead of log2(longest repeat) doubling rounds over all of them.
  if (dc_state >= 0 && M > 0) {
    const bool wanted = dc_state == 1 ? c.dc_mode > 0 && max_group <= (u32)(DT_CAP - DT_TILE + 1)
                                      : dc_pass_applies(c, N, M, max_group, K0, skip_direct);
    if (dc_state == 1 && !wanted) {
      *redo = REDO_NO_DC;
      return false;
    }
    if (wanted && e != b) {
      *redo = REDO_DC;
      return false;
    }
    if (wanted) {
      ensure_dc_sample(c);
      const u32 nx = c.dc.v - 1 > (u32)K0 ? c.dc.v - 1 - (u32)K0 : 0u;  // symbols behind the key that a comparison may need
      Phase ph(c, "leaf_dc_emit", (double)N * (e_w ? 8 : 12) + (double)M * (c.dc.nd * 4 + nx) + (double)(b - a) * 2);
      LeafEmit<View> em;
      memset(&em, 0, sizeof(em));
      if (E) static_cast<EmitParams &>(em) = *E;
      em.sa_out = sa_out ? res.get() : nullptr;
      em.v = v;
      em.N = N;
      em.sym_mask = sym_mask;
      em.samp_ratio = em.m > 1 ? (double)(em.ns - 1) / (double)(i64)(em.m - 1) : 1.0;
      em.h = (u64)K0;
      em.leaf_end = (u32)b;
      em.fail = d_fail;
      em.T = c.T;
      em.n = (u64)c.n;
      HIP_CHECK(hipMemsetAsync(d_fail, 0, sizeof(u32), c.stream));
      const DcView dv = dc_view(c);
      const u32 tile = max_group <= (u32)(DT_CAP - DT_TILE_WIDE + 1) ? DT_TILE_WIDE : DT_TILE;
      const dim3 grid(ceil_div(N, tile)), block(DT_THREADS);
      const u32 min_sym = (u32)(c.min_sym > 0 ? c.min_sym : 1);
      const u32 xbits = (u32)bits_for((u64)(c.max_sym - (int)min_sym + 1));  // codes symbol - min_sym + 1
      const bool pk = nx > 0 && nx * xbits <= 58;
#define DC_EMIT(XW, PK) hipLaunchKernelGGL((k_dc_emit<View, XW, PK>), grid, block, 0, c.stream, em, dv, (u32)K0, nx, xbits, min_sym, tile)
      if (nx == 0) DC_EMIT(0, false);
      else if (nx <= 8) { if (pk) DC_EMIT(1, true); else DC_EMIT(1, false); }
      else if (nx <= 16) { if (pk) DC_EMIT(2, true); else DC_EMIT(2, false); }
      else if (nx <= 24 && pk) DC_EMIT(3, true);
      else DC_EMIT(4, false);
#undef DC_EMIT
      HIP_CHECK(hipGetLastError());
      if (read_u32(c, d_fail) != 0) {
        *redo = REDO_NO_DC;
        return false;
      }
      if (sa_out) *sa_out = std::move(res);
      return true;
    }
  }
  RefineRounds R;
  R.a = (u64)a;
  R.b = (u64)b;
  R.e = (u64)e;
  R.n = (u64)c.n;
  R.N = N;
  R.h = (u64)K0;
  R.wl_slot = wl_slot;
  R.wl_pos = wl_pos;
  R.wl_grp = wl_grp;
  R.M = M;
  DBuf<u32> isa, sa_mat;
  if (M > 0 && !sa_w) {
    // the refinement rounds rewrite positions: give the packed layout a position array of its own
    // (tied suffixes then take their BWT symbol from the text, as in the pairs layout)
    sa_mat = DBuf<u32>(c, N);
    hipLaunchKernelGGL(k_unpack_pos, dim3(ceil_div(N, 256)), dim3(256), 0, c.stream, e_w, (u32)N, (u32)a,
                       (u32)((1ull << bits_for(N - 1)) - 1), sa_mat.get());
    HIP_CHECK(hipGetLastError());
    sa_w = sa_mat.get();
    set_positions(v, sa_w);
  }
  R.sa_w = sa_w;
  // Text rounds cost O(M) and need no inverse array, but deepen the comparison by only K1 symbols;
  // doubling rounds double it but need the inverse array of the whole window (one random write per
  // window position).  Sparse ties get a few text rounds first; dense or deep ties go to doubling.
  while (R.M > 0 && R.round < 8 && (u64)R.M * 16 < N) {
    Phase ph(c, "leaf_text_round", (double)R.M * 8 * 32);
    radix_round(c, R, c.T, K1);
  }
  if (R.M > 0) {
    // ranks at this depth came from reading the text itself: exact whatever the window is
    R.h_exact = R.h;
    isa = DBuf<u32>(c, N);
    R.isa = isa.get();
    if (R.round == 0)  // the list is still the one the keys gave: tied = shares its key with a neighbour
      hipLaunchKernelGGL(k_isa_init_untied<View>, dim3(ceil_div(N, 256)), dim3(256), 0, c.stream, ViewTies<View>{v, N, sym_mask},
                         (const u32 *)sa_w, (u32)a, (u32)N, isa.get());
    else
      hipLaunchKernelGGL(k_isa_init, dim3(ceil_div(N, 256)), dim3(256), 0, c.stream, sa_w, (u32)a, (u32)N, isa.get());
    hipLaunchKernelGGL(k_isa_fix, dim3(ceil_div(R.M, 256)), dim3(256), 0, c.stream, R.wl_pos, R.wl_grp, R.M, (u32)a, isa.get());
    HIP_CHECK(hipGetLastError());
    // a window with long runs of one symbol: the rounds step over them (k_comp_isa, RunJump)
    if (c.run_jump && (c.window_long_run || c.run_jump == 2) && N < (1ull << 31)) {
      u32 tail = 0;
      if ((u64)e + 1 < (u64)c.n) {  // does the run at the window's end go on behind it, and how far?
        u8 last[2];
        HIP_CHECK(hipMemcpyAsync(last, c.T + e - 1, 2, hipMemcpyDeviceToHost, c.stream));
        c.sync();
        if (last[0] == last[1]) {
          u32 *d_ext = c.d_scalars + 38;
          hipLaunchKernelGGL(k_run_tail, dim3(1), dim3(256), 0, c.stream, c.T, (u64)c.n, (u64)e, (u32)last[0],
                             (const u32 *)(c.runs_valid ? c.runs : nullptr), c.runs_valid ? c.n_runs : 0u, d_ext);
          HIP_CHECK(hipGetLastError());
          tail = read_u32(c, d_ext);
        }
      }
      if (N + tail < (1ull << 31)) {
        R.run_ends = DBuf<u32>(c, N);
        hipLaunchKernelGGL(k_run_end_marks, dim3(ceil_div(N, 256)), dim3(256), 0, c.stream, c.T, (u64)a, N, (u64)c.n, R.run_ends.get());
        HIP_CHECK(hipGetLastError());
        scan_inclusive_max_u32(c, R.run_ends.get(), R.run_ends.get(), (size_t)N);
        R.J = RunJump{1, 0, R.run_ends.get(), N, tail, c.T};
      }
    }
    if (!doubling_rounds(c, R, "leaf_doubling_round")) return false;
  }
  LeafEmit<View> em;
  memset(&em, 0, sizeof(em));
  if (E) static_cast<EmitParams &>(em) = *E;
  em.sa_out = sa_out ? res.get() : nullptr;
  em.v = v;
  em.N = N;
  em.sym_mask = sym_mask;
  em.samp_ratio = em.m > 1 ? (double)(em.ns - 1) / (double)(i64)(em.m - 1) : 1.0;
  {
    Phase ph(c, "leaf_emit", (double)N * (e_w ? 8 : 12) + (double)(b - a) * 2);
    if (e == b) {
      hipLaunchKernelGGL(k_emit_all<View>, dim3(ceil_div(ceil_div(N, 4), 256)), dim3(256), 0, c.stream, em);
      HIP_CHECK(hipGetLastError());
    } else {
      scan_generic<u32, SumU32>(c, N, LeafFilterIn<View>{v, (u32)b}, LeafFilterOut<View>{em});
    }
  }
  if (sa_out) *sa_out = std::move(res);
  return true;
}

// Wide bucket route: windows that the packed 8-byte elements cannot index (more than 2^29 suffixes of DNA) but
// whose symbols are small.  Elements of 8 + 1 bytes (TextKeys::wide) go through three radix passes on their
// top bbits key bits; a bucket then holds a few thousand suffixes, is ordered in LDS on the next 16 key bits
// and emitted by k_bucket_emit like a bucket of the packed route; whatever those bbits + 16 bits leave tied
// is decided on the text.  1: done.  0: not applicable or given up (skewed or repetitive text), nothing
// usable was written and the caller sorts the window as (key, position) pairs.
static int try_wide_route(Ctx &c, i64 a, i64 b, i64 e, u64 N, u32 min_sym, int span, const EmitParams *E, DBuf<u32> *sa_out) {
  if (c.max_sym > 15 || N < c.bucket_min_n || N < 2) return 0;
  const bool pow2 = ((u32)(span + 1) & (u32)span) == 0;  // 2^k symbols: zero-based codes save a bit (TextKeys)
  const int one_based = pow2 ? 0 : 1;
  const int w = bits_for((u64)(span + one_based));
  // Buckets of 3500 to 7000 suffixes: larger buckets leave more ties to the text comparisons but cost less per
  // suffix (1.55e9 suffixes: 21 ms with 2^18 buckets, 25 ms with 2^19, 42 ms with 2^20).
  int bbits = 17;
  while (bbits < 24 && (N >> bbits) > 7000) bbits++;
  if (c.wide_bbits) bbits = c.wide_bbits;
  if ((N >> bbits) > 7000) return 0;
  const int h = (bbits + 16) / w;  // symbols covered by the bits the buckets are ordered on
  const int K0 = 36 / w;
  if (h < 1 || h > K0) return 0;
  const double sigma = span + 1 < 2 ? 2.0 : (double)(span + 1);
  if (expected_ties(sigma, h, N) > 0.25) return 0;
  constexpr u32 CAP = 8 * BK_THREADS;
  TextKeys tk{c.T, (u64)c.n, (u64)a, w, K0, min_sym, one_based, 0};
  tk.wide = 1;
  DBuf<u64> eb0(c, N), eb1(c, N);
  DBuf<u8> sb0(c, N), sb1(c, N);
  u64 *el[2] = {eb0.get(), eb1.get()};
  u8 *sd[2] = {sb0.get(), sb1.get()};
  BucketState bs;
  int cur;
  {
    Phase ph(c, "leaf_radix_sort", (double)N * (1 + 1 + 10 + 1 + 18 + 8 + 18));
    cur = leaf_sort_text_wide(c, el, sd, N, bbits, CAP, tk, &bs);
  }
  if (cur < 0) return 0;
  const u32 n_buckets = 1u << bbits;
  u32 *d_fail = c.d_scalars + 10;
  DBuf<u32> res;
  if (sa_out) res = DBuf<u32>(c, (size_t)(b - a));
  Phase ph(c, "leaf_direct_emit", (double)N * 9 + (double)(b - a) * 2);
  LeafEmit<WideView> em;
  memset(&em, 0, sizeof(em));
  if (E) static_cast<EmitParams &>(em) = *E;
  em.sa_out = sa_out ? res.get() : nullptr;
  em.v = WideView{el[cur], sd[cur], 20 - bbits};
  em.N = N;
  em.sym_mask = ~0xFFull;  // BucketWindow<WideView>::key: the ordered key bits above the preceding symbol
  em.samp_ratio = em.m > 1 ? (double)(em.ns - 1) / (double)(i64)(em.m - 1) : 1.0;
  em.h = (u64)h;
  em.leaf_end = (u32)b;
  em.fail = d_fail;
  em.T = c.T;
  em.n = (u64)c.n;
  HIP_CHECK(hipMemsetAsync(d_fail, 0, sizeof(u32), c.stream));
  {
    KernelTimer kt(c, "leaf:k_bucket_emit", (double)N * 9 + (double)(b - a));
    if (e == b) {
      launch_bucket_emit<WideView, false>(c, em, n_buckets, bs.max_bucket, bs.start.get(), nullptr);
    } else {
      DBuf<u32> sb(c, n_buckets);
      HIP_CHECK(hipMemsetAsync(sb.get(), 0, (size_t)n_buckets * sizeof(u32), c.stream));
      bucket_histogram(c, tk, (u64)b, (u64)(e - b), 16, sb.get(), bbits);
      scan_generic<u32, SumU32>(c, n_buckets, ArrayIn<u32>{sb.get()}, ArrayOutExcl<u32>{sb.get(), nullptr, n_buckets});
      launch_bucket_emit<WideView, true>(c, em, n_buckets, bs.max_bucket, bs.start.get(), sb.get());
    }
  }
  if (read_u32(c, d_fail) != 0) return 0;
  if (sa_out) *sa_out = std::move(res);
  return 1;
}

// ----------------------------------------------------------------------------
// Long runs of one symbol (the N of an assembly's gaps: millions of symbols) in closed form.  The suffixes that begin
// with kf copies of the run symbol c share one key; there are as many of them as the runs are long, far more than a
// bucket holds, and prefix doubling keeps a run of R symbols tied for log2 R rounds.  But their order is known:
// such a suffix is c^m X with X not beginning with c, and of two of them with m < m' the first is the smaller one
// exactly when X begins with a symbol below c -- so those whose run is followed by a smaller symbol come first, in
// the order of m, then those followed by a larger one, in the opposite order of m; equal m are decided by X.  The lean
// route leaves these suffixes out of its sort (TextKeys::run_span), ranks them with one radix sort on
// (follower below or above c, m or -m, the first symbols of X) plus text comparisons of the few that still agree, and
// inserts them as one block of consecutive ranks where their key falls among the others.
// ----------------------------------------------------------------------------
struct RunPlan {
  bool on = false;
  u32 sym = 0;  // the symbol whose runs are taken out
  // The symbol is rare outside its runs, or rare altogether (the N of an assembly): the keys then give it no code of
  // its own (TextKeys::code_max) and use the key space of the alphabet without it -- with a code that hardly ever
  // occurs, a fifth symbol leaves four fifths of the key space empty at every digit, and the buckets that are used
  // overflow.  Every suffix that begins with it is then in the block.
  bool merge = false;
  i64 longest = 0;  // the leaf's part of the longest listed run of one symbol
};

// out[0] += the positions of [a, b) that hold sym, out[1] += those that hold a smaller symbol (the terminator too)
__global__ void __launch_bounds__(256) k_count_symbol(const u8 *__restrict__ T, u64 a, u64 b, u32 sym, u32 *__restrict__ out) {
  u32 mine = 0, below = 0;
  for (u64 p = a + ((u64)blockIdx.x * 256 + threadIdx.x) * 8; p < b; p += (u64)gridDim.x * 256 * 8) {
    u64 w = 0;
    u32 count = 8;
    if (p + 8 <= b) {
      __builtin_memcpy(&w, T + p, 8);
    } else {
      count = (u32)(b - p);
      for (u32 k = 0; k < count; k++) w |= (u64)T[p + k] << (8 * k);
    }
    for (u32 k = 0; k < count; k++) {
      const u32 v = (u32)(w >> (8 * k)) & 255u;
      mine += v == sym;
      below += v < sym;
    }
  }
  mine = wave_reduce_sum(mine);
  below = wave_reduce_sum(below);
  if (lane_id() == 0) {
    if (mine) atomicAdd(out, mine);
    if (below) atomicAdd(out + 1, below);
  }
}

// out[v] += the positions of [a, b) that hold symbol v, for v < 16 (alphabets of a few symbols)
__global__ void __launch_bounds__(256) k_count_symbols16(const u8 *__restrict__ T, u64 a, u64 b, u32 *__restrict__ out) {
  __shared__ u32 tot[16];
  if (threadIdx.x < 16) tot[threadIdx.x] = 0;
  __syncthreads();
  u64 lo = 0, hi = 0;  // sixteen 8-bit counters, flushed before they can wrap
  u32 since = 0;
  auto flush = [&]() {
    for (u32 v = 0; v < 8; v++) {
      const u32 x = (u32)(lo >> (8 * v)) & 255u, y = (u32)(hi >> (8 * v)) & 255u;
      if (x) atomicAdd(&tot[v], x);
      if (y) atomicAdd(&tot[8 + v], y);
    }
    lo = hi = 0;
    since = 0;
  };
  for (u64 p = a + ((u64)blockIdx.x * 256 + threadIdx.x) * 8; p < b; p += (u64)gridDim.x * 256 * 8) {
    const u32 count = p + 8 <= b ? 8u : (u32)(b - p);
    for (u32 k = 0; k < count; k++) {
      const u32 v = T[p + k];
      if (v < 8) lo += 1ull << (8 * v);
      else if (v < 16) hi += 1ull << (8 * (v - 8));
    }
    if (++since == 31) flush();
  }
  flush();
  __syncthreads();
  if (threadIdx.x < 16 && tot[threadIdx.x]) atomicAdd(&out[threadIdx.x], tot[threadIdx.x]);
}

// the listed runs of one symbol that meet [a, b): worth it from 4096 positions on; the symbol of the longest
static RunPlan plan_run_filter(Ctx &c, i64 a, i64 b) {
  RunPlan rp;
  if (!c.run_filter || !c.chunks.empty()) return rp;  // (a text that is still arriving has not been scanned for runs)
  ensure_run_table(c);
  const size_t cnt = c.n_runs;
  i64 best = 0;
  for (size_t i = 0; i < cnt; i++) {
    if (c.runs_host[2 * cnt + i] != 1) continue;
    const i64 s0 = c.runs_host[i], e0 = c.runs_host[cnt + i];
    const i64 lo = s0 > a ? s0 : a, hi = e0 < b ? e0 : b;
    if (hi - lo > best) {
      best = hi - lo;
      rp.sym = c.runs_host[3 * cnt + i];
    }
  }
  rp.on = best >= 4096 && rp.sym != 0;
  rp.longest = best;
  // No code of its own for a symbol that is rare (RunPlan::merge): only where the alphabet without it has 2^k symbols
  // (bit-field codes), the symbol is not the smallest (its place in a key is taken by the symbol below it) and all
  // symbols are below 128 (the key roll's byte arithmetic).
  const int span = c.max_sym - c.min_sym;
  const bool fits = c.min_sym > 0 && span >= 2 && (span & (span - 1)) == 0 && c.max_sym < 128;
  if (rp.on && fits && (int)rp.sym > c.min_sym) {
    i64 covered = 0;  // by the listed runs of this symbol
    for (size_t i = 0; i < cnt; i++) {
      if (c.runs_host[2 * cnt + i] != 1 || c.runs_host[3 * cnt + i] != rp.sym) continue;
      const i64 s0 = c.runs_host[i], e0 = c.runs_host[cnt + i];
      const i64 lo = s0 > a ? s0 : a, hi = e0 < b ? e0 : b;
      if (hi > lo) covered += hi - lo;
    }
    u32 *d_count = c.d_scalars + 34;
    HIP_CHECK(hipMemsetAsync(d_count, 0, 2 * sizeof(u32), c.stream));
    hipLaunchKernelGGL(k_count_symbol, dim3(1024), dim3(256), 0, c.stream, c.T, (u64)a, (u64)b, rp.sym, d_count);
    HIP_CHECK(hipGetLastError());
    const i64 outside = (i64)read_u32(c, d_count) - covered;
    rp.merge = outside * 512 < b - a;
  } else if (!rp.on && fits && c.max_sym < 16 && (u64)(b - a) >= c.bucket_min_n) {
    // no long runs, but maybe a symbol that hardly occurs (a few N scattered over a genome): the rarest one
    u32 *d_counts = c.d_scalars + 48;
    HIP_CHECK(hipMemsetAsync(d_counts, 0, 16 * sizeof(u32), c.stream));
    hipLaunchKernelGGL(k_count_symbols16, dim3(1024), dim3(256), 0, c.stream, c.T, (u64)a, (u64)b, d_counts);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipMemcpyAsync(c.h_scalars + 48, d_counts, 16 * sizeof(u32), hipMemcpyDeviceToHost, c.stream));
    c.sync();
    int rare = 0;
    for (int v = c.min_sym + 1; v <= c.max_sym; v++)
      if (rare == 0 || c.h_scalars[48 + v] < c.h_scalars[48 + rare]) rare = v;
    if (rare && (u64)c.h_scalars[48 + rare] * 512 < (u64)(b - a)) {
      rp.on = true;
      rp.sym = (u32)rare;
      rp.merge = true;
    }
  }
  return rp;
}

struct RunMatch {
  const u8 *T;
  u64 n;
  u32 sym;
  // bit k (k < count <= 64): position p0 + k holds the symbol
  __device__ __forceinline__ u64 bits(u64 p0, int count) const {
    u64 out = 0;
    for (int k = 0; k < count; k += 8) {
      u64 eight;
      if (p0 + k + 9 <= n) {  // eight positions before the terminator
        u64 w;
        __builtin_memcpy(&w, T + p0 + k, 8);
        const u64 x = w ^ (0x0101010101010101ull * sym);
        u64 t = (x & 0x7F7F7F7F7F7F7F7Full) + 0x7F7F7F7F7F7F7F7Full;
        t = ~(t | x | 0x7F7F7F7F7F7F7F7Full);                    // 0x80 in every byte that matched
        eight = ((t >> 7) * 0x0102040810204080ull) >> 56;        // those eight bits, first position lowest
      } else {
        eight = 0;
        for (int j = 0; j < 8; j++) {
          const u64 p = p0 + k + j;
          eight |= (u64)(p + 1 < n && T[p] == sym) << j;
        }
      }
      out |= eight << k;
    }
    return count < 64 ? out & ((1ull << count) - 1ull) : out;
  }
  // of the 16 positions from p0 on (those below `end`): which begin kf copies of the symbol
  __device__ __forceinline__ u32 left_out16(u64 p0, u64 end, int kf) const {
    u64 r = bits(p0, 16 + kf - 1);
    for (int len = 1; len < kf;) {
      const int sft = len < kf - len ? len : kf - len;
      r &= r >> sft;
      len += sft;
    }
    u32 f = (u32)r & 0xFFFFu;
    if (p0 + 16 > end) f &= (1u << (end - p0)) - 1u;
    return f;
  }
};

struct RunCountIn {
  RunMatch rm;
  u64 a, b;
  int kf;
  __device__ __forceinline__ u32 operator()(size_t i) const { return __popc(rm.left_out16(a + 16 * (u64)i, b, kf)); }
};
struct RunListOut {
  RunMatch rm;
  u64 a, b;
  int kf;
  u32 *P, cap;
  u32 *total;
  size_t n;
  __device__ __forceinline__ void operator()(size_t i, u32 excl, u32 incl) const {
    u32 f = rm.left_out16(a + 16 * (u64)i, b, kf);
    u32 o = excl;
    while (f) {
      const u32 k = (u32)__ffs(f) - 1u;
      if (o < cap) P[o] = (u32)(a + 16 * (u64)i + k);
      o++;
      f &= f - 1u;
    }
    if (i + 1 == n) *total = incl;
  }
};

// *ext = how many positions from x0 on hold the symbol (x0 < n).  One workgroup: 2048 positions a step, and after every
// step that found nothing else a look at the list of long runs, which says where the one around the position ends.
__global__ void __launch_bounds__(256) k_run_ext(RunMatch rm, u64 x0, const u32 *__restrict__ runs, u32 n_runs, u32 *__restrict__ ext) {
  __shared__ u32 first_s;
  u64 p = x0;
  for (;;) {
    if (threadIdx.x == 0) first_s = 2048;
    __syncthreads();
    const u64 mine = p + 8 * threadIdx.x;
    u64 m8 = 0;
    for (int j = 0; j < 8; j++) m8 |= (u64)(mine + j + 1 < rm.n && rm.T[mine + j] == rm.sym) << j;
    if (m8 != 0xFF) atomicMin(&first_s, 8 * threadIdx.x + (u32)__ffsll((unsigned long long)~m8) - 1u);
    __syncthreads();
    const u32 first = first_s;
    __syncthreads();
    if (first < 2048) {
      if (threadIdx.x == 0) *ext = (u32)(p + first - x0);
      return;
    }
    p += 2048;
    u64 end = 0;
    if (n_runs && run_around(runs, n_runs, p, &end) == 1 && end > p + 16) p = end - 8;
  }
}

struct LastSetU32 {
  __device__ __forceinline__ static u32 identity() { return 0; }
  __device__ __forceinline__ static u32 apply(u32 earlier, u32 later) { return later ? later : earlier; }
};
// the list from its end: where a stretch of consecutive positions ends, that position + 1
struct RunEndIn {
  const u32 *P;
  u32 cnt;
  __device__ __forceinline__ u32 operator()(size_t j) const {
    const u32 i = cnt - 1u - (u32)j;
    return (i + 1 == cnt || P[i + 1] != P[i] + 1u) ? P[i] + 1u : 0u;
  }
};
// key of list entry i: | follower above c, 1 bit | m, or 2^32 - 1 - m behind a larger follower, 32 bits | codes of the
// first kx symbols behind the run, 31 bits |
struct RunKeyOut {
  const u32 *P;
  u32 cnt;
  const u8 *T;
  u64 n, b;
  u32 kf, sym, min_sym;
  const u32 *ext;
  int wx, kx;
  u64 *keys;
  u32 *vals;
  __device__ __forceinline__ void operator()(size_t j, u32, u32 incl) const {
    const u32 i = cnt - 1u - (u32)j;
    const u64 q = P[i], last = (u64)incl - 1u;  // the stretch's last position that begins kf copies
    u64 m = last + kf - q;
    if (last + 1 == b) m += *ext;               // the run goes on behind the leaf
    const u64 x = q + m;
    const u32 f = x < n ? T[x] : 0u;
    const u64 above = f > sym;
    const u64 mm = above ? 0xFFFFFFFFull - m : m;
    u64 xp = 0;
    for (int k = 0; k < kx; k++) {
      const u32 ch = x + k < n ? T[x + k] : 0u;
      xp = (xp << wx) | (ch ? ch - min_sym + 1u : 0u);
    }
    xp <<= 31 - kx * wx;
    keys[i] = (above << 63) | (mm << 31) | xp;
    vals[i] = (u32)q;
  }
};

// *out = leaf rank of the block: the sorted suffixes whose key lies below the keys left out.  They sit in one
// bucket (bkt) and before it; an element shows the key's bits below the first digit down to a precision that may
// be coarser than the key, so an element inside the range those bits leave open is compared on the text.
__global__ void __launch_bounds__(256) k_run_block_start(const u64 *__restrict__ el, const u32 *__restrict__ start, u32 bkt, TextKeys tk,
                                                         u32 posmask, u32 sym, u32 kf, u32 *__restrict__ out) {
  __shared__ u32 scan_s[17];
  const u32 s0 = start[bkt], s1 = start[bkt + 1];
  const u64 topmask = ~0ull << (64 - tk.b2 - tk.lean_deep);
  const u64 t_lo = (tk.run_lo << tk.b1) & topmask, t_hi = ((tk.run_lo + (tk.run_span - 256)) << tk.b1) & topmask;
  u32 below = 0;
  for (u32 i = s0 + threadIdx.x; i < s1; i += 256) {
    const u64 e = el[i], t = e & topmask;
    if (t < t_lo) {
      below++;
    } else if (t <= t_hi) {
      const u64 p = tk.a + ((u32)e & posmask);
      for (u32 j = 0; j < kf; j++) {  // (it does not begin with kf copies of the symbol: it would have been left out)
        const u32 ch = p + j < tk.n ? tk.T[p + j] : 0u;
        if (ch != sym) {
          below += ch < sym;
          break;
        }
      }
    }
  }
  u32 tot;
  block_scan_incl<OpSum>(below, scan_s, &tot);
  if (threadIdx.x == 0) *out = s0 + tot;
}

// the sorted suffixes' samples and special ranks, written with ranks that do not count the block: those behind it move up
__global__ void k_run_shift_samples(u64 *__restrict__ samp, i64 ns, u32 *__restrict__ special, const u32 *__restrict__ at, u32 cnt) {
  const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  const u32 B = *at;
  if (i < 2 && special && special[i] >= B) special[i] += cnt;
  if ((i64)i >= ns || !samp) return;
  const u64 v = samp[i];
  if (v != ~0ull && (u32)v >= B) samp[i] = v + cnt;
}

// Entry i of the block (sorted on RunKeyOut's keys) takes rank *at + i, or, where keys agree, the rank that the text
// behind the run and the key's symbols gives it among those.
template <class View>
__global__ void k_run_place(LeafEmit<View> E, const u64 *__restrict__ keys, const u32 *__restrict__ pos, u32 cnt,
                            const u32 *__restrict__ at, int kx) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= cnt) return;
  const u64 key = keys[i];
  u32 rank = i;
  if ((i > 0 && keys[i - 1] == key) || (i + 1 < cnt && keys[i + 1] == key)) {
    u32 gs = i, ge = i;
    while (gs > 0 && i - gs < DF_GROUP && keys[gs - 1] == key) gs--;
    while (ge + 1 < cnt && ge - i < DF_GROUP && keys[ge + 1] == key) ge++;
    if (i - gs >= DF_GROUP || ge - i >= DF_GROUP) {
      *E.fail = 1;
      return;
    }
    const u64 mm = (key >> 31) & 0xFFFFFFFFull;
    const u64 depth = ((key >> 63) ? 0xFFFFFFFFull - mm : mm) + (u64)kx;
    u32 smaller = 0;
    for (u32 j = gs; j <= ge; j++) {
      if (j == i) continue;
      bool decided;
      const bool less = text_less(E.T, E.n, (u64)pos[j] + depth, (u64)pos[i] + depth, &decided, E.runs, E.n_runs);
      if (!decided) {
        *E.fail = 1;
        return;
      }
      smaller += less;
    }
    rank = gs + smaller;
  }
  E.place((u64)*at + rank, pos[i], E.sym_before(pos[i]));
}

// Lean bucket route: windows of up to 2^32 - 9 suffixes of small symbols (DNA), 8-byte elements with nothing on the
// side (TextKeys::lean).  Two radix passes -- the text pass on the lower b1 bucket bits, which it does not store,
// the second on the top b2 -- leave 2^(b1+b2) buckets of a few thousand suffixes; k_bucket_emit orders a bucket in
// LDS on the next 16 key bits, settles what those leave tied on the key bits below them and only then on the
// text.  1: done.  0: not applicable or given up (skewed or repetitive text), nothing usable was written.
static int try_lean_route(Ctx &c, i64 a, i64 b, i64 e, u64 N, u32 min_sym, int span, const EmitParams *E, DBuf<u32> *sa_out,
                          LeanShard *shard = nullptr, const RunPlan *rp = nullptr) {
  if (N < c.bucket_min_n || N < 2 || N >= (1ull << 32) - 8) return 0;
  // a largest symbol that shares the code of the one below it (RunPlan::merge): the keys' alphabet is one smaller
  // (only where the alphabet without it has 2^k symbols: bit-field codes)
  const bool merged = rp && rp->on && rp->merge && !shard && e == b && span >= 2 && rp->sym > min_sym && rp->sym <= min_sym + (u32)span &&
                      ((u32)span & (u32)(span - 1)) == 0 && c.max_sym < 128;
  if (merged) span--;
  const bool pow2 = ((u32)(span + 1) & (u32)span) == 0;  // 2^k symbols: zero-based codes save a bit (TextKeys)
  // Alphabets of 3, 5, 6 or 7 symbols (DNA with N): the codes as digits of a number in that base, scaled to the top of
  // 56 bits (TextKeys::base) -- bit fields would spend 2 or 3 bits on 1.6 to 2.8 bits of information and leave dozens
  // of suffixes per key value.
  const u32 base = (!pow2 && span + 1 >= 3 && span + 1 <= 7 && c.lean_base) ? (u32)(span + 1) : 0u;
  const int one_based = (pow2 || base) ? 0 : 1;
  const int w = bits_for((u64)(span + one_based));
  int K0b = 0;
  u64 lead = 1, scale = 1;
  double fill = 1.0;  // the part of the key space that keys of a text with all symbols equally frequent reach
  if (base) {
    u64 pw = 1;
    while (pw <= (1ull << 56) / base) {
      lead = pw;
      pw *= base;
      K0b++;
    }
    scale = (1ull << 56) / pw;
    fill = (double)(scale * pw) / (double)(1ull << 56);
  }
  // buckets of 4100 to 8200 suffixes (the largest bucket of a random text then stays below the 9216 a workgroup holds)
  // (2^19 buckets for windows of 2^31 suffixes and more: a 10-bit second digit)
  const u64 Nf = (u64)((double)N / fill);
  int bbits = 16;
  // (a symbol without a code of its own leaves the suffixes that hold it early with few distinct keys: those gather in
  // a few buckets, which then need room)
  while (bbits < 19 && (Nf >> bbits) > (merged ? 4100u : 8200u)) bbits++;
  if (c.wide_bbits >= 2 && c.wide_bbits <= 19) bbits = c.wide_bbits;
  if ((Nf >> bbits) > 8200) return 0;
  const int b2 = bbits > 18 ? 10 : (bbits > 16 ? 9 : (bbits + 1) / 2), b1 = bbits - b2;
  const int vbits = c.sym_bits;
  int pbits = bits_for(N - 1);
  if (pbits < 13 + b1) pbits = 13 + b1;
  int deep = 64 - b2 - vbits - pbits;
  if (deep > 48 - bbits) deep = 48 - bbits;
  if (!base && deep > 56 / w * w - bbits) deep = 56 / w * w - bbits;
  if (deep < 16) return 0;
  const int K0 = base ? K0b : (bbits + deep + w - 1) / w;  // covers every stored key bit (K0 * w <= 56)
  const int h16 = (bbits + 16) / w;                        // symbols covered by the bits the buckets are ordered on
  // whole symbols that two elements with the same key bits are known to share: none with digits of another base, or
  // with two symbols behind one code
  const int h = (base || merged) ? 0 : (bbits + deep) / w;
  const double sigma = span + 1 < 2 ? 2.0 : (double)(span + 1);
  if (base) {
    if ((double)N / (ldexp(1.0, bbits + 16) * fill) > 0.25) return 0;
  } else if (h16 < 1 || expected_ties(sigma, h16, N) > 0.25) {
    return 0;
  }
  TextKeys tk{c.T, (u64)c.n, (u64)a, w, K0, min_sym, one_based, pbits};
  tk.lean = 1;
  tk.b1 = b1;
  tk.b2 = b2;
  tk.lean_deep = deep;
  tk.vbits = vbits;
  tk.base = base;
  tk.scale = scale;
  tk.lead = lead;
  tk.fill = (float)(base ? fill : (one_based ? pow((double)(span + 1) / (double)(1u << w), (double)bbits / (double)w) : 1.0));
  tk.fill_even = base || !one_based;
  // long runs of one symbol: the suffixes that begin with kf copies of it stay out of the sort (RunPlan)
  // How many copies, kf: few enough that a suffix with kf - 1 of them still shows some symbols of what follows the
  // run in the 16 key bits its bucket is ordered on (the bucket kernel gives up on 16 equal values: one such suffix per
  // run would do), enough that the keys left out fall inside one bucket.
  bool runf = rp && rp->on && !shard && e == b && rp->sym >= min_sym && (int)(rp->sym - min_sym) <= span + (merged ? 1 : 0);
  int kf = h16 > 7 ? h16 - 6 : 1;
  if (merged) tk.code_max = 1;
  if (runf) {
    const u64 B = base ? base : 1ull << w, scl = base ? scale : 1ull << (56 - K0 * w);
    u64 code_c = rp->sym - min_sym + (u32)one_based;
    if (merged) code_c = rp->sym - min_sym - 1;  // (the code of the symbol below it; zero-based: 2^k symbols)
    const int kmax = lean_count_symbols(tk);  // what the counting pass sees of a key
    // (a symbol without a code of its own: every suffix that begins with it is left out -- their keys are all the
    // same -- and, the symbol being the largest, their block comes behind everything else)
    if (merged) kf = 1;
    for (runf = false; kf <= kmax && !runf; kf++) {
      u64 lead_kf = 0, rest = 1;
      for (int j = 0; j < kf; j++) lead_kf = lead_kf * B + code_c;
      for (int j = kf; j < K0; j++) rest *= B;
      tk.run_lo = (lead_kf * rest * scl) << 8;
      tk.run_span = (rest * scl) << 8;
      runf = merged || (tk.run_lo >> (64 - bbits)) == ((tk.run_lo + (tk.run_span - 256)) >> (64 - bbits));
    }
    kf--;
    if (!runf) tk.run_lo = tk.run_span = 0;
    tk.run_sym = rp->sym;
    tk.run_k = (u32)kf;
  }
  if (merged && !runf) return 0;
  DBuf<u64> el[2];
  BucketState bs;
  int cur;
  size_t kept = N;
  {
    Phase ph(c, "leaf_radix_sort", (double)N * (1 + 1 + 8 + 8 + 16));
    cur = leaf_sort_text_lean(c, el, N, BK_CAP, tk, &bs, shard, &kept);
  }
  if (cur < 0) return 0;
  const u32 n_buckets = 1u << bbits;
  u32 *d_fail = c.d_scalars + 10;
  // the block of suffixes left out: listed, keyed and sorted while the others wait for their bucket kernel
  const u32 run_cnt = (u32)(N - kept);
  u32 *d_run_at = c.d_scalars + 32, *d_run_ext = c.d_scalars + 33;
  DBuf<u64> rk0, rk1;
  DBuf<u32> rv0, rv1;
  int rcur = 0;
  const int wx = bits_for((u64)span + 1), kx = 31 / wx;
  if (runf && run_cnt > 0) {
    Phase ph(c, "leaf_run_block", (double)N + (double)run_cnt * 8 * 24);
    u32 *d_total = c.d_scalars + 8;
    const RunMatch rm{c.T, (u64)c.n, rp->sym};
    DBuf<u32> P(c, run_cnt);
    const size_t chunks = (size_t)((N + 15) / 16);
    scan_generic<u32, SumU32>(c, chunks, RunCountIn{rm, (u64)a, (u64)b, kf},
                              RunListOut{rm, (u64)a, (u64)b, kf, P.get(), run_cnt, d_total, chunks});
    if (read_u32(c, d_total) != run_cnt)
      fail(BWT_HIP_EINTERNAL, "leaf sort: %u suffixes left out of the sort, %u begin with %d copies of symbol %u", run_cnt,
           c.h_scalars[0], kf, rp->sym);
    HIP_CHECK(hipMemsetAsync(d_run_ext, 0, sizeof(u32), c.stream));
    const u64 x0 = (u64)b - 1 + (u64)kf;
    if (x0 < (u64)c.n) {
      hipLaunchKernelGGL(k_run_ext, dim3(1), dim3(256), 0, c.stream, rm, x0, (const u32 *)c.runs, c.n_runs, d_run_ext);
      HIP_CHECK(hipGetLastError());
    }
    rk0 = DBuf<u64>(c, run_cnt);
    rk1 = DBuf<u64>(c, run_cnt);
    rv0 = DBuf<u32>(c, run_cnt);
    rv1 = DBuf<u32>(c, run_cnt);
    scan_generic<u32, LastSetU32>(c, run_cnt, RunEndIn{P.get(), run_cnt},
                                  RunKeyOut{P.get(), run_cnt, c.T, (u64)c.n, (u64)b, (u32)kf, rp->sym, min_sym, d_run_ext, wx, kx,
                                            rk0.get(), rv0.get()});
    u64 *keys[2] = {rk0.get(), rk1.get()};
    u32 *vals[2] = {rv0.get(), rv1.get()};
    c.sort_tag = "run:";
    rcur = radix_sort_u64(c, keys, vals, run_cnt, 0, 64);
    c.sort_tag = "";
    if (merged) {
      // every suffix that begins with the symbol is in the block: it comes behind the suffixes that begin with a smaller one
      u32 *d_count = c.d_scalars + 34;
      HIP_CHECK(hipMemsetAsync(d_count, 0, 2 * sizeof(u32), c.stream));
      hipLaunchKernelGGL(k_count_symbol, dim3(1024), dim3(256), 0, c.stream, c.T, (u64)a, (u64)b, rp->sym, d_count);
      HIP_CHECK(hipGetLastError());
      HIP_CHECK(hipMemcpyAsync(d_run_at, d_count + 1, sizeof(u32), hipMemcpyDeviceToDevice, c.stream));
    } else {
      hipLaunchKernelGGL(k_run_block_start, dim3(1), dim3(256), 0, c.stream, (const u64 *)el[cur].get(), (const u32 *)bs.start.get(),
                         (u32)(tk.run_lo >> (64 - bbits)), tk, (u32)((1ull << pbits) - 1ull), rp->sym, (u32)kf, d_run_at);
      HIP_CHECK(hipGetLastError());
    }
  }
  // A GPU that holds a share of the leaf (shard) has its elements, its stretch of L and of the suffix array at local
  // index = leaf rank - first_rank; the bucket bounds are leaf ranks, so the three arrays are addressed from where
  // rank 0 would lie.
  const u64 first_rank = shard ? shard->first_rank : 0;
  const u64 mine = shard ? shard->count : (u64)(b - a);
  DBuf<u32> res;
  if (sa_out) res = DBuf<u32>(c, (size_t)(mine ? mine : 1));
  Phase ph(c, "leaf_direct_emit", (double)mine * 8 + (double)mine * 2);
  LeafEmit<LeanView> em;
  memset(&em, 0, sizeof(em));
  if (E) static_cast<EmitParams &>(em) = *E;
  if (shard && em.samp) {  // a leaf build: this share's stretch of L is made here, now that its length is known
    shard->L = DBuf<u8>(c, (size_t)(mine ? mine : 1));
    em.L = shard->L.get();
  }
  if (em.L) em.L -= first_rank;
  em.sa_out = sa_out ? res.get() - first_rank : nullptr;
  em.v = LeanView{el[cur].get() - first_rank, (u32)a, (u32)((1ull << pbits) - 1ull), (1u << vbits) - 1u, pbits, b2, deep - 16, bs.slot};
  em.N = first_rank + kept;  // (ranks: a share's elements begin at its first rank)
  em.sym_mask = ~0xFFull;  // BucketWindow<LeanView>::key: the ordered key bits above the preceding symbol
  em.samp_ratio = em.m > 1 ? (double)(em.ns - 1) / (double)(i64)(em.m - 1) : 1.0;
  em.h = (u64)h;
  em.leaf_end = (u32)b;
  em.fail = d_fail;
  em.T = c.T;
  em.n = (u64)c.n;
  if (rp && rp->on) {  // (the list exists: RunPlan comes from it)
    em.runs = c.runs;
    em.n_runs = c.n_runs;
  }
  HIP_CHECK(hipMemsetAsync(d_fail, 0, sizeof(u32), c.stream));
  // buckets with too many equal keys for the bucket kernel are listed and taken by a general one afterwards
  DBuf<u32> deferred;
  u32 *d_deferred = c.d_scalars + 36;
  if (e == b && c.defer_buckets) {
    deferred = DBuf<u32>(c, n_buckets);
    HIP_CHECK(hipMemsetAsync(d_deferred, 0, sizeof(u32), c.stream));
    em.defer_list = deferred.get();
    em.defer_count = d_deferred;
  }
  {
    KernelTimer kt(c, "leaf:k_bucket_emit", (double)N * 8 + (double)(b - a));
    if (e == b) {
      launch_bucket_emit<LeanView, false>(c, em, n_buckets, bs.max_bucket, bs.start.get(), nullptr);
    } else {
      DBuf<u32> sb(c, n_buckets);
      HIP_CHECK(hipMemsetAsync(sb.get(), 0, (size_t)n_buckets * sizeof(u32), c.stream));
      bucket_histogram(c, tk, (u64)b, (u64)(e - b), 16, sb.get(), bbits);
      scan_generic<u32, SumU32>(c, n_buckets, ArrayIn<u32>{sb.get()}, ArrayOutExcl<u32>{sb.get(), nullptr, n_buckets});
      launch_bucket_emit<LeanView, true>(c, em, n_buckets, bs.max_bucket, bs.start.get(), sb.get());
    }
  }
  if (read_u32(c, d_fail) != 0) return 0;
  if (em.defer_list) {
    const u32 n_def = read_u32(c, d_deferred);
    if (n_def > 0) {
      KernelTimer kt(c, "leaf:k_bucket_general", (double)n_def * bs.max_bucket * 8);
      hipLaunchKernelGGL(k_bucket_general, dim3(n_def), dim3(BG_THREADS), 0, c.stream, em, (const u32 *)bs.start.get(), (const u32 *)deferred.get());
      HIP_CHECK(hipGetLastError());
      if (read_u32(c, d_fail) != 0) return 0;
    }
  }
  if (runf && run_cnt > 0) {
    // the sorted suffixes behind the block move up by its length, then the block takes its ranks
    Phase ph(c, "leaf_run_block", (double)kept * 2 + (double)run_cnt * 16);
    const u32 at = read_u32(c, d_run_at);
    if (at > kept) fail(BWT_HIP_EINTERNAL, "leaf sort: the block of run suffixes at rank %u of %zu", at, kept);
    const size_t tail = kept - at;
    if (tail > 0 && em.L) {
      DBuf<u8> tmp(c, tail);
      HIP_CHECK(hipMemcpyAsync(tmp.get(), em.L + at, tail, hipMemcpyDeviceToDevice, c.stream));
      HIP_CHECK(hipMemcpyAsync(em.L + at + run_cnt, tmp.get(), tail, hipMemcpyDeviceToDevice, c.stream));
      c.sync();  // tmp goes back to the pool
    }
    if (tail > 0 && em.sa_out) {
      DBuf<u32> tmp(c, tail);
      HIP_CHECK(hipMemcpyAsync(tmp.get(), em.sa_out + at, tail * sizeof(u32), hipMemcpyDeviceToDevice, c.stream));
      HIP_CHECK(hipMemcpyAsync(em.sa_out + at + run_cnt, tmp.get(), tail * sizeof(u32), hipMemcpyDeviceToDevice, c.stream));
      c.sync();
    }
    if (em.L) {
      const u64 slots = em.ns > 2 ? (u64)em.ns : 2;
      hipLaunchKernelGGL(k_run_shift_samples, dim3(ceil_div(slots, 256)), dim3(256), 0, c.stream, em.samp, em.ns, em.special,
                         (const u32 *)d_run_at, run_cnt);
      HIP_CHECK(hipGetLastError());
    }
    hipLaunchKernelGGL(k_run_place<LeanView>, dim3(ceil_div(run_cnt, 256)), dim3(256), 0, c.stream, em, (const u64 *)(rcur ? rk1.get() : rk0.get()),
                       (const u32 *)(rcur ? rv1.get() : rv0.get()), run_cnt, (const u32 *)d_run_at, kx);
    HIP_CHECK(hipGetLastError());
    if (read_u32(c, d_fail) != 0) return 0;
  }
  if (sa_out) *sa_out = std::move(res);
  return 1;
}

// Sorts the window [a,e).  On success: if E is not null the leaf's L, special ranks and samples have
// been written through it; if sa_out is not null it holds the leaf suffix array (b-a entries).
static bool sort_window(Ctx &c, i64 a, i64 b, i64 e, const EmitParams *E, DBuf<u32> *sa_out, bool repeats, int dc_state, int *redo,
                        const RunPlan *rp = nullptr) {
  const u64 N = (u64)(e - a);
  const u32 min_sym = (u32)(c.min_sym > 0 ? c.min_sym : 1);
  const int span = c.max_sym > (int)min_sym ? c.max_sym - (int)min_sym : 0;
  const KeyPlan kp = plan_keys(span, c.sym_bits, N);
  const int K0 = kp.K0;
  const int low_bit = 64 - K0 * kp.w;  // >= 8 (pairs), >= 32 (packed)
  const u64 sym_mask = ~0ull << low_bit;
  const int begin_bit = low_bit & ~7;  // whole digits; the extra low bits are zero padding
  const int passes = (64 - begin_bit) / 8;
  TextKeys tk{c.T, (u64)c.n, (u64)a, kp.w, K0, min_sym, kp.one_based, kp.pbits};
  // (repeats: a text with long repeats goes straight to the plain passes and what follows them)
  // Large windows: lean elements; wide ones where those do not apply.  These routes settle every tie on the text
  // itself, so they sort the leaf's own suffixes and no window slack.  (force_wide 3: never the lean route; windows
  // the packed elements can index then take those elements' own bucket route, the lean route's predecessor.)
  const u64 NL = (u64)(b - a);
  if (!repeats && c.force_wide != 1 && c.force_wide != 3 && try_lean_route(c, a, b, b, NL, min_sym, span, E, sa_out, nullptr, rp) == 1) return true;
  if (!repeats && (!kp.pbits || c.force_wide == 1) && try_wide_route(c, a, b, b, NL, min_sym, span, E, sa_out) == 1) return true;
  if (kp.pbits) {
    // packed elements; the spare buffer later holds the first active list (2N words)
    DBuf<u64> eb0(c, N), eb1(c, N);
    DBuf<u32> grp(c, N);
    BucketState bs;
    u64 *el[2] = {eb0.get(), eb1.get()};
    int cur;
    {
      // Algorithmic bytes per window suffix: the text pass reads a symbol twice (tile histogram, scatter) and
      // writes the 8-byte element plus the next pass's digit; the second pass reads that digit and moves the
      // element (16); every further pass reads the element for its histogram (8) and moves it (16).  The
      // bucket route stops after the second pass.
      Phase ph(c, "leaf_radix_sort", 0);
      // the packed elements' own bucket route is what the lean route replaced: only where that one is switched off
      const bool own_buckets = !repeats && c.force_wide == 3;
      cur = leaf_sort_text_packed(c, el, N, begin_bit, tk, own_buckets ? &bs : nullptr, own_buckets);
      const int done = bs.pending ? 2 : passes;
      ph.add_work((double)N * (done <= 1 ? 10 : 11 + 17 + (done - 2) * 24));
    }
    PackedView v{el[cur], nullptr, (u32)a, (u32)((1ull << kp.pbits) - 1), kp.pbits};
    return finish_window<PackedView>(c, a, b, e, N, K0, sym_mask, v, nullptr, el[cur], (u32 *)el[cur ^ 1], grp.get(), E,
                                     sa_out, dc_state, redo, &bs, &tk, repeats);
  }
  const u32 stride = (u32)((N + 3) & ~(u64)3);  // 32-bit words between a key's low and high word
  DBuf<u64> kb0(c, stride), kb1(c, stride);
  DBuf<u32> sb0(c, N), sb1(c, N);
  u64 *keys[2] = {kb0.get(), kb1.get()};
  u32 *vals[2] = {sb0.get(), sb1.get()};
  int cur;
  {
    Phase ph(c, "leaf_radix_sort", (double)N * (passes * 28 - 15));
    cur = leaf_sort_text_keys(c, keys, vals, N, stride, begin_bit, tk);
  }
  PairsView v{reinterpret_cast<const u32 *>(keys[cur]), reinterpret_cast<const u32 *>(keys[cur]) + stride, vals[cur]};
  return finish_window<PairsView>(c, a, b, e, N, K0, sym_mask, v, vals[cur], nullptr, (u32 *)keys[cur ^ 1],
                                  vals[cur ^ 1], E, sa_out, dc_state, redo, nullptr, nullptr, repeats);
}

static void run_leaf_sort(Ctx &c, i64 a, i64 b, const EmitParams *E, DBuf<u32> *sa_out) {
  if (!c.T) fail(BWT_HIP_EINVAL, "no text set");
  if (a < 0 || b > c.n || a >= b) fail(BWT_HIP_EINVAL, "bad leaf range [%ld,%ld) for n=%ld", (long)a, (long)b, (long)c.n);
  i64 slack = 0;
  if (b < c.n) {
    slack = (b - a) / 256;
    if (slack < 4096) slack = 4096;
    // A long periodic stretch (a run of one symbol above all) that crosses the leaf's end, or ends less than its own
    // length before it, keeps the leaf's suffixes inside it tied until the doubling depth h exceeds half of what is
    // left of the stretch, and a round reads ranks up to 2h beyond a suffix: the window reaches that far from the
    // start, instead of finding out after log2 of the slack rounds, twice.  (A text that is still arriving is scanned
    // for stretches at its first merge only.)
    if (c.chunks.empty()) {
      ensure_run_table(c);
      const size_t cnt = c.n_runs;
      for (size_t i = 0; i < cnt; i++) {
        const i64 s0 = c.runs_host[i], e0 = c.runs_host[cnt + i];
        if (s0 < b && e0 > a) {
          const i64 from = s0 > a ? s0 : a;
          const i64 want = (e0 - b) + (e0 - from) + 4096;
          if (want > slack) slack = want;
        }
      }
    }
    if (slack > c.n - b) slack = c.n - b;
  }
  // A text whose sample ranks exist (an earlier leaf found it repetitive) is sorted leaf by leaf without slack: ties
  // are settled by those ranks, which are the whole text's.
  int dc_state = (c.dc_mode > 0 && c.dc.valid) ? 1 : 0;
  // a large leaf is probed for long repeats: such a text skips the bucket routes and the direct pass, and, where the
  // sample ranks may be used, is sorted without slack from the start
  c.wait_text((u64)b + 128);
  const RunPlan rp = plan_run_filter(c, a, b);
  const bool repeats = window_has_long_repeats(c, a, (u64)(b - a), rp.on);
  if (repeats && c.dc_mode > 0) dc_state = 1;
  // (a long run of one symbol is one tied group far beyond what the sample ranks are used for: that attempt would sort
  // the window, look at its ties and give way to the doubling rounds, which then sort the window again)
  if (dc_state == 1 && c.dc_mode == 1 && !c.dc.valid && rp.longest >= 4096) dc_state = -1;
  c.window_long_run = rp.longest >= 4096;  // (the doubling rounds then step over runs: RunJump)
  for (;;) {
    const i64 e = dc_state == 1 ? b : b + slack;
    // a text that is still being uploaded: the window, and what the direct comparisons may read beyond it
    c.wait_text((u64)e + DF_DEPTH + 128);
    int redo = 0;
    if (sort_window(c, a, b, e, E, sa_out, repeats, dc_state, &redo, &rp)) return;
    if (redo == REDO_DC) {
      dc_state = 1;
      continue;
    }
    if (redo == REDO_NO_DC) {
      dc_state = -1;
      continue;
    }
    if (e >= c.n) fail(BWT_HIP_EINTERNAL, "suffix sort overflowed a window that ends at n");
    if (dc_state == 1) {  // (ties so sparse that the rounds ran instead, and looked past the leaf's end)
      dc_state = -1;
      continue;
    }
    slack = slack * 8;
    if (slack > c.n - b) slack = c.n - b;
  }
}

DBuf<u32> leaf_suffix_array(Ctx &c, i64 a, i64 b) {
  DBuf<u32> out;
  run_leaf_sort(c, a, b, nullptr, &out);
  return out;
}

struct SampFilledIn {
  const u64 *samp;
  __device__ __forceinline__ u32 operator()(size_t i) const { return samp[i] != ~0ull; }
};
struct SampCompactOut {
  const u64 *samp;
  u32 *rank_out, *pos_out;
  u32 *total;
  size_t n;
  u32 rank_base = 0;
  __device__ __forceinline__ void operator()(size_t i, u32 ex, u32 in) const {
    if (in != ex) {
      const u64 x = samp[i];
      rank_out[ex] = (u32)x + rank_base;
      pos_out[ex] = (u32)(x >> 32);
    }
    if (i + 1 == n) *total = in;
  }
};

i64 compact_sample_slots(Ctx &c, const u64 *slots, i64 n_slots, u32 rank_base, u32 *rank_out, u32 *pos_out) {
  if (n_slots <= 0) return 0;
  u32 *d_total = c.d_scalars + 8;
  scan_generic<u32, SumU32>(c, (size_t)n_slots, SampFilledIn{slots}, SampCompactOut{slots, rank_out, pos_out, d_total, (size_t)n_slots, rank_base});
  return (i64)read_u32(c, d_total);
}

bool build_leaf_share(Ctx &c, i64 a, i64 b, i64 ns_req, int part, int parts, LeafShare &out) {
  const i64 m = b - a;
  if (!c.T) fail(BWT_HIP_EINVAL, "no text set");
  if (a < 0 || b > c.n || m < 2 || parts < 1 || part < 0 || part >= parts) fail(BWT_HIP_EINVAL, "bad leaf share %d of %d over [%ld,%ld)", part, parts, (long)a, (long)b);
  const i64 ns = ns_req < m ? ns_req : m;
  if (ns < 2) fail(BWT_HIP_EINVAL, "ssa_n_samples must be at least 2 (got %ld)", (long)ns_req);
  c.wait_text((u64)b + DF_DEPTH + 128);
  if (window_has_long_repeats(c, a, (u64)m, false)) return false;
  out.samp = DBuf<u64>(c, (size_t)ns);
  out.ns = ns;
  HIP_CHECK(hipMemsetAsync(out.samp.get(), 0xFF, (size_t)ns * 8, c.stream));
  u32 *special = c.d_scalars + 16;
  HIP_CHECK(hipMemsetAsync(special, 0xFF, 8, c.stream));
  EmitParams E;
  memset(&E, 0, sizeof(E));
  E.T = c.T;
  E.n = (u64)c.n;
  E.a = (u64)a;
  E.m = (u64)m;
  E.ns = ns;
  E.L = nullptr;  // made by the route once the share's length is known
  E.samp = out.samp.get();
  E.special = special;
  const u32 min_sym = (u32)(c.min_sym > 0 ? c.min_sym : 1);
  const int span = c.max_sym > (int)min_sym ? c.max_sym - (int)min_sym : 0;
  LeanShard sh;
  sh.part = part;
  sh.parts = parts;
  if (try_lean_route(c, a, b, b, (u64)m, min_sym, span, &E, nullptr, &sh) != 1) return false;
  out.L = std::move(sh.L);
  out.bounds = sh.bounds;
  HIP_CHECK(hipMemcpyAsync(c.h_scalars, special, 8, hipMemcpyDeviceToHost, c.stream));
  c.sync();
  out.s_rank = c.h_scalars[0] == 0xFFFFFFFFu ? -1 : (i64)c.h_scalars[0];
  out.e_rank = c.h_scalars[1] == 0xFFFFFFFFu ? -1 : (i64)c.h_scalars[1];
  return true;
}

// bwt_leaf of the reference (include/bwt.h:615-635)
Block *build_leaf(Ctx &c, i64 a, i64 b, i64 ns_req) {
  const i64 m = b - a;
  i64 ns = ns_req < m ? ns_req : m;  // include/bwt.h:607
  if (m >= 2 && ns < 2) fail(BWT_HIP_EINVAL, "ssa_n_samples must be at least 2 (got %ld)", (long)ns_req);
  if (m < 1) fail(BWT_HIP_EINVAL, "empty leaf");
  Block *blk = new Block();
  try {
    blk->a = a;
    blk->b = b;
    blk->L = DBuf<u8>(c, (size_t)m);
    DBuf<u64> sv(c, (size_t)ns);
    HIP_CHECK(hipMemsetAsync(sv.get(), 0xFF, (size_t)ns * 8, c.stream));
    u32 *special = c.d_scalars + 16;
    HIP_CHECK(hipMemsetAsync(special, 0xFF, 8, c.stream));
    EmitParams E;
    memset(&E, 0, sizeof(E));
    E.T = c.T;
    E.n = (u64)c.n;
    E.a = (u64)a;
    E.m = (u64)m;
    E.ns = ns;
    E.L = blk->L.get();
    E.samp = sv.get();
    E.special = special;
    run_leaf_sort(c, a, b, &E, nullptr);
    blk->samp_rank = DBuf<u32>(c, (size_t)ns);
    blk->samp_pos = DBuf<u32>(c, (size_t)ns);
    u32 *d_total = c.d_scalars + 8;
    scan_generic<u32, SumU32>(c, (size_t)ns, SampFilledIn{sv.get()},
                              SampCompactOut{sv.get(), blk->samp_rank.get(), blk->samp_pos.get(), d_total, (size_t)ns});
    HIP_CHECK(hipMemcpyAsync(c.h_scalars, special, 8, hipMemcpyDeviceToHost, c.stream));
    HIP_CHECK(hipMemcpyAsync(c.h_scalars + 2, d_total, 4, hipMemcpyDeviceToHost, c.stream));
    HIP_CHECK(hipMemcpyAsync(c.h_scalars + 3, c.T + (b - 1), 1, hipMemcpyDeviceToHost, c.stream));
    c.sync();
    if (c.h_scalars[0] == 0xFFFFFFFFu || c.h_scalars[1] == 0xFFFFFFFFu)
      fail(BWT_HIP_EINTERNAL, "leaf [%ld,%ld): suffix array is not a permutation of the range", (long)a, (long)b);
    blk->s_rank = c.h_scalars[0];
    blk->e_rank = c.h_scalars[1];
    blk->n_samp = c.h_scalars[2];
    blk->e_char = *(u8 *)&c.h_scalars[3];
  } catch (...) {
    delete blk;
    throw;
  }
  return blk;
}

}  // namespace bwtamd
