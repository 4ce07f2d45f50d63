// Leaf suffix array: positions a..b-1 of the text ordered by their FULL-TEXT
// suffixes (what the reference's idx_less comparison sort produces,
// include/suffix_array.h:14-49), by radix sort + prefix doubling.
//
//   round 0   key = first K0 symbols packed into 64 bits; LSD radix sort of
//             (key, position) over the window [a, e), e = b + slack
//   tie list  suffixes whose key is shared form groups; only those stay active
//   text rnd  active suffixes are ordered inside their group by the next K1
//             symbols read from the text (no inverse array needed; on random
//             text this ends the sort)
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
#include "device_utils.h"
#include "internal.h"
#include "scan_tmpl.h"

namespace bwtamd {

// ----------------------------------------------------------------------------
// round 0 keys
// ----------------------------------------------------------------------------
constexpr int KB_THREADS = 256;
constexpr int KB_ITEMS = 16;
constexpr int KB_TILE = KB_THREADS * KB_ITEMS;

// keys[i] = symbols T[a+i .. a+i+K0) packed w bits each, first symbol highest,
// left-aligned in 64 bits; positions >= n read as 0.  sa[i] = a + i.
__global__ void __launch_bounds__(KB_THREADS)
    k_build_keys(const u8 *__restrict__ T, u64 n, u64 a, u64 N, int w, int K0, u64 *__restrict__ keys,
                 u32 *__restrict__ sa) {
  __shared__ u8 s[KB_TILE + 64];
  const u32 tid = threadIdx.x;
  const u64 tile0 = (u64)blockIdx.x * KB_TILE;
  for (u32 i = tid; i < KB_TILE + 64; i += KB_THREADS) {
    u64 p = a + tile0 + i;
    s[i] = p < n ? T[p] : (u8)0;
  }
  __syncthreads();
  const u32 t0 = tid * KB_ITEMS;
  const int used = K0 * w;
  const u64 maskall = used == 64 ? ~0ull : ((1ull << used) - 1);
  const int lsh = 64 - used;
  u64 key = 0;
  for (int k = 0; k < K0; k++) key = (key << w) | s[t0 + k];
#pragma unroll
  for (int j = 0; j < KB_ITEMS; j++) {
    u64 idx = tile0 + t0 + j;
    if (idx < N) {
      keys[idx] = key << lsh;
      sa[idx] = (u32)(a + idx);
    }
    key = ((key << w) | s[t0 + j + K0]) & maskall;
  }
}

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

struct GroupIn {
  const u64 *key;
  size_t n;
  __device__ __forceinline__ TieHead operator()(size_t i) const {
    u64 k = key[i];
    bool head = (i == 0) || key[i - 1] != k;
    bool next_head = (i + 1 == n) || key[i + 1] != k;
    return TieHead{(u32)!(head && next_head), head ? (u32)(i + 1) : 0u};
  }
};

// after round 0: emit the active list (slot = rank index, pos, group = head rank)
struct FirstListOut {
  const u32 *sa;
  u32 *slot, *pos, *grp;
  u32 *total;
  size_t n;
  __device__ __forceinline__ void operator()(size_t i, TieHead ex, TieHead in) const {
    if (in.cnt != ex.cnt) {
      u32 k = ex.cnt;
      slot[k] = (u32)i;
      pos[k] = sa[i];
      grp[k] = in.hmax - 1;
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

// doubling round: composite = group << 32 | (rank of suffix pos+h, plus one; 0 past the window)
__global__ void k_comp_isa(const u32 *__restrict__ isa, u64 a, u64 b, u64 e, u64 n,
                           const u32 *__restrict__ pos, const u32 *__restrict__ grp, u32 M, u64 h,
                           u64 *__restrict__ comp, u32 *__restrict__ pout, u32 *__restrict__ overflow) {
  u32 k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= M) return;
  u32 p = pos[k];
  u64 q = (u64)p + h;
  u32 sec = 0;
  if (q < e) {
    sec = isa[q - a] + 1;
  } else if (p < b && e < n) {
    *overflow = 1;  // a leaf suffix needs text ranks beyond the window
  }
  comp[k] = ((u64)grp[k] << 32) | sec;
  pout[k] = p;
}

__global__ void k_isa_init(const u32 *__restrict__ sa, u32 a, u32 N, u32 *__restrict__ isa) {
  u32 r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r < N) isa[sa[r] - a] = r;
}

__global__ void k_isa_fix(const u32 *__restrict__ pos, const u32 *__restrict__ grp, u32 M, u32 a,
                          u32 *__restrict__ isa) {
  u32 k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < M) isa[pos[k] - a] = grp[k];
}

struct LeafFilterIn {
  const u32 *sa;
  u32 b;
  __device__ __forceinline__ u32 operator()(size_t i) const { return sa[i] < b; }
};
struct LeafFilterOut {
  const u32 *sa;
  u32 *out;
  __device__ __forceinline__ void operator()(size_t i, u32 ex, u32 in) const {
    if (in != ex) out[ex] = sa[i];
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

// Sorts the window [a,e); on success the leaf suffix array (b-a entries) is in *out.
static bool sort_window(Ctx &c, i64 a, i64 b, i64 e, DBuf<u32> *out, i64 *depth_out) {
  const u64 N = (u64)(e - a);
  const int w = c.sym_bits;
  const int K0 = 64 / w;
  const int K1 = 32 / w;
  const int low_bit = 64 - K0 * w;
  DBuf<u64> kb0(c, N), kb1(c, N);
  DBuf<u32> sb0(c, N), sb1(c, N);
  u64 *keys[2] = {kb0.get(), kb1.get()};
  u32 *vals[2] = {sb0.get(), sb1.get()};
  {
    Phase ph(c, "leaf_key_build", (double)N * 13);
    hipLaunchKernelGGL(k_build_keys, dim3(ceil_div(N, KB_TILE)), dim3(KB_THREADS), 0, c.stream, c.T, (u64)c.n,
                       (u64)a, N, w, K0, keys[0], vals[0]);
    HIP_CHECK(hipGetLastError());
  }
  int cur;
  {
    int passes = (64 - low_bit + 7) / 8;
    Phase ph(c, "leaf_radix_sort", (double)N * passes * 32);
    c.sort_tag = "leaf:";
    cur = radix_sort_u64(c, keys, vals, N, low_bit, 64, 0);
    c.sort_tag = "";
  }
  u32 *sa = vals[cur];
  u32 *d_total = c.d_scalars + 8;
  u32 *d_overflow = c.d_scalars + 9;
  HIP_CHECK(hipMemsetAsync(d_overflow, 0, sizeof(u32), c.stream));

  // first active list lives in the sort's scratch buffers
  u32 *wl_slot = (u32 *)keys[cur ^ 1];
  u32 *wl_pos = wl_slot + N;
  u32 *wl_grp = vals[cur ^ 1];
  u32 M;
  {
    Phase ph(c, "leaf_group", (double)N * 16);
    scan_generic<TieHead, TieHeadOp>(c, N, GroupIn{keys[cur], N},
                                     FirstListOut{sa, wl_slot, wl_pos, wl_grp, d_total, N});
    M = read_u32(c, d_total);
  }
  u64 h = (u64)K0;
  int round = 0;
  DBuf<u32> isa;
  DBuf<u32> nl_slot, nl_pos, nl_grp;  // owners of the current list after the first round
  const int gbits = bits_for(N);
  while (M > 0) {
    Phase ph(c, round == 0 ? "leaf_text_round" : "leaf_doubling_round", (double)M * 8 * 32);
    DBuf<u64> ck0(c, M), ck1(c, M);
    DBuf<u32> cp0(c, M), cp1(c, M);
    if (round == 0) {
      hipLaunchKernelGGL(k_comp_text, dim3(ceil_div(M, 256)), dim3(256), 0, c.stream, c.T, (u64)c.n, wl_pos,
                         wl_grp, M, (u32)h, w, K1, ck0.get(), cp0.get());
    } else {
      if (!isa.get()) {
        isa = DBuf<u32>(c, N);
        hipLaunchKernelGGL(k_isa_init, dim3(ceil_div(N, 256)), dim3(256), 0, c.stream, sa, (u32)a, (u32)N,
                           isa.get());
        hipLaunchKernelGGL(k_isa_fix, dim3(ceil_div(M, 256)), dim3(256), 0, c.stream, wl_pos, wl_grp, M,
                           (u32)a, isa.get());
      }
      hipLaunchKernelGGL(k_comp_isa, dim3(ceil_div(M, 256)), dim3(256), 0, c.stream, isa.get(), (u64)a, (u64)b,
                         (u64)e, (u64)c.n, wl_pos, wl_grp, M, h, ck0.get(), cp0.get(), d_overflow);
    }
    HIP_CHECK(hipGetLastError());
    u64 *ck[2] = {ck0.get(), ck1.get()};
    u32 *cp[2] = {cp0.get(), cp1.get()};
    c.sort_tag = "refine:";
    int cc = radix_sort_u64(c, ck, cp, M, 0, 32 + gbits);
    c.sort_tag = "";
    DBuf<u32> ns(c, M), np(c, M), ng(c, M);
    scan_generic<TieHead, TieHeadOp>(
        c, M, GroupIn{ck[cc], M},
        RefineOut{wl_slot, cp[cc], sa, isa.get(), (u32)a, ns.get(), np.get(), ng.get(), d_total, M});
    M = read_u32(c, d_total);
    nl_slot = std::move(ns);
    nl_pos = std::move(np);
    nl_grp = std::move(ng);
    wl_slot = nl_slot.get();
    wl_pos = nl_pos.get();
    wl_grp = nl_grp.get();
    h = (round == 0) ? (u64)(K0 + K1) : 2 * h;
    round++;
    if (round > 1 && read_u32(c, d_overflow)) return false;
    if (round > 80) fail(BWT_HIP_EINTERNAL, "suffix sort did not converge");
  }
  *depth_out = (i64)h;
  if (e == b) {
    *out = std::move(cur ? sb1 : sb0);
  } else {
    Phase ph(c, "leaf_window_filter", (double)N * 8 + (double)(b - a) * 4);
    DBuf<u32> res(c, (size_t)(b - a));
    scan_generic<u32, SumU32>(c, N, LeafFilterIn{sa, (u32)b}, LeafFilterOut{sa, res.get()});
    *out = std::move(res);
  }
  return true;
}

DBuf<u32> leaf_suffix_array(Ctx &c, i64 a, i64 b) {
  if (!c.T) fail(BWT_HIP_EINVAL, "no text set");
  if (a < 0 || b > c.n || a >= b) fail(BWT_HIP_EINVAL, "bad leaf range [%ld,%ld) for n=%ld", (long)a, (long)b, (long)c.n);
  i64 slack = 0;
  if (b < c.n) {
    slack = (b - a) / 256;
    if (slack < 4096) slack = 4096;
    if (slack > c.n - b) slack = c.n - b;
  }
  for (;;) {
    DBuf<u32> out;
    i64 depth = 0;
    if (sort_window(c, a, b, b + slack, &out, &depth)) return out;
    if (b + slack >= c.n) fail(BWT_HIP_EINTERNAL, "suffix sort overflowed a window that ends at n");
    slack = slack * 8;
    if (slack > c.n - b) slack = c.n - b;
  }
}

}  // namespace bwtamd
