// Pieces of a merge that is spread over several GPUs (one process per GPU; the
// exchange between them is done by the caller).  A merge of a left block of m
// suffixes and a right block of r suffixes is cut into G parts by LEFT RANK: with
//   part(v) = ((v >> shift) * scale) >> 32,   scale = floor(2^32 * G / ((m >> shift) + 1)),
// (shift = 0, or 13 where the g values travel grouped by gap bucket: parts then begin at multiples of 8192)
// part k owns the left ranks i with part(i) == k and the right suffixes j whose
// g_j (number of smaller left suffixes) has part(g_j) == k.  Those right suffixes
// are consecutive in the right block's order, and part k's output -- its left
// and right symbols interleaved -- is a contiguous slice of the merged block; the
// slices of parts 0, 1, ... follow each other.  g_j depends on the left block and
// the text only, so the LF chains of the right positions can be divided among the
// GPUs in any way.  (reference: include/bwt.h:730-847, build_gap + bwt_merge)
#include "device_utils.h"
#include "internal.h"

namespace bwtamd {

constexpr int PT_THREADS = 256;
constexpr int PT_ITEMS = 16;
constexpr int PT_MAX_PARTS = 256;

__device__ __forceinline__ u32 part_of(u32 v, u64 scale, int shift) { return (u32)(((u64)(v >> shift) * scale) >> 32); }

__global__ void __launch_bounds__(PT_THREADS)
    k_part_count(const u32 *__restrict__ in, u64 count, u64 scale, int shift, u32 parts, unsigned long long *__restrict__ counts) {
  __shared__ u32 cnt[PT_MAX_PARTS];
  const u32 tid = threadIdx.x;
  cnt[tid] = 0;
  __syncthreads();
  const u64 base = (u64)blockIdx.x * (PT_THREADS * PT_ITEMS);
#pragma unroll 4
  for (int k = 0; k < PT_ITEMS; k++) {
    u64 i = base + (u64)k * PT_THREADS + tid;
    if (i < count) atomicAdd(&cnt[part_of(in[i], scale, shift)], 1u);
  }
  __syncthreads();
  if (tid < parts && cnt[tid]) atomicAdd(&counts[tid], (unsigned long long)cnt[tid]);
}

// out: the values grouped by part (order inside a part is arbitrary), each minus the first left rank of
// its part.  cursors[k] starts as the first output index of part k.
__global__ void __launch_bounds__(PT_THREADS)
    k_part_scatter(const u32 *__restrict__ in, u64 count, u64 scale, int shift, u32 parts, const u32 *__restrict__ part_lo,
                   unsigned long long *__restrict__ cursors, u32 *__restrict__ out, u32 *__restrict__ perm) {
  __shared__ u32 cnt[PT_MAX_PARTS];
  __shared__ unsigned long long start[PT_MAX_PARTS];
  const u32 tid = threadIdx.x;
  cnt[tid] = 0;
  __syncthreads();
  const u64 base = (u64)blockIdx.x * (PT_THREADS * PT_ITEMS);
  u32 v[PT_ITEMS], slot[PT_ITEMS];
#pragma unroll
  for (int k = 0; k < PT_ITEMS; k++) {
    u64 i = base + (u64)k * PT_THREADS + tid;
    v[k] = i < count ? in[i] : 0u;
    if (i < count) slot[k] = atomicAdd(&cnt[part_of(v[k], scale, shift)], 1u);
  }
  __syncthreads();
  if (tid < parts && cnt[tid]) start[tid] = atomicAdd(&cursors[tid], (unsigned long long)cnt[tid]);
  __syncthreads();
#pragma unroll
  for (int k = 0; k < PT_ITEMS; k++) {
    u64 i = base + (u64)k * PT_THREADS + tid;
    if (i < count) {
      u32 p = part_of(v[k], scale, shift);
      out[start[p] + slot[k]] = v[k] - part_lo[p];
      if (perm) perm[start[p] + slot[k]] = (u32)i;  // where the value came from
    }
  }
}

// first value v with part(v) >= k
static u64 part_lower_bound(u64 k, u64 scale) { return ((k << 32) + scale - 1) / scale; }

void partition_by_left_rank(Ctx &c, const u32 *in, u64 count, u64 m, u32 parts, u32 *out, i64 *counts_host, u32 *perm, int shift) {
  if (parts < 1 || parts > PT_MAX_PARTS) fail(BWT_HIP_EINVAL, "partition: 1..%d parts", PT_MAX_PARTS);
  const u64 scale = (((u64)parts) << 32) / ((m >> shift) + 1);
  DBuf<unsigned long long> d(c, 2 * (size_t)parts);
  DBuf<u32> lo(c, parts);
  u32 h_lo[PT_MAX_PARTS];
  for (u32 k = 0; k < parts; k++) h_lo[k] = (u32)(part_lower_bound(k, scale) << shift);
  HIP_CHECK(hipMemsetAsync(d.get(), 0, 2 * (size_t)parts * sizeof(unsigned long long), c.stream));
  HIP_CHECK(hipMemcpyAsync(lo.get(), h_lo, parts * sizeof(u32), hipMemcpyHostToDevice, c.stream));
  const u32 nblk = ceil_div(count, PT_THREADS * PT_ITEMS);
  Phase ph(c, "dist_partition", (double)count * 12);
  unsigned long long h_cnt[PT_MAX_PARTS];
  if (count > 0) {
    hipLaunchKernelGGL(k_part_count, dim3(nblk), dim3(PT_THREADS), 0, c.stream, in, count, scale, shift, parts, d.get());
    HIP_CHECK(hipGetLastError());
  }
  HIP_CHECK(hipMemcpyAsync(h_cnt, d.get(), parts * sizeof(unsigned long long), hipMemcpyDeviceToHost, c.stream));
  c.sync();  // also: h_lo is on the stack
  unsigned long long h_cur[PT_MAX_PARTS], acc = 0;
  for (u32 k = 0; k < parts; k++) {
    counts_host[k] = (i64)h_cnt[k];
    h_cur[k] = acc;
    acc += h_cnt[k];
  }
  if (acc != count) fail(BWT_HIP_EINTERNAL, "partition: counted %llu of %lu values", acc, (unsigned long)count);
  if (count > 0) {
    HIP_CHECK(hipMemcpyAsync(d.get() + parts, h_cur, parts * sizeof(unsigned long long), hipMemcpyHostToDevice, c.stream));
    hipLaunchKernelGGL(k_part_scatter, dim3(nblk), dim3(PT_THREADS), 0, c.stream, in, count, scale, shift, parts, lo.get(),
                       d.get() + parts, out, perm);
    HIP_CHECK(hipGetLastError());
    c.sync();  // h_cur is on the stack
  }
}

// ----------------------------------------------------------------------------
// Symbols on the wire between GPUs.  When the text's symbols span at most four consecutive values (DNA), a
// stretch of L travels as two bits per symbol: code = (symbol - min_sym) & 3, four symbols per byte, the
// first in the lowest bits.  Source and destination stretches start at arbitrary symbol offsets (slices of a
// block are cut by rank), so both kernels work from a byte pointer of any alignment; the packed stream itself
// always starts at a byte.  A symbol outside the range (the terminator that the block starting at text position
// 0 holds at s_rank) packs as garbage and is put back by the receiver.
// ----------------------------------------------------------------------------
__global__ void k_wire_pack2(const u8 *__restrict__ src, u64 count, u8 *__restrict__ packed, u32 min_sym) {
  const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  const u64 i0 = t * 16;
  if (i0 >= count) return;
  if (i0 + 16 <= count) {
    u64 v[2];
    __builtin_memcpy(v, src + i0, 16);  // any alignment
    u32 w = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const u32 in = (u32)(v[q >> 1] >> (32 * (q & 1)));
      // (symbol - min_sym) mod 4 per byte; the 4 added to every byte keeps a borrow from crossing into its neighbour
      const u32 x = ((in & 0x03030303u) | 0x04040404u) - (min_sym & 3u) * 0x01010101u;
      w |= ((x & 3u) | (((x >> 8) & 3u) << 2) | (((x >> 16) & 3u) << 4) | (((x >> 24) & 3u) << 6)) << (8 * q);
    }
    __builtin_memcpy(packed + t * 4, &w, 4);
  } else {
    for (u64 i = i0; i < count; i += 4) {
      u32 b = 0;
      for (u64 j = i; j < i + 4 && j < count; j++) b |= ((u32)(src[j] - min_sym) & 3u) << (2 * (j - i));
      packed[i >> 2] = (u8)b;
    }
  }
}

// dst[i] = min_sym + code i of the packed stream, i in [0, count).  A thread owns one 16-byte aligned group of
// destination bytes; groups that the stretch covers only partly (its two ends) are written byte by byte.
__global__ void k_wire_unpack2(const u8 *__restrict__ packed, u8 *__restrict__ dst, u64 count, u32 min_sym) {
  const u64 lead = (u64)(reinterpret_cast<uintptr_t>(dst) & 15);  // bytes between the aligned group start and dst
  const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (t * 16 >= lead + count) return;
  const i64 s0 = (i64)(t * 16) - (i64)lead;  // stream index of the group's first byte
  if (s0 >= 0 && (u64)s0 + 16 <= count) {
    u64 w;
    __builtin_memcpy(&w, packed + (s0 >> 2), 8);  // 32 + 6 bits are needed; the caller pads the stream by 8 bytes
    const u32 bits = (u32)(w >> (2 * (s0 & 3)));
    u32 o[4];
    // min_sym + code modulo 256 in every byte: the terminator travels as code (0 - min_sym) mod 4, which with min_sym
    // above 252 sums to 256 and more, and a plain 32-bit sum would carry that into the neighbouring symbol.  The low
    // seven bits are added (3 + 127 carries nowhere), the top bit by exclusive or.
    const u32 M = min_sym * 0x01010101u;
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const u32 b = (bits >> (8 * q)) & 255u;
      const u32 codes = (b & 3u) | (((b >> 2) & 3u) << 8) | (((b >> 4) & 3u) << 16) | ((b >> 6) << 24);
      o[q] = (codes + (M & 0x7F7F7F7Fu)) ^ (M & 0x80808080u);
    }
    *reinterpret_cast<uint4 *>(dst + s0) = make_uint4(o[0], o[1], o[2], o[3]);
  } else {
    const i64 lo = s0 < 0 ? 0 : s0, hi = s0 + 16 < (i64)count ? s0 + 16 : (i64)count;
    for (i64 i = lo; i < hi; i++) dst[i] = (u8)(min_sym + ((packed[i >> 2] >> (2 * (i & 3))) & 3u));
  }
}

bool wire_packable(const Ctx &c) { return c.min_sym >= 1 && c.max_sym >= c.min_sym && c.max_sym - c.min_sym <= 3; }

void wire_pack2(Ctx &c, const u8 *src, u64 count, u8 *packed) {
  if (count == 0) return;
  hipLaunchKernelGGL(k_wire_pack2, dim3(ceil_div((count + 15) / 16, 256)), dim3(256), 0, c.stream, src, count, packed, (u32)c.min_sym);
  HIP_CHECK(hipGetLastError());
}

void wire_unpack2(Ctx &c, const u8 *packed, u64 count, u8 *dst) {
  if (count == 0) return;
  const u64 lead = (u64)(reinterpret_cast<uintptr_t>(dst) & 15);
  hipLaunchKernelGGL(k_wire_unpack2, dim3(ceil_div((lead + count + 15) / 16, 256)), dim3(256), 0, c.stream, packed, dst, count, (u32)c.min_sym);
  HIP_CHECK(hipGetLastError());
}

void partition_bounds(u64 m, u32 parts, i64 *lo_out, int shift) {
  const u64 scale = (((u64)parts) << 32) / ((m >> shift) + 1);
  for (u32 k = 0; k < parts; k++) lo_out[k] = (i64)(part_lower_bound(k, scale) << shift);
  lo_out[parts] = (i64)(m + 1);
}

int partition_shift(const Ctx &c, u64 m) { return gap_grouping_applies(c, m) ? 13 : 0; }

// samples kept among the first i of a merge's sample sequence (left samples by position, then right samples by
// position): the reference's serial selection (include/bwt.h:412, :436) in closed form, as in k_resample of merge.hip
__device__ __forceinline__ u64 taken_before_d(u64 i, u64 A, u64 B) {
  if (i == 0) return 0;
  if (B == 0) return 1;
  u64 t = 1 + div_floor_u64((i - 1) * A, B);
  return t < i ? t : i;
}

// ----------------------------------------------------------------------------
// The samples stay where they are.  Every member of a block keeps the samples whose positions lie
// in the text range it built its own subtree on (sorted by position; members in order = the block's samples in
// order).  In a merge a sample does not move: its holder asks the member that knows for the one number the new
// rank needs -- for a left sample the owner of its left rank's part (prefix sums), for a right sample the member
// whose chains covered its position (g) -- gets it back, and thins its own samples by the reference's rule, which
// depends on a sample's index in the whole sequence only.  4 bytes out and 4 bytes back per sample.
// ----------------------------------------------------------------------------
__global__ void k_rel_u32(const u32 *__restrict__ in, u64 n, u32 base, u32 *__restrict__ out) {
  const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = in[i] - base;
}

// first index of the ascending array v[0:n) whose value is >= bound[t], for t in [0, nb)
__global__ void k_lower_bounds(const u32 *__restrict__ v, u32 n, const u32 *__restrict__ bound, u32 nb, u32 *__restrict__ out) {
  const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nb) return;
  u32 lo = 0, hi = n;
  const u32 b = bound[t];
  while (lo < hi) {
    const u32 mid = lo + (hi - lo) / 2;
    if (v[mid] < b) lo = mid + 1;
    else hi = mid;
  }
  out[t] = lo;
}

// queries of a right holder: sample i asks member d, the one with first[d] <= i < first[d+1], for g at pos[i] - bound[d]
__global__ void k_right_queries(const u32 *__restrict__ pos, u32 n, const u32 *__restrict__ bound, const u32 *__restrict__ first,
                                u32 members, u32 *__restrict__ q) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  u32 lo = 0, hi = members;  // the last d with first[d] <= i
  while (hi - lo > 1) {
    const u32 mid = (lo + hi) / 2;
    if (first[mid] <= i) lo = mid;
    else hi = mid;
  }
  q[i] = pos[i] - bound[lo];
}

// answers of one member: n_left look-ups into its part's prefix sums, then n_right into its chains' g values
__global__ void k_sample_answers(const u32 *__restrict__ q, u64 n_left, u64 n_right, const u32 *__restrict__ S_local, u32 left_base,
                                 const u32 *__restrict__ g_local, u32 *__restrict__ ans) {
  const u64 x = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (x >= n_left + n_right) return;
  const u32 v = q[x];
  // left: new rank = a + rk + (right suffixes in lower parts) + S'[rk - i_lo], with left_base = a + i_lo + right_before
  ans[x] = x < n_left ? left_base + v + S_local[v] : g_local[v];
}

__global__ void k_sample_place(SamplePlace P) {
  const u64 x = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (x >= P.n) return;
  const u64 i = P.perm ? P.perm[x] : x;  // answer x belongs to the holder's sample i
  const u64 gi = P.first_index + i;
  const u64 t0 = taken_before_d(gi, P.A, P.B), t1 = taken_before_d(gi + 1, P.A, P.B);
  if (t1 == t0) return;
  const u64 slot = t0 - P.first_slot;
  // left: the answer is the new rank; right: rank in the right block + g, include/bwt.h:778-781 then :437
  P.o_rank[slot] = P.right_shift ? P.rank[i] - P.right_shift + P.ans[x] : P.ans[x];
  P.o_pos[slot] = P.pos[i];
}

u64 samples_taken_before(u64 i, u64 A, u64 B) {
  if (i == 0) return 0;
  if (B == 0) return 1;
  const u64 t = 1 + (u64)(((unsigned __int128)(i - 1) * A) / B);
  return t < i ? t : i;
}

void samples_route_left(Ctx &c, const u32 *rank_abs, u64 n, i64 a, u64 m, u32 members, u32 *queries, u32 *perm, i64 *counts, int shift) {
  DBuf<u32> rel(c, n ? n : 1);
  if (n) {
    hipLaunchKernelGGL(k_rel_u32, dim3(ceil_div(n, 256)), dim3(256), 0, c.stream, rank_abs, n, (u32)a, rel.get());
    HIP_CHECK(hipGetLastError());
  }
  partition_by_left_rank(c, rel.get(), n, m, members, queries, counts, perm, shift);
  c.sync();
}

void samples_route_right(Ctx &c, const u32 *pos, u64 n, i64 cpos, u64 r, u32 members, u32 *queries, i64 *counts) {
  if (members < 1 || members > PT_MAX_PARTS) fail(BWT_HIP_EINVAL, "samples: 1..%d members", PT_MAX_PARTS);
  u32 h_bound[PT_MAX_PARTS + 1], h_first[PT_MAX_PARTS + 1];
  for (u32 d = 0; d <= members; d++) h_bound[d] = (u32)((u64)cpos + (u64)(((unsigned __int128)d * r) / members));
  DBuf<u32> d_tab(c, 2 * (size_t)(members + 1));
  HIP_CHECK(hipMemcpyAsync(d_tab.get(), h_bound, (members + 1) * sizeof(u32), hipMemcpyHostToDevice, c.stream));
  hipLaunchKernelGGL(k_lower_bounds, dim3(ceil_div(members + 1, 256)), dim3(256), 0, c.stream, pos, (u32)n, (const u32 *)d_tab.get(),
                     members + 1, d_tab.get() + (members + 1));
  HIP_CHECK(hipGetLastError());
  HIP_CHECK(hipMemcpyAsync(h_first, d_tab.get() + (members + 1), (members + 1) * sizeof(u32), hipMemcpyDeviceToHost, c.stream));
  c.sync();
  if (h_first[0] != 0 || h_first[members] != (u32)n) fail(BWT_HIP_EINTERNAL, "samples: a right sample lies outside [%ld, %ld)", (long)cpos, (long)(cpos + (i64)r));
  for (u32 d = 0; d < members; d++) counts[d] = (i64)h_first[d + 1] - (i64)h_first[d];
  if (n) {
    hipLaunchKernelGGL(k_right_queries, dim3(ceil_div(n, 256)), dim3(256), 0, c.stream, pos, (u32)n, (const u32 *)d_tab.get(),
                       (const u32 *)(d_tab.get() + (members + 1)), members, queries);
    HIP_CHECK(hipGetLastError());
  }
  c.sync();  // the table dies here
}

void samples_answer(Ctx &c, const u32 *q, u64 n_left, u64 n_right, const u32 *S_local, u64 i_lo, u64 right_before, i64 a,
                    const u32 *g_local, u32 *ans) {
  if (n_left + n_right == 0) return;
  hipLaunchKernelGGL(k_sample_answers, dim3(ceil_div(n_left + n_right, 256)), dim3(256), 0, c.stream, q, n_left, n_right, S_local,
                     (u32)((u64)a + i_lo + right_before), g_local, ans);
  HIP_CHECK(hipGetLastError());
}

void samples_place(Ctx &c, const SamplePlace &P) {
  if (P.n == 0) return;
  hipLaunchKernelGGL(k_sample_place, dim3(ceil_div(P.n, 256)), dim3(256), 0, c.stream, P);
  HIP_CHECK(hipGetLastError());
}

}  // namespace bwtamd
