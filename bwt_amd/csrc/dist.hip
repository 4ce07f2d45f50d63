// Pieces of a merge that is spread over several GPUs (one process per GPU; the
// exchange between them is done by the caller).  A merge of a left block of m
// suffixes and a right block of r suffixes is cut into G parts by LEFT RANK: with
//   part(v) = (v * scale) >> 32,   scale = floor(2^32 * G / (m + 1)),
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

__device__ __forceinline__ u32 part_of(u32 v, u64 scale) { return (u32)(((u64)v * scale) >> 32); }

__global__ void __launch_bounds__(PT_THREADS)
    k_part_count(const u32 *__restrict__ in, u64 count, u64 scale, u32 parts, unsigned long long *__restrict__ counts) {
  __shared__ u32 cnt[PT_MAX_PARTS];
  const u32 tid = threadIdx.x;
  cnt[tid] = 0;
  __syncthreads();
  const u64 base = (u64)blockIdx.x * (PT_THREADS * PT_ITEMS);
#pragma unroll 4
  for (int k = 0; k < PT_ITEMS; k++) {
    u64 i = base + (u64)k * PT_THREADS + tid;
    if (i < count) atomicAdd(&cnt[part_of(in[i], scale)], 1u);
  }
  __syncthreads();
  if (tid < parts && cnt[tid]) atomicAdd(&counts[tid], (unsigned long long)cnt[tid]);
}

// out: the values grouped by part (order inside a part is arbitrary), each minus the first left rank of
// its part.  cursors[k] starts as the first output index of part k.
__global__ void __launch_bounds__(PT_THREADS)
    k_part_scatter(const u32 *__restrict__ in, u64 count, u64 scale, u32 parts, const u32 *__restrict__ part_lo,
                   unsigned long long *__restrict__ cursors, u32 *__restrict__ out) {
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
    if (i < count) slot[k] = atomicAdd(&cnt[part_of(v[k], scale)], 1u);
  }
  __syncthreads();
  if (tid < parts && cnt[tid]) start[tid] = atomicAdd(&cursors[tid], (unsigned long long)cnt[tid]);
  __syncthreads();
#pragma unroll
  for (int k = 0; k < PT_ITEMS; k++) {
    u64 i = base + (u64)k * PT_THREADS + tid;
    if (i < count) {
      u32 p = part_of(v[k], scale);
      out[start[p] + slot[k]] = v[k] - part_lo[p];
    }
  }
}

// first value v with part(v) >= k
static u64 part_lower_bound(u64 k, u64 scale) { return ((k << 32) + scale - 1) / scale; }

void partition_by_left_rank(Ctx &c, const u32 *in, u64 count, u64 m, u32 parts, u32 *out, i64 *counts_host) {
  if (parts < 1 || parts > PT_MAX_PARTS) fail(BWT_HIP_EINVAL, "partition: 1..%d parts", PT_MAX_PARTS);
  const u64 scale = (((u64)parts) << 32) / (m + 1);
  DBuf<unsigned long long> d(c, 2 * (size_t)parts);
  DBuf<u32> lo(c, parts);
  u32 h_lo[PT_MAX_PARTS];
  for (u32 k = 0; k < parts; k++) h_lo[k] = (u32)part_lower_bound(k, scale);
  HIP_CHECK(hipMemsetAsync(d.get(), 0, 2 * (size_t)parts * sizeof(unsigned long long), c.stream));
  HIP_CHECK(hipMemcpyAsync(lo.get(), h_lo, parts * sizeof(u32), hipMemcpyHostToDevice, c.stream));
  const u32 nblk = ceil_div(count, PT_THREADS * PT_ITEMS);
  Phase ph(c, "dist_partition", (double)count * 12);
  unsigned long long h_cnt[PT_MAX_PARTS];
  if (count > 0) {
    hipLaunchKernelGGL(k_part_count, dim3(nblk), dim3(PT_THREADS), 0, c.stream, in, count, scale, parts, d.get());
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
    hipLaunchKernelGGL(k_part_scatter, dim3(nblk), dim3(PT_THREADS), 0, c.stream, in, count, scale, parts, lo.get(),
                       d.get() + parts, out);
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
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const u32 b = (bits >> (8 * q)) & 255u;
      o[q] = ((b & 3u) | (((b >> 2) & 3u) << 8) | (((b >> 4) & 3u) << 16) | ((b >> 6) << 24)) + min_sym * 0x01010101u;
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

void partition_bounds(u64 m, u32 parts, i64 *lo_out) {
  const u64 scale = (((u64)parts) << 32) / (m + 1);
  for (u32 k = 0; k < parts; k++) lo_out[k] = (i64)part_lower_bound(k, scale);
  lo_out[parts] = (i64)(m + 1);
}

// ----------------------------------------------------------------------------
// this GPU's share of the merged block's samples (reference: include/bwt.h:366-453
// resample_sa; selection rule as in k_resample of merge.hip).  Sample i of the
// concatenation (left samples by position, then right samples by position) is
// handled by the GPU that knows its new rank: a left sample by the owner of its
// left rank, a right sample by the GPU whose chains covered its position.  Kept
// samples are written to their slot; the caller sums the arrays of all GPUs.
// Ranks are block-relative plus the block's first position, as everywhere in the C interface.
// ----------------------------------------------------------------------------
__device__ __forceinline__ u64 taken_before_d(u64 i, u64 A, u64 B) {
  if (i == 0) return 0;
  if (B == 0) return 1;
  u64 t = 1 + div_floor_u64((i - 1) * A, B);
  return t < i ? t : i;
}

__global__ void k_resample_part(ResamplePart R) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= R.nl + R.nr) return;
  u64 t0 = taken_before_d(i, R.A, R.B), t1 = taken_before_d(i + 1, R.A, R.B);
  if (t1 == t0) return;
  if (i < R.nl) {
    u64 rk = (u64)(R.l_rank[i] - R.a);
    if (rk < R.i_lo || rk >= R.i_hi) return;
    R.o_rank[t0] = R.a + (i64)(rk + R.right_before + R.S_local[rk - R.i_lo]);
    R.o_pos[t0] = R.l_pos[i];
  } else {
    u64 k = i - R.nl;
    u64 p = (u64)R.r_pos[k];
    if (p < R.p_lo || p >= R.p_hi) return;
    R.o_rank[t0] = R.a + (R.r_rank[k] - R.cc) + (i64)R.g_local[p - R.p_lo];
    R.o_pos[t0] = R.r_pos[k];
  }
}

void resample_part(Ctx &c, const ResamplePart &R) {
  u64 nn = R.nl + R.nr;
  if (nn == 0) return;
  hipLaunchKernelGGL(k_resample_part, dim3(ceil_div(nn, 256)), dim3(256), 0, c.stream, R);
  HIP_CHECK(hipGetLastError());
}

}  // namespace bwtamd
