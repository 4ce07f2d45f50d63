// Device-side helpers shared by the kernels: 64-lane wave scans/reductions and
// block-wide combinations of them.  gfx950 only: a wave is 64 lanes.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bwtamd {

typedef uint8_t u8;
typedef uint16_t u16;
typedef uint32_t u32;
typedef uint64_t u64;
typedef int64_t i64;

#define WAVE 64

__device__ __forceinline__ u32 lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ u32 wave_id() { return threadIdx.x >> 6; }
__device__ __forceinline__ u64 lanemask_lt() { return (1ull << lane_id()) - 1ull; }

__device__ __forceinline__ u32 wave_reduce_sum(u32 v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__device__ __forceinline__ u32 wave_reduce_max(u32 v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = max(v, (u32)__shfl_xor(v, o, 64));
  return v;
}

// inclusive scans across the 64 lanes of a wave
__device__ __forceinline__ u32 wave_scan_incl_sum(u32 v) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    u32 t = __shfl_up(v, o, 64);
    if (lane_id() >= (u32)o) v += t;
  }
  return v;
}

__device__ __forceinline__ u32 wave_scan_incl_max(u32 v) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    u32 t = __shfl_up(v, o, 64);
    if (lane_id() >= (u32)o) v = max(v, t);
  }
  return v;
}

struct OpSum {
  __device__ __forceinline__ static u32 identity() { return 0; }
  __device__ __forceinline__ static u32 apply(u32 a, u32 b) { return a + b; }
  __device__ __forceinline__ static u32 wave_incl(u32 v) { return wave_scan_incl_sum(v); }
};
struct OpMax {
  __device__ __forceinline__ static u32 identity() { return 0; }
  __device__ __forceinline__ static u32 apply(u32 a, u32 b) { return max(a, b); }
  __device__ __forceinline__ static u32 wave_incl(u32 v) { return wave_scan_incl_max(v); }
};

// Block-wide inclusive scan of one value per thread (blockDim.x <= 1024, a
// multiple of 64).  `smem` must hold 17 words.  Returns the inclusive value;
// *block_total receives the combination of all threads' values.  Contains
// two __syncthreads(); every thread of the block must call it.
template <class Op>
__device__ __forceinline__ u32 block_scan_incl(u32 v, u32 *smem, u32 *block_total) {
  u32 incl = Op::wave_incl(v);
  u32 w = wave_id(), nw = blockDim.x >> 6;
  __syncthreads();  // smem may still be read from a previous call
  if (lane_id() == 63) smem[w] = incl;
  __syncthreads();
  u32 carry = Op::identity(), tot = Op::identity();
  for (u32 k = 0; k < nw; k++) {
    u32 s = smem[k];
    if (k < w) carry = Op::apply(carry, s);
    tot = Op::apply(tot, s);
  }
  *block_total = tot;
  return Op::apply(carry, incl);
}

// Which lanes of the wave hold the same BITS-bit digit as this lane.  lo/hi come in
// as the two halves of the mask of participating lanes and go out as the mask of
// this lane's peers (itself included).  One ballot per digit bit; per bit and
// half the update  m &= ~(ballot ^ my_bit_replicated)  is a single three-input
// bit operation on gfx950.
template <int BITS>
__device__ __forceinline__ void match_digit(u32 digit, u32 &lo, u32 &hi) {
#pragma unroll
  for (int b = 0; b < BITS; b++) {
    const int32_t bm = __builtin_amdgcn_sbfe((int32_t)digit, b, 1);  // all ones if bit b is set
    const u64 bal = __ballot(bm < 0);
    lo = __builtin_amdgcn_bitop3_b32(lo, (u32)bal, (u32)bm, 0x90);  // lo & ~(bal ^ bm)
    hi = __builtin_amdgcn_bitop3_b32(hi, (u32)(bal >> 32), (u32)bm, 0x90);
  }
}

__device__ __forceinline__ void match_digit8(u32 digit, u32 &lo, u32 &hi) { match_digit<8>(digit, lo, hi); }

// floor(x / d) for d > 0.  64-bit integer division is a long instruction sequence on the GPU; below 2^53 a
// double division is off by at most one, which two multiplications put right.
__device__ __forceinline__ u64 div_floor_u64(u64 x, u64 d) {
  if (x >= (1ull << 53) || d >= (1ull << 53)) return x / d;
  u64 q = (u64)((double)x / (double)d);
  if (q * d > x) q--;
  else if ((q + 1) * d <= x) q++;
  return q;
}

// number of set bits of (hi:lo) below this lane
__device__ __forceinline__ u32 count_below_lane(u32 lo, u32 hi) {
  return __builtin_amdgcn_mbcnt_hi(hi, __builtin_amdgcn_mbcnt_lo(lo, 0u));
}

// Workgroup barrier that orders LDS traffic only (__syncthreads() also waits for every outstanding
// global access of the wave).
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// The listed stretch (start, end, period p: T[j] = T[j-p] for start + p <= j < end) that holds x with at least its
// period behind x inside it: *end_out and the period, 0 if there is none.  runs: n_runs starts, ends, periods.
constexpr u32 RUN_PERIOD_MAX = 8;
__device__ __forceinline__ u32 run_around(const u32 *__restrict__ runs, u32 n_runs, u64 x, u64 *end_out) {
  if (x < RUN_PERIOD_MAX) return 0;
  u32 lo = 0, hi = n_runs;  // the last stretch that starts at or below x - RUN_PERIOD_MAX
  while (lo < hi) {
    const u32 mid = (lo + hi) / 2;
    if ((u64)runs[mid] <= x - RUN_PERIOD_MAX) lo = mid + 1;
    else hi = mid;
  }
  if (lo == 0) return 0;
  const u64 end = runs[n_runs + lo - 1];
  if (x >= end || x < (u64)runs[lo - 1] + RUN_PERIOD_MAX) return 0;
  *end_out = end;
  return runs[2 * n_runs + lo - 1];
}

// Workgroups are dealt round-robin over the 8 XCDs (each with its own L2).  This
// bijection gives every XCD one contiguous range of tiles, so that the runs that
// neighbouring tiles append to the same digit's output region meet in one L2
// and leave it as whole lines.  Speed only: any mapping is correct (measured on
// the 5e8-suffix leaf sort: scatter pass 2.39 ms with the mapping, 2.90 ms without).
__device__ __forceinline__ u32 xcd_tile(u32 b, u32 nblk) {
  u32 q = nblk >> 3, r = nblk & 7, x = b & 7;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
}

}  // namespace bwtamd
