// Ranking of one bucket of the leaf sort inside a workgroup (bucket route: the elements that share their
// top 16 key bits are contiguous after two radix passes and fit LDS).
#pragma once

#include "device_utils.h"

namespace bwtamd {

constexpr int BK_THREADS = 1024;
constexpr int BK_WAVES = BK_THREADS / 64;
constexpr int BK_ITEMS = 9;
constexpr int BK_CAP = BK_THREADS * BK_ITEMS;  // 9216 elements = 72 KB of LDS
constexpr int BK_PER_WAVE = 64 * BK_ITEMS;
constexpr u32 BK_BUCKETS = 1u << 16;


// Stable rank of the workgroup's `cnt` keys by the digit at `shift`.  Wave w holds the keys at indices
// w * BK_PER_WAVE + r * 64 + lane in key[r]; pos[r] receives each key's index in digit order.
// ctr: (BK_WAVES + 1) rows of 256 counters, dead when the function returns.
__device__ __forceinline__ void bk_rank(const u64 (&key)[BK_ITEMS], int shift, u32 cnt, u32 (*ctr)[256], u32 *scan_s,
                                        u32 (&pos)[BK_ITEMS]) {
  const u32 tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const u32 wbase = w * BK_PER_WAVE;
  u32 *myc = ctr[w];
  u32 *dstart = ctr[BK_WAVES];
  for (u32 d = lane; d < 256; d += 64) myc[d] = 0;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  u32 dig[BK_ITEMS];
#pragma unroll
  for (int r = 0; r < BK_ITEMS; r++) {
    dig[r] = (u32)(key[r] >> shift) & 255u;
    pos[r] = 0;
    if (wbase + r * 64 < cnt) {  // the same for the whole wave
      const bool valid = wbase + r * 64 + lane < cnt;
      const u64 v = __ballot(valid);
      u32 lo = (u32)v, hi = (u32)(v >> 32);
      match_digit8(dig[r], lo, hi);
      const u32 before = count_below_lane(lo, hi);
      const u32 base = __hip_atomic_load(&myc[dig[r]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
      if (valid && before == 0)
        __hip_atomic_fetch_add(&myc[dig[r]], (u32)(__popc(lo) + __popc(hi)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      pos[r] = base + before;
    }
  }
  lds_barrier();
  u32 tot = 0;
  if (tid < 256) {
#pragma unroll
    for (int k = 0; k < BK_WAVES; k++) {
      const u32 v = ctr[k][tid];
      ctr[k][tid] = tot;
      tot += v;
    }
  }
  u32 incl = wave_scan_incl_sum(tot);
  if (lane == 63 && w < 4) scan_s[w] = incl;
  lds_barrier();
  for (u32 k = 0; k < w && k < 4; k++) incl += scan_s[k];
  if (tid < 256) dstart[tid] = incl - tot;
  lds_barrier();
#pragma unroll
  for (int r = 0; r < BK_ITEMS; r++) pos[r] += dstart[dig[r]] + myc[dig[r]];
  lds_barrier();  // the counters are dead from here
}

}  // namespace bwtamd
