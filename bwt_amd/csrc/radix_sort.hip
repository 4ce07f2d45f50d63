// Stable LSD radix sort, 8 bits per pass, for (u64|u32 key, optional u32 value).
//
// One pass = three launches:
//   k_rs_hist     per-tile digit counts          reads keys
//   scan          exclusive scan of [digit][tile] counts
//   k_rs_scatter  ranks the tile's keys per digit, orders the tile in LDS and
//                 writes each digit's run to its global offset
//                                                reads keys+vals, writes keys+vals
// Algorithmic HBM bytes per element and pass: 2*sizeof(K) + sizeof(K) + 2*4 (with values).
//
// Ranking inside a tile: a wave owns 512 consecutive keys and walks them 64 at a
// time; eight ballots find the lanes that share a digit, the lowest of them
// bumps the wave's private LDS counter, so equal digits keep their input order
// (the sort is stable, which LSD needs).  The histogram kernel only counts, so
// it uses LDS atomic adds on per-wave counter rows.
#include "device_utils.h"
#include "internal.h"

namespace bwtamd {

constexpr int RS_THREADS = 512;
constexpr int RS_WAVES = RS_THREADS / 64;
constexpr int RS_ITEMS = 8;
constexpr int RS_TILE = RS_THREADS * RS_ITEMS;  // 4096 keys per workgroup
constexpr int RS_PER_WAVE = RS_TILE / RS_WAVES; // 512 consecutive keys per wave
constexpr int RH_THREADS = 256;                 // histogram kernel: 256 threads x 16 keys
constexpr int RH_ITEMS = RS_TILE / RH_THREADS;

// Workgroups are dealt round-robin over the 8 XCDs (each with its own L2).  This
// bijection gives every XCD one contiguous range of tiles, so that the runs that
// neighbouring tiles append to the same digit's output region meet in one L2
// and leave it as whole lines.  Speed only: any mapping is correct.
__device__ __forceinline__ u32 xcd_tile(u32 b, u32 nblk) {
  u32 q = nblk >> 3, r = nblk & 7, x = b & 7;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
}

// USE only names the caller in the kernel symbol (0 leaf sort, 1 refinement
// rounds, 2 gap histogram), so that a profiler lists their launches apart.
template <class K, int USE>
__global__ void __launch_bounds__(RH_THREADS)
    k_rs_hist(const K *__restrict__ keys, size_t n, int shift, u32 mask, u32 *__restrict__ counts, u32 nblk) {
  __shared__ u32 cnt[RH_THREADS / 64][256];  // one row of LDS counters per wave
  const u32 tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  for (u32 d = lane; d < 256; d += 64) cnt[w][d] = 0;
  const u32 tile = xcd_tile(blockIdx.x, nblk);
  const size_t tile0 = (size_t)tile * RS_TILE;
  K key[RH_ITEMS];
  const bool full = tile0 + RS_TILE <= n;
  if (full) {  // all loads issued before the first use: 16 bytes per lane and load
    constexpr int PER = 16 / sizeof(K);
    const uint4 *src = reinterpret_cast<const uint4 *>(keys + tile0);
#pragma unroll
    for (int r = 0; r < RH_ITEMS / PER; r++) {
      uint4 v = src[r * RH_THREADS + tid];
      if (sizeof(K) == 8) {
        key[2 * r] = (K)(((u64)v.y << 32) | v.x);
        key[2 * r + 1] = (K)(((u64)v.w << 32) | v.z);
      } else {
        key[4 * r] = (K)v.x;
        key[4 * r + 1] = (K)v.y;
        key[4 * r + 2] = (K)v.z;
        key[4 * r + 3] = (K)v.w;
      }
    }
  } else {
#pragma unroll
    for (int r = 0; r < RH_ITEMS; r++) {
      size_t i = tile0 + (size_t)r * RH_THREADS + tid;
      key[r] = i < n ? keys[i] : (K)0;
    }
  }
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int r = 0; r < RH_ITEMS; r++) {
    bool valid = full || (tile0 + (size_t)r * RH_THREADS + tid < n);
    u32 d = (u32)(key[r] >> shift) & mask;
    if (valid) atomicAdd(&cnt[w][d], 1u);  // LDS counter of this wave; order is irrelevant for a count
  }
  __syncthreads();
  u32 tot = 0;
#pragma unroll
  for (int k = 0; k < RH_THREADS / 64; k++) tot += cnt[k][tid];
  counts[(size_t)tid * nblk + tile] = tot;
}

template <class K, bool HAS_VALS, int USE>
__global__ void __launch_bounds__(RS_THREADS)
    k_rs_scatter(const K *__restrict__ kin, const u32 *__restrict__ vin, K *__restrict__ kout,
                 u32 *__restrict__ vout, size_t n, int shift, u32 mask, const u32 *__restrict__ offs,
                 u32 nblk) {
  // One LDS buffer, used twice: first the per-wave digit counters and the digit
  // starts, then (after every thread has turned its ranks into tile positions)
  // the tile's keys and values in digit order.
  constexpr size_t STAGE_BYTES = RS_TILE * sizeof(K) + (HAS_VALS ? RS_TILE * 4 : 0);
  constexpr size_t COUNT_BYTES = (RS_WAVES + 1) * 256 * 4;
  constexpr size_t BUF_BYTES = STAGE_BYTES > COUNT_BYTES ? STAGE_BYTES : COUNT_BYTES;
  __shared__ __align__(16) unsigned char buf[BUF_BYTES];
  __shared__ u32 gbase[256];
  __shared__ u32 scan_s[17];
  u32(*cnt)[256] = reinterpret_cast<u32(*)[256]>(buf);
  u32 *dstart = reinterpret_cast<u32 *>(buf) + RS_WAVES * 256;
  K *keys_s = reinterpret_cast<K *>(buf);
  u32 *vals_s = reinterpret_cast<u32 *>(buf + RS_TILE * sizeof(K));

  const u32 tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const u32 tile = xcd_tile(blockIdx.x, nblk);
  const size_t tile0 = (size_t)tile * RS_TILE;
  const u32 tile_n = (u32)((n - tile0) < (size_t)RS_TILE ? (n - tile0) : (size_t)RS_TILE);
  for (u32 d = lane; d < 256; d += 64) cnt[w][d] = 0;

  K key[RS_ITEMS];
  u32 val[RS_ITEMS];
  u32 rk[RS_ITEMS];
  const u32 wbase = w * RS_PER_WAVE;
#pragma unroll
  for (int r = 0; r < RS_ITEMS; r++) {
    u32 idx = wbase + r * 64 + lane;
    bool valid = idx < tile_n;
    key[r] = valid ? kin[tile0 + idx] : (K)0;
    if (HAS_VALS) val[r] = valid ? vin[tile0 + idx] : 0u;
  }
  volatile u32 *myc = cnt[w];
#pragma unroll
  for (int r = 0; r < RS_ITEMS; r++) {
    u32 idx = wbase + r * 64 + lane;
    bool valid = idx < tile_n;
    u32 d = (u32)(key[r] >> shift) & mask;
    u64 peers = match_digit8(d, valid);
    u32 before = __popcll(peers & lanemask_lt());
    u32 base = valid ? myc[d] : 0u;
    rk[r] = base + before;
    if (valid && before == 0) myc[d] = base + __popcll(peers);
  }
  __syncthreads();
  {
    // thread d < 256 owns digit d: wave counts -> wave offsets, digit totals -> tile-local digit starts
    u32 tot = 0;
    if (tid < 256) {
#pragma unroll
      for (int k = 0; k < RS_WAVES; k++) {
        u32 v = cnt[k][tid];
        cnt[k][tid] = tot;
        tot += v;
      }
    }
    u32 all;
    u32 incl = block_scan_incl<OpSum>(tot, scan_s, &all);
    if (tid < 256) {
      u32 ds = incl - tot;
      dstart[tid] = ds;
      gbase[tid] = offs[(size_t)tid * nblk + tile] - ds;
    }
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < RS_ITEMS; r++) {
    u32 d = (u32)(key[r] >> shift) & mask;
    rk[r] += dstart[d] + cnt[w][d];  // position inside the tile
  }
  __syncthreads();  // counters are dead from here: the buffer becomes the staging area
#pragma unroll
  for (int r = 0; r < RS_ITEMS; r++) {
    u32 idx = wbase + r * 64 + lane;
    if (idx < tile_n) {
      keys_s[rk[r]] = key[r];
      if (HAS_VALS) vals_s[rk[r]] = val[r];
    }
  }
  __syncthreads();
  for (u32 p = tid; p < tile_n; p += RS_THREADS) {
    K k = keys_s[p];
    u32 d = (u32)(k >> shift) & mask;
    u32 dst = gbase[d] + p;
    kout[dst] = k;
    if (HAS_VALS) vout[dst] = vals_s[p];
  }
}

template <class K, int USE>
static int radix_sort_impl(Ctx &c, K *keys[2], u32 *vals[2], size_t n, int begin_bit, int end_bit) {
  if (n == 0 || end_bit <= begin_bit) return 0;
  if (n >= (1ull << 32)) fail(BWT_HIP_EINVAL, "radix sort: %zu elements exceed 2^32-1", n);
  u32 nblk = ceil_div(n, RS_TILE);
  DBuf<u32> counts(c, (size_t)256 * nblk);
  const bool has_vals = vals[0] != nullptr;
  int cur = 0;
  for (int shift = begin_bit; shift < end_bit; shift += 8) {
    int bits = end_bit - shift < 8 ? end_bit - shift : 8;
    u32 mask = (1u << bits) - 1;
    {
      KernelTimer kt(c, (std::string(c.sort_tag) + (sizeof(K) == 8 ? "k_rs_hist_u64" : "k_rs_hist_u32")).c_str(),
                     (double)n * sizeof(K));
      hipLaunchKernelGGL((k_rs_hist<K, USE>), dim3(nblk), dim3(RH_THREADS), 0, c.stream, keys[cur], n, shift, mask,
                         counts.get(), nblk);
      HIP_CHECK(hipGetLastError());
    }
    scan_exclusive_u32(c, counts.get(), counts.get(), (size_t)256 * nblk, nullptr);
    {
      const char *nm = sizeof(K) == 8 ? (has_vals ? "k_rs_scatter_u64_pairs" : "k_rs_scatter_u64_keys")
                                      : (has_vals ? "k_rs_scatter_u32_pairs" : "k_rs_scatter_u32_keys");
      KernelTimer kt(c, (std::string(c.sort_tag) + nm).c_str(), (double)n * 2 * (sizeof(K) + (has_vals ? 4 : 0)));
      if (has_vals)
        hipLaunchKernelGGL((k_rs_scatter<K, true, USE>), dim3(nblk), dim3(RS_THREADS), 0, c.stream, keys[cur],
                           vals[cur], keys[cur ^ 1], vals[cur ^ 1], n, shift, mask, counts.get(), nblk);
      else
        hipLaunchKernelGGL((k_rs_scatter<K, false, USE>), dim3(nblk), dim3(RS_THREADS), 0, c.stream, keys[cur],
                           (const u32 *)nullptr, keys[cur ^ 1], (u32 *)nullptr, n, shift, mask, counts.get(),
                           nblk);
      HIP_CHECK(hipGetLastError());
    }
    cur ^= 1;
  }
  return cur;
}

int radix_sort_u64(Ctx &c, u64 *keys[2], u32 *vals[2], size_t n, int begin_bit, int end_bit, int use) {
  if (use == 0) return radix_sort_impl<u64, 0>(c, keys, vals, n, begin_bit, end_bit);
  return radix_sort_impl<u64, 1>(c, keys, vals, n, begin_bit, end_bit);
}

int radix_sort_u32(Ctx &c, u32 *keys[2], u32 *vals[2], size_t n, int begin_bit, int end_bit, int use) {
  if (use == 2) return radix_sort_impl<u32, 2>(c, keys, vals, n, begin_bit, end_bit);
  return radix_sort_impl<u32, 1>(c, keys, vals, n, begin_bit, end_bit);
}

}  // namespace bwtamd
