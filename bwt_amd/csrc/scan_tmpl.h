// Generic device-wide scan in three launches (tile reduce, scan of the tile
// aggregates, tile scan + emit).  Inputs come from a functor and results go to
// a functor, so flag computation, compaction and scatter fuse into the scan's
// two passes over the data instead of materialising flag and offset arrays.
//
//   T   value type; a multiple of 4 bytes, trivially copyable
//   Op  struct { static T identity(); static T apply(T earlier, T later); }
//       associative
//   In  struct { __device__ T operator()(size_t i) const; }
//       optionally also  void vec4(size_t i0, T out[4]) const  for four consecutive
//       inputs starting at a multiple of four (all below n), e.g. through 16-byte loads
//   Out may likewise offer  void vec4(size_t i0, const T excl[4], const T incl[4]) const
//   Out struct { __device__ void operator()(size_t i, T excl, T incl) const; }
//
// HBM traffic: whatever In reads, twice, plus whatever Out writes.
#pragma once

#include <type_traits>
#include <utility>

#include "device_utils.h"
#include "internal.h"

namespace bwtamd {

template <class Out, class T, class = void>
struct has_out_vec4 : std::false_type {};
template <class Out, class T>
struct has_out_vec4<Out, T,
                    std::void_t<decltype(std::declval<const Out &>().vec4((size_t)0, (const T *)nullptr, (const T *)nullptr))>>
    : std::true_type {};

template <class In, class T, class = void>
struct has_vec4 : std::false_type {};
template <class In, class T>
struct has_vec4<In, T, std::void_t<decltype(std::declval<const In &>().vec4((size_t)0, (T *)nullptr))>>
    : std::true_type {};

constexpr int SCAN_THREADS = 256;
constexpr int SCAN_VEC = 4;     // consecutive items per thread per round
constexpr int SCAN_ROUNDS = 4;  // rounds per tile
constexpr int SCAN_TILE = SCAN_THREADS * SCAN_VEC * SCAN_ROUNDS;

template <class T>
__device__ __forceinline__ T shfl_up_any(T v, int delta) {
  static_assert(sizeof(T) % 4 == 0, "T must be a multiple of 4 bytes");
  union {
    T t;
    u32 w[sizeof(T) / 4];
  } a, b;
  a.t = v;
#pragma unroll
  for (unsigned k = 0; k < sizeof(T) / 4; k++) b.w[k] = __shfl_up(a.w[k], delta, 64);
  return b.t;
}

template <class T, class Op>
__device__ __forceinline__ T wave_incl_any(T v) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    T t = shfl_up_any(v, o);
    if (lane_id() >= (u32)o) v = Op::apply(t, v);
  }
  return v;
}

// Block scan of one T per thread (SCAN_THREADS threads).  smem: one T per wave.
// Two __syncthreads(); every thread of the block must call it.
template <class T, class Op>
__device__ __forceinline__ void block_scan_any(T v, T *smem, T &excl, T &incl, T &total) {
  T wincl = wave_incl_any<T, Op>(v);
  T wprev = shfl_up_any(wincl, 1);
  u32 w = wave_id();
  __syncthreads();  // smem may still be read by a previous call
  if (lane_id() == 63) smem[w] = wincl;
  __syncthreads();
  T carry = Op::identity(), tot = Op::identity();
#pragma unroll
  for (u32 k = 0; k < SCAN_THREADS / 64; k++) {
    T s = smem[k];
    if (k < w) carry = Op::apply(carry, s);
    tot = Op::apply(tot, s);
  }
  excl = (lane_id() == 0) ? carry : Op::apply(carry, wprev);
  incl = Op::apply(carry, wincl);
  total = tot;
}

// the SCAN_VEC inputs of one thread and round; identity beyond n
template <class T, class Op, class In>
__device__ __forceinline__ void scan_load(const In &in, size_t i0, size_t n, T (&v)[SCAN_VEC]) {
  static_assert(SCAN_VEC == 4, "vec4 inputs");
  if constexpr (has_vec4<In, T>::value) {
    if (i0 + SCAN_VEC <= n) {
      in.vec4(i0, v);
      return;
    }
  }
#pragma unroll
  for (int k = 0; k < SCAN_VEC; k++) v[k] = (i0 + k < n) ? in(i0 + k) : Op::identity();
}

template <class T, class Op, class In>
__global__ void __launch_bounds__(SCAN_THREADS) k_scan_reduce(size_t n, In in, T *tile_agg) {
  __shared__ T smem[SCAN_THREADS / 64];
  size_t base = (size_t)blockIdx.x * SCAN_TILE;
  T tile_tot = Op::identity();
  // the same order of combination as k_scan_apply, so a non-commutative Op works too
#pragma unroll
  for (int r = 0; r < SCAN_ROUNDS; r++) {
    size_t i0 = base + ((size_t)r * SCAN_THREADS + threadIdx.x) * SCAN_VEC;
    T v[SCAN_VEC];
    scan_load<T, Op, In>(in, i0, n, v);
    T mine = Op::identity();
#pragma unroll
    for (int k = 0; k < SCAN_VEC; k++) mine = Op::apply(mine, v[k]);
    T e, i, tot;
    block_scan_any<T, Op>(mine, smem, e, i, tot);
    tile_tot = Op::apply(tile_tot, tot);
  }
  if (threadIdx.x == 0) tile_agg[blockIdx.x] = tile_tot;
}

template <class T, class Op, class In, class Out>
__global__ void __launch_bounds__(SCAN_THREADS)
    k_scan_apply(size_t n, In in, Out out, const T *tile_excl) {
  __shared__ T smem[SCAN_THREADS / 64];
  size_t base = (size_t)blockIdx.x * SCAN_TILE;
  T carry = tile_excl ? tile_excl[blockIdx.x] : Op::identity();
#pragma unroll
  for (int r = 0; r < SCAN_ROUNDS; r++) {
    size_t i0 = base + ((size_t)r * SCAN_THREADS + threadIdx.x) * SCAN_VEC;
    T v[SCAN_VEC];
    scan_load<T, Op, In>(in, i0, n, v);
    T mine = Op::identity();
#pragma unroll
    for (int k = 0; k < SCAN_VEC; k++) mine = Op::apply(mine, v[k]);
    T excl, incl, tot;
    block_scan_any<T, Op>(mine, smem, excl, incl, tot);
    T run = Op::apply(carry, excl);
    T ex[SCAN_VEC], inc[SCAN_VEC];
#pragma unroll
    for (int k = 0; k < SCAN_VEC; k++) {
      ex[k] = run;
      run = Op::apply(run, v[k]);
      inc[k] = run;
    }
    bool done = false;
    if constexpr (has_out_vec4<Out, T>::value) {
      if (i0 + SCAN_VEC <= n) {
        out.vec4(i0, ex, inc);
        done = true;
      }
    }
    if (!done) {
#pragma unroll
      for (int k = 0; k < SCAN_VEC; k++)
        if (i0 + k < n) out(i0 + k, ex[k], inc[k]);
    }
    carry = Op::apply(carry, tot);
  }
}

template <class T>
struct ArrayIn {
  const T *p;
  __device__ __forceinline__ T operator()(size_t i) const { return p[i]; }
};
template <class T>
struct ArrayOutExcl {
  T *p;
  T *total;  // may be null; receives the combination of all n inputs
  size_t n;
  __device__ __forceinline__ void operator()(size_t i, T excl, T incl) const {
    p[i] = excl;
    if (total && i + 1 == n) *total = incl;
  }
};
template <class T>
struct ArrayOutIncl {
  T *p;
  __device__ __forceinline__ void operator()(size_t i, T, T incl) const { p[i] = incl; }
};

// Device-wide scan of in(0..n) with results handed to out(i, exclusive, inclusive).
template <class T, class Op, class In, class Out>
void scan_generic(Ctx &c, size_t n, In in, Out out) {
  if (n == 0) return;
  size_t tiles = (n + SCAN_TILE - 1) / SCAN_TILE;
  if (tiles == 1) {
    hipLaunchKernelGGL((k_scan_apply<T, Op, In, Out>), dim3(1), dim3(SCAN_THREADS), 0, c.stream, n, in,
                       out, (const T *)nullptr);
    HIP_CHECK(hipGetLastError());
    return;
  }
  DBuf<T> agg(c, tiles);
  hipLaunchKernelGGL((k_scan_reduce<T, Op, In>), dim3((u32)tiles), dim3(SCAN_THREADS), 0, c.stream, n,
                     in, agg.get());
  HIP_CHECK(hipGetLastError());
  // exclusive scan of the tile aggregates in place (the recursion ends at one tile)
  scan_generic<T, Op>(c, tiles, ArrayIn<T>{agg.get()}, ArrayOutExcl<T>{agg.get(), nullptr, tiles});
  hipLaunchKernelGGL((k_scan_apply<T, Op, In, Out>), dim3((u32)tiles), dim3(SCAN_THREADS), 0, c.stream, n,
                     in, out, (const T *)agg.get());
  HIP_CHECK(hipGetLastError());
}

struct SumU32 {
  __device__ __forceinline__ static u32 identity() { return 0; }
  __device__ __forceinline__ static u32 apply(u32 a, u32 b) { return a + b; }
};
struct MaxU32 {
  __device__ __forceinline__ static u32 identity() { return 0; }
  __device__ __forceinline__ static u32 apply(u32 a, u32 b) { return a > b ? a : b; }
};

}  // namespace bwtamd
