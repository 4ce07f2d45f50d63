// Device context: caching allocator, phase timers, text utilities.
#include <stdarg.h>

#include <mutex>

#include "internal.h"
#include "device_utils.h"

namespace bwtamd {

void fail(int code, const char *fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  throw Error(code, buf);
}

// ---------------------------------------------------------------------------
// caching allocator.  hipMalloc/hipFree synchronise the device and cost
// milliseconds at GB sizes, and one BWT build allocates dozens of GB-sized
// temporaries; freed blocks are therefore kept and reused (best fit, up to
// 25 % slack) until trim() or context destruction.
// ---------------------------------------------------------------------------
static size_t round_size(size_t b) {
  if (b < 256) return 256;
  if (b < (1u << 20)) return (b + 255) & ~(size_t)255;
  return (b + ((1u << 21) - 1)) & ~(size_t)((1u << 21) - 1);
}

// ---------------------------------------------------------------------------
// Scratch memory of the entry points that must not allocate (bwt_hip_dist_merge_slice_noalloc): one static
// array per GPU, held from the moment the library's code is loaded on it until the process ends.  In host
// code the array's name is only a handle for hipGetSymbolAddress: every use of the memory -- kernel
// arguments, memsets, copies -- goes through the address looked up here, once per device with that
// device current (each GPU has its own copy at its own address).
// ---------------------------------------------------------------------------
// Sized for the largest call bwt_hip_dist_merge_slice_noalloc accepts, whatever the left block's length: 4 Mi right
// symbols (16 MiB for the sort's second buffer) next to a left block of 2^30 to 2^31 suffixes, whose grouping has the most
// buckets (2^18 + 1 bounds, 1 MiB) and the most tiles (512 counters for each of 512 + 513 tiles, 3 MiB), plus the tile
// map and the scan's partial sums (a few KiB), and the merge's chunk bounds (a word per 4096 output symbols: 2 MiB for a
// slice of 2^31 left symbols): 22.2 MiB.
constexpr size_t SCRATCH_BYTES = (size_t)24 << 20;
static __device__ __align__(256) unsigned char scratch[SCRATCH_BYTES];

constexpr int MAX_DEVICES = 64;
static unsigned char *g_scratch_ptr[MAX_DEVICES];
static std::mutex g_scratch_mu;

unsigned char *device_scratch(int device, size_t *bytes) {
  if (device < 0 || device >= MAX_DEVICES) fail(BWT_HIP_EINVAL, "device %d: at most %d devices", device, MAX_DEVICES);
  std::lock_guard<std::mutex> lock(g_scratch_mu);
  if (!g_scratch_ptr[device]) {
    int cur = -1;
    HIP_CHECK(hipGetDevice(&cur));
    if (cur != device) fail(BWT_HIP_EINTERNAL, "scratch look-up for device %d while device %d is current", device, cur);
    void *p = nullptr;
    HIP_CHECK(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
    if (!p || ((uintptr_t)p & 255)) fail(BWT_HIP_EINTERNAL, "scratch array of device %d at %p", device, p);
    g_scratch_ptr[device] = (unsigned char *)p;
  }
  if (bytes) *bytes = SCRATCH_BYTES;
  return g_scratch_ptr[device];
}

void *Ctx::alloc(size_t bytes) {
  if (arena_on) {
    const size_t need = (bytes + 255) & ~(size_t)255;
    if (need > arena_cap - arena_used)
      fail(BWT_HIP_EINVAL, "no-alloc call needs %zu more bytes of scratch than the %zu the library holds", need - (arena_cap - arena_used),
           arena_cap);
    void *p = arena + arena_used;
    arena_used += need;
    return p;
  }
  size_t sz = round_size(bytes);
  auto it = free_blocks.lower_bound(sz);
  void *p = nullptr;
  if (it != free_blocks.end() && it->first <= sz + sz / 4 + (1u << 20)) {
    p = it->second;
    sz = it->first;
    free_blocks.erase(it);
  } else {
    hipError_t e = hipMalloc(&p, sz);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      trim();
      e = hipMalloc(&p, sz);
    }
    if (e != hipSuccess)
      fail((int)e, "hipMalloc of %zu bytes failed: %s", sz, hipGetErrorString(e));
    bytes_reserved += sz;
  }
  live_blocks[p] = sz;
  bytes_live += sz;
  if (bytes_live > bytes_peak) bytes_peak = bytes_live;
  return p;
}

void Ctx::release(void *p) {
  if (!p) return;
  if (arena && (unsigned char *)p >= arena && (unsigned char *)p < arena + arena_cap) return;  // part of the static array
  auto it = live_blocks.find(p);
  if (it == live_blocks.end()) return;
  size_t sz = it->second;
  live_blocks.erase(it);
  bytes_live -= sz;
  // Work queued on the stream may still read or write the block; any later
  // user of the block is queued on the same stream, behind that work.
  free_blocks.insert({sz, p});
}

void Ctx::trim() {
  if (free_blocks.empty()) return;
  (void)hipStreamSynchronize(stream);
  for (auto &kv : free_blocks) {
    (void)hipFree(kv.second);
    bytes_reserved -= kv.first;
  }
  free_blocks.clear();
}

void Ctx::sync() { HIP_CHECK(hipStreamSynchronize(stream)); }

void Ctx::wait_text(u64 upto) {
  for (size_t i = 0; i < chunks.size(); i++) {
    TextChunk &ch = chunks[i];
    if (!ch.waited) {
      {
        std::unique_lock<std::mutex> lk(chunk_mu);
        chunk_cv.wait(lk, [&] { return chunks_recorded > i; });
      }
      if (xfer_mismatch.load()) throw TextMismatch();
      HIP_CHECK(hipStreamWaitEvent(stream, ch.ev, 0));
      ch.waited = true;
    }
    if (ch.end >= upto) break;
  }
}

// ---------------------------------------------------------------------------
// phase timers
// ---------------------------------------------------------------------------
int Ctx::timer_index(const char *name) {
  for (size_t i = 0; i < timers.size(); i++)
    if (timers[i].name == name) return (int)i;
  PhaseTimer t;
  t.name = name;
  timers.push_back(t);
  return (int)timers.size() - 1;
}

static hipEvent_t get_event(Ctx &c) {
  if (!c.event_pool.empty()) {
    hipEvent_t e = c.event_pool.back();
    c.event_pool.pop_back();
    return e;
  }
  hipEvent_t e;
  HIP_CHECK(hipEventCreate(&e));
  return e;
}

Phase::Phase(Ctx &ctx, const char *name, double bytes) : c(ctx) {
  if (c.timing_level <= 0) return;
  idx = c.timer_index(name);
  c.timers[idx].bytes += bytes;
  c.timers[idx].launches++;
  e0 = get_event(c);
  (void)hipEventRecord(e0, c.stream);
}

Phase::~Phase() {
  if (idx < 0) return;
  hipEvent_t e1 = nullptr;
  try {
    e1 = get_event(c);
  } catch (...) {
    return;
  }
  (void)hipEventRecord(e1, c.stream);
  c.pending.push_back({idx, e0, e1});
}

void Ctx::flush_timers() {
  if (pending.empty()) return;
  sync();
  for (auto &p : pending) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, p.e0, p.e1) == hipSuccess) timers[p.idx].ms += ms;
    event_pool.push_back(p.e0);
    event_pool.push_back(p.e1);
  }
  pending.clear();
}

// ---------------------------------------------------------------------------
// text utilities
// ---------------------------------------------------------------------------

// T[i] = 1 + splitmix64(seed, i) % sigma for i < n-1, T[n-1] = 0: the same
// stream as the CPU generators of the tests, one element per counter value.
__global__ void k_gen_text(u8 *T, u64 n, u64 seed, u32 sigma) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  u64 stride = (u64)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    u64 z = seed + (i + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    T[i] = (i + 1 == n) ? (u8)0 : (u8)(1 + z % sigma);
  }
}

void device_gen_text(Ctx &c, u8 *dT, i64 n, u64 seed, int sigma) {
  u32 blocks = (u32)std::min<u64>(ceil_div(n, 256), (u64)c.n_cu * 32);
  hipLaunchKernelGGL(k_gen_text, dim3(blocks), dim3(256), 0, c.stream, dT, (u64)n, seed, (u32)sigma);
  HIP_CHECK(hipGetLastError());
}

// out[0] = max symbol, out[1] = number of zeros, out[2] = 255 - (smallest non-zero symbol)
__global__ void k_text_stats(const u8 *T, u64 n, u32 *out) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  u64 stride = (u64)gridDim.x * blockDim.x;
  u32 mx = 0, nz = 0, inv_mn = 0;
  for (; i < n; i += stride) {
    u32 v = T[i];
    mx = max(mx, v);
    nz += (v == 0);
    if (v) inv_mn = max(inv_mn, 255u - v);
  }
  mx = wave_reduce_max(mx);
  nz = wave_reduce_sum(nz);
  inv_mn = wave_reduce_max(inv_mn);
  if ((threadIdx.x & 63) == 0) {
    atomicMax(&out[0], mx);
    atomicAdd(&out[1], nz);
    atomicMax(&out[2], inv_mn);
  }
}

void text_stats(Ctx &c, const u8 *dT, i64 n, int *max_sym, int *min_sym, i64 *n_zero, int *last) {
  HIP_CHECK(hipMemsetAsync(c.d_scalars, 0, 12, c.stream));
  u32 blocks = (u32)std::min<u64>(ceil_div(n, 256 * 16), (u64)c.n_cu * 16);
  hipLaunchKernelGGL(k_text_stats, dim3(blocks), dim3(256), 0, c.stream, dT, (u64)n, c.d_scalars);
  HIP_CHECK(hipGetLastError());
  HIP_CHECK(hipMemcpyAsync(c.h_scalars, c.d_scalars, 12, hipMemcpyDeviceToHost, c.stream));
  u8 lastc = 0;
  HIP_CHECK(hipMemcpyAsync(&c.h_scalars[3], dT + n - 1, 1, hipMemcpyDeviceToHost, c.stream));
  c.sync();
  lastc = *(u8 *)&c.h_scalars[3];
  *min_sym = c.h_scalars[0] ? 255 - (int)c.h_scalars[2] : 0;
  *max_sym = (int)c.h_scalars[0];
  *n_zero = (i64)c.h_scalars[1];
  *last = lastc;
}

__global__ void __launch_bounds__(256) k_symbol_presence(const u8 *__restrict__ T, u64 n, u32 *__restrict__ present) {
  u64 seen[4] = {0, 0, 0, 0};
  const u64 words = n / 16;
  for (u64 q = (u64)blockIdx.x * 256 + threadIdx.x; q <= words; q += (u64)gridDim.x * 256) {
    u32 w[4] = {0, 0, 0, 0};
    u32 count = 16;
    if (q < words) {
      const uint4 v = reinterpret_cast<const uint4 *>(T)[q];  // (the text is 16-byte aligned)
      w[0] = v.x, w[1] = v.y, w[2] = v.z, w[3] = v.w;
    } else {
      count = (u32)(n - words * 16);
      for (u32 k = 0; k < count; k++) w[k >> 2] |= (u32)T[words * 16 + k] << (8 * (k & 3));
    }
    for (u32 k = 0; k < count; k++) {
      const u32 v = (w[k >> 2] >> (8 * (k & 3))) & 255u;
      const u64 bit = 1ull << (v & 63u);
      seen[0] |= (v >> 6) == 0 ? bit : 0ull;
      seen[1] |= (v >> 6) == 1 ? bit : 0ull;
      seen[2] |= (v >> 6) == 2 ? bit : 0ull;
      seen[3] |= (v >> 6) == 3 ? bit : 0ull;
    }
  }
  for (int k = 0; k < 4; k++) {
    u32 lo = (u32)seen[k], hi = (u32)(seen[k] >> 32);
    for (int o = 32; o > 0; o >>= 1) {
      lo |= (u32)__shfl_xor(lo, o, 64);
      hi |= (u32)__shfl_xor(hi, o, 64);
    }
    if ((threadIdx.x & 63) == 0) {
      if (lo) atomicOr(&present[2 * k], lo);
      if (hi) atomicOr(&present[2 * k + 1], hi);
    }
  }
}

void symbol_presence(Ctx &c, const u8 *dT, i64 n, u32 present[8]) {
  u32 *d = c.d_scalars + 40;
  HIP_CHECK(hipMemsetAsync(d, 0, 32, c.stream));
  const u32 blocks = (u32)std::min<u64>(ceil_div((u64)n / 16 + 1, 256 * 8), (u64)c.n_cu * 16);
  hipLaunchKernelGGL(k_symbol_presence, dim3(blocks ? blocks : 1), dim3(256), 0, c.stream, dT, (u64)n, d);
  HIP_CHECK(hipGetLastError());
  HIP_CHECK(hipMemcpyAsync(c.h_scalars + 40, d, 32, hipMemcpyDeviceToHost, c.stream));
  c.sync();
  for (int k = 0; k < 8; k++) present[k] = c.h_scalars[40 + k];
}

// (src and dst 16-byte aligned)
__global__ void __launch_bounds__(256) k_map_symbols(const u8 *src, u8 *dst, u64 n, const u8 *__restrict__ table) {
  __shared__ u8 tab[256];
  tab[threadIdx.x] = table[threadIdx.x];
  __syncthreads();
  const u64 words = n / 16;
  for (u64 q = (u64)blockIdx.x * 256 + threadIdx.x; q <= words; q += (u64)gridDim.x * 256) {
    if (q < words) {
      const uint4 v = reinterpret_cast<const uint4 *>(src)[q];
      u32 w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int k = 0; k < 4; k++)
        w[k] = (u32)tab[w[k] & 255u] | ((u32)tab[(w[k] >> 8) & 255u] << 8) | ((u32)tab[(w[k] >> 16) & 255u] << 16) | ((u32)tab[w[k] >> 24] << 24);
      reinterpret_cast<uint4 *>(dst)[q] = make_uint4(w[0], w[1], w[2], w[3]);
    } else {
      for (u64 i = words * 16; i < n; i++) dst[i] = tab[src[i]];
    }
  }
}

void map_symbols(Ctx &c, const u8 *src, u8 *dst, u64 n, const u8 *d_table) {
  if (n == 0) return;
  if (((uintptr_t)src & 15) || ((uintptr_t)dst & 15)) fail(BWT_HIP_EINTERNAL, "map_symbols: unaligned buffers");
  const u32 blocks = (u32)std::min<u64>(ceil_div(n / 16 + 1, 256 * 4), (u64)c.n_cu * 32);
  hipLaunchKernelGGL(k_map_symbols, dim3(blocks ? blocks : 1), dim3(256), 0, c.stream, src, dst, n, d_table);
  HIP_CHECK(hipGetLastError());
}

void symbols_to_host(Ctx &c, const u8 *src_dev, u64 count, u8 *dst_host) {
  if (count == 0) return;
  if (!c.compact) {
    HIP_CHECK(hipMemcpyAsync(dst_host, src_dev, (size_t)count, hipMemcpyDeviceToHost, c.stream));
    c.sync();
    return;
  }
  // through a buffer of at most 256 MB, piece by piece (the source may begin at any address: a slice of L)
  const u64 piece = std::min<u64>(count, 256ull << 20);
  DBuf<u8> tmp(c, (size_t)piece + 32);
  for (u64 at = 0; at < count; at += piece) {
    const u64 len = std::min<u64>(piece, count - at);
    const u32 off = (u32)((uintptr_t)(src_dev + at) & 15);  // the aligned block in front of the piece is the source's own memory or the pool's
    map_symbols(c, src_dev + at - off, tmp.get(), len + off, c.d_sym_tab + 256);
    HIP_CHECK(hipMemcpyAsync(dst_host + at, tmp.get() + off, (size_t)len, hipMemcpyDeviceToHost, c.stream));
    c.sync();
  }
}

}  // namespace bwtamd
