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
//
// The first pass of the leaf suffix sort does not read a key array at all: its
// keys are a function of the text (TextKeys in internal.h), so the histogram and
// scatter kernels of that pass stage the tile's stretch of text in LDS and roll
// the keys out of it.  That removes the key-building pass (13 bytes written and
// 20 read back per suffix) from the leaf sort.
#include <string.h>

#include "device_utils.h"
#include "bucket_rank.h"
#include "internal.h"

namespace bwtamd {

constexpr int RS_THREADS = 1024;
constexpr int RS_WAVES = RS_THREADS / 64;
constexpr int RS_ITEMS = 8;
constexpr int RS_TILE = RS_THREADS * RS_ITEMS;  // 4096 keys per workgroup
constexpr int RS_PER_WAVE = RS_TILE / RS_WAVES; // 512 consecutive keys per wave
constexpr int RH_THREADS = 512;                 // histogram kernel: 512 threads x 16 keys
constexpr int RH_ITEMS = RS_TILE / RH_THREADS;



// ----------------------------------------------------------------------------
// keys rolled out of the text (first pass of the leaf suffix sort)
//   key(i) = | codes of the K0 symbols T[a+i .. a+i+K0), w bits each, first symbol highest,
//              left-aligned | zero padding | T[a+i-1] in the low 8 bits |   (TextKeys in internal.h)
//   value(i) = a + i
// Positions >= n read as 0; T[-1] is T[n-1].
// ----------------------------------------------------------------------------
constexpr int TK_LEAD = 4;  // staged bytes in front of the tile's first symbol (the preceding symbol is the last of them)
constexpr int TK_STAGE_WORDS = (TK_LEAD + RS_TILE + 64 + 4 + 3) / 4 + 1;

// S[k] = symbol at text position a + tile0 - TK_LEAD + k, for every k a key of the tile can touch
__device__ __forceinline__ void tk_stage(u32 *S32, const TextKeys &tk, u64 tile0, u32 tid, u32 nthreads) {
  const i64 p0 = (i64)(tk.a + tile0) - TK_LEAD;
  const u32 words = (u32)(TK_LEAD + RS_TILE + tk.K0 + 4 + 3) / 4;
  if (p0 >= 0 && (u64)p0 + (u64)words * 4 + 8 <= tk.n) {
    // aligned dword loads, shifted to the tile's own alignment
    const uintptr_t ad = reinterpret_cast<uintptr_t>(tk.T + p0);
    const u32 off = (u32)(ad & 3);
    const u32 *G = reinterpret_cast<const u32 *>(ad - off);
    for (u32 q = tid; q < words; q += nthreads) {
      u64 two = ((u64)G[q + 1] << 32) | G[q];
      S32[q] = (u32)(two >> (8 * off));
    }
  } else {
    u8 *S = reinterpret_cast<u8 *>(S32);
    for (u32 k = tid; k < words * 4; k += nthreads) {
      i64 p = p0 + (i64)k;
      u8 ch = 0;
      if (p == -1) ch = tk.T[tk.n - 1];
      else if (p >= 0 && (u64)p < tk.n) ch = tk.T[p];
      S[k] = ch;
    }
  }
}

// keys k0 .. k0+ITEMS-1 of the tile (k0 a multiple of 4), from the staged text
// RUNF: the text may hold a symbol without a code of its own (TextKeys::code_max); compiled apart, because the few
// registers it takes cost the common case a quarter of its waves.
template <int ITEMS, bool RUNF = false>
__device__ __forceinline__ void tk_roll(const u32 *S32, u32 k0, const TextKeys &tk, u64 key_out[ITEMS]) {
  const int w = tk.w, K0 = tk.K0;
  const int used = K0 * w;  // <= 56
  const u64 maskall = (1ull << used) - 1;
  const int lsh = 64 - used;
  const u32 ms = tk.min_sym - (u32)tk.one_based;  // code(c) = c - ms for a symbol, 0 for the terminator
  const u32 code_max = RUNF ? tk.code_max : 0u;
  const u32 skip = code_max ? tk.run_sym : 0x100u;  // (TextKeys::code_max: symbols from here on are coded one lower)
  const u32 *src = S32 + (TK_LEAD + k0) / 4;
  const u8 *raw = reinterpret_cast<const u8 *>(S32) + TK_LEAD + k0;
  if (w == 2 && !tk.one_based && !tk.base && K0 + ITEMS <= 32) {
    // Four symbols with 2-bit codes: four symbols per step.  Subtracting min_sym from every byte leaves
    // 0..3 in each; one multiplication gathers the four 2-bit fields, first symbol highest, in bits
    // 24..31 (the partial products land on distinct bits, so nothing carries).  A word with a zero byte
    // (the terminator or beyond the text) takes the byte loop below.
    constexpr int WORDS = (ITEMS + 32 + 3) / 4;
    const int words = (K0 + ITEMS + 3) / 4;
    u64 codes = 0;  // 4 * words symbols, first symbol highest
    u32 wds[WORDS / 2 + 1];
    bool plain = true;
#pragma unroll
    for (int q = 0; q < WORDS; q++) {
      if (q < words) {
        const u32 x = src[q];
        if (q < WORDS / 2 + 1) wds[q] = x;
        plain = plain && !((x - 0x01010101u) & ~x & 0x80808080u);
        u32 y = x - tk.min_sym * 0x01010101u;
        if (code_max) {  // a word that holds the symbol without a code takes the byte loop; larger symbols are coded one lower
          const u32 z = x ^ (tk.run_sym * 0x01010101u);
          plain = plain && !((z - 0x01010101u) & ~z & 0x80808080u);
          y -= (((x | 0x80808080u) - tk.run_sym * 0x01010101u) & 0x80808080u) >> 7;  // (symbols below 128: codes that fit two bits)
        }
        codes = (codes << 8) | (((y & 0x03030303u) * 0x40100401u) >> 24);
      }
    }
    if (plain) {
      const u32 before = src[-1] >> 24;
      const int top = 8 * words;  // bits in `codes`
#pragma unroll
      for (int j = 0; j < ITEMS; j++) {
        const u64 field = (codes >> (top - used - 2 * j)) & maskall;
        const u32 prev = j == 0 ? before : (wds[(j - 1) >> 2] >> (8 * ((j - 1) & 3))) & 0xFFu;
        key_out[j] = (field << lsh) | prev;
      }
      return;
    }
  }
  u64 sf = 0;
  u32 wd = 0;
  if (tk.base) {
    // the codes as digits of a number: roll = take the leading digit off, shift by one digit, add the new one
    const u64 B = tk.base;
    for (int k = 0; k < K0; k++) {
      if ((k & 3) == 0) wd = src[k >> 2];
      const u32 ch = wd & 255u;
      sf = sf * B + (ch ? ch - ms - (u32)(ch >= skip) : 0u);
      wd >>= 8;
    }
#pragma unroll
    for (int j = 0; j < ITEMS; j++) {
      key_out[j] = ((sf * tk.scale) << 8) | raw[j - 1];
      const u32 out = raw[j], in = raw[K0 + j];
      sf = (sf - (u64)(out ? out - ms - (u32)(out >= skip) : 0u) * tk.lead) * B + (in ? in - ms - (u32)(in >= skip) : 0u);
    }
    return;
  }
  // code_max: which of the key's symbols are the one that shares a code (bit k: the k-th) -- behind the first of
  // them the key shows the largest codes only, whatever follows (TextKeys::code_max)
  u64 shared = 0;
  for (int k = 0; k < K0; k++) {
    if ((k & 3) == 0) wd = src[k >> 2];
    u32 ch = wd & 255u;
    sf = (sf << w) | (ch ? ch - ms - (u32)(ch >= skip) : 0u);
    if (code_max) shared |= (u64)(ch == tk.run_sym) << k;
    wd >>= 8;
  }
#pragma unroll
  for (int j = 0; j < ITEMS; j++) {
    u64 field = sf;
    if (code_max && shared) field |= (1ull << (used - w * (__builtin_ctzll(shared) + 1))) - 1ull;
    key_out[j] = (field << lsh) | raw[j - 1];  // the preceding symbol as it is
    const int k = K0 + j;
    if ((k & 3) == 0) wd = src[k >> 2];
    u32 ch = wd & 255u;
    sf = ((sf << w) | (ch ? ch - ms - (u32)(ch >= skip) : 0u)) & maskall;
    if (code_max) shared = (shared >> 1) | ((u64)(ch == tk.run_sym) << (K0 - 1));
    wd >>= 8;
  }
}

// Is the suffix at tile index idx, whose key has been rolled, left out of the sort (TextKeys::run_span)?
__device__ __forceinline__ bool tk_left_out(const u32 *S32, u32 idx, u64 key, const TextKeys &tk) {
  if (!((key & ~0xFFull) - tk.run_lo < tk.run_span)) return false;  // (run_span 0: never)
  const u8 *raw = reinterpret_cast<const u8 *>(S32) + TK_LEAD + idx;
  for (u32 j = 0; j < tk.run_k; j++)
    if (raw[j] != tk.run_sym) return false;
  return true;
}

// USE only names the caller in the kernel symbol (0 leaf sort, 1 refinement
// rounds, 2 gap histogram), so that a profiler lists their launches apart.
// DB: digit bits (8; 9 where two wider passes replace three; 10 for the second pass of lean windows above 2^31 suffixes)
template <class K, int USE, bool TEXT, int DB = 8, bool RUNF = false>
__global__ void __launch_bounds__(RH_THREADS)
    k_rs_hist(const K *__restrict__ keys, size_t n, int shift, u32 mask, u32 *__restrict__ counts, u32 nblk,
              TextKeys tk, const uint2 *__restrict__ tmap = nullptr) {
  constexpr int DIG = 1 << DB;
  __shared__ u32 cnt[RH_THREADS / 64][DIG];  // one row of LDS counters per wave
  __shared__ u32 stage[TEXT ? TK_STAGE_WORDS : 1];
  const u32 tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  for (u32 d = lane; d < (u32)DIG; d += 64) cnt[w][d] = 0;
  const u32 tile = xcd_tile(blockIdx.x, nblk);
  const size_t tile0 = (size_t)tile * RS_TILE;
  if constexpr (TEXT) {
    // thread t counts keys 16t .. 16t+15 of the tile
    tk_stage(stage, tk, tile0, tid, RH_THREADS);
    __syncthreads();
    u64 key[RH_ITEMS];
    tk_roll<RH_ITEMS, RUNF>(stage, tid * RH_ITEMS, tk, key);
#pragma unroll
    for (int r = 0; r < RH_ITEMS; r++) {
      u32 d = (u32)(key[r] >> shift) & mask;
      bool counts = tile0 + tid * RH_ITEMS + r < n;
      if constexpr (RUNF) counts = counts && !tk_left_out(stage, tid * RH_ITEMS + r, key[r], tk);
      if (counts) atomicAdd(&cnt[w][d], 1u);
    }
    __syncthreads();
    for (u32 d = tid; d < (u32)DIG; d += RH_THREADS) {
      u32 tot = 0;
#pragma unroll
      for (int k = 0; k < RH_THREADS / 64; k++) tot += cnt[k][d];
      counts[(size_t)d * nblk + (nblk - 1 - tile)] = tot;  // tiles in reverse: this pass is reverse-stable
    }
    return;
  }
  K key[RH_ITEMS];
  if (tmap) {  // tiles of any start and length (TextKeys::lean): tmap[tile] = (first element, elements)
    const uint2 tm = tmap[tile];
    const K *src = keys + tm.x;
#pragma unroll
    for (int r = 0; r < RH_ITEMS; r++) {
      const u32 i = (u32)r * RH_THREADS + tid;
      key[r] = i < tm.y ? src[i] : (K)0;
    }
#pragma unroll
    for (int r = 0; r < RH_ITEMS; r++) {
      const u32 d = (u32)(key[r] >> shift) & mask;
      if ((u32)r * RH_THREADS + tid < tm.y) atomicAdd(&cnt[w][d], 1u);
    }
    __syncthreads();
    for (u32 d = tid; d < (u32)DIG; d += RH_THREADS) {
      u32 tot = 0;
#pragma unroll
      for (int k = 0; k < RH_THREADS / 64; k++) tot += cnt[k][d];
      counts[(size_t)d * nblk + tile] = tot;
    }
    return;
  }
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
  for (u32 d = tid; d < (u32)DIG; d += RH_THREADS) {
    u32 tot = 0;
#pragma unroll
    for (int k = 0; k < RH_THREADS / 64; k++) tot += cnt[k][d];
    counts[(size_t)d * nblk + tile] = tot;
  }
}

template <class K>
__device__ __forceinline__ u32 digit_of(K key, int shift, u32 mask) {
  return (u32)(key >> shift) & mask;
}

// Ranking, ordering in LDS and writing of one tile.  FULL: the tile has all RS_TILE keys.
// Tried and dropped: a persistent variant that loops over tiles and issues the next tile's loads while the
// current one is written out -- 80 registers do not hold both tiles' state (spills), and at 2 workgroups
// per CU it was 50 % slower than one tile per workgroup.
// REVERSE: equal digits leave in the reverse of their input order, over the whole array (the caller's
// histogram is then laid out with the tiles in reverse, see k_rs_hist).
// ANY_ORDER: equal digits may leave in any order (a first pass whose consumer does not care): the rank
// within the tile is then the old value of one LDS atomic add per key instead of the ballot ranking, which
// costs a fifth of its vector instructions (gap histogram: 1.19 -> 1.08 ms per pass).  The leaf sort's first
// pass gains as little and has to stay reverse-stable for the refinement rounds.
// SIDE: every key travels with one more byte (sd -> side_out), staged in side_s (RS_TILE bytes of LDS).
template <class K, bool HAS_VALS, bool FULL, bool SPLIT, bool REVERSE, bool ANY_ORDER = false, bool SIDE = false, int DB = 8>
__device__ __forceinline__ void rs_scatter_tile(K (&key)[RS_ITEMS], u32 (&val)[RS_ITEMS], unsigned char *buf,
                                                u32 *gbase, u32 *scan_s, K *__restrict__ kout,
                                                u32 *__restrict__ vout, u32 tile_n, int shift, u32 mask,
                                                const u32 *__restrict__ offs, u32 nblk, u32 tile, u32 stride,
                                                u8 *__restrict__ next_digits = nullptr, int next_shift = 0,
                                                const u8 *sd = nullptr, u8 *side_s = nullptr,
                                                u8 *__restrict__ side_out = nullptr, u32 lean_posmask = 0,
                                                u32 lean_tile0 = 0, u32 narrow_mask = 0, u32 keep_lo = 0, u32 keep_hi = 0,
                                                u32 dropm = 0, u32 *__restrict__ cursor = nullptr, u32 cell0 = 0, u32 slot = 0,
                                                u32 *__restrict__ over = nullptr, u32 dmul = 1) {
  constexpr int DIG = 1 << DB;
  u32(*cnt)[DIG] = reinterpret_cast<u32(*)[DIG]>(buf);
  u32 *dstart = reinterpret_cast<u32 *>(buf) + RS_WAVES * DIG;
  K *keys_s = reinterpret_cast<K *>(buf);
  u32 *vals_s = reinterpret_cast<u32 *>(buf + RS_TILE * sizeof(K));
  const u32 tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const u32 wbase = w * RS_PER_WAVE;
  u32 *myc = cnt[w];  // this wave's counters: no other wave touches them before the barrier
  u32 dig[RS_ITEMS], rk[RS_ITEMS];
  // A text pass that keeps only the digits keep_lo .. keep_hi-1 (a leaf shared by several GPUs) and leaves out the
  // elements whose bit in dropm is set (TextKeys::run_span): the others are neither ranked nor staged, the tile
  // shrinks to what is kept (tile_kept).
  const bool filt = REVERSE && sizeof(K) == 8 && !HAS_VALS && !SIDE && keep_hi != 0;
  u32 keepm = ~0u, tile_kept = tile_n;
  if constexpr (ANY_ORDER) {
    u32 *all = cnt[0];  // one row of counters for the whole tile
    if (tid < (u32)DIG) all[tid] = 0;
    if (cursor && tid == 0) scan_s[16] = 0;  // (never together with filt, which keeps the tile's count there)
    lds_barrier();
#pragma unroll
    for (int r = 0; r < RS_ITEMS; r++) {
      dig[r] = digit_of<K>(key[r], shift, mask);
      rk[r] = 0;
      bool valid = FULL || wbase + r * 64 + lane < tile_n;
      if (filt && (dig[r] < keep_lo || dig[r] >= keep_hi || ((dropm >> r) & 1u))) {
        valid = false;
        keepm &= ~(1u << r);
      }
      if (valid) rk[r] = atomicAdd(&all[dig[r]], 1u);
    }
    lds_barrier();
    const u32 tot = tid < (u32)DIG ? all[tid] : 0u;
    u32 incl = wave_scan_incl_sum(tot);
    if (lane == 63 && w < (u32)DIG / 64) scan_s[w] = incl;
    lds_barrier();
    for (u32 k = 0; k < w && k < (u32)DIG / 64; k++) incl += scan_s[k];
    if (tid < (u32)DIG) {
      const u32 ds = incl - tot;
      dstart[tid] = REVERSE ? ds + tot - 1 : ds;
      if (cursor) {
        // No table of offsets: digit d of this tile's region owns the `slot` elements from cell * slot on (cell =
        // cell0 + d * dmul), and the tile takes its place among them with one atomic add on the cell's fill count.
        u32 g = 0;
        if (tot) {
          const u32 cell = cell0 + tid * dmul;
          const u32 old = atomicAdd(&cursor[cell], tot);
          if (old + tot > slot) scan_s[16] = 1;
          g = cell * slot + old - ds;
        }
        gbase[tid] = g;
      } else {
        gbase[tid] = offs[(size_t)tid * nblk + (REVERSE ? nblk - 1 - tile : tile)] - ds;
      }
    }
    if (filt && tid == (u32)DIG - 1) scan_s[16] = incl;  // what the tile keeps
    lds_barrier();
    if (filt) tile_kept = scan_s[16];
    if (cursor && scan_s[16]) {  // a digit's elements do not fit its slots: the tile writes nothing, the pass is void
      if (tid == 0) *over = 1;
      return;
    }
#pragma unroll
    for (int r = 0; r < RS_ITEMS; r++) rk[r] = REVERSE ? dstart[dig[r]] - rk[r] : dstart[dig[r]] + rk[r];
  } else {
  for (u32 d = lane; d < (u32)DIG; d += 64) myc[d] = 0;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
#pragma unroll
  for (int r = 0; r < RS_ITEMS; r++) {
    const bool valid = FULL || (wbase + r * 64 + lane < tile_n);
    const u32 d = digit_of<K>(key[r], shift, mask);
    dig[r] = d;
    u32 lo = ~0u, hi = ~0u;
    if (!FULL) {
      u64 v = __ballot(valid);
      lo = (u32)v;
      hi = (u32)(v >> 32);
    }
    match_digit<DB>(d, lo, hi);
    const u32 before = count_below_lane(lo, hi);
    // The wave's LDS operations execute in program order: every peer reads the counter, then the
    // first peer adds the group's size; the next 64 keys see the sum.
    const u32 base = __hip_atomic_load(&myc[d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    if (valid && before == 0)
      __hip_atomic_fetch_add(&myc[d], (u32)(__popc(lo) + __popc(hi)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    rk[r] = base + before;
  }
  lds_barrier();
  {
    // thread d < 256 owns digit d: wave counts -> wave offsets, digit totals -> tile-local digit starts
    u32 tot = 0;
    if (tid < (u32)DIG) {
#pragma unroll
      for (int k = 0; k < RS_WAVES; k++) {
        u32 v = cnt[k][tid];
        cnt[k][tid] = tot;
        tot += v;
      }
    }
    // exclusive scan of the 256 digit totals: they sit in the first four waves
    u32 incl = wave_scan_incl_sum(tot);
    if (lane == 63 && w < (u32)DIG / 64) scan_s[w] = incl;
    lds_barrier();
    for (u32 k = 0; k < w && k < (u32)DIG / 64; k++) incl += scan_s[k];
    if (tid < (u32)DIG) {
      u32 ds = incl - tot;
      dstart[tid] = REVERSE ? ds + tot - 1 : ds;  // reverse: the digit's last slot in the tile
      gbase[tid] = offs[(size_t)tid * nblk + (REVERSE ? nblk - 1 - tile : tile)] - ds;
    }
  }
  lds_barrier();
#pragma unroll
  for (int r = 0; r < RS_ITEMS; r++) {  // position inside the tile
    if (REVERSE) rk[r] = dstart[dig[r]] - (myc[dig[r]] + rk[r]);
    else rk[r] += dstart[dig[r]] + myc[dig[r]];
  }
  }
  lds_barrier();  // counters are dead from here: the buffer becomes the staging area
#pragma unroll
  for (int r = 0; r < RS_ITEMS; r++) {
    if ((FULL || wbase + r * 64 + lane < tile_n) && ((keepm >> r) & 1u)) {
      keys_s[rk[r]] = key[r];
      if (HAS_VALS) vals_s[rk[r]] = val[r];
      if constexpr (SIDE) side_s[rk[r]] = sd[r];
    }
  }
  lds_barrier();
#pragma unroll
  for (int r = 0; r < RS_ITEMS; r++) {
    const u32 p = r * RS_THREADS + tid;
    if ((FULL && !filt) || p < tile_kept) {
      K k = keys_s[p];
      const u32 dg = digit_of<K>(k, shift, mask);
      if constexpr (REVERSE && sizeof(K) == 8 && !HAS_VALS && !SIDE) {
        // a text pass that keeps only the digits keep_lo .. keep_hi-1 (the caller has zeroed the counts of the others)
        if (keep_hi && (dg < keep_lo || dg >= keep_hi)) continue;
      }
      u32 dst = gbase[dg] + p;
      if constexpr (SPLIT) {
        u32 *klo = reinterpret_cast<u32 *>(kout);
        klo[dst] = (u32)k;
        klo[(size_t)stride + dst] = (u32)((u64)k >> 32);  // beyond 2^32 words for windows above 2^31
      } else {
        if constexpr (REVERSE && sizeof(K) == 8 && !HAS_VALS && !SIDE) {
          // lean elements (TextKeys::lean) were staged with this pass's digit above the low 13 bits of their
          // position field; the element proper has its window index there
          if (lean_posmask) k = (K)(((u64)k & ~(u64)lean_posmask) | (u64)(lean_tile0 + ((u32)k & (u32)(RS_TILE - 1))));
        }
        if constexpr (sizeof(K) == 4 && !HAS_VALS) {
          // last pass of a sort whose higher key bits are known from where a key lies: only the low bits go out, as
          // 16-bit words (the output array is then an array of those)
          if (narrow_mask) reinterpret_cast<u16 *>(kout)[dst] = (u16)((u32)k & narrow_mask);
          else kout[dst] = k;
        } else {
          kout[dst] = k;
        }
        if constexpr (REVERSE && sizeof(K) == 8) {  // the text pass: the next pass's digits on the side
          if (next_digits) next_digits[dst] = (u8)((u64)k >> next_shift);
        }
      }
      if (HAS_VALS) vout[dst] = vals_s[p];
      if constexpr (SIDE) side_out[dst] = side_s[p];
    }
  }
}

// SPLIT: a 64-bit key lives as two 32-bit words `stride` words apart (all low words, then all high
// words), so that a histogram pass reads only the half that holds its digit.
// SIDE: 8-byte keys-only elements with a side byte each (the wide layout of the leaf sort, TextKeys::wide):
// side_in[i] belongs to kin[i] and moves with it to side_out.  The text pass makes both from the text.
template <class K, bool HAS_VALS, int USE, bool TEXT, bool SPLIT, bool ANY_ORDER = false, bool SIDE = false, int DB = 8, bool RUNF = false>
__global__ void __launch_bounds__(RS_THREADS, (HAS_VALS && sizeof(K) == 8) ? 4 : 8)
    k_rs_scatter(const K *__restrict__ kin, const u32 *__restrict__ vin, K *__restrict__ kout,
                 u32 *__restrict__ vout, size_t n, int shift, u32 mask, const u32 *__restrict__ offs,
                 u32 nblk, TextKeys tk, u32 stride, const u8 *__restrict__ side_in = nullptr,
                 u8 *__restrict__ side_out = nullptr, const uint2 *__restrict__ tmap = nullptr, u32 narrow_mask = 0,
                 u32 *__restrict__ cursor = nullptr, u32 slot = 0, const u32 *__restrict__ treg = nullptr,
                 u32 *__restrict__ over = nullptr, u32 dmul = 1) {
  static_assert(!SIDE || (sizeof(K) == 8 && !HAS_VALS && !SPLIT), "side bytes go with 8-byte keys-only elements");
  u8 sd[RS_ITEMS];
  static_assert(!SPLIT || sizeof(K) == 8, "only 64-bit keys are split");
  // One LDS buffer, used twice: first the per-wave digit counters and the digit
  // starts, then (after every thread has turned its ranks into tile positions)
  // the tile's keys and values in digit order.
  constexpr size_t STAGE_BYTES = RS_TILE * sizeof(K) + (HAS_VALS ? RS_TILE * 4 : 0);
  constexpr int DIG = 1 << DB;
  constexpr size_t COUNT_BYTES = (RS_WAVES + 1) * DIG * 4;
  constexpr size_t TEXT_BYTES = TEXT ? (size_t)RS_TILE * 8 + TK_STAGE_WORDS * 4 : 0;  // key exchange + staged text
  constexpr size_t BUF0 = (STAGE_BYTES > COUNT_BYTES ? STAGE_BYTES : COUNT_BYTES) + (SIDE ? RS_TILE : 0);
  constexpr size_t BUF_BYTES = BUF0 > TEXT_BYTES ? BUF0 : TEXT_BYTES;
  __shared__ __align__(16) unsigned char buf[BUF_BYTES];
  u8 *side_s = buf + STAGE_BYTES;  // SIDE: behind the staged keys (where the text pass kept its stretch of text before)
  __shared__ u32 gbase[DIG];
  __shared__ u32 scan_s[17];

  const u32 tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const u32 tile = xcd_tile(blockIdx.x, nblk);
  size_t tile0 = (size_t)tile * RS_TILE;
  u32 tile_n = (u32)((n - tile0) < (size_t)RS_TILE ? (n - tile0) : (size_t)RS_TILE);
  if constexpr (!TEXT && !SPLIT && !HAS_VALS && !SIDE) {
    if (tmap) {  // tiles of any start and length (TextKeys::lean)
      const uint2 tm = tmap[tile];
      tile0 = tm.x;
      tile_n = tm.y;
      if (tile_n == 0) return;
    }
  }
  const bool full = tile_n == RS_TILE;

  K key[RS_ITEMS];
  u32 val[RS_ITEMS];
  u32 dropm = 0;  // elements of the text pass that are left out (TextKeys::run_span)
  const u32 wbase = w * RS_PER_WAVE;
  if constexpr (TEXT) {
    // The same buffer first holds the staged text (behind the key exchange area) and the keys that
    // thread t rolled for tile indices 8t .. 8t+7; every thread then picks up the keys of its ranking
    // slots.  Thread t keeps key j in slot j ^ (t/4 mod 8) of its eight: 32 neighbouring threads then
    // write to 32 different bank pairs, and the readers' 64 consecutive indices stay 64 distinct slots.
    static_assert(sizeof(K) == 8 && !(SPLIT && !HAS_VALS), "text keys: 64-bit with a position, or packed elements");
    static_assert(RS_ITEMS == 8, "slot swizzle of the key exchange");
    constexpr u32 EX_WORDS = RS_TILE;
    static_assert(EX_WORDS * 8 + TK_STAGE_WORDS * 4 <= BUF_BYTES, "exchange area and staged text share the buffer");
    K *ex = reinterpret_cast<K *>(buf);
    u32 *stage = reinterpret_cast<u32 *>(buf + EX_WORDS * 8);
    tk_stage(stage, tk, tile0, tid, RS_THREADS);
    __syncthreads();
    u64 mine[RS_ITEMS];
    tk_roll<RS_ITEMS, RUNF>(stage, tid * RS_ITEMS, tk, mine);
    // Where the order inside a digit is free (ANY_ORDER) a thread ranks the keys it rolled itself: no exchange.
    // (Only whole tiles: the ranking's bookkeeping of a short tile assumes the slots of the exchange.)
    const bool direct = ANY_ORDER && full;
    if (!direct) {
#pragma unroll
      for (int j = 0; j < RS_ITEMS; j++) ex[tid * RS_ITEMS + (j ^ ((tid >> 2) & 7))] = mine[j];
      __syncthreads();
    }
    const u32 pos0 = (u32)(tk.a + tile0);
#pragma unroll
    for (int r = 0; r < RS_ITEMS; r++) {
      u32 idx = direct ? tid * RS_ITEMS + r : wbase + r * 64 + lane;
      bool valid = idx < tile_n;
      u64 k = direct ? mine[r] : (valid ? ex[(idx & ~7u) | ((idx & 7) ^ ((idx >> 5) & 7))] : 0ull);
      if constexpr (HAS_VALS) {
        key[r] = (K)k;
        val[r] = pos0 + idx;
      } else if constexpr (SIDE) {  // wide element: key bits 63..32 | absolute position; side: key bits 31..28 | preceding symbol
        key[r] = (K)((k & 0xFFFFFFFF00000000ull) | (u32)(pos0 + idx));
        sd[r] = (u8)((((u32)k >> 28) << 4) | ((u32)k & 0xFu));
      } else if (tk.lean) {
        // lean element as it is staged: | d2 | deep | preceding symbol | d1 above bit 13 | index in the tile |
        // (d1 = the key's top b1 bits, which this pass sorts on and does not store; d2 = the b2 bits below them)
        const u64 hi = (k << tk.b1) & (~0ull << (64 - tk.b2 - tk.lean_deep));
        const u32 d1 = (u32)(k >> (64 - tk.b1));
        if constexpr (RUNF) dropm |= (u32)tk_left_out(stage, idx, k, tk) << r;
        key[r] = (K)(hi | ((u64)((u32)k & 0xFFu) << tk.pbits) | (u64)(d1 << 13) | idx);
      } else {  // packed element: codes | preceding symbol | window index
        key[r] = (K)((k & 0xFFFFFFFF00000000ull) | (((u32)k & 0xFFu) << tk.pbits) | (u32)(tile0 + idx));
      }
    }
    if (!direct) __syncthreads();  // the exchange area becomes the counters
  } else if constexpr (SPLIT) {
    const u32 *lo = reinterpret_cast<const u32 *>(kin) + tile0;
    const u32 *hi = lo + stride;
    const u32 *vt = vin + tile0;
    if (full) {  // no per-lane conditions: all loads are issued back to back
#pragma unroll
      for (int r = 0; r < RS_ITEMS; r++) {
        u32 idx = wbase + r * 64 + lane;
        key[r] = (K)(((u64)hi[idx] << 32) | lo[idx]);
        if (HAS_VALS) val[r] = vt[idx];
      }
    } else {
#pragma unroll
      for (int r = 0; r < RS_ITEMS; r++) {
        u32 idx = wbase + r * 64 + lane;
        bool valid = idx < tile_n;
        u32 l = valid ? lo[idx] : 0u, h = valid ? hi[idx] : 0u;
        key[r] = (K)(((u64)h << 32) | l);
        if (HAS_VALS) val[r] = valid ? vt[idx] : 0u;
      }
    }
  } else {
    const K *kt = kin + tile0;      // uniform base + 32-bit lane offset
    const u32 *vt = vin + tile0;
    const u8 *st = side_in + tile0;
    if (full) {
#pragma unroll
      for (int r = 0; r < RS_ITEMS; r++) {
        u32 idx = wbase + r * 64 + lane;
        key[r] = kt[idx];
        if (HAS_VALS) val[r] = vt[idx];
        if constexpr (SIDE) sd[r] = st[idx];
      }
    } else {
#pragma unroll
      for (int r = 0; r < RS_ITEMS; r++) {
        u32 idx = wbase + r * 64 + lane;
        bool valid = idx < tile_n;
        key[r] = valid ? kt[idx] : (K)0;
        if (HAS_VALS) val[r] = valid ? vt[idx] : 0u;
        if constexpr (SIDE) sd[r] = valid ? st[idx] : (u8)0;
      }
    }
  }
  const u32 lean_posmask = (TEXT && tk.lean) ? (u32)((1ull << tk.pbits) - 1ull) : 0u;
  // Slots instead of offsets (cursor): cell = region * 2^DB + digit (dmul 1: the leaf sort, region-major), or
  // digit * dmul + region (the gap grouping: its second digit is the more significant one); no regions: cell = digit.
  // A pass over plain tiles (no treg): cell = digit * dmul + workgroup mod dmul, dmul areas per digit.  With eight
  // areas that is an area per XCD (workgroups go round the XCDs): the runs that neighbour each other in an area then
  // come from one XCD, whose L2 joins them into whole lines before they leave for memory -- with the areas taken by
  // tile number instead, neighbouring runs come from different L2s and the pass takes 5.3 ms instead of 3.7.
  const u32 cell0 = !cursor ? 0u : (treg ? (dmul == 1 ? treg[tile] << DB : treg[tile]) : blockIdx.x % dmul);
  if (full)
    rs_scatter_tile<K, HAS_VALS, true, SPLIT, TEXT, ANY_ORDER, SIDE, DB>(key, val, buf, gbase, scan_s, kout, vout, tile_n, shift, mask, offs, nblk,
                                              tile, stride, TEXT ? tk.next_digits : nullptr, tk.next_shift, sd, side_s, side_out,
                                              lean_posmask, (u32)tile0, narrow_mask, TEXT ? tk.keep_lo : 0u, TEXT ? tk.keep_hi : 0u, dropm, cursor, cell0, slot, over, dmul);
  else
    rs_scatter_tile<K, HAS_VALS, false, SPLIT, TEXT, ANY_ORDER, SIDE, DB>(key, val, buf, gbase, scan_s, kout, vout, tile_n, shift, mask, offs,
                                               nblk, tile, stride, TEXT ? tk.next_digits : nullptr, tk.next_shift, sd, side_s, side_out,
                                               lean_posmask, (u32)tile0, narrow_mask, TEXT ? tk.keep_lo : 0u, TEXT ? tk.keep_hi : 0u, dropm, cursor, cell0, slot, over, dmul);
}

// tile histogram of digits that are already bytes (TextKeys::next_digits): 16 of them per thread
__global__ void __launch_bounds__(RH_THREADS)
    k_rs_hist_bytes(const u8 *__restrict__ digits, size_t n, u32 *__restrict__ counts, u32 nblk,
                    const uint2 *__restrict__ tmap = nullptr) {
  __shared__ u32 cnt[RH_THREADS / 64][256];
  static_assert(RS_TILE == RH_THREADS * 16, "one 16-byte load per thread covers the tile");
  const u32 tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  for (u32 d = lane; d < 256; d += 64) cnt[w][d] = 0;
  const u32 tile = xcd_tile(blockIdx.x, nblk);
  const size_t i0 = (size_t)tile * RS_TILE + (size_t)tid * 16;
  __builtin_amdgcn_wave_barrier();
  if (tmap) {
    // a tile of any start and length (TextKeys::lean): the aligned 16-byte pieces that cover it, at most 513 of
    // them (the array ends in 16 spare bytes), each byte counted if it lies inside the tile
    const uint2 tm = tmap[tile];
    const size_t first = (size_t)tm.x & ~(size_t)15, end = (size_t)tm.x + tm.y;
    for (size_t p0 = first + (size_t)tid * 16; p0 < end; p0 += (size_t)RH_THREADS * 16) {
      const uint4 v = *reinterpret_cast<const uint4 *>(digits + p0);
      const u32 wd[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int q = 0; q < 4; q++)
#pragma unroll
        for (int b = 0; b < 4; b++) {
          const size_t p = p0 + (size_t)(4 * q + b);
          if (p >= tm.x && p < end) atomicAdd(&cnt[w][(wd[q] >> (8 * b)) & 255u], 1u);
        }
    }
  } else if (i0 + 16 <= n) {
    const uint4 v = *reinterpret_cast<const uint4 *>(digits + i0);
    const u32 wd[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int q = 0; q < 4; q++)
#pragma unroll
      for (int b = 0; b < 4; b++) atomicAdd(&cnt[w][(wd[q] >> (8 * b)) & 255u], 1u);
  } else {
    for (size_t i = i0; i < n && i < i0 + 16; i++) atomicAdd(&cnt[w][digits[i]], 1u);
  }
  __syncthreads();
  if (tid < 256) {
    u32 tot = 0;
#pragma unroll
    for (int k = 0; k < RH_THREADS / 64; k++) tot += cnt[k][tid];
    counts[(size_t)tid * nblk + tile] = tot;
  }
}

template <class K, int USE, int DB = 8>
static int radix_sort_impl(Ctx &c, K *keys[2], u32 *vals[2], size_t n, int begin_bit, int end_bit) {
  if (n == 0 || end_bit <= begin_bit) return 0;
  if (n >= (1ull << 32)) fail(BWT_HIP_EINVAL, "radix sort: %zu elements exceed 2^32-1", n);
  u32 nblk = ceil_div(n, RS_TILE);
  constexpr int DIG = 1 << DB;
  DBuf<u32> counts(c, (size_t)DIG * nblk);
  const bool has_vals = vals[0] != nullptr;
  TextKeys none;
  memset(&none, 0, sizeof(none));
  int cur = 0;
  for (int shift = begin_bit; shift < end_bit; shift += DB) {
    int bits = end_bit - shift < DB ? end_bit - shift : DB;
    u32 mask = (1u << bits) - 1;
    {
      KernelTimer kt(c, (std::string(c.sort_tag) + (sizeof(K) == 8 ? "k_rs_hist_u64" : "k_rs_hist_u32")).c_str(),
                     (double)n * sizeof(K));
      hipLaunchKernelGGL((k_rs_hist<K, USE, false, DB>), dim3(nblk), dim3(RH_THREADS), 0, c.stream, keys[cur], n, shift,
                         mask, counts.get(), nblk, none);
      HIP_CHECK(hipGetLastError());
    }
    scan_exclusive_u32(c, counts.get(), counts.get(), (size_t)DIG * nblk, nullptr);
    {
      const char *nm = sizeof(K) == 8 ? (has_vals ? "k_rs_scatter_u64_pairs" : "k_rs_scatter_u64_keys")
                                      : (has_vals ? "k_rs_scatter_u32_pairs" : "k_rs_scatter_u32_keys");
      KernelTimer kt(c, (std::string(c.sort_tag) + nm).c_str(), (double)n * 2 * (sizeof(K) + (has_vals ? 4 : 0)));
      if (has_vals)
        hipLaunchKernelGGL((k_rs_scatter<K, true, USE, false, false, false, false, DB>), dim3(nblk), dim3(RS_THREADS), 0, c.stream,
                           keys[cur], vals[cur], keys[cur ^ 1], vals[cur ^ 1], n, shift, mask, counts.get(), nblk,
                           none, 0u);
      else if (USE == 2 && shift == begin_bit)  // the gap histogram only counts inside buckets: the first pass may reorder
        hipLaunchKernelGGL((k_rs_scatter<K, false, USE, false, false, USE == 2, false, DB>), dim3(nblk), dim3(RS_THREADS), 0, c.stream,
                           keys[cur], (const u32 *)nullptr, keys[cur ^ 1], (u32 *)nullptr, n, shift, mask,
                           counts.get(), nblk, none, 0u);
      else
        hipLaunchKernelGGL((k_rs_scatter<K, false, USE, false, false, false, false, DB>), dim3(nblk), dim3(RS_THREADS), 0, c.stream,
                           keys[cur], (const u32 *)nullptr, keys[cur ^ 1], (u32 *)nullptr, n, shift, mask,
                           counts.get(), nblk, none, 0u);
      HIP_CHECK(hipGetLastError());
    }
    cur ^= 1;
  }
  return cur;
}

int radix_sort_u64(Ctx &c, u64 *keys[2], u32 *vals[2], size_t n, int begin_bit, int end_bit) {
  return radix_sort_impl<u64, 1>(c, keys, vals, n, begin_bit, end_bit);
}

int radix_sort_u32(Ctx &c, u32 *keys[2], u32 *vals[2], size_t n, int begin_bit, int end_bit, int use) {
  // 17 or 18 bits: two passes of 9-bit digits instead of three of 8 (shorter output runs per pass, but a pass less)
  if (use == 2 && !vals[0] && end_bit - begin_bit > 16 && end_bit - begin_bit <= 18)
    return radix_sort_impl<u32, 2, 9>(c, keys, vals, n, begin_bit, end_bit);
  if (use == 2) return radix_sort_impl<u32, 2>(c, keys, vals, n, begin_bit, end_bit);
  return radix_sort_impl<u32, 1>(c, keys, vals, n, begin_bit, end_bit);
}

// Round 0 of the leaf suffix sort: (key, position) pairs of TextKeys, sorted on key bits
// [begin_bit, 64).  Keys are kept split (see k_rs_scatter); begin_bit is a multiple of 8 so that no
// digit straddles the two words.  Buffer b holds low words at keys[b][0..n) and high words at
// keys[b][stride..stride+n) counted in 32-bit words; returns the buffer that holds the result.
int leaf_sort_text_keys(Ctx &c, u64 *keys[2], u32 *vals[2], size_t n, u32 stride, int begin_bit,
                        const TextKeys &text) {
  if (n == 0) return 0;
  if (n >= (1ull << 32)) fail(BWT_HIP_EINVAL, "radix sort: %zu elements exceed 2^32-1", n);
  if (begin_bit % 8 != 0 || begin_bit < 0 || begin_bit >= 64 || stride < n || stride % 4 != 0)
    fail(BWT_HIP_EINTERNAL, "leaf sort: bad key layout (begin_bit %d, stride %u, n %zu)", begin_bit, stride, n);
  u32 nblk = ceil_div(n, RS_TILE);
  DBuf<u32> counts(c, (size_t)256 * nblk);
  TextKeys none;
  memset(&none, 0, sizeof(none));
  int cur = 0;
  for (int shift = begin_bit; shift < 64; shift += 8) {
    const u32 mask = 255u;
    if (shift == begin_bit) {  // keys and positions of this pass come out of the text
      {
        KernelTimer kt(c, "leaf:k_rs_hist_text", (double)n);
        hipLaunchKernelGGL((k_rs_hist<u64, 0, true>), dim3(nblk), dim3(RH_THREADS), 0, c.stream, (const u64 *)nullptr,
                           n, shift, mask, counts.get(), nblk, text);
        HIP_CHECK(hipGetLastError());
      }
      scan_exclusive_u32(c, counts.get(), counts.get(), (size_t)256 * nblk, nullptr);
      {
        KernelTimer kt(c, "leaf:k_rs_scatter_text", (double)n * 13);
        hipLaunchKernelGGL((k_rs_scatter<u64, true, 0, true, true>), dim3(nblk), dim3(RS_THREADS), 0, c.stream,
                           (const u64 *)nullptr, (const u32 *)nullptr, keys[cur ^ 1], vals[cur ^ 1], n, shift, mask,
                           counts.get(), nblk, text, stride);
        HIP_CHECK(hipGetLastError());
      }
    } else {
      {
        // the digit's half of the keys only
        const u32 *half = reinterpret_cast<const u32 *>(keys[cur]) + (shift >= 32 ? stride : 0u);
        KernelTimer kt(c, "leaf:k_rs_hist_half", (double)n * 4);
        hipLaunchKernelGGL((k_rs_hist<u32, 0, false>), dim3(nblk), dim3(RH_THREADS), 0, c.stream, half, n, shift & 31,
                           mask, counts.get(), nblk, none);
        HIP_CHECK(hipGetLastError());
      }
      scan_exclusive_u32(c, counts.get(), counts.get(), (size_t)256 * nblk, nullptr);
      {
        KernelTimer kt(c, "leaf:k_rs_scatter_u64_pairs", (double)n * 24);
        hipLaunchKernelGGL((k_rs_scatter<u64, true, 0, false, true>), dim3(nblk), dim3(RS_THREADS), 0, c.stream,
                           keys[cur], vals[cur], keys[cur ^ 1], vals[cur ^ 1], n, shift, mask, counts.get(), nblk,
                           none, stride);
        HIP_CHECK(hipGetLastError());
      }
    }
    cur ^= 1;
  }
  return cur;
}

// ----------------------------------------------------------------------------
// Bucket route for 8-byte elements with 32 sorted key bits: after two passes on the top 16 bits the
// elements with the same top 16 bits (a bucket) are contiguous; one workgroup loads a bucket, sorts it on
// the other bits with one or two ranking rounds in LDS and writes it back in place.  The result is the one
// four stable passes give, for half the trips through memory of the last two.  Needs every bucket to
// fit a workgroup (BK_CAP elements): texts with a skewed distribution of 8-symbol words take the passes.
// ----------------------------------------------------------------------------
// start[k] = number of elements whose top `bbits` bits are below k, k = 0 .. 2^bbits (e is sorted on them)
__global__ void k_bucket_bounds(const u64 *__restrict__ e, u32 n, u32 *__restrict__ start, u32 n_buckets, int bbits) {
  const u32 k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k > n_buckets) return;
  u32 lo = 0, hi = n;
  while (lo < hi) {
    const u32 mid = lo + (hi - lo) / 2;
    if ((u32)(e[mid] >> (64 - bbits)) < k) lo = mid + 1;
    else hi = mid;
  }
  start[k] = lo;
}

__global__ void k_bucket_max(const u32 *__restrict__ start, u32 *__restrict__ out, u32 n_buckets) {
  const u32 k = blockIdx.x * blockDim.x + threadIdx.x;
  u32 v = k < n_buckets ? start[k + 1] - start[k] : 0u;
  v = wave_reduce_max(v);
  if (lane_id() == 0) atomicMax(out, v);
}

__global__ void __launch_bounds__(BK_THREADS) k_bucket_sort(u64 *__restrict__ e, const u32 *__restrict__ start, int begin_bit) {
  // the bucket's elements in digit order; the ranking counters use the same memory while the keys are in registers
  __shared__ __align__(16) u64 buf[BK_CAP];
  __shared__ u32 scan_s[4];
  static_assert(sizeof(u64) * BK_CAP >= (BK_WAVES + 1) * 256 * 4, "counters fit the staging area");
  const u32 s = start[blockIdx.x], cnt = start[blockIdx.x + 1] - s;  // cnt <= BK_CAP: the caller has checked
  if (cnt < 2) return;
  const u32 tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const u32 wbase = w * BK_PER_WAVE;
  u64 *mine = e + s;
  u32(*ctr)[256] = reinterpret_cast<u32(*)[256]>(buf);
  u64 key[BK_ITEMS];
  u32 pos[BK_ITEMS];
#pragma unroll
  for (int r = 0; r < BK_ITEMS; r++) {
    const u32 idx = wbase + r * 64 + lane;
    key[r] = idx < cnt ? mine[idx] : 0ull;
  }
#pragma unroll 1
  for (int shift = begin_bit; shift < 48; shift += 8) {
    bk_rank(key, shift, cnt, ctr, scan_s, pos);
#pragma unroll
    for (int r = 0; r < BK_ITEMS; r++)
      if (wbase + r * 64 + lane < cnt) buf[pos[r]] = key[r];
    lds_barrier();
    if (shift + 8 < 48) {
#pragma unroll
      for (int r = 0; r < BK_ITEMS; r++) {
        const u32 idx = wbase + r * 64 + lane;
        key[r] = idx < cnt ? buf[idx] : 0ull;
      }
      lds_barrier();  // the next round's counters overwrite the buffer
    }
  }
  for (u32 i = tid; i < cnt; i += BK_THREADS) mine[i] = buf[i];
}

// per_bucket[k] += the sampled text positions whose key, as the first radix pass builds it (TextKeys), has top
// 16 bits k.  Samples: `count` positions from `first` on, in runs of 16 consecutive positions that start
// `stride` apart (stride 16: every position).  One position per thread.
__global__ void k_bucket_histogram(TextKeys tk, u64 first, u64 count, u64 stride, u32 *__restrict__ per_bucket, int bbits) {
  const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= count) return;
  const u64 p = first + (t >> 4) * stride + (t & 15);
  const u32 ms = tk.min_sym - (u32)tk.one_based;
  u64 sf = 0;
  const u32 skip = tk.code_max ? tk.run_sym : 0x100u, top = (1u << tk.w) - 1u;
  u32 copies = 0;  // of the run symbol at the suffix's start
  bool seen = false;  // code_max: a symbol that shares its code has gone by; the largest code from there on
  for (int k = 0; k < tk.K0; k++) {
    const u32 ch = p + k < tk.n ? tk.T[p + k] : 0u;
    const u32 code = seen ? top : (ch ? ch - ms - (u32)(ch >= skip) : 0u);
    if (tk.base) sf = sf * tk.base + code;
    else sf = (sf << tk.w) | code;
    copies += (copies == (u32)k && ch == tk.run_sym);
    seen = seen || (tk.code_max && ch == tk.run_sym);
  }
  const u64 key = tk.base ? (sf * tk.scale) << 8 : sf << (64 - tk.K0 * tk.w);
  if (key - tk.run_lo < tk.run_span && copies >= tk.run_k) return;  // left out of the sort (run_span 0: never)
  atomicAdd(&per_bucket[(u32)(key >> (64 - bbits))], 1u);
}

void bucket_histogram(Ctx &c, const TextKeys &tk, u64 first, u64 count, u64 stride, u32 *per_bucket, int bbits) {
  if (count == 0) return;
  hipLaunchKernelGGL(k_bucket_histogram, dim3(ceil_div(count, 256)), dim3(256), 0, c.stream, tk, first, count, stride,
                     per_bucket, bbits);
  HIP_CHECK(hipGetLastError());
}

__global__ void k_max_u32(const u32 *__restrict__ v, u32 n, u32 *__restrict__ out) {
  const u32 k = blockIdx.x * blockDim.x + threadIdx.x;
  u32 x = wave_reduce_max(k < n ? v[k] : 0u);
  if (lane_id() == 0) atomicMax(out, x);
}

// A sample of the window (16 positions out of every 4096) says whether some bucket will be far too large
// for a workgroup (twice its capacity: sampling noise must not turn an even text away); the two passes on
// the top 16 bits are then not worth trying.
// out[r] = sum of the `per` values of v from r * per on (one wave per r)
__global__ void __launch_bounds__(64) k_group_sums(const u32 *__restrict__ v, u32 per, u32 *__restrict__ out) {
  u32 sum = 0;
  for (u32 k = threadIdx.x; k < per; k += 64) sum += v[(size_t)blockIdx.x * per + k];
  sum = wave_reduce_sum(sum);
  if (threadIdx.x == 0) out[blockIdx.x] = sum;
}

// regions (may be null): out, the sampled positions whose key's top `rbits` bits are r, for every r, and in
// *sampled how many positions were sampled in all (left empty when there are too few samples).
static bool buckets_look_even(Ctx &c, const TextKeys &tk, size_t n, int bbits = 16, u32 cap = BK_CAP, std::vector<u64> *regions = nullptr,
                              int rbits = 0, u64 *sampled = nullptr) {
  constexpr u64 STRIDE = 256;  // one position in 256
  const u32 n_buckets = 1u << bbits;
  const u64 count = n / (16 * STRIDE) * 16;
  if (regions) regions->clear();
  if (count < 4 * (u64)n_buckets) return true;  // too few samples to say anything
  KernelTimer kt(c, "leaf:k_bucket_histogram", (double)count * 8);
  DBuf<u32> hist(c, n_buckets);
  u32 *d_max = c.d_scalars + 12;
  HIP_CHECK(hipMemsetAsync(hist.get(), 0, n_buckets * sizeof(u32), c.stream));
  HIP_CHECK(hipMemsetAsync(d_max, 0, sizeof(u32), c.stream));
  bucket_histogram(c, tk, tk.a, count, 16 * STRIDE, hist.get(), bbits);
  hipLaunchKernelGGL(k_max_u32, dim3(n_buckets / 256), dim3(256), 0, c.stream, hist.get(), n_buckets, d_max);
  HIP_CHECK(hipGetLastError());
  HIP_CHECK(hipMemcpyAsync(c.h_scalars, d_max, sizeof(u32), hipMemcpyDeviceToHost, c.stream));
  std::vector<u32> h;
  DBuf<u32> sums;
  if (regions) {
    const u32 nr = 1u << rbits;
    h.resize(nr);
    sums = DBuf<u32>(c, nr);
    hipLaunchKernelGGL(k_group_sums, dim3(nr), dim3(64), 0, c.stream, (const u32 *)hist.get(), n_buckets >> rbits, sums.get());
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipMemcpyAsync(h.data(), sums.get(), (size_t)nr * sizeof(u32), hipMemcpyDeviceToHost, c.stream));
  }
  c.sync();
  if (regions) {
    regions->assign(h.begin(), h.end());
    *sampled = count;
  }
  return (u64)c.h_scalars[0] * STRIDE <= (u64)cap * 2;
}

// Bucket bounds of e (already sorted on its top bbits bits).  False if some bucket holds more than cap elements.
static bool bucket_bounds(Ctx &c, const u64 *e, size_t n, DBuf<u32> &start, u32 *max_bucket, int bbits = 16, u32 cap = BK_CAP) {
  const u32 n_buckets = 1u << bbits;
  start = DBuf<u32>(c, (size_t)n_buckets + 1);
  u32 *d_max = c.d_scalars + 12;
  {
    KernelTimer kt(c, "leaf:k_bucket_bounds", (double)n_buckets * 8 * 29);
    HIP_CHECK(hipMemsetAsync(d_max, 0, sizeof(u32), c.stream));
    hipLaunchKernelGGL(k_bucket_bounds, dim3(n_buckets / 256 + 1), dim3(256), 0, c.stream, e, (u32)n, start.get(), n_buckets, bbits);
    hipLaunchKernelGGL(k_bucket_max, dim3(n_buckets / 256), dim3(256), 0, c.stream, start.get(), d_max, n_buckets);
    HIP_CHECK(hipGetLastError());
  }
  HIP_CHECK(hipMemcpyAsync(c.h_scalars, d_max, sizeof(u32), hipMemcpyDeviceToHost, c.stream));
  c.sync();
  *max_bucket = c.h_scalars[0];
  return c.h_scalars[0] <= cap;
}

// Round 0 with WIDE elements, for windows whose positions need all 32 bits (TextKeys::wide): radix passes on
// the top bbits key bits only (16 < bbits <= 24: two passes of 9-bit digits up to 18 bits, three of 8-bit digits
// beyond; the first one -- from the text, on the lowest of those bits -- reverse-stable as always), then the
// bucket bounds.  The caller finishes the
// buckets from LDS (k_bucket_emit).  elems[b] / sides[b]: the two element and side-byte buffers; returns the
// one that holds the result, or -1 if some bucket holds more than `cap` elements (nothing usable then).
int leaf_sort_text_wide(Ctx &c, u64 *elems[2], u8 *sides[2], size_t n, int bbits, u32 cap, const TextKeys &text,
                        BucketState *bs) {
  if (bbits <= 16 || bbits > 24 || !text.wide || text.K0 * text.w > 36 || n >= (1ull << 32))
    fail(BWT_HIP_EINTERNAL, "leaf sort: bad wide layout (bbits %d, K0 %d, w %d, n %zu)", bbits, text.K0, text.w, n);
  if (!buckets_look_even(c, text, n, bbits, cap)) return -1;
  const u32 nblk = ceil_div(n, RS_TILE);
  TextKeys none;
  memset(&none, 0, sizeof(none));
  int cur = 0;
  if (bbits <= 18) {
    // two passes of 9-bit digits (the first one on the lower bbits - 9 of them)
    DBuf<u32> counts(c, (size_t)512 * nblk);
    TextKeys first = text;
    first.next_digits = nullptr;
    TextKeys count_keys = first;  // the histogram only needs the top bbits bits: 16 symbols keep its fast key roll
    if (count_keys.K0 * count_keys.w > 32) count_keys.K0 = 32 / count_keys.w;
    const int shifts[2] = {64 - bbits, 55};
    for (int p = 0; p < 2; p++) {
      const int shift = shifts[p];
      const u32 mask = p == 0 ? (1u << (bbits - 9)) - 1u : 511u;
      if (p == 0) {
        {
          KernelTimer kt(c, "leaf:k_rs_hist_text", (double)n);
          hipLaunchKernelGGL((k_rs_hist<u64, 0, true, 9>), dim3(nblk), dim3(RH_THREADS), 0, c.stream, (const u64 *)nullptr, n,
                             shift, mask, counts.get(), nblk, count_keys);
          HIP_CHECK(hipGetLastError());
        }
        scan_exclusive_u32(c, counts.get(), counts.get(), (size_t)512 * nblk, nullptr);
        KernelTimer kt(c, "leaf:k_rs_scatter_text", (double)n * 10);
        hipLaunchKernelGGL((k_rs_scatter<u64, false, 0, true, false, false, true, 9>), dim3(nblk), dim3(RS_THREADS), 0, c.stream,
                           (const u64 *)nullptr, (const u32 *)nullptr, elems[cur ^ 1], (u32 *)nullptr, n, shift, mask,
                           counts.get(), nblk, first, 0u, (const u8 *)nullptr, sides[cur ^ 1]);
        HIP_CHECK(hipGetLastError());
      } else {
        {
          KernelTimer kt(c, "leaf:k_rs_hist_u64", (double)n * 8);
          hipLaunchKernelGGL((k_rs_hist<u64, 0, false, 9>), dim3(nblk), dim3(RH_THREADS), 0, c.stream, elems[cur], n, shift, mask,
                             counts.get(), nblk, none);
          HIP_CHECK(hipGetLastError());
        }
        scan_exclusive_u32(c, counts.get(), counts.get(), (size_t)512 * nblk, nullptr);
        KernelTimer kt(c, "leaf:k_rs_scatter_u64_side", (double)n * 18);
        hipLaunchKernelGGL((k_rs_scatter<u64, false, 0, false, false, false, true, 9>), dim3(nblk), dim3(RS_THREADS), 0, c.stream,
                           elems[cur], (const u32 *)nullptr, elems[cur ^ 1], (u32 *)nullptr, n, shift, mask, counts.get(), nblk,
                           none, 0u, (const u8 *)sides[cur], sides[cur ^ 1]);
        HIP_CHECK(hipGetLastError());
      }
      cur ^= 1;
    }
  } else {
  DBuf<u32> counts(c, (size_t)256 * nblk);
  DBuf<u8> digits(c, n);
  TextKeys first = text;
  first.next_digits = digits.get();
  first.next_shift = 48;
  TextKeys count_keys = first;  // the histogram only needs the top bbits bits: 16 symbols keep its fast key roll
  if (count_keys.K0 * count_keys.w > 32) count_keys.K0 = 32 / count_keys.w;
  const int shifts[3] = {64 - bbits, 48, 56};
  for (int p = 0; p < 3; p++) {
    const int shift = shifts[p];
    const u32 mask = p == 0 ? (1u << (bbits - 16)) - 1u : 255u;
    if (p == 0) {
      {
        KernelTimer kt(c, "leaf:k_rs_hist_text", (double)n);
        hipLaunchKernelGGL((k_rs_hist<u64, 0, true>), dim3(nblk), dim3(RH_THREADS), 0, c.stream, (const u64 *)nullptr, n,
                           shift, mask, counts.get(), nblk, count_keys);
        HIP_CHECK(hipGetLastError());
      }
      scan_exclusive_u32(c, counts.get(), counts.get(), (size_t)256 * nblk, nullptr);
      KernelTimer kt(c, "leaf:k_rs_scatter_text", (double)n * 11);
      hipLaunchKernelGGL((k_rs_scatter<u64, false, 0, true, false, false, true>), dim3(nblk), dim3(RS_THREADS), 0, c.stream,
                         (const u64 *)nullptr, (const u32 *)nullptr, elems[cur ^ 1], (u32 *)nullptr, n, shift, mask,
                         counts.get(), nblk, first, 0u, (const u8 *)nullptr, sides[cur ^ 1]);
      HIP_CHECK(hipGetLastError());
    } else {
      if (p == 1) {
        KernelTimer kt(c, "leaf:k_rs_hist_bytes", (double)n);
        hipLaunchKernelGGL(k_rs_hist_bytes, dim3(nblk), dim3(RH_THREADS), 0, c.stream, (const u8 *)digits.get(), n,
                           counts.get(), nblk);
        HIP_CHECK(hipGetLastError());
      } else {
        KernelTimer kt(c, "leaf:k_rs_hist_u64", (double)n * 8);
        hipLaunchKernelGGL((k_rs_hist<u64, 0, false>), dim3(nblk), dim3(RH_THREADS), 0, c.stream, elems[cur], n, shift, mask,
                           counts.get(), nblk, none);
        HIP_CHECK(hipGetLastError());
      }
      scan_exclusive_u32(c, counts.get(), counts.get(), (size_t)256 * nblk, nullptr);
      KernelTimer kt(c, "leaf:k_rs_scatter_u64_side", (double)n * 18);
      hipLaunchKernelGGL((k_rs_scatter<u64, false, 0, false, false, false, true>), dim3(nblk), dim3(RS_THREADS), 0, c.stream,
                         elems[cur], (const u32 *)nullptr, elems[cur ^ 1], (u32 *)nullptr, n, shift, mask, counts.get(), nblk,
                         none, 0u, (const u8 *)sides[cur], sides[cur ^ 1]);
      HIP_CHECK(hipGetLastError());
    }
    cur ^= 1;
  }
  }
  if (!bucket_bounds(c, elems[cur], n, bs->start, &bs->max_bucket, bbits, cap)) {
    bs->start = DBuf<u32>();
    return -1;
  }
  bs->begin_bit = 64 - bbits - 16;
  bs->bbits = bbits;
  bs->pending = true;
  return cur;
}

// ----------------------------------------------------------------------------
// Lean elements (TextKeys::lean): the second pass runs over tiles that never cross the regions of equal first
// digit, so the first digit need not be stored and the bucket bounds fall out of the second pass's scanned
// tile counts.
// ----------------------------------------------------------------------------
// One workgroup.  offs1: the first pass's scanned counts, [digit][tile]: region d begins at offs1[d * nblk].
// Cuts every region into tiles of RS_TILE elements (the last one shorter): tfirst[d] = number of tiles in the
// regions below d (nd1 + 1 entries), tmap[t] = (first element, elements) of tile t; the tiles from tfirst[nd1] on
// are empty (the caller's nblk2 is an upper bound that leaves at least one of them).
// keep_lo, keep_hi: only the rows of these digits were scanned (a share of a leaf): the regions below are empty at 0,
// those from keep_hi on empty at n.
__global__ void __launch_bounds__(1024) k_lean_tiles(const u32 *__restrict__ offs1, u32 nblk, u32 nd1, u32 n,
                                                     u32 *__restrict__ tfirst, uint2 *__restrict__ tmap, u32 nblk2,
                                                     u32 *__restrict__ rstart = nullptr, u32 keep_lo = 0, u32 keep_hi = ~0u,
                                                     u32 *__restrict__ treg = nullptr) {
  __shared__ u32 rs[513], tf[513];
  __shared__ u32 scan_s[17];
  const u32 tid = threadIdx.x;
  if (tid < nd1) rs[tid] = tid < keep_lo ? 0u : (tid >= keep_hi ? n : offs1[(size_t)tid * nblk]);
  if (tid == nd1) rs[tid] = n;
  __syncthreads();
  const u32 nt = tid < nd1 ? (rs[tid + 1] - rs[tid] + RS_TILE - 1) / RS_TILE : 0u;
  u32 total;
  const u32 incl = block_scan_incl<OpSum>(nt, scan_s, &total);
  if (tid < nd1) tf[tid] = incl - nt;
  if (tid == nd1) tf[tid] = total;
  __syncthreads();
  if (tid <= nd1) tfirst[tid] = tf[tid];
  if (rstart && tid <= nd1) rstart[tid] = rs[tid];  // first element of every region
  for (u32 t = tid; t < nblk2; t += 1024) {
    if (t >= total) {
      tmap[t] = make_uint2(n, 0u);
      if (treg) treg[t] = 0;
      continue;
    }
    u32 lo = 0, hi = nd1;  // the last region d with tf[d] <= t (empty regions share their tf with the next one)
    while (hi - lo > 1) {
      const u32 mid = (lo + hi) / 2;
      if (tf[mid] <= t) lo = mid;
      else hi = mid;
    }
    const u32 t0 = rs[lo] + (t - tf[lo]) * RS_TILE;
    const u32 left = rs[lo + 1] - t0;
    tmap[t] = make_uint2(t0, left < (u32)RS_TILE ? left : (u32)RS_TILE);
    if (treg) treg[t] = lo;  // the region the tile lies in
  }
}

// The same after a first pass with slots instead of offsets: region d lies in `sub` areas of slot1 elements each, area
// a = d * sub + j from a * slot1 on with fill1[a] of them in use (a workgroup of the first pass wrote into the areas
// j = its number mod sub: see k_rs_scatter).  Tiles never cross an area; treg[t] = the region.
constexpr u32 LT_AREAS = 8192;
// tfa[a] = tiles in the areas below a (na + 1 entries), tfirst[d] = tiles in the regions below d, and (rstart not null)
// rstart[d] = elements of the regions below d: where region d begins once the elements lie densely.  One workgroup.
__global__ void __launch_bounds__(1024) k_lean_areas_scan(const u32 *__restrict__ fill1, u32 nd1, u32 sub, u32 *__restrict__ tfa,
                                                          u32 *__restrict__ tfirst, u32 *__restrict__ rstart) {
  __shared__ u32 tf[LT_AREAS + 1];
  __shared__ u32 scan_s[17];
  const u32 tid = threadIdx.x, na = nd1 * sub, per = (na + 1023) / 1024;
  u32 mine = 0;
  for (u32 k = 0; k < per; k++) {
    const u32 a = tid * per + k;
    if (a < na) {
      tf[a] = mine;
      mine += (fill1[a] + RS_TILE - 1) / RS_TILE;
    }
  }
  u32 total;
  const u32 base = block_scan_incl<OpSum>(mine, scan_s, &total) - mine;
  for (u32 k = 0; k < per; k++) {
    const u32 a = tid * per + k;
    if (a < na) tf[a] += base;
  }
  if (tid == 0) tf[na] = total;
  __syncthreads();
  for (u32 a = tid; a <= na; a += 1024) tfa[a] = tf[a];
  for (u32 d = tid; d <= nd1; d += 1024) tfirst[d] = tf[d * sub];
  if (rstart) {
    __syncthreads();  // (scan_s is used again)
    u32 reg = 0;
    if (tid < nd1)
      for (u32 j = 0; j < sub; j++) reg += fill1[tid * sub + j];
    u32 all;
    const u32 incl = block_scan_incl<OpSum>(reg, scan_s, &all);
    if (tid < nd1) rstart[tid] = incl - reg;
    if (tid == nd1) rstart[tid] = all;
  }
}

// The tiles themselves: workgroup a < na cuts area a; the last one marks the tiles behind the last area as empty.
__global__ void __launch_bounds__(64) k_lean_areas_map(const u32 *__restrict__ fill1, const u32 *__restrict__ tfa, u32 na, u32 sub,
                                                       u32 slot1, uint2 *__restrict__ tmap, u32 *__restrict__ treg, u32 nblk2) {
  const u32 a = blockIdx.x;
  if (a == na) {
    for (u32 t = tfa[na] + threadIdx.x; t < nblk2; t += 64) {
      tmap[t] = make_uint2(0u, 0u);
      treg[t] = 0;
    }
    return;
  }
  const u32 t0 = tfa[a], t1 = tfa[a + 1], len = fill1[a];
  for (u32 t = t0 + threadIdx.x; t < t1; t += 64) {
    const u32 done = (t - t0) * RS_TILE, left = len - done;
    tmap[t] = make_uint2(a * slot1 + done, left < (u32)RS_TILE ? left : (u32)RS_TILE);
    treg[t] = a / sub;
  }
}

static void lean_tiles_areas(Ctx &c, const u32 *fill1, u32 nd1, u32 sub, u32 slot1, u32 *tfirst, uint2 *tmap, u32 *treg, u32 nblk2,
                             u32 *rstart) {
  const u32 na = nd1 * sub;
  DBuf<u32> tfa(c, (size_t)na + 1);
  hipLaunchKernelGGL(k_lean_areas_scan, dim3(1), dim3(1024), 0, c.stream, fill1, nd1, sub, tfa.get(), tfirst, rstart);
  hipLaunchKernelGGL(k_lean_areas_map, dim3(na + 1), dim3(64), 0, c.stream, fill1, (const u32 *)tfa.get(), na, sub, slot1, tmap, treg, nblk2);
  HIP_CHECK(hipGetLastError());
}

// start[d2 << b1 | d1] = first element of that bucket after the second pass: the scanned count of digit d2 at the
// first tile of region d1 (the pass is stable and its tiles follow the regions in order); start[n_buckets] = n
__global__ void k_lean_bucket_starts(const u32 *__restrict__ offs2, u32 nblk2, const u32 *__restrict__ tfirst, int b1,
                                     u32 n_buckets, u32 n, u32 *__restrict__ start) {
  const u32 k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k > n_buckets) return;
  start[k] = k == n_buckets ? n : offs2[(size_t)(k >> b1) * nblk2 + tfirst[k & ((1u << b1) - 1u)]];
}

// The leaf sort's second pass writes REGION by region (the first pass sorted on the key's top digit, so the final order
// is region, then second digit): the offsets of its tile counts, [digit][tile], are therefore scanned per region -- all
// tiles of the region's digit 0, then digit 1, ... -- from the region's first element on, in place.  One workgroup per
// region; every wave sums, then scans, the digits it owns, so that no barrier separates the digits.
constexpr int LR_THREADS = 256;
constexpr u32 LR_DIGITS = 32;  // digits per workgroup (eight per wave)

// tot[r * ndig + d] = elements of digit d in region r (a wave per digit)
__global__ void __launch_bounds__(LR_THREADS) k_lean_region_sums(const u32 *__restrict__ counts, u32 nblk2, const u32 *__restrict__ tfirst,
                                                                 u32 ndig, u32 *__restrict__ tot) {
  const u32 chunks = ndig / LR_DIGITS, r = blockIdx.x / chunks, d0 = (blockIdx.x % chunks) * LR_DIGITS;
  const u32 lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const u32 t0 = tfirst[r], t1 = tfirst[r + 1];
  for (u32 d = d0 + w; d < d0 + LR_DIGITS; d += LR_THREADS / 64) {
    u32 sum = 0;
    for (u32 t = t0 + lane; t < t1; t += 64) sum += counts[(size_t)d * nblk2 + t];
    sum = wave_reduce_sum(sum);
    if (lane == 0) tot[(size_t)r * ndig + d] = sum;
  }
}

// the scan itself: a workgroup takes LR_DIGITS digits of a region, starting where the digits below them end
__global__ void __launch_bounds__(LR_THREADS) k_lean_region_scan(u32 *__restrict__ counts, u32 nblk2, const u32 *__restrict__ tfirst,
                                                                 const u32 *__restrict__ rstart, u32 ndig, const u32 *__restrict__ tot) {
  __shared__ u32 base_s[LR_DIGITS];
  __shared__ u32 scan_s[17];
  const u32 chunks = ndig / LR_DIGITS, r = blockIdx.x / chunks, d0 = (blockIdx.x % chunks) * LR_DIGITS;
  const u32 tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const u32 t0 = tfirst[r], t1 = tfirst[r + 1];
  if (t0 == t1) return;
  // elements of the region's digits below d0, then the exclusive sums of this workgroup's own digits
  u32 below = 0;
  for (u32 d = tid; d < d0; d += LR_THREADS) below += tot[(size_t)r * ndig + d];
  u32 all_below;
  block_scan_incl<OpSum>(below, scan_s, &all_below);
  if (tid < 64) {
    const u32 v = tid < LR_DIGITS ? tot[(size_t)r * ndig + d0 + tid] : 0u;
    const u32 in = wave_scan_incl_sum(v);
    if (tid < LR_DIGITS) base_s[tid] = rstart[r] + all_below + in - v;
  }
  __syncthreads();
  for (u32 d = d0 + w; d < d0 + LR_DIGITS; d += LR_THREADS / 64) {
    u32 run = base_s[d - d0];
    for (u32 tb = t0; tb < t1; tb += 64) {
      const u32 t = tb + lane;
      const u32 v = t < t1 ? counts[(size_t)d * nblk2 + t] : 0u;
      const u32 in = wave_scan_incl_sum(v);
      if (t < t1) counts[(size_t)d * nblk2 + t] = run + in - v;
      run += __shfl(in, 63, 64);
    }
  }
}

// start[d1 << b2 | d2] = first element of that bucket after the second pass of the leaf sort, plus `base`: the scanned
// count of digit d2 at the first tile of region d1; an empty region's buckets all begin where the region does.
// start[n_buckets] = base + n.
__global__ void k_lean_bucket_starts_msd(const u32 *__restrict__ offs2, u32 nblk2, const u32 *__restrict__ tfirst,
                                         const u32 *__restrict__ rstart, int b2, u32 n_buckets, u32 n, u32 base, u32 *__restrict__ start) {
  const u32 k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k > n_buckets) return;
  if (k == n_buckets) {
    start[k] = base + n;
    return;
  }
  const u32 r = k >> b2, d = k & ((1u << b2) - 1u);
  start[k] = base + (tfirst[r] == tfirst[r + 1] ? rstart[r] : offs2[(size_t)d * nblk2 + tfirst[r]]);
}

// Neither pass needs a stable order: a bucket is ordered in LDS afterwards, and when that gives up nothing of this
// sort is used.  So both rank their tiles with one LDS atomic per element (ANY_ORDER).
// per_digit[d] = sum over the tiles of counts[d][.] (one wave per digit)
__global__ void __launch_bounds__(256) k_row_sums(const u32 *__restrict__ counts, u32 nblk, u32 ndig, u32 *__restrict__ per_digit) {
  const u32 d = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (d >= ndig) return;
  u32 sum = 0;
  for (u32 t = lane; t < nblk; t += 64) sum += counts[(size_t)d * nblk + t];
  sum = wave_reduce_sum(sum);
  if (lane == 0) per_digit[d] = sum;
}

// The text pass in two steps, so that a caller can look at the digits' totals in between (LeanShard).
template <int DB>
static void lean_text_hist(Ctx &c, size_t n, u32 nblk, const TextKeys &tk, const TextKeys &count_keys, u32 *counts) {
  const int shift = 64 - tk.b1;
  const u32 mask = (1u << tk.b1) - 1u;
  KernelTimer kt(c, "leaf:k_rs_hist_text", (double)n);
  if (tk.run_span)
    hipLaunchKernelGGL((k_rs_hist<u64, 0, true, DB, true>), dim3(nblk), dim3(RH_THREADS), 0, c.stream, (const u64 *)nullptr, n, shift,
                       mask, counts, nblk, count_keys, (const uint2 *)nullptr);
  else
    hipLaunchKernelGGL((k_rs_hist<u64, 0, true, DB>), dim3(nblk), dim3(RH_THREADS), 0, c.stream, (const u64 *)nullptr, n, shift,
                       mask, counts, nblk, count_keys, (const uint2 *)nullptr);
  HIP_CHECK(hipGetLastError());
}

// fill1 (may be null): no offsets (counts is not read): digit d's elements go to the `sub` areas of slot1 elements each
// from (d * sub) * slot1 on, fill1[d * sub + j] counts what area j holds and *over is raised when one outgrows slot1.
template <int DB>
static void lean_text_scatter(Ctx &c, u64 *out, size_t n, u32 nblk, const TextKeys &tk, u32 *counts, u32 *fill1 = nullptr, u32 slot1 = 0,
                              u32 sub = 1, u32 *over = nullptr) {
  const u32 mask = (1u << tk.b1) - 1u;
  KernelTimer kt(c, "leaf:k_rs_scatter_text", (double)n * 9);
  if (fill1) {
    hipLaunchKernelGGL((k_rs_scatter<u64, false, 0, true, false, true, false, DB>), dim3(nblk), dim3(RS_THREADS), 0, c.stream,
                       (const u64 *)nullptr, (const u32 *)nullptr, out, (u32 *)nullptr, n, 13, mask, (const u32 *)nullptr, nblk, tk, 0u,
                       (const u8 *)nullptr, (u8 *)nullptr, (const uint2 *)nullptr, 0u, fill1, slot1, (const u32 *)nullptr, over, sub);
    HIP_CHECK(hipGetLastError());
    return;
  }
  // the staged elements carry the digit above the low 13 bits of their position field
  // (only the any-order ranking knows how to leave elements out)
  if (tk.run_span)
    hipLaunchKernelGGL((k_rs_scatter<u64, false, 0, true, false, true, false, DB, true>), dim3(nblk), dim3(RS_THREADS), 0, c.stream,
                       (const u64 *)nullptr, (const u32 *)nullptr, out, (u32 *)nullptr, n, 13, mask, counts, nblk, tk, 0u,
                       (const u8 *)nullptr, (u8 *)nullptr, (const uint2 *)nullptr);
  else if (c.lean_any_order || tk.keep_hi)
    hipLaunchKernelGGL((k_rs_scatter<u64, false, 0, true, false, true, false, DB>), dim3(nblk), dim3(RS_THREADS), 0, c.stream,
                       (const u64 *)nullptr, (const u32 *)nullptr, out, (u32 *)nullptr, n, 13, mask, counts, nblk, tk, 0u,
                       (const u8 *)nullptr, (u8 *)nullptr, (const uint2 *)nullptr);
  else
  hipLaunchKernelGGL((k_rs_scatter<u64, false, 0, true, false, false, false, DB>), dim3(nblk), dim3(RS_THREADS), 0, c.stream,
                     (const u64 *)nullptr, (const u32 *)nullptr, out, (u32 *)nullptr, n, 13, mask, counts, nblk, tk, 0u,
                     (const u8 *)nullptr, (u8 *)nullptr, (const uint2 *)nullptr);
  HIP_CHECK(hipGetLastError());
}

template <int DB>
static void lean_second_pass(Ctx &c, const u64 *in, u64 *out, size_t n, u32 nblk2, const TextKeys &tk, u32 *counts,
                             const uint2 *tmap, const u8 *digits, const u32 *tfirst, const u32 *rstart) {
  TextKeys none;
  memset(&none, 0, sizeof(none));
  const int shift = 64 - tk.b2;
  const u32 mask = (1u << tk.b2) - 1u;
  if (digits) {  // the text pass left this pass's digits as bytes on the side
    KernelTimer kt(c, "leaf:k_rs_hist_bytes", (double)n);
    hipLaunchKernelGGL(k_rs_hist_bytes, dim3(nblk2), dim3(RH_THREADS), 0, c.stream, digits, n, counts, nblk2, tmap);
    HIP_CHECK(hipGetLastError());
  } else {
    KernelTimer kt(c, "leaf:k_rs_hist_u64", (double)n * 8);
    hipLaunchKernelGGL((k_rs_hist<u64, 0, false, DB>), dim3(nblk2), dim3(RH_THREADS), 0, c.stream, in, n, shift, mask, counts,
                       nblk2, none, tmap);
    HIP_CHECK(hipGetLastError());
  }
  {
    KernelTimer kt(c, "leaf:k_lean_region_scan", (double)(1 << DB) * nblk2 * 12);
    constexpr u32 ndig = 1u << DB;
    static_assert(ndig % LR_DIGITS == 0 && LR_DIGITS <= 64, "whole chunks of digits; one wave scans a chunk's totals");
    const u32 regions = 1u << tk.b1, grid = regions * (ndig / LR_DIGITS);
    DBuf<u32> tot(c, (size_t)regions * ndig);
    hipLaunchKernelGGL(k_lean_region_sums, dim3(grid), dim3(LR_THREADS), 0, c.stream, (const u32 *)counts, nblk2, tfirst, ndig, tot.get());
    hipLaunchKernelGGL(k_lean_region_scan, dim3(grid), dim3(LR_THREADS), 0, c.stream, counts, nblk2, tfirst, rstart, ndig, (const u32 *)tot.get());
    HIP_CHECK(hipGetLastError());
  }
  KernelTimer kt(c, "leaf:k_rs_scatter_u64_keys", (double)n * 16);
  if (c.lean_any_order)
    hipLaunchKernelGGL((k_rs_scatter<u64, false, 0, false, false, true, false, DB>), dim3(nblk2), dim3(RS_THREADS), 0, c.stream, in,
                       (const u32 *)nullptr, out, (u32 *)nullptr, n, shift, mask, counts, nblk2, none, 0u, (const u8 *)nullptr,
                       (u8 *)nullptr, tmap);
  else
  hipLaunchKernelGGL((k_rs_scatter<u64, false, 0, false, false, false, false, DB>), dim3(nblk2), dim3(RS_THREADS), 0, c.stream, in,
                     (const u32 *)nullptr, out, (u32 *)nullptr, n, shift, mask, counts, nblk2, none, 0u, (const u8 *)nullptr,
                     (u8 *)nullptr, tmap);
  HIP_CHECK(hipGetLastError());
}

// The second pass with a fixed number of output elements per bucket (slot) instead of scanned offsets: no histogram,
// no scan; fill[region << DB | digit] counts what the bucket holds, *over is raised when one of them outgrows its slots.
template <int DB>
static void lean_second_pass_slots(Ctx &c, const u64 *in, u64 *out, size_t n, u32 nblk2, const TextKeys &tk, const uint2 *tmap,
                                   const u32 *treg, u32 *fill, u32 slot, u32 *over) {
  TextKeys none;
  memset(&none, 0, sizeof(none));
  const int shift = 64 - tk.b2;
  const u32 mask = (1u << tk.b2) - 1u;
  KernelTimer kt(c, "leaf:k_rs_scatter_u64_keys", (double)n * 16);
  hipLaunchKernelGGL((k_rs_scatter<u64, false, 0, false, false, true, false, DB>), dim3(nblk2), dim3(RS_THREADS), 0, c.stream, in,
                     (const u32 *)nullptr, out, (u32 *)nullptr, n, shift, mask, (const u32 *)nullptr, nblk2, none, 0u, (const u8 *)nullptr,
                     (u8 *)nullptr, tmap, 0u, fill, slot, treg, over);
  HIP_CHECK(hipGetLastError());
}

// The gap histogram's sort with the same device (gap.hip): 32-bit values grouped by their bits [low, low + b1 + b2).
// The first pass sorts on the lower b1 of those bits, the second on the upper b2 over tiles that follow the first
// pass's regions; neither needs a stable order (the buckets are only counted), and the second writes just the low
// `low` bits of a value, as 16-bit words: bucket (d2, d1) = the words [start[d2 << b1 | d1], start[.. + 1]) of the
// result, which is in keys[0] (keys[1] is scratch).  start: 2^(b1+b2) + 1 entries.
static u32 read_flag(Ctx &c, const u32 *d) {
  HIP_CHECK(hipMemcpyAsync(c.h_scalars, d, sizeof(u32), hipMemcpyDeviceToHost, c.stream));
  c.sync();
  return c.h_scalars[0];
}

// The same with slots instead of offsets (see leaf_sort_text_lean): neither pass has a histogram.  The first pass
// writes region d1 into the slot1 words from d1 * slot1 on of a buffer of its own, the second bucket k = d2 << b1 | d1
// into the slot2 16-bit words from k * slot2 on of keys[0]; start[k] = values in the buckets below k.  Gap indices of
// blocks that resemble each other spread evenly; where they do not, some cell outgrows its slots and the pass is
// void: 0 then, with keys[0] as it was (the first pass), or 2 with the second pass done the other way (contiguous
// buckets); 1: done, with slots.
template <int DB1, int DB2>
static int gap_sort_slots_impl(Ctx &c, u32 *keys[2], size_t n, int low, int b1, int b2, u32 *start, u32 slot2, u32 used_buckets) {
  const u32 nblk = ceil_div(n, RS_TILE);
  const u32 nd1 = 1u << b1, n_buckets = 1u << (b1 + b2);
  // a first digit's share: the buckets that have it (one more for some digits when the buckets in use are not a
  // multiple of the digits), and room for a few thousand filler values, which all fall into one bucket
  // (every digit's values spread over `sub` areas, one per XCD)
  const u32 sub = c.slot_sub, na = nd1 * sub;
  const double avg1 = (double)n / (double)used_buckets * (double)ceil_div(used_buckets, nd1) / (double)sub;
  const u32 slot1 = c.lean_slots == 4 ? (u32)avg1 + 1 : (u32)(((u64)(avg1 + 8.0 * sqrt(avg1) + 4096.0 + 64.0) + 63) & ~63ull);  // (4: tests)
  if ((u64)na * slot1 >= (1ull << 32) - (1ull << 20)) return 0;  // (32-bit indices)
  const u32 nblk2 = ceil_div(n, RS_TILE) + na + 1;
  DBuf<u32> mid(c, (size_t)na * slot1);
  DBuf<u32> fill1(c, na), fill2(c, (size_t)n_buckets + 1), tfirst(c, nd1 + 1), treg(c, nblk2);
  DBuf<uint2> tmap(c, nblk2);
  u32 *d_over = c.d_scalars + 37;
  TextKeys none;
  memset(&none, 0, sizeof(none));
  HIP_CHECK(hipMemsetAsync(fill1.get(), 0, na * sizeof(u32), c.stream));
  HIP_CHECK(hipMemsetAsync(fill2.get(), 0, ((size_t)n_buckets + 1) * sizeof(u32), c.stream));
  HIP_CHECK(hipMemsetAsync(d_over, 0, 2 * sizeof(u32), c.stream));
  {
    KernelTimer kt(c, "gap:k_rs_scatter_u32_slots", (double)n * 8);
    hipLaunchKernelGGL((k_rs_scatter<u32, false, 2, false, false, true, false, DB1>), dim3(nblk), dim3(RS_THREADS), 0, c.stream,
                       (const u32 *)keys[0], (const u32 *)nullptr, mid.get(), (u32 *)nullptr, n, low, nd1 - 1, (const u32 *)nullptr, nblk, none, 0u,
                       (const u8 *)nullptr, (u8 *)nullptr, (const uint2 *)nullptr, 0u, fill1.get(), slot1, (const u32 *)nullptr, d_over, sub);
    HIP_CHECK(hipGetLastError());
  }
  lean_tiles_areas(c, fill1.get(), nd1, sub, slot1, tfirst.get(), tmap.get(), treg.get(), nblk2, nullptr);
  if (read_flag(c, d_over) != 0) return 0;  // the first pass is void: nothing has been written over
  {
    KernelTimer kt(c, "gap:k_rs_scatter_u32_slots2", (double)n * 6);
    hipLaunchKernelGGL((k_rs_scatter<u32, false, 2, false, false, true, false, DB2>), dim3(nblk2), dim3(RS_THREADS), 0, c.stream,
                       (const u32 *)mid.get(), (const u32 *)nullptr, keys[0], (u32 *)nullptr, n, low + b1, (1u << b2) - 1u, (const u32 *)nullptr,
                       nblk2, none, 0u, (const u8 *)nullptr, (u8 *)nullptr, (const uint2 *)tmap.get(), (1u << low) - 1u, fill2.get(), slot2,
                       (const u32 *)treg.get(), d_over + 1, nd1);
    HIP_CHECK(hipGetLastError());
  }
  scan_exclusive_u32(c, fill2.get(), start, (size_t)n_buckets + 1, nullptr);
  if (read_flag(c, d_over + 1) != 0) {
    // the second pass again, with exact offsets, from where the first one left the values
    DBuf<u32> counts(c, (size_t)(1 << DB2) * nblk2);
    {
      KernelTimer kt(c, "gap:k_rs_hist_u32", (double)n * 4);
      hipLaunchKernelGGL((k_rs_hist<u32, 2, false, DB2>), dim3(nblk2), dim3(RH_THREADS), 0, c.stream, (const u32 *)mid.get(), n, low + b1,
                         (1u << b2) - 1u, counts.get(), nblk2, none, (const uint2 *)tmap.get());
      HIP_CHECK(hipGetLastError());
    }
    scan_exclusive_u32(c, counts.get(), counts.get(), (size_t)(1 << DB2) * nblk2, nullptr);
    {
      KernelTimer kt(c, "gap:k_rs_scatter_u32_keys", (double)n * 6);
      hipLaunchKernelGGL((k_rs_scatter<u32, false, 2, false, false, true, false, DB2>), dim3(nblk2), dim3(RS_THREADS), 0, c.stream,
                         (const u32 *)mid.get(), (const u32 *)nullptr, keys[0], (u32 *)nullptr, n, low + b1, (1u << b2) - 1u, counts.get(),
                         nblk2, none, 0u, (const u8 *)nullptr, (u8 *)nullptr, (const uint2 *)tmap.get(), (1u << low) - 1u);
      HIP_CHECK(hipGetLastError());
    }
    hipLaunchKernelGGL(k_lean_bucket_starts, dim3(n_buckets / 256 + 1), dim3(256), 0, c.stream, (const u32 *)counts.get(), nblk2,
                       (const u32 *)tfirst.get(), b1, n_buckets, (u32)n, start);
    HIP_CHECK(hipGetLastError());
    c.sync();  // (counts goes back to the pool)
    return 2;
  }
  return 1;
}

template <int DB1, int DB2>
static void gap_sort_lean_impl(Ctx &c, u32 *keys[2], size_t n, int low, int b1, int b2, u32 *start) {
  const u32 nblk = ceil_div(n, RS_TILE);
  const u32 nd1 = 1u << b1, n_buckets = 1u << (b1 + b2);
  const u32 nblk2 = nblk + nd1 + 1;
  DBuf<u32> counts(c, (size_t)512 * nblk2);
  DBuf<u32> tfirst(c, nd1 + 1);
  DBuf<uint2> tmap(c, nblk2);
  TextKeys none;
  memset(&none, 0, sizeof(none));
  {
    KernelTimer kt(c, "gap:k_rs_hist_u32", (double)n * 4);
    hipLaunchKernelGGL((k_rs_hist<u32, 2, false, DB1>), dim3(nblk), dim3(RH_THREADS), 0, c.stream, (const u32 *)keys[0], n, low,
                       nd1 - 1, counts.get(), nblk, none, (const uint2 *)nullptr);
    HIP_CHECK(hipGetLastError());
  }
  scan_exclusive_u32(c, counts.get(), counts.get(), (size_t)(1 << DB1) * nblk, nullptr);
  {
    KernelTimer kt(c, "gap:k_rs_scatter_u32_keys", (double)n * 8);
    hipLaunchKernelGGL((k_rs_scatter<u32, false, 2, false, false, true, false, DB1>), dim3(nblk), dim3(RS_THREADS), 0, c.stream,
                       (const u32 *)keys[0], (const u32 *)nullptr, keys[1], (u32 *)nullptr, n, low, nd1 - 1, counts.get(), nblk, none, 0u,
                       (const u8 *)nullptr, (u8 *)nullptr, (const uint2 *)nullptr, 0u);
    HIP_CHECK(hipGetLastError());
  }
  hipLaunchKernelGGL(k_lean_tiles, dim3(1), dim3(1024), 0, c.stream, (const u32 *)counts.get(), nblk, nd1, (u32)n, tfirst.get(),
                     tmap.get(), nblk2);
  HIP_CHECK(hipGetLastError());
  {
    KernelTimer kt(c, "gap:k_rs_hist_u32", (double)n * 4);
    hipLaunchKernelGGL((k_rs_hist<u32, 2, false, DB2>), dim3(nblk2), dim3(RH_THREADS), 0, c.stream, (const u32 *)keys[1], n, low + b1,
                       (1u << b2) - 1u, counts.get(), nblk2, none, (const uint2 *)tmap.get());
    HIP_CHECK(hipGetLastError());
  }
  scan_exclusive_u32(c, counts.get(), counts.get(), (size_t)(1 << DB2) * nblk2, nullptr);
  {
    KernelTimer kt(c, "gap:k_rs_scatter_u32_keys", (double)n * 6);
    hipLaunchKernelGGL((k_rs_scatter<u32, false, 2, false, false, true, false, DB2>), dim3(nblk2), dim3(RS_THREADS), 0, c.stream,
                       (const u32 *)keys[1], (const u32 *)nullptr, keys[0], (u32 *)nullptr, n, low + b1, (1u << b2) - 1u, counts.get(),
                       nblk2, none, 0u, (const u8 *)nullptr, (u8 *)nullptr, (const uint2 *)tmap.get(), (1u << low) - 1u);
    HIP_CHECK(hipGetLastError());
  }
  hipLaunchKernelGGL(k_lean_bucket_starts, dim3(n_buckets / 256 + 1), dim3(256), 0, c.stream, (const u32 *)counts.get(), nblk2,
                     (const u32 *)tfirst.get(), b1, n_buckets, (u32)n, start);
  HIP_CHECK(hipGetLastError());
}

void gap_sort_lean(Ctx &c, u32 *keys[2], size_t n, int low, int b1, int b2, u32 *start, u32 *slot_out, u32 used_buckets) {
  if (n == 0 || n >= (1ull << 32) || b1 < 1 || b1 > 9 || b2 < 1 || b2 > 9 || low < 1 || low > 16)
    fail(BWT_HIP_EINTERNAL, "gap sort: bad layout (n %zu, low %d, b1 %d, b2 %d)", n, low, b1, b2);
  if (slot_out) *slot_out = 0;
  if (slot_out && c.lean_slots && !c.arena_on) {  // (a call that may not allocate: the leaner of the two in memory)
    // slots where they cost little more memory than the values: the second pass's 16-bit words fit keys[0]
    const u32 n_buckets = 1u << (b1 + b2);
    if (!used_buckets || used_buckets > n_buckets) used_buckets = n_buckets;
    const double avg = (double)n / (double)used_buckets;
    u64 s = ((u64)(avg + 8.0 * sqrt(avg) + 64.0) + 63) & ~63ull;
    if (c.lean_slots == 2 || c.lean_slots == 5) s = avg < 1.0 ? 1 : (u64)avg;
    // (the kernels index both outputs with 32 bits)
    if (((u64)used_buckets + 1) * s * 2 <= (u64)n * 4 && s < 65536 && ((u64)used_buckets + 1) * s < (1ull << 32) - (1ull << 20)) {
      int done;
      if (b1 > 8 && b2 > 8) done = gap_sort_slots_impl<9, 9>(c, keys, n, low, b1, b2, start, (u32)s, used_buckets);
      else if (b2 > 8) done = gap_sort_slots_impl<8, 9>(c, keys, n, low, b1, b2, start, (u32)s, used_buckets);
      else done = gap_sort_slots_impl<8, 8>(c, keys, n, low, b1, b2, start, (u32)s, used_buckets);
      if (done) {
        *slot_out = done == 1 ? (u32)s : 0u;
        return;
      }
    }
  }
  if (b1 > 8 && b2 > 8) gap_sort_lean_impl<9, 9>(c, keys, n, low, b1, b2, start);
  else if (b2 > 8) gap_sort_lean_impl<8, 9>(c, keys, n, low, b1, b2, start);
  else gap_sort_lean_impl<8, 8>(c, keys, n, low, b1, b2, start);
}

int lean_count_symbols(const TextKeys &text) {
  const int bbits = text.b1 + text.b2;
  // the text pass's histogram only needs the top bbits bits: 16 symbols keep its fast key roll
  // (not with digits of another base: there every symbol of the key can reach the top bits)
  if (!text.base && text.K0 * text.w > 32 && 32 / text.w * text.w >= bbits) return 32 / text.w;
  return text.K0;
}

int leaf_sort_text_lean(Ctx &c, DBuf<u64> (&elems)[2], size_t n, u32 cap, const TextKeys &text, BucketState *bs, LeanShard *shard,
                        size_t *kept_out) {
  const int bbits = text.b1 + text.b2;
  if (!text.lean || text.b1 < 1 || text.b1 > 9 || text.b2 < 1 || text.b2 > 10 || text.pbits < 13 + text.b1 || text.pbits > 32 ||
      n >= (1ull << 32) - 8 || ((n - 1) >> text.pbits) || text.lean_deep < 16 || text.b1 + text.b2 + text.lean_deep > 48 ||
      64 - text.b2 - text.lean_deep < text.pbits + text.vbits ||
      (text.base ? (bbits + text.lean_deep > 56 || text.K0 > 56) : (text.K0 * text.w < bbits + text.lean_deep || text.K0 * text.w > 56)))
    fail(BWT_HIP_EINTERNAL, "leaf sort: bad lean layout (b1 %d, b2 %d, deep %d, pbits %d, vbits %d, K0 %d, w %d, n %zu)", text.b1,
         text.b2, text.lean_deep, text.pbits, text.vbits, text.K0, text.w, n);
  std::vector<u64> reg_est;
  u64 sampled = 0;
  if (!buckets_look_even(c, text, n, bbits, cap, &reg_est, text.b1, &sampled)) return -1;
  const u32 nblk = ceil_div(n, RS_TILE);
  const u32 nd1 = 1u << text.b1, n_buckets = 1u << bbits;
  TextKeys first = text;
  first.next_digits = nullptr;
  TextKeys count_keys = first;
  count_keys.K0 = lean_count_symbols(text);
  DBuf<u32> tfirst(c, nd1 + 1), rstart(c, nd1 + 1);
  const u32 dig1 = text.b1 > 8 ? 512u : 256u;  // counter rows of the text pass
  size_t kept = n;
  u64 first_rank = 0;
  u32 scan_lo = 0, scan_hi = dig1;
  // Slots instead of offsets, where the whole leaf is sorted here and nothing is left out.
  // The second pass without its histogram: every bucket owns `slot` elements of the output, a tile reserves its
  // place in a bucket with one atomic add per digit, and what the buckets hold falls out of the fill counts.  For the
  // even texts this route is made for a bucket stays within eight standard deviations of the mean; one that does
  // not voids the pass, and the pass with exact offsets runs instead.
  // The first pass the same way, with the regions' sizes estimated from the sample above (eight areas per region,
  // one per XCD: see k_rs_scatter); a region that outgrows the estimate voids it, and the counted pass runs.
  const bool plain = !(shard && shard->parts > 1) && !text.run_span;
  u32 slot = 0, slot1 = 0;
  const u32 sub = c.slot_sub, na = nd1 * sub;
  // (only where the keys of an even text fill the buckets they reach evenly: all of them, or the lower part that keys
  // as numbers in a base reach; bit-field codes with unused values fill them by a pattern of their own)
  const double reach = text.fill > 0 ? (double)text.fill : 1.0;
  if (plain && c.lean_slots && !c.arena_on && c.lean_any_order && (text.fill_even || reach > 0.97 || text.fill == 0)) {
    const double avg = (double)n / ((double)n_buckets * reach);
    u64 s = ((u64)(avg + 8.0 * sqrt(avg) + 64.0) + 63) & ~63ull;
    if (c.lean_slots == 2 || c.lean_slots == 5) s = avg < 1.0 ? 1 : (u64)avg;  // (tests: many buckets outgrow this)
    if (s <= cap && (u64)n_buckets * s < (1ull << 32) - (1ull << 20) &&
        (c.lean_slots > 1 || (double)n_buckets * (double)s * reach <= 1.25 * (double)n + 65536.0))
      slot = (u32)s;
    double est = 0, rel = 0;  // the largest region, and how far the sample may be off (six standard deviations)
    if (!reg_est.empty()) {
      u64 k = 1;
      for (u64 v : reg_est) k = v > k ? v : k;
      est = (double)k * ((double)n / (double)sampled);
      rel = 6.0 / sqrt((double)k);
      // A largest region well above the mean (symbols of unequal frequency) says the buckets are as uneven: their
      // slots would overflow and the second pass be run twice.  Offsets, then.
      if (c.lean_slots == 1 && est > (1.0 + rel + 0.03) * (double)n / ((double)nd1 * reach)) slot = 0;
    } else if (c.lean_slots > 2) {  // (tests, on texts too small for a sample)
      est = 2.0 * (double)n / (double)nd1 + 1024.0;
    }
    if (slot && est > 0) {
      const double area = est / (double)sub;
      u64 s1 = ((u64)(area * (1.0 + rel) + 8.0 * sqrt(area) + 64.0) + 63) & ~63ull;
      if (c.lean_slots == 4) s1 = (u64)(area / 8) + 1;  // (tests: too small)
      if ((u64)na * s1 < (1ull << 32) - (1ull << 20) && na <= LT_AREAS &&
          (c.lean_slots > 2 || (double)na * (double)s1 <= 1.15 * (double)n + 1048576.0))
        slot1 = (u32)s1;
    }
  }
  bool first_done = false;
  u32 nblk2 = 0;
  DBuf<uint2> tmap;
  DBuf<u32> treg;
  DBuf<u32> counts;
  u32 *d_over = c.d_scalars + 37;
  if (slot1) {
    elems[1] = DBuf<u64>(c, (size_t)na * slot1);
    DBuf<u32> fill1(c, na);
    HIP_CHECK(hipMemsetAsync(fill1.get(), 0, na * sizeof(u32), c.stream));
    HIP_CHECK(hipMemsetAsync(d_over, 0, sizeof(u32), c.stream));
    if (text.b1 > 8) lean_text_scatter<9>(c, elems[1].get(), n, nblk, first, nullptr, fill1.get(), slot1, sub, d_over);
    else lean_text_scatter<8>(c, elems[1].get(), n, nblk, first, nullptr, fill1.get(), slot1, sub, d_over);
    nblk2 = nblk + na + 1;
    tmap = DBuf<uint2>(c, nblk2);
    treg = DBuf<u32>(c, nblk2);
    lean_tiles_areas(c, fill1.get(), nd1, sub, slot1, tfirst.get(), tmap.get(), treg.get(), nblk2, rstart.get());
    first_done = read_flag(c, d_over) == 0;  // (fill1 goes back to the pool after this)
    if (first_done && shard) {
      shard->first_rank = 0;
      shard->count = n;
      shard->bounds.assign(2, 0);
      shard->bounds[1] = n;
    }
    if (!first_done) elems[1] = DBuf<u64>();
  }
  DBuf<u8> digits;
  if (!first_done) {
    counts = DBuf<u32>(c, (size_t)(text.b1 > 8 ? 512 : 256) * nblk);
    if (text.b1 > 8) lean_text_hist<9>(c, n, nblk, first, count_keys, counts.get());
    else lean_text_hist<8>(c, n, nblk, first, count_keys, counts.get());
    if (shard && shard->parts > 1) {
      // This GPU keeps a stretch of the first digit's values: the `part`-th of `parts` stretches that hold about the
      // same number of suffixes each.  Every GPU counts the whole text and cuts the same way; nothing is exchanged.
      DBuf<u32> per(c, dig1);
      hipLaunchKernelGGL(k_row_sums, dim3(ceil_div(dig1, 4)), dim3(256), 0, c.stream, (const u32 *)counts.get(), nblk, dig1, per.get());
      HIP_CHECK(hipGetLastError());
      std::vector<u32> h(dig1);
      HIP_CHECK(hipMemcpyAsync(h.data(), per.get(), dig1 * sizeof(u32), hipMemcpyDeviceToHost, c.stream));
      c.sync();
      std::vector<u64> cum(dig1 + 1, 0);
      for (u32 d = 0; d < dig1; d++) cum[d + 1] = cum[d] + h[d];
      auto cut = [&](int j) {  // first digit of part j: the first one at which j / parts of the suffixes lie below
        if (j <= 0) return 0u;
        if (j >= shard->parts) return nd1;
        const u64 want = (u64)n * (u64)j / (u64)shard->parts;
        u32 d = 0;
        while (d < nd1 && cum[d] < want) d++;
        return d;
      };
      shard->bounds.assign((size_t)shard->parts + 1, 0);
      for (int j = 0; j <= shard->parts; j++) shard->bounds[(size_t)j] = cum[cut(j)];
      const u32 lo = cut(shard->part), hi = cut(shard->part + 1);
      shard->dig_lo = lo;
      shard->dig_hi = hi;
      first_rank = cum[lo];
      kept = (size_t)(cum[hi] - cum[lo]);
      shard->first_rank = first_rank;
      shard->count = kept;
      // (an empty share -- both of its cuts fall inside one heavy digit, which the next share takes -- keeps a digit
      // value that no element has)
      first.keep_lo = hi > lo ? lo : dig1;
      first.keep_hi = hi > lo ? hi : dig1 + 1;
      scan_lo = first.keep_lo;  // only the rows of the digits kept are scanned, and read
      scan_hi = hi > lo ? hi : dig1;
    } else if (shard) {
      shard->first_rank = 0;
      shard->count = n;
      shard->bounds.assign(2, 0);
      shard->bounds[1] = n;
    }
    if (text.run_span) {
      // suffixes left out of the sort: what is kept is what the histogram counted
      if (shard && shard->parts > 1) fail(BWT_HIP_EINTERNAL, "leaf sort: a share of a leaf with suffixes left out");
      first.keep_lo = 0;
      first.keep_hi = dig1;
      u32 *d_kept = c.d_scalars + 12;
      scan_exclusive_u32(c, counts.get(), counts.get(), (size_t)dig1 * nblk, d_kept);
      HIP_CHECK(hipMemcpyAsync(c.h_scalars, d_kept, sizeof(u32), hipMemcpyDeviceToHost, c.stream));
      c.sync();
      kept = c.h_scalars[0];
      if (shard) {
        shard->count = kept;
        shard->bounds[1] = kept;
      }
    } else {
      u32 *rows = counts.get() + (size_t)scan_lo * nblk;
      scan_exclusive_u32(c, rows, rows, (size_t)(scan_hi - scan_lo) * nblk, nullptr);
    }
    const size_t room = kept > 0 ? kept : 1;
    elems[1] = DBuf<u64>(c, room);
    if (text.b2 <= 8 && !slot) {  // the second pass's digits as bytes on the side: its histogram reads a byte per element, not eight
      digits = DBuf<u8>(c, room + 16);
      first.next_digits = digits.get();
      first.next_shift = 64 - text.b2;
    }
    if (text.b1 > 8) lean_text_scatter<9>(c, elems[1].get(), n, nblk, first, counts.get());
    else lean_text_scatter<8>(c, elems[1].get(), n, nblk, first, counts.get());
    // the second pass's tiles: those of what was kept, cut at the regions' ends (at most one short tile per region)
    nblk2 = ceil_div(room, RS_TILE) + nd1 + 1;
    tmap = DBuf<uint2>(c, nblk2);
    if (slot) treg = DBuf<u32>(c, nblk2);
    hipLaunchKernelGGL(k_lean_tiles, dim3(1), dim3(1024), 0, c.stream, (const u32 *)counts.get(), nblk, nd1, (u32)kept, tfirst.get(),
                       tmap.get(), nblk2, rstart.get(), scan_lo, scan_hi, slot ? treg.get() : (u32 *)nullptr);
    HIP_CHECK(hipGetLastError());
  }
  if (kept_out) *kept_out = kept;
  const size_t room = kept > 0 ? kept : 1;
  u32 *d_max = c.d_scalars + 12;
  if (slot) {
    elems[0] = DBuf<u64>(c, (size_t)n_buckets * slot);
    DBuf<u32> fill(c, (size_t)n_buckets + 1);
    HIP_CHECK(hipMemsetAsync(fill.get(), 0, ((size_t)n_buckets + 1) * sizeof(u32), c.stream));
    HIP_CHECK(hipMemsetAsync(d_over, 0, sizeof(u32), c.stream));
    HIP_CHECK(hipMemsetAsync(d_max, 0, sizeof(u32), c.stream));
    if (text.b2 > 9) lean_second_pass_slots<10>(c, elems[1].get(), elems[0].get(), kept, nblk2, text, tmap.get(), treg.get(), fill.get(), slot, d_over);
    else if (text.b2 > 8) lean_second_pass_slots<9>(c, elems[1].get(), elems[0].get(), kept, nblk2, text, tmap.get(), treg.get(), fill.get(), slot, d_over);
    else lean_second_pass_slots<8>(c, elems[1].get(), elems[0].get(), kept, nblk2, text, tmap.get(), treg.get(), fill.get(), slot, d_over);
    // bucket k = region << b2 | digit is cell k of the fill counts; its first rank: the counts below it
    bs->start = DBuf<u32>(c, (size_t)n_buckets + 1);
    {
      KernelTimer kt(c, "leaf:k_bucket_bounds", (double)n_buckets * 12);
      scan_exclusive_u32(c, fill.get(), bs->start.get(), (size_t)n_buckets + 1, nullptr);
      hipLaunchKernelGGL(k_bucket_max, dim3(n_buckets / 256), dim3(256), 0, c.stream, bs->start.get(), d_max, n_buckets);
      HIP_CHECK(hipGetLastError());
    }
    HIP_CHECK(hipMemcpyAsync(c.h_scalars, d_max, sizeof(u32), hipMemcpyDeviceToHost, c.stream));
    HIP_CHECK(hipMemcpyAsync(c.h_scalars + 1, d_over, sizeof(u32), hipMemcpyDeviceToHost, c.stream));
    c.sync();
    if (!c.h_scalars[1]) {
      bs->max_bucket = c.h_scalars[0];
      if (bs->max_bucket > cap) {
        bs->start = DBuf<u32>();
        return -1;
      }
      bs->begin_bit = 0;
      bs->bbits = bbits;
      bs->slot = slot;
      bs->pending = true;
      return 0;
    }
    slot = 0;  // some bucket outgrew its slots: the same pass with exact offsets
  }
  elems[0] = DBuf<u64>(c, room);
  DBuf<u32> counts2(c, (size_t)(text.b2 > 9 ? 1024 : (text.b2 > 8 ? 512 : 256)) * nblk2);
  if (text.b2 > 9) lean_second_pass<10>(c, elems[1].get(), elems[0].get(), kept, nblk2, text, counts2.get(), tmap.get(), nullptr, tfirst.get(), rstart.get());
  else if (text.b2 > 8) lean_second_pass<9>(c, elems[1].get(), elems[0].get(), kept, nblk2, text, counts2.get(), tmap.get(), nullptr, tfirst.get(), rstart.get());
  else lean_second_pass<8>(c, elems[1].get(), elems[0].get(), kept, nblk2, text, counts2.get(), tmap.get(), digits.get(), tfirst.get(), rstart.get());
  counts = std::move(counts2);
  bs->start = DBuf<u32>(c, (size_t)n_buckets + 1);
  {
    KernelTimer kt(c, "leaf:k_bucket_bounds", (double)n_buckets * 12);
    HIP_CHECK(hipMemsetAsync(d_max, 0, sizeof(u32), c.stream));
    hipLaunchKernelGGL(k_lean_bucket_starts_msd, dim3(n_buckets / 256 + 1), dim3(256), 0, c.stream, (const u32 *)counts.get(), nblk2,
                       (const u32 *)tfirst.get(), (const u32 *)rstart.get(), text.b2, n_buckets, (u32)kept, (u32)first_rank, bs->start.get());
    hipLaunchKernelGGL(k_bucket_max, dim3(n_buckets / 256), dim3(256), 0, c.stream, bs->start.get(), d_max, n_buckets);
    HIP_CHECK(hipGetLastError());
  }
  HIP_CHECK(hipMemcpyAsync(c.h_scalars, d_max, sizeof(u32), hipMemcpyDeviceToHost, c.stream));
  c.sync();
  bs->max_bucket = c.h_scalars[0];
  if (bs->max_bucket > cap) {
    bs->start = DBuf<u32>();
    return -1;
  }
  bs->begin_bit = 0;
  bs->bbits = bbits;
  bs->slot = 0;
  bs->pending = true;
  return 0;
}

// sorts every bucket of e on bits begin_bit..47, in place
void bucket_sort_in_place(Ctx &c, u64 *e, size_t n, const BucketState &bs) {
  KernelTimer kt(c, "leaf:k_bucket_sort", (double)n * 16);
  hipLaunchKernelGGL(k_bucket_sort, dim3(BK_BUCKETS), dim3(BK_THREADS), 0, c.stream, e, bs.start.get(), bs.begin_bit);
  HIP_CHECK(hipGetLastError());
}

// Round 0 with packed elements (TextKeys::pbits != 0): the first pass builds the elements from the text
// (reverse-stable, see TextKeys), the others are plain stable passes over 8-byte keys without values.
// bs: if not null and the bucket route applies, the sort stops after the passes on the top 16 bits and
// hands the buckets over (bs->pending); the caller finishes them (bucket_sort_in_place, or a kernel of its
// own that consumes the sorted buckets straight from LDS).
int leaf_sort_text_packed(Ctx &c, u64 *elems[2], size_t n, int begin_bit, const TextKeys &text, BucketState *bs, bool try_buckets) {
  if (n == 0) return 0;
  if (begin_bit % 8 != 0 || begin_bit < 32 || begin_bit >= 64 || text.pbits <= 0 || text.pbits > 31 ||
      (n - 1) >> text.pbits)
    fail(BWT_HIP_EINTERNAL, "leaf sort: bad packed layout (begin_bit %d, pbits %d, n %zu)", begin_bit, text.pbits, n);
  if (try_buckets && begin_bit < 48 && n >= c.bucket_min_n && buckets_look_even(c, text, n)) {
    // top 16 bits by two passes (the first one reverse-stable from the text, as always), the rest per bucket
    const int cur = leaf_sort_text_packed(c, elems, n, 48, text, nullptr, false);
    BucketState local;
    BucketState &st = bs ? *bs : local;
    if (bucket_bounds(c, elems[cur], n, st.start, &st.max_bucket)) {
      st.bbits = 16;
      st.begin_bit = begin_bit;
      st.pending = true;
      if (!bs) bucket_sort_in_place(c, elems[cur], n, st);
      return cur;
    }
    st.start = DBuf<u32>();
  }
  u32 nblk = ceil_div(n, RS_TILE);
  DBuf<u32> counts(c, (size_t)256 * nblk);
  TextKeys none;
  memset(&none, 0, sizeof(none));
  // the text pass leaves the second pass's digits in a byte array of their own: its histogram then reads one
  // byte per element instead of the whole element
  DBuf<u8> digits;
  TextKeys first = text;
  first.next_digits = nullptr;
  if (begin_bit + 8 < 64) {
    digits = DBuf<u8>(c, n);
    first.next_digits = digits.get();
    first.next_shift = begin_bit + 8;
  }
  int cur = 0;
  for (int shift = begin_bit; shift < 64; shift += 8) {
    const u32 mask = 255u;
    if (shift == begin_bit) {
      {
        KernelTimer kt(c, "leaf:k_rs_hist_text", (double)n);
        hipLaunchKernelGGL((k_rs_hist<u64, 0, true>), dim3(nblk), dim3(RH_THREADS), 0, c.stream, (const u64 *)nullptr,
                           n, shift, mask, counts.get(), nblk, first);
        HIP_CHECK(hipGetLastError());
      }
      scan_exclusive_u32(c, counts.get(), counts.get(), (size_t)256 * nblk, nullptr);
      {
        KernelTimer kt(c, "leaf:k_rs_scatter_text", (double)n * 9);
        hipLaunchKernelGGL((k_rs_scatter<u64, false, 0, true, false>), dim3(nblk), dim3(RS_THREADS), 0, c.stream,
                           (const u64 *)nullptr, (const u32 *)nullptr, elems[cur ^ 1], (u32 *)nullptr, n, shift, mask,
                           counts.get(), nblk, first, 0u);
        HIP_CHECK(hipGetLastError());
      }
    } else {
      if (shift == begin_bit + 8 && digits.get()) {
        KernelTimer kt(c, "leaf:k_rs_hist_bytes", (double)n);
        hipLaunchKernelGGL(k_rs_hist_bytes, dim3(nblk), dim3(RH_THREADS), 0, c.stream, (const u8 *)digits.get(), n,
                           counts.get(), nblk);
        HIP_CHECK(hipGetLastError());
      } else {
        KernelTimer kt(c, "leaf:k_rs_hist_u64", (double)n * 8);
        hipLaunchKernelGGL((k_rs_hist<u64, 0, false>), dim3(nblk), dim3(RH_THREADS), 0, c.stream, elems[cur], n, shift,
                           mask, counts.get(), nblk, none);
        HIP_CHECK(hipGetLastError());
      }
      scan_exclusive_u32(c, counts.get(), counts.get(), (size_t)256 * nblk, nullptr);
      {
        KernelTimer kt(c, "leaf:k_rs_scatter_u64_keys", (double)n * 16);
        hipLaunchKernelGGL((k_rs_scatter<u64, false, 0, false, false>), dim3(nblk), dim3(RS_THREADS), 0, c.stream,
                           elems[cur], (const u32 *)nullptr, elems[cur ^ 1], (u32 *)nullptr, n, shift, mask,
                           counts.get(), nblk, none, 0u);
        HIP_CHECK(hipGetLastError());
      }
    }
    cur ^= 1;
  }
  return cur;
}

}  // namespace bwtamd
