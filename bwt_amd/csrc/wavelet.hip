// count_alphabets and the wavelet matrix of a block's L (reference:
// include/bwt.h:200-250, include/wavelet_matrix.h:198-241,
// include/succinct_bit_vector.h:193-250), rebuilt 4-ary with one 64-byte rank
// block per level and query (layout in wavelet.h).
#include "wavelet.h"

#include <string.h>

#include "internal.h"
#include "scan_tmpl.h"

namespace bwtamd {

// ----------------------------------------------------------------------------
// symbol histogram of L: per-workgroup LDS counters, per-workgroup partials in
// global memory, then a 256-thread column sum.  No global atomics.
// ----------------------------------------------------------------------------
constexpr int HIST_THREADS = 256;

__global__ void __launch_bounds__(HIST_THREADS)
    k_sym_hist(const u8 *__restrict__ L, u64 m, u32 *__restrict__ partial) {
  __shared__ u32 h[HIST_THREADS / 64][256];
  const u32 tid = threadIdx.x, w = tid >> 6;
  for (u32 i = tid; i < 256 * (HIST_THREADS / 64); i += HIST_THREADS) (&h[0][0])[i] = 0;
  __syncthreads();
  u64 n16 = m / 16;
  const uint4 *L16 = reinterpret_cast<const uint4 *>(L);
  for (u64 i = (u64)blockIdx.x * HIST_THREADS + tid; i < n16; i += (u64)gridDim.x * HIST_THREADS) {
    uint4 v = L16[i];
    u32 ws[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int k = 0; k < 4; k++) {
#pragma unroll
      for (int b = 0; b < 4; b++) atomicAdd(&h[w][(ws[k] >> (8 * b)) & 255], 1u);
    }
  }
  if (blockIdx.x == 0)
    for (u64 i = n16 * 16 + tid; i < m; i += HIST_THREADS) atomicAdd(&h[w][L[i]], 1u);
  __syncthreads();
  u32 tot = 0;
#pragma unroll
  for (int k = 0; k < HIST_THREADS / 64; k++) tot += h[k][tid];
  partial[(u64)blockIdx.x * 256 + tid] = tot;
}

// The same for a text with at most 7 distinct non-zero symbols (DNA): LDS counters would serialise on
// the few addresses, so every thread counts in registers instead.  A symbol's code (0 for the
// terminator, c - min_sym + 1 otherwise) selects one byte of a 64-bit accumulator, which is spilled
// into eight 32-bit counters before a byte can overflow.
__global__ void __launch_bounds__(HIST_THREADS)
    k_sym_hist_small(const u8 *__restrict__ L, u64 m, u32 min_sym, u32 *__restrict__ partial) {
  __shared__ u32 h[HIST_THREADS / 64][8];
  const u32 tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  u32 cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  u64 acc = 0;
  u32 pending = 0;
  auto add = [&](u32 c) {
    u32 code = c ? c - min_sym + 1 : 0u;
    acc += 1ull << (8 * code);
  };
  auto spill = [&]() {
#pragma unroll
    for (int k = 0; k < 8; k++) cnt[k] += (u32)(acc >> (8 * k)) & 255u;
    acc = 0;
    pending = 0;
  };
  const u64 n16 = m / 16;
  const uint4 *L16 = reinterpret_cast<const uint4 *>(L);
  for (u64 i = (u64)blockIdx.x * HIST_THREADS + tid; i < n16; i += (u64)gridDim.x * HIST_THREADS) {
    uint4 v = L16[i];
    u32 ws[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int k = 0; k < 4; k++) {
#pragma unroll
      for (int b = 0; b < 4; b++) add((ws[k] >> (8 * b)) & 255u);
    }
    pending += 16;
    if (pending > 255 - 16) spill();
  }
  spill();
  if (blockIdx.x == 0)
    for (u64 i = n16 * 16 + tid; i < m; i += HIST_THREADS) {
      add(L[i]);
      spill();
    }
#pragma unroll
  for (int k = 0; k < 8; k++) cnt[k] = wave_reduce_sum(cnt[k]);
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < 8; k++) h[w][k] = cnt[k];
  }
  __syncthreads();
  u32 tot = 0;
  // column `tid` of the partial table is symbol `tid`: code 0 -> symbol 0, code k -> symbol min_sym + k - 1
  u32 code = tid == 0 ? 0u : (tid >= min_sym && tid - min_sym + 1 < 8 ? tid - min_sym + 1 : 8u);
  if (code < 8) {
#pragma unroll
    for (int k = 0; k < HIST_THREADS / 64; k++) tot += h[k][code];
  }
  partial[(u64)blockIdx.x * 256 + tid] = tot;
}

// column sums of the [workgroup][256] table: 1024 threads, four row groups, rows eight at a time
__global__ void __launch_bounds__(1024) k_hist_reduce(const u32 *__restrict__ partial, u32 nblk, u64 *__restrict__ out) {
  __shared__ u64 part[4][256];
  const u32 c = threadIdx.x & 255, grp = threadIdx.x >> 8;
  u64 s = 0;
  u32 b = grp;
  for (; b + 28 < nblk; b += 32) {
    u32 v[8];
#pragma unroll
    for (int k = 0; k < 8; k++) v[k] = partial[(u64)(b + 4 * k) * 256 + c];
#pragma unroll
    for (int k = 0; k < 8; k++) s += v[k];
  }
  for (; b < nblk; b += 4) s += partial[(u64)b * 256 + c];
  part[grp][c] = s;
  __syncthreads();
  if (grp == 0) out[c] = part[0][c] + part[1][c] + part[2][c] + part[3][c];
}

void count_alphabets(Ctx &c, Block &blk) {
  if (blk.counted) return;
  Phase ph(c, "count_alphabets", (double)blk.len());
  const u64 m = (u64)blk.len();
  u32 nblk = (u32)std::min<u64>((u64)c.n_cu * 8, (u64)ceil_div(m, 16 * HIST_THREADS));
  if (nblk == 0) nblk = 1;
  DBuf<u32> partial(c, (size_t)nblk * 256);
  DBuf<u64> tot(c, 256);
  const u32 min_sym = (u32)(c.min_sym > 0 ? c.min_sym : 1);
  if (c.max_sym >= (int)min_sym && c.max_sym - (int)min_sym + 1 < 8)
    hipLaunchKernelGGL(k_sym_hist_small, dim3(nblk), dim3(HIST_THREADS), 0, c.stream, blk.L.get(), m, min_sym,
                       partial.get());
  else
    hipLaunchKernelGGL(k_sym_hist, dim3(nblk), dim3(HIST_THREADS), 0, c.stream, blk.L.get(), m, partial.get());
  hipLaunchKernelGGL(k_hist_reduce, dim3(1), dim3(1024), 0, c.stream, partial.get(), nblk, tot.get());
  HIP_CHECK(hipGetLastError());
  u64 h[256];
  u8 hidden = 0;
  HIP_CHECK(hipMemcpyAsync(h, tot.get(), sizeof(h), hipMemcpyDeviceToHost, c.stream));
  HIP_CHECK(hipMemcpyAsync(&hidden, blk.L.get() + blk.s_rank, 1, hipMemcpyDeviceToHost, c.stream));
  c.sync();
  for (int k = 0; k < 256; k++) blk.hist[k] = (i64)h[k];
  // L holds T[a-1] (not in the block) and lacks T[b-1]: include/bwt.h:215-217
  i64 cnt[256];
  for (int k = 0; k < 256; k++) cnt[k] = blk.hist[k];
  cnt[hidden]--;
  cnt[blk.e_char]++;
  i64 acc = 0;
  for (int k = 0; k < 256; k++) {
    blk.C[k] = acc;
    acc += cnt[k];
  }
  blk.C[256] = acc;
  blk.counted = true;
}

// ----------------------------------------------------------------------------
// one level of the matrix
// ----------------------------------------------------------------------------
constexpr int WM_THREADS = 256;
constexpr int WM_ROUNDS = 12;                                 // 4 waves x 12 rounds = 48 chunks
constexpr int WM_TILE_WORDS = (WM_THREADS / 64) * WM_ROUNDS;  // 48 = 16 rank blocks x 3 chunks
constexpr int WM_TILE_BLOCKS = WM_TILE_WORDS / WM_CHUNKS;
constexpr int WM_TILE = WM_TILE_WORDS * 64;                   // 3072 positions
static_assert(WM_TILE_WORDS % WM_CHUNKS == 0, "a tile holds whole rank blocks");

// The code of position idx on this level.  Level 0 reads L itself (raw symbols;
// the foreign symbol at `hidden` and anything below min_sym become code 0),
// deeper levels read the partitioned code arrays.
__device__ __forceinline__ u32 wm_code(const u8 *__restrict__ cur, u32 idx, u32 m, u32 hidden, u32 min_sym,
                                      bool raw) {
  if (idx >= m) return 0;
  u32 ch = cur[idx];
  if (!raw) return ch;
  return (idx == hidden || ch < min_sym) ? 0u : ch - min_sym;
}

// Both bit planes of one tile by wave ballots, and the tile's digit counts.
__global__ void __launch_bounds__(WM_THREADS)
    k_wm_plane(const u8 *__restrict__ cur, u32 m, int shift, u32 hidden, u32 min_sym, int raw,
               u64 *__restrict__ plane, u32 *__restrict__ tile_cnt, u32 tiles) {
  __shared__ u32 wc[WM_THREADS / 64][4];
  __shared__ __align__(16) u8 stage[WM_TILE];  // the tile's symbols: 16-byte global loads, byte reads from LDS
  const u32 tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const u32 tile0 = blockIdx.x * WM_TILE;
  static_assert(WM_TILE % 16 == 0 && WM_TILE / 16 <= WM_THREADS, "one 16-byte load per thread covers the tile");
  if (tid < WM_TILE / 16) {
    const u32 g0 = tile0 + tid * 16;  // tile0 is a multiple of 16 and `cur` is 256-byte aligned
    if (g0 + 16 <= m) {
      reinterpret_cast<uint4 *>(stage)[tid] = *reinterpret_cast<const uint4 *>(cur + g0);
    } else {
      for (u32 k = 0; k < 16; k++) stage[tid * 16 + k] = g0 + k < m ? cur[g0 + k] : (u8)0;
    }
  }
  __syncthreads();
  u32 c0 = 0, c1 = 0, c2 = 0, c3 = 0;
#pragma unroll
  for (int r = 0; r < WM_ROUNDS; r++) {
    u32 wi = r * (WM_THREADS / 64) + w;
    u32 idx = tile0 + wi * 64 + lane;
    u32 ch = stage[wi * 64 + lane];  // zero beyond m
    u32 code = (!raw) ? ch : ((idx == hidden || ch < min_sym) ? 0u : ch - min_sym);
    u32 d = (idx < m ? code >> shift : 0u) & 3;
    u64 b0 = __ballot(d & 1), b1 = __ballot(d >> 1), v = __ballot(idx < m);
    c0 += __popcll(~b1 & ~b0 & v);
    c1 += __popcll(~b1 & b0 & v);
    c2 += __popcll(b1 & ~b0 & v);
    c3 += __popcll(b1 & b0 & v);
    if (lane == 0) {
      ulonglong2 *dst = reinterpret_cast<ulonglong2 *>(plane + ((u64)blockIdx.x * WM_TILE_BLOCKS + wi / WM_CHUNKS) *
                                                                   WM_BLOCK_WORDS +
                                                       2 + 2 * (wi % WM_CHUNKS));
      *dst = make_ulonglong2(b0, b1);
    }
  }
  if (lane == 0) {
    wc[w][0] = c0;
    wc[w][1] = c1;
    wc[w][2] = c2;
    wc[w][3] = c3;
  }
  __syncthreads();
  if (tid < 4) tile_cnt[(u64)tid * tiles + blockIdx.x] = wc[0][tid] + wc[1][tid] + wc[2][tid] + wc[3][tid];
}

// Stable 4-way split of the tile's codes: digit d goes to its bucket, after
// everything of digit d in earlier tiles (tile_off is the scanned [digit][tile] table).
__global__ void __launch_bounds__(WM_THREADS)
    k_wm_split(const u8 *__restrict__ cur, u32 m, int shift, u32 hidden, u32 min_sym, int raw,
               const u32 *__restrict__ tile_off, u32 tiles, u8 *__restrict__ next) {
  __shared__ u32 wcnt[4][WM_TILE_WORDS];  // per word: count of each digit, then exclusive prefix over words
  const u32 tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const u32 tile0 = blockIdx.x * WM_TILE;
  u8 code[WM_ROUNDS];
  u64 lo[WM_ROUNDS], hi[WM_ROUNDS];
#pragma unroll
  for (int r = 0; r < WM_ROUNDS; r++) {
    u32 wi = r * (WM_THREADS / 64) + w;
    u32 idx = tile0 + wi * 64 + lane;
    u32 cd = wm_code(cur, idx, m, hidden, min_sym, raw);
    code[r] = (u8)cd;
    u32 d = (cd >> shift) & 3;
    u64 b0 = __ballot(d & 1), b1 = __ballot(d >> 1), v = __ballot(idx < m);
    lo[r] = b0;
    hi[r] = b1;
    if (lane < 4) {
      u64 sel = ((lane & 1) ? b0 : ~b0) & ((lane & 2) ? b1 : ~b1) & v;
      wcnt[lane][wi] = __popcll(sel);
    }
  }
  __syncthreads();
  if (tid < 4) {
    u32 run = tile_off[(u64)tid * tiles + blockIdx.x];
    for (int k = 0; k < WM_TILE_WORDS; k++) {
      u32 v = wcnt[tid][k];
      wcnt[tid][k] = run;
      run += v;
    }
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < WM_ROUNDS; r++) {
    u32 wi = r * (WM_THREADS / 64) + w;
    u32 idx = tile0 + wi * 64 + lane;
    if (idx < m) {
      u32 d = ((u32)code[r] >> shift) & 3;
      u64 sel = ((d & 1) ? lo[r] : ~lo[r]) & ((d & 2) ? hi[r] : ~hi[r]);
      next[wcnt[d][wi] + __popcll(sel & lanemask_lt())] = code[r];
    }
  }
}

// rank-block headers: exclusive scan of the blocks' digit counts
struct U4 {
  u32 v[4];
};
struct SumU4 {
  __device__ __forceinline__ static U4 identity() { return U4{{0, 0, 0, 0}}; }
  __device__ __forceinline__ static U4 apply(U4 a, U4 b) {
    return U4{{a.v[0] + b.v[0], a.v[1] + b.v[1], a.v[2] + b.v[2], a.v[3] + b.v[3]}};
  }
};
struct BlockDigitsIn {
  const u64 *plane;
  __device__ __forceinline__ U4 operator()(size_t b) const {
    const u64 *p = plane + b * WM_BLOCK_WORDS + 2;
    U4 s = SumU4::identity();
#pragma unroll
    for (int k = 0; k < (int)WM_CHUNKS; k++) {
      u64 lo = p[2 * k], hi = p[2 * k + 1];
      s.v[0] += __popcll(~hi & ~lo);
      s.v[1] += __popcll(~hi & lo);
      s.v[2] += __popcll(hi & ~lo);
      s.v[3] += __popcll(hi & lo);
    }
    return s;
  }
};
struct BlockHeaderOut {
  u64 *plane;
  __device__ __forceinline__ void operator()(size_t b, U4 ex, U4) const {
    u32 *h = reinterpret_cast<u32 *>(plane + b * WM_BLOCK_WORDS);
    h[0] = ex.v[0];
    h[1] = ex.v[1];
    h[2] = ex.v[2];
    h[3] = ex.v[3];
  }
};

// bucket starts of the next level = the scanned table's entries of tile 0
__global__ void k_wm_starts(const u32 *__restrict__ tile_off, u32 tiles, u32 *__restrict__ starts) {
  if (threadIdx.x < 4) starts[threadIdx.x] = tile_off[(u64)threadIdx.x * tiles];
}

// ----------------------------------------------------------------------------
// 16-ary levels (WM16_*): the same three steps per level with four bit planes
// ----------------------------------------------------------------------------
constexpr int WM16_TILE_BLOCKS = WM_TILE / WM16_BLOCK_POS;  // 24 rank blocks of 128 positions
static_assert(WM_TILE % WM16_BLOCK_POS == 0, "a tile holds whole rank blocks");

// this lane's digit among the wave's 64: the lanes that hold digit `d`, from the four ballots
__device__ __forceinline__ u64 wm16_select(const u64 (&bal)[4], u32 d) {
  u64 sel = ~0ull;
#pragma unroll
  for (int b = 0; b < 4; b++) sel &= ((d >> b) & 1) ? bal[b] : ~bal[b];
  return sel;
}

// the four bit planes of one tile by wave ballots, and the tile's 16 digit counts
__global__ void __launch_bounds__(WM_THREADS)
    k_wm_plane16(const u8 *__restrict__ cur, u32 m, int shift, u32 hidden, u32 min_sym, int raw,
                 u64 *__restrict__ plane, u32 *__restrict__ tile_cnt, u32 tiles) {
  __shared__ u32 wc[WM_THREADS / 64][16];
  const u32 tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const u32 tile0 = blockIdx.x * WM_TILE;
  u32 mine = 0;  // lane d < 16: occurrences of digit d in this wave's rounds
#pragma unroll
  for (int r = 0; r < WM_ROUNDS; r++) {
    const u32 wi = r * (WM_THREADS / 64) + w;  // 64-position word of the tile
    const u32 idx = tile0 + wi * 64 + lane;
    const u32 d = (wm_code(cur, idx, m, hidden, min_sym, raw) >> shift) & 15;
    const u64 bal[4] = {__ballot(d & 1), __ballot(d & 2), __ballot(d & 4), __ballot(d & 8)};
    const u64 v = __ballot(idx < m);
    if (lane < 16) mine += __popcll(wm16_select(bal, lane) & v);
    if (lane < 4) {
      // plane `lane` of the block: words 8 + 2 * lane (positions 0..63) and 9 + 2 * lane (64..127)
      u64 *blk = plane + ((u64)blockIdx.x * WM16_TILE_BLOCKS + wi / 2) * WM16_BLOCK_WORDS;
      blk[8 + 2 * lane + (wi & 1)] = lane == 0 ? bal[0] : (lane == 1 ? bal[1] : (lane == 2 ? bal[2] : bal[3]));
    }
  }
  if (lane < 16) wc[w][lane] = mine;
  __syncthreads();
  if (tid < 16) tile_cnt[(u64)tid * tiles + blockIdx.x] = wc[0][tid] + wc[1][tid] + wc[2][tid] + wc[3][tid];
}

// rank-block headers of one tile: digit counts before each block = the scanned [digit][tile] entry plus the
// tile's earlier blocks
__global__ void __launch_bounds__(64)
    k_wm_headers16(u64 *__restrict__ plane, const u32 *__restrict__ tile_off, u32 tiles) {
  const u32 d = threadIdx.x;
  if (d >= 16) return;
  // the scanned table counts every smaller digit too: take the digit's own start out again
  u32 run = tile_off[(u64)d * tiles + blockIdx.x] - tile_off[(u64)d * tiles];
  for (int b = 0; b < WM16_TILE_BLOCKS; b++) {
    u64 *blk = plane + ((u64)blockIdx.x * WM16_TILE_BLOCKS + b) * WM16_BLOCK_WORDS;
    reinterpret_cast<u32 *>(blk)[d] = run;
    u64 m0 = ~0ull, m1 = ~0ull;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const u64 inv = ((d >> k) & 1) ? 0ull : ~0ull;
      m0 &= blk[8 + 2 * k] ^ inv;
      m1 &= blk[9 + 2 * k] ^ inv;
    }
    run += __popcll(m0) + __popcll(m1);
  }
}

// stable 16-way split of the tile's codes (k_wm_split with four planes)
__global__ void __launch_bounds__(WM_THREADS)
    k_wm_split16(const u8 *__restrict__ cur, u32 m, int shift, u32 hidden, u32 min_sym, int raw,
                 const u32 *__restrict__ tile_off, u32 tiles, u8 *__restrict__ next) {
  __shared__ u32 wcnt[16][WM_TILE_WORDS];  // per word: count of each digit, then exclusive prefix over words
  const u32 tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const u32 tile0 = blockIdx.x * WM_TILE;
  u8 code[WM_ROUNDS];
#pragma unroll
  for (int r = 0; r < WM_ROUNDS; r++) {
    const u32 wi = r * (WM_THREADS / 64) + w;
    const u32 idx = tile0 + wi * 64 + lane;
    const u32 cd = wm_code(cur, idx, m, hidden, min_sym, raw);
    code[r] = (u8)cd;
    const u32 d = (cd >> shift) & 15;
    const u64 bal[4] = {__ballot(d & 1), __ballot(d & 2), __ballot(d & 4), __ballot(d & 8)};
    const u64 v = __ballot(idx < m);
    if (lane < 16) wcnt[lane][wi] = __popcll(wm16_select(bal, lane) & v);
  }
  __syncthreads();
  if (tid < 16) {
    u32 run = tile_off[(u64)tid * tiles + blockIdx.x];
    for (int k = 0; k < WM_TILE_WORDS; k++) {
      const u32 v = wcnt[tid][k];
      wcnt[tid][k] = run;
      run += v;
    }
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < WM_ROUNDS; r++) {
    const u32 wi = r * (WM_THREADS / 64) + w;
    const u32 idx = tile0 + wi * 64 + lane;
    const u32 d = ((u32)code[r] >> shift) & 15;
    const u64 bal[4] = {__ballot(d & 1), __ballot(d & 2), __ballot(d & 4), __ballot(d & 8)};
    const u64 v = __ballot(idx < m);
    if (idx < m) next[wcnt[d][wi] + __popcll(wm16_select(bal, d) & v & lanemask_lt())] = code[r];
  }
}

__global__ void k_wm_starts16(const u32 *__restrict__ tile_off, u32 tiles, u32 *__restrict__ starts) {
  if (threadIdx.x < 16) starts[threadIdx.x] = tile_off[(u64)threadIdx.x * tiles];
}

// p_end[c]: where the walk of position 0 ends for symbol c; lf_base[c] = C[c] - p_end[c]
__global__ void k_wm_tables(WmView wm, u32 n_codes, const u32 *__restrict__ C, u32 *__restrict__ tab) {
  u32 c = threadIdx.x;
  u32 p = 0;
  if (c >= wm.min_sym && c - wm.min_sym < n_codes) p = wm_walk(wm, c - wm.min_sym, 0);
  tab[c] = C[c] - p;
  tab[256 + c] = p;
}

WmView make_view(const Block &blk) {
  const WaveletDev &wm = blk.wm;
  WmView v;
  memset(&v, 0, sizeof(v));
  v.planes = wm.planes.get();
  v.level_words = wm.n_blk * (wm.dbits == 4 ? WM16_BLOCK_WORDS : WM_BLOCK_WORDS);
  v.levels = wm.levels;
  v.dbits = wm.dbits;
  v.min_sym = wm.min_sym;
  v.max_pos = (u32)(wm.n_blk * (wm.dbits == 4 ? WM16_BLOCK_POS : WM_BLOCK_POS) - 1);
  v.hidden = (u32)blk.s_rank;
  for (int l = 0; l < WM_MAX_LEVELS; l++)
    for (int d = 0; d < 16; d++) v.start[l][d] = wm.start[l][d];
  return v;
}

void build_wavelet(Ctx &c, Block &blk) {
  if (blk.wm.built) return;
  count_alphabets(c, blk);
  const u64 m = (u64)blk.len();
  // code width: symbols min_sym..max_sym become 0..max_sym-min_sym; two bits per level
  const u32 min_sym = (u32)(c.min_sym > 0 ? c.min_sym : 1);
  int bits = 1;
  while (((c.max_sym > (int)min_sym ? c.max_sym - (int)min_sym : 0) >> bits) != 0) bits++;
  // up to four symbols: one 2-bit level of 64-byte blocks; larger alphabets: 4-bit levels of 128-byte blocks
  const int dbits = bits <= 2 ? 2 : 4;
  const int levels = (bits + dbits - 1) / dbits;
  const int fan = 1 << dbits;
  Phase ph(c, "wavelet_build", (double)m * levels * (dbits == 4 ? 4.0 : 3.3));
  const u32 tiles = (u32)(m / WM_TILE) + 1;
  const u64 n_blk = (u64)tiles * (dbits == 4 ? WM16_TILE_BLOCKS : WM_TILE_BLOCKS);
  const u64 blk_words = dbits == 4 ? WM16_BLOCK_WORDS : WM_BLOCK_WORDS;
  WaveletDev &wm = blk.wm;
  wm.levels = levels;
  wm.dbits = dbits;
  wm.min_sym = min_sym;
  wm.n_blk = n_blk;
  wm.planes = DBuf<u64>(c, (size_t)levels * n_blk * blk_words);
  DBuf<u8> t0, t1;
  if (levels > 1) {
    t0 = DBuf<u8>(c, m);
    t1 = DBuf<u8>(c, m);
  }
  DBuf<u32> tz(c, (size_t)fan * tiles);
  DBuf<u32> starts(c, 16 * WM_MAX_LEVELS);
  u32 *d_starts = starts.get();  // `fan` words per level
  const u8 *cur = blk.L.get();
  u8 *bufs[2] = {t0.get(), t1.get()};
  const u32 hidden = (u32)blk.s_rank;
  for (int l = 0; l < levels; l++) {
    const int shift = dbits * (levels - 1 - l);
    const int raw = l == 0;
    u64 *plane = wm.planes.get() + (u64)l * n_blk * blk_words;
    if (dbits == 4) {
      hipLaunchKernelGGL(k_wm_plane16, dim3(tiles), dim3(WM_THREADS), 0, c.stream, cur, (u32)m, shift, hidden, min_sym,
                         raw, plane, tz.get(), tiles);
      HIP_CHECK(hipGetLastError());
      scan_exclusive_u32(c, tz.get(), tz.get(), (size_t)16 * tiles, nullptr);
      hipLaunchKernelGGL(k_wm_starts16, dim3(1), dim3(64), 0, c.stream, tz.get(), tiles, d_starts + 16 * l);
      hipLaunchKernelGGL(k_wm_headers16, dim3(tiles), dim3(64), 0, c.stream, plane, tz.get(), tiles);
      HIP_CHECK(hipGetLastError());
      if (l + 1 < levels) {
        u8 *nxt = bufs[l & 1];
        hipLaunchKernelGGL(k_wm_split16, dim3(tiles), dim3(WM_THREADS), 0, c.stream, cur, (u32)m, shift, hidden, min_sym,
                           raw, tz.get(), tiles, nxt);
        HIP_CHECK(hipGetLastError());
        cur = nxt;
      }
      continue;
    }
    hipLaunchKernelGGL(k_wm_plane, dim3(tiles), dim3(WM_THREADS), 0, c.stream, cur, (u32)m, shift, hidden, min_sym, raw,
                       plane, tz.get(), tiles);
    HIP_CHECK(hipGetLastError());
    scan_exclusive_u32(c, tz.get(), tz.get(), (size_t)4 * tiles, nullptr);
    hipLaunchKernelGGL(k_wm_starts, dim3(1), dim3(64), 0, c.stream, tz.get(), tiles, d_starts + 4 * l);
    scan_generic<U4, SumU4>(c, (size_t)n_blk, BlockDigitsIn{plane}, BlockHeaderOut{plane});
    if (l + 1 < levels) {
      u8 *nxt = bufs[l & 1];
      hipLaunchKernelGGL(k_wm_split, dim3(tiles), dim3(WM_THREADS), 0, c.stream, cur, (u32)m, shift, hidden, min_sym,
                         raw, tz.get(), tiles, nxt);
      HIP_CHECK(hipGetLastError());
      cur = nxt;
    }
  }
  HIP_CHECK(hipMemcpyAsync(c.h_scalars, d_starts, sizeof(u32) * fan * levels, hipMemcpyDeviceToHost, c.stream));
  c.sync();
  for (int l = 0; l < levels; l++)
    for (int d = 0; d < fan; d++) wm.start[l][d] = c.h_scalars[fan * l + d];
  // LF tables
  u32 hC[256];
  for (int k = 0; k < 256; k++) hC[k] = (u32)blk.C[k];
  DBuf<u32> dC(c, 256);
  HIP_CHECK(hipMemcpyAsync(dC.get(), hC, sizeof(hC), hipMemcpyHostToDevice, c.stream));
  wm.lf_base = DBuf<u32>(c, 512);
  hipLaunchKernelGGL(k_wm_tables, dim3(1), dim3(256), 0, c.stream, make_view(blk), 1u << (dbits * levels), dC.get(),
                     wm.lf_base.get());
  HIP_CHECK(hipGetLastError());
  c.sync();  // hC is on the stack
  wm.built = true;
}

// ----------------------------------------------------------------------------
// batched rank / LF queries (test taps and host-side helpers)
//   lf == false: out = wm.rank(c, i)                  include/wavelet_matrix.h:260-290
//   lf == true : out = LF_map_mod(c, i) relative to a include/bwt.h:253-259
// ----------------------------------------------------------------------------
__global__ void k_rank_batch(WmView wm, const u32 *__restrict__ tab, const u8 *__restrict__ sym,
                             const u32 *__restrict__ pos, u32 *__restrict__ out, u32 nq, int lf, u32 e_char,
                             u32 e_rank) {
  u32 k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= nq) return;
  u32 c = sym[k];
  u32 v = wm_lf(wm, tab, c, pos[k]);  // C[c] + occ(c, pos)
  if (!lf) {
    out[k] = v - (tab[c] + tab[256 + c]);  // minus C[c]
  } else {
    v += (v >= e_rank && c == e_char);
    out[k] = v;
  }
}

void rank_batch(Ctx &c, Block &blk, const u8 *d_sym, const u32 *d_pos, u32 *d_out, size_t nq, bool lf) {
  build_wavelet(c, blk);
  hipLaunchKernelGGL(k_rank_batch, dim3(ceil_div(nq, 256)), dim3(256), 0, c.stream, make_view(blk),
                     blk.wm.lf_base.get(), d_sym, d_pos, d_out, (u32)nq, lf ? 1 : 0, (u32)blk.e_char, (u32)blk.e_rank);
  HIP_CHECK(hipGetLastError());
}

}  // namespace bwtamd
