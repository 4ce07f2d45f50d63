// count_alphabets and the wavelet matrix of a block's L (reference:
// include/bwt.h:200-250, include/wavelet_matrix.h:198-241,
// include/succinct_bit_vector.h:193-250), rebuilt for one-transaction ranks.
#include "wavelet.h"

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

__global__ void k_hist_reduce(const u32 *__restrict__ partial, u32 nblk, u64 *__restrict__ out) {
  u32 c = threadIdx.x;
  u64 s = 0;
  for (u32 b = 0; b < nblk; b++) s += partial[(u64)b * 256 + c];
  out[c] = s;
}

void count_alphabets(Ctx &c, Block &blk) {
  if (blk.counted) return;
  Phase ph(c, "count_alphabets", (double)blk.len());
  const u64 m = (u64)blk.len();
  u32 nblk = (u32)std::min<u64>((u64)c.n_cu * 8, (u64)ceil_div(m, 16 * HIST_THREADS));
  if (nblk == 0) nblk = 1;
  DBuf<u32> partial(c, (size_t)nblk * 256);
  DBuf<u64> tot(c, 256);
  hipLaunchKernelGGL(k_sym_hist, dim3(nblk), dim3(HIST_THREADS), 0, c.stream, blk.L.get(), m, partial.get());
  hipLaunchKernelGGL(k_hist_reduce, dim3(1), dim3(256), 0, c.stream, partial.get(), nblk, tot.get());
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
constexpr int WM_ROUNDS = 14;                                   // 4 waves x 14 rounds = 56 plane words
constexpr int WM_TILE_WORDS = (WM_THREADS / 64) * WM_ROUNDS;    // 56 = 8 rank blocks x 7 words
constexpr int WM_TILE = WM_TILE_WORDS * 64;                     // 3584 symbols

// Plane words of one tile by wave ballots, and the tile's number of zeros.
// `hidden` is an index whose symbol reads as 0 (T[a-1], include/bwt.h:234-236).
__global__ void __launch_bounds__(WM_THREADS)
    k_wm_plane(const u8 *__restrict__ cur, u32 m, int shift, u32 hidden, u64 *__restrict__ plane,
               u32 *__restrict__ tile_zeros) {
  __shared__ u32 wz[WM_THREADS / 64];
  const u32 tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const u32 tile0 = blockIdx.x * WM_TILE;
  u32 zeros = 0;
#pragma unroll
  for (int r = 0; r < WM_ROUNDS; r++) {
    u32 wi = r * (WM_THREADS / 64) + w;
    u32 idx = tile0 + wi * 64 + lane;
    bool valid = idx < m;
    u32 ch = valid ? cur[idx] : 0u;
    if (idx == hidden) ch = 0;
    u64 word = __ballot((ch >> shift) & 1);
    u64 vmask = __ballot(valid);
    zeros += __popcll(vmask & ~word);
    if (lane == 0) plane[((u64)blockIdx.x * 8 + wi / 7) * WM_BLOCK_WORDS + 1 + wi % 7] = word;
  }
  if (lane == 0) wz[w] = zeros;
  __syncthreads();
  if (tid == 0) tile_zeros[blockIdx.x] = wz[0] + wz[1] + wz[2] + wz[3];
}

// Stable split of the tile's symbols: zeros to [zoff..), ones to [nz + ooff..).
__global__ void __launch_bounds__(WM_THREADS)
    k_wm_split(const u8 *__restrict__ cur, u32 m, int shift, u32 hidden, const u32 *__restrict__ tile_zoff,
               const u32 *__restrict__ d_nz, u8 *__restrict__ next) {
  __shared__ u32 wzero[WM_TILE_WORDS + 1];  // zeros in each word, then exclusive prefix
  __shared__ u32 wvalid[WM_TILE_WORDS + 1];
  const u32 tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const u32 tile0 = blockIdx.x * WM_TILE;
  u8 sym[WM_ROUNDS];
  u64 words[WM_ROUNDS];
#pragma unroll
  for (int r = 0; r < WM_ROUNDS; r++) {
    u32 wi = r * (WM_THREADS / 64) + w;
    u32 idx = tile0 + wi * 64 + lane;
    bool valid = idx < m;
    u32 ch = valid ? cur[idx] : 0u;
    if (idx == hidden) ch = 0;
    sym[r] = (u8)ch;
    u64 word = __ballot((ch >> shift) & 1);
    u64 vmask = __ballot(valid);
    words[r] = word;
    if (lane == 0) {
      wzero[wi] = __popcll(vmask & ~word);
      wvalid[wi] = __popcll(vmask);
    }
  }
  __syncthreads();
  if (tid == 0) {
    u32 az = 0, av = 0;
    for (int k = 0; k < WM_TILE_WORDS; k++) {
      u32 z = wzero[k], v = wvalid[k];
      wzero[k] = az;
      wvalid[k] = av;
      az += z;
      av += v;
    }
  }
  __syncthreads();
  const u32 nz = *d_nz;
  const u32 zoff = tile_zoff[blockIdx.x];
  const u32 ooff = tile0 - zoff;  // ones before this tile
#pragma unroll
  for (int r = 0; r < WM_ROUNDS; r++) {
    u32 wi = r * (WM_THREADS / 64) + w;
    u32 idx = tile0 + wi * 64 + lane;
    if (idx < m) {
      u64 word = words[r];
      u32 zb = wzero[wi], ob = wvalid[wi] - wzero[wi];
      bool bit = (word >> lane) & 1;
      u32 dst = bit ? nz + ooff + ob + __popcll(word & lanemask_lt())
                    : zoff + zb + __popcll(~word & lanemask_lt());
      next[dst] = sym[r];
    }
  }
}

// rank-block headers: exclusive scan of the blocks' popcounts
struct BlockOnesIn {
  const u64 *plane;
  __device__ __forceinline__ u32 operator()(size_t b) const {
    const u64 *p = plane + b * WM_BLOCK_WORDS;
    u32 s = 0;
#pragma unroll
    for (int k = 1; k < 8; k++) s += __popcll(p[k]);
    return s;
  }
};
struct BlockHeaderOut {
  u64 *plane;
  __device__ __forceinline__ void operator()(size_t b, u32 ex, u32) const { plane[b * WM_BLOCK_WORDS] = ex; }
};

// p_end[c]: where the walk of position 0 ends for symbol c; lf_base[c] = C[c] - p_end[c]
__global__ void k_wm_tables(WmView wm, const u32 *__restrict__ C, u32 *__restrict__ tab) {
  u32 c = threadIdx.x;
  u32 p = (c >> wm.levels) ? 0u : wm_walk(wm, c, 0);
  tab[c] = C[c] - p;
  tab[256 + c] = p;
}

void build_wavelet(Ctx &c, Block &blk) {
  if (blk.wm.built) return;
  count_alphabets(c, blk);
  const u64 m = (u64)blk.len();
  const int levels = c.sym_bits;
  Phase ph(c, "wavelet_build", (double)m * levels * 3.2);
  const u32 tiles = (u32)(m / WM_TILE) + 1;
  const u64 n_blk = (u64)tiles * 8;
  WaveletDev &wm = blk.wm;
  wm.levels = levels;
  wm.n_blk = n_blk;
  wm.planes = DBuf<u64>(c, (size_t)levels * n_blk * WM_BLOCK_WORDS);
  DBuf<u8> t0(c, m), t1(c, m);
  DBuf<u32> tz(c, tiles);
  u32 *d_nz = c.d_scalars + 24;  // one word per level
  const u8 *cur = blk.L.get();
  u8 *bufs[2] = {t0.get(), t1.get()};
  u32 hidden = (u32)blk.s_rank;
  for (int l = 0; l < levels; l++) {
    int shift = levels - 1 - l;
    u64 *plane = wm.planes.get() + (u64)l * n_blk * WM_BLOCK_WORDS;
    hipLaunchKernelGGL(k_wm_plane, dim3(tiles), dim3(WM_THREADS), 0, c.stream, cur, (u32)m, shift, hidden, plane,
                       tz.get());
    HIP_CHECK(hipGetLastError());
    scan_exclusive_u32(c, tz.get(), tz.get(), tiles, d_nz + l);
    scan_generic<u32, SumU32>(c, (size_t)n_blk, BlockOnesIn{plane}, BlockHeaderOut{plane});
    if (l + 1 < levels) {
      u8 *nxt = bufs[l & 1];
      hipLaunchKernelGGL(k_wm_split, dim3(tiles), dim3(WM_THREADS), 0, c.stream, cur, (u32)m, shift, hidden,
                         tz.get(), d_nz + l, nxt);
      HIP_CHECK(hipGetLastError());
      cur = nxt;
      hidden = 0xFFFFFFFFu;  // the copy already holds the hidden symbol as 0
    }
  }
  HIP_CHECK(hipMemcpyAsync(c.h_scalars, d_nz, sizeof(u32) * levels, hipMemcpyDeviceToHost, c.stream));
  c.sync();
  for (int l = 0; l < levels; l++) wm.n_zeros[l] = c.h_scalars[l];
  // LF tables
  u32 hC[256];
  for (int k = 0; k < 256; k++) hC[k] = (u32)blk.C[k];
  DBuf<u32> dC(c, 256);
  HIP_CHECK(hipMemcpyAsync(dC.get(), hC, sizeof(hC), hipMemcpyHostToDevice, c.stream));
  wm.lf_base = DBuf<u32>(c, 512);
  WmView v;
  v.planes = wm.planes.get();
  v.level_words = n_blk * WM_BLOCK_WORDS;
  v.levels = levels;
  v.max_pos = (u32)(n_blk * WM_BLOCK_BITS - 1);
  for (int l = 0; l < 8; l++) v.n_zeros[l] = wm.n_zeros[l];
  hipLaunchKernelGGL(k_wm_tables, dim3(1), dim3(256), 0, c.stream, v, dC.get(), wm.lf_base.get());
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
  u32 q = wm_walk(wm, c, pos[k]);
  if (!lf) {
    out[k] = q - tab[256 + c];
  } else {
    u32 v = tab[c] + q;
    v += (v >= e_rank && c == e_char);
    out[k] = v;
  }
}

void rank_batch(Ctx &c, Block &blk, const u8 *d_sym, const u32 *d_pos, u32 *d_out, size_t nq, bool lf) {
  build_wavelet(c, blk);
  WmView v;
  v.planes = blk.wm.planes.get();
  v.level_words = blk.wm.n_blk * WM_BLOCK_WORDS;
  v.levels = blk.wm.levels;
  v.max_pos = (u32)(blk.wm.n_blk * WM_BLOCK_BITS - 1);
  for (int l = 0; l < 8; l++) v.n_zeros[l] = blk.wm.n_zeros[l];
  hipLaunchKernelGGL(k_rank_batch, dim3(ceil_div(nq, 256)), dim3(256), 0, c.stream, v, blk.wm.lf_base.get(), d_sym,
                     d_pos, d_out, (u32)nq, lf ? 1 : 0, (u32)blk.e_char, (u32)blk.e_rank);
  HIP_CHECK(hipGetLastError());
}

}  // namespace bwtamd
