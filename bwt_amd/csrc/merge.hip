// Merge of two adjacent partial BWTs given the gap prefix sums (reference:
// include/bwt.h:791-847 bwt_merge, :366-453 resample_sa).
//
// With S[i] = gap[0] + ... + gap[i], left symbol i lands at F(i) = i + S[i] and
// the right symbols fill the other positions in order.  The kernel is driven by
// OUTPUT chunks, so every workgroup writes the same number of contiguous bytes
// whatever the gap distribution (one huge gap does not unbalance it).
#include "device_utils.h"
#include "internal.h"
#include "scan_tmpl.h"

namespace bwtamd {

constexpr int MG_THREADS = 256;
constexpr int MG_PER_THREAD = 16;
constexpr int MG_CHUNK = MG_THREADS * MG_PER_THREAD;  // 4096 output bytes per workgroup

// Number of i in [0, m) with i + S[i] < P; all 64 lanes cooperate (64-ary search).
// The first round probes around the position a uniform interleaving would give (left_share = m / total), which
// on most inputs leaves a range of a few hundred candidates instead of m.
__device__ __forceinline__ u32 wave_count_below(const u32 *__restrict__ S, u32 m, u64 P, float left_share) {
  u32 lo = 0, hi = m;  // answer in [lo, hi]
  const u32 lane = lane_id();
  if (m > 64 * 1024) {
    constexpr long long W = 512;
    long long est = (long long)((double)P * (double)left_share);
    long long idx = est + ((long long)lane - 32) * W;
    bool t = idx < 0 || (idx < (long long)m && (u64)idx + S[idx] < P);
    u32 cnt = (u32)__popcll(__ballot(t));
    long long first = est - 32 * W;  // probe of lane 0
    if (cnt > 0) {
      long long l = first + (long long)(cnt - 1) * W + 1;  // one past the last probe that is below P
      lo = l < 0 ? 0u : (l > (long long)m ? m : (u32)l);
    }
    if (cnt < 64) {
      long long h = first + (long long)cnt * W;  // the first probe that is not below P
      hi = h < 0 ? 0u : (h > (long long)m ? m : (u32)h);
    }
  }
  for (;;) {
    u32 span = hi - lo;
    if (span <= 64) {
      bool t = false;
      if (lane < span) {
        u32 i = lo + lane;
        t = (u64)i + S[i] < P;
      }
      return lo + (u32)__popcll(__ballot(t));
    }
    u32 step = span / 64;
    u32 i = lo + (lane + 1) * step - 1;
    bool t = (u64)i + S[i] < P;
    u32 cnt = (u32)__popcll(__ballot(t));
    u32 nlo = lo + cnt * step;
    if (cnt < 64) hi = lo + (cnt + 1) * step - 1;
    lo = nlo;
  }
}

// bounds[k] = number of left symbols placed before output position k * MG_CHUNK (k = 0 .. chunks): one wavefront
// per boundary, so that the merge kernel's workgroups start with their stretch of S known instead of each searching
// for both of its ends (three dependent rounds of loads at the head of every workgroup).
__global__ void __launch_bounds__(256) k_merge_bounds(const u32 *__restrict__ S, u32 m, u64 total, float left_share, u32 chunks,
                                                      u32 *__restrict__ bounds) {
  const u32 k = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (k > chunks) return;
  const u64 P = (u64)k * MG_CHUNK < total ? (u64)k * MG_CHUNK : total;
  const u32 v = wave_count_below(S, m, P, left_share);
  if (lane_id() == 0) bounds[k] = v;
}

__global__ void __launch_bounds__(MG_THREADS)
    k_merge(const u8 *__restrict__ lL, const u8 *__restrict__ rL, const u32 *__restrict__ S, u64 total,
            const u32 *__restrict__ bounds, u8 *__restrict__ out) {
  __shared__ u8 is_left[MG_CHUNK];
  __shared__ __align__(16) u8 src_l[MG_CHUNK + 32], src_r[MG_CHUNK + 32];  // the chunk's left and right symbols
  __shared__ u32 scan_s[17];
  const u32 tid = threadIdx.x;
  // every XCD takes a contiguous range of chunks: the sectors that neighbouring chunks share then meet in one L2
  const u32 chunk = xcd_tile(blockIdx.x, gridDim.x);
  const u64 p0 = (u64)chunk * MG_CHUNK;
  const u64 p1 = (p0 + MG_CHUNK < total) ? p0 + MG_CHUNK : total;
  const u32 tlo = bounds[chunk], thi = bounds[chunk + 1];
  reinterpret_cast<uint4 *>(is_left)[tid] = make_uint4(0, 0, 0, 0);
  const u32 n_left = thi - tlo, n_right = (u32)(p1 - p0) - n_left;
  const u64 r0 = p0 - tlo;  // right symbols placed before p0
  // All loads of the chunk are issued before anything waits for one of them: its stretch of S (at most MG_PER_THREAD
  // words a thread), its left and its right symbols (aligned 16-byte pieces, at most two a thread each; an aligned
  // 16-byte block never straddles a page, so reading the whole block of a valid byte is safe).
  const uintptr_t al = reinterpret_cast<uintptr_t>(lL + tlo), ar = reinterpret_cast<uintptr_t>(rL + r0);
  const u32 off_l = (u32)(al & 15), off_r = (u32)(ar & 15);
  const uint4 *bl = reinterpret_cast<const uint4 *>(al - off_l), *br = reinterpret_cast<const uint4 *>(ar - off_r);
  const u32 wl = n_left ? (off_l + n_left + 15) / 16 : 0u, wr = n_right ? (off_r + n_right + 15) / 16 : 0u;  // <= MG_THREADS + 1
  u32 sv[MG_PER_THREAD];
#pragma unroll
  for (int k = 0; k < MG_PER_THREAD; k++) {
    const u32 t = tlo + (u32)k * MG_THREADS + tid;
    sv[k] = t < thi ? S[t] : 0u;
  }
  uint4 vl[2], vr[2];
#pragma unroll
  for (int k = 0; k < 2; k++) {
    const u32 q = (u32)k * MG_THREADS + tid;
    vl[k] = q < wl ? bl[q] : make_uint4(0, 0, 0, 0);
    vr[k] = q < wr ? br[q] : make_uint4(0, 0, 0, 0);
  }
  __syncthreads();  // the flags are cleared
#pragma unroll
  for (int k = 0; k < MG_PER_THREAD; k++) {
    const u32 t = tlo + (u32)k * MG_THREADS + tid;
    if (t < thi) is_left[(u64)t + sv[k] - p0] = 1;
  }
#pragma unroll
  for (int k = 0; k < 2; k++) {
    const u32 q = (u32)k * MG_THREADS + tid;
    if (q < wl) reinterpret_cast<uint4 *>(src_l)[q] = vl[k];
    if (q < wr) reinterpret_cast<uint4 *>(src_r)[q] = vr[k];
  }
  __syncthreads();
  const u32 q0 = tid * MG_PER_THREAD;
  uint4 fl = reinterpret_cast<const uint4 *>(is_left)[tid];
  u32 fw[4] = {fl.x, fl.y, fl.z, fl.w};
  u32 mine = __popc(fw[0]) + __popc(fw[1]) + __popc(fw[2]) + __popc(fw[3]);  // flags are 0/1 bytes
  u32 all;
  u32 incl = block_scan_incl<OpSum>(mine, scan_s, &all);
  u32 li = off_l + (incl - mine);        // next left symbol, in src_l
  u32 ri = off_r + q0 - (incl - mine);   // next right symbol, in src_r
  if (p0 + q0 >= p1) return;
  u32 ow[4];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    u32 word = 0;
#pragma unroll
    for (int b = 0; b < 4; b++) {
      u64 p = p0 + q0 + k * 4 + b;
      u32 ch = 0;
      if (p < p1) {
        if ((fw[k] >> (8 * b)) & 1) ch = src_l[li++];
        else ch = src_r[ri++];
      }
      word |= ch << (8 * b);
    }
    ow[k] = word;
  }
  if (p0 + q0 + MG_PER_THREAD <= p1) {
    reinterpret_cast<uint4 *>(out + p0)[tid] = make_uint4(ow[0], ow[1], ow[2], ow[3]);
  } else {
    for (u32 q = 0; p0 + q0 + q < p1; q++) out[p0 + q0 + q] = (u8)(ow[q >> 2] >> (8 * (q & 3)));
  }
}

// ----------------------------------------------------------------------------
// resample: the i-th sample (left samples by position, then right samples by
// position) is kept iff n_taken * (nn-1) <= i * (ns-1) in the reference's
// serial walk (include/bwt.h:412, :436).  n_taken before sample i has the
// closed form t(i) = min(i, 1 + floor((i-1)(ns-1)/(nn-1))) for i >= 1, t(0) = 0,
// so sample i is kept iff t(i+1) > t(i) and lands in slot t(i).
// ----------------------------------------------------------------------------
__device__ __forceinline__ u64 taken_before(u64 i, u64 A, u64 B) {
  if (i == 0) return 0;
  if (B == 0) return 1;
  u64 t = 1 + div_floor_u64((i - 1) * A, B);
  return t < i ? t : i;
}

__global__ void k_resample(const u32 *__restrict__ l_rank, const u32 *__restrict__ l_pos, u32 nl,
                           const u32 *__restrict__ r_rank, const u32 *__restrict__ r_pos, u32 nr,
                           const u32 *__restrict__ S, const u32 *__restrict__ r_g, u64 A, u64 B,
                           u32 *__restrict__ o_rank, u32 *__restrict__ o_pos) {
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nl + nr) return;
  u64 t0 = taken_before(i, A, B), t1 = taken_before((u64)i + 1, A, B);
  if (t1 == t0) return;
  u32 rank, pos;
  if (i < nl) {
    u32 rk = l_rank[i];
    rank = rk + S[rk];  // rank + gap.sum(rank), include/bwt.h:413-417
    pos = l_pos[i];
  } else {
    u32 k = i - nl;
    rank = r_rank[k] + r_g[k];  // include/bwt.h:778-781 then :437
    pos = r_pos[k];
  }
  o_rank[t0] = rank;
  o_pos[t0] = pos;
}

// out[0 : m + rn) = lL and rL interleaved as S says
void merge_symbols(Ctx &c, const u8 *lL, u64 m, const u8 *rL, u64 rn, const u32 *S, u8 *out) {
  const u64 total = m + rn;
  if (total == 0) return;
  Phase ph(c, "merge_scatter", (double)total * 2 + (double)(m + 1) * 4);
  const u32 chunks = ceil_div(total, MG_CHUNK);
  DBuf<u32> bounds(c, (size_t)chunks + 1);
  hipLaunchKernelGGL(k_merge_bounds, dim3(chunks / 4 + 1), dim3(256), 0, c.stream, S, (u32)m, total, (float)((double)m / (double)total),
                     chunks, bounds.get());
  hipLaunchKernelGGL(k_merge, dim3(chunks), dim3(MG_THREADS), 0, c.stream, lL, rL, S, total, (const u32 *)bounds.get(), out);
  HIP_CHECK(hipGetLastError());
}

Block *merge_blocks(Ctx &c, Block &l, Block &r, GapResult &gap, i64 ns_req) {
  const u64 m = (u64)l.len(), rn = (u64)r.len();
  const u64 total = m + rn;
  Block *blk = new Block();
  try {
    blk->a = l.a;
    blk->b = r.b;
    blk->e_char = r.e_char;
    blk->L = DBuf<u8>(c, total);
    merge_symbols(c, l.L.get(), m, r.L.get(), rn, gap.S.get(), blk->L.get());
    // samples
    i64 ns = ns_req < (i64)total ? ns_req : (i64)total;  // include/bwt.h:840
    const u64 nn = (u64)(l.n_samp + r.n_samp);
    const u64 A = (u64)(ns - 1), B = nn - 1;
    u64 kept = nn;
    if (nn >= 1) {
      u64 t = 1 + (B ? ((nn - 1) * A) / B : 0);
      kept = t < nn ? t : nn;
    }
    blk->samp_rank = DBuf<u32>(c, kept ? kept : 1);
    blk->samp_pos = DBuf<u32>(c, kept ? kept : 1);
    blk->n_samp = (i64)kept;
    {
      Phase ph(c, "merge_resample", (double)nn * 16);
      hipLaunchKernelGGL(k_resample, dim3(ceil_div(nn, 256)), dim3(256), 0, c.stream, l.samp_rank.get(),
                         l.samp_pos.get(), (u32)l.n_samp, r.samp_rank.get(), r.samp_pos.get(), (u32)r.n_samp,
                         gap.S.get(), gap.samp_g.get(), A, B, blk->samp_rank.get(), blk->samp_pos.get());
      HIP_CHECK(hipGetLastError());
    }
    // s_rank: new rank of T[a:], e_rank: new rank of T[b-1:]  (include/bwt.h:418-421, :438-441)
    HIP_CHECK(hipMemcpyAsync(c.h_scalars, gap.S.get() + l.s_rank, 4, hipMemcpyDeviceToHost, c.stream));
    HIP_CHECK(hipMemcpyAsync(c.h_scalars + 1, blk->samp_rank.get() + (kept - 1), 4, hipMemcpyDeviceToHost, c.stream));
    HIP_CHECK(hipMemcpyAsync(c.h_scalars + 2, blk->samp_pos.get() + (kept - 1), 4, hipMemcpyDeviceToHost, c.stream));
    c.sync();
    blk->s_rank = l.s_rank + (i64)c.h_scalars[0];
    if ((i64)c.h_scalars[2] != r.b - 1)
      fail(BWT_HIP_EINTERNAL, "merge [%ld,%ld): last sample is at %u, not at b-1", (long)l.a, (long)r.b, c.h_scalars[2]);
    blk->e_rank = c.h_scalars[1];
  } catch (...) {
    delete blk;
    throw;
  }
  return blk;
}

}  // namespace bwtamd
