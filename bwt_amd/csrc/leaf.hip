// Leaf BWT from the leaf suffix array: L[r] = T[SA[r]-1], the two special
// ranks, and the sampled suffix array (reference: include/bwt.h:589-609
// sa_to_bwt, :294-361 sample_sa).
#include <math.h>

#include "device_utils.h"
#include "internal.h"
#include "scan_tmpl.h"

namespace bwtamd {

// One thread per rank r.  A sampled position i lands in slot
// y = floor((i-a)(ns-1)/(b-a-1)): the slots are in position order, so the
// sample arrays come out sorted by position with no sort.  The two floor()
// expressions are the reference's own (include/bwt.h:339-340), evaluated in
// IEEE double exactly as there: 64-bit integer product, one conversion, one
// correctly rounded division.
__global__ void k_sa_to_bwt(const u8 *__restrict__ T, u64 n, u64 a, u64 m, const u32 *__restrict__ sa,
                            u8 *__restrict__ L, i64 ns, u32 *__restrict__ samp_rank,
                            u32 *__restrict__ samp_pos, u32 *__restrict__ special) {
  u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= m) return;
  u64 i = sa[r];
  if (i < a || i >= a + m) {  // not a permutation of the range: report, touch nothing
    special[2] = 1;
    return;
  }
  L[r] = i ? T[i - 1] : T[n - 1];
  if (i == a) special[0] = (u32)r;          // s_rank
  if (i + 1 == a + m) special[1] = (u32)r;  // e_rank
  if (m == 1) {
    samp_rank[0] = (u32)r;
    samp_pos[0] = (u32)i;
    return;
  }
  i64 rel = (i64)(i - a);
  i64 A = ns - 1;
  double D = (double)(i64)(m - 1);
  i64 x = (i64)floor((double)((rel - 1) * A) / D);
  i64 y = (i64)floor((double)(rel * A) / D);
  if (x != y) {
    samp_rank[y] = (u32)r;
    samp_pos[y] = (u32)i;
  }
}

struct SampFilledIn {
  const u32 *pos;
  __device__ __forceinline__ u32 operator()(size_t i) const { return pos[i] != 0xFFFFFFFFu; }
};
struct SampCompactOut {
  const u32 *rank_in, *pos_in;
  u32 *rank_out, *pos_out;
  u32 *total;
  size_t n;
  __device__ __forceinline__ void operator()(size_t i, u32 ex, u32 in) const {
    if (in != ex) {
      rank_out[ex] = rank_in[i];
      pos_out[ex] = pos_in[i];
    }
    if (i + 1 == n) *total = in;
  }
};

Block *leaf_from_sa(Ctx &c, i64 a, i64 b, const u32 *sa, i64 ns_req) {
  const i64 m = b - a;
  i64 ns = ns_req < m ? ns_req : m;  // include/bwt.h:607
  if (m >= 2 && ns < 2) fail(BWT_HIP_EINVAL, "ssa_n_samples must be at least 2 (got %ld)", (long)ns_req);
  Block *blk = new Block();
  try {
    blk->a = a;
    blk->b = b;
    blk->L = DBuf<u8>(c, (size_t)m);
    DBuf<u32> sr(c, (size_t)ns), sp(c, (size_t)ns);
    HIP_CHECK(hipMemsetAsync(sr.get(), 0xFF, (size_t)ns * 4, c.stream));
    HIP_CHECK(hipMemsetAsync(sp.get(), 0xFF, (size_t)ns * 4, c.stream));
    u32 *special = c.d_scalars + 16;
    HIP_CHECK(hipMemsetAsync(special, 0xFF, 8, c.stream));
    HIP_CHECK(hipMemsetAsync(special + 2, 0, 4, c.stream));
    {
      Phase ph(c, "leaf_sa_to_bwt", (double)m * 6);
      hipLaunchKernelGGL(k_sa_to_bwt, dim3(ceil_div(m, 256)), dim3(256), 0, c.stream, c.T, (u64)c.n, (u64)a,
                         (u64)m, sa, blk->L.get(), ns, sr.get(), sp.get(), special);
      HIP_CHECK(hipGetLastError());
    }
    blk->samp_rank = DBuf<u32>(c, (size_t)ns);
    blk->samp_pos = DBuf<u32>(c, (size_t)ns);
    u32 *d_total = c.d_scalars + 8;
    scan_generic<u32, SumU32>(c, (size_t)ns, SampFilledIn{sp.get()},
                              SampCompactOut{sr.get(), sp.get(), blk->samp_rank.get(), blk->samp_pos.get(),
                                             d_total, (size_t)ns});
    HIP_CHECK(hipMemcpyAsync(c.h_scalars, special, 8, hipMemcpyDeviceToHost, c.stream));
    HIP_CHECK(hipMemcpyAsync(c.h_scalars + 2, d_total, 4, hipMemcpyDeviceToHost, c.stream));
    HIP_CHECK(hipMemcpyAsync(c.h_scalars + 3, c.T + (b - 1), 1, hipMemcpyDeviceToHost, c.stream));
    HIP_CHECK(hipMemcpyAsync(c.h_scalars + 4, special + 2, 4, hipMemcpyDeviceToHost, c.stream));
    c.sync();
    if (c.h_scalars[0] == 0xFFFFFFFFu || c.h_scalars[1] == 0xFFFFFFFFu || c.h_scalars[4] != 0)
      fail(BWT_HIP_EINTERNAL, "leaf [%ld,%ld): suffix array is not a permutation of the range", (long)a, (long)b);
    blk->s_rank = c.h_scalars[0];
    blk->e_rank = c.h_scalars[1];
    blk->n_samp = c.h_scalars[2];
    blk->e_char = *(u8 *)&c.h_scalars[3];
  } catch (...) {
    delete blk;
    throw;
  }
  return blk;
}

}  // namespace bwtamd
