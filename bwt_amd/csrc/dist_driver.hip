// bwt::pmbwt over several GPUs (declared in include/bwt_hip.h, "the whole build over several GPUs").
//
// The recursion tree of bwt_rec (reference include/bwt.h:854-865: split at (a+b)/2) is cut at depth
// log2(world); rank k builds subtree k on its own GPU.  A block of the tree that spans q ranks stays spread
// over them as q slices of its L (rank ranges); each keeps the block's samples whose positions lie in its own
// text range; s_rank is known to the first of them, e_rank to the last.  A merge
// of two such blocks (include/bwt.h:791-847) runs on all G = 2q ranks:
//   1. every rank receives the left block's L (each holder sends its slice to all others) and builds the
//      left rank structure;
//   2. the right block's text positions are split evenly; every rank runs the LF chains of its share
//      (g_j depends on the left block and the text only, and every rank holds the text);
//   3. the merge is cut by left rank into G parts (partition_bounds): one all-to-all moves every g value
//      to the rank that owns its part, a second one the right symbols that part interleaves;
//   4. every rank builds its part's prefix sums and writes its slice of the merged L;
//   5. the samples do not move: every rank keeps those of its own text range, asks the rank that knows for
//      the number each new rank needs (the part's prefix sums for a left sample, g for a right one) and thins
//      its own by the reference's rule.
// Same tree, same rules: the slices concatenated are the L of the single-GPU build, bit for bit.
//
// Ranks move device memory through a bwt_hip_transport (one primitive: a batch of sends and receives).  Two
// ship here: RCCL (a process or thread per GPU, ncclSend/ncclRecv over xGMI) and an in-process one (a thread
// per GPU, peer copies), which bwt_hip_pmbwt_multi uses.
#include <dlfcn.h>
#include <string.h>

#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <thread>

#include "../../include/bwt_hip.h"
#include "internal.h"

namespace bwtamd {

// ---------------------------------------------------------------------------
// the tree
// ---------------------------------------------------------------------------
static void split_ranges(i64 a, i64 b, int depth, std::vector<std::pair<i64, i64>> &out) {
  if (depth == 0) {
    out.push_back({a, b});
    return;
  }
  const i64 mid = (a + b) / 2;
  split_ranges(a, mid, depth - 1, out);
  split_ranges(mid, b, depth - 1, out);
}

// Does bwt_rec with this threshold split every range down to depth `depth`?  (Then the tree has 2^depth
// subtrees there, one per rank.)
static bool tree_reaches(i64 n, i64 th, int depth) {
  if (depth == 0) return true;
  std::vector<std::pair<i64, i64>> parents;
  split_ranges(0, n, depth - 1, parents);
  for (auto &r : parents)
    if (r.second - r.first <= th || r.second - r.first < 2) return false;
  return true;
}

// ---------------------------------------------------------------------------
// small kernels
// ---------------------------------------------------------------------------
__global__ void k_u32_to_i64(const u32 *__restrict__ in, u64 n, i64 *__restrict__ out) {
  const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = (i64)in[i];
}
__global__ void k_abs_ranks(const u32 *__restrict__ rel, u64 n, u32 a, u32 *__restrict__ out) {
  const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = rel[i] + a;
}

template <class K, class... A>
static void launch1d(Ctx &c, K kernel, u64 n, A... args) {
  if (n == 0) return;
  hipLaunchKernelGGL(kernel, dim3(ceil_div(n, 256)), dim3(256), 0, c.stream, args...);
  HIP_CHECK(hipGetLastError());
}

// ---------------------------------------------------------------------------
// a block spread over ranks
// ---------------------------------------------------------------------------
struct DistBlock {
  i64 a = 0, b = 0;
  int first = 0, q = 1;       // held by the ranks first .. first+q-1
  std::vector<i64> sizes;     // symbols of every holder's slice
  DBuf<u8> L;                 // this rank's slice
  i64 s_rank = -1, e_rank = -1;  // absolute; s_rank on the first holder only, e_rank on the last only
  DBuf<u32> samp_rank, samp_pos; // absolute ranks, positions: the block's samples inside this rank's own text range, by position
  i64 n_samp = 0;                // of this rank
};

struct Group {
  bwt_hip_transport *t;
  Ctx &c;
  int first, G, me, k;  // ranks first .. first+G-1; k = me - first

  void exchange(std::vector<bwt_hip_p2p> &ops) {
    c.sync();  // what is sent has been written; what is received into is no longer read
    std::vector<bwt_hip_p2p> live;
    for (auto &o : ops)
      if (o.bytes > 0) live.push_back(o);
    if (live.empty()) return;
    const int rc = t->exchange(t->self, (int)live.size(), live.data());
    if (rc != 0) fail(rc, "transport: exchange of %zu transfers failed on rank %d", live.size(), me);
  }

  // out[j * count + i] = value i of group member j
  std::vector<i64> all_gather(const std::vector<i64> &mine) {
    const size_t cnt = mine.size();
    DBuf<i64> d(c, cnt * (size_t)G);
    HIP_CHECK(hipMemcpyAsync(d.get() + cnt * (size_t)k, mine.data(), cnt * 8, hipMemcpyHostToDevice, c.stream));
    std::vector<bwt_hip_p2p> ops;
    for (int j = 0; j < G; j++)
      if (j != k) ops.push_back({first + j, 1, d.get() + cnt * (size_t)k, (i64)cnt * 8});
    for (int j = 0; j < G; j++)
      if (j != k) ops.push_back({first + j, 0, d.get() + cnt * (size_t)j, (i64)cnt * 8});
    exchange(ops);
    std::vector<i64> out(cnt * (size_t)G);
    HIP_CHECK(hipMemcpyAsync(out.data(), d.get(), out.size() * 8, hipMemcpyDeviceToHost, c.stream));
    c.sync();
    return out;
  }

  // piece j of `send` (send_bytes[j], consecutive) goes to member j; recv_bytes[j] arrive from member j, consecutive in recv
  void all_to_all(const u8 *send, const std::vector<i64> &send_bytes, u8 *recv, const std::vector<i64> &recv_bytes) {
    std::vector<bwt_hip_p2p> ops;
    i64 so = 0, ro = 0, my_so = 0, my_ro = 0;
    for (int j = 0; j < G; j++) {
      if (j == k) my_so = so;
      else ops.push_back({first + j, 1, const_cast<u8 *>(send) + so, send_bytes[(size_t)j]});
      so += send_bytes[(size_t)j];
    }
    for (int j = 0; j < G; j++) {
      if (j == k) my_ro = ro;
      else ops.push_back({first + j, 0, recv + ro, recv_bytes[(size_t)j]});
      ro += recv_bytes[(size_t)j];
    }
    if (send_bytes[(size_t)k] != recv_bytes[(size_t)k]) fail(BWT_HIP_EINTERNAL, "all-to-all: own piece %ld != %ld", (long)send_bytes[(size_t)k], (long)recv_bytes[(size_t)k]);
    if (send_bytes[(size_t)k] > 0)
      HIP_CHECK(hipMemcpyAsync(recv + my_ro, send + my_so, (size_t)send_bytes[(size_t)k], hipMemcpyDeviceToDevice, c.stream));
    exchange(ops);
  }
};

static i64 overlap(i64 lo0, i64 hi0, i64 lo1, i64 hi1) {
  const i64 lo = lo0 > lo1 ? lo0 : lo1, hi = hi0 < hi1 ? hi0 : hi1;
  return hi > lo ? hi - lo : 0;
}

// Merges the block held by the group's first half (left) with the one held by its second half (right); every
// member calls this with the block it holds a slice of.  Consumes blk.
static std::unique_ptr<DistBlock> dist_merge(Group &g, std::unique_ptr<DistBlock> blk, i64 ns_req, i64 seg) {
  Ctx &c = g.c;
  const int G = g.G, q = G / 2, k = g.k;
  const int mine_at = g.me - blk->first;
  std::vector<i64> meta = g.all_gather({blk->a, blk->b, blk->s_rank, blk->e_rank, blk->n_samp, blk->sizes[(size_t)mine_at]});
  auto M = [&](int j, int f) { return meta[(size_t)j * 6 + (size_t)f]; };
  const i64 a = M(0, 0), cpos = M(0, 1), b = M(q, 1);
  if (M(q, 0) != cpos) fail(BWT_HIP_EINTERNAL, "blocks are not adjacent: [%ld,%ld) and [%ld,%ld)", (long)a, (long)cpos, (long)M(q, 0), (long)b);
  const i64 m = cpos - a, r = b - cpos;
  std::vector<i64> loff(1, 0), roff(1, 0);
  for (int j = 0; j < q; j++) loff.push_back(loff.back() + M(j, 5));
  for (int j = 0; j < q; j++) roff.push_back(roff.back() + M(q + j, 5));
  if (loff.back() != m || roff.back() != r) fail(BWT_HIP_EINTERNAL, "slice sizes do not add up to the blocks");
  const i64 s_rank_l = M(0, 2), e_rank_l = M(q - 1, 3);

  // 1. the left block's L on every member
  std::unique_ptr<Block> lb(new Block());
  lb->a = a;
  lb->b = cpos;
  lb->s_rank = s_rank_l - a;
  lb->e_rank = e_rank_l - a;
  lb->L = DBuf<u8>(c, (size_t)m);
  const bool packed_wire = wire_packable(c);  // the same on every member: they hold the same text
  {
    // two bits per symbol on the wire where the alphabet allows it: a quarter of the bytes, which at the top
    // of the tree are the largest transfer of the build (every member receives the whole left block)
    std::vector<bwt_hip_p2p> ops;
    std::vector<size_t> in_at((size_t)q + 1, 0);
    for (int j = 0; j < q; j++)
      in_at[(size_t)j + 1] = in_at[(size_t)j] + ((packed_wire && j != k) ? ((wire_bytes((u64)M(j, 5)) + WIRE_PAD + 15) & ~(size_t)15) : 0);
    DBuf<u8> in_pk(c, in_at[(size_t)q] + 16), out_pk(c, (packed_wire && k < q ? wire_bytes((u64)M(k, 5)) : 0) + 16);
    if (k < q) {
      HIP_CHECK(hipMemcpyAsync(lb->L.get() + loff[(size_t)k], blk->L.get(), (size_t)M(k, 5), hipMemcpyDeviceToDevice, c.stream));
      if (packed_wire) wire_pack2(c, blk->L.get(), (u64)M(k, 5), out_pk.get());
      for (int j = 0; j < G; j++)
        if (j != k) ops.push_back({g.first + j, 1, packed_wire ? out_pk.get() : blk->L.get(), packed_wire ? (i64)wire_bytes((u64)M(k, 5)) : M(k, 5)});
    }
    for (int j = 0; j < q; j++)
      if (j != k) {
        if (packed_wire) ops.push_back({g.first + j, 0, in_pk.get() + in_at[(size_t)j], (i64)wire_bytes((u64)M(j, 5))});
        else ops.push_back({g.first + j, 0, lb->L.get() + loff[(size_t)j], M(j, 5)});
      }
    g.exchange(ops);
    if (packed_wire) {
      for (int j = 0; j < q; j++)
        if (j != k) wire_unpack2(c, in_pk.get() + in_at[(size_t)j], (u64)M(j, 5), lb->L.get() + loff[(size_t)j]);
      // the block that starts at text position 0 holds the terminator at s_rank, which has no two-bit code
      if (a == 0) HIP_CHECK(hipMemsetAsync(lb->L.get() + (s_rank_l - a), 0, 1, c.stream));
      c.sync();  // the packed buffers die here
    }
  }
  HIP_CHECK(hipMemcpyAsync(c.h_scalars, c.T + (cpos - 1), 1, hipMemcpyDeviceToHost, c.stream));
  c.sync();
  lb->e_char = *(u8 *)c.h_scalars;

  // Where the gap histogram groups its values in two passes (left blocks of 2^21 suffixes and more), the g values
  // travel grouped: the sender runs that grouping on the values of its own chains and sends each part the 16-bit
  // words of the part's buckets plus the buckets' bounds, half the bytes of the values themselves, and the receiver
  // counts the words of all senders bucket by bucket.  Parts then begin at multiples of 8192 left ranks.
  const int shift = partition_shift(c, (u64)m);
  const bool grouped = shift != 0;
  std::vector<i64> lo((size_t)G + 1);
  partition_bounds((u64)m, (u32)G, lo.data(), shift);

  // 2. LF chains of this member's share of the right positions
  const i64 p_lo = cpos + (i64)k * r / G, p_hi = cpos + (i64)(k + 1) * r / G;
  const i64 cnt = p_hi - p_lo;
  DBuf<u32> gt(c, (size_t)(cnt > 0 ? cnt : 1)), g0;
  u64 total = 0;  // entries of g0: the chains' values in step-major order, the short last chain filled up with m + 1
  if (cnt > 0) {
    ChainLayout lay = run_chains(c, *lb, p_lo, p_hi, seg, g0);
    chains_text_order(c, lay, g0.get(), gt.get());
    total = (u64)lay.seg * lay.n_chains;
  }
  if (!grouped) g0.reset();
  lb->wm = WaveletDev();  // the rank structure is not needed any more; L is (step 4)

  // 3. every g value to the member that owns its part, then the right symbols each part interleaves
  std::vector<i64> R((size_t)G, 0), rb(1, 0), send_b((size_t)G), recv_b((size_t)G);
  DBuf<u32> g_in, st_in, w_off;
  DBuf<u16> w_in;
  u32 nbk = 0;  // buckets of this member's part
  if (!grouped) {
    DBuf<u32> g_parts(c, (size_t)(cnt > 0 ? cnt : 1));
    std::vector<i64> counts((size_t)G, 0);
    partition_by_left_rank(c, gt.get(), (u64)cnt, (u64)m, (u32)G, g_parts.get(), counts.data());
    std::vector<i64> table = g.all_gather(counts);  // table[src * G + part]
    for (int d = 0; d < G; d++)
      for (int s = 0; s < G; s++) R[(size_t)d] += table[(size_t)s * (size_t)G + (size_t)d];
    g_in = DBuf<u32>(c, (size_t)(R[(size_t)k] > 0 ? R[(size_t)k] : 1));
    for (int j = 0; j < G; j++) {
      send_b[(size_t)j] = counts[(size_t)j] * 4;
      recv_b[(size_t)j] = table[(size_t)j * (size_t)G + (size_t)k] * 4;
    }
    g.all_to_all((const u8 *)g_parts.get(), send_b, (u8 *)g_in.get(), recv_b);
  } else {
    const u32 nb = (u32)(((u64)m >> shift) + 1);  // buckets of the whole left block
    std::vector<u32> B((size_t)G + 1, nb);        // part d owns the buckets B[d] .. B[d+1]-1
    for (int d = 0; d < G; d++) B[(size_t)d] = (u32)((u64)lo[(size_t)d] >> shift) < nb ? (u32)((u64)lo[(size_t)d] >> shift) : nb;
    const size_t nst = gap_group16_starts((u64)m);
    DBuf<u32> start(c, nst);
    if (total > 0) {
      DBuf<u32> g1(c, (size_t)total);
      gap_group16(c, g0.get(), g1.get(), total, (u64)m, start.get());
      c.sync();  // g1 dies here
    } else {
      HIP_CHECK(hipMemsetAsync(start.get(), 0, nst * sizeof(u32), c.stream));
    }
    std::vector<u32> h_st((size_t)G + 1);
    for (int d = 0; d <= G; d++) HIP_CHECK(hipMemcpyAsync(&h_st[(size_t)d], start.get() + B[(size_t)d], sizeof(u32), hipMemcpyDeviceToHost, c.stream));
    c.sync();
    // the fillers lie in the bucket of m + 1: the last part's last bucket (its receiver skips them), or one past all
    const i64 fillers_in = ((((u64)m + 1) >> shift) < nb) ? (i64)total - cnt : 0;
    std::vector<i64> words((size_t)G + 1, 0);
    for (int d = 0; d < G; d++) words[(size_t)d] = (i64)h_st[(size_t)d + 1] - (i64)h_st[(size_t)d];
    words[(size_t)G] = fillers_in;
    std::vector<i64> table = g.all_gather(words);  // table[src * (G + 1) + part], [.. + G] = the sender's fillers
    const size_t W = (size_t)G + 1;
    i64 in_words = 0;
    std::vector<u32> h_off((size_t)G, 0);
    for (int s = 0; s < G; s++) {
      for (int d = 0; d < G; d++) R[(size_t)d] += table[(size_t)s * W + (size_t)d];
      R[(size_t)G - 1] -= table[(size_t)s * W + (size_t)G];
      h_off[(size_t)s] = (u32)in_words;
      in_words += table[(size_t)s * W + (size_t)k];
    }
    w_in = DBuf<u16>(c, (size_t)in_words + 8);
    for (int j = 0; j < G; j++) {
      send_b[(size_t)j] = words[(size_t)j] * 2;
      recv_b[(size_t)j] = table[(size_t)j * W + (size_t)k] * 2;
    }
    g.all_to_all((const u8 *)g0.get(), send_b, (u8 *)w_in.get(), recv_b);  // a part's buckets follow the part before it
    // the bounds of every part's buckets (piece d = start[B[d] .. B[d+1]], one entry more than buckets)
    nbk = B[(size_t)k + 1] - B[(size_t)k];
    size_t st_total = 0;
    for (int d = 0; d < G; d++) st_total += (size_t)(B[(size_t)d + 1] - B[(size_t)d]) + 1;
    DBuf<u32> st_out(c, st_total);
    size_t at = 0;
    for (int d = 0; d < G; d++) {
      const size_t len = (size_t)(B[(size_t)d + 1] - B[(size_t)d]) + 1;
      HIP_CHECK(hipMemcpyAsync(st_out.get() + at, start.get() + B[(size_t)d], len * sizeof(u32), hipMemcpyDeviceToDevice, c.stream));
      send_b[(size_t)d] = (i64)len * 4;
      recv_b[(size_t)d] = ((i64)nbk + 1) * 4;
      at += len;
    }
    st_in = DBuf<u32>(c, (size_t)G * ((size_t)nbk + 1));
    g.all_to_all((const u8 *)st_out.get(), send_b, (u8 *)st_in.get(), recv_b);
    w_off = DBuf<u32>(c, (size_t)G);
    HIP_CHECK(hipMemcpyAsync(w_off.get(), h_off.data(), (size_t)G * sizeof(u32), hipMemcpyHostToDevice, c.stream));
    c.sync();  // h_off is on the stack; the grouped values and the bounds that were sent die here
    g0.reset();
  }
  for (int d = 0; d < G; d++) rb.push_back(rb.back() + R[(size_t)d]);
  if (rb.back() != r) fail(BWT_HIP_EINTERNAL, "parts hold %ld of %ld right suffixes", (long)rb.back(), (long)r);
  const i64 rn = R[(size_t)k];
  DBuf<u8> rL(c, (size_t)(rn > 0 ? rn : 1));
  for (int d = 0; d < G; d++) {
    send_b[(size_t)d] = k >= q ? overlap(roff[(size_t)(k - q)], roff[(size_t)(k - q + 1)], rb[(size_t)d], rb[(size_t)d + 1]) : 0;
    recv_b[(size_t)d] = d >= q ? overlap(roff[(size_t)(d - q)], roff[(size_t)(d - q + 1)], rb[(size_t)k], rb[(size_t)k + 1]) : 0;
  }
  if (!packed_wire) {
    g.all_to_all(k >= q ? blk->L.get() : (const u8 *)rL.get(), send_b, rL.get(), recv_b);
  } else {
    // the same pieces as two bits per symbol: a holder's pieces follow each other in its slice, a part's in rL
    std::vector<size_t> s_at((size_t)G + 1, 0), r_at((size_t)G + 1, 0), s_sym((size_t)G + 1, 0), r_sym((size_t)G + 1, 0);
    for (int d = 0; d < G; d++) {
      s_sym[(size_t)d + 1] = s_sym[(size_t)d] + (size_t)send_b[(size_t)d];
      r_sym[(size_t)d + 1] = r_sym[(size_t)d] + (size_t)recv_b[(size_t)d];
      s_at[(size_t)d + 1] = s_at[(size_t)d] + (d != k ? ((wire_bytes((u64)send_b[(size_t)d]) + 15) & ~(size_t)15) : 0);
      r_at[(size_t)d + 1] = r_at[(size_t)d] + (d != k ? ((wire_bytes((u64)recv_b[(size_t)d]) + WIRE_PAD + 15) & ~(size_t)15) : 0);
    }
    DBuf<u8> s_pk(c, s_at[(size_t)G] + 16), r_pk(c, r_at[(size_t)G] + 16);
    std::vector<bwt_hip_p2p> ops;
    for (int d = 0; d < G; d++)
      if (d != k && send_b[(size_t)d] > 0) {
        wire_pack2(c, blk->L.get() + s_sym[(size_t)d], (u64)send_b[(size_t)d], s_pk.get() + s_at[(size_t)d]);
        ops.push_back({g.first + d, 1, s_pk.get() + s_at[(size_t)d], (i64)wire_bytes((u64)send_b[(size_t)d])});
      }
    for (int d = 0; d < G; d++)
      if (d != k && recv_b[(size_t)d] > 0) ops.push_back({g.first + d, 0, r_pk.get() + r_at[(size_t)d], (i64)wire_bytes((u64)recv_b[(size_t)d])});
    if (send_b[(size_t)k] != recv_b[(size_t)k]) fail(BWT_HIP_EINTERNAL, "right symbols: own piece %ld != %ld", (long)send_b[(size_t)k], (long)recv_b[(size_t)k]);
    if (send_b[(size_t)k] > 0)
      HIP_CHECK(hipMemcpyAsync(rL.get() + r_sym[(size_t)k], blk->L.get() + s_sym[(size_t)k], (size_t)send_b[(size_t)k], hipMemcpyDeviceToDevice, c.stream));
    g.exchange(ops);
    for (int d = 0; d < G; d++)
      if (d != k && recv_b[(size_t)d] > 0) wire_unpack2(c, r_pk.get() + r_at[(size_t)d], (u64)recv_b[(size_t)d], rL.get() + r_sym[(size_t)d]);
    c.sync();  // the packed buffers die here
  }

  // 4. this part's slice of the merged block
  auto clampm = [&](i64 v) { return v < m ? v : m; };
  const i64 i_lo = clampm(lo[(size_t)k]), i_hi = clampm(lo[(size_t)k + 1]), mL = i_hi - i_lo;
  std::unique_ptr<DistBlock> out(new DistBlock());
  out->a = a;
  out->b = b;
  out->first = g.first;
  out->q = G;
  for (int d = 0; d < G; d++) out->sizes.push_back(clampm(lo[(size_t)d + 1]) - clampm(lo[(size_t)d]) + R[(size_t)d]);
  out->L = DBuf<u8>(c, (size_t)(mL + rn > 0 ? mL + rn : 1));
  DBuf<u32> S(c, (size_t)mL + 1);
  if (!grouped) {
    DBuf<u32> scratch(c, (size_t)(rn > 0 ? rn : 1));
    gap_prefix(c, g_in.get(), scratch.get(), (u64)rn, (u64)mL, S.get());
  } else {
    gap_prefix_grouped(c, w_in.get(), w_off.get(), st_in.get(), (u32)G, nbk, (u64)mL, S.get());
  }
  merge_symbols(c, lb->L.get() + i_lo, (u64)mL, rL.get(), (u64)rn, S.get(), out->L.get());

  // 5. samples stay with their holders: requests out, answers back, each holder thins its own
  {
    u64 nl = 0, nr = 0, first_index = 0;
    for (int j = 0; j < G; j++) {
      if (j < k) first_index += (u64)M(j, 4);
      (j < q ? nl : nr) += (u64)M(j, 4);
    }
    const u64 nn = nl + nr, mine = (u64)blk->n_samp;
    const i64 ns_out = ns_req < m + r ? ns_req : m + r;
    const u64 A = (u64)(ns_out - 1), B = nn ? nn - 1 : 0;
    DBuf<u32> queries(c, mine ? mine : 1), perm(c, mine ? mine : 1);
    std::vector<i64> ask((size_t)G, 0);
    if (k < q) samples_route_left(c, blk->samp_rank.get(), mine, a, (u64)m, (u32)G, queries.get(), perm.get(), ask.data(), shift);
    else samples_route_right(c, blk->samp_pos.get(), mine, cpos, (u64)r, (u32)G, queries.get(), ask.data());
    std::vector<i64> asked = g.all_gather(ask);  // asked[src * G + member]
    u64 in_left = 0, in_right = 0;
    std::vector<i64> ask_b((size_t)G), asked_b((size_t)G);
    for (int j = 0; j < G; j++) {
      const i64 x = asked[(size_t)j * (size_t)G + (size_t)k];
      (j < q ? in_left : in_right) += (u64)x;
      asked_b[(size_t)j] = x * 4;
      ask_b[(size_t)j] = ask[(size_t)j] * 4;
    }
    const u64 in_all = in_left + in_right;
    DBuf<u32> q_in(c, in_all ? in_all : 1), a_out(c, in_all ? in_all : 1), a_in(c, mine ? mine : 1);
    g.all_to_all((const u8 *)queries.get(), ask_b, (u8 *)q_in.get(), asked_b);
    samples_answer(c, q_in.get(), in_left, in_right, S.get(), (u64)i_lo, (u64)rb[(size_t)k], a, gt.get(), a_out.get());
    g.all_to_all((const u8 *)a_out.get(), asked_b, (u8 *)a_in.get(), ask_b);
    const u64 first_slot = samples_taken_before(first_index, A, B);
    const u64 kept = samples_taken_before(first_index + mine, A, B) - first_slot;
    DBuf<u32> o_rank(c, kept ? kept : 1), o_pos(c, kept ? kept : 1);
    SamplePlace P;
    P.rank = blk->samp_rank.get();
    P.pos = blk->samp_pos.get();
    P.n = mine;
    P.ans = a_in.get();
    P.perm = k < q ? perm.get() : nullptr;
    P.right_shift = k < q ? 0u : (u32)m;
    P.first_index = first_index;
    P.first_slot = first_slot;
    P.A = A;
    P.B = B;
    P.o_rank = o_rank.get();
    P.o_pos = o_pos.get();
    samples_place(c, P);
    // the block's first sample is at position a (the first holder's first), its last at b - 1 (the last holder's last)
    if (k == 0 || k == G - 1) {
      if (kept == 0) fail(BWT_HIP_EINTERNAL, "merge [%ld,%ld): member %d kept no sample", (long)a, (long)b, k);
      const u64 at = k == 0 ? 0 : kept - 1;
      HIP_CHECK(hipMemcpyAsync(c.h_scalars, o_rank.get() + at, 4, hipMemcpyDeviceToHost, c.stream));
      HIP_CHECK(hipMemcpyAsync(c.h_scalars + 1, o_pos.get() + at, 4, hipMemcpyDeviceToHost, c.stream));
      c.sync();
      if ((i64)c.h_scalars[1] != (k == 0 ? a : b - 1))
        fail(BWT_HIP_EINTERNAL, "merge [%ld,%ld): the %s sample is at %u", (long)a, (long)b, k == 0 ? "first" : "last", c.h_scalars[1]);
      (k == 0 ? out->s_rank : out->e_rank) = (i64)c.h_scalars[0];
    }
    c.sync();  // the temporaries die here
    out->samp_rank = std::move(o_rank);
    out->samp_pos = std::move(o_pos);
    out->n_samp = (i64)kept;
  }
  c.sync();
  return out;
}

// out[y] = the filled one of in[0][y] .. in[sources-1][y] (~0 = empty: the minimum)
__global__ void k_merge_slots(const u64 *__restrict__ in, u64 n, u32 sources, u64 *__restrict__ out) {
  const u64 y = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (y >= n) return;
  u64 v = ~0ull;
  for (u32 s = 0; s < sources; s++) {
    const u64 x = in[(u64)s * n + y];
    v = x < v ? x : v;
  }
  out[y] = v;
}

// The whole text as ONE leaf over all ranks (a tree without merges: bwt_rec_threshold >= n).  The reference sorts a
// leaf in one address space (include/bwt.h:615-635, include/suffix_array.h:34-49); here every rank holds the text,
// builds the stretch of the leaf's suffix order whose keys begin with its share of first digits (build_leaf_share:
// the shares are cut the same way by every rank, from its own count of the whole text) and ends up with slice `rank`
// of L.  What is exchanged: the special ranks (two numbers per rank) and the sample slots -- a rank's share holds
// suffixes from all over the text, so every rank sends each other rank the slots of that rank's position range and
// keeps the filled ones of what it receives.
static std::unique_ptr<DistBlock> dist_leaf(Ctx &c, bwt_hip_transport *t, i64 ns_req) {
  const int world = t->world, me = t->rank;
  const i64 n = c.n;
  LeafShare sh;
  if (!build_leaf_share(c, 0, n, ns_req, me, world, sh))
    fail(BWT_HIP_ENOROUTE, "a leaf over %d GPUs needs a text that takes the lean bucket route (1e8 symbols and more, no long repeats): build this one as a tree of at least %d leaves", world, world);
  Group g{t, c, 0, world, me, me};
  std::unique_ptr<DistBlock> out(new DistBlock());
  out->a = 0;
  out->b = n;
  out->first = 0;
  out->q = world;
  for (int j = 0; j < world; j++) out->sizes.push_back((i64)(sh.bounds[(size_t)j + 1] - sh.bounds[(size_t)j]));
  out->L = std::move(sh.L);
  std::vector<i64> sp = g.all_gather({sh.s_rank, sh.e_rank});
  for (int j = 0; j < world; j++) {
    if (sp[(size_t)j * 2] >= 0) out->s_rank = sp[(size_t)j * 2];
    if (sp[(size_t)j * 2 + 1] >= 0) out->e_rank = sp[(size_t)j * 2 + 1];
  }
  if (out->s_rank < 0 || out->e_rank < 0) fail(BWT_HIP_EINTERNAL, "leaf over %d GPUs: no share holds the first or the last suffix", world);
  // sample slots: rank j gets the slots [ns * j / world, ns * (j + 1) / world) of every share
  const i64 ns = sh.ns;
  auto cut = [&](int j) { return ns * (i64)j / (i64)world; };
  const i64 my_lo = cut(me), my_cnt = cut(me + 1) - my_lo;
  DBuf<u64> in(c, (size_t)(my_cnt > 0 ? my_cnt : 1) * (size_t)world);
  std::vector<i64> send_b((size_t)world), recv_b((size_t)world, my_cnt * 8);
  for (int j = 0; j < world; j++) send_b[(size_t)j] = (cut(j + 1) - cut(j)) * 8;
  g.all_to_all((const u8 *)sh.samp.get(), send_b, (u8 *)in.get(), recv_b);
  DBuf<u64> merged(c, (size_t)(my_cnt > 0 ? my_cnt : 1));
  launch1d(c, k_merge_slots, (u64)my_cnt, (const u64 *)in.get(), (u64)my_cnt, (u32)world, merged.get());
  out->samp_rank = DBuf<u32>(c, (size_t)(my_cnt > 0 ? my_cnt : 1));
  out->samp_pos = DBuf<u32>(c, (size_t)(my_cnt > 0 ? my_cnt : 1));
  out->n_samp = compact_sample_slots(c, merged.get(), my_cnt, 0u, out->samp_rank.get(), out->samp_pos.get());
  c.sync();
  return out;
}

static std::unique_ptr<DistBlock> dist_build(Ctx &c, bwt_hip_transport *t, i64 th, i64 ns, i64 seg) {
  const int world = t->world, me = t->rank;
  if (world < 1 || (world & (world - 1)) || me < 0 || me >= world) fail(BWT_HIP_EINVAL, "rank %d of %d: the number of ranks must be a power of two", me, world);
  if (!c.T) fail(BWT_HIP_EINVAL, "no text set");
  if (world > 1 && (th >= c.n || c.n < 2)) {  // the tree is one leaf: the ranks share it by key range
    drop_dc_sample(c);
    return dist_leaf(c, t, ns);
  }
  int depth = 0;
  while ((1 << depth) < world) depth++;
  if (!tree_reaches(c.n, th, depth))
    fail(BWT_HIP_EINVAL, "bwt_rec_threshold %ld leaves fewer than %d subtrees at depth %d of a text of %ld symbols", (long)th, world, depth, (long)c.n);
  std::vector<std::pair<i64, i64>> ranges;
  split_ranges(0, c.n, depth, ranges);
  const i64 a = ranges[(size_t)me].first, b = ranges[(size_t)me].second;
  drop_dc_sample(c);  // a build of the whole text starts from the text alone
  std::unique_ptr<Block> own(build_rec(c, a, b, th, ns, seg));
  std::unique_ptr<DistBlock> blk(new DistBlock());
  blk->a = a;
  blk->b = b;
  blk->first = me;
  blk->q = 1;
  blk->sizes.push_back(b - a);
  blk->s_rank = a + own->s_rank;
  blk->e_rank = a + own->e_rank;
  blk->n_samp = own->n_samp;
  blk->samp_rank = DBuf<u32>(c, (size_t)(own->n_samp ? own->n_samp : 1));
  launch1d(c, k_abs_ranks, (u64)own->n_samp, (const u32 *)own->samp_rank.get(), (u64)own->n_samp, (u32)a, blk->samp_rank.get());
  blk->samp_pos = std::move(own->samp_pos);
  blk->L = std::move(own->L);
  c.sync();
  own.reset();
  for (int G = 2; G <= world; G *= 2) {
    Group g{t, c, (me / G) * G, G, me, me % G};
    blk = dist_merge(g, std::move(blk), ns, seg);
  }
  return blk;
}

// ---------------------------------------------------------------------------
// transport: ranks that are threads of one process
// ---------------------------------------------------------------------------
struct LocalHub {
  struct Post {
    const void *ptr;
    i64 bytes;
    int dev;
    bool done;
  };
  std::mutex mu;
  std::condition_variable cv;
  int world, refs;
  std::vector<std::deque<Post *>> box;  // box[src * world + dst]: what src has offered to dst, oldest first
  bool broken = false;                  // a member failed: nobody waits any longer
};

struct LocalEnd {
  bwt_hip_transport iface;
  LocalHub *hub;
  hipStream_t stream = nullptr;
  int stream_dev = -1;
};

static int local_exchange(void *self, int n_ops, const bwt_hip_p2p *ops) {
  LocalEnd *e = (LocalEnd *)self;
  LocalHub *h = e->hub;
  const int me = e->iface.rank;
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess) return BWT_HIP_EINTERNAL;
  if (!e->stream || e->stream_dev != dev) {
    if (e->stream) (void)hipStreamDestroy(e->stream);
    if (hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking) != hipSuccess) return BWT_HIP_EINTERNAL;
    e->stream_dev = dev;
  }
  std::vector<LocalHub::Post *> mine;
  int rc = 0;
  {
    std::unique_lock<std::mutex> lock(h->mu);
    // every transfer is checked before the first is posted; a bad list fails the whole group (the peers' matching
    // calls would otherwise wait for ever)
    for (int i = 0; i < n_ops; i++)
      if (ops[i].peer < 0 || ops[i].peer >= h->world || ops[i].peer == me || ops[i].bytes < 0 || (ops[i].bytes > 0 && !ops[i].buf)) {
        h->broken = true;
        h->cv.notify_all();
        return BWT_HIP_EINVAL;
      }
    for (int i = 0; i < n_ops; i++) {
      if (!ops[i].is_send) continue;
      LocalHub::Post *p = new LocalHub::Post{ops[i].buf, ops[i].bytes, dev, false};
      h->box[(size_t)me * (size_t)h->world + (size_t)ops[i].peer].push_back(p);
      mine.push_back(p);
    }
    h->cv.notify_all();
    for (int i = 0; i < n_ops && rc == 0; i++) {
      if (ops[i].is_send) continue;
      auto &q = h->box[(size_t)ops[i].peer * (size_t)h->world + (size_t)me];
      h->cv.wait(lock, [&] { return !q.empty() || h->broken; });
      if (h->broken) {
        rc = BWT_HIP_EINTERNAL;
        break;
      }
      LocalHub::Post *p = q.front();
      q.pop_front();
      if (p->bytes != ops[i].bytes) {
        rc = BWT_HIP_EINTERNAL;
      } else {
        lock.unlock();
        hipError_t he = p->dev == dev ? hipMemcpyAsync(ops[i].buf, p->ptr, (size_t)p->bytes, hipMemcpyDeviceToDevice, e->stream)
                                      : hipMemcpyPeerAsync(ops[i].buf, dev, p->ptr, p->dev, (size_t)p->bytes, e->stream);
        if (he == hipSuccess) he = hipStreamSynchronize(e->stream);
        lock.lock();
        if (he != hipSuccess) rc = (int)he;
      }
      p->done = true;
      h->cv.notify_all();
    }
    if (rc != 0) {
      h->broken = true;
      h->cv.notify_all();
    }
    h->cv.wait(lock, [&] {
      if (h->broken) return true;
      for (auto *p : mine)
        if (!p->done) return false;
      return true;
    });
    if (h->broken && rc == 0) rc = BWT_HIP_EINTERNAL;
    if (!h->broken)
      for (auto *p : mine) delete p;  // (after a failure the posts may still sit in a mailbox: leaked on purpose)
  }
  return rc;
}

static void local_destroy(void *self) {
  LocalEnd *e = (LocalEnd *)self;
  if (e->stream) (void)hipStreamDestroy(e->stream);
  LocalHub *h = e->hub;
  bool last;
  {
    std::lock_guard<std::mutex> lock(h->mu);
    last = --h->refs == 0;
  }
  if (last) delete h;
  delete e;
}

// ---------------------------------------------------------------------------
// transport: RCCL
// ---------------------------------------------------------------------------
struct Id128 {  // ncclUniqueId: 128 opaque bytes, passed by value
  char b[128];
};
struct RcclApi {
  void *lib = nullptr;
  int (*GetUniqueId)(void *) = nullptr;
  int (*CommInitRank)(void **, int, Id128, int) = nullptr;
  int (*CommDestroy)(void *) = nullptr;
  int (*Send)(const void *, size_t, int, int, void *, hipStream_t) = nullptr;
  int (*Recv)(void *, size_t, int, int, void *, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  const char *(*GetErrorString)(int) = nullptr;
};
static RcclApi &rccl() {
  static RcclApi api;
  static std::once_flag once;
  std::call_once(once, [] {
    // a copy the process already holds (for example the one PyTorch brought) is taken before a second one is loaded
    for (const char *name : {"librccl.so.1", "librccl.so"}) {
      api.lib = dlopen(name, RTLD_NOW | RTLD_NOLOAD);
      if (api.lib) break;
    }
    if (!api.lib)
      for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        api.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (api.lib) break;
      }
    if (!api.lib) return;
    auto sym = [&](const char *s) { return dlsym(api.lib, s); };
    api.GetUniqueId = (decltype(api.GetUniqueId))sym("ncclGetUniqueId");
    api.CommInitRank = (decltype(api.CommInitRank))sym("ncclCommInitRank");
    api.CommDestroy = (decltype(api.CommDestroy))sym("ncclCommDestroy");
    api.Send = (decltype(api.Send))sym("ncclSend");
    api.Recv = (decltype(api.Recv))sym("ncclRecv");
    api.GroupStart = (decltype(api.GroupStart))sym("ncclGroupStart");
    api.GroupEnd = (decltype(api.GroupEnd))sym("ncclGroupEnd");
    api.GetErrorString = (decltype(api.GetErrorString))sym("ncclGetErrorString");
  });
  if (!api.lib || !api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.Send || !api.Recv || !api.GroupStart || !api.GroupEnd) {
    const char *why = dlerror();  // (a second call would return null: the message is cleared by the first)
    fail(BWT_HIP_EINTERNAL, "librccl.so could not be loaded: %s", why ? why : "symbols missing");
  }
  return api;
}

static void rccl_ok(int rc, const char *what) {
  if (rc != 0) fail(BWT_HIP_EINTERNAL, "%s failed: %s", what, rccl().GetErrorString ? rccl().GetErrorString(rc) : "RCCL error");
}

struct RcclEnd {
  bwt_hip_transport iface;
  void *comm = nullptr;
  hipStream_t stream = nullptr;  // the context's stream: transfers queue behind the kernels that made the data
  int device = 0;
};

static int rccl_exchange(void *self, int n_ops, const bwt_hip_p2p *ops) {
  RcclEnd *e = (RcclEnd *)self;
  try {
    RcclApi &api = rccl();
    constexpr int NCCL_UINT8 = 1;  // ncclUint8 / ncclChar share the size; value of ncclUint8 in rccl.h
    rccl_ok(api.GroupStart(), "ncclGroupStart");
    for (int i = 0; i < n_ops; i++) {
      if (ops[i].is_send) rccl_ok(api.Send(ops[i].buf, (size_t)ops[i].bytes, NCCL_UINT8, ops[i].peer, e->comm, e->stream), "ncclSend");
      else rccl_ok(api.Recv(ops[i].buf, (size_t)ops[i].bytes, NCCL_UINT8, ops[i].peer, e->comm, e->stream), "ncclRecv");
    }
    rccl_ok(api.GroupEnd(), "ncclGroupEnd");
    HIP_CHECK(hipStreamSynchronize(e->stream));
  } catch (const Error &err) {
    fprintf(stderr, "bwt_hip RCCL transport: %s\n", err.what());
    return err.code ? err.code : BWT_HIP_EINTERNAL;
  }
  return 0;
}

static void rccl_destroy(void *self) {
  RcclEnd *e = (RcclEnd *)self;
  if (e->comm) (void)rccl().CommDestroy(e->comm);
  delete e;
}

struct Slice {
  std::unique_ptr<DistBlock> d;
  int rank = 0;
  i64 first = 0;
};

}  // namespace bwtamd

using namespace bwtamd;

// ---------------------------------------------------------------------------
// C-ABI
// ---------------------------------------------------------------------------
#define DIST_BEGIN try {
#define DIST_END                                   \
  }                                                \
  catch (const Error &e) {                         \
    set_last_error(e.what());             \
    return e.code ? e.code : BWT_HIP_EINTERNAL;    \
  }                                                \
  catch (const std::exception &e) {                \
    set_last_error(e.what());             \
    return BWT_HIP_EINTERNAL;                      \
  }                                                \
  return 0;

int bwt_hip_transport_local(int world, bwt_hip_transport **out) {
  DIST_BEGIN
  if (world < 1 || world > 256 || !out) fail(BWT_HIP_EINVAL, "invalid argument: 1 <= world <= 256, out");
  LocalHub *h = new LocalHub();
  h->world = world;
  h->refs = world;
  h->box.resize((size_t)world * (size_t)world);
  for (int k = 0; k < world; k++) {
    LocalEnd *e = new LocalEnd();
    e->hub = h;
    e->iface = bwt_hip_transport{e, k, world, local_exchange, local_destroy};
    out[k] = &e->iface;
  }
  DIST_END
}

int bwt_hip_rccl_unique_id(uint8_t id[128]) {
  DIST_BEGIN
  if (!id) fail(BWT_HIP_EINVAL, "invalid argument: id");
  rccl_ok(rccl().GetUniqueId(id), "ncclGetUniqueId");
  DIST_END
}

int bwt_hip_transport_rccl(bwt_hip_ctx *ctx, const uint8_t id[128], int rank, int world, bwt_hip_transport **out) {
  DIST_BEGIN
  if (!ctx || !id || !out || world < 1 || rank < 0 || rank >= world) fail(BWT_HIP_EINVAL, "invalid argument: ctx, id, out, 0 <= rank < world");
  Ctx &c = ctx->c;
  HIP_CHECK(hipSetDevice(c.device));
  std::unique_ptr<RcclEnd> e(new RcclEnd());
  Id128 uid;
  ::memcpy(uid.b, id, 128);
  rccl_ok(rccl().CommInitRank(&e->comm, world, uid, rank), "ncclCommInitRank");
  e->stream = c.stream;
  e->device = c.device;
  e->iface = bwt_hip_transport{e.get(), rank, world, rccl_exchange, rccl_destroy};
  *out = &e.release()->iface;
  DIST_END
}

void bwt_hip_transport_free(bwt_hip_transport *t) {
  if (t && t->destroy) t->destroy(t->self);
}

int bwt_hip_dist_pmbwt(bwt_hip_ctx *ctx, bwt_hip_transport *t, int64_t th, int64_t ns, int64_t seg, bwt_hip_slice **out) {
  DIST_BEGIN
  if (!ctx || !t || !out || !t->exchange) fail(BWT_HIP_EINVAL, "invalid argument: ctx, transport, out");
  if (th < 1) fail(BWT_HIP_EINVAL, "invalid argument: bwt_rec_threshold >= 1");
  Ctx &c = ctx->c;
  HIP_CHECK(hipSetDevice(c.device));
  std::unique_ptr<Slice> s(new Slice());
  s->d = dist_build(c, t, th, ns, seg);
  s->rank = t->rank;
  i64 first = 0;
  for (int j = 0; j < t->rank; j++) first += s->d->sizes[(size_t)j];
  s->first = first;
  *out = reinterpret_cast<bwt_hip_slice *>(s.release());
  DIST_END
}

int bwt_hip_slice_info(const bwt_hip_slice *s_, int64_t *first, int64_t *size, int64_t *s_rank, int64_t *e_rank, int64_t *n_samples) {
  DIST_BEGIN
  if (!s_) fail(BWT_HIP_EINVAL, "invalid argument: slice");
  const Slice *s = reinterpret_cast<const Slice *>(s_);
  if (first) *first = s->first;
  if (size) *size = s->d->sizes[(size_t)s->rank];
  if (s_rank) *s_rank = s->rank == 0 ? s->d->s_rank : -1;
  if (e_rank) *e_rank = s->rank == s->d->first + s->d->q - 1 ? s->d->e_rank : -1;
  if (n_samples) *n_samples = s->d->n_samp;
  DIST_END
}

int bwt_hip_slice_L_device(const bwt_hip_slice *s_, const uint8_t **L_dev) {
  DIST_BEGIN
  if (!s_ || !L_dev) fail(BWT_HIP_EINVAL, "invalid argument: slice, L_dev");
  *L_dev = reinterpret_cast<const Slice *>(s_)->d->L.get();
  DIST_END
}

int bwt_hip_slice_get_L(bwt_hip_ctx *ctx, const bwt_hip_slice *s_, uint8_t *L_host) {
  DIST_BEGIN
  if (!ctx || !s_ || !L_host) fail(BWT_HIP_EINVAL, "invalid argument: ctx, slice, L_host");
  const Slice *s = reinterpret_cast<const Slice *>(s_);
  const i64 size = s->d->sizes[(size_t)s->rank];
  HIP_CHECK(hipSetDevice(ctx->c.device));
  if (size > 0) symbols_to_host(ctx->c, s->d->L.get(), (u64)size, L_host);
  DIST_END
}

int bwt_hip_slice_get_samples(bwt_hip_ctx *ctx, const bwt_hip_slice *s_, int64_t *rank_out, int64_t *pos_out, int64_t cap) {
  DIST_BEGIN
  if (!ctx || !s_ || !rank_out || !pos_out) fail(BWT_HIP_EINVAL, "invalid argument: ctx, slice, rank_out, pos_out");
  const Slice *s = reinterpret_cast<const Slice *>(s_);
  Ctx &c = ctx->c;
  HIP_CHECK(hipSetDevice(c.device));
  const i64 k = s->d->n_samp < cap ? s->d->n_samp : cap;
  if (k > 0) {
    DBuf<i64> wr(c, (size_t)k), wp(c, (size_t)k);
    launch1d(c, k_u32_to_i64, (u64)k, (const u32 *)s->d->samp_rank.get(), (u64)k, wr.get());
    launch1d(c, k_u32_to_i64, (u64)k, (const u32 *)s->d->samp_pos.get(), (u64)k, wp.get());
    HIP_CHECK(hipMemcpyAsync(rank_out, wr.get(), (size_t)k * 8, hipMemcpyDeviceToHost, c.stream));
    HIP_CHECK(hipMemcpyAsync(pos_out, wp.get(), (size_t)k * 8, hipMemcpyDeviceToHost, c.stream));
    c.sync();
  }
  DIST_END
}

void bwt_hip_slice_free(bwt_hip_slice *s) { delete reinterpret_cast<Slice *>(s); }

static void pmbwt_multi_impl(const int *devices, int n_devices, const uint8_t *T_host, int64_t n, int64_t th, int64_t ns,
                             int64_t seg, uint8_t *L_host, int64_t *s_rank, int64_t *e_rank, int64_t *samp_rank,
                             int64_t *samp_pos, int64_t samp_cap, int64_t *n_samples, int *used, bool one_device_only) {
  {
  if (!devices || n_devices < 1 || !T_host || !L_host || n < 1 || th < 1) fail(BWT_HIP_EINVAL, "invalid argument: devices, T_host, L_host, n >= 1, bwt_rec_threshold >= 1");
  int world = 1, depth = 0;
  while (world * 2 <= n_devices && tree_reaches(n, th, depth + 1)) {
    world *= 2;
    depth++;
  }
  // a tree of one leaf: the GPUs share the leaf by key range (dist_leaf), where the text takes the route that needs
  const bool one_leaf = world == 1 && n_devices >= 2 && th >= n && !one_device_only;
  if (one_leaf)
    while (world * 2 <= n_devices) world *= 2;
  if (used) *used = world;
  std::vector<bwt_hip_transport *> tr((size_t)world, nullptr);
  int rc0 = bwt_hip_transport_local(world, tr.data());
  if (rc0 != 0) fail(rc0, "%s", bwt_hip_last_error());
  std::vector<int> rcs((size_t)world, 0);
  std::vector<std::string> msgs((size_t)world);
  std::vector<bwt_hip_ctx *> ctxs((size_t)world, nullptr);
  std::vector<bwt_hip_slice *> slices((size_t)world, nullptr);
  auto body = [&](int k) {
    bwt_hip_ctx *&ctx = ctxs[(size_t)k];
    bwt_hip_slice *&sl = slices[(size_t)k];
    int rc = bwt_hip_create(devices[k], &ctx);
    if (rc == 0) rc = bwt_hip_set_text(ctx, T_host, n);
    // a rank that cannot start still has to meet the others: without its transfers they would wait for ever;
    // the in-process transport wakes them when one member reports a failure
    if (rc == 0) rc = bwt_hip_dist_pmbwt(ctx, tr[(size_t)k], th, ns, seg, &sl);
    if (rc != 0) {
      LocalEnd *e = (LocalEnd *)tr[(size_t)k]->self;
      std::lock_guard<std::mutex> lock(e->hub->mu);
      e->hub->broken = true;
      e->hub->cv.notify_all();
    }
    if (rc == 0) {
      int64_t first = 0;
      rc = bwt_hip_slice_info(sl, &first, nullptr, nullptr, nullptr, nullptr);
      if (rc == 0) rc = bwt_hip_slice_get_L(ctx, sl, L_host + first);
    }
    if (rc != 0) msgs[(size_t)k] = bwt_hip_last_error();
    rcs[(size_t)k] = rc;
  };
  std::vector<std::thread> th_;
  for (int k = 1; k < world; k++) th_.push_back(std::thread(body, k));
  body(0);
  for (auto &x : th_) x.join();
  // every rank holds the samples of its own text range: they follow each other in the caller's arrays
  bool ok = true;
  for (int k = 0; k < world; k++) ok = ok && rcs[(size_t)k] == 0;
  int64_t at = 0;
  for (int k = 0; k < world && ok; k++) {
    int64_t sr = -1, er = -1, nsmp = 0;
    int rc = bwt_hip_slice_info(slices[(size_t)k], nullptr, nullptr, &sr, &er, &nsmp);
    if (rc == 0 && k == 0 && s_rank) *s_rank = sr;
    if (rc == 0 && k == world - 1 && e_rank) *e_rank = er;
    if (rc == 0 && samp_rank && samp_pos && at < samp_cap)
      rc = bwt_hip_slice_get_samples(ctxs[(size_t)k], slices[(size_t)k], samp_rank + at, samp_pos + at, samp_cap - at);
    at += nsmp;
    if (rc != 0) {
      msgs[(size_t)k] = bwt_hip_last_error();
      rcs[(size_t)k] = rc;
      ok = false;
    }
  }
  if (n_samples) *n_samples = at;
  for (int k = 0; k < world; k++) {
    bwt_hip_slice_free(slices[(size_t)k]);
    bwt_hip_transport_free(tr[(size_t)k]);
    if (ctxs[(size_t)k]) bwt_hip_destroy(ctxs[(size_t)k]);
  }
  for (int k = 0; k < world; k++)
    if (rcs[(size_t)k] != 0) fail(rcs[(size_t)k], "rank %d: %s", k, msgs[(size_t)k].c_str());
  }
}

int bwt_hip_pmbwt_multi(const int *devices, int n_devices, const uint8_t *T_host, int64_t n, int64_t th, int64_t ns,
                        int64_t seg, uint8_t *L_host, int64_t *s_rank, int64_t *e_rank, int64_t *samp_rank,
                        int64_t *samp_pos, int64_t samp_cap, int64_t *n_samples, int *used) {
  DIST_BEGIN
  try {
    pmbwt_multi_impl(devices, n_devices, T_host, n, th, ns, seg, L_host, s_rank, e_rank, samp_rank, samp_pos, samp_cap, n_samples, used, false);
  } catch (const Error &e) {
    if (e.code != BWT_HIP_ENOROUTE) throw;
    // a tree of one leaf whose text does not take the route that GPUs can share: one GPU builds it
    pmbwt_multi_impl(devices, n_devices, T_host, n, th, ns, seg, L_host, s_rank, e_rank, samp_rank, samp_pos, samp_cap, n_samples, used, true);
  }
  DIST_END
}
