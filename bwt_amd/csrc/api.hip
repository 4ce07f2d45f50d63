// The extern "C" boundary (declared in include/bwt_hip.h).  Thin: argument
// checks, exception -> status conversion, host<->device staging.
#include <string.h>

#include <chrono>
#include <memory>
#include <thread>

#include "../../include/bwt_hip.h"
#include "internal.h"

using namespace bwtamd;

struct bwt_hip_block {
  Block b;
};

static thread_local std::string g_last_error;

const char *bwt_hip_last_error(void) { return g_last_error.c_str(); }
namespace bwtamd {
void set_last_error(const char *msg) { g_last_error = msg; }
Block *build_rec(Ctx &c, i64 a, i64 b, i64 th, i64 ns, i64 seg);
}

#define API_BEGIN try {
#define API_END                                  \
  }                                              \
  catch (const Error &e) {                       \
    g_last_error = e.what();                     \
    return e.code ? e.code : BWT_HIP_EINTERNAL;  \
  }                                              \
  catch (const std::exception &e) {              \
    g_last_error = e.what();                     \
    return BWT_HIP_EINTERNAL;                    \
  }                                              \
  return 0;

static void need(bool cond, const char *what) {
  if (!cond) fail(BWT_HIP_EINVAL, "invalid argument: %s", what);
}

// Every block handed out through the C-ABI, with its context: destroying a context frees the blocks that are still
// alive in it, and freeing a handle that is no longer registered (its context went first) does nothing.
#include <mutex>
#include <set>
static std::mutex g_blocks_mu;
static std::map<Block *, Ctx *> g_blocks;

static Block *as_block(bwt_hip_block *b) { return reinterpret_cast<Block *>(b); }
static const Block *as_block(const bwt_hip_block *b) { return reinterpret_cast<const Block *>(b); }
static bwt_hip_block *as_handle(Block *b) { return reinterpret_cast<bwt_hip_block *>(b); }

static bwt_hip_block *publish(Ctx &c, Block *b) {
  std::lock_guard<std::mutex> lock(g_blocks_mu);
  g_blocks[b] = &c;
  return as_handle(b);
}
// a block that a call consumes (merge) leaves the registry with the call
static Block *take(bwt_hip_block *h) {
  Block *b = as_block(h);
  std::lock_guard<std::mutex> lock(g_blocks_mu);
  if (g_blocks.erase(b) == 0) fail(BWT_HIP_EINVAL, "block handle %p is not alive (freed, merged, or its context was destroyed)", (void *)h);
  return b;
}

// ---------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------
int bwt_hip_create(int device, bwt_hip_ctx **out) {
  API_BEGIN
  need(out != nullptr, "out");
  int count = 0;
  HIP_CHECK(hipGetDeviceCount(&count));
  if (device < 0 || device >= count) fail(BWT_HIP_EINVAL, "device %d out of range (%d visible)", device, count);
  HIP_CHECK(hipSetDevice(device));
  std::unique_ptr<bwt_hip_ctx> h(new bwt_hip_ctx());
  Ctx &c = h->c;
  c.device = device;
  hipDeviceProp_t prop;
  HIP_CHECK(hipGetDeviceProperties(&prop, device));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    fail(BWT_HIP_EINVAL, "device %d is %s; this library is built for gfx950 only", device, prop.gcnArchName);
  c.n_cu = prop.multiProcessorCount;
  HIP_CHECK(hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking));
  HIP_CHECK(hipHostMalloc((void **)&c.h_scalars, 64 * sizeof(u32), hipHostMallocDefault));
  HIP_CHECK(hipMalloc((void **)&c.d_scalars, 64 * sizeof(u32)));
  if (const char *v = getenv("BWT_HIP_BUCKET_MIN_N")) c.bucket_min_n = strtoull(v, nullptr, 10);  // tuning and tests
  if (const char *v = getenv("BWT_HIP_FORCE_WIDE")) c.force_wide = atoi(v);
  if (const char *v = getenv("BWT_HIP_WIDE_BBITS")) c.wide_bbits = atoi(v);
  if (const char *v = getenv("BWT_HIP_LEAN_ANY")) c.lean_any_order = atoi(v) != 0;
  if (const char *v = getenv("BWT_HIP_LEAN_SLOTS")) c.lean_slots = atoi(v);
  if (const char *v = getenv("BWT_HIP_SLOT_SUB")) {
    const int k = atoi(v);
    c.slot_sub = (u32)(k < 1 ? 1 : (k > 16 ? 16 : k));
  }
  if (const char *v = getenv("BWT_HIP_GAP_LEAN")) c.gap_lean = atoi(v) != 0;
  if (const char *v = getenv("BWT_HIP_HOST_PACK")) c.host_pack = atoi(v);
  if (const char *v = getenv("BWT_HIP_XFER_CHUNK")) c.xfer_chunk = atoll(v);
  if (const char *v = getenv("BWT_HIP_DEFER_CHAINS")) c.defer_chains = atoi(v) != 0;
  if (const char *v = getenv("BWT_HIP_REPEAT_PROBE")) c.repeat_probe = atoi(v) != 0;
  if (const char *v = getenv("BWT_HIP_RUN_FILTER")) c.run_filter = atoi(v) != 0;
  if (const char *v = getenv("BWT_HIP_COMPACT")) c.compact_allowed = atoi(v) != 0;
  if (const char *v = getenv("BWT_HIP_DEFER_BUCKETS")) c.defer_buckets = atoi(v) != 0;
  if (const char *v = getenv("BWT_HIP_RUN_TABLE")) c.run_table = atoi(v) != 0;
  if (const char *v = getenv("BWT_HIP_RUN_JUMP")) c.run_jump = atoi(v);
  if (const char *v = getenv("BWT_HIP_DC")) c.dc_mode = atoi(v);
  if (const char *v = getenv("BWT_HIP_LEAN_BASE")) c.lean_base = atoi(v) != 0;
  (void)device_scratch(device, nullptr);  // this GPU's address of the static scratch array, looked up while it is current
  *out = h.release();
  API_END
}

void bwt_hip_destroy(bwt_hip_ctx *ctx) {
  if (!ctx) return;
  Ctx &c = ctx->c;
  (void)hipSetDevice(c.device);
  (void)hipStreamSynchronize(c.stream);
  {
    std::vector<Block *> mine;
    std::lock_guard<std::mutex> lock(g_blocks_mu);
    for (auto it = g_blocks.begin(); it != g_blocks.end();) {
      if (it->second == &c) {
        mine.push_back(it->first);
        it = g_blocks.erase(it);
      } else {
        ++it;
      }
    }
    for (Block *b : mine) delete b;
  }
  for (auto &p : c.pending) {
    (void)hipEventDestroy(p.e0);
    (void)hipEventDestroy(p.e1);
  }
  for (auto &ch : c.chunks) (void)hipEventDestroy(ch.ev);
  for (auto e : c.chunk_events) (void)hipEventDestroy(e);
  for (int k = 0; k < 3; k++) {
    if (c.xfer_stage[k]) (void)hipHostFree(c.xfer_stage[k]);
    if (c.xfer_ev[k]) (void)hipEventDestroy(c.xfer_ev[k]);
  }
  if (c.copy_stream) {
    (void)hipStreamSynchronize(c.copy_stream);
    (void)hipStreamDestroy(c.copy_stream);
  }
  for (auto e : c.event_pool) (void)hipEventDestroy(e);
  for (auto &kv : c.free_blocks) (void)hipFree(kv.second);
  for (auto &kv : c.live_blocks) (void)hipFree(kv.first);
  if (c.h_scalars) (void)hipHostFree(c.h_scalars);
  if (c.d_scalars) (void)hipFree(c.d_scalars);
  if (c.d_sym_tab) (void)hipFree(c.d_sym_tab);
  (void)hipStreamDestroy(c.stream);
  delete ctx;
}

int bwt_hip_host_alloc(int64_t bytes, void **out) {
  API_BEGIN
  need(out != nullptr && bytes > 0, "out, bytes > 0");
  HIP_CHECK(hipHostMalloc(out, (size_t)bytes, hipHostMallocDefault));
  API_END
}

void bwt_hip_host_free(void *p) {
  if (p) (void)hipHostFree(p);
}

static void drop_text(Ctx &c) {
  drop_dc_sample(c);
  if (c.runs) c.release(c.runs);
  c.runs = nullptr;
  c.n_runs = 0;
  c.runs_host.clear();
  c.runs_valid = false;
  if (c.T_owned) c.release(c.T_owned);
  c.T_owned = nullptr;
  c.T = nullptr;
  c.n = 0;
  c.compact = false;
}

static void adopt_text(Ctx &c, const u8 *dT, i64 n) {
  int max_sym = 0, min_sym = 0, last = 0;
  i64 n_zero = 0;
  text_stats(c, dT, n, &max_sym, &min_sym, &n_zero, &last);
  if (last != 0) fail(BWT_HIP_ETEXT, "text must end with a 0 byte (T[n-1] = %d)", last);
  if (n_zero != 1) fail(BWT_HIP_ETEXT, "text must hold exactly one 0 byte, found %ld", (long)n_zero);
  c.compact = false;
  // Symbols scattered over the byte range (Ctx::compact): renumbered 1 .. sigma in their order, on the device copy.
  // (Not for alphabets of up to four values wide -- nothing to gain -- and not where every value of the range occurs.)
  if (c.compact_allowed && max_sym - min_sym >= 4) {
    u32 present[8];
    symbol_presence(c, dT, n, present);
    int sigma = 0;
    memset(c.sym_fwd, 0, sizeof(c.sym_fwd));
    memset(c.sym_back, 0, sizeof(c.sym_back));
    for (int v = 1; v < 256; v++)
      if ((present[v >> 5] >> (v & 31)) & 1u) {
        sigma++;
        c.sym_fwd[v] = (u8)sigma;
        c.sym_back[sigma] = (u8)v;
      }
    if (sigma < max_sym - min_sym + 1) {
      if (!c.d_sym_tab) HIP_CHECK(hipMalloc((void **)&c.d_sym_tab, 512));
      u8 both[512];
      memcpy(both, c.sym_fwd, 256);
      memcpy(both + 256, c.sym_back, 256);
      HIP_CHECK(hipMemcpyAsync(c.d_sym_tab, both, 512, hipMemcpyHostToDevice, c.stream));
      c.sync();  // `both` is on the stack
      if (dT != c.T_owned) {  // the caller's memory stays as it is: a copy of the text, in codes
        if (c.T_owned) c.release(c.T_owned);
        c.T_owned = (u8 *)c.alloc((size_t)n + 64);
        HIP_CHECK(hipMemsetAsync(c.T_owned + n, 0, 64, c.stream));
      }
      map_symbols(c, dT, c.T_owned, (u64)n, c.d_sym_tab);
      dT = c.T_owned;
      c.compact = true;
      min_sym = 1;
      max_sym = sigma;
    }
  }
  c.T = dT;
  c.n = n;
  c.max_sym = max_sym;
  c.min_sym = min_sym;
  int bits = 1;
  while ((max_sym >> bits) != 0) bits++;
  c.sym_bits = bits;
}

static void check_n(i64 n) {
  if (n < 1) fail(BWT_HIP_EINVAL, "n must be at least 1");
  // Device indices are 32-bit words, and kernels index whole tiles (up to 16384 positions) with them: the last tile of a
  // text that ends closer than that to 2^32 would wrap (seen in the rank structure's build at n = 2^32 - 16).
  if ((u64)n > 0xFFFFFFFFull - 0xFFFFull) fail(BWT_HIP_EINVAL, "n = %ld exceeds 2^32 - 65536", (long)n);
}

int bwt_hip_set_text(bwt_hip_ctx *ctx, const uint8_t *T_host, int64_t n) {
  API_BEGIN
  need(ctx && T_host, "ctx, T_host");
  check_n(n);
  Ctx &c = ctx->c;
  HIP_CHECK(hipSetDevice(c.device));
  drop_text(c);
  c.T_owned = (u8 *)c.alloc((size_t)n + 64);
  HIP_CHECK(hipMemsetAsync(c.T_owned + n, 0, 64, c.stream));
  HIP_CHECK(hipMemcpyAsync(c.T_owned, T_host, (size_t)n, hipMemcpyHostToDevice, c.stream));
  c.sync();
  try {
    adopt_text(c, c.T_owned, n);
  } catch (...) {
    drop_text(c);
    throw;
  }
  API_END
}

int bwt_hip_set_text_device(bwt_hip_ctx *ctx, const uint8_t *T_dev, int64_t n) {
  API_BEGIN
  need(ctx && T_dev, "ctx, T_dev");
  need(((uintptr_t)T_dev & 15) == 0, "T_dev must be 16-byte aligned");
  check_n(n);
  Ctx &c = ctx->c;
  HIP_CHECK(hipSetDevice(c.device));
  drop_text(c);
  adopt_text(c, T_dev, n);
  API_END
}

int bwt_hip_gen_text(bwt_hip_ctx *ctx, int64_t n, uint64_t seed, int sigma) {
  API_BEGIN
  need(ctx != nullptr, "ctx");
  need(sigma >= 1 && sigma <= 255, "sigma in 1..255");
  check_n(n);
  Ctx &c = ctx->c;
  HIP_CHECK(hipSetDevice(c.device));
  drop_text(c);
  c.T_owned = (u8 *)c.alloc((size_t)n + 64);
  HIP_CHECK(hipMemsetAsync(c.T_owned + n, 0, 64, c.stream));
  device_gen_text(c, c.T_owned, n, seed, sigma);
  try {
    adopt_text(c, c.T_owned, n);
  } catch (...) {
    drop_text(c);
    throw;
  }
  API_END
}

int bwt_hip_get_text(bwt_hip_ctx *ctx, uint8_t *T_host) {
  API_BEGIN
  need(ctx && T_host, "ctx, T_host");
  Ctx &c = ctx->c;
  need(c.T != nullptr, "no text set");
  HIP_CHECK(hipSetDevice(c.device));
  symbols_to_host(c, c.T, (u64)c.n, T_host);
  API_END
}

int bwt_hip_symbol_codes(bwt_hip_ctx *ctx, int *compact, uint8_t code_of[256], uint8_t symbol_of[256]) {
  API_BEGIN
  need(ctx != nullptr, "ctx");
  Ctx &c = ctx->c;
  need(c.T != nullptr, "no text set");
  if (compact) *compact = c.compact ? 1 : 0;
  for (int v = 0; v < 256; v++) {
    if (code_of) code_of[v] = c.compact ? c.sym_fwd[v] : (u8)((v >= c.min_sym && v <= c.max_sym && v > 0) ? v : 0);
    if (symbol_of) symbol_of[v] = c.compact ? c.sym_back[v] : (u8)((v >= c.min_sym && v <= c.max_sym && v > 0) ? v : 0);
  }
  API_END
}

// ---------------------------------------------------------------------------
// pipeline
// ---------------------------------------------------------------------------
static Block *do_leaf(Ctx &c, i64 a, i64 b, i64 ns) { return build_leaf(c, a, b, ns); }

static Block *do_merge(Ctx &c, Block *l, Block *r, i64 ns, i64 seg) {
  std::unique_ptr<Block> L(l), R(r);
  GapResult gap = build_gap(c, *l, *r, seg, false);
  return merge_blocks(c, *l, *r, gap, ns);
}

static Block *do_rec(Ctx &c, i64 a, i64 b, i64 th, i64 ns, i64 seg) {
  if (b - a <= th || b - a < 2) return do_leaf(c, a, b, ns);
  i64 mid = (a + b) / 2;
  std::unique_ptr<Block> l(do_rec(c, a, mid, th, ns, seg));
  std::unique_ptr<Block> r(do_rec(c, mid, b, th, ns, seg));
  return do_merge(c, l.release(), r.release(), ns, seg);
}
Block *bwtamd::build_rec(Ctx &c, i64 a, i64 b, i64 th, i64 ns, i64 seg) { return do_rec(c, a, b, th, ns, seg); }

int bwt_hip_leaf(bwt_hip_ctx *ctx, int64_t a, int64_t b, int64_t ns, bwt_hip_block **out) {
  API_BEGIN
  need(ctx && out, "ctx, out");
  HIP_CHECK(hipSetDevice(ctx->c.device));
  *out = publish(ctx->c, do_leaf(ctx->c, a, b, ns));
  API_END
}

int bwt_hip_merge(bwt_hip_ctx *ctx, bwt_hip_block *l, bwt_hip_block *r, int64_t ns, int64_t seg,
                  bwt_hip_block **out) {
  API_BEGIN
  need(ctx && l && r && out, "ctx, l, r, out");
  HIP_CHECK(hipSetDevice(ctx->c.device));
  need(l != r, "l and r are the same block");
  Block *lb = take(l), *rb = nullptr;
  try {
    rb = take(r);
  } catch (...) {
    publish(ctx->c, lb);
    throw;
  }
  *out = publish(ctx->c, do_merge(ctx->c, lb, rb, ns, seg));
  API_END
}

int bwt_hip_rec(bwt_hip_ctx *ctx, int64_t a, int64_t b, int64_t th, int64_t ns, int64_t seg, bwt_hip_block **out) {
  API_BEGIN
  need(ctx && out, "ctx, out");
  need(ctx->c.T != nullptr, "no text set");
  need(a >= 0 && a < b && b <= ctx->c.n, "0 <= a < b <= n");
  need(th >= 1, "bwt_rec_threshold >= 1");
  HIP_CHECK(hipSetDevice(ctx->c.device));
  *out = publish(ctx->c, do_rec(ctx->c, a, b, th, ns, seg));
  API_END
}

int bwt_hip_pmbwt(bwt_hip_ctx *ctx, int64_t th, int64_t ns, int64_t seg, bwt_hip_block **out) {
  if (!ctx) {
    g_last_error = "invalid argument: ctx";
    return BWT_HIP_EINVAL;
  }
  // a build of the whole text starts from the text alone: what an earlier build derived from it (the ranks of its
  // difference-cover sample) is made again if this one needs it
  drop_dc_sample(ctx->c);
  return bwt_hip_rec(ctx, 0, ctx->c.n, th, ns, seg, out);
}

int bwt_hip_leaf_capacity(bwt_hip_ctx *ctx, int64_t n, int64_t *max_leaf) {
  API_BEGIN
  need(ctx && max_leaf && n >= 1, "ctx, max_leaf, n >= 1");
  Ctx &c = ctx->c;
  HIP_CHECK(hipSetDevice(c.device));
  size_t free_b = 0, total_b = 0;
  HIP_CHECK(hipMemGetInfo(&free_b, &total_b));
  // what the pool already holds from the driver is free to a build as well
  const double avail = 0.9 * ((double)free_b + (double)c.bytes_reserved - (double)c.bytes_live);
  // measured peaks (tools/leaf_memory.py, 3.1e9 symbols of DNA): 19.6 bytes per symbol for one leaf of the whole text
  // (its two element buffers are 6 and 12 % larger than the elements: radix_sort.hip, slots), 11.1 for two leaves
  // and their merge, 10.5 for eight leaves (the top merge)
  const double leaf = (avail - 3.0 * (double)n) / 18.5;
  *max_leaf = avail < 10.7 * (double)n || leaf < 1 ? 0 : (leaf > (double)n ? n : (int64_t)leaf);
  API_END
}

// max symbol, smallest non-zero symbol and number of zeros of a host text, on a few host threads
// (*distinct, if not null: how many different non-zero symbols)
static void host_text_stats(const u8 *T, i64 n, int *max_sym, int *min_sym, i64 *n_zero, int max_threads, int *distinct = nullptr) {
  unsigned hw = std::thread::hardware_concurrency();
  const int nt = n < (1 << 22) ? 1 : (hw > (unsigned)max_threads ? max_threads : (hw < 1 ? 1 : (int)hw));
  std::vector<u32> mx((size_t)nt, 0), mn((size_t)nt, 255);
  std::vector<i64> nz((size_t)nt, 0);
  std::vector<u8> seen((size_t)nt * 256, 0);
  auto body = [&](int t) {
    const i64 lo = n * t / nt, hi = n * (t + 1) / nt;
    u8 m1 = 0, m0 = 255;  // m0: the smallest symbol counting 0 as 255 (v - 1 wraps)
    i64 z = 0;
    u8 *mine = seen.data() + (size_t)t * 256;
    for (i64 i = lo; i < hi; i++) {
      const u8 v = T[i], w = (u8)(v - 1);
      m1 = v > m1 ? v : m1;
      m0 = w < m0 ? w : m0;
      z += (v == 0);
      mine[v] = 1;
    }
    mx[(size_t)t] = m1;
    mn[(size_t)t] = m0;  // smallest (v - 1) over v != 0, or 255 if a 0 ... see below
    nz[(size_t)t] = z;
  };
  std::vector<std::thread> th;
  for (int t = 1; t < nt; t++) th.push_back(std::thread(body, t));
  body(0);
  for (auto &x : th) x.join();
  u32 M = 0, W = 255;
  i64 Z = 0;
  for (int t = 0; t < nt; t++) {
    M = mx[(size_t)t] > M ? mx[(size_t)t] : M;
    W = mn[(size_t)t] < W ? mn[(size_t)t] : W;  // min over all symbols of (v - 1) mod 256: zeros map to 255 and never win
    Z += nz[(size_t)t];
  }
  *max_sym = (int)M;
  *min_sym = M ? (int)W + 1 : 0;
  *n_zero = Z;
  if (distinct) {
    *distinct = 0;
    for (int v = 1; v < 256; v++) {
      bool any = false;
      for (int t = 0; t < nt; t++) any = any || seen[(size_t)t * 256 + (size_t)v];
      *distinct += any;
    }
  }
}

int bwt_hip_pmbwt_host(bwt_hip_ctx *ctx, const uint8_t *T_host, int64_t n, int64_t th, int64_t ns, int64_t seg,
                       uint8_t *L_host, bwt_hip_block **out) {
  API_BEGIN
  need(ctx && T_host && out, "ctx, T_host, out");
  need(th >= 1, "bwt_rec_threshold >= 1");
  check_n(n);
  Ctx &c = ctx->c;
  HIP_CHECK(hipSetDevice(c.device));
  drop_text(c);
  if (!c.copy_stream) HIP_CHECK(hipStreamCreateWithFlags(&c.copy_stream, hipStreamNonBlocking));
  if (T_host[n - 1] != 0) fail(BWT_HIP_ETEXT, "text must end with a 0 byte (T[n-1] = %d)", (int)T_host[n - 1]);
  c.T_owned = (u8 *)c.alloc((size_t)n + 64);
  // The terminator is in place before anything else: the leaf that starts at 0 reads T[n-1] as the symbol before
  // position 0, long before the last chunk of the text arrives (which writes the same 0 again).
  HIP_CHECK(hipMemsetAsync(c.T_owned + n - 1, 0, 65, c.stream));
  // BWT_HIP_POISON_TEXT=1 (tests): the text's memory starts as a byte no test text holds, so that a kernel which
  // reads a chunk before it has arrived shows in the result instead of finding a previous call's copy of the text
  if (n > 1 && getenv("BWT_HIP_POISON_TEXT") && atoi(getenv("BWT_HIP_POISON_TEXT")) != 0)
    HIP_CHECK(hipMemsetAsync(c.T_owned, 0xA5, (size_t)n - 1, c.stream));
  c.sync();  // whatever used this memory before is done: the copy stream may write it
  // Two ways up.  Packed (host_transfer.hip): a text of at most four consecutive symbol values, judged on a sample,
  // goes up as two bits per symbol from a thread of its own, which also checks every symbol against the sample's
  // range, so that the build starts without a pass over the text; a symbol outside that range makes wait_text
  // throw TextMismatch before any kernel has read the chunk that holds it, and the call starts over the other way.
  // Raw: the symbol statistics from a pass over the host text, first chunk = the leftmost leaf's window, then pieces
  // of 256 MB, all enqueued here.
  // BWT_HIP_XFER_TRACE=1: the host's view of this call's steps, in ms since its start, on stderr
  const bool trace = getenv("BWT_HIP_XFER_TRACE") != nullptr && atoi(getenv("BWT_HIP_XFER_TRACE")) != 0;
  const auto t_start = std::chrono::steady_clock::now();
  auto mark = [&](const char *what) {
    if (trace) fprintf(stderr, "[pmbwt_host] %8.2f ms  %s\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count(), what);
  };
  int plo = 0, phi = 0;
  const bool pack_size = (c.host_pack == 1 && n >= ((i64)1 << 24)) || (c.host_pack >= 2 && n >= 2);
  if (pack_size) probe_symbol_range(T_host, n, &plo, &phi);
  const bool try_packed = pack_size && plo >= 1 && plo <= 252 && phi - plo <= 3;
  i64 b0 = n;
  while (b0 > th && b0 >= 2) b0 = b0 / 2;
  i64 first = b0 + (b0 / 256 > 4096 ? b0 / 256 : 4096) + 4096;
  if (first > n) first = n;
  std::thread uploader;
  std::string upload_err;
  auto drain = [&]() {
    c.xfer_abort.store(true);
    if (uploader.joinable()) uploader.join();
    c.xfer_abort.store(false);
    c.xfer_mismatch.store(false);
    (void)hipStreamSynchronize(c.copy_stream);
    for (auto &ch : c.chunks) c.chunk_events.push_back(ch.ev);
    c.chunks.clear();
    c.chunks_recorded = 0;
  };
  auto chunk_event = [&]() {
    hipEvent_t ev;
    if (!c.chunk_events.empty()) {
      ev = c.chunk_events.back();
      c.chunk_events.pop_back();
    } else {
      HIP_CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    }
    return ev;
  };
  for (int attempt = 0;; attempt++) {
    const bool packed = try_packed && attempt == 0;
    try {
      DBuf<u8> dpacked;
      c.chunks_recorded = 0;
      int max_sym = 0, min_sym = 0;
      if (packed) {
        prepare_packed_upload(c);
        dpacked = DBuf<u8>(c, (size_t)(n / 4 + 64));
        c.sync();
        const i64 piece = c.xfer_chunk;
        for (i64 off = 0; off < n;) {
          const i64 len = n - off < piece ? n - off : piece;
          off += len;
          c.chunks.push_back({(u64)off, chunk_event(), false});
        }
        uploader = std::thread(upload_text_packed, std::ref(c), T_host, n, (u32)plo, (u32)(phi - plo), dpacked.get(), &upload_err);
        // the sample saw plo and phi; the uploader lets nothing else through: these are the text's statistics
        min_sym = plo;
        max_sym = phi;
        mark("uploader started (packed)");
      } else {
        constexpr i64 PIECE = (i64)256 << 20;
        for (i64 off = 0; off < n;) {
          const i64 len = off == 0 ? first : (n - off < PIECE ? n - off : PIECE);
          HIP_CHECK(hipMemcpyAsync(c.T_owned + off, T_host + off, (size_t)len, hipMemcpyHostToDevice, c.copy_stream));
          hipEvent_t ev = chunk_event();
          HIP_CHECK(hipEventRecord(ev, c.copy_stream));
          off += len;
          c.chunks.push_back({(u64)off, ev, false});
        }
        c.chunks_recorded = c.chunks.size();
        mark("upload enqueued (raw)");
        // the text's statistics (which symbols occur decides the layout of every sort key and rank block) come from
        // the host copy while the first chunk is on its way
        i64 n_zero = 0;
        int distinct = 0;
        host_text_stats(T_host, n, &max_sym, &min_sym, &n_zero, 16, &distinct);
        mark("text statistics done");
        if (n_zero != 1) fail(BWT_HIP_ETEXT, "text must hold exactly one 0 byte, found %ld", (long)n_zero);
        if (c.compact_allowed && max_sym - min_sym >= 4 && distinct < max_sym - min_sym + 1) {
          // Symbols scattered over the byte range (Ctx::compact): the whole text first, renumbered on the device, then
          // the build; nothing of it runs under the upload.
          drain();
          adopt_text(c, c.T_owned, n);
          std::unique_ptr<Block> blk(do_rec(c, 0, n, th, ns, seg));
          if (L_host) symbols_to_host(c, blk->L.get(), (u64)n, L_host);
          c.sync();
          mark("L in host memory (renumbered alphabet)");
          *out = publish(c, blk.release());
          break;
        }
      }
      c.T = c.T_owned;
      c.n = n;
      c.max_sym = max_sym;
      c.min_sym = min_sym;
      int bits = 1;
      while ((max_sym >> bits) != 0) bits++;
      c.sym_bits = bits;
      std::unique_ptr<Block> blk(do_rec(c, 0, n, th, ns, seg));
      mark("build enqueued");
      c.wait_text((u64)n);
      if (uploader.joinable()) uploader.join();
      if (trace) {
        c.sync();
        mark("build finished on the device");
      }
      if (!upload_err.empty()) fail(BWT_HIP_EINTERNAL, "upload of the text: %s", upload_err.c_str());
      if (L_host) {
        if (pack_size && min_sym >= 1 && max_sym - min_sym <= 3)
          download_packed(c, blk->L.get(), n, (u32)(min_sym > 252 ? 252 : min_sym), blk->s_rank, L_host);
        else
          HIP_CHECK(hipMemcpyAsync(L_host, blk->L.get(), (size_t)n, hipMemcpyDeviceToHost, c.stream));
      }
      c.sync();
      mark("L in host memory");
      drain();
      *out = publish(c, blk.release());
      break;
    } catch (const TextMismatch &) {
      drain();
      (void)hipStreamSynchronize(c.stream);
      c.T = nullptr;
      if (!packed) {  // only a packed upload raises it
        drop_text(c);
        fail(BWT_HIP_EINTERNAL, "text mismatch outside a packed upload");
      }
      mark("a symbol outside the sampled range: starting over, unpacked");
    } catch (...) {
      drain();
      (void)hipStreamSynchronize(c.stream);
      drop_text(c);
      throw;
    }
  }
  API_END
}

int bwt_hip_block_upload(bwt_hip_ctx *ctx, int64_t a, int64_t b, const uint8_t *L_host, int64_t s_rank,
                         int64_t e_rank, const int64_t *samp_rank, const int64_t *samp_pos, int64_t ns,
                         bwt_hip_block **out) {
  API_BEGIN
  need(ctx && L_host && out, "ctx, L_host, out");
  Ctx &c = ctx->c;
  need(c.T != nullptr, "no text set");
  need(a >= 0 && a < b && b <= c.n, "0 <= a < b <= n");
  need(s_rank >= a && s_rank < b && e_rank >= a && e_rank < b, "ranks inside [a,b)");
  need(ns >= 1 && samp_rank && samp_pos, "samples");
  HIP_CHECK(hipSetDevice(c.device));
  std::unique_ptr<Block> blk(new Block());
  blk->a = a;
  blk->b = b;
  blk->s_rank = s_rank - a;
  blk->e_rank = e_rank - a;
  blk->L = DBuf<u8>(c, (size_t)(b - a));
  blk->samp_rank = DBuf<u32>(c, (size_t)ns);
  blk->samp_pos = DBuf<u32>(c, (size_t)ns);
  blk->n_samp = ns;
  std::vector<u32> hr((size_t)ns), hp((size_t)ns);
  for (i64 k = 0; k < ns; k++) {
    need(samp_rank[k] >= a && samp_rank[k] < b && samp_pos[k] >= a && samp_pos[k] < b, "sample inside [a,b)");
    need(k == 0 || samp_pos[k] > samp_pos[k - 1], "samples ordered by pos");
    hr[(size_t)k] = (u32)(samp_rank[k] - a);
    hp[(size_t)k] = (u32)samp_pos[k];
  }
  HIP_CHECK(hipMemcpyAsync(blk->L.get(), L_host, (size_t)(b - a), hipMemcpyHostToDevice, c.stream));
  if (c.compact) map_symbols(c, blk->L.get(), blk->L.get(), (u64)(b - a), c.d_sym_tab);  // the text's symbols -> codes
  HIP_CHECK(hipMemcpyAsync(blk->samp_rank.get(), hr.data(), (size_t)ns * 4, hipMemcpyHostToDevice, c.stream));
  HIP_CHECK(hipMemcpyAsync(blk->samp_pos.get(), hp.data(), (size_t)ns * 4, hipMemcpyHostToDevice, c.stream));
  HIP_CHECK(hipMemcpyAsync(c.h_scalars, c.T + (b - 1), 1, hipMemcpyDeviceToHost, c.stream));
  c.sync();
  blk->e_char = *(u8 *)c.h_scalars;
  *out = publish(c, blk.release());
  API_END
}

__global__ void k_widen_samples(const u32 *__restrict__ r, const u32 *__restrict__ p, u32 n, i64 a,
                                i64 *__restrict__ ro, i64 *__restrict__ po) {
  u32 k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n) {
    ro[k] = a + (i64)r[k];
    po[k] = (i64)p[k];
  }
}

__global__ void k_narrow_samples(const i64 *__restrict__ r, const i64 *__restrict__ p, u32 n, i64 a,
                                 u32 *__restrict__ ro, u32 *__restrict__ po) {
  u32 k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n) {
    ro[k] = (u32)(r[k] - a);
    po[k] = (u32)p[k];
  }
}

int bwt_hip_block_export_device(bwt_hip_ctx *ctx, const bwt_hip_block *blk_, uint8_t *L_dev, int64_t *rank_dev,
                                int64_t *pos_dev) {
  API_BEGIN
  need(ctx && blk_ && L_dev && rank_dev && pos_dev, "ctx, blk, L_dev, samp_rank_dev, samp_pos_dev");
  Ctx &c = ctx->c;
  const Block *blk = as_block(blk_);
  HIP_CHECK(hipSetDevice(c.device));
  HIP_CHECK(hipMemcpyAsync(L_dev, blk->L.get(), (size_t)blk->len(), hipMemcpyDeviceToDevice, c.stream));
  if (blk->n_samp > 0) {
    hipLaunchKernelGGL(k_widen_samples, dim3(ceil_div(blk->n_samp, 256)), dim3(256), 0, c.stream,
                       blk->samp_rank.get(), blk->samp_pos.get(), (u32)blk->n_samp, blk->a, rank_dev, pos_dev);
    HIP_CHECK(hipGetLastError());
  }
  c.sync();
  API_END
}

int bwt_hip_block_import_device(bwt_hip_ctx *ctx, int64_t a, int64_t b, const uint8_t *L_dev, int64_t s_rank,
                                int64_t e_rank, const int64_t *rank_dev, const int64_t *pos_dev, int64_t ns,
                                bwt_hip_block **out) {
  API_BEGIN
  need(ctx && L_dev && out && (ns == 0 || (rank_dev && pos_dev)), "ctx, L_dev, samp_rank_dev, samp_pos_dev, out");
  Ctx &c = ctx->c;
  need(c.T != nullptr, "no text set");
  need(a >= 0 && a < b && b <= c.n, "0 <= a < b <= n");
  need(s_rank >= a && s_rank < b && e_rank >= a && e_rank < b, "ranks inside [a,b)");
  need(ns >= 0, "n_samples >= 0 (0: a block that only serves as the left side of LF chains)");
  HIP_CHECK(hipSetDevice(c.device));
  std::unique_ptr<Block> blk(new Block());
  blk->a = a;
  blk->b = b;
  blk->s_rank = s_rank - a;
  blk->e_rank = e_rank - a;
  blk->L = DBuf<u8>(c, (size_t)(b - a));
  blk->samp_rank = DBuf<u32>(c, (size_t)(ns ? ns : 1));
  blk->samp_pos = DBuf<u32>(c, (size_t)(ns ? ns : 1));
  blk->n_samp = ns;
  HIP_CHECK(hipMemcpyAsync(blk->L.get(), L_dev, (size_t)(b - a), hipMemcpyDeviceToDevice, c.stream));
  if (ns > 0) {
    hipLaunchKernelGGL(k_narrow_samples, dim3(ceil_div(ns, 256)), dim3(256), 0, c.stream, rank_dev, pos_dev, (u32)ns, a,
                       blk->samp_rank.get(), blk->samp_pos.get());
    HIP_CHECK(hipGetLastError());
  }
  HIP_CHECK(hipMemcpyAsync(c.h_scalars, c.T + (b - 1), 1, hipMemcpyDeviceToHost, c.stream));
  c.sync();
  blk->e_char = *(u8 *)c.h_scalars;
  *out = publish(c, blk.release());
  API_END
}

void bwt_hip_block_free(bwt_hip_ctx *ctx, bwt_hip_block *blk) {
  (void)ctx;
  if (!blk) return;
  {
    std::lock_guard<std::mutex> lock(g_blocks_mu);
    if (g_blocks.erase(as_block(blk)) == 0) return;  // already gone with its context
  }
  delete as_block(blk);
}

// ---------------------------------------------------------------------------
// reading a block
// ---------------------------------------------------------------------------
int bwt_hip_block_info(const bwt_hip_block *blk_, int64_t *a, int64_t *b, int64_t *s_rank, int64_t *e_rank,
                       int *e_char, int64_t *n_samples) {
  API_BEGIN
  need(blk_ != nullptr, "blk");
  const Block *blk = as_block(blk_);
  if (a) *a = blk->a;
  if (b) *b = blk->b;
  if (s_rank) *s_rank = blk->a + blk->s_rank;
  if (e_rank) *e_rank = blk->a + blk->e_rank;
  if (e_char) {
    *e_char = blk->e_char;
    std::lock_guard<std::mutex> lock(g_blocks_mu);
    auto it = g_blocks.find(const_cast<Block *>(blk));
    if (it != g_blocks.end() && it->second->compact) *e_char = it->second->sym_back[blk->e_char];  // the text's own symbol
  }
  if (n_samples) *n_samples = blk->n_samp;
  API_END
}

int bwt_hip_block_get_L(bwt_hip_ctx *ctx, const bwt_hip_block *blk_, uint8_t *L_host) {
  API_BEGIN
  need(ctx && blk_ && L_host, "ctx, blk, L_host");
  const Block *blk = as_block(blk_);
  HIP_CHECK(hipSetDevice(ctx->c.device));
  symbols_to_host(ctx->c, blk->L.get(), (u64)blk->len(), L_host);
  API_END
}

int bwt_hip_block_L_device(const bwt_hip_block *blk_, const uint8_t **L_dev) {
  API_BEGIN
  need(blk_ && L_dev, "blk, L_dev");
  *L_dev = as_block(blk_)->L.get();
  API_END
}

int bwt_hip_block_get_samples(bwt_hip_ctx *ctx, const bwt_hip_block *blk_, int64_t *rank_out, int64_t *pos_out,
                              int64_t cap) {
  API_BEGIN
  need(ctx && blk_ && rank_out && pos_out, "ctx, blk, rank_out, pos_out");
  const Block *blk = as_block(blk_);
  Ctx &c = ctx->c;
  HIP_CHECK(hipSetDevice(c.device));
  i64 k = blk->n_samp < cap ? blk->n_samp : cap;
  if (k >= ((i64)1 << 20)) {
    // large sets cross the link as the 32-bit words they are on the device and are widened by host threads
    download_samples32(c, blk->samp_rank.get(), blk->samp_pos.get(), k, blk->a, rank_out, pos_out);
  } else if (k > 0) {
    // widened to the reference's idx_t on the device, then one copy per array straight into the caller's memory
    DBuf<i64> wr(c, (size_t)k), wp(c, (size_t)k);
    hipLaunchKernelGGL(k_widen_samples, dim3(ceil_div(k, 256)), dim3(256), 0, c.stream, blk->samp_rank.get(),
                       blk->samp_pos.get(), (u32)k, blk->a, wr.get(), wp.get());
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipMemcpyAsync(rank_out, wr.get(), (size_t)k * 8, hipMemcpyDeviceToHost, c.stream));
    HIP_CHECK(hipMemcpyAsync(pos_out, wp.get(), (size_t)k * 8, hipMemcpyDeviceToHost, c.stream));
    c.sync();
  }
  API_END
}

// ---------------------------------------------------------------------------
// taps
// ---------------------------------------------------------------------------
int bwt_hip_leaf_sa(bwt_hip_ctx *ctx, int64_t a, int64_t b, int64_t *SA_host) {
  API_BEGIN
  need(ctx && SA_host, "ctx, SA_host");
  Ctx &c = ctx->c;
  HIP_CHECK(hipSetDevice(c.device));
  DBuf<u32> sa = leaf_suffix_array(c, a, b);
  std::vector<u32> h((size_t)(b - a));
  HIP_CHECK(hipMemcpyAsync(h.data(), sa.get(), h.size() * 4, hipMemcpyDeviceToHost, c.stream));
  c.sync();
  for (size_t i = 0; i < h.size(); i++) SA_host[i] = (i64)h[i];
  API_END
}

int bwt_hip_block_init_extra(bwt_hip_ctx *ctx, bwt_hip_block *blk) {
  API_BEGIN
  need(ctx && blk, "ctx, blk");
  HIP_CHECK(hipSetDevice(ctx->c.device));
  build_wavelet(ctx->c, *as_block(blk));
  API_END
}

int bwt_hip_block_get_C(bwt_hip_ctx *ctx, bwt_hip_block *blk_, int alpha_max, int64_t *C_out) {
  API_BEGIN
  need(ctx && blk_ && C_out, "ctx, blk, C_out");
  need(alpha_max >= 0 && alpha_max <= 255, "alpha_max in 0..255");
  Block *blk = as_block(blk_);
  HIP_CHECK(hipSetDevice(ctx->c.device));
  count_alphabets(ctx->c, *blk);
  const int top = ctx->c.compact ? (int)ctx->c.sym_back[ctx->c.max_sym] : ctx->c.max_sym;
  if (top > alpha_max) fail(BWT_HIP_ETEXT, "text holds symbol %d above alpha_max %d", top, alpha_max);
  if (ctx->c.compact) {
    // C[k] = symbols below k: that of the smallest symbol of the text that is not below k
    int code = ctx->c.max_sym + 1;  // (none: all of them are below)
    for (int k = 255; k >= 0; k--) {
      if (ctx->c.sym_fwd[k]) code = ctx->c.sym_fwd[k];
      if (k == 0) code = 0;
      if (k <= alpha_max) C_out[k] = blk->C[code];
    }
  } else {
    for (int k = 0; k <= alpha_max; k++) C_out[k] = blk->C[k];
  }
  API_END
}

static void run_rank(Ctx &c, Block &blk, i64 nq, const uint8_t *sym, const int64_t *pos, int64_t *out, bool lf) {
  if (nq <= 0) return;
  const i64 m = blk.len();
  std::vector<u32> hp((size_t)nq);
  for (i64 k = 0; k < nq; k++) {
    i64 p = lf ? pos[k] - blk.a : pos[k];
    if (p < 0 || p > m) fail(BWT_HIP_EINVAL, "query %ld: position %ld outside the block", (long)k, (long)pos[k]);
    if (!c.compact && ((int)sym[k] >> c.sym_bits)) fail(BWT_HIP_EINVAL, "query %ld: symbol %d not in the text's alphabet", (long)k, (int)sym[k]);
    hp[(size_t)k] = (u32)p;
  }
  DBuf<u8> ds(c, (size_t)nq);
  DBuf<u32> dp(c, (size_t)nq), dout(c, (size_t)nq);
  std::vector<u8> codes;
  if (c.compact) {  // the text's symbols -> codes
    codes.resize((size_t)nq);
    for (i64 k = 0; k < nq; k++) {
      if (sym[k] && !c.sym_fwd[sym[k]]) fail(BWT_HIP_EINVAL, "query %ld: symbol %d not in the text's alphabet", (long)k, (int)sym[k]);
      codes[(size_t)k] = c.sym_fwd[sym[k]];
    }
    sym = codes.data();
  }
  HIP_CHECK(hipMemcpyAsync(ds.get(), sym, (size_t)nq, hipMemcpyHostToDevice, c.stream));
  HIP_CHECK(hipMemcpyAsync(dp.get(), hp.data(), (size_t)nq * 4, hipMemcpyHostToDevice, c.stream));
  rank_batch(c, blk, ds.get(), dp.get(), dout.get(), (size_t)nq, lf);
  HIP_CHECK(hipMemcpyAsync(hp.data(), dout.get(), (size_t)nq * 4, hipMemcpyDeviceToHost, c.stream));
  c.sync();
  for (i64 k = 0; k < nq; k++) out[k] = (i64)hp[(size_t)k] + (lf ? blk.a : 0);
}

int bwt_hip_block_rank(bwt_hip_ctx *ctx, bwt_hip_block *blk, int64_t nq, const uint8_t *sym, const int64_t *pos,
                       int64_t *out) {
  API_BEGIN
  need(ctx && blk && sym && pos && out, "ctx, blk, sym, pos, out");
  HIP_CHECK(hipSetDevice(ctx->c.device));
  run_rank(ctx->c, *as_block(blk), nq, sym, pos, out, false);
  API_END
}

int bwt_hip_block_LF(bwt_hip_ctx *ctx, bwt_hip_block *blk, int64_t nq, const uint8_t *sym, const int64_t *rank,
                     int64_t *out) {
  API_BEGIN
  need(ctx && blk && sym && rank && out, "ctx, blk, sym, rank, out");
  HIP_CHECK(hipSetDevice(ctx->c.device));
  run_rank(ctx->c, *as_block(blk), nq, sym, rank, out, true);
  API_END
}

int bwt_hip_block_ibwt(bwt_hip_ctx *ctx, bwt_hip_block *blk_, uint8_t *I_host, int64_t *mismatches) {
  API_BEGIN
  need(ctx && blk_, "ctx, blk");
  Ctx &c = ctx->c;
  need(c.T != nullptr, "no text set");
  HIP_CHECK(hipSetDevice(c.device));
  Block &blk = *as_block(blk_);
  DBuf<u8> out;
  if (I_host) out = DBuf<u8>(c, (size_t)blk.len());
  i64 bad = 0;
  inverse_block(c, blk, out.get(), mismatches ? &bad : nullptr);
  if (mismatches) *mismatches = bad;
  if (I_host) symbols_to_host(c, out.get(), (u64)blk.len(), I_host);
  API_END
}

int bwt_hip_block_sa(bwt_hip_ctx *ctx, bwt_hip_block *blk_, int64_t nq, const int64_t *rank, int64_t *pos_out) {
  API_BEGIN
  need(ctx && blk_ && nq >= 0 && (nq == 0 || (rank && pos_out)), "ctx, blk, nq >= 0, rank, pos_out");
  Ctx &c = ctx->c;
  HIP_CHECK(hipSetDevice(c.device));
  Block &blk = *as_block(blk_);
  if (nq > 0) {
    std::vector<u32> h((size_t)nq);
    for (i64 k = 0; k < nq; k++) {
      if (rank[k] < blk.a || rank[k] >= blk.b) fail(BWT_HIP_EINVAL, "query %ld: rank %ld outside [a,b)", (long)k, (long)rank[k]);
      h[(size_t)k] = (u32)(rank[k] - blk.a);
    }
    DBuf<u32> dq(c, (size_t)nq), dout(c, (size_t)nq);
    HIP_CHECK(hipMemcpyAsync(dq.get(), h.data(), (size_t)nq * 4, hipMemcpyHostToDevice, c.stream));
    sa_batch(c, blk, dq.get(), dout.get(), (size_t)nq);
    HIP_CHECK(hipMemcpyAsync(h.data(), dout.get(), (size_t)nq * 4, hipMemcpyDeviceToHost, c.stream));
    c.sync();
    for (i64 k = 0; k < nq; k++) pos_out[k] = (i64)h[(size_t)k];
  }
  API_END
}

int bwt_hip_dc_sample(bwt_hip_ctx *ctx, int64_t cap, int64_t *pos_out, int64_t *rank_out, int64_t *n_out, int *period) {
  API_BEGIN
  need(ctx != nullptr && cap >= 0, "ctx, cap >= 0");
  Ctx &c = ctx->c;
  need(c.T != nullptr, "no text set");
  HIP_CHECK(hipSetDevice(c.device));
  ensure_dc_sample(c);
  if (n_out) *n_out = c.dc.n_s;
  if (period) *period = (int)c.dc.v;
  const i64 k = (i64)c.dc.n_s < cap ? (i64)c.dc.n_s : cap;
  if (k > 0 && rank_out) {
    std::vector<u32> h((size_t)k);
    HIP_CHECK(hipMemcpyAsync(h.data(), c.dc.rank, (size_t)k * 4, hipMemcpyDeviceToHost, c.stream));
    c.sync();
    for (i64 i = 0; i < k; i++) rank_out[i] = (i64)h[(size_t)i];
  }
  if (pos_out)
    for (i64 i = 0; i < k; i++) pos_out[i] = (i / c.dc.nd) * (i64)c.dc.v + (i64)c.dc.D[i % c.dc.nd];
  API_END
}

int bwt_hip_gap(bwt_hip_ctx *ctx, bwt_hip_block *l_, bwt_hip_block *r_, int64_t seg, int64_t *gap_out,
                int64_t *g_out) {
  API_BEGIN
  need(ctx && l_ && r_, "ctx, l, r");
  Ctx &c = ctx->c;
  HIP_CHECK(hipSetDevice(c.device));
  Block &l = *as_block(l_), &r = *as_block(r_);
  GapResult G = build_gap(c, l, r, seg, g_out != nullptr);
  const size_t m = (size_t)l.len(), rn = (size_t)r.len();
  if (gap_out) {
    std::vector<u32> S(m + 1);
    HIP_CHECK(hipMemcpyAsync(S.data(), G.S.get(), (m + 1) * 4, hipMemcpyDeviceToHost, c.stream));
    c.sync();
    for (size_t i = 0; i <= m; i++) gap_out[i] = (i64)S[i] - (i ? (i64)S[i - 1] : 0);
  }
  if (g_out) {
    std::vector<u32> g(rn);
    HIP_CHECK(hipMemcpyAsync(g.data(), G.g_text.get(), rn * 4, hipMemcpyDeviceToHost, c.stream));
    c.sync();
    for (size_t j = 0; j < rn; j++) g_out[j] = l.a + (i64)g[j];
  }
  API_END
}

// ---------------------------------------------------------------------------
// a merge spread over several GPUs
// ---------------------------------------------------------------------------
int bwt_hip_dist_bounds(int64_t m, int parts, int64_t *lo) {
  API_BEGIN
  need(m >= 0 && parts >= 1 && parts <= 256 && lo, "m >= 0, 1 <= parts <= 256, lo");
  partition_bounds((u64)m, (u32)parts, lo);
  API_END
}

int bwt_hip_dist_layout(bwt_hip_ctx *ctx, int64_t m, int parts, int64_t *lo, int *shift) {
  API_BEGIN
  need(ctx && m >= 0 && parts >= 1 && parts <= 256 && lo && shift, "ctx, m >= 0, 1 <= parts <= 256, lo, shift");
  *shift = partition_shift(ctx->c, (u64)m);
  partition_bounds((u64)m, (u32)parts, lo, *shift);
  API_END
}

int bwt_hip_dist_chains_grouped(bwt_hip_ctx *ctx, bwt_hip_block *l_, int64_t ra, int64_t rb, int64_t seg, uint32_t *g_text_dev,
                                uint16_t *words_dev, int64_t words_cap, uint32_t *starts_dev, int64_t *n_words,
                                int64_t *n_buckets, int64_t *fillers) {
  API_BEGIN
  need(ctx && l_ && g_text_dev && words_dev && starts_dev && n_words && n_buckets && fillers, "ctx, l, device buffers, outputs");
  Ctx &c = ctx->c;
  HIP_CHECK(hipSetDevice(c.device));
  Block &l = *as_block(l_);
  const u64 m = (u64)l.len();
  need(gap_grouping_applies(c, m), "a left block whose g values travel grouped (bwt_hip_dist_layout: shift != 0)");
  const u32 nb = (u32)((m >> 13) + 1);
  *n_buckets = nb;
  DBuf<u32> start(c, gap_group16_starts(m));
  u64 total = 0;
  if (rb > ra) {
    DBuf<u32> g0;
    ChainLayout lay = run_chains(c, l, ra, rb, seg, g0);
    chains_text_order(c, lay, g0.get(), g_text_dev);
    total = (u64)lay.seg * lay.n_chains;
    need((i64)total <= words_cap, "words_cap >= the positions rounded up to whole chains");
    DBuf<u32> g1(c, (size_t)total);
    gap_group16(c, g0.get(), g1.get(), total, m, start.get());
    HIP_CHECK(hipMemcpyAsync(c.h_scalars, start.get() + nb, sizeof(u32), hipMemcpyDeviceToHost, c.stream));
    c.sync();
    *n_words = (i64)c.h_scalars[0];
    if (*n_words > 0) HIP_CHECK(hipMemcpyAsync(words_dev, g0.get(), (size_t)*n_words * 2, hipMemcpyDeviceToDevice, c.stream));
  } else {
    HIP_CHECK(hipMemsetAsync(start.get(), 0, gap_group16_starts(m) * sizeof(u32), c.stream));
    *n_words = 0;
  }
  HIP_CHECK(hipMemcpyAsync(starts_dev, start.get(), ((size_t)nb + 1) * sizeof(u32), hipMemcpyDeviceToDevice, c.stream));
  // fillers (value m + 1) inside the buckets 0 .. nb-1: all of them, or none when m + 1 opens a bucket of its own
  *fillers = (((m + 1) >> 13) < nb) ? (i64)total - (rb - ra) : 0;
  c.sync();
  API_END
}

int bwt_hip_dist_merge_slice_grouped(bwt_hip_ctx *ctx, const uint8_t *lL_dev, int64_t mL, const uint8_t *rL_dev, int64_t rn,
                                     const uint16_t *words_dev, const int64_t *word_off, const uint32_t *starts_dev, int sources,
                                     int64_t nbk, uint8_t *out_dev, uint32_t *S_dev) {
  API_BEGIN
  need(ctx && S_dev && mL >= 0 && rn >= 0 && word_off && starts_dev && sources >= 1 && sources <= 256 && nbk >= 0, "ctx, S_dev, mL >= 0, rn >= 0, word_off, starts_dev, 1 <= sources <= 256");
  need((mL == 0 || lL_dev) && (rn == 0 || (rL_dev && words_dev)) && (mL + rn == 0 || out_dev), "device buffers");
  need(mL + rn < (int64_t)((1ull << 32) - 2), "slice below 2^32-2 symbols");
  Ctx &c = ctx->c;
  HIP_CHECK(hipSetDevice(c.device));
  std::vector<u32> h_off((size_t)sources);
  for (int s = 0; s < sources; s++) {
    need(word_off[s] >= 0 && word_off[s] < (int64_t)1 << 32 && (s == 0 || word_off[s] >= word_off[s - 1]),
         "word_off: non-decreasing offsets below 2^32");
    h_off[(size_t)s] = (u32)word_off[s];
  }
  DBuf<u32> d_off(c, (size_t)sources);
  HIP_CHECK(hipMemcpyAsync(d_off.get(), h_off.data(), (size_t)sources * sizeof(u32), hipMemcpyHostToDevice, c.stream));
  gap_prefix_grouped(c, words_dev, d_off.get(), starts_dev, (u32)sources, (u32)nbk, (u64)mL, S_dev);
  merge_symbols(c, lL_dev, (u64)mL, rL_dev, (u64)rn, S_dev, out_dev);
  c.sync();
  API_END
}

int bwt_hip_dist_leaf(bwt_hip_ctx *ctx, int64_t a, int64_t b, int64_t ns, int part, int parts, uint8_t *L_dev, int64_t cap,
                      uint64_t *slots_dev, int64_t *first_rank, int64_t *count, int64_t *s_rank, int64_t *e_rank,
                      int64_t *bounds, int *applies) {
  API_BEGIN
  need(ctx && first_rank && count && s_rank && e_rank && applies, "ctx, first_rank, count, s_rank, e_rank, applies");
  need(parts >= 1 && parts <= 256 && part >= 0 && part < parts, "0 <= part < parts <= 256");
  Ctx &c = ctx->c;
  HIP_CHECK(hipSetDevice(c.device));
  LeafShare sh;
  *applies = build_leaf_share(c, a, b, ns, part, parts, sh) ? 1 : 0;
  if (*applies) {
    const i64 lo = (i64)sh.bounds[(size_t)part], cnt = (i64)sh.bounds[(size_t)part + 1] - lo;
    need(!L_dev || cnt <= cap, "cap: room for this share of L");
    *first_rank = lo;
    *count = cnt;
    *s_rank = sh.s_rank;
    *e_rank = sh.e_rank;
    if (bounds)
      for (int j = 0; j <= parts; j++) bounds[j] = (int64_t)sh.bounds[(size_t)j];
    if (L_dev && cnt > 0) HIP_CHECK(hipMemcpyAsync(L_dev, sh.L.get(), (size_t)cnt, hipMemcpyDeviceToDevice, c.stream));
    if (slots_dev) HIP_CHECK(hipMemcpyAsync(slots_dev, sh.samp.get(), (size_t)sh.ns * 8, hipMemcpyDeviceToDevice, c.stream));
    c.sync();
  }
  API_END
}

int bwt_hip_dist_chains(bwt_hip_ctx *ctx, bwt_hip_block *l_, int64_t ra, int64_t rb, int64_t seg, uint32_t *g_dev) {
  API_BEGIN
  need(ctx && l_ && g_dev, "ctx, l, g_dev");
  Ctx &c = ctx->c;
  HIP_CHECK(hipSetDevice(c.device));
  DBuf<u32> g0;
  ChainLayout lay = run_chains(c, *as_block(l_), ra, rb, seg, g0);
  chains_text_order(c, lay, g0.get(), g_dev);
  c.sync();
  API_END
}

int bwt_hip_dist_partition(bwt_hip_ctx *ctx, const uint32_t *g_dev, int64_t count, int64_t m, int parts,
                           uint32_t *out_dev, int64_t *counts) {
  API_BEGIN
  need(ctx && counts && count >= 0 && m >= 0, "ctx, counts, count >= 0, m >= 0");
  need(count == 0 || (g_dev && out_dev), "g_dev, out_dev");
  Ctx &c = ctx->c;
  HIP_CHECK(hipSetDevice(c.device));
  partition_by_left_rank(c, g_dev, (u64)count, (u64)m, (u32)parts, out_dev, counts);
  API_END
}

int bwt_hip_dist_wire_packable(bwt_hip_ctx *ctx, int *packable) {
  API_BEGIN
  need(ctx && packable, "ctx, packable");
  *packable = wire_packable(ctx->c) ? 1 : 0;
  API_END
}

int bwt_hip_dist_wire_pack(bwt_hip_ctx *ctx, const uint8_t *src_dev, int64_t count, uint8_t *packed_dev) {
  API_BEGIN
  need(ctx && count >= 0 && (count == 0 || (src_dev && packed_dev)), "ctx, count >= 0, src_dev, packed_dev");
  Ctx &c = ctx->c;
  need(wire_packable(c), "a text whose symbols span at most four consecutive values");
  HIP_CHECK(hipSetDevice(c.device));
  wire_pack2(c, src_dev, (u64)count, packed_dev);
  c.sync();
  API_END
}

int bwt_hip_dist_wire_unpack(bwt_hip_ctx *ctx, const uint8_t *packed_dev, int64_t count, uint8_t *dst_dev) {
  API_BEGIN
  need(ctx && count >= 0 && (count == 0 || (packed_dev && dst_dev)), "ctx, count >= 0, packed_dev, dst_dev");
  Ctx &c = ctx->c;
  need(wire_packable(c), "a text whose symbols span at most four consecutive values");
  HIP_CHECK(hipSetDevice(c.device));
  wire_unpack2(c, packed_dev, (u64)count, dst_dev);
  c.sync();
  API_END
}

int bwt_hip_dist_merge_slice(bwt_hip_ctx *ctx, const uint8_t *lL_dev, int64_t mL, const uint8_t *rL_dev, int64_t rn,
                             uint32_t *g_dev, uint8_t *out_dev, uint32_t *S_dev) {
  API_BEGIN
  need(ctx && S_dev && mL >= 0 && rn >= 0, "ctx, S_dev, mL >= 0, rn >= 0");
  need((mL == 0 || lL_dev) && (rn == 0 || (rL_dev && g_dev)) && (mL + rn == 0 || out_dev), "device buffers");
  need(mL + rn < (int64_t)((1ull << 32) - 2), "slice below 2^32-2 symbols");
  Ctx &c = ctx->c;
  HIP_CHECK(hipSetDevice(c.device));
  DBuf<u32> scratch(c, (size_t)(rn > 0 ? rn : 1));
  gap_prefix(c, g_dev, scratch.get(), (u64)rn, (u64)mL, S_dev);
  merge_symbols(c, lL_dev, (u64)mL, rL_dev, (u64)rn, S_dev, out_dev);
  c.sync();
  API_END
}

int bwt_hip_dist_merge_slice_noalloc(bwt_hip_ctx *ctx, const uint8_t *lL_dev, int64_t mL, const uint8_t *rL_dev,
                                     int64_t rn, uint32_t *g_dev, uint8_t *out_dev, uint32_t *S_dev) {
  API_BEGIN
  need(ctx && S_dev && mL >= 0 && rn >= 0, "ctx, S_dev, mL >= 0, rn >= 0");
  need((mL == 0 || lL_dev) && (rn == 0 || (rL_dev && g_dev)) && (mL + rn == 0 || out_dev), "device buffers");
  need(mL + rn < (int64_t)((1ull << 32) - 2), "slice below 2^32-2 symbols");
  if (rn > (int64_t)BWT_HIP_NOALLOC_MAX_RIGHT)
    fail(BWT_HIP_EINVAL, "bwt_hip_dist_merge_slice_noalloc: rn = %ld exceeds the %ld right symbols the static scratch holds",
         (long)rn, (long)BWT_HIP_NOALLOC_MAX_RIGHT);
  Ctx &c = ctx->c;
  HIP_CHECK(hipSetDevice(c.device));
  {
    ArenaScope arena(c);  // the sort's second buffer, tile counts and scan partials come out of the static array
    DBuf<u32> second(c, (size_t)(rn > 0 ? rn : 1));
    gap_prefix(c, g_dev, second.get(), (u64)rn, (u64)mL, S_dev);
    merge_symbols(c, lL_dev, (u64)mL, rL_dev, (u64)rn, S_dev, out_dev);
    c.sync();
  }
  API_END
}

int bwt_hip_dist_samples_route(bwt_hip_ctx *ctx, int is_left, const uint32_t *rank_dev, const uint32_t *pos_dev, int64_t n,
                               int64_t a, int64_t cpos, int64_t m, int64_t r, int members, int shift, uint32_t *queries_dev,
                               uint32_t *perm_dev, int64_t *counts) {
  API_BEGIN
  need(ctx && counts && n >= 0 && members >= 1 && members <= 256, "ctx, counts, n >= 0, 1 <= members <= 256");
  need(n == 0 || (queries_dev && (is_left ? (rank_dev && perm_dev) : pos_dev != nullptr)), "device buffers");
  Ctx &c = ctx->c;
  HIP_CHECK(hipSetDevice(c.device));
  need(shift == 0 || shift == 13, "shift as bwt_hip_dist_layout gives it");
  if (is_left) samples_route_left(c, rank_dev, (u64)n, a, (u64)m, (u32)members, queries_dev, perm_dev, counts, shift);
  else samples_route_right(c, pos_dev, (u64)n, cpos, (u64)r, (u32)members, queries_dev, counts);
  API_END
}

int bwt_hip_dist_samples_answer(bwt_hip_ctx *ctx, const uint32_t *queries_dev, int64_t n_left, int64_t n_right,
                                const uint32_t *S_dev, int64_t i_lo, int64_t right_before, int64_t a, const uint32_t *g_dev,
                                uint32_t *answers_dev) {
  API_BEGIN
  need(ctx && n_left >= 0 && n_right >= 0, "ctx, n_left >= 0, n_right >= 0");
  need(n_left + n_right == 0 || (queries_dev && answers_dev), "queries_dev, answers_dev");
  need((n_left == 0 || S_dev) && (n_right == 0 || g_dev), "S_dev, g_dev");
  Ctx &c = ctx->c;
  HIP_CHECK(hipSetDevice(c.device));
  samples_answer(c, queries_dev, (u64)n_left, (u64)n_right, S_dev, (u64)i_lo, (u64)right_before, a, g_dev, answers_dev);
  c.sync();
  API_END
}

int bwt_hip_dist_samples_place(bwt_hip_ctx *ctx, int is_left, const uint32_t *rank_dev, const uint32_t *pos_dev, int64_t n,
                               const uint32_t *answers_dev, const uint32_t *perm_dev, int64_t a, int64_t cpos,
                               int64_t first_index, int64_t n_all, int64_t ns_out, uint32_t *o_rank_dev, uint32_t *o_pos_dev,
                               int64_t *kept) {
  API_BEGIN
  need(ctx && kept && n >= 0 && first_index >= 0 && first_index + n <= n_all && ns_out >= 1, "ctx, kept, 0 <= first_index, first_index + n <= n_all, ns_out >= 1");
  Ctx &c = ctx->c;
  HIP_CHECK(hipSetDevice(c.device));
  SamplePlace P;
  P.A = (u64)(ns_out - 1);
  P.B = n_all ? (u64)(n_all - 1) : 0;
  P.first_index = (u64)first_index;
  P.first_slot = samples_taken_before(P.first_index, P.A, P.B);
  *kept = (i64)(samples_taken_before(P.first_index + (u64)n, P.A, P.B) - P.first_slot);
  if (o_rank_dev && o_pos_dev && n > 0) {
    need(rank_dev && pos_dev && answers_dev && (!is_left || perm_dev), "device buffers");
    P.rank = rank_dev;
    P.pos = pos_dev;
    P.n = (u64)n;
    P.ans = answers_dev;
    P.perm = is_left ? perm_dev : nullptr;
    P.right_shift = is_left ? 0u : (u32)(cpos - a);
    P.o_rank = o_rank_dev;
    P.o_pos = o_pos_dev;
    samples_place(c, P);
    c.sync();
  }
  API_END
}

// ---------------------------------------------------------------------------
// observability
// ---------------------------------------------------------------------------
int bwt_hip_set_timing(bwt_hip_ctx *ctx, int level) {
  API_BEGIN
  need(ctx != nullptr, "ctx");
  ctx->c.flush_timers();
  ctx->c.timing_level = level;
  API_END
}

int bwt_hip_reset_timers(bwt_hip_ctx *ctx) {
  API_BEGIN
  need(ctx != nullptr, "ctx");
  ctx->c.flush_timers();
  for (auto &t : ctx->c.timers) {
    t.ms = 0;
    t.bytes = 0;
    t.launches = 0;
  }
  API_END
}

int bwt_hip_timer_count(bwt_hip_ctx *ctx) {
  if (!ctx) return 0;
  try {
    ctx->c.flush_timers();
  } catch (...) {
  }
  return (int)ctx->c.timers.size();
}

int bwt_hip_timer_get(bwt_hip_ctx *ctx, int idx, const char **name, double *ms, double *work, int64_t *calls) {
  API_BEGIN
  need(ctx != nullptr, "ctx");
  need(idx >= 0 && idx < (int)ctx->c.timers.size(), "idx");
  const PhaseTimer &t = ctx->c.timers[(size_t)idx];
  if (name) *name = t.name.c_str();
  if (ms) *ms = t.ms;
  if (work) *work = t.bytes;
  if (calls) *calls = t.launches;
  API_END
}

int bwt_hip_mem_stats(bwt_hip_ctx *ctx, int64_t *live, int64_t *peak, int64_t *reserved) {
  API_BEGIN
  need(ctx != nullptr, "ctx");
  if (live) *live = (i64)ctx->c.bytes_live;
  if (peak) *peak = (i64)ctx->c.bytes_peak;
  if (reserved) *reserved = (i64)ctx->c.bytes_reserved;
  API_END
}

int bwt_hip_trim(bwt_hip_ctx *ctx) {
  API_BEGIN
  need(ctx != nullptr, "ctx");
  ctx->c.trim();
  API_END
}

void *bwt_hip_stream(bwt_hip_ctx *ctx) { return ctx ? (void *)ctx->c.stream : nullptr; }

int bwt_hip_sync(bwt_hip_ctx *ctx) {
  API_BEGIN
  need(ctx != nullptr, "ctx");
  ctx->c.sync();
  API_END
}

// ---------------------------------------------------------------------------
// primitives for unit tests
// ---------------------------------------------------------------------------
template <class K>
static void test_sort(Ctx &c, K *keys, u32 *vals, i64 n, int b0, int b1) {
  if (n <= 0) return;
  DBuf<K> k0(c, (size_t)n), k1(c, (size_t)n);
  DBuf<u32> v0, v1;
  HIP_CHECK(hipMemcpyAsync(k0.get(), keys, (size_t)n * sizeof(K), hipMemcpyHostToDevice, c.stream));
  if (vals) {
    v0 = DBuf<u32>(c, (size_t)n);
    v1 = DBuf<u32>(c, (size_t)n);
    HIP_CHECK(hipMemcpyAsync(v0.get(), vals, (size_t)n * 4, hipMemcpyHostToDevice, c.stream));
  }
  K *kk[2] = {k0.get(), k1.get()};
  u32 *vv[2] = {v0.get(), v1.get()};
  int cur;
  if (sizeof(K) == 8)
    cur = radix_sort_u64(c, (u64 **)kk, vv, (size_t)n, b0, b1);
  else
    cur = radix_sort_u32(c, (u32 **)kk, vv, (size_t)n, b0, b1, (!vals && b1 - b0 > 16 && b1 - b0 <= 18) ? 2 : 1);
  HIP_CHECK(hipMemcpyAsync(keys, kk[cur], (size_t)n * sizeof(K), hipMemcpyDeviceToHost, c.stream));
  if (vals) HIP_CHECK(hipMemcpyAsync(vals, vv[cur], (size_t)n * 4, hipMemcpyDeviceToHost, c.stream));
  c.sync();
}

int bwt_hip_test_sort_u64(bwt_hip_ctx *ctx, uint64_t *keys, uint32_t *vals, int64_t n, int b0, int b1) {
  API_BEGIN
  need(ctx && keys, "ctx, keys");
  need(b0 >= 0 && b0 <= b1 && b1 <= 64, "bit range");
  HIP_CHECK(hipSetDevice(ctx->c.device));
  test_sort<u64>(ctx->c, keys, vals, n, b0, b1);
  API_END
}

int bwt_hip_test_sort_u32(bwt_hip_ctx *ctx, uint32_t *keys, uint32_t *vals, int64_t n, int b0, int b1) {
  API_BEGIN
  need(ctx && keys, "ctx, keys");
  need(b0 >= 0 && b0 <= b1 && b1 <= 32, "bit range");
  HIP_CHECK(hipSetDevice(ctx->c.device));
  test_sort<u32>(ctx->c, keys, vals, n, b0, b1);
  API_END
}

int bwt_hip_test_scan_u32(bwt_hip_ctx *ctx, uint32_t *data, int64_t n, int op) {
  API_BEGIN
  need(ctx && data, "ctx, data");
  need(op == 0 || op == 1, "op");
  Ctx &c = ctx->c;
  HIP_CHECK(hipSetDevice(c.device));
  if (n > 0) {
    DBuf<u32> d(c, (size_t)n);
    HIP_CHECK(hipMemcpyAsync(d.get(), data, (size_t)n * 4, hipMemcpyHostToDevice, c.stream));
    if (op == 0)
      scan_exclusive_u32(c, d.get(), d.get(), (size_t)n, nullptr);
    else
      scan_inclusive_max_u32(c, d.get(), d.get(), (size_t)n);
    HIP_CHECK(hipMemcpyAsync(data, d.get(), (size_t)n * 4, hipMemcpyDeviceToHost, c.stream));
    c.sync();
  }
  API_END
}
