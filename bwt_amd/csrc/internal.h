// Internal C++ interface between the translation units of libbwt_hip.so.
// Nothing here is part of the C-ABI (see include/bwt_hip.h for that).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <atomic>
#include <condition_variable>
#include <map>
#include <mutex>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

namespace bwtamd {

typedef uint8_t u8;
typedef uint16_t u16;
typedef uint32_t u32;
typedef uint64_t u64;
typedef int64_t i64;

// ---------------------------------------------------------------------------
// errors: internal code throws, the C-ABI layer converts to a status + message
// ---------------------------------------------------------------------------
struct Error : std::runtime_error {
  int code;
  Error(int code_, const std::string &what) : std::runtime_error(what), code(code_) {}
};

[[noreturn]] void fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));

#define HIP_CHECK(expr)                                                                        \
  do {                                                                                         \
    hipError_t e__ = (expr);                                                                   \
    if (e__ != hipSuccess)                                                                     \
      ::bwtamd::fail((int)e__, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, \
                     __LINE__);                                                                \
  } while (0)

// status codes of the C-ABI that are not hipError_t values (same values as include/bwt_hip.h)
#ifndef BWT_HIP_EINVAL
#define BWT_HIP_EINVAL 10001
#define BWT_HIP_ETEXT 10002
#define BWT_HIP_EINTERNAL 10003
#endif
#define BWT_HIP_ENOROUTE 10004  // internal: a build that needs a particular route was asked of a text that does not take it

// ---------------------------------------------------------------------------
// device context: one per (process, GPU).  All work runs on `stream`.
// ---------------------------------------------------------------------------
struct PhaseTimer {
  std::string name;
  double ms = 0;
  double bytes = 0;  // algorithmic bytes (or transactions) of the phase
  long launches = 0;
};

// Thrown by Ctx::wait_text when the thread that uploads a packed text has met a symbol outside the range the build
// was started with; nothing has read that part of the text yet.
struct TextMismatch {};

struct Ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  int n_cu = 256;

  // text
  const u8 *T = nullptr;  // device
  u8 *T_owned = nullptr;
  i64 n = 0;
  int max_sym = 0;  // largest symbol in T
  int min_sym = 0;  // smallest symbol in T besides the terminator (0 if there is none)
  int sym_bits = 0; // bits per symbol = floor(log2(max_sym)) + 1
  // A text whose symbols are scattered over the byte range (letters: A, C, G, T are 65, 67, 71, 84) is held on the
  // device with its symbols renumbered 1 .. sigma in their order (compact): keys, rank structures and wire formats then
  // see a dense alphabet -- as ASCII, DNA would spend five bits a symbol where two do.  Everything inside the library,
  // the blocks' L included, is in those codes; symbols are translated where they enter or leave through host memory
  // (sym_fwd: symbol -> code, 0 for symbols the text does not hold; sym_back: code -> symbol; d_sym_tab: both on the
  // device, 256 bytes each).
  bool compact_allowed = true, compact = false;
  u8 sym_fwd[256] = {}, sym_back[256] = {};
  u8 *d_sym_tab = nullptr;

  // A text that is still arriving (bwt_hip_pmbwt_host): its upload runs on `copy_stream` in chunks, each with
  // an event; wait_text(upto) makes `stream` wait for the chunks that cover T[0:upto) before work that reads
  // them is queued.  Empty when the text is resident: wait_text then does nothing.
  hipStream_t copy_stream = nullptr;
  struct TextChunk {
    u64 end;
    hipEvent_t ev;
    bool waited;
  };
  std::vector<TextChunk> chunks;
  std::vector<hipEvent_t> chunk_events;  // spare events of finished uploads (made without timing: not for the timers' pool)
  void wait_text(u64 upto);
  // A packed upload (host_transfer.hip) records the chunks' events from a thread of its own: wait_text first waits,
  // on the host, until the chunk's event has been recorded at all.
  std::mutex chunk_mu;
  std::condition_variable chunk_cv;
  size_t chunks_recorded = 0;
  std::atomic<bool> xfer_abort{false};
  std::atomic<bool> xfer_mismatch{false};  // the uploader met a symbol outside the range the build was started with
  unsigned char *xfer_stage[3] = {nullptr, nullptr, nullptr};  // page-locked staging buffers, made at first use
  hipEvent_t xfer_ev[3] = {nullptr, nullptr, nullptr};
  i64 xfer_chunk = (i64)256 << 20;  // symbols per chunk of a packed transfer (BWT_HIP_XFER_CHUNK; tests make it small)
  int host_pack = 1;  // texts of up to four consecutive symbol values cross the link as two bits per symbol: 0 never, 1 from 2^24 symbols, 2 always

  // caching allocator (sizes rounded up; freed blocks are kept for reuse)
  std::multimap<size_t, void *> free_blocks;
  std::unordered_map<void *, size_t> live_blocks;
  size_t bytes_live = 0, bytes_peak = 0, bytes_reserved = 0;

  // phase timing (hipEvent pairs on `stream`), enabled by timing_level > 0
  int timing_level = 0;
  const char *sort_tag = "";  // prefix of the radix sort's per-kernel timer names (who is sorting)
  // leaf windows of at least this many suffixes may take the bucket route of the leaf sort (two passes
  // on the top 16 key bits, the rest in LDS); below it the 65536 buckets are too small to pay
  u64 bucket_min_n = 100000000;  // measured crossover on random DNA: 7e7 suffixes 2.3 vs 1.9 ms, 1.4e8 suffixes 3.1 vs 3.6 ms
  // tests and tuning: take the wide bucket route wherever the alphabet allows it (not only where the packed
  // elements run out of position bits); bucket bits of that route (0 = by window length)
  // force_wide = 1 prefers the wide route, 2 the lean route (below), both wherever the alphabet allows them
  int force_wide = 0;
  int wide_bbits = 0;
  bool lean_any_order = true;
  u32 slot_sub = 8;    // areas per first digit where a first pass reserves its output that way (1 to 16; 8: one per XCD)
  int lean_slots = 1;  // the lean route's second pass reserves its output by atomic adds (no histogram of its digits);
                       // tests: 3: also where the slots cost far more memory than the elements (small texts), 2: with too
                       // few slots in the second passes, so that they are void and the ones with exact offsets follow,
                       // 4: the same for the first passes, 5: 3 and 2 together
  bool lean_base = true;  // alphabets of 3, 5, 6, 7 symbols: lean keys as numbers in that base (off: such windows are sorted as pairs)
  bool gap_lean = true;  // the gap histogram's grouping in two passes that keep neither digit (gap_sort_lean)
  // chains whose head does not come out of a short backward search are finished by a second launch that searches
  // the left block's suffix array (gap.hip); off: they restart from further and further right instead
  bool defer_chains = true;
  // Long stretches of T that repeat a unit of at most 8 symbols (runs of one symbol above all; every such stretch that
  // holds a whole aligned 4096-symbol block), found on the first merge of a text: the chains' suffix comparisons step
  // over them (gap.hip).  n_runs starts, then n_runs ends (exclusive), then n_runs periods, then n_runs last symbols.
  u32 *runs = nullptr;
  u32 n_runs = 0;
  std::vector<u32> runs_host;  // the same list on the host (the leaf sort sizes its window by it)
  bool runs_valid = false;
  bool window_long_run = false;  // the leaf window being sorted holds a listed run of one symbol of 4096 symbols or more
  int run_jump = 1;  // 0: the doubling rounds take runs of one symbol like anything else, 2: step over them in every window (tests)
  bool run_table = true;  // off: comparisons read every symbol (tests)
  // large leaf windows are probed for long repeats before a bucket route or the direct pass is tried on them
  bool repeat_probe = true;
  bool defer_buckets = true;  // the lean route's buckets with too many equal keys go to a general kernel instead of failing the route (off: tests)
  bool run_filter = true;  // long runs of one symbol leave the lean route's sort and are ranked in closed form (off: tests)
  // Ranks of a difference-cover sample of the whole text (DcSample below), built on first use by a leaf whose ties are
  // dense: 0 never, 1 when a leaf asks for them, 2 for every leaf that has ties at all (tests)
  int dc_mode = 1;
  struct DcSample {
    u32 v = 0, nd = 0;   // period and size of the cover
    u32 D[8] = {};       // its residues, ascending
    u32 n_s = 0;         // sample positions below n
    u32 *rank = nullptr;      // rank[k] of the suffix at sample k among the sample suffixes (n_s entries, pool memory)
    u16 *pair_tab = nullptr;  // v * v entries, see dc_pair_entry()
    u8 *lo_tab = nullptr;     // lo_tab[r] = residues of the cover below r (v entries, behind pair_tab in one block)
    bool valid = false;
  } dc;
  std::vector<PhaseTimer> timers;
  struct Pending {
    int idx;
    hipEvent_t e0, e1;
  };
  std::vector<Pending> pending;
  std::vector<hipEvent_t> event_pool;

  // small pinned host scratch for scalar read-backs
  u32 *h_scalars = nullptr;  // 64 words, pinned
  u32 *d_scalars = nullptr;  // 64 words, device

  // While `arena_on` is set, alloc() carves its blocks out of [arena, arena + arena_cap) -- this GPU's copy of
  // the library's static device array -- and never reaches the driver; release() of such a block does
  // nothing.  ArenaScope turns it on for the duration of one call.
  unsigned char *arena = nullptr;
  size_t arena_cap = 0, arena_used = 0;
  bool arena_on = false;

  void *alloc(size_t bytes);
  void release(void *p);
  void trim();  // return cached blocks to the driver
  void sync();

  int timer_index(const char *name);
  void flush_timers();
};

// RAII handle on pool memory
template <class T>
struct DBuf {
  Ctx *c = nullptr;
  T *p = nullptr;
  size_t n = 0;
  DBuf() {}
  DBuf(Ctx &ctx, size_t count) : c(&ctx), n(count) { p = (T *)ctx.alloc(count * sizeof(T)); }
  DBuf(const DBuf &) = delete;
  DBuf &operator=(const DBuf &) = delete;
  DBuf(DBuf &&o) : c(o.c), p(o.p), n(o.n) { o.p = nullptr; }
  DBuf &operator=(DBuf &&o) {
    if (this != &o) {
      reset();
      c = o.c;
      p = o.p;
      n = o.n;
      o.p = nullptr;
    }
    return *this;
  }
  ~DBuf() { reset(); }
  void reset() {
    if (p) c->release(p);
    p = nullptr;
  }
  T *get() const { return p; }
  operator T *() const { return p; }
};

// This GPU's copy of the library's static device array (context.hip) and its size.  The address is looked up
// once per device, with that device current; the array's name itself is never used as an address.
unsigned char *device_scratch(int device, size_t *bytes);

// All pool allocations inside the scope come from the static device array; phase timing (which creates
// events) is off.  The array is shared by every call on the GPU and keeps the last call's contents.
struct ArenaScope {
  Ctx &c;
  int saved_timing;
  explicit ArenaScope(Ctx &ctx) : c(ctx), saved_timing(ctx.timing_level) {
    c.arena = device_scratch(c.device, &c.arena_cap);
    c.arena_used = 0;
    c.arena_on = true;
    c.timing_level = 0;
  }
  ~ArenaScope() {
    c.arena_on = false;
    c.timing_level = saved_timing;
  }
};

// scoped phase timer: records two events on the stream when timing is on
struct Phase {
  Ctx &c;
  int idx = -1;
  hipEvent_t e0 = nullptr;
  Phase(Ctx &ctx, const char *name, double bytes = 0);
  ~Phase();
  // more algorithmic work, for phases whose route (and with it their work) is known only afterwards
  void add_work(double bytes) {
    if (idx >= 0) c.timers[(size_t)idx].bytes += bytes;
  }
};

// the same around ONE kernel launch, recorded only at timing_level >= 2
struct KernelTimer {
  Phase *ph = nullptr;
  KernelTimer(Ctx &ctx, const char *name, double bytes) {
    if (ctx.timing_level >= 2) ph = new Phase(ctx, name, bytes);
  }
  ~KernelTimer() { delete ph; }
};

static inline u32 ceil_div(u64 a, u64 b) { return (u32)((a + b - 1) / b); }

// ---------------------------------------------------------------------------
// primitives (scan.hip, radix_sort.hip)
// ---------------------------------------------------------------------------

// out[i] = sum_{k<i} in[k]  (u32, wraps); in and out may alias.  If total is
// not null, *total (device) receives the sum of all entries.
void scan_exclusive_u32(Ctx &c, const u32 *in, u32 *out, size_t n, u32 *total);
// out[i] = max_{k<=i} in[k]; in and out may alias.
void scan_inclusive_max_u32(Ctx &c, const u32 *in, u32 *out, size_t n);

// Stable LSD radix sort of n (key, val) pairs on key bits [begin_bit,end_bit).
// keys[0]/vals[0] hold the input; the other pair is scratch of the same size.
// vals may be {nullptr,nullptr} for keys only.  Returns which pair (0 or 1)
// holds the result.  n < 2^32.  `use` picks a kernel instantiation named after
// the caller (0 leaf sort, 1 refinement rounds and tests, 2 gap histogram).
// Keys that are a function of the text (round 0 of the leaf suffix sort), value(i) = a+i:
//   key(i) = | codes of the K0 symbols T[a+i .. a+i+K0), w bits each, first symbol highest | zero padding |
//            | T[a+i-1] as it is, 8 bits |
// The terminator and positions past the end have code 0.  Two codings:
//   one_based = 1: code(c) = c - min_sym + 1.  Exact: 0 is below every symbol.
//   one_based = 0: code(c) = c - min_sym, which saves a bit per symbol when the number of distinct symbols is
//     a power of two (DNA: 2 bits).  The terminator then looks like the smallest symbol.  Where keys differ
//     their order is still the order of the suffixes (a terminator coded as the smallest symbol can only turn
//     a strict inequality into an equality).  Among equal keys, the (at most K0) suffixes whose K0 symbols
//     reach the terminator precede all others, a shorter one before a longer one.  That is the order of
//     DECREASING position, which the sort produces without looking: its first pass leaves equal digits in
//     reverse input order and the later passes are stable.
struct TextKeys {
  const u8 *T;
  u64 n, a;
  int w, K0;
  u32 min_sym;
  int one_based;
  // 0: (key, value) pairs as above.  Otherwise PACKED 8-byte elements (K0 * w <= 32, window below 2^pbits):
  //   | codes, 32 bits | T[a+i-1], 32 - pbits bits | i, pbits bits |
  int pbits;
  // Packed elements only: if not null, the pass that builds the elements also writes the next pass's digit of
  // every element (bits next_shift .. next_shift+7) into this byte array, in output order, so that the next
  // pass's histogram reads one byte per element instead of eight.
  u8 *next_digits;
  int next_shift;
  // WIDE elements, for windows whose positions need all 32 bits (K0 * w <= 36, symbols below 16):
  //   element | key bits 63..32 | absolute text position, 32 bits |   side byte | key bits 31..28 | T[a+i-1], 4 bits |
  // made by the text pass when set; the side bytes travel with their elements through the later passes.
  int wide;
  // LEAN elements (lean != 0), for windows below 2^31 suffixes of small symbols: 8 bytes and nothing on the side,
  // because a radix pass need not keep the digit it sorts on -- after the pass that digit is known from where an
  // element lies.  With the key's top bits cut into | d1, b1 bits | d2, b2 bits | deep, lean_deep bits |:
  //   element  | d2 | deep | zero padding | T[a+i-1], vbits | i, pbits |
  // The text pass sorts on d1 (the key's top b1 bits) without storing it, the second pass on d2 over tiles that never
  // cross the regions the first pass made (equal d1) and writes region by region, so that the result is in the order
  // of (d1, d2) and the bucket of (d1, d2) starts where the second pass's tile counts, scanned per region, say.  The
  // top digit first: a GPU that keeps only a stretch of d1 values (LeanShard) ends up with a stretch of the leaf's
  // ranks.  K0 covers b1 + b2 + lean_deep bits.
  int lean, b1, b2, lean_deep, vbits;
  // base != 0 (lean elements of alphabets of 3, 5, 6 or 7 symbols): the K0 codes c - min_sym are the digits of a number
  // in that base instead of bit fields -- 24 symbols of a five-letter alphabet fit 56 bits where 3-bit fields hold 18 --
  // and key = number * scale in the bits 8..63 (scale = floor(2^56 / base^K0): the keys then reach up to the top of
  // their 56 bits and the top bits make buckets of even size).  lead = base^(K0-1).  Two keys that agree on their top
  // bits need not agree on any whole symbol: ties on such keys are compared on the text from the suffixes' first symbol.
  u32 base;
  u64 scale, lead;
  // keep_hi != 0 (lean text pass of a leaf that is spread over several GPUs): only the elements whose first digit lies
  // in [keep_lo, keep_hi) are written
  u32 keep_lo, keep_hi;
  // run_span != 0 (lean text pass of a window with long runs of one symbol): the suffixes whose key lies in
  // [run_lo, run_lo + run_span) -- those that begin with run_k copies of that symbol -- are left out of the sort (their
  // order is known in closed form, suffix_sort.hip); low byte of run_lo zero
  u64 run_lo, run_span;
  // ... and whose first run_k symbols really are run_sym (a key alone cannot always tell: the terminator, or
  // code_max).  code_max != 0 (bit-field codes only): run_sym, rare outside its runs or altogether, has no code of its
  // own, so that the keys use the key space of the alphabet without it: the symbols above it are coded one lower, and
  // where it stands itself a key shows the code of the symbol below it and, for every symbol behind it, the largest
  // code -- "that symbol, then nothing larger" -- which keeps the keys in the order of the suffixes; equal keys then
  // need not be equal symbols.
  u32 run_sym, run_k, code_max;
  // lean: the part of the buckets that the keys of an even text reach (0: all).  Keys as numbers in a base: the buckets
  // below that part of the key space, evenly; bit-field codes that leave code values unused: unevenly.
  float fill;
  int fill_even;
};
int radix_sort_u64(Ctx &c, u64 *keys[2], u32 *vals[2], size_t n, int begin_bit, int end_bit);
// Round 0 of the leaf suffix sort: sorts the (key, position) pairs of `text` on key bits [begin_bit, 64).
// Keys are stored split: buffer b holds the low words at ((u32*)keys[b])[0..n) and the high words
// `stride` 32-bit words further.  Returns the buffer that holds the result.
int leaf_sort_text_keys(Ctx &c, u64 *keys[2], u32 *vals[2], size_t n, u32 stride, int begin_bit,
                        const TextKeys &text);
// The same for packed elements (text.pbits != 0): sorts them on bits [begin_bit, 64), begin_bit >= 32.
// Bucket route of the packed leaf sort, handed from the sort to whoever finishes the buckets: elements sorted on
// their top 16 key bits, start[k] = first element of bucket k (65537 entries), bits begin_bit..47 still unsorted.
struct BucketState {
  DBuf<u32> start;
  int begin_bit = 0;
  int bbits = 16;      // a bucket = the elements that share their top bbits key bits
  u32 max_bucket = 0;  // elements of the largest bucket
  u32 slot = 0;        // lean elements only; not 0: bucket k's elements lie from k * slot on (start[] still holds its ranks)
  bool pending = false;
};
int leaf_sort_text_packed(Ctx &c, u64 *elems[2], size_t n, int begin_bit, const TextKeys &text, BucketState *bs,
                          bool try_buckets = true);
void bucket_sort_in_place(Ctx &c, u64 *e, size_t n, const BucketState &bs);
void bucket_histogram(Ctx &c, const TextKeys &tk, u64 first, u64 count, u64 stride, u32 *per_bucket, int bbits = 16);
int leaf_sort_text_wide(Ctx &c, u64 *elems[2], u8 *sides[2], size_t n, int bbits, u32 cap, const TextKeys &text,
                        BucketState *bs);
// Round 0 with lean elements: two passes, then bs->start holds the bounds of the 2^(b1+b2) buckets (bucket
// index = d1 << b2 | d2: the key's top b1 + b2 bits).  Returns the buffer with the result, or -1 if a bucket holds more than cap elements.
// shard (may be null): in: this GPU's part of `parts` (1: everything); out: the first digits it keeps, the leaf rank of
// its first suffix and how many it holds.  With parts > 1 the element buffers hold only this GPU's suffixes and
// bs->start[k] is the leaf rank of bucket k's first suffix (local index + first_rank).  elems: allocated here.
struct LeanShard {
  int part = 0, parts = 1;
  u32 dig_lo = 0, dig_hi = 0;
  u64 first_rank = 0, count = 0;
  std::vector<u64> bounds;  // out: leaf rank of every part's first suffix, parts + 1 entries (the same on every GPU)
  DBuf<u8> L;               // out (leaf builds that ask for L): this part's stretch of L, `count` symbols
};
int leaf_sort_text_lean(Ctx &c, DBuf<u64> (&elems)[2], size_t n, u32 cap, const TextKeys &text, BucketState *bs, LeanShard *shard = nullptr,
                        size_t *kept_out = nullptr);
// symbols of the key that the lean text pass's histogram rolls (what TextKeys::run_lo / run_span may be cut from)
int lean_count_symbols(const TextKeys &text);
int radix_sort_u32(Ctx &c, u32 *keys[2], u32 *vals[2], size_t n, int begin_bit, int end_bit, int use = 1);
// The gap histogram's grouping: the values of keys[0] grouped by their bits [low, low + b1 + b2), in two passes
// that keep neither digit; afterwards keys[0] holds the values' low `low` bits as 16-bit words, bucket k of them
// at [start[k], start[k+1]) (start: 2^(b1+b2) + 1 entries).
// slot_out (may be null): the caller can read buckets that are not contiguous; *slot_out != 0 then says that bucket k's
// words lie from k * *slot_out on, and start[k] is only the number of values in the buckets below k.  used_buckets: how
// many of the 2^(b1+b2) buckets the values can fall into (0: all), for the size of a slot.
void gap_sort_lean(Ctx &c, u32 *keys[2], size_t n, int low, int b1, int b2, u32 *start, u32 *slot_out = nullptr, u32 used_buckets = 0);

// ---------------------------------------------------------------------------
// a partial BWT resident on the device (include/bwt.h:122-149 of the reference)
// ---------------------------------------------------------------------------
struct WaveletDev {
  int levels = 0;
  int dbits = 2;         // code bits per level: 2 (alphabets of up to four symbols) or 4
  u32 min_sym = 1;       // code = symbol - min_sym
  u64 n_blk = 0;         // rank blocks per level
  DBuf<u64> planes;      // levels * n_blk * (8 or 16) words
  u32 start[4][16] = {}; // per level: first position of each digit's bucket on the next level
  DBuf<u32> lf_base;     // 512 entries: C[c] - p_end[c], then p_end[c]
  bool built = false;
};

struct Block {
  i64 a = 0, b = 0;
  u8 e_char = 0;
  i64 s_rank = -1, e_rank = -1;  // relative to a
  DBuf<u8> L;                    // b-a symbols
  DBuf<u32> samp_rank;           // relative to a, ordered by pos
  DBuf<u32> samp_pos;            // absolute text positions, ascending
  i64 n_samp = 0;
  i64 hist[256];                 // symbol counts of L (filled by count_alphabets)
  i64 C[257];                    // C[c] = #symbols < c in T[a:b)
  bool counted = false;
  WaveletDev wm;
  i64 len() const { return b - a; }
};

// What device code needs of Ctx::dc to compare two suffixes in at most v symbols and two look-ups (rank == null:
// there is no sample).  pair_tab[(i mod v) * v + (j mod v)]: bits 0..4 = the shift d that makes i + d and j + d
// sample positions, bits 5..8 / 9..12 = the sample positions in [i, i + d) / [j, j + d).  The compact index of a
// sample position p is (p / v) * nd + lo_tab[p mod v].
struct DcView {
  const u32 *rank;
  const u16 *pair_tab;
  const u8 *lo_tab;
  u32 v, nd, n_s;
};
static inline DcView dc_view(const Ctx &c) {
  if (!c.dc.valid) return DcView{nullptr, nullptr, nullptr, 0, 0, 0};
  return DcView{c.dc.rank, c.dc.pair_tab, c.dc.lo_tab, c.dc.v, c.dc.nd, c.dc.n_s};
}
// suffix_sort.hip: the ranks of the text's difference-cover sample (Ctx::dc), built on first use; the whole text must
// have arrived.  drop_dc_sample releases them (a new text).
void ensure_dc_sample(Ctx &c);
void drop_dc_sample(Ctx &c);
// suffix_sort.hip: SA of T[a:b) in full-text suffix order, absolute positions
DBuf<u32> leaf_suffix_array(Ctx &c, i64 a, i64 b);
// suffix_sort.hip: bwt_leaf -- the leaf's L, special ranks and samples
Block *build_leaf(Ctx &c, i64 a, i64 b, i64 ns);
// suffix_sort.hip: one GPU's share of a leaf that `parts` GPUs build together, cut by key range: every GPU reads the
// whole text, keeps the suffixes whose key's first digit falls into its stretch (stretches of about equal counts, cut
// the same way by every GPU from its own histogram: nothing is exchanged) and finishes them, which gives it the stretch
// [bounds[part], bounds[part+1]) of the leaf's L.  samp: the leaf's ns position-ordered sample slots (position << 32 |
// leaf rank), filled where the sampled suffix belongs to this share, ~0 elsewhere.  s_rank / e_rank: leaf-relative,
// -1 where the suffix is another share's.  False: the text or the leaf does not take the route this needs (the lean
// bucket route: large leaves without long repeats); nothing was built.
struct LeafShare {
  DBuf<u8> L;
  DBuf<u64> samp;
  std::vector<u64> bounds;
  i64 ns = 0, s_rank = -1, e_rank = -1;
};
bool build_leaf_share(Ctx &c, i64 a, i64 b, i64 ns, int part, int parts, LeafShare &out);
// the filled slots of a sample slot array, in order, as (rank + rank_base, position) pairs; returns their number
i64 compact_sample_slots(Ctx &c, const u64 *slots, i64 n_slots, u32 rank_base, u32 *rank_out, u32 *pos_out);
// wavelet.hip
void count_alphabets(Ctx &c, Block &blk);
void build_wavelet(Ctx &c, Block &blk);
void rank_batch(Ctx &c, Block &blk, const u8 *d_sym, const u32 *d_pos, u32 *d_out, size_t nq, bool lf);
// inverse.hip: the block decoded back to text (ibwt), SA[r] for a batch of ranks (sa)
void inverse_block(Ctx &c, Block &blk, u8 *out_dev, i64 *mismatches);
void sa_batch(Ctx &c, Block &blk, const u32 *d_rank, u32 *d_out, size_t nq);
// inverse.hip: what SA[rank] look-ups need besides the rank structure -- a bit per rank (is it sampled?) and the
// samples ordered by rank
struct RankIndex {
  DBuf<u32> bits, by_rank, by_rank_pos;
  u32 ns = 0;
};
void build_rank_index(Ctx &c, Block &blk, RankIndex &ix);
// gap.hip: the inclusive prefix sums S[i] = #{j : g_j <= i}, i in [0, m], of
// the gap array, plus g_j at the right block's sample positions
struct GapResult {
  DBuf<u32> S;       // m + 1 entries
  DBuf<u32> samp_g;  // one per sample of the right block
  DBuf<u32> g_text;  // optional tap: g_j for j = r.a .. r.b-1
  i64 seg = 0, n_chains = 0;
};
GapResult build_gap(Ctx &c, Block &l, Block &r, i64 seg, bool want_g);
// pieces of build_gap, used on their own when a merge is spread over several GPUs
struct ChainLayout {
  u64 ra, rb;  // right positions covered
  u32 seg, n_chains;
};
ChainLayout run_chains(Ctx &c, Block &l, i64 ra, i64 rb, i64 seg, DBuf<u32> &g0);
// the list of long periodic stretches of the text (Ctx::runs), built on first use; the whole text must have arrived
void ensure_run_table(Ctx &c);
void chains_text_order(Ctx &c, const ChainLayout &lay, const u32 *g0, u32 *out);
void gap_prefix(Ctx &c, u32 *g0, u32 *g1, u64 total, u64 m, u32 *S);
void merge_symbols(Ctx &c, const u8 *lL, u64 m, const u8 *rL, u64 rn, const u32 *S, u8 *out);
// The histogram's grouping on its own, for merges spread over GPUs whose g values travel grouped (gap.hip).
// gap_group16: g0 (total values in [0, m + 1], destroyed; g1 scratch of the same size) -> the values' low 13 bits
// as 16-bit words in g0, grouped by bucket = value >> 13; start[k] = first word of bucket k (gap_group16_starts(m)
// entries).  gap_prefix_grouped: S[i], i in [0, m], from the grouped words of `sources` senders (k_bucket_prefix16_multi).
bool gap_grouping_applies(const Ctx &c, u64 m);
size_t gap_group16_starts(u64 m);
void gap_group16(Ctx &c, u32 *g0, u32 *g1, u64 total, u64 m, u32 *start);
void gap_prefix_grouped(Ctx &c, const u16 *words, const u32 *word_off, const u32 *starts, u32 sources, u32 nbk, u64 m, u32 *S);
// merge.hip
Block *merge_blocks(Ctx &c, Block &l, Block &r, GapResult &gap, i64 ns);

// dist.hip: pieces of a merge spread over several GPUs
// shift: parts are cut at multiples of 2^shift left ranks (0: anywhere)
void partition_bounds(u64 m, u32 parts, i64 *lo_out, int shift = 0);
int partition_shift(const Ctx &c, u64 m);  // 13 where the g values of a merge travel grouped by gap bucket, else 0
// perm (may be null): perm[x] = index in `in` of the value that went to out[x]
void partition_by_left_rank(Ctx &c, const u32 *in, u64 count, u64 m, u32 parts, u32 *out, i64 *counts_host, u32 *perm = nullptr,
                            int shift = 0);
// Symbols between GPUs as two bits each (texts whose symbols span at most four consecutive values): `count`
// symbols at src (any alignment) -> wire_bytes(count) bytes at packed, and back to dst (any alignment).  The
// unpacking kernel reads up to WIRE_PAD bytes past the packed stream: buffers that receive one are that much longer.
bool wire_packable(const Ctx &c);
static inline size_t wire_bytes(u64 count) { return (size_t)((count + 3) / 4); }
constexpr size_t WIRE_PAD = 8;
void wire_pack2(Ctx &c, const u8 *src, u64 count, u8 *packed);
void wire_unpack2(Ctx &c, const u8 *packed, u64 count, u8 *dst);
// Samples that stay with their holders (dist.hip): look-up requests grouped by the member that answers them.
// left: queries[x] = left rank - first left rank of the answering part, perm[x] = the sample it belongs to;
// right: queries[i] = pos[i] - first position of the answering member's chains (samples are in position order, so
// the groups are too).  counts[d] = requests for member d.
void samples_route_left(Ctx &c, const u32 *rank_abs, u64 n, i64 a, u64 m, u32 members, u32 *queries, u32 *perm, i64 *counts,
                        int shift = 0);
void samples_route_right(Ctx &c, const u32 *pos, u64 n, i64 cpos, u64 r, u32 members, u32 *queries, i64 *counts);
// ans[x]: for the first n_left requests the new absolute rank of the left sample, for the n_right after them g
void samples_answer(Ctx &c, const u32 *q, u64 n_left, u64 n_right, const u32 *S_local, u64 i_lo, u64 right_before, i64 a,
                    const u32 *g_local, u32 *ans);
struct SamplePlace {
  const u32 *rank, *pos;  // the holder's samples (absolute ranks)
  u64 n;
  const u32 *ans, *perm;  // answers in request order; perm null: request x = sample x
  u32 right_shift;        // 0 for a left holder; for a right holder c - a: new rank = rank - right_shift + g
  u64 first_index;        // index of the holder's first sample in the merge's whole sequence (left by position, then right)
  u64 first_slot;         // samples_taken_before(first_index)
  u64 A, B;               // ns_out - 1, nl + nr - 1
  u32 *o_rank, *o_pos;    // the holder's kept samples, in order
};
void samples_place(Ctx &c, const SamplePlace &P);
u64 samples_taken_before(u64 i, u64 A, u64 B);  // samples kept among the first i of the sequence (include/bwt.h:412, :436 in closed form)

// api.hip: bwt_rec (leaf iff b-a <= th, else split at (a+b)/2 and merge); the calling thread's error message
Block *build_rec(Ctx &c, i64 a, i64 b, i64 th, i64 ns, i64 seg);
void set_last_error(const char *msg);

// host_transfer.hip
void probe_symbol_range(const u8 *T, i64 n, int *lo, int *hi);
void prepare_packed_upload(Ctx &c);
void upload_text_packed(Ctx &c, const u8 *T_host, i64 n, u32 min_sym, u32 span, u8 *dpacked, std::string *err);
void download_packed(Ctx &c, const u8 *L_dev, i64 n, u32 min_sym, i64 foreign, u8 *L_host);
void download_samples32(Ctx &c, const u32 *rank, const u32 *pos, i64 k, i64 a, i64 *rank_out, i64 *pos_out);

void device_gen_text(Ctx &c, u8 *dT, i64 n, u64 seed, int sigma);
void text_stats(Ctx &c, const u8 *dT, i64 n, int *max_sym, int *min_sym, i64 *n_zero, int *last);
// present[v >> 5] bit v & 31: the text holds symbol v
void symbol_presence(Ctx &c, const u8 *dT, i64 n, u32 present[8]);
// dst[i] = table[src[i]] (table: 256 bytes on the device; dst may be src)
void map_symbols(Ctx &c, const u8 *src, u8 *dst, u64 n, const u8 *d_table);
// count symbols from device memory to host memory, codes turned back into the text's own symbols where it is compact
void symbols_to_host(Ctx &c, const u8 *src_dev, u64 count, u8 *dst_host);
struct WmView;
WmView make_view(const Block &blk);

}  // namespace bwtamd

// the C-ABI's context handle
struct bwt_hip_ctx {
  bwtamd::Ctx c;
};
