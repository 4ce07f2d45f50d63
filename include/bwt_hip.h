/*
 * bwt_hip.h -- C-ABI of libbwt_hip.so, the MI355X (gfx950) implementation of
 * taura/bwt's build-leaf-then-merge BWT path.
 *
 * The reference has no FFI; its boundary is the header API of include/bwt.h
 * (bwt::pmbwt, bwt_rec, bwt_leaf, bwt_merge, struct bwt).  The drop-in C++
 * headers shipped next to this file (bwt.h, opts.h, util.h, ...) keep that API
 * and implement it with the entry points below.  Every entry point names the
 * reference interface it replaces (paths relative to the reference's root).
 *
 * Conventions
 *   - plain pointers and sizes only; int64_t indices (the reference's idx_t).
 *   - return value: 0 on success, otherwise a hipError_t value or one of
 *     BWT_HIP_E*; bwt_hip_last_error() returns the message of the calling
 *     thread's last failure.  Nothing aborts and nothing throws across the ABI.
 *   - one context per (process, GPU); a context and its blocks are used by one
 *     host thread at a time, like the reference (its global `stat`, util.h:209).
 *   - the text obeys the reference's contract (README.md:66-68): T[n-1] == 0
 *     and no other 0.  n <= 2^32 - 65536.
 *   - ranks taken or returned are absolute (rank r of block [a,b) lies in
 *     [a,b)), as in the reference.
 */
#ifndef BWT_HIP_H
#define BWT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct bwt_hip_ctx bwt_hip_ctx;     /* device context */
typedef struct bwt_hip_block bwt_hip_block; /* partial BWT of T[a:b) resident in HBM (struct bwt, include/bwt.h:122-149) */

#define BWT_HIP_EINVAL 10001    /* bad argument */
#define BWT_HIP_ETEXT 10002     /* text breaks the contract above */
#define BWT_HIP_EINTERNAL 10003 /* internal invariant failed */

const char *bwt_hip_last_error(void);

/* ---- context ---------------------------------------------------------- */

/* Creates a context on HIP device `device`.  Replaces nothing in the
   reference (which has no device); lifetime brackets what `mallocator mem`
   brackets in src/pmbwt.cc:317-332. */
int bwt_hip_create(int device, bwt_hip_ctx **out);
/* Also frees every block of the context that is still alive; their handles become invalid, and
   bwt_hip_block_free on such a handle does nothing. */
void bwt_hip_destroy(bwt_hip_ctx *ctx);

/* Page-locked host memory for T and L: copies between such buffers and the device run at the link's rate
   (about 55 GB/s here) instead of the 7-28 GB/s of pageable memory.  Any context's calls accept these
   pointers wherever a host pointer is taken.  No counterpart in the reference (the caller's new[]/malloc,
   src/pmbwt.cc:300-306). */
int bwt_hip_host_alloc(int64_t bytes, void **out);
void bwt_hip_host_free(void *p);

/* Copies the host text T[0:n) to the device and validates it.
   Replaces the `const alpha_t * T, idx_t n` arguments of bwt::pmbwt
   (include/bwt.h:867). */
int bwt_hip_set_text(bwt_hip_ctx *ctx, const uint8_t *T_host, int64_t n);

/* Uses a text already in device memory (borrowed, 16-byte aligned, must stay
   valid while the context uses it).  Same role as bwt_hip_set_text. */
int bwt_hip_set_text_device(bwt_hip_ctx *ctx, const uint8_t *T_dev, int64_t n);

/* Fills a device text with the synthetic stream of the benchmarks:
   x_i = splitmix64(seed + (i+1)*0x9E3779B97F4A7C15), T[i] = 1 + x_i % sigma,
   T[n-1] = 0.  Replaces gen_input of src/pmbwt.cc:34-48 (whose nrand48() % 4
   text is periodic and unusable for 4-letter alphabets). */
int bwt_hip_gen_text(bwt_hip_ctx *ctx, int64_t n, uint64_t seed, int sigma);

/* Copies the context's text back to the host (n bytes). */
int bwt_hip_get_text(bwt_hip_ctx *ctx, uint8_t *T_host);

/* A text whose symbols are scattered over the byte range (letters: A, C, G, T are 65, 67, 71, 84) is held on the
   device with its symbols renumbered 1 .. sigma in their order, so that sort keys, rank structures and the transfers
   between GPUs see a dense alphabet; the reference instead sizes everything by alpha_max (include/bwt.h:200-226,
   wavelet_matrix.h:95-110).  Every entry point that takes or returns symbols in HOST memory (the text, L, e_char,
   C[], the symbols of rank and LF queries, decoded blocks) uses the text's own symbols.  DEVICE memory that the
   library hands out or takes (bwt_hip_block_L_device, _export_device, _import_device, the bwt_hip_dist_* steps,
   bwt_hip_slice_L_device) holds the codes.  *compact = 1 if the current text is renumbered; code_of[v] = code of
   symbol v (0: not in the text), symbol_of[k] = symbol of code k; either table may be null.  No counterpart in the
   reference.  BWT_HIP_COMPACT=0 in the environment switches the renumbering off. */
int bwt_hip_symbol_codes(bwt_hip_ctx *ctx, int *compact, uint8_t code_of[256], uint8_t symbol_of[256]);

/* ---- the pipeline ------------------------------------------------------ */

/* bwt::bwt_leaf (include/bwt.h:615-635): BWT of T[a:b) from its suffix array
   in full-text suffix order, plus min(b-a, ssa_n_samples) samples. */
int bwt_hip_leaf(bwt_hip_ctx *ctx, int64_t a, int64_t b, int64_t ssa_n_samples, bwt_hip_block **out);

/* bwt::bwt_merge (include/bwt.h:791-847): merges adjacent blocks l=[a,c) and
   r=[c,b) into [a,b).  Consumes (frees) l and r.  build_gap_segment_sz is
   bwt_opt's field of that name (include/opts.h:112); 0 = 1024. */
int bwt_hip_merge(bwt_hip_ctx *ctx, bwt_hip_block *l, bwt_hip_block *r, int64_t ssa_n_samples,
                  int64_t build_gap_segment_sz, bwt_hip_block **out);

/* bwt::bwt_rec (include/bwt.h:854-865) over [a,b): leaf iff
   b-a <= bwt_rec_threshold, else split at (a+b)/2 and merge. */
int bwt_hip_rec(bwt_hip_ctx *ctx, int64_t a, int64_t b, int64_t bwt_rec_threshold, int64_t ssa_n_samples,
                int64_t build_gap_segment_sz, bwt_hip_block **out);

/* bwt::pmbwt (include/bwt.h:867-883): bwt_hip_rec over [0,n). */
int bwt_hip_pmbwt(bwt_hip_ctx *ctx, int64_t bwt_rec_threshold, int64_t ssa_n_samples,
                  int64_t build_gap_segment_sz, bwt_hip_block **out);

/* The longest leaf this GPU's free memory holds next to a text of n symbols (two 8-byte sort elements per leaf suffix
   plus the text, L and a merge's buffers): what bwt_opt::fit_leaves_to_device sizes the recursion tree with.  The
   reference derives its leaf size from mem_budget alone (include/opts.h:278-289). */
int bwt_hip_leaf_capacity(bwt_hip_ctx *ctx, int64_t n, int64_t *max_leaf);

/* bwt::pmbwt as the reference's caller sees it (include/bwt.h:867-883): T in host memory, L delivered to host
   memory (L_host may be null: the block stays on the device only).  Equivalent to bwt_hip_set_text +
   bwt_hip_pmbwt + bwt_hip_block_get_L, with the upload overlapped: the text goes up in chunks on a stream
   of its own, the first leaf's sort starts when its window has arrived, and the text's statistics are taken from
   the host copy meanwhile.  Page-locked buffers (bwt_hip_host_alloc) make both copies run at the link's rate.
   A text of 2^24 symbols or more whose symbols span at most four consecutive values (DNA) crosses the link as two
   bits per symbol in both directions, through page-locked staging the context keeps (192 MB): T and L may then be
   pageable memory at no cost, and host threads of this call (up to 16) do the packing, which also validates the
   text, so no separate pass over it precedes the build.
   The context holds the text afterwards, as after bwt_hip_set_text. */
int bwt_hip_pmbwt_host(bwt_hip_ctx *ctx, const uint8_t *T_host, int64_t n, int64_t bwt_rec_threshold,
                       int64_t ssa_n_samples, int64_t build_gap_segment_sz, uint8_t *L_host, bwt_hip_block **out);

/* Uploads a block computed elsewhere (host L[a:b) as the reference stores it in
   L + a, samples ordered by pos).  Lets bwt::bwt_merge run on blocks whose
   device copy was dropped.  Counterpart of passing `bwt& l` by reference
   (include/bwt.h:791). */
int bwt_hip_block_upload(bwt_hip_ctx *ctx, int64_t a, int64_t b, const uint8_t *L_host, int64_t s_rank,
                         int64_t e_rank, const int64_t *samp_rank, const int64_t *samp_pos, int64_t n_samples,
                         bwt_hip_block **out);

/* The same two transfers with DEVICE buffers owned by the caller (for example
   tensors that RCCL sends between GPUs): export copies L[a:b) and the samples
   (absolute ranks, positions) out of a block; import builds a block from such
   buffers.  The buffers must not be in use by other streams during the call;
   the copies are complete when the call returns.  No counterpart in the
   reference (it has one address space). */
int bwt_hip_block_export_device(bwt_hip_ctx *ctx, const bwt_hip_block *blk, uint8_t *L_dev, int64_t *samp_rank_dev,
                                int64_t *samp_pos_dev);
int bwt_hip_block_import_device(bwt_hip_ctx *ctx, int64_t a, int64_t b, const uint8_t *L_dev, int64_t s_rank,
                                int64_t e_rank, const int64_t *samp_rank_dev, const int64_t *samp_pos_dev,
                                int64_t n_samples, bwt_hip_block **out);

/* bwt::fini (include/bwt.h:174-182) */
void bwt_hip_block_free(bwt_hip_ctx *ctx, bwt_hip_block *blk);

/* ---- reading a block ---------------------------------------------------- */

/* The scalar fields of struct bwt (include/bwt.h:135-145); any out pointer may be null. */
int bwt_hip_block_info(const bwt_hip_block *blk, int64_t *a, int64_t *b, int64_t *s_rank, int64_t *e_rank,
                       int *e_char, int64_t *n_samples);

/* Copies L[a:b) (b-a bytes) to the host: what the reference leaves in L + a. */
int bwt_hip_block_get_L(bwt_hip_ctx *ctx, const bwt_hip_block *blk, uint8_t *L_host);

/* Device address of L[a:b) (valid until the block is freed or merged). */
int bwt_hip_block_L_device(const bwt_hip_block *blk, const uint8_t **L_dev);

/* The sampled suffix array (include/sampled_suffix_array.h:36-39 entries),
   ordered by pos; writes min(n_samples, cap) pairs. */
int bwt_hip_block_get_samples(bwt_hip_ctx *ctx, const bwt_hip_block *blk, int64_t *rank_out, int64_t *pos_out,
                              int64_t cap);

/* ---- per-phase taps (parity tests, host-side helpers) -------------------- */

/* sa_range (include/suffix_array.h:34-49): SA_host[r] = position of rank a+r. */
int bwt_hip_leaf_sa(bwt_hip_ctx *ctx, int64_t a, int64_t b, int64_t *SA_host);

/* bwt::init_extra (include/bwt.h:169-172): count_alphabets + wavelet matrix. */
int bwt_hip_block_init_extra(bwt_hip_ctx *ctx, bwt_hip_block *blk);

/* C[0..alpha_max] of include/bwt.h:200-226 (alpha_max + 1 entries, alpha_max <= 255). */
int bwt_hip_block_get_C(bwt_hip_ctx *ctx, bwt_hip_block *blk, int alpha_max, int64_t *C_out);

/* Batched wavelet_matrix::rank (include/wavelet_matrix.h:260-290):
   out[k] = occurrences of sym[k] in the block's L[0:pos[k]) with T[a-1] hidden;
   pos relative to a. */
int bwt_hip_block_rank(bwt_hip_ctx *ctx, bwt_hip_block *blk, int64_t nq, const uint8_t *sym, const int64_t *pos,
                       int64_t *out);

/* Batched bwt::LF_map_mod (include/bwt.h:253-259): absolute ranks in and out. */
int bwt_hip_block_LF(bwt_hip_ctx *ctx, bwt_hip_block *blk, int64_t nq, const uint8_t *sym, const int64_t *rank,
                     int64_t *out);

/* bwt::ibwt (include/bwt.h:276-285; the round trip of src/pmbwt.cc:64-81): decodes the block back to
   T[a:b).  The serial LF walk of the reference is cut at the block's samples, one device thread per
   sample.  I_host (b-a bytes) may be null; *mismatches (may be null) = positions where the decoded text
   differs from the context's text, plus samples whose rank the walk does not reach: 0 iff L and the sample
   set describe T[a:b). */
int bwt_hip_block_ibwt(bwt_hip_ctx *ctx, bwt_hip_block *blk, uint8_t *I_host, int64_t *mismatches);

/* Batched bwt::sa (include/bwt.h:456-481): pos_out[k] = text position of the suffix of absolute rank rank[k]. */
int bwt_hip_block_sa(bwt_hip_ctx *ctx, bwt_hip_block *blk, int64_t nq, const int64_t *rank, int64_t *pos_out);

/* The ranks of a difference-cover sample of the whole text: what README.md:109 of the reference lists as missing there
   ("a suffix array of sampled characters from the whole input to guarantee a single string comparison does not take too
   long").  Sample positions are those p with p mod *period in a perfect difference set, so that any two suffixes i, j
   have a shift d < *period with i + d and j + d both sampled: they compare in at most *period symbols plus two ranks.
   The library builds the sample by itself when a leaf's ties are dense (texts with long repeats) and then settles the
   leaf's ties and the chains' suffix comparisons with it; this call builds it if need be and copies it out:
   pos_out[k] ascending, rank_out[k] = rank of T[pos_out[k]:] among the sampled suffixes, min(*n_out, cap) entries.
   Any output pointer may be null. */
int bwt_hip_dc_sample(bwt_hip_ctx *ctx, int64_t cap, int64_t *pos_out, int64_t *rank_out, int64_t *n_out, int *period);

/* build_gap (include/bwt.h:730-787) without consuming the blocks:
   gap_out[i - l.a] = gap.get(i) for i in [l.a, l.b] (l.b-l.a+1 entries);
   g_out[j - r.a] = insertion point of T[j:] among l's suffixes, absolute
   (r.b-r.a entries).  Either may be null. */
int bwt_hip_gap(bwt_hip_ctx *ctx, bwt_hip_block *l, bwt_hip_block *r, int64_t build_gap_segment_sz,
                int64_t *gap_out, int64_t *g_out);

/* ---- one merge spread over several GPUs ---------------------------------- */
/* The reference merges in one address space (include/bwt.h:791-847); these entry points are the same
   steps -- build_gap's LF chains (:730-787), the gap prefix sums, the interleaving of the two blocks and
   resample_sa (:366-453) -- cut so that G processes, one per GPU, each do a share and exchange data in
   between (bwt_amd/multi_gpu.py drives them).  The merge of a left block of m suffixes and a right block is
   cut by LEFT RANK into `parts` parts: part k owns the left ranks [lo[k], lo[k+1]) and the right suffixes
   whose insertion point g falls there; its output is a contiguous slice of the merged block.
   All *_dev pointers are device memory of the context's GPU. */

/* lo[0..parts]: first left rank (equally: first g value) of every part; lo[parts] = m + 1. */
int bwt_hip_dist_bounds(int64_t m, int parts, int64_t *lo);
/* The same bounds as the context's GPU would cut them, and *shift: 0, or 13 for left blocks whose g values travel
   GROUPED (those of 2^21 suffixes and more, where the gap histogram groups its values in two passes): parts then
   begin at multiples of 2^shift left ranks, part(v) = ((v >> shift) * scale) >> 32, and the merge's steps are
   bwt_hip_dist_chains_grouped and bwt_hip_dist_merge_slice_grouped instead of dist_chains + dist_partition and
   dist_merge_slice.  All processes of a merge get the same answer (it depends on m only). */
int bwt_hip_dist_layout(bwt_hip_ctx *ctx, int64_t m, int parts, int64_t *lo, int *shift);
/* dist_chains plus the grouping of the values on the sending side: g_text_dev as dist_chains' g_dev; words_dev
   receives the values' low 13 bits as 16-bit words grouped by bucket = value >> 13 (any order inside a bucket),
   starts_dev[k] = first word of bucket k for k in [0, *n_buckets], *n_buckets = (m >> 13) + 1, *n_words = their number.
   The words of part d are those of its buckets [lo[d] >> 13, lo[d+1] >> 13) (the last part: up to *n_buckets); they and
   the matching stretch of starts_dev (one entry more than buckets) are what travels.  The last chain, if short, is
   filled up with the value m + 1: *fillers of those lie in the last bucket sent (the receiver skips them: they are
   above its last gap index) and have to be taken off the last part's count of right suffixes.  words_cap: capacity of
   words_dev in words, at least rb - ra rounded up to whole chains. */
int bwt_hip_dist_chains_grouped(bwt_hip_ctx *ctx, bwt_hip_block *l, int64_t ra, int64_t rb, int64_t build_gap_segment_sz,
                                uint32_t *g_text_dev, uint16_t *words_dev, int64_t words_cap, uint32_t *starts_dev,
                                int64_t *n_words, int64_t *n_buckets, int64_t *fillers);
/* dist_merge_slice from grouped values: sender s's words for this part lie at words_dev[word_off[s] ...) with the
   bounds of the part's nbk buckets at starts_dev[s * (nbk + 1) ...] (as the sender's starts_dev held them: only
   differences are used). */
int bwt_hip_dist_merge_slice_grouped(bwt_hip_ctx *ctx, const uint8_t *lL_dev, int64_t mL, const uint8_t *rL_dev, int64_t rn,
                                     const uint16_t *words_dev, const int64_t *word_off, const uint32_t *starts_dev, int sources,
                                     int64_t nbk, uint8_t *out_dev, uint32_t *S_dev);
/* g_dev[j - ra] = number of suffixes of block l smaller than T[j:], for the positions j in [ra, rb), ra >= l.b
   (the block needs L, s_rank, e_rank only: one imported with n_samples = 0 will do). */
int bwt_hip_dist_chains(bwt_hip_ctx *ctx, bwt_hip_block *l, int64_t ra, int64_t rb, int64_t build_gap_segment_sz,
                        uint32_t *g_dev);
/* Groups `count` g values by part: out_dev holds part 0's values, then part 1's, ... (any order inside a
   part), each minus lo[part]; counts[k] = size of part k. */
int bwt_hip_dist_partition(bwt_hip_ctx *ctx, const uint32_t *g_dev, int64_t count, int64_t m, int parts,
                           uint32_t *out_dev, int64_t *counts);
/* Symbols between GPUs.  The blocks of a merge travel as they are in the reference's one address space
   (include/bwt.h:791-847 reads l.L and r.L in place); between GPUs a stretch of L crosses the link as two bits
   per symbol when the text allows it: *packable = 1 iff the text's symbols (the terminator aside) span at most
   four consecutive values.  pack: `count` symbols at src_dev (any alignment) -> (count + 3) / 4 bytes at
   packed_dev, code = (symbol - smallest symbol) & 3, first symbol in the lowest bits.  unpack: the reverse into
   dst_dev (any alignment); it reads up to BWT_HIP_WIRE_PAD bytes past the packed stream, so the buffer that holds
   one is that much longer.  The terminator, which the block that starts at text position 0 holds at s_rank, has no
   code: the receiver puts it back. */
#define BWT_HIP_WIRE_PAD 8
int bwt_hip_dist_wire_packable(bwt_hip_ctx *ctx, int *packable);
int bwt_hip_dist_wire_pack(bwt_hip_ctx *ctx, const uint8_t *src_dev, int64_t count, uint8_t *packed_dev);
int bwt_hip_dist_wire_unpack(bwt_hip_ctx *ctx, const uint8_t *packed_dev, int64_t count, uint8_t *dst_dev);
/* One part: lL_dev = the part's mL left symbols (rank order), rL_dev = its rn right symbols (rank order),
   g_dev = their rn insertion points relative to the part's first left rank, any order (destroyed).
   out_dev[0 : mL+rn) = the part's slice of the merged block; S_dev[i] (mL+1 entries) = right symbols of
   the part placed before its left symbol i. */
int bwt_hip_dist_merge_slice(bwt_hip_ctx *ctx, const uint8_t *lL_dev, int64_t mL, const uint8_t *rL_dev, int64_t rn,
                             uint32_t *g_dev, uint8_t *out_dev, uint32_t *S_dev);
/* bwt_hip_dist_merge_slice for callers that must not reach hipMalloc / hipFree (tight serving loops, harnesses
   that forbid allocation).  No counterpart in the reference.  Same arguments and results;
   the scratch the kernels need (the histogram sort's second buffer, its tile counts and scan partials) lies in
   a static device array that the library holds on every GPU it runs on (24 MiB, until the process ends)
   instead of pool memory: the call reaches no hipMalloc / hipFree.  Limit: rn <= BWT_HIP_NOALLOC_MAX_RIGHT for any
   mL, larger calls fail with BWT_HIP_EINVAL before any work.  The call synchronises the context's stream (it is not
   for use under HIP graph capture).  Phase timers are not recorded for this call.  All calls
   on one GPU share the array: two that overlap in time (two contexts or streams on one GPU) race. */
#define BWT_HIP_NOALLOC_MAX_RIGHT 4194304
int bwt_hip_dist_merge_slice_noalloc(bwt_hip_ctx *ctx, const uint8_t *lL_dev, int64_t mL, const uint8_t *rL_dev,
                                     int64_t rn, uint32_t *g_dev, uint8_t *out_dev, uint32_t *S_dev);
/* The merged block's samples (resample_sa, include/bwt.h:366-453), with the samples left where they are: every
   process keeps the samples of a block whose positions lie in the text range it built its own subtree on (sorted
   by position; the processes in order hold the block's samples in order; ranks are block-relative + the block's
   first position, 32-bit words).  In a merge of [a,c) and [c,b) over `members` processes a holder asks, per sample,
   the process that knows the number its new rank needs:
   route   is_left: queries_dev[x] = left rank - first left rank of the part that owns it (parts as bwt_hip_dist_layout
           cuts them: its shift), grouped by part, and
           perm_dev[x] = the sample the request belongs to; otherwise queries_dev[i] = pos[i] - first position of the
           member whose chains cover pos[i] (member d covers [c + d*r/members, c + (d+1)*r/members)), which are grouped
           already.  counts[d] = requests for member d.
   answer  the requests a process received, those of left holders first (n_left), then those of right holders
           (n_right): answers_dev[x] = new rank of the left sample (from the part's S_dev, its first left rank i_lo and
           the right suffixes in lower parts) resp. g of the right sample (g_dev: text order, from the first position
           of this process's chains).
   place   the holder's answers, in request order, become its share of the merged samples: sample i of the holder
           has index first_index + i among the n_all samples of both blocks; the kept ones (the reference's thinning
           to ssa_n_samples_out) go to o_rank_dev / o_pos_dev in order, *kept of them.  With null outputs only *kept. */
int bwt_hip_dist_samples_route(bwt_hip_ctx *ctx, int is_left, const uint32_t *rank_dev, const uint32_t *pos_dev, int64_t n,
                               int64_t a, int64_t c, int64_t m, int64_t r, int members, int shift, uint32_t *queries_dev,
                               uint32_t *perm_dev, int64_t *counts);
int bwt_hip_dist_samples_answer(bwt_hip_ctx *ctx, const uint32_t *queries_dev, int64_t n_left, int64_t n_right,
                                const uint32_t *S_dev, int64_t i_lo, int64_t right_before, int64_t a, const uint32_t *g_dev,
                                uint32_t *answers_dev);
int bwt_hip_dist_samples_place(bwt_hip_ctx *ctx, int is_left, const uint32_t *rank_dev, const uint32_t *pos_dev, int64_t n,
                               const uint32_t *answers_dev, const uint32_t *perm_dev, int64_t a, int64_t c,
                               int64_t first_index, int64_t n_all, int64_t ssa_n_samples_out, uint32_t *o_rank_dev,
                               uint32_t *o_pos_dev, int64_t *kept);

/* bwt::bwt_leaf (include/bwt.h:615-635) of T[a:b) shared by `parts` GPUs, cut by KEY RANGE: every GPU reads the whole
   text and builds the stretch of the leaf's suffix order whose sort keys begin with its share of first digits.  The
   shares are cut into about equal numbers of suffixes by every GPU from its own count over the text -- the same cut
   everywhere, nothing exchanged -- so GPU `part` ends up with the ranks [*first_rank, *first_rank + *count) of the
   leaf (relative to a): L_dev receives that stretch of L (up to cap symbols; a larger share fails with BWT_HIP_EINVAL),
   slots_dev (min(b-a, ssa_n_samples) entries, may be null) the leaf's position-ordered sample slots as
   (position << 32 | leaf rank), filled where the sampled suffix belongs to this share and all ones elsewhere;
   *s_rank / *e_rank are -1 where another share holds that suffix; bounds (parts + 1 entries, may be null) receives
   every share's first rank.  *applies = 0 (and nothing else written): the leaf does not take the route this needs
   (the lean bucket route: 1e8 suffixes and more, no long repeats); it is then built whole, or as a tree.
   bwt_hip_dist_pmbwt does this when its tree is a single leaf. */
int bwt_hip_dist_leaf(bwt_hip_ctx *ctx, int64_t a, int64_t b, int64_t ssa_n_samples, int part, int parts, uint8_t *L_dev,
                      int64_t cap, uint64_t *slots_dev, int64_t *first_rank, int64_t *count, int64_t *s_rank,
                      int64_t *e_rank, int64_t *bounds, int *applies);

/* ---- the whole build over several GPUs -------------------------------------------------------------- */
/* The reference builds in one address space.  Here the recursion tree of bwt::bwt_rec (include/bwt.h:854-865) is cut
   at depth log2(world): rank k builds the k-th subtree on its own GPU, and every merge above (include/bwt.h:791-847)
   runs on all GPUs of its subtree through the bwt_hip_dist_* steps, with the left block, the g values, the right
   symbols and the samples exchanged in between.  A rank is a (context, transport) pair: a process per GPU over
   RCCL, or a thread per GPU inside one process over peer copies.  Every rank holds the whole text. */

/* One send or receive of device memory (bytes) to / from rank `peer`. */
typedef struct bwt_hip_p2p {
  int peer;
  int is_send;
  void *buf;
  int64_t bytes;
} bwt_hip_p2p;

/* How ranks move device memory between them.  exchange() performs all `n_ops` transfers of the calling rank and
   returns when they are complete; the matching calls of the peers list the same transfers from their side, in the
   same order per pair of ranks.  A caller may bring its own (MPI, a test harness); two ship with the library. */
typedef struct bwt_hip_transport {
  void *self;
  int rank, world;
  int (*exchange)(void *self, int n_ops, const bwt_hip_p2p *ops);
  void (*destroy)(void *self);
} bwt_hip_transport;

/* RCCL over xGMI, one process (or thread) per GPU: ncclSend/ncclRecv grouped per exchange.  id: 128 bytes from
   bwt_hip_rccl_unique_id on rank 0, handed to the other ranks by the launcher (file, MPI, torch.distributed store).
   librccl.so is loaded when the first of these is called. */
int bwt_hip_rccl_unique_id(uint8_t id[128]);
int bwt_hip_transport_rccl(bwt_hip_ctx *ctx, const uint8_t id[128], int rank, int world, bwt_hip_transport **out);
/* Ranks that are threads of this process: `world` transports that copy peer to peer (hipMemcpyPeerAsync over
   xGMI between GPUs; plain device copies where ranks share a GPU).  out[k] belongs to rank k. */
int bwt_hip_transport_local(int world, bwt_hip_transport **out);
void bwt_hip_transport_free(bwt_hip_transport *t);

/* This rank's share of a block that is spread over the ranks: slice `rank` of L (by rank range) and the block's
   samples whose positions lie in the text range this rank built its own subtree on (ranks in order hold the
   samples in position order).  s_rank is known to rank 0, e_rank to the last rank. */
typedef struct bwt_hip_slice bwt_hip_slice;
/* bwt::pmbwt (include/bwt.h:867-883) over t->world ranks (a power of two); called by every rank with its own
   context, all of which hold the same text.  The tree is the reference's for this bwt_rec_threshold; it must have
   at least `world` leaves at depth log2(world) (BWT_HIP_EINVAL otherwise) -- or be a single leaf
   (bwt_rec_threshold >= n), which the ranks then share by key range (bwt_hip_dist_leaf: no merge at all; rank k ends
   up with slice k of L and the samples of the k-th stretch of sample slots). */
int bwt_hip_dist_pmbwt(bwt_hip_ctx *ctx, bwt_hip_transport *t, int64_t bwt_rec_threshold, int64_t ssa_n_samples,
                       int64_t build_gap_segment_sz, bwt_hip_slice **out);
/* first: rank of the slice's first symbol; s_rank / e_rank as bwt_hip_block_info, -1 on the ranks that do not
   know them; n_samples: samples this rank holds.  Any output pointer may be null. */
int bwt_hip_slice_info(const bwt_hip_slice *s, int64_t *first, int64_t *size, int64_t *s_rank, int64_t *e_rank,
                       int64_t *n_samples);
int bwt_hip_slice_L_device(const bwt_hip_slice *s, const uint8_t **L_dev);
int bwt_hip_slice_get_L(bwt_hip_ctx *ctx, const bwt_hip_slice *s, uint8_t *L_host);
int bwt_hip_slice_get_samples(bwt_hip_ctx *ctx, const bwt_hip_slice *s, int64_t *rank_out, int64_t *pos_out, int64_t cap);
void bwt_hip_slice_free(bwt_hip_slice *s);

/* The same in one call for a host program that owns several GPUs: one thread and one context per entry of
   `devices` (a device may be listed more than once: its ranks then share it), the text uploaded to each, slices
   and samples gathered into L_host[0:n) and the sample arrays.  Sample outputs as bwt_hip_block_get_samples;
   *n_samples receives their number.
   n_devices = 1 is the plain single-GPU build.  If the tree has fewer than n_devices leaves at that depth, the
   largest power of two of GPUs that it feeds is used; *used (may be null) says how many. */
int bwt_hip_pmbwt_multi(const int *devices, int n_devices, const uint8_t *T_host, int64_t n, int64_t bwt_rec_threshold,
                        int64_t ssa_n_samples, int64_t build_gap_segment_sz, uint8_t *L_host, int64_t *s_rank,
                        int64_t *e_rank, int64_t *samp_rank, int64_t *samp_pos, int64_t samp_cap, int64_t *n_samples,
                        int *used);

/* ---- observability ------------------------------------------------------- */

/* level > 0 records a hipEvent pair around every phase on the context's stream. */
int bwt_hip_set_timing(bwt_hip_ctx *ctx, int level);
int bwt_hip_reset_timers(bwt_hip_ctx *ctx);
/* Number of named phases; then per phase its name, accumulated milliseconds,
   algorithmic bytes (transactions for gap_chains) and launch count.  Stands in
   for ts_stat (include/util.h:138-209). */
int bwt_hip_timer_count(bwt_hip_ctx *ctx);
int bwt_hip_timer_get(bwt_hip_ctx *ctx, int idx, const char **name, double *ms, double *work, int64_t *calls);

/* Device memory accounting, the counterpart of mallocator's counters
   (include/util.h:267-276): bytes live now, high-water mark, bytes held from the driver. */
int bwt_hip_mem_stats(bwt_hip_ctx *ctx, int64_t *live, int64_t *peak, int64_t *reserved);
/* Returns cached device memory to the driver. */
int bwt_hip_trim(bwt_hip_ctx *ctx);

/* The HIP stream all of the context's work is queued on (a hipStream_t). */
void *bwt_hip_stream(bwt_hip_ctx *ctx);
/* Waits for all queued work. */
int bwt_hip_sync(bwt_hip_ctx *ctx);

/* ---- primitives exposed for unit tests ----------------------------------- */

/* Stable LSD radix sort of host arrays on the device, key bits [begin_bit,end_bit);
   vals may be null.  u32 keys without values on 17 or 18 bits take the gap histogram's variant (two passes of
   9-bit digits, the first in any order): sorted on those bits, not stable. */
int bwt_hip_test_sort_u64(bwt_hip_ctx *ctx, uint64_t *keys, uint32_t *vals, int64_t n, int begin_bit, int end_bit);
int bwt_hip_test_sort_u32(bwt_hip_ctx *ctx, uint32_t *keys, uint32_t *vals, int64_t n, int begin_bit, int end_bit);
/* op 0: exclusive sum, op 1: inclusive max; in place on a host array. */
int bwt_hip_test_scan_u32(bwt_hip_ctx *ctx, uint32_t *data, int64_t n, int op);

#ifdef __cplusplus
}
#endif
#endif /* BWT_HIP_H */
