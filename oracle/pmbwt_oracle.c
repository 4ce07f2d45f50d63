/*
 * pmbwt_oracle.c -- CPU restatement of the reference's build-leaf-then-merge
 * BWT pipeline (taura/bwt).
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the checker that the parity tests,
 * the smoke test and the bench's cpu_baseline leg compare the HIP path with.
 * Nothing in the product path links, loads or calls it.
 *
 * Parity pinned: every function below was checked against the reference itself,
 * compiled from its own sources into oracle/_ref (see oracle/Makefile and
 * tests/golden/make_golden.py); the committed fixtures in tests/golden/ are
 * outputs of that reference build.
 *
 * Each function cites the reference lines it restates (paths are relative to
 * the reference's root).  The storage tricks of the reference that do not
 * change any observable value (8-bit gap counters with an overflow hash,
 * include/gap_array.h:20-29) are restated by their meaning only.
 *
 * Plain C99, serial.
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef int64_t idx_t;  /* include/typedefs.h:21 */
typedef uint8_t alpha_t; /* include/typedefs.h:14 */

#define ORC_SUPERBLOCK_BITS (1 << 15) /* include/util.h:236 */
#define ORC_BLOCK_BITS (1 << 6)       /* include/util.h:237 */

static idx_t imin(idx_t x, idx_t y) { return x < y ? x : y; }
static idx_t imax(idx_t x, idx_t y) { return x > y ? x : y; }

static void die(const char *msg) {
  fprintf(stderr, "pmbwt_oracle: %s\n", msg);
  abort();
}

static void *xmalloc(size_t sz) {
  void *p = malloc(sz ? sz : 1);
  if (!p) die("out of memory");
  return p;
}

/* ------------------------------------------------------------------ */
/* options: include/opts.h:36-202 (fields), :214-291 (set_defaults)    */
/* ------------------------------------------------------------------ */
typedef struct {
  double mem_budget;
  int alpha_min, alpha_max;
  idx_t ssa_n_samples;
  idx_t bwt_rec_threshold;
  idx_t build_gap_segment_sz;
  idx_t gap_sum_gran;
} orc_opt;

/* include/opts.h:204-209 */
static idx_t max_num_ssa(idx_t n, idx_t th) {
  idx_t k = 0;
  while (n > th) {
    n = n - n / 2;
    k++;
  }
  return k;
}

/* include/opts.h:214-291.  Zero fields mean "unset".  Only the fields that
   change a computed value are kept; the *_gran fields are CPU grain sizes. */
void orc_set_defaults(orc_opt *o, idx_t n) {
  if (o->alpha_min == 0 && o->alpha_max == 0) {
    o->alpha_min = 1;
    o->alpha_max = 255;
  }
  if (o->build_gap_segment_sz == 0) o->build_gap_segment_sz = 1024;
  if (o->gap_sum_gran == 0) o->gap_sum_gran = 128;
  if (o->mem_budget == 0) o->mem_budget = 4.0;
  if (o->mem_budget < 2.5) o->mem_budget = 4.0;
  if (o->mem_budget >= 2 + sizeof(idx_t) * 2) {
    o->bwt_rec_threshold = n;
    o->ssa_n_samples = 2;
  } else {
    double p = 0.8;
    o->bwt_rec_threshold = (idx_t)(((o->mem_budget - 2) * p * n) / (sizeof(idx_t) * 2));
    o->bwt_rec_threshold = imax(1000, o->bwt_rec_threshold);
    idx_t ssa_total_bytes = (idx_t)((o->mem_budget - 2) * (1 - p) * n);
    idx_t ssa_bytes = ssa_total_bytes / (1 + max_num_ssa(n, o->bwt_rec_threshold));
    o->ssa_n_samples = ssa_bytes / (sizeof(idx_t) * 2);
    o->ssa_n_samples = imax(1000, o->ssa_n_samples);
  }
}

/* ------------------------------------------------------------------ */
/* suffix comparison and leaf suffix array: include/suffix_array.h     */
/* ------------------------------------------------------------------ */

/* include/suffix_array.h:20-27: T[a:] < T[b:], run to the end of the text
   (the unique 0 at T[n-1] ends every comparison of two distinct suffixes). */
int orc_suffix_less(const alpha_t *T, idx_t n, idx_t a, idx_t b) {
  (void)n;
  for (;; a++, b++) {
    if (T[a] < T[b]) return 1;
    if (T[a] > T[b]) return 0;
  }
}

typedef struct {
  const alpha_t *T;
  idx_t n;
} cmp_ctx;

static int cmp_suffix_qsort(const void *x, const void *y, void *c_) {
  const cmp_ctx *c = (const cmp_ctx *)c_;
  idx_t a = *(const idx_t *)x, b = *(const idx_t *)y;
  if (a == b) return 0;
  return orc_suffix_less(c->T, c->n, a, b) ? -1 : 1;
}

/* include/suffix_array.h:34-49: SA[0:b-a) = positions a..b-1 ordered by the
   full-text suffix order.  (The reference indexes SA from a; here from 0.) */
void orc_sa_range(const alpha_t *T, idx_t n, idx_t a, idx_t b, idx_t *SA) {
  for (idx_t i = a; i < b; i++) SA[i - a] = i;
  cmp_ctx c = {T, n};
  qsort_r(SA, (size_t)(b - a), sizeof(idx_t), cmp_suffix_qsort, &c);
}

/* ------------------------------------------------------------------ */
/* succinct bit vector: include/succinct_bit_vector.h                  */
/* ------------------------------------------------------------------ */
typedef struct {
  const uint8_t *bits; /* LSB-first inside a byte, :185-190 */
  idx_t begin, end;    /* bit offsets of the slice */
  idx_t n_super, n_blk;
  idx_t *super;  /* ones before each superblock */
  uint16_t *blk; /* ones before each block, inside its superblock */
  idx_t n_zeros;
} orc_sbv;

/* ones in slice bits [b,e): :149-182 (the reference counts by bytes and
   words; the value is the plain bit count). */
static idx_t sbv_count(const orc_sbv *s, idx_t b, idx_t e) {
  idx_t c = 0;
  idx_t p = b + s->begin, q = e + s->begin;
  while (p < q && (p & 7)) {
    c += (s->bits[p >> 3] >> (p & 7)) & 1;
    p++;
  }
  while (p + 8 <= q) {
    c += __builtin_popcount(s->bits[p >> 3]);
    p += 8;
  }
  while (p < q) {
    c += (s->bits[p >> 3] >> (p & 7)) & 1;
    p++;
  }
  return c;
}

static idx_t sbv_rank1(const orc_sbv *s, idx_t i) { /* :264-283 */
  idx_t c0 = s->super[i / ORC_SUPERBLOCK_BITS];
  idx_t c1 = s->blk[i / ORC_BLOCK_BITS];
  idx_t c2 = sbv_count(s, i - i % ORC_BLOCK_BITS, i);
  return c0 + c1 + c2;
}

static idx_t sbv_rank0(const orc_sbv *s, idx_t i) { return i - sbv_rank1(s, i); } /* :286-288 */

static void sbv_init(orc_sbv *s, const uint8_t *bits, idx_t begin, idx_t end) { /* :193-250 */
  s->bits = bits;
  s->begin = begin;
  s->end = end;
  idx_t n = end - begin;
  idx_t ns = n / ORC_SUPERBLOCK_BITS + 1;
  idx_t nb = n / ORC_BLOCK_BITS + 1;
  idx_t per = ORC_SUPERBLOCK_BITS / ORC_BLOCK_BITS;
  s->n_super = ns;
  s->n_blk = nb;
  s->super = (idx_t *)xmalloc(sizeof(idx_t) * ns);
  s->blk = (uint16_t *)xmalloc(sizeof(uint16_t) * nb);
  for (idx_t sb = 0; sb < ns; sb++) { /* :225-243 */
    idx_t b0 = sb * per, b1 = imin(b0 + per, nb);
    idx_t t = 0;
    for (idx_t b = b0; b < b1; b++) {
      idx_t i0 = b * ORC_BLOCK_BITS, i1 = imin(i0 + ORC_BLOCK_BITS, n);
      s->blk[b] = (uint16_t)t;
      t += sbv_count(s, i0, i1);
    }
    s->super[sb] = t;
  }
  idx_t acc = 0; /* exclusive scan, :246 and include/conc.h:199-206 */
  for (idx_t sb = 0; sb < ns; sb++) {
    idx_t t = s->super[sb];
    s->super[sb] = acc;
    acc += t;
  }
  s->n_zeros = sbv_rank0(s, n); /* :249 */
}

static void sbv_fini(orc_sbv *s) {
  free(s->super);
  free(s->blk);
}

/* ------------------------------------------------------------------ */
/* wavelet matrix: include/wavelet_matrix.h                            */
/* ------------------------------------------------------------------ */
typedef struct {
  idx_t n;
  int width; /* int_log2(alpha_max), include/util.h:244-255 */
  orc_sbv *lv;
  uint8_t *planes; /* level l occupies bits [l*n,(l+1)*n), :185,:189 */
} orc_wm;

int orc_int_log2(int x) { /* include/util.h:244-255 */
  int l = 0;
  idx_t y = 1;
  while (y <= x) {
    l++;
    y += y;
  }
  return l;
}

/* include/wavelet_matrix.h:72-147: stable split of t by bit (width-1-l).
   The element written at d keeps the l high bits already in place at d and
   takes the low bits of the moved element (:143), so after all levels t
   holds, at level l, bit (width-1-l) of the sequence ordered by the bit
   reversal of its l higher bits. */
static void wm_partition(const alpha_t *t, alpha_t *u, idx_t n, int width, int l) {
  int shift = width - l - 1;
  alpha_t mask0 = (alpha_t)((1 << shift) - 1);
  alpha_t mask1 = (alpha_t)((1 << width) - (1 << shift));
  idx_t z = 0;
  for (idx_t i = 0; i < n; i++) z += !((t[i] >> shift) & 1);
  idx_t off[2] = {0, z};
  for (idx_t i = 0; i < n; i++) {
    int b = (t[i] >> shift) & 1;
    idx_t d = off[b]++;
    u[d] = (alpha_t)((t[d] & mask1) | (t[i] & mask0));
  }
}

/* include/wavelet_matrix.h:198-241 (init) and :175-191 (transpose_bits). */
static void wm_init(orc_wm *w, const alpha_t *s, idx_t n, int alpha_max) {
  w->n = n;
  w->width = orc_int_log2(alpha_max);
  int W = w->width;
  alpha_t *t0 = (alpha_t *)xmalloc((size_t)n + 1);
  alpha_t *t1 = (alpha_t *)xmalloc((size_t)n + 1);
  alpha_t *t[2] = {t0, t1};
  memcpy(t[0], s, (size_t)n);
  for (int l = 0; l < W; l++) wm_partition(t[l % 2], t[(l + 1) % 2], n, W, l);
  const alpha_t *fin = t[W % 2];
  size_t nbytes = (size_t)(((idx_t)W * n + 7) / 8 + 16);
  w->planes = (uint8_t *)xmalloc(nbytes);
  memset(w->planes, 0, nbytes);
  for (idx_t i = 0; i < n; i++) { /* :180-187 */
    alpha_t c = fin[i];
    for (int l = 0; l < W; l++) {
      if ((c >> (W - l - 1)) & 1) {
        idx_t bit = (idx_t)l * n + i;
        w->planes[bit >> 3] |= (uint8_t)(1u << (bit & 7));
      }
    }
  }
  w->lv = (orc_sbv *)xmalloc(sizeof(orc_sbv) * (size_t)W);
  for (int l = 0; l < W; l++) sbv_init(&w->lv[l], w->planes, (idx_t)l * n, (idx_t)(l + 1) * n);
  free(t0);
  free(t1);
}

static void wm_fini(orc_wm *w) {
  if (!w->lv) return;
  for (int l = 0; l < w->width; l++) sbv_fini(&w->lv[l]);
  free(w->lv);
  free(w->planes);
  w->lv = 0;
  w->planes = 0;
}

/* include/wavelet_matrix.h:260-290: occurrences of c in s[0:i). */
static idx_t wm_rank(const orc_wm *w, alpha_t c, idx_t i) {
  idx_t p = 0, q = i;
  for (int l = 0; l < w->width; l++) {
    const orc_sbv *b = &w->lv[l];
    if ((c >> (w->width - l - 1)) & 1) {
      p = b->n_zeros + sbv_rank1(b, p);
      q = b->n_zeros + sbv_rank1(b, q);
    } else {
      p = sbv_rank0(b, p);
      q = sbv_rank0(b, q);
    }
  }
  return q - p;
}

/* ------------------------------------------------------------------ */
/* sampled suffix array: include/sampled_suffix_array.h                */
/* ------------------------------------------------------------------ */
typedef struct {
  idx_t rank, pos;
} orc_sample; /* :36-39 */

typedef struct {
  idx_t n, sz;
  orc_sample *a;
  orc_sample *begin, *end; /* valid after sort_by_pos */
} orc_ssa;

static void ssa_init(orc_ssa *s, idx_t req) { /* :83-97 */
  idx_t log_sz = 0, sz = 1;
  while (sz < req) { /* include/util.h:257-265 */
    sz *= 2;
    log_sz++;
  }
  s->n = 0;
  s->sz = (idx_t)1 << (log_sz + 1);
  s->a = (orc_sample *)xmalloc(sizeof(orc_sample) * (size_t)s->sz);
  for (idx_t i = 0; i < s->sz; i++) s->a[i].rank = s->a[i].pos = -1;
  s->begin = 0;
  s->end = s->a + s->sz;
}

static void ssa_fini(orc_ssa *s) {
  free(s->a);
  s->a = 0;
}

static void ssa_put(orc_ssa *s, idx_t rank, idx_t pos) { /* :183-212 */
  idx_t h = rank & (s->sz - 1);
  for (idx_t k = 0; k < s->sz; k++) {
    orc_sample *e = &s->a[(h + k) & (s->sz - 1)];
    if (e->rank == rank) return;
    if (e->rank == -1) {
      e->rank = rank;
      e->pos = pos;
      s->n++;
      return;
    }
  }
  die("ssa_put: table full");
}

static idx_t ssa_get(const orc_ssa *s, idx_t rank) { /* :148-171 */
  idx_t h = rank & (s->sz - 1);
  for (idx_t k = 0; k < s->sz; k++) {
    const orc_sample *e = &s->a[(h + k) & (s->sz - 1)];
    if (e->rank == rank) return e->pos;
    if (e->rank == -1) return -1;
  }
  return -1;
}

static int cmp_sample_pos(const void *x, const void *y) { /* :41-45 */
  idx_t a = ((const orc_sample *)x)->pos, b = ((const orc_sample *)y)->pos;
  return (a > b) - (a < b);
}

static void ssa_sort_by_pos(orc_ssa *s) { /* :104-122 */
  qsort(s->a, (size_t)s->sz, sizeof(orc_sample), cmp_sample_pos);
  orc_sample *p = s->a;
  while (p < s->a + s->sz && p->pos == -1) p++;
  s->begin = p;
  s->end = s->a + s->sz;
}

/* ------------------------------------------------------------------ */
/* gap array: include/gap_array.h (counters + sparse prefix sums)      */
/* ------------------------------------------------------------------ */
typedef struct {
  idx_t a, b, interval;
  idx_t *cnt; /* cnt[i-a] for i in [a,b]; the reference keeps 8-bit counters
                 plus an overflow hash for values >= 255 (:24-25, :206-278) */
  idx_t ps_a, ps_b;
  idx_t *ps; /* exclusive sums of whole intervals, :281-303 */
} orc_gap;

static void gap_init(orc_gap *g, idx_t a, idx_t b, idx_t interval) { /* :32-60 */
  g->a = a;
  g->b = b;
  g->interval = interval;
  g->cnt = (idx_t *)xmalloc(sizeof(idx_t) * (size_t)(b - a + 1));
  memset(g->cnt, 0, sizeof(idx_t) * (size_t)(b - a + 1));
  g->ps_a = a / interval;
  g->ps_b = b / interval;
  g->ps = (idx_t *)xmalloc(sizeof(idx_t) * (size_t)(g->ps_b - g->ps_a + 1));
}

static void gap_fini(orc_gap *g) {
  free(g->cnt);
  free(g->ps);
}

static void gap_inc(orc_gap *g, idx_t i) { g->cnt[i - g->a]++; }        /* :256-260 */
static idx_t gap_get(const orc_gap *g, idx_t i) { return g->cnt[i - g->a]; } /* :262-278 */

static void gap_set_prefix_sum(orc_gap *g) { /* :281-303 */
  for (idx_t k = g->ps_a; k <= g->ps_b; k++) {
    idx_t s = 0;
    for (idx_t i = imax(g->a, k * g->interval); i < imin(g->b, (k + 1) * g->interval); i++)
      s += gap_get(g, i);
    g->ps[k - g->ps_a] = s;
  }
  idx_t acc = 0;
  for (idx_t k = g->ps_a; k <= g->ps_b; k++) {
    idx_t t = g->ps[k - g->ps_a];
    g->ps[k - g->ps_a] = acc;
    acc += t;
  }
}

/* gap[a] + ... + gap[rank], rank included: :307-332 */
static idx_t gap_sum(const orc_gap *g, idx_t rank) {
  idx_t j = rank / g->interval;
  idx_t s = g->ps[j - g->ps_a];
  idx_t i0 = rank - rank % g->interval;
  if (i0 < g->a) i0 = g->a;
  for (idx_t i = i0; i <= rank; i++) s += gap_get(g, i);
  return s;
}

/* ------------------------------------------------------------------ */
/* partial BWT of T[a:b): include/bwt.h:122-539                        */
/* ------------------------------------------------------------------ */
typedef struct orc_block {
  const alpha_t *T;
  idx_t n;
  alpha_t *L;
  idx_t a, b;
  alpha_t e_char; /* T[b-1], :140 */
  orc_ssa ssa;
  idx_t s_rank, e_rank; /* rank of T[a:], of T[b-1:], :144-145 */
  idx_t *C;             /* :147 */
  orc_wm wm;
  int alpha_max;
} orc_block;

static void blk_init(orc_block *t, const alpha_t *T, idx_t n, alpha_t *L, idx_t a, idx_t b,
                     int alpha_max) { /* :155-167 */
  memset(t, 0, sizeof(*t));
  t->T = T;
  t->n = n;
  t->L = L;
  t->a = a;
  t->b = b;
  t->e_char = T[b - 1];
  t->s_rank = t->e_rank = -1;
  t->alpha_max = alpha_max;
}

/* include/bwt.h:200-226 */
static void blk_count_alphabets(orc_block *t) {
  int A = t->alpha_max;
  t->C = (idx_t *)xmalloc(sizeof(idx_t) * (size_t)(A + 1));
  for (int c = 0; c <= A; c++) t->C[c] = 0;
  for (idx_t i = t->a; i < t->b; i++) t->C[t->L[i]]++;
  t->C[t->L[t->s_rank]]--; /* T[a-1] is in L but not in T[a:b) */
  t->C[t->e_char]++;       /* T[b-1] is in T[a:b) but not in L */
  idx_t acc = 0;
  for (int c = 0; c <= A; c++) {
    idx_t s = t->C[c];
    t->C[c] = acc;
    acc += s;
  }
}

/* include/bwt.h:231-250: T[a-1] is hidden as 0 while the matrix is built. */
static void blk_build_wm(orc_block *t) {
  alpha_t save = t->L[t->s_rank];
  t->L[t->s_rank] = 0;
  wm_init(&t->wm, t->L + t->a, t->b - t->a, t->alpha_max);
  t->L[t->s_rank] = save;
}

void orc_block_init_extra(orc_block *t) { /* :169-172 */
  blk_count_alphabets(t);
  blk_build_wm(t);
}

static void blk_fini_extra(orc_block *t) {
  wm_fini(&t->wm);
  free(t->C);
  t->C = 0;
}

static idx_t blk_LF_map_mod(const orc_block *t, alpha_t c, idx_t i) { /* :253-259 */
  idx_t k = t->C[c] + wm_rank(&t->wm, c, i - t->a) + t->a;
  k += ((k >= t->e_rank) & (c == t->e_char));
  return k;
}

static idx_t blk_LF_map_mod2(const orc_block *t, alpha_t c, idx_t i, idx_t j) { /* :262-273 */
  idx_t k = t->C[c] + wm_rank(&t->wm, c, i - t->a) + t->a;
  if (c == t->e_char && k >= t->e_rank) {
    if (orc_suffix_less(t->T, t->n, t->b - 1, j)) k++;
  }
  return k;
}

void orc_block_ibwt(const orc_block *t, alpha_t *I) { /* :276-285 */
  idx_t i = t->e_rank;
  alpha_t c = t->e_char;
  for (idx_t j = t->b; j > t->a; j--) {
    I[j - 1] = c;
    c = t->L[i];
    i = blk_LF_map_mod(t, c, i);
  }
}

/* include/bwt.h:294-361.  sa is indexed from 0 (rank r-a). */
static void blk_sample_sa(orc_block *t, const idx_t *sa, idx_t ns) {
  idx_t a = t->a, b = t->b;
  ssa_init(&t->ssa, ns);
  for (idx_t r = a; r < b; r++) {
    idx_t i = sa[r - a];
    if (i == a) t->s_rank = r;
    if (i + 1 == b) t->e_rank = r;
    if (b - a == 1) {
      ssa_put(&t->ssa, r, i);
    } else {
      idx_t x = (idx_t)floor((i - 1 - a) * (ns - 1) / (double)(b - a - 1)); /* :339 */
      idx_t y = (idx_t)floor((i - a) * (ns - 1) / (double)(b - a - 1));     /* :340 */
      if (x != y) ssa_put(&t->ssa, r, i);
    }
  }
  t->ssa.n = imin(ns, b - a); /* :358 */
}

/* include/bwt.h:456-481 */
static idx_t blk_sa(const orc_block *t, idx_t r) {
  idx_t x = 0, j = r;
  for (;;) {
    idx_t p = ssa_get(&t->ssa, j);
    if (p != -1) return p + x;
    j = blk_LF_map_mod(t, t->L[j], j);
    x++;
  }
}

/* include/bwt.h:489-537: i such that T[sa[i-1]:] < T[x:] < T[sa[i]:]. */
static idx_t blk_bin_search(const orc_block *t, idx_t x) {
  if (orc_suffix_less(t->T, t->n, x, blk_sa(t, t->a))) return t->a;
  if (orc_suffix_less(t->T, t->n, blk_sa(t, t->b - 1), x)) return t->b;
  idx_t p = t->a, q = t->b - 1;
  while (q - p > 1) {
    idx_t r = (p + q) / 2;
    if (orc_suffix_less(t->T, t->n, x, blk_sa(t, r)))
      q = r;
    else
      p = r;
  }
  return q;
}

/* include/bwt.h:589-609 (sa_to_bwt) and :615-635 (bwt_leaf).
   If sa_out is not null it receives the leaf suffix array (b-a entries). */
static void blk_leaf(orc_block *t, const alpha_t *T, idx_t n, idx_t a, idx_t b, alpha_t *L,
                     const orc_opt *o, idx_t *sa_out) {
  idx_t *sa = (idx_t *)xmalloc(sizeof(idx_t) * (size_t)(b - a));
  orc_sa_range(T, n, a, b, sa);
  for (idx_t r = a; r < b; r++) { /* :591-603 */
    idx_t i = sa[r - a];
    L[r] = (i == 0) ? T[n - 1] : T[i - 1];
  }
  blk_init(t, T, n, L, a, b, o->alpha_max);
  blk_sample_sa(t, sa, imin(b - a, o->ssa_n_samples)); /* :607 */
  if (sa_out) memcpy(sa_out, sa, sizeof(idx_t) * (size_t)(b - a));
  free(sa);
}

/* include/bwt.h:730-787.  g_out, if not null, receives for every j in
   [r.a,r.b) the gap index g_j (entry j-r.a). */
static void blk_build_gap(orc_block *l, orc_block *r, orc_gap *gap, const orc_opt *o,
                          idx_t *g_out) {
  ssa_sort_by_pos(&r->ssa); /* :742 */
  idx_t seg = o->build_gap_segment_sz;
  for (idx_t begin = r->a; begin < r->b; begin += seg) { /* :745 */
    idx_t g = -1;
    idx_t end = imin(begin + seg, r->b);
    /* include/sampled_suffix_array.h:126-134: last sample with pos <= end-1 */
    orc_sample *it = r->ssa.end - 1;
    while (it >= r->ssa.begin && it->pos > end - 1) it--;
    for (idx_t j = end - 1; j >= begin; j--) {
      if (j == end - 1)
        g = blk_bin_search(l, j); /* :753 */
      else
        g = blk_LF_map_mod2(l, r->T[j], g, j); /* :760 */
      gap_inc(gap, g);                         /* :777 */
      if (g_out) g_out[j - r->a] = g;
      if (it >= r->ssa.begin && it->pos == j) { /* :778-781 */
        it->rank += g;
        it--;
      }
    }
  }
  gap_set_prefix_sum(gap); /* :784 */
}

/* include/bwt.h:366-453 */
static void blk_resample_sa(orc_block *m, orc_ssa *l, orc_ssa *r, idx_t c, idx_t ns,
                            const orc_gap *gap) {
  ssa_init(&m->ssa, ns);
  idx_t nn = l->n + r->n;
  idx_t n_all = 0, n_taken = 0;
  if (!l->begin) ssa_sort_by_pos(l); /* include/sampled_suffix_array.h:136-141 */
  for (orc_sample *e = l->begin; e < l->end; e++) { /* :395-427 */
    if (n_taken * (nn - 1) <= n_all * (ns - 1)) {
      idx_t g = gap_sum(gap, e->rank);
      ssa_put(&m->ssa, e->rank + g, e->pos);
      if (e->pos == m->a) m->s_rank = e->rank + g;
      n_taken++;
    }
    n_all++;
  }
  if (!r->begin) ssa_sort_by_pos(r);
  for (orc_sample *e = r->begin; e < r->end; e++) { /* :431-447 */
    if (n_taken * (nn - 1) <= n_all * (ns - 1)) {
      ssa_put(&m->ssa, e->rank - c, e->pos);
      if (e->pos == m->b - 1) m->e_rank = e->rank - c;
      n_taken++;
    }
    n_all++;
  }
}

/* include/bwt.h:791-847.  W is the scratch of the merge (n bytes, indexed
   like L).  gap_out (l.b-l.a+1 entries) and g_out (r.b-r.a entries) are
   optional taps for the parity tests. */
static void blk_merge(orc_block *m, orc_block *l, orc_block *r, alpha_t *W, const orc_opt *o,
                      idx_t *gap_out, idx_t *g_out) {
  orc_block_init_extra(l); /* :795 */
  orc_gap gap;
  gap_init(&gap, l->a, l->b, o->gap_sum_gran); /* :797 */
  blk_build_gap(l, r, &gap, o, g_out);         /* :799 */
  if (gap_out)
    for (idx_t i = l->a; i <= l->b; i++) gap_out[i - l->a] = gap_get(&gap, i);
  for (idx_t i = l->a; i < l->b + 1; i++) { /* :815-829 */
    idx_t e = r->a + gap_sum(&gap, i);
    for (idx_t j = e - gap_get(&gap, i); j < e; j++) W[i - r->a + j] = r->L[j];
    if (i < l->b) W[i - r->a + e] = l->L[i];
  }
  memcpy(&l->L[l->a], &W[l->a], (size_t)(r->b - l->a)); /* :833-834 */
  blk_init(m, l->T, l->n, l->L, l->a, r->b, o->alpha_max);
  blk_resample_sa(m, &l->ssa, &r->ssa, l->b, imin(m->b - m->a, o->ssa_n_samples), &gap); /* :840 */
  gap_fini(&gap);
  blk_fini_extra(l);
  ssa_fini(&l->ssa);
  ssa_fini(&r->ssa);
}

/* include/bwt.h:854-865 */
static void blk_rec(orc_block *t, const alpha_t *T, idx_t n, idx_t a, idx_t b, alpha_t *L,
                    alpha_t *W, const orc_opt *o) {
  if (b - a <= o->bwt_rec_threshold) {
    blk_leaf(t, T, n, a, b, L, o, 0);
  } else {
    idx_t c = (a + b) / 2;
    orc_block l, r;
    blk_rec(&l, T, n, a, c, L, W, o);
    blk_rec(&r, T, n, c, b, L, W, o);
    blk_merge(t, &l, &r, W, o, 0, 0);
  }
}

/* ------------------------------------------------------------------ */
/* exported handle API (ctypes)                                        */
/* ------------------------------------------------------------------ */

orc_block *orc_leaf(const alpha_t *T, idx_t n, idx_t a, idx_t b, alpha_t *L, const orc_opt *o,
                    idx_t *sa_out) {
  orc_block *t = (orc_block *)xmalloc(sizeof(orc_block));
  blk_leaf(t, T, n, a, b, L, o, sa_out);
  return t;
}

/* merges and frees l and r; returns the merged block */
orc_block *orc_merge(orc_block *l, orc_block *r, const orc_opt *o, idx_t *gap_out, idx_t *g_out) {
  orc_block *m = (orc_block *)xmalloc(sizeof(orc_block));
  alpha_t *W = (alpha_t *)xmalloc((size_t)l->n);
  blk_merge(m, l, r, W, o, gap_out, g_out);
  free(W);
  free(l);
  free(r);
  return m;
}

/* include/bwt.h:867-883 */
orc_block *orc_pmbwt(const alpha_t *T, idx_t n, alpha_t *L, const orc_opt *o) {
  orc_block *t = (orc_block *)xmalloc(sizeof(orc_block));
  alpha_t *W = (alpha_t *)xmalloc((size_t)n);
  blk_rec(t, T, n, 0, n, L, W, o);
  free(W);
  return t;
}

void orc_block_free(orc_block *t) {
  blk_fini_extra(t);
  ssa_fini(&t->ssa);
  free(t);
}

idx_t orc_block_a(const orc_block *t) { return t->a; }
idx_t orc_block_b(const orc_block *t) { return t->b; }
idx_t orc_block_s_rank(const orc_block *t) { return t->s_rank; }
idx_t orc_block_e_rank(const orc_block *t) { return t->e_rank; }
int orc_block_e_char(const orc_block *t) { return t->e_char; }
idx_t orc_block_n_samples(const orc_block *t) { return t->ssa.n; }

/* copies the samples, ordered by pos; returns how many were written */
idx_t orc_block_samples(const orc_block *t, idx_t *rank_out, idx_t *pos_out, idx_t cap) {
  idx_t k = 0;
  orc_sample *tmp = (orc_sample *)xmalloc(sizeof(orc_sample) * (size_t)t->ssa.sz);
  for (idx_t i = 0; i < t->ssa.sz; i++)
    if (t->ssa.a[i].pos != -1) tmp[k++] = t->ssa.a[i];
  qsort(tmp, (size_t)k, sizeof(orc_sample), cmp_sample_pos);
  idx_t w = imin(k, cap);
  for (idx_t i = 0; i < w; i++) {
    rank_out[i] = tmp[i].rank;
    pos_out[i] = tmp[i].pos;
  }
  free(tmp);
  return k;
}

/* C[] of include/bwt.h:200-226; requires orc_block_init_extra */
void orc_block_C(const orc_block *t, idx_t *C_out) {
  for (int c = 0; c <= t->alpha_max; c++) C_out[c] = t->C[c];
}

/* wm.rank(c, i) for i relative to a; requires orc_block_init_extra */
idx_t orc_block_rank(const orc_block *t, int c, idx_t i) { return wm_rank(&t->wm, (alpha_t)c, i); }

idx_t orc_block_LF(const orc_block *t, int c, idx_t i) { return blk_LF_map_mod(t, (alpha_t)c, i); }

idx_t orc_block_sa(const orc_block *t, idx_t r) { return blk_sa(t, r); }

idx_t orc_block_bin_search(const orc_block *t, idx_t x) { return blk_bin_search(t, x); }

/* The BWT by its definition: L[r] = T[SA[r]-1] over the whole text, from one
   comparison sort.  Used to cross-check the pipeline above on small inputs. */
void orc_bwt_direct(const alpha_t *T, idx_t n, alpha_t *L, idx_t *sa_out) {
  idx_t *sa = (idx_t *)xmalloc(sizeof(idx_t) * (size_t)n);
  orc_sa_range(T, n, 0, n, sa);
  for (idx_t r = 0; r < n; r++) L[r] = sa[r] ? T[sa[r] - 1] : T[n - 1];
  if (sa_out) memcpy(sa_out, sa, sizeof(idx_t) * (size_t)n);
  free(sa);
}

uint64_t orc_fnv1a64(const uint8_t *p, idx_t n) {
  uint64_t h = 0xcbf29ce484222325ull;
  for (idx_t i = 0; i < n; i++) {
    h ^= p[i];
    h *= 0x100000001b3ull;
  }
  return h;
}

/* Synthetic text: x = splitmix64 stream from `seed`, T[i] = 1 + x % sigma,
   T[n-1] = 0. */
void orc_gen_text(alpha_t *T, idx_t n, uint64_t seed, int sigma) {
  uint64_t s = seed;
  for (idx_t i = 0; i + 1 < n; i++) {
    s += 0x9E3779B97F4A7C15ull;
    uint64_t z = s;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    T[i] = (alpha_t)(1 + z % (uint64_t)sigma);
  }
  T[n - 1] = 0;
}
