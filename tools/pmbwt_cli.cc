// pmbwt -- command line front end of the MI355X BWT builder.
//
// Takes the flags of the reference's src/pmbwt.cc (:148-177) with the same
// meaning, generates or reads a text, builds its BWT with bwt::pmbwt and, with
// --check 1 (the default), inverts it and compares (src/pmbwt.cc:64-81): on the device, cut at the samples;
// --check 2 takes the reference's serial walk on the host (init_extra + ibwt).
//
// Differences from the reference's program, on purpose:
//   --generator splitmix (default) draws symbols from a splitmix64 stream;
//       nrand48 reproduces the reference's generator, whose low bits make
//       4-letter texts periodic with period 2^19 (every suffix comparison of
//       its CPU sort then runs for ~n symbols);
//   --input FILE reads the text from a file (a 0 byte is appended);
//   --output FILE writes L;
//   -t / --ssa_n_samples given explicitly are applied AFTER set_defaults, which
//       otherwise recomputes both from --mem_budget (include/opts.h:278-290 of
//       the reference overwrites them);
//   --algo_alpha 1 passes --alpha_min/--alpha_max to the algorithm as well
//       (the reference uses them for the generated input only).
//   --n_workers is accepted and ignored: the parallel work runs on the GPU.
#include <getopt.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "bwt.h"

struct cli_opt {
  bwt::idx_t n_chars = 100000;
  unsigned long seed = 918729723;
  bwt::alpha_t alpha_min = 'a', alpha_max = 'z';
  int stat_level = 0, mstat_level = 0, n_workers = 1, check = 1, repeat = 1, help = 0, algo_alpha = 0;
  std::string generator = "splitmix", input, output;
  bwt::idx_t bwt_rec_threshold = 0, ssa_n_samples = 0;
};

struct flag {
  const char * name;
  int short_name;   // 0 if none
  char kind;        // 'i' idx_t field of bwt_opt, 'd' double, 'n' int field, 'c' cli field
  size_t offset;    // into bwt_opt for i/d/n
};

#define OPT_I(f) { #f, 0, 'i', offsetof(bwt::bwt_opt, f) }
static const flag opt_flags[] = {
  { "mem_budget", 'm', 'd', offsetof(bwt::bwt_opt, mem_budget) },
  OPT_I(sort_rec_threshold), OPT_I(merge_rec_threshold), OPT_I(build_gap_segment_sz), OPT_I(gap_sum_gran),
  OPT_I(wavelet_matrix_sum_interval), OPT_I(count_alpha_gran), OPT_I(merge_gran), OPT_I(memcpy_gran),
  OPT_I(gap_init_gran), OPT_I(ssa_init_gran), OPT_I(wavelet_matrix_add_zero_count_gran),
  { "n_approx_workers", 0, 'n', offsetof(bwt::bwt_opt, n_approx_workers) },
  { "assert_level", 0, 'n', offsetof(bwt::bwt_opt, assert_level) },
  { "device", 0, 'n', offsetof(bwt::bwt_opt, device) },
  { "n_gpus", 0, 'n', offsetof(bwt::bwt_opt, n_gpus) },
  { "fit_leaves_to_device", 0, 'n', offsetof(bwt::bwt_opt, fit_leaves_to_device) },
};
static const char * cli_flags[] = { "n_workers", "alpha_min", "alpha_max", "stat_level", "mstat_level", "check",
                                    "repeat", "ssa_n_samples", "bwt_rec_threshold", "seed", "generator", "input",
                                    "output", "algo_alpha", "help" };

static void usage() {
  for (const flag & f : opt_flags) fprintf(stderr, "--%s%s X\n", f.name, f.short_name ? (std::string(",-") + (char)f.short_name).c_str() : "");
  for (const char * f : cli_flags) fprintf(stderr, "--%s X\n", f);
  fprintf(stderr, "short forms: -w n_workers, -m mem_budget, -t bwt_rec_threshold, -r repeat, -h help\n");
  fprintf(stderr, "positional: number of characters (default 100000)\n");
}

static bool parse(int argc, char ** argv, bwt::bwt_opt & opt, cli_opt & c) {
  std::vector<option> longs;
  const int n_opt = sizeof(opt_flags) / sizeof(opt_flags[0]);
  const int n_cli = sizeof(cli_flags) / sizeof(cli_flags[0]);
  for (int i = 0; i < n_opt; i++) longs.push_back({ opt_flags[i].name, required_argument, 0, 1000 + i });
  for (int i = 0; i < n_cli; i++) longs.push_back({ cli_flags[i], required_argument, 0, 2000 + i });
  longs.push_back({ 0, 0, 0, 0 });
  for (;;) {
    int k = getopt_long(argc, argv, "hw:m:t:r:", longs.data(), 0);
    if (k == -1) break;
    std::string name;
    if (k == 'h') name = "help";
    else if (k == 'w') name = "n_workers";
    else if (k == 'm') name = "mem_budget";
    else if (k == 't') name = "bwt_rec_threshold";
    else if (k == 'r') name = "repeat";
    else if (k >= 2000 && k < 2000 + n_cli) name = cli_flags[k - 2000];
    else if (k >= 1000 && k < 1000 + n_opt) name = opt_flags[k - 1000].name;
    else return false;
    bool done = false;
    for (const flag & f : opt_flags) {
      if (name != f.name) continue;
      char * base = (char *)&opt + f.offset;
      if (f.kind == 'i') *(bwt::idx_t *)base = atol(optarg);
      else if (f.kind == 'd') *(double *)base = atof(optarg);
      else *(int *)base = atoi(optarg);
      done = true;
    }
    if (done) continue;
    if (name == "help") { c.help = 1; return true; }
    else if (name == "n_workers") c.n_workers = atoi(optarg);
    else if (name == "alpha_min") c.alpha_min = (bwt::alpha_t)optarg[0];
    else if (name == "alpha_max") c.alpha_max = (bwt::alpha_t)optarg[0];
    else if (name == "stat_level") c.stat_level = atoi(optarg);
    else if (name == "mstat_level") c.mstat_level = atoi(optarg);
    else if (name == "check") c.check = atoi(optarg);
    else if (name == "repeat") c.repeat = atoi(optarg);
    else if (name == "ssa_n_samples") c.ssa_n_samples = atol(optarg);
    else if (name == "bwt_rec_threshold") c.bwt_rec_threshold = atol(optarg);
    else if (name == "seed") c.seed = strtoul(optarg, 0, 10);
    else if (name == "generator") c.generator = optarg;
    else if (name == "input") c.input = optarg;
    else if (name == "output") c.output = optarg;
    else if (name == "algo_alpha") c.algo_alpha = atoi(optarg);
    else return false;
  }
  if (optind < argc) c.n_chars = atol(argv[optind]);
  return true;
}

static bwt::alpha_t * make_text(cli_opt & c, bwt::idx_t & n) {
  if (!c.input.empty()) {
    FILE * f = fopen(c.input.c_str(), "rb");
    if (!f) { perror(c.input.c_str()); exit(1); }
    fseek(f, 0, SEEK_END);
    long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    bwt::alpha_t * T = new bwt::alpha_t[sz + 1];
    if (fread(T, 1, sz, f) != (size_t)sz) { fprintf(stderr, "short read of %s\n", c.input.c_str()); exit(1); }
    fclose(f);
    for (long i = 0; i < sz; i++)
      if (T[i] == 0) { fprintf(stderr, "%s holds a 0 byte at offset %ld\n", c.input.c_str(), i); exit(1); }
    T[sz] = 0;
    n = sz + 1;
    return T;
  }
  n = c.n_chars;
  bwt::alpha_t * T = new bwt::alpha_t[n];
  unsigned span = c.alpha_max - c.alpha_min + 1;
  if (c.generator == "nrand48") { /* the reference's generator, src/pmbwt.cc:34-48 */
    unsigned short rg[3] = { (unsigned short)(c.seed >> 32), (unsigned short)(c.seed >> 16), (unsigned short)c.seed };
    for (bwt::idx_t i = 0; i + 1 < n; i++) T[i] = c.alpha_min + (nrand48(rg) % span);
  } else {
    uint64_t s = c.seed;
    for (bwt::idx_t i = 0; i + 1 < n; i++) {
      s += 0x9E3779B97F4A7C15ull;
      uint64_t z = s;
      z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
      z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
      z ^= z >> 31;
      T[i] = c.alpha_min + (bwt::alpha_t)(z % span);
    }
  }
  T[n - 1] = 0;
  return T;
}

int main(int argc, char ** argv) {
  bwt::bwt_opt opt;
  cli_opt c;
  if (!parse(argc, argv, opt, c)) { usage(); return 1; }
  if (c.help) { usage(); return 0; }
  bwt::idx_t n = 0;
  bwt::alpha_t * T = make_text(c, n);
  bwt::alpha_t * L = new bwt::alpha_t[n];
  if (c.algo_alpha) { opt.alpha_min = c.alpha_min; opt.alpha_max = c.alpha_max; }
  opt.set_defaults(T, n);
  if (c.bwt_rec_threshold) opt.bwt_rec_threshold = c.bwt_rec_threshold;
  if (c.ssa_n_samples) opt.ssa_n_samples = c.ssa_n_samples;

  bwt::mallocator mem(n, opt);
  mem.level = c.mstat_level;
  int ok = 1;
  for (int rep = 0; rep < c.repeat && ok; rep++) {
    printf("==== repeat %d ====\n", rep);
    printf("building bwt of %ld characters\n", (long)n);
    printf(" n_workers = %d\n repeat = %d\n ssa_n_samples = %ld\n bwt_rec_threshold = %ld\n", c.n_workers, c.repeat,
           (long)opt.ssa_n_samples, (long)opt.bwt_rec_threshold);
    printf(" build_gap_segment_sz = %ld\n gap_sum_gran = %ld\n wavelet_matrix_sum_interval = %ld\n n_approx_workers = %d\n",
           (long)opt.build_gap_segment_sz, (long)opt.gap_sum_gran, (long)opt.wavelet_matrix_sum_interval, opt.n_approx_workers);
    printf(" input_alpha_min = %d\n input_alpha_max = %d\n algo_alpha_min = %d\n algo_alpha_max = %d\n", c.alpha_min,
           c.alpha_max, opt.alpha_min, opt.alpha_max);
    printf(" stat_level = %d\n mstat_level = %d\n assert_level = %d\n seed = %ld\n", c.stat_level, c.mstat_level,
           opt.assert_level, (long)c.seed);
    bwt::bwt t = bwt::pmbwt(T, n, L, c.stat_level, mem, opt);
    if (c.check == 1 && t.dev) {
      // decoded on the device from the block that is still there: one LF walk per sample instead of one serial walk
      printf("checking result ... ");
      fflush(stdout);
      ok = (t.round_trip_mismatches() == 0);
      printf(ok ? "OK\n" : "NG\n");
    } else if (c.check) {
      bwt::alpha_t * I = new bwt::alpha_t[n];
      memset(I, 0xA5, n);
      printf("checking result ... ");
      fflush(stdout);
      t.init_extra(0, mem, opt);
      t.ibwt(I);
      ok = (memcmp(T, I, n) == 0);
      printf(ok ? "OK\n" : "NG\n");
      delete[] I;
    } else {
      printf("result not checked\n");
    }
    t.fini(mem, opt);
  }
  if (ok && !c.output.empty()) {
    FILE * f = fopen(c.output.c_str(), "wb");
    if (!f || fwrite(L, 1, n, f) != (size_t)n) { perror(c.output.c_str()); ok = 0; }
    if (f) fclose(f);
  }
  mem.print();
  delete[] T;
  delete[] L;
  return ok ? 0 : 1;
}
