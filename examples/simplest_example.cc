// Smallest use of the bwt:: header API on MI355X: BWT of one short string.
//   g++ -std=c++11 -I../include simplest_example.cc -L../bwt_amd/lib -lbwt_hip
#include "bwt.h"

int main() {
  const char * text = "abracadabra";
  const bwt::alpha_t * T = (const bwt::alpha_t *)text;
  bwt::idx_t n = (bwt::idx_t)strlen(text) + 1; /* the terminating 0 is part of the input */
  bwt::alpha_t * L = new bwt::alpha_t[n];

  bwt::bwt_opt opt;
  opt.set_defaults(T, n);
  bwt::mallocator mem(n, opt);

  bwt::bwt t = bwt::pmbwt(T, n, L, 0, mem, opt);

  printf("input: %s\nbwt:   ", text);
  for (bwt::idx_t i = 0; i < n; i++) putchar(L[i] ? L[i] : '$');
  printf("\nrank of the whole text: %ld\n", (long)t.s_rank);
  t.fini(mem, opt);
  delete[] L;
  return 0;
}
