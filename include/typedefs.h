/* typedefs.h -- the two scalar types of the bwt:: API (same names and meaning
   as the reference's include/typedefs.h:14-31). */
#pragma once
#include <stddef.h>

namespace bwt {
  typedef unsigned char alpha_t; /* one text symbol; 0 ends the text */
  typedef ptrdiff_t idx_t;       /* signed index into text / suffix array; -1 is "none" */
  inline idx_t min(idx_t x, idx_t y) { return y < x ? y : x; }
  inline idx_t max(idx_t x, idx_t y) { return y > x ? y : x; }
}
