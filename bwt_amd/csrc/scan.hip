// Concrete scans over plain u32 arrays.
#include "scan_tmpl.h"

namespace bwtamd {

void scan_exclusive_u32(Ctx &c, const u32 *in, u32 *out, size_t n, u32 *total) {
  if (n == 0) {
    if (total) HIP_CHECK(hipMemsetAsync(total, 0, sizeof(u32), c.stream));
    return;
  }
  scan_generic<u32, SumU32>(c, n, ArrayIn<u32>{in}, ArrayOutExcl<u32>{out, total, n});
}

void scan_inclusive_max_u32(Ctx &c, const u32 *in, u32 *out, size_t n) {
  scan_generic<u32, MaxU32>(c, n, ArrayIn<u32>{in}, ArrayOutIncl<u32>{out});
}

}  // namespace bwtamd
