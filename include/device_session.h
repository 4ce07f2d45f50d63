/* device_session.h -- the process-wide link between the bwt:: header API and
   libbwt_hip.so: one device context, and the text it currently holds.

   The reference needs nothing like this (it has no device).  The header API
   passes `const alpha_t * T, idx_t n` to every call.  pmbwt, bwt_leaf and bwt_rec
   upload T every time (n bytes: little next to the build), because a caller may
   have rewritten the buffer in place between two calls; bwt_merge, which can
   only follow them, reuses the device copy while (pointer, length, sampled
   signature) stay the same.  A caller that builds many leaves of one unchanged
   text sets `device_session::get().text_is_stable = true` to skip the repeated
   uploads.  Switching the device (opt.device) destroys the old context together
   with every device block still alive in it: struct bwt objects from before the
   switch keep their host data, and their fini() stays valid.  Failures of the C-ABI are reported the way the
   reference reports failures: message on stderr, abort() (include/util.h:26-36
   of the reference). */
#pragma once

#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "bwt_hip.h"
#include "typedefs.h"

namespace bwt {

  inline void hip_ok_(int rc, const char * what, const char * file, int line) {
    if (rc != 0) {
      fprintf(stderr, "error:%s:%d: %s failed (%d): %s\n", file, line, what, rc, bwt_hip_last_error());
      abort();
    }
  }
#define bwt_hip_ok(call) ::bwt::hip_ok_((call), #call, __FILE__, __LINE__)

  struct device_session {
    bwt_hip_ctx * ctx;
    int device;
    const alpha_t * T;
    idx_t n;
    uint64_t sig;
    bool text_is_stable; /* the caller promises not to modify T between calls */

    device_session() : ctx(0), device(-1), T(0), n(0), sig(0), text_is_stable(false) {}
    ~device_session() { if (ctx) bwt_hip_destroy(ctx); }

    static device_session & get() { static device_session s; return s; }

    bwt_hip_ctx * context(int dev) {
      if (ctx && device != dev) { bwt_hip_destroy(ctx); ctx = 0; T = 0; }
      if (!ctx) { bwt_hip_ok(bwt_hip_create(dev, &ctx)); device = dev; }
      return ctx;
    }

    /* FNV-1a over the ends and a stride of the text: cheap, and enough to notice a reused buffer */
    static uint64_t signature(const alpha_t * T, idx_t n) {
      uint64_t h = 0xcbf29ce484222325ull;
      idx_t step = n / 4096 + 1;
      for (idx_t i = 0; i < n; i += step) { h ^= T[i]; h *= 0x100000001b3ull; }
      for (idx_t i = (n > 4096 ? n - 4096 : 0); i < n; i++) { h ^= T[i]; h *= 0x100000001b3ull; }
      return h ^ (uint64_t)n;
    }

    /* the context has just been given T[0:n) by a call that uploads on its own (bwt_hip_pmbwt_host) */
    void note_text(const alpha_t * T_, idx_t n_) { T = T_; n = n_; sig = signature(T_, n_); }

    /* makes the context hold T[0:n); `force` uploads even if it looks unchanged */
    bwt_hip_ctx * with_text(const alpha_t * T_, idx_t n_, int dev, bool force = false) {
      bwt_hip_ctx * c = context(dev);
      uint64_t s = signature(T_, n_);
      if (force || T != T_ || n != n_ || sig != s) {
        bwt_hip_ok(bwt_hip_set_text(c, T_, n_));
        T = T_; n = n_; sig = s;
      }
      return c;
    }
  };

}
