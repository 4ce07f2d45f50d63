// Host <-> device transfer of a text and of its BWT for bwt_hip_pmbwt_host (reference: the caller's T and L of
// bwt::pmbwt, include/bwt.h:867-883, live in host memory).
//
// A text whose symbols span at most four consecutive values (DNA) crosses the link as two bits per symbol: host
// threads pack a chunk into a page-locked staging buffer, the chunk goes up and a kernel expands it into the
// device text; L comes back the same way (packed by a kernel, expanded by host threads).  That is a quarter of
// the bytes on the link, and the caller's own buffers need not be page-locked to reach the link's rate.  The
// upload runs on a thread of its own so that the build can start on the chunks that have arrived.  That thread also
// checks every symbol against the range a sample of the text showed, which spares the build a pass over the text
// before it starts; a text with a symbol outside that range is taken again, unpacked.
#include <string.h>

#include <atomic>
#include <string>
#include <thread>
#include <vector>

#include "device_utils.h"
#include "internal.h"
#include "pack2.h"

namespace bwtamd {

constexpr i64 XF_CHUNK = (i64)256 << 20;  // most symbols per chunk (64 MB packed); Ctx::xfer_chunk is the size in use
constexpr int XF_SLOTS = 3;

// 16 symbols per thread: T[i] = min_sym + two bits of the packed text
__global__ void k_unpack2(const u8 *__restrict__ packed, u8 *__restrict__ T, u64 count, u32 min_sym) {
  const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  const u64 i0 = t * 16;
  if (i0 >= count) return;
  if (i0 + 16 <= count) {
    const u32 w = *reinterpret_cast<const u32 *>(packed + t * 4);
    u32 o[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const u32 b = (w >> (8 * q)) & 255u;
      // no byte overflows: the packer saw symbol - min_sym <= 3 for every symbol, and min_sym + 3 <= 255 by the caller's test
      o[q] = ((b & 3u) | (((b >> 2) & 3u) << 8) | (((b >> 4) & 3u) << 16) | ((b >> 6) << 24)) + min_sym * 0x01010101u;
    }
    *reinterpret_cast<uint4 *>(T + i0) = make_uint4(o[0], o[1], o[2], o[3]);
  } else {
    for (u64 i = i0; i < count; i++) T[i] = (u8)(min_sym + ((packed[i >> 2] >> (2 * (i & 3))) & 3u));
  }
}

// the reverse: 16 symbols of L per thread into four packed bytes (a symbol outside the range packs as garbage:
// the one foreign symbol of L, the terminator, is put back by the host)
__global__ void k_pack2(const u8 *__restrict__ L, u8 *__restrict__ packed, u64 count, u32 min_sym) {
  const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  const u64 i0 = t * 16;
  if (i0 >= count) return;
  u32 w = 0;
  if (i0 + 16 <= count) {
    const uint4 v = *reinterpret_cast<const uint4 *>(L + i0);
    const u32 in[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int q = 0; q < 4; q++) {
      // (symbol - min_sym) mod 4 per byte; the 4 added to every byte keeps a borrow from crossing into its neighbour
      const u32 x = ((in[q] & 0x03030303u) | 0x04040404u) - (min_sym & 3u) * 0x01010101u;
      const u32 b = (x & 3u) | (((x >> 8) & 3u) << 2) | (((x >> 16) & 3u) << 4) | (((x >> 24) & 3u) << 6);
      w |= b << (8 * q);
    }
    *reinterpret_cast<u32 *>(packed + t * 4) = w;
  } else {
    for (u64 i = i0; i < count; i += 4) {
      u32 b = 0;
      for (u64 j = i; j < i + 4 && j < count; j++) b |= ((u32)(L[j] - min_sym) & 3u) << (2 * (j - i));
      packed[i >> 2] = (u8)b;
    }
  }
}

static int host_threads(i64 work) {
  unsigned hw = std::thread::hardware_concurrency();
  int nt = hw > 16 ? 16 : (hw < 1 ? 1 : (int)hw);
  if (work < ((i64)1 << 22)) nt = 1;
  return nt;
}

template <class F>
static void on_threads(int nt, F body) {
  std::vector<std::thread> th;
  for (int t = 1; t < nt; t++) th.push_back(std::thread(body, t));
  body(0);
  for (auto &x : th) x.join();
}

// packs src[0:count) into dst on nt threads; false if a symbol lies outside [min_sym, min_sym + span]
static bool pack_chunk(const u8 *src, i64 count, u8 *dst, u32 min_sym, u32 span, int nt) {
  std::atomic<int> bad(0);
  on_threads(nt, [&](int t) {
    const i64 groups = (count + 63) / 64;  // threads split at multiples of 64 symbols: whole bytes, whole steps
    const i64 lo = groups * t / nt * 64, hi = groups * (t + 1) / nt * 64 < count ? groups * (t + 1) / nt * 64 : count;
    if (lo < hi && !pack2::pack(src + lo, hi - lo, dst + lo / 4, min_sym, span)) bad.store(1);
  });
  return bad.load() == 0;
}

static void unpack_chunk(const u8 *src, i64 count, u8 *dst, u32 min_sym, int nt) {
  on_threads(nt, [&](int t) {
    const i64 groups = (count + 63) / 64;
    const i64 lo = groups * t / nt * 64, hi = groups * (t + 1) / nt * 64 < count ? groups * (t + 1) / nt * 64 : count;
    if (lo < hi) pack2::unpack(src + lo / 4, hi - lo, dst + lo, min_sym);
  });
}

static void ensure_staging(Ctx &c) {
  if (c.xfer_stage[0]) return;
  for (int s = 0; s < XF_SLOTS; s++) {
    HIP_CHECK(hipHostMalloc((void **)&c.xfer_stage[s], (size_t)(c.xfer_chunk / 4), hipHostMallocDefault));
    HIP_CHECK(hipEventCreateWithFlags(&c.xfer_ev[s], hipEventDisableTiming));
  }
}

// smallest and largest non-zero symbol of a sample of the text (ends and strided positions)
void probe_symbol_range(const u8 *T, i64 n, int *lo, int *hi) {
  int mn = 256, mx = 0;
  auto see = [&](u8 v) {
    if (v) {
      mn = v < mn ? v : mn;
      mx = v > mx ? v : mx;
    }
  };
  const i64 edge = n < 8192 ? n : 4096;
  for (i64 i = 0; i < edge; i++) see(T[i]);
  for (i64 i = n - edge; i < n; i++) see(T[i]);
  const i64 step = n / 65536 + 1;
  for (i64 i = 0; i < n; i += step) see(T[i]);
  *lo = mn == 256 ? 0 : mn;
  *hi = mx;
}

// The uploader thread of a packed upload: c.chunks holds the chunks' ends and events (none recorded yet).  A chunk
// with a symbol outside [min_sym, min_sym + span] is not sent: c.xfer_mismatch is raised instead and the upload ends
// there.  Every chunk's event is recorded whatever happens, so that nobody waits forever; an error is reported
// through *err.
void upload_text_packed(Ctx &c, const u8 *T_host, i64 n, u32 min_sym, u32 span, u8 *dpacked, std::string *err) {
  bool stopped = false;
  try {
    HIP_CHECK(hipSetDevice(c.device));
  } catch (const std::exception &e) {
    *err = e.what();
    stopped = true;
  }
  const i64 body = n - 1;  // everything but the terminator
  const int nt = host_threads(body);
  i64 off = 0;
  for (size_t k = 0; k < c.chunks.size(); k++) {
    const i64 end = (i64)c.chunks[k].end;
    const i64 upto = end < body ? end : body;  // symbols of this chunk that are packed
    const i64 len = upto - off;
    if (!stopped && !c.xfer_abort.load()) {
      try {
        const int slot = (int)(k % XF_SLOTS);
        if (k >= (size_t)XF_SLOTS) HIP_CHECK(hipEventSynchronize(c.xfer_ev[slot]));  // the slot's last copy has left it
        if (len > 0) {
          if (pack_chunk(T_host + off, len, c.xfer_stage[slot], min_sym, span, nt)) {
            HIP_CHECK(hipMemcpyAsync(dpacked + off / 4, c.xfer_stage[slot], (size_t)((len + 3) / 4), hipMemcpyHostToDevice, c.copy_stream));
            HIP_CHECK(hipEventRecord(c.xfer_ev[slot], c.copy_stream));
            hipLaunchKernelGGL(k_unpack2, dim3(ceil_div((u64)(len + 15) / 16, 256)), dim3(256), 0, c.copy_stream,
                               (const u8 *)(dpacked + off / 4), c.T_owned + off, (u64)len, min_sym);
            HIP_CHECK(hipGetLastError());
          } else {
            c.xfer_mismatch.store(true);
            stopped = true;
          }
        }
        if (!stopped && end >= n) HIP_CHECK(hipMemsetAsync(c.T_owned + n - 1, 0, 1, c.copy_stream));  // the terminator
      } catch (const std::exception &e) {
        *err = e.what();
        stopped = true;
      }
    }
    (void)hipEventRecord(c.chunks[k].ev, c.copy_stream);
    {
      std::lock_guard<std::mutex> lk(c.chunk_mu);
      c.chunks_recorded = k + 1;
    }
    c.chunk_cv.notify_all();
    off = upto;
  }
}

// L[0:n) from the device to host memory as two bits per symbol; the symbol at `foreign` (the terminator) is put back
void download_packed(Ctx &c, const u8 *L_dev, i64 n, u32 min_sym, i64 foreign, u8 *L_host) {
  prepare_packed_upload(c);
  DBuf<u8> packed(c, (size_t)((n + 3) / 4 + 16));
  hipLaunchKernelGGL(k_pack2, dim3(ceil_div((u64)(n + 15) / 16, 256)), dim3(256), 0, c.stream, L_dev, packed.get(), (u64)n, min_sym);
  HIP_CHECK(hipGetLastError());
  const int nt = host_threads(n);
  const i64 CH = c.xfer_chunk;
  const i64 chunks = (n + CH - 1) / CH;
  auto issue = [&](i64 k) {
    const i64 off = k * CH, len = n - off < CH ? n - off : CH;
    const int slot = (int)(k % XF_SLOTS);
    HIP_CHECK(hipMemcpyAsync(c.xfer_stage[slot], packed.get() + off / 4, (size_t)((len + 3) / 4), hipMemcpyDeviceToHost, c.stream));
    HIP_CHECK(hipEventRecord(c.xfer_ev[slot], c.stream));
  };
  auto finish = [&](i64 k) {
    const i64 off = k * CH, len = n - off < CH ? n - off : CH;
    const int slot = (int)(k % XF_SLOTS);
    HIP_CHECK(hipEventSynchronize(c.xfer_ev[slot]));
    unpack_chunk(c.xfer_stage[slot], len, L_host + off, min_sym, nt);
  };
  // two copies in flight while the host expands the one before them
  for (i64 k = 0; k < chunks + 2; k++) {
    if (k < chunks) issue(k);
    if (k >= 2) finish(k - 2);
  }
  if (foreign >= 0 && foreign < n) L_host[foreign] = 0;
  c.sync();
}

// A block's samples to host memory: 32-bit words over the link (half the bytes of the caller's 64-bit arrays),
// widened by host threads while the next chunk travels.  rank_out[i] = rank[i] + a, pos_out[i] = pos[i].
void download_samples32(Ctx &c, const u32 *rank, const u32 *pos, i64 k, i64 a, i64 *rank_out, i64 *pos_out) {
  prepare_packed_upload(c);
  const i64 per = (c.xfer_chunk / 4) / 8;  // samples per staging buffer: 4 bytes of rank + 4 of position each
  const i64 chunks = (k + per - 1) / per;
  const int nt = host_threads(k);
  auto issue = [&](i64 j) {
    const i64 off = j * per, len = k - off < per ? k - off : per;
    u32 *st = reinterpret_cast<u32 *>(c.xfer_stage[j % XF_SLOTS]);
    HIP_CHECK(hipMemcpyAsync(st, rank + off, (size_t)len * 4, hipMemcpyDeviceToHost, c.stream));
    HIP_CHECK(hipMemcpyAsync(st + per, pos + off, (size_t)len * 4, hipMemcpyDeviceToHost, c.stream));
    HIP_CHECK(hipEventRecord(c.xfer_ev[j % XF_SLOTS], c.stream));
  };
  auto finish = [&](i64 j) {
    const i64 off = j * per, len = k - off < per ? k - off : per;
    const u32 *st = reinterpret_cast<const u32 *>(c.xfer_stage[j % XF_SLOTS]);
    HIP_CHECK(hipEventSynchronize(c.xfer_ev[j % XF_SLOTS]));
    on_threads(nt, [&](int t) {
      const i64 lo = len * t / nt, hi = len * (t + 1) / nt;
      for (i64 i = lo; i < hi; i++) {
        rank_out[off + i] = (i64)st[i] + a;
        pos_out[off + i] = (i64)st[per + i];
      }
    });
  };
  for (i64 j = 0; j < chunks + 2; j++) {  // two chunks in flight while the host widens the one before them
    if (j < chunks) issue(j);
    if (j >= 2) finish(j - 2);
  }
  c.sync();
}

void prepare_packed_upload(Ctx &c) {
  if (c.xfer_chunk > XF_CHUNK) c.xfer_chunk = XF_CHUNK;
  if (c.xfer_chunk < 4096) c.xfer_chunk = 4096;
  c.xfer_chunk &= ~(i64)63;
  ensure_staging(c);  // sized by xfer_chunk, which does not change once the buffers exist (read when the context is made)
}

}  // namespace bwtamd
