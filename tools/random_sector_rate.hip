// How many isolated random 64-byte reads per second the memory side of the GPU serves, independent of the LF-chain
// kernel: every lane reads whole 64-byte blocks of a table at pseudo-random block indices.  Two forms:
//   independent: the next index comes from the lane's own generator (as many reads in flight as the hardware takes)
//   dependent:   the next index depends on the data just read (one read in flight per lane, like an LF step)
// usage: random_sector_rate [table_MB=167] [threads=500000] [steps=1024]
//        random_sector_rate stream      (a plain copy, for the streaming rate of the same chip)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x)                                                                              \
  do {                                                                                        \
    hipError_t e_ = (x);                                                                      \
    if (e_ != hipSuccess) {                                                                   \
      fprintf(stderr, "%s:%d: %s\n", __FILE__, __LINE__, hipGetErrorString(e_));              \
      exit(1);                                                                                \
    }                                                                                         \
  } while (0)

// the same reads issued so that four neighbouring lanes fetch the four 16-byte pieces of one block: a wave
// instruction then touches 16 blocks instead of 64 (four instructions still cover the wave's 64 blocks)
__global__ void __launch_bounds__(256) k_reads_quad(const uint4 *__restrict__ table, uint32_t blocks, uint32_t threads,
                                                    int steps, uint32_t *__restrict__ sink) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= threads) return;
  const uint32_t lane = threadIdx.x & 63;
  uint64_t x = 0x9E3779B97F4A7C15ull * (t + 1);
  uint32_t acc = 0;
  for (int s = 0; s < steps; s++) {
    x = x * 6364136223846793005ull + 1442695040888963407ull;
    const uint32_t b = (uint32_t)(((x >> 32) * (uint64_t)blocks) >> 32);
    uint32_t sum = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const uint32_t owner = 16 * k + (lane >> 2);  // the lane whose block this quad fetches now
      const uint32_t bb = __shfl(b, owner, 64);
      const uint4 v = table[(size_t)bb * 4 + (lane & 3)];
      sum ^= v.x ^ v.y;
    }
    acc += sum;
  }
  if (acc == 0xFFFFFFFFu) sink[0] = acc;
}

// whole 128-byte lines instead of 64-byte sectors, each read waiting for the one before
__global__ void __launch_bounds__(256) k_reads_line(const uint4 *__restrict__ table, uint32_t blocks, uint32_t threads,
                                                    int steps, uint32_t *__restrict__ sink) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= threads) return;
  uint64_t x = 0x9E3779B97F4A7C15ull * (t + 1);
  uint32_t acc = 0;
  for (int s = 0; s < steps; s++) {
    x = x * 6364136223846793005ull + 1442695040888963407ull;
    const uint32_t b = (uint32_t)(((x >> 32) * (uint64_t)(blocks / 2)) >> 32);
    const uint4 *p = table + (size_t)b * 8;
    uint32_t sum = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) sum ^= p[k].x;
    acc += sum;
    x += sum;  // zeros: the sequence stays, the next read waits for this one
  }
  if (acc == 0xFFFFFFFFu) sink[0] = acc;
}

template <bool DEPENDENT>
__global__ void __launch_bounds__(256) k_reads(const uint4 *__restrict__ table, uint32_t blocks, uint32_t threads,
                                               int steps, uint32_t *__restrict__ sink) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= threads) return;
  uint64_t x = 0x9E3779B97F4A7C15ull * (t + 1);
  uint32_t acc = 0;
  for (int s = 0; s < steps; s++) {
    x = x * 6364136223846793005ull + 1442695040888963407ull;
    uint32_t b = (uint32_t)(((x >> 32) * (uint64_t)blocks) >> 32);
    const uint4 *p = table + (size_t)b * 4;
    const uint4 v0 = p[0], v1 = p[1], v2 = p[2], v3 = p[3];
    const uint32_t sum = v0.x ^ v1.y ^ v2.z ^ v3.w;
    acc += sum;
    if (DEPENDENT) x += sum;  // the table holds zeros: the sequence is the same, the dependence is real
  }
  if (acc == 0xFFFFFFFFu) sink[0] = acc;
}

// plain streaming copy, 16 bytes per lane: what the chip gives a kernel that only streams
__global__ void __launch_bounds__(256) k_copy(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n16) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}

__global__ void __launch_bounds__(256) k_read(const uint4 *__restrict__ src, size_t n16, uint32_t *__restrict__ sink) {
  uint32_t acc = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
    const uint4 v = src[i];
    acc ^= v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0xFFFFFFFFu) sink[0] = acc;
}

__global__ void __launch_bounds__(256) k_write(uint4 *__restrict__ dst, size_t n16) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256)
    dst[i] = make_uint4((uint32_t)i, 1, 2, 3);
}

static void stream_rate() {
  const size_t bytes = (size_t)2 << 30;
  uint4 *a, *b;
  CHECK(hipMalloc(&a, bytes));
  CHECK(hipMalloc(&b, bytes));
  CHECK(hipMemset(a, 1, bytes));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int rep = 0; rep < 5; rep++) {
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_copy, dim3(256 * 32), dim3(256), 0, 0, a, b, bytes / 16);
    CHECK(hipGetLastError());
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  printf("streaming copy of 2 GiB (read + write 4 GiB): %.3f ms = %.2f TB/s\n", best, 2.0 * bytes / (best * 1e-3) / 1e12);
  best = 1e30f;
  for (int rep = 0; rep < 5; rep++) {
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_read, dim3(256 * 32), dim3(256), 0, 0, a, bytes / 16, (uint32_t *)b);
    CHECK(hipGetLastError());
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  printf("streaming read of 2 GiB: %.3f ms = %.2f TB/s\n", best, 1.0 * bytes / (best * 1e-3) / 1e12);
  best = 1e30f;
  for (int rep = 0; rep < 5; rep++) {
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_write, dim3(256 * 32), dim3(256), 0, 0, b, bytes / 16);
    CHECK(hipGetLastError());
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  printf("streaming write of 2 GiB: %.3f ms = %.2f TB/s\n", best, 1.0 * bytes / (best * 1e-3) / 1e12);
  CHECK(hipFree(a));
  CHECK(hipFree(b));
}

int main(int argc, char **argv) {
  if (argc > 1 && argv[1][0] == 's') {
    stream_rate();
    return 0;
  }
  const size_t mb = argc > 1 ? strtoull(argv[1], 0, 10) : 167;
  const uint32_t threads = argc > 2 ? (uint32_t)strtoul(argv[2], 0, 10) : 500000;
  const int steps = argc > 3 ? atoi(argv[3]) : 1024;
  const size_t bytes = mb << 20;
  const uint32_t blocks = (uint32_t)(bytes / 64);
  uint4 *table;
  uint32_t *sink;
  CHECK(hipMalloc(&table, bytes));
  CHECK(hipMalloc(&sink, 4));
  CHECK(hipMemset(table, 0, bytes));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  for (int dep = 0; dep < 4; dep++) {
    float best = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
      CHECK(hipEventRecord(e0));
      if (dep == 3) hipLaunchKernelGGL(k_reads_line, dim3((threads + 255) / 256), dim3(256), 0, 0, table, blocks, threads, steps, sink);
      else if (dep == 2) hipLaunchKernelGGL(k_reads_quad, dim3((threads + 255) / 256), dim3(256), 0, 0, table, blocks, threads, steps, sink);
      else if (dep) hipLaunchKernelGGL(k_reads<true>, dim3((threads + 255) / 256), dim3(256), 0, 0, table, blocks, threads, steps, sink);
      else hipLaunchKernelGGL(k_reads<false>, dim3((threads + 255) / 256), dim3(256), 0, 0, table, blocks, threads, steps, sink);
      CHECK(hipGetLastError());
      CHECK(hipEventRecord(e1));
      CHECK(hipEventSynchronize(e1));
      float ms;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best) best = ms;
    }
    const double reads = (double)threads * steps;
    printf("table %zu MB, %u threads x %d reads, %s: %.3f ms, %.2fe10 reads/s = %.2f TB/s\n", mb, threads, steps,
           dep == 3 ? "dependent, 128-byte lines (bytes: x2)" : dep == 2 ? "independent, four lanes per block" : dep ? "dependent" : "independent", best, reads / (best * 1e-3) / 1e10, reads * 64 / (best * 1e-3) / 1e12);
  }
  return 0;
}
