// stream_probe.hip -- what can this chip stream with the gradient kernel's read/write mix?
// (measurement helper, not part of the product)   hipcc --offload-arch=gfx950 -O3 stream_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_read(const uint4* __restrict__ a, size_t n, uint4* sink) {
  uint4 acc = {0, 0, 0, 0};
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    uint4 v = a[i]; acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
  }
  if (acc.x == 0x12345678u && acc.y == 77u) sink[0] = acc;
}
__global__ void k_write(uint4* __restrict__ o, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    o[i] = make_uint4((unsigned)i, 1, 2, 3);
}
// read 2 streams of nr each, write one stream of nw  (nr*2 : nw  ~  720 : 352)
__global__ void k_mix(const uint4* __restrict__ a, const uint4* __restrict__ b, uint4* __restrict__ o, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    uint4 v = a[i], w = b[i];
    o[i] = make_uint4(v.x ^ w.x, v.y ^ w.y, v.z ^ w.z, v.w ^ w.w);
  }
}
// tile-wise: one block handles contiguous chunks of `chunk` uint4 (like a tile): read 2 chunks, write 1
__global__ void k_mix_tiles(const uint4* __restrict__ a, const uint4* __restrict__ b, uint4* __restrict__ o,
                            size_t n, int chunk) {
  size_t ntiles = n / chunk;
  for (size_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
    for (int i = threadIdx.x; i < chunk; i += blockDim.x) {
      size_t j = t * chunk + i;
      uint4 v = a[j], w = b[j];
      o[j] = make_uint4(v.x ^ w.x, v.y ^ w.y, v.z ^ w.z, v.w ^ w.w);
    }
  }
}
int main(int argc, char **argv) {
  // default: 360 MB per read stream, 352 MB written (the 128^3 mix); `stream_probe 45` = the
  // Infinity-Cache-resident 64^3 mix
  const size_t mb = argc > 1 ? (size_t)atoi(argv[1]) : 360;
  const size_t nA = (mb << 20) / 16, nO = nA;
  uint4 *a, *b, *o, *sink;
  CK(hipMalloc(&a, nA * 16)); CK(hipMalloc(&b, nA * 16)); CK(hipMalloc(&o, nA * 16)); CK(hipMalloc(&sink, 64));
  CK(hipMemset(a, 1, nA * 16)); CK(hipMemset(b, 2, nA * 16)); CK(hipMemset(o, 0, nA * 16));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto timeit = [&](const char* name, double bytes, auto launch) {
    for (int w = 0; w < 3; w++) launch();
    hipEventRecord(e0); const int it = 20; for (int i = 0; i < it; i++) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= it;
    printf("%-28s %8.1f us  %7.0f GB/s\n", name, ms * 1e3, bytes / ms / 1e6);
  };
  for (int blocks : {2048, 8192}) {
    printf("blocks %d x 256, %zu MB per stream\n", blocks, mb);
    timeit("read 2 streams", 2.0 * nA * 16, [&] { k_read<<<blocks, 256>>>(a, nA, sink); k_read<<<blocks, 256>>>(b, nA, sink); });
    timeit("write 1 stream", (double)nO * 16, [&] { k_write<<<blocks, 256>>>(o, nO); });
    timeit("mix 2 read : 1 write", 3.0 * nA * 16, [&] { k_mix<<<blocks, 256>>>(a, b, o, nA); });
  }
  for (int chunk : {1024, 2048, 4096})
    timeit(chunk == 1024 ? "mix tiles 16KB" : chunk == 2048 ? "mix tiles 32KB" : "mix tiles 64KB", 3.0 * nA * 16,
           [&] { k_mix_tiles<<<2048, 512>>>(a, b, o, nA, chunk); });
  return 0;
}
