// lds_rmw_probe.hip -- what an LDS accumulator costs on gfx950 (round 5, EXPERIMENTS.md D.1): the throughput per CU of
//   A  ds_add_f64 without return (an LDS fp64 atomic add: the race-free scatter of a face-major loop without a colouring)
//   B  ds_read_b64 + v_add_f64 + ds_write_b64 (the same update under a colouring: plain read-modify-write)
//   C  ds_read_b64 alone (what the point-centric loops do: operands gathered, sums in registers)
//   D  ds_read_b128 alone
//   E  as A, but no two lanes of a wave touch the same row (what a colouring guarantees)
// at the occupancy of the fused pass: 256-thread workgroups, 4 per CU (36 KiB of LDS each), 4 waves per SIMD.  Addresses
// are pseudo-random 8-byte slots of a [64][21]-double accumulator (a tile's gradient rows), different per lane.
//   hipcc -O3 --offload-arch=gfx950 tools/lds_rmw_probe.hip -o /tmp/lds_rmw_probe && /tmp/lds_rmw_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void probe(double *out, int iters) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  double *acc = reinterpret_cast<double *>(smem);  // 64 x 21 doubles (+ padding up to the 36 KiB image)
  const int tid = threadIdx.x;
  for (int i = tid; i < 36 * 1024 / 8; i += 256) acc[i] = 0.0;
  __syncthreads();
  unsigned x = 2463534242u ^ (tid * 2654435761u) ^ (blockIdx.x * 40503u);
  double s0 = 0.0, s1 = 0.0;
  for (int it = 0; it < iters; it++) {
    x ^= x << 13; x ^= x >> 17; x ^= x << 5;
    // one accumulator row per lane and step, 21 (A/B/C/E) or 20 (D) doubles of it; E: no two lanes of a wave share a row
    const int row = MODE == 4 ? (tid + it * 17) & 63 : x % 64u;
    double *r = acc + row * 21;
    if constexpr (MODE == 0 || MODE == 4) {
#pragma unroll
      for (int c = 0; c < 21; c++) __hip_atomic_fetch_add(r + c, 1.0 + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else if constexpr (MODE == 1) {
      double v[21];
#pragma unroll
      for (int c = 0; c < 21; c++) v[c] = r[c];
#pragma unroll
      for (int c = 0; c < 21; c++) r[c] = v[c] + (1.0 + c);
    } else if constexpr (MODE == 2) {
#pragma unroll
      for (int c = 0; c < 21; c += 3) { s0 += r[c]; s1 += r[c + 1]; s0 += r[c + 2]; }
    } else {
      const double2 *q = reinterpret_cast<const double2 *>(acc + (row * 21 & ~1));
#pragma unroll
      for (int c = 0; c < 10; c++) { const double2 v = q[c]; s0 += v.x; s1 += v.y; }
    }
  }
  __syncthreads();
  if (MODE >= 2 || tid == 0) out[blockIdx.x * 256 + tid] = s0 + s1 + acc[tid];
}

template <int MODE> void run(const char *what, double bytes_per_lane_step, int cus) {
  const int blocks = cus * 4, iters = 2000;
  double *out;
  CK(hipMalloc(&out, sizeof(double) * blocks * 256));
  CK(hipFuncSetAttribute(reinterpret_cast<const void *>(probe<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 36 * 1024));
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int rep = 0; rep < 3; rep++) {
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(256), 36 * 1024, 0, out, iters);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
  }
  float ms = 0;
  CK(hipEventElapsedTime(&ms, a, b));
  const double lane_steps = (double)blocks * 256 * iters;
  const double wave_instr_per_cu = lane_steps / 64 / cus * (MODE == 3 ? 10 : 21) * (MODE == 1 ? 2 : 1);
  printf("%-58s %8.3f ms  %7.1f B/clk/CU (at 2.4 GHz)  %6.2f clk per wave-instruction per CU\n", what, ms,
         lane_steps * bytes_per_lane_step / (ms * 1e-3) / cus / 2.4e9, ms * 1e-3 * 2.4e9 / wave_instr_per_cu);
  CK(hipFree(out));
}

int main() {
  int cus = 0;
  CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
  printf("%d CUs; 4 workgroups of 256 threads per CU, one accumulator row [21 doubles] per lane and step\n", cus);
  run<0>("A ds_add_f64, no return (21 per lane and step)", 21 * 8.0, cus);
  run<1>("B ds_read_b64 + v_add_f64 + ds_write_b64 (21 + 21)", 2 * 21 * 8.0, cus);
  run<2>("C ds_read_b64 (21 per lane and step)", 21 * 8.0, cus);
  run<3>("D ds_read_b128 (10 per lane and step)", 10 * 16.0, cus);
  run<4>("E ds_add_f64, no return, a row of its own per lane (a colour)", 21 * 8.0, cus);
  return 0;
}
