// face_loop_probe.hip -- the two forms of a tile's GRADIENT face loop side by side, the loop alone, everything in LDS
// (round 5, EXPERIMENTS.md D.1): what the loop costs a CU per tile
//   P  point-centric, as the product kernels have it: 4 lanes per owned point (2 equations each), the point's incidences in
//      a list, normal + the other end's var row gathered from LDS, 6 sums per lane in REGISTERS, one store at the end
//   F  face-major with an intra-tile colouring, as the reference loops (src/gradients.c:65-133 -- loop the faces, update both
//      ends; colours src/rangelist.c:654-703): 4 lanes per face, normal and both ends' var rows read once, the face term
//      formed once, +t added to p0's row and -t to p1's row of an LDS accumulator [64][4][6] by plain read-modify-write
//      (ds_read_b128 x3, v_add_f64 x6, ds_write_b128 x3 per owned end and lane); faces of one colour share no owned end, one
//      barrier per colour (the four waves of a tile are not in lockstep)
//   A  as F without colours and barriers, the sums by ds_add_f64 (NOT acceptable in the product: the order of an fp64 sum
//      would differ from run to run -- here as the lower bound of any face-major form)
//   S4 / S2 / S1  point-centric with the point's INCIDENCES split over 4 / 2 / 1 lanes, every lane all equations, the lanes'
//      sums meeting in xor-shuffles: a quarter of P's address arithmetic per incidence
// and, for every form, the socket's power read from rocm-smi while it runs back to back: joules per tile (EXPERIMENTS D.8)
// on a synthetic tile with the bench mesh's counts: a 4 x 4 x 4 block of points with 14 neighbours each (6 along the axes, 8
// along the body diagonals): 896 incidences, 252 faces with both ends owned, 392 cut faces, 152 halo rows (bench mesh, per
// 64-point tile: 870 / 278 / 314 / ~140).  The three kernels compute the same sums (checked, 1e-12 relative).
//   hipcc -O3 --offload-arch=gfx950 tools/face_loop_probe.hip -o /tmp/face_loop_probe && /tmp/face_loop_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <map>
#include <string>
#include <thread>
#include <atomic>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

enum { NPT = 64, NNB = 14, NROWS = 216, NFACES = 644, MAXCOL = 32 };

// what a workgroup copies into LDS once: the rows and normals, and the lists of its form (the point-centric image has to
// fit 32 KiB, the product's five workgroups per CU)
struct geom {
  double var[NROWS][8];   // rows 0..63 owned, then halo
  double nrm[NFACES][3];  // in face-major order (by colour)
};
struct image_p {
  geom g;
  unsigned short pface[NPT * NNB];  // face | (sign < 0) << 15
  unsigned char prow[NPT * NNB];    // the other end's row
};
struct image_f {
  geom g;
  unsigned char f0[656], f1[656];  // the two ends' rows, faces sorted by colour
  int coff[MAXCOL + 4];            // first face of a colour
  int ncol, nfaces, pad[2];
};
static_assert(sizeof(image_p) % 16 == 0 && sizeof(image_p) <= 32 * 1024 && sizeof(image_f) % 16 == 0, "LDS images");

template <typename IMG> __device__ __forceinline__ void copy_in(unsigned char *smem, const IMG *img, int tid) {
  const uint4 *s = reinterpret_cast<const uint4 *>(img);
  uint4 *d = reinterpret_cast<uint4 *>(smem);
  for (int i = tid; i < (int)(sizeof(IMG) / 16); i += 256) d[i] = s[i];
  __syncthreads();
}

__global__ __launch_bounds__(256) void loop_points(const image_p *img, double *out, int iters) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, p = tid >> 2, sub = tid & 3;
  copy_in(smem, img, tid);
  const image_p *t = reinterpret_cast<const image_p *>(smem);
  double acc[6] = {0, 0, 0, 0, 0, 0};
  for (int it = 0; it < iters; it++) {
    asm volatile("" ::: "memory");
    const double2 v = *reinterpret_cast<const double2 *>(&t->g.var[p][2 * sub]);
    double a[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll 2
    for (int j = 0; j < NNB; j++) {
      const int e = t->pface[p * NNB + j], q = t->prow[p * NNB + j];
      const int f = e & 1023;
      const int sgn = (e >> 15) << 31;  // the sign rides on the sum (xor), the 0.5 is applied once at the end
      const double2 w = *reinterpret_cast<const double2 *>(&t->g.var[q][2 * sub]);
      const double n0 = t->g.nrm[f][0], n1 = t->g.nrm[f][1], n2 = t->g.nrm[f][2];
      const double b0 = v.x + w.x, b1 = v.y + w.y;
      const double a0 = __hiloint2double(__double2hiint(b0) ^ sgn, __double2loint(b0));
      const double a1 = __hiloint2double(__double2hiint(b1) ^ sgn, __double2loint(b1));
      a[0] += a0 * n0; a[1] += a0 * n1; a[2] += a0 * n2;
      a[3] += a1 * n0; a[4] += a1 * n1; a[5] += a1 * n2;
    }
#pragma unroll
    for (int c = 0; c < 6; c++) acc[c] += 0.5 * a[c];
  }
  if (blockIdx.x == 0)
    for (int c = 0; c < 6; c++) out[(p * 4 + sub) * 6 + c] = acc[c] / iters;
  else if (acc[0] == 1.2345e300) out[0] = acc[1];
}

// S<LPP>: the point's INCIDENCES split over LPP lanes, every lane all eight slots of a row (7 equations + the volume's
// slot, as the product's row): the word and the normal are read once per incidence instead of once per lane of the point,
// and the sums of the lanes meet in xor-shuffles at the end (24 values, log2(LPP) steps).  Lanes >= 64 * LPP idle.
template <int LPP>
__global__ __launch_bounds__(256) void loop_points_split(const image_p *img, double *out, int iters) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, p = tid / LPP, sub = tid % LPP;
  copy_in(smem, img, tid);
  const image_p *t = reinterpret_cast<const image_p *>(smem);
  const bool active = p < NPT;
  double keep[24];
#pragma unroll
  for (int c = 0; c < 24; c++) keep[c] = 0.0;
  for (int it = 0; it < iters; it++) {
    asm volatile("" ::: "memory");
    double a[8][3];
#pragma unroll
    for (int e = 0; e < 8; e++) a[e][0] = a[e][1] = a[e][2] = 0.0;
    if (active) {
      double v[8];
#pragma unroll
      for (int e = 0; e < 8; e += 2) {
        const double2 x = *reinterpret_cast<const double2 *>(&t->g.var[p][e]);
        v[e] = x.x; v[e + 1] = x.y;
      }
#pragma unroll 2
      for (int j = sub; j < NNB; j += LPP) {
        const int e = t->pface[p * NNB + j], q = t->prow[p * NNB + j];
        const int f = e & 1023, sgn = (e >> 15) << 31;
        const double n0 = t->g.nrm[f][0], n1 = t->g.nrm[f][1], n2 = t->g.nrm[f][2];
#pragma unroll
        for (int h = 0; h < 8; h += 2) {
          const double2 w = *reinterpret_cast<const double2 *>(&t->g.var[q][h]);
          const double b0 = v[h] + w.x, b1 = v[h + 1] + w.y;
          const double a0 = __hiloint2double(__double2hiint(b0) ^ sgn, __double2loint(b0));
          const double a1 = __hiloint2double(__double2hiint(b1) ^ sgn, __double2loint(b1));
          a[h][0] += a0 * n0; a[h][1] += a0 * n1; a[h][2] += a0 * n2;
          a[h + 1][0] += a1 * n0; a[h + 1][1] += a1 * n1; a[h + 1][2] += a1 * n2;
        }
      }
    }
#pragma unroll
    for (int m = 1; m < LPP; m <<= 1)
#pragma unroll
      for (int e = 0; e < 8; e++)
#pragma unroll
        for (int c = 0; c < 3; c++) a[e][c] += __shfl_xor(a[e][c], m, 64);
#pragma unroll
    for (int c = 0; c < 24; c++) keep[c] += 0.5 * a[c / 3][c % 3];
  }
  // the same layout as the other forms write: [point][4 lane groups of 2 equations][6]
  if (blockIdx.x == 0 && active && sub == 0)
    for (int c = 0; c < 24; c++) out[p * 24 + c] = keep[c] / iters;
  else if (keep[0] == 1.2345e300) out[0] = keep[1];
}

template <bool ATOMIC>
__global__ __launch_bounds__(256) void loop_faces(const image_f *img, double *out, int iters) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, g = tid >> 2, sub = tid & 3;
  copy_in(smem, img, tid);
  const image_f *t = reinterpret_cast<const image_f *>(smem);
  double *accum = reinterpret_cast<double *>(smem + sizeof(image_f));  // [64][4][6]
  const int ncol = t->ncol, nfaces = t->nfaces;
  double keep[6] = {0, 0, 0, 0, 0, 0};
  for (int it = 0; it < iters; it++) {
    double2 *z = reinterpret_cast<double2 *>(accum + tid * 6);
    z[0] = z[1] = z[2] = double2{0.0, 0.0};
    __syncthreads();
    auto face = [&](int f) {
      const int p0 = t->f0[f], p1 = t->f1[f];
      const double2 w0 = *reinterpret_cast<const double2 *>(&t->g.var[p0][2 * sub]);
      const double2 w1 = *reinterpret_cast<const double2 *>(&t->g.var[p1][2 * sub]);
      const double n0 = t->g.nrm[f][0], n1 = t->g.nrm[f][1], n2 = t->g.nrm[f][2];
      const double a0 = (w0.x + w1.x) * 0.5, a1 = (w0.y + w1.y) * 0.5;
      const double tm[6] = {a0 * n0, a0 * n1, a0 * n2, a1 * n0, a1 * n1, a1 * n2};
      if (p0 < NPT) {
        double *r = accum + (p0 * 4 + sub) * 6;
        if constexpr (ATOMIC) {
#pragma unroll
          for (int c = 0; c < 6; c++) __hip_atomic_fetch_add(r + c, tm[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else {
          double2 *r2 = reinterpret_cast<double2 *>(r);
          double2 x0 = r2[0], x1 = r2[1], x2 = r2[2];
          x0.x += tm[0]; x0.y += tm[1]; x1.x += tm[2]; x1.y += tm[3]; x2.x += tm[4]; x2.y += tm[5];
          r2[0] = x0; r2[1] = x1; r2[2] = x2;
        }
      }
      if (p1 < NPT) {
        double *r = accum + (p1 * 4 + sub) * 6;
        if constexpr (ATOMIC) {
#pragma unroll
          for (int c = 0; c < 6; c++) __hip_atomic_fetch_add(r + c, -tm[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else {
          double2 *r2 = reinterpret_cast<double2 *>(r);
          double2 x0 = r2[0], x1 = r2[1], x2 = r2[2];
          x0.x -= tm[0]; x0.y -= tm[1]; x1.x -= tm[2]; x1.y -= tm[3]; x2.x -= tm[4]; x2.y -= tm[5];
          r2[0] = x0; r2[1] = x1; r2[2] = x2;
        }
      }
    };
    if constexpr (ATOMIC) {
      for (int f = g; f < nfaces; f += 64) face(f);
      __syncthreads();
    } else {
      for (int c = 0; c < ncol; c++) {
        const int f = t->coff[c] + g;
        if (f < t->coff[c + 1]) face(f);
        __syncthreads();
      }
    }
#pragma unroll
    for (int c = 0; c < 6; c++) keep[c] += accum[tid * 6 + c];
    __syncthreads();
  }
  if (blockIdx.x == 0)
    for (int c = 0; c < 6; c++) out[tid * 6 + c] = keep[c] / iters;
  else if (keep[0] == 1.2345e300) out[0] = keep[1];
}

static void build(image_p &P, image_f &T) {
  static const int off[NNB][3] = {{1, 0, 0}, {-1, 0, 0}, {0, 1, 0}, {0, -1, 0}, {0, 0, 1}, {0, 0, -1}, {1, 1, 1}, {-1, -1, -1},
                                  {1, 1, -1}, {-1, -1, 1}, {1, -1, 1}, {-1, 1, -1}, {-1, 1, 1}, {1, -1, -1}};
  std::map<int, int> rowof;  // coordinate key -> row
  auto key = [](int x, int y, int z) { return (x + 1) + 6 * (y + 1) + 36 * (z + 1); };
  for (int z = 0; z < 4; z++) for (int y = 0; y < 4; y++) for (int x = 0; x < 4; x++) rowof[key(x, y, z)] = x + 4 * y + 16 * z;
  int nrows = NPT;
  struct face { int p0, p1; };
  std::vector<face> faces;
  std::map<std::pair<int, int>, int> faceof;
  std::vector<int> plist(NPT * NNB);
  srand(7);
  for (int z = 0; z < 4; z++) for (int y = 0; y < 4; y++) for (int x = 0; x < 4; x++) {
    const int p = x + 4 * y + 16 * z;
    for (int j = 0; j < NNB; j++) {
      const int k = key(x + off[j][0], y + off[j][1], z + off[j][2]);
      if (!rowof.count(k)) rowof[k] = nrows++;
      const int q = rowof[k];
      const std::pair<int, int> pr(std::min(p, q), std::max(p, q));
      if (!faceof.count(pr)) { faceof[pr] = (int)faces.size(); faces.push_back({pr.first, pr.second}); }
      const int f = faceof[pr];
      plist[p * NNB + j] = f | q << 10 | (faces[f].p0 == p ? 0 : 1) << 18;
    }
  }
  if (nrows != NROWS || (int)faces.size() != NFACES) { printf("unexpected tile: %d rows, %zu faces\n", nrows, faces.size()); exit(1); }
  // greedy colouring: no two faces of a colour share an owned end
  std::vector<int> colour(faces.size(), -1);
  std::vector<std::vector<char>> used(MAXCOL, std::vector<char>(NPT, 0));
  int ncol = 0;
  for (size_t f = 0; f < faces.size(); f++)
    for (int c = 0; c < MAXCOL; c++) {
      const bool u0 = faces[f].p0 < NPT && used[c][faces[f].p0], u1 = faces[f].p1 < NPT && used[c][faces[f].p1];
      if (u0 || u1) continue;
      int cnt = 0;
      for (size_t h = 0; h < f; h++) cnt += colour[h] == c;
      if (cnt >= 64) continue;  // 4 lanes per face, 256 threads: one step per colour
      colour[f] = c;
      if (faces[f].p0 < NPT) used[c][faces[f].p0] = 1;
      if (faces[f].p1 < NPT) used[c][faces[f].p1] = 1;
      ncol = std::max(ncol, c + 1);
      break;
    }
  for (size_t f = 0; f < faces.size(); f++) if (colour[f] < 0) { printf("colouring needs more than %d colours\n", MAXCOL); exit(1); }
  // the face-major order: by colour; the point lists name faces by their position in that order
  std::vector<int> order(faces.size()), pos(faces.size());
  for (size_t f = 0; f < faces.size(); f++) order[f] = (int)f;
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return colour[a] < colour[b]; });
  memset(&T, 0, sizeof T);
  memset(&P, 0, sizeof P);
  for (size_t i = 0; i < order.size(); i++) {
    pos[order[i]] = (int)i;
    T.f0[i] = (unsigned char)faces[order[i]].p0;
    T.f1[i] = (unsigned char)faces[order[i]].p1;
    for (int k = 0; k < 3; k++) T.g.nrm[i][k] = (rand() % 2001 - 1000) * 1e-3;
  }
  for (int c = 0; c <= ncol; c++) T.coff[c] = 0;
  for (size_t i = 0; i < order.size(); i++) T.coff[colour[order[i]] + 1]++;
  for (int c = 0; c < ncol; c++) T.coff[c + 1] += T.coff[c];
  for (int i = 0; i < NPT * NNB; i++) {
    P.pface[i] = (unsigned short)(pos[plist[i] & 1023] | ((plist[i] >> 18) & 1) << 15);
    P.prow[i] = (unsigned char)((plist[i] >> 10) & 255);
  }
  for (int r = 0; r < nrows; r++) for (int e = 0; e < 8; e++) T.g.var[r][e] = (rand() % 2001 - 1000) * 1e-3;
  P.g = T.g;
  T.ncol = ncol; T.nfaces = (int)faces.size();
  int inner = 0;
  for (auto &f : faces) inner += f.p1 < NPT;
  printf("tile: %d owned points, %d rows, %zu faces (%d with both ends owned, %zu cut), %d incidences, %d colours of <= 64 faces\n", NPT,
         nrows, faces.size(), inner, faces.size() - inner, NPT * NNB, ncol);
}

// socket power (W) and shader clock (MHz) right now, from rocm-smi (0 when it cannot be read)
static void smi(double &watt, double &mhz) {
  watt = mhz = 0;
  FILE *f = popen("rocm-smi --showclocks --showpower --csv 2>/dev/null", "r");
  if (!f) return;
  char hdr[4096] = "", val[4096] = "";
  if (fgets(hdr, sizeof hdr, f) && fgets(val, sizeof val, f)) {
    std::vector<std::string> h, v;
    for (char *tk = strtok(hdr, ",\n"); tk; tk = strtok(nullptr, ",\n")) h.push_back(tk);
    for (char *tk = strtok(val, ",\n"); tk; tk = strtok(nullptr, ",\n")) v.push_back(tk);
    for (size_t i = 0; i < h.size() && i < v.size(); i++) {
      if (h[i].find("sclk clock speed") != std::string::npos) { std::string d; for (char c : v[i]) if (c >= '0' && c <= '9') d += c; mhz = atof(d.c_str()); }
      if (h[i].find("Power") != std::string::npos) watt = atof(v[i].c_str());
    }
  }
  pclose(f);
}

template <typename K, typename IMG> static double run(const char *what, K kern, const IMG *d_img, double *d_out, size_t lds, int occ, int cus,
                                        std::vector<double> &res) {
  const int blocks = cus * occ, iters = 400;
  CK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  int got = 0;
  CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&got, kern, 256, lds));
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  float best = 1e30f;
  for (int rep = 0; rep < 5; rep++) {
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, 0, d_img, d_out, iters);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    best = std::min(best, ms);
  }
  res.resize(NPT * 24);
  CK(hipMemcpy(res.data(), d_out, sizeof(double) * NPT * 24, hipMemcpyDeviceToHost));
  const double clk = best * 1e-3 * 2.4e9 / ((double)occ * iters);
  // the same launch back to back for ~3 s with the socket's power read beside it: joules per tile, the quantity the
  // product's time follows
  std::atomic<bool> stop{false};
  std::vector<double> ws, fs;
  std::thread th([&] { while (!stop) { double w, m; smi(w, m); if (w > 0) { ws.push_back(w); fs.push_back(m); } } });
  const auto t0 = std::chrono::steady_clock::now();
  long launches = 0;
  while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < 3.0) {
    for (int i = 0; i < 20; i++) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, 0, d_img, d_out, iters);
    CK(hipDeviceSynchronize());
    launches += 20;
  }
  const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  stop = true; th.join();
  double w = 0, m = 0;
  if (ws.size() > 2) { std::sort(ws.begin() + 1, ws.end()); std::sort(fs.begin() + 1, fs.end()); w = ws[1 + (ws.size() - 1) / 2]; m = fs[1 + (fs.size() - 1) / 2]; }
  const double tiles = (double)launches * blocks * iters;
  printf("%-64s %d workgroups per CU (occupancy query %d, %5.1f KiB LDS): %7.0f clk per tile and CU (at 2.4 GHz); back to back %5.0f W at %4.0f MHz, %6.1f ns and %6.2f uJ per tile above the idle socket\n",
         what, occ, got, lds / 1024.0, clk, w, m, secs / tiles * 1e9, (w - 239.0) * secs / tiles * 1e6);
  return clk;
}

int main() {
  int cus = 0;
  CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
  static image_p P;
  static image_f T;
  build(P, T);
  image_p *d_p; image_f *d_f; double *d_out;
  CK(hipMalloc(&d_p, sizeof P)); CK(hipMalloc(&d_f, sizeof T)); CK(hipMalloc(&d_out, sizeof(double) * NPT * 24));
  CK(hipMemcpy(d_p, &P, sizeof P, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_f, &T, sizeof T, hipMemcpyHostToDevice));
  const size_t img = sizeof(image_p), imgf = sizeof(image_f), acc = NPT * 24 * sizeof(double);
  printf("LDS image: point-centric %zu bytes, face-major %zu + %zu of accumulators; %d CUs\n", img, imgf, acc, cus);
  std::vector<double> rp, rf, ra;
  for (int occ = 3; occ <= 5; occ++) {
    // occupancy forced by the LDS request (160 KiB per CU): 3 -> 48 KiB, 4 -> 36 KiB, 5 -> 32 KiB
    const size_t want = occ == 3 ? 48 * 1024 : occ == 4 ? 36 * 1024 : 32 * 1024;
    if (img <= want) run("P point-centric, 4 lanes per point x 2 equations, sums in registers", loop_points, d_p, d_out, want, occ, cus, rp);
    if (img <= want && occ == 5) {
      std::vector<double> rs;
      auto same = [&](const char *nm) {
        double worst = 0, scale = 0;
        for (int i = 0; i < NPT * 24; i++) { scale = std::max(scale, fabs(rp[i])); worst = std::max(worst, fabs(rp[i] - rs[i])); }
        printf("    (%s against P: largest difference %.3g of %.3g)\n", nm, worst, scale);
      };
      run("S4 incidences split over 4 lanes per point, all equations per lane", loop_points_split<4>, d_p, d_out, want, occ, cus, rs); same("S4");
      run("S2 incidences split over 2 lanes per point (half the lanes idle)", loop_points_split<2>, d_p, d_out, want, occ, cus, rs); same("S2");
      run("S1 one lane per point (three quarters of the lanes idle)", loop_points_split<1>, d_p, d_out, want, occ, cus, rs); same("S1");
    }
    if (imgf + acc <= want) {
      run("F face-major, coloured, plain read-modify-write, barrier per colour", loop_faces<false>, d_f, d_out, want, occ, cus, rf);
      run("A face-major, ds_add_f64, no colours (order not reproducible)", loop_faces<true>, d_f, d_out, want, occ, cus, ra);
    } else
      printf("  (the face-major forms do not fit %zu KiB: image %zu + accumulators %zu bytes)\n", want / 1024, imgf, acc);
  }
  double worst = 0, worst_a = 0, scale = 0;
  for (int i = 0; i < NPT * 24 && !rf.empty(); i++) {
    if ((i % 6) / 3 + 2 * ((i / 6) % 4) >= 8) continue;
    scale = std::max(scale, fabs(rp[i]));
    worst = std::max(worst, fabs(rp[i] - rf[i]));
    worst_a = std::max(worst_a, fabs(rp[i] - ra[i]));
  }
  printf("largest difference of the sums, face-major vs point-centric: %.3g (coloured), %.3g (atomic); largest sum %.3g\n", worst, worst_a, scale);
  return (worst <= 1e-12 * scale && worst_a <= 1e-12 * scale) ? 0 : 1;
}
