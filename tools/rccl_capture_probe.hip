// Does ncclSend/ncclRecv survive hipStream capture with THIS librccl?  (DESIGN appendix A: with the RCCL 2.26.6 PyTorch bundles,
// hipStreamEndCapture crashed.)   usage: rccl_capture_probe /path/to/librccl.so
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <dlfcn.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 2; } } while (0)
#define NK(x) do { ncclResult_t r_ = (x); if (r_ != ncclSuccess) { printf("RCCL error %d at %s:%d\n", (int)r_, __FILE__, __LINE__); return 3; } } while (0)
int main(int argc, char **argv) {
  const char *path = argc > 1 ? argv[1] : "librccl.so";
  void *lib = dlopen(path, RTLD_NOW | RTLD_GLOBAL);
  if (!lib) { printf("cannot load %s: %s\n", path, dlerror()); return 1; }
  auto GetVersion = (ncclResult_t(*)(int *))dlsym(lib, "ncclGetVersion");
  auto GetUniqueId = (ncclResult_t(*)(ncclUniqueId *))dlsym(lib, "ncclGetUniqueId");
  auto CommInitRank = (ncclResult_t(*)(ncclComm_t *, int, ncclUniqueId, int))dlsym(lib, "ncclCommInitRank");
  auto GroupStart = (ncclResult_t(*)())dlsym(lib, "ncclGroupStart");
  auto GroupEnd = (ncclResult_t(*)())dlsym(lib, "ncclGroupEnd");
  auto Send = (ncclResult_t(*)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t))dlsym(lib, "ncclSend");
  auto Recv = (ncclResult_t(*)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t))dlsym(lib, "ncclRecv");
  auto CommDestroy = (ncclResult_t(*)(ncclComm_t))dlsym(lib, "ncclCommDestroy");
  int ver = 0;
  if (GetVersion) GetVersion(&ver);
  printf("librccl %s version %d\n", path, ver); fflush(stdout);
  CK(hipSetDevice(0));
  ncclUniqueId id; NK(GetUniqueId(&id));
  ncclComm_t comm; NK(CommInitRank(&comm, 1, id, 0));
  const size_t n = 1 << 16;
  double *a, *b; CK(hipMalloc(&a, n * 8)); CK(hipMalloc(&b, n * 8));
  CK(hipMemset(a, 1, n * 8)); CK(hipMemset(b, 0, n * 8));
  hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  // eager once (warms the communicator)
  NK(GroupStart()); NK(Send(a, n, ncclDouble, 0, comm, st)); NK(Recv(b, n, ncclDouble, 0, comm, st)); NK(GroupEnd());
  CK(hipStreamSynchronize(st));
  printf("eager self send/recv ok\n"); fflush(stdout);
  hipGraph_t gr = nullptr; hipGraphExec_t ge = nullptr;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
  for (int i = 0; i < 4; i++) { NK(GroupStart()); NK(Send(a, n, ncclDouble, 0, comm, st)); NK(Recv(b, n, ncclDouble, 0, comm, st)); NK(GroupEnd()); }
  printf("captured 4 groups; ending capture ...\n"); fflush(stdout);
  CK(hipStreamEndCapture(st, &gr));
  printf("capture ended; instantiating ...\n"); fflush(stdout);
  CK(hipGraphInstantiate(&ge, gr, nullptr, nullptr, 0));
  for (int r = 0; r < 3; r++) CK(hipGraphLaunch(ge, st));
  CK(hipStreamSynchronize(st));
  double h = 0; CK(hipMemcpy(&h, b, 8, hipMemcpyDeviceToHost));
  printf("GRAPH REPLAY OK (b[0] bytes %s)\n", h != 0.0 ? "copied" : "NOT copied");
  NK(CommDestroy(comm));
  return 0;
}
