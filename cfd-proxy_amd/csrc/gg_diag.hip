// gg_diag.hip -- the DIAGNOSTIC instantiations of the phase-split fused pass (gg_device.h), in a library of their own,
// lib/libcfdproxy_diag.so, which libcfdproxy_hip.so loads only when a diagnostic is asked for:
//   DIAG = 1   phase stamps: thread 0 of every workgroup writes shader-clock stamps of its phase boundaries
//              (cfdp_gpu_debug_phase_stamps, tools/phase_stamps.py)
//   DIAG = 2   data movement only: every load and every store of the pass, neither face loop -- the floor bench.py reports
//              beside the pass (cfdp_gpu_time_fused_movement, roofline.movement_only_us)
//   DIAG = 3   the timing experiment of EXPERIMENTS.md D.2 (CFDP_EXP_SKIP_PRE, an experiment switch: values wrong)
// at the two capacities of the pass.  None of them is a kernel a timed run can execute, so none of them is in the product
// library; merely compiled into the product kernels and switched off, the stamps cost 1-2.5 %.
#include "gg_device.h"

namespace {
template <typename K, typename... A>
hipError_t launch(K *kernel, int grid, int block, size_t lds, hipStream_t stream, A... args) {
  if (lds > 64 * 1024) return hipErrorInvalidConfiguration;  // (both capacities are below the default dynamic-LDS limit)
  hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), lds, stream, args...);
  return hipGetLastError();
}
template <int CB, int KV, int KG, int KX, int D, bool P, bool N>
hipError_t go(const gg_args &a, const gg_grad_view &gnew, int tile_begin, int ntiles, int block, hipStream_t stream, int dbgf,
              const gg_push_args &pa) {
  return launch(gg_fused_split_kernel<false, N, CB, KV, KG, KX, D, true, P>, ntiles, block, (size_t)(CB + KX) * block * 16, stream, a.tiles,
                tile_begin, a.blob, a.halo_idx, a.rowlist, a.rowlist_stride, a.var, a.grad.a, a.grad.ghost, a.flux, a.nown, gnew, dbgf, pa);
}
}  // namespace

extern "C" {

// diag: 1 stamps, 2 movement only, 3 skip-pre; large: the <6,4,3,4> capacity instead of <5,3,3,3>; needs the fixed-stride row
// lists (a.rowlist).  hipErrorNotSupported: no such instantiation
hipError_t gg_diag_launch_fused(int diag, int large, int nt, const gg_args *a, const gg_grad_view *gnew, int tile_begin, int ntiles,
                                int block, hipStream_t stream, int dbgf, const gg_push_args *pa) {
  if (!a || !gnew || !pa || !a->rowlist) return hipErrorNotSupported;
#define GO(CB, KV, KG, KX, D, P) \
  (nt ? go<CB, KV, KG, KX, D, P, true>(*a, *gnew, tile_begin, ntiles, block, stream, dbgf, *pa) \
      : go<CB, KV, KG, KX, D, P, false>(*a, *gnew, tile_begin, ntiles, block, stream, dbgf, *pa))
  if (diag == 1) return large ? GO(6, 4, 3, 4, 1, true) : GO(5, 3, 3, 3, 1, true);
  if (diag == 2) return large ? GO(6, 4, 3, 4, 2, false) : GO(5, 3, 3, 3, 2, false);
  if (diag == 3 && !large) return GO(5, 3, 3, 3, 3, false);
#undef GO
  return hipErrorNotSupported;
}

hipError_t gg_diag_set_stamp_buffer(unsigned long long *dev) {
  return hipMemcpyToSymbol(HIP_SYMBOL(gg_stamp_buf), &dev, sizeof dev);
}

}  // extern "C"
