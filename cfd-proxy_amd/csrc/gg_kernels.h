// gg_kernels.h -- launch wrappers of the gfx950 kernels (internal to libcfdproxy_hip.so).
#ifndef CFDP_GG_KERNELS_H
#define CFDP_GG_KERNELS_H

#include <hip/hip_runtime.h>

#include "cfdproxy_host.h"  // cfdp_tile_desc

struct gg_args {
  const cfdp_tile_desc *tiles;  // device copies
  const uint4 *blob;
  const int *halo_idx;
  const double *var;            // [nall][8]: 7 variables + the dual volume in slot 7
  double *grad;                 // [nall][21]
  double *flux;                 // [nown][3]
  int nown;
};

// pipeline: 0 = one workgroup per tile; k > 0 = persistent double-buffered LDS-DMA kernel
// with at most k workgroups per CU (falls back to 0 when two buffers do not fit in LDS)
hipError_t gg_launch_gradient(const gg_args &a, int lanes, int tile_begin, int ntiles,
                              int tile_points, size_t lds, int max_halo, int max_blob_qw,
                              int pipeline, bool nt, hipStream_t stream);
hipError_t gg_launch_flux(const gg_args &a, int lanes, bool refmode, int tile_begin, int ntiles,
                          int tile_points, size_t lds, int max_halo, int max_blob_qw, bool nt,
                          hipStream_t stream);
hipError_t gg_launch_pack(const int *send_idx, int nsend, const double *grad, double *sendbuf,
                          hipStream_t stream);
hipError_t gg_launch_unpack(const double *recvbuf, int nrecv, int nown, double *grad,
                            hipStream_t stream);
extern int gg_debug_flags;
hipError_t gg_set_max_lds(size_t lds_grad, size_t lds_flux);

#endif
