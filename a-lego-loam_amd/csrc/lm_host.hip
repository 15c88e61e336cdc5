// lm_host.hip — placeholder until the LaserMapping kernels land: /odom_aft_mapped = /odom/lidar.
#include "lm_host.h"

struct LmHost { int n_slots; hipStream_t st; };

__global__ void lm_passthrough(DevCtx d) {
  const int slot = blockIdx.x * blockDim.x + threadIdx.x;
  if (slot >= d.n_launch) return;
  double* po = d.poses + (size_t)(slot + d.slot0) * 16;
  for (int k = 0; k < 7; ++k) po[7 + k] = po[k];
}

LmHost* lm_host_create(const alego_params&, const DevCtx&, int n_slots, hipStream_t st, std::string*) { return new LmHost{n_slots, st}; }
void lm_host_destroy(LmHost* lm) { delete lm; }
int lm_host_enqueue(LmHost* lm, const DevCtx& d, int, std::string*) {
  hipLaunchKernelGGL(lm_passthrough, dim3((d.n_launch + 63) / 64), dim3(64), 0, lm->st, d);
  return 0;
}
int lm_host_process_host(LmHost*, const DevCtx&, const alego_point*, int, const alego_point*, int, const alego_point*, int,
                         const alego_pose*, alego_pose*, std::string* err) { *err = "alego_lm_process: not implemented yet"; return ALEGO_ERR_ARG; }
void lm_host_get_params(LmHost*, int, double* p6) { for (int i = 0; i < 6; ++i) p6[i] = 0; }
int lm_host_set_params(LmHost*, int, const double*, std::string*) { return 0; }
void lm_host_get_counts(LmHost*, int, int* o) { for (int i = 0; i < 6; ++i) o[i] = 0; }
int lm_host_debug_get(LmHost*, int, const char* name, void*, int, int*, int*, std::string* err) { *err = std::string("debug_get: unknown name ") + name; return ALEGO_ERR_ARG; }
