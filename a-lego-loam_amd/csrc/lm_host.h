// lm_host.h — LaserMapping scan-to-map registration: host sequencing + HBM state.
#ifndef ALEGO_LM_HOST_H_
#define ALEGO_LM_HOST_H_
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "../../include/alego_mi355x.h"
#include "dev_common.h"

struct LmHost;
LmHost* lm_host_create(const alego_params& P, const DevCtx& d, int n_slots, int gsize, const std::vector<hipStream_t>& streams, std::string* err);
void lm_host_destroy(LmHost* lm);
// LaserMapping for the scan just processed by LO, for the slots of view `d`
// st_override: enqueue on this stream instead of the slot's group stream (alego_stream_run pipelines LaserMapping on a stream of its own)
int lm_host_enqueue(LmHost* lm, const DevCtx& d, const std::vector<char>& odom_valid, std::string* err, hipStream_t st_override = nullptr);
// the batch path: LaserMapping of this scan on `back`, behind the hand-over kernel (lm_stage) on `front`; k = scans handed over so far for
// this stream group, back_same / back_other = events recorded on `back` after the LaserMapping of scans k - 2 / k - 1
int lm_host_enqueue_async(LmHost* lm, const DevCtx& d, const std::vector<char>& odom_valid, std::string* err, hipStream_t front, hipStream_t back,
                          hipEvent_t staged, hipEvent_t back_same, hipEvent_t back_other, long k);
int lm_host_get_flags(LmHost* lm, int slot);
int lm_host_process_host(LmHost* lm, const DevCtx& d, const alego_point* corner_last, int n_corner, const alego_point* surf_last,
                         int n_surf, const alego_point* outlier, int n_outlier, const alego_pose* odom, alego_pose* map_pose,
                         std::string* err);
void lm_host_get_params(LmHost* lm, int slot, double* p6);
int lm_host_set_params(LmHost* lm, int slot, const double* p6, std::string* err);
void lm_host_get_counts(LmHost* lm, int slot, int* out6);
int lm_host_keyframe_count(LmHost* lm, int slot);
int lm_host_get_keyframe(LmHost* lm, int slot, int kf_id, alego_keyframe* out, std::string* err);
int lm_host_set_keypose(LmHost* lm, const DevCtx& d, int slot, int kf_id, const float* pose6, std::string* err);
int lm_host_reset_window(LmHost* lm, int slot, std::string* err);
int lm_host_apply_correction(LmHost* lm, const DevCtx& d, int slot, const double* rc12, std::string* err);
int lm_host_add_keyframe(LmHost* lm, const DevCtx& d, int slot, const float* pose6, const alego_point* corner, int nc, const alego_point* surf, int ns,
                         const alego_point* outlier, int no, std::string* err);
int lm_host_dist_unique_id(char* id128);
int lm_host_dist_init(LmHost* lm, int rank, int world, const char* id128, std::string* err);
int lm_host_dist_shutdown(LmHost* lm);
int lm_host_dist_probe(LmHost* lm, int iters, double* usec, std::string* err);
int lm_host_debug_slice(LmHost* lm, int rank, int world, std::string* err);
int lm_host_set_map_merge(LmHost* lm, int on, std::string* err);
int lm_host_debug_get(LmHost* lm, int slot, const char* name, void* out, int cap_bytes, int* count, int* dtype, std::string* err);
#endif
