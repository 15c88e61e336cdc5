// voxel.h — batched pcl::VoxelGrid<PointXYZI> on the device.
//
// A "job" is one VoxelGrid::filter call (one cloud, one leaf size).  All jobs of a round
// are processed by the same launches (blockIdx.y = job):
//   vox_bbox      getMinMax3D (ordered-int atomics)
//   vox_keys      voxel index per point (f32 floor arithmetic exactly as PCL) + bucket histogram
//   vox_bscan / vox_bscatter / vox_bsort   two-level bucket sort of (voxel id, position)
//   vox_vscan / vox_bcentroid   one output point per voxel in ascending voxel id, f32 sums accumulated
//                 in sorted (= original) order, divided by the count
#ifndef ALEGO_VOXEL_H_
#define ALEGO_VOXEL_H_
#include <hip/hip_runtime.h>

#include <string>

struct VoxJob {
  const float4* in;     // input cloud
  const int* n_in;      // device count
  float4* out;          // output cloud
  int* n_out;           // device count
  const int* enable;    // device flag (nullptr = always); a disabled job keeps its previous output
  float leaf;
  int cap;              // capacity of in / out (points)
  int off;              // offset of this job's region in the key / pair scratch arrays
  int nbcap, boff0;     // bucket capacity (power of two) and offset of this job's region in the bucket arrays
};

#define VX_GEOM 16   // ints of per-job geometry: 0-2 min_b, 3 mul1, 4 mul2, 5 n, 6 passthrough, 7 bucket shift, 8 bucket count

struct VoxCtx {
  VoxJob* jobs;         // device array [njobs]
  int njobs, max_cap, gx;  // gx = blocks per job (kernels grid-stride over chunks)
  unsigned* bbox;       // [job][8] ordered-int encoded min xyz (0..2) and ~max xyz (4..6)
  int* geom;            // [job][VX_GEOM]
  unsigned* keys;       // voxel id per input point
  unsigned long long *pairs_a, *pairs_b;  // (voxel id << 32 | position): bucketed, then sorted
  int *bcnt, *boff, *bcur, *bvox, *voff;  // per-job regions: bucket histogram / offsets / cursors / voxel counts / output ranks
  int *pt_items, *bk_items;  // [njobs+1] exclusive scans of the per-job work-item counts (points / buckets)
  unsigned total;       // total scratch elements
};

// host: allocate scratch for `jobs` (host copy of the job table), upload the table
int vox_create(VoxCtx* V, const VoxJob* jobs, int njobs, std::string* err);
void vox_destroy(VoxCtx* V);
// enqueue one round over all jobs of V
int vox_run(const VoxCtx& V, hipStream_t st, std::string* err);

#endif
