// voxel.h — batched pcl::VoxelGrid<PointXYZI> on the device.
//
// A "job" is one VoxelGrid::filter call (one cloud, one leaf size).  All jobs of a round
// are processed by the same launches (blockIdx.y = job):
//   vox_bbox      getMinMax3D (ordered-int atomics)
//   vox_keys      voxel index per point (f32 floor arithmetic exactly as PCL), value = position
//   rocprim::segmented_radix_sort_pairs (stable: equal voxel ids keep the input order)
//   vox_heads / vox_scan / vox_centroid   one output point per voxel in ascending voxel id,
//                 f32 sums accumulated in sorted (= original) order, divided by the count
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
  int off;              // offset of this job's region in the key / value scratch arrays
};

struct VoxCtx {
  VoxJob* jobs;         // device array [njobs]
  int njobs, max_cap, gx;  // gx = blocks per job (kernels grid-stride over chunks)
  unsigned* bbox;       // [job][8] ordered-int encoded min xyz (0..2) and ~max xyz (4..6)
  int* geom;            // [job][8] min_b xyz, mul1, mul2, n, passthrough
  unsigned *keys_a, *keys_b;
  int *vals_a, *vals_b;
  int *seg_begin, *seg_end;  // [njobs]
  int* blk_cnt;         // [job][blk_stride]
  int blk_stride;
  void* sort_tmp;
  size_t sort_tmp_bytes;
  unsigned total;       // total scratch elements
};

// host: allocate scratch for `jobs` (host copy of the job table), upload the table
int vox_create(VoxCtx* V, const VoxJob* jobs, int njobs, std::string* err);
void vox_destroy(VoxCtx* V);
// enqueue one round over all jobs of V
int vox_run(const VoxCtx& V, hipStream_t st, std::string* err);

#endif
