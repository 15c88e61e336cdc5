// voxel.h — batched pcl::VoxelGrid<PointXYZI> on the device.
//
// A "job" is one VoxelGrid::filter call (one cloud, one leaf size); a round processes all jobs of a
// context with three launches (work lists, then one workgroup per job; kernels_voxel.hip): vox_small for clouds of up to
// 8192 points (LDS resident), vox_big for the local maps (stable LSD radix sort of (voxel id, position)).
// Output: one point per voxel in ascending voxel id, f32 sums accumulated in original order, divided by the count.
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
  int cap;              // capacity of in (points)
  int out_cap;          // capacity of out: further voxels are dropped and *overflow is raised
  int* overflow;        // device flag (may be nullptr), set to 1 when the output was truncated
  int off;              // offset of this job's region in the key / pair scratch arrays (filled by vox_create)
  // mode 1 ("sort only"): out = the input points in ascending voxel id (stable: the order inside a voxel is the input order),
  // n_out = n.  Used when a key frame enters LaserMapping's ring: PCL's voxel id orders points by (floor(z/leaf), floor(y/leaf),
  // floor(x/leaf)) whatever the bounding box, so key frames sorted once can be merged into every local map they belong to.
  int mode;
  const int* out_sel;   // mode 1 (may be nullptr): the output goes to out + *out_sel * out_stride (ring entry chosen on the device)
  int out_stride;
  float* box_out;       // mode 1 (may be nullptr): min xyz / max xyz of the cloud as 8 floats at box_out + *out_sel * 8
  int* n_sel_out;       // mode 1 (may be nullptr): n is also stored at n_sel_out[*out_sel * n_sel_stride]
  int n_sel_stride;
};

struct VoxCtx {
  VoxJob* jobs;         // device array [njobs]
  int njobs;
  unsigned* bbox;       // [job][8] ordered-int encoded min xyz (0..2) and ~max xyz (4..6) of the input cloud
  unsigned* keys;       // voxel id per input point; later the list of voxel run starts
  unsigned long long *pairs_a, *pairs_b;  // (voxel id << 32 | position) ping-pong buffers of the radix passes
  int *list_small, *list_big, *cnt;       // work lists of a round (vox_plan): enabled jobs of <= 8192 / more points; cnt[2]
  int grid_small, grid_big;               // workgroups launched for each list (default njobs; persistent loop over the list)
  unsigned total;       // total scratch elements
};

// host: allocate scratch for `jobs` (host copy of the job table), upload the table
int vox_create(VoxCtx* V, const VoxJob* jobs, int njobs, std::string* err);
void vox_destroy(VoxCtx* V);
// enqueue one round over all jobs of V
int vox_run(const VoxCtx& V, hipStream_t st, std::string* err);

#endif
