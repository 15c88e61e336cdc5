// kernels_icp.hip — the numeric part of loop closure (src/laserMapping.cpp:652-824): sub-map assembly of detectLoopClosure
// (:794-812) and pcl::IterativeClosestPoint as performLoopClosure configures it (:670-692).  The pose graph (GTSAM) and the
// decision logic stay on the host (alego_loop_detect is a host helper); this is the per-point work.
//
//   icp_build     transformPointCloud of the newest key frame (source: surf, corner, outlier) and of the history frames (raw target)
//   (voxel.h)     VoxelGrid(lc_leaf) of the raw target -> near_history_keyframes_
//   icp_corr      one thread per source point: apply the last iteration's transformation (f32, as pcl::transformPointCloud), exact
//                 1-NN in the target (f32 squared distance, ties -> lowest index; target tiled through LDS), correspondences within the
//                 maximum distance feed the sums of the closed-form alignment (f64): per-workgroup partials in fixed order
//   icp_step      one workgroup: partials -> sums, Horn's closed-form rigid transform (largest eigenvector of the 4x4 quaternion
//                 matrix; PCL's TransformationEstimationSVD solves the same least-squares problem by SVD in f32), accumulation of
//                 final_transformation_ (Matrix4f), DefaultConvergenceCriteria (iterations, transformation epsilon, absolute and
//                 relative MSE)
//   icp_fitness   getFitnessScore(): mean squared distance of the source under the final transformation to its nearest target point
// The host enqueues icp_max_iters x (icp_corr, icp_step); kernels of a finished alignment return at once.
#include <vector>

#include "dev_common.h"
#include "lm_ctx.h"
#include "prof.h"
#include "voxel.h"

#define ICP_T 256
#define ICP_TILE 2048

struct IcpState {
  float M[16];        // transformation_ of the last iteration (applied to the source by the next icp_corr)
  float Tf[16];       // final_transformation_
  double prev_mse, fitness;
  int iter, done, converged, apply, n_src, n_tgt;
};

struct IcpFrame { float pose[6]; int off[4]; int dst[3]; };   // raw cloud offsets (corner, surf, outlier, end) and destination offsets of the three clouds

__global__ void __launch_bounds__(ICP_T) icp_build(const IcpFrame* frames, int nframes, const float4* raw, float4* src, float4* tgt_raw) {
  const int f = blockIdx.y;
  const IcpFrame F = frames[f];
  float m[3][4];
  float kp[8] = {F.pose[0], F.pose[1], F.pose[2], F.pose[3], F.pose[4], F.pose[5], 0.f, 0.f};
  keypose_matrix(kp, m);
  float4* dst = f == 0 ? src : tgt_raw;
  for (int kind = 0; kind < 3; ++kind) {
    const int n = F.off[kind + 1] - F.off[kind];
    for (int i = blockIdx.x * ICP_T + threadIdx.x; i < n; i += gridDim.x * ICP_T) dst[F.dst[kind] + i] = kf_transform(m, raw[F.off[kind] + i]);
  }
}

// symmetric 4x4 eigen-decomposition, cyclic Jacobi (same algorithm as oracle_icp.h)
DEV_INLINE void jacobi4(double A[4][4], double V[4][4], double lam[4]) {
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) V[i][j] = i == j ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 60; ++sweep) {
    double off = 0;
    for (int p = 0; p < 4; ++p) for (int q = p + 1; q < 4; ++q) off += A[p][q] * A[p][q];
    if (off < 1e-300) break;
    for (int p = 0; p < 3; ++p)
      for (int q = p + 1; q < 4; ++q) {
        const double apq = A[p][q];
        if (apq == 0.0) continue;
        const double theta = (A[q][q] - A[p][p]) / (2.0 * apq);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < 4; ++k) { const double akp = A[k][p], akq = A[k][q]; A[k][p] = c * akp - s * akq; A[k][q] = s * akp + c * akq; }
        for (int k = 0; k < 4; ++k) { const double apk = A[p][k], aqk = A[q][k]; A[p][k] = c * apk - s * aqk; A[q][k] = s * apk + c * aqk; }
        for (int k = 0; k < 4; ++k) { const double vkp = V[k][p], vkq = V[k][q]; V[k][p] = c * vkp - s * vkq; V[k][q] = s * vkp + c * vkq; }
      }
  }
  for (int i = 0; i < 4; ++i) lam[i] = A[i][i];
}

// exact 1-NN of p in tgt[0..n): f32 squared distance as flann::L2_Simple, ties -> lowest index; the target goes through LDS in tiles
DEV_INLINE void icp_nn(const float4 p, const float4* tgt, int n, float4* s_tile, float* best_d, int* best_i) {
  float bd = 3.402823466e+38f;
  int bi = -1;
  for (int t0 = 0; t0 < n; t0 += ICP_TILE) {
    const int m = min(ICP_TILE, n - t0);
    __syncthreads();
    for (int j = threadIdx.x; j < m; j += ICP_T) s_tile[j] = tgt[t0 + j];
    __syncthreads();
    for (int j = 0; j < m; ++j) {
      const float4 q = s_tile[j];
      float r = 0.f, df;
      df = q.x - p.x; r += df * df; df = q.y - p.y; r += df * df; df = q.z - p.z; r += df * df;
      if (r < bd) { bd = r; bi = t0 + j; }
    }
  }
  *best_d = bd; *best_i = bi;
}

// grid ceil(n_src / ICP_T); partial[wg][18]: 15 sums, mse sum, count
__global__ void __launch_bounds__(ICP_T) icp_corr(IcpState* S, float4* cur, const float4* tgt, double max_d2, double* partial) {
  if (S->done) return;
  __shared__ float4 s_tile[ICP_TILE];
  __shared__ double s_red[ICP_T / 64][17];
  const int n_src = S->n_src, n_tgt = S->n_tgt;
  const int i = blockIdx.x * ICP_T + threadIdx.x;
  float4 p = cur[min(i, n_src - 1)];
  if (S->apply) {   // transformCloud of the previous iteration (f32)
    const float* M = S->M;
    const float x = p.x, y = p.y, z = p.z;
    p.x = M[0] * x + M[1] * y + M[2] * z + M[3]; p.y = M[4] * x + M[5] * y + M[6] * z + M[7]; p.z = M[8] * x + M[9] * y + M[10] * z + M[11];
    if (i < n_src) cur[i] = p;
  }
  float d2;
  int idx;
  icp_nn(p, tgt, n_tgt, s_tile, &d2, &idx);
  double v[17];
#pragma unroll
  for (int k = 0; k < 17; ++k) v[k] = 0.0;
  if (i < n_src && idx >= 0 && (double)d2 <= max_d2) {
    const float4 q = tgt[idx];
    const double a[3] = {p.x, p.y, p.z}, b[3] = {q.x, q.y, q.z};
#pragma unroll
    for (int k = 0; k < 3; ++k) { v[k] = a[k]; v[3 + k] = b[k]; }
#pragma unroll
    for (int u = 0; u < 3; ++u)
#pragma unroll
      for (int w = 0; w < 3; ++w) v[6 + u * 3 + w] = a[u] * b[w];
    v[15] = (double)d2; v[16] = 1.0;
  }
#pragma unroll
  for (int k = 0; k < 17; ++k) v[k] = wave_sum_f64(v[k]);
  if (lane_id() == 0) for (int k = 0; k < 17; ++k) s_red[threadIdx.x >> 6][k] = v[k];
  __syncthreads();
  if (threadIdx.x < 17) {
    double t = 0;
    for (int w = 0; w < ICP_T / 64; ++w) t += s_red[w][threadIdx.x];
    partial[(size_t)blockIdx.x * 18 + threadIdx.x] = t;
  }
}

__global__ void icp_step(IcpState* S, const double* partial, int nwg, alego_params P) {
  if (threadIdx.x != 0 || S->done) return;
  double T[17];
  for (int k = 0; k < 17; ++k) { double t = 0; for (int w = 0; w < nwg; ++w) t += partial[(size_t)w * 18 + k]; T[k] = t; }
  const double n = T[16];
  if (n < 3.0) { S->done = 1; S->converged = 0; return; }   // "Not enough correspondences found"
  const double mse = T[15] / n;
  const double ms[3] = {T[0] / n, T[1] / n, T[2] / n}, mt[3] = {T[3] / n, T[4] / n, T[5] / n};
  double Mc[3][3];
  for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) Mc[a][b] = T[6 + a * 3 + b] - n * ms[a] * mt[b];
  double Nq[4][4] = {{Mc[0][0] + Mc[1][1] + Mc[2][2], Mc[1][2] - Mc[2][1], Mc[2][0] - Mc[0][2], Mc[0][1] - Mc[1][0]},
                     {Mc[1][2] - Mc[2][1], Mc[0][0] - Mc[1][1] - Mc[2][2], Mc[0][1] + Mc[1][0], Mc[2][0] + Mc[0][2]},
                     {Mc[2][0] - Mc[0][2], Mc[0][1] + Mc[1][0], -Mc[0][0] + Mc[1][1] - Mc[2][2], Mc[1][2] + Mc[2][1]},
                     {Mc[0][1] - Mc[1][0], Mc[2][0] + Mc[0][2], Mc[1][2] + Mc[2][1], -Mc[0][0] - Mc[1][1] + Mc[2][2]}};
  double V[4][4], lam[4];
  jacobi4(Nq, V, lam);
  int best = 0;
  for (int i = 1; i < 4; ++i) if (lam[i] > lam[best]) best = i;
  double q[4] = {V[0][best], V[1][best], V[2][best], V[3][best]};
  if (q[0] < 0) for (int i = 0; i < 4; ++i) q[i] = -q[i];
  const double nn = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  const double w = q[0] / nn, x = q[1] / nn, y = q[2] / nn, z = q[3] / nn;
  const double R[9] = {1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y), 2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                       2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)};
  float M[16];
  for (int a = 0; a < 3; ++a) {
    for (int b = 0; b < 3; ++b) M[a * 4 + b] = (float)R[a * 3 + b];
    M[a * 4 + 3] = (float)(mt[a] - (R[a * 3 + 0] * ms[0] + R[a * 3 + 1] * ms[1] + R[a * 3 + 2] * ms[2]));
  }
  M[12] = 0.f; M[13] = 0.f; M[14] = 0.f; M[15] = 1.f;
  float Nf[16];   // final_transformation_ = transformation_ * final_transformation_ (Matrix4f)
  for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) Nf[r * 4 + c] = M[r * 4 + 0] * S->Tf[0 * 4 + c] + M[r * 4 + 1] * S->Tf[1 * 4 + c] + M[r * 4 + 2] * S->Tf[2 * 4 + c] + M[r * 4 + 3] * S->Tf[3 * 4 + c];
  for (int k = 0; k < 16; ++k) { S->M[k] = M[k]; S->Tf[k] = Nf[k]; }
  S->apply = 1;
  const int it = ++S->iter;
  // DefaultConvergenceCriteria::hasConverged (absolute MSE 1e-12 is PCL's default; IterativeClosestPoint::computeTransformation sets the
  // rotation threshold to 1 - transformation_epsilon_ (PCL 1.8: setRotationThreshold(1.0 - transformation_epsilon_)), i.e. 0.999999 with laserMapping.cpp:673)
  bool conv = false;
  if (it >= P.icp_max_iters) conv = true;
  else {
    const double cos_angle = 0.5 * ((double)M[0] + (double)M[5] + (double)M[10] - 1.0);
    const double tr2 = (double)M[3] * M[3] + (double)M[7] * M[7] + (double)M[11] * M[11];
    if (cos_angle >= 1.0 - P.icp_trans_eps && tr2 <= P.icp_trans_eps) conv = true;
    else if (fabs(mse - S->prev_mse) < 1e-12) conv = true;
    else if (fabs(mse - S->prev_mse) / S->prev_mse < P.icp_fitness_eps) conv = true;
    else S->prev_mse = mse;
  }
  if (conv) { S->done = 1; S->converged = 1; }
}

__global__ void __launch_bounds__(ICP_T) icp_fitness(const IcpState* S, const float4* src, const float4* tgt, double* partial) {
  __shared__ float4 s_tile[ICP_TILE];
  __shared__ double s_red[ICP_T / 64][2];
  const int n_src = S->n_src, n_tgt = S->n_tgt;
  const int i = blockIdx.x * ICP_T + threadIdx.x;
  const float4 p0 = src[min(i, n_src - 1)];
  const float* T = S->Tf;
  float4 p;
  p.x = T[0] * p0.x + T[1] * p0.y + T[2] * p0.z + T[3]; p.y = T[4] * p0.x + T[5] * p0.y + T[6] * p0.z + T[7]; p.z = T[8] * p0.x + T[9] * p0.y + T[10] * p0.z + T[11]; p.w = p0.w;
  float d2;
  int idx;
  icp_nn(p, tgt, n_tgt, s_tile, &d2, &idx);
  double s = (i < n_src && idx >= 0) ? (double)d2 : 0.0, c = (i < n_src && idx >= 0) ? 1.0 : 0.0;
  s = wave_sum_f64(s); c = wave_sum_f64(c);
  if (lane_id() == 0) { s_red[threadIdx.x >> 6][0] = s; s_red[threadIdx.x >> 6][1] = c; }
  __syncthreads();
  if (threadIdx.x < 2) { double t = 0; for (int w = 0; w < ICP_T / 64; ++w) t += s_red[w][threadIdx.x]; partial[(size_t)blockIdx.x * 18 + threadIdx.x] = t; }
}
__global__ void icp_fitness_final(IcpState* S, const double* partial, int nwg) {
  if (threadIdx.x != 0) return;
  double s = 0, c = 0;
  for (int w = 0; w < nwg; ++w) { s += partial[(size_t)w * 18]; c += partial[(size_t)w * 18 + 1]; }
  S->fitness = c > 0 ? s / c : 1.7976931348623157e308;
}
__global__ void icp_init(IcpState* S, int n_src, const int* n_tgt) {
  if (threadIdx.x != 0) return;
  for (int k = 0; k < 16; ++k) { S->M[k] = (k % 5 == 0) ? 1.f : 0.f; S->Tf[k] = (k % 5 == 0) ? 1.f : 0.f; }
  S->prev_mse = 1.7976931348623157e308; S->fitness = 1.7976931348623157e308;
  S->iter = 0; S->done = (n_src == 0 || *n_tgt == 0) ? 1 : 0; S->converged = 0; S->apply = 0; S->n_src = n_src; S->n_tgt = *n_tgt;
}

// ---- host --------------------------------------------------------------------------------------------------------------------
#include "../../include/alego_mi355x.h"

namespace {
struct Temps {
  std::vector<void*> p;
  template <class T> hipError_t get(T** q, size_t bytes) { void* v = nullptr; hipError_t e = hipMalloc(&v, bytes ? bytes : 16); if (e == hipSuccess) p.push_back(v); *q = (T*)v; return e; }
  ~Temps() { for (void* v : p) (void)hipFree(v); }
};
}  // namespace

int icp_run(const alego_params& P, const alego_kf_in* latest, const alego_kf_in* history, int n_history, alego_icp_result* out,
            alego_point* target_out, int target_cap, hipStream_t st, std::string* err) {
  const int nf = 1 + n_history;
  std::vector<IcpFrame> F(nf);
  std::vector<alego_point> raw;
  int n_src = 0, n_traw = 0;
  for (int f = 0; f < nf; ++f) {
    const alego_kf_in& k = f == 0 ? *latest : history[f - 1];
    if (k.n_corner < 0 || k.n_surf < 0 || k.n_outlier < 0 || (k.n_corner && !k.corner) || (k.n_surf && !k.surf) || (k.n_outlier && !k.outlier)) { *err = "loop closure: null cloud / negative count"; return ALEGO_ERR_ARG; }
    for (int a = 0; a < 6; ++a) F[f].pose[a] = k.pose[a];
    F[f].off[0] = (int)raw.size(); raw.insert(raw.end(), k.corner, k.corner + k.n_corner);
    F[f].off[1] = (int)raw.size(); raw.insert(raw.end(), k.surf, k.surf + k.n_surf);
    F[f].off[2] = (int)raw.size(); raw.insert(raw.end(), k.outlier, k.outlier + k.n_outlier);
    F[f].off[3] = (int)raw.size();
    int& base = f == 0 ? n_src : n_traw;   // destination order: surf, corner, outlier (laserMapping.cpp:794-796,:805-807)
    F[f].dst[1] = base; base += k.n_surf;
    F[f].dst[0] = base; base += k.n_corner;
    F[f].dst[2] = base; base += k.n_outlier;
  }
  Temps T;
  IcpFrame* dF; float4 *draw, *dsrc, *dcur, *dtraw, *dtgt; int* dcnt; IcpState* dS; double* dpart;
  const int nwg = std::max(1, (n_src + ICP_T - 1) / ICP_T);
  hipError_t e = T.get(&dF, sizeof(IcpFrame) * nf);
  if (e == hipSuccess) e = T.get(&draw, raw.size() * 16);
  if (e == hipSuccess) e = T.get(&dsrc, (size_t)std::max(n_src, 1) * 16);
  if (e == hipSuccess) e = T.get(&dcur, (size_t)std::max(n_src, 1) * 16);
  if (e == hipSuccess) e = T.get(&dtraw, (size_t)std::max(n_traw, 1) * 16);
  if (e == hipSuccess) e = T.get(&dtgt, (size_t)std::max(n_traw, 1) * 16);
  if (e == hipSuccess) e = T.get(&dcnt, 8);
  if (e == hipSuccess) e = T.get(&dS, sizeof(IcpState));
  if (e == hipSuccess) e = T.get(&dpart, (size_t)nwg * 18 * 8);
  if (e == hipSuccess) e = hipMemcpyAsync(dF, F.data(), sizeof(IcpFrame) * nf, hipMemcpyHostToDevice, st);
  if (e == hipSuccess && !raw.empty()) e = hipMemcpyAsync(draw, raw.data(), raw.size() * 16, hipMemcpyHostToDevice, st);
  const int hc[2] = {n_traw, 0};
  if (e == hipSuccess) e = hipMemcpyAsync(dcnt, hc, 8, hipMemcpyHostToDevice, st);
  if (e != hipSuccess) { *err = std::string("loop closure: ") + hipGetErrorString(e); return ALEGO_ERR_HIP; }
  hipLaunchKernelGGL(icp_build, dim3(8, nf), dim3(ICP_T), 0, st, dF, nf, draw, dsrc, dtraw);
  VoxJob job{dtraw, dcnt, dtgt, dcnt + 1, nullptr, P.lc_leaf, std::max(n_traw, 1), std::max(n_traw, 1), nullptr, 0};
  VoxCtx V;
  if (vox_create(&V, &job, 1, err)) return ALEGO_ERR_HIP;
  int rc = vox_run(V, st, err);
  (void)hipMemcpyAsync(dcur, dsrc, (size_t)n_src * 16, hipMemcpyDeviceToDevice, st);
  hipLaunchKernelGGL(icp_init, dim3(1), dim3(64), 0, st, dS, n_src, dcnt + 1);
  const double max_d2 = P.icp_max_corr_dist * P.icp_max_corr_dist;
  for (int it = 0; it < P.icp_max_iters && rc == 0; ++it) {
    hipLaunchKernelGGL(icp_corr, dim3(nwg), dim3(ICP_T), 0, st, dS, dcur, dtgt, max_d2, dpart);
    hipLaunchKernelGGL(icp_step, dim3(1), dim3(64), 0, st, dS, dpart, nwg, P);
  }
  hipLaunchKernelGGL(icp_fitness, dim3(nwg), dim3(ICP_T), 0, st, dS, dsrc, dtgt, dpart);
  hipLaunchKernelGGL(icp_fitness_final, dim3(1), dim3(64), 0, st, dS, dpart, nwg);
  IcpState S;
  e = hipMemcpyAsync(&S, dS, sizeof(S), hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  if (e == hipSuccess && target_out && S.n_tgt > 0) e = hipMemcpy(target_out, dtgt, (size_t)std::min(S.n_tgt, target_cap) * 16, hipMemcpyDeviceToHost);
  vox_destroy(&V);
  if (e != hipSuccess || rc) { *err = std::string("loop closure: ") + (rc ? "VoxelGrid failed" : hipGetErrorString(e)); return ALEGO_ERR_HIP; }
  out->converged = S.converged; out->iterations = S.iter; out->fitness = S.n_src && S.n_tgt ? S.fitness : 1.7976931348623157e308;
  for (int k = 0; k < 16; ++k) out->correction[k] = S.Tf[k];
  out->n_source = S.n_src; out->n_target = S.n_tgt;
  return 0;
}
