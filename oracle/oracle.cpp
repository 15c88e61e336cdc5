// oracle.cpp — TEST INFRASTRUCTURE: CPU restatement of the A-LeGO-LOAM per-scan
// hot path (ImageProjection -> feature extraction + LaserOdometry -> LaserMapping
// scan-to-map registration).  It is the parity checker for the HIP path and the
// `cpu_baseline` leg of bench.py.  Only tests/, __graft_entry__.smoke() and
// bench.py's cpu_baseline may load it; the product library never does.
//
// PARITY UNPINNED: the reference has no tests, fixtures or golden vectors
//  and cannot be compiled here (ROS/PCL/FLANN/Eigen/Ceres/GTSAM are
// absent), so nothing but the reference source text pins this restatement.  The
// third-party pieces it re-implements from their published algorithms are:
//   pcl::VoxelGrid<PointXYZI>::applyFilter (PCL 1.8-era)          -> voxel_grid()
//   pcl::KdTreeFLANN / FLANN KDTreeSingleIndex, L2_Simple, exact  -> KdTree
//   ceres::Solve, TrustRegionMinimizer + LevenbergMarquardtStrategy + DENSE_QR +
//     HuberLoss corrector (Ceres 1.13/1.14 defaults)              -> ceres_like_solve()
//   Eigen AngleAxis/Quaternion composition, SelfAdjointEigenSolver<Matrix3d>,
//     colPivHouseholderQr 5x3 least squares                       -> quat_*, eig3(), colpiv_qr_solve()
//   GTSAM iSAM2 with a prior + chain of between factors and no loop closure
//     (estimate == initial values)                                -> key pose pass-through
// Where those libraries leave an order unspecified (unstable std::sort ties,
// kd-tree ties) the rule is: stable / lowest index first.
//
// Every function cites the reference file:line it follows (paths relative to the
// reference repository root).
#include <algorithm>
#include <cfloat>
#include <chrono>
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <condition_variable>
#include <deque>
#include <map>
#include <mutex>
#include <thread>
#include <queue>
#include <string>
#include <vector>

#include "../include/alego_params.h"
#include "oracle_math.h"
#include "oracle_icp.h"

namespace {

typedef alego_point Pt;
typedef alego_params Params;

// RAD2ANGLE / ANGLE2RAD, utility.h:47-48
inline double rad2angle(double x) { return x * 180.0 / M_PI; }

// ===========================================================================
// 1. ImageProjection  (src/imageProjection.cpp:49-316; options from src/IP.cpp)
// ===========================================================================
struct ImageProjection {
  Params P;
  int N = 0;
  // persistent members, imageProjection.cpp:16-35
  std::vector<Pt> full_cloud;          // full_cloud_
  std::vector<double> range_mat;       // range_mat_ (row-major here; Eigen col-major in the reference)
  std::vector<int> label_mat;          // label_mat_
  std::vector<uint8_t> ground_mat;     // ground_mat_
  std::vector<int> start_ring, end_ring;
  std::vector<uint8_t> seg_ground;     // segmentedCloudGroundFlag (length N, stale tail)
  std::vector<int> seg_col;            // segmentedCloudColInd
  std::vector<float> seg_range;        // segmentedCloudRange
  float ori[3] = {0, 0, 0};            // start/end/diff orientation
  std::vector<Pt> seg_cloud, outlier_cloud;
  int label_cnt = 1;
  // copies of the images taken before the end-of-callback reset (:197-205)
  std::vector<float> out_range_img;    // f32 range, -1 for empty
  std::vector<int> out_label_img;
  std::vector<uint8_t> out_ground_img;
  std::vector<int> out_owner_dummy;

  void init(const Params& p) {
    P = p;
    N = P.n_scan * P.horizon_scan;
    Pt nan_p{0, 0, 0, -1.0f};
    full_cloud.assign(N, nan_p);
    range_mat.assign(N, DBL_MAX);
    label_mat.assign(N, 0);
    ground_mat.assign(N, 0);
    start_ring.assign(P.n_scan, 0);
    end_ring.assign(P.n_scan, 0);
    seg_ground.assign(N, 0);
    seg_col.assign(N, 0);
    seg_range.assign(N, 0.f);
    label_cnt = 1;
  }

  // labelComponents, imageProjection.cpp:210-316
  void label_components(int row, int col) {
    const int H = P.horizon_scan, NS = P.n_scan;
    std::vector<uint8_t> line_cnt_flag(NS, 0);
    std::queue<int> que_i, que_j, all_i, all_j;
    que_i.push(row); que_j.push(col);
    line_cnt_flag[row] = 1;
    all_i.push(row); all_j.push(col);
    static const int nb[4][2] = {{-1, 0}, {1, 0}, {0, -1}, {0, 1}};  // :37-40
    while (!que_i.empty()) {
      int fi = que_i.front(), fj = que_j.front();
      que_i.pop(); que_j.pop();
      label_mat[fi * H + fj] = label_cnt;
      line_cnt_flag[fi] = 1;
      for (int k = 0; k < 4; ++k) {
        int ti = fi + nb[k][0], tj = fj + nb[k][1];
        if (ti < 0 || ti >= NS) continue;
        if (tj < 0) tj = H - 1; else if (tj >= H) tj = 0;
        if (label_mat[ti * H + tj]) continue;
        double d1 = std::max(range_mat[fi * H + fj], range_mat[ti * H + tj]);
        double d2 = std::min(range_mat[fi * H + fj], range_mat[ti * H + tj]);
        double alpha = (nb[k][0] == 0) ? P.seg_alpha_x : P.seg_alpha_y;
        double angle = std::atan2(d2 * std::sin(alpha), (d1 - d2 * std::cos(alpha)));
        if (angle > P.seg_theta) {
          que_i.push(ti); que_j.push(tj);
          label_mat[ti * H + tj] = label_cnt;
          line_cnt_flag[ti] = 1;
          all_i.push(ti); all_j.push(tj);
        }
      }
    }
    bool feasible = false;
    if ((int)all_i.size() >= P.seg_big_num) feasible = true;
    else if ((int)all_i.size() >= P.seg_valid_point_num) {
      int line_cnt = 0;
      for (int i = 0; i < NS; ++i) if (line_cnt_flag[i]) ++line_cnt;
      if (line_cnt >= P.seg_valid_line_num) feasible = true;
    }
    if (feasible) ++label_cnt;
    else while (!all_i.empty()) { label_mat[all_i.front() * H + all_j.front()] = 999999; all_i.pop(); all_j.pop(); }
  }

  // pcCB, imageProjection.cpp:49-208 (IP.cpp:106-304 for the near filter / RFANS table)
  void process(const Pt* in, int n_in) {
    const int H = P.horizon_scan, NS = P.n_scan;
    // a1: pcl::removeNaNFromPointCloud :58-59: filters non-finite points only when the message says is_dense == false,
    // otherwise it is a plain copy and the row test below rejects them; IP.cpp:77-104,117
    std::vector<Pt> cloud;
    cloud.reserve(n_in);
    for (int i = 0; i < n_in; ++i) {
      const Pt& p = in[i];
      if (!P.input_is_dense && (!std::isfinite(p.x) || !std::isfinite(p.y) || !std::isfinite(p.z))) continue;
      if (P.near_filter) {
        float th = (float)P.near_thres;
        if (p.x * p.x + p.y * p.y + p.z * p.z < th * th) continue;  // IP.cpp:91
      }
      cloud.push_back(p);
    }
    seg_cloud.clear(); outlier_cloud.clear();
    int cloud_size = (int)cloud.size();
    // a2: orientation :62-72 (float fields, double intermediates)
    if (cloud_size > 0) {
      float so = -omath::o_atan2f(cloud[0].y, cloud[0].x);
      float eo = (float)((double)(-omath::o_atan2f(cloud[cloud_size - 1].y, cloud[cloud_size - 1].x)) + 2 * M_PI);
      if ((double)(eo - so) > 3 * M_PI) eo = (float)((double)eo - 2 * M_PI);
      else if ((double)(eo - so) < M_PI) eo = (float)((double)eo + 2 * M_PI);
      ori[0] = so; ori[1] = eo; ori[2] = eo - so;
    }
    // a3: projection :76-104
    for (int i = 0; i < cloud_size; ++i) {
      Pt p = cloud[i];
      // is_dense == true with a non-finite point: NaN angles; `int(NaN)` is INT_MIN on x86-64 (cvttsd2si), so :81-85 / :93-97
      // reject it (an infinite coordinate gives a finite angle but an infinite range: the reference would keep such a cell;
      // treated like NaN here — no driver emits it)
      if (!std::isfinite(p.x) || !std::isfinite(p.y) || !std::isfinite(p.z)) continue;
      double vertical_ang = rad2angle((double)omath::o_atan2f(p.z, omath::o_hypotf(p.x, p.y)));
      int row_id;
      if (P.laser_type == ALEGO_LASER_UNIFORM) {
        row_id = (int)((vertical_ang + P.ang_bottom) / P.ang_res_y + 0.5);  // :80
      } else {  // IP.cpp:142-172
        if (vertical_ang > 4.5) row_id = (int)(13 + (vertical_ang - 5.) / 3 + 0.5);
        else if (vertical_ang > 0.5) row_id = (int)(11 + (vertical_ang - 1.0) / 2 + 0.5);
        else if (vertical_ang > -7.) row_id = (int)(10.5 + vertical_ang);
        else if (vertical_ang > -8.5) row_id = 3;
        else if (vertical_ang > -10.5) row_id = 2;
        else if (vertical_ang > -13.5) row_id = 1;
        else row_id = 0;
      }
      if (row_id < 0 || row_id >= NS) continue;
      double horizon_ang = rad2angle((double)(-omath::o_atan2f(p.y, p.x)) + 2 * M_PI);
      int col_id = (int)(horizon_ang / P.ang_res_x);
      if (col_id >= H) col_id -= H;
      if (col_id < 0 || col_id >= H) continue;
      range_mat[row_id * H + col_id] = (double)std::sqrt(p.x * p.x + p.y * p.y + p.z * p.z);  // sqrtf :99
      p.intensity = (float)(row_id + col_id / 10000.0);
      full_cloud[col_id + row_id * H] = p;
    }
    // a4: ground :107-132
    for (int j = 0; j < H; ++j) {
      for (int i = 0; i < P.ground_scan_id; ++i) {
        if (i + 1 >= NS) break;
        int lower_id = j + i * H, upper_id = j + (i + 1) * H;
        if (-1 == full_cloud[lower_id].intensity || -1 == full_cloud[upper_id].intensity) continue;
        double dx = full_cloud[upper_id].x - full_cloud[lower_id].x;  // f32 subtraction, widened
        double dy = full_cloud[upper_id].y - full_cloud[lower_id].y;
        double dz = full_cloud[upper_id].z - full_cloud[lower_id].z;
        double angle = rad2angle(std::atan2(dz, std::hypot(dx, dy)));
        if (std::abs(angle - P.sensor_mount_ang) < P.ground_angle_thres)
          ground_mat[i * H + j] = ground_mat[(i + 1) * H + j] = 1;
      }
    }
    // label init :134-143
    for (int i = 0; i < NS; ++i)
      for (int j = 0; j < H; ++j)
        if (ground_mat[i * H + j] == 1 || range_mat[i * H + j] == DBL_MAX) label_mat[i * H + j] = -1;
    // a5: segmentation :147-156
    for (int i = 0; i < NS; ++i)
      for (int j = 0; j < H; ++j)
        if (label_mat[i * H + j] == 0) label_components(i, j);
    // a6: compaction :158-191
    int line_size = 0;
    for (int i = 0; i < NS; ++i) {
      start_ring[i] = line_size + 5;
      for (int j = 0; j < H; ++j) {
        if (label_mat[i * H + j] > 0 || ground_mat[i * H + j] == 1) {
          if (label_mat[i * H + j] == 999999) {
            if (i > P.ground_scan_id && j % 5 == 0) outlier_cloud.push_back(full_cloud[j + i * H]);
            continue;
          } else if (ground_mat[i * H + j] == 1) {
            if (j % 5 != 0 && j > 4 && j < H - 5) continue;
          }
          seg_ground[line_size] = (ground_mat[i * H + j] == 1);
          seg_col[line_size] = j;
          seg_range[line_size] = (float)range_mat[i * H + j];
          seg_cloud.push_back(full_cloud[j + i * H]);
          ++line_size;
        }
      }
      end_ring[i] = line_size - 1 - 5;
    }
    // snapshot of the images for parity checks, then the reset of :197-205
    out_range_img.resize(N); out_label_img = label_mat; out_ground_img = ground_mat;
    for (int c = 0; c < N; ++c) out_range_img[c] = (range_mat[c] == DBL_MAX) ? -1.0f : (float)range_mat[c];
    std::fill(range_mat.begin(), range_mat.end(), DBL_MAX);
    std::fill(label_mat.begin(), label_mat.end(), 0);
    std::fill(ground_mat.begin(), ground_mat.end(), 0);
    label_cnt = 1;
    Pt nan_p{0, 0, 0, -1.0f};
    std::fill(full_cloud.begin(), full_cloud.end(), nan_p);
  }
};

// ===========================================================================
// 2. pcl::VoxelGrid<PointXYZI>::applyFilter  [upstream, SURVEY.md B.1]
//    call sites: laserOdometry.cpp:289-292, laserMapping.cpp:316-319,329-342
// ===========================================================================
void voxel_grid(const std::vector<Pt>& in, float leaf, std::vector<Pt>& out, int sort_mode) {
  out.clear();
  if (in.empty()) return;
  const float inv = 1.0f / leaf;
  float minp[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, maxp[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  for (const Pt& p : in) {  // getMinMax3D
    minp[0] = std::min(minp[0], p.x); minp[1] = std::min(minp[1], p.y); minp[2] = std::min(minp[2], p.z);
    maxp[0] = std::max(maxp[0], p.x); maxp[1] = std::max(maxp[1], p.y); maxp[2] = std::max(maxp[2], p.z);
  }
  int64_t dx = (int64_t)((maxp[0] - minp[0]) * inv) + 1;
  int64_t dy = (int64_t)((maxp[1] - minp[1]) * inv) + 1;
  int64_t dz = (int64_t)((maxp[2] - minp[2]) * inv) + 1;
  if (dx * dy * dz > (int64_t)INT_MAX) { out = in; return; }  // "leaf size too small": input returned
  int minb[3], maxb[3], divb[3];
  for (int a = 0; a < 3; ++a) {
    minb[a] = (int)std::floor(minp[a] * inv);
    maxb[a] = (int)std::floor(maxp[a] * inv);
    divb[a] = maxb[a] - minb[a] + 1;
  }
  const int mul[3] = {1, divb[0], divb[0] * divb[1]};
  struct Key { unsigned idx; int pt; };
  std::vector<Key> keys(in.size());
  for (size_t i = 0; i < in.size(); ++i) {
    int ijk0 = (int)(std::floor(in[i].x * inv) - (float)minb[0]);
    int ijk1 = (int)(std::floor(in[i].y * inv) - (float)minb[1]);
    int ijk2 = (int)(std::floor(in[i].z * inv) - (float)minb[2]);
    keys[i] = Key{(unsigned)(ijk0 * mul[0] + ijk1 * mul[1] + ijk2 * mul[2]), (int)i};
  }
  if (sort_mode == 1) std::sort(keys.begin(), keys.end(), [](const Key& a, const Key& b) { return a.idx < b.idx; });
  else std::stable_sort(keys.begin(), keys.end(), [](const Key& a, const Key& b) { return a.idx < b.idx; });
  size_t index = 0;
  while (index < keys.size()) {
    size_t i = index + 1;
    while (i < keys.size() && keys[i].idx == keys[index].idx) ++i;
    // pcl::CentroidPoint<PointXYZI>: f32 accumulators, divided by the count
    float sx = 0, sy = 0, sz = 0, si = 0;
    for (size_t li = index; li < i; ++li) {
      const Pt& p = in[keys[li].pt];
      sx += p.x; sy += p.y; sz += p.z; si += p.intensity;
    }
    float n = (float)(i - index);
    out.push_back(Pt{sx / n, sy / n, sz / n, si / n});
    index = i;
  }
}

// ===========================================================================
// 3. Exact k-NN (pcl::KdTreeFLANN::nearestKSearch -> FLANN KDTreeSingleIndex,
//    L2_Simple<float>, leaf 15, eps 0)  [upstream, SURVEY.md B.2]
// ===========================================================================
inline float dist2_f32(const Pt& a, const Pt& b) {  // flann::L2_Simple<float>
  float r = 0.f, d;
  d = a.x - b.x; r += d * d;
  d = a.y - b.y; r += d * d;
  d = a.z - b.z; r += d * d;
  return r;
}

struct KdTree {
  struct Node { int lo, hi, left, right; float bmin[3], bmax[3]; };
  const Pt* pts = nullptr;
  int n = 0;
  std::vector<int> order;
  std::vector<Node> nodes;
  std::vector<Pt> store;

  void build(const std::vector<Pt>& cloud) {
    store = cloud; pts = store.data(); n = (int)store.size();
    order.resize(n);
    for (int i = 0; i < n; ++i) order[i] = i;
    nodes.clear();
    if (n > 0) { nodes.reserve(2 * n / 8 + 8); build_rec(0, n); }
  }
  int build_rec(int lo, int hi) {
    Node nd; nd.lo = lo; nd.hi = hi; nd.left = nd.right = -1;
    for (int a = 0; a < 3; ++a) { nd.bmin[a] = FLT_MAX; nd.bmax[a] = -FLT_MAX; }
    for (int i = lo; i < hi; ++i) {
      const Pt& p = pts[order[i]];
      const float v[3] = {p.x, p.y, p.z};
      for (int a = 0; a < 3; ++a) { nd.bmin[a] = std::min(nd.bmin[a], v[a]); nd.bmax[a] = std::max(nd.bmax[a], v[a]); }
    }
    int id = (int)nodes.size();
    nodes.push_back(nd);
    if (hi - lo > 15) {
      int dim = 0; float ext = nd.bmax[0] - nd.bmin[0];
      for (int a = 1; a < 3; ++a) if (nd.bmax[a] - nd.bmin[a] > ext) { ext = nd.bmax[a] - nd.bmin[a]; dim = a; }
      int mid = (lo + hi) / 2;
      std::nth_element(order.begin() + lo, order.begin() + mid, order.begin() + hi, [&](int a, int b) {
        const float va = dim == 0 ? pts[a].x : dim == 1 ? pts[a].y : pts[a].z;
        const float vb = dim == 0 ? pts[b].x : dim == 1 ? pts[b].y : pts[b].z;
        return va < vb || (va == vb && a < b);
      });
      int l = build_rec(lo, mid), r = build_rec(mid, hi);
      nodes[id].left = l; nodes[id].right = r;
    }
    return id;
  }
  struct Cand { float d; int i; bool operator<(const Cand& o) const { return d < o.d || (d == o.d && i < o.i); } };
  double box_dist(const Node& nd, const Pt& q) const {
    const float v[3] = {q.x, q.y, q.z};
    double s = 0;
    for (int a = 0; a < 3; ++a) {
      double d = 0;
      if (v[a] < nd.bmin[a]) d = (double)nd.bmin[a] - v[a]; else if (v[a] > nd.bmax[a]) d = (double)v[a] - nd.bmax[a];
      s += d * d;
    }
    return s;
  }
  // result set: ascending (distance, index), at most k entries, in a caller-provided array (as FLANN's KNNResultSet)
  struct ResultSet { Cand* c; int n, k; };
  void search_rec(int id, const Pt& q, ResultSet& best) const {
    const Node& nd = nodes[id];
    if (best.n == best.k && box_dist(nd, q) * (1.0 - 1e-5) > (double)best.c[best.n - 1].d) return;
    if (nd.left < 0) {
      for (int i = nd.lo; i < nd.hi; ++i) {
        Cand c{dist2_f32(pts[order[i]], q), order[i]};
        if (best.n < best.k || c < best.c[best.n - 1]) {
          int pos = best.n < best.k ? best.n : best.n - 1;   // insert, dropping the last entry of a full set
          while (pos > 0 && c < best.c[pos - 1]) { best.c[pos] = best.c[pos - 1]; --pos; }
          best.c[pos] = c;
          if (best.n < best.k) ++best.n;
        }
      }
      return;
    }
    double dl = box_dist(nodes[nd.left], q), dr = box_dist(nodes[nd.right], q);
    if (dl <= dr) { search_rec(nd.left, q, best); search_rec(nd.right, q, best); }
    else { search_rec(nd.right, q, best); search_rec(nd.left, q, best); }
  }
  // returns number found (min(k, n)); ascending (distance, index)
  int knn(const Pt& q, int k, int* idx, float* dist) const {
    Cand small[16];
    std::vector<Cand> big;
    if (k > 16) big.resize(k);
    ResultSet best{k > 16 ? big.data() : small, 0, k};
    if (n > 0 && k > 0) search_rec(0, q, best);
    for (int i = 0; i < best.n; ++i) { idx[i] = best.c[i].i; dist[i] = best.c[i].d; }
    return best.n;
  }
};

// ===========================================================================
// 4. Eigen pieces  [upstream, SURVEY.md B.4]
// ===========================================================================
struct Quat { double w, x, y, z; };
inline Quat quat_axis(double angle, int axis) {  // Quaternion(AngleAxisd(angle, Unit<axis>))
  double ha = 0.5 * angle, s = std::sin(ha);
  Quat q{std::cos(ha), 0, 0, 0};
  if (axis == 0) q.x = s; else if (axis == 1) q.y = s; else q.z = s;
  return q;
}
inline Quat quat_mul(const Quat& a, const Quat& b) {
  return Quat{a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z,
              a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
              a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z,
              a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x};
}
inline Quat quat_zyx(double yaw, double pitch, double roll) {  // AngleAxis(Z)*AngleAxis(Y)*AngleAxis(X)
  return quat_mul(quat_mul(quat_axis(yaw, 2), quat_axis(pitch, 1)), quat_axis(roll, 0));
}
inline void quat_rotate(const Quat& q, const double v[3], double out[3]) {  // QuaternionBase::_transformVector
  double uv[3] = {2 * (q.y * v[2] - q.z * v[1]), 2 * (q.z * v[0] - q.x * v[2]), 2 * (q.x * v[1] - q.y * v[0])};
  out[0] = v[0] + q.w * uv[0] + (q.y * uv[2] - q.z * uv[1]);
  out[1] = v[1] + q.w * uv[1] + (q.z * uv[0] - q.x * uv[2]);
  out[2] = v[2] + q.w * uv[2] + (q.x * uv[1] - q.y * uv[0]);
}
inline void quat_to_mat(const Quat& q, double R[9]) {  // QuaternionBase::toRotationMatrix
  double tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z;
  double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
  double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x, tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
  R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
  R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}
inline Quat mat_to_quat(const double m[9]) {  // Quaternion(Matrix3d)
  Quat q;
  double t = m[0] + m[4] + m[8];
  if (t > 0) {
    t = std::sqrt(t + 1.0); q.w = 0.5 * t; t = 0.5 / t;
    q.x = (m[7] - m[5]) * t; q.y = (m[2] - m[6]) * t; q.z = (m[3] - m[1]) * t;
  } else {
    int i = 0;
    if (m[4] > m[0]) i = 1;
    if (m[8] > m[i * 3 + i]) i = 2;
    int j = (i + 1) % 3, k = (j + 1) % 3;
    t = std::sqrt(m[i * 3 + i] - m[j * 3 + j] - m[k * 3 + k] + 1.0);
    double v[3];
    v[i] = 0.5 * t; t = 0.5 / t;
    q.w = (m[k * 3 + j] - m[j * 3 + k]) * t;
    v[j] = (m[j * 3 + i] + m[i * 3 + j]) * t;
    v[k] = (m[k * 3 + i] + m[i * 3 + k]) * t;
    q.x = v[0]; q.y = v[1]; q.z = v[2];
  }
  return q;
}
inline Quat quat_inverse(const Quat& q) {
  double n2 = q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z;
  return Quat{q.w / n2, -q.x / n2, -q.y / n2, -q.z / n2};
}

// Householder QR least squares: min ||A x - b||, A is m x n column-major (overwritten), v = scratch of m doubles.
void qr_solve(double* A, double* b, int m, int n, double* x, double* v) {
  for (int k = 0; k < n; ++k) {
    double* col = A + (size_t)k * m;
    double norm2 = 0;
    for (int i = k; i < m; ++i) norm2 += col[i] * col[i];
    double norm = std::sqrt(norm2);
    if (norm == 0.0) continue;
    double alpha = col[k] > 0 ? -norm : norm;
    for (int i = k; i < m; ++i) v[i - k] = col[i];
    v[0] -= alpha;
    double vn2 = 0;
    for (int i = 0; i < m - k; ++i) vn2 += v[i] * v[i];
    if (vn2 == 0.0) continue;
    for (int j = k; j < n; ++j) {
      double* cj = A + (size_t)j * m;
      double dot = 0;
      for (int i = k; i < m; ++i) dot += v[i - k] * cj[i];
      double f = 2.0 * dot / vn2;
      for (int i = k; i < m; ++i) cj[i] -= f * v[i - k];
    }
    double dot = 0;
    for (int i = k; i < m; ++i) dot += v[i - k] * b[i];
    double f = 2.0 * dot / vn2;
    for (int i = k; i < m; ++i) b[i] -= f * v[i - k];
  }
  for (int k = n - 1; k >= 0; --k) {
    double s = b[k];
    for (int j = k + 1; j < n; ++j) s -= A[(size_t)j * m + k] * x[j];
    x[k] = s / A[(size_t)k * m + k];
  }
}
// Eigen::ColPivHouseholderQR<Matrix<double, M, N>>::compute() + solve() (Eigen 3.3.x: ColPivHouseholderQR.h computeInPlace / _solve_impl, Householder.h
// makeHouseholder / applyHouseholderOnTheLeft) — what `matA0.colPivHouseholderQr().solve(matB0)` of laserMapping.cpp:435 runs.  Restated from the published
// algorithm (Eigen is not in this image; SURVEY.md B: 3.3.4 is an era guess, 3.3.0 - 3.4.0 share this code):
//   * column norms up front; at step k the remaining column of largest (down-dated) norm is swapped into place;
//   * nonzero_pivots = first k whose pivot column has  norm^2 < (max initial norm * eps)^2 / rows * (rows - k)   ("Track the number of meaningful pivots", bug 941);
//   * Householder reflector with beta = -sign(c0) * ||col||, essential = tail / (c0 - beta), tau = (beta - c0) / beta; nothing to do when the tail is <= DBL_MIN;
//   * LAPACK xGEQP3's norm down-date (LAWN 176) with recomputation when the estimate has lost half its digits;
//   * solve: c = H_{r-1} ... H_0 b with r = nonzero_pivots, back-substitution on the leading r x r triangle, x[perm[i]] = c[i] for i < r and **0 for i >= r**
//     (a rank-deficient neighbourhood — collinear or coincident map points — gets a FINITE basic solution, not a division by a ~1e-17 pivot).
// Note solve() uses nonzeroPivots() (the tiny threshold above), not rank() / setThreshold(): that is Eigen's code, and what this follows.
// Sums run in index order (Eigen's unrolled / SSE reductions of these 3- to 5-element vectors may associate differently: last-bit differences, no test can pin
// them without Eigen).  A is m x n column-major and is overwritten by the factorisation; m <= 8, n <= 8.
int colpiv_qr_solve(double* A, const double* b_in, int m, int n, double* x) {
  const double eps = 2.220446049250313e-16, tiny = 2.2250738585072014e-308;
  const int size = m < n ? m : n;
  double hc[8], nu[8], nd[8], c[8];
  int trans[8], perm[8];
  double maxn = 0;
  for (int k = 0; k < n; ++k) {
    double s2 = 0;
    for (int i = 0; i < m; ++i) s2 += A[k * m + i] * A[k * m + i];
    nd[k] = nu[k] = std::sqrt(s2);
    if (nu[k] > maxn) maxn = nu[k];
  }
  const double threshold_helper = (maxn * eps) * (maxn * eps) / (double)m;
  const double downdate_threshold = std::sqrt(eps);
  int nonzero = size;
  for (int k = 0; k < size; ++k) {
    int big = k;
    for (int j = k + 1; j < n; ++j) if (nu[j] > nu[big]) big = j;            // maxCoeff: the first of equal maxima
    const double big_sq = nu[big] * nu[big];
    if (nonzero == size && big_sq < threshold_helper * (double)(m - k)) nonzero = k;
    trans[k] = big;
    if (big != k) {
      for (int i = 0; i < m; ++i) std::swap(A[k * m + i], A[big * m + i]);
      std::swap(nu[k], nu[big]); std::swap(nd[k], nd[big]);
    }
    double* col = A + k * m;
    double tail2 = 0;
    for (int i = k + 1; i < m; ++i) tail2 += col[i] * col[i];
    const double c0 = col[k];
    double beta, tau;
    if (tail2 <= tiny) { tau = 0; beta = c0; for (int i = k + 1; i < m; ++i) col[i] = 0; }
    else {
      beta = std::sqrt(c0 * c0 + tail2);
      if (c0 >= 0) beta = -beta;
      for (int i = k + 1; i < m; ++i) col[i] = col[i] / (c0 - beta);
      tau = (beta - c0) / beta;
    }
    hc[k] = tau;
    col[k] = beta;
    if (m - k == 1) { for (int j = k + 1; j < n; ++j) A[j * m + k] *= 1.0 - tau; }
    else if (tau != 0) {
      for (int j = k + 1; j < n; ++j) {
        double* cj = A + j * m;
        double t = 0;
        for (int i = k + 1; i < m; ++i) t += col[i] * cj[i];
        t += cj[k];
        cj[k] -= tau * t;
        for (int i = k + 1; i < m; ++i) cj[i] -= tau * col[i] * t;
      }
    }
    for (int j = k + 1; j < n; ++j) {
      if (nu[j] != 0) {
        double t = std::fabs(A[j * m + k]) / nu[j];
        t = (1.0 + t) * (1.0 - t);
        t = t < 0 ? 0 : t;
        const double q = nu[j] / nd[j];
        const double t2 = t * (q * q);
        if (t2 <= downdate_threshold) {
          double s2 = 0;
          for (int i = k + 1; i < m; ++i) s2 += A[j * m + i] * A[j * m + i];
          nd[j] = nu[j] = std::sqrt(s2);
        } else nu[j] *= std::sqrt(t);
      }
    }
  }
  for (int k = 0; k < n; ++k) perm[k] = k;
  for (int k = 0; k < size; ++k) std::swap(perm[k], perm[trans[k]]);
  if (nonzero == 0) { for (int k = 0; k < n; ++k) x[k] = 0; return 0; }
  for (int i = 0; i < m; ++i) c[i] = b_in[i];
  for (int k = 0; k < nonzero; ++k) {       // (H_0 H_1 ...)^T b : H_0 first
    const double tau = hc[k];
    const double* col = A + k * m;
    if (m - k == 1) c[k] *= 1.0 - tau;
    else if (tau != 0) {
      double t = 0;
      for (int i = k + 1; i < m; ++i) t += col[i] * c[i];
      t += c[k];
      c[k] -= tau * t;
      for (int i = k + 1; i < m; ++i) c[i] -= tau * col[i] * t;
    }
  }
  for (int k = nonzero - 1; k >= 0; --k) {  // triangularView<Upper>().solveInPlace
    double sum = c[k];
    for (int j = k + 1; j < nonzero; ++j) sum -= A[j * m + k] * c[j];
    c[k] = sum / A[k * m + k];
  }
  for (int i = 0; i < nonzero; ++i) x[perm[i]] = c[i];
  for (int i = nonzero; i < n; ++i) x[perm[i]] = 0;
  return nonzero;
}
void qr_solve(std::vector<double>& A, std::vector<double>& b, int m, int n, double* x) {
  std::vector<double> v(m);
  qr_solve(A.data(), b.data(), m, n, x, v.data());
}

// Symmetric 3x3 eigen-decomposition (cyclic Jacobi), eigenvalues ascending,
// V column c = eigenvector c  (stands in for SelfAdjointEigenSolver<Matrix3d>).
void eig3(const double Ain[9], double lam[3], double V[9]) {
  double A[9];
  std::memcpy(A, Ain, sizeof(A));
  for (int i = 0; i < 9; ++i) V[i] = (i % 4 == 0) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 60; ++sweep) {
    double off = A[1] * A[1] + A[2] * A[2] + A[5] * A[5];
    if (off < 1e-300) break;
    // converged to working precision: the off-diagonal mass is below 1e-40 of the diagonal's (a further rotation moves no entry by more than 1e-20
    // of the largest; Eigen's own solver stops on a relative test as well).  The device's d_eig3 has the same test in the same place.
    if (off <= 1e-40 * (A[0] * A[0] + A[4] * A[4] + A[8] * A[8])) break;
    for (int p = 0; p < 2; ++p)
      for (int q = p + 1; q < 3; ++q) {
        double apq = A[p * 3 + q];
        if (apq == 0.0) continue;
        double theta = (A[q * 3 + q] - A[p * 3 + p]) / (2.0 * apq);
        double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < 3; ++k) {
          double akp = A[k * 3 + p], akq = A[k * 3 + q];
          A[k * 3 + p] = c * akp - s * akq; A[k * 3 + q] = s * akp + c * akq;
        }
        for (int k = 0; k < 3; ++k) {
          double apk = A[p * 3 + k], aqk = A[q * 3 + k];
          A[p * 3 + k] = c * apk - s * aqk; A[q * 3 + k] = s * apk + c * aqk;
        }
        for (int k = 0; k < 3; ++k) {
          double vkp = V[k * 3 + p], vkq = V[k * 3 + q];
          V[k * 3 + p] = c * vkp - s * vkq; V[k * 3 + q] = s * vkp + c * vkq;
        }
      }
  }
  int ord[3] = {0, 1, 2};
  double d[3] = {A[0], A[4], A[8]};
  std::sort(ord, ord + 3, [&](int a, int b) { return d[a] < d[b]; });
  double Vs[9];
  for (int c = 0; c < 3; ++c) { lam[c] = d[ord[c]]; for (int k = 0; k < 3; ++k) Vs[k * 3 + c] = V[k * 3 + ord[c]]; }
  std::memcpy(V, Vs, sizeof(Vs));
}

// ===========================================================================
// 5. Cost functors (include/alego/utility.h:122-349) and the Ceres restatement
// ===========================================================================
enum BlockType { BLK_SURF = 0, BLK_CORNER = 1, BLK_EDGE = 2, BLK_PLANE = 3 };
struct Block {
  int type;
  double cp[3];
  double a[3];  // lpj  | plane unit normal
  double b[3];  // lpl
  double c[3];  // lpm
  double d;     // negative_OA_dot_norm
};

// Evaluate(): residual and (uncorrected) 1x6 Jacobian.
void eval_block(const Block& B, const double* p, double* res, double* J) {
  Quat q = quat_zyx(p[5], p[4], p[3]);
  double lp[3];
  quat_rotate(q, B.cp, lp);
  lp[0] += p[0]; lp[1] += p[1]; lp[2] += p[2];
  const double X = B.cp[0], Y = B.cp[1], Z = B.cp[2];
  double sr = std::sin(p[3]), cr = std::cos(p[3]), sp = std::sin(p[4]), cp = std::cos(p[4]);
  double sy = std::sin(p[5]), cy = std::cos(p[5]);
  // shared derivative table utility.h:148-158 (dy_dp keeps the reference's `cr*sr*cp` term)
  double dx_dr = (cy * sp * cr + sr * sy) * Y + (sy * cr - cy * sr * sp) * Z;
  double dy_dr = (-cy * sr + sy * sp * cr) * Y + (-sr * sy * sp - cy * cr) * Z;
  double dz_dr = cp * cr * Y - cp * sr * Z;
  double dx_dp = -cy * sp * X + cy * cp * sr * Y + cy * cr * cp * Z;
  double dy_dp = -sp * sy * X + sy * cp * sr * Y + cr * sr * cp * Z;
  double dz_dp = -cp * X - sp * sr * Y - sp * cr * Z;
  double dx_dy = -sy * cp * X - (sy * sp * sr + cr * cy) * Y + (cy * sr - sy * cr * sp) * Z;
  double dy_dy = cp * cy * X + (-sy * cr + cy * sp * sr) * Y + (cy * cr * sp + sy * sr) * Z;
  double dz_dy = 0.;
  if (B.type == BLK_CORNER || B.type == BLK_EDGE) {  // utility.h:126-174, :246-294
    const double* j = B.a; const double* l = B.b;
    double k = std::sqrt(std::pow(j[0] - l[0], 2) + std::pow(j[1] - l[1], 2) + std::pow(j[2] - l[2], 2));
    double a = (lp[1] - j[1]) * (lp[2] - l[2]) - (lp[2] - j[2]) * (lp[1] - l[1]);
    double b = (lp[2] - j[2]) * (lp[0] - l[0]) - (lp[0] - j[0]) * (lp[2] - l[2]);
    double c = (lp[0] - j[0]) * (lp[1] - l[1]) - (lp[1] - j[1]) * (lp[0] - l[0]);
    double m = std::sqrt(a * a + b * b + c * c);
    *res = m / k;
    if (!J) return;
    double dm_dx = (b * (l[2] - j[2]) + c * (j[1] - l[1])) / m;
    double dm_dy = (a * (j[2] - l[2]) - c * (j[0] - l[0])) / m;
    double dm_dz = (-a * (j[1] - l[1]) + b * (j[0] - l[0])) / m;
    if (B.type == BLK_CORNER) {
      J[0] = dm_dx / k; J[1] = dm_dy / k; J[2] = 0.; J[3] = 0.; J[4] = 0.;
      J[5] = (dm_dx * dx_dy + dm_dy * dy_dy + dm_dz * dz_dy) / k;
    } else {
      J[0] = dm_dx / k; J[1] = dm_dy / k; J[2] = dm_dz / k;
      J[3] = (dm_dx * dx_dr + dm_dy * dy_dr + dm_dz * dz_dr) / k;
      J[4] = (dm_dx * dx_dp + dm_dy * dy_dp + dm_dz * dz_dp) / k;
      J[5] = (dm_dx * dx_dy + dm_dy * dy_dy + dm_dz * dz_dy) / k;
    }
  } else if (B.type == BLK_SURF) {  // utility.h:185-235
    const double* j = B.a; const double* l = B.b; const double* mm = B.c;
    double a = (j[1] - l[1]) * (j[2] - mm[2]) - (j[2] - l[2]) * (j[1] - mm[1]);
    double b = (j[2] - l[2]) * (j[0] - mm[0]) - (j[0] - l[0]) * (j[2] - mm[2]);
    double c = (j[0] - l[0]) * (j[1] - mm[1]) - (j[1] - l[1]) * (j[0] - mm[0]);
    a *= a; b *= b; c *= c;
    double m = std::sqrt(std::pow((lp[0] - j[0]), 2) * a + std::pow((lp[1] - j[1]), 2) * b + std::pow((lp[2] - j[2]), 2) * c);
    double k = std::sqrt(a + b + c);
    *res = m / k;
    if (!J) return;
    double tmp = m * k;
    double dm_dz = ((lp[2] - j[2]) * c) / tmp;
    J[0] = 0.; J[1] = 0.; J[2] = dm_dz / k; J[3] = 0.; J[4] = 0.; J[5] = 0.;
  } else {  // BLK_PLANE utility.h:307-343
    const double* n = B.a;
    *res = (n[0] * lp[0] + n[1] * lp[1] + n[2] * lp[2]) + B.d;
    if (!J) return;
    J[0] = n[0]; J[1] = n[1]; J[2] = n[2];
    J[3] = n[0] * dx_dr + n[1] * dy_dr + n[2] * dz_dr;
    J[4] = n[0] * dx_dp + n[1] * dy_dp + n[2] * dz_dp;
    J[5] = n[0] * dx_dy + n[1] * dy_dy + n[2] * dz_dy;
  }
}

struct SolveSummary {
  int iterations = 0, successful = 0, termination = 0;  // 0 max-iter, 1 gradient, 2 parameter, 3 function, 4 failure, 5 radius
  double initial_cost = 0, final_cost = 0;
};

// ceres::Solve with options {DENSE_QR, max_num_iterations}, HuberLoss(a) on every
// block, one 6-parameter block  [upstream, SURVEY.md B.3].
// call sites: laserOdometry.cpp:413-418,487-492; laserMapping.cpp:468-475.
struct CeresLike {
  const std::vector<Block>* blocks;
  double huber_a;
  int R;
  std::vector<double> residuals, jac;  // jac row-major R x 6
  double gradient[6];

  // Evaluator::Evaluate + ResidualBlock::Evaluate + Corrector (rho'' <= 0 path)
  bool evaluate(const double* x, double* cost, bool want_jac) {
    double c = 0;
    if (want_jac) for (int k = 0; k < 6; ++k) gradient[k] = 0;
    const double b2 = huber_a * huber_a;
    for (int i = 0; i < R; ++i) {
      double r, J[6];
      eval_block((*blocks)[i], x, &r, want_jac ? J : nullptr);
      double s = r * r, rho0, rho1;
      if (s > b2) { double rr = std::sqrt(s); rho0 = 2.0 * huber_a * rr - b2; rho1 = std::max(DBL_MIN, huber_a / rr); }
      else { rho0 = s; rho1 = 1.0; }
      c += 0.5 * rho0;
      if (want_jac) {
        double sq = std::sqrt(rho1);
        for (int k = 0; k < 6; ++k) J[k] *= sq;
        r *= sq;
        residuals[i] = r;
        for (int k = 0; k < 6; ++k) { jac[(size_t)i * 6 + k] = J[k]; gradient[k] += J[k] * r; }
      }
    }
    *cost = c;
    return std::isfinite(c);
  }

  SolveSummary solve(const std::vector<Block>& blk, double* params, int max_iter, double a) {
    SolveSummary sum;
    blocks = &blk; huber_a = a; R = (int)blk.size();
    residuals.assign(R, 0); jac.assign((size_t)R * 6, 0);
    double x[6], x_cost, x_norm = 0, scale[6], diagonal[6];
    std::memcpy(x, params, sizeof(x));
    for (int k = 0; k < 6; ++k) x_norm += x[k] * x[k];
    x_norm = std::sqrt(x_norm);
    if (R == 0 || !evaluate(x, &x_cost, true)) { sum.termination = 4; return sum; }
    sum.initial_cost = sum.final_cost = x_cost;
    // jacobi scaling (iteration 0 only)
    for (int k = 0; k < 6; ++k) {
      double s = 0;
      for (int i = 0; i < R; ++i) s += jac[(size_t)i * 6 + k] * jac[(size_t)i * 6 + k];
      scale[k] = 1.0 / (1.0 + std::sqrt(s));
    }
    auto scale_jac = [&]() { for (int i = 0; i < R; ++i) for (int k = 0; k < 6; ++k) jac[(size_t)i * 6 + k] *= scale[k]; };
    auto grad_max = [&]() { double g = 0; for (int k = 0; k < 6; ++k) g = std::max(g, std::fabs(x[k] - (x[k] - gradient[k]))); return g; };
    scale_jac();
    double gmax = grad_max();
    double radius = 1e4, decrease_factor = 2.0;
    bool reuse_diagonal = false, step_successful = true;
    int iter = 0, num_invalid = 0;
    while (true) {
      if (iter >= max_iter) { sum.termination = 0; break; }
      if (step_successful && gmax <= 1e-10) { sum.termination = 1; break; }
      if (radius <= 1e-32) { sum.termination = 5; break; }
      ++iter;
      // LevenbergMarquardtStrategy::ComputeStep
      if (!reuse_diagonal) {
        for (int k = 0; k < 6; ++k) {
          double s = 0;
          for (int i = 0; i < R; ++i) s += jac[(size_t)i * 6 + k] * jac[(size_t)i * 6 + k];
          diagonal[k] = std::min(std::max(s, 1e-6), 1e32);
        }
      }
      const int m = R + 6;
      std::vector<double> A((size_t)m * 6, 0.0), rhs(m, 0.0);
      for (int i = 0; i < R; ++i) { rhs[i] = residuals[i]; for (int k = 0; k < 6; ++k) A[(size_t)k * m + i] = jac[(size_t)i * 6 + k]; }
      for (int k = 0; k < 6; ++k) A[(size_t)k * m + R + k] = std::sqrt(diagonal[k] / radius);
      double step[6];
      // ORACLE_SOLVER=normal (diagnostic, tests/diagnostics/solver_family.py): the same damped least-squares step through the normal
      // equations + Cholesky (what the device's lm_propose does) instead of Householder QR of the stacked system (Ceres DENSE_QR).
      // Mathematically the same step; the two differ by rounding only — the experiment asks how far that alone moves a free run.
      static const int normal_eq = [] { const char* e = std::getenv("ORACLE_SOLVER"); return !e ? 0 : std::strcmp(e, "normal") == 0 ? 1 : std::strcmp(e, "normal_ld") == 0 ? 2 : std::strcmp(e, "normal_ld_lo") == 0 ? 3 : std::strcmp(e, "tsqr") == 0 ? 4 : 0; }();
      // 2: the same in extended precision (x87 long double, 64-bit mantissa) — a proxy for "normal equations formed and factored more
      // accurately than fp64" (double-double on the device); 3: extended only for problems of LaserOdometry's size (R < 1000)
      const bool ext = normal_eq == 2 || (normal_eq == 3 && R < 1000);
      if (normal_eq == 4) {
        // 4: what a QR solve ON THE DEVICE would be — TSQR: the rows split into 64 strided blocks (one per thread), a Householder QR per block,
        // then a binary tree of QRs of stacked 6 x 6 triangles.  Same family as DENSE_QR, another order of operations.
        const int NB = 64;
        std::vector<std::vector<double>> Rk(NB, std::vector<double>(36, 0.0)), qk(NB, std::vector<double>(6, 0.0));
        for (int t = 0; t < NB; ++t) {
          std::vector<int> rows;
          for (int i = t; i < m; i += NB) rows.push_back(i);
          const int mb = std::max((int)rows.size(), 6);
          std::vector<double> Ab((size_t)mb * 6, 0.0), bb(mb, 0.0);
          for (size_t r = 0; r < rows.size(); ++r) { bb[r] = rhs[rows[r]]; for (int k = 0; k < 6; ++k) Ab[(size_t)k * mb + r] = A[(size_t)k * m + rows[r]]; }
          double dummy[6];
          qr_solve(Ab, bb, mb, 6, dummy);   // leaves R in the upper triangle of Ab and Q^T b in bb
          for (int r = 0; r < 6; ++r) { qk[t][r] = bb[r]; for (int k = r; k < 6; ++k) Rk[t][r * 6 + k] = Ab[(size_t)k * mb + r]; }
        }
        for (int stride = 1; stride < NB; stride *= 2)
          for (int t = 0; t + stride < NB; t += 2 * stride) {
            std::vector<double> Ab(12 * 6, 0.0), bb(12, 0.0);
            for (int r = 0; r < 6; ++r) { bb[r] = qk[t][r]; bb[6 + r] = qk[t + stride][r]; for (int k = 0; k < 6; ++k) { Ab[(size_t)k * 12 + r] = Rk[t][r * 6 + k]; Ab[(size_t)k * 12 + 6 + r] = Rk[t + stride][r * 6 + k]; } }
            double dummy[6];
            qr_solve(Ab, bb, 12, 6, dummy);
            for (int r = 0; r < 6; ++r) { qk[t][r] = bb[r]; for (int k = 0; k < 6; ++k) Rk[t][r * 6 + k] = k >= r ? Ab[(size_t)k * 12 + r] : 0.0; }
          }
        for (int k = 5; k >= 0; --k) { double sacc = qk[0][k]; for (int j = k + 1; j < 6; ++j) sacc -= Rk[0][k * 6 + j] * step[j]; step[k] = sacc / Rk[0][k * 6 + k]; }
      } else if (normal_eq && ext) {
        long double H[6][6] = {{0}}, g[6] = {0};
        for (int i = 0; i < R; ++i)
          for (int a = 0; a < 6; ++a) { g[a] += (long double)jac[(size_t)i * 6 + a] * residuals[i]; for (int b2_ = 0; b2_ <= a; ++b2_) H[a][b2_] += (long double)jac[(size_t)i * 6 + a] * jac[(size_t)i * 6 + b2_]; }
        for (int a = 0; a < 6; ++a) { for (int b2_ = a + 1; b2_ < 6; ++b2_) H[a][b2_] = H[b2_][a]; H[a][a] += (long double)diagonal[a] / radius; }
        long double L[6][6] = {{0}}, y[6];
        bool ok = true;
        for (int j = 0; j < 6; ++j) {
          long double sj = H[j][j];
          for (int k = 0; k < j; ++k) sj -= L[j][k] * L[j][k];
          if (!(sj > 0)) { ok = false; sj = 1; }
          L[j][j] = sqrtl(sj);
          for (int i = j + 1; i < 6; ++i) { long double t = H[i][j]; for (int k = 0; k < j; ++k) t -= L[i][k] * L[j][k]; L[i][j] = t / L[j][j]; }
        }
        for (int i = 0; i < 6; ++i) { long double t = g[i]; for (int k = 0; k < i; ++k) t -= L[i][k] * y[k]; y[i] = t / L[i][i]; }
        for (int i = 5; i >= 0; --i) { long double t = y[i]; for (int k = i + 1; k < 6; ++k) t -= L[k][i] * y[k]; y[i] = t / L[i][i]; }
        for (int k = 0; k < 6; ++k) step[k] = ok ? (double)y[k] : NAN;
      } else if (normal_eq) {
        double H[6][6] = {{0}}, g[6] = {0};
        for (int i = 0; i < R; ++i)
          for (int a = 0; a < 6; ++a) { g[a] += jac[(size_t)i * 6 + a] * residuals[i]; for (int b2_ = 0; b2_ <= a; ++b2_) H[a][b2_] += jac[(size_t)i * 6 + a] * jac[(size_t)i * 6 + b2_]; }
        for (int a = 0; a < 6; ++a) { for (int b2_ = a + 1; b2_ < 6; ++b2_) H[a][b2_] = H[b2_][a]; H[a][a] += diagonal[a] / radius; }
        double L[6][6] = {{0}}, y[6];
        bool ok = true;
        for (int j = 0; j < 6; ++j) {
          double sj = H[j][j];
          for (int k = 0; k < j; ++k) sj -= L[j][k] * L[j][k];
          if (!(sj > 0)) { ok = false; sj = 1; }
          L[j][j] = std::sqrt(sj);
          for (int i = j + 1; i < 6; ++i) { double t = H[i][j]; for (int k = 0; k < j; ++k) t -= L[i][k] * L[j][k]; L[i][j] = t / L[j][j]; }
        }
        for (int i = 0; i < 6; ++i) { double t = g[i]; for (int k = 0; k < i; ++k) t -= L[i][k] * y[k]; y[i] = t / L[i][i]; }
        for (int i = 5; i >= 0; --i) { double t = y[i]; for (int k = i + 1; k < 6; ++k) t -= L[k][i] * y[k]; y[i] = t / L[i][i]; }
        for (int k = 0; k < 6; ++k) step[k] = ok ? y[k] : NAN;
      } else
      qr_solve(A, rhs, m, 6, step);
      reuse_diagonal = true;
      bool finite = true;
      for (int k = 0; k < 6; ++k) { step[k] = -step[k]; finite = finite && std::isfinite(step[k]); }
      double model_cost_change = 0;
      if (finite) {
        for (int i = 0; i < R; ++i) {
          double mr = 0;
          for (int k = 0; k < 6; ++k) mr += jac[(size_t)i * 6 + k] * step[k];
          model_cost_change += mr * (residuals[i] + mr / 2.0);
        }
        model_cost_change = -model_cost_change;
      }
      if (!finite || !(model_cost_change > 0.0)) {  // HandleInvalidStep
        if (++num_invalid >= 5) { sum.termination = 4; break; }
        radius /= decrease_factor; decrease_factor *= 2.0; reuse_diagonal = true; step_successful = false;
        continue;
      }
      num_invalid = 0;
      double cand[6], cand_cost, step_norm = 0;
      for (int k = 0; k < 6; ++k) cand[k] = x[k] + step[k] * scale[k];
      if (!evaluate(cand, &cand_cost, false)) cand_cost = DBL_MAX;
      for (int k = 0; k < 6; ++k) step_norm += (x[k] - cand[k]) * (x[k] - cand[k]);
      step_norm = std::sqrt(step_norm);
      if (step_norm <= 1e-8 * (x_norm + 1e-8)) { sum.termination = 2; break; }
      double cost_change = x_cost - cand_cost;
      if (std::fabs(cost_change) <= 1e-6 * x_cost) { sum.termination = 3; break; }
      double relative_decrease = cost_change / model_cost_change;
      if (relative_decrease > 1e-3) {  // HandleSuccessfulStep
        std::memcpy(x, cand, sizeof(x));
        x_norm = 0;
        for (int k = 0; k < 6; ++k) x_norm += x[k] * x[k];
        x_norm = std::sqrt(x_norm);
        if (!evaluate(x, &x_cost, true)) { sum.termination = 4; break; }
        scale_jac();
        gmax = grad_max();
        step_successful = true;
        ++sum.successful;
        sum.final_cost = x_cost;
        std::memcpy(params, x, sizeof(x));
        radius = radius / std::max(1.0 / 3.0, 1.0 - std::pow(2.0 * relative_decrease - 1.0, 3));
        radius = std::min(1e16, radius);
        decrease_factor = 2.0;
        reuse_diagonal = false;
      } else {  // HandleUnsuccessfulStep
        step_successful = false;
        radius /= decrease_factor; decrease_factor *= 2.0; reuse_diagonal = true;
      }
    }
    sum.iterations = iter;
    return sum;
  }
};

// ===========================================================================
// 6. Feature extraction + LaserOdometry  (src/laserOdometry.cpp:111-535,728-740)
// ===========================================================================
struct LaserOdometry {
  Params P;
  int N = 0;
  // persistent state, laserOdometry.cpp:31-47
  std::vector<double> curvature;
  std::vector<int> picked, label, sort_idx;
  bool initialized = false;
  double params[6] = {0, 0, 0, 0, 0, 0};
  double t_w[3] = {0, 0, 0};
  double r_w[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  std::vector<Pt> surf_last, corner_last;
  KdTree kd_surf, kd_corner;
  // per-scan outputs
  std::vector<Pt> sharp, less_sharp, flat, less_flat;
  std::vector<int> sharp_idx, less_sharp_idx, flat_idx;  // indices into the segmented cloud
  std::vector<float> curv_d;                              // the f32 11-tap sum (diff_range) per point
  std::vector<uint8_t> picked_occl;                       // picked[] right after occlusion marking
  std::vector<int> out_label;
  std::vector<int> surf_corr, corner_corr;                // (j, closest, idx2, idx3) / (j, closest, idx2)
  double params_after_surf[6], params_after_corner[6];
  SolveSummary sum_surf, sum_corner;
  bool odom_valid = false;
  int n_surf_corr = 0, n_corner_corr = 0;
  double t_fe_ms = 0, t_assoc_ms = 0, t_solve_ms = 0;

  // ---- IMU ring + motion de-skew (laserOdometry.cpp:17-29,557-726,761-802; the call at :115 is commented out in the reference:
  //      deskew_mode = 0 is the reference as shipped) ----
  static const int IMU_Q = 200;   // imu_queue_length, utility.h:70
  int imu_ptr_front = 0, imu_ptr_last = -1, imu_ptr_last_iter = 0;
  double imu_time[IMU_Q], imu_roll[IMU_Q], imu_pitch[IMU_Q], imu_yaw[IMU_Q], imu_shift[3][IMU_Q], imu_velo[3][IMU_Q];
  double scan_time = 0;            // t1: stamp of the segmented cloud (:111-115)
  std::vector<Pt> undistorted;     // the LO's copy of the segmented cloud after adjustDistortion (/undistorted, :719-726)
  bool deskew_aborted = false;

  // imuHandler :761-802.  sample = stamp, orientation (w x y z), linear_acceleration, angular_velocity (unused: :787-789)
  void imu_handler(const double* smp) {
    const double stamp = smp[0], qw = smp[1], qx = smp[2], qy = smp[3], qz = smp[4];
    // tf::Matrix3x3(ori).getRPY(roll, pitch, yaw)  [upstream tf: setRotation + getEulerYPR, solution 1]
    double roll, pitch, yaw;
    {
      const double d = qx * qx + qy * qy + qz * qz + qw * qw, sc = 2.0 / d;
      const double xs = qx * sc, ys = qy * sc, zs = qz * sc;
      const double wx = qw * xs, wy = qw * ys, wz = qw * zs, xx = qx * xs, xy = qx * ys, xz = qx * zs, yy = qy * ys, yz = qy * zs, zz = qz * zs;
      const double m00 = 1.0 - (yy + zz), m10 = xy + wz, m20 = xz - wy, m21 = yz + wx, m22 = 1.0 - (xx + yy), m01 = xy - wz, m02 = xz + wy;
      if (std::fabs(m20) >= 1) {
        yaw = 0;
        const double delta = std::atan2(m21, m22);
        if (m20 < 0) { pitch = M_PI / 2.0; roll = delta; } else { pitch = -M_PI / 2.0; roll = delta; }
        (void)m01; (void)m02;
      } else {
        pitch = -std::asin(m20);
        roll = std::atan2(m21 / std::cos(pitch), m22 / std::cos(pitch));
        yaw = std::atan2(m10 / std::cos(pitch), m00 / std::cos(pitch));
      }
    }
    const double acc_x = smp[5] + 9.81 * std::sin(pitch);
    const double acc_y = smp[6] - 9.81 * std::cos(pitch) * std::sin(roll);
    const double acc_z = smp[7] - 9.81 * std::cos(pitch) * std::cos(roll);
    imu_ptr_last = (imu_ptr_last + 1) % IMU_Q;
    if ((imu_ptr_last + 1) % IMU_Q == imu_ptr_front) imu_ptr_front = (imu_ptr_front + 1) % IMU_Q;
    imu_time[imu_ptr_last] = stamp; imu_roll[imu_ptr_last] = roll; imu_pitch[imu_ptr_last] = pitch; imu_yaw[imu_ptr_last] = yaw;
    // Eigen::Quaternionf(w, x, y, z).toRotationMatrix() * Vector3f(acc): f32 (:785-786)
    const float w = (float)qw, x = (float)qx, y = (float)qy, z = (float)qz;
    const float tx = 2 * x, ty = 2 * y, tz = 2 * z;
    const float twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y, tyz = tz * y, tzz = tz * z;
    const float R[3][3] = {{1 - (tyy + tzz), txy - twz, txz + twy}, {txy + twz, 1 - (txx + tzz), tyz - twx}, {txz - twy, tyz + twx, 1 - (txx + tyy)}};
    const float a[3] = {(float)acc_x, (float)acc_y, (float)acc_z};
    float acc[3];
    for (int i = 0; i < 3; ++i) acc[i] = R[i][0] * a[0] + (R[i][1] * a[1] + R[i][2] * a[2]);   // Eigen's unrolled 3-term reduction: a0 + (a1 + a2)
    const int back = (imu_ptr_last - 1 + IMU_Q) % IMU_Q;
    const double time_diff = imu_time[imu_ptr_last] - imu_time[back];
    if (time_diff < 1.) {
      for (int k = 0; k < 3; ++k) {
        imu_shift[k][imu_ptr_last] = imu_shift[k][back] + imu_velo[k][back] * time_diff + acc[k] * time_diff * time_diff * 0.5;
        imu_velo[k][imu_ptr_last] = imu_velo[k][back] + acc[k] * time_diff;
      }
    }
  }

  static void rpy_matrix(const float rpy[3], float m[3][3]) {   // (AngleAxisf(yaw, Z) * AngleAxisf(pitch, Y) * AngleAxisf(roll, X)).toRotationMatrix()
    auto qaxis = [](float angle, int axis, float q[4]) {
      float ha = 0.5f * angle, sn = omath::o_sinf(ha);
      q[0] = omath::o_cosf(ha); q[1] = q[2] = q[3] = 0.f; q[1 + axis] = sn;
    };
    auto qmul = [](const float a[4], const float b[4], float o[4]) {
      o[0] = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
      o[1] = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
      o[2] = a[0] * b[2] + a[2] * b[0] + a[3] * b[1] - a[1] * b[3];
      o[3] = a[0] * b[3] + a[3] * b[0] + a[1] * b[2] - a[2] * b[1];
    };
    float qz[4], qy[4], qx[4], qzy[4], q[4];
    qaxis(rpy[2], 2, qz); qaxis(rpy[1], 1, qy); qaxis(rpy[0], 0, qx);
    qmul(qz, qy, qzy); qmul(qzy, qx, q);
    const float w = q[0], x = q[1], y = q[2], z = q[3];
    const float tx = 2 * x, ty = 2 * y, tz = 2 * z;
    const float twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y, tyz = tz * y, tzz = tz * z;
    m[0][0] = 1 - (tyy + tzz); m[0][1] = txy - twz; m[0][2] = txz + twy;
    m[1][0] = txy + twz; m[1][1] = 1 - (txx + tzz); m[1][2] = tyz - twx;
    m[2][0] = txz - twy; m[2][1] = tyz + twx; m[2][2] = 1 - (txx + tyy);
  }
  static void inverse3(const float m[3][3], float r[3][3]) {   // Eigen Matrix3f::inverse(): cofactors / determinant (Inverse.h, size 3)
    auto cof = [&](int i, int j) { return m[(i + 1) % 3][(j + 1) % 3] * m[(i + 2) % 3][(j + 2) % 3] - m[(i + 1) % 3][(j + 2) % 3] * m[(i + 2) % 3][(j + 1) % 3]; };
    const float c0[3] = {cof(0, 0), cof(1, 0), cof(2, 0)};
    const float det = c0[0] * m[0][0] + (c0[1] * m[1][0] + c0[2] * m[2][0]);
    const float invdet = 1.0f / det;
    for (int j = 0; j < 3; ++j) r[0][j] = c0[j] * invdet;
    r[1][0] = cof(0, 1) * invdet; r[1][1] = cof(1, 1) * invdet; r[1][2] = cof(2, 1) * invdet;
    r[2][0] = cof(0, 2) * invdet; r[2][1] = cof(1, 2) * invdet; r[2][2] = cof(2, 2) * invdet;
  }
  static void mul3(const float m[3][3], const float v[3], float o[3]) {
    for (int i = 0; i < 3; ++i) o[i] = m[i][0] * v[0] + (m[i][1] * v[1] + m[i][2] * v[2]);
  }

  // adjustDistortion :557-726, IMU branch (use_imu = true, utility.h:68)
  void adjust_distortion(std::vector<Pt>& cloud, const ImageProjection& ip) {
    deskew_aborted = false;
    const int H = P.horizon_scan;
    const int cloud_size = (int)cloud.size();
    int start_ori = (int)((ip.ori[0] + 2 * M_PI) / H);   // :562-563 (sic: an angle in radians divided by Horizon_SCAN)
    int end_ori = (int)((ip.ori[1] + 2 * M_PI) / H);
    if (start_ori >= H) start_ori -= H;
    if (end_ori >= H) end_ori -= H;
    int ori_diff = end_ori - start_ori;
    if (ori_diff <= 0) ori_diff = H;
    float rpy_start[3] = {0, 0, 0}, shift_start[3] = {0, 0, 0}, velo_start[3] = {0, 0, 0}, rpy_cur[3], shift_cur[3], velo_cur[3];
    float r_s_i[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}}, r_c[3][3];
    (void)rpy_start;
    for (int i = 0; i < cloud_size; ++i) {
      Pt& p = cloud[i];
      const double rel_time = (ip.seg_col[i] - start_ori) * P.scan_period / ori_diff;
      const double cur_time = scan_time + rel_time;
      if (imu_ptr_last > 0) {
        imu_ptr_front = imu_ptr_last_iter;
        while (imu_ptr_front != imu_ptr_last) {
          if (cur_time < imu_time[imu_ptr_front]) break;
          imu_ptr_front = (imu_ptr_front + 1) % IMU_Q;
        }
        if (std::abs(cur_time - imu_time[imu_ptr_front]) > P.scan_period) { deskew_aborted = true; return; }   // :604-608
        const int fr = imu_ptr_front;
        if (cur_time > imu_time[fr]) {
          rpy_cur[0] = (float)imu_roll[fr]; rpy_cur[1] = (float)imu_pitch[fr]; rpy_cur[2] = (float)imu_yaw[fr];
          for (int k = 0; k < 3; ++k) { shift_cur[k] = (float)imu_shift[k][fr]; velo_cur[k] = (float)imu_velo[k][fr]; }
        } else {
          const int bk = (fr - 1 + IMU_Q) % IMU_Q;
          const double ratio_front = (cur_time - imu_time[bk]) / (imu_time[fr] - imu_time[bk]);
          const double ratio_back = 1. - ratio_front;
          rpy_cur[0] = (float)(imu_roll[fr] * ratio_front + imu_roll[bk] * ratio_back);
          rpy_cur[1] = (float)(imu_pitch[fr] * ratio_front + imu_pitch[bk] * ratio_back);
          rpy_cur[2] = (float)(imu_yaw[fr] * ratio_front + imu_yaw[bk] * ratio_back);
          for (int k = 0; k < 3; ++k) {
            shift_cur[k] = (float)(imu_shift[k][fr] * ratio_front + imu_shift[k][bk] * ratio_back);
            velo_cur[k] = (float)(imu_velo[k][fr] * ratio_front + imu_velo[k][bk] * ratio_back);
          }
        }
        rpy_matrix(rpy_cur, r_c);
        if (i == 0) {
          for (int k = 0; k < 3; ++k) { rpy_start[k] = rpy_cur[k]; shift_start[k] = shift_cur[k]; velo_start[k] = velo_cur[k]; }
          inverse3(r_c, r_s_i);
        } else {
          const float rt = (float)rel_time;   // Vector3f * double: the scalar is converted to the vector's scalar type
          float sfs[3], v[3], w[3], o[3];
          for (int k = 0; k < 3; ++k) sfs[k] = (shift_cur[k] - shift_start[k]) - velo_start[k] * rt;
          const float pv[3] = {p.x, p.y, p.z};
          mul3(r_c, pv, v);
          for (int k = 0; k < 3; ++k) w[k] = v[k] + sfs[k];
          mul3(r_s_i, w, o);
          p.x = o[0]; p.y = o[1]; p.z = o[2];
        }
        imu_ptr_last_iter = imu_ptr_front;
      }
    }
  }

  void init(const Params& p) {
    P = p; N = P.n_scan * P.horizon_scan;
    curvature.assign(N, 0); picked.assign(N, 0); label.assign(N, 0); sort_idx.assign(N, 0);
    initialized = false;
    for (int i = 0; i < 6; ++i) params[i] = 0;
    t_w[0] = t_w[1] = t_w[2] = 0;
    for (int i = 0; i < 9; ++i) r_w[i] = (i % 4 == 0) ? 1 : 0;
    surf_last.clear(); corner_last.clear();
    imu_ptr_front = 0; imu_ptr_last = -1; imu_ptr_last_iter = 0;   // :17-29
    for (int i = 0; i < IMU_Q; ++i) {
      imu_time[i] = imu_roll[i] = imu_pitch[i] = imu_yaw[i] = 0;
      for (int k = 0; k < 3; ++k) imu_shift[k][i] = imu_velo[k][i] = 0;
    }
  }

  // transformToStart, laserOdometry.cpp:728-740
  void transform_to_start(const Pt& pi, Pt& po) const {
    Quat q = quat_zyx(params[5], params[4], params[3]);
    double R[9];
    quat_to_mat(q, R);
    double x = pi.x, y = pi.y, z = pi.z;
    po.x = (float)(R[0] * x + R[1] * y + R[2] * z + params[0]);
    po.y = (float)(R[3] * x + R[4] * y + R[5] * z + params[1]);
    po.z = (float)(R[6] * x + R[7] * y + R[8] * z + params[2]);
    po.intensity = pi.intensity;
  }

  void extract_features(const ImageProjection& ip) {
    if (P.deskew_mode == 1) { undistorted = ip.seg_cloud; adjust_distortion(undistorted, ip); }   // :115 (commented out in the reference)
    const std::vector<Pt>& seg = P.deskew_mode == 1 ? undistorted : ip.seg_cloud;
    const int cloud_size = (int)seg.size();
    const float* rng = ip.seg_range.data();
    const int* colv = ip.seg_col.data();
    const uint8_t* gnd = ip.seg_ground.data();
    curv_d.assign(cloud_size, 0.f);
    // step2 calculateSmoothness :122-129
    for (int i = 5; i < cloud_size - 5; ++i) {
      float d = rng[i - 5] + rng[i - 4] + rng[i - 3] + rng[i - 2] + rng[i - 1] - rng[i] * 10 + rng[i + 1] + rng[i + 2] + rng[i + 3] + rng[i + 4] + rng[i + 5];
      double diff_range = d;
      curv_d[i] = d;
      curvature[i] = diff_range * diff_range;
      picked[i] = 0; label[i] = 0; sort_idx[i] = i;
    }
    // step3 markOccludedPoints :131-159 (LO.cpp:203-230 when occl_f32)
    for (int i = 5; i < cloud_size - 5; ++i) {
      int col_diff = std::abs(colv[i] - colv[i + 1]);
      bool c1, c2;
      double diff1, diff2;
      if (P.occl_f32) {
        float depth1 = rng[i], depth2 = rng[i + 1];
        c1 = (double)(depth1 - depth2) > P.occl_depth; c2 = (double)(depth2 - depth1) > P.occl_depth;
        diff1 = std::abs(rng[i - 1] - depth1); diff2 = std::abs(depth2 - depth1);
      } else {
        double depth1 = rng[i], depth2 = rng[i + 1];
        c1 = depth1 - depth2 > P.occl_depth; c2 = depth2 - depth1 > P.occl_depth;
        diff1 = std::abs(rng[i - 1] - depth1); diff2 = std::abs(depth2 - depth1);
      }
      if (col_diff < P.occl_col_diff) {
        if (c1) { for (int l = 0; l <= 5; ++l) picked[i - l] = 1; continue; }
        else if (c2) { for (int l = 1; l <= 5; ++l) picked[i + l] = 1; }
      }
      if (diff1 > P.parallel_ratio * rng[i] && diff2 > P.parallel_ratio * rng[i]) picked[i] = 1;
    }
    picked_occl.assign(cloud_size, 0);
    for (int i = 0; i < cloud_size; ++i) picked_occl[i] = (uint8_t)picked[i];
    // step4 extractFeatures :164-294
    sharp.clear(); less_sharp.clear(); flat.clear(); less_flat.clear();
    sharp_idx.clear(); less_sharp_idx.clear(); flat_idx.clear();
    std::vector<Pt> less_flat_scan, less_flat_scan_ds;
    const int SR = P.suppress_radius;
    auto suppress = [&](int idx) {
      for (int l = 1; l <= SR; ++l) {
        int cd = std::abs(colv[idx + l] - colv[idx + l - 1]);
        if (cd > P.suppress_col_diff) break; else picked[idx + l] = 1;
      }
      for (int l = -1; l >= -SR; --l) {
        int cd = std::abs(colv[idx + l] - colv[idx + l + 1]);
        if (cd > P.suppress_col_diff) break; else picked[idx + l] = 1;
      }
    };
    for (int i = 0; i < P.n_scan; ++i) {
      less_flat_scan.clear();
      const int S = ip.start_ring[i], E = ip.end_ring[i], NSEC = P.n_sectors;
      for (int j = 0; j < NSEC; ++j) {
        int sp, ep;
        if (P.sector_formula == 0) { sp = (S * (NSEC - j) + E * j) / NSEC; ep = (S * (NSEC - 1 - j) + E * (j + 1)) / NSEC - 1; }
        else { int diff = E - S; sp = S + j * diff / NSEC; ep = S + (j + 1) * diff / NSEC - 1; }
        if (sp >= ep) continue;
        auto cmp = [this](int a, int b) { return curvature[a] < curvature[b]; };
        if (P.sort_mode == 1 || P.sort_mode == 2) std::sort(sort_idx.begin() + sp, sort_idx.begin() + ep + 1, cmp);   // :185
        else std::stable_sort(sort_idx.begin() + sp, sort_idx.begin() + ep + 1, cmp);
        int picked_num = 0;
        for (int k = ep; k >= sp; --k) {
          int idx = sort_idx[k];
          if (picked[idx] == 0 && curvature[idx] > P.edge_thres && gnd[idx] == 0) {
            ++picked_num;
            picked[idx] = 1;
            if (picked_num <= P.n_sharp) {
              label[idx] = 2; sharp.push_back(seg[idx]); sharp_idx.push_back(idx);
              less_sharp.push_back(seg[idx]); less_sharp_idx.push_back(idx);
            } else if (picked_num <= P.n_less_sharp) {
              label[idx] = 1; less_sharp.push_back(seg[idx]); less_sharp_idx.push_back(idx);
            } else break;
            suppress(idx);
          }
        }
        picked_num = 0;
        for (int k = sp; k <= ep; ++k) {
          int idx = sort_idx[k];
          if (picked[idx] == 0 && curvature[idx] < P.surf_thres && gnd[idx] == 1) {
            label[idx] = -1; flat.push_back(seg[idx]); flat_idx.push_back(idx);
            ++picked_num;
            picked[idx] = 1;
            if (picked_num >= P.n_flat) break;
            suppress(idx);
          }
        }
        for (int k = sp; k <= ep; ++k) if (label[k] <= 0) less_flat_scan.push_back(seg[k]);
      }
      voxel_grid(less_flat_scan, P.less_flat_leaf, less_flat_scan_ds, P.sort_mode);
      less_flat.insert(less_flat.end(), less_flat_scan_ds.begin(), less_flat_scan_ds.end());
    }
    out_label.assign(cloud_size, 0);
    for (int i = 5; i < cloud_size - 5; ++i) out_label[i] = label[i];
  }

  static inline double walk_dist(const Pt& a, const Pt& q) {  // :354 pow(float diff, 2) in double
    return std::pow(a.x - q.x, 2) + std::pow(a.y - q.y, 2) + std::pow(a.z - q.z, 2);
  }

  void associate_surf(std::vector<Block>& blocks) {  // :337-407
    surf_corr.clear(); n_surf_corr = 0;
    for (int j = 0; j < (int)flat.size(); ++j) {
      Pt sel; transform_to_start(flat[j], sel);
      int sidx[1]; float sdist[1];
      if (kd_surf.knn(sel, 1, sidx, sdist) < 1) continue;
      int closest = -1, min2 = -1, min3 = -1;
      if ((double)sdist[0] < P.nearest_feature_dist) {
        closest = sidx[0];
        double min_dist2 = P.nearest_feature_dist, min_dist3 = P.nearest_feature_dist;
        int closest_scan = (int)surf_last[closest].intensity;
        for (int k = closest + 1; k < (int)surf_last.size(); ++k) {
          if ((int)surf_last[k].intensity > closest_scan + P.ring_window + 0.5) break;
          double pd = walk_dist(surf_last[k], sel);
          if ((int)surf_last[k].intensity == closest_scan) { if (pd < min_dist2) { min_dist2 = pd; min2 = k; } }
          else { if (pd < min_dist3) { min_dist3 = pd; min3 = k; } }
        }
        for (int k = closest - 1; k >= 0; --k) {
          if ((int)surf_last[k].intensity < closest_scan - P.ring_window - 0.5) break;
          double pd = walk_dist(surf_last[k], sel);
          if ((int)surf_last[k].intensity == closest_scan) { if (pd < min_dist2) { min_dist2 = pd; min2 = k; } }
          else { if (pd < min_dist3) { min_dist3 = pd; min3 = k; } }
        }
        if (min2 >= 0 && min3 >= 0) {
          Block B; B.type = BLK_SURF; B.d = 0;
          const Pt& cp = flat[j]; const Pt &a = surf_last[closest], &b = surf_last[min2], &c = surf_last[min3];
          B.cp[0] = cp.x; B.cp[1] = cp.y; B.cp[2] = cp.z;
          B.a[0] = a.x; B.a[1] = a.y; B.a[2] = a.z; B.b[0] = b.x; B.b[1] = b.y; B.b[2] = b.z; B.c[0] = c.x; B.c[1] = c.y; B.c[2] = c.z;
          blocks.push_back(B);
          surf_corr.push_back(j); surf_corr.push_back(closest); surf_corr.push_back(min2); surf_corr.push_back(min3);
          ++n_surf_corr;
        }
      }
    }
  }

  void associate_corner(std::vector<Block>& blocks) {  // :427-481
    corner_corr.clear(); n_corner_corr = 0;
    for (int j = 0; j < (int)sharp.size(); ++j) {
      Pt sel; transform_to_start(sharp[j], sel);
      int sidx[1]; float sdist[1];
      if (kd_corner.knn(sel, 1, sidx, sdist) < 1) continue;
      int closest = -1, min2 = -1;
      if ((double)sdist[0] < P.nearest_feature_dist) {
        closest = sidx[0];
        int closest_scan = (int)corner_last[closest].intensity;
        double min_d2 = P.nearest_feature_dist;
        for (int k = closest + 1; k < (int)corner_last.size(); ++k) {
          if ((int)corner_last[k].intensity > closest_scan + P.ring_window) break;
          double pd = walk_dist(corner_last[k], sel);
          if ((int)corner_last[k].intensity > closest_scan) { if (pd < min_d2) { min_d2 = pd; min2 = k; } }
        }
        for (int k = closest - 1; k >= 0; --k) {
          if ((int)corner_last[k].intensity < closest_scan - P.ring_window) break;
          double pd = walk_dist(corner_last[k], sel);
          if ((int)corner_last[k].intensity < closest_scan) { if (pd < min_d2) { min_d2 = pd; min2 = k; } }
        }
      }
      if (min2 >= 0) {
        Block B; B.type = BLK_CORNER; B.d = 0;
        const Pt& cp = sharp[j]; const Pt &a = corner_last[closest], &b = corner_last[min2];
        B.cp[0] = cp.x; B.cp[1] = cp.y; B.cp[2] = cp.z;
        B.a[0] = a.x; B.a[1] = a.y; B.a[2] = a.z; B.b[0] = b.x; B.b[1] = b.y; B.b[2] = b.z;
        B.c[0] = B.c[1] = B.c[2] = 0;
        blocks.push_back(B);
        corner_corr.push_back(j); corner_corr.push_back(closest); corner_corr.push_back(min2);
        ++n_corner_corr;
      }
    }
  }

  // mainLoop body for one synced triple, laserOdometry.cpp:111-535.  Returns true when an
  // odometry message is published (not on the initialising scan, :316-324).
  bool process(const ImageProjection& ip) {
    auto t0 = std::chrono::steady_clock::now();
    extract_features(ip);
    auto t1 = std::chrono::steady_clock::now();
    t_fe_ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
    t_assoc_ms = t_solve_ms = 0;
    odom_valid = false;
    surf_corr.clear(); corner_corr.clear(); n_surf_corr = n_corner_corr = 0;
    if (!initialized) {
      initialized = true;
    } else {
      std::vector<Block> blocks;
      CeresLike solver;
      auto a0 = std::chrono::steady_clock::now();
      associate_surf(blocks);
      auto a1 = std::chrono::steady_clock::now();
      if (n_surf_corr >= P.lo_min_corr) sum_surf = solver.solve(blocks, params, P.lo_iters_surf, P.huber_delta);
      auto a2 = std::chrono::steady_clock::now();
      std::memcpy(params_after_surf, params, sizeof(params));
      associate_corner(blocks);
      auto a3 = std::chrono::steady_clock::now();
      if (n_corner_corr >= P.lo_min_corr) sum_corner = solver.solve(blocks, params, P.lo_iters_corner, P.huber_delta);
      auto a4 = std::chrono::steady_clock::now();
      std::memcpy(params_after_corner, params, sizeof(params));
      t_assoc_ms = std::chrono::duration<double, std::milli>((a1 - a0) + (a3 - a2)).count();
      t_solve_ms = std::chrono::duration<double, std::milli>((a2 - a1) + (a4 - a3)).count();
      // pose integration :504-508
      double c = std::cos(params[5]), s = std::sin(params[5]);
      double rl[9] = {c, -s, 0, s, c, 0, 0, 0, 1};
      double nt[3], nr[9];
      for (int i = 0; i < 3; ++i) nt[i] = t_w[i] + (r_w[i * 3 + 0] * params[0] + r_w[i * 3 + 1] * params[1] + r_w[i * 3 + 2] * params[2]);
      for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) nr[i * 3 + j] = r_w[i * 3 + 0] * rl[0 * 3 + j] + r_w[i * 3 + 1] * rl[1 * 3 + j] + r_w[i * 3 + 2] * rl[2 * 3 + j];
      std::memcpy(t_w, nt, sizeof(nt)); std::memcpy(r_w, nr, sizeof(nr));
      odom_valid = true;
    }
    surf_last = less_flat; corner_last = less_sharp;  // :319-322, :531-534
    kd_surf.build(surf_last); kd_corner.build(corner_last);
    return odom_valid;
  }
};

// ===========================================================================
// 7. LaserMapping scan-to-map registration (src/laserMapping.cpp:102-131,188-192,
//    194-244,315-559; include/alego/laserMapping.h:164-194)
// ===========================================================================
struct KeyPose { float x, y, z, roll, pitch, yaw; };  // PointXYZIRPYT, utility.h:83-92

struct LaserMapping {
  Params P;
  double params[6] = {0, 0, 0, 0, 0, 0};
  double t_map2odom[3] = {0, 0, 0}; Quat q_map2odom{1, 0, 0, 0};
  double t_odom2laser[3] = {0, 0, 0}; Quat q_odom2laser{1, 0, 0, 0};
  double t_map2laser[3] = {0, 0, 0}; Quat q_map2laser{1, 0, 0, 0};
  std::vector<KeyPose> keyposes;
  std::vector<std::vector<Pt>> corner_frames, surf_frames, outlier_frames;
  std::deque<std::vector<Pt>> recent_corner, recent_surf, recent_outlier;
  int latest_frame_id = -1;
  std::vector<Pt> corner_from_map, surf_from_map, corner_from_map_ds, surf_from_map_ds;
  std::vector<Pt> laser_corner_ds, laser_surf_ds, laser_outlier_ds, laser_surf_total, laser_surf_total_ds;
  KdTree kd_corner_map, kd_surf_map;
  int frame_cnt = 0;
  // outputs / debug
  bool ran_body = false, optimized = false, keyframe_added = false;
  int n_corner_corr = 0, n_surf_corr = 0;
  std::vector<int> corner_corr_q, surf_corr_q;  // accepted query indices (first outer iteration)
  std::vector<double> blocks14;                 // the residual blocks of the first outer iteration: type, cp, a, b, c, d
  std::vector<int> plane_rank_hist = std::vector<int>(4, 0);   // first outer iteration: plane fits by nonzeroPivots() (tests: were rank-deficient neighbourhoods reached?)
  SolveSummary sums[2];
  double params_iter[2][6];
  double t_map_ms = 0, t_ds_ms = 0, t_tree_ms = 0, t_assoc_ms = 0, t_solve_ms = 0;

  void init(const Params& p) { *this = LaserMapping(); P = p; }

  // transformPointCloud, laserMapping.h:164-177 (f32 4x4, pcl::transformPointCloud PCL 1.8 dense path)
  static void transform_cloud(const std::vector<Pt>& in, const KeyPose& kp, std::vector<Pt>& out) {
    // Eigen's Quaternionf(AngleAxisf) evaluates std::cos / std::sin on the float half angle: glibc's cosf / sinf,
    // restated in oracle_math.h (pinned bit for bit against this host's libm) and shared with the HIP path.
    auto qaxis = [](float angle, int axis, float q[4]) {
      float ha = 0.5f * angle, s = omath::o_sinf(ha);
      q[0] = omath::o_cosf(ha); q[1] = q[2] = q[3] = 0.f; q[1 + axis] = s;
    };
    auto qmul = [](const float a[4], const float b[4], float o[4]) {
      o[0] = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
      o[1] = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
      o[2] = a[0] * b[2] + a[2] * b[0] + a[3] * b[1] - a[1] * b[3];
      o[3] = a[0] * b[3] + a[3] * b[0] + a[1] * b[2] - a[2] * b[1];
    };
    float qz[4], qy[4], qx[4], qzy[4], q[4];
    qaxis(kp.yaw, 2, qz); qaxis(kp.pitch, 1, qy); qaxis(kp.roll, 0, qx);
    qmul(qz, qy, qzy); qmul(qzy, qx, q);
    float w = q[0], x = q[1], y = q[2], z = q[3];
    float tx = 2 * x, ty = 2 * y, tz = 2 * z;
    float twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y, tyz = tz * y, tzz = tz * z;
    float m[3][4] = {{1 - (tyy + tzz), txy - twz, txz + twy, kp.x},
                     {txy + twz, 1 - (txx + tzz), tyz - twx, kp.y},
                     {txz - twy, tyz + twx, 1 - (txx + tyy), kp.z}};
    out.resize(in.size());
    for (size_t i = 0; i < in.size(); ++i) {
      const Pt& p = in[i];
      out[i].x = m[0][0] * p.x + m[0][1] * p.y + m[0][2] * p.z + m[0][3];
      out[i].y = m[1][0] * p.x + m[1][1] * p.y + m[1][2] * p.z + m[1][3];
      out[i].z = m[2][0] * p.x + m[2][1] * p.y + m[2][2] * p.z + m[2][3];
      out[i].intensity = p.intensity;
    }
  }

  void transform_associate_to_map() {  // :188-192
    double r[3];
    quat_rotate(q_map2odom, t_odom2laser, r);
    for (int i = 0; i < 3; ++i) t_map2laser[i] = r[i] + t_map2odom[i];
    q_map2laser = quat_mul(q_map2odom, q_odom2laser);
  }

  void extract_surrounding_keyframes() {  // :194-244 (loop_closure_enabled_ branch), :315-319
    surf_from_map.clear(); corner_from_map.clear(); surf_from_map_ds.clear(); corner_from_map_ds.clear();
    if (keyposes.empty()) return;
    const int K = P.recent_keyframe_num;
    if ((int)recent_corner.size() < K) {
      recent_corner.clear(); recent_surf.clear(); recent_outlier.clear();
      for (int i = (int)keyposes.size() - 1; i >= 0; --i) {
        std::vector<Pt> c, s, o;
        transform_cloud(corner_frames[i], keyposes[i], c);
        transform_cloud(surf_frames[i], keyposes[i], s);
        transform_cloud(outlier_frames[i], keyposes[i], o);
        recent_corner.push_front(std::move(c)); recent_surf.push_front(std::move(s)); recent_outlier.push_front(std::move(o));
        if ((int)recent_corner.size() >= K) break;
      }
    } else if (latest_frame_id != (int)keyposes.size() - 1) {
      recent_corner.pop_front(); recent_surf.pop_front(); recent_outlier.pop_front();
      latest_frame_id = (int)keyposes.size() - 1;
      std::vector<Pt> c, s, o;
      transform_cloud(corner_frames[latest_frame_id], keyposes[latest_frame_id], c);
      transform_cloud(surf_frames[latest_frame_id], keyposes[latest_frame_id], s);
      transform_cloud(outlier_frames[latest_frame_id], keyposes[latest_frame_id], o);
      recent_corner.push_back(std::move(c)); recent_surf.push_back(std::move(s)); recent_outlier.push_back(std::move(o));
    }
    for (size_t i = 0; i < recent_corner.size(); ++i) {
      corner_from_map.insert(corner_from_map.end(), recent_corner[i].begin(), recent_corner[i].end());
      surf_from_map.insert(surf_from_map.end(), recent_surf[i].begin(), recent_surf[i].end());
      surf_from_map.insert(surf_from_map.end(), recent_outlier[i].begin(), recent_outlier[i].end());
    }
    auto t0 = std::chrono::steady_clock::now();
    voxel_grid(surf_from_map, P.lm_leaf_surf, surf_from_map_ds, P.sort_mode);
    voxel_grid(corner_from_map, P.lm_leaf_corner, corner_from_map_ds, P.sort_mode);
    t_ds_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  }

  void downsample_current_scan(const std::vector<Pt>& corner, const std::vector<Pt>& surf, const std::vector<Pt>& outlier) {  // :325-346
    auto t0 = std::chrono::steady_clock::now();
    voxel_grid(corner, P.lm_leaf_corner, laser_corner_ds, P.sort_mode);
    voxel_grid(surf, P.lm_leaf_surf, laser_surf_ds, P.sort_mode);
    voxel_grid(outlier, P.lm_leaf_outlier, laser_outlier_ds, P.sort_mode);
    laser_surf_total = laser_surf_ds;
    laser_surf_total.insert(laser_surf_total.end(), laser_outlier_ds.begin(), laser_outlier_ds.end());
    voxel_grid(laser_surf_total, P.lm_leaf_surf, laser_surf_total_ds, P.sort_mode);
    t_ds_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  }

  void point_associate_to_map(const Pt& in, Pt& out) const {  // laserMapping.h:187-194
    double v[3] = {in.x, in.y, in.z}, r[3];
    quat_rotate(q_map2laser, v, r);
    out.x = (float)(r[0] + t_map2laser[0]); out.y = (float)(r[1] + t_map2laser[1]); out.z = (float)(r[2] + t_map2laser[2]);
    out.intensity = in.intensity;
  }

  void scan2map_optimization() {  // :348-479
    optimized = false; n_corner_corr = n_surf_corr = 0;
    corner_corr_q.clear(); surf_corr_q.clear();
    if ((int)laser_corner_ds.size() < P.lm_min_corner || (int)laser_surf_total.size() < P.lm_min_surf ||
        (int)corner_from_map_ds.size() < P.lm_min_map_corner) return;
    auto t0 = std::chrono::steady_clock::now();
    kd_corner_map.build(corner_from_map_ds);
    kd_surf_map.build(surf_from_map_ds);
    auto t1 = std::chrono::steady_clock::now();
    t_tree_ms += std::chrono::duration<double, std::milli>(t1 - t0).count();
    for (int iter_cnt = 0; iter_cnt < P.lm_outer_iters; ++iter_cnt) {
      auto a0 = std::chrono::steady_clock::now();
      std::vector<Block> blocks;
      int cc = 0, sc = 0;
      int nidx[5]; float ndist[5];
      for (int i = 0; i < (int)laser_corner_ds.size(); ++i) {  // :371-417
        Pt sel; point_associate_to_map(laser_corner_ds[i], sel);
        if (kd_corner_map.knn(sel, 5, nidx, ndist) < 5) continue;
        if ((double)ndist[4] < P.knn_max_dist) {
          double near[5][3], center[3] = {0, 0, 0};
          for (int j = 0; j < 5; ++j) {
            const Pt& m = corner_from_map_ds[nidx[j]];
            near[j][0] = m.x; near[j][1] = m.y; near[j][2] = m.z;
            for (int a = 0; a < 3; ++a) center[a] = center[a] + near[j][a];
          }
          for (int a = 0; a < 3; ++a) center[a] = center[a] / 5.0;
          double cov[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
          for (int j = 0; j < 5; ++j) {
            double zm[3] = {near[j][0] - center[0], near[j][1] - center[1], near[j][2] - center[2]};
            for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) cov[a * 3 + b] = cov[a * 3 + b] + zm[a] * zm[b];
          }
          double lam[3], V[9];
          eig3(cov, lam, V);
          if (lam[2] > P.line_ratio * lam[1]) {
            Block B; B.type = BLK_EDGE; B.d = 0;
            const Pt& cp = laser_corner_ds[i];
            B.cp[0] = cp.x; B.cp[1] = cp.y; B.cp[2] = cp.z;
            for (int a = 0; a < 3; ++a) {
              double u = V[a * 3 + 2];
              B.a[a] = P.line_half_len * u + center[a];
              B.b[a] = -P.line_half_len * u + center[a];
              B.c[a] = 0;
            }
            blocks.push_back(B);
            ++cc;
            if (iter_cnt == 0) corner_corr_q.push_back(i);
          }
        }
      }
      if (iter_cnt == 0) plane_rank_hist.assign(4, 0);
      for (int i = 0; i < (int)laser_surf_total_ds.size(); ++i) {  // :419-462
        Pt sel; point_associate_to_map(laser_surf_total_ds[i], sel);
        if (kd_surf_map.knn(sel, 5, nidx, ndist) < 5) continue;
        if ((double)ndist[4] < P.knn_max_dist) {
          double A[15], b[5] = {-1.0, -1.0, -1.0, -1.0, -1.0};   // Matrix<double, 5, 3> / Matrix<double, 5, 1>: fixed size, no heap
          for (int j = 0; j < 5; ++j) {
            const Pt& m = surf_from_map_ds[nidx[j]];
            A[0 * 5 + j] = m.x; A[1 * 5 + j] = m.y; A[2 * 5 + j] = m.z;
          }
          double norm[3];
          const int rank = colpiv_qr_solve(A, b, 5, 3, norm);   // matA0.colPivHouseholderQr().solve(matB0), :435
          if (iter_cnt == 0) ++plane_rank_hist[rank];
          double nn = std::sqrt(norm[0] * norm[0] + norm[1] * norm[1] + norm[2] * norm[2]);
          double negative_OA_dot_norm = 1 / nn;
          for (int a = 0; a < 3; ++a) norm[a] /= nn;
          bool plane_valid = true;
          for (int j = 0; j < 5; ++j) {
            const Pt& m = surf_from_map_ds[nidx[j]];
            if (std::fabs(norm[0] * m.x + norm[1] * m.y + norm[2] * m.z + negative_OA_dot_norm) > P.plane_tol) { plane_valid = false; break; }
          }
          if (plane_valid) {
            Block B; B.type = BLK_PLANE;
            const Pt& cp = laser_surf_total_ds[i];
            B.cp[0] = cp.x; B.cp[1] = cp.y; B.cp[2] = cp.z;
            for (int a = 0; a < 3; ++a) { B.a[a] = norm[a]; B.b[a] = 0; B.c[a] = 0; }
            B.d = negative_OA_dot_norm;
            blocks.push_back(B);
            ++sc;
            if (iter_cnt == 0) surf_corr_q.push_back(i);
          }
        }
      }
      auto a1 = std::chrono::steady_clock::now();
      n_corner_corr = cc; n_surf_corr = sc;
      if (iter_cnt == 0) {
        blocks14.clear();
        for (const Block& B : blocks) {
          blocks14.push_back((double)B.type);
          for (int a = 0; a < 3; ++a) blocks14.push_back(B.cp[a]);
          for (int a = 0; a < 3; ++a) blocks14.push_back(B.a[a]);
          for (int a = 0; a < 3; ++a) blocks14.push_back(B.b[a]);
          for (int a = 0; a < 3; ++a) blocks14.push_back(B.c[a]);
          blocks14.push_back(B.d);
        }
      }
      CeresLike solver;
      SolveSummary s = solver.solve(blocks, params, P.lm_max_iters, P.huber_delta);
      if (iter_cnt < 2) { sums[iter_cnt] = s; std::memcpy(params_iter[iter_cnt], params, sizeof(params)); }
      auto a2 = std::chrono::steady_clock::now();
      t_assoc_ms += std::chrono::duration<double, std::milli>(a1 - a0).count();
      t_solve_ms += std::chrono::duration<double, std::milli>(a2 - a1).count();
    }
    optimized = true;
  }

  // saveKeyFramesAndFactor, :491-559, with the GTSAM part reduced to the no-loop-closure
  // pass-through: the iSAM2 estimate of the new node equals its
  // initial value, which is then stored in the f32 fields of PointXYZIRPYT.
  bool save_keyframes_and_factor() {
    if (!keyposes.empty()) {
      const KeyPose& pre = keyposes.back();
      if (std::pow(t_map2laser[0] - pre.x, 2) + std::pow(t_map2laser[1] - pre.y, 2) + std::pow(t_map2laser[2] - pre.z, 2) < P.min_keyframe_dist)
        return false;
    }
    double R[9];
    quat_to_mat(q_map2laser, R);
    // gtsam::Rot3::roll/pitch/yaw (ZYX Euler angles of the rotation matrix)
    double roll = std::atan2(R[7], R[8]);
    double pitch = std::atan2(-R[6], std::sqrt(R[7] * R[7] + R[8] * R[8]));
    double yaw = std::atan2(R[3], R[0]);
    KeyPose kp{(float)t_map2laser[0], (float)t_map2laser[1], (float)t_map2laser[2], (float)roll, (float)pitch, (float)yaw};
    keyposes.push_back(kp);
    params[0] = kp.x; params[1] = kp.y; params[2] = kp.z; params[3] = kp.roll; params[4] = kp.pitch; params[5] = kp.yaw;
    corner_frames.push_back(laser_corner_ds); surf_frames.push_back(laser_surf_ds); outlier_frames.push_back(laser_outlier_ds);
    return true;
  }

  void transform_update() {  // :481-489
    q_map2laser = quat_zyx(params[5], params[4], params[3]);
    t_map2laser[0] = params[0]; t_map2laser[1] = params[1]; t_map2laser[2] = params[2];
    q_map2odom = quat_mul(q_map2laser, quat_inverse(q_odom2laser));
    double r[3];
    quat_rotate(q_map2odom, t_odom2laser, r);
    for (int i = 0; i < 3; ++i) t_map2odom[i] = t_map2laser[i] - r[i];
  }

  // laserOdomHandler :154-166 + mainLoop gate :107-127.  Called once per scan for which LO
  // published odometry.  `pose_out` = /odom_aft_mapped (t xyz, q wxyz) as of the odom callback.
  void process(const std::vector<Pt>& corner_last, const std::vector<Pt>& surf_last, const std::vector<Pt>& outlier,
               const double t_odom[3], const Quat& q_odom, double pose_out[7]) {
    for (int i = 0; i < 3; ++i) t_odom2laser[i] = t_odom[i];
    q_odom2laser = q_odom;
    transform_associate_to_map();
    pose_out[0] = t_map2laser[0]; pose_out[1] = t_map2laser[1]; pose_out[2] = t_map2laser[2];
    pose_out[3] = q_map2laser.w; pose_out[4] = q_map2laser.x; pose_out[5] = q_map2laser.y; pose_out[6] = q_map2laser.z;
    ran_body = false; keyframe_added = false; optimized = false;
    t_map_ms = t_ds_ms = t_tree_ms = t_assoc_ms = t_solve_ms = 0;
    if (frame_cnt % P.lm_every == 0) {
      ran_body = true;
      auto t0 = std::chrono::steady_clock::now();
      transform_associate_to_map();
      extract_surrounding_keyframes();
      t_map_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      downsample_current_scan(corner_last, surf_last, outlier);
      scan2map_optimization();
      keyframe_added = save_keyframes_and_factor();
      transform_update();
    }
    ++frame_cnt;
  }
};

// ===========================================================================
// 8. Context + C API
// ===========================================================================
struct Ctx {
  Params P;
  ImageProjection ip;
  LaserOdometry lo;
  LaserMapping lm;
  long scans = 0;
  double odom_pose[7] = {0, 0, 0, 1, 0, 0, 0};  // /odom/lidar: t xyz, q wxyz
  double map_pose[7] = {0, 0, 0, 1, 0, 0, 0};   // /odom_aft_mapped
  double t_ip_ms = 0, t_lo_ms = 0, t_lm_ms = 0;
  std::vector<double> scratch_d;
  std::vector<int> scratch_i;
  std::vector<float> scratch_f;
};

template <class T>
int ret(const std::vector<T>& v, const void** p, int* n, int mult = 1) { *p = v.data(); *n = (int)v.size() * mult; return 0; }

}  // namespace

extern "C" {

enum { ORACLE_F32 = 0, ORACLE_F64 = 1, ORACLE_I32 = 2, ORACLE_U8 = 3 };

void* oracle_create(const alego_params* p) {
  Ctx* c = new Ctx();
  c->P = *p;
  c->ip.init(*p); c->lo.init(*p); c->lm.init(*p);
  return c;
}
void oracle_destroy(void* h) { delete (Ctx*)h; }

// a1-a6: one ImageProjection::pcCB
int oracle_ip(void* h, const alego_point* pts, int n) {
  Ctx* c = (Ctx*)h;
  auto t0 = std::chrono::steady_clock::now();
  c->ip.process(pts, n);
  c->t_ip_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  return (int)c->ip.seg_cloud.size();
}

// a7-a17: one LaserOdometry::mainLoop body on the last oracle_ip() result.
// Returns 1 when odometry was produced, 0 on the initialising scan.
int oracle_lo(void* h) {
  Ctx* c = (Ctx*)h;
  auto t0 = std::chrono::steady_clock::now();
  bool ok = c->lo.process(c->ip);
  c->t_lo_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  if (ok) {
    Quat q = mat_to_quat(c->lo.r_w);
    c->odom_pose[0] = c->lo.t_w[0]; c->odom_pose[1] = c->lo.t_w[1]; c->odom_pose[2] = c->lo.t_w[2];
    c->odom_pose[3] = q.w; c->odom_pose[4] = q.x; c->odom_pose[5] = q.y; c->odom_pose[6] = q.z;
  }
  return ok ? 1 : 0;
}

// sensor_msgs/Imu samples for the LO's IMU ring (imuHandler :761-802): smp[i] = stamp, orientation w x y z, linear_acceleration xyz,
// angular_velocity xyz (11 doubles); oracle_set_scan_time = the stamp of the next segmented cloud (t1, :111)
void oracle_push_imu(void* h, const double* smp, int n) {
  Ctx* c = (Ctx*)h;
  for (int i = 0; i < n; ++i) c->lo.imu_handler(smp + 11 * i);
}
void oracle_set_scan_time(void* h, double t) { ((Ctx*)h)->lo.scan_time = t; }

// feature extraction only (a7-a10), no odometry state change (config 1)
int oracle_fe(void* h) {
  Ctx* c = (Ctx*)h;
  c->lo.extract_features(c->ip);
  return (int)c->lo.less_flat.size();
}

// a18-a24: LaserMapping on the last LO outputs (call only when oracle_lo returned 1).
// Returns bit0 = mapping body ran, bit1 = optimisation ran, bit2 = key frame added.
int oracle_lm(void* h) {
  Ctx* c = (Ctx*)h;
  auto t0 = std::chrono::steady_clock::now();
  Quat q{c->odom_pose[3], c->odom_pose[4], c->odom_pose[5], c->odom_pose[6]};
  c->lm.process(c->lo.corner_last, c->lo.surf_last, c->ip.outlier_cloud, c->odom_pose, q, c->map_pose);
  c->t_lm_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  return (c->lm.ran_body ? 1 : 0) | (c->lm.optimized ? 2 : 0) | (c->lm.keyframe_added ? 4 : 0);
}

// LaserMapping fed with host clouds and an /odom/lidar pose (t xyz, q wxyz) — the checker's counterpart of alego_lm_process, for tests that construct the
// /corner_last, /surf_last, /outlier messages themselves (laserOdomHandler + one mainLoop body, laserMapping.cpp:102-131,154-166).
int oracle_lm_process(void* h, const alego_point* cl, int nc, const alego_point* sl, int ns, const alego_point* ol, int no, const double* odom7) {
  Ctx* c = (Ctx*)h;
  std::vector<Pt> vc(cl, cl + nc), vs(sl, sl + ns), vo(ol, ol + no);
  for (int i = 0; i < 7; ++i) c->odom_pose[i] = odom7[i];
  Quat q{odom7[3], odom7[4], odom7[5], odom7[6]};
  c->lm.process(vc, vs, vo, c->odom_pose, q, c->map_pose);
  return (c->lm.ran_body ? 1 : 0) | (c->lm.optimized ? 2 : 0) | (c->lm.keyframe_added ? 4 : 0);
}

// Full loop for one scan: IP -> LO -> LM.  stages: bit0 IP, bit1 LO, bit2 LM.
int oracle_process_scan(void* h, const alego_point* pts, int n, int stages) {
  Ctx* c = (Ctx*)h;
  int r = 0;
  if (stages & 1) oracle_ip(h, pts, n);
  if (stages & 2) {
    int ok = oracle_lo(h);
    r |= ok;
    if (ok && (stages & 4)) r |= oracle_lm(h) << 1;
  }
  ++c->scans;
  return r;
}

// cpu_pipe3: the reference's deployment shape — ImageProjection, LaserOdometry and LaserMapping as three
// threads (three nodelets of one manager, launch/test.launch:7-10) connected by message queues.  Every scan goes through
// all three stages (the ROS nodes drop frames when they fall behind; here the queues block, so the figure is the
// throughput of the slowest stage, with the hand-over copies a nodelet manager's shared_ptr messages avoid).
// Returns the wall-clock seconds for the n scans; the context ends in the same state as n oracle_process_scan calls.
}  // extern "C"
namespace {
template <class T>
struct BoundedQueue {
  std::mutex m; std::condition_variable cv_put, cv_get; std::deque<T> q; size_t cap = 4; bool closed = false;
  void put(T&& v) { std::unique_lock<std::mutex> l(m); cv_put.wait(l, [&] { return q.size() < cap; }); q.push_back(std::move(v)); cv_get.notify_one(); }
  bool get(T& v) { std::unique_lock<std::mutex> l(m); cv_get.wait(l, [&] { return !q.empty() || closed; }); if (q.empty()) return false; v = std::move(q.front()); q.pop_front(); cv_put.notify_one(); return true; }
  void close() { std::unique_lock<std::mutex> l(m); closed = true; cv_get.notify_all(); }
};
struct SegMsg {   // /segmented_cloud + /seg_info + /outlier
  std::vector<Pt> seg, outlier; std::vector<uint8_t> ground; std::vector<int> col, start, end; std::vector<float> range;
};
struct LoMsg {    // /corner_last + /surf_last + /outlier + /odom/lidar
  std::vector<Pt> corner, surf, outlier; double odom[7]; bool valid;
};
}  // namespace
extern "C" {
double oracle_run_pipelined(void* h, const alego_point* const* scans, const int* counts, int n) {
  Ctx* c = (Ctx*)h;
  BoundedQueue<SegMsg> q1;
  BoundedQueue<LoMsg> q2;
  const auto t0 = std::chrono::steady_clock::now();
  std::thread t_ip([&] {
    for (int k = 0; k < n; ++k) {
      c->ip.process(scans[k], counts[k]);
      SegMsg m;
      const int M = (int)c->ip.seg_cloud.size();
      m.seg = c->ip.seg_cloud; m.outlier = c->ip.outlier_cloud;
      m.ground.assign(c->ip.seg_ground.begin(), c->ip.seg_ground.begin() + M);
      m.col.assign(c->ip.seg_col.begin(), c->ip.seg_col.begin() + M);
      m.range.assign(c->ip.seg_range.begin(), c->ip.seg_range.begin() + M);
      m.start = c->ip.start_ring; m.end = c->ip.end_ring;
      q1.put(std::move(m));
    }
    q1.close();
  });
  std::thread t_lo([&] {
    ImageProjection view;   // LaserOdometry reads the message through the same members it reads in the sequential path
    view.P = c->P;
    SegMsg m;
    while (q1.get(m)) {
      view.seg_cloud.swap(m.seg); view.seg_ground.swap(m.ground); view.seg_col.swap(m.col); view.seg_range.swap(m.range);
      view.start_ring.swap(m.start); view.end_ring.swap(m.end);
      view.seg_ground.resize(view.seg_cloud.size() + 16); view.seg_col.resize(view.seg_cloud.size() + 16); view.seg_range.resize(view.seg_cloud.size() + 16);
      const bool ok = c->lo.process(view);
      LoMsg o;
      o.valid = ok;
      if (ok) {
        Quat q = mat_to_quat(c->lo.r_w);
        c->odom_pose[0] = c->lo.t_w[0]; c->odom_pose[1] = c->lo.t_w[1]; c->odom_pose[2] = c->lo.t_w[2];
        c->odom_pose[3] = q.w; c->odom_pose[4] = q.x; c->odom_pose[5] = q.y; c->odom_pose[6] = q.z;
        o.corner = c->lo.corner_last; o.surf = c->lo.surf_last; o.outlier.swap(m.outlier);
        std::memcpy(o.odom, c->odom_pose, sizeof(o.odom));
      }
      q2.put(std::move(o));
    }
    q2.close();
  });
  std::thread t_lm([&] {
    LoMsg o;
    while (q2.get(o)) {
      if (!o.valid) continue;
      Quat q{o.odom[3], o.odom[4], o.odom[5], o.odom[6]};
      c->lm.process(o.corner, o.surf, o.outlier, o.odom, q, c->map_pose);
    }
  });
  t_ip.join(); t_lo.join(); t_lm.join();
  c->scans += n;
  return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

// teacher forcing: overwrite LO params_ / LM params_ (6 doubles)
void oracle_set_lo_params(void* h, const double* p6) { std::memcpy(((Ctx*)h)->lo.params, p6, 48); }
void oracle_set_lm_params(void* h, const double* p6) { std::memcpy(((Ctx*)h)->lm.params, p6, 48); }

// ---- host pose-graph pass-through (laserMapping.cpp:561-584 correctPoses; the GTSAM side stays outside) ----
// correctPoses :569-578 for one frame: overwrite the stored key pose (f32 fields of PointXYZIRPYT)
void oracle_lm_set_keypose(void* h, int id, const float* pose6) {
  LaserMapping& lm = ((Ctx*)h)->lm;
  if (id < 0 || id >= (int)lm.keyposes.size()) return;
  lm.keyposes[id] = KeyPose{pose6[0], pose6[1], pose6[2], pose6[3], pose6[4], pose6[5]};
}
// correctPoses :563-565: recent_*_keyframes_.clear()
void oracle_lm_reset_window(void* h) {
  LaserMapping& lm = ((Ctx*)h)->lm;
  lm.recent_corner.clear(); lm.recent_surf.clear(); lm.recent_outlier.clear();
}
// correctPoses :579-580 with correction_ = [R | c] (row-major 3x4)
void oracle_lm_apply_correction(void* h, const double* rc) {
  LaserMapping& lm = ((Ctx*)h)->lm;
  double R[9], M[9], t[3];
  quat_to_mat(lm.q_map2odom, R);
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) M[i * 3 + j] = rc[i * 4 + 0] * R[0 * 3 + j] + rc[i * 4 + 1] * R[1 * 3 + j] + rc[i * 4 + 2] * R[2 * 3 + j];
  for (int i = 0; i < 3; ++i) t[i] = rc[i * 4 + 0] * lm.t_map2odom[0] + rc[i * 4 + 1] * lm.t_map2odom[1] + rc[i * 4 + 2] * lm.t_map2odom[2] + rc[i * 4 + 3];
  lm.q_map2odom = mat_to_quat(M);
  for (int i = 0; i < 3; ++i) lm.t_map2odom[i] = t[i];
}
// saveKeyFramesAndFactor :531-555 with a host-provided pose and clouds
void oracle_lm_add_keyframe(void* h, const float* pose6, const alego_point* c, int nc, const alego_point* s, int ns, const alego_point* o, int no) {
  LaserMapping& lm = ((Ctx*)h)->lm;
  lm.keyposes.push_back(KeyPose{pose6[0], pose6[1], pose6[2], pose6[3], pose6[4], pose6[5]});
  lm.corner_frames.emplace_back(c, c + nc); lm.surf_frames.emplace_back(s, s + ns); lm.outlier_frames.emplace_back(o, o + no);
}
// key frame `id` as stored by saveKeyFramesAndFactor: kind 0 corner_frames_, 1 surf_frames_, 2 outlier_frames_
int oracle_lm_keyframe(void* h, int id, int kind, const void** ptr, int* n_points) {
  LaserMapping& lm = ((Ctx*)h)->lm;
  if (id < 0 || id >= (int)lm.keyposes.size() || kind < 0 || kind > 2) return -1;
  const std::vector<Pt>& v = kind == 0 ? lm.corner_frames[id] : (kind == 1 ? lm.surf_frames[id] : lm.outlier_frames[id]);
  *ptr = v.data(); *n_points = (int)v.size();
  return 0;
}

// Generic read-only access to the last scan's intermediates.  count = number of scalars.
int oracle_get(void* h, const char* name, const void** ptr, int* count, int* dtype) {
  Ctx* c = (Ctx*)h;
  std::string s(name);
  ImageProjection& ip = c->ip; LaserOdometry& lo = c->lo; LaserMapping& lm = c->lm;
  auto cloud = [&](const std::vector<Pt>& v) { *ptr = v.data(); *count = (int)v.size() * 4; *dtype = ORACLE_F32; return 0; };
  const int M = (int)ip.seg_cloud.size();
  if (s == "range_img") { *dtype = ORACLE_F32; return ret(ip.out_range_img, ptr, count); }
  if (s == "label_img") { *dtype = ORACLE_I32; return ret(ip.out_label_img, ptr, count); }
  if (s == "ground_img") { *dtype = ORACLE_U8; return ret(ip.out_ground_img, ptr, count); }
  if (s == "seg_cloud") return cloud(ip.seg_cloud);
  if (s == "undistorted") return cloud(lo.undistorted);
  if (s == "imu_ptr") {
    static thread_local int iptr[3];
    iptr[0] = lo.imu_ptr_last; iptr[1] = lo.imu_ptr_front; iptr[2] = lo.imu_ptr_last_iter;
    *dtype = ORACLE_I32; *ptr = iptr; *count = 3; return 0;
  }
  if (s == "imu_ring") {   // [200][10]: time, roll, pitch, yaw, shift xyz, velo xyz
    static thread_local double ring[LaserOdometry::IMU_Q * 10];
    for (int i = 0; i < LaserOdometry::IMU_Q; ++i) {
      double* r = ring + i * 10;
      r[0] = lo.imu_time[i]; r[1] = lo.imu_roll[i]; r[2] = lo.imu_pitch[i]; r[3] = lo.imu_yaw[i];
      for (int k = 0; k < 3; ++k) { r[4 + k] = lo.imu_shift[k][i]; r[7 + k] = lo.imu_velo[k][i]; }
    }
    *dtype = ORACLE_F64; *ptr = ring; *count = LaserOdometry::IMU_Q * 10; return 0;
  }
  if (s == "outlier") return cloud(ip.outlier_cloud);
  if (s == "seg_ground") { *dtype = ORACLE_U8; *ptr = ip.seg_ground.data(); *count = M; return 0; }
  if (s == "seg_col") { *dtype = ORACLE_I32; *ptr = ip.seg_col.data(); *count = M; return 0; }
  if (s == "seg_range") { *dtype = ORACLE_F32; *ptr = ip.seg_range.data(); *count = M; return 0; }
  if (s == "ring_start") { *dtype = ORACLE_I32; return ret(ip.start_ring, ptr, count); }
  if (s == "ring_end") { *dtype = ORACLE_I32; return ret(ip.end_ring, ptr, count); }
  if (s == "orientation") { *dtype = ORACLE_F32; *ptr = ip.ori; *count = 3; return 0; }
  if (s == "curv_d") { *dtype = ORACLE_F32; return ret(lo.curv_d, ptr, count); }
  if (s == "picked_occl") { *dtype = ORACLE_U8; return ret(lo.picked_occl, ptr, count); }
  if (s == "point_label") { *dtype = ORACLE_I32; return ret(lo.out_label, ptr, count); }
  if (s == "sharp") return cloud(lo.sharp);
  if (s == "less_sharp") return cloud(lo.less_sharp);
  if (s == "flat") return cloud(lo.flat);
  if (s == "less_flat") return cloud(lo.less_flat);
  if (s == "sharp_idx") { *dtype = ORACLE_I32; return ret(lo.sharp_idx, ptr, count); }
  if (s == "less_sharp_idx") { *dtype = ORACLE_I32; return ret(lo.less_sharp_idx, ptr, count); }
  if (s == "flat_idx") { *dtype = ORACLE_I32; return ret(lo.flat_idx, ptr, count); }
  if (s == "surf_last") return cloud(lo.surf_last);
  if (s == "corner_last") return cloud(lo.corner_last);
  if (s == "lo_surf_corr") { *dtype = ORACLE_I32; return ret(lo.surf_corr, ptr, count); }
  if (s == "lo_corner_corr") { *dtype = ORACLE_I32; return ret(lo.corner_corr, ptr, count); }
  if (s == "lo_params") { *dtype = ORACLE_F64; *ptr = lo.params; *count = 6; return 0; }
  if (s == "lo_params_after_surf") { *dtype = ORACLE_F64; *ptr = lo.params_after_surf; *count = 6; return 0; }
  if (s == "lo_t_w") { *dtype = ORACLE_F64; *ptr = lo.t_w; *count = 3; return 0; }
  if (s == "lo_r_w") { *dtype = ORACLE_F64; *ptr = lo.r_w; *count = 9; return 0; }
  if (s == "odom_pose") { *dtype = ORACLE_F64; *ptr = c->odom_pose; *count = 7; return 0; }
  if (s == "map_pose") { *dtype = ORACLE_F64; *ptr = c->map_pose; *count = 7; return 0; }
  if (s == "lo_solve_info") {
    c->scratch_i = {lo.sum_surf.iterations, lo.sum_surf.successful, lo.sum_surf.termination,
                    lo.sum_corner.iterations, lo.sum_corner.successful, lo.sum_corner.termination,
                    lo.n_surf_corr, lo.n_corner_corr};
    *dtype = ORACLE_I32; return ret(c->scratch_i, ptr, count);
  }
  if (s == "lo_costs") {
    c->scratch_d = {lo.sum_surf.initial_cost, lo.sum_surf.final_cost, lo.sum_corner.initial_cost, lo.sum_corner.final_cost};
    *dtype = ORACLE_F64; return ret(c->scratch_d, ptr, count);
  }
  if (s == "lm_params") { *dtype = ORACLE_F64; *ptr = lm.params; *count = 6; return 0; }
  if (s == "lm_params_iter") { *dtype = ORACLE_F64; *ptr = lm.params_iter; *count = 12; return 0; }
  if (s == "lm_corner_map_ds") return cloud(lm.corner_from_map_ds);
  if (s == "lm_surf_map_ds") return cloud(lm.surf_from_map_ds);
  if (s == "lm_corner_map") return cloud(lm.corner_from_map);
  if (s == "lm_surf_map") return cloud(lm.surf_from_map);
  if (s == "lm_corner_ds") return cloud(lm.laser_corner_ds);
  if (s == "lm_surf_ds") return cloud(lm.laser_surf_ds);
  if (s == "lm_outlier_ds") return cloud(lm.laser_outlier_ds);
  if (s == "lm_surf_total_ds") return cloud(lm.laser_surf_total_ds);
  if (s == "lm_corner_corr_q") { *dtype = ORACLE_I32; return ret(lm.corner_corr_q, ptr, count); }
  if (s == "lm_surf_corr_q") { *dtype = ORACLE_I32; return ret(lm.surf_corr_q, ptr, count); }
  if (s == "lm_plane_rank_hist") { *dtype = ORACLE_I32; return ret(lm.plane_rank_hist, ptr, count); }
  if (s == "lm_blocks14") { *dtype = ORACLE_F64; return ret(lm.blocks14, ptr, count); }
  if (s == "lm_keyposes") { *dtype = ORACLE_F32; *ptr = lm.keyposes.data(); *count = (int)lm.keyposes.size() * 6; return 0; }
  if (s == "lm_map2odom") {
    c->scratch_d = {lm.t_map2odom[0], lm.t_map2odom[1], lm.t_map2odom[2], lm.q_map2odom.w, lm.q_map2odom.x, lm.q_map2odom.y, lm.q_map2odom.z};
    *dtype = ORACLE_F64; return ret(c->scratch_d, ptr, count);
  }
  if (s == "lm_info") {
    c->scratch_i = {lm.ran_body, lm.optimized, lm.keyframe_added, lm.n_corner_corr, lm.n_surf_corr,
                    lm.sums[0].iterations, lm.sums[0].successful, lm.sums[0].termination,
                    lm.sums[1].iterations, lm.sums[1].successful, lm.sums[1].termination, (int)lm.keyposes.size()};
    *dtype = ORACLE_I32; return ret(c->scratch_i, ptr, count);
  }
  if (s == "timing_ms") {
    c->scratch_d = {c->t_ip_ms, c->t_lo_ms, c->t_lm_ms, lo.t_fe_ms, lo.t_assoc_ms, lo.t_solve_ms,
                    lm.t_map_ms, lm.t_ds_ms, lm.t_tree_ms, lm.t_assoc_ms, lm.t_solve_ms};
    *dtype = ORACLE_F64; return ret(c->scratch_d, ptr, count);
  }
  return -1;
}

// ---- stand-alone pieces for unit tests ----
float oracle_atan2f(float y, float x) { return omath::o_atan2f(y, x); }
float oracle_hypotf(float x, float y) { return omath::o_hypotf(x, y); }
void oracle_atan2f_array(const float* y, const float* x, float* out, int n) { for (int i = 0; i < n; ++i) out[i] = omath::o_atan2f(y[i], x[i]); }
void oracle_hypotf_array(const float* x, const float* y, float* out, int n) { for (int i = 0; i < n; ++i) out[i] = omath::o_hypotf(x[i], y[i]); }
void oracle_libm_atan2f_array(const float* y, const float* x, float* out, int n) { for (int i = 0; i < n; ++i) out[i] = std::atan2(y[i], x[i]); }
void oracle_libm_hypotf_array(const float* x, const float* y, float* out, int n) { for (int i = 0; i < n; ++i) out[i] = std::hypot(x[i], y[i]); }
// mode 0: restated sinf, 1: restated cosf, 2: this host's libm sinf, 3: libm cosf
void oracle_sincosf_array(const float* x, float* out, int n, int mode) {
  for (int i = 0; i < n; ++i) out[i] = mode == 0 ? omath::o_sinf(x[i]) : mode == 1 ? omath::o_cosf(x[i]) : mode == 2 ? std::sin(x[i]) : std::cos(x[i]);
}

// std::sort itself (this container's libstdc++) on idx = 0..n-1 with the reference's kind of comparator (keys only):
// laserOdometry.cpp:185.  depth_limit >= 0 calls the two phases of std::sort (bits/stl_algo.h) with that depth limit instead of
// 2 floor(log2 n), so that the heap-sort branch of __introsort_loop can be reached without adversarial input.
void oracle_std_sort_order(const uint32_t* keys, int n, int depth_limit, int* order) {
  std::vector<int> idx(n);
  for (int i = 0; i < n; ++i) idx[i] = i;
  auto cmp = [keys](int a, int b) { return keys[a] < keys[b]; };
  if (depth_limit < 0) std::sort(idx.begin(), idx.end(), cmp);
  else if (n > 0) {
    std::__introsort_loop(idx.begin(), idx.end(), (long)depth_limit, __gnu_cxx::__ops::__iter_comp_iter(cmp));
    std::__final_insertion_sort(idx.begin(), idx.end(), __gnu_cxx::__ops::__iter_comp_iter(cmp));
  }
  for (int i = 0; i < n; ++i) order[i] = idx[i];
}

int oracle_voxel_grid(const alego_point* in, int n, float leaf, int sort_mode, alego_point* out, int cap) {
  std::vector<Pt> v(in, in + n), o;
  voxel_grid(v, leaf, o, sort_mode);
  int m = std::min((int)o.size(), cap);
  std::memcpy(out, o.data(), sizeof(Pt) * m);
  return (int)o.size();
}

// type: 0 surf, 1 corner, 2 edge, 3 plane; geom = cp[3], a[3], b[3], c[3], d  (13 doubles)
void oracle_eval_block(int type, const double* geom, const double* params, double* res, double* J6) {
  Block B; B.type = type;
  std::memcpy(B.cp, geom, 24); std::memcpy(B.a, geom + 3, 24); std::memcpy(B.b, geom + 6, 24); std::memcpy(B.c, geom + 9, 24);
  B.d = geom[12];
  eval_block(B, params, res, J6);
}

// Solve on caller-provided blocks (n x 14 doubles: type, geom[13]); returns iterations.
int oracle_solve(const double* blocks14, int n, double* params6, int max_iter, double huber, double* costs2) {
  std::vector<Block> bl(n);
  for (int i = 0; i < n; ++i) {
    const double* g = blocks14 + (size_t)i * 14;
    bl[i].type = (int)g[0];
    std::memcpy(bl[i].cp, g + 1, 24); std::memcpy(bl[i].a, g + 4, 24); std::memcpy(bl[i].b, g + 7, 24); std::memcpy(bl[i].c, g + 10, 24);
    bl[i].d = g[13];
  }
  CeresLike s;
  SolveSummary sum = s.solve(bl, params6, max_iter, huber);
  if (costs2) { costs2[0] = sum.initial_cost; costs2[1] = sum.final_cost; }
  return sum.iterations | (sum.termination << 8) | (sum.successful << 16);
}

int oracle_knn(const alego_point* cloud, int n, const alego_point* q, int nq, int k, int* idx, float* dist) {
  std::vector<Pt> v(cloud, cloud + n);
  KdTree t; t.build(v);
  for (int i = 0; i < nq; ++i) {
    int f = t.knn(q[i], k, idx + (size_t)i * k, dist + (size_t)i * k);
    for (int j = f; j < k; ++j) { idx[(size_t)i * k + j] = -1; dist[(size_t)i * k + j] = -1.f; }
  }
  return 0;
}

void oracle_eig3(const double* A9, double* lam3, double* V9) { eig3(A9, lam3, V9); }
// Eigen::ColPivHouseholderQR compute + solve of an m x n column-major system (m, n <= 8); A is overwritten.  Returns nonzeroPivots().
int oracle_colpiv_qr_solve(double* A, const double* b, int m, int n, double* x) { return (m < 1 || n < 1 || m > 8 || n > 8) ? -1 : colpiv_qr_solve(A, b, m, n, x); }

// ---- loop closure (src/laserMapping.cpp:652-824) ------------------------------------------------------------------------------
// detectLoopClosure :760-790: the key pose nearest to the current position (radius search, ascending distance) that is more than
// lc_min_time_gap older than the newest key frame; -1 if there is none.  keyposes6 = n x (x y z roll pitch yaw) f32, times = n stamps.
int oracle_loop_detect(const alego_params* P, const float* keyposes6, const double* times, int n, const double* cur_xyz) {
  if (n <= 0) return -1;
  const Pt cur{(float)cur_xyz[0], (float)cur_xyz[1], (float)cur_xyz[2], 0.f};
  std::vector<std::pair<float, int>> cand;
  const float r2 = (float)(P->lc_search_radius * P->lc_search_radius);
  for (int i = 0; i < n; ++i) {
    const Pt kp{keyposes6[i * 6 + 0], keyposes6[i * 6 + 1], keyposes6[i * 6 + 2], 0.f};
    const float d = dist2_f32(kp, cur);
    if (d < r2) cand.emplace_back(d, i);   // flann RadiusResultSet: dist < radius^2
  }
  std::sort(cand.begin(), cand.end());
  for (const auto& c : cand)
    if (times[n - 1] - times[c.second] > P->lc_min_time_gap) return c.second;
  return -1;
}

// performLoopClosure :670-697 on caller-provided key frames.  poses6 = (1 + nh) x 6 f32 (the newest key frame first, then the
// history frames closest_history_frame_id_ - 25 .. + 25 in ascending order); pts = their clouds back to back in the order corner,
// surf, outlier per frame with offs[(1 + nh) * 3 + 1] prefix offsets.  out[0..] = converged, iterations, n_source, n_target (as
// doubles), fitness, then the 16 entries of getFinalTransformation() (row-major, f32 values).  target_out (may be NULL, cap points)
// receives near_history_keyframes_.
int oracle_loop_icp(const alego_params* P, const float* poses6, const alego_point* pts, const int* offs, int nh, double* out21,
                    alego_point* target_out, int target_cap) {
  auto frame = [&](int f, int kind, std::vector<Pt>& dst) {   // transformPointCloud(frames_[f], pose[f]) appended
    std::vector<Pt> in(pts + offs[f * 3 + kind], pts + offs[f * 3 + kind + 1]), tr;
    const float* kp = poses6 + f * 6;
    LaserMapping::transform_cloud(in, KeyPose{kp[0], kp[1], kp[2], kp[3], kp[4], kp[5]}, tr);
    dst.insert(dst.end(), tr.begin(), tr.end());
  };
  std::vector<Pt> src, raw, tgt;
  frame(0, 1, src); frame(0, 0, src); frame(0, 2, src);                                   // :794-796: surf, corner, outlier
  for (int f = 1; f <= nh; ++f) { frame(f, 1, raw); frame(f, 0, raw); frame(f, 2, raw); }   // :805-807
  voxel_grid(raw, P->lc_leaf, tgt, P->sort_mode);                                            // :811-812
  if (target_out) std::memcpy(target_out, tgt.data(), sizeof(Pt) * std::min((size_t)target_cap, tgt.size()));
  for (int k = 0; k < 21; ++k) out21[k] = 0;
  out21[2] = (double)src.size(); out21[3] = (double)tgt.size();
  float Tf[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};   // final_transformation_ (Matrix4f)
  auto finish = [&](int conv, int it, double fit) { out21[0] = conv; out21[1] = it; out21[4] = fit; for (int k = 0; k < 16; ++k) out21[5 + k] = Tf[k]; return 0; };
  if (src.empty() || tgt.empty()) return finish(0, 0, DBL_MAX);
  KdTree kd; kd.build(tgt);
  std::vector<Pt> cur = src;   // input_transformed
  const double max_d2 = P->icp_max_corr_dist * P->icp_max_corr_dist;
  double prev_mse = DBL_MAX;
  int it = 0, converged = 0;
  while (true) {
    double S[15] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, mse = 0;
    long n = 0;
    for (const Pt& p : cur) {   // determineCorrespondences: 1-NN in the target, kept within the maximum correspondence distance
      int idx; float d2;
      if (kd.knn(p, 1, &idx, &d2) < 1 || (double)d2 > max_d2) continue;
      const Pt& q = tgt[idx];
      const double a[3] = {p.x, p.y, p.z}, b[3] = {q.x, q.y, q.z};
      for (int k = 0; k < 3; ++k) { S[k] += a[k]; S[3 + k] += b[k]; }
      for (int u = 0; u < 3; ++u) for (int v = 0; v < 3; ++v) S[6 + u * 3 + v] += a[u] * b[v];
      mse += (double)d2;
      ++n;
    }
    if (n < 3) { converged = 0; break; }   // "Not enough correspondences found"
    mse /= (double)n;
    double RT[12];
    oicp::horn_transform(S, (double)n, RT);
    float M[16] = {(float)RT[0], (float)RT[1], (float)RT[2], (float)RT[3], (float)RT[4], (float)RT[5], (float)RT[6], (float)RT[7],
                   (float)RT[8], (float)RT[9], (float)RT[10], (float)RT[11], 0, 0, 0, 1};   // transformation_ (Matrix4f)
    for (Pt& p : cur) {   // transformCloud (f32)
      const float x = p.x, y = p.y, z = p.z;
      p.x = M[0] * x + M[1] * y + M[2] * z + M[3]; p.y = M[4] * x + M[5] * y + M[6] * z + M[7]; p.z = M[8] * x + M[9] * y + M[10] * z + M[11];
    }
    float N[16];   // final_transformation_ = transformation_ * final_transformation_
    for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) N[r * 4 + c] = M[r * 4 + 0] * Tf[0 * 4 + c] + M[r * 4 + 1] * Tf[1 * 4 + c] + M[r * 4 + 2] * Tf[2 * 4 + c] + M[r * 4 + 3] * Tf[3 * 4 + c];
    std::memcpy(Tf, N, sizeof(N));
    ++it;
    // DefaultConvergenceCriteria::hasConverged (absolute MSE 1e-12: PCL's default; rotation threshold = 1 - transformation_epsilon_ as
    // IterativeClosestPoint::computeTransformation sets it in PCL 1.8, 0.999999 with laserMapping.cpp:673)
    if (it >= P->icp_max_iters) { converged = 1; break; }
    const double cos_angle = 0.5 * ((double)M[0] + (double)M[5] + (double)M[10] - 1.0);
    const double tr2 = (double)M[3] * M[3] + (double)M[7] * M[7] + (double)M[11] * M[11];
    if (cos_angle >= 1.0 - P->icp_trans_eps && tr2 <= P->icp_trans_eps) { converged = 1; break; }
    if (std::fabs(mse - prev_mse) < 1e-12) { converged = 1; break; }
    if (std::fabs(mse - prev_mse) / prev_mse < P->icp_fitness_eps) { converged = 1; break; }
    prev_mse = mse;
  }
  // getFitnessScore(): the source under the final transformation against the target
  double fit = 0; long nf = 0;
  for (const Pt& p0 : src) {
    Pt p;
    p.x = Tf[0] * p0.x + Tf[1] * p0.y + Tf[2] * p0.z + Tf[3]; p.y = Tf[4] * p0.x + Tf[5] * p0.y + Tf[6] * p0.z + Tf[7]; p.z = Tf[8] * p0.x + Tf[9] * p0.y + Tf[10] * p0.z + Tf[11];
    p.intensity = p0.intensity;
    int idx; float d2;
    if (kd.knn(p, 1, &idx, &d2) == 1) { fit += (double)d2; ++nf; }
  }
  return finish(converged, it, nf ? fit / (double)nf : DBL_MAX);
}

// Normal equations of a set of residual blocks at params6 with HuberLoss(huber) + corrector: out28 = upper triangle of J^T J (21, row by
// row), J^T r (6), cost (1) — what one rank of a sharded registration contributes to the all-reduce.
void oracle_normal_eq(const double* blocks14, int n, const double* params6, double huber, double* out28) {
  for (int k = 0; k < 28; ++k) out28[k] = 0;
  const double b2 = huber * huber;
  for (int i = 0; i < n; ++i) {
    const double* g = blocks14 + (size_t)i * 14;
    Block B; B.type = (int)g[0];
    std::memcpy(B.cp, g + 1, 24); std::memcpy(B.a, g + 4, 24); std::memcpy(B.b, g + 7, 24); std::memcpy(B.c, g + 10, 24); B.d = g[13];
    double r, J[6];
    eval_block(B, params6, &r, J);
    const double s = r * r;
    double rho0, rho1;
    if (s > b2) { const double rr = std::sqrt(s); rho0 = 2.0 * huber * rr - b2; rho1 = std::max(DBL_MIN, huber / rr); } else { rho0 = s; rho1 = 1.0; }
    const double sq = std::sqrt(rho1);
    for (int k = 0; k < 6; ++k) J[k] *= sq;
    r *= sq;
    int t = 0;
    for (int a = 0; a < 6; ++a) for (int b = a; b < 6; ++b) out28[t++] += J[a] * J[b];
    for (int k = 0; k < 6; ++k) out28[21 + k] += J[k] * r;
    out28[27] += 0.5 * rho0;
  }
}

// transformToStart (laserOdometry.cpp:728-740) with params_ = params6
void oracle_transform_to_start(const double* params6, const alego_point* in, int n, alego_point* out) {
  LaserOdometry lo;
  std::memcpy(lo.params, params6, 48);
  for (int i = 0; i < n; ++i) lo.transform_to_start(in[i], out[i]);
}
// transformPointCloud (laserMapping.h:164-177) by an f32 key pose (x y z roll pitch yaw)
void oracle_transform_cloud(const float* pose6, const alego_point* in, int n, alego_point* out) {
  std::vector<Pt> v(in, in + n), o;
  LaserMapping::transform_cloud(v, KeyPose{pose6[0], pose6[1], pose6[2], pose6[3], pose6[4], pose6[5]}, o);
  if (n) std::memcpy(out, o.data(), sizeof(Pt) * (size_t)n);
}

}  // extern "C"
