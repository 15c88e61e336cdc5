// oracle_icp.h — TEST INFRASTRUCTURE (parity oracle): loop-closure detection and the PCL ICP of
// LaserMapping::performLoopClosure (src/laserMapping.cpp:652-824), restated.  PARITY UNPINNED like the rest of the oracle; on
// top of that pcl::IterativeClosestPoint / TransformationEstimationSVD / DefaultConvergenceCriteria are [upstream] (PCL 1.8,
// un-vendored): restated from their published algorithm with two declared choices — the rigid transform of an iteration is Horn's
// closed form (largest eigenvector of the 4x4 quaternion matrix, f64 sums in input order) instead of Eigen::umeyama in f32, whose
// summation order is Eigen's own, and nearest-neighbour ties go to the lowest index.
#ifndef ORACLE_ICP_H_
#define ORACLE_ICP_H_
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <vector>

namespace oicp {

// symmetric n x n eigen-decomposition, cyclic Jacobi; V column c = eigenvector c (unsorted)
template <int N>
inline void jacobi_eig(double A[N][N], double V[N][N], double lam[N]) {
  for (int i = 0; i < N; ++i) for (int j = 0; j < N; ++j) V[i][j] = i == j ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 60; ++sweep) {
    double off = 0;
    for (int p = 0; p < N; ++p) for (int q = p + 1; q < N; ++q) off += A[p][q] * A[p][q];
    if (off < 1e-300) break;
    for (int p = 0; p < N - 1; ++p)
      for (int q = p + 1; q < N; ++q) {
        const double apq = A[p][q];
        if (apq == 0.0) continue;
        const double theta = (A[q][q] - A[p][p]) / (2.0 * apq);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < N; ++k) { const double akp = A[k][p], akq = A[k][q]; A[k][p] = c * akp - s * akq; A[k][q] = s * akp + c * akq; }
        for (int k = 0; k < N; ++k) { const double apk = A[p][k], aqk = A[q][k]; A[p][k] = c * apk - s * aqk; A[q][k] = s * apk + c * aqk; }
        for (int k = 0; k < N; ++k) { const double vkp = V[k][p], vkq = V[k][q]; V[k][p] = c * vkp - s * vkq; V[k][q] = s * vkp + c * vkq; }
      }
  }
  for (int i = 0; i < N; ++i) lam[i] = A[i][i];
}

// Least-squares rigid transform tgt ~ R src + t from the sums of n correspondences (Horn 1987): S[0..2] = sum src, S[3..5] = sum tgt,
// S[6..14] = sum src_a * tgt_b (row-major a, b).  Output row-major 3x4 [R | t] in double.
inline void horn_transform(const double S[15], double n, double RT[12]) {
  const double ms[3] = {S[0] / n, S[1] / n, S[2] / n}, mt[3] = {S[3] / n, S[4] / n, S[5] / n};
  double M[3][3];
  for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) M[a][b] = S[6 + a * 3 + b] - n * ms[a] * mt[b];   // sum (src - ms)_a (tgt - mt)_b
  double Nq[4][4] = {{M[0][0] + M[1][1] + M[2][2], M[1][2] - M[2][1], M[2][0] - M[0][2], M[0][1] - M[1][0]},
                     {M[1][2] - M[2][1], M[0][0] - M[1][1] - M[2][2], M[0][1] + M[1][0], M[2][0] + M[0][2]},
                     {M[2][0] - M[0][2], M[0][1] + M[1][0], -M[0][0] + M[1][1] - M[2][2], M[1][2] + M[2][1]},
                     {M[0][1] - M[1][0], M[2][0] + M[0][2], M[1][2] + M[2][1], -M[0][0] - M[1][1] + M[2][2]}};
  double V[4][4], lam[4];
  jacobi_eig<4>(Nq, V, lam);
  int best = 0;
  for (int i = 1; i < 4; ++i) if (lam[i] > lam[best]) best = i;
  double q[4] = {V[0][best], V[1][best], V[2][best], V[3][best]};
  if (q[0] < 0) for (int i = 0; i < 4; ++i) q[i] = -q[i];
  const double nn = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  const double w = q[0] / nn, x = q[1] / nn, y = q[2] / nn, z = q[3] / nn;
  double R[9] = {1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
                 2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                 2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)};
  for (int a = 0; a < 3; ++a) {
    for (int b = 0; b < 3; ++b) RT[a * 4 + b] = R[a * 3 + b];
    RT[a * 4 + 3] = mt[a] - (R[a * 3 + 0] * ms[0] + R[a * 3 + 1] * ms[1] + R[a * 3 + 2] * ms[2]);
  }
}

}  // namespace oicp
#endif
