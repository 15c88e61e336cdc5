// examples/replay.cpp — the C ABI used from plain C++ (no Python, no ROS): what a host program that links libalego_mi355x.so looks like.
//
// Replays `n` scans of the synthetic S0/T0 stream (libalego_synth.so stands in for the bag reader) through
// ImageProjection -> LaserOdometry -> LaserMapping, one alego_scan_process call per scan as a single nodelet manager would
// (launch/test.launch:6-10), pulls every new key frame across the boundary the way the reference's pose-graph thread reads
// cloud_keyposes_6d_ (laserMapping.cpp:586-596), and prints one JSON line with the final poses.
//
//   g++ -O2 -std=c++17 -Iinclude examples/replay.cpp -o examples/replay -La-lego-loam_amd -lalego_mi355x -lalego_synth
//       -Wl,-rpath,'$ORIGIN/../a-lego-loam_amd'                                  (__graft_entry__.build() does this)
//   examples/replay [n_scans] [n_scan] [horizon_scan]
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "alego_mi355x.h"

extern "C" int alego_synth_scan(const alego_params* P, int stream, long scan_index, int flags, alego_point* out, int cap);

int main(int argc, char** argv) {
  const int n_scans = argc > 1 ? std::atoi(argv[1]) : 40;
  const int n_scan = argc > 2 ? std::atoi(argv[2]) : 16, horizon = argc > 3 ? std::atoi(argv[3]) : 1800;
  alego_params P;
  alego_default_params(&P, n_scan, horizon);
  if (alego_params_sizeof() != (int)sizeof(alego_params)) { std::fprintf(stderr, "header / library mismatch\n"); return 2; }
  alego_handle* h = nullptr;
  if (int rc = alego_create(&P, /*device*/ 0, /*slots*/ 1, /*ring*/ 1, &h)) {
    std::fprintf(stderr, "alego_create failed (%d): there is no CPU fallback, an MI355X is required\n", rc);
    return 1;
  }
  const int N = P.n_scan * P.horizon_scan;
  std::vector<alego_point> pts(N), kc(N), ks(N), ko(N);
  alego_pose odom{}, mapped{};
  int key_frames = 0, last_flags = 0;
  float last_key_pose[6] = {0, 0, 0, 0, 0, 0};
  for (int k = 0; k < n_scans; ++k) {
    const int n = alego_synth_scan(&P, 0, k, 0, pts.data(), N);
    alego_scan_in in{pts.data(), n, 0.1 * k};
    const int flags = alego_scan_process(h, 0, &in, /*IP | LO | LM*/ 7, nullptr, nullptr, &odom, &mapped);
    if (flags < 0) { std::fprintf(stderr, "scan %d: %s\n", k, alego_last_error(h)); alego_destroy(h); return 1; }
    last_flags = flags;
    if (flags & ALEGO_FLAG_LM_KEYFRAME) {   // saveKeyFramesAndFactor stored a frame: fetch it as the pose-graph thread would
      alego_keyframe kf{};
      kf.corner = kc.data(); kf.corner_cap = N; kf.surf = ks.data(); kf.surf_cap = N; kf.outlier = ko.data(); kf.outlier_cap = N;
      if (alego_lm_get_keyframe(h, 0, -1, &kf) < 0) { std::fprintf(stderr, "get_keyframe: %s\n", alego_last_error(h)); alego_destroy(h); return 1; }
      ++key_frames;
      for (int i = 0; i < 6; ++i) last_key_pose[i] = kf.pose[i];
    }
  }
  std::printf("{\"scans\": %d, \"flags\": %d, \"key_frames\": %d, \"resident_key_frames\": %d, "
              "\"odom_t\": [%.17g, %.17g, %.17g], \"map_t\": [%.17g, %.17g, %.17g], \"map_params\": [%.17g, %.17g, %.17g, %.17g, %.17g, %.17g], "
              "\"last_key_pose\": [%.9g, %.9g, %.9g, %.9g, %.9g, %.9g]}\n",
              n_scans, last_flags, key_frames, alego_lm_keyframe_count(h, 0), odom.t[0], odom.t[1], odom.t[2], mapped.t[0], mapped.t[1], mapped.t[2],
              mapped.params[0], mapped.params[1], mapped.params[2], mapped.params[3], mapped.params[4], mapped.params[5],
              last_key_pose[0], last_key_pose[1], last_key_pose[2], last_key_pose[3], last_key_pose[4], last_key_pose[5]);
  alego_destroy(h);
  return 0;
}
