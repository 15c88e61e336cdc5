// examples/replay.cpp — the C ABI used from plain C++ (no Python, no ROS): what a host program that links libalego_mi355x.so looks like.
//
// Two sources of scans:
//   examples/replay [n_scans] [n_scan] [horizon_scan]
//       the synthetic S0/T0 stream (libalego_synth.so stands in for the sensor);
//   examples/replay --bag FILE.bag [--topic /lslidar_point_cloud] [--list] [--scans N] [--n-scan 16] [--horizon 4000] [--standalone]
//       a recorded rosbag (format 2.0; uncompressed, bz2 or lz4 chunks) read by the library's own reader: what
//       `rosbag play test_0515.bag` + `roslaunch alego test2.launch` do in the reference (README.md:33-37, launch/test2.launch:6-14).
//       --standalone selects the IP.cpp twin of ImageProjection: RFANS-16M ring table + removeClosedPointCloud(1.0 m)
//       (IP.cpp:77-104,117,142-172; utility.h:81); the default is the nodelet (imageProjection.cpp) on the reference geometry
//       16 x 4000 (utility.h:50-55).  --list prints the bag's topics and needs no GPU.
// Every scan goes through ImageProjection -> LaserOdometry -> LaserMapping with one alego_scan_process call, as a single nodelet
// manager would run them (launch/test.launch:6-10); every new key frame is pulled across the boundary the way the reference's
// pose-graph thread reads cloud_keyposes_6d_ (laserMapping.cpp:586-596); one JSON line with the final poses is printed.
//
//   g++ -O2 -std=c++17 -Iinclude examples/replay.cpp -o examples/replay -La-lego-loam_amd -lalego_mi355x -lalego_synth
//       -Wl,-rpath,'$ORIGIN/../a-lego-loam_amd'                                  (__graft_entry__.build() does this)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "alego_mi355x.h"

extern "C" int alego_synth_scan(const alego_params* P, int stream, long scan_index, int flags, alego_point* out, int cap);

int main(int argc, char** argv) {
  std::string bag_path, topic = "/lslidar_point_cloud";
  bool list_only = false, standalone = false;
  long max_scans = -1;
  int n_scan = 16, horizon = -1;
  std::vector<const char*> pos;
  for (int i = 1; i < argc; ++i) {
    const std::string a = argv[i];
    auto val = [&]() -> const char* { if (i + 1 >= argc) { std::fprintf(stderr, "%s needs a value\n", a.c_str()); std::exit(2); } return argv[++i]; };
    if (a == "--bag") bag_path = val();
    else if (a == "--topic") topic = val();
    else if (a == "--list") list_only = true;
    else if (a == "--standalone") standalone = true;
    else if (a == "--scans") max_scans = std::atol(val());
    else if (a == "--n-scan") n_scan = std::atoi(val());
    else if (a == "--horizon") horizon = std::atoi(val());
    else pos.push_back(argv[i]);
  }
  alego_bag* bag = nullptr;
  long n_scans = pos.size() > 0 ? std::atol(pos[0]) : 40;
  if (!bag_path.empty()) {
    if (alego_bag_open(bag_path.c_str(), &bag) != ALEGO_OK) return 1;
    if (list_only) {
      for (int i = 0; i < alego_bag_topic_count(bag); ++i) {
        const char *t, *ty; int64_t n;
        alego_bag_topic_info(bag, i, &t, &ty, &n);
        std::printf("%-32s %-32s %lld\n", t, ty, (long long)n);
      }
      alego_bag_close(bag);
      return 0;
    }
    n_scans = (long)alego_bag_message_count(bag, topic.c_str());
    if (n_scans <= 0) { std::fprintf(stderr, "no messages on %s (try --list)\n", topic.c_str()); alego_bag_close(bag); return 1; }
    if (max_scans >= 0 && max_scans < n_scans) n_scans = max_scans;
    if (horizon < 0) horizon = 4000;           // the geometry compiled into the reference (utility.h:50-55)
  } else {
    if (pos.size() > 1) n_scan = std::atoi(pos[1]);
    horizon = pos.size() > 2 ? std::atoi(pos[2]) : (horizon < 0 ? 1800 : horizon);
  }
  alego_params P;
  alego_default_params(&P, n_scan, horizon);
  if (standalone) { P.laser_type = ALEGO_LASER_RFANS_16M; P.near_filter = 1; }
  if (alego_params_sizeof() != (int)sizeof(alego_params)) { std::fprintf(stderr, "header / library mismatch\n"); return 2; }
  const int N = P.n_scan * P.horizon_scan;
  const int cap_in = bag ? (1 << 20) : N;      // a driver may publish more returns than cells; the library takes at most N per scan
  std::vector<alego_point> pts(cap_in), kc(N), ks(N), ko(N);
  double stamp0 = 0.0;
  if (bag) {   // pcl::removeNaNFromPointCloud follows the message's is_dense (alego_params.input_is_dense): read it off the first message
    int32_t dense = 0;
    if (alego_bag_read_pc2(bag, topic.c_str(), 0, pts.data(), cap_in, &stamp0, &dense) < 0) { std::fprintf(stderr, "%s\n", alego_bag_last_error(bag)); alego_bag_close(bag); return 1; }
    P.input_is_dense = dense;
  }
  alego_handle* h = nullptr;
  if (int rc = alego_create(&P, /*device*/ 0, /*slots*/ 1, /*ring*/ 1, &h)) {
    std::fprintf(stderr, "alego_create failed (%d): there is no CPU fallback, an MI355X is required\n", rc);
    return 1;
  }
  alego_pose odom{}, mapped{};
  int key_frames = 0, last_flags = 0, dropped = 0;
  float last_key_pose[6] = {0, 0, 0, 0, 0, 0};
  for (long k = 0; k < n_scans; ++k) {
    int n;
    double stamp = 0.1 * k;
    if (bag) {
      n = alego_bag_read_pc2(bag, topic.c_str(), k, pts.data(), cap_in, &stamp, nullptr);
      if (n < 0) { std::fprintf(stderr, "message %ld: %s\n", k, alego_bag_last_error(bag)); ++dropped; continue; }   // pcCB would warn and return
      if (n > N) { std::fprintf(stderr, "message %ld: %d points > n_scan * horizon_scan = %d (use --n-scan / --horizon)\n", k, n, N); ++dropped; continue; }
    } else {
      n = alego_synth_scan(&P, 0, k, 0, pts.data(), N);
    }
    alego_scan_in in{pts.data(), n, stamp};
    const int flags = alego_scan_process(h, 0, &in, /*IP | LO | LM*/ 7, nullptr, nullptr, &odom, &mapped);
    if (flags < 0) { std::fprintf(stderr, "scan %ld: %s\n", k, alego_last_error(h)); alego_destroy(h); return 1; }
    last_flags = flags;
    if (flags & ALEGO_FLAG_LM_KEYFRAME) {   // saveKeyFramesAndFactor stored a frame: fetch it as the pose-graph thread would
      alego_keyframe kf{};
      kf.corner = kc.data(); kf.corner_cap = N; kf.surf = ks.data(); kf.surf_cap = N; kf.outlier = ko.data(); kf.outlier_cap = N;
      if (alego_lm_get_keyframe(h, 0, -1, &kf) < 0) { std::fprintf(stderr, "get_keyframe: %s\n", alego_last_error(h)); alego_destroy(h); return 1; }
      ++key_frames;
      for (int i = 0; i < 6; ++i) last_key_pose[i] = kf.pose[i];
    }
  }
  std::printf("{\"scans\": %ld, \"dropped\": %d, \"flags\": %d, \"key_frames\": %d, \"resident_key_frames\": %d, "
              "\"odom_t\": [%.17g, %.17g, %.17g], \"map_t\": [%.17g, %.17g, %.17g], \"map_params\": [%.17g, %.17g, %.17g, %.17g, %.17g, %.17g], "
              "\"last_key_pose\": [%.9g, %.9g, %.9g, %.9g, %.9g, %.9g]}\n",
              n_scans, dropped, last_flags, key_frames, alego_lm_keyframe_count(h, 0), odom.t[0], odom.t[1], odom.t[2], mapped.t[0], mapped.t[1], mapped.t[2],
              mapped.params[0], mapped.params[1], mapped.params[2], mapped.params[3], mapped.params[4], mapped.params[5],
              last_key_pose[0], last_key_pose[1], last_key_pose[2], last_key_pose[3], last_key_pose[4], last_key_pose[5]);
  alego_destroy(h);
  if (bag) alego_bag_close(bag);
  return 0;
}
