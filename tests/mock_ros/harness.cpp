// harness.cpp — the three nodelet adapters (ros_adapter/*.cpp), compiled against the in-process ROS mock and DRIVEN: scans of the
// synthetic stream are published on /lslidar_point_cloud as sensor_msgs/PointCloud2 (PCL's 32-byte layout), the nodelets talk to
// each other over their topics exactly as under a nodelet manager (launch/test.launch:6-10) — ImageProjection's callback on the
// publisher's thread, LaserOdometry's and LaserMapping's 100 Hz main loops on threads of their own, all three on ONE shared
// alego_handle — and what arrives on /odom/lidar and /odom_aft_mapped is compared with a second handle that runs the same scans
// through alego_scan_process.  Prints one JSON line.  Test infrastructure (tests/test_ros_adapter.py); needs an MI355X.
#include "../../ros_adapter/imageProjection.cpp"
#include "../../ros_adapter/laserOdometry.cpp"
#include "../../ros_adapter/laserMapping.cpp"

#include <cstdlib>
#include <cstring>
#include <unistd.h>

extern "C" int alego_synth_scan(const alego_params* P, int stream, long scan_index, int flags, alego_point* out, int cap);

int main(int argc, char** argv) {
  const int n_scans = argc > 1 ? std::atoi(argv[1]) : 12;
  const int n_scan = 16, horizon = 1800;
  ros::Bus::get().params["n_scan"] = n_scan;
  ros::Bus::get().params["horizon_scan"] = horizon;
  const bool standalone = argc > 2 && std::strcmp(argv[2], "standalone") == 0;   // LO.cpp's frame convention: /odom -> /base_link (tf_b2l = identity)
  if (standalone) ros::Bus::get().params["standalone_frames"] = 1;
  loam::ImageProjection ip;
  loam::LaserOdometry lo;
  loam::LaserMapping lm;
  ros::NodeHandle nh;
  std::mutex m;
  std::vector<nav_msgs::Odometry> odom, mapped;
  int n_surf_last = 0, n_undistorted = 0;
  nh.subscribe_fn<nav_msgs::Odometry>("/odom/lidar", [&](const nav_msgs::OdometryConstPtr& o) { std::lock_guard<std::mutex> l(m); odom.push_back(*o); });
  nh.subscribe_fn<nav_msgs::Odometry>("/odom_aft_mapped", [&](const nav_msgs::OdometryConstPtr& o) { std::lock_guard<std::mutex> l(m); mapped.push_back(*o); });
  ip.init(); lo.init(); lm.init();
  // (subscribed after the nodelets: LaserMapping's own handler for /surf_last has run when this one fires)
  nh.subscribe_fn<sensor_msgs::PointCloud2>("/surf_last", [&](const sensor_msgs::PointCloud2ConstPtr&) { std::lock_guard<std::mutex> l(m); ++n_surf_last; });
  int topics_advertised = 0;
  { std::lock_guard<std::mutex> l(ros::Bus::get().m); topics_advertised = (int)ros::Bus::get().topics.count("/undistorted") + (int)ros::Bus::get().topics.count("/outlier_last"); }
  if (standalone) nh.subscribe_fn<sensor_msgs::PointCloud2>("/undistorted", [&](const sensor_msgs::PointCloud2ConstPtr&) { std::lock_guard<std::mutex> l(m); ++n_undistorted; });
  ros::Publisher pub = nh.advertise<sensor_msgs::PointCloud2>("/lslidar_point_cloud", 10);

  alego_params P;
  alego_default_params(&P, n_scan, horizon);
  const int N = n_scan * horizon;
  std::vector<alego_point> pts(N);
  alego_handle* ref = nullptr;
  if (alego_create(&P, 0, 1, 1, &ref) != ALEGO_OK) { std::fprintf(stderr, "alego_create failed\n"); return 1; }
  alego_pose ro{}, rm{};
  int bad = 0, lm_frames = 0;
  double worst_odom = 0, worst_map = 0;
  std::string child_frame;
  auto wait_for = [&](auto cond, const char* what) {
    for (int i = 0; i < 20000; ++i) { { std::lock_guard<std::mutex> l(m); if (cond()) return true; } usleep(500); }
    std::fprintf(stderr, "timeout waiting for %s\n", what);
    return false;
  };
  for (int k = 0; k < n_scans; ++k) {
    const int n = alego_synth_scan(&P, 0, k, 0, pts.data(), N);
    sensor_msgs::PointCloud2Ptr msg(new sensor_msgs::PointCloud2);
    msg->header.seq = k; msg->header.stamp.fromSec(100.0 + 0.1 * k); msg->header.frame_id = "laser";
    alego_ros::to_ros(pts.data(), n, msg->header, *msg);
    msg->is_dense = false;
    const size_t odom_before = odom.size(), mapped_before = mapped.size();
    pub.publish(msg);   // ImageProjection::pcCB runs here; LaserOdometry / LaserMapping pick the results up on their threads
    alego_scan_in in{pts.data(), n, 100.0 + 0.1 * k};
    const int flags = alego_scan_process(ref, 0, &in, 7, nullptr, nullptr, &ro, &rm);
    if (flags < 0) { std::fprintf(stderr, "reference handle: %s\n", alego_last_error(ref)); return 1; }
    if (!wait_for([&] { return n_surf_last > k; }, "/surf_last")) return 1;
    if (!(flags & ALEGO_FLAG_LO_INIT)) {
      if (!wait_for([&] { return odom.size() > odom_before; }, "/odom/lidar")) return 1;
      std::lock_guard<std::mutex> l(m);
      const auto& p = odom.back().pose.pose;
      const double e = std::fabs(p.position.x - ro.t[0]) + std::fabs(p.position.y - ro.t[1]) + std::fabs(p.position.z - ro.t[2]) +
                       std::fabs(p.orientation.w - ro.q[0]) + std::fabs(p.orientation.z - ro.q[3]);
      worst_odom = std::max(worst_odom, e);
      if (standalone ? e > 1e-14 : e != 0.0) ++bad;   // (standalone: the quaternion went through a rotation matrix and back)
      child_frame = odom.back().child_frame_id;
    }
    if (k >= 1) {   // LaserMapping's odom handler + gate fire for every scan with an /odom/lidar message; it publishes on every one of them
      if (!wait_for([&] { return mapped.size() > mapped_before; }, "/odom_aft_mapped")) return 1;
      std::lock_guard<std::mutex> l(m);
      const auto& p = mapped.back().pose.pose;
      const double e = std::fabs(p.position.x - rm.t[0]) + std::fabs(p.position.y - rm.t[1]) + std::fabs(p.position.z - rm.t[2]) +
                       std::fabs(p.orientation.w - rm.q[0]) + std::fabs(p.orientation.z - rm.q[3]);
      worst_map = std::max(worst_map, e);
      if (standalone ? e > 1e-12 : e != 0.0) ++bad;   // (standalone: LaserMapping received the re-derived quaternion)
      ++lm_frames;
    }
  }
  std::printf("{\"scans\": %d, \"odom_msgs\": %zu, \"mapped_msgs\": %zu, \"lm_frames\": %d, \"differing\": %d, \"worst_odom_abs\": %.3e, \"worst_map_abs\": %.3e, "
              "\"odom_t\": [%.17g, %.17g, %.17g], \"map_t\": [%.17g, %.17g, %.17g], \"child_frame\": \"%s\", \"topics_advertised\": %d, \"undistorted_msgs\": %d}\n",
              n_scans, odom.size(), mapped.size(), lm_frames, bad, worst_odom, worst_map, ro.t[0], ro.t[1], ro.t[2], rm.t[0], rm.t[1], rm.t[2],
              child_frame.c_str(), topics_advertised, n_undistorted);
  std::fflush(stdout);
  ros::shutdown();
  _exit(bad ? 3 : 0);   // (the nodelets' polling threads are never joined, as in the reference: leave without running destructors)
}
