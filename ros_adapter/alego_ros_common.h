// alego_ros_common.h — shared pieces of the three nodelet adapters: the process-wide handle and
// sensor_msgs/PointCloud2 <-> alego_point marshalling (through the ROS-free parser of the C ABI).
//
// The adapters keep the reference's plugin names, topics, queue sizes, frames and stamps
// (nodelet_plugins.xml:1-10; imageProjection.cpp:42-45; laserOdometry.cpp:52-64; laserMapping.cpp:82-93) and contain no
// numerics: every callback body is one call into libalego_mi355x.so.  They only compile where ROS is installed
// (`__has_include(<ros/ros.h>)`); this image has no ROS, so here they preprocess to empty translation units
// (tests/test_host.py::test_ros_adapter_sources_are_guarded).
#ifndef ALEGO_ROS_COMMON_H_
#define ALEGO_ROS_COMMON_H_
#if defined(__has_include)
#if __has_include(<ros/ros.h>) && __has_include(<nodelet/nodelet.h>)
#define ALEGO_HAVE_ROS 1
#endif
#endif

#ifdef ALEGO_HAVE_ROS
#include <mutex>
#include <vector>

#include <nav_msgs/Odometry.h>
#include <nodelet/nodelet.h>
#include <pluginlib/class_list_macros.h>
#include <ros/ros.h>
#include <sensor_msgs/Imu.h>
#include <sensor_msgs/PointCloud2.h>
#include <sensor_msgs/PointField.h>
#include <tf/transform_broadcaster.h>

#include "alego_mi355x.h"

namespace alego_ros {

// One handle per process: the three nodelets of launch/test.launch share a manager, so IP -> LO -> LM can hand their
// intermediates over in HBM; as three processes (launch/test2.launch) every process owns a handle and the messages carry the data.
inline alego_handle* shared_handle(ros::NodeHandle& pnh) {
  static std::mutex m;
  static alego_handle* h = nullptr;
  std::lock_guard<std::mutex> lock(m);
  if (!h) {
    alego_params P;
    int n_scan = 16, horizon = 0, device = 0;   // horizon 0 = the reference geometry 16 x 4000 (utility.h:50-55)
    pnh.param("n_scan", n_scan, n_scan);
    pnh.param("horizon_scan", horizon, horizon);
    pnh.param("device", device, device);
    alego_default_params(&P, n_scan, horizon);
    if (alego_create(&P, device, 1, 1, &h) != ALEGO_OK) { ROS_FATAL("alego_create failed: no gfx950 device (there is no CPU fallback)"); h = nullptr; }
  }
  return h;
}

// The shared handle is single-threaded by contract, but ImageProjection's callback, LaserOdometry's main loop + IMU handler and
// LaserMapping's main loop run on different threads of the nodelet manager: every alego_* call on the handle is bracketed by
// the handle's own lock (alego_handle_lock: one mutex per handle, whoever the caller is — a mutex that is a member of one
// nodelet would not exclude the other two).
struct HandleLock {
  alego_handle* h;
  explicit HandleLock(alego_handle* handle) : h(handle) { if (h) alego_handle_lock(h); }
  ~HandleLock() { if (h) alego_handle_unlock(h); }
  HandleLock(const HandleLock&) = delete;
  HandleLock& operator=(const HandleLock&) = delete;
};

// pcl::fromROSMsg<PointXYZI> without PCL (imageProjection.cpp:54-55)
inline int from_ros(const sensor_msgs::PointCloud2& msg, std::vector<alego_point>& out) {
  std::vector<alego_pc2_field> f(msg.fields.size());
  for (size_t i = 0; i < f.size(); ++i) f[i] = alego_pc2_field{msg.fields[i].name.c_str(), msg.fields[i].offset, msg.fields[i].datatype, msg.fields[i].count};
  out.resize((size_t)msg.width * msg.height);
  const int n = alego_pc2_to_points(msg.data.data(), msg.data.size(), msg.width, msg.height, msg.point_step, msg.row_step, msg.is_bigendian,
                                    f.data(), (int)f.size(), out.data(), (int32_t)out.size());
  if (n >= 0) out.resize(n);
  return n;
}

// pcl::toROSMsg<PointXYZI>: PCL's 32-byte point (x y z @0/4/8, intensity @16)
inline void to_ros(const alego_point* pts, int n, const std_msgs::Header& header, sensor_msgs::PointCloud2& msg) {
  msg.header = header;
  msg.height = 1; msg.width = n; msg.is_bigendian = false; msg.is_dense = true;
  msg.point_step = 32; msg.row_step = 32u * n;
  msg.fields.resize(4);
  const char* names[4] = {"x", "y", "z", "intensity"};
  const uint32_t offs[4] = {0, 4, 8, 16};
  for (int i = 0; i < 4; ++i) { msg.fields[i].name = names[i]; msg.fields[i].offset = offs[i]; msg.fields[i].datatype = sensor_msgs::PointField::FLOAT32; msg.fields[i].count = 1; }
  msg.data.assign((size_t)n * 32, 0);
  for (int i = 0; i < n; ++i) {
    float* p = reinterpret_cast<float*>(&msg.data[(size_t)i * 32]);
    p[0] = pts[i].x; p[1] = pts[i].y; p[2] = pts[i].z; p[4] = pts[i].intensity;
  }
}

}  // namespace alego_ros
#endif  // ALEGO_HAVE_ROS
#endif
