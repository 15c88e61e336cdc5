// mock_ros_core.h — just enough of roscpp / nodelet / the message headers, in-process, for the three adapters under ros_adapter/
// to COMPILE and RUN in an image without ROS (tests/test_ros_adapter.py).  Test infrastructure only: topics are a process-local
// bus with synchronous delivery in the publisher's thread; names, signatures and message fields follow ROS 1 (melodic).
#ifndef ALEGO_MOCK_ROS_CORE_H_
#define ALEGO_MOCK_ROS_CORE_H_
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace ros {
struct Time {
  uint32_t sec = 0, nsec = 0;
  double toSec() const { return (double)sec + 1e-9 * (double)nsec; }
  Time& fromSec(double t) { sec = (uint32_t)std::floor(t); nsec = (uint32_t)std::llround((t - (double)sec) * 1e9); if (nsec >= 1000000000u) { nsec -= 1000000000u; ++sec; } return *this; }
};
inline std::atomic<bool>& ok_flag() { static std::atomic<bool> f{true}; return f; }
inline bool ok() { return ok_flag().load(); }
inline void shutdown() { ok_flag().store(false); }
struct Rate {
  double hz;
  explicit Rate(double h) : hz(h) {}
  void sleep() { std::this_thread::sleep_for(std::chrono::microseconds((long)(1e6 / hz))); }
};

struct Bus {
  struct Topic { std::vector<std::function<void(const std::shared_ptr<const void>&)>> subs; };
  std::mutex m;
  std::map<std::string, Topic> topics;
  std::map<std::string, int> params;
  static Bus& get() { static Bus b; return b; }
};

class Publisher {
 public:
  Publisher() = default;
  explicit Publisher(std::string t) : topic_(std::move(t)) {}
  template <class M> void publish(const std::shared_ptr<M>& msg) const {
    std::vector<std::function<void(const std::shared_ptr<const void>&)>> subs;
    { std::lock_guard<std::mutex> l(Bus::get().m); subs = Bus::get().topics[topic_].subs; }
    const std::shared_ptr<const void> p = std::static_pointer_cast<const void>(std::shared_ptr<const M>(msg));
    for (auto& f : subs) f(p);
  }
  uint32_t getNumSubscribers() const { std::lock_guard<std::mutex> l(Bus::get().m); return (uint32_t)Bus::get().topics[topic_].subs.size(); }
 private:
  std::string topic_;
};
class Subscriber {};

class NodeHandle {
 public:
  template <class M> Publisher advertise(const std::string& topic, uint32_t /*queue*/) { std::lock_guard<std::mutex> l(Bus::get().m); (void)Bus::get().topics[topic]; return Publisher(topic); }
  template <class M, class T> Subscriber subscribe(const std::string& topic, uint32_t /*queue*/, void (T::*fp)(const std::shared_ptr<const M>&), T* obj) {
    std::lock_guard<std::mutex> l(Bus::get().m);
    Bus::get().topics[topic].subs.push_back([fp, obj](const std::shared_ptr<const void>& p) { (obj->*fp)(std::static_pointer_cast<const M>(p)); });
    return Subscriber();
  }
  // a free function / lambda subscriber (the test harness)
  template <class M> Subscriber subscribe_fn(const std::string& topic, std::function<void(const std::shared_ptr<const M>&)> f) {
    std::lock_guard<std::mutex> l(Bus::get().m);
    Bus::get().topics[topic].subs.push_back([f](const std::shared_ptr<const void>& p) { f(std::static_pointer_cast<const M>(p)); });
    return Subscriber();
  }
  bool param(const std::string& name, int& v, const int& dflt) {
    std::lock_guard<std::mutex> l(Bus::get().m);
    auto it = Bus::get().params.find(name);
    v = it == Bus::get().params.end() ? dflt : it->second;
    return it != Bus::get().params.end();
  }
};
}  // namespace ros

#define ROS_FATAL(...) do { std::fprintf(stderr, "[FATAL] " __VA_ARGS__); std::fprintf(stderr, "\n"); } while (0)
#define ROS_WARN(...) do { std::fprintf(stderr, "[WARN] " __VA_ARGS__); std::fprintf(stderr, "\n"); } while (0)
#define NODELET_WARN(...) ROS_WARN(__VA_ARGS__)
#define NODELET_ERROR(...) do { std::fprintf(stderr, "[ERROR] " __VA_ARGS__); std::fprintf(stderr, "\n"); } while (0)
#define NODELET_WARN_THROTTLE(period, ...) ROS_WARN(__VA_ARGS__)

namespace nodelet {
class Nodelet {
 public:
  virtual ~Nodelet() {}
  void init() { onInit(); }
 protected:
  virtual void onInit() = 0;
  ros::NodeHandle& getMTNodeHandle() { return nh_; }
  ros::NodeHandle& getMTPrivateNodeHandle() { return pnh_; }
 private:
  ros::NodeHandle nh_, pnh_;
};
}  // namespace nodelet
#define PLUGINLIB_EXPORT_CLASS(cls, base)

namespace std_msgs { struct Header { uint32_t seq = 0; ros::Time stamp; std::string frame_id; }; }
namespace geometry_msgs {
struct Point { double x = 0, y = 0, z = 0; };
struct Vector3 { double x = 0, y = 0, z = 0; };
struct Quaternion { double x = 0, y = 0, z = 0, w = 1; };
struct Pose { Point position; Quaternion orientation; };
struct PoseWithCovariance { Pose pose; double covariance[36] = {0}; };
}
namespace sensor_msgs {
struct PointField { enum { INT8 = 1, UINT8, INT16, UINT16, INT32, UINT32, FLOAT32, FLOAT64 }; std::string name; uint32_t offset = 0; uint8_t datatype = 0; uint32_t count = 0; };
struct PointCloud2 {
  std_msgs::Header header; uint32_t height = 0, width = 0; std::vector<PointField> fields; bool is_bigendian = false;
  uint32_t point_step = 0, row_step = 0; std::vector<uint8_t> data; bool is_dense = false;
};
typedef std::shared_ptr<PointCloud2> PointCloud2Ptr;
typedef std::shared_ptr<const PointCloud2> PointCloud2ConstPtr;
struct Imu { std_msgs::Header header; geometry_msgs::Quaternion orientation; geometry_msgs::Vector3 angular_velocity, linear_acceleration; };
typedef std::shared_ptr<const Imu> ImuConstPtr;
}
namespace nav_msgs {
struct Odometry { std_msgs::Header header; std::string child_frame_id; geometry_msgs::PoseWithCovariance pose; };
typedef std::shared_ptr<Odometry> OdometryPtr;
typedef std::shared_ptr<const Odometry> OdometryConstPtr;
}
namespace tf {
struct Transform {};
struct StampedTransform { StampedTransform(const Transform&, const ros::Time&, const std::string&, const std::string&) {} };
inline void poseMsgToTF(const geometry_msgs::Pose&, Transform&) {}
struct TransformBroadcaster { void sendTransform(const StampedTransform&) {} };
}
namespace alego {
struct cloud_info {
  std_msgs::Header header;
  std::vector<int32_t> startRingIndex, endRingIndex;
  float startOrientation = 0, endOrientation = 0, orientationDiff = 0;
  std::vector<uint8_t> segmentedCloudGroundFlag;
  std::vector<int32_t> segmentedCloudColInd;
  std::vector<float> segmentedCloudRange;
};
typedef std::shared_ptr<cloud_info> cloud_infoPtr;
typedef std::shared_ptr<const cloud_info> cloud_infoConstPtr;
}
#endif
