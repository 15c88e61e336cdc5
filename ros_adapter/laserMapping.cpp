// ros_adapter/laserMapping.cpp — nodelet loam/LaserMapping on top of alego_lm_process and the key-frame pass-through.
// Same plugin name, topics, 100 Hz poll and 5 ms sync window as src/laserMapping.cpp:82-131; transformAssociateToMap ...
// transformUpdate (:116-122) incl. the every-2nd-frame gate (:112) and the key-frame bookkeeping without loop closure are the
// library call.  Every new key frame is read back (alego_lm_get_keyframe) into host-side stores that play the part of
// cloud_keyposes_3d_ / *_frames_ (:531-555): they feed /keyposes (:586-596) and are what a GTSAM + ICP loop-closure thread
// (:633-824, not part of the hot path) would work on, writing corrections back with alego_lm_set_keypose /
// alego_lm_reset_window / alego_lm_apply_correction (:561-584).
#include "alego_ros_common.h"
#ifdef ALEGO_HAVE_ROS
#include <cmath>
#include <thread>

namespace loam {

class LaserMapping : public nodelet::Nodelet {
 public:
  void onInit() override {
    nh_ = getMTNodeHandle();
    ros::NodeHandle pnh = getMTPrivateNodeHandle();
    h_ = alego_ros::shared_handle(pnh);
    pub_odom_ = nh_.advertise<nav_msgs::Odometry>("/odom_aft_mapped", 10);
    pub_keyposes_ = nh_.advertise<sensor_msgs::PointCloud2>("/keyposes", 10);
    sub_surf_ = nh_.subscribe<sensor_msgs::PointCloud2>("/surf_last", 10, &LaserMapping::surfHandler, this);
    sub_corner_ = nh_.subscribe<sensor_msgs::PointCloud2>("/corner_last", 10, &LaserMapping::cornerHandler, this);
    sub_outlier_ = nh_.subscribe<sensor_msgs::PointCloud2>("/outlier", 10, &LaserMapping::outlierHandler, this);
    sub_odom_ = nh_.subscribe<nav_msgs::Odometry>("/odom/lidar", 10, &LaserMapping::odomHandler, this);
    main_thread_ = std::thread(&LaserMapping::mainLoop, this);
  }

 private:
  // latest-value registers, not queues (:133-186)
  void surfHandler(const sensor_msgs::PointCloud2ConstPtr& m) { std::lock_guard<std::mutex> l(mtx_); alego_ros::from_ros(*m, surf_); t_surf_ = m->header.stamp.toSec(); new_surf_ = true; }
  void cornerHandler(const sensor_msgs::PointCloud2ConstPtr& m) { std::lock_guard<std::mutex> l(mtx_); alego_ros::from_ros(*m, corner_); t_corner_ = m->header.stamp.toSec(); new_corner_ = true; }
  void outlierHandler(const sensor_msgs::PointCloud2ConstPtr& m) { std::lock_guard<std::mutex> l(mtx_); alego_ros::from_ros(*m, outlier_); t_outlier_ = m->header.stamp.toSec(); new_outlier_ = true; }
  void odomHandler(const nav_msgs::OdometryConstPtr& m) {
    std::lock_guard<std::mutex> l(mtx_);
    odom_.t[0] = m->pose.pose.position.x; odom_.t[1] = m->pose.pose.position.y; odom_.t[2] = m->pose.pose.position.z;
    odom_.q[0] = m->pose.pose.orientation.w; odom_.q[1] = m->pose.pose.orientation.x; odom_.q[2] = m->pose.pose.orientation.y; odom_.q[3] = m->pose.pose.orientation.z;
    odom_.valid = 1; t_odom_ = m->header.stamp.toSec(); new_odom_ = true;
  }

  void mainLoop() {
    ros::Rate rate(100);
    while (ros::ok()) {
      rate.sleep();
      std::lock_guard<std::mutex> l(mtx_);
      if (!(new_surf_ && new_corner_ && new_outlier_ && new_odom_) || std::fabs(t_surf_ - t_corner_) >= 0.005 || std::fabs(t_surf_ - t_outlier_) >= 0.005 ||
          std::fabs(t_surf_ - t_odom_) >= 0.005) continue;
      new_surf_ = new_corner_ = new_outlier_ = new_odom_ = false;
      if (!h_) continue;
      alego_pose mapped;
      alego_ros::HandleLock lock(h_);   // until the end of this iteration: alego_lm_process and the key-frame fetch see one handle state
      const int flags = alego_lm_process(h_, corner_.data(), (int32_t)corner_.size(), surf_.data(), (int32_t)surf_.size(), outlier_.data(),
                                         (int32_t)outlier_.size(), &odom_, &mapped);
      if (flags < 0) { NODELET_ERROR("alego_lm_process: %s", alego_last_error(h_)); continue; }
      nav_msgs::OdometryPtr o(new nav_msgs::Odometry);                                  // :167-181
      o->header.frame_id = "map"; o->child_frame_id = "/laser"; o->header.stamp.fromSec(t_odom_);
      o->pose.pose.position.x = mapped.t[0]; o->pose.pose.position.y = mapped.t[1]; o->pose.pose.position.z = mapped.t[2];
      o->pose.pose.orientation.w = mapped.q[0]; o->pose.pose.orientation.x = mapped.q[1]; o->pose.pose.orientation.y = mapped.q[2]; o->pose.pose.orientation.z = mapped.q[3];
      pub_odom_.publish(o);
      if (flags & ALEGO_FLAG_LM_KEYFRAME) {                                             // host copy of the new key frame (:531-555)
        alego_keyframe kf{};
        kf_c_.resize(1 << 16); kf_s_.resize(1 << 16); kf_o_.resize(1 << 16);
        kf.corner = kf_c_.data(); kf.corner_cap = (int32_t)kf_c_.size(); kf.surf = kf_s_.data(); kf.surf_cap = (int32_t)kf_s_.size();
        kf.outlier = kf_o_.data(); kf.outlier_cap = (int32_t)kf_o_.size();
        if (alego_lm_get_keyframe(h_, 0, -1, &kf) == 0) keyposes_.push_back(alego_point{kf.pose[0], kf.pose[1], kf.pose[2], (float)kf.id + 0.1f});   // :529
      }
      if (pub_keyposes_.getNumSubscribers() > 0) {                                      // publish() :586-596
        sensor_msgs::PointCloud2Ptr m(new sensor_msgs::PointCloud2);
        std_msgs::Header hd; hd.frame_id = "map"; hd.stamp.fromSec(t_odom_);
        alego_ros::to_ros(keyposes_.data(), (int)keyposes_.size(), hd, *m);
        pub_keyposes_.publish(m);
      }
    }
  }

  ros::NodeHandle nh_;
  ros::Subscriber sub_surf_, sub_corner_, sub_outlier_, sub_odom_;
  ros::Publisher pub_odom_, pub_keyposes_;
  std::thread main_thread_;
  std::mutex mtx_;
  alego_handle* h_ = nullptr;
  std::vector<alego_point> surf_, corner_, outlier_, kf_c_, kf_s_, kf_o_, keyposes_;
  alego_pose odom_{};
  double t_surf_ = 0, t_corner_ = 0, t_outlier_ = 0, t_odom_ = 0;
  bool new_surf_ = false, new_corner_ = false, new_outlier_ = false, new_odom_ = false;
};

}  // namespace loam
PLUGINLIB_EXPORT_CLASS(loam::LaserMapping, nodelet::Nodelet)
#endif
