// ros_adapter/laserOdometry.cpp — nodelet loam/LaserOdometry on top of alego_lo_process.
// Same plugin name, topics, queues, 100 Hz poll, 0.2 s sync window, frames and stamps as src/laserOdometry.cpp:6-109,513-553;
// the body of mainLoop between the sync check and the publishers (:111-508,:531-534) is the library call.
#include "alego_ros_common.h"
#ifdef ALEGO_HAVE_ROS
#include <cmath>
#include <queue>
#include <thread>

#include <alego/cloud_info.h>

namespace loam {

class LaserOdometry : public nodelet::Nodelet {
 public:
  void onInit() override {
    nh_ = getMTNodeHandle();
    ros::NodeHandle pnh = getMTPrivateNodeHandle();
    h_ = alego_ros::shared_handle(pnh);
    int n_scan = 16, horizon = 4000;
    pnh.param("n_scan", n_scan, n_scan); pnh.param("horizon_scan", horizon, horizon);
    n_ = n_scan * (horizon > 0 ? horizon : 4000);
    for (auto* v : {&sharp_, &less_sharp_, &flat_, &less_flat_}) v->resize(n_);
    pub_corner_ = nh_.advertise<sensor_msgs::PointCloud2>("/corner", 10);
    pub_corner_less_ = nh_.advertise<sensor_msgs::PointCloud2>("/corner_less", 10);
    pub_surf_ = nh_.advertise<sensor_msgs::PointCloud2>("/surf", 10);
    pub_surf_less_ = nh_.advertise<sensor_msgs::PointCloud2>("/surf_less", 10);
    pub_undistorted_pc_ = nh_.advertise<sensor_msgs::PointCloud2>("/undistorted", 10);   // :56 (published by adjustDistortion, :718-725: only with deskew_mode = 1)
    pub_odom_ = nh_.advertise<nav_msgs::Odometry>("/odom/lidar", 10);
    pub_surf_last_ = nh_.advertise<sensor_msgs::PointCloud2>("/surf_last", 10);
    pub_corner_last_ = nh_.advertise<sensor_msgs::PointCloud2>("/corner_last", 10);
    pub_outlier_last_ = nh_.advertise<sensor_msgs::PointCloud2>("/outlier_last", 10);   // :60 (advertised; the reference never publishes on it)
    // frames of /odom/lidar: 0 (default) the nodelet's /odom -> /laser (:513-529); 1 the standalone node's /odom -> /base_link with
    // tf_o2b = tf_o2l * tf_b2l^-1 (LO.cpp:588-608; tf_b2l_ is the identity there, LO.cpp:121 — a host with a calibrated mount sets it here)
    pnh.param("standalone_frames", standalone_frames_, standalone_frames_);
    for (int i = 0; i < 16; ++i) tf_b2l_[i] = (i % 5 == 0) ? 1.0 : 0.0;
    undist_.resize(n_);
    sub_seg_ = nh_.subscribe<sensor_msgs::PointCloud2>("/segmented_cloud", 10, &LaserOdometry::segCloudHandler, this);
    sub_info_ = nh_.subscribe<alego::cloud_info>("/seg_info", 10, &LaserOdometry::segInfoHandler, this);
    sub_outlier_ = nh_.subscribe<sensor_msgs::PointCloud2>("/outlier", 10, &LaserOdometry::outlierHandler, this);
    sub_imu_ = nh_.subscribe<sensor_msgs::Imu>("/imu/data", 100, &LaserOdometry::imuHandler, this);   // :65-68 (use_imu)
    static std::thread main_thread(&LaserOdometry::mainLoop, this);   // :76
  }

 private:
  void segCloudHandler(const sensor_msgs::PointCloud2ConstPtr& m) { std::lock_guard<std::mutex> l(m_buf_); seg_buf_.push(m); }
  void segInfoHandler(const alego::cloud_infoConstPtr& m) { std::lock_guard<std::mutex> l(m_buf_); info_buf_.push(m); }
  void outlierHandler(const sensor_msgs::PointCloud2ConstPtr& m) { std::lock_guard<std::mutex> l(m_buf_); outlier_buf_.push(m); }
  // :761-802: the handler's ring of dead-reckoned samples lives in the library; it is only read when alego_params.deskew_mode = 1
  // (the reference's adjustDistortion call at :115 is commented out)
  void imuHandler(const sensor_msgs::ImuConstPtr& m) {
    if (!h_) return;
    alego_imu s{};
    s.stamp = m->header.stamp.toSec();
    s.orientation[0] = m->orientation.w; s.orientation[1] = m->orientation.x; s.orientation[2] = m->orientation.y; s.orientation[3] = m->orientation.z;
    s.linear_acceleration[0] = m->linear_acceleration.x; s.linear_acceleration[1] = m->linear_acceleration.y; s.linear_acceleration[2] = m->linear_acceleration.z;
    s.angular_velocity[0] = m->angular_velocity.x; s.angular_velocity[1] = m->angular_velocity.y; s.angular_velocity[2] = m->angular_velocity.z;
    alego_ros::HandleLock lock(h_);   // a handle is single-threaded: the main loop, ImageProjection and LaserMapping take the same lock
    if (alego_lo_push_imu(h_, 0, &s, 1) < 0) NODELET_WARN_THROTTLE(1.0, "alego_lo_push_imu: %s", alego_last_error(h_));
  }

  void mainLoop() {
    ros::Rate rate(100);
    while (ros::ok()) {
      rate.sleep();
      sensor_msgs::PointCloud2ConstPtr seg;
      alego::cloud_infoConstPtr info;
      {
        std::lock_guard<std::mutex> l(m_buf_);
        if (seg_buf_.empty() || info_buf_.empty() || outlier_buf_.empty()) continue;
        const double t1 = seg_buf_.front()->header.stamp.toSec(), t2 = info_buf_.front()->header.stamp.toSec(), t3 = outlier_buf_.front()->header.stamp.toSec();
        if (std::fabs(t1 - t2) > 0.2 || std::fabs(t1 - t3) > 0.2) {   // :91-108: unsynchronised -> drop everything
          while (!seg_buf_.empty()) seg_buf_.pop();
          while (!info_buf_.empty()) info_buf_.pop();
          while (!outlier_buf_.empty()) outlier_buf_.pop();
          continue;
        }
        seg = seg_buf_.front(); info = info_buf_.front();
        seg_buf_.pop(); info_buf_.pop(); outlier_buf_.pop();
      }
      if (!h_ || alego_ros::from_ros(*seg, seg_pts_) < 0) continue;
      alego_seg_out in{};
      in.seg = seg_pts_.data(); in.seg_cap = in.m = (int32_t)seg_pts_.size();
      in.ground = const_cast<uint8_t*>(info->segmentedCloudGroundFlag.data()); in.col = const_cast<int32_t*>(info->segmentedCloudColInd.data());
      in.range = const_cast<float*>(info->segmentedCloudRange.data());
      in.ring_start = const_cast<int32_t*>(info->startRingIndex.data()); in.ring_end = const_cast<int32_t*>(info->endRingIndex.data());
      alego_feat_out f{};
      f.sharp = sharp_.data(); f.sharp_cap = n_; f.less_sharp = less_sharp_.data(); f.less_sharp_cap = n_;
      f.flat = flat_.data(); f.flat_cap = n_; f.less_flat = less_flat_.data(); f.less_flat_cap = n_;
      in.orientation[0] = info->startOrientation; in.orientation[1] = info->endOrientation; in.orientation[2] = info->orientationDiff;
      in.stamp = seg->header.stamp.toSec();   // t1 (:111)
      alego_pose odom;
      int flags, nu = -1;
      {
        alego_ros::HandleLock lock(h_);
        flags = alego_lo_process(h_, &in, &f, &odom);
        if (flags < 0) { NODELET_ERROR("alego_lo_process: %s", alego_last_error(h_)); continue; }
        // /undistorted is fetched under the SAME lock as the scan's LaserOdometry step: the three nodelets share slot 0, and ImageProjection of the
        // next scan may otherwise run in between (the getter also reads lo_deskew's own point count, SC_M_DSK, not ImageProjection's)
        if (pub_undistorted_pc_.getNumSubscribers() > 0) nu = alego_lo_get_undistorted(h_, 0, undist_.data(), n_);
      }
      std_msgs::Header hd = seg->header;
      hd.frame_id = "/laser";
      auto pub = [&](ros::Publisher& p, const alego_point* pts, int n) {
        if (p.getNumSubscribers() == 0) return;
        sensor_msgs::PointCloud2Ptr m(new sensor_msgs::PointCloud2);
        alego_ros::to_ros(pts, n, hd, *m);
        p.publish(m);
      };
      pub(pub_corner_, f.sharp, f.n_sharp); pub(pub_corner_less_, f.less_sharp, f.n_less_sharp);   // :299-314
      pub(pub_surf_, f.flat, f.n_flat); pub(pub_surf_less_, f.less_flat, f.n_less_flat);
      if (nu >= 0) pub(pub_undistorted_pc_, undist_.data(), nu);   // :718-725 (ALEGO_ERR_ARG: deskew_mode = 0, adjustDistortion is not run — nothing to publish, as in the reference)
      if (!(flags & ALEGO_FLAG_LO_INIT)) {                                                           // :513-529 / LO.cpp:588-608
        const char* child = standalone_frames_ ? "/base_link" : "/laser";
        if (standalone_frames_) { alego_pose b; if (alego_pose_o2b(&odom, tf_b2l_, &b) == ALEGO_OK) odom = b; }
        nav_msgs::OdometryPtr o(new nav_msgs::Odometry);
        o->header.frame_id = "/odom"; o->child_frame_id = child; o->header.stamp = seg->header.stamp;
        o->pose.pose.position.x = odom.t[0]; o->pose.pose.position.y = odom.t[1]; o->pose.pose.position.z = odom.t[2];
        o->pose.pose.orientation.w = odom.q[0]; o->pose.pose.orientation.x = odom.q[1]; o->pose.pose.orientation.y = odom.q[2]; o->pose.pose.orientation.z = odom.q[3];
        pub_odom_.publish(o);
        tf::Transform t;
        tf::poseMsgToTF(o->pose.pose, t);
        tf_.sendTransform(tf::StampedTransform(t, o->header.stamp, "/odom", child));
      }
      // /surf_last and /corner_last are published on every frame, also the initialising one (:537-546)
      sensor_msgs::PointCloud2Ptr ms(new sensor_msgs::PointCloud2), mc(new sensor_msgs::PointCloud2);
      alego_ros::to_ros(f.less_flat, f.n_less_flat, hd, *ms); alego_ros::to_ros(f.less_sharp, f.n_less_sharp, hd, *mc);
      pub_surf_last_.publish(ms); pub_corner_last_.publish(mc);
    }
  }

  ros::NodeHandle nh_;
  ros::Subscriber sub_seg_, sub_info_, sub_outlier_, sub_imu_;
  ros::Publisher pub_corner_, pub_corner_less_, pub_surf_, pub_surf_less_, pub_undistorted_pc_, pub_odom_, pub_surf_last_, pub_corner_last_, pub_outlier_last_;
  int standalone_frames_ = 0;
  double tf_b2l_[16];
  tf::TransformBroadcaster tf_;
  std::mutex m_buf_;
  std::queue<sensor_msgs::PointCloud2ConstPtr> seg_buf_, outlier_buf_;
  std::queue<alego::cloud_infoConstPtr> info_buf_;
  alego_handle* h_ = nullptr;
  int n_ = 0;
  std::vector<alego_point> seg_pts_, sharp_, less_sharp_, flat_, less_flat_, undist_;
};

}  // namespace loam
PLUGINLIB_EXPORT_CLASS(loam::LaserOdometry, nodelet::Nodelet)
#endif
