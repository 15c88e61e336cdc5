// ros_adapter/imageProjection.cpp — nodelet loam/ImageProjection on top of alego_ip_process.
// Same plugin name, topics and queue sizes as src/imageProjection.cpp:6-47,318-336; the numeric body of pcCB (:49-208) is the
// library call.  Needs the reference package's alego/cloud_info message (msg/cloud_info.msg).
#include "alego_ros_common.h"
#ifdef ALEGO_HAVE_ROS
#include <alego/cloud_info.h>

namespace loam {

class ImageProjection : public nodelet::Nodelet {
 public:
  void onInit() override {
    nh_ = getMTNodeHandle();
    ros::NodeHandle pnh = getMTPrivateNodeHandle();
    h_ = alego_ros::shared_handle(pnh);
    int n_scan = 16, horizon = 4000;
    pnh.param("n_scan", n_scan, n_scan); pnh.param("horizon_scan", horizon, horizon);
    n_ = n_scan * (horizon > 0 ? horizon : 4000);
    info_.reset(new alego::cloud_info());
    info_->startRingIndex.resize(n_scan); info_->endRingIndex.resize(n_scan);          // :16-20: pre-sized, never shrunk
    info_->segmentedCloudGroundFlag.assign(n_, 0); info_->segmentedCloudColInd.assign(n_, 0); info_->segmentedCloudRange.assign(n_, 0);
    seg_.resize(n_); outlier_.resize(n_);
    pub_seg_ = nh_.advertise<sensor_msgs::PointCloud2>("/segmented_cloud", 10);
    pub_info_ = nh_.advertise<alego::cloud_info>("/seg_info", 10);
    pub_outlier_ = nh_.advertise<sensor_msgs::PointCloud2>("/outlier", 10);
    sub_ = nh_.subscribe<sensor_msgs::PointCloud2>("/lslidar_point_cloud", 10, &ImageProjection::pcCB, this);
  }

 private:
  void pcCB(const sensor_msgs::PointCloud2ConstPtr& msg) {
    if (!h_ || alego_ros::from_ros(*msg, in_) < 0) { NODELET_WARN("unusable PointCloud2"); return; }
    alego_scan_in in{in_.data(), (int32_t)in_.size(), msg->header.stamp.toSec()};
    alego_seg_out out{};
    out.seg = seg_.data(); out.seg_cap = n_; out.outlier = outlier_.data(); out.outlier_cap = n_;
    out.ground = info_->segmentedCloudGroundFlag.data(); out.col = info_->segmentedCloudColInd.data(); out.range = info_->segmentedCloudRange.data();
    out.ring_start = info_->startRingIndex.data(); out.ring_end = info_->endRingIndex.data();
    {
      alego_ros::HandleLock lock(h_);   // LaserOdometry / LaserMapping drive the same handle from their own threads
      if (alego_ip_process(h_, &in, &out) < 0) { NODELET_ERROR("alego_ip_process: %s", alego_last_error(h_)); return; }
    }
    info_->header = msg->header;
    info_->startOrientation = out.orientation[0]; info_->endOrientation = out.orientation[1]; info_->orientationDiff = out.orientation[2];
    if (pub_info_.getNumSubscribers() > 0) pub_info_.publish(info_);                    // :320-335: only with subscribers
    if (pub_seg_.getNumSubscribers() > 0) { sensor_msgs::PointCloud2Ptr m(new sensor_msgs::PointCloud2); alego_ros::to_ros(seg_.data(), out.m, msg->header, *m); pub_seg_.publish(m); }
    if (pub_outlier_.getNumSubscribers() > 0) { sensor_msgs::PointCloud2Ptr m(new sensor_msgs::PointCloud2); alego_ros::to_ros(outlier_.data(), out.n_outlier, msg->header, *m); pub_outlier_.publish(m); }
  }

  ros::NodeHandle nh_;
  ros::Subscriber sub_;
  ros::Publisher pub_seg_, pub_info_, pub_outlier_;
  alego_handle* h_ = nullptr;
  int n_ = 0;
  alego::cloud_infoPtr info_;
  std::vector<alego_point> in_, seg_, outlier_;
};

}  // namespace loam
PLUGINLIB_EXPORT_CLASS(loam::ImageProjection, nodelet::Nodelet)
#endif
