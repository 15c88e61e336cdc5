// pc2.cpp — sensor_msgs/PointCloud2 -> alego_point without ROS or PCL (host side of the ROS adapter).
// Replaces pcl::fromROSMsg<PointXYZI> at the head of ImageProjection::pcCB (src/imageProjection.cpp:54-55, src/IP.cpp:109-110):
// PCL's field mapper matches the message fields to the point type BY NAME and requires identical datatype and count;
// PointXYZI = x, y, z, intensity, all FLOAT32 (sensor_msgs/PointField datatype 7).  A field that is missing (or has
// another type) is left out of the mapping: PCL warns and the member keeps the value of a default-constructed point (0).
// PCL's FieldMatches accepts count == 1 and, for single-element fields, count == 0 (drivers that leave it unset).
#include <cstdint>
#include <cstring>

#include "../../include/alego_mi355x.h"

namespace {
inline float load_f32(const uint8_t* p, bool swap) {
  uint8_t b[4];
  if (swap) { b[0] = p[3]; b[1] = p[2]; b[2] = p[1]; b[3] = p[0]; } else { std::memcpy(b, p, 4); }
  float f;
  std::memcpy(&f, b, 4);
  return f;
}
}  // namespace

extern "C" int alego_pc2_to_points(const uint8_t* data, uint64_t data_len, uint32_t width, uint32_t height, uint32_t point_step,
                                   uint32_t row_step, int is_bigendian, const alego_pc2_field* fields, int n_fields,
                                   alego_point* out, int32_t cap) {
  if ((!data && data_len) || !fields || n_fields < 0 || !out || cap < 0) return ALEGO_ERR_ARG;
  const uint64_t n = (uint64_t)width * height;
  if (n > (uint64_t)cap) return ALEGO_ERR_CAPACITY;
  int64_t off[4] = {-1, -1, -1, -1};
  static const char* names[4] = {"x", "y", "z", "intensity"};
  for (int f = 0; f < n_fields; ++f) {
    if (!fields[f].name) return ALEGO_ERR_ARG;
    for (int k = 0; k < 4; ++k)
      if (std::strcmp(fields[f].name, names[k]) == 0 && fields[f].datatype == 7 && fields[f].count <= 1) off[k] = fields[f].offset;
  }
  if (off[0] < 0 || off[1] < 0 || off[2] < 0) return ALEGO_ERR_ARG;   // fromROSMsg cannot fill an XYZ point without x, y, z
  for (int k = 0; k < 4; ++k)
    if (off[k] >= 0 && (uint64_t)off[k] + 4 > point_step) return ALEGO_ERR_ARG;
  if (height > 0 && row_step < (uint64_t)width * point_step) return ALEGO_ERR_ARG;
  if (n > 0 && (uint64_t)(height - 1) * row_step + (uint64_t)width * point_step > data_len) return ALEGO_ERR_ARG;
  const uint16_t probe = 1;
  const bool host_big = *reinterpret_cast<const uint8_t*>(&probe) == 0;
  const bool swap = (is_bigendian != 0) != host_big;
  uint64_t i = 0;
  for (uint32_t r = 0; r < height; ++r) {
    const uint8_t* row = data + (uint64_t)r * row_step;
    for (uint32_t c = 0; c < width; ++c, ++i) {
      const uint8_t* p = row + (uint64_t)c * point_step;
      out[i].x = load_f32(p + off[0], swap);
      out[i].y = load_f32(p + off[1], swap);
      out[i].z = load_f32(p + off[2], swap);
      out[i].intensity = off[3] >= 0 ? load_f32(p + off[3], swap) : 0.0f;
    }
  }
  return (int)n;
}
