// DECLARATIONS ONLY -- see ../README.md.  pcl/point_types.h + pcl/point_cloud.h + pcl/PointIndices.h as far
// as include/KeypointLearning.h uses them.
#pragma once
#include <cstdint>
#include <cstdio>
#include <memory>
#include <string>
#include <vector>

#define PCL_ERROR(...) std::fprintf(stderr, __VA_ARGS__)          /* pcl/console/print.h */

namespace boost {                                                  /* PCL 1.8 uses boost::shared_ptr */
template <class T> using shared_ptr = std::shared_ptr<T>;
}
namespace Eigen {
struct Vector4f {
    float coeff(int i) const;
};
}  // namespace Eigen

namespace pcl {
struct PointXYZ { float x, y, z, data_pad; };
struct PointXYZI { float x, y, z, data_pad, intensity, pad2[3]; };
struct Normal { float normal_x, normal_y, normal_z, data_pad, curvature, pad2[3]; };

template <typename PointT>
class PointCloud {
public:
    typedef boost::shared_ptr<PointCloud<PointT>> Ptr;
    typedef boost::shared_ptr<const PointCloud<PointT>> ConstPtr;
    std::vector<PointT> points;
    uint32_t width, height;
    bool is_dense;
    Eigen::Vector4f sensor_origin_;
    size_t size() const;
    bool isOrganized() const;
};

struct PointIndices {
    typedef boost::shared_ptr<PointIndices> Ptr;
    typedef boost::shared_ptr<const PointIndices> ConstPtr;
    std::vector<int> indices;
};
typedef boost::shared_ptr<PointIndices> PointIndicesPtr;
typedef boost::shared_ptr<const PointIndices> PointIndicesConstPtr;
}  // namespace pcl
