// DECLARATIONS ONLY -- see ../../README.md.  pcl::IntegralImageNormalEstimation (PCL 1.8) as far as
// tools/refgen/refgen_driver.cpp drives it (the calls of /root/reference/include/impl/KeypointLearning.hpp:138-145).
#pragma once
#include <pcl/point_types.h>
namespace pcl {
template <typename PointInT, typename PointOutT>
class IntegralImageNormalEstimation {
public:
    typedef typename pcl::PointCloud<PointInT>::ConstPtr PointCloudConstPtr;
    enum NormalEstimationMethod { COVARIANCE_MATRIX, AVERAGE_3D_GRADIENT, AVERAGE_DEPTH_CHANGE, SIMPLE_3D_GRADIENT };
    IntegralImageNormalEstimation();
    void setNormalEstimationMethod(NormalEstimationMethod normal_estimation_method);
    void setInputCloud(const PointCloudConstPtr &cloud);
    void setNormalSmoothingSize(float normal_smoothing_size);
    void compute(pcl::PointCloud<PointOutT> &output);
};
}  // namespace pcl
