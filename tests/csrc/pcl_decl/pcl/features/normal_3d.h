// DECLARATIONS ONLY -- see ../../README.md.  pcl::NormalEstimation (PCL 1.8: pcl/features/normal_3d.h, feature.h) as far as
// tools/refgen/refgen_driver.cpp drives it (the calls of /root/reference/src/main_test_detector.cpp:162-169 and
// include/impl/KeypointLearning.hpp:130-137).
#pragma once
#include <pcl/search/kdtree.h>
namespace pcl {
template <typename PointInT, typename PointOutT>
class NormalEstimation {
public:
    typedef typename pcl::PointCloud<PointInT>::ConstPtr PointCloudConstPtr;
    typedef typename pcl::search::Search<PointInT>::Ptr KdTreePtr;
    NormalEstimation();
    void setInputCloud(const PointCloudConstPtr &cloud);
    void setSearchMethod(const KdTreePtr &tree);
    void setKSearch(int k);
    void setRadiusSearch(double radius);
    void setViewPoint(float vpx, float vpy, float vpz);
    void compute(pcl::PointCloud<PointOutT> &output);
};
}  // namespace pcl
