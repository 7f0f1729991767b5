// DECLARATIONS ONLY -- see ../../README.md.  pcl/io/pcd_io.h (PCL 1.8) as far as tools/refgen/refgen_driver.cpp uses it.
#pragma once
#include <pcl/point_types.h>
namespace pcl {
namespace io {
template <typename PointT>
int loadPCDFile(const std::string &file_name, pcl::PointCloud<PointT> &cloud);
}  // namespace io
}  // namespace pcl
