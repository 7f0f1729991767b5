// DECLARATIONS ONLY -- see ../README.md.  pcl::PointCloud lives in point_types.h of this directory.
#pragma once
#include <pcl/point_types.h>
