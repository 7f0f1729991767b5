// DECLARATIONS ONLY -- see ../../README.md.  pcl::search::Search / KdTree are declared in pcl/keypoints/keypoint.h here.
#pragma once
#include <pcl/keypoints/keypoint.h>
