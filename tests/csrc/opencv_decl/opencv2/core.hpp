// DECLARATIONS ONLY: cv::Mat (OpenCV 3.x, opencv2/core/mat.hpp) as far as include/KeypointLearning.h touches it under
// -DKPL_USE_OPENCV and tools/refgen/refgen_driver.cpp reads the matrix computePointsForTrainingFeatures returns.
// Purpose: a SYNTAX check of those configurations of this repo's own code (tests/test_header_syntax.py,
// tests/test_refgen_kit.py); no OpenCV exists in this image.  Nothing is linked or executed, nothing of the reference is
// compiled against it, and it pins no behaviour.
#pragma once
#define CV_32F 5
namespace cv {
class Mat {
public:
    Mat();
    Mat(int rows, int cols, int type);
    int rows, cols;
    template <typename T> T *ptr(int row = 0);
    template <typename T> const T *ptr(int row = 0) const;
    template <typename T> T &at(int row, int col);
    bool empty() const;
};
}  // namespace cv
