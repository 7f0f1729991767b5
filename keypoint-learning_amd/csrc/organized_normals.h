// Normals of an organized cloud (pcl::IntegralImageNormalEstimation, SIMPLE_3D_GRADIENT) on the device:
// see organized_normals.hip.  Internal to libkpl; the C-ABI entry points are in include/kpl.h.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>

namespace kpl {

struct OrganizedView {
    const char *xyz;           // device, row-major width x height, byte stride xs (>= 12)
    size_t xs;
    int W, H;
    float smoothing;           // setNormalSmoothingSize
    float vp[3];               // viewpoint (sensor origin)
    char *normals;             // device, byte stride ns: 3 floats per pixel (NaN where PCL leaves NaN)
    size_t ns;
    char *curvature;           // device, byte stride cs, or null: all NaN (SIMPLE_3D_GRADIENT computes none)
    size_t cs;
    // scratch, set by launch_organized_normals
    unsigned char *change;
    float *dist;
    double *ii;
};

size_t organized_normals_scratch_bytes(int W, int H);
void launch_organized_normals(OrganizedView v, void *scratch, hipStream_t st);

}  // namespace kpl
