// Translation unit of the -DKPL_USE_PCL syntax check (tests/test_header_syntax.py): instantiates the drop-in
// class the way /root/reference/src/main_test_detector.cpp:93-95,123-187 does, against declaration-only PCL
// headers (tests/csrc/pcl_decl).  Compiled with -fsyntax-only; never linked, never run.
#include "KeypointLearning.h"

typedef pcl::keypoints::KeypointLearningDetector<pcl::PointXYZ, pcl::PointXYZI> Detector;

void use(pcl::PointCloud<pcl::PointXYZ>::Ptr cloud, pcl::PointCloud<pcl::Normal>::Ptr normals,
         pcl::PointCloud<pcl::PointXYZI> &keypoints, pcl::PointIndicesConstPtr some) {
    Detector::Ptr detector(new Detector());
    detector->setNAnnulus(5);
    detector->setNBins(10);
    detector->setNonMaxima(true);
    detector->setNonMaxRadius(4.0);
    detector->setNonMaximaDrawsRemove(false);
    detector->setNonMaximaDrawsThreshold(0.0f);
    detector->setPredictionThreshold(0.85f);
    detector->setRadiusSearch(20.0);
    detector->setSearchMethod(Detector::KdTree::Ptr(new pcl::search::KdTree<pcl::PointXYZ>()));
    if (!detector->loadForest("forest.yaml.gz")) return;
    detector->setInputCloud(cloud);
    detector->setNormals(normals);
    detector->compute(keypoints);
    pcl::PointIndicesConstPtr idx = detector->getKeypointsIndices();
    kpl::FeatureMat rows = detector->computePointsForTrainingFeatures(some);
    (void)idx;
    (void)rows;
}

template class pcl::keypoints::KeypointLearningDetector<pcl::PointXYZ, pcl::PointXYZI>;   // every member body
