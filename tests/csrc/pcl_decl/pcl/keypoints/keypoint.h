// DECLARATIONS ONLY -- see ../../README.md.  pcl/pcl_base.h, pcl/search/search.h, pcl/search/kdtree.h and
// pcl/keypoints/keypoint.h (PCL 1.8) as far as include/KeypointLearning.h uses them.
#pragma once
#include <pcl/point_types.h>

namespace pcl {
namespace search {
template <typename PointT>
class Search {
public:
    typedef boost::shared_ptr<Search<PointT>> Ptr;
    typedef boost::shared_ptr<const Search<PointT>> ConstPtr;
    Search(const std::string &name = "", bool sorted = false);
    virtual ~Search();
    virtual void setSortedResults(bool sorted_results);
    virtual bool getSortedResults();
};
template <typename PointT>
class KdTree : public Search<PointT> {
public:
    typedef boost::shared_ptr<KdTree<PointT>> Ptr;
    KdTree(bool sorted = true);
};
}  // namespace search

template <typename PointT>
class PCLBase {
public:
    typedef pcl::PointCloud<PointT> PointCloud;
    typedef typename PointCloud::ConstPtr PointCloudConstPtr;
    virtual ~PCLBase();
    virtual void setInputCloud(const PointCloudConstPtr &cloud);

protected:
    PointCloudConstPtr input_;
    bool initCompute();
    bool deinitCompute();
};

template <typename PointInT, typename PointOutT>
class Keypoint : public PCLBase<PointInT> {
public:
    typedef boost::shared_ptr<Keypoint<PointInT, PointOutT>> Ptr;
    typedef typename pcl::search::Search<PointInT> KdTree;
    typedef typename pcl::search::Search<PointInT>::Ptr KdTreePtr;
    typedef pcl::PointCloud<PointInT> PointCloudIn;
    typedef typename PointCloudIn::Ptr PointCloudInPtr;
    typedef typename PointCloudIn::ConstPtr PointCloudInConstPtr;
    typedef pcl::PointCloud<PointOutT> PointCloudOut;

    Keypoint();
    virtual ~Keypoint();
    virtual void setSearchSurface(const PointCloudInConstPtr &cloud);
    inline void setSearchMethod(const KdTreePtr &tree) { tree_ = tree; }
    inline KdTreePtr getSearchMethod() { return tree_; }
    inline void setKSearch(int k) { k_ = k; }
    inline void setRadiusSearch(double radius) { search_radius_ = radius; }
    inline pcl::PointIndicesConstPtr getKeypointsIndices() { return keypoints_indices_; }
    void compute(PointCloudOut &output);

protected:
    using PCLBase<PointInT>::input_;
    virtual bool initCompute();
    std::string name_;
    PointCloudInConstPtr surface_;
    KdTreePtr tree_;
    double search_parameter_;
    double search_radius_;
    int k_;
    pcl::PointIndicesPtr keypoints_indices_;
    virtual void detectKeypoints(PointCloudOut &output) = 0;
};
}  // namespace pcl
