// TEST DOUBLE -- see tests/cpp/mock/README.md
#pragma once
#include <memory>
#include <vector>
namespace pcl {
template <typename T>
using shared_ptr = std::shared_ptr<T>;
template <typename PointT>
class PointCloud {
 public:
  using Ptr = shared_ptr<PointCloud<PointT>>;
  using ConstPtr = shared_ptr<const PointCloud<PointT>>;
  std::vector<PointT> points;
  unsigned width = 0, height = 1;
  bool is_dense = true;
  size_t size() const { return points.size(); }
  bool empty() const { return points.empty(); }
  void resize(size_t n) { points.resize(n); }
  PointT& operator[](size_t i) { return points[i]; }
  const PointT& operator[](size_t i) const { return points[i]; }
  void push_back(const PointT& p) { points.push_back(p); }
  Ptr makeShared() const { return Ptr(new PointCloud<PointT>(*this)); }
};
}  // namespace pcl
