// TEST DOUBLE -- see tests/cpp/mock/README.md
#pragma once
namespace pcl {
struct PointXYZ { float x = 0, y = 0, z = 0, pad = 1; PointXYZ() = default; PointXYZ(float a, float b, float c) : x(a), y(b), z(c) {} };
struct PointXYZI { float x = 0, y = 0, z = 0, pad = 1; float intensity = 0, p1 = 0, p2 = 0, p3 = 0; };
static_assert(sizeof(PointXYZ) == 16 && sizeof(PointXYZI) == 32, "PCL point layouts");
}  // namespace pcl
