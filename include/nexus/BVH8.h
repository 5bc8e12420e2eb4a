// nexus/BVH8.h — compressed wide BVH (Ylitie, Karras, Laine 2017), 80-byte nodes.
// Mirrors /root/reference/Nexus/src/Geometry/BVH/BVH8.h:18-75.  The host node IS the device node
// (nx_bvh8_node): upload is a raw copy.
#pragma once

#include <cstdint>
#include <vector>

#include "../nexus_pod.h"
#include "Triangle.h"

namespace nexus {

constexpr float C_PRIM = 0.3f;  // cost of a ray-primitive intersection
constexpr float C_NODE = 1.0f;  // cost of a ray-node intersection
constexpr int P_MAX = 3;        // maximum leaf size
constexpr int N_Q = 8;          // bits per quantised child coordinate

using BVH8Node = nx_bvh8_node;

struct BVH8 {
    BVH8() = default;
    explicit BVH8(const std::vector<Triangle>& tri);

    std::vector<Triangle> triangles;
    std::vector<uint32_t> triangleIdx;  // triangle (BLAS) or instance (TLAS) ids in leaf order
    std::vector<BVH8Node> nodes;

    // Device handle once uploaded through the C-ABI device layer (nxhip_upload_blas); -1 before.
    int32_t deviceBlasId = -1;
};

}  // namespace nexus
