// nexus/BVH8Builder.h — BVH2 -> BVH8 collapse by SAH dynamic programming.
// Same algorithm and output bytes as /root/reference/Nexus/src/Geometry/BVH/BVH8Builder.h:8-75,
// BVH8Builder.cpp:7-393; the DP table is a flat bottom-up array instead of a memoised recursion over
// vector<vector<>>, and the TLAS variant (TLASBuilder) shares the implementation.
// One deliberate deviation: quantised upper bounds are clamped to 255 (the reference BLAS path lets
// ceil() == 256 wrap to 0, which would drop geometry; its TLAS path clamps, TLASBuilder.cpp:314-316).
#pragma once

#include <vector>

#include "BVH.h"
#include "BVH8.h"

namespace nexus {

class BVH8Builder {
public:
    explicit BVH8Builder(const std::vector<Triangle>& triangles);

    enum struct Decision : int8_t { UNDEFINED = -1, LEAF, INTERNAL, DISTRIBUTE };

    void Init(unsigned threads = 0);  // builds the BVH2 and the cost table
    BVH8 Build();

    const BVH2& GetBVH2() const { return m_Bvh2; }

private:
    BVH2 m_Bvh2;
    std::vector<uint8_t> m_EvalStorage;  // collapse::Eval[nodes][7]
};

}  // namespace nexus
