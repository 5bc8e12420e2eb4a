// BVH.cpp — BVH2 binned-SAH builder, one triangle per leaf, task-parallel.
//
// Algorithm (and therefore output) of /root/reference/Nexus/src/Geometry/BVH/BVH.cpp:13-210:
// 8 bins per axis, split = argmin over 3 axes x 7 planes of leftCount*leftArea + rightCount*rightArea,
// no termination on cost, midpoint ("split in half") fallback when binning cannot separate, in-place
// swap-with-last partition.
//
// Re-design: the reference recurses while push_back-ing into one vector, which serialises the build.
// Because every leaf holds exactly one triangle, a subtree over k triangles has exactly 2k-1 nodes, so
// the index the serial recursion would give to every node is known up front: if the vector held S
// nodes when a node over k triangles with kL left triangles is subdivided, its children land at S and
// S+1, the left subtree's descendants start at S+2 and the right subtree's at S+2*kL.  Subtrees are
// therefore built by independent threads straight into a pre-sized array, and the three per-axis
// passes of the reference are fused into one pass over the node's triangles (same numbers: min/max
// and integer counts do not depend on visiting order).
#include "nexus/BVH.h"

#include <atomic>
#include <numeric>
#include <thread>

namespace nexus {

namespace {

constexpr int BINS = 8;

struct Builder {
    BVH2& bvh;
    std::vector<float3> centroid;
    std::atomic<int> spareThreads;
    static constexpr uint32_t kSpawnThreshold = 1u << 14;

    explicit Builder(BVH2& b, int threads) : bvh(b), spareThreads(threads - 1) {}

    void UpdateNodeBounds(uint32_t nodeIdx)
    {
        BVH2Node& node = bvh.nodes[nodeIdx];
        float3 mn = make_float3(1e30f), mx = make_float3(-1e30f);
        const uint32_t* idx = bvh.triangleIdx.data() + node.firstTriIdx;
        for (uint32_t i = 0; i < node.triCount; i++) {
            const AABB& b = bvh.trianglesAABB[idx[i]];
            mn = fminf(mn, b.bMin);
            mx = fmaxf(mx, b.bMax);
        }
        node.aabbMin = mn;
        node.aabbMax = mx;
    }

    // Returns the axis (-1: no axis separates the centroids) and the split position.
    int FindBestSplitPlane(const BVH2Node& node, double& splitPos) const
    {
        const uint32_t* idx = bvh.triangleIdx.data() + node.firstTriIdx;
        const uint32_t n = node.triCount;

        float3 cmin = make_float3(1e30f), cmax = make_float3(-1e30f);
        for (uint32_t i = 0; i < n; i++) {
            const float3 c = centroid[idx[i]];
            cmin = fminf(cmin, c);
            cmax = fmaxf(cmax, c);
        }

        struct Bin { AABB bounds; int triCount = 0; };
        Bin bins[3][BINS];
        double scale[3];
        bool live[3];
        for (int a = 0; a < 3; a++) {
            live[a] = comp(cmin, a) != comp(cmax, a);
            scale[a] = live[a] ? static_cast<double>(static_cast<float>(BINS) / (comp(cmax, a) - comp(cmin, a))) : 0.0;
        }
        for (uint32_t i = 0; i < n; i++) {
            const uint32_t t = idx[i];
            const float3 c = centroid[t];
            const AABB& tb = bvh.trianglesAABB[t];
            for (int a = 0; a < 3; a++) {
                if (!live[a]) continue;
                // (clamped before the conversion: a NaN or infinite centroid — a damaged mesh — must not index outside the bins;
                //  finite centroids give 0 <= f as before)
                const double f = (comp(c, a) - comp(cmin, a)) * scale[a];
                const int binIdx = !(f >= 0.0) ? 0 : (f >= static_cast<double>(BINS) ? BINS - 1 : static_cast<int>(f));
                Bin& bin = bins[a][binIdx];
                bin.triCount++;
                bin.bounds.bMin = fminf(bin.bounds.bMin, tb.bMin);
                bin.bounds.bMax = fmaxf(bin.bounds.bMax, tb.bMax);
            }
        }

        // Cost of cutting after bin k = (triangles at or below k) x (area of their box) + the same for the bins above k.
        // The seven "at or below" states are one running sweep upwards, the seven "above" states one sweep downwards;
        // both are kept so that the cost loop below reads them side by side.
        struct Side {
            AABB box;
            int tris = 0;
            void Take(const Bin& b) { box.Grow(b.bounds); tris += b.triCount; }
            float Weighted() const { return static_cast<float>(tris) * box.Area(); }
        };
        float bestCost = 1e30f;
        int axis = -1;
        for (int a = 0; a < 3; a++) {
            if (!live[a]) continue;
            Side below[BINS - 1], above[BINS - 1];
            Side run;
            for (int k = 0; k < BINS - 1; k++) { run.Take(bins[a][k]); below[k] = run; }
            run = Side();
            for (int k = BINS - 1; k >= 1; k--) { run.Take(bins[a][k]); above[k - 1] = run; }
            const float lo = comp(cmin, a);
            const double step = static_cast<double>((comp(cmax, a) - lo) / static_cast<float>(BINS));
            for (int k = 0; k < BINS - 1; k++) {
                const float cost = below[k].Weighted() + above[k].Weighted();
                if (cost < bestCost) {
                    bestCost = cost;
                    axis = a;
                    splitPos = lo + step * (k + 1);
                }
            }
        }
        return axis;
    }

    // `base` = number of nodes the serial recursion would already have emitted when it reaches nodeIdx.
    void Subdivide(uint32_t nodeIdx, uint32_t base)
    {
        for (;;) {
            BVH2Node& node = bvh.nodes[nodeIdx];
            const uint32_t first = node.firstTriIdx, count = node.triCount;
            if (count == 1) return;

            uint32_t leftCount = 0;
            double splitPos = 0.0;
            const int axis = FindBestSplitPlane(node, splitPos);
            if (axis != -1) {
                uint32_t* idx = bvh.triangleIdx.data();
                int i = static_cast<int>(first);
                int j = i + static_cast<int>(count) - 1;
                while (i <= j) {
                    if (static_cast<double>(comp(centroid[idx[i]], axis)) < splitPos) i++;
                    else std::swap(idx[i], idx[j--]);
                }
                leftCount = static_cast<uint32_t>(i) - first;
            }
            if (leftCount == 0 || leftCount == count) leftCount = count / 2;  // SplitNodeInHalf

            const uint32_t l = base, r = base + 1;
            BVH2Node& left = bvh.nodes[l];
            BVH2Node& right = bvh.nodes[r];
            left.firstTriIdx = first;
            left.triCount = leftCount;
            right.firstTriIdx = first + leftCount;
            right.triCount = count - leftCount;
            node.leftNode = l;
            node.triCount = 0;
            UpdateNodeBounds(l);
            UpdateNodeBounds(r);

            const uint32_t leftBase = base + 2, rightBase = base + 2 * leftCount;
            if (count >= kSpawnThreshold && spareThreads.fetch_sub(1) > 0) {
                std::thread t([this, l, leftBase] { Subdivide(l, leftBase); });
                Subdivide(r, rightBase);
                t.join();
                spareThreads.fetch_add(1);
                return;
            }
            if (count >= kSpawnThreshold) spareThreads.fetch_add(1);
            Subdivide(l, leftBase);
            nodeIdx = r;  // tail call on the right child
            base = rightBase;
        }
    }
};

}  // namespace

BVH2::BVH2(const std::vector<Triangle>& tri) : triangles(tri), triangleIdx(tri.size()) {}

void BVH2::Build(unsigned threads)
{
    const size_t n = triangles.size();
    nodes.clear();
    trianglesAABB.clear();
    if (n == 0) return;
    if (threads == 0) threads = std::max(1u, std::thread::hardware_concurrency());

    std::iota(triangleIdx.begin(), triangleIdx.end(), 0u);
    trianglesAABB.resize(n);
    Builder b(*this, static_cast<int>(threads));
    b.centroid.resize(n);
    for (size_t i = 0; i < n; i++) {
        AABB box;
        box.Grow(triangles[i].pos0);
        box.Grow(triangles[i].pos1);
        box.Grow(triangles[i].pos2);
        trianglesAABB[i] = box;
        b.centroid[i] = triangles[i].centroid;
    }
    nodes.assign(2 * n - 1, BVH2Node{});
    nodes[0].firstTriIdx = 0;
    nodes[0].triCount = static_cast<uint32_t>(n);
    b.UpdateNodeBounds(0);
    b.Subdivide(0, 1);
}

}  // namespace nexus
