// nx_refit.hip — dynamic transforms on the device: new object-to-world matrices of existing instances go in, and the
// instance table, the traversal records and the TLAS BVH8 (same topology, refitted bounds) are updated in HBM without the
// scene or the tree making a round trip through the host.
//
// Reference behaviour being replaced: every edit re-runs MeshInstance::SetTransform -> BVHInstance::SetTransform
// (/root/reference/Nexus/src/Geometry/BVH/BVHInstance.cpp:4-29: inverse matrix, world bounds from the 8 transformed corners
// of the BLAS root's quantisation frame), then an O(n^2) agglomerative TLAS rebuild on the CPU and a full re-upload
// (Scene/Scene.cpp:29-55, Geometry/BVH/TLAS.cpp:13-100).  Here: one thread per moved instance does the SetTransform
// arithmetic (same operation order as the host classes: results are bit-identical to nexus::BVHInstance::SetTransform), then
// one workgroup sweeps the TLAS bottom-up level by level (levels are independent inside; a barrier between them),
// re-deriving each node's quantisation frame and child boxes exactly as nexus::collapse::Refit does on the host.
#define NX_KERNEL_TU 1
#include "nx_device.h"
#include "nx_instbox.h"
#include "nx_math.h"

namespace nxd {

constexpr int kRefitBlock = 1024;

struct Box {
    float lo[3], hi[3];
};

NXD void box_empty(Box& b)
{
    for (int a = 0; a < 3; a++) { b.lo[a] = 1e30f; b.hi[a] = -1e30f; }  // nexus::AABB's empty box
}
NXD void box_grow(Box& b, const float* lo, const float* hi)
{
    for (int a = 0; a < 3; a++) { b.lo[a] = fminf(b.lo[a], lo[a]); b.hi[a] = fmaxf(b.hi[a], hi[a]); }
}

// nexus::Mat4::Inverted (host/Math.cpp): cofactor expansion, every product and sum in the same order
NXD float cof(const float* m, int a, int b, int c) { return m[a] * m[b] * m[c]; }
// returns whether the matrix is singular (det == 0: the identity comes out, as nexus::Mat4's default)
NXD bool mat4_invert(const float* m, float* out)
{
    float inv[16];
    inv[0] = cof(m, 5, 10, 15) - cof(m, 5, 11, 14) - cof(m, 9, 6, 15) + cof(m, 9, 7, 14) + cof(m, 13, 6, 11) - cof(m, 13, 7, 10);
    inv[1] = -cof(m, 1, 10, 15) + cof(m, 1, 11, 14) + cof(m, 9, 2, 15) - cof(m, 9, 3, 14) - cof(m, 13, 2, 11) + cof(m, 13, 3, 10);
    inv[2] = cof(m, 1, 6, 15) - cof(m, 1, 7, 14) - cof(m, 5, 2, 15) + cof(m, 5, 3, 14) + cof(m, 13, 2, 7) - cof(m, 13, 3, 6);
    inv[3] = -cof(m, 1, 6, 11) + cof(m, 1, 7, 10) + cof(m, 5, 2, 11) - cof(m, 5, 3, 10) - cof(m, 9, 2, 7) + cof(m, 9, 3, 6);
    inv[4] = -cof(m, 4, 10, 15) + cof(m, 4, 11, 14) + cof(m, 8, 6, 15) - cof(m, 8, 7, 14) - cof(m, 12, 6, 11) + cof(m, 12, 7, 10);
    inv[5] = cof(m, 0, 10, 15) - cof(m, 0, 11, 14) - cof(m, 8, 2, 15) + cof(m, 8, 3, 14) + cof(m, 12, 2, 11) - cof(m, 12, 3, 10);
    inv[6] = -cof(m, 0, 6, 15) + cof(m, 0, 7, 14) + cof(m, 4, 2, 15) - cof(m, 4, 3, 14) - cof(m, 12, 2, 7) + cof(m, 12, 3, 6);
    inv[7] = cof(m, 0, 6, 11) - cof(m, 0, 7, 10) - cof(m, 4, 2, 11) + cof(m, 4, 3, 10) + cof(m, 8, 2, 7) - cof(m, 8, 3, 6);
    inv[8] = cof(m, 4, 9, 15) - cof(m, 4, 11, 13) - cof(m, 8, 5, 15) + cof(m, 8, 7, 13) + cof(m, 12, 5, 11) - cof(m, 12, 7, 9);
    inv[9] = -cof(m, 0, 9, 15) + cof(m, 0, 11, 13) + cof(m, 8, 1, 15) - cof(m, 8, 3, 13) - cof(m, 12, 1, 11) + cof(m, 12, 3, 9);
    inv[10] = cof(m, 0, 5, 15) - cof(m, 0, 7, 13) - cof(m, 4, 1, 15) + cof(m, 4, 3, 13) + cof(m, 12, 1, 7) - cof(m, 12, 3, 5);
    inv[11] = -cof(m, 0, 5, 11) + cof(m, 0, 7, 9) + cof(m, 4, 1, 11) - cof(m, 4, 3, 9) - cof(m, 8, 1, 7) + cof(m, 8, 3, 5);
    inv[12] = -cof(m, 4, 9, 14) + cof(m, 4, 10, 13) + cof(m, 8, 5, 14) - cof(m, 8, 6, 13) - cof(m, 12, 5, 10) + cof(m, 12, 6, 9);
    inv[13] = cof(m, 0, 9, 14) - cof(m, 0, 10, 13) - cof(m, 8, 1, 14) + cof(m, 8, 2, 13) + cof(m, 12, 1, 10) - cof(m, 12, 2, 9);
    inv[14] = -cof(m, 0, 5, 14) + cof(m, 0, 6, 13) + cof(m, 4, 1, 14) - cof(m, 4, 2, 13) - cof(m, 12, 1, 6) + cof(m, 12, 2, 5);
    inv[15] = cof(m, 0, 5, 10) - cof(m, 0, 6, 9) - cof(m, 4, 1, 10) + cof(m, 4, 2, 9) + cof(m, 8, 1, 6) - cof(m, 8, 2, 5);
    const float det = m[0] * inv[0] + m[1] * inv[4] + m[2] * inv[8] + m[3] * inv[12];
    if (det != 0) {
        const float invdet = 1.0f / det;
        for (int i = 0; i < 16; ++i) out[i] = inv[i] * invdet;
        return false;
    }
    for (int i = 0; i < 16; ++i) out[i] = (i % 5 == 0) ? 1.0f : 0.0f;  // nexus::Mat4's default: identity
    return true;
}

// BVHInstance::SetTransform for `count` instances: thread k handles instance ids[k] with matrix transforms[16 k .. 16 k + 15]
// (kernel parameters are plain pointers: an address-space qualifier in a kernel signature would mangle the device symbol
//  differently from the host stub, which is compiled without it)
__global__ void __launch_bounds__(256) instance_transform_kernel(const DeviceState* __restrict__ S, nx_bvh_instance* instancesArg, InstTrav* travArg,
                                                                 const uint32_t* __restrict__ leafOfInstance, const uint32_t* __restrict__ ids,
                                                                 const float* __restrict__ transforms, const uint32_t count, InstBox* __restrict__ tightBoxes,
                                                                 ShadeInst* shadeInstArg)
{
    NX_G ShadeInst* shadeInst = (NX_G ShadeInst*)shadeInstArg;  // nullptr: the host rebuilds the shading records before the next render
    NX_G nx_bvh_instance* instances = (NX_G nx_bvh_instance*)instancesArg;
    NX_G InstTrav* trav = (NX_G InstTrav*)travArg;
    for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < count; k += gridDim.x * blockDim.x) {
        const uint32_t id = ids[k];
        float m[16], inv[16];
        for (int i = 0; i < 16; i++) m[i] = transforms[16 * (size_t)k + i];
        const bool singular = mat4_invert(m, inv);
        NX_G nx_bvh_instance* inst = &instances[id];
        // world bounds: the 8 transformed corners of the BLAS root's quantisation frame [p, p + 2^(e-127) * 255]
        const NX_G uint4* root = S->blas[inst->bvhIdx].nodes;
        const uint4 n0 = root[0];
        const float lo[3] = {__uint_as_float(n0.x), __uint_as_float(n0.y), __uint_as_float(n0.z)};
        const float steps = 255.0f;
        float hi[3];
        for (int a = 0; a < 3; a++) hi[a] = lo[a] + ldexpf(1.0f, (int)((n0.w >> (8 * a)) & 0xffu) - 127) * steps;
        Box wb;
        box_empty(wb);
        for (int corner = 0; corner < 8; corner++) {
            const float cx = (corner & 1) ? hi[0] : lo[0], cy = (corner & 2) ? hi[1] : lo[1], cz = (corner & 4) ? hi[2] : lo[2];
            const float p[3] = {m[0] * cx + m[1] * cy + m[2] * cz + m[3] * 1.0f, m[4] * cx + m[5] * cy + m[6] * cz + m[7] * 1.0f,
                                m[8] * cx + m[9] * cy + m[10] * cz + m[11] * 1.0f};
            box_grow(wb, p, p);
        }
        for (int i = 0; i < 16; i++) { inst->transform.cell[i] = m[i]; inst->invTransform.cell[i] = inv[i]; }
        if (shadeInst)
            for (int i = 0; i < 12; i++) { shadeInst[id].transform[i] = m[i]; shadeInst[id].invTransform[i] = inv[i]; }
        for (int a = 0; a < 3; a++) { inst->boundsMin[a] = wb.lo[a]; inst->boundsMax[a] = wb.hi[a]; }
        if (tightBoxes && kNodeStride == 5) {  // a TLAS built on the device keeps its tighter boxes (nx_instbox.h) through the refit
            InstBox tb;
            for (int a = 0; a < 3; a++) { tb.lo[a] = wb.lo[a]; tb.hi[a] = wb.hi[a]; }
            tighten_instance_box(reinterpret_cast<const NX_G nx_bvh8_node*>(root), m, tb, singular);
            tightBoxes[id] = tb;
        }
        NX_G InstTrav* t = &trav[leafOfInstance[id]];
        t->r0 = make_float4(inv[0], inv[1], inv[2], inv[3]);
        t->r1 = make_float4(inv[4], inv[5], inv[6], inv[7]);
        t->r2 = make_float4(inv[8], inv[9], inv[10], inv[11]);
        t->flags = rows_are_identity(inv) ? kInstIdentity : 0u;
    }
}

NXD uint32_t quantize(float v)  // nexus::collapse Quantize
{
    if (!(v == v)) return 0u;
    if (v <= 0.0f) return 0u;
    if (v >= 255.0f) return 255u;
    return (uint32_t)v;
}

// ceil(log2f(x)) as the host computes it (std::log2 on a float, then std::ceil).  glibc's log2f is within 0.75 ulp, the
// double-precision logarithm rounded to float is the correctly rounded value: the two agree except for arguments within a
// few float steps of a power of two whose logarithm falls near a rounding midpoint (probability ~1e-7 per call).
NXD float ceil_log2(float x) { return ceilf((float)log2((double)x)); }

// nexus::collapse::Refit on the device.  `order` lists the node indices grouped by depth, deepest level first;
// levelStart[l] .. levelStart[l + 1] is level l of that list.  One workgroup: levels are separated by a barrier.
__global__ void __launch_bounds__(kRefitBlock) tlas_refit_kernel(nx_bvh8_node* nodesArg, const uint32_t* __restrict__ primIdx,
                                                                 const nx_bvh_instance* instancesArg, const uint32_t* __restrict__ order,
                                                                 const uint32_t* __restrict__ levelStart, const uint32_t levels, Box* __restrict__ nodeBox,
                                                                 const InstBox* __restrict__ tightBoxes)
{
    NX_G nx_bvh8_node* nodes = (NX_G nx_bvh8_node*)nodesArg;
    const NX_G nx_bvh_instance* instances = (const NX_G nx_bvh_instance*)instancesArg;
    for (uint32_t l = 0; l < levels; l++) {
        for (uint32_t i = levelStart[l] + threadIdx.x; i < levelStart[l + 1]; i += blockDim.x) {
            const uint32_t k = order[i];
            NX_G nx_bvh8_node* node = &nodes[k];
            const uint32_t imask = node->imask, childBase = node->childBaseIdx, primBase = node->triangleBaseIdx;
            Box childBox[8], nb;
            bool used[8];
            box_empty(nb);
            for (int s = 0; s < 8; s++) {
                used[s] = false;
                Box cb;
                box_empty(cb);
                const uint32_t meta = node->meta[s];
                if (imask & (1u << s)) {
                    cb = nodeBox[childBase + (uint32_t)__popc(imask & ((1u << s) - 1u))];
                } else if (meta) {
                    const uint32_t first = primBase + (meta & 0x1fu);
                    const int cnt = __popc(meta >> 5);
                    for (int j = 0; j < cnt; j++) {
                        const uint32_t instanceId = primIdx[first + (uint32_t)j];
                        if (tightBoxes) {  // (a device-built TLAS: the boxes it was built from, kept up to date by the transform kernel)
                            const InstBox tb = tightBoxes[instanceId];
                            box_grow(cb, tb.lo, tb.hi);
                            continue;
                        }
                        const NX_G nx_bvh_instance* in = &instances[instanceId];
                        const float lo[3] = {in->boundsMin[0], in->boundsMin[1], in->boundsMin[2]}, hi[3] = {in->boundsMax[0], in->boundsMax[1], in->boundsMax[2]};
                        box_grow(cb, lo, hi);
                    }
                } else {
                    continue;
                }
                used[s] = true;
                childBox[s] = cb;
                box_grow(nb, cb.lo, cb.hi);
            }
            nodeBox[k] = nb;
            const float denom = 1.0f / 255.0f;
            float ex[3], invScale[3];
            uint32_t ebytes = 0;
            for (int a = 0; a < 3; a++) {
                ex[a] = ceil_log2((nb.hi[a] - nb.lo[a]) * denom);
                // exponent byte of exp2f(ex): 0 below the normal range (and for -inf: a degenerate axis), 255 above it
                uint32_t e = 0;
                if (ex[a] == ex[a] && ex[a] > -127.0f) e = ex[a] >= 128.0f ? 255u : (uint32_t)((int)ex[a] + 127);
                ebytes |= e << (8 * a);
                // 1 / 2^ex, exactly: inf for ex = -inf (degenerate axis), as 1.0f / std::pow(2.0f, ex)
                const float pw = ex[a] == ex[a] ? (ex[a] < -200.0f ? 0.0f : (ex[a] > 200.0f ? __uint_as_float(0x7f800000u) : ldexpf(1.0f, (int)ex[a]))) : ex[a];
                invScale[a] = 1.0f / pw;
            }
            node->p[0] = nb.lo[0]; node->p[1] = nb.lo[1]; node->p[2] = nb.lo[2];
            node->e[0] = (uint8_t)(ebytes & 0xffu); node->e[1] = (uint8_t)((ebytes >> 8) & 0xffu); node->e[2] = (uint8_t)((ebytes >> 16) & 0xffu);
            for (int s = 0; s < 8; s++) {
                if (!used[s]) continue;
                const Box& cb = childBox[s];
                node->qlox[s] = (uint8_t)quantize(floorf((cb.lo[0] - nb.lo[0]) * invScale[0]));
                node->qloy[s] = (uint8_t)quantize(floorf((cb.lo[1] - nb.lo[1]) * invScale[1]));
                node->qloz[s] = (uint8_t)quantize(floorf((cb.lo[2] - nb.lo[2]) * invScale[2]));
                node->qhix[s] = (uint8_t)quantize(ceilf((cb.hi[0] - nb.lo[0]) * invScale[0]));
                node->qhiy[s] = (uint8_t)quantize(ceilf((cb.hi[1] - nb.lo[1]) * invScale[1]));
                node->qhiz[s] = (uint8_t)quantize(ceilf((cb.hi[2] - nb.lo[2]) * invScale[2]));
            }
        }
        __syncthreads();  // the next (shallower) level reads the boxes just written; one workgroup, so this orders them
    }
}

// The material code of every traversal record (nx_device.h, kHitCodeShift): record k — TLAS leaf order — belongs to instance
// instIdx, whose shading record holds a copy of its material; type + 1 goes above the instance index, where the closest-hit
// kernel finds it in the register that carries the hit's instance (nx_trace.hip, flush).  Run by the host after it has rebuilt
// the shading records (refresh_shade_inst: an instance, BLAS or material table changed).
__global__ void __launch_bounds__(256) inst_code_kernel(const DeviceState* __restrict__ S, InstTrav* travArg, const ShadeInst* shadeInstArg, const uint32_t count)
{
    NX_G InstTrav* trav = (NX_G InstTrav*)travArg;
    const NX_G ShadeInst* shadeInst = (const NX_G ShadeInst*)shadeInstArg;
    for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < count; k += gridDim.x * blockDim.x) {
        const uint32_t inst = trav[k].instIdx & kHitInstMask;
        const int type = (int)shadeInst[inst].material.type;
        trav[k].instIdx = inst | ((type >= 0 && type <= 3) ? (uint32_t)(type + 1) << kHitCodeShift : 0u);
    }
}

const void* inst_code_kernel_ptr() { return (const void*)inst_code_kernel; }
const void* instance_transform_kernel_ptr() { return (const void*)instance_transform_kernel; }
const void* tlas_refit_kernel_ptr() { return (const void*)tlas_refit_kernel; }

// the device-side layouts this translation unit was compiled with (nx_device.h layout_stamp; compared by nxhip_create)
uint64_t layout_stamp_refit() { return layout_stamp(); }

}  // namespace nxd
