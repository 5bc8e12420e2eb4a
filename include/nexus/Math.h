// nexus/Math.h — host-side vector / matrix / AABB types of the kept C++ API surface.
//
// Mirrors the semantics (not the text) of the reference's host math:
//   float3 helpers   /root/reference/Nexus/src/Utils/cuda_math.h (helper_math derivative)
//   Mat4             /root/reference/Nexus/src/Math/Mat4.h:9-231, Math/Mat4.cpp:3-73 (row major)
//   AABB             /root/reference/Nexus/src/Geometry/AABB.h:5-36
// Host arithmetic is plain IEEE float with contraction off (the library is built with -ffp-contract=off),
// so builder output is reproducible byte for byte.
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>

#include "../nexus_pod.h"

namespace nexus {

struct float2 { float x = 0, y = 0; };
struct float3 { float x = 0, y = 0, z = 0; };
struct float4 { float x = 0, y = 0, z = 0, w = 0; };

inline float3 make_float3(float x, float y, float z) { return float3{x, y, z}; }
inline float3 make_float3(float s) { return float3{s, s, s}; }
inline float3 make_float3(const float* p) { return float3{p[0], p[1], p[2]}; }
inline float2 make_float2(float x, float y) { return float2{x, y}; }
inline void store(float* p, const float3& v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }

inline float3 operator+(const float3& a, const float3& b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline float3 operator-(const float3& a, const float3& b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline float3 operator-(const float3& a) { return {-a.x, -a.y, -a.z}; }
inline float3 operator*(const float3& a, const float3& b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
inline float3 operator*(const float3& a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline float3 operator*(float s, const float3& a) { return {s * a.x, s * a.y, s * a.z}; }
inline float3 operator/(const float3& a, float s) { return {a.x / s, a.y / s, a.z / s}; }
inline float3& operator+=(float3& a, const float3& b) { a = a + b; return a; }
inline float3& operator-=(float3& a, const float3& b) { a = a - b; return a; }
inline float3& operator*=(float3& a, float s) { a = a * s; return a; }
inline float3& operator/=(float3& a, float s) { a = a / s; return a; }
inline float dot(const float3& a, const float3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline float3 cross(const float3& a, const float3& b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline float length(const float3& a) { return std::sqrt(dot(a, a)); }
inline float3 normalize(const float3& a) { return a * (1.0f / std::sqrt(dot(a, a))); }
inline float3 fminf(const float3& a, const float3& b) { return {std::fmin(a.x, b.x), std::fmin(a.y, b.y), std::fmin(a.z, b.z)}; }
inline float3 fmaxf(const float3& a, const float3& b) { return {std::fmax(a.x, b.x), std::fmax(a.y, b.y), std::fmax(a.z, b.z)}; }
inline float fmaxf(const float3& a) { return std::fmax(a.x, std::fmax(a.y, a.z)); }
inline float comp(const float3& a, int axis) { return axis == 0 ? a.x : (axis == 1 ? a.y : a.z); }

constexpr double PI = 3.14159265358979323846;

namespace Utils {
inline float ToRadians(float deg) { return static_cast<float>(deg * PI / 180.0f); }
inline float ToDegrees(float rad) { return static_cast<float>(rad * 180.0f / PI); }
template <typename T> inline T SgnE(T v) { return v < T(0) ? T(-1) : T(1); }
}  // namespace Utils

// Row-major 4x4 matrix.  Same cell layout as nx_mat4 so it can be memcpy'd into nx_bvh_instance.
class Mat4 {
public:
    float cell[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};

    float& operator[](int i) { return cell[i]; }
    float operator()(int i, int j) const { return cell[i * 4 + j]; }
    float& operator()(int i, int j) { return cell[i * 4 + j]; }

    static Mat4 Identity() { return Mat4{}; }
    static Mat4 Translate(const float3& p) { Mat4 r; r.cell[3] = p.x; r.cell[7] = p.y; r.cell[11] = p.z; return r; }
    static Mat4 Scale(const float3& s) { Mat4 r; r.cell[0] = s.x; r.cell[5] = s.y; r.cell[10] = s.z; return r; }
    static Mat4 Scale(float s) { return Scale(make_float3(s)); }
    static Mat4 RotateX(float a) { Mat4 r; r.cell[5] = std::cos(a); r.cell[6] = -std::sin(a); r.cell[9] = std::sin(a); r.cell[10] = std::cos(a); return r; }
    static Mat4 RotateY(float a) { Mat4 r; r.cell[0] = std::cos(a); r.cell[2] = std::sin(a); r.cell[8] = -std::sin(a); r.cell[10] = std::cos(a); return r; }
    static Mat4 RotateZ(float a) { Mat4 r; r.cell[0] = std::cos(a); r.cell[1] = -std::sin(a); r.cell[4] = std::sin(a); r.cell[5] = std::cos(a); return r; }

    float3 GetTranslation() const { return make_float3(cell[3], cell[7], cell[11]); }

    Mat4 Transposed() const;  // 3x3 block only, as the reference (Mat4.h:142-149)
    Mat4 Inverted() const;    // cofactor expansion; identity if singular (Mat4.h:151-194)

    float3 TransformVector(const float3& v) const
    {
        return make_float3(cell[0] * v.x + cell[1] * v.y + cell[2] * v.z, cell[4] * v.x + cell[5] * v.y + cell[6] * v.z,
                           cell[8] * v.x + cell[9] * v.y + cell[10] * v.z);
    }
    float3 TransformPoint(const float3& v) const
    {
        return make_float3(cell[0] * v.x + cell[1] * v.y + cell[2] * v.z + cell[3], cell[4] * v.x + cell[5] * v.y + cell[6] * v.z + cell[7],
                           cell[8] * v.x + cell[9] * v.y + cell[10] * v.z + cell[11]);
    }
};

Mat4 operator*(const Mat4& a, const Mat4& b);
bool operator==(const Mat4& a, const Mat4& b);
inline bool operator!=(const Mat4& a, const Mat4& b) { return !(a == b); }
// float4(a, 1) * M, Mat4.cpp:58-69
float3 TransformPosition(const float3& a, const Mat4& M);

struct AABB {
    float3 bMin = make_float3(1e30f);
    float3 bMax = make_float3(-1e30f);

    AABB() = default;
    AABB(const float3& mn, const float3& mx) : bMin(mn), bMax(mx) {}
    void Grow(const float3& p) { bMin = fminf(bMin, p); bMax = fmaxf(bMax, p); }
    void Grow(const AABB& o)
    {
        if (o.bMin.x != 1e30f) { bMin = fminf(bMin, o.bMin); bMax = fmaxf(bMax, o.bMax); }
    }
    // Half surface area (three faces), as the reference uses for SAH.
    float Area() const
    {
        const float3 d = bMax - bMin;
        return d.x * d.y + d.y * d.z + d.x * d.z;
    }
};

}  // namespace nexus
