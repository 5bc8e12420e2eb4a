// Math.cpp — Mat4 operations of the host API (see include/nexus/Math.h).
// Semantics follow /root/reference/Nexus/src/Math/Mat4.h:142-194 and Math/Mat4.cpp:3-73.
#include "nexus/Math.h"

namespace nexus {

Mat4 operator*(const Mat4& a, const Mat4& b)
{
    Mat4 r;
    for (int row = 0; row < 4; ++row)
        for (int col = 0; col < 4; ++col)
            r.cell[row * 4 + col] = (a.cell[row * 4 + 0] * b.cell[col + 0]) + (a.cell[row * 4 + 1] * b.cell[col + 4]) +
                                    (a.cell[row * 4 + 2] * b.cell[col + 8]) + (a.cell[row * 4 + 3] * b.cell[col + 12]);
    return r;
}

bool operator==(const Mat4& a, const Mat4& b)
{
    for (int i = 0; i < 16; ++i)
        if (a.cell[i] != b.cell[i]) return false;
    return true;
}

float3 TransformPosition(const float3& a, const Mat4& M)
{
    const float* c = M.cell;
    return make_float3(c[0] * a.x + c[1] * a.y + c[2] * a.z + c[3] * 1.0f, c[4] * a.x + c[5] * a.y + c[6] * a.z + c[7] * 1.0f,
                       c[8] * a.x + c[9] * a.y + c[10] * a.z + c[11] * 1.0f);
}

Mat4 Mat4::Transposed() const
{
    Mat4 M;
    M.cell[0] = cell[0]; M.cell[1] = cell[4]; M.cell[2] = cell[8];
    M.cell[4] = cell[1]; M.cell[5] = cell[5]; M.cell[6] = cell[9];
    M.cell[8] = cell[2]; M.cell[9] = cell[6]; M.cell[10] = cell[10];
    return M;
}

namespace {
// 3x3 minor of the 4x4 matrix m built from rows (r0,r1,r2) and columns (c0,c1,c2), expanded in the term
// order of the classic MESA gluInvertMatrix cofactor formula so results match the reference bit for bit.
struct Cof {
    const float* m;
    float operator()(int a, int b, int c) const { return m[a] * m[b] * m[c]; }
};
}  // namespace

Mat4 Mat4::Inverted() const
{
    const float* m = cell;
    const Cof t{m};
    float inv[16];
    inv[0] = t(5, 10, 15) - t(5, 11, 14) - t(9, 6, 15) + t(9, 7, 14) + t(13, 6, 11) - t(13, 7, 10);
    inv[1] = -t(1, 10, 15) + t(1, 11, 14) + t(9, 2, 15) - t(9, 3, 14) - t(13, 2, 11) + t(13, 3, 10);
    inv[2] = t(1, 6, 15) - t(1, 7, 14) - t(5, 2, 15) + t(5, 3, 14) + t(13, 2, 7) - t(13, 3, 6);
    inv[3] = -t(1, 6, 11) + t(1, 7, 10) + t(5, 2, 11) - t(5, 3, 10) - t(9, 2, 7) + t(9, 3, 6);
    inv[4] = -t(4, 10, 15) + t(4, 11, 14) + t(8, 6, 15) - t(8, 7, 14) - t(12, 6, 11) + t(12, 7, 10);
    inv[5] = t(0, 10, 15) - t(0, 11, 14) - t(8, 2, 15) + t(8, 3, 14) + t(12, 2, 11) - t(12, 3, 10);
    inv[6] = -t(0, 6, 15) + t(0, 7, 14) + t(4, 2, 15) - t(4, 3, 14) - t(12, 2, 7) + t(12, 3, 6);
    inv[7] = t(0, 6, 11) - t(0, 7, 10) - t(4, 2, 11) + t(4, 3, 10) + t(8, 2, 7) - t(8, 3, 6);
    inv[8] = t(4, 9, 15) - t(4, 11, 13) - t(8, 5, 15) + t(8, 7, 13) + t(12, 5, 11) - t(12, 7, 9);
    inv[9] = -t(0, 9, 15) + t(0, 11, 13) + t(8, 1, 15) - t(8, 3, 13) - t(12, 1, 11) + t(12, 3, 9);
    inv[10] = t(0, 5, 15) - t(0, 7, 13) - t(4, 1, 15) + t(4, 3, 13) + t(12, 1, 7) - t(12, 3, 5);
    inv[11] = -t(0, 5, 11) + t(0, 7, 9) + t(4, 1, 11) - t(4, 3, 9) - t(8, 1, 7) + t(8, 3, 5);
    inv[12] = -t(4, 9, 14) + t(4, 10, 13) + t(8, 5, 14) - t(8, 6, 13) - t(12, 5, 10) + t(12, 6, 9);
    inv[13] = t(0, 9, 14) - t(0, 10, 13) - t(8, 1, 14) + t(8, 2, 13) + t(12, 1, 10) - t(12, 2, 9);
    inv[14] = -t(0, 5, 14) + t(0, 6, 13) + t(4, 1, 14) - t(4, 2, 13) - t(12, 1, 6) + t(12, 2, 5);
    inv[15] = t(0, 5, 10) - t(0, 6, 9) - t(4, 1, 10) + t(4, 2, 9) + t(8, 1, 6) - t(8, 2, 5);
    const float det = m[0] * inv[0] + m[1] * inv[4] + m[2] * inv[8] + m[3] * inv[12];
    Mat4 r;
    if (det != 0) {
        const float invdet = 1.0f / det;
        for (int i = 0; i < 16; ++i) r.cell[i] = inv[i] * invdet;
    }
    return r;
}

}  // namespace nexus
