// nexus/Triangle.h — host triangle of the kept API surface.
// Mirrors /root/reference/Nexus/src/Geometry/Triangle.h:7-43 (fields, centroid, ToDevice -> 96-byte device POD).
#pragma once

#include "Math.h"

namespace nexus {

struct Triangle {
    float3 pos0, pos1, pos2;
    float3 centroid;
    float3 normal0, normal1, normal2;
    float2 texCoord0, texCoord1, texCoord2;

    Triangle() = default;
    Triangle(float3 p0, float3 p1, float3 p2, float3 n0 = {}, float3 n1 = {}, float3 n2 = {}, float2 t0 = {}, float2 t1 = {},
             float2 t2 = {})
        : pos0(p0), pos1(p1), pos2(p2), centroid((p0 + p1 + p2) / 3.0f), normal0(n0), normal1(n1), normal2(n2), texCoord0(t0),
          texCoord1(t1), texCoord2(t2)
    {
    }
    explicit Triangle(const nx_triangle& t)
        : Triangle(make_float3(t.pos0), make_float3(t.pos1), make_float3(t.pos2), make_float3(t.normal0), make_float3(t.normal1),
                   make_float3(t.normal2), make_float2(t.texCoord0[0], t.texCoord0[1]), make_float2(t.texCoord1[0], t.texCoord1[1]),
                   make_float2(t.texCoord2[0], t.texCoord2[1]))
    {
    }

    static nx_triangle ToDevice(const Triangle& t)
    {
        nx_triangle d;
        store(d.pos0, t.pos0); store(d.pos1, t.pos1); store(d.pos2, t.pos2);
        store(d.normal0, t.normal0); store(d.normal1, t.normal1); store(d.normal2, t.normal2);
        d.texCoord0[0] = t.texCoord0.x; d.texCoord0[1] = t.texCoord0.y;
        d.texCoord1[0] = t.texCoord1.x; d.texCoord1[1] = t.texCoord1.y;
        d.texCoord2[0] = t.texCoord2.x; d.texCoord2[1] = t.texCoord2.y;
        return d;
    }
};

}  // namespace nexus
