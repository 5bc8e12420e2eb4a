"""Deterministic procedural geometry for the bench and the parity tests.

The reference ships no 1M-triangle asset (its demo scenes are missing large blobs, /root/reference/.MISSING_LARGE_BLOBS),
so BASELINE.json's configs are regenerated from a seed on every box instead of being committed.
"""
import numpy as np

from . import pod


def _grid_surface_triangles(P, N, uv, wrap_u, wrap_v):
    """P, N: (U, V, 3) vertex grid; returns TRI_DT triangles of the (optionally periodic) quad grid."""
    U, V, _ = P.shape
    iu = np.arange(U if wrap_u else U - 1)
    iv = np.arange(V if wrap_v else V - 1)
    a, b = np.meshgrid(iu, iv, indexing="ij")
    a1 = (a + 1) % U
    b1 = (b + 1) % V
    idx00 = (a, b)
    idx10 = (a1, b)
    idx01 = (a, b1)
    idx11 = (a1, b1)

    def tri(i0, i1, i2):
        t = np.zeros(a.size, dtype=pod.TRI_DT)
        for k, ii in enumerate((i0, i1, i2)):
            t["pos%d" % k] = P[ii].reshape(-1, 3)
            t["normal%d" % k] = N[ii].reshape(-1, 3)
            t["texCoord%d" % k] = uv[ii].reshape(-1, 2)
        return t

    # winding chosen so that the geometric normal cross(p1-p0, p2-p0) agrees with the vertex normals cross(dv, du)
    t0 = tri(idx00, idx11, idx10)
    t1 = tri(idx00, idx01, idx11)
    out = np.empty(2 * a.size, dtype=pod.TRI_DT)
    out[0::2] = t0
    out[1::2] = t1
    return out


def displaced_torus(nu=1024, nv=512, seed=1, major=1.0, minor=0.45, amp=0.06, center=(0.0, 0.0, 0.0)):
    """Closed mesh with exactly 2*nu*nv triangles (1024 x 512 -> 1 048 576): a torus whose tube radius is displaced
    by a seeded sum of sines; smooth per-vertex normals from central differences of the vertex grid."""
    rng = np.random.RandomState(seed)
    u = (np.arange(nu, dtype=np.float64) / nu) * 2 * np.pi
    v = (np.arange(nv, dtype=np.float64) / nv) * 2 * np.pi
    uu, vv = np.meshgrid(u, v, indexing="ij")
    disp = np.zeros_like(uu)
    for _ in range(6):
        fu, fv = rng.randint(1, 9), rng.randint(1, 9)
        ph = rng.uniform(0, 2 * np.pi)
        disp += np.sin(fu * uu + fv * vv + ph) / (fu + fv)
    r = minor * (1.0 + amp * disp / max(1e-9, np.abs(disp).max()) * 3.0)
    x = (major + r * np.cos(vv)) * np.cos(uu)
    z = (major + r * np.cos(vv)) * np.sin(uu)
    y = r * np.sin(vv)
    P = np.stack([x, y, z], axis=-1)
    du = np.roll(P, -1, axis=0) - np.roll(P, 1, axis=0)
    dv = np.roll(P, -1, axis=1) - np.roll(P, 1, axis=1)
    N = np.cross(dv, du)
    N /= np.maximum(np.linalg.norm(N, axis=-1, keepdims=True), 1e-30)
    P = (P + np.asarray(center, np.float64)).astype(np.float32)
    N = N.astype(np.float32)
    uv = np.stack([uu / (2 * np.pi), vv / (2 * np.pi)], axis=-1).astype(np.float32)
    return _grid_surface_triangles(P, N, uv, True, True)


def quad(p0, p1, p2, p3):
    """Two triangles (p0,p1,p2), (p0,p2,p3) with the face normal and unit-square texture coordinates."""
    p = np.asarray([[p0, p1, p2], [p0, p2, p3]], np.float32)
    uv = np.asarray([[(0, 0), (1, 0), (1, 1)], [(0, 0), (1, 1), (0, 1)]], np.float32)
    return pod.make_triangles(p, uvs=uv)


def random_soup(n, seed=0, extent=1.0, size=0.05):
    """n small random triangles in a cube (stress case for the BVH: no spatial coherence)."""
    rng = np.random.RandomState(seed)
    c = rng.uniform(-extent, extent, size=(n, 1, 3))
    p = c + rng.uniform(-size, size, size=(n, 3, 3))
    return pod.make_triangles(p.astype(np.float32))


def height_field(m, seed=0, amp=0.15):
    """2*m*m triangles over [-1,1]^2 with a seeded sine height."""
    rng = np.random.RandomState(seed)
    g = np.linspace(-1, 1, m + 1)
    xx, zz = np.meshgrid(g, g, indexing="ij")
    yy = np.zeros_like(xx)
    for _ in range(5):
        fx, fz = rng.uniform(1, 6, 2)
        yy += np.sin(fx * xx + rng.uniform(0, 6)) * np.cos(fz * zz + rng.uniform(0, 6))
    yy *= amp / 5
    P = np.stack([xx, yy, zz], -1)
    du = np.gradient(P, axis=0)
    dv = np.gradient(P, axis=1)
    N = np.cross(dv, du)
    N /= np.maximum(np.linalg.norm(N, axis=-1, keepdims=True), 1e-30)
    uv = np.stack([(xx + 1) / 2, (zz + 1) / 2], -1).astype(np.float32)
    return _grid_surface_triangles(P.astype(np.float32), N.astype(np.float32), uv, False, False)


def random_rays(n, seed=0, radius=3.0, target_extent=1.0):
    """Rays from a sphere of `radius` towards random points in the target cube (unit directions)."""
    rng = np.random.RandomState(seed)
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    o = d * radius
    t = rng.uniform(-target_extent, target_extent, size=(n, 3))
    dirs = t - o
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    rays = np.zeros(n, dtype=pod.RAY_DT)
    rays["origin"] = o.astype(np.float32)
    rays["direction"] = dirs.astype(np.float32)
    return rays


def interior_rays(n, seed=0, extent=1.0):
    """Incoherent rays: random origins inside the cube, uniformly random unit directions."""
    rng = np.random.RandomState(seed)
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.zeros(n, dtype=pod.RAY_DT)
    rays["origin"] = rng.uniform(-extent, extent, size=(n, 3)).astype(np.float32)
    rays["direction"] = d.astype(np.float32)
    return rays
