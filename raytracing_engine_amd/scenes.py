"""Synthetic triangle scenes for path B (BASELINE.json configs[2..4]).  The
reference has no triangle scenes: these are build-defined inputs, generated from a counter hash so
they are identical on every host and numpy version."""
import numpy as np


def _hash32(x):
    x = np.asarray(x, np.uint32).copy()
    x ^= x >> np.uint32(16)
    x *= np.uint32(0x7FEB352D)
    x ^= x >> np.uint32(15)
    x *= np.uint32(0x846CA68B)
    x ^= x >> np.uint32(16)
    return x


def _uniform(seed, stream, n):
    """n floats in [0,1) on a 2^-24 grid: hash32(i + hash32(seed*0x9e3779b9 + stream))."""
    with np.errstate(over="ignore"):
        base = _hash32(np.uint32((seed * 0x9E3779B9 + stream) & 0xFFFFFFFF))
        h = _hash32(np.arange(n, dtype=np.uint32) + base)
    return (h >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24)


def _quad(p0, p1, p2, p3):
    return [np.concatenate([p0, p1, p2]), np.concatenate([p0, p2, p3])]


def soup_scene(n_tris, seed=1, edge=0.25, light_emission=(30.0, 30.0, 30.0)):
    """n_tris random triangles: v0 ~ U([-10,10] x [5,25] x [-10,10]), edges ~ U(-edge,edge)^3,
    albedo ~ U(0.2,0.9); the last two triangles are replaced by one emissive quad overhead
    (z = 12, facing down).  Returns (verts[n,9], albedo[n,3], emission[n,3]) float32."""
    n = int(n_tris)
    assert n >= 3
    u = [_uniform(seed, k, n) for k in range(12)]
    v0 = np.stack([u[0] * 20 - 10, u[1] * 20 + 5, u[2] * 20 - 10], 1).astype(np.float32)
    e1 = (np.stack(u[3:6], 1) * 2 - 1).astype(np.float32) * np.float32(edge)
    e2 = (np.stack(u[6:9], 1) * 2 - 1).astype(np.float32) * np.float32(edge)
    verts = np.concatenate([v0, v0 + e1, v0 + e2], 1).astype(np.float32)
    albedo = (np.stack(u[9:12], 1) * np.float32(0.7) + np.float32(0.2)).astype(np.float32)
    emission = np.zeros((n, 3), np.float32)
    f = np.float32
    quad = _quad(np.array([-4, 11, 12], f), np.array([4, 11, 12], f), np.array([4, 19, 12], f), np.array([-4, 19, 12], f))
    verts[n - 2], verts[n - 1] = quad
    albedo[n - 2:] = 0.0
    emission[n - 2:] = np.asarray(light_emission, f)
    return verts, albedo, emission


def cornell_tri_scene():
    """A small closed room (x,z in [-6,6], y in [0,22]) with a ceiling light and two boxes: 7 quads
    + 2 boxes = 38 triangles.  Camera at (0,1,0) looking +Y sees the whole room."""
    f = np.float32
    tris, alb, emi = [], [], []

    def add(q, a, e=(0, 0, 0)):
        for t in _quad(*[np.array(p, f) for p in q]):
            tris.append(t)
            alb.append(a)
            emi.append(e)

    X, Z, Y0, Y1 = 6.0, 6.0, 0.0, 22.0
    white, red, green = (0.75, 0.75, 0.75), (0.75, 0.15, 0.15), (0.15, 0.75, 0.15)
    add([(-X, Y0, -Z), (X, Y0, -Z), (X, Y1, -Z), (-X, Y1, -Z)], white)  # floor
    add([(-X, Y0, Z), (-X, Y1, Z), (X, Y1, Z), (X, Y0, Z)], white)      # ceiling
    add([(-X, Y1, -Z), (X, Y1, -Z), (X, Y1, Z), (-X, Y1, Z)], white)    # back wall
    add([(-X, Y0, -Z), (-X, Y1, -Z), (-X, Y1, Z), (-X, Y0, Z)], red)    # left
    add([(X, Y0, -Z), (X, Y0, Z), (X, Y1, Z), (X, Y1, -Z)], green)      # right
    add([(-X, Y0, -Z), (-X, Y0, Z), (X, Y0, Z), (X, Y0, -Z)], white)    # wall behind the camera
    add([(-2, 9, Z - 0.01), (-2, 13, Z - 0.01), (2, 13, Z - 0.01), (2, 9, Z - 0.01)], (0, 0, 0), (15, 15, 15))  # light

    def box(cx, cy, cz, sx, sy, sz, a):
        x0, x1, y0, y1, z0, z1 = cx - sx, cx + sx, cy - sy, cy + sy, cz - sz, cz + sz
        add([(x0, y0, z0), (x1, y0, z0), (x1, y0, z1), (x0, y0, z1)], a)
        add([(x0, y1, z0), (x0, y1, z1), (x1, y1, z1), (x1, y1, z0)], a)
        add([(x0, y0, z0), (x0, y0, z1), (x0, y1, z1), (x0, y1, z0)], a)
        add([(x1, y0, z0), (x1, y1, z0), (x1, y1, z1), (x1, y0, z1)], a)
        add([(x0, y0, z0), (x0, y1, z0), (x1, y1, z0), (x1, y0, z0)], a)
        add([(x0, y0, z1), (x1, y0, z1), (x1, y1, z1), (x0, y1, z1)], a)

    box(-2.5, 14, -3.0, 1.8, 1.8, 3.0, (0.7, 0.7, 0.3))
    box(2.5, 10, -4.5, 1.5, 1.5, 1.5, (0.3, 0.5, 0.8))
    return np.array(tris, f), np.array(alb, f), np.array(emi, f)
