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


def terrain_scene(grid=708, seed=1, light_emission=(6.0, 6.0, 6.0)):
    """A closed-surface scene for context (the soup is a participating-medium-like worst case): a
    height field of grid x grid cells = 2*grid^2 triangles over x in [-30,30], y in [2,62] (the camera at
    the origin looks along +Y and slightly down onto it), heights from three octaves of hashed value
    noise, plus the same emissive quad overhead.  grid=708 gives 1 002 528 + 2 triangles."""
    f = np.float32
    n = int(grid)

    def lattice(cells, stream):
        return _uniform(seed, stream, (cells + 1) * (cells + 1)).reshape(cells + 1, cells + 1)

    def value_noise(cells, stream):
        lat = lattice(cells, stream)
        t = np.linspace(0, cells, n + 1, dtype=np.float64)
        i = np.minimum(t.astype(np.int64), cells - 1)
        fr = (t - i)
        fr = fr * fr * (3 - 2 * fr)
        a = lat[i][:, i] * (1 - fr)[None, :] + lat[i][:, i + 1] * fr[None, :]
        b = lat[i + 1][:, i] * (1 - fr)[None, :] + lat[i + 1][:, i + 1] * fr[None, :]
        return a * (1 - fr)[:, None] + b * fr[:, None]

    hgt = (6.0 * value_noise(6, 20) + 2.0 * value_noise(24, 21) + 0.5 * value_noise(96, 22)).astype(f) - f(9.0)
    xs = np.linspace(-30, 30, n + 1, dtype=f)
    ys = np.linspace(2, 62, n + 1, dtype=f)
    X, Y = np.meshgrid(xs, ys)
    P = np.stack([X, Y, hgt], -1)  # (n+1, n+1, 3)
    p00, p10, p01, p11 = P[:-1, :-1], P[:-1, 1:], P[1:, :-1], P[1:, 1:]
    t1 = np.concatenate([p00, p10, p11], -1).reshape(-1, 9)
    t2 = np.concatenate([p00, p11, p01], -1).reshape(-1, 9)
    verts = np.concatenate([t1, t2]).astype(f)
    m = len(verts)
    u = [_uniform(seed, 30 + k, m) for k in range(3)]
    albedo = (np.stack(u, 1) * f(0.4) + f(0.4)).astype(f)
    emission = np.zeros((m, 3), f)
    quad = _quad(np.array([-6, 20, 14], f), np.array([6, 20, 14], f), np.array([6, 32, 14], f), np.array([-6, 32, 14], f))
    verts = np.concatenate([verts, np.array(quad, f)])
    albedo = np.concatenate([albedo, np.zeros((2, 3), f)])
    emission = np.concatenate([emission, np.tile(np.asarray(light_emission, f), (2, 1))])
    return verts, albedo, emission
