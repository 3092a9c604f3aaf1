"""Deterministic SPD-style scene generators emitting NFF text.

The Standard Procedural Databases programs (balls, tetra, rings, teapot, mount) are
not available offline, so these generators produce scenes of the same *shape of work*
(primitive kinds and counts, light counts, surface parameters, camera) using only the
NFF commands the reference parser accepts (``v b l f s c p pp``,
/root/reference/src/nff.rs:381-410).  They are not byte-for-byte the SPD files; every
parity check renders the same generated text with both the GPU path and the oracle.

All generators are seed-free except ``mount`` which uses a fixed LCG.
"""
from __future__ import annotations

import math
from typing import List


def _fmt(x: float) -> str:
    return repr(round(float(x), 6)) if abs(x) >= 1e-9 else "0"


def _view(frm, at, up, angle, hither, res) -> List[str]:
    return [
        "v",
        "from " + " ".join(_fmt(c) for c in frm),
        "at " + " ".join(_fmt(c) for c in at),
        "up " + " ".join(_fmt(c) for c in up),
        f"angle {_fmt(angle)}",
        f"hither {_fmt(hither)}",
        f"resolution {int(res[0])} {int(res[1])}",
    ]


def _norm(v):
    m = math.sqrt(sum(c * c for c in v))
    return tuple(c / m for c in v)


def _cross(a, b):
    return (a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0])


def _frame(axis):
    """Two unit vectors orthogonal to ``axis`` (unit)."""
    ref = (1.0, 0.0, 0.0) if abs(axis[0]) < 0.9 else (0.0, 1.0, 0.0)
    u = _norm(_cross(axis, ref))
    v = _cross(axis, u)
    return u, v


def balls(size: int = 4, width: int = 512, height: int = 512) -> str:
    """Sphereflake: sum_{k<=size} 9^k shiny spheres + one floor polygon, 3 lights.

    size 4 -> 7,381 spheres + 1 polygon (the SPD ``balls`` headline scene).  Each
    sphere of radius r carries 9 children of radius r/3 touching its surface: a ring
    of 6 at 80 degrees from its axis and a ring of 3 at 30 degrees.
    """
    out = _view((2.1, 1.3, 1.7), (0, 0, 0), (0, 0, 1), 45, 1.0, (width, height))
    out.append("b 0.078 0.361 0.753")
    out += ["l 4 3 2", "l 1 -4 4", "l -3 1 5"]
    out.append("f 1 0.75 0.33 0.8 0 0 0 0")
    out += ["p 4", "12 12 -0.5", "-12 12 -0.5", "-12 -12 -0.5", "12 -12 -0.5"]
    out.append("f 1 0.9 0.7 0.5 0.5 3 0 0")

    spheres = []

    def rec(center, radius, axis, depth):
        spheres.append((center, radius))
        if depth == 0:
            return
        u, v = _frame(axis)
        cr = radius / 3.0
        dist = radius + cr
        rings = ((6, math.radians(80.0), 0.0), (3, math.radians(30.0), math.radians(30.0)))
        for count, polar, phase in rings:
            for k in range(count):
                az = phase + 2.0 * math.pi * k / count
                d = tuple(
                    math.cos(polar) * axis[i] + math.sin(polar) * (math.cos(az) * u[i] + math.sin(az) * v[i])
                    for i in range(3)
                )
                d = _norm(d)
                c = tuple(center[i] + dist * d[i] for i in range(3))
                rec(c, cr, d, depth - 1)

    rec((0.0, 0.0, 0.0), 0.5, (0.0, 0.0, 1.0), size)
    for c, r in spheres:
        out.append("s " + " ".join(_fmt(x) for x in c) + " " + _fmt(r))
    return "\n".join(out) + "\n"


def tetra(size: int = 6, width: int = 512, height: int = 512) -> str:
    """Sierpinski tetrahedron: 4^size triangles (``p 3``) facing outwards, two lights."""
    out = _view((1.022846, -3.177154, -2.174512), (-0.004103, -0.004103, 0.216539), (-0.816497, -0.816497, 0.816497),
                45, 1.0, (width, height))
    out.append("b 0.078 0.361 0.753")
    out.append("l 1.876066 -18.123936 -5.000422")
    out.append("l 6 -4 -9 0.6 0.6 0.5")
    out.append("f 1 0.2 0.2 1 0 0 0 0")
    tris = []

    def rec(cx, cy, cz, s, depth):
        if depth == 0:
            p = [
                (cx + s, cy + s, cz + s),
                (cx + s, cy - s, cz - s),
                (cx - s, cy + s, cz - s),
                (cx - s, cy - s, cz + s),
            ]
            # (counter-clockwise as seen from outside: the reference takes (v1 - v0) x (v2 - v0) for the front,
            # src/polygon.rs:61-63, and shades no back face)
            for a, b, c in ((0, 3, 1), (0, 1, 2), (0, 2, 3), (1, 3, 2)):
                tris.append((p[a], p[b], p[c]))
            return
        h = s / 2.0
        for dx, dy, dz in ((1, 1, 1), (1, -1, -1), (-1, 1, -1), (-1, -1, 1)):
            rec(cx + dx * h, cy + dy * h, cz + dz * h, h, depth - 1)

    rec(0.0, 0.0, 0.0, 1.0, size - 1 if size > 0 else 0)
    for t in tris:
        out.append("p 3")
        for p in t:
            out.append(" ".join(_fmt(c) for c in p))
    return "\n".join(out) + "\n"


def rings(size: int = 3, width: int = 512, height: int = 512) -> str:
    """Pyramid of rings built from short cylinders (``c``, equal radii) with joint
    spheres, a few true cones, refractive and reflective surfaces, 3 lights."""
    out = _view((-6.0, -7.0, 5.5), (0, 0, 0.8), (0, 0, 1), 45, 0.5, (width, height))
    out.append("b 0.078 0.361 0.753")
    out += ["l 3 -10 9", "l -9 -3 7 0.9 0.9 1.0", "l 4 5 12"]
    out.append("f 1 0.75 0.33 0.8 0.1 5 0 0")
    out += ["p 4", "14 14 -0.6", "-14 14 -0.6", "-14 -14 -0.6", "14 -14 -0.6"]
    fills = [
        "f 0.9 0.2 0.2 0.6 0.4 20 0 0",
        "f 0.2 0.9 0.3 0.5 0.5 30 0 0",
        "f 0.9 0.9 1.0 0.1 0.3 60 0.8 1.5",
        "f 0.3 0.4 0.9 0.7 0.3 10 0 0",
    ]
    seg = 10
    ring_index = 0
    for level in range(size):
        count = size - level
        for a in range(count):
            for b in range(count):
                cx = (a - (count - 1) / 2.0) * 2.2
                cy = (b - (count - 1) / 2.0) * 2.2
                cz = level * 1.3
                out.append(fills[ring_index % len(fills)])
                ring_index += 1
                R, r = 1.0, 0.16
                tilt = 0.35 * ((a + b + level) % 3 - 1)
                pts = []
                for k in range(seg):
                    ang = 2.0 * math.pi * k / seg
                    x, y, z = R * math.cos(ang), R * math.sin(ang), 0.0
                    y, z = y * math.cos(tilt) - z * math.sin(tilt), y * math.sin(tilt) + z * math.cos(tilt)
                    pts.append((cx + x, cy + y, cz + z))
                for k in range(seg):
                    p, q = pts[k], pts[(k + 1) % seg]
                    out.append("c")
                    out.append(" ".join(_fmt(c) for c in p) + " " + _fmt(r))
                    out.append(" ".join(_fmt(c) for c in q) + " " + _fmt(r))
                    out.append("s " + " ".join(_fmt(c) for c in p) + " " + _fmt(r))
    # true cones and a frustum (one-line form) on top, and a glass ball
    top = size * 1.3
    out.append("f 0.9 0.8 0.2 0.6 0.4 15 0 0")
    out.append(f"c 0 0 {_fmt(top - 0.4)} 0.7 0 0 {_fmt(top + 1.0)} 0")
    out.append(f"c 2.5 -2.5 -0.6 0.5 2.5 -2.5 0.6 0.2")
    out.append("f 1 1 1 0.05 0.2 80 0.9 1.5")
    out.append(f"s -2.6 -2.8 0.3 0.9")
    return "\n".join(out) + "\n"


def lathe(nu: int = 48, nv: int = 24, width: int = 512, height: int = 512) -> str:
    """Teapot-class scene: a lathed pot + torus handle as ``pp`` patches with smooth
    per-vertex normals (quads and triangles), on a two-colour checker floor of ``p 4``
    quads, 2 lights, Phong highlights.  nu=96, nv=48 gives ~9.2k patches."""
    out = _view((4.8, 3.2, 2.6), (0, 0, 0.6), (0, 0, 1), 45, 1.0, (width, height))
    out.append("b 0.1 0.1 0.1")
    out += ["l 6 2 8", "l -3 7 6 0.8 0.8 0.7"]
    # checker floor
    n = 8
    for i in range(-n, n):
        for j in range(-n, n):
            out.append("f 1 1 1 0.7 0.2 8 0 0" if (i + j) & 1 else "f 0.5 0.1 0.1 0.7 0.2 8 0 0")
            out.append("p 4")
            for x, y in ((i, j), (i + 1, j), (i + 1, j + 1), (i, j + 1)):
                out.append(f"{_fmt(x * 0.75)} {_fmt(y * 0.75)} 0")

    def profile(t):  # radius, height along the pot body, t in [0,1]
        r = 0.25 + 1.0 * math.sin(math.pi * (0.08 + 0.84 * t)) ** 1.0 * (0.75 + 0.25 * math.cos(2.2 * t))
        return r, 0.02 + 1.3 * t

    def body(u, v):
        ang = 2.0 * math.pi * u
        r, z = profile(v)
        return (r * math.cos(ang), r * math.sin(ang), z)

    def torus(u, v):
        a, b = 2.0 * math.pi * u, 2.0 * math.pi * v
        R, r = 0.45, 0.09
        x = 1.25 + (R + r * math.cos(b)) * math.cos(a)
        z = 0.75 + (R + r * math.cos(b)) * math.sin(a)
        return (x, r * math.sin(b), z)

    def emit(fn, nu_, nv_, closed_v):
        eps = 1e-4

        def nrm(u, v):
            p0 = fn(u, v)
            pu = fn(u + eps, v)
            pv = fn(u, v + eps)
            du = tuple(pu[i] - p0[i] for i in range(3))
            dv = tuple(pv[i] - p0[i] for i in range(3))
            c = _cross(du, dv)
            m = math.sqrt(sum(x * x for x in c))
            return tuple(x / m for x in c) if m > 0 else (0.0, 0.0, 1.0)

        for i in range(nu_):
            for j in range(nv_):
                u0, u1 = i / nu_, (i + 1) / nu_
                v0, v1 = j / nv_, (j + 1) / nv_
                corners = [(u0, v0), (u1, v0), (u1, v1), (u0, v1)]
                if (i + j) % 5 == 0:  # mix in triangles
                    polys = [corners[:3], [corners[0], corners[2], corners[3]]]
                else:
                    polys = [corners]
                for poly in polys:
                    out.append(f"pp {len(poly)}")
                    for (u, v) in poly:
                        p = fn(u, v)
                        nn = nrm(u, v)
                        out.append(" ".join(_fmt(c) for c in p) + " " + " ".join(_fmt(c) for c in nn))

    out.append("f 0.9 0.85 0.7 0.6 0.4 25 0 0")
    emit(body, nu, nv, False)
    out.append("f 0.8 0.3 0.2 0.6 0.4 25 0 0")
    emit(torus, max(8, nu // 3), max(6, nv // 3), True)
    return "\n".join(out) + "\n"


def mount(size: int = 5, lights: int = 4, width: int = 512, height: int = 512) -> str:
    """Fractal height field: 2*4^size triangles (``p 3``) + 4 refractive spheres,
    ``lights`` point lights.  Midpoint displacement with a fixed LCG."""
    n = 1 << size
    state = [12345]

    def rnd():
        state[0] = (1103515245 * state[0] + 12345) & 0x7FFFFFFF
        return state[0] / float(0x7FFFFFFF) - 0.5

    hgt = [[0.0] * (n + 1) for _ in range(n + 1)]
    step, amp = n, 0.9
    while step > 1:
        half = step // 2
        for i in range(0, n, step):
            for j in range(0, n, step):
                c = (hgt[i][j] + hgt[i + step][j] + hgt[i][j + step] + hgt[i + step][j + step]) / 4.0
                hgt[i + half][j + half] = c + amp * rnd()
        for i in range(0, n + 1, half):
            for j in range((i + half) % step, n + 1, step):
                acc, cnt = 0.0, 0
                for di, dj in ((-half, 0), (half, 0), (0, -half), (0, half)):
                    a, b = i + di, j + dj
                    if 0 <= a <= n and 0 <= b <= n:
                        acc += hgt[a][b]
                        cnt += 1
                hgt[i][j] = acc / cnt + amp * rnd() * 0.7
        step = half
        amp *= 0.55

    out = _view((-1.6, 1.6, 1.7), (0, 0, 0), (0, 0, 1), 45, 0.5, (width, height))
    out.append("b 0.078 0.361 0.753")
    for k in range(lights):
        ang = 2.0 * math.pi * k / max(1, lights)
        out.append(f"l {_fmt(6.0 * math.cos(ang))} {_fmt(6.0 * math.sin(ang))} {_fmt(5.0 + (k % 3))}")
    out.append("f 0.5 0.45 0.35 1 0 0 0 0")

    def P(i, j):
        return (-1.0 + 2.0 * i / n, -1.0 + 2.0 * j / n, hgt[i][j] * 0.6)

    for i in range(n):
        for j in range(n):
            a, b, c, d = P(i, j), P(i + 1, j), P(i + 1, j + 1), P(i, j + 1)
            for tri in ((a, b, c), (a, c, d)):
                out.append("p 3")
                for p in tri:
                    out.append(" ".join(_fmt(x) for x in p))
    out.append("f 1 1 1 0.05 0.3 40 0.9 1.5")
    for (x, y, z, r) in ((-0.8, 0.8, 0.8, 0.15), (-0.5, 0.3, 0.9, 0.2), (0.1, 0.5, 0.85, 0.15), (-0.3, 0.9, 0.7, 0.1)):
        out.append(f"s {x} {y} {z} {r}")
    return "\n".join(out) + "\n"


GENERATORS = {"balls": balls, "tetra": tetra, "rings": rings, "lathe": lathe, "mount": mount}
