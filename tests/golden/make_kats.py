#!/usr/bin/env python3
"""Known-answer pixels and known-answer space partitions for the reference's eight fixtures
(nff/test-*.nff), the added fixtures with interior nodes (nff/extra-*.nff) and a few small scenes
written here, derived from the reference's formulas and from nothing under oracle/ or the
product: this file imports numpy and json only.

    python tests/golden/make_kats.py            # rewrites tests/golden/kats.json

How the answers are made.  `Tracer` below is a scalar evaluation of the reference's render path,
written from the reference's source, formula by formula, with the line each one comes from:
camera src/render.rs:81-108, Whitted recursion src/scene.rs:82-200, bounding boxes
src/sphere.rs:12-25, src/cone.rs:15-95, src/polygon.rs:11-56, box algebra and the slab test
src/shape.rs:37-172, construction of the space partition src/space_partition.rs:40-258 (candidate
order, strict `<`, the two different pairs of inequalities of appraise_split and split, the axis
cycle), its recursive walk :268-346 (the far child's result REPLACES the near one's) and the
linear search of a leaf :348-385, sphere src/sphere.rs:49-76, cone src/cone.rs:115-214, polygon
src/polygon.rs:61-289, Phong src/phong.rs:35-62, Blinn-Phong src/blinn_phong.rs:35-56, reflect /
refract (with total internal reflection) / interpolate src/vector_math.rs:51-73 and :440-448, the
scene reader's one-line `c` and six-argument `l` src/nff.rs:206-224, :320-333, :381-410,
quantisation src/ppm.rs:20-22.  For every scene whose partition has interior nodes the float32
partition itself (boxes, axes, planes, leaf lists, as bit patterns) is stored as well, and the
tests hold the product's builder and the oracle's to it.
Every pixel is evaluated twice, in numpy float32 (the reference's arithmetic type, one rounding
per operation as in its scalar code) and in float64; a pixel becomes a known answer only when
both give the same byte within 1 -- so no answer hangs on a rounding accident -- and the float32
byte is what is stored.  On top of that, the expectations the fixtures' own header comments
state, and values that follow in closed form from the scene files, are asserted here (see
`closed_form_checks`): they hold for the stored answers independently of the tracer's details.

Tests assert the stored bytes, to within 1, on the CPU oracle and on the HIP path.
"""
import json
import math
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
NFF_DIR = os.path.join(HERE, "nff")

# Two scenes the reference's fixtures do not cover (they have no transmitting surface and no
# `pp` patch).  window: a head-on square, lit head-on by a light between it and a small clear pane
# (transmittance 0.8, refraction index 1: the refracted ray goes straight on, src/vector_math.rs:
# 62-73 with index 1) through which the middle of the view sees it; the light sits behind the
# pane because a transmitting blocker still blocks a shadow ray completely (src/scene.rs:108-117).
# patch: one `pp` triangle whose vertex normals differ.
EXTRA_SCENES = {
    "kat-window": """b 0 0 0
v
from 0 0 0
at 1 0 0
up 0 1 0
angle 45
hither 0.1
resolution 512 512
l 3 0 0
f 1 1 1 0 0 1 0.8 1
p 4
2 0.5 0.5
2 0.5 -0.5
2 -0.5 -0.5
2 -0.5 0.5
f 0.2 0.6 1 0.75 0 1 0 0
p 4
4 2 2
4 2 -2
4 -2 -2
4 -2 2
""",
    "kat-patch": """b 0 0 0
v
from 0 0 0
at 1 0 0
up 0 1 0
angle 45
hither 0.1
resolution 512 512
l -1 2 1
f 1 0.5 0.25 0.8 0.2 6 0 0
pp 3
3 -1 1.2 -1 -0.3 0.4
3 1.3 0 -1 0.5 0
3 -1 -1.2 -1 -0.3 -0.4
""",
    # Total internal reflection (src/vector_math.rs:62-73, the `else` branch).  The middle of the view
    # meets a glass face head-on (index 1.5: refracted(n, 1.5) of a head-on ray is the ray itself), then
    # the back of a second glass face turned 45 degrees: leaving glass, refracted(-n, 1/1.5) has
    # a = 1 - (1 - 0.5) * 2.25 < 0, so the ray is mirrored instead, straight down onto a matt floor lit
    # from the side.  Both faces shade to nothing themselves (kd = ks = 0).
    "kat-tir": """b 0 0 0
v
from 0 0 0
at 1 0 0
up 0 1 0
angle 45
hither 0.1
resolution 512 512
l 3 -0.5 2.5
f 1 1 1 0 0 1 0.9 1.5
p 4
2 0.5 0.5
2 0.5 -0.5
2 -0.5 -0.5
2 -0.5 0.5
p 4
2.5 0.5 -0.6
2.5 0.5 0.6
3.5 -0.5 0.6
3.5 -0.5 -0.6
f 0.2 0.8 0.4 1 0 1 0 0
p 4
0 -2 -3
0 -2 3
6 -2 3
6 -2 -3
""",
    # A cylinder (`c` with equal radii: dr = 0 in src/cone.rs:155-171) given on one line
    # (src/nff.rs:206-224) under a coloured light (six-argument `l`, :320-333) at the eye.
    "kat-cylinder": """b 0 0 0
v
from 0 0 0
at 1 0 0
up 0 1 0
angle 45
hither 0.1
resolution 512 512
l 0 0 0 1 0.5 0.25
f 1 1 1 0.6 0.4 4 0 0
c 3 -1 0 0.5 3 1 0 0.5
""",
    # Fifteen primitives of every kind (quads, a pentagon, a patch, spheres, a glass sphere, cylinders
    # on one and two lines, a cone proper) in a strip along x, so that the construction splits it.
    "kat-mixed": """b 0.05 0.1 0.2
v
from 2 -10 5
at 0 0 0.5
up 0 0 1
angle 58
hither 0.5
resolution 256 256
l 8 -6 10
l -6 -8 7 0.9 0.7 0.5
f 0.8 0.8 0.8 0.8 0.2 8 0 0
p 4
-6 -1.5 0
-4 -1.5 0
-4 1.5 0
-6 1.5 0
p 4
-4 -1.5 0
-2 -1.5 0
-2 1.5 0
-4 1.5 0
p 4
-2 -1.5 0
0 -1.5 0
0 1.5 0
-2 1.5 0
p 4
0 -1.5 0
2 -1.5 0
2 1.5 0
0 1.5 0
p 4
2 -1.5 0
4 -1.5 0
4 1.5 0
2 1.5 0
p 4
4 -1.5 0
6 -1.5 0
6 1.5 0
4 1.5 0
f 0.9 0.3 0.2 0.4 0.6 30 0 0
s -5 0 0.8 0.8
s 5 0.3 0.8 0.8
f 0.2 0.4 0.9 0.6 0.4 12 0 0
s 2.9 -0.6 0.7 0.7
f 1 1 1 0.05 0.1 40 0.85 1.5
s 0 0 1 1
f 0.3 0.8 0.3 0.7 0.3 6 0 0
c -3 0.4 0 0.45 -3 0.4 1.6 0.45
c
-1.9 -1.0 0.4 0.3
-1.1 1.0 0.4 0.3
f 0.9 0.8 0.2 0.6 0.4 20 0 0
c 1.5 0.8 0 0.6 1.5 0.8 1.5 0
f 0.8 0.3 0.8 0.9 0.1 3 0 0
pp 3
3.6 1.2 0.1 -0.2 -0.5 1
4.4 1.4 1.4 0.3 -0.6 0.7
3.2 1.4 1.2 -0.4 -0.6 0.6
p 5
-4.2 -1.2 0.2
-3.6 -1.3 1.0
-4.4 -1.0 1.6
-5.2 -0.7 1.1
-5.0 -0.8 0.4
""",
}


class Tracer:
    """The reference's render path in arithmetic type T.  blinn: the --blinn-phong flag
    (src/main.rs:24-34): every `f` line then makes a Blinn-Phong surface (src/nff.rs:286-306); the
    surface in force before the first `f` is Phong either way (:352-360)."""

    def __init__(self, text, T, blinn=False):
        self.T = T
        self.blinn = blinn
        self.lights, self.prims = [], []
        self.background = self.vec(1, 1, 1)  # src/scene.rs:26
        surface = dict(color=self.vec(1, 1, 1), kd=T(1), ks=T(0), shine=T(1), refl=T(0), trans=T(0), ior=T(1), blinn=False)  # src/nff.rs:352-360
        lines = text.splitlines()
        k = 0
        while k < len(lines):
            line = lines[k]
            k += 1
            if line.startswith("#") or not line.strip():
                continue
            tok = line.split()
            cmd, args = tok[0], [T(x) for x in tok[1:]] if tok[0] != "v" else []
            if cmd == "v":
                self.view = {}
                while len(self.view) < 6:
                    t = lines[k].split()
                    k += 1
                    self.view[t[0]] = [T(x) for x in t[1:]]
            elif cmd == "b":
                self.background = self.vec(*args)
            elif cmd == "l":
                self.lights.append((self.vec(*args[:3]), self.vec(*args[3:6]) if len(args) == 6 else self.vec(1, 1, 1)))
            elif cmd == "f":  # src/nff.rs:270-300: reflectance := ks
                surface = dict(color=self.vec(*args[:3]), kd=args[3], ks=args[4], shine=args[5], refl=args[4], trans=args[6], ior=args[7], blinn=blinn)
            elif cmd == "s":
                self.prims.append(("sphere", self.vec(*args[:3]), args[3], surface))
            elif cmd == "c" and len(args) == 8:  # src/nff.rs:206-224, :405-407: base and apex on the command's own line
                self.prims.append(("cone", self.vec(*args[:3]), args[3], self.vec(*args[4:7]), args[7], surface))
            elif cmd == "c":
                b = [T(x) for x in lines[k].split()]
                a = [T(x) for x in lines[k + 1].split()]
                k += 2
                self.prims.append(("cone", self.vec(*b[:3]), b[3], self.vec(*a[:3]), a[3], surface))
            elif cmd in ("p", "pp"):
                n = int(tok[1])
                rows = [[T(x) for x in lines[k + i].split()] for i in range(n)]
                k += n
                pts = [self.vec(*r[:3]) for r in rows]
                if cmd == "p":  # src/nff.rs:249-260: one normal for all vertices
                    nrm = self.normalized(self.cross(pts[1] - pts[0], pts[2] - pts[0]))
                    nrms = [nrm] * n
                else:           # src/nff.rs:177-178
                    nrms = [self.normalized(self.vec(*r[3:6])) for r in rows]
                self.prims.append(("polygon", pts, nrms, surface))
        self.root = self.build_partition()

    # ---- vector helpers: every operation rounds once in T, in the reference's order
    def vec(self, x, y, z):
        return np.array([x, y, z], dtype=self.T)

    def dot(self, a, b):  # src/vector_math.rs:436-438
        return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]

    def cross(self, a, b):  # :428-434
        return self.vec(a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0])

    def magnitude(self, a):  # :25-28
        return np.sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2])

    def normalized(self, a):  # :35-38: three true divisions
        return a / self.magnitude(a)

    def reflected(self, v, n):  # :51-53
        return v - self.T(2) * self.dot(v, n) * n

    def refracted(self, v, n, ri):  # :62-73
        dp = self.dot(v, n)
        a = self.T(1) - ((self.T(1) - dp * dp) / (ri * ri))
        if a >= 0:
            b = np.sqrt(a) + dp / ri
            return v / ri - b * n
        return self.reflected(v, n)

    def interpolate(self, v1, v2, s):  # :440-448
        one = self.T(1)
        return self.normalized(self.vec(v1[0] * s + v2[0] * (one - s), v1[1] * s + v2[1] * (one - s), v1[2] * s + v2[2] * (one - s)))

    # ---- shapes
    def hit_sphere(self, p, src, ray, near):  # src/sphere.rs:49-76
        T = self.T
        _, center, radius, _ = p
        sc = src - center
        a = self.dot(ray, ray)
        b = T(2) * self.dot(ray, sc)
        c = self.dot(sc, sc) - radius * radius
        disc = b * b - T(4) * a * c
        if disc >= 0:
            sq = np.sqrt(disc)
            t = (-b - sq) / (T(2) * a)
            if t < near:
                t = (-b + sq) / (T(2) * a)
            if t >= near:
                return t, self.normalized(src + t * ray - center)
        return None

    def hit_cone(self, p, src, ray, near):  # src/cone.rs:115-214
        T = self.T
        _, base, base_r, apex, apex_r, _ = p
        base_to_apex = apex - base
        w = self.normalized(base_to_apex)
        if abs(w[0]) < abs(w[1]) and abs(w[0]) < abs(w[2]):
            e = self.vec(1, 0, 0)
        elif abs(w[1]) < abs(w[2]):
            e = self.vec(0, 1, 0)
        else:
            e = self.vec(0, 0, 1)
        u = self.cross(w, e)
        v = self.cross(w, u)
        bs = src - base
        s = self.vec(self.dot(bs, u), self.dot(bs, v), self.dot(bs, w))
        r = self.vec(self.dot(ray, u), self.dot(ray, v), self.dot(ray, w))
        mag = self.magnitude(base_to_apex)
        dr = base_r - apex_r
        two = T(2)
        a = r[0] * r[0] + r[1] * r[1] - dr * dr * r[2] * r[2] / (mag * mag)
        b = two * s[0] * r[0] + two * s[1] * r[1] - two * dr * dr * s[2] * r[2] / (mag * mag) + two * base_r * dr * r[2] / mag
        c = s[0] * s[0] + s[1] * s[1] - base_r * base_r - dr * dr * s[2] * s[2] / (mag * mag) + two * base_r * dr * s[2] / mag
        disc = b * b - T(4) * a * c
        if disc >= 0:
            sq = np.sqrt(disc)
            r1 = (-b - sq) / (two * a)
            r2 = (-b + sq) / (two * a)
            w1 = r1 * r[2] + s[2]
            w2 = r2 * r[2] + s[2]
            if r1 < r2 and r1 >= near and w1 >= 0 and w1 <= mag:
                res = r1
            elif r2 >= near and w2 >= 0 and w2 <= mag:
                res = r2
            else:
                res = T(-1)
            if res >= near:
                nu, nv, nw = (res * r[0] + s[0]) * mag, (res * r[1] + s[1]) * mag, dr
                n = self.vec(nu * u[0] + nv * v[0] + nw * w[0], nu * u[1] + nv * v[1] + nw * w[1], nu * u[2] + nv * v[2] + nw * w[2])
                return res, self.normalized(n)
        return None

    def hit_polygon(self, p, src, ray, near):  # src/polygon.rs:61-289
        T = self.T
        _, pts, nrms, _ = p
        n = len(pts)
        g = self.cross(pts[1] - pts[0], pts[2] - pts[0])
        den = self.dot(ray, g)
        if abs(den) < T(0.000001):
            return None
        t = self.dot(pts[0] - src, g) / den
        if t < near:
            return None
        q = src + ray * t
        # (minor, major): the coordinate the edge is cut at, the coordinate traced along (:109-257)
        if abs(g[2]) > abs(g[0]) and abs(g[2]) > abs(g[1]):
            mi, ma = 1, 0
        elif abs(g[1]) > abs(g[0]):
            mi, ma = 2, 0
        else:
            mi, ma = 2, 1
        count = 0
        fwd = rev = None  # (distance, index, scale)
        for i in range(n):
            a, b = pts[i], pts[(i + 1) % n]
            edge = a - b
            if abs(edge[mi]) < T(0.000001):
                continue
            scale = (q[mi] - b[mi]) / edge[mi]
            if scale < 0 or scale > 1:
                continue
            d = scale * edge[ma] + b[ma] - q[ma]
            if d >= 0:
                count += 1
                if fwd is None or d < fwd[0]:
                    fwd = (d, i, scale)
            else:
                if rev is None or d > rev[0]:
                    rev = (d, i, scale)
        if count % 2 == 0 or fwd is None or rev is None:
            return None
        fn = self.normalized(self.interpolate(nrms[fwd[1]], nrms[(fwd[1] + 1) % n], fwd[2]))
        rn = self.normalized(self.interpolate(nrms[rev[1]], nrms[(rev[1] + 1) % n], rev[2]))
        s = rev[0] / (rev[0] - fwd[0])
        return t, self.normalized(self.interpolate(fn, rn, s))

    # ---- bounding boxes: (corner, extent), the far corner always re-derived as corner + extent (src/shape.rs:37-39)
    def bbox(self, p):
        T = self.T
        if p[0] == "sphere":  # src/sphere.rs:12-25
            _, c, r, _ = p
            return self.vec(c[0] - r, c[1] - r, c[2] - r), self.vec(r * T(2), r * T(2), r * T(2))
        if p[0] == "cone":  # src/cone.rs:30-94
            _, base, base_r, apex, apex_r, _ = p
            n = self.normalized(apex - base)
            scale = []
            for k in range(3):
                unit = self.vec(*[1 if j == k else 0 for j in range(3)])
                pk = unit - n * n[k]
                mk = self.magnitude(pk)
                scale.append(pk[k] / mk if mk > 0 else T(0))
            lo, hi = [], []
            for k in range(3):
                be, ae = base_r * scale[k], apex_r * scale[k]
                lo.append(min(base[k] - be, apex[k] - ae))
                hi.append(max(base[k] + be, apex[k] + ae))
            return self.vec(*lo), self.vec(hi[0] - lo[0], hi[1] - lo[1], hi[2] - lo[2])
        _, pts, _, _ = p  # src/polygon.rs:11-56
        lo = [pts[0][k] for k in range(3)]
        hi = list(lo)
        for q in pts[1:]:
            for k in range(3):
                if q[k] < lo[k]:
                    lo[k] = q[k]
                if q[k] > hi[k]:
                    hi[k] = q[k]
        return self.vec(*lo), self.vec(hi[0] - lo[0], hi[1] - lo[1], hi[2] - lo[2])

    def expand_to_fit(self, a, b):  # src/shape.rs:59-82
        lo = [min(a[0][k], b[0][k]) for k in range(3)]
        fa, fb = a[0] + a[1], b[0] + b[1]
        hi = [max(fa[k], fb[k]) for k in range(3)]
        return self.vec(*lo), self.vec(hi[0] - lo[0], hi[1] - lo[1], hi[2] - lo[2])

    def surface_area(self, box):  # src/shape.rs:45-49
        e, two = box[1], self.T(2)
        return e[0] * e[1] * two + e[1] * e[2] * two + e[0] * e[2] * two

    def box_hit(self, box, src, ray, near_cull):  # src/shape.rs:84-172
        T = self.T
        corner, extent = box
        largest_near, smallest_far = T(np.finfo(T).min), T(np.finfo(T).max)
        to_min = corner - src
        to_max = corner + extent - src
        for k in range(3):
            if ray[k] != 0:
                a, b = to_min[k] / ray[k], to_max[k] / ray[k]
                lo, hi = (a, b) if a < b else (b, a)
                if k == 0:  # the x slab assigns, y and z tighten
                    largest_near, smallest_far = lo, hi
                else:
                    if lo > largest_near:
                        largest_near = lo
                    if hi < smallest_far:
                        smallest_far = hi
            elif src[k] < corner[k] or src[k] > corner[k] + extent[k]:
                return False
        return bool(smallest_far >= largest_near and smallest_far >= near_cull)

    # ---- construction of the space partition (src/space_partition.rs:40-258)
    def appraise_split(self, items, axis, plane):  # :100-149
        T = self.T
        under_count = over_count = 0
        under_box = over_box = None
        for _, box in items:
            if box[0][axis] < plane:
                under_count += 1
                under_box = box if under_box is None else self.expand_to_fit(under_box, box)
            if (box[0] + box[1])[axis] >= plane:
                over_count += 1
                over_box = box if over_box is None else self.expand_to_fit(over_box, box)
        if under_box is not None and over_box is not None:
            cost = self.surface_area(under_box) * T(under_count) + self.surface_area(over_box) * T(over_count)
        elif under_box is not None:
            cost = self.surface_area(under_box) * T(under_count)
        elif over_box is not None:
            cost = self.surface_area(over_box) * T(over_count)
        else:
            cost = T(0)
        return under_box, over_box, cost

    def find_splitting_plane(self, items, axis, no_split_cost):  # :40-97
        if len(items) < 4:
            return None
        min_cost, best_plane, best_over, best_under = no_split_cost, self.T(0), None, None
        for _, box in items:
            for plane in (box[0][axis], (box[0] + box[1])[axis]):  # the box's min corner, then its max corner
                under_box, over_box, cost = self.appraise_split(items, axis, plane)
                if cost < min_cost:
                    best_under, best_over, best_plane, min_cost = under_box, over_box, plane, cost
        if best_over is None or best_under is None:
            return None
        if min_cost < no_split_cost:
            return best_under, best_over, best_plane
        return None

    def build_node(self, items, axis, box, depth=0):  # :223-258
        assert depth < 64, "the reference's construction does not terminate on this scene"
        no_split_cost = self.surface_area(box) * self.T(len(items))
        decision = self.find_splitting_plane(items, axis, no_split_cost)
        if decision is None:
            return dict(box=box, leaf=[i for i, _ in items])
        under_box, over_box, plane = decision
        over = [it for it in items if (it[1][0] + it[1][1])[axis] > plane]   # split, :151-172: NOT the
        under = [it for it in items if it[1][0][axis] <= plane]             # inequalities of appraise_split
        nxt = (axis + 1) % 3
        return dict(box=box, axis=axis, plane=plane, over=self.build_node(over, nxt, over_box, depth + 1),
                    under=self.build_node(under, nxt, under_box, depth + 1))

    def build_partition(self):  # :197-221
        if not self.prims:
            return dict(box=(self.vec(0, 0, 0), self.vec(0, 0, 0)), leaf=[])
        items = [(i, self.bbox(p)) for i, p in enumerate(self.prims)]
        total = items[0][1]
        for _, box in items[1:]:
            total = self.expand_to_fit(total, box)
        return self.build_node(items, 0, total)

    def flat_tree(self):
        """Preorder (node, under subtree, over subtree): boxes, axis (-1 = leaf), plane, leaf sizes, leaf lists."""
        boxes, axis, plane, counts, refs = [], [], [], [], []
        def go(n):
            boxes.append([float(x) for x in n["box"][0]] + [float(x) for x in n["box"][1]])
            if "leaf" in n:
                axis.append(-1); plane.append(0.0); counts.append(len(n["leaf"])); refs.extend(n["leaf"])
            else:
                axis.append(n["axis"]); plane.append(float(n["plane"])); counts.append(0)
                go(n["under"]); go(n["over"])
        go(self.root)
        return boxes, axis, plane, counts, refs

    # ---- the walk (src/space_partition.rs:268-385)
    def leaf_closest(self, leaf, src, ray, near, ignore):  # :348-385
        best = None
        for idx in leaf:
            if idx == ignore:
                continue
            p = self.prims[idx]
            h = {"sphere": self.hit_sphere, "cone": self.hit_cone, "polygon": self.hit_polygon}[p[0]](p, src, ray, near)
            if h is not None and (best is None or h[0] < best[0]):
                best = (h[0], h[1], idx)
        return best

    def walk(self, node, src, ray, near, ignore):  # :268-346
        if not self.box_hit(node["box"], src, ray, near):
            return None
        if "leaf" in node:
            return self.leaf_closest(node["leaf"], src, ray, near, ignore)
        axis, plane = node["axis"], node["plane"]
        if src[axis] < plane:
            first, second, beyond = node["under"], node["over"], (lambda e: e > plane)
        else:
            first, second, beyond = node["over"], node["under"], (lambda e: e < plane)
        r = self.walk(first, src, ray, near, ignore)
        if r is None or beyond((src + ray * r[0])[axis]):
            return self.walk(second, src, ray, near, ignore)  # instead of r, whatever it returns
        return r

    def closest(self, src, ray, near, ignore):
        return self.walk(self.root, src, ray, near, ignore)

    # ---- shading and recursion
    def phong(self, s, normal, view, ldir, lcol):  # src/phong.rs:35-62
        T = self.T
        ndv = self.dot(normal, view)
        if ndv > 0:
            return self.vec(0, 0, 0)
        rv = view - T(2) * ndv * normal
        spec = T(0)
        ldr = self.dot(rv, ldir)
        if ldr > 0:
            spec = s["ks"] * np.power(ldr, s["shine"])
        diff = T(0)
        ndl = self.dot(normal, ldir)
        if ndl > 0:
            diff = s["kd"] * ndl
        return self.vec(lcol[0] * (spec + diff * s["color"][0]), lcol[1] * (spec + diff * s["color"][1]),
                        lcol[2] * (spec + diff * s["color"][2]))

    def blinn_phong(self, s, normal, view, ldir, lcol):  # src/blinn_phong.rs:35-56 (no back-face way out)
        T = self.T
        half = self.normalized(ldir - view)
        spec = T(0)
        ndh = self.dot(normal, half)
        if ndh > 0:
            spec = s["ks"] * np.power(ndh, s["shine"])
        diff = T(0)
        ndl = self.dot(normal, ldir)
        if ndl > 0:
            diff = s["kd"] * ndl
        return self.vec(lcol[0] * (spec + diff * s["color"][0]), lcol[1] * (spec + diff * s["color"][1]),
                        lcol[2] * (spec + diff * s["color"][2]))

    def sub_trace(self, src, ray, near, ignore, contribution, depth, max_depth=5, cutoff=0.003):  # src/scene.rs:82-200
        T = self.T
        cut = T(cutoff)
        h = self.closest(src, ray, near, ignore)
        if h is None:
            return self.background.copy()
        dist, normal, idx = h
        s = self.prims[idx][-1]
        pos = src + ray * dist
        back = self.dot(normal, ray) > 0
        total = self.vec(0, 0, 0)
        if not back:
            for lpos, lcol in self.lights:
                to_l = lpos - pos
                ld = self.magnitude(to_l)
                ldir = to_l / ld
                b = self.closest(pos, ldir, T(0), idx)
                if not (b is not None and b[0] <= ld):
                    total = total + (self.blinn_phong if s["blinn"] else self.phong)(s, normal, ray, ldir, lcol)
            if depth < max_depth:
                rc = contribution * s["refl"]
                if rc > cut:
                    c = self.sub_trace(pos, self.reflected(ray, normal), T(0), idx, rc, depth + 1, max_depth, cutoff)
                    total = total + c * s["refl"]
        if depth < max_depth:
            tr = (T(1) if s["trans"] > cut else T(0)) if back else s["trans"]
            tc = contribution * tr
            if tc > cut:
                rr = self.refracted(ray, -normal, T(1) / s["ior"]) if back else self.refracted(ray, normal, s["ior"])
                c = self.sub_trace(pos, rr, T(0.0001), None, tc, depth + 1, max_depth, cutoff)
                total = total + c * tr
        return np.minimum(total, T(1))  # Color::clamp: upper bound only (src/color.rs:11-21)

    def camera(self):  # src/render.rs:81-92
        T = self.T
        v = self.view
        w, h = int(v["resolution"][0]), int(v["resolution"][1])
        up_len = np.tan(T(math.pi) * v["angle"][0] / T(360))  # tan in T
        right_len = up_len * (T(w) / T(h))
        frm, at, upv = self.vec(*v["from"]), self.vec(*v["at"]), self.vec(*v["up"])
        fwd = self.normalized(at - frm)
        right = self.normalized(self.cross(fwd, upv)) * right_len
        up = self.normalized(self.cross(right, fwd)) * up_len
        return frm, fwd, right, up, w, h

    def pixel(self, i, j):  # src/render.rs:101-111, then src/ppm.rs:20-22
        T = self.T
        frm, fwd, right, up, w, h = self.camera()
        sy = T(1) - T(2 * j + 1) / T(h)
        sx = T(-1) + T(2 * i + 1) / T(w)
        ray = self.normalized(fwd + up * sy + right * sx)
        c = self.sub_trace(frm, ray, T(self.view["hither"][0]), None, T(1), 0)
        return c

    @staticmethod
    def quantize(c):  # (c * 255.9) as u8, saturating
        out = []
        for x in c:
            y = float(np.float32(x) * np.float32(255.9)) if isinstance(x, np.float32) else float(x) * 255.9
            out.append(0 if not (y > 0) else 255 if y >= 255 else int(y))
        return out


def scene_text(name):
    if name in EXTRA_SCENES:
        return EXTRA_SCENES[name]
    with open(os.path.join(NFF_DIR, name + ".nff")) as f:
        return f.read()


# Pixels (i = column, j = row) per scene, chosen inside regions -- not on silhouettes -- with what
# the fixture's header comment (or, for the extra scenes, the scene text) says about them.
PIXELS = {
    "test-cone": [(256, 256, "frustum, mid height"), (256, 380, "frustum, wide lower part"), (256, 125, "frustum, narrow upper part"),
                  (300, 256, "right of the axis, towards the light"), (212, 256, "left of the axis"), (256, 20, "above the frustum: background"),
                  (120, 400, "beside the frustum: background")],
    "test-side-lit": [(256, 256, "triangle centre"), (256, 200, "upper part"), (230, 300, "lower left"), (20, 20, "background")],
    "test-overlap": [(256, 256, "red triangle in front"), (256, 200, "red, upper"), (250, 55, "green only, above the red apex"),
                     (256, 360, "green only, below the red base"), (20, 20, "background")],
    "test-shadow-1": [(300, 250, "red square"), (270, 240, "red square, left part"), (280, 180, "green square in the red one's shadow"),
                      (280, 110, "green square, lit, beyond the shadow"), (170, 180, "green square beside the shadow, lit"),
                      (20, 20, "background")],
    "test-shadow-2": [(256, 150, "green triangle, lit, above the shadow"), (250, 260, "in the off-screen square's shadow"),
                      (256, 335, "green, lit, below the shadow"), (190, 260, "green, lit, left of the shadow"),
                      (310, 260, "green, lit, right of the shadow"), (20, 20, "background")],
    "test-triangle": [(256, 256, "white triangle, centre"), (256, 150, "upper part"), (200, 300, "lower left"), (20, 20, "background"),
                      (256, 60, "above the apex: background")],
    "test-sphere": [(255, 255, "sphere centre"), (256, 150, "highlight"), (256, 100, "upper limb"), (0, 0, "background")],
    "test-square": [(256, 256, "square centre"), (110, 110, "inside, near the corner"), (20, 256, "background")],
    "kat-window": [(256, 256, "square seen through the pane"), (200, 300, "the same"), (108, 256, "through the pane, just inside its edge"),
                   (96, 256, "square seen directly, just outside the pane's edge"), (10, 10, "square seen directly, far corner")],
    "kat-patch": [(256, 256, "patch centre"), (256, 200, "towards the upper vertex"), (230, 290, "towards a lower vertex"),
                  (290, 290, "towards the other"), (10, 10, "background")],
}


# Scenes sampled on a grid: (scene, blinn, first column, first row, step, columns, rows).  Of the grid's
# pixels those become known answers whose float32 and float64 bytes agree within 1 (which leaves out
# silhouettes and the chaotic end of mirror paths); at most GRID_KEEP of them per scene, those that
# show an object first.
GRID_SCENES = [
    ("extra-onplane", False, 20, 20, 24, 10, 10),
    ("extra-patch", False, 20, 20, 24, 10, 10),
    ("extra-glass", False, 20, 20, 24, 10, 10),
    ("extra-glass", True, 32, 32, 24, 9, 9),
    ("extra-cones", False, 20, 20, 24, 10, 10),
    ("extra-cones", True, 32, 32, 24, 9, 9),
    ("kat-mixed", False, 8, 102, 12, 21, 6),
    ("kat-mixed", True, 14, 106, 12, 20, 5),
    ("test-sphere", True, 100, 100, 52, 7, 7),
]
GRID_KEEP = 12

# Explicit pixels of the scenes written here
PIXELS.update({
    "kat-tir": [(256, 256, "the floor, by way of a total internal reflection inside the glass"), (250, 262, "the same"),
                (256, 200, "the floor, seen directly above the glass"), (256, 40, "background")],
    "kat-cylinder": [(256, 256, "cylinder, head-on, light at the eye"), (256, 200, "cylinder, same column"), (230, 256, "towards the limb"),
                     (256, 4, "beyond the open end: background")],
})


def kat_key(name, blinn):
    return name + ("+blinn" if blinn else "")


def make():
    kats, trees = [], {}
    def one(name, blinn, t32, t64, i, j, what, must):
        c32 = t32.pixel(i, j)
        b32, b64 = Tracer.quantize(c32), Tracer.quantize(t64.pixel(i, j))
        ok = max(abs(a - b) for a, b in zip(b32, b64)) <= 1
        assert ok or not must, (name, i, j, b32, b64, "float32 and float64 disagree: not a known answer")
        if ok:
            return {"scene": name, "blinn": blinn, "x": i, "y": j, "rgb": b32, "f32": [float(x) for x in c32], "what": what}
        return None
    def keep_tree(name, t32):
        boxes, axis, plane, counts, refs = t32.flat_tree()
        if len(axis) > 1 and name not in trees:  # bit patterns: JSON numbers would have to round-trip
            bits = lambda v: [int(np.float32(x).view(np.uint32)) for x in v]
            trees[name] = {"boxes": [bits(b) for b in boxes], "axis": axis, "plane": bits(plane), "count": counts, "refs": refs}
    for name, pixels in PIXELS.items():
        text = scene_text(name)
        t32, t64 = Tracer(text, np.float32), Tracer(text, np.float64)
        keep_tree(name, t32)
        for (i, j, what) in pixels:
            kats.append(one(name, False, t32, t64, i, j, what, True))
    for name, blinn, x0, y0, step, nx, ny in GRID_SCENES:
        text = scene_text(name)
        t32, t64 = Tracer(text, np.float32, blinn), Tracer(text, np.float64, blinn)
        keep_tree(name, t32)
        bg = Tracer.quantize(t32.background)
        found = [one(name, blinn, t32, t64, x0 + step * a, y0 + step * b, "grid sample", False) for b in range(ny) for a in range(nx)]
        found = [k for k in found if k is not None]
        objects = [k for k in found if k["rgb"] != bg]
        sky = [k for k in found if k["rgb"] == bg]
        stride = max(1, len(objects) // (GRID_KEEP - 1))
        chosen = objects[::stride][: GRID_KEEP - 1] + sky[:1]
        assert len(chosen) >= 6, (name, blinn, len(found), len(objects))
        kats.extend(chosen)
    closed_form_checks({(kat_key(k["scene"], k["blinn"]), k["x"], k["y"]): k["rgb"] for k in kats})
    return kats, trees


def closed_form_checks(px):
    """What the fixtures' header comments promise, and values that follow from the scene files
    in closed form (plain float64 arithmetic below, no tracer)."""
    near = lambda a, b: abs(a - b) <= 1
    black = [0, 0, 0]
    for key, rgb in px.items():
        if key[0].startswith("test-") and "+" not in key[0] and key[0] not in ("test-overlap", "test-shadow-1", "test-shadow-2"):
            assert rgb[0] == rgb[1] == rgb[2], (key, rgb)  # "white ... on a black background": grey levels only
    # test-triangle / test-square: head-on, light behind the eye on the axis.  At the centre pixel the
    # view direction is (1, +-1/512 tan 22.5deg...) ~ the axis, N.L ~ 1: kd 0.5 + ks 0.5 * (R.L)^1 ~ 1 -> 255.
    assert px[("test-triangle", 256, 256)] == [255, 255, 255] and px[("test-square", 256, 256)] == [255, 255, 255]
    assert px[("test-triangle", 20, 20)] == black and px[("test-triangle", 256, 60)] == black
    # test-side-lit: the triangle's plane contains (2, 0.577, 0) and is turned 45 degrees about y; eye at the
    # origin looking along x, light at (2, 0, 2).  Centre pixel: hit P ~ (2, 0, 0) (the plane passes through
    # x = 2 on the axis), N = (-1, 0, 1)/sqrt2 (towards eye and light), L = (0, 0, 1), V = (1, 0, 0):
    # N.L = 1/sqrt2, R = V - 2(N.V)N = (0, 0, 1), R.L = 1: 0.5 * 0.7071 + 0.5 * 1 = 0.85355 -> 218.
    assert near(px[("test-side-lit", 256, 256)][0], int((0.5 / math.sqrt(2) + 0.5) * 255.9))
    # test-overlap: red (kd 0.5, ks 0.5, colour (1, 0, 0)) in front of green.  Centre: N.L ~ 1, R.L ~ 1:
    # r = 0.5 * 1 + 0.5 = 1 -> 255, g = b = specular only = 0.5 -> 127.
    assert px[("test-overlap", 256, 256)] == [255, 127, 127] or all(near(a, b) for a, b in zip(px[("test-overlap", 256, 256)], [255, 127, 127]))
    for key in (("test-overlap", 250, 55), ("test-overlap", 256, 360)):
        r, g, b = px[key]
        assert g > r and r == b  # green only where the red triangle does not cover
    # test-shadow-1: the light (-5, 0, 0), the red square (x = 0, |y|,|z| <= 1) and the green square (x = 2,
    # |y|,|z| <= 2) are in line: the shadow on the green square is |y|,|z| <= 1.4; inside it no light reaches
    # the surface and the mirror ray (ks 0.5) leaves the scene: black.
    assert px[("test-shadow-1", 280, 180)] == black
    for key in (("test-shadow-1", 280, 110), ("test-shadow-1", 170, 180)):
        r, g, b = px[key]
        assert g > 0 and g > r and r == b  # lit green: diffuse in g only, the same specular in all three
    r, g, b = px[("test-shadow-1", 300, 250)]
    assert r > 0 and r > g and g == b  # red square
    # test-shadow-2: "a green triangle ... with a shadow cast by an off-screen square": some pixel of the
    # triangle is lit (green over the specular grey), and the blocker itself is not in view
    for key in (("test-shadow-2", 256, 150), ("test-shadow-2", 256, 335), ("test-shadow-2", 190, 260), ("test-shadow-2", 310, 260)):
        r, g, b = px[key]
        assert g > r and r == b
    assert px[("test-shadow-2", 250, 260)] == black  # the shadow: no light, and the mirror ray leaves the scene
    # kat-window: behind the pane the square's colour arrives times 0.8 (the pane itself adds nothing: kd = ks = 0);
    # square: kd 0.75, colour (0.2, 0.6, 1), head-on light: centre = 0.75 * (0.2, 0.6, 1) * N.L (~1)
    direct = [0.75 * c for c in (0.2, 0.6, 1.0)]
    assert all(near(a, int(0.8 * d * 255.9)) or near(a, int(0.8 * d * 255.9 * 0.999)) for a, d in zip(px[("kat-window", 256, 256)], direct))
    seen, bare = px[("kat-window", 108, 256)], px[("kat-window", 96, 256)]
    assert all(0.76 < s / b_ < 0.84 for s, b_ in zip(seen[1:], bare[1:]))  # 0.8 of the light on either side of the pane's edge
    # kat-tir: 0.9 (the front face's transmittance) x 1 (a back face passes everything on, src/scene.rs:151-165)
    # x the floor's own colour: kd 1 x N.L x (0.2, 0.8, 0.4), N = (0, 1, 0), L = (0, 1.5, 2.5) / |.| at the
    # point (3, -2, 0) under the mirror face: N.L = 0.5145.  Without the total reflection the ray would
    # leave through the tilted face towards +x and find the background: black.
    ndl = 1.5 / math.hypot(1.5, 2.5)
    assert all(near(a, int(0.9 * ndl * c * 255.9)) for a, c in zip(px[("kat-tir", 256, 256)], (0.2, 0.8, 0.4))), px[("kat-tir", 256, 256)]
    assert px[("kat-tir", 256, 40)] == black
    # kat-cylinder: axis along y through (3, 0, 0), radius 0.5, light at the eye with colour (1, 0.5, 0.25): at the
    # centre N = -V = L, so N.L = 1 and R.L = 1: light x (ks + kd) = light -> (255, 127, 63); the same all along the
    # column (N.L = cos of the elevation only through L's y component: N stays (-1, 0, 0), L tilts)
    assert all(near(a, b) for a, b in zip(px[("kat-cylinder", 256, 256)], (255, 127, 63))), px[("kat-cylinder", 256, 256)]
    r, g, b = px[("kat-cylinder", 230, 256)]
    assert r < 255 and near(g, r // 2) and near(b, r // 4)  # dimmer towards the limb, the light's colour ratios kept
    assert px[("kat-cylinder", 256, 4)] == black
    # test-cone: "lit from the front-right": brighter right of the axis than left of it
    assert px[("test-cone", 300, 256)][0] > px[("test-cone", 212, 256)][0]
    assert px[("test-cone", 256, 20)] == black and px[("test-cone", 120, 400)] == black


if __name__ == "__main__":
    kats, trees = make()
    out = os.path.join(HERE, "kats.json")
    with open(out, "w") as f:
        json.dump({"made_by": "tests/golden/make_kats.py", "scenes": EXTRA_SCENES, "trees": trees, "pixels": kats}, f, indent=1)
    print("wrote %d known-answer pixels over %d scenes and %d partitions to %s"
          % (len(kats), len({(k["scene"], k["blinn"]) for k in kats}), len(trees), out))
