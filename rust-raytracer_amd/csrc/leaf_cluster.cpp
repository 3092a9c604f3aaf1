// Bounding-sphere hierarchy over the spheres of one leaf (host side).
//
// The reference walks every primitive of a leaf (src/space_partition.rs:348-385); when its
// tree degenerates into leaves of thousands of spheres that walk is the whole cost.  The
// device may skip a group of spheres when it can PROVE that the reference's test would reject
// each of them for this ray.  The (spatially re-ordered) sphere run is grouped 8 entries to a
// level-1 record, 8 level-1 records to a level-2 record, and so on up to level 5; every record
// is a bounding sphere {centre C, q} of all entries below it.  Re-ordering is free: list
// order only matters for ties, and every entry keeps its list position for that.
//
// What the reference can accept.  With u = 2^-24, e = src - c, a = |ray|^2, Sphere::intersect
// (src/sphere.rs:49-56) returns a hit only if fl(b*b - 4*a*c) >= 0.  Bounding the rounding of
// its 3 subtractions, two dot products, r*r and the products gives
//     disc/4 >= (ray.e)^2 - a*(|e|^2 - r^2) - a*(19u |e|^2 + 6u r^2),
// so acceptance implies  dist(line, c)^2 <= r^2 (1 + 6u) + 19u |e|^2,  i.e.
//     dist(line, c) <= r (1 + 3u) + k0 |e|,   k0 = sqrt(19u) = 1.07e-3.
// For a sphere inside a bounding sphere (C, R) (|c - C| + r <= R) and E = src - C this gives
//     dist(line, C) <= R' + k0 |E|,   R' = R (1 + 1.07e-3).
// Since 2 k0 R' |E| <= alpha R'^2 + (k0^2 / alpha) |E|^2 for any alpha > 0,
//     (R' + k0 |E|)^2 <= R'^2 (1 + alpha) + k0^2 (1 + 1/alpha) |E|^2.
// The device rejects the group iff  |E|^2 (1 - beta) - q > (n.E)^2  with n = ray / |ray|, which
// in exact arithmetic means dist(line, C)^2 > q + beta |E|^2.  With alpha = 1/16, q = R'^2
// (1 + alpha) rounded up, and beta = 2.2e-5 >= k0^2 * 17 (1.95e-5) + 22.1u (the f32 error of
// evaluating the test itself, relative to |E|^2: 9u in (n.E)^2, 6.1u from n's own rounding,
// 7u in the left side), a rejected group holds no sphere the reference could accept.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "rt_host.h"

namespace rtamd {

namespace {

const double CULL_ALPHA = 1.0 / 16.0;
const double CULL_K0 = 1.07e-3;  // device side: beta = 2.2e-5 in rt_traverse.h

float round_up(double x) {
    float f = (float)x;
    if ((double)f < x) f = std::nextafter(f, INFINITY);
    return std::nextafter(f, INFINITY);
}

float round_down(double x) {
    float f = (float)x;
    if ((double)f > x) f = std::nextafter(f, -INFINITY);
    return std::nextafter(f, -INFINITY);
}

// Radius of the smallest sphere about `c` that holds entries [b, e) of v.
double radius_about(const std::vector<LeafSphere>& v, size_t b, size_t e, const double c[3], size_t* farthest) {
    double R = 0.0;
    for (size_t i = b; i < e; ++i) {
        double d2 = 0.0;
        for (int a = 0; a < 3; ++a) {
            const double d = (double)v[i].c[a] - c[a];
            d2 += d * d;
        }
        const double r = std::sqrt(d2) + std::fabs((double)v[i].r);
        if (!(r <= R)) {  // also catches NaN
            R = r;
            if (farthest) *farthest = i;
        }
    }
    return R;
}

// Centre of a small sphere around entries [b, e): starts at the middle of their box and walks
// towards the farthest point with a shrinking step (Badoiu-Clarkson), keeping the best centre.
double enclose(const std::vector<LeafSphere>& v, size_t b, size_t e, double c[3], int iterations) {
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (size_t i = b; i < e; ++i)
        for (int a = 0; a < 3; ++a) {
            lo[a] = std::min(lo[a], (double)v[i].c[a] - std::fabs((double)v[i].r));
            hi[a] = std::max(hi[a], (double)v[i].c[a] + std::fabs((double)v[i].r));
        }
    double cur[3];
    for (int a = 0; a < 3; ++a) cur[a] = c[a] = 0.5 * (lo[a] + hi[a]);
    size_t far = b;
    double best = radius_about(v, b, e, cur, &far);
    for (int it = 1; it <= iterations && best == best && best < 1e300; ++it) {
        double dir[3], len = 0.0;
        for (int a = 0; a < 3; ++a) {
            dir[a] = (double)v[far].c[a] - cur[a];
            len += dir[a] * dir[a];
        }
        len = std::sqrt(len);
        const double reach = len + std::fabs((double)v[far].r);  // distance of the farthest point
        for (int a = 0; a < 3; ++a) cur[a] += (len > 0.0 ? dir[a] / len : 0.0) * reach / (it + 1);
        const double R = radius_about(v, b, e, cur, &far);
        if (R < best) {
            best = R;
            for (int a = 0; a < 3; ++a) c[a] = cur[a];
        }
    }
    return best;
}

// {C, q} bounding entries [b, e) of v.  The centre is rounded to f32 first and the radius
// taken about the rounded centre, so the record bounds what it says whatever the search did.
void bound(const std::vector<LeafSphere>& v, size_t b, size_t e, float out[4]) {
    double c[3];
    enclose(v, b, e, c, 64);
    double cf[3];
    for (int a = 0; a < 3; ++a) {
        out[a] = (float)c[a];
        cf[a] = (double)out[a];
    }
    const double R = radius_about(v, b, e, cf, nullptr);
    const double Rp = R * (1.0 + CULL_K0) * (1.0 + 1e-6);
    out[3] = round_up(Rp * Rp * (1.0 + CULL_ALPHA));
    if (!(out[3] == out[3]) || !(R == R)) out[3] = INFINITY;  // never cull what cannot be bounded
}

// Cost of keeping entries [b, e) in one group: a ray's chance of passing a bounding sphere goes
// with its cross-section, and passing costs work in proportion to the entries inside.
double group_cost(const std::vector<LeafSphere>& v, size_t b, size_t e) {
    double c[3];
    const double R = enclose(v, b, e, c, 0);
    return R * R * (double)(e - b);
}

// Recursive two-way splits.  Records group entries by absolute index (8i .. 8i+7, 64i .. 64i+63,
// ...), so split points are multiples of the largest group size that has one inside (b, e): no
// group of that size straddles a split.  Among the three axes (entries sorted by centre) and up
// to 15 such points around the middle, the split with the least summed cost wins.
void order(std::vector<LeafSphere>& v, size_t b, size_t e) {
    size_t g = 0, first = 0;
    for (size_t cand : {(size_t)4096, (size_t)512, (size_t)64, (size_t)8}) {
        first = (b / cand + 1) * cand;  // first multiple above b
        if (first < e) {
            g = cand;
            break;
        }
    }
    if (g == 0) return;
    std::vector<size_t> points;
    for (size_t m = first; m < e; m += g) points.push_back(m);
    if (points.size() > 15) {  // keep 15 evenly spread
        std::vector<size_t> some;
        for (int k = 0; k < 15; ++k) some.push_back(points[(points.size() - 1) * k / 14]);
        points.swap(some);
    }
    std::vector<LeafSphere> best_order;
    double best = INFINITY;
    size_t best_m = points[points.size() / 2];
    std::vector<LeafSphere> tmp(v.begin() + b, v.begin() + e);
    for (int ax = 0; ax < 3; ++ax) {
        auto key = [ax](const LeafSphere& s) { return s.c[ax] == s.c[ax] ? s.c[ax] : INFINITY; };  // total order
        std::sort(tmp.begin(), tmp.end(), [&key](const LeafSphere& x, const LeafSphere& y) {
            const float kx = key(x), ky = key(y);
            if (kx != ky) return kx < ky;
            return x.pos < y.pos;  // deterministic
        });
        for (size_t m : points) {
            const double cost = group_cost(tmp, 0, m - b) + group_cost(tmp, m - b, e - b);
            if (cost < best) {
                best = cost;
                best_m = m;
                best_order = tmp;
            }
        }
    }
    if (!best_order.empty()) std::copy(best_order.begin(), best_order.end(), v.begin() + b);
    order(v, b, best_m);
    order(v, best_m, e);
}

// Pads the run to a multiple of 8 entries with copies of its last sphere, then orders it.
void arrange(std::vector<LeafSphere>& v) {
    while (v.size() % 8) {  // a copy never wins: same distance as its original, later position
        v.push_back(v.back());
        v.back().pos |= 0x80000000u;
    }
    order(v, 0, v.size());
}

}  // namespace

// ---------------------------------------------------------------------------------------
// Polygons.  What the reference can accept (src/polygon.rs:58-293, restated in rt_shapes.h
// isect_polygon): with gn = fl((v1-v0) x (v2-v0)), t = fl(fl((v0-src).gn) / fl(ray.gn)) >= near
// and P = fl(src + ray t), the crossing test only passes if P's two kept coordinates lie in
// the polygon's coordinate ranges (a crossing edge has scale in [0,1], and P's major
// coordinate lies between two crossing points), each up to 8u(M + |src| + |X - src|), where
// X = src + ray t exactly -- a point of the ray's LINE -- and M bounds the vertex coordinates.
// The dropped coordinate D is the one where |gn| is largest, and (X - v0).gn is a rounding
// residue of at most u |gn| (5|v0 - src| + 3|X - src|), so
//     |X_D - v0_D| <= |gn_m/gn_D| |X_m - v0_m| + |gn_M/gn_D| |X_M - v0_M| + sqrt(3) residue/|gn|,
// whatever the quality of gn (the ratios are at most 1).  X therefore lies in a box: the
// polygon's ranges on the kept axes and v0_D +- h on the dropped one, all widened by at most
// 32u (M + |src| + |X - src| + |v0 - src|).  With C the box's centre and R its half diagonal
// this gives  |X - C| <= R (1 + 2e-5) + 2e-5 |C| + 1.1e-5 |src - C|,  the form the sphere
// hierarchy's test is built for (its margins are wider than these).  Which axis is dropped is
// decided here in binary64 with the worst f32 deviation of either kernel build (plain or fused
// multiply-add) as a dead band; inside the band every axis is allowed for.
// ---------------------------------------------------------------------------------------
bool polygon_bound(const rt_vertex* vertices, uint32_t first, uint32_t count, float c_out[3], float* r_out, float* box6) {
    const double u = 5.9604644775390625e-8;
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (uint32_t i = 0; i < count; ++i)
        for (int a = 0; a < 3; ++a) {
            const double x = vertices[first + i].point[a];
            if (!(std::fabs(x) < 1e30)) return false;
            lo[a] = std::min(lo[a], x);
            hi[a] = std::max(hi[a], x);
        }
    const float* p0 = vertices[first].point;
    const float* p1 = vertices[first + 1].point;
    const float* p2 = vertices[first + 2].point;
    double ea[3], eb[3];
    for (int a = 0; a < 3; ++a) {
        ea[a] = (double)(float)((double)p1[a] - (double)p0[a]);  // the kernels' f32 edge vectors
        eb[a] = (double)(float)((double)p2[a] - (double)p0[a]);
    }
    const double g[3] = {ea[1] * eb[2] - ea[2] * eb[1], ea[2] * eb[0] - ea[0] * eb[2], ea[0] * eb[1] - ea[1] * eb[0]};
    const double la = std::sqrt(ea[0] * ea[0] + ea[1] * ea[1] + ea[2] * ea[2]);
    const double lb = std::sqrt(eb[0] * eb[0] + eb[1] * eb[1] + eb[2] * eb[2]);
    const double noise = 4.0 * u * la * lb;  // |kernel's gn - g| per component, either build
    int order[3] = {0, 1, 2};
    std::sort(order, order + 3, [&g](int x, int y) { return std::fabs(g[x]) > std::fabs(g[y]); });
    const bool clear = std::fabs(g[order[0]]) - std::fabs(g[order[1]]) > 4.0 * noise && std::fabs(g[order[0]]) > 8.0 * noise;
    double blo[3] = {1e300, 1e300, 1e300}, bhi[3] = {-1e300, -1e300, -1e300};  // union of the allowed boxes
    for (int D = 0; D < 3; ++D) {
        if (clear && D != order[0]) continue;
        const int m = (D + 1) % 3, M = (D + 2) % 3;
        double rm = 1.0, rM = 1.0;
        if (clear) {
            rm = std::min(1.0, (std::fabs(g[m]) + noise) / (std::fabs(g[D]) - noise));
            rM = std::min(1.0, (std::fabs(g[M]) + noise) / (std::fabs(g[D]) - noise));
        }
        const double Em = std::max(std::fabs(lo[m] - p0[m]), std::fabs(hi[m] - p0[m]));
        const double EM = std::max(std::fabs(lo[M] - p0[M]), std::fabs(hi[M] - p0[M]));
        const double h = rm * Em + rM * EM;
        double l3[3], h3[3];
        l3[m] = lo[m]; h3[m] = hi[m];
        l3[M] = lo[M]; h3[M] = hi[M];
        l3[D] = std::min(lo[D], (double)p0[D] - h);  // (the vertices themselves stay inside too)
        h3[D] = std::max(hi[D], (double)p0[D] + h);
        for (int a = 0; a < 3; ++a) {
            blo[a] = std::min(blo[a], l3[a]);
            bhi[a] = std::max(bhi[a], h3[a]);
        }
    }
    double cc[3], R = 0.0, cn = 0.0;
    for (int a = 0; a < 3; ++a) {
        c_out[a] = (float)(0.5 * (blo[a] + bhi[a]));
        cc[a] = c_out[a];
        const double half = std::max(bhi[a] - cc[a], cc[a] - blo[a]);
        R += half * half;
        cn += cc[a] * cc[a];
    }
    R = std::sqrt(R);
    const double Rp = R * (1.0 + 2e-5) + 2e-5 * std::sqrt(cn);
    *r_out = round_up(Rp);
    if (box6) {
        // The same box, as a box: the widening of 32u (M + |src| + |X - src| + |v0 - src|) is at most
        // 1.9e-6 (3 R + 2 |C| + 3 |src - C|) + ..., of which 2e-5 (R + |C|) goes into the box here and
        // 1.1e-5 |src - C| is left to the ray (the shares the sphere above takes).
        const double s = 2e-5 * (R + std::sqrt(cn));
        for (int a = 0; a < 3; ++a) {
            box6[a] = round_down(blo[a] - s);
            box6[3 + a] = round_up(bhi[a] + s);
        }
    }
    return Rp == Rp && *r_out < 1e30f;
}

// ---------------------------------------------------------------------------------------
// Cylinders: cones whose two radii are equal (as f32 numbers).  What the reference can accept
// (src/cone.rs:97-221, restated in rt_shapes.h cone_frame + isect_cone), with u = 2^-24:
//
// The frame.  w = fl(normalized(apex - base)); U = cross(w, e) for the axis e of w's smallest
// component is exact in f32 (products with 0 and 1), so U is orthogonal to w and |U|^2 =
// |w|^2 - w_e^2 >= 2/3; V = fl(cross(w, U)) is within 3.5u |U| of w x U.  The test works in
// the coordinates S = ((src - base).U, .V, .w), R = (ray.U, .V, .w), each computed within
// 4u |src - base| |U| resp. 3u |ray| |U| of its exact value.
//
// The quadratic.  With equal radii dr = 0 exactly, every term that carries dr is an exact zero,
// and with Sr = (S.x, S.y), Rr = (R.x, R.y) the computed coefficients are
//     a = |Rr|^2 (1 + 2u')            -- a sum of squares: no cancellation,
//     b = 2 Sr.Rr + eb,   |eb| <= 4u |Sr| |Rr|,
//     c = |Sr|^2 - br^2 + ec,   |ec| <= 3u |Sr|^2 + 2u br^2,
//     D = fl(b b - 4 a c) = b^2 - 4 a c + eD,   |eD| <= 2u b^2 + 2u |4 a c|.
// Let t be the exact root (-b -+ sqrt D) / (2a) of these computed numbers; it solves
// a t^2 + b t + (c - eD/(4a)) = 0, so the point of the computed line at t has
//     |Sr + t Rr|^2 = br^2 - ec - (a - |Rr|^2) t^2 - eb t + eD/(4a)
//                  <= br^2 (1 + 4u) + u (13 |Sr|^2 + 8 |Sr| rho + 2 rho^2),   rho = |Sr + t Rr|,
// which gives rho^2 <= br^2 (1 + 11u) + 17.1u |Sr|^2.  The root the reference really uses is
// fl((-b -+ fl(sqrt D)) / (2a)) = t + er with |er| |Rr| <= 3.1u (br + |Sr|), so the accepted
// point X = src + ray r lies, in these coordinates, within
//     br (1 + 10u) + (sqrt(17.1u) + 3.1u) |Sr|  =  br (1 + 10u) + 1.0098e-3 |Sr|
// of the axis, and its third coordinate passes the reference's 0 <= w <= |axis| test up to
// 2u (|axis| + |S.z|).  (near >= 0 is required: for a ray along the axis the reference falls
// through to a result of -1, which a negative near would accept.)
//
// Back to space.  (U, w x U, w) is an orthogonal triple with lengths |U|, |w||U|, |w|, so a
// coordinate distance d from the axis is a distance of at most d (1 + 6u) / |U| in space, and
// |Sr| <= |U| |src - base| (1 + 10u).  With C the axis's midpoint, Rc^2 = (|axis|/2)^2 +
// (br/|U|)^2 and E = src - C this puts X -- a point of the ray's line -- within
//     Rc (1 + 1.014e-3) + 1.011e-3 |E|
// of C: the form the sphere hierarchy is built for (r (1 + 3u) + 1.07e-3 |e|) with r = Rc
// (1 + 1.03e-3).  The same bounds hold for the build that contracts multiply-adds (fewer
// roundings).  They assume that no product underflows, which would take a ray parallel to a
// cylinder's axis to within 1e-15 of its length.
// ---------------------------------------------------------------------------------------
bool cylinder_bound(const rt_cone& cone, float c_out[3], float* r_out) {
    if (!(cone.base_radius == cone.apex_radius)) return false;
    double axis[3], len2 = 0.0;
    for (int a = 0; a < 3; ++a) {
        if (!(std::fabs((double)cone.base[a]) < 1e15) || !(std::fabs((double)cone.apex[a]) < 1e15)) return false;
        axis[a] = (double)cone.apex[a] - (double)cone.base[a];
        len2 += axis[a] * axis[a];
    }
    const double len = std::sqrt(len2), br = std::fabs((double)cone.base_radius);
    if (!(len > 1e-15) || !(br < 1e15)) return false;
    double w_min = 1.0;
    for (int a = 0; a < 3; ++a) w_min = std::min(w_min, std::fabs(axis[a]) / len);
    const double u_len = std::sqrt(1.0 - w_min * w_min) * (1.0 - 1e-6);  // |U|, from below (>= 0.81)
    double cc[3], off2 = 0.0;
    for (int a = 0; a < 3; ++a) {
        const double mid = (double)cone.base[a] + 0.5 * axis[a];
        c_out[a] = (float)mid;
        cc[a] = c_out[a];
        off2 += (cc[a] - mid) * (cc[a] - mid);
    }
    const double Rc = std::sqrt(0.25 * len2 + (br / u_len) * (br / u_len));
    const double Rp = Rc * (1.0 + 1.03e-3) + std::sqrt(off2);
    *r_out = round_up(Rp);
    return Rp == Rp && *r_out < 1e30f;
}

void sphere_record(const LeafSphere& e, float out[4]) {
    const std::vector<LeafSphere> one(1, e);
    bound(one, 0, 1, out);
}

// Block layout in `bnd` (4 floats per record, indices in records):
//   [h]            header, as integers: {first level-2 record, first level-3, first level-4,
//                  top level | padded entry count << 3}
//   [h+1]          one record bounding the whole run
//   [h+2 ...]      level-1 records (one per 8 entries), then levels 2, 3, 4, 5, one after the other
//                  (level 5 starts where level 4 ends: the header has no word for it).
// Record i of level L bounds records 8i .. 8i+7 of level L-1 (entries 8i .. 8i+7 for L = 1).
// The top level is the lowest one with at most 16 records (level `max_level`, 4 or 5, if none is that small); the
// device tests it linearly and descends from there.
uint32_t cluster_spheres(std::vector<LeafSphere>& v, std::vector<float>& bnd, int max_level) {
    const uint32_t h = (uint32_t)(bnd.size() / 4);
    if (v.empty()) return h;
    arrange(v);
    const size_t n = v.size();  // a multiple of 8
    bnd.insert(bnd.end(), 4, 0.0f);
    {
        float rec[4];
        bound(v, 0, n, rec);
        bnd.insert(bnd.end(), rec, rec + 4);
    }
    uint32_t first[6] = {0, h + 2, 0, 0, 0, 0};
    size_t count[6] = {0, 0, 0, 0, 0, 0};
    size_t span = 8;  // entries per record of the level
    for (int L = 1; L <= 5; ++L, span *= 8) {
        first[L] = (uint32_t)(bnd.size() / 4);
        count[L] = (n + span - 1) / span;
        for (size_t i = 0; i < count[L]; ++i) {
            float rec[4];
            bound(v, i * span, std::min(n, (i + 1) * span), rec);
            bnd.insert(bnd.end(), rec, rec + 4);
        }
    }
    uint32_t top = (uint32_t)max_level;
    for (int L = max_level; L >= 1; --L)
        if (count[L] <= 16) top = (uint32_t)L;
    const uint32_t hdr[4] = {first[2], first[3], first[4], top | ((uint32_t)n << 3)};
    memcpy(&bnd[(size_t)h * 4], hdr, sizeof(hdr));
    return h;
}

void cluster_boxes(const std::vector<LeafSphere>& v, const std::vector<float>& bnd, uint32_t h, std::vector<float>& box) {
    const size_t records = bnd.size() / 4;
    const size_t had = box.size() / 8;
    box.resize(records * 8);
    for (size_t i = had; i < records; ++i) {
        float* b = &box[i * 8];
        b[0] = b[1] = b[2] = -INFINITY;
        b[4] = b[5] = b[6] = INFINITY;
        b[3] = b[7] = 0.0f;
    }
    const size_t n = v.size();
    if (n == 0 || (size_t)h + 2 > records) return;
    uint32_t hdr[4];
    memcpy(hdr, &bnd[(size_t)h * 4], sizeof(hdr));
    uint32_t first[6] = {0, h + 2, hdr[0], hdr[1], hdr[2], 0};
    first[5] = first[4] + (uint32_t)((n + 4095) / 4096);
    auto unite = [&](size_t rec, size_t b, size_t e) {
        float* o = &box[rec * 8];
        for (int a = 0; a < 3; ++a) {
            float lo = INFINITY, hi = -INFINITY;
            for (size_t i = b; i < e; ++i) {
                lo = std::min(lo, v[i].lo[a]);  // (min and max of f32 values: exact)
                hi = std::max(hi, v[i].hi[a]);
                if (!(v[i].lo[a] == v[i].lo[a]) || !(v[i].hi[a] == v[i].hi[a])) { lo = -INFINITY; hi = INFINITY; break; }
            }
            o[a] = lo;
            o[4 + a] = hi;
        }
    };
    unite((size_t)h + 1, 0, n);
    size_t span = 8;
    for (int L = 1; L <= 5; ++L, span *= 8) {
        const size_t count = (n + span - 1) / span;
        for (size_t i = 0; i < count; ++i) unite((size_t)first[L] + i, i * span, std::min(n, (i + 1) * span));
    }
}

}  // namespace rtamd
