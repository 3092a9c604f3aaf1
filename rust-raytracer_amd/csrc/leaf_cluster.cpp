// Bounding-sphere hierarchy over the spheres of one leaf (host side).
//
// The reference walks every primitive of a leaf (src/space_partition.rs:348-385); when its
// tree degenerates into leaves of thousands of spheres that walk is the whole cost.  The
// device may skip a group of spheres when it can PROVE that the reference's test would reject
// each of them for this ray.  The (spatially re-ordered) sphere run is grouped 8 entries to a
// level-1 record, 8 level-1 records to a level-2 record, and so on up to level 4; every record
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
// The device rejects the group iff  a * (|E|^2 (1 - beta) - q) > (ray.E)^2,  which in exact
// arithmetic means dist(line, C)^2 > q + beta |E|^2.  With alpha = 1/16, q = R'^2 (1 + alpha)
// rounded up, and beta = 2.2e-5 >= k0^2 * 17 (1.95e-5) + 16u (the f32 error of evaluating the
// test itself, relative to a |E|^2: 9u in (ray.E)^2, 7u in the left side), a rejected group
// holds no sphere the reference could accept.
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

// Block layout in `bnd` (4 floats per record, indices in records):
//   [h]            header, as integers: {first level-2 record, first level-3, first level-4, top level}
//   [h+1]          one record bounding the whole run
//   [h+2 ...]      level-1 records (one per 8 entries), then levels 2, 3, 4.
// Record i of level L bounds records 8i .. 8i+7 of level L-1 (entries 8i .. 8i+7 for L = 1).
// The top level is the lowest one with at most 16 records (level 4 if none is that small); the
// device tests it linearly and descends from there.
uint32_t cluster_spheres(std::vector<LeafSphere>& v, std::vector<float>& bnd) {
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
    uint32_t first[5] = {0, h + 2, 0, 0, 0};
    size_t count[5] = {0, 0, 0, 0, 0};
    size_t span = 8;  // entries per record of the level
    for (int L = 1; L <= 4; ++L, span *= 8) {
        first[L] = (uint32_t)(bnd.size() / 4);
        count[L] = (n + span - 1) / span;
        for (size_t i = 0; i < count[L]; ++i) {
            float rec[4];
            bound(v, i * span, std::min(n, (i + 1) * span), rec);
            bnd.insert(bnd.end(), rec, rec + 4);
        }
    }
    uint32_t top = 4;
    for (int L = 4; L >= 1; --L)
        if (count[L] <= 16) top = (uint32_t)L;
    const uint32_t hdr[4] = {first[2], first[3], first[4], top};
    memcpy(&bnd[(size_t)h * 4], hdr, sizeof(hdr));
    return h;
}

}  // namespace rtamd
