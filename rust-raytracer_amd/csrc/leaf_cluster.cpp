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
#include <cstdlib>
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

// {C, q} bounding entries [b, e) of v
void bound(const std::vector<LeafSphere>& v, size_t b, size_t e, float out[4]) {
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (size_t i = b; i < e; ++i)
        for (int a = 0; a < 3; ++a) {
            lo[a] = std::min(lo[a], (double)v[i].c[a] - std::fabs((double)v[i].r));
            hi[a] = std::max(hi[a], (double)v[i].c[a] + std::fabs((double)v[i].r));
        }
    float C[3];
    for (int a = 0; a < 3; ++a) C[a] = (float)(0.5 * (lo[a] + hi[a]));
    double R = 0.0;
    for (size_t i = b; i < e; ++i) {
        double d2 = 0.0;
        for (int a = 0; a < 3; ++a) {
            double d = (double)v[i].c[a] - (double)C[a];
            d2 += d * d;
        }
        R = std::max(R, std::sqrt(d2) + std::fabs((double)v[i].r));
    }
    const double Rp = R * (1.0 + CULL_K0) * (1.0 + 1e-6);
    out[0] = C[0]; out[1] = C[1]; out[2] = C[2];
    out[3] = round_up(Rp * Rp * (1.0 + CULL_ALPHA));
    if (!(out[3] == out[3]) || !(R == R)) out[3] = INFINITY;  // never cull what cannot be bounded
}

// Median splits on the longest axis of the centres.  Records group entries by absolute index
// (8i .. 8i+7, 64i .. 64i+63, ...), so the split point is the multiple of the largest group
// size that lies inside (b, e), nearest to the middle: no group straddles a split.
void order(std::vector<LeafSphere>& v, size_t b, size_t e) {
    size_t m = 0;
    for (size_t g : {(size_t)4096, (size_t)512, (size_t)64, (size_t)8}) {
        const size_t lo = (b / g + 1) * g;  // first multiple of g above b
        if (lo >= e) continue;
        const size_t mid = (b + e) / 2;
        m = std::max(lo, (mid + g / 2) / g * g);
        while (m >= e) m -= g;
        break;
    }
    if (m <= b || m >= e) return;
    float lo[3], hi[3];
    for (int a = 0; a < 3; ++a) lo[a] = hi[a] = v[b].c[a];
    for (size_t i = b; i < e; ++i)
        for (int a = 0; a < 3; ++a) {
            lo[a] = std::min(lo[a], v[i].c[a]);
            hi[a] = std::max(hi[a], v[i].c[a]);
        }
    int ax = 0;
    for (int a = 1; a < 3; ++a)
        if (hi[a] - lo[a] > hi[ax] - lo[ax]) ax = a;
    auto key = [ax](const LeafSphere& s) { return s.c[ax] == s.c[ax] ? s.c[ax] : INFINITY; };  // total order
    std::nth_element(v.begin() + b, v.begin() + m, v.begin() + e, [&key](const LeafSphere& x, const LeafSphere& y) {
        const float kx = key(x), ky = key(y);
        if (kx != ky) return kx < ky;
        return x.pos < y.pos;  // deterministic
    });
    order(v, b, m);
    order(v, m, e);
}

// Spheres of very different size make poor neighbours: one big sphere inflates the bound of
// every group it sits in.  The run is therefore cut into segments of similar radius (largest
// first, a new segment when the radius has halved), each padded to a multiple of 8 entries
// with copies of its last sphere and ordered spatially on its own.
void arrange(std::vector<LeafSphere>& v) {
    static const float CLASS_RATIO = getenv("RT_CLASS_RATIO") ? (float)atof(getenv("RT_CLASS_RATIO")) : 0.0f;
    auto size_of = [](const LeafSphere& s) { const float r = std::fabs(s.r); return r == r ? r : INFINITY; };
    std::stable_sort(v.begin(), v.end(), [&](const LeafSphere& x, const LeafSphere& y) { return size_of(x) > size_of(y); });
    std::vector<LeafSphere> out;
    out.reserve(v.size() + 64);
    size_t i = 0;
    while (i < v.size()) {
        const size_t b = out.size();
        const float r0 = size_of(v[i]);
        size_t j = i;
        while (j < v.size() && (j - i < 8 || size_of(v[j]) >= CLASS_RATIO * r0)) ++j;
        out.insert(out.end(), v.begin() + i, v.begin() + j);
        while (out.size() % 8) {  // a copy never wins: same distance as its original, later position
            out.push_back(out.back());
            out.back().pos |= 0x80000000u;
        }
        order(out, b, out.size());
        i = j;
    }
    v.swap(out);
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
