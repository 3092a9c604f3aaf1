// Internal host-side declarations shared by the translation units of librtamd.so.
#pragma once
#include <cmath>
#include <cstdint>
#include <functional>
#include <string>
#include <vector>

#include "../../include/rtamd.h"

namespace rtamd {

struct HostTree {
    std::vector<rt_kd_node> nodes;
    std::vector<uint32_t> refs;
    rt_tree_info info{};
};

// kd_build.cpp
int prim_box(const rt_scene_desc& d, uint32_t prim, float corner[3], float extent[3]);
// Costs of all 2n candidate planes of one node (candidate 2k = min, 2k+1 = max of the k-th listed
// primitive's box on `axis`), evaluated somewhere faster than the host threads; boxes6 holds
// {corner, extent} per primitive of the scene.  Returns false if it cannot take the node.
using SplitCostFn = std::function<bool(const float* boxes6, uint32_t n_prims, const uint32_t* list, uint32_t n, int axis, float* cost)>;
int build_tree(const rt_scene_desc& d, HostTree& out, std::string& err, const SplitCostFn* wide_cost = nullptr,
               uint32_t wide_min = 0);

// leaf_cluster.cpp: bounding-sphere groups over one kind of primitive of one leaf
struct LeafSphere {      // a sphere, or the sphere that stands in for a polygon or a cylinder (*_bound)
    float c[3];
    float r;
    uint32_t prim, pos;  // primitive id, position in the leaf's list
    uint32_t a0, a1;     // polygons: first vertex, vertex count; cylinders: cone index, -
    // polygons: the box of polygon_bound (everything by default: an entry without one is never ruled out by it)
    float lo[3] = {-INFINITY, -INFINITY, -INFINITY}, hi[3] = {INFINITY, INFINITY, INFINITY};
};
// Re-orders `v` spatially, pads it to a multiple of 8 entries (padding = copies of entries with
// bit 31 of `pos` set), and appends the run's block of bounding records to `bnd` (4 floats each);
// returns the index of the block's header record.
// max_level (4 or 5): the highest level that may be the run's top level (the device walks five-level runs of
// polygons only).
uint32_t cluster_spheres(std::vector<LeafSphere>& v, std::vector<float>& bnd, int max_level = 4);
// The bounding record {C, q} of one entry alone (q = inf when its radius is not finite: never culled).
void sphere_record(const LeafSphere& e, float out[4]);
// Sphere {c, r} such that whenever Polygon::intersect (reference src/polygon.rs:58-293) accepts
// a ray, the ray's line passes within r + 1.1e-5 |src - c| of c.  False if no such bound exists
// (coordinates that are not finite).
// box6 (may be null): an axis-aligned box {lo xyz, hi xyz} about c such that the accepted point src + ray t lies inside
// it widened by 1.1e-5 |src - c| on every side.
bool polygon_bound(const rt_vertex* vertices, uint32_t first, uint32_t count, float c[3], float* r, float* box6 = nullptr);
// Box records beside the sphere records of the block cluster_spheres appended at header `h` for `v` (as it left
// them): `box` holds 8 floats {lo xyz, -, hi xyz, -} per record of `bnd`, same index; a record's box is the union of
// the boxes of the entries below it.  Records of other blocks keep (or get) the box that holds everything.
void cluster_boxes(const std::vector<LeafSphere>& v, const std::vector<float>& bnd, uint32_t h, std::vector<float>& box);
// The same for a cone with equal radii and a query with near >= 0 (Cone::intersect, reference
// src/cone.rs:97-221): the line passes within r + 1.011e-3 |src - c| of c.  False for any other
// cone (its quadratic's leading coefficient can cancel, and then no such bound holds).
bool cylinder_bound(const rt_cone& cone, float c[3], float* r);

// Validates counts, pointers and every index in a scene description.
int validate_desc(const rt_scene_desc& d, std::string& err);

// Owning copy of a scene description.
struct SceneData {
    std::vector<rt_sphere> spheres;
    std::vector<rt_cone> cones;
    std::vector<rt_polygon> polygons;
    std::vector<rt_vertex> vertices;
    std::vector<rt_primitive> primitives;
    std::vector<rt_surface> surfaces;
    std::vector<rt_light> lights;
    float background[3] = {1.0f, 1.0f, 1.0f};  // reference src/scene.rs:26
    rt_scene_desc desc{};
    void refresh_desc();  // points desc at the vectors
    void assign(const rt_scene_desc& d);
};

// Camera basis (reference src/render.rs:81-92)
struct Camera {
    float forward[3], right[3], up[3];
};
Camera camera_basis(const rt_view& v);

// tiles.cpp: interleaved 8x8 tiles
struct TileGrid {
    uint32_t width, height, tiles_x, tiles_y, count;
    uint64_t n_tiles;
    TileGrid(uint32_t w, uint32_t h, uint32_t c)
        : width(w), height(h), tiles_x((w + 7) / 8), tiles_y((h + 7) / 8), count(c ? c : 1),
          n_tiles((uint64_t)((w + 7) / 8) * ((h + 7) / 8)) {}
    uint64_t tiles_of(uint32_t rank) const { return rank < n_tiles % count ? n_tiles / count + 1 : n_tiles / count; }
};

}  // namespace rtamd

struct rt_parsed_scene {
    rtamd::SceneData data;
    rt_view view{};
    std::vector<std::string> warnings;
};

struct rt_tree {
    rtamd::HostTree tree;
};
