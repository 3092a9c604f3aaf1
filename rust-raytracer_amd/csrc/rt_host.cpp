// Host-only pieces of librtamd.so that need no GPU: scene description copies and
// validation, the camera basis, the interleaved-tile index math and the exported
// host tree build.
#include "rt_host.h"

#include <cmath>
#include <cstring>

namespace rtamd {

void SceneData::refresh_desc() {
    desc.n_spheres = (uint32_t)spheres.size();
    desc.n_cones = (uint32_t)cones.size();
    desc.n_polygons = (uint32_t)polygons.size();
    desc.n_vertices = (uint32_t)vertices.size();
    desc.n_primitives = (uint32_t)primitives.size();
    desc.n_surfaces = (uint32_t)surfaces.size();
    desc.n_lights = (uint32_t)lights.size();
    desc.spheres = spheres.data();
    desc.cones = cones.data();
    desc.polygons = polygons.data();
    desc.vertices = vertices.data();
    desc.primitives = primitives.data();
    desc.surfaces = surfaces.data();
    desc.lights = lights.data();
    memcpy(desc.background, background, sizeof(background));
}

void SceneData::assign(const rt_scene_desc& d) {
    spheres.assign(d.spheres, d.spheres + d.n_spheres);
    cones.assign(d.cones, d.cones + d.n_cones);
    polygons.assign(d.polygons, d.polygons + d.n_polygons);
    vertices.assign(d.vertices, d.vertices + d.n_vertices);
    primitives.assign(d.primitives, d.primitives + d.n_primitives);
    surfaces.assign(d.surfaces, d.surfaces + d.n_surfaces);
    lights.assign(d.lights, d.lights + d.n_lights);
    memcpy(background, d.background, sizeof(background));
    desc = d;
    refresh_desc();
    desc.nodes = nullptr;
    desc.leaf_refs = nullptr;
    desc.n_nodes = desc.n_leaf_refs = 0;
}

int validate_desc(const rt_scene_desc& d, std::string& err) {
    auto need = [&](const void* p, uint32_t n, const char* what) {
        if (n && !p) { err = std::string("null ") + what + " array with nonzero count"; return false; }
        return true;
    };
    if (!need(d.spheres, d.n_spheres, "spheres") || !need(d.cones, d.n_cones, "cones") ||
        !need(d.polygons, d.n_polygons, "polygons") || !need(d.vertices, d.n_vertices, "vertices") ||
        !need(d.primitives, d.n_primitives, "primitives") || !need(d.surfaces, d.n_surfaces, "surfaces") ||
        !need(d.lights, d.n_lights, "lights"))
        return RT_ERR_INVALID;
    for (uint32_t i = 0; i < d.n_polygons; ++i) {
        const rt_polygon& p = d.polygons[i];
        if (p.vertex_count < 3) { err = "polygon with fewer than 3 vertices"; return RT_ERR_INVALID; }
        if ((uint64_t)p.first_vertex + p.vertex_count > d.n_vertices) { err = "polygon vertex range out of bounds"; return RT_ERR_INVALID; }
    }
    for (uint32_t i = 0; i < d.n_primitives; ++i) {
        const rt_primitive& p = d.primitives[i];
        uint32_t lim = p.kind == RT_KIND_SPHERE ? d.n_spheres : p.kind == RT_KIND_CONE ? d.n_cones
                     : p.kind == RT_KIND_POLYGON ? d.n_polygons : 0;
        if (p.kind > RT_KIND_POLYGON || p.index >= lim) { err = "primitive kind/index out of range"; return RT_ERR_INVALID; }
        if (p.surface >= d.n_surfaces) { err = "primitive surface index out of range"; return RT_ERR_INVALID; }
    }
    for (uint32_t i = 0; i < d.n_surfaces; ++i)
        if (d.surfaces[i].model > RT_MODEL_BLINN_PHONG) { err = "unknown shading model"; return RT_ERR_INVALID; }
    if (d.nodes) {
        if (d.n_nodes == 0) { err = "prebuilt partition with zero nodes"; return RT_ERR_INVALID; }
        if (d.n_leaf_refs && !d.leaf_refs) { err = "null leaf_refs"; return RT_ERR_INVALID; }
        std::vector<uint8_t> has_parent(d.n_nodes, 0);
        for (uint32_t i = 0; i < d.n_nodes; ++i) {
            const rt_kd_node& n = d.nodes[i];
            if (n.axis > 3) { err = "node axis out of range"; return RT_ERR_INVALID; }
            if (n.axis == 3) {
                if ((uint64_t)n.under + n.over > d.n_leaf_refs) { err = "leaf reference range out of bounds"; return RT_ERR_INVALID; }
            } else if (n.under >= d.n_nodes || n.over >= d.n_nodes || n.under <= i || n.over <= i) {
                // children must come after their parent: this also rules out cycles
                err = "interior node child index out of range";
                return RT_ERR_INVALID;
            } else {
                // a tree: the device's traversal stack is sized by the depth, which is only well
                // defined when every node has one parent
                if (n.under == n.over || has_parent[n.under] || has_parent[n.over]) {
                    err = "node with more than one parent";
                    return RT_ERR_INVALID;
                }
                has_parent[n.under] = has_parent[n.over] = 1;
            }
        }
        for (uint32_t i = 0; i < d.n_leaf_refs; ++i)
            if (d.leaf_refs[i] >= d.n_primitives) { err = "leaf reference out of range"; return RT_ERR_INVALID; }
    }
    return RT_OK;
}

// reference src/render.rs:81-92
Camera camera_basis(const rt_view& v) {
    const float PI_F = 3.14159265358979323846f;
    float up_len = tanf(PI_F * v.angle / 360.0f);
    float right_len = up_len * ((float)v.width / (float)v.height);
    auto norm = [](float a[3]) {
        float m = sqrtf(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
        a[0] /= m; a[1] /= m; a[2] /= m;
    };
    auto cross = [](const float a[3], const float b[3], float o[3]) {
        o[0] = a[1] * b[2] - a[2] * b[1];
        o[1] = a[2] * b[0] - a[0] * b[2];
        o[2] = a[0] * b[1] - a[1] * b[0];
    };
    Camera c;
    for (int i = 0; i < 3; ++i) c.forward[i] = v.at[i] - v.from[i];
    norm(c.forward);
    cross(c.forward, v.up, c.right);
    norm(c.right);
    for (int i = 0; i < 3; ++i) c.right[i] = c.right[i] * right_len;
    cross(c.right, c.forward, c.up);
    norm(c.up);
    for (int i = 0; i < 3; ++i) c.up[i] = c.up[i] * up_len;
    return c;
}

}  // namespace rtamd

using namespace rtamd;

extern "C" int rt_debug_bound(uint32_t kind, const rt_cone* cone, const rt_vertex* verts, uint32_t n_verts, float* out4) {
    if (!out4) return 0;
    if (kind == RT_KIND_CONE && cone) return cylinder_bound(*cone, out4, out4 + 3) ? 1 : 0;
    if (kind == RT_KIND_POLYGON && verts && n_verts >= 3) return polygon_bound(verts, 0, n_verts, out4, out4 + 3) ? 1 : 0;
    return 0;
}

extern "C" int rt_debug_polygon_box(const rt_vertex* verts, uint32_t n_verts, float* out6) {
    float c[3], r;
    if (!out6 || !verts || n_verts < 3) return 0;
    return polygon_bound(verts, 0, n_verts, c, &r, out6) ? 1 : 0;
}

extern "C" void rt_camera_basis(const rt_view* view, float* out9) {
    Camera c = camera_basis(*view);
    memcpy(out9, c.forward, 12);
    memcpy(out9 + 3, c.right, 12);
    memcpy(out9 + 6, c.up, 12);
}

// ---- interleaved tiles: tile t (row-major over the 8x8 tile grid) belongs to rank
// t % count and is that rank's local tile t / count; a tile is 64 consecutive slots,
// slot = ly * 8 + lx.  Edge tiles keep all 64 slots (those outside the image unused).
extern "C" uint64_t rt_tile_pixels(uint32_t width, uint32_t height, uint32_t rank, uint32_t count) {
    TileGrid g(width, height, count);
    return rank < g.count ? g.tiles_of(rank) * 64 : 0;
}

extern "C" uint64_t rt_tile_pixels_max(uint32_t width, uint32_t height, uint32_t count) {
    TileGrid g(width, height, count);
    return g.tiles_of(0) * 64;
}

extern "C" uint64_t rt_tile_slot(uint32_t width, uint32_t height, uint32_t count, uint32_t x, uint32_t y,
                                 uint32_t* rank_out) {
    TileGrid g(width, height, count);
    uint64_t t = (uint64_t)(y / 8) * g.tiles_x + (x / 8);
    if (rank_out) *rank_out = (uint32_t)(t % g.count);
    return (t / g.count) * 64 + (y % 8) * 8 + (x % 8);
}

extern "C" int rt_deinterleave_host(uint32_t width, uint32_t height, uint32_t count, uint64_t stride_pixels,
                                    uint32_t elem_size, const void* gathered, void* image) {
    if (!gathered || !image || (elem_size != 1 && elem_size != 4)) return RT_ERR_INVALID;
    TileGrid g(width, height, count);
    if (stride_pixels < g.tiles_of(0) * 64) return RT_ERR_INVALID;
    const size_t px = (size_t)elem_size * 3;
    const uint8_t* src = (const uint8_t*)gathered;
    uint8_t* dst = (uint8_t*)image;
    for (uint32_t y = 0; y < height; ++y)
        for (uint32_t x = 0; x < width; ++x) {
            uint32_t rank;
            uint64_t slot = rt_tile_slot(width, height, count, x, y, &rank);
            memcpy(dst + ((size_t)y * width + x) * px, src + ((size_t)rank * stride_pixels + slot) * px, px);
        }
    return RT_OK;
}

extern "C" int rt_tree_build(const rt_scene_desc* desc, rt_tree** out, char* errbuf, size_t errlen) {
    auto fail = [&](int code, const std::string& msg) {
        if (errbuf && errlen) {
            size_t n = msg.size() < errlen - 1 ? msg.size() : errlen - 1;
            memcpy(errbuf, msg.data(), n);
            errbuf[n] = 0;
        }
        return code;
    };
    if (!desc || !out) return fail(RT_ERR_INVALID, "null argument");
    *out = nullptr;
    std::string err;
    rt_scene_desc d = *desc;
    d.nodes = nullptr;
    int rc = validate_desc(d, err);
    if (rc != RT_OK) return fail(rc, err);
    rt_tree* t = new rt_tree();
    rc = build_tree(d, t->tree, err);
    if (rc != RT_OK) {
        delete t;
        return fail(rc, err);
    }
    *out = t;
    return RT_OK;
}

extern "C" int rt_tree_get(const rt_tree* t, rt_tree_info* info, const rt_kd_node** nodes, const uint32_t** leaf_refs) {
    if (!t) return RT_ERR_INVALID;
    if (info) *info = t->tree.info;
    if (nodes) *nodes = t->tree.nodes.data();
    if (leaf_refs) *leaf_refs = t->tree.refs.data();
    return RT_OK;
}

extern "C" void rt_tree_free(rt_tree* t) { delete t; }
