// C++ host API over librtamd.so with the reference's shapes of things: View, Scene
// (add_primitive / add_light / background), Sphere, Cone, Polygon, Phong, BlinnPhong,
// RenderTarget and render(view, scene, target) -- reference src/render.rs:8-114,
// src/scene.rs:17-48, src/sphere.rs:6-9, src/cone.rs:7-12, src/polygon.rs:6-8,
// src/phong.rs:6-14, src/blinn_phong.rs:6-14 -- so a caller written against the reference
// switches to the GPU path by changing the include.  render() flattens the scene into the
// plain arrays of rt_scene_desc and calls rt_scene_create + rt_render.
#pragma once
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/rtamd.h"

namespace rtamd_host {

struct Point { float x, y, z; };
struct Vector { float dx, dy, dz; };
struct Color { float r, g, b; };
struct PointNormal { Point point; Vector normal; };

struct View {  // src/render.rs:8-23
    Point from{0, 0, 0}, at{0, 0, 0};
    Vector up{0, 0, 0};
    float angle = 0, hither = 0;
    uint32_t width = 0, height = 0;
};

struct Sphere { Point center; float radius; };
struct Cone { Point base; Point apex; float base_radius; float apex_radius; };
struct Polygon { std::vector<PointNormal> vertices; };

// The two surface types share one field list (src/phong.rs:6-14, src/blinn_phong.rs:6-14).
struct Surface {
    Color color{1, 1, 1};
    float diffuse_component = 1, specular_component = 0, shine = 1;
    float reflectance = 0, transmittance = 0, refraction_index = 1;
    uint32_t model = RT_MODEL_PHONG;
};
inline Surface Phong(Color c, float kd, float ks, float shine, float refl, float t, float ior) {
    return Surface{c, kd, ks, shine, refl, t, ior, RT_MODEL_PHONG};
}
inline Surface BlinnPhong(Color c, float kd, float ks, float shine, float refl, float t, float ior) {
    return Surface{c, kd, ks, shine, refl, t, ior, RT_MODEL_BLINN_PHONG};
}

struct Light { Point position; Color color; };  // src/scene.rs:11-15

class Scene {  // src/scene.rs:17-48
public:
    Color background{1, 1, 1};
    // Surfaces are shared between primitives by index (the reference shares an Rc).
    uint32_t add_surface(const Surface& s) {
        rt_surface r{{s.color.r, s.color.g, s.color.b}, s.diffuse_component, s.specular_component, s.shine,
                     s.reflectance, s.transmittance, s.refraction_index, s.model};
        surfaces_.push_back(r);
        return (uint32_t)surfaces_.size() - 1;
    }
    void add_primitive(const Sphere& s, uint32_t surface) {
        spheres_.push_back(rt_sphere{{s.center.x, s.center.y, s.center.z}, s.radius});
        prims_.push_back(rt_primitive{RT_KIND_SPHERE, (uint32_t)spheres_.size() - 1, surface});
    }
    void add_primitive(const Cone& c, uint32_t surface) {
        cones_.push_back(rt_cone{{c.base.x, c.base.y, c.base.z}, c.base_radius, {c.apex.x, c.apex.y, c.apex.z}, c.apex_radius});
        prims_.push_back(rt_primitive{RT_KIND_CONE, (uint32_t)cones_.size() - 1, surface});
    }
    void add_primitive(const Polygon& p, uint32_t surface) {
        if (p.vertices.size() < 3) throw std::invalid_argument("polygon needs at least 3 vertices");
        polygons_.push_back(rt_polygon{(uint32_t)vertices_.size(), (uint32_t)p.vertices.size()});
        for (const PointNormal& v : p.vertices)
            vertices_.push_back(rt_vertex{{v.point.x, v.point.y, v.point.z}, {v.normal.dx, v.normal.dy, v.normal.dz}});
        prims_.push_back(rt_primitive{RT_KIND_POLYGON, (uint32_t)polygons_.size() - 1, surface});
    }
    void add_light(const Light& l) {
        lights_.push_back(rt_light{{l.position.x, l.position.y, l.position.z}, {l.color.r, l.color.g, l.color.b}});
    }
    rt_scene_desc desc() const {
        rt_scene_desc d{};
        d.n_spheres = (uint32_t)spheres_.size(); d.spheres = spheres_.data();
        d.n_cones = (uint32_t)cones_.size(); d.cones = cones_.data();
        d.n_polygons = (uint32_t)polygons_.size(); d.polygons = polygons_.data();
        d.n_vertices = (uint32_t)vertices_.size(); d.vertices = vertices_.data();
        d.n_primitives = (uint32_t)prims_.size(); d.primitives = prims_.data();
        d.n_surfaces = (uint32_t)surfaces_.size(); d.surfaces = surfaces_.data();
        d.n_lights = (uint32_t)lights_.size(); d.lights = lights_.data();
        d.background[0] = background.r; d.background[1] = background.g; d.background[2] = background.b;
        return d;
    }

private:
    std::vector<rt_sphere> spheres_;
    std::vector<rt_cone> cones_;
    std::vector<rt_polygon> polygons_;
    std::vector<rt_vertex> vertices_;
    std::vector<rt_primitive> prims_;
    std::vector<rt_surface> surfaces_;
    std::vector<rt_light> lights_;
};

class RenderTarget {  // src/render.rs:50-70
public:
    size_t width, height;
    RenderTarget(size_t w, size_t h) : width(w), height(h), values_(w * h * 3, 0.0f), bytes_(w * h * 3, 0) {}
    Color get(size_t x, size_t y) const {
        const float* p = &values_[(y * width + x) * 3];
        return Color{p[0], p[1], p[2]};
    }
    float* f32() { return values_.data(); }
    uint8_t* u8() { return bytes_.data(); }
    const uint8_t* u8() const { return bytes_.data(); }

private:
    std::vector<float> values_;
    std::vector<uint8_t> bytes_;  // quantised as src/ppm.rs:20-22
};

struct RenderSettings {
    uint32_t max_depth = 5;           // src/scene.rs:8
    float min_contribution = 0.003f;  // src/scene.rs:9
    uint32_t fp_mode = RT_FP_EXACT;
    uint32_t build_threads = 1;
    int device = 0;
};

// render::render (src/render.rs:72): builds the space partition, traces, fills the target.
inline rt_stats render(const View& view, const Scene& scene, RenderTarget& target, const RenderSettings& rs = {}) {
    rt_context* ctx = nullptr;
    int dev = rs.device;
    auto check = [&](int rc) {
        if (rc != RT_OK) {
            std::string msg = ctx ? rt_last_error(ctx) : "rt_context_create failed";
            if (ctx) rt_context_destroy(ctx);
            throw std::runtime_error("librtamd: " + msg);
        }
    };
    check(rt_context_create(1, &dev, &ctx));
    rt_scene_desc d = scene.desc();
    d.build_threads = rs.build_threads;
    rt_scene* sc = nullptr;
    check(rt_scene_create(ctx, &d, &sc));
    rt_view v{{view.from.x, view.from.y, view.from.z}, {view.at.x, view.at.y, view.at.z},
              {view.up.dx, view.up.dy, view.up.dz}, view.angle, view.hither, view.width, view.height};
    rt_options o{rs.max_depth, rs.min_contribution, rs.fp_mode, 0, 1, 0, 0};
    rt_stats st{};
    int rc = rt_render(ctx, sc, &v, &o, target.f32(), target.u8(), &st);
    if (rc != RT_OK) rt_scene_destroy(sc);
    check(rc);
    rt_scene_destroy(sc);
    rt_context_destroy(ctx);
    return st;
}

}  // namespace rtamd_host
