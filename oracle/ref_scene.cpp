// CPU ORACLE -- TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (see ref_math.h).
//
// Space partition build + traversal, integrator and camera.
#include "ref_scene.h"

#include <atomic>
#include <cstring>
#include <stdexcept>
#include <string>
#include <thread>

namespace ref {

// reference src/space_partition.rs:40-97
bool find_splitting_plane(const std::vector<BoxedPrim>& list, int axis, float no_split_cost,
                          SplitDecision& out, BuildStats& st, int threads) {
    if (list.size() < 4) return false;  // :44
    st.split_attempts++;

    // Candidate 2k is prim k's min corner on the axis, 2k+1 its max corner (:55-75).
    const size_t ncand = list.size() * 2;
    std::vector<SplitAppraisal> apps(ncand);
    std::vector<float> planes(ncand);
    auto eval = [&](size_t c) {
        const BBox& b = list[c >> 1].box;
        float plane = (c & 1) ? b.max_corner().comp(axis) : b.corner.comp(axis);
        planes[c] = plane;
        apps[c] = appraise_split(list, axis, plane);
    };
    if (threads > 1 && list.size() >= 256) {
        std::atomic<size_t> next{0};
        std::vector<std::thread> pool;
        for (int t = 0; t < threads; ++t)
            pool.emplace_back([&] {
                for (;;) {
                    size_t c0 = next.fetch_add(64);
                    if (c0 >= ncand) break;
                    size_t c1 = c0 + 64 < ncand ? c0 + 64 : ncand;
                    for (size_t c = c0; c < c1; ++c) eval(c);
                }
            });
        for (auto& th : pool) th.join();
    } else {
        for (size_t c = 0; c < ncand; ++c) eval(c);
    }

    float min_cost = no_split_cost;
    float best_plane = 0.0f;
    bool best_have_under = false, best_have_over = false;
    BBox best_under{}, best_over{};
    for (size_t c = 0; c < ncand; ++c) {
        if (apps[c].cost < min_cost) {  // strict: the first best candidate wins (:59, :69)
            best_have_under = apps[c].have_under;
            best_under = apps[c].under_box;
            best_have_over = apps[c].have_over;
            best_over = apps[c].over_box;
            best_plane = planes[c];
            min_cost = apps[c].cost;
        }
    }
    if (!best_have_over) return false;   // :79-82
    if (!best_have_under) return false;  // :83-86
    if (min_cost < no_split_cost) {      // :88-96
        out.under_box = best_under;
        out.over_box = best_over;
        out.plane = best_plane;
        return true;
    }
    return false;
}

// reference src/space_partition.rs:223-258 (+ split :151-172, advance :174-180)
std::unique_ptr<Partition> build_partition(const std::vector<BoxedPrim>& list, int axis, const BBox& bbox,
                                           BuildStats& st, uint64_t depth, int threads, const SameListCall* up) {
    // The reference recursion has no depth bound and does not terminate on some
    // inputs: when the chosen plane is the largest min corner on the axis, split()
    // (:163-166) puts every primitive in `under` again while appraise_split (:111)
    // boxed one fewer, and the same decision can repeat on all three axes forever
    // (the reference then overflows its stack).  Such scenes have no reference
    // output; report them instead of recursing.  Proof of non-termination, not a depth
    // guess: this function depends on (list, axis, bbox) alone, so a call that has the
    // arguments of an enclosing call repeats its decisions and meets itself again.  `up`
    // chains the enclosing calls whose list had this length -- a child's list is a
    // sub-sequence of its parent's, so those had this very list.
    for (const SameListCall* a = up; a; a = a->up)
        if (a->axis == axis && std::memcmp(&a->bbox, &bbox, sizeof(BBox)) == 0)
            throw std::runtime_error("space partition recursion does not terminate on this scene: a call at depth " +
                                     std::to_string(depth) + " has the list, axis and box of an enclosing call "
                                     "(the reference overflows its stack here)");
    std::unique_ptr<Partition> node(new Partition());
    node->bbox = bbox;
    st.nodes++;
    if (depth > st.max_depth) st.max_depth = depth;

    float no_split_cost = appraise(list.size(), bbox);
    SplitDecision d;
    if (!find_splitting_plane(list, axis, no_split_cost, d, st, threads)) {
        node->leaf = true;
        for (const BoxedPrim& bp : list) node->prims.push_back(bp.prim);
        st.leaves++;
        st.leaf_refs += list.size();
        if (list.size() > st.max_leaf) st.max_leaf = list.size();
        return node;
    }

    std::vector<BoxedPrim> over, under;
    for (const BoxedPrim& bp : list) {  // split(), :160-169 -- note > and <= here vs < and >= in appraise_split
        if (bp.box.max_corner().comp(axis) > d.plane) over.push_back(bp);
        if (bp.box.corner.comp(axis) <= d.plane) under.push_back(bp);
    }
    int next_axis = axis == AX ? AY : (axis == AY ? AZ : AX);

    node->leaf = false;
    node->axis = axis;
    node->plane = d.plane;
    const SameListCall me{axis, bbox, up};
    node->over = build_partition(over, next_axis, d.over_box, st, depth + 1, threads, over.size() == list.size() ? &me : nullptr);
    node->under = build_partition(under, next_axis, d.under_box, st, depth + 1, threads, under.size() == list.size() ? &me : nullptr);
    return node;
}

// reference src/space_partition.rs:197-221
void Scene::build_space_partition(int threads) {
    build_stats = BuildStats();
    if (primitives.empty()) {
        partition.reset(new Partition());
        partition->bbox = BBox{V3{0, 0, 0}, V3{0, 0, 0}};
        partition->leaf = true;
        build_stats.nodes = build_stats.leaves = 1;
        return;
    }
    std::vector<BoxedPrim> boxed;
    BBox first = primitives[0].shape.bounding_box();
    boxed.push_back(BoxedPrim{0, first});
    BBox total = first;
    for (uint32_t i = 1; i < primitives.size(); ++i) {
        BBox b = primitives[i].shape.bounding_box();
        total = total.expand_to_fit(b);
        boxed.push_back(BoxedPrim{i, b});
    }
    partition = build_partition(boxed, AX, total, build_stats, 0, threads, nullptr);
}

// reference src/space_partition.rs:268-289 (node), :292-346 (interior), :348-385 (leaf)
bool Scene::intersect(const Partition& node, const V3& src, const V3& ray, float near, uint32_t ignore,
                      HitRecord& out, Counters& ctr) const {
    ctr.node_visits++;
    if (!node.bbox.intersect(src, ray, near)) return false;

    if (node.leaf) {
        bool have = false;
        for (uint32_t p : node.prims) {
            if (p == ignore) continue;  // identity of the shape object (:361-366)
            Hit h;
            if (primitives[p].shape.intersect(src, ray, near, h, ctr)) {
                if (!have || h.dist < out.dist) {  // strict <: first wins ties (:370-381)
                    have = true;
                    out.normal = h.normal;
                    out.dist = h.dist;
                    out.prim = p;
                }
            }
        }
        return have;
    }

    ctr.interior_steps++;
    if (src.comp(node.axis) < node.plane) {  // :303
        HitRecord r;
        bool hit = intersect(*node.under, src, ray, near, ignore, r, ctr);
        bool check_over = true;
        if (hit) {
            V3 endpoint = add(src, mul(ray, r.dist));
            check_over = endpoint.comp(node.axis) > node.plane;
        }
        if (check_over) return intersect(*node.over, src, ray, near, ignore, out, ctr);
        out = r;
        return true;
    } else {
        HitRecord r;
        bool hit = intersect(*node.over, src, ray, near, ignore, r, ctr);
        bool check_under = true;
        if (hit) {
            V3 endpoint = add(src, mul(ray, r.dist));
            check_under = endpoint.comp(node.axis) < node.plane;
        }
        if (check_under) return intersect(*node.under, src, ray, near, ignore, out, ctr);
        out = r;
        return true;
    }
}

// reference src/scene.rs:82-200
Color Scene::sub_trace(const V3& src, const V3& ray, float near, uint32_t ignore, float contribution,
                       uint32_t depth, const TraceParams& tp, Counters& ctr) const {
    HitRecord hit;
    if (!intersect(*partition, src, ray, near, ignore, hit, ctr)) return background;  // :199, not clamped

    ctr.surface_hits++;
    const Surface& surface = surfaces[primitives[hit.prim].surface];
    const V3 normal = hit.normal;
    V3 pos = add(src, mul(ray, hit.dist));      // :95
    bool back_face = dot(normal, ray) > 0.0f;   // :96
    Color total{0.0f, 0.0f, 0.0f};

    if (!back_face) {
        for (const Light& light : lights) {  // :103-125
            ctr.light_evals++;
            V3 to_light = sub(light.position, pos);
            float ldist = magnitude(to_light);
            V3 ldir = divs(to_light, ldist);
            ctr.rays_shadow++;
            HitRecord blocker;
            bool blocked = false;
            if (intersect(*partition, pos, ldir, 0.0f, hit.prim, blocker, ctr)) blocked = blocker.dist <= ldist;
            if (!blocked) {
                ctr.shade_evals++;
                Color direct = surface.blinn ? blinn_visible(surface, normal, ray, ldir, light.color, tp.pow_mode)
                                             : phong_visible(surface, normal, ray, ldir, light.color, tp.pow_mode);
                acc(total, direct);
            }
        }
        if (depth < tp.max_depth) {  // :127-144
            float rc = contribution * surface.reflectance;
            if (rc > tp.min_contribution) {
                V3 rr = reflected(ray, normal);
                ctr.rays_reflect++;
                Color c = sub_trace(pos, rr, 0.0f, hit.prim, rc, depth + 1, tp, ctr);
                acc(total, scale(c, surface.reflectance));
            }
        }
    }

    if (depth < tp.max_depth) {  // :147-193
        float transmittance;
        if (back_face) transmittance = surface.transmittance > tp.min_contribution ? 1.0f : 0.0f;
        else transmittance = surface.transmittance;
        float tc = contribution * transmittance;
        if (tc > tp.min_contribution) {
            V3 rt = back_face ? refracted(ray, neg(normal), 1.0f / surface.ior) : refracted(ray, normal, surface.ior);
            ctr.rays_refract++;
            Color c = sub_trace(pos, rt, 0.0001f, NO_PRIM, tc, depth + 1, tp, ctr);
            acc(total, scale(c, transmittance));
        }
    }

    clamp_upper(total);  // :195
    return total;
}

// reference src/render.rs:81-92
Camera make_camera(const View& view, uint32_t width, uint32_t height) {
    const float PI_F = 3.14159265358979323846f;  // std::f32::consts::PI
    float up_len = tanf(PI_F * view.angle / 360.0f);
    float aspect = (float)width / (float)height;  // :26-28
    float right_len = up_len * aspect;
    Camera c;
    c.forward = normalized(sub(view.at, view.from));
    c.right = mul(normalized(cross(c.forward, view.up)), right_len);
    c.up = mul(normalized(cross(c.right, c.forward)), up_len);
    return c;
}

}  // namespace ref
