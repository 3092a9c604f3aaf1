// Host-side exact construction of the space partition, emitted directly as the flat
// 64-byte node array + leaf reference array the GPU walks.
//
// It reproduces SpacePartition::from_primitives (reference src/space_partition.rs:197-258)
// decision for decision: candidate planes are every primitive box's min then max on
// the current axis in list order (:55-75), a candidate's cost is
// SA(under box) * n_under + SA(over box) * n_over with "under" = min < plane and
// "over" = max >= plane (:100-149), the first strictly cheapest candidate wins, the
// actual split then uses max > plane / min <= plane (:151-172), children inherit the
// appraisal's boxes, and the axis cycles X->Y->Z (:174-180).  Boxes are kept in the
// reference's (corner, extent) form so that every union rounds as BoundingBox::
// expand_to_fit does (src/shape.rs:59-82).
//
// Unlike the reference this is iterative (explicit work stack), evaluates the
// candidates of one node on several threads or, for big nodes when a scene is created,
// on the GPU (each candidate is independent; the arg-min scan is done in candidate order
// afterwards), keeps only the cost per candidate and re-derives the winner's boxes, and
// refuses scenes on which the reference's recursion never ends.
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "rt_host.h"

namespace rtamd {

namespace {

struct Box {
    float c[3];  // min corner
    float e[3];  // extent
};

inline float box_max(const Box& b, int a) { return b.c[a] + b.e[a]; }  // src/shape.rs:37-39

inline float surface_area(const Box& b) {  // src/shape.rs:45-49
    return b.e[0] * b.e[1] * 2.0f + b.e[1] * b.e[2] * 2.0f + b.e[0] * b.e[2] * 2.0f;
}

inline void expand(Box& acc, const Box& o) {  // src/shape.rs:59-82
    for (int a = 0; a < 3; ++a) {
        float mn = fminf(acc.c[a], o.c[a]);
        float sf = acc.c[a] + acc.e[a];
        float of = o.c[a] + o.e[a];
        float mx = fmaxf(sf, of);
        acc.c[a] = mn;
        acc.e[a] = mx - mn;
    }
}

inline float vmag(const float v[3]) { return sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); }

Box sphere_box(const rt_sphere& s) {  // src/sphere.rs:12-25
    Box b;
    for (int a = 0; a < 3; ++a) {
        b.c[a] = s.center[a] - s.radius;
        b.e[a] = s.radius * 2.0f;
    }
    return b;
}

Box cone_box(const rt_cone& k) {  // src/cone.rs:15-95
    float d[3] = {k.apex[0] - k.base[0], k.apex[1] - k.base[1], k.apex[2] - k.base[2]};
    float m = vmag(d);
    float n[3] = {d[0] / m, d[1] / m, d[2] / m};
    Box b;
    for (int a = 0; a < 3; ++a) {
        // projection of the axis unit vector onto the base plane: e_a - n * n[a]
        float p[3];
        for (int j = 0; j < 3; ++j) p[j] = (j == a ? 1.0f : 0.0f) - n[j] * n[a];
        float pm = vmag(p);
        float scale = pm > 0.0f ? p[a] / pm : 0.0f;
        float base_extra = k.base_radius * scale;
        float apex_extra = k.apex_radius * scale;
        float mn = fminf(k.base[a] - base_extra, k.apex[a] - apex_extra);
        float mx = fmaxf(k.base[a] + base_extra, k.apex[a] + apex_extra);
        b.c[a] = mn;
        b.e[a] = mx - mn;
    }
    return b;
}

Box polygon_box(const rt_vertex* v, uint32_t n) {  // src/polygon.rs:11-56
    float mn[3], mx[3];
    for (int a = 0; a < 3; ++a) mn[a] = mx[a] = v[0].point[a];
    for (uint32_t i = 1; i < n; ++i)
        for (int a = 0; a < 3; ++a) {
            float x = v[i].point[a];
            if (x < mn[a]) mn[a] = x;
            if (x > mx[a]) mx[a] = x;
        }
    Box b;
    for (int a = 0; a < 3; ++a) {
        b.c[a] = mn[a];
        b.e[a] = mx[a] - mn[a];
    }
    return b;
}

// What from_boxed_primitives (src/space_partition.rs:223) is called with, less the list.
struct CallState {
    uint32_t axis;
    Box box;
};

struct Work {
    uint32_t node;
    uint32_t axis;
    uint32_t depth;
    Box box;
    std::vector<uint32_t> list;  // primitive ids, builder list order
    // Ancestors' calls that had this very list (consecutive ones up the path: a child's list is a
    // sub-sequence of its parent's, so equal length means equal list).
    std::vector<CallState> same_list;
};

inline bool same_call(const CallState& a, uint32_t axis, const Box& box) {
    return a.axis == axis && memcmp(&a.box, &box, sizeof(Box)) == 0;
}

// Cost of splitting `list` at `plane` (src/space_partition.rs:100-149); optionally
// returns the two boxes.
float appraise_split(const std::vector<uint32_t>& list, const std::vector<Box>& boxes, int axis, float plane,
                     bool* have_under, bool* have_over, Box* under_box, Box* over_box) {
    size_t nu = 0, no = 0;
    Box ub{}, ob{};
    for (uint32_t p : list) {
        const Box& b = boxes[p];
        if (b.c[axis] < plane) {
            if (nu == 0) ub = b; else expand(ub, b);
            ++nu;
        }
        if (box_max(b, axis) >= plane) {
            if (no == 0) ob = b; else expand(ob, b);
            ++no;
        }
    }
    if (have_under) *have_under = nu > 0;
    if (have_over) *have_over = no > 0;
    if (under_box) *under_box = ub;
    if (over_box) *over_box = ob;
    // appraise(), :184-193: surface area * count
    if (nu > 0 && no > 0) return surface_area(ub) * (float)nu + surface_area(ob) * (float)no;
    if (nu > 0) return surface_area(ub) * (float)nu;
    if (no > 0) return surface_area(ob) * (float)no;
    return 0.0f;
}

}  // namespace

int prim_box(const rt_scene_desc& d, uint32_t prim, float corner[3], float extent[3]) {
    const rt_primitive& p = d.primitives[prim];
    Box b;
    if (p.kind == RT_KIND_SPHERE) b = sphere_box(d.spheres[p.index]);
    else if (p.kind == RT_KIND_CONE) b = cone_box(d.cones[p.index]);
    else b = polygon_box(d.vertices + d.polygons[p.index].first_vertex, d.polygons[p.index].vertex_count);
    memcpy(corner, b.c, sizeof(b.c));
    memcpy(extent, b.e, sizeof(b.e));
    return 0;
}

int build_tree(const rt_scene_desc& d, HostTree& out, std::string& err, const SplitCostFn* wide_cost, uint32_t wide_min) {
    auto t0 = std::chrono::steady_clock::now();
    out.nodes.clear();
    out.refs.clear();
    out.info = rt_tree_info{};
    const int threads = d.build_threads < 1 ? 1 : (int)d.build_threads;

    auto make_node = [&](const Box& b, uint32_t parent, uint32_t depth) {
        rt_kd_node n;
        memset(&n, 0, sizeof(n));
        for (int a = 0; a < 3; ++a) {
            n.bmin[a] = b.c[a];
            n.bmax[a] = b.c[a] + b.e[a];
            n.extent[a] = b.e[a];
        }
        n.axis = 3;
        n.parent = parent;
        n.depth = depth;
        out.nodes.push_back(n);
        if (depth > out.info.max_depth) out.info.max_depth = depth;
        return (uint32_t)out.nodes.size() - 1;
    };

    if (d.n_primitives == 0) {  // :200-204
        Box z{};
        make_node(z, 0xffffffffu, 0);
        out.info.n_nodes = out.info.n_leaves = 1;
        return RT_OK;
    }

    std::vector<Box> boxes(d.n_primitives);
    for (uint32_t i = 0; i < d.n_primitives; ++i) prim_box(d, i, boxes[i].c, boxes[i].e);
    Box total = boxes[0];  // :206-216
    for (uint32_t i = 1; i < d.n_primitives; ++i) expand(total, boxes[i]);

    std::vector<Work> stack;
    {
        Work w;
        w.node = make_node(total, 0xffffffffu, 0);
        w.axis = 0;
        w.depth = 0;
        w.box = total;
        w.list.resize(d.n_primitives);
        for (uint32_t i = 0; i < d.n_primitives; ++i) w.list[i] = i;
        stack.push_back(std::move(w));
    }

    std::vector<float> cost;
    while (!stack.empty()) {
        Work w = std::move(stack.back());
        stack.pop_back();
        const size_t n = w.list.size();
        const int axis = (int)w.axis;
        const float no_split_cost = surface_area(w.box) * (float)n;  // :228

        bool split = false;
        float plane = 0.0f;
        Box under_box{}, over_box{};
        if (n >= 4) {  // :44
            const size_t ncand = 2 * n;
            cost.assign(ncand, 0.0f);
            auto plane_of = [&](size_t c) {
                const Box& b = boxes[w.list[c >> 1]];
                return (c & 1) ? box_max(b, axis) : b.c[axis];
            };
            auto eval_range = [&](size_t c0, size_t c1) {
                for (size_t c = c0; c < c1; ++c)
                    cost[c] = appraise_split(w.list, boxes, axis, plane_of(c), nullptr, nullptr, nullptr, nullptr);
            };
            static_assert(sizeof(Box) == 6 * sizeof(float), "boxes are handed over as 6 floats");
            if (wide_cost && n >= wide_min &&
                (*wide_cost)(&boxes[0].c[0], d.n_primitives, w.list.data(), (uint32_t)n, axis, cost.data())) {
                // (evaluated on the GPU: rt_kernels.hip k_appraise, the same arithmetic)
            } else if (threads > 1 && n >= 256) {
                std::atomic<size_t> next{0};
                std::vector<std::thread> pool;
                for (int t = 0; t < threads; ++t)
                    pool.emplace_back([&] {
                        for (;;) {
                            size_t c0 = next.fetch_add(32);
                            if (c0 >= ncand) break;
                            eval_range(c0, c0 + 32 < ncand ? c0 + 32 : ncand);
                        }
                    });
                for (auto& th : pool) th.join();
            } else {
                eval_range(0, ncand);
            }
            // first strictly cheapest candidate (:59, :69)
            float min_cost = no_split_cost;
            size_t best = ncand;
            for (size_t c = 0; c < ncand; ++c)
                if (cost[c] < min_cost) { min_cost = cost[c]; best = c; }
            if (best < ncand) {
                bool hu, ho;
                plane = plane_of(best);
                appraise_split(w.list, boxes, axis, plane, &hu, &ho, &under_box, &over_box);
                split = hu && ho && min_cost < no_split_cost;  // :79-96
            }
        }

        rt_kd_node& node = out.nodes[w.node];
        if (!split) {  // leaf, :231-238
            node.axis = 3;
            node.under = (uint32_t)out.refs.size();
            node.over = (uint32_t)n;
            out.refs.insert(out.refs.end(), w.list.begin(), w.list.end());
            out.info.n_leaves++;
            if (n > out.info.max_leaf) out.info.max_leaf = (uint32_t)n;
            continue;
        }

        Work wu, wo;
        for (uint32_t p : w.list) {  // split(), :160-169
            const Box& b = boxes[p];
            if (box_max(b, axis) > plane) wo.list.push_back(p);
            if (b.c[axis] <= plane) wu.list.push_back(p);
        }
        const uint32_t next_axis = (w.axis + 1) % 3;
        // The reference recursion has no depth bound, and from_boxed_primitives is a function of its
        // arguments alone: a call whose (list, axis, box) equals an ancestor's repeats that
        // ancestor's decisions, reaches the same call again and never returns (the reference
        // overflows its stack).  That is decided here exactly: a child whose list is as long as the
        // parent's has the parent's list, and is compared with the chain of calls that had it.  The
        // test is also complete -- with the list unchanged the plane is the same for an axis whatever
        // the box, so a path either shortens its list or revisits a call within seven levels.
        for (Work* c : {&wu, &wo}) {
            const Box& cb = c == &wu ? under_box : over_box;
            if (c->list.size() != n) continue;
            c->same_list = w.same_list;
            c->same_list.push_back(CallState{w.axis, w.box});
            for (const CallState& a : c->same_list)
                if (same_call(a, next_axis, cb)) {
                    err = "space partition construction does not terminate on this scene: at depth " +
                          std::to_string(w.depth + 1) + " the reference's split() (src/space_partition.rs:163-166) hands a "
                          "list of " + std::to_string(n) + " primitives, axis and box to from_boxed_primitives that an "
                          "enclosing call already had (it overflows its stack on such input)";
                    return RT_ERR_BUILD;
                }
        }
        const uint32_t me = w.node;
        const uint32_t iu = make_node(under_box, me, w.depth + 1);
        const uint32_t io = make_node(over_box, me, w.depth + 1);
        rt_kd_node& nd = out.nodes[me];  // (make_node may have reallocated)
        nd.axis = w.axis;
        nd.plane = plane;
        nd.under = iu;
        nd.over = io;
        wu.node = iu; wu.axis = next_axis; wu.depth = w.depth + 1; wu.box = under_box;
        wo.node = io; wo.axis = next_axis; wo.depth = w.depth + 1; wo.box = over_box;
        stack.push_back(std::move(wo));
        stack.push_back(std::move(wu));
    }

    out.info.n_nodes = (uint32_t)out.nodes.size();
    out.info.n_leaf_refs = (uint32_t)out.refs.size();
    out.info.build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return RT_OK;
}

}  // namespace rtamd
