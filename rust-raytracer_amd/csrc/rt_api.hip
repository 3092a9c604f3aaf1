// C ABI of librtamd.so: contexts, scene upload, frame orchestration, diagnostics.
// Host code only; kernels live in rt_kernels.hip (built twice: rt_exact, rt_fast).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "rt_host.h"
#include "rt_launch.h"

// Polygon and cylinder runs of at least this many entries get a hierarchy of bounding records.
#ifndef RT_HIER_MIN
#define RT_HIER_MIN 12
#endif

using namespace rtamd;

// Workspace of rt_render_device_static, the entry point that allocates nothing: ray pools, shadow
// queues and counters of one frame live in this array instead of hipMalloc'd buffers.  Every GPU
// the library runs on holds its own copy, at its own address, until the process ends.  In host
// code the name is only a handle for hipGetSymbolAddress: what kernels, memsets and copies are
// given is the address looked up per device in scratch_address().
static __device__ __align__(256) unsigned char scratch[RT_STATIC_WORKSPACE_BYTES];

namespace {

template <class T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    hipError_t alloc(size_t count) {
        release();
        if (count == 0) count = 1;
        hipError_t e = hipMalloc((void**)&p, count * sizeof(T));
        if (e == hipSuccess) n = count;
        else p = nullptr;
        return e;
    }
    hipError_t upload(const std::vector<T>& v) {
        hipError_t e = alloc(v.size());
        if (e != hipSuccess) return e;
        if (!v.empty()) e = hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
        return e;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
    ~DevBuf() { release(); }
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
};

}  // namespace

// The device memory a frame works in: where its kernels find it.  A frame slot's pools own theirs
// (hipMalloc, grown on demand); the slot of the entry point that allocates nothing points into
// the library's static device array instead.
struct Workspace {
    float4 *ray_o = nullptr, *ray_d = nullptr, *node_col = nullptr, *hit_a = nullptr, *sray_o = nullptr;
    uint32_t *hit_p = nullptr, *sray_blocked = nullptr;
    uint2* node_sh = nullptr;
    uint4* node_kid = nullptr;
    // level_count[RT_MAX_LEVELS] | level_base[RT_MAX_LEVELS] | shadow_count[RT_MAX_LEVELS] | overflow | work_next[2 * RT_MAX_LEVELS] | deep_max
    uint32_t* levels = nullptr;
    unsigned long long* counters = nullptr;
    size_t pool_cap = 0;    // ray-tree nodes
    size_t shadow_slots = 0;  // lit hits the shadow queue holds (sray_o: a hit's point and primitive, once for all lights)
    size_t shadow_cap = 0;    // shadow rays, all lights' together (sray_blocked)
};

// Everything one frame in flight owns.  A context has four, so that the small, latency-bound
// launches at the end of one frame can run beside the big first launches of the next.
struct FrameSlot {
    // workspace, kept across frames and grown on demand
    DevBuf<float4> ray_o, ray_d, node_col, hit_a, sray_o;
    DevBuf<uint32_t> hit_p, sray_blocked;
    DevBuf<uint2> node_sh;
    DevBuf<uint4> node_kid;
    DevBuf<uint32_t> levels;
    DevBuf<unsigned long long> counters;
    Workspace ws;           // what the kernels are given: the buffers above, or a share of the static array
    bool fixed = false;     // ws is a share of the static array: never grown, never re-run
    hipStream_t main_stream = nullptr;    // the frame's kernels
    hipStream_t shadow_stream = nullptr;  // shadow rays of a level run here, beside the next levels
    hipEvent_t ev[3] = {nullptr, nullptr, nullptr};  // timing: start, traced, resolved
    hipEvent_t ev_level[RT_MAX_LEVELS] = {};         // level l intersected and spawned
    hipEvent_t ev_shadow = nullptr;                  // all shadow launches done
    hipEvent_t ev_caller = nullptr;                  // what the caller's stream held when the frame began
    hipEvent_t ev_done = nullptr;                    // frame finished and its counters copied to the host
    hipEvent_t ev_t[RT_MAX_LEVELS + 2] = {};         // serial schedule: after each intersect level, the shadow launch, the light launch
    uint32_t timed_levels = 0;                       // ev_t[0 .. timed_levels) are intersect levels of this frame; 0 = not a timed frame
    uint32_t* h_levels = nullptr;                    // pinned copies of `levels` and `counters`, filled by the frame
    unsigned long long* h_counters = nullptr;        // itself (a blocking copy would wait for the other frame too)
    // the frame in flight
    bool busy = false;
    bool unsettled = false;   // queued with pools of unproven size and not waited for yet (settle_frame)
    bool finished = false;    // waited for and its statistics kept in `stats` already (rt_render_device_begin did)
    int finish_rc = 0;
    rt_stats stats{};
    bool one_stream = false;  // this frame's shadow launches go to the main stream too
    bool fused = false;       // this frame's levels below the camera rays ran as one launch
    const rt_scene* scene = nullptr;
    rt_view view{};
    rt_options opt{};
    void* out_f32 = nullptr;
    void* out_u8 = nullptr;
    hipStream_t caller = nullptr;
    int variant_deep = -1, variant_shadow = -1;  // CULL values of the kernel instances this frame launched
    // Camera rays of a scene with the hit-list tables whose tree partitions well (choose_level0): which way this
    // frame's go (0 leaf by leaf, 1 by the wave's beam where that is cheap), whether the frame is one of the two that
    // are timed to find out, and the events round its level-0 launch.
    int l0_variant = 0;
    bool l0_trial = false;
    uint32_t l0_epoch = 0;
    hipEvent_t ev_l0[2] = {nullptr, nullptr};
    // A context over several devices: the frame of ticket t uses slot t of every rank's context.  Each
    // rank's slot has its own compact tile buffers (the root's: the gathered buffers), the stream its
    // share of the frame ends on and its copy to the root runs on, and the event that ends both.
    DevBuf<float> tile_f32, gathered_f32;
    DevBuf<uint8_t> tile_u8, gathered_u8;
    hipStream_t gather_stream = nullptr;
    hipEvent_t ev_gather = nullptr;
    bool multi = false;        // the root's slot: a frame over all the context's devices is in flight under this ticket
    void* multi_f32 = nullptr; // ... into these images of the caller's, on this stream of the caller's
    void* multi_u8 = nullptr;
    hipStream_t multi_stream = nullptr;
};

constexpr uint32_t RT_MAX_FRAME_SLOTS = 4;

struct rt_context {
    int device = 0;
    std::string err;
    FrameSlot slot[RT_MAX_FRAME_SLOTS];  // all set up, the first n_slots in use
    uint32_t n_slots = RT_MAX_FRAME_SLOTS;
    uint32_t next_slot = 0;
    // A context over several devices is this context (the root: rank 0, where images are put
    // together) plus one one-device context per further rank.  Every rank keeps its own compact
    // tile buffers, a stream for its frame and the copy to the root, and the event that ends both.
    std::vector<rt_context*> peers;
    // per further rank: 1 = its device reaches the root's memory directly (peer access enabled, or the same
    // device), 0 = it does not and the runtime stages the copies; and what the runtime said
    std::vector<int> peer_direct;
    std::vector<std::string> peer_note;
    FrameSlot fixed_slot;                // rt_render_device_static: its workspace is a share of `scratch`
    unsigned char* scratch_ptr = nullptr;  // this device's copy of `scratch` (hipGetSymbolAddress)
    // Pool sizes every slot is brought to before a frame LIKE the finished ones they come from (same
    // scene, image share and depth: the hint's key below): the largest any of those needed, with
    // headroom, so that a slot's first frame does not overflow where another's already has.  Frames of
    // another kind start from their own floor and reset these.
    size_t want_pool = 0, want_shadow1 = 0;
    uint64_t workspace_limit = 0;        // bytes of pools per frame slot the caller allows; 0 = no limit
    DevBuf<float> out_f32;
    DevBuf<uint8_t> out_u8;
    bool kernels_ready = false;
    uint32_t deep_exit_polls = 4;        // looks at an empty stack after which an idle wave of k_deep leaves (RTAMD_DEEP_EXIT_POLLS; 0: never)
    bool fuse_deep = true;               // all levels below the camera rays in one launch where the kernels allow (RTAMD_FUSE_DEEP=0: never)
    // Sizes of the last finished frame's levels: how many blocks the next frame of the same scene and
    // size launches per level.  Only a hint -- every kernel strides over its level whatever the grid.
    uint64_t hint_scene = 0;  // rt_scene::serial (a number, not an address: a scene created where a destroyed one lay is another scene)
    uint32_t hint_slots = 0, hint_depth = 0;
    // ([1]: of frames whose deeper levels ran as one launch -- there "level 2" stands for all levels from 2 on)
    uint32_t hint_rays[2][RT_MAX_LEVELS] = {}, hint_lit[2][RT_MAX_LEVELS] = {};
    bool hint_valid[2] = {false, false};
    // Which way camera rays go (FrameSlot::l0_variant) is measured, per scene, image share and arithmetic mode: the
    // first frame of a kind goes one way, the second the other, each alone on the device with its level-0 launch
    // timed, and the frames after them the faster way.  (Both ways give the same image.)
    int beam_level0 = 1;  // RTAMD_BEAM_LEVEL0: 0 never the beam, 1 measured, 2 always
    uint64_t l0_scene = 0;
    uint32_t l0_slots = 0, l0_mode = 0, l0_epoch = 0;
    float l0_ms[2] = {-1.0f, -1.0f};
    bool l0_pending[2] = {false, false};
};

static std::atomic<uint64_t> g_scene_serial{0};

struct rt_scene {
    rt_context* ctx = nullptr;
    uint64_t serial = ++g_scene_serial;  // never 0, never reused
    std::vector<rt_scene*> peer_scenes;  // the copies on the context's further devices
    ~rt_scene() {
        for (rt_scene* p : peer_scenes) delete p;
    }
    SceneData data;
    HostTree tree;
    DevBuf<uint4> nodes, cone_items, poly_items;
    DevBuf<float4> sph;
    DevBuf<uint2> sph_id;
    DevBuf<float> bnd, bnd_box;
    DevBuf<float4> poly_box;
    DevBuf<float4> cones, poly_rec, poly_bound, cone_bound, vpos, vnrm, surfaces, lights;
    DevBuf<uint32_t> prim_surface, prim_entry;
    DevBuf<float4> prim_box;
    DevScene dev{};
};

namespace {

constexpr uint32_t LV_SHADOW_COUNT = 2 * RT_MAX_LEVELS, LV_OVERFLOW = 3 * RT_MAX_LEVELS, LV_WORK_NEXT = 3 * RT_MAX_LEVELS + 1,
                   LV_DEEP_MAX = 5 * RT_MAX_LEVELS + 1, LV_WORDS = 5 * RT_MAX_LEVELS + 2;
constexpr uint32_t N_COUNTERS = 40;  // see FrameArgs::counters
constexpr uint32_t SHADOW_GROUP_FROM = 4;  // levels from here on are too small to fill the GPU: one shadow launch for all

int fail(rt_context* ctx, int code, const std::string& msg) {
    if (ctx) ctx->err = msg;
    return code;
}

// (The launchers report through hipGetLastError, which also returns what an earlier failed call of
// anybody in this thread left behind -- another library probing device ordinals, say: cleared first.)
#define RT_HIP(ctx, call)                                                                              \
    do {                                                                                               \
        (void)hipGetLastError();                                                                       \
        hipError_t e_ = (call);                                                                        \
        if (e_ != hipSuccess)                                                                          \
            return fail(ctx, RT_ERR_DEVICE, std::string(#call) + ": " + hipGetErrorString(e_));        \
    } while (0)

float4 f4(const float* v, float w) { return make_float4(v[0], v[1], v[2], w); }
float comp4(const float4& v, uint32_t axis) { return axis == 0 ? v.x : (axis == 1 ? v.y : v.z); }
float bits_to_float(uint32_t u) {
    float f;
    memcpy(&f, &u, 4);
    return f;
}

constexpr size_t NODE_BYTES = 4 * sizeof(float4) + sizeof(uint4) + sizeof(uint32_t) + sizeof(uint2);  // per ray-tree node
constexpr size_t SHADOW_SLOT_BYTES = sizeof(float4);  // per lit hit: the queue entry its shadow rays share
constexpr size_t SHADOW_RAY_BYTES = sizeof(uint32_t);  // per lit hit and light: the ray's answer
size_t shadow_bytes(size_t slots, size_t rays) { return slots * SHADOW_SLOT_BYTES + rays * SHADOW_RAY_BYTES; }
// lit hits a workspace holds when a scene has `lights1` lights
size_t shadow_slots_of(size_t slots, size_t rays, size_t lights1) { return std::min(slots, rays / lights1); }

// Both leave the slot consistent when an allocation fails: the whole set released, capacity 0.
int ensure_pool(rt_context* ctx, FrameSlot& f, size_t cap) {
    if (f.ws.pool_cap >= cap) return RT_OK;
    if (f.fixed) return fail(ctx, RT_ERR_CAPACITY, "the static workspace holds fewer ray-tree nodes than this frame needs");
    f.ws.pool_cap = 0;
    hipError_t e = f.ray_o.alloc(cap);
    if (e == hipSuccess) e = f.ray_d.alloc(cap);
    if (e == hipSuccess) e = f.node_col.alloc(cap);
    if (e == hipSuccess) e = f.node_kid.alloc(cap);
    if (e == hipSuccess) e = f.hit_a.alloc(cap);
    if (e == hipSuccess) e = f.hit_p.alloc(cap);
    if (e == hipSuccess) e = f.node_sh.alloc(cap);
    if (e != hipSuccess) {
        f.ray_o.release(); f.ray_d.release(); f.node_col.release(); f.node_kid.release();
        f.hit_a.release(); f.hit_p.release(); f.node_sh.release();
        (void)hipGetLastError();
        return fail(ctx, RT_ERR_DEVICE, std::string("ray pool of ") + std::to_string(cap) + " nodes: " + hipGetErrorString(e));
    }
    f.ws.ray_o = f.ray_o.p; f.ws.ray_d = f.ray_d.p; f.ws.node_col = f.node_col.p; f.ws.node_kid = f.node_kid.p;
    f.ws.hit_a = f.hit_a.p; f.ws.hit_p = f.hit_p.p; f.ws.node_sh = f.node_sh.p;
    f.ws.pool_cap = cap;
    return RT_OK;
}

int ensure_shadow(rt_context* ctx, FrameSlot& f, size_t slots, size_t lights1) {
    const size_t cap = slots * lights1;
    if (f.ws.shadow_slots >= slots && f.ws.shadow_cap >= cap) return RT_OK;
    if (f.fixed) return fail(ctx, RT_ERR_CAPACITY, "the static workspace holds fewer shadow rays than this frame needs");
    // (each array only ever grows: a scene with fewer lights and more hits keeps the answers' array)
    const size_t new_slots = std::max(slots, f.ws.shadow_slots), new_cap = std::max(cap, f.ws.shadow_cap);
    hipError_t e = hipSuccess;
    if (new_slots != f.ws.shadow_slots) {
        f.ws.shadow_slots = 0;
        e = f.sray_o.alloc(new_slots);
    }
    if (e == hipSuccess && new_cap != f.ws.shadow_cap) {
        f.ws.shadow_cap = 0;
        e = f.sray_blocked.alloc(new_cap);
    }
    if (e != hipSuccess) {
        f.sray_o.release(); f.sray_blocked.release();
        f.ws.shadow_slots = f.ws.shadow_cap = 0;
        (void)hipGetLastError();
        return fail(ctx, RT_ERR_DEVICE, std::string("shadow queues of ") + std::to_string(cap) + " rays: " + hipGetErrorString(e));
    }
    f.ws.sray_o = f.sray_o.p; f.ws.sray_blocked = f.sray_blocked.p;
    f.ws.shadow_slots = new_slots;
    f.ws.shadow_cap = new_cap;
    return RT_OK;
}

int prepare_kernels(rt_context* ctx) {
    if (ctx->kernels_ready) return RT_OK;
    RT_HIP(ctx, rt_exact::init_kernels());
    RT_HIP(ctx, rt_fast::init_kernels());
    ctx->kernels_ready = true;
    return RT_OK;
}

// Device address of `scratch` on `device`, which must be the current device: looked up once per
// device and kept (each GPU has its own copy of the array at its own address).
constexpr int MAX_DEVICES = 64;
unsigned char* g_scratch_ptr[MAX_DEVICES] = {};
std::mutex g_scratch_mutex;

int scratch_address(rt_context* ctx, int device, unsigned char** out) {
    if (device < 0 || device >= MAX_DEVICES) return fail(ctx, RT_ERR_INVALID, "device id out of range");
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[device]) {
        void* p = nullptr;
        RT_HIP(ctx, hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        if (!p) return fail(ctx, RT_ERR_DEVICE, "no device address for the static workspace");
        g_scratch_ptr[device] = (unsigned char*)p;
    }
    *out = g_scratch_ptr[device];
    return RT_OK;
}

// Streams, events and the pinned copies of a slot's counters.
int setup_slot(rt_context* ctx, FrameSlot& f) {
    for (auto& e : f.ev) RT_HIP(ctx, hipEventCreate(&e));
    RT_HIP(ctx, hipStreamCreateWithFlags(&f.main_stream, hipStreamNonBlocking));
    RT_HIP(ctx, hipStreamCreateWithFlags(&f.shadow_stream, hipStreamNonBlocking));
    for (auto& e : f.ev_level) RT_HIP(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    RT_HIP(ctx, hipEventCreateWithFlags(&f.ev_shadow, hipEventDisableTiming));
    RT_HIP(ctx, hipEventCreateWithFlags(&f.ev_caller, hipEventDisableTiming));
    RT_HIP(ctx, hipEventCreateWithFlags(&f.ev_done, hipEventDisableTiming));
    RT_HIP(ctx, hipStreamCreateWithFlags(&f.gather_stream, hipStreamNonBlocking));
    RT_HIP(ctx, hipEventCreateWithFlags(&f.ev_gather, hipEventDisableTiming));
    for (auto& e : f.ev_t) RT_HIP(ctx, hipEventCreate(&e));
    for (auto& e : f.ev_l0) RT_HIP(ctx, hipEventCreate(&e));
    RT_HIP(ctx, hipHostMalloc((void**)&f.h_levels, LV_WORDS * sizeof(uint32_t), hipHostMallocDefault));
    RT_HIP(ctx, hipHostMalloc((void**)&f.h_counters, N_COUNTERS * sizeof(unsigned long long), hipHostMallocDefault));
    return RT_OK;
}

void teardown_slot(FrameSlot& f) {
    if (f.busy && f.main_stream) (void)hipStreamSynchronize(f.main_stream);
    for (auto& e : f.ev)
        if (e) (void)hipEventDestroy(e);
    for (auto& e : f.ev_level)
        if (e) (void)hipEventDestroy(e);
    if (f.ev_shadow) (void)hipEventDestroy(f.ev_shadow);
    if (f.ev_caller) (void)hipEventDestroy(f.ev_caller);
    if (f.ev_done) (void)hipEventDestroy(f.ev_done);
    if (f.ev_gather) (void)hipEventDestroy(f.ev_gather);
    if (f.gather_stream) (void)hipStreamDestroy(f.gather_stream);
    for (auto& e : f.ev_t)
        if (e) (void)hipEventDestroy(e);
    for (auto& e : f.ev_l0)
        if (e) (void)hipEventDestroy(e);
    if (f.h_levels) (void)hipHostFree(f.h_levels);
    if (f.h_counters) (void)hipHostFree(f.h_counters);
    if (f.shadow_stream) (void)hipStreamDestroy(f.shadow_stream);
    if (f.main_stream) (void)hipStreamDestroy(f.main_stream);
}

}  // namespace

extern "C" int rt_context_create(int n_devices, const int* device_ids, rt_context** out) {
    if (!out) return RT_ERR_INVALID;
    *out = nullptr;
    rt_context* ctx = new rt_context();
    *out = ctx;  // returned even on failure so the message can be read; destroy it afterwards
    if (n_devices < 1 || n_devices > MAX_DEVICES) return fail(ctx, RT_ERR_INVALID, "n_devices out of range");
    int count = 0;
    RT_HIP(ctx, hipGetDeviceCount(&count));
    ctx->device = device_ids ? device_ids[0] : 0;
    if (ctx->device >= MAX_DEVICES) return fail(ctx, RT_ERR_INVALID, "device id out of range");
    if (ctx->device < 0 || ctx->device >= count) return fail(ctx, RT_ERR_INVALID, "device id out of range");
    RT_HIP(ctx, hipSetDevice(ctx->device));
    hipDeviceProp_t prop;
    RT_HIP(ctx, hipGetDeviceProperties(&prop, ctx->device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(ctx, RT_ERR_UNSUPPORTED, std::string("librtamd is built for gfx950 only; device is ") + prop.gcnArchName);
    // Frames in flight: up to four workspaces (the caller decides how many it keeps queued; which
    // number is fastest depends on the scene, see enqueue_frame).
    if (const char* s = getenv("RTAMD_FRAMES_IN_FLIGHT"))
        ctx->n_slots = std::min<uint32_t>(RT_MAX_FRAME_SLOTS, std::max<uint32_t>(1, (uint32_t)strtoul(s, nullptr, 10)));
    for (FrameSlot& f : ctx->slot) {
        int rc = setup_slot(ctx, f);
        if (rc != RT_OK) return rc;
        RT_HIP(ctx, f.levels.alloc(LV_WORDS));
        RT_HIP(ctx, f.counters.alloc(N_COUNTERS));
        f.ws.levels = f.levels.p;
        f.ws.counters = f.counters.p;
    }
    if (const char* s = getenv("RTAMD_FUSE_DEEP")) ctx->fuse_deep = strtoul(s, nullptr, 10) != 0;
    if (const char* s = getenv("RTAMD_BEAM_LEVEL0")) ctx->beam_level0 = (int)strtoul(s, nullptr, 10);
    if (const char* s = getenv("RTAMD_DEEP_EXIT_POLLS")) ctx->deep_exit_polls = (uint32_t)strtoul(s, nullptr, 10);
    if (const char* s = getenv("RTAMD_WORKSPACE_MAX_MB")) ctx->workspace_limit = (uint64_t)strtoull(s, nullptr, 10) << 20;
    // The entry point that allocates nothing works in the static array: this device's copy of it
    // is looked up here, with the device current, and its slot's streams, events and pinned
    // counters are made now, so that the entry point itself has nothing left to set up.
    {
        int rc = scratch_address(ctx, ctx->device, &ctx->scratch_ptr);
        if (rc != RT_OK) return rc;
        ctx->fixed_slot.fixed = true;
        rc = setup_slot(ctx, ctx->fixed_slot);
        if (rc != RT_OK) return rc;
    }
    {
        int rc = prepare_kernels(ctx);
        if (rc != RT_OK) return rc;
    }
    // further devices: one one-device context each; their tile buffers are copied into this device
    for (int r = 1; r < n_devices; ++r) {
        const int id = device_ids ? device_ids[r] : r;
        rt_context* peer = nullptr;
        const int rc = rt_context_create(1, &id, &peer);
        if (peer) ctx->peers.push_back(peer);
        if (rc != RT_OK) return fail(ctx, rc, std::string("device ") + std::to_string(id) + ": " + (peer ? peer->err : "no context"));
        // Direct copies over the GPUs' links where the platform allows them; what the runtime answered is
        // kept (rt_context_peer_access): without peer access the copies still work, staged by the runtime.
        int direct = 1;
        std::string note = "same device";
        if (id != ctx->device) {
            int can = 0;
            hipError_t e = hipDeviceCanAccessPeer(&can, id, ctx->device);
            if (e == hipSuccess && can) {
                e = hipSetDevice(id);
                if (e == hipSuccess) e = hipDeviceEnablePeerAccess(ctx->device, 0);
                if (e == hipErrorPeerAccessAlreadyEnabled) e = hipSuccess;
            }
            (void)hipGetLastError();
            direct = (e == hipSuccess && can) ? 1 : 0;
            note = e != hipSuccess ? std::string("peer access refused: ") + hipGetErrorString(e)
                                   : (can ? "peer access enabled" : "the device cannot reach the root's memory: copies are staged");
        }
        ctx->peer_direct.push_back(direct);
        ctx->peer_note.push_back(note);
    }
    RT_HIP(ctx, hipSetDevice(ctx->device));
    return RT_OK;
}

extern "C" uint32_t rt_context_device_count(const rt_context* ctx) { return ctx ? 1u + (uint32_t)ctx->peers.size() : 0u; }

extern "C" int rt_context_peer_access(const rt_context* ctx, uint32_t rank, const char** note_out) {
    if (!ctx || rank > ctx->peers.size()) return RT_ERR_INVALID;
    if (note_out) *note_out = rank == 0 ? "the root device" : ctx->peer_note[rank - 1].c_str();
    return rank == 0 ? 1 : ctx->peer_direct[rank - 1];
}

extern "C" void rt_context_destroy(rt_context* ctx) {
    if (!ctx) return;
    for (rt_context* p : ctx->peers) rt_context_destroy(p);
    (void)hipSetDevice(ctx->device);
    for (FrameSlot& f : ctx->slot) teardown_slot(f);
    teardown_slot(ctx->fixed_slot);
    delete ctx;
}

extern "C" const char* rt_last_error(const rt_context* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

extern "C" int rt_scene_create(rt_context* ctx, const rt_scene_desc* desc, rt_scene** out) {
    if (!ctx || !desc || !out) return fail(ctx, RT_ERR_INVALID, "null argument");
    *out = nullptr;
    std::string err;
    int rc = validate_desc(*desc, err);
    if (rc != RT_OK) return fail(ctx, rc, err);
    RT_HIP(ctx, hipSetDevice(ctx->device));

    std::unique_ptr<rt_scene> sc(new rt_scene());
    sc->ctx = ctx;
    sc->data.assign(*desc);
    if (desc->nodes) {  // adopt the caller's partition
        sc->tree.nodes.assign(desc->nodes, desc->nodes + desc->n_nodes);
        sc->tree.refs.assign(desc->leaf_refs, desc->leaf_refs + desc->n_leaf_refs);
        rt_tree_info& ti = sc->tree.info;
        ti.n_nodes = desc->n_nodes;
        ti.n_leaf_refs = desc->n_leaf_refs;
        // depth by walking down from the root (children always follow their parent)
        std::vector<uint32_t> depth(desc->n_nodes, 0);
        for (uint32_t i = 0; i < desc->n_nodes; ++i) {
            const rt_kd_node& n = desc->nodes[i];
            if (n.axis == 3) {
                ti.n_leaves++;
                ti.max_leaf = std::max(ti.max_leaf, n.over);
            } else {
                depth[n.under] = std::max(depth[n.under], depth[i] + 1);
                depth[n.over] = std::max(depth[n.over], depth[i] + 1);
            }
            ti.max_depth = std::max(ti.max_depth, depth[i]);
        }
    } else {
        rt_scene_desc d = sc->data.desc;
        d.build_threads = desc->build_threads;
        // Nodes with many primitives have their candidate planes costed on the GPU (2n candidates
        // x n primitives each: the O(n^2) part of the reference's construction).
        DevBuf<float> d_boxes, d_cost;
        DevBuf<uint32_t> d_list;
        std::string dev_err;
        const SplitCostFn on_gpu = [&](const float* boxes6, uint32_t n_prims, const uint32_t* list, uint32_t n, int axis,
                                       float* cost) -> bool {
            if (!dev_err.empty()) return false;
            hipError_t e = hipSuccess;
            if (!d_boxes.p) {
                e = d_boxes.alloc((size_t)n_prims * 6);
                if (e == hipSuccess) e = hipMemcpy(d_boxes.p, boxes6, (size_t)n_prims * 6 * sizeof(float), hipMemcpyHostToDevice);
                if (e == hipSuccess) e = d_list.alloc(n_prims);  // a node's list never exceeds the scene (:160-169)
                if (e == hipSuccess) e = d_cost.alloc((size_t)n_prims * 2);
            }
            if (e == hipSuccess && n > n_prims) return false;
            if (e == hipSuccess) e = hipMemcpy(d_list.p, list, (size_t)n * sizeof(uint32_t), hipMemcpyHostToDevice);
            if (e == hipSuccess) e = rt_exact::launch_appraise(d_boxes.p, d_list.p, n, (uint32_t)axis, d_cost.p, nullptr);
            if (e == hipSuccess) e = hipMemcpy(cost, d_cost.p, (size_t)n * 2 * sizeof(float), hipMemcpyDeviceToHost);
            if (e != hipSuccess) {
                dev_err = hipGetErrorString(e);
                return false;  // the host threads take over
            }
            return true;
        };
        const char* wm = getenv("RTAMD_GPU_BUILD_MIN");  // primitives in a node from which the GPU costs it
        const uint32_t wide_min = wm ? (uint32_t)strtoul(wm, nullptr, 10) : 2048u;
        rc = build_tree(d, sc->tree, err, &on_gpu, wide_min);
        if (rc != RT_OK) return fail(ctx, rc, err);
        if (!dev_err.empty()) return fail(ctx, RT_ERR_DEVICE, "tree construction on the GPU: " + dev_err);
    }

    const SceneData& sd = sc->data;
    // Device nodes + leaf runs.  A leaf's list (src/space_partition.rs:348-385 walks it in order)
    // becomes one run per primitive kind, each in list order with the list position attached.
    static_assert(sizeof(rt_kd_node) == 64, "node layout");
    if (sc->tree.nodes.size() >= (1u << 29)) return fail(ctx, RT_ERR_UNSUPPORTED, "more than 2^29 partition nodes");
    std::vector<uint4> nodes(sc->tree.nodes.size() * 4, make_uint4(0, 0, 0, 0));
    std::vector<float4> sph;
    std::vector<uint2> sph_id;
    std::vector<float> bnd;  // 4 floats per record
    std::vector<float> bnd_box;  // 8 floats per record of bnd: the box records of polygon runs (cluster_boxes)
    std::vector<float4> poly_box;  // 2 per polygon item: its own box {lo, -} {hi, -}
    std::vector<uint4> cone_items, poly_items;
    std::vector<float4> cone_bound, poly_bound;  // one record per item, for the item alone
    auto push_record = [](std::vector<float4>& to, const LeafSphere& e) {
        float rec[4];
        sphere_record(e, rec);
        to.push_back(make_float4(rec[0], rec[1], rec[2], rec[3]));
    };
    bool any_poly_walk = false;
    // One leaf's runs: `list` holds {primitive, position in the leaf's list}; returns the leaf record's words.
    auto emit_leaf = [&](const std::vector<std::pair<uint32_t, uint32_t>>& list, uint4 rec[3]) {
        const uint32_t s0 = (uint32_t)sph.size(), p0 = (uint32_t)poly_items.size();
        std::vector<LeafSphere> ls, lp, lc;
        std::vector<uint4> true_cones;
        bool poly_bounds = true;
        for (const auto& pk : list) {
            const uint32_t prim = pk.first, k = pk.second;
            const rt_primitive& p = sd.primitives[prim];
            if (p.kind == RT_KIND_SPHERE) {
                const rt_sphere& s = sd.spheres[p.index];
                ls.push_back(LeafSphere{{s.center[0], s.center[1], s.center[2]}, s.radius, prim, k, 0, 0});
            } else if (p.kind == RT_KIND_CONE) {
                LeafSphere e{{0.0f, 0.0f, 0.0f}, 0.0f, prim, k, p.index, 0};
                if (cylinder_bound(sd.cones[p.index], e.c, &e.r)) lc.push_back(e);
                else true_cones.push_back(make_uint4(prim, k, p.index, 0));
            } else {
                const rt_polygon& pg = sd.polygons[p.index];
                LeafSphere e{{0.0f, 0.0f, 0.0f}, 0.0f, prim, k, pg.first_vertex, pg.vertex_count};
                float box6[6];
                if (!polygon_bound(sd.vertices.data(), pg.first_vertex, pg.vertex_count, e.c, &e.r, box6)) {
                    poly_bounds = false;
                    e.c[0] = e.c[1] = e.c[2] = 0.0f;
                    e.r = INFINITY;
                } else {
                    memcpy(e.lo, box6, 12);
                    memcpy(e.hi, box6 + 3, 12);
                }
                lp.push_back(e);
            }
        }
        // polygon runs worth a hierarchy get one over the spheres that stand in for the polygons
        uint32_t pu0 = 0xffffffffu;
        if (poly_bounds && lp.size() >= RT_HIER_MIN) {
            pu0 = cluster_spheres(lp, bnd, 5);
            cluster_boxes(lp, bnd, pu0, bnd_box);
            any_poly_walk = true;
        }
        for (const LeafSphere& e : lp) {
            poly_items.push_back(make_uint4(e.prim, e.pos, e.a0, e.a1));
            push_record(poly_bound, e);
            poly_box.push_back(make_float4(e.lo[0], e.lo[1], e.lo[2], 0.0f));
            poly_box.push_back(make_float4(e.hi[0], e.hi[1], e.hi[2], 0.0f));
        }
        // Cone items: the cylinders first (under a hierarchy of their own when there are enough of
        // them), then the cones proper, which no sphere can stand in for.
        uint32_t cu0 = 0xffffffffu;
        if (lc.size() >= RT_HIER_MIN) {
            cu0 = cluster_spheres(lc, bnd);
        }
        const uint32_t c0 = (uint32_t)cone_items.size() + (cu0 != 0xffffffffu ? (uint32_t)lc.size() : 0u);
        for (const LeafSphere& e : lc) {
            cone_items.push_back(make_uint4(e.prim, e.pos, e.a0, 0));
            push_record(cone_bound, e);
        }
        cone_items.insert(cone_items.end(), true_cones.begin(), true_cones.end());
        cone_bound.insert(cone_bound.end(), true_cones.size(), make_float4(0.0f, 0.0f, 0.0f, INFINITY));
        const uint32_t u0 = cluster_spheres(ls, bnd);  // re-ordered, padded to 8, bounding records appended
        const uint32_t ns = (uint32_t)ls.size();
        for (const LeafSphere& e : ls) {
            sph.push_back(make_float4(e.c[0], e.c[1], e.c[2], e.r * e.r));  // src/sphere.rs:53
            sph_id.push_back(make_uint2(e.prim, e.pos));
        }
        rec[0] = make_uint4(cu0, 3u, 0u, 0u);
        rec[1] = make_uint4(s0, ns, c0, (uint32_t)cone_items.size() - c0);
        rec[2] = make_uint4(p0, (uint32_t)poly_items.size() - p0, u0, pu0);
    };
    std::vector<std::pair<uint32_t, uint32_t>> leaf_list;
    for (size_t ni = 0; ni < sc->tree.nodes.size(); ++ni) {
        const rt_kd_node& n = sc->tree.nodes[ni];
        if (n.axis != 3) {  // {plane, axis, under, over} + the children's boxes
            const rt_kd_node& u = sc->tree.nodes[n.under];
            const rt_kd_node& o = sc->tree.nodes[n.over];
            auto fb = [](float f) { uint32_t b; memcpy(&b, &f, 4); return b; };
            nodes[ni * 4 + 0] = make_uint4(fb(n.plane), n.axis, n.under, n.over);
            nodes[ni * 4 + 1] = make_uint4(fb(u.bmin[0]), fb(u.bmin[1]), fb(u.bmin[2]), fb(u.bmax[0]));
            nodes[ni * 4 + 2] = make_uint4(fb(u.bmax[1]), fb(u.bmax[2]), fb(o.bmin[0]), fb(o.bmin[1]));
            nodes[ni * 4 + 3] = make_uint4(fb(o.bmin[2]), fb(o.bmax[0]), fb(o.bmax[1]), fb(o.bmax[2]));
            continue;
        }
        leaf_list.clear();
        for (uint32_t k = 0; k < n.over; ++k) leaf_list.emplace_back(sc->tree.refs[n.under + k], k);
        emit_leaf(leaf_list, &nodes[ni * 4]);
    }
    const bool lean_scene = cone_items.empty() && !any_poly_walk && poly_items.size() <= 8;  // (what the real leaves hold)

    // The hit-list walk (rt_traverse.h, HitWalk) needs: every primitive once, as one more "leaf" whose list
    // is the scene's (a leaf's list is a sub-sequence of it, src/space_partition.rs:160-169, so the position
    // there orders two primitives as any leaf's list does); each primitive's box as split() read it
    // (:163-166: corner, fl(corner + extent)); where a primitive's entry lies in that leaf's runs; and the
    // certainty that this partition's leaves ARE what split() makes of those boxes (a partition the caller
    // handed in need not be).
    uint4 all_rec[3] = {make_uint4(0xffffffffu, 3u, 0u, 0u), make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0xffffffffu)};
    std::vector<float4> pboxes;
    std::vector<uint32_t> prim_entry;
    // (Lean scenes get the tables too: their camera rays may go a tile at a time, choose_level0; every other ray of
    // theirs keeps the lean kernels, which never see the extra leaf's records.)
    bool hit_walk = sd.primitives.size() > 0 && sd.primitives.size() < (1u << 29) && sc->tree.nodes.size() < (1u << HITWALK_NODE_BITS);
    const size_t bnd_real = bnd.size();  // the real leaves' bounding records end here
    bool force_hit_walk = false;  // (experiment, RTAMD_HIT_WALK=2: also scenes of spheres alone, which then run the general kernels)
    if (const char* e = getenv("RTAMD_HIT_WALK")) {
        hit_walk = hit_walk && atoi(e) != 0;
        if (atoi(e) == 2 && sd.primitives.size() > 0) hit_walk = force_hit_walk = true;
    }
    if (hit_walk) {
        const uint32_t np = (uint32_t)sd.primitives.size();
        pboxes.resize((size_t)np * 2);
        for (uint32_t i = 0; i < np; ++i) {
            float c[3], e[3];
            prim_box(sd.desc, i, c, e);
            pboxes[i * 2] = make_float4(c[0], c[1], c[2], 0.0f);
            pboxes[i * 2 + 1] = make_float4(c[0] + e[0], c[1] + e[1], c[2] + e[2], 0.0f);
        }
        // the leaves follow from the boxes and the planes?
        std::vector<std::pair<uint32_t, std::vector<uint32_t>>> todo;
        {
            std::vector<uint32_t> all(np);
            for (uint32_t i = 0; i < np; ++i) all[i] = i;
            todo.emplace_back(0u, std::move(all));
        }
        while (hit_walk && !todo.empty()) {
            const uint32_t ni = todo.back().first;
            const std::vector<uint32_t> list = std::move(todo.back().second);
            todo.pop_back();
            const rt_kd_node& n = sc->tree.nodes[ni];
            if (n.axis == 3) {
                hit_walk = n.over == list.size() && (list.empty() || memcmp(&sc->tree.refs[n.under], list.data(), list.size() * sizeof(uint32_t)) == 0);
                continue;
            }
            if (n.axis > 2) { hit_walk = false; break; }
            std::vector<uint32_t> under, over;
            for (uint32_t p : list) {  // src/space_partition.rs:160-169
                if (comp4(pboxes[p * 2 + 1], n.axis) > n.plane) over.push_back(p);
                if (comp4(pboxes[p * 2], n.axis) <= n.plane) under.push_back(p);
            }
            todo.emplace_back(n.over, std::move(over));
            todo.emplace_back(n.under, std::move(under));
        }
    }
    if (hit_walk) {
        leaf_list.clear();
        for (uint32_t i = 0; i < (uint32_t)sd.primitives.size(); ++i) leaf_list.emplace_back(i, i);
        const size_t s_from = sph_id.size(), c_from = cone_items.size(), p_from = poly_items.size();
        emit_leaf(leaf_list, all_rec);
        prim_entry.assign(sd.primitives.size(), 0xffffffffu);
        for (size_t i = s_from; i < sph_id.size(); ++i)
            if ((sph_id[i].y >> 31) == 0u) prim_entry[sph_id[i].x] = (uint32_t)i;  // (not the padding copies)
        for (size_t i = c_from; i < cone_items.size(); ++i)
            if ((cone_items[i].y >> 31) == 0u) prim_entry[cone_items[i].x] = (uint32_t)i | (1u << 30);
        for (size_t i = p_from; i < poly_items.size(); ++i)
            if ((poly_items[i].y >> 31) == 0u) prim_entry[poly_items[i].x] = (uint32_t)i | (2u << 30);
        for (uint32_t e : prim_entry)
            if (e == 0xffffffffu || (e & 0x3fffffffu) >= (1u << 26)) hit_walk = false;  // (a queued test names its item in 26 bits)
    }

    std::vector<float4> cones, vpos, vnrm, surfaces, lights;
    // (one frame per cone ENTRY, beside cone_items: an entry's loads do not wait for its item)
    for (const uint4& it : cone_items) {
        const rt_cone& c = sd.cones[it.z];
        cones.push_back(f4(c.base, c.base_radius));
        cones.push_back(f4(c.apex, c.apex_radius));
    }
    for (const rt_vertex& v : sd.vertices) {
        vpos.push_back(f4(v.point, 0.0f));
        vnrm.push_back(f4(v.normal, 0.0f));
    }
    for (const rt_surface& s : sd.surfaces) {
        surfaces.push_back(f4(s.color, s.kd));
        surfaces.push_back(make_float4(s.ks, s.shine, s.reflectance, s.transmittance));
        surfaces.push_back(make_float4(s.ior, bits_to_float(s.model), 0.0f, 0.0f));
    }
    for (const rt_light& l : sd.lights) {
        lights.push_back(f4(l.position, 0.0f));
        lights.push_back(f4(l.color, 0.0f));
    }
    std::vector<uint32_t> prim_surface;
    for (const rt_primitive& p : sd.primitives) prim_surface.push_back(p.surface);

    RT_HIP(ctx, sc->nodes.upload(nodes));
    RT_HIP(ctx, sc->sph.upload(sph));
    RT_HIP(ctx, sc->sph_id.upload(sph_id));
    bnd.insert(bnd.end(), 8 * 4, 0.0f);  // the walk reads bounding records eight at a time
    RT_HIP(ctx, sc->bnd.upload(bnd));
    RT_HIP(ctx, sc->cone_items.upload(cone_items));
    RT_HIP(ctx, sc->poly_items.upload(poly_items));
    // (the walks read these records eight at a time)
    poly_bound.insert(poly_bound.end(), 8, make_float4(0.0f, 0.0f, 0.0f, INFINITY));
    cone_bound.insert(cone_bound.end(), 8, make_float4(0.0f, 0.0f, 0.0f, INFINITY));
    RT_HIP(ctx, sc->poly_bound.upload(poly_bound));
    RT_HIP(ctx, sc->cone_bound.upload(cone_bound));
    cluster_boxes(std::vector<LeafSphere>(), bnd, 0u, bnd_box);  // (the records of the other blocks, and the padding: boxes that hold everything)
    for (int k = 0; k < 8; ++k) {
        poly_box.push_back(make_float4(-INFINITY, -INFINITY, -INFINITY, 0.0f));
        poly_box.push_back(make_float4(INFINITY, INFINITY, INFINITY, 0.0f));
    }
    RT_HIP(ctx, sc->bnd_box.upload(bnd_box));
    RT_HIP(ctx, sc->poly_box.upload(poly_box));
    RT_HIP(ctx, sc->vpos.upload(vpos));
    {
        // The ray-independent part of the cone and polygon tests, by the device's own arithmetic
        // (the exact build's: no contraction, correctly rounded divide and square root).
        DevBuf<float4> cones_in;
        RT_HIP(ctx, cones_in.upload(cones));
        RT_HIP(ctx, sc->cones.alloc(cone_items.size() * 4));
        RT_HIP(ctx, sc->poly_rec.alloc(poly_items.size() * 4));
        RT_HIP(ctx, rt_exact::launch_prepare_shapes(cones_in.p, (uint32_t)cone_items.size(), sc->cones.p, sc->poly_items.p,
                                                    (uint32_t)poly_items.size(), sc->vpos.p, sc->poly_rec.p,
                                                    nullptr));
        RT_HIP(ctx, hipStreamSynchronize(nullptr));
    }
    RT_HIP(ctx, sc->vnrm.upload(vnrm));
    RT_HIP(ctx, sc->surfaces.upload(surfaces));
    RT_HIP(ctx, sc->lights.upload(lights));
    RT_HIP(ctx, sc->prim_surface.upload(prim_surface));
    if (hit_walk) {
        RT_HIP(ctx, sc->prim_box.upload(pboxes));
        RT_HIP(ctx, sc->prim_entry.upload(prim_entry));
    }

    DevScene& d = sc->dev;
    d.nodes = sc->nodes.p;
    d.sph = sc->sph.p;
    d.sph_id = sc->sph_id.p;
    d.bnd = (const float4*)sc->bnd.p;
    d.cone_items = sc->cone_items.p;
    d.poly_rec = sc->poly_rec.p;
    d.poly_bound = sc->poly_bound.p;
    d.bnd_box = (const float4*)sc->bnd_box.p;
    d.poly_box = sc->poly_box.p;
    d.cone_bound = sc->cone_bound.p;
    d.cones = sc->cones.p;
    d.vpos = sc->vpos.p;
    d.vnrm = sc->vnrm.p;
    d.surfaces = sc->surfaces.p;
    d.prim_surface = sc->prim_surface.p;
    d.lights = sc->lights.p;
    d.n_lights = (uint32_t)sd.lights.size();
    d.stack_depth = std::max<uint32_t>(1, sc->tree.info.max_depth);
    // lean: no cones and next to no polygons (a floor): kernels without the code for either's records
    d.bnd_records = (uint32_t)(bnd_real / 4) + 8u;  // (what a lean kernel's blocks copy: the real leaves' records, read eight at a time)
    d.lean = (lean_scene && !force_hit_walk && d.stack_depth <= RT_STACK_RING &&
              (size_t)d.bnd_records * sizeof(float4) <= RT_LEAN_RECORD_BYTES) ? 1u : 0u;
    d.n_nodes = (uint32_t)sc->tree.nodes.size();
    d.hit_tables = hit_walk ? 1u : 0u;
    d.poly_runs = any_poly_walk ? 1u : 0u;
    d.cone_count = (uint32_t)sd.cones.size();
    d.hit_walk = (hit_walk && d.lean == 0u) ? 1u : 0u;
    d.hit_walk_level0 = (d.hit_walk != 0u && (uint64_t)sc->tree.info.max_leaf * 4u >= sd.primitives.size()) ? 1u : 0u;
    if (const char* e = getenv("RTAMD_HIT_LEVEL0")) d.hit_walk_level0 = (d.hit_walk != 0u && atoi(e) != 0) ? 1u : 0u;  // (experiment)
    d.prim_box = sc->prim_box.p;
    d.prim_entry = sc->prim_entry.p;
    d.all_w = all_rec[0].x;
    d.all_q2 = all_rec[1];
    d.all_q3 = all_rec[2];
    d.beam_shadow = (d.hit_walk != 0u && all_rec[2].w != 0xffffffffu && all_rec[1].y < 24u && all_rec[0].x == 0xffffffffu) ? 1u : 0u;
    if (const char* e = getenv("RTAMD_BEAM_SHADOW")) d.beam_shadow = (d.beam_shadow != 0u && atoi(e) != 0) ? 1u : 0u;
    d.nodes_lds = (d.lean == 0u && (size_t)d.n_nodes * 64u <= RT_NODE_LDS_BYTES) ? 1u : 0u;
    // Measured at 3840x2160 (k_shadow, ms, lanes waiting for their wave / lanes taking the next ray): rings 4
    // (largest leaf 212) 23.1 / 18.4; lathe 96 at 1920x1080 (273) 2.41 / 2.34; mount 8 (45685) 6.4 / 8.0;
    // balls 4 (lean) 2.79 / 3.44.
    d.refill = (d.lean == 0u && sc->tree.info.max_leaf < 1024u) ? 1u : 0u;
    memcpy(d.bg, sd.background, sizeof(d.bg));
    memcpy(d.root_min, sc->tree.nodes[0].bmin, sizeof(d.root_min));
    memcpy(d.root_max, sc->tree.nodes[0].bmax, sizeof(d.root_max));
    d.reach = 0.0f;
    for (int k = 0; k < 3; ++k) d.reach = std::max(d.reach, std::max(std::fabs(d.root_min[k]), std::fabs(d.root_max[k])));
    if (d.stack_depth > RT_STACK_RING + RT_STACK_DEEP)
        return fail(ctx, RT_ERR_UNSUPPORTED, "space partition deeper than the device traversal stack (64 levels)");
    // a copy of the scene, with this partition, on each further device of the context
    for (rt_context* peer : ctx->peers) {
        rt_scene_desc pd = sc->data.desc;
        pd.n_nodes = (uint32_t)sc->tree.nodes.size();
        pd.n_leaf_refs = (uint32_t)sc->tree.refs.size();
        pd.nodes = sc->tree.nodes.data();
        pd.leaf_refs = sc->tree.refs.data();
        rt_scene* ps = nullptr;
        rc = rt_scene_create(peer, &pd, &ps);
        if (rc != RT_OK) return fail(ctx, rc, "device " + std::to_string(peer->device) + ": " + peer->err);
        sc->peer_scenes.push_back(ps);
    }
    RT_HIP(ctx, hipSetDevice(ctx->device));
    *out = sc.release();
    return RT_OK;
}

extern "C" void rt_scene_destroy(rt_scene* scene) { delete scene; }

extern "C" int rt_scene_tree_info(const rt_scene* scene, rt_tree_info* out) {
    if (!scene || !out) return RT_ERR_INVALID;
    *out = scene->tree.info;
    return RT_OK;
}

extern "C" int rt_scene_tree_copy(const rt_scene* scene, rt_kd_node* nodes, uint32_t* leaf_refs) {
    if (!scene) return RT_ERR_INVALID;
    if (nodes) memcpy(nodes, scene->tree.nodes.data(), scene->tree.nodes.size() * sizeof(rt_kd_node));
    if (leaf_refs) memcpy(leaf_refs, scene->tree.refs.data(), scene->tree.refs.size() * sizeof(uint32_t));
    return RT_OK;
}

namespace {

struct FramePlan {  // what enqueue and finish both need about a frame
    uint32_t n_slots = 0, n_lights = 0;
    uint64_t n_primary = 0;
    size_t lights1 = 1;
};

int plan_frame(rt_context* ctx, const FrameSlot& f, FramePlan& p) {
    const rt_view& view = f.view;
    const rt_options& opt = f.opt;
    const uint32_t tile_count = opt.tile_count ? opt.tile_count : 1;
    const TileGrid grid(view.width, view.height, tile_count);
    const uint64_t n_slots64 = grid.tiles_of(opt.tile_rank) * 64;
    if (n_slots64 > 0x7fffffffull) return fail(ctx, RT_ERR_UNSUPPORTED, "image too large");
    p.n_slots = (uint32_t)n_slots64;
    p.n_lights = f.scene->dev.n_lights;
    p.lights1 = std::max<uint32_t>(1, p.n_lights);
    // owned pixels = primary rays
    if (tile_count == 1) {
        p.n_primary = (uint64_t)view.width * view.height;
    } else {
        p.n_primary = 0;
        for (uint64_t lt = 0; lt < grid.tiles_of(opt.tile_rank); ++lt) {
            const uint64_t t = lt * tile_count + opt.tile_rank;
            const uint32_t tx = (uint32_t)(t % grid.tiles_x), ty = (uint32_t)(t / grid.tiles_x);
            const uint32_t w = std::min<uint32_t>(8, view.width - tx * 8), h = std::min<uint32_t>(8, view.height - ty * 8);
            p.n_primary += (uint64_t)w * h;
        }
    }
    return RT_OK;
}

// Puts the frame of slot `f` on the slot's streams with pools of at least (cap, scap) entries.
// Nothing here waits for the GPU.  The kernels run on the slot's own stream; only the last one
// (resolve of level 0, the one that writes the caller's buffers) is ordered after what the
// caller's stream held when the frame began, and the caller's stream then waits for the frame.
// Which way the camera rays of the frame in slot `f` go, and whether the frame is timed for it (rt_context::l0_ms).
void choose_level0(rt_context* ctx, FrameSlot& f, const FramePlan& p) {
    f.l0_variant = 0;
    f.l0_trial = false;
    const DevScene& d = f.scene->dev;
    const rt_options& o = f.opt;
    if (ctx->beam_level0 == 0 || o.collect_work != 0 || o.exhaustive_leaves != 0 || o.leaf_walk == 1u || d.hit_tables == 0u || d.hit_walk_level0 != 0u)
        return;
    if (ctx->beam_level0 >= 2 || o.leaf_walk == 2u) {
        f.l0_variant = 1;
        return;
    }
    if (ctx->l0_scene != f.scene->serial || ctx->l0_slots != p.n_slots || ctx->l0_mode != o.fp_mode) {
        ctx->l0_scene = f.scene->serial;
        ctx->l0_slots = p.n_slots;
        ctx->l0_mode = o.fp_mode;
        ctx->l0_ms[0] = ctx->l0_ms[1] = -1.0f;
        ctx->l0_pending[0] = ctx->l0_pending[1] = false;
        ++ctx->l0_epoch;
    }
    f.l0_epoch = ctx->l0_epoch;
    for (int v = 0; v < 2; ++v)
        if (ctx->l0_ms[v] < 0.0f && !ctx->l0_pending[v]) {
            f.l0_variant = v;
            f.l0_trial = true;
            ctx->l0_pending[v] = true;
            return;
        }
    f.l0_variant = (ctx->l0_ms[0] >= 0.0f && ctx->l0_ms[1] >= 0.0f && ctx->l0_ms[1] < ctx->l0_ms[0]) ? 1 : 0;
}

int enqueue_frame(rt_context* ctx, FrameSlot& f, const FramePlan& p, size_t cap, size_t scap) {
    const rt_scene* scene = f.scene;
    const rt_view* view = &f.view;
    const rt_options* opt = &f.opt;
    const uint32_t tile_count = opt->tile_count ? opt->tile_count : 1;
    const TileGrid grid(view->width, view->height, tile_count);
    const uint32_t n_slots = p.n_slots, n_lights = p.n_lights;
    // (room above the largest index: the work counters of the lean kernels run past their ends by 512 per wave)
    if (cap > 0xf0000000ull || scap * p.lights1 > 0xf0000000ull)
        return fail(ctx, RT_ERR_CAPACITY, "ray pool would exceed 2^32 entries");
    if (!f.fixed && ctx->workspace_limit != 0 &&
        (uint64_t)cap * NODE_BYTES + shadow_bytes(scap, scap * p.lights1) <= ctx->workspace_limit &&
        (uint64_t)f.ws.pool_cap * NODE_BYTES + shadow_bytes(f.ws.shadow_slots, f.ws.shadow_cap) > ctx->workspace_limit) {
        // The slot holds more than the limit allows -- pools grown for an earlier, larger frame, before
        // the limit was lowered -- and this frame fits it: the pools are given back and made anew.
        f.ray_o.release(); f.ray_d.release(); f.node_col.release(); f.node_kid.release();
        f.hit_a.release(); f.hit_p.release(); f.node_sh.release();
        f.sray_o.release(); f.sray_blocked.release();
        f.ws.pool_cap = 0;
        f.ws.shadow_slots = f.ws.shadow_cap = 0;
    }
    if (!f.fixed && ctx->workspace_limit != 0 &&
        (uint64_t)std::max(cap, f.ws.pool_cap) * NODE_BYTES + shadow_bytes(std::max(scap, f.ws.shadow_slots), std::max(scap * p.lights1, f.ws.shadow_cap)) >
            ctx->workspace_limit)
        return fail(ctx, RT_ERR_CAPACITY,
                    "this frame needs " + std::to_string((cap * NODE_BYTES + shadow_bytes(scap, scap * p.lights1)) >> 20) +
                        " MiB of ray pools, more than the workspace limit of " + std::to_string(ctx->workspace_limit >> 20) +
                        " MiB per frame slot (rt_context_set_workspace_limit, RTAMD_WORKSPACE_MAX_MB)");
    int rc = ensure_pool(ctx, f, cap);
    if (rc != RT_OK) return rc;
    rc = ensure_shadow(ctx, f, scap, p.lights1);
    if (rc != RT_OK) return rc;

    FrameArgs A;
    memset(&A, 0, sizeof(A));
    A.scene = scene->dev;
    if (opt->leaf_walk == 1u) A.scene.hit_walk = 0u;
    if (A.scene.hit_tables != 0u && A.scene.hit_walk_level0 == 0u && f.l0_variant == 1) A.scene.hit_walk_level0 = 2u;  // (2: the beam, or leaf by leaf)
    const Camera cam = camera_basis(*view);
    memcpy(A.from, view->from, 12);
    memcpy(A.fwd, cam.forward, 12);
    memcpy(A.up, cam.up, 12);
    memcpy(A.right, cam.right, 12);
    A.hither = view->hither;
    A.width = view->width;
    A.height = view->height;
    A.tiles_x = grid.tiles_x;
    A.tile_rank = opt->tile_rank;
    A.tile_count = tile_count;
    A.compact = (tile_count > 1 || opt->compact_tiles != 0) ? 1u : 0u;
    A.shadow_mode = opt->shadow_mode;
    A.n_slots = n_slots;
    A.max_depth = opt->max_depth;
    A.min_contribution = opt->min_contribution;
    A.out_f32 = (float*)f.out_f32;
    A.out_u8 = (uint8_t*)f.out_u8;
    const Workspace& ws = f.ws;
    A.ray_o = ws.ray_o;
    A.ray_d = ws.ray_d;
    A.hit_a = ws.hit_a;
    A.hit_p = ws.hit_p;
    A.node_sh = ws.node_sh;
    A.node_col = ws.node_col;
    A.node_kid = ws.node_kid;
    A.pool_cap = (uint32_t)ws.pool_cap;
    A.sray_o = ws.sray_o;
    A.sray_blocked = ws.sray_blocked;
    A.shadow_cap = (uint32_t)shadow_slots_of(ws.shadow_slots, ws.shadow_cap, p.lights1);
    A.level_count = ws.levels;
    A.level_base = ws.levels + RT_MAX_LEVELS;
    A.shadow_count = ws.levels + LV_SHADOW_COUNT;
    A.overflow = ws.levels + LV_OVERFLOW;
    A.work_next = ws.levels + LV_WORK_NEXT;
    A.deep_max = ws.levels + LV_DEEP_MAX;
    A.deep_exit_polls = ctx->deep_exit_polls;
    A.counters = ws.counters;

    // Levels that cannot hold a ray are not launched.  A ray of level k carries a contribution of
    // at most w^k, w the largest reflectance in the scene, when nothing transmits (a surface with
    // transmittance above the cutoff passes rays on at weight 1 from its back, src/scene.rs:151-165,
    // so then there is no bound); level k + 1 needs contribution * weight > min_contribution
    // (src/scene.rs:127-130, :166-167).  The bound is taken with room for the f32 products' rounding.
    uint32_t last_level = opt->max_depth;
    if (opt->min_contribution >= 0.0f) {
        double w = 0.0;
        bool bounded = true;
        for (const rt_surface& sf : scene->data.surfaces) {
            if (!(sf.reflectance >= 0.0f) || !(sf.reflectance < 1e30f) || !(sf.transmittance <= opt->min_contribution)) bounded = false;
            w = std::max(w, (double)sf.reflectance * (1.0 + 1e-6));
        }
        if (bounded) {
            uint32_t k = 0;
            double c = 1.0;  // bound on a level-k ray's contribution
            while (k < opt->max_depth && c * w > (double)opt->min_contribution) {
                c *= w;
                ++k;
            }
            last_level = k;
        }
    }
    const bool fast = opt->fp_mode == RT_FP_FAST;
    const bool work = opt->collect_work != 0;
    const bool cull = opt->exhaustive_leaves == 0;
    // Two streams per frame let a level's shadow rays run beside the next levels' intersections; HIP
    // gives a process four hardware queues, so that is the schedule for up to two frames in flight.
    // A caller that keeps three or four queued gets one stream per frame instead: the frames fill
    // each other's gaps, and more streams than queues would only take turns on them.
    const bool serial = opt->serial_schedule != 0;
    const bool one_stream = f.one_stream || serial;
    hipStream_t st = f.main_stream, st2 = one_stream ? f.main_stream : f.shadow_stream;
    f.timed_levels = 0;
    f.variant_deep = f.variant_shadow = -1;
    // What the caller's stream holds now comes before this attempt's writes to the caller's buffers.
    // (A re-run from rt_render_device_end records it afresh: frames begun in the meantime into the
    // same buffers, and whatever consumes them, are on that stream by then.)
    RT_HIP(ctx, hipEventRecord(f.ev_caller, f.caller));
    RT_HIP(ctx, hipMemsetAsync(ws.levels, 0, LV_WORDS * sizeof(uint32_t), st));
    RT_HIP(ctx, hipMemsetAsync(ws.counters, 0, N_COUNTERS * sizeof(unsigned long long), st));
    RT_HIP(ctx, hipEventRecord(f.ev[0], st));

    const uint32_t grid0 = std::max<uint32_t>(1, (n_slots + RT_BLOCK - 1) / RT_BLOCK);
    // Deeper levels' sizes are only known on the device; launch enough blocks for the
    // largest plausible level and let blocks past the end exit (grid-stride inside).
    const uint32_t gridn = std::min<uint32_t>(grid0, 16384);
    const uint32_t grids = std::min<uint32_t>(grid0, 65535);  // per light: the launch is (grids, n_lights)
    // With the sizes of the last frame of this scene at this size known, a level gets the blocks its
    // rays need (a quarter more, at least one block per CU) instead of the most it could need:
    // the blocks past a level's end only start, read the level's size and leave, but there are
    // tens of thousands of them per launch and they hold wave slots while they do.
    // All levels below the camera rays in one launch, where the kernels for that exist (lean scenes): the
    // frame is then level 0, that launch, and the shadow rays of both.  It shortens a frame (balls at
    // 3840x2160: the deeper levels 1.63 -> 1.13 ms, one frame at a time 5.21 -> 4.96 ms, an eighth of the
    // frame 1.40 -> 1.10 ms) at the price of blocks that stay while their rays dwindle; a frame begun with two
    // or more others in flight, whose gaps those fill anyway, keeps one launch per level (4.63 ms per frame
    // against 4.72 with four in flight).
    // (The per-level launches of the hit-list walk are one-wave blocks that take twice the time of its one
    // launch for all levels, rings 4: 15.3 against 8.0 ms: those scenes keep the one launch whatever is in flight.)
    const bool fused = !work && cull && ctx->fuse_deep && !(f.one_stream && !serial && A.scene.hit_walk == 0u) &&
                       (fast ? rt_fast::can_fuse(A, last_level) : rt_exact::can_fuse(A, last_level));
    const int hm = fused ? 1 : 0;  // which set of level sizes applies
    const bool hinted = ctx->hint_valid[hm] && ctx->hint_scene == scene->serial && ctx->hint_slots == n_slots && ctx->hint_depth == opt->max_depth;
    auto blocks_for = [&](uint64_t rays, uint32_t most) -> uint32_t {
        if (!hinted) return most;
        const uint64_t b = (rays + (rays >> 2) + RT_BLOCK - 1) / RT_BLOCK;
        return (uint32_t)std::min<uint64_t>(most, std::max<uint64_t>(b, 256));
    };
    auto shadow_rays = [&](uint32_t first, uint32_t last) -> int {
        uint64_t lit = 0;
        for (uint32_t l = first; l <= last; ++l) lit += ctx->hint_lit[hm][l];
        const uint32_t gs = blocks_for(lit, first == 0 ? grids : gridn);
        RT_HIP(ctx, hipStreamWaitEvent(st2, f.ev_level[last], 0));
        RT_HIP(ctx, fast ? rt_fast::launch_shadow(A, first, last, work, cull, gs, st2, &f.variant_shadow)
                         : rt_exact::launch_shadow(A, first, last, work, cull, gs, st2, &f.variant_shadow));
        return RT_OK;
    };
    bool lit_level0 = false;  // level 0's lighting is queued already
    A.fused = fused ? 1u : 0u;
    f.fused = fused;
    if (fused) {
        if (f.l0_trial) RT_HIP(ctx, hipEventRecord(f.ev_l0[0], st));
        RT_HIP(ctx, fast ? rt_fast::launch_intersect(A, 0, work, cull, grid0, st, nullptr) : rt_exact::launch_intersect(A, 0, work, cull, grid0, st, nullptr));
        if (f.l0_trial) RT_HIP(ctx, hipEventRecord(f.ev_l0[1], st));
        if (serial) RT_HIP(ctx, hipEventRecord(f.ev_t[0], st));
        if (n_lights) {
            RT_HIP(ctx, hipEventRecord(f.ev_level[0], st));
            if (!one_stream) {
                rc = shadow_rays(0, 0);
                if (rc != RT_OK) return rc;
                RT_HIP(ctx, fast ? rt_fast::launch_light(A, 0, 0, work, grid0, st2) : rt_exact::launch_light(A, 0, 0, work, grid0, st2));
                lit_level0 = true;
            }
        }
        // (blocks stay and take rays from the launch's counter: as many as the level-1 rays can use)
        const uint32_t gd = blocks_for(ctx->hint_rays[hm][1], gridn);
        RT_HIP(ctx, fast ? rt_fast::launch_deep(A, gd, 0u, st, &f.variant_deep) : rt_exact::launch_deep(A, gd, 0u, st, &f.variant_deep));
        if (serial) {
            RT_HIP(ctx, hipEventRecord(f.ev_t[1], st));
            f.timed_levels = 2;
        }
        if (n_lights) {
            RT_HIP(ctx, hipEventRecord(f.ev_level[last_level], st));
            rc = shadow_rays(one_stream ? 0 : 1, last_level);
            if (rc != RT_OK) return rc;
            if (serial) RT_HIP(ctx, hipEventRecord(f.ev_t[f.timed_levels], st));
        }
    }
    for (uint32_t level = 0; level <= last_level && !fused; ++level) {
        const uint32_t g = level == 0 ? grid0 : blocks_for(ctx->hint_rays[hm][level], gridn);
        int* variant = level > 0 ? &f.variant_deep : nullptr;
        if (level == 0 && f.l0_trial) RT_HIP(ctx, hipEventRecord(f.ev_l0[0], st));
        RT_HIP(ctx, fast ? rt_fast::launch_intersect(A, level, work, cull, g, st, variant) : rt_exact::launch_intersect(A, level, work, cull, g, st, variant));
        if (level == 0 && f.l0_trial) RT_HIP(ctx, hipEventRecord(f.ev_l0[1], st));
        if (serial) {
            RT_HIP(ctx, hipEventRecord(f.ev_t[level], st));
            f.timed_levels = level + 1;
        }
        if (n_lights) {  // this level's shadow rays and lighting: beside the next levels, on the second stream
            RT_HIP(ctx, hipEventRecord(f.ev_level[level], st));
            if (one_stream) {
                // one stream: the chain of intersections first, without a shadow launch between two
                // levels, then every level's shadow rays in one launch (the other frames in flight
                // fill the GPU under the small launches of the deep levels)
                if (level == last_level) {
                    rc = shadow_rays(0, level);
                    if (rc != RT_OK) return rc;
                    if (serial) RT_HIP(ctx, hipEventRecord(f.ev_t[f.timed_levels], st));
                }
            } else if (level < SHADOW_GROUP_FROM || level == last_level) {
                const uint32_t first = level < SHADOW_GROUP_FROM ? level : SHADOW_GROUP_FROM;
                rc = shadow_rays(first, level);
                if (rc != RT_OK) return rc;
                if (level == 0 && last_level > 0) {
                    // Two streams: the camera rays' hits (most of the frame's nodes) are lit as soon as their shadow
                    // rays are traced, beside the deeper levels -- the lighting kernel waits for HBM, the trace
                    // kernels for the vector ALUs.
                    RT_HIP(ctx, fast ? rt_fast::launch_light(A, 0, 0, work, grid0, st2) : rt_exact::launch_light(A, 0, 0, work, grid0, st2));
                    lit_level0 = true;
                }
            }
        }
    }
    if (n_lights) {
        {   // direct light of every node of the frame, once all shadow rays are traced: one launch, and
            // no level's shadow rays wait behind an earlier level's lighting
            uint64_t rays = lit_level0 ? 0u : n_slots;
            for (uint32_t l = 1; l <= last_level; ++l) rays += ctx->hint_rays[hm][l];
            const uint32_t gl = hinted ? blocks_for(rays, 0x7fffffffu) : grid0 + gridn;
            const uint32_t from = lit_level0 ? 1u : 0u;
            RT_HIP(ctx, fast ? rt_fast::launch_light(A, from, last_level, work, gl, st2)
                             : rt_exact::launch_light(A, from, last_level, work, gl, st2));
        }
        if (serial) RT_HIP(ctx, hipEventRecord(f.ev_t[f.timed_levels + 1], st));
        RT_HIP(ctx, hipEventRecord(f.ev_shadow, st2));
        RT_HIP(ctx, hipStreamWaitEvent(st, f.ev_shadow, 0));
    }
    RT_HIP(ctx, hipEventRecord(f.ev[1], st));
    for (uint32_t level = last_level + 1; level-- > 0;) {
        // (fused: hint_rays[2] is the size of the range all levels from 2 on share, which each of their launches goes over)
        const uint32_t g = level == 0 ? grid0 : blocks_for(ctx->hint_rays[hm][fused && level > 2 ? 2 : level], gridn);
        if (level == 0) RT_HIP(ctx, hipStreamWaitEvent(st, f.ev_caller, 0));  // the caller's buffers are written from here on
        RT_HIP(ctx, fast ? rt_fast::launch_resolve(A, level, g, st) : rt_exact::launch_resolve(A, level, g, st));
    }
    RT_HIP(ctx, hipEventRecord(f.ev[2], st));
    RT_HIP(ctx, hipStreamWaitEvent(f.caller, f.ev[2], 0));
    RT_HIP(ctx, hipMemcpyAsync(f.h_levels, ws.levels, LV_WORDS * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    RT_HIP(ctx, hipMemcpyAsync(f.h_counters, ws.counters, N_COUNTERS * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    RT_HIP(ctx, hipEventRecord(f.ev_done, st));
    return RT_OK;
}

// Waits for the frame of slot `f`, grows the pools and runs it again if they overflowed, and
// fills the statistics.  An attempt that overflowed has written nothing to the caller's buffers
// (the last kernel checks the flag).
int finish_frame(rt_context* ctx, FrameSlot& f, rt_stats* stats_out) {
    FramePlan p;
    int rc = plan_frame(ctx, f, p);
    if (rc != RT_OK) return rc;
    const rt_options* opt = &f.opt;
    rt_stats stats;
    memset(&stats, 0, sizeof(stats));
    uint32_t lv[LV_WORDS];
    for (uint32_t attempt = 0;; ++attempt) {
        RT_HIP(ctx, hipEventSynchronize(f.ev_done));
        memcpy(lv, f.h_levels, sizeof(lv));
        const size_t shadow_cap1 = shadow_slots_of(f.ws.shadow_slots, f.ws.shadow_cap, p.lights1);
        uint64_t need = p.n_slots, sneed = 0;
        for (uint32_t l = 0; l <= opt->max_depth; ++l) sneed += lv[LV_SHADOW_COUNT + l];
        for (uint32_t l = 1; l <= opt->max_depth + 1 && l < RT_MAX_LEVELS; ++l) need += lv[l];
        if (lv[LV_OVERFLOW] >= 0x100u)  // (k_deep: a wave waited longer than any walk can last and gave up)
            return fail(ctx, RT_ERR_DEVICE, "internal: the kernel of the deeper levels gave up a wait (" + std::to_string(lv[LV_OVERFLOW] - 0x100u) +
                                                "); nothing was written to the outputs");
        if (lv[LV_OVERFLOW] != 0) {  // a pool overflowed: grow and run the frame again
            if (f.fixed)
                return fail(ctx, RT_ERR_CAPACITY,
                            "the frame's rays do not fit the static workspace (" + std::to_string(need) + "+ ray-tree nodes of " +
                                std::to_string(f.ws.pool_cap) + ", " + std::to_string(sneed) + "+ lit hits of " +
                                std::to_string(shadow_cap1) + "); nothing was written to the outputs");
            size_t cap = f.ws.pool_cap, scap = shadow_cap1;
            if (need > f.ws.pool_cap) cap = std::max<size_t>(f.ws.pool_cap * 2, (size_t)need + (need >> 2));
            if (sneed > shadow_cap1) scap = std::max<size_t>(shadow_cap1 * 2, (size_t)sneed + (sneed >> 2));
            if (need <= f.ws.pool_cap && sneed <= shadow_cap1) { cap = f.ws.pool_cap * 2; scap = shadow_cap1 * 2; }
            stats.retries = attempt + 1;
            if (attempt >= 8) return fail(ctx, RT_ERR_CAPACITY, "ray pool still too small after 8 growth steps");
            rc = enqueue_frame(ctx, f, p, cap, scap);
            if (rc != RT_OK) return rc;
            continue;
        }
        // what the next frames of this kind should find in their pools, on every slot: this frame's need and
        // a quarter more (the largest over the finished frames of the kind; another kind starts afresh)
        if (ctx->hint_scene != f.scene->serial || ctx->hint_slots != p.n_slots || ctx->hint_depth != opt->max_depth)
            ctx->want_pool = ctx->want_shadow1 = 0;
        ctx->want_pool = std::max<size_t>(ctx->want_pool, (size_t)need + (need >> 2));
        ctx->want_shadow1 = std::max<size_t>(ctx->want_shadow1, (size_t)sneed + (sneed >> 2));
        unsigned long long c[N_COUNTERS];
        memcpy(c, f.h_counters, sizeof(c));
        stats.rays_primary = p.n_primary;
        stats.rays_shadow = 0;
        for (uint32_t l = 0; l <= opt->max_depth; ++l) stats.rays_shadow += (uint64_t)lv[LV_SHADOW_COUNT + l] * p.n_lights;
        stats.rays_reflect = c[2];
        stats.rays_refract = c[3];
        for (int k = 0; k < 16; ++k) {
            stats.work[k] = c[4 + k] + c[24 + k];
            stats.work_shadow[k] = c[24 + k];
        }
        if (ctx->hint_scene != f.scene->serial || ctx->hint_slots != p.n_slots || ctx->hint_depth != opt->max_depth)
            ctx->hint_valid[0] = ctx->hint_valid[1] = false;  // (another kind of frame: neither set of level sizes is its)
        ctx->hint_scene = f.scene->serial;
        ctx->hint_slots = p.n_slots;
        ctx->hint_depth = opt->max_depth;
        const int hm = f.fused ? 1 : 0;
        ctx->hint_valid[hm] = true;
        for (uint32_t l = 0; l < RT_MAX_LEVELS; ++l) {
            ctx->hint_rays[hm][l] = l <= opt->max_depth ? lv[l] : 0u;
            ctx->hint_lit[hm][l] = l <= opt->max_depth ? lv[LV_SHADOW_COUNT + l] : 0u;
        }
        if (f.l0_trial) {
            f.l0_trial = false;
            if (f.l0_epoch == ctx->l0_epoch) {
                float ms = 0;
                RT_HIP(ctx, hipEventElapsedTime(&ms, f.ev_l0[0], f.ev_l0[1]));
                ctx->l0_ms[f.l0_variant] = ms;
                ctx->l0_pending[f.l0_variant] = false;
            }
        }
        stats.kernel_shadow = f.variant_shadow;
        stats.kernel_deep = f.variant_deep;
        stats.workspace_bytes = (uint64_t)f.ws.pool_cap * NODE_BYTES + shadow_bytes(f.ws.shadow_slots, f.ws.shadow_cap);
        stats.ray_nodes = p.n_slots;
        stats.levels = 1;
        for (uint32_t l = 1; l <= opt->max_depth; ++l) {
            stats.ray_nodes += lv[l];
            if (lv[l]) stats.levels = l + 1;
        }
        if (f.fused && lv[1] != 0) stats.levels = std::max<uint32_t>(2, lv[LV_DEEP_MAX] + 1);  // (levels from 2 on share lv[2])
        float ms01 = 0, ms12 = 0, ms02 = 0;
        RT_HIP(ctx, hipEventElapsedTime(&ms01, f.ev[0], f.ev[1]));
        RT_HIP(ctx, hipEventElapsedTime(&ms12, f.ev[1], f.ev[2]));
        RT_HIP(ctx, hipEventElapsedTime(&ms02, f.ev[0], f.ev[2]));
        stats.ms_trace = ms01;
        stats.ms_resolve = ms12;
        stats.ms_frame = ms02;
        if (f.timed_levels != 0) {  // serial schedule: the kernels' own times
            float ms = 0;
            RT_HIP(ctx, hipEventElapsedTime(&ms, f.ev[0], f.ev_t[0]));
            stats.ms_intersect0 = ms;
            if (f.timed_levels > 1) {
                RT_HIP(ctx, hipEventElapsedTime(&ms, f.ev_t[0], f.ev_t[f.timed_levels - 1]));
                stats.ms_intersect_deep = ms;
            }
            if (p.n_lights) {
                RT_HIP(ctx, hipEventElapsedTime(&ms, f.ev_t[f.timed_levels - 1], f.ev_t[f.timed_levels]));
                stats.ms_shadow = ms;
                RT_HIP(ctx, hipEventElapsedTime(&ms, f.ev_t[f.timed_levels], f.ev_t[f.timed_levels + 1]));
                stats.ms_light = ms;
            }
        }
        break;
    }
    if (stats_out) *stats_out = stats;
    return RT_OK;
}

// Argument checks shared by the entry points that start a frame.
int check_frame_args(rt_context* ctx, const rt_scene* scene, const rt_view* view, const rt_options* opt) {
    if (!ctx || !scene || !view || !opt) return fail(ctx, RT_ERR_INVALID, "null argument");
    if (scene->ctx != ctx) return fail(ctx, RT_ERR_INVALID, "scene belongs to another context");
    if (view->width == 0 || view->height == 0) return fail(ctx, RT_ERR_INVALID, "empty view");
    if ((uint64_t)view->width * view->height > 0x7fffffffull / 3) return fail(ctx, RT_ERR_UNSUPPORTED, "image too large");
    if (opt->max_depth + 2 > RT_MAX_LEVELS) return fail(ctx, RT_ERR_UNSUPPORTED, "max_depth too large");
    if (opt->fp_mode > RT_FP_FAST) return fail(ctx, RT_ERR_INVALID, "unknown fp_mode");
    if (opt->shadow_mode > RT_SHADOW_ANY_HIT) return fail(ctx, RT_ERR_INVALID, "unknown shadow_mode");
    const uint32_t tile_count = opt->tile_count ? opt->tile_count : 1;
    if (opt->tile_rank >= tile_count) return fail(ctx, RT_ERR_INVALID, "tile_rank >= tile_count");
    if (scene->dev.n_lights > 65535) return fail(ctx, RT_ERR_UNSUPPORTED, "more than 65535 lights");
    return RT_OK;
}

}  // namespace

namespace {

// A frame of unproven size, queued by begin_frame: waited for and re-run until its pools hold it.
int settle_frame(rt_context* ctx, uint32_t id) {
    FrameSlot& f = ctx->slot[id];
    if (!f.busy || !f.unsettled) return RT_OK;
    f.unsettled = false;
    RT_HIP(ctx, hipSetDevice(ctx->device));
    f.finish_rc = finish_frame(ctx, f, &f.stats);
    f.finished = true;
    if (f.finish_rc != RT_OK) f.busy = false;
    return f.finish_rc;
}

// Ends the frame of slot `id` whatever its state (error paths of a frame over several devices).
int begin_end_quietly(rt_context* ctx, uint32_t id) {
    FrameSlot& f = ctx->slot[id];
    if (!f.busy) return RT_OK;
    (void)hipSetDevice(ctx->device);
    int rc = RT_OK;
    if (!f.finished) rc = finish_frame(ctx, f, nullptr);
    f.busy = false;
    f.unsettled = false;
    return rc;
}

// (rt_render_device_begin, also for one rank of a context over several devices)
// slot < 0: the context's next slot, which then moves on; else that slot (a rank of a context over several
// devices follows the root's ticket).  settle_now false: a frame of unproven size is only queued, and the
// caller settles it (settle_frame) once it has queued the other ranks' frames as well.
int begin_frame(rt_context* ctx, const rt_scene* scene, const rt_view* view, const rt_options* opt,
                                      void* d_rgb_f32_out, void* d_rgb_u8_out, void* stream, uint32_t* ticket_out,
                                      int slot = -1, bool settle_now = true) {
    if (!ticket_out) return fail(ctx, RT_ERR_INVALID, "null argument");
    int rc = check_frame_args(ctx, scene, view, opt);
    if (rc != RT_OK) return rc;
    RT_HIP(ctx, hipSetDevice(ctx->device));
    rc = prepare_kernels(ctx);
    if (rc != RT_OK) return rc;
    const uint32_t id = slot < 0 ? ctx->next_slot : (uint32_t)slot;
    FrameSlot& f = ctx->slot[id];
    if (f.busy) return fail(ctx, RT_ERR_INVALID, "every frame slot is in flight already: call rt_render_device_end first");
    {
        uint32_t busy = 0;
        for (uint32_t k = 0; k < ctx->n_slots; ++k) busy += ctx->slot[k].busy ? 1u : 0u;
        f.one_stream = busy >= 2;
    }
    f.scene = scene;
    f.view = *view;
    f.opt = *opt;
    f.out_f32 = d_rgb_f32_out;
    f.out_u8 = d_rgb_u8_out;
    f.caller = (hipStream_t)stream;
    FramePlan p;
    rc = plan_frame(ctx, f, p);
    if (rc != RT_OK) return rc;
    // Pools: what this slot has, what finished frames on any slot needed (and a quarter more), and
    // room for the camera rays and as many again.  Shadow queues hold the whole frame's lit hits,
    // times the number of lights.
    // A frame like the last finished one (same scene, image share and depth) is known to fit pools of
    // that size, or to come close.
    choose_level0(ctx, f, p);
    // (a frame whose camera rays are timed runs alone: it is waited for like one of unproven size)
    const bool known = ctx->hint_scene == scene->serial && ctx->hint_slots == p.n_slots && ctx->hint_depth == opt->max_depth;
    const bool proven = known && !f.l0_trial;
    const size_t floor_cap = std::max<size_t>((size_t)p.n_slots * 2, 1u << 16);
    // (The wanted sizes carry a quarter of headroom and creep up with every frame that needs a little more than
    // any before -- a camera going round a scene.  A slot that holds all but an eighth of them keeps its pools:
    // growing them means freeing them, which waits for everything in flight on the device.  lathe 96 at 1920x1080,
    // twenty views round the scene with four frames in flight: 3.6 -> 1.8 ms per frame.)
    auto wanted = [&](size_t have, size_t want) { return known && have < want - want / 8 ? want : have; };
    const size_t cap = std::max(wanted(f.ws.pool_cap, ctx->want_pool), floor_cap);
    const size_t scap = std::max(wanted(shadow_slots_of(f.ws.shadow_slots, f.ws.shadow_cap, p.lights1), ctx->want_shadow1), floor_cap);
    // Any other frame is waited for here, pools grown and the frame re-run as often as it takes, so that
    // what the caller queues on `stream` after this call finds the finished image.
    rc = enqueue_frame(ctx, f, p, cap, scap);
    if (rc != RT_OK) return rc;
    f.busy = true;
    f.finished = false;
    f.unsettled = !proven;
    if (settle_now) {
        rc = settle_frame(ctx, id);
        if (rc != RT_OK) return rc;
    }
    if (slot < 0) ctx->next_slot = (id + 1u) % ctx->n_slots;
    *ticket_out = id;
    return RT_OK;
}

}  // namespace

namespace {
int multi_begin(rt_context* ctx, const rt_scene* scene, const rt_view* view, const rt_options* opt, void* d_f32, void* d_u8,
                void* stream, uint32_t* ticket_out);
int multi_end(rt_context* ctx, uint32_t ticket, rt_stats* stats_out);
}  // namespace

extern "C" int rt_render_device_begin(rt_context* ctx, const rt_scene* scene, const rt_view* view, const rt_options* opt,
                                      void* d_rgb_f32_out, void* d_rgb_u8_out, void* stream, uint32_t* ticket_out) {
    if (ctx && !ctx->peers.empty()) return multi_begin(ctx, scene, view, opt, d_rgb_f32_out, d_rgb_u8_out, stream, ticket_out);
    return begin_frame(ctx, scene, view, opt, d_rgb_f32_out, d_rgb_u8_out, stream, ticket_out);
}

extern "C" int rt_render_device_end(rt_context* ctx, uint32_t ticket, rt_stats* stats_out) {
    if (!ctx || ticket >= ctx->n_slots) return fail(ctx, RT_ERR_INVALID, "bad ticket");
    if (!ctx->peers.empty()) return multi_end(ctx, ticket, stats_out);
    FrameSlot& f = ctx->slot[ticket];
    if (!f.busy) return fail(ctx, RT_ERR_INVALID, "no frame in flight under this ticket");
    RT_HIP(ctx, hipSetDevice(ctx->device));
    int rc = RT_OK;
    if (f.finished) {
        if (stats_out) *stats_out = f.stats;
    } else {
        rc = finish_frame(ctx, f, stats_out);
    }
    f.busy = false;
    return rc;
}

extern "C" int rt_context_set_workspace_limit(rt_context* ctx, uint64_t bytes_per_frame_slot) {
    if (!ctx) return RT_ERR_INVALID;
    ctx->workspace_limit = bytes_per_frame_slot;
    return RT_OK;
}

// The same frame as rt_render_device with every byte of device workspace taken from `scratch`:
// no hipMalloc, hipFree or other allocating call, here or below.  The array is shared by every
// call on this GPU and keeps the last call's contents; the frame clears the counters it needs at
// zero itself (enqueue_frame), everything else is written before it is read.
extern "C" int rt_render_device_static(rt_context* ctx, const rt_scene* scene, const rt_view* view, const rt_options* opt,
                                       void* d_rgb_f32_out, void* d_rgb_u8_out, void* stream, rt_stats* stats_out) {
    int rc = check_frame_args(ctx, scene, view, opt);
    if (rc != RT_OK) return rc;
    if (!ctx->peers.empty()) return fail(ctx, RT_ERR_UNSUPPORTED, "the static workspace is for one-device contexts");
    RT_HIP(ctx, hipSetDevice(ctx->device));
    FrameSlot& f = ctx->fixed_slot;
    if (f.busy || !ctx->scratch_ptr || !ctx->kernels_ready) return fail(ctx, RT_ERR_INVALID, "context not ready");
    f.scene = scene;
    f.view = *view;
    f.opt = *opt;
    f.out_f32 = d_rgb_f32_out;
    f.out_u8 = d_rgb_u8_out;
    f.caller = (hipStream_t)stream;
    f.one_stream = false;
    FramePlan p;
    rc = plan_frame(ctx, f, p);
    if (rc != RT_OK) return rc;
    // Shares of the array: counters first, then the node arrays and the shadow queues, all for the
    // same number of entries (a frame has at most one lit hit per ray-tree node).  That number is a
    // multiple of 64, which keeps every array on a 256-byte boundary.
    constexpr size_t HEAD = 4096;
    static_assert(LV_WORDS * sizeof(uint32_t) <= 2048 && N_COUNTERS * sizeof(unsigned long long) <= 2048, "counters fit the head");
    const size_t per_entry = NODE_BYTES + SHADOW_SLOT_BYTES + SHADOW_RAY_BYTES * p.lights1;
    const size_t cap = ((RT_STATIC_WORKSPACE_BYTES - HEAD) / per_entry) & ~(size_t)63;
    if (cap < p.n_slots || cap < 64)
        return fail(ctx, RT_ERR_CAPACITY,
                    "frame too large for the static workspace: " + std::to_string(p.n_slots) + " camera rays, room for " +
                        std::to_string(cap) + " ray-tree nodes with " + std::to_string(p.n_lights) + " lights in " +
                        std::to_string(RT_STATIC_WORKSPACE_BYTES >> 20) + " MiB");
    unsigned char* at = ctx->scratch_ptr;
    Workspace& ws = f.ws;
    ws.levels = (uint32_t*)at;
    ws.counters = (unsigned long long*)(at + 2048);
    at += HEAD;
    auto take = [&](size_t bytes) { unsigned char* q = at; at += bytes; return q; };
    ws.ray_o = (float4*)take(cap * sizeof(float4));
    ws.ray_d = (float4*)take(cap * sizeof(float4));
    ws.node_col = (float4*)take(cap * sizeof(float4));
    ws.hit_a = (float4*)take(cap * sizeof(float4));
    ws.node_kid = (uint4*)take(cap * sizeof(uint4));
    ws.node_sh = (uint2*)take(cap * sizeof(uint2));
    ws.hit_p = (uint32_t*)take(cap * sizeof(uint32_t));
    ws.sray_o = (float4*)take(cap * sizeof(float4));
    ws.sray_blocked = (uint32_t*)take(cap * p.lights1 * sizeof(uint32_t));
    ws.pool_cap = cap;
    ws.shadow_slots = cap;
    ws.shadow_cap = cap * p.lights1;
    if ((size_t)(at - ctx->scratch_ptr) > RT_STATIC_WORKSPACE_BYTES) return fail(ctx, RT_ERR_CAPACITY, "static workspace layout");
    choose_level0(ctx, f, p);
    rc = enqueue_frame(ctx, f, p, cap, cap);
    if (rc != RT_OK) return rc;
    f.busy = true;
    rc = finish_frame(ctx, f, stats_out);
    f.busy = false;
    return rc;
}

extern "C" uint32_t rt_frames_in_flight(const rt_context* ctx) { return ctx ? ctx->n_slots : 0; }

namespace {

// A frame on a context over several devices: every rank (device) traces its interleaved 8x8 tiles into
// a compact buffer (pixels are independent, src/render.rs:104-112; every device holds the whole scene),
// ranks 1.. copy theirs into the root device's gathered buffer (hipMemcpyPeerAsync on the rank's own
// stream: over the GPUs' links where peer access is enabled, rt_context_peer_access), and the root
// de-interleaves into the caller's image on the caller's stream.  The frame of ticket t uses slot t of
// every rank's context, with tile and gathered buffers of its own, so up to rt_frames_in_flight frames
// may be queued.  Every rank's share is queued before any is waited for.
int multi_begin(rt_context* ctx, const rt_scene* scene, const rt_view* view, const rt_options* opt, void* d_f32, void* d_u8,
                void* stream, uint32_t* ticket_out) {
    if (!ticket_out) return fail(ctx, RT_ERR_INVALID, "null argument");
    int rc = check_frame_args(ctx, scene, view, opt);
    if (rc != RT_OK) return rc;
    const uint32_t n = 1u + (uint32_t)ctx->peers.size();
    if ((opt->tile_count > 1) || opt->compact_tiles || opt->collect_work)
        return fail(ctx, RT_ERR_UNSUPPORTED, "a context over several devices tiles the image itself and keeps no work counters");
    if (scene->peer_scenes.size() + 1 != n) return fail(ctx, RT_ERR_INVALID, "scene was not created on this context");
    const uint32_t id = ctx->next_slot;
    auto rank_ctx = [&](uint32_t r) { return r == 0 ? ctx : ctx->peers[r - 1]; };
    for (uint32_t r = 0; r < n; ++r)
        if (rank_ctx(r)->slot[id].busy)
            return fail(ctx, RT_ERR_INVALID, "every frame slot is in flight already: call rt_render_device_end first");
    const uint32_t W = view->width, H = view->height;
    const size_t stride = (size_t)rt_tile_pixels_max(W, H, n);
    RT_HIP(ctx, hipSetDevice(ctx->device));
    FrameSlot& root = ctx->slot[id];
    if (d_f32 && root.gathered_f32.n < stride * n * 3) RT_HIP(ctx, root.gathered_f32.alloc(stride * n * 3));
    if (d_u8 && root.gathered_u8.n < stride * n * 3) RT_HIP(ctx, root.gathered_u8.alloc(stride * n * 3));
    uint32_t begun = 0;
    std::string err;
    // 1. every rank's share on its own device's streams; nothing is waited for
    for (uint32_t r = 0; r < n && rc == RT_OK; ++r) {
        rt_context* c = rank_ctx(r);
        FrameSlot& fs = c->slot[id];
        const rt_scene* sc = r == 0 ? scene : scene->peer_scenes[r - 1];
        hipError_t e = hipSetDevice(c->device);
        float* pf = nullptr;
        uint8_t* pu = nullptr;
        if (r == 0) {  // the root's share goes straight into its place
            pf = d_f32 ? root.gathered_f32.p : nullptr;
            pu = d_u8 ? root.gathered_u8.p : nullptr;
        } else {
            if (e == hipSuccess && d_f32 && fs.tile_f32.n < stride * 3) e = fs.tile_f32.alloc(stride * 3);
            if (e == hipSuccess && d_u8 && fs.tile_u8.n < stride * 3) e = fs.tile_u8.alloc(stride * 3);
            pf = d_f32 ? fs.tile_f32.p : nullptr;
            pu = d_u8 ? fs.tile_u8.p : nullptr;
        }
        if (e != hipSuccess) { rc = RT_ERR_DEVICE; err = hipGetErrorString(e); break; }
        rt_options o = *opt;
        o.tile_rank = r;
        o.tile_count = n;
        uint32_t t = 0;
        rc = begin_frame(c, sc, view, &o, pf, pu, fs.gather_stream, &t, (int)id, false);
        if (rc != RT_OK) { err = "device " + std::to_string(c->device) + ": " + c->err; break; }
        ++begun;
    }
    // 2. shares of unproven size (the first frame of a scene, size or depth) are waited for and re-run
    // until their pools hold them, now that every device has work
    for (uint32_t r = 0; r < begun && rc == RT_OK; ++r) {
        rt_context* c = rank_ctx(r);
        rc = settle_frame(c, id);
        if (rc != RT_OK) err = "device " + std::to_string(c->device) + ": " + c->err;
    }
    // 3. the copies to the root, each behind its rank's share (the share's last kernel is what the rank's
    // stream waits for), and the event that ends share and copy
    for (uint32_t r = 0; r < begun && rc == RT_OK; ++r) {
        rt_context* c = rank_ctx(r);
        FrameSlot& fs = c->slot[id];
        hipError_t e = hipSetDevice(c->device);
        const size_t mine = (size_t)rt_tile_pixels(W, H, r, n) * 3;
        if (r > 0) {
            if (e == hipSuccess && d_f32)
                e = hipMemcpyPeerAsync(root.gathered_f32.p + (size_t)r * stride * 3, ctx->device, fs.tile_f32.p, c->device,
                                       mine * sizeof(float), fs.gather_stream);
            if (e == hipSuccess && d_u8)
                e = hipMemcpyPeerAsync(root.gathered_u8.p + (size_t)r * stride * 3, ctx->device, fs.tile_u8.p, c->device, mine, fs.gather_stream);
        }
        if (e == hipSuccess) e = hipEventRecord(fs.ev_gather, fs.gather_stream);
        if (e != hipSuccess) { rc = RT_ERR_DEVICE; err = hipGetErrorString(e); }
    }
    (void)hipSetDevice(ctx->device);
    // 4. the image, on the caller's stream, once every share has arrived
    if (rc == RT_OK) {
        hipError_t e = hipSuccess;
        for (uint32_t r = 0; r < n && e == hipSuccess; ++r) e = hipStreamWaitEvent((hipStream_t)stream, rank_ctx(r)->slot[id].ev_gather, 0);
        if (e != hipSuccess) { rc = RT_ERR_DEVICE; err = hipGetErrorString(e); }
        if (rc == RT_OK && d_f32) rc = rt_deinterleave_device(ctx, W, H, n, stride, 4, root.gathered_f32.p, d_f32, stream);
        if (rc == RT_OK && d_u8) rc = rt_deinterleave_device(ctx, W, H, n, stride, 1, root.gathered_u8.p, d_u8, stream);
        if (rc != RT_OK && err.empty()) err = ctx->err;
    }
    if (rc != RT_OK) {  // every share that was begun is ended, whatever happened since
        for (uint32_t r = 0; r < begun; ++r) {
            rt_context* c = rank_ctx(r);
            if (c->slot[id].busy) (void)begin_end_quietly(c, id);
        }
        (void)hipSetDevice(ctx->device);
        return fail(ctx, rc, err);
    }
    root.multi = true;
    root.multi_f32 = d_f32;
    root.multi_u8 = d_u8;
    root.multi_stream = (hipStream_t)stream;
    ctx->next_slot = (id + 1u) % ctx->n_slots;
    *ticket_out = id;
    return RT_OK;
}

int multi_end(rt_context* ctx, uint32_t ticket, rt_stats* stats_out) {
    FrameSlot& root = ctx->slot[ticket];
    if (!root.multi) return fail(ctx, RT_ERR_INVALID, "no frame in flight under this ticket");
    root.multi = false;
    const uint32_t n = 1u + (uint32_t)ctx->peers.size();
    auto rank_ctx = [&](uint32_t r) { return r == 0 ? ctx : ctx->peers[r - 1]; };
    const uint32_t W = root.view.width, H = root.view.height;
    const size_t stride = (size_t)rt_tile_pixels_max(W, H, n);
    void* d_f32 = root.multi_f32;
    void* d_u8 = root.multi_u8;
    hipStream_t stream = root.multi_stream;
    int rc = RT_OK;
    std::string err;
    rt_stats total;
    memset(&total, 0, sizeof(total));
    bool again = false;  // some share was run again after its copy had been queued
    for (uint32_t r = 0; r < n; ++r) {
        rt_context* c = rank_ctx(r);
        FrameSlot& fs = c->slot[ticket];
        if (!fs.busy) continue;
        (void)hipSetDevice(c->device);
        rt_stats st;
        memset(&st, 0, sizeof(st));
        int rc_end = RT_OK;
        if (fs.finished) st = fs.stats;
        else rc_end = finish_frame(c, fs, &st);
        fs.busy = false;
        if (rc_end != RT_OK) {
            if (rc == RT_OK) { rc = rc_end; err = "device " + std::to_string(c->device) + ": " + c->err; }
            continue;
        }
        if (st.retries > 0 && !fs.finished) {
            again = true;
            if (r > 0 && rc == RT_OK) {  // (the re-run is complete: finish_frame waited for it)
                const size_t mine = (size_t)rt_tile_pixels(W, H, r, n) * 3;
                hipError_t e = hipSuccess;
                if (d_f32) e = hipMemcpyPeer(root.gathered_f32.p + (size_t)r * stride * 3, ctx->device, fs.tile_f32.p, c->device, mine * sizeof(float));
                if (e == hipSuccess && d_u8) e = hipMemcpyPeer(root.gathered_u8.p + (size_t)r * stride * 3, ctx->device, fs.tile_u8.p, c->device, mine);
                if (e != hipSuccess) { rc = RT_ERR_DEVICE; err = hipGetErrorString(e); }
            }
        }
        total.rays_primary += st.rays_primary; total.rays_shadow += st.rays_shadow;
        total.rays_reflect += st.rays_reflect; total.rays_refract += st.rays_refract;
        total.ray_nodes += st.ray_nodes;
        total.workspace_bytes += st.workspace_bytes;
        total.kernel_shadow = st.kernel_shadow;
        total.kernel_deep = st.kernel_deep;
        total.levels = std::max(total.levels, st.levels);
        total.retries = std::max(total.retries, st.retries);
        total.ms_trace = std::max(total.ms_trace, st.ms_trace);
        total.ms_resolve = std::max(total.ms_resolve, st.ms_resolve);
        total.ms_frame = std::max(total.ms_frame, st.ms_frame);
    }
    (void)hipSetDevice(ctx->device);
    if (rc == RT_OK && again) {  // the image once more, from the shares as they are now
        if (d_f32) rc = rt_deinterleave_device(ctx, W, H, n, stride, 4, root.gathered_f32.p, d_f32, stream);
        if (rc == RT_OK && d_u8) rc = rt_deinterleave_device(ctx, W, H, n, stride, 1, root.gathered_u8.p, d_u8, stream);
        if (rc != RT_OK) err = ctx->err;
    }
    if (rc != RT_OK) return fail(ctx, rc, err);
    if (stats_out) *stats_out = total;
    return RT_OK;
}

}  // namespace

extern "C" int rt_render_device(rt_context* ctx, const rt_scene* scene, const rt_view* view, const rt_options* opt,
                                void* d_rgb_f32_out, void* d_rgb_u8_out, void* stream, rt_stats* stats_out) {
    uint32_t ticket = 0;
    int rc = rt_render_device_begin(ctx, scene, view, opt, d_rgb_f32_out, d_rgb_u8_out, stream, &ticket);
    if (rc != RT_OK) return rc;
    rc = rt_render_device_end(ctx, ticket, stats_out);
    // (several devices: the image is put together by a kernel on `stream`, after the last share's copy)
    if (rc == RT_OK && !ctx->peers.empty()) RT_HIP(ctx, hipStreamSynchronize((hipStream_t)stream));
    return rc;
}

extern "C" int rt_render(rt_context* ctx, const rt_scene* scene, const rt_view* view, const rt_options* opt,
                         float* rgb_f32_out, uint8_t* rgb_u8_out, rt_stats* stats_out) {
    if (!ctx || !view || !opt) return fail(ctx, RT_ERR_INVALID, "null argument");
    const uint32_t tile_count = opt->tile_count ? opt->tile_count : 1;
    if (opt->tile_rank >= tile_count) return fail(ctx, RT_ERR_INVALID, "tile_rank >= tile_count");
    RT_HIP(ctx, hipSetDevice(ctx->device));
    const bool compact = tile_count > 1 || opt->compact_tiles != 0;
    const size_t px = !compact ? (size_t)view->width * view->height
                               : (size_t)rt_tile_pixels(view->width, view->height, opt->tile_rank, tile_count);
    if (rgb_f32_out && ctx->out_f32.n < px * 3) RT_HIP(ctx, ctx->out_f32.alloc(px * 3));
    if (rgb_u8_out && ctx->out_u8.n < px * 3) RT_HIP(ctx, ctx->out_u8.alloc(px * 3));
    // padding slots of edge tiles are never written by the kernels: clear them
    if (rgb_f32_out && compact) RT_HIP(ctx, hipMemset(ctx->out_f32.p, 0, px * 3 * sizeof(float)));
    if (rgb_u8_out && compact) RT_HIP(ctx, hipMemset(ctx->out_u8.p, 0, px * 3));
    rt_stats stats;
    int rc = rt_render_device(ctx, scene, view, opt, rgb_f32_out ? ctx->out_f32.p : nullptr,
                              rgb_u8_out ? ctx->out_u8.p : nullptr, nullptr, &stats);
    if (rc != RT_OK) return rc;
    auto t0 = std::chrono::steady_clock::now();
    if (rgb_f32_out) RT_HIP(ctx, hipMemcpy(rgb_f32_out, ctx->out_f32.p, px * 3 * sizeof(float), hipMemcpyDeviceToHost));
    if (rgb_u8_out) RT_HIP(ctx, hipMemcpy(rgb_u8_out, ctx->out_u8.p, px * 3, hipMemcpyDeviceToHost));
    stats.ms_d2h = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (stats_out) *stats_out = stats;
    return RT_OK;
}

extern "C" int rt_deinterleave_device(rt_context* ctx, uint32_t width, uint32_t height, uint32_t count,
                                      uint64_t stride_pixels, uint32_t elem_size, const void* d_gathered,
                                      void* d_image, void* stream) {
    if (!ctx || !d_gathered || !d_image) return fail(ctx, RT_ERR_INVALID, "null argument");
    if (elem_size != 1 && elem_size != 4) return fail(ctx, RT_ERR_INVALID, "elem_size must be 1 or 4");
    if (width == 0 || height == 0) return fail(ctx, RT_ERR_INVALID, "empty image");
    const TileGrid g(width, height, count);
    if (stride_pixels < g.tiles_of(0) * 64) return fail(ctx, RT_ERR_INVALID, "stride_pixels smaller than a rank's tile buffer");
    RT_HIP(ctx, hipSetDevice(ctx->device));
    DeinterleaveArgs D;
    D.width = width; D.height = height; D.tiles_x = g.tiles_x; D.count = g.count; D.elem_size = elem_size;
    D.stride_pixels = stride_pixels;
    D.gathered = d_gathered;
    D.image = d_image;
    RT_HIP(ctx, rt_exact::launch_deinterleave(D, (hipStream_t)stream));
    return RT_OK;
}

extern "C" int rt_debug_intersect(rt_context* ctx, const rt_scene* scene, uint32_t fp_mode_flags, uint64_t n,
                                  const float* src, const float* ray, const float* near, const uint32_t* ignore,
                                  uint32_t* out_prim, float* out_dist, float* out_normal) {
    if (!ctx || !scene || !src || !ray || !near || !ignore || !out_prim || !out_dist || !out_normal)
        return fail(ctx, RT_ERR_INVALID, "null argument");
    const uint32_t fp_mode = fp_mode_flags & ~(RT_DEBUG_EXHAUSTIVE | RT_DEBUG_LEAF_WALK);
    if (fp_mode > RT_FP_FAST) return fail(ctx, RT_ERR_INVALID, "unknown fp_mode");
    if (n == 0) return RT_OK;
    DevScene dev = scene->dev;
    if ((fp_mode_flags & RT_DEBUG_LEAF_WALK) != 0u) dev.hit_walk = 0u;
    for (uint64_t i = 0; i < n; ++i)
        if (ignore[i] != 0xffffffffu && ignore[i] >= scene->data.primitives.size())
            return fail(ctx, RT_ERR_INVALID, "ignore index out of range");
    RT_HIP(ctx, hipSetDevice(ctx->device));
    int rc = prepare_kernels(ctx);
    if (rc != RT_OK) return rc;
    DevBuf<float> d_src, d_ray, d_near, d_dist, d_nrm;
    DevBuf<uint32_t> d_ign, d_prim;
    RT_HIP(ctx, d_src.alloc(n * 3));
    RT_HIP(ctx, d_ray.alloc(n * 3));
    RT_HIP(ctx, d_near.alloc(n));
    RT_HIP(ctx, d_ign.alloc(n));
    RT_HIP(ctx, d_prim.alloc(n));
    RT_HIP(ctx, d_dist.alloc(n));
    RT_HIP(ctx, d_nrm.alloc(n * 3));
    RT_HIP(ctx, hipMemcpy(d_src.p, src, n * 12, hipMemcpyHostToDevice));
    RT_HIP(ctx, hipMemcpy(d_ray.p, ray, n * 12, hipMemcpyHostToDevice));
    RT_HIP(ctx, hipMemcpy(d_near.p, near, n * 4, hipMemcpyHostToDevice));
    RT_HIP(ctx, hipMemcpy(d_ign.p, ignore, n * 4, hipMemcpyHostToDevice));
    if (fp_mode == RT_FP_FAST)
        RT_HIP(ctx, rt_fast::launch_debug_intersect(dev, (fp_mode_flags & RT_DEBUG_EXHAUSTIVE) == 0, n, d_src.p, d_ray.p, d_near.p, d_ign.p, d_prim.p, d_dist.p, d_nrm.p, nullptr));
    else
        RT_HIP(ctx, rt_exact::launch_debug_intersect(dev, (fp_mode_flags & RT_DEBUG_EXHAUSTIVE) == 0, n, d_src.p, d_ray.p, d_near.p, d_ign.p, d_prim.p, d_dist.p, d_nrm.p, nullptr));
    RT_HIP(ctx, hipDeviceSynchronize());
    RT_HIP(ctx, hipMemcpy(out_prim, d_prim.p, n * 4, hipMemcpyDeviceToHost));
    RT_HIP(ctx, hipMemcpy(out_dist, d_dist.p, n * 4, hipMemcpyDeviceToHost));
    RT_HIP(ctx, hipMemcpy(out_normal, d_nrm.p, n * 12, hipMemcpyDeviceToHost));
    return RT_OK;
}

extern "C" int rt_debug_eval(rt_context* ctx, uint32_t op, uint32_t fp_mode, uint64_t n, const float* in,
                             uint32_t in_stride, float* out, uint32_t out_stride, const rt_vertex* verts,
                             uint32_t n_verts) {
    static const uint32_t need_in[10] = {13, 11, 15, 7, 21, 21, 6, 7, 2, 1};
    static const uint32_t need_out[10] = {1, 5, 5, 5, 3, 3, 3, 3, 1, 1};
    if (!ctx || !in || !out) return fail(ctx, RT_ERR_INVALID, "null argument");
    if (op > RT_OP_QUANTIZE || fp_mode > RT_FP_FAST) return fail(ctx, RT_ERR_INVALID, "unknown op or fp_mode");
    if (in_stride < need_in[op] || out_stride < need_out[op]) return fail(ctx, RT_ERR_INVALID, "stride too small for op");
    if (op == RT_OP_POLYGON && (!verts || n_verts < 3)) return fail(ctx, RT_ERR_INVALID, "polygon needs >= 3 vertices");
    if (n == 0) return RT_OK;
    RT_HIP(ctx, hipSetDevice(ctx->device));
    DevBuf<float> d_in, d_out;
    DevBuf<float4> d_vp, d_vn;
    RT_HIP(ctx, d_in.alloc(n * in_stride));
    RT_HIP(ctx, d_out.alloc(n * out_stride));
    RT_HIP(ctx, hipMemcpy(d_in.p, in, n * in_stride * 4, hipMemcpyHostToDevice));
    RT_HIP(ctx, hipMemset(d_out.p, 0, n * out_stride * 4));
    std::vector<float4> vp, vn;
    for (uint32_t k = 0; k < n_verts && verts; ++k) {
        vp.push_back(f4(verts[k].point, 0.0f));
        vn.push_back(f4(verts[k].normal, 0.0f));
    }
    RT_HIP(ctx, d_vp.upload(vp));
    RT_HIP(ctx, d_vn.upload(vn));
    if (fp_mode == RT_FP_FAST)
        RT_HIP(ctx, rt_fast::launch_debug_eval(op, n, d_in.p, in_stride, d_out.p, out_stride, d_vp.p, d_vn.p, n_verts, nullptr));
    else
        RT_HIP(ctx, rt_exact::launch_debug_eval(op, n, d_in.p, in_stride, d_out.p, out_stride, d_vp.p, d_vn.p, n_verts, nullptr));
    RT_HIP(ctx, hipDeviceSynchronize());
    RT_HIP(ctx, hipMemcpy(out, d_out.p, n * out_stride * 4, hipMemcpyDeviceToHost));
    return RT_OK;
}
