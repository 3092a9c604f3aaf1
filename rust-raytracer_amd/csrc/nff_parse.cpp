// NFF front end: text -> flat scene description + view.
//
// Accepts exactly what the reference's reader accepts (reference src/nff.rs:348-421):
//   '#' in column 0 starts a comment line (:370); blank lines are skipped (:375);
//   a line is dispatched on its first token AND its argument count (:381-410):
//     v                      then from/at/up/angle/hither/resolution lines, matched by
//                            line prefix in any order; the block ends when all six have
//                            been seen (:102-157)
//     b r g b                background (:77-100)
//     l x y z [r g b]        light, white by default (:309-333)
//     f r g b kd ks shine t ior   surface; reflectance := ks (:267-307)
//     s x y z r              sphere (:335-346)
//     c                      cone: next two lines "x y z r" = base, apex (:188-204)
//     c bx by bz br ax ay az ar   cone on one line (:206-224)
//     p n                    polygon: n lines "x y z"; one flat normal
//                            normalize((v1-v0) x (v2-v0)) for every vertex (:226-265)
//     pp n                   patch: n lines "x y z nx ny nz"; normals normalised (:159-186)
//   n < 3 vertices is an error (:162, :229); a vertex/cone/view line with the wrong
//   number of values is an error (a panic in the reference, :57); anything else is a
//   warning and is skipped (:409); a file without a view is an error (:417-419).
// Primitives take the surface in force when they appear; before any `f` that is a
// Phong surface, white, kd 1, ks 0, shine 1, no reflection or transmission (:352-360),
// also when Blinn-Phong shading was asked for.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "rt_host.h"

namespace {

using rtamd::SceneData;

struct Cursor {
    const char* pos;
    const char* end;
    // Next line without its terminator; `raw` keeps the untrimmed text for prefix tests.
    bool next(std::string& raw) {
        if (pos >= end) return false;
        const char* nl = static_cast<const char*>(memchr(pos, '\n', static_cast<size_t>(end - pos)));
        const char* stop = nl ? nl : end;
        raw.assign(pos, stop);
        pos = nl ? nl + 1 : end;
        return true;
    }
};

bool is_space(char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '\v' || c == '\f'; }

void tokenize(const std::string& s, std::vector<std::string>& out) {
    out.clear();
    size_t i = 0;
    const size_t n = s.size();
    while (i < n) {
        if (is_space(s[i])) { ++i; continue; }
        size_t j = i;
        while (j < n && !is_space(s[j])) ++j;
        out.emplace_back(s, i, j - i);
        i = j;
    }
}

bool all_digits(const char* p, const char* e) {
    if (p == e) return false;
    for (; p < e; ++p)
        if (*p < '0' || *p > '9') return false;
    return true;
}

// Decimal float literal as Rust's f32::from_str takes it: sign, then inf / infinity / nan
// (any case) or digits with optional fraction and exponent; no hex, no trailing junk.
bool to_f32(const std::string& t, float& v) {
    const char* p = t.c_str();
    const char* e = p + t.size();
    if (p < e && (*p == '+' || *p == '-')) ++p;
    if (p == e) return false;
    auto ieq = [&](const char* lit) {
        size_t n = strlen(lit);
        if (static_cast<size_t>(e - p) != n) return false;
        for (size_t k = 0; k < n; ++k)
            if ((p[k] | 0x20) != lit[k]) return false;
        return true;
    };
    bool ok = ieq("inf") || ieq("infinity") || ieq("nan");
    if (!ok) {
        const char* q = p;
        size_t nd = 0;
        while (q < e && *q >= '0' && *q <= '9') { ++q; ++nd; }
        if (q < e && *q == '.') {
            ++q;
            while (q < e && *q >= '0' && *q <= '9') { ++q; ++nd; }
        }
        if (nd == 0) return false;
        if (q < e && (*q == 'e' || *q == 'E')) {
            ++q;
            if (q < e && (*q == '+' || *q == '-')) ++q;
            if (!all_digits(q, e)) return false;
            q = e;
        }
        if (q != e) return false;
    }
    v = strtof(t.c_str(), nullptr);  // round-to-nearest, like the reference's parser
    return true;
}

bool to_u32(const std::string& t, uint32_t& v) {
    const char* p = t.c_str();
    const char* e = p + t.size();
    if (p < e && *p == '+') ++p;
    if (!all_digits(p, e)) return false;
    uint64_t acc = 0;
    for (; p < e; ++p) {
        acc = acc * 10 + static_cast<uint64_t>(*p - '0');
        if (acc > 0xffffffffull) return false;
    }
    v = static_cast<uint32_t>(acc);
    return true;
}

// `count` floats from tokens[first...]; the line must hold exactly first + count tokens.
bool floats_exact(const std::vector<std::string>& tok, size_t first, size_t count, float* out, std::string& err) {
    if (tok.size() != first + count) { err = "wrong value count"; return false; }
    for (size_t k = 0; k < count; ++k)
        if (!to_f32(tok[first + k], out[k])) { err = "invalid number '" + tok[first + k] + "'"; return false; }
    return true;
}

bool has_prefix(const std::string& s, const char* pre) { return s.compare(0, strlen(pre), pre) == 0; }

void normalize3(float v[3]) {  // Vector::normalize, reference src/vector_math.rs:30-33
    float m = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    v[0] /= m; v[1] /= m; v[2] /= m;
}

bool read_view(Cursor& cur, rt_view& view, std::string& err) {
    enum { FROM = 1, AT = 2, UP = 4, ANGLE = 8, HITHER = 16, RES = 32, ALL = 63 };
    unsigned seen = 0;
    std::string raw;
    std::vector<std::string> tok;
    while (seen != ALL) {
        if (!cur.next(raw)) { err = "Error parsing command v: missing parameters"; return false; }
        tokenize(raw, tok);
        if (has_prefix(raw, "from")) {
            if (!floats_exact(tok, 1, 3, view.from, err)) return false;
            seen |= FROM;
        } else if (has_prefix(raw, "at")) {
            if (!floats_exact(tok, 1, 3, view.at, err)) return false;
            seen |= AT;
        } else if (has_prefix(raw, "up")) {
            if (!floats_exact(tok, 1, 3, view.up, err)) return false;
            seen |= UP;
        } else if (has_prefix(raw, "angle")) {
            if (!floats_exact(tok, 1, 1, &view.angle, err)) return false;
            seen |= ANGLE;
        } else if (has_prefix(raw, "hither")) {
            if (!floats_exact(tok, 1, 1, &view.hither, err)) return false;
            seen |= HITHER;
        } else if (has_prefix(raw, "resolution")) {
            if (tok.size() != 3) { err = "wrong value count"; return false; }
            if (!to_u32(tok[1], view.width) || !to_u32(tok[2], view.height)) { err = "invalid resolution"; return false; }
            seen |= RES;
        }
    }
    return true;
}

bool read_polygon(Cursor& cur, bool patch, const std::string& count_tok, uint32_t surface, SceneData& sd,
                  std::string& err) {
    const char* name = patch ? "pp" : "p";
    uint32_t n;
    if (!to_u32(count_tok, n)) { err = std::string("Error parsing command ") + name + ": invalid vertex count"; return false; }
    if (n < 3) { err = std::string("Error parsing command ") + name + ": insufficient vertex count"; return false; }
    const uint32_t first = static_cast<uint32_t>(sd.vertices.size());
    std::string raw;
    std::vector<std::string> tok;
    for (uint32_t k = 0; k < n; ++k) {
        if (!cur.next(raw)) { err = std::string("Error parsing command ") + name + ": missing parameters"; return false; }
        tokenize(raw, tok);
        float f[6] = {0, 0, 0, 0, 0, 0};
        if (!floats_exact(tok, 0, patch ? 6 : 3, f, err)) return false;
        rt_vertex v;
        memcpy(v.point, f, sizeof(v.point));
        memcpy(v.normal, f + 3, sizeof(v.normal));
        if (patch) normalize3(v.normal);
        sd.vertices.push_back(v);
    }
    if (!patch) {
        const rt_vertex* v = &sd.vertices[first];
        float a[3], b[3];
        for (int i = 0; i < 3; ++i) { a[i] = v[1].point[i] - v[0].point[i]; b[i] = v[2].point[i] - v[0].point[i]; }
        float nrm[3] = {a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]};
        normalize3(nrm);
        for (uint32_t k = 0; k < n; ++k) memcpy(sd.vertices[first + k].normal, nrm, sizeof(nrm));
    }
    sd.polygons.push_back(rt_polygon{first, n});
    sd.primitives.push_back(rt_primitive{RT_KIND_POLYGON, static_cast<uint32_t>(sd.polygons.size() - 1), surface});
    return true;
}

}  // namespace

extern "C" int rt_nff_parse(const char* text, size_t len, int use_phong, rt_parsed_scene** out, char* errbuf,
                            size_t errlen) {
    auto fail = [&](const std::string& msg) {
        if (errbuf && errlen) {
            size_t n = msg.size() < errlen - 1 ? msg.size() : errlen - 1;
            memcpy(errbuf, msg.data(), n);
            errbuf[n] = 0;
        }
        return static_cast<int>(RT_ERR_PARSE);
    };
    if (!out || (!text && len)) return fail("null argument");
    *out = nullptr;

    rt_parsed_scene* ps = new rt_parsed_scene();
    SceneData& sd = ps->data;
    rt_surface deflt{};
    deflt.color[0] = deflt.color[1] = deflt.color[2] = 1.0f;
    deflt.kd = 1.0f; deflt.ks = 0.0f; deflt.shine = 1.0f;
    deflt.reflectance = 0.0f; deflt.transmittance = 0.0f; deflt.ior = 1.0f;
    deflt.model = RT_MODEL_PHONG;
    sd.surfaces.push_back(deflt);
    uint32_t surface = 0;
    bool have_view = false;

    Cursor cur{text, text + len};
    std::string raw, err;
    std::vector<std::string> tok;
    while (cur.next(raw)) {
        if (!raw.empty() && raw[0] == '#') continue;
        tokenize(raw, tok);
        if (tok.empty()) continue;
        const std::string& cmd = tok[0];
        const size_t nargs = tok.size() - 1;
        bool ok = true;

        if (cmd == "v" && nargs == 0) {
            ok = read_view(cur, ps->view, err);
            have_view = ok;
        } else if (cmd == "b" && nargs == 3) {
            static const char* chan[3] = {"red", "green", "blue"};
            for (int k = 0; k < 3 && ok; ++k)
                if (!to_f32(tok[1 + k], sd.background[k])) {
                    err = std::string("Error parsing command b: invalid ") + chan[k] + " value";
                    ok = false;
                }
        } else if ((cmd == "pp" || cmd == "p") && nargs == 1) {
            ok = read_polygon(cur, cmd.size() == 2, tok[1], surface, sd, err);
        } else if (cmd == "f" && nargs == 8) {
            float f[8];
            ok = floats_exact(tok, 1, 8, f, err);
            if (ok) {
                rt_surface s{};
                memcpy(s.color, f, sizeof(s.color));
                s.kd = f[3]; s.ks = f[4]; s.shine = f[5];
                s.reflectance = f[4];
                s.transmittance = f[6]; s.ior = f[7];
                s.model = use_phong ? RT_MODEL_PHONG : RT_MODEL_BLINN_PHONG;
                sd.surfaces.push_back(s);
                surface = static_cast<uint32_t>(sd.surfaces.size() - 1);
            }
        } else if (cmd == "l" && (nargs == 3 || nargs == 6)) {
            float f[6] = {0, 0, 0, 1.0f, 1.0f, 1.0f};
            ok = floats_exact(tok, 1, nargs, f, err);
            if (ok) {
                rt_light l;
                memcpy(l.position, f, sizeof(l.position));
                memcpy(l.color, f + 3, sizeof(l.color));
                sd.lights.push_back(l);
            }
        } else if (cmd == "s" && nargs == 4) {
            float f[4];
            ok = floats_exact(tok, 1, 4, f, err);
            if (ok) {
                rt_sphere s;
                memcpy(s.center, f, sizeof(s.center));
                s.radius = f[3];
                sd.spheres.push_back(s);
                sd.primitives.push_back(rt_primitive{RT_KIND_SPHERE, static_cast<uint32_t>(sd.spheres.size() - 1), surface});
            }
        } else if (cmd == "c" && (nargs == 0 || nargs == 8)) {
            float f[8];
            if (nargs == 8) {
                ok = floats_exact(tok, 1, 8, f, err);
            } else {
                std::vector<std::string> t2;
                for (int half = 0; half < 2 && ok; ++half) {
                    std::string l2;
                    cur.next(l2);  // at end of input the empty line fails the count check below
                    tokenize(l2, t2);
                    ok = floats_exact(t2, 0, 4, f + 4 * half, err);
                }
            }
            if (ok) {
                rt_cone c;
                memcpy(c.base, f, sizeof(c.base));
                c.base_radius = f[3];
                memcpy(c.apex, f + 4, sizeof(c.apex));
                c.apex_radius = f[7];
                sd.cones.push_back(c);
                sd.primitives.push_back(rt_primitive{RT_KIND_CONE, static_cast<uint32_t>(sd.cones.size() - 1), surface});
            }
        } else {
            ps->warnings.push_back("unrecognized command: " + raw);
        }
        if (!ok) {
            delete ps;
            return fail(err);
        }
    }
    if (!have_view) {
        delete ps;
        return fail("Error parsing command v: missing view");
    }
    sd.refresh_desc();
    *out = ps;
    return RT_OK;
}

extern "C" const rt_scene_desc* rt_parsed_desc(const rt_parsed_scene* p) { return p ? &p->data.desc : nullptr; }
extern "C" const rt_view* rt_parsed_view(const rt_parsed_scene* p) { return p ? &p->view : nullptr; }
extern "C" uint32_t rt_parsed_warning_count(const rt_parsed_scene* p) {
    return p ? static_cast<uint32_t>(p->warnings.size()) : 0;
}
extern "C" void rt_parsed_free(rt_parsed_scene* p) { delete p; }

// PPM back end, reference src/ppm.rs:8-26: "P6\nW H\n255\n" then rows top to bottom.
extern "C" int rt_ppm_write(const char* path, uint32_t width, uint32_t height, const uint8_t* rgb) {
    if (!path || !rgb) return RT_ERR_INVALID;
    FILE* f = fopen(path, "wb");
    if (!f) return RT_ERR_IO;
    bool ok = fprintf(f, "P6\n%u %u\n%d\n", width, height, 255) > 0;
    const size_t row = static_cast<size_t>(width) * 3;
    for (uint32_t j = 0; ok && j < height; ++j) ok = fwrite(rgb + static_cast<size_t>(j) * row, 1, row, f) == row;
    ok = (fclose(f) == 0) && ok;
    return ok ? RT_OK : RT_ERR_IO;
}
