// CPU ORACLE -- TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (see ref_math.h).
//
// NFF text -> (View, Scene), restating reference src/nff.rs line by line:
// read() :348-421, parse_view :102-157, parse_polygon :226-265,
// parse_polygon_patch :159-186, parse_cone :188-204, parse_cone_one_line :206-224,
// parse_fill :267-307, lights :309-333, parse_sphere :335-346.
#pragma once
#include <cctype>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "ref_scene.h"

namespace ref {

struct NffReader {
    const char* p;
    const char* end;
    // std::io::Stdin::read_line: returns the line including its '\n'; false at EOF.
    bool read_line(std::string& line) {
        if (p >= end) return false;
        const char* q = (const char*)memchr(p, '\n', (size_t)(end - p));
        const char* stop = q ? q + 1 : end;
        line.assign(p, stop);
        p = stop;
        return true;
    }
};

static inline std::vector<std::string> split_ws(const std::string& s) {  // str::split_whitespace
    std::vector<std::string> out;
    size_t i = 0;
    while (i < s.size()) {
        while (i < s.size() && isspace((unsigned char)s[i])) ++i;
        size_t j = i;
        while (j < s.size() && !isspace((unsigned char)s[j])) ++j;
        if (j > i) out.emplace_back(s.substr(i, j - i));
        i = j;
    }
    return out;
}

// Grammar of Rust's <f32 as FromStr>: [+-]? ( inf | infinity | nan | digits[.digits]?[exp] | .digits[exp] )
static inline bool parse_f32(const std::string& w, float& out) {
    size_t i = 0, n = w.size();
    if (n == 0) return false;
    if (w[i] == '+' || w[i] == '-') ++i;
    if (i >= n) return false;
    std::string rest = w.substr(i);
    std::string lower;
    for (char c : rest) lower.push_back((char)tolower((unsigned char)c));
    if (lower == "inf" || lower == "infinity" || lower == "nan") {
        out = strtof(w.c_str(), nullptr);
        return true;
    }
    size_t digits = 0;
    while (i < n && isdigit((unsigned char)w[i])) { ++i; ++digits; }
    if (i < n && w[i] == '.') {
        ++i;
        while (i < n && isdigit((unsigned char)w[i])) { ++i; ++digits; }
    }
    if (digits == 0) return false;
    if (i < n && (w[i] == 'e' || w[i] == 'E')) {
        ++i;
        if (i < n && (w[i] == '+' || w[i] == '-')) ++i;
        size_t ed = 0;
        while (i < n && isdigit((unsigned char)w[i])) { ++i; ++ed; }
        if (ed == 0) return false;
    }
    if (i != n) return false;
    out = strtof(w.c_str(), nullptr);  // correctly rounded, as Rust's parser is
    return true;
}

static inline bool parse_u32(const std::string& w, uint32_t& out) {
    size_t i = 0, n = w.size();
    if (n == 0) return false;
    if (w[i] == '+') ++i;
    if (i >= n) return false;
    uint64_t v = 0;
    for (; i < n; ++i) {
        if (!isdigit((unsigned char)w[i])) return false;
        v = v * 10 + (uint64_t)(w[i] - '0');
        if (v > 0xffffffffull) return false;
    }
    out = (uint32_t)v;
    return true;
}

// parse_values, src/nff.rs:48-75.  A wrong word count is a panic in the reference;
// here it is reported as an error.
static inline bool parse_floats(const std::string& line, size_t start_word, size_t count, std::vector<float>& vals,
                                std::string& err) {
    std::vector<std::string> words = split_ws(line);
    if (words.size() != start_word + count) { err = "wrong value count"; return false; }
    vals.clear();
    for (size_t i = start_word; i < words.size(); ++i) {
        float f;
        if (!parse_f32(words[i], f)) { err = "invalid float literal"; return false; }
        vals.push_back(f);
    }
    return true;
}

static inline bool starts_with(const std::string& s, const char* prefix) {
    return s.compare(0, strlen(prefix), prefix) == 0;
}

// src/nff.rs:102-157
static inline bool parse_view(NffReader& rd, View& view, std::string& err) {
    bool have_from = false, have_at = false, have_up = false, have_angle = false, have_hither = false,
         have_res = false;
    std::vector<float> v;
    for (;;) {
        std::string line;
        if (!rd.read_line(line)) { err = "Error parsing command v: missing parameters"; return false; }
        if (starts_with(line, "from")) {
            if (!parse_floats(line, 1, 3, v, err)) return false;
            view.from = V3{v[0], v[1], v[2]}; have_from = true;
        } else if (starts_with(line, "at")) {
            if (!parse_floats(line, 1, 3, v, err)) return false;
            view.at = V3{v[0], v[1], v[2]}; have_at = true;
        } else if (starts_with(line, "up")) {
            if (!parse_floats(line, 1, 3, v, err)) return false;
            view.up = V3{v[0], v[1], v[2]}; have_up = true;
        } else if (starts_with(line, "angle")) {
            if (!parse_floats(line, 1, 1, v, err)) return false;
            view.angle = v[0]; have_angle = true;
        } else if (starts_with(line, "hither")) {
            if (!parse_floats(line, 1, 1, v, err)) return false;
            view.hither = v[0]; have_hither = true;
        } else if (starts_with(line, "resolution")) {
            std::vector<std::string> words = split_ws(line);
            if (words.size() != 3) { err = "wrong value count"; return false; }
            uint32_t w, h;
            if (!parse_u32(words[1], w) || !parse_u32(words[2], h)) { err = "invalid integer literal"; return false; }
            view.width = w; view.height = h; have_res = true;
        }
        if (have_from && have_at && have_up && have_angle && have_hither && have_res) return true;
    }
}

// src/nff.rs:348-421
static inline bool read_nff(const char* text, size_t len, bool use_phong, View& view, Scene& scene,
                            std::string& err, std::vector<std::string>* warnings = nullptr) {
    NffReader rd{text, text + len};
    bool have_view = false;
    scene = Scene();
    scene.surfaces.push_back(Surface());  // default surface, :352-360 (Phong even under --blinn-phong)
    int surface = 0;
    std::vector<float> v;

    for (;;) {
        std::string line;
        if (!rd.read_line(line)) break;
        if (starts_with(line, "#")) continue;  // :370
        std::vector<std::string> tok = split_ws(line);
        if (tok.empty()) continue;
        const std::string& cmd = tok[0];
        const size_t nargs = tok.size() - 1;

        if (cmd == "v" && nargs == 0) {
            if (!parse_view(rd, view, err)) return false;
            have_view = true;
        } else if (cmd == "b" && nargs == 3) {  // :77-100
            float c[3];
            const char* names[3] = {"red", "green", "blue"};
            for (int k = 0; k < 3; ++k)
                if (!parse_f32(tok[1 + k], c[k])) {
                    err = std::string("Error parsing command b: invalid ") + names[k] + " value";
                    return false;
                }
            scene.background = Color{c[0], c[1], c[2]};
        } else if ((cmd == "pp" || cmd == "p") && nargs == 1) {
            const bool patch = cmd == "pp";
            uint32_t count;
            if (!parse_u32(tok[1], count)) { err = "invalid vertex count"; return false; }
            if (count < 3) {
                err = std::string("Error parsing command ") + (patch ? "pp" : "p") + ": insufficient vertex count";
                return false;
            }
            Shape poly;
            poly.kind = KIND_POLYGON;
            for (uint32_t k = 0; k < count; ++k) {
                std::string vl;
                if (!rd.read_line(vl)) {
                    err = std::string("Error parsing command ") + (patch ? "pp" : "p") + ": missing parameters";
                    return false;
                }
                if (!parse_floats(vl, 0, patch ? 6 : 3, v, err)) return false;
                PointNormal pn;
                pn.point = V3{v[0], v[1], v[2]};
                if (patch) {
                    V3 nrm{v[3], v[4], v[5]};
                    pn.normal = normalized(nrm);  // :177-178
                } else {
                    pn.normal = V3{0, 0, 0};
                }
                poly.vertices.push_back(pn);
            }
            if (!patch) {  // :249-260 one flat normal for every vertex
                V3 v1 = sub(poly.vertices[1].point, poly.vertices[0].point);
                V3 v2 = sub(poly.vertices[2].point, poly.vertices[0].point);
                V3 nrm = normalized(cross(v1, v2));
                for (PointNormal& pn : poly.vertices) pn.normal = nrm;
            }
            scene.primitives.push_back(Primitive{poly, surface});
        } else if (cmd == "f" && nargs == 8) {  // :267-307
            float a[8];
            for (int k = 0; k < 8; ++k)
                if (!parse_f32(tok[1 + k], a[k])) { err = "invalid float literal"; return false; }
            Surface s;
            s.color = Color{a[0], a[1], a[2]};
            s.kd = a[3];
            s.ks = a[4];
            s.shine = a[5];
            s.reflectance = a[4];  // reflectance := ks, :292
            s.transmittance = a[6];
            s.ior = a[7];
            s.blinn = !use_phong;
            scene.surfaces.push_back(s);
            surface = (int)scene.surfaces.size() - 1;
        } else if (cmd == "l" && (nargs == 3 || nargs == 6)) {  // :309-333
            float a[6] = {0, 0, 0, 1.0f, 1.0f, 1.0f};
            for (size_t k = 0; k < nargs; ++k)
                if (!parse_f32(tok[1 + k], a[k])) { err = "invalid float literal"; return false; }
            scene.lights.push_back(Light{V3{a[0], a[1], a[2]}, Color{a[3], a[4], a[5]}});
        } else if (cmd == "s" && nargs == 4) {  // :335-346
            float a[4];
            for (int k = 0; k < 4; ++k)
                if (!parse_f32(tok[1 + k], a[k])) { err = "invalid float literal"; return false; }
            Shape s;
            s.kind = KIND_SPHERE;
            s.center = V3{a[0], a[1], a[2]};
            s.radius = a[3];
            scene.primitives.push_back(Primitive{s, surface});
        } else if (cmd == "c" && nargs == 0) {  // :188-204
            std::string bl, al;
            rd.read_line(bl);
            std::vector<float> bv, av;
            if (!parse_floats(bl, 0, 4, bv, err)) return false;
            rd.read_line(al);
            if (!parse_floats(al, 0, 4, av, err)) return false;
            Shape s;
            s.kind = KIND_CONE;
            s.base = V3{bv[0], bv[1], bv[2]};
            s.apex = V3{av[0], av[1], av[2]};
            s.base_radius = bv[3];
            s.apex_radius = av[3];
            scene.primitives.push_back(Primitive{s, surface});
        } else if (cmd == "c" && nargs == 8) {  // :206-224
            float a[8];
            for (int k = 0; k < 8; ++k)
                if (!parse_f32(tok[1 + k], a[k])) { err = "invalid float literal"; return false; }
            Shape s;
            s.kind = KIND_CONE;
            s.base = V3{a[0], a[1], a[2]};
            s.base_radius = a[3];
            s.apex = V3{a[4], a[5], a[6]};
            s.apex_radius = a[7];
            scene.primitives.push_back(Primitive{s, surface});
        } else {
            if (warnings) warnings->push_back("unrecognized command: " + line);  // :409
        }
    }
    if (!have_view) { err = "Error parsing command v: missing view"; return false; }  // :417-419
    return true;
}

}  // namespace ref
