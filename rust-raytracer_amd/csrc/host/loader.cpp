// Scene-file front end: data/<name>.json | .yaml  ->  rt_scene (+ rt_camera).
//
// The reference ships these files (data/scene_10.json, scene_200_no_bvh.json,
// scene_500.json and YAML twins) but NO loader (README.md:86-89 leaves it as
// "Track 5"); the schema is derived from the files themselves:
//   top:        { objects: <node>, camera: {look_from, look_at, vup, vfov, aspect, aperture, focus_dist} }
//   HitableList { items: [node...] }            -> impl Hitable for Vec<Arc<dyn Hitable>>
//   BVHNode     { left, right, bounding_box }   -> BVHNode::construct(left, right); the box is redundant
//                                                  (README.md:88) and recomputed from the children
//   Sphere      { center, radius, material }
//   Lambertian{albedo:<tex>} Metal{albedo:{x,y,z},fuzz} Dielectric{ref_idx} DiffuseLight{emit:<tex>}
//   ConstantTexture{color} CheckerTexture{t0,t1}
// data/test.json's older schema (bare "objects" array, "object_type" key) is accepted
// structurally; a Sphere without a material is rejected with RT_ERR_SCHEMA.
// Both parsers are hand-written (no third-party dependency): a strict JSON reader and
// a reader for the block/flow YAML subset these files use.
#include <cerrno>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>

#include "scene.h"

namespace rtamd {

namespace {

struct Value {
    enum T { NUL, BOOL, NUM, STR, ARR, OBJ } t = NUL;
    double num = 0;
    bool b = false;
    std::string str;
    std::vector<Value> arr;
    std::vector<std::pair<std::string, Value>> obj;
    const Value* get(const char* k) const {
        if (t != OBJ) return nullptr;
        for (auto& kv : obj)
            if (kv.first == k) return &kv.second;
        return nullptr;
    }
};

[[noreturn]] void schema(const std::string& m) { throw RtError(RT_ERR_SCHEMA, m); }

// ---------------------------------------------------------------- JSON ----
struct Json {
    const char* p;
    const char* e;
    int line = 1;
    void ws() {
        while (p < e && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) {
            if (*p == '\n') line++;
            p++;
        }
    }
    [[noreturn]] void fail(const char* m) { schema(std::string("JSON: ") + m + " at line " + std::to_string(line)); }
    std::string string_() {
        std::string s;
        p++;  // opening quote
        while (p < e && *p != '"') {
            if (*p == '\\') {
                p++;
                if (p >= e) fail("bad escape");
                switch (*p) {
                    case 'n': s += '\n'; break;
                    case 't': s += '\t'; break;
                    case 'r': s += '\r'; break;
                    case 'b': s += '\b'; break;
                    case 'f': s += '\f'; break;
                    case 'u': {
                        if (e - p < 5) fail("bad \\u escape");
                        unsigned cp = (unsigned)strtoul(std::string(p + 1, p + 5).c_str(), nullptr, 16);
                        p += 4;
                        if (cp < 0x80) s += (char)cp;
                        else if (cp < 0x800) { s += (char)(0xC0 | (cp >> 6)); s += (char)(0x80 | (cp & 0x3F)); }
                        else { s += (char)(0xE0 | (cp >> 12)); s += (char)(0x80 | ((cp >> 6) & 0x3F)); s += (char)(0x80 | (cp & 0x3F)); }
                        break;
                    }
                    default: s += *p;
                }
                p++;
            } else {
                s += *p++;
            }
        }
        if (p >= e) fail("unterminated string");
        p++;
        return s;
    }
    Value value(int depth) {
        if (depth > 512) fail("nesting too deep");
        ws();
        if (p >= e) fail("unexpected end");
        Value v;
        if (*p == '{') {
            v.t = Value::OBJ;
            p++;
            ws();
            if (p < e && *p == '}') { p++; return v; }
            for (;;) {
                ws();
                if (p >= e || *p != '"') fail("expected key");
                std::string k = string_();
                ws();
                if (p >= e || *p != ':') fail("expected ':'");
                p++;
                v.obj.emplace_back(std::move(k), value(depth + 1));
                ws();
                if (p < e && *p == ',') { p++; continue; }
                if (p < e && *p == '}') { p++; break; }
                fail("expected ',' or '}'");
            }
        } else if (*p == '[') {
            v.t = Value::ARR;
            p++;
            ws();
            if (p < e && *p == ']') { p++; return v; }
            for (;;) {
                v.arr.push_back(value(depth + 1));
                ws();
                if (p < e && *p == ',') { p++; continue; }
                if (p < e && *p == ']') { p++; break; }
                fail("expected ',' or ']'");
            }
        } else if (*p == '"') {
            v.t = Value::STR;
            v.str = string_();
        } else if (!strncmp(p, "true", 4) && e - p >= 4) {
            v.t = Value::BOOL; v.b = true; p += 4;
        } else if (!strncmp(p, "false", 5) && e - p >= 5) {
            v.t = Value::BOOL; v.b = false; p += 5;
        } else if (!strncmp(p, "null", 4) && e - p >= 4) {
            p += 4;
        } else {
            char* end = nullptr;
            errno = 0;
            double d = strtod(p, &end);  // correctly rounded decimal -> f64
            if (end == p) fail("unexpected character");
            v.t = Value::NUM;
            v.num = d;
            p = end;
        }
        return v;
    }
};

// ---------------------------------------------------------------- YAML ----
// Block mappings / block sequences by indentation, "- " items, plain or quoted
// scalars, '#' comments, and JSON-like flow collections ({a: 1, b: [2, 3]}).
struct YLine {
    int indent;
    std::string text;  // without indentation / trailing comment
    int lineno;
};
struct Yaml {
    std::vector<YLine> lines;
    size_t i = 0;
    [[noreturn]] void fail(const std::string& m, int ln) { schema("YAML: " + m + " at line " + std::to_string(ln)); }

    static std::string strip_comment(const std::string& s) {
        bool sq = false, dq = false;
        for (size_t k = 0; k < s.size(); k++) {
            char c = s[k];
            if (c == '\'' && !dq) sq = !sq;
            else if (c == '"' && !sq) dq = !dq;
            else if (c == '#' && !sq && !dq && (k == 0 || s[k - 1] == ' ' || s[k - 1] == '\t')) return s.substr(0, k);
        }
        return s;
    }
    static std::string rtrim(std::string s) {
        while (!s.empty() && (s.back() == ' ' || s.back() == '\t' || s.back() == '\r')) s.pop_back();
        return s;
    }
    void load(const std::string& text) {
        std::istringstream in(text);
        std::string ln;
        int no = 0;
        while (std::getline(in, ln)) {
            no++;
            ln = rtrim(strip_comment(ln));
            size_t k = 0;
            while (k < ln.size() && ln[k] == ' ') k++;
            if (k < ln.size() && ln[k] == '\t') fail("tab indentation", no);
            if (k == ln.size()) continue;
            std::string body = ln.substr(k);
            if (body == "---" || body == "...") continue;
            lines.push_back(YLine{(int)k, body, no});
        }
    }
    static Value scalar(const std::string& raw) {
        Value v;
        std::string s = raw;
        if (s.size() >= 2 && ((s.front() == '"' && s.back() == '"') || (s.front() == '\'' && s.back() == '\''))) {
            v.t = Value::STR;
            v.str = s.substr(1, s.size() - 2);
            return v;
        }
        if (s == "~" || s == "null" || s.empty()) return v;
        if (s == "true" || s == "True") { v.t = Value::BOOL; v.b = true; return v; }
        if (s == "false" || s == "False") { v.t = Value::BOOL; v.b = false; return v; }
        if (s == ".inf" || s == "+.inf") { v.t = Value::NUM; v.num = INFINITY; return v; }
        if (s == "-.inf") { v.t = Value::NUM; v.num = -INFINITY; return v; }
        char* end = nullptr;
        double d = strtod(s.c_str(), &end);
        if (end != s.c_str() && *end == 0 && (isdigit((unsigned char)s[0]) || s[0] == '-' || s[0] == '+' || s[0] == '.')) {
            v.t = Value::NUM;
            v.num = d;
            return v;
        }
        v.t = Value::STR;
        v.str = s;
        return v;
    }
    // flow collection parser on a single string
    struct Flow {
        const std::string& s;
        size_t p = 0;
        int ln;
        Yaml* y;
        void ws() { while (p < s.size() && (s[p] == ' ' || s[p] == '\t')) p++; }
        std::string token(const char* stops) {
            ws();
            std::string t;
            if (p < s.size() && (s[p] == '"' || s[p] == '\'')) {
                char q = s[p];
                t += s[p++];
                while (p < s.size() && s[p] != q) t += s[p++];
                if (p < s.size()) t += s[p++];
                return t;
            }
            while (p < s.size() && !strchr(stops, s[p])) t += s[p++];
            return rtrim(t);
        }
        Value value() {
            ws();
            Value v;
            if (p < s.size() && s[p] == '{') {
                v.t = Value::OBJ;
                p++;
                ws();
                if (p < s.size() && s[p] == '}') { p++; return v; }
                for (;;) {
                    std::string k = token(":,}");
                    if (k.size() >= 2 && (k.front() == '"' || k.front() == '\'')) k = k.substr(1, k.size() - 2);
                    ws();
                    if (p >= s.size() || s[p] != ':') y->fail("expected ':' in flow mapping", ln);
                    p++;
                    v.obj.emplace_back(k, value());
                    ws();
                    if (p < s.size() && s[p] == ',') { p++; continue; }
                    if (p < s.size() && s[p] == '}') { p++; break; }
                    y->fail("expected ',' or '}' in flow mapping", ln);
                }
                return v;
            }
            if (p < s.size() && s[p] == '[') {
                v.t = Value::ARR;
                p++;
                ws();
                if (p < s.size() && s[p] == ']') { p++; return v; }
                for (;;) {
                    v.arr.push_back(value());
                    ws();
                    if (p < s.size() && s[p] == ',') { p++; continue; }
                    if (p < s.size() && s[p] == ']') { p++; break; }
                    y->fail("expected ',' or ']' in flow sequence", ln);
                }
                return v;
            }
            return scalar(token(",}]"));
        }
    };
    static int flow_balance(const std::string& s) {
        int depth = 0;
        bool sq = false, dq = false;
        for (char c : s) {
            if (c == '\'' && !dq) sq = !sq;
            else if (c == '"' && !sq) dq = !dq;
            else if (!sq && !dq && (c == '{' || c == '[')) depth++;
            else if (!sq && !dq && (c == '}' || c == ']')) depth--;
        }
        return depth;
    }
    // `i` must already point past the line `s` came from: a flow collection may continue on following lines
    Value inline_value(const std::string& s0, int ln) {
        if (!s0.empty() && (s0[0] == '{' || s0[0] == '[')) {
            std::string s = s0;
            while (flow_balance(s) > 0 && i < lines.size()) {
                s += " " + lines[i].text;
                i++;
            }
            Flow f{s, 0, ln, this};
            return f.value();
        }
        return scalar(s0);
    }
    // split "key: rest" ; returns false if the text is not a mapping entry
    static bool split_key(const std::string& t, std::string& key, std::string& rest) {
        if (t.empty() || t[0] == '{' || t[0] == '[') return false;
        size_t k = 0;
        if (t[0] == '"' || t[0] == '\'') {
            k = t.find(t[0], 1);
            if (k == std::string::npos) return false;
            k++;
        } else {
            while (k < t.size() && !(t[k] == ':' && (k + 1 == t.size() || t[k + 1] == ' '))) k++;
        }
        if (k >= t.size() || t[k] != ':') return false;
        key = rtrim(t.substr(0, k));
        if (key.size() >= 2 && (key.front() == '"' || key.front() == '\'')) key = key.substr(1, key.size() - 2);
        rest = k + 1 < t.size() ? t.substr(k + 1) : "";
        size_t a = 0;
        while (a < rest.size() && rest[a] == ' ') a++;
        rest = rest.substr(a);
        return true;
    }
    Value block(int indent, int depth) {
        if (depth > 512) fail("nesting too deep", lines[i].lineno);
        if (i >= lines.size()) return Value();
        const YLine& first = lines[i];
        if (first.text[0] == '-' && (first.text.size() == 1 || first.text[1] == ' ')) return sequence(indent, depth);
        std::string k, r;
        if (split_key(first.text, k, r)) return mapping(indent, depth);
        const std::string text = first.text;
        const int lineno = first.lineno;
        i++;
        return inline_value(text, lineno);
    }
    Value mapping(int indent, int depth) {
        Value v;
        v.t = Value::OBJ;
        while (i < lines.size() && lines[i].indent == indent) {
            const YLine ln = lines[i];
            if (ln.text[0] == '-' && (ln.text.size() == 1 || ln.text[1] == ' ')) break;
            std::string key, rest;
            if (!split_key(ln.text, key, rest)) fail("expected 'key: value'", ln.lineno);
            i++;
            if (!rest.empty()) {
                v.obj.emplace_back(key, inline_value(rest, ln.lineno));
            } else if (i < lines.size() && (lines[i].indent > indent ||
                                            (lines[i].indent == indent && lines[i].text[0] == '-' &&
                                             (lines[i].text.size() == 1 || lines[i].text[1] == ' ')))) {
                v.obj.emplace_back(key, block(lines[i].indent, depth + 1));
            } else {
                v.obj.emplace_back(key, Value());
            }
        }
        if (i < lines.size() && lines[i].indent > indent) fail("bad indentation", lines[i].lineno);
        return v;
    }
    Value sequence(int indent, int depth) {
        Value v;
        v.t = Value::ARR;
        while (i < lines.size() && lines[i].indent == indent && lines[i].text[0] == '-' &&
               (lines[i].text.size() == 1 || lines[i].text[1] == ' ')) {
            YLine& ln = lines[i];
            size_t k = 1;
            while (k < ln.text.size() && ln.text[k] == ' ') k++;
            if (k >= ln.text.size()) {  // "-" alone: nested block on following lines
                i++;
                if (i < lines.size() && lines[i].indent > indent) v.arr.push_back(block(lines[i].indent, depth + 1));
                else v.arr.push_back(Value());
                continue;
            }
            // rewrite "- rest" as a line holding `rest` at the deeper indentation and parse it as a block
            ln.indent = indent + (int)k;
            ln.text = ln.text.substr(k);
            v.arr.push_back(block(ln.indent, depth + 1));
        }
        return v;
    }
    Value parse() {
        if (lines.empty()) return Value();
        Value v = block(lines[0].indent, 0);
        if (i < lines.size()) fail("unexpected content", lines[i].lineno);
        return v;
    }
};

// ------------------------------------------------------- Value -> scene ----
double num(const Value& o, const char* k, const char* where) {
    const Value* v = o.get(k);
    if (!v || v->t != Value::NUM) schema(std::string(where) + ": missing number '" + k + "'");
    return v->num;
}
void vec3(const Value& o, const char* k, const char* where, double out[3]) {
    const Value* v = o.get(k);
    if (!v || v->t != Value::OBJ) schema(std::string(where) + ": missing vector '" + k + "'");
    out[0] = num(*v, "x", k);
    out[1] = num(*v, "y", k);
    out[2] = num(*v, "z", k);
}
std::string type_of(const Value& o) {
    const Value* t = o.get("type");
    if (!t) t = o.get("object_type");  // data/test.json's older key
    if (!t || t->t != Value::STR) schema("node without a 'type'");
    return t->str;
}
const Value& child(const Value& o, const char* k, const char* where) {
    const Value* v = o.get(k);
    if (!v || v->t != Value::OBJ) schema(std::string(where) + ": missing '" + k + "'");
    return *v;
}

int build_texture(rt_scene& s, const Value& d) {
    std::string t = type_of(d);
    if (t == "ConstantTexture") {
        double c[3];
        vec3(d, "color", "ConstantTexture", c);
        return add_texture_constant(s, c);
    }
    if (t == "CheckerTexture") {
        int t0 = build_texture(s, child(d, "t0", "CheckerTexture"));
        int t1 = build_texture(s, child(d, "t1", "CheckerTexture"));
        return add_texture_checker(s, t0, t1);
    }
    schema("unknown texture type '" + t + "'");
}
int build_material(rt_scene& s, const Value& d) {
    std::string t = type_of(d);
    if (t == "Lambertian") return add_material(s, MAT_LAMBERTIAN, build_texture(s, child(d, "albedo", "Lambertian")), 0.);
    if (t == "Metal") {  // albedo is a bare {x,y,z}, not a texture node
        double c[3];
        vec3(d, "albedo", "Metal", c);
        return add_material(s, MAT_METAL, add_texture_constant(s, c), num(d, "fuzz", "Metal"));
    }
    if (t == "Dielectric") {  // only ref_idx in the files; Dielectric.albedo (material.rs:143) defaults to white
        const double one[3] = {1., 1., 1.};
        return add_material(s, MAT_DIELECTRIC, add_texture_constant(s, one), num(d, "ref_idx", "Dielectric"));
    }
    if (t == "DiffuseLight") return add_material(s, MAT_DIFFUSE_LIGHT, build_texture(s, child(d, "emit", "DiffuseLight")), 0.);
    schema("unknown material type '" + t + "'");
}
int build_object(rt_scene& s, const Value& d, int depth) {
    if (depth > 512) schema("object nesting too deep");
    if (d.t != Value::OBJ) schema("object node is not a mapping");
    std::string t = type_of(d);
    if (t == "HitableList") {
        const Value* items = d.get("items");
        if (!items || items->t != Value::ARR) schema("HitableList without 'items'");
        std::vector<int> ids;
        for (auto& it : items->arr) ids.push_back(build_object(s, it, depth + 1));
        return add_list(s, (int)ids.size(), ids.data());
    }
    if (t == "BVHNode") {
        int l = build_object(s, child(d, "left", "BVHNode"), depth + 1);
        int r = build_object(s, child(d, "right", "BVHNode"), depth + 1);
        return add_bvh_node(s, l, r);
    }
    if (t == "Sphere") {
        double c[3];
        vec3(d, "center", "Sphere", c);
        const Value* m = d.get("material");
        if (!m || m->t != Value::OBJ) schema("Sphere without material");
        return add_sphere(s, c, num(d, "radius", "Sphere"), build_material(s, *m));
    }
    schema("unknown object type '" + t + "'");
}

}  // namespace

rt_scene* load_scene_file(const char* path, rt_camera* cam) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw RtError(RT_ERR_IO, std::string("cannot open scene file ") + path);
    std::stringstream ss;
    ss << f.rdbuf();
    std::string text = ss.str();
    std::string p(path);
    bool yaml = p.size() > 5 && (p.rfind(".yaml") == p.size() - 5 || p.rfind(".yml") == p.size() - 4);
    Value doc;
    if (yaml) {
        Yaml y;
        y.load(text);
        doc = y.parse();
    } else {
        Json j{text.data(), text.data() + text.size()};
        doc = j.value(0);
        j.ws();
        if (j.p != j.e) j.fail("trailing characters");
    }
    if (doc.t != Value::OBJ) schema("top level is not a mapping");
    const Value* objs = doc.get("objects");
    if (!objs) schema("missing 'objects'");
    std::unique_ptr<rt_scene> s(new rt_scene());
    int root;
    if (objs->t == Value::ARR) {  // older schema: bare array == HitableList
        std::vector<int> ids;
        for (auto& it : objs->arr) ids.push_back(build_object(*s, it, 1));
        root = add_list(*s, (int)ids.size(), ids.data());
    } else {
        root = build_object(*s, *objs, 0);
    }
    s->root = root;
    const Value& c = child(doc, "camera", "scene");
    if (cam) {
        vec3(c, "look_from", "camera", cam->look_from);
        vec3(c, "look_at", "camera", cam->look_at);
        vec3(c, "vup", "camera", cam->vup);
        cam->vfov = num(c, "vfov", "camera");
        cam->aspect = num(c, "aspect", "camera");
        cam->aperture = num(c, "aperture", "camera");
        cam->focus_dist = num(c, "focus_dist", "camera");
    }
    flatten(*s);
    return s.release();
}

}  // namespace rtamd
