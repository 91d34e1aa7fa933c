// rt_scene_create_from_json: the text Scene.MarshalJSON writes (scene.go:82-114), read into an rt_scene_desc and handed to
// rt_scene_create.  The reader is a pull parser over the text itself, not a document tree: a scan's mesh is millions of
// faces, each written once under "Face" and once more inside the "BVH" field, and only the first copy is needed.
//
// Every interface value is wrapped as {"Data": ..., "Type": "<Go type name>"} (marshalInterface, json.go:52-79); the
// keys arrive sorted, so "Data" precedes the "Type" that says how to read it: the wrapper reader remembers where Data
// starts, skips it, and comes back once it knows the type.  Interface fields of types without a marshaller of their own
// (Lambertian.Texture, TransformedShape.Shape) are written bare; they are typed by their "Type" member when they have
// one and by their keys otherwise.  PinholeCamera, FocusableCamera and Triangle, which the reference writes but never
// registers for reading (json.go:82-100), are read too.
#include <charconv>
#include <cmath>
#include <cstring>
#include <set>
#include <string>
#include <vector>

#include "host_common.h"

namespace {

struct JsonError {
    int code;
    std::string msg;
};
[[noreturn]] void bad(const std::string& m, int code = RT_ERR_INVALID) { throw JsonError{code, m}; }

struct Cur {
    const char* p;
    const char* end;
    const char* base;
};

void ws(Cur& c) {
    while (c.p < c.end && (*c.p == ' ' || *c.p == '\t' || *c.p == '\n' || *c.p == '\r')) ++c.p;
}
char peek(Cur& c) {
    ws(c);
    if (c.p >= c.end) bad("unexpected end of JSON input");
    return *c.p;
}
[[noreturn]] void bad_char(Cur& c, const char* looking_for) {
    bad(std::string("invalid character '") + *c.p + "' at offset " + std::to_string(c.p - c.base) + " " + looking_for);
}
bool eat(Cur& c, char ch) {
    if (peek(c) != ch) return false;
    ++c.p;
    return true;
}
void need(Cur& c, char ch, const char* looking_for) {
    if (!eat(c, ch)) bad_char(c, looking_for);
}
bool eat_word(Cur& c, const char* w) {
    ws(c);
    size_t n = std::strlen(w);
    if (size_t(c.end - c.p) < n || std::memcmp(c.p, w, n) != 0) return false;
    c.p += n;
    return true;
}
bool eat_null(Cur& c) { return peek(c) == 'n' && eat_word(c, "null"); }

void put_utf8(std::string& s, uint32_t cp) {
    if (cp < 0x80) {
        s += char(cp);
    } else if (cp < 0x800) {
        s += char(0xC0 | (cp >> 6));
        s += char(0x80 | (cp & 63));
    } else if (cp < 0x10000) {
        s += char(0xE0 | (cp >> 12));
        s += char(0x80 | ((cp >> 6) & 63));
        s += char(0x80 | (cp & 63));
    } else {
        s += char(0xF0 | (cp >> 18));
        s += char(0x80 | ((cp >> 12) & 63));
        s += char(0x80 | ((cp >> 6) & 63));
        s += char(0x80 | (cp & 63));
    }
}
uint32_t hex4(Cur& c) {
    if (c.end - c.p < 4) bad("unexpected end of JSON input");
    uint32_t v = 0;
    for (int i = 0; i < 4; i++) {
        char h = *c.p++;
        v <<= 4;
        if (h >= '0' && h <= '9') v |= h - '0';
        else if (h >= 'a' && h <= 'f') v |= h - 'a' + 10;
        else if (h >= 'A' && h <= 'F') v |= h - 'A' + 10;
        else bad("invalid character in \\u hexadecimal character escape");
    }
    return v;
}
// A JSON string; `out` may be null to skip it.
void read_string(Cur& c, std::string* out) {
    need(c, '"', "looking for beginning of string");
    if (out) out->clear();
    for (;;) {
        const char* run = c.p;
        while (c.p < c.end && *c.p != '"' && *c.p != '\\' && static_cast<unsigned char>(*c.p) >= 0x20) ++c.p;
        if (out) out->append(run, c.p);
        if (c.p >= c.end) bad("unexpected end of JSON input");
        char ch = *c.p++;
        if (ch == '"') return;
        if (ch != '\\') bad("invalid character in string literal");
        if (c.p >= c.end) bad("unexpected end of JSON input");
        char e = *c.p++;
        uint32_t cp;
        switch (e) {
            case '"': cp = '"'; break;
            case '\\': cp = '\\'; break;
            case '/': cp = '/'; break;
            case 'b': cp = '\b'; break;
            case 'f': cp = '\f'; break;
            case 'n': cp = '\n'; break;
            case 'r': cp = '\r'; break;
            case 't': cp = '\t'; break;
            case 'u': {
                cp = hex4(c);
                if (cp >= 0xD800 && cp < 0xDC00 && c.end - c.p >= 6 && c.p[0] == '\\' && c.p[1] == 'u') {
                    Cur look = c;
                    look.p += 2;
                    uint32_t lo = hex4(look);
                    if (lo >= 0xDC00 && lo < 0xE000) {
                        cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
                        c = look;
                    }
                }
                if (cp >= 0xD800 && cp < 0xE000) cp = 0xFFFD;  // a lone surrogate becomes U+FFFD, as in encoding/json
                break;
            }
            default: bad("invalid character in string escape code");
        }
        if (out) put_utf8(*out, cp);
    }
}

// The extent of a number by JSON's grammar; the value by from_chars, which like strconv.ParseFloat rounds correctly.
double read_number(Cur& c, const char* what) {
    ws(c);
    const char* s = c.p;
    const char* p = s;
    auto digits = [&]() {
        const char* q = p;
        while (p < c.end && *p >= '0' && *p <= '9') ++p;
        return p > q;
    };
    if (p < c.end && *p == '-') ++p;
    if (p < c.end && *p == '0') ++p;
    else if (!digits()) bad(std::string("json: cannot unmarshal a value that is not a number into ") + what);
    if (p < c.end && *p == '.') {
        ++p;
        if (!digits()) bad("invalid character after decimal point in numeric literal");
    }
    if (p < c.end && (*p == 'e' || *p == 'E')) {
        ++p;
        if (p < c.end && (*p == '+' || *p == '-')) ++p;
        if (!digits()) bad("invalid character in exponent of numeric literal");
    }
    double v = 0;
    auto r = std::from_chars(s, p, v);
    if (r.ec == std::errc::result_out_of_range) bad(std::string("json: cannot unmarshal number ") + std::string(s, p) + " into " + what);
    c.p = p;
    return v;
}

// Steps over one value of any kind, checking that brackets pair up.
void skip_value(Cur& c) {
    std::vector<char> open;
    do {
        char ch = peek(c);
        if (ch == '"') {
            read_string(c, nullptr);
        } else if (ch == '{' || ch == '[') {
            open.push_back(ch);
            ++c.p;
            char nx = peek(c);
            if (nx == '}' || nx == ']') continue;  // closed by the branch below on the next turn
            if (ch == '{') {
                read_string(c, nullptr);
                need(c, ':', "after object key");
            }
            continue;
        } else if (ch == '}' || ch == ']') {
            if (open.empty() || open.back() != (ch == '}' ? '{' : '[')) bad_char(c, "looking for beginning of value");
            open.pop_back();
            ++c.p;
        } else if (ch == 't' || ch == 'f' || ch == 'n') {
            if (!eat_word(c, "true") && !eat_word(c, "false") && !eat_word(c, "null")) bad_char(c, "looking for beginning of value");
        } else {
            read_number(c, "a number");
        }
        // after a complete value: a comma and, inside an object, the next key
        while (!open.empty()) {
            char nx = peek(c);
            if (nx == ',') {
                ++c.p;
                if (open.back() == '{') {
                    read_string(c, nullptr);
                    need(c, ':', "after object key");
                }
                break;
            }
            if (nx == '}' || nx == ']') {
                if (open.back() != (nx == '}' ? '{' : '[')) bad_char(c, "after array element");
                open.pop_back();
                ++c.p;
                continue;
            }
            bad_char(c, "after object key:value pair");
        }
    } while (!open.empty());
}

// Calls f(key, cursor) for every member of the object at c; f must consume the member's value.  null = no members.
template <typename F>
void members(Cur& c, const char* what, F f) {
    if (eat_null(c)) return;
    if (peek(c) != '{') bad(std::string("json: cannot unmarshal a value that is not an object into ") + what);
    ++c.p;
    if (eat(c, '}')) return;
    std::string key;
    for (;;) {
        read_string(c, &key);
        need(c, ':', "after object key");
        f(key, c);
        if (eat(c, ',')) continue;
        need(c, '}', "after object key:value pair");
        return;
    }
}
template <typename F>
void elements(Cur& c, const char* what, F f) {
    if (eat_null(c)) return;
    if (peek(c) != '[') bad(std::string("json: cannot unmarshal a value that is not an array into ") + what);
    ++c.p;
    if (eat(c, ']')) return;
    for (;;) {
        f(c);
        if (eat(c, ',')) continue;
        need(c, ']', "after array element");
        return;
    }
}

double number_or_null(Cur& c, const char* what) { return eat_null(c) ? 0.0 : read_number(c, what); }
int64_t integer(Cur& c, const char* what) {
    if (eat_null(c)) return 0;
    double v = read_number(c, what);
    if (v != std::floor(v) || std::fabs(v) > 9.2e18) bad(std::string("json: cannot unmarshal a number with a fraction into ") + what);
    return int64_t(v);
}
void xyz(Cur& c, double* out, const char* what) {  // r3.Point / r3.Vec / Spectrum
    members(c, what, [&](const std::string& k, Cur& v) {
        if (k == "X") out[0] = number_or_null(v, what);
        else if (k == "Y") out[1] = number_or_null(v, what);
        else if (k == "Z") out[2] = number_or_null(v, what);
        else skip_value(v);
    });
}
void xy(Cur& c, double* out, const char* what) {  // r2.Point
    members(c, what, [&](const std::string& k, Cur& v) {
        if (k == "X") out[0] = number_or_null(v, what);
        else if (k == "Y") out[1] = number_or_null(v, what);
        else skip_value(v);
    });
}
std::string text_or_null(Cur& c) {
    std::string s;
    if (!eat_null(c)) read_string(c, &s);
    return s;
}

const char* const kKnownTypes[] = {"Cylinder", "TextureImage", "TransformedShape", "CalibratedCamera", "TriangleUV", "Dielectric", "BVH",
                                   "Group", "DebugNormal", "Metal", "PointLight", "Sphere", "TextureUniform", "Quad", "Lambertian",
                                   "OrthographicCamera", "TextureCheckerboard", "Mesh", "Face", "DebugUV", "Emitter",
                                   // written by the reference, never registered by it
                                   "PinholeCamera", "FocusableCamera", "Triangle"};

struct Typed {
    std::string type;
    Cur data;  // positioned at the value to read as `type`
    bool has_data = false;
};

// {"Data": ..., "Type": "..."} (unmarshalInterface, json.go:82-100).
Typed wrapper(Cur& c) {
    Typed t;
    if (peek(c) != '{') bad("json: cannot unmarshal a value that is not an object into Go value of type struct { Type string; Data json.RawMessage }");
    members(c, "a typed value", [&](const std::string& k, Cur& v) {
        if (k == "Type") {
            t.type = text_or_null(v);
        } else if (k == "Data") {
            t.data = v;
            t.has_data = true;
            skip_value(v);
        } else {
            skip_value(v);
        }
    });
    bool known = false;
    for (const char* n : kKnownTypes) known = known || t.type == n;
    if (!known) bad("unsupported type: `" + t.type + "`; has it been registered?");
    if (!t.has_data) {
        static const char kNull[] = "null";
        t.data = Cur{kNull, kNull + 4, kNull};
    }
    return t;
}

// An interface field that may be wrapped, carry a "Type" member of its own, or be bare; guess() types a bare one by its keys.
template <typename G>
Typed field(Cur& c, const char* what, G guess) {
    Cur start = c;
    if (peek(c) != '{') bad(std::string(what) + ": expected an object");
    std::set<std::string> keys;
    std::string own_type;
    bool type_is_text = false;
    members(c, what, [&](const std::string& k, Cur& v) {
        keys.insert(k);
        if (k == "Type" && peek(v) == '"') {
            read_string(v, &own_type);
            type_is_text = true;
        } else {
            skip_value(v);
        }
    });
    bool only_wrapper_keys = true;
    for (const std::string& k : keys) only_wrapper_keys = only_wrapper_keys && (k == "Type" || k == "Data");
    if (keys.count("Data") && only_wrapper_keys) {
        Cur again = start;
        return wrapper(again);
    }
    Typed t;
    t.data = start;
    t.has_data = true;
    if (type_is_text) {
        t.type = own_type;
        return t;
    }
    const char* g = guess(keys);
    if (!g) {
        std::string all;
        for (const std::string& k : keys) all += (all.empty() ? "" : ", ") + k;
        bad(std::string(what) + ": cannot tell the type of an untagged value with keys [" + all + "]");
    }
    t.type = g;
    return t;
}

void expect_type(const std::string& got, const char* name) {  // the checks of texture_uniform.go:49-51 and its like
    if (got != name) bad(std::string("invalid type: expected ") + name + ", got " + got);
}

int b64_value(unsigned char ch) {
    if (ch >= 'A' && ch <= 'Z') return ch - 'A';
    if (ch >= 'a' && ch <= 'z') return ch - 'a' + 26;
    if (ch >= '0' && ch <= '9') return ch - '0' + 52;
    if (ch == '+') return 62;
    if (ch == '/') return 63;
    return -1;
}
std::vector<uint8_t> base64(const std::string& s) {  // []byte as encoding/json writes it: standard alphabet, padded
    std::vector<uint8_t> out;
    out.reserve(s.size() / 4 * 3);
    uint32_t acc = 0;
    int bits = 0;
    for (unsigned char ch : s) {
        if (ch == '=' || ch == '\n' || ch == '\r') continue;
        int v = b64_value(ch);
        if (v < 0) bad("illegal base64 data in TextureImage.Image.Pix");
        acc = (acc << 6) | uint32_t(v);
        bits += 6;
        if (bits >= 8) {
            bits -= 8;
            out.push_back(uint8_t(acc >> bits));
        }
    }
    return out;
}

struct Builder {
    std::vector<rt_node> nodes;
    std::vector<rt_sphere> spheres;
    std::vector<rt_quad> quads;
    std::vector<rt_cylinder> cylinders;
    std::vector<rt_triangle> triangles;
    std::vector<rt_mesh> meshes;
    std::vector<rt_face> faces;
    std::vector<rt_transformed> transformed;
    std::vector<rt_material> materials;
    std::vector<rt_texture> textures;
    std::vector<rt_image> images;
    std::vector<std::vector<uint8_t>> pixels;  // images[i].pixels points into pixels[i]
    std::vector<rt_light> lights;
    std::vector<rt_camera_desc> cameras;
    rt_render_opts opts{};

    // -- textures --
    static const char* guess_texture(const std::set<std::string>& k) {
        if (k.count("Color")) return "TextureUniform";
        if (k.count("Odd") || k.count("Even")) return "TextureCheckerboard";
        if (k.count("Image") || k.count("FilePath")) return "TextureImage";
        return nullptr;
    }
    // image.NRGBA / image.RGBA64 as encoding/json writes the struct: Pix (base64), Stride, Rect{Min, Max}
    int image(Cur& c) {
        std::string pix;
        int64_t stride = 0, x0 = 0, y0 = 0, x1 = 0, y1 = 0;
        members(c, "image.Image", [&](const std::string& k, Cur& v) {
            if (k == "Pix") pix = text_or_null(v);
            else if (k == "Stride") stride = integer(v, "Stride");
            else if (k == "Rect") {
                members(v, "image.Rectangle", [&](const std::string& rk, Cur& rv) {
                    bool mn = rk == "Min";
                    if (!mn && rk != "Max") return skip_value(rv);
                    members(rv, "image.Point", [&](const std::string& pk, Cur& pv) {
                        if (pk == "X") (mn ? x0 : x1) = integer(pv, "X");
                        else if (pk == "Y") (mn ? y0 : y1) = integer(pv, "Y");
                        else skip_value(pv);
                    });
                });
            } else {
                skip_value(v);
            }
        });
        std::vector<uint8_t> raw = base64(pix);
        int64_t w = x1 - x0, h = y1 - y0, n = int64_t(raw.size());
        // every product below stays far inside int64: w, h <= 65536 and stride <= n (the decoded bytes: at most the text's length)
        if (w <= 0 || h <= 0 || w > 65536 || h > 65536 || stride < 4 * w || stride > n || (h - 1) > (n - 4 * w) / stride)
            bad("TextureImage.Image: " + std::to_string(n) + " bytes of pixels for " + std::to_string(w) + "x" + std::to_string(h) +
                ", stride " + std::to_string(stride));
        bool wide = stride >= 8 * w && n >= 8 * w && (h - 1) <= (n - 8 * w) / stride;  // 16 bits per channel, big-endian (image.RGBA64)
        std::vector<uint8_t> px(size_t(w) * size_t(h) * (wide ? 8 : 4));
        for (int64_t y = 0; y < h; y++) {
            const uint8_t* row = raw.data() + y * stride;
            if (!wide) {
                std::memcpy(px.data() + y * w * 4, row, size_t(w) * 4);
            } else {
                uint16_t* dst = reinterpret_cast<uint16_t*>(px.data()) + y * w * 4;
                for (int64_t i = 0; i < 4 * w; i++) dst[i] = uint16_t(row[2 * i] << 8 | row[2 * i + 1]);
            }
        }
        rt_image im{};
        im.width = int32_t(w);
        im.height = int32_t(h);
        im.format = wide ? RT_IMG_RGBA16 : RT_IMG_NRGBA8;
        pixels.push_back(std::move(px));
        images.push_back(im);  // the pointers are set once nothing is appended any more
        return int(images.size()) - 1;
    }
    // Textures and shapes nest (TextureCheckerboard.Odd / Even, TransformedShape.Shape): a text may not drive the reader's
    // own recursion deeper than this (encoding/json gives up at 10000 levels; rt_scene_create takes 4 for shapes anyway).
    static constexpr int kMaxNesting = 64;
    int nesting = 0;
    struct Nest {
        int& n;
        explicit Nest(int& depth) : n(depth) {
            if (++n > kMaxNesting) { --n; bad("scene JSON: values nested deeper than " + std::to_string(kMaxNesting)); }
        }
        ~Nest() { --n; }
    };
    int texture(Typed t) {
        Nest guard(nesting);
        rt_texture x{};
        Cur c = t.data;
        if (t.type == "TextureUniform") {
            std::string own;
            x.kind = RT_TEX_UNIFORM;
            members(c, "TextureUniform", [&](const std::string& k, Cur& v) {
                if (k == "Type") own = text_or_null(v);
                else if (k == "Color") xyz(v, x.color, "TextureUniform.Color");
                else skip_value(v);
            });
            expect_type(own, "TextureUniform");
        } else if (t.type == "TextureCheckerboard") {
            std::string own;
            x.kind = RT_TEX_CHECKERBOARD;
            x.odd = x.even = -1;
            Cur odd{}, even{};  // read once the tag has been checked, as texture_checkerboard.go:107-135 does
            members(c, "TextureCheckerboard", [&](const std::string& k, Cur& v) {
                if (k == "Type") own = text_or_null(v);
                else if (k == "Frequency") x.frequency = number_or_null(v, "TextureCheckerboard.Frequency");
                else {
                    if (k == "Odd") odd = v;
                    if (k == "Even") even = v;
                    skip_value(v);
                }
            });
            expect_type(own, "TextureCheckerboard");
            if (!odd.p || !even.p) bad("TextureCheckerboard: Odd and Even must both be set");
            x.odd = texture(wrapper(odd));
            x.even = texture(wrapper(even));
        } else if (t.type == "TextureImage") {
            x.kind = RT_TEX_IMAGE;
            x.image = -1;
            std::string path;
            members(c, "TextureImage", [&](const std::string& k, Cur& v) {
                if (k == "Image") {
                    if (!eat_null(v)) x.image = image(v);
                } else if (k == "FilePath") path = text_or_null(v);
                else if (k == "Interp") x.interp = text_or_null(v) == "bilinear" ? RT_INTERP_BILINEAR : RT_INTERP_NEAREST;
                else if (k == "WrapMode") x.wrap = text_or_null(v) == "clamp" ? RT_WRAP_CLAMP : RT_WRAP_REPEAT;
                else skip_value(v);
            });
            if (x.image < 0)
                bad("TextureImage" + (path.empty() ? std::string() : " (" + path + ")") +
                        " carries no pixels: the library reads no image files, the host must decode FilePath and pass the pixels",
                    RT_ERR_UNSUPPORTED);
        } else {
            bad("expected Texture, got *phys." + t.type);
        }
        textures.push_back(x);
        return int(textures.size()) - 1;
    }

    // -- materials --
    int material(Typed t) {
        rt_material m{};
        m.texture = -1;
        Cur c = t.data;
        if (t.type == "Emitter") {
            std::string own;
            m.kind = RT_MAT_EMITTER;
            Cur tex{};  // read once the tag has been checked, as material_emitter.go:50-68 does
            members(c, "Emitter", [&](const std::string& k, Cur& v) {
                if (k == "Type") own = text_or_null(v);
                else {
                    if (k == "Texture") tex = v;
                    skip_value(v);
                }
            });
            expect_type(own, "Emitter");
            if (!tex.p) bad("Emitter.Texture is missing");
            m.texture = texture(wrapper(tex));
        } else if (t.type == "Lambertian") {
            m.kind = RT_MAT_LAMBERTIAN;
            members(c, "Lambertian", [&](const std::string& k, Cur& v) {
                if (k == "Texture" && !eat_null(v)) m.texture = texture(field(v, "Texture", guess_texture));
                else if (k != "Texture") skip_value(v);
            });
            if (m.texture < 0) bad("Lambertian.Texture is nil");
        } else if (t.type == "Metal") {
            m.kind = RT_MAT_METAL;
            members(c, "Metal", [&](const std::string& k, Cur& v) {
                if (k == "Albedo") xyz(v, m.albedo, "Metal.Albedo");
                else if (k == "Fuzz") m.fuzz = number_or_null(v, "Metal.Fuzz");
                else skip_value(v);
            });
        } else if (t.type == "Dielectric") {
            m.kind = RT_MAT_DIELECTRIC;
            members(c, "Dielectric", [&](const std::string& k, Cur& v) {
                if (k == "RefractiveIndexInterior") m.ior_interior = number_or_null(v, "Dielectric.RefractiveIndexInterior");
                else if (k == "RefractiveIndexExterior") m.ior_exterior = number_or_null(v, "Dielectric.RefractiveIndexExterior");
                else if (k == "Roughness") m.roughness = number_or_null(v, "Dielectric.Roughness");
                else skip_value(v);
            });
        } else if (t.type == "DebugUV" || t.type == "DebugNormal") {
            m.kind = t.type == "DebugUV" ? RT_MAT_DEBUG_UV : RT_MAT_DEBUG_NORMAL;
            skip_value(c);
        } else {
            bad("expected Material, got *phys." + t.type);  // node.go:89-92
        }
        materials.push_back(m);
        return int(materials.size()) - 1;
    }

    // -- shapes --
    static const char* guess_shape(const std::set<std::string>& k) {
        if (k.count("Face")) return "Mesh";
        if (k.count("Vertex")) return "Face";
        if (k.count("Shape")) return "TransformedShape";
        if (k.count("UV0")) return "TriangleUV";
        if (k.count("P0")) return "Triangle";
        if (k.count("Normal")) return "Quad";
        if (k.count("Direction")) return "Cylinder";
        if (k.count("Center")) return "Sphere";
        return nullptr;
    }
    void face(Cur& c, rt_face& f) {  // Face{Vertex [3]Vertex{Position, UV}}, shape_mesh.go:12-21
        members(c, "Face", [&](const std::string& k, Cur& v) {
            if (k != "Vertex") return skip_value(v);
            int i = 0;
            elements(v, "Face.Vertex", [&](Cur& e) {
                if (i >= 3) return skip_value(e);  // encoding/json drops what does not fit a fixed-size array
                double* dst = f.v[i++];
                members(e, "Vertex", [&](const std::string& vk, Cur& vv) {
                    if (vk == "Position") xyz(vv, dst, "Vertex.Position");
                    else if (vk == "UV") xy(vv, dst + 3, "Vertex.UV");
                    else skip_value(vv);
                });
            });
        });
    }
    void shape(Typed t, int32_t& kind, int32_t& index) {
        Nest guard(nesting);
        Cur c = t.data;
        if (t.type == "Sphere") {
            rt_sphere s{};
            members(c, "Sphere", [&](const std::string& k, Cur& v) {
                if (k == "Center") xyz(v, s.center, "Sphere.Center");
                else if (k == "Radius") s.radius = number_or_null(v, "Sphere.Radius");
                else if (k == "UVMap") s.uvmap = int32_t(integer(v, "Sphere.UVMap"));
                else skip_value(v);
            });
            spheres.push_back(s);
            kind = RT_SHAPE_SPHERE, index = int32_t(spheres.size()) - 1;
        } else if (t.type == "Quad") {
            rt_quad q{};
            members(c, "Quad", [&](const std::string& k, Cur& v) {
                if (k == "Center") xyz(v, q.center, "Quad.Center");
                else if (k == "Normal") xyz(v, q.normal, "Quad.Normal");
                else if (k == "Width") q.width = number_or_null(v, "Quad.Width");
                else if (k == "Height") q.height = number_or_null(v, "Quad.Height");
                else skip_value(v);
            });
            quads.push_back(q);
            kind = RT_SHAPE_QUAD, index = int32_t(quads.size()) - 1;
        } else if (t.type == "Cylinder") {
            rt_cylinder cy{};
            members(c, "Cylinder", [&](const std::string& k, Cur& v) {
                if (k == "Origin") xyz(v, cy.origin, "Cylinder.Origin");
                else if (k == "Direction") xyz(v, cy.direction, "Cylinder.Direction");
                else if (k == "Radius") cy.radius = number_or_null(v, "Cylinder.Radius");
                else if (k == "Height") cy.height = number_or_null(v, "Cylinder.Height");
                else skip_value(v);
            });
            cylinders.push_back(cy);
            kind = RT_SHAPE_CYLINDER, index = int32_t(cylinders.size()) - 1;
        } else if (t.type == "Triangle" || t.type == "TriangleUV") {
            rt_triangle tr{};
            tr.kind = t.type == "Triangle" ? RT_TRI_TRIANGLE : RT_TRI_TRIANGLE_UV;
            members(c, "Triangle", [&](const std::string& k, Cur& v) {
                if (k.size() == 2 && k[0] == 'P' && k[1] >= '0' && k[1] <= '2') xyz(v, tr.p[k[1] - '0'], "Triangle.P");
                else if (k.size() == 3 && k[0] == 'U' && k[1] == 'V' && k[2] >= '0' && k[2] <= '2') xy(v, tr.uv[k[2] - '0'], "TriangleUV.UV");
                else if (k == "Normal") xyz(v, tr.normal, "TriangleUV.Normal");
                else skip_value(v);
            });
            triangles.push_back(tr);
            kind = RT_SHAPE_TRIANGLE, index = int32_t(triangles.size()) - 1;
        } else if (t.type == "Face") {
            rt_face f{};
            face(c, f);
            rt_triangle tr{};
            tr.kind = RT_TRI_FACE;
            for (int i = 0; i < 3; i++) {
                std::memcpy(tr.p[i], f.v[i], 3 * sizeof(double));
                std::memcpy(tr.uv[i], f.v[i] + 3, 2 * sizeof(double));
            }
            triangles.push_back(tr);
            kind = RT_SHAPE_TRIANGLE, index = int32_t(triangles.size()) - 1;
        } else if (t.type == "Mesh") {
            rt_mesh m{};
            m.first_face = faces.size();
            members(c, "Mesh", [&](const std::string& k, Cur& v) {
                if (k == "Face") {
                    elements(v, "Mesh.Face", [&](Cur& e) {
                        faces.emplace_back();
                        std::memset(&faces.back(), 0, sizeof(rt_face));
                        face(e, faces.back());
                    });
                } else if (k == "BVH" && !eat_null(v)) {
                    // (*BVH).UnmarshalJSON checks the tag (shape_bvh.go:311-313); the tree is rebuilt from the faces
                    std::string own;
                    members(v, "BVH", [&](const std::string& bk, Cur& bv) {
                        if (bk == "Type") own = text_or_null(bv);
                        else skip_value(bv);
                    });
                    expect_type(own, "BVH");
                } else if (k != "BVH") {
                    skip_value(v);
                }
            });
            m.n_faces = faces.size() - m.first_face;
            meshes.push_back(m);
            kind = RT_SHAPE_MESH, index = int32_t(meshes.size()) - 1;
        } else if (t.type == "TransformedShape") {
            rt_transformed x{};
            x.child_kind = -1;
            members(c, "TransformedShape", [&](const std::string& k, Cur& v) {
                if (k == "Shape") {
                    if (!eat_null(v)) shape(field(v, "Shape", guess_shape), x.child_kind, x.child_index);
                } else if (k == "Transform") {
                    members(v, "Transform", [&](const std::string& tk, Cur& tv) {
                        if (tk == "Translation") xyz(tv, x.translation, "Transform.Translation");
                        else if (tk == "Scale") xyz(tv, x.scale, "Transform.Scale");
                        else if (tk == "Rotation") {
                            members(tv, "r3.Mat3x3", [&](const std::string& mk, Cur& mv) {
                                if (mk != "M") return skip_value(mv);
                                int row = 0;
                                elements(mv, "Mat3x3.M", [&](Cur& rv) {
                                    if (row >= 3) return skip_value(rv);
                                    int col = 0, r = row++;
                                    elements(rv, "Mat3x3.M", [&](Cur& cv) {
                                        if (col >= 3) return skip_value(cv);
                                        x.rotation[3 * r + col++] = number_or_null(cv, "Mat3x3.M");
                                    });
                                });
                            });
                        } else {
                            skip_value(tv);
                        }
                    });
                } else {
                    skip_value(v);
                }
            });
            if (x.child_kind < 0) bad("TransformedShape.Shape is nil");
            transformed.push_back(x);
            kind = RT_SHAPE_TRANSFORMED, index = int32_t(transformed.size()) - 1;
        } else if (t.type == "BVH" || t.type == "Group") {
            bad(t.type + " as a node's shape is not supported: it is read only inside a Mesh", RT_ERR_UNSUPPORTED);
        } else {
            bad("expected Shape, got *phys." + t.type);  // node.go:80-83
        }
    }

    // -- cameras, lights, nodes --
    void look(const std::string& k, Cur& v, rt_camera_desc& d, bool& taken) {
        taken = true;
        if (k == "LookFrom") xyz(v, d.look_from, "LookFrom");
        else if (k == "LookAt") xyz(v, d.look_at, "LookAt");
        else if (k == "VUp") xyz(v, d.vup, "VUp");
        else taken = false;
    }
    void camera(Typed t) {
        rt_camera_desc d{};
        Cur c = t.data;
        if (t.type == "OrthographicCamera" || t.type == "FocusableCamera") {
            d.kind = t.type == "OrthographicCamera" ? RT_CAM_ORTHOGRAPHIC : RT_CAM_FOCUSABLE;
            members(c, "camera", [&](const std::string& k, Cur& v) {
                bool taken;
                look(k, v, d, taken);
                if (taken) return;
                if (k == "FOVHeight") d.fov_height = number_or_null(v, "FOVHeight");
                else if (k == "FOVWidth") d.fov_width = number_or_null(v, "FOVWidth");
                else if (k == "Aperture") d.aperture = number_or_null(v, "Aperture");
                else if (k == "WorkingDistance") d.working_distance = number_or_null(v, "WorkingDistance");
                else skip_value(v);
            });
        } else if (t.type == "PinholeCamera") {
            d.kind = RT_CAM_PINHOLE;
            members(c, "PinholeCamera", [&](const std::string& k, Cur& v) {
                if (k == "LowerLeftCorner") xyz(v, d.lower_left_corner, "LowerLeftCorner");
                else if (k == "Origin") xyz(v, d.origin, "Origin");
                else if (k == "Horizontal") xyz(v, d.horizontal, "Horizontal");
                else if (k == "Vertical") xyz(v, d.vertical, "Vertical");
                else skip_value(v);
            });
        } else if (t.type == "CalibratedCamera") {
            d.kind = RT_CAM_CALIBRATED;
            members(c, "CalibratedCamera", [&](const std::string& k, Cur& v) {
                if (k == "Extrinsics") {
                    members(v, "CameraExtrinsics", [&](const std::string& ek, Cur& ev) {
                        bool taken;
                        look(ek, ev, d, taken);
                        if (!taken) skip_value(ev);
                    });
                } else if (k == "Intrinsics") {
                    members(v, "CameraIntrinsics", [&](const std::string& ik, Cur& iv) {
                        struct { const char* name; double* dst; } f[] = {{"Fx", &d.fx}, {"Fy", &d.fy}, {"Cx", &d.cx}, {"Cy", &d.cy},
                                                                         {"K1", &d.k1}, {"K2", &d.k2}, {"P1", &d.p1}, {"P2", &d.p2},
                                                                         {"K3", &d.k3}, {"K4", &d.k4}, {"K5", &d.k5}, {"K6", &d.k6}};
                        if (ik == "Width") { d.width = int32_t(integer(iv, "CameraIntrinsics.Width")); return; }
                        if (ik == "Height") { d.height = int32_t(integer(iv, "CameraIntrinsics.Height")); return; }
                        for (auto& e : f)
                            if (ik == e.name) { *e.dst = number_or_null(iv, e.name); return; }
                        skip_value(iv);
                    });
                } else {
                    skip_value(v);
                }
            });
        } else {
            bad("expected Camera, got *phys." + t.type);
        }
        cameras.push_back(d);
    }
    void light(Typed t) {
        if (t.type != "PointLight") bad("expected Light, got *phys." + t.type);
        rt_light l{};
        Cur c = t.data;
        members(c, "PointLight", [&](const std::string& k, Cur& v) {
            if (k == "Position") xyz(v, l.position, "PointLight.Position");
            else if (k == "RadiantIntensity") xyz(v, l.intensity, "PointLight.RadiantIntensity");
            else skip_value(v);
        });
        lights.push_back(l);
    }
    void node(Cur& c, std::set<std::string>& names) {  // Node.UnmarshalJSON, node.go:66-97
        rt_node n{};
        n.shape_kind = n.material = -1;
        std::string name;
        if (peek(c) != '{') bad("json: cannot unmarshal a node that is not an object");
        members(c, "Node", [&](const std::string& k, Cur& v) {
            if (k == "Shape") shape(wrapper(v), n.shape_kind, n.shape_index);
            else if (k == "Material") n.material = material(wrapper(v));
            else if (k == "Name") name = text_or_null(v);
            else skip_value(v);
        });
        if (n.shape_kind < 0 || n.material < 0) bad("node " + name + ": Shape and Material must both be present");
        if (!names.insert(name).second) bad("duplicate node name: " + name);  // scene.go:51-54
        nodes.push_back(n);
    }
    void scene(Cur& c) {  // Scene.UnmarshalJSON, scene.go:117-159
        if (peek(c) != '{') bad("json: cannot unmarshal a scene that is not an object");
        std::set<std::string> names;
        members(c, "Scene", [&](const std::string& k, Cur& v) {
            if (k == "Camera") elements(v, "Scene.Camera", [&](Cur& e) { camera(wrapper(e)); });
            else if (k == "Light") elements(v, "Scene.Light", [&](Cur& e) { light(wrapper(e)); });
            else if (k == "Node") elements(v, "Scene.Node", [&](Cur& e) { node(e, names); });
            else if (k == "RenderOptions") {
                members(v, "RenderOptions", [&](const std::string& ok, Cur& ov) {
                    if (ok == "Seed") opts.seed = integer(ov, "RenderOptions.Seed");
                    else if (ok == "RaysPerPixel") opts.rays_per_pixel = int32_t(integer(ov, "RenderOptions.RaysPerPixel"));
                    else if (ok == "MaxRayDepth") opts.max_ray_depth = int32_t(integer(ov, "RenderOptions.MaxRayDepth"));
                    else if (ok == "Dx") opts.dx = int32_t(integer(ov, "RenderOptions.Dx"));
                    else if (ok == "Dy") opts.dy = int32_t(integer(ov, "RenderOptions.Dy"));
                    else skip_value(ov);
                });
            } else {
                skip_value(v);
            }
        });
        ws(c);
        if (c.p != c.end) bad_char(c, "after top-level value");
    }
};

}  // namespace

extern "C" int rt_scene_create_from_json(const char* text, size_t len, rt_scene** out, rt_json_scene_info* info) {
    if (!text || !out) return rt::fail(RT_ERR_INVALID, "rt_scene_create_from_json: null argument");
    *out = nullptr;
    Builder b;
    try {
        Cur c{text, text + len, text};
        b.scene(c);
    } catch (const JsonError& e) {
        return rt::fail(e.code, "scene JSON: " + e.msg);
    } catch (const std::bad_alloc&) {
        return rt::fail(RT_ERR_NOMEM, "scene JSON: out of host memory");
    } catch (const std::exception& e) {  // nothing may cross the C boundary (length_error of a container, system_error, ...)
        return rt::fail(RT_ERR_INVALID, std::string("scene JSON: ") + e.what());
    }
    for (size_t i = 0; i < b.images.size(); i++) b.images[i].pixels = b.pixels[i].data();
    rt_scene_desc d{};
    d.nodes = b.nodes.data(), d.n_nodes = b.nodes.size();
    d.spheres = b.spheres.data(), d.n_spheres = b.spheres.size();
    d.quads = b.quads.data(), d.n_quads = b.quads.size();
    d.cylinders = b.cylinders.data(), d.n_cylinders = b.cylinders.size();
    d.triangles = b.triangles.data(), d.n_triangles = b.triangles.size();
    d.meshes = b.meshes.data(), d.n_meshes = b.meshes.size();
    d.faces = b.faces.data(), d.n_faces = b.faces.size();
    d.transformed = b.transformed.data(), d.n_transformed = b.transformed.size();
    d.materials = b.materials.data(), d.n_materials = b.materials.size();
    d.textures = b.textures.data(), d.n_textures = b.textures.size();
    d.images = b.images.data(), d.n_images = b.images.size();
    d.lights = b.lights.data(), d.n_lights = b.lights.size();
    int rc = rt_scene_create(&d, out);
    if (rc != RT_OK) return rc;
    if (info) {
        std::memset(info, 0, sizeof(*info));
        info->opts = b.opts;
        info->opts.mode = RT_MODE_FAST;
        info->opts.band_world = 1;
        info->n_cameras = b.cameras.size();
        info->n_nodes = b.nodes.size();
        info->n_lights = b.lights.size();
        for (size_t i = 0; i < b.cameras.size() && i < sizeof(info->cameras) / sizeof(info->cameras[0]); i++) info->cameras[i] = b.cameras[i];
    }
    return RT_OK;
}
