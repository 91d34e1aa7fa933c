// Native scanner for the geometry of a Wavefront OBJ text: the statements the reference's obj package reads
// line by line (obj/obj.go:236-468: v, vt, vn, f, usemtl, mtllib; everything else is ignored), for files of
// millions of lines.  Host code only; it needs no GPU and no rt_init.
//
// The scanner handles exactly the well-formed subset and says so: at the first line it is not sure to read the way
// the line parser does (a number that is not plain decimal notation, fields not separated by single spaces, an index
// out of range, a face of fewer than three corners, bytes outside printable ASCII, ...) it stops and reports that
// line; the caller then runs the line parser over the whole text, so every error message and every corner case is
// the line parser's.
#include <charconv>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/rt_hip.h"

namespace {

struct Owner {
    std::vector<double> vertices, texcoords, normals;
    std::vector<int32_t> corners;
    std::vector<uint64_t> face_start;
    std::vector<int32_t> face_event;
    std::vector<rt_obj_event> events;
};

inline bool is_digit(char c) { return c >= '0' && c <= '9'; }

// [+-]?(digits[.digits*]|.digits)([eE][+-]?digits)? exactly filling [p, e): the notations float() of the line parser and
// strtod agree on (both round correctly).  false: let the line parser decide.
bool plain_decimal(const char* p, const char* e) {
    if (p < e && (*p == '+' || *p == '-')) p++;
    const char* d0 = p;
    while (p < e && is_digit(*p)) p++;
    size_t digits = (size_t)(p - d0);
    if (p < e && *p == '.') {
        p++;
        const char* f0 = p;
        while (p < e && is_digit(*p)) p++;
        digits += (size_t)(p - f0);
    }
    if (digits == 0) return false;
    if (p < e && (*p == 'e' || *p == 'E')) {
        p++;
        if (p < e && (*p == '+' || *p == '-')) p++;
        const char* x0 = p;
        while (p < e && is_digit(*p)) p++;
        if (p == x0) return false;
    }
    return p == e;
}

bool parse_double(const char* p, const char* e, double& out) {
    if (!plain_decimal(p, e)) return false;
    if (*p == '+') p++;  // from_chars takes no plus sign
    const std::from_chars_result r = std::from_chars(p, e, out, std::chars_format::general);  // correctly rounded, like strtod / float()
    if (r.ec == std::errc::result_out_of_range) return false;  // 1e400: float() gives inf, leave it to the line parser
    return r.ec == std::errc() && r.ptr == e;
}

// [+-]?digits, at most 9 digits (fits int32 with room for the negative-index arithmetic)
bool parse_int(const char* p, const char* e, long long& out) {
    bool neg = false;
    if (p < e && (*p == '+' || *p == '-')) { neg = *p == '-'; p++; }
    if (p == e || e - p > 9) return false;
    long long v = 0;
    for (; p < e; p++) {
        if (!is_digit(*p)) return false;
        v = v * 10 + (*p - '0');
    }
    out = neg ? -v : v;
    return true;
}

// resolveIndex (obj.go:431-439): negative counts back from the end; 0 (false) = out of range
bool resolve(long long v, size_t size, int32_t& out) {
    if (v < 0) v = (long long)size + v + 1;
    if (v < 1 || v > (long long)size) return false;
    out = (int32_t)v;
    return true;
}

// `count` fields separated by single spaces filling [p, e) (strings.SplitN(rest, " ", count) with every field a
// plain number and nothing left over)
bool numbers(const char* p, const char* e, int count, double* out) {
    for (int i = 0; i < count; i++) {
        const char* q = p;
        while (q < e && *q != ' ') q++;
        if (i + 1 < count && q == e) return false;
        if (i + 1 == count && q != e) return false;  // more text after the last field
        if (!parse_double(p, q, out[i])) return false;
        p = q + 1;
    }
    return true;
}

}  // namespace

extern "C" {

int rt_obj_scan_text(const char* text, uint64_t len, rt_obj_scan* out) {
    if (!out || (!text && len)) return RT_ERR_INVALID;
    std::memset(out, 0, sizeof(*out));
    Owner* own = new (std::nothrow) Owner();
    if (!own) return RT_ERR_NOMEM;
    out->owner = own;
    own->face_start.push_back(0);
    int32_t current_event = -1;  // usemtl in force
    uint32_t line_no = 0;
    const char* p = text;
    const char* const end = text + len;
    try {
        while (p <= end) {
            const char* nl = static_cast<const char*>(std::memchr(p, '\n', (size_t)(end - p)));
            const char* le = nl ? nl : end;
            line_no++;
            const char* a = p;
            const char* b = le;
            // str.strip() of the line parser: here blanks, tabs and carriage returns; a statement holding any other byte
            // outside printable ASCII goes to the line parser (whose strip() knows more kinds of white space)
            while (a < b && (*a == ' ' || *a == '\t' || *a == '\r')) a++;
            while (b > a && (b[-1] == ' ' || b[-1] == '\t' || b[-1] == '\r')) b--;
            const char* next = nl ? nl + 1 : end + 1;
            if (a == b || *a == '#') { p = next; continue; }  // blank line or comment (obj.go:236-241)
            bool ascii = true;
            for (const char* q = a; q < b; q++)
                if ((unsigned char)*q >= 0x7f || (unsigned char)*q < 0x20) { ascii = false; break; }
            if (!ascii) { out->fallback_line = line_no; break; }
            const char* sp = static_cast<const char*>(std::memchr(a, ' ', (size_t)(b - a)));
            if (!sp) { p = next; continue; }  // a lone token: ignored (obj.go:242-246)
            const size_t dl = (size_t)(sp - a);
            const char* rest = sp + 1;
            bool ok = true;
            if (dl == 1 && a[0] == 'v') {  // obj.go:270-290
                double v[3];
                ok = numbers(rest, b, 3, v);
                if (ok) own->vertices.insert(own->vertices.end(), v, v + 3);
            } else if (dl == 2 && a[0] == 'v' && a[1] == 'n') {  // obj.go:312-332
                double v[3];
                ok = numbers(rest, b, 3, v);
                if (ok) own->normals.insert(own->normals.end(), v, v + 3);
            } else if (dl == 2 && a[0] == 'v' && a[1] == 't') {  // obj.go:293-309
                double v[2];
                ok = numbers(rest, b, 2, v);
                if (ok) own->texcoords.insert(own->texcoords.end(), v, v + 2);
            } else if (dl == 1 && a[0] == 'f') {  // obj.go:335-428
                const size_t nv = own->vertices.size() / 3, nt = own->texcoords.size() / 2, nn = own->normals.size() / 3;
                const size_t before = own->corners.size();
                const char* q = rest;
                int corners = 0;
                while (ok) {
                    while (q < b && *q == ' ') q++;  // strings.Fields
                    if (q == b) break;
                    const char* t = q;
                    while (t < b && *t != ' ') t++;
                    const char* s1 = static_cast<const char*>(std::memchr(q, '/', (size_t)(t - q)));
                    const char* s2 = s1 ? static_cast<const char*>(std::memchr(s1 + 1, '/', (size_t)(t - s1 - 1))) : nullptr;
                    if (s2 && std::memchr(s2 + 1, '/', (size_t)(t - s2 - 1))) { ok = false; break; }
                    long long iv = 0, it = 0, in = 0;
                    int32_t rv = 0, rt = 0, rn = 0;
                    ok = parse_int(q, s1 ? s1 : t, iv) && resolve(iv, nv, rv);
                    if (ok && s1 && !s2) ok = parse_int(s1 + 1, t, it) && resolve(it, nt, rt);  // v/vt: vt is required
                    if (ok && s2) {
                        if (s2 > s1 + 1) ok = parse_int(s1 + 1, s2, it) && resolve(it, nt, rt);
                        if (ok && t > s2 + 1) ok = parse_int(s2 + 1, t, in) && resolve(in, nn, rn);
                    }
                    if (ok) {
                        own->corners.push_back(rv); own->corners.push_back(rt); own->corners.push_back(rn);
                        corners++;
                    }
                    q = t;
                }
                if (ok && corners < 3) ok = false;
                if (ok) {
                    own->face_start.push_back(own->corners.size() / 3);
                    own->face_event.push_back(current_event);
                } else {
                    own->corners.resize(before);
                }
            } else if ((dl == 6 && std::memcmp(a, "usemtl", 6) == 0) || (dl == 6 && std::memcmp(a, "mtllib", 6) == 0)) {  // obj.go:442-468
                if (rest == b) ok = false;  // missing name: an error of the line parser
                if (ok) {
                    rt_obj_event ev;
                    ev.kind = a[0] == 'u' ? RT_OBJ_USEMTL : RT_OBJ_MTLLIB;
                    ev.line = line_no;
                    ev.line_off = (uint64_t)(a - text);
                    ev.line_len = (uint64_t)(b - a);
                    ev.arg_off = (uint64_t)(rest - text);
                    ev.arg_len = (uint64_t)(b - rest);
                    own->events.push_back(ev);
                    if (ev.kind == RT_OBJ_USEMTL) current_event = (int32_t)own->events.size() - 1;
                }
            }
            if (!ok) { out->fallback_line = line_no; break; }
            p = next;
        }
    } catch (const std::bad_alloc&) {
        delete own;
        std::memset(out, 0, sizeof(*out));
        return RT_ERR_NOMEM;
    }
    out->n_vertices = own->vertices.size() / 3;
    out->n_texcoords = own->texcoords.size() / 2;
    out->n_normals = own->normals.size() / 3;
    out->n_corners = own->corners.size() / 3;
    out->n_faces = own->face_event.size();
    out->n_events = own->events.size();
    out->vertices = own->vertices.data();
    out->texcoords = own->texcoords.data();
    out->normals = own->normals.data();
    out->corners = own->corners.data();
    out->face_start = own->face_start.data();
    out->face_event = own->face_event.data();
    out->events = own->events.data();
    return RT_OK;
}

void rt_obj_scan_free(rt_obj_scan* scan) {
    if (!scan) return;
    delete static_cast<Owner*>(scan->owner);
    std::memset(scan, 0, sizeof(*scan));
}

}  // extern "C"
