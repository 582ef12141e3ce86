/*
 * ASCII STL -> .bin converter (pure host code).
 *
 * Behaviour follows the reference script prepare_data/data_formatter.js:9-18 token for token,
 * because the .bin layout it writes is the renderer's data contract:
 *   - the text is split at '\n'; every line that CONTAINS "vertex" is kept (:13);
 *   - of such a line, the text between the first and the second occurrence of "vertex " is taken
 *     (`line.split("vertex ")[1]`); a kept line without "vertex " makes the script throw, here it
 *     is reported as PT_EINVAL;
 *   - that text is split at every single space; each piece goes through JavaScript's Number():
 *     surrounding white space (a trailing '\r' too) is ignored, an empty piece is 0, anything
 *     that is not a numeric literal is NaN.  Two consecutive spaces therefore yield an extra 0,
 *     exactly as in the script;
 *   - each number is multiplied by `scale` in double precision (:5,13) and stored as a
 *     little-endian f32 (`writeFloatLE`, :18): one rounding, to nearest.
 * Nothing checks that the count is a multiple of 9; the caller decides what to do with a ragged
 * tail (the script writes it out as it is).
 */
#include "../../include/pt_api.h"

#include <cerrno>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <new>
#include <string>
#include <vector>

namespace {

bool js_space(unsigned char c) {
    return c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '\v' || c == '\f';
}

/* ECMAScript StringToNumber for the inputs an STL file can hold */
double js_number(const char *s, size_t n) {
    while (n && js_space((unsigned char)s[0])) { s++; n--; }
    while (n && js_space((unsigned char)s[n - 1])) n--;
    if (n == 0) return 0.0;
    std::string t(s, n);
    const double nan = std::numeric_limits<double>::quiet_NaN();
    const char *p = t.c_str();
    bool neg = false;
    const char *body = p;
    if (*body == '+' || *body == '-') { neg = *body == '-'; body++; }
    if (strcmp(body, "Infinity") == 0) return neg ? -INFINITY : INFINITY;
    if (body == p && body[0] == '0' && (body[1] == 'x' || body[1] == 'X' || body[1] == 'b' || body[1] == 'B' || body[1] == 'o' || body[1] == 'O')) {
        int base = (body[1] == 'x' || body[1] == 'X') ? 16 : (body[1] == 'b' || body[1] == 'B') ? 2 : 8;
        if (!body[2]) return nan;
        double v = 0.0;
        for (const char *q = body + 2; *q; q++) {
            int d = (*q >= '0' && *q <= '9') ? *q - '0' : (*q >= 'a' && *q <= 'f') ? *q - 'a' + 10 : (*q >= 'A' && *q <= 'F') ? *q - 'A' + 10 : 99;
            if (d >= base) return nan;
            v = v * base + d;
        }
        return v;
    }
    /* decimal literal: digits [. digits] [e[+-]digits]; at least one digit in the mantissa */
    const char *q = body;
    bool digits = false;
    while (*q >= '0' && *q <= '9') { q++; digits = true; }
    if (*q == '.') { q++; while (*q >= '0' && *q <= '9') { q++; digits = true; } }
    if (!digits) return nan;
    if (*q == 'e' || *q == 'E') {
        const char *e = q + 1;
        if (*e == '+' || *e == '-') e++;
        if (!(*e >= '0' && *e <= '9')) return nan;
        while (*e >= '0' && *e <= '9') e++;
        q = e;
    }
    if (*q) return nan;
    return strtod(p, nullptr);     /* correctly rounded decimal -> double, like V8 */
}

int format_text(const char *text, size_t len, double scale, std::vector<float> &out, std::string &why) {
    size_t pos = 0, line_no = 0;
    while (pos <= len) {
        const char *line = text + pos;
        const char *nl = (const char *)memchr(line, '\n', len - pos);
        size_t n = nl ? (size_t)(nl - line) : len - pos;
        line_no++;
        std::string L(line, n);
        if (L.find("vertex") != std::string::npos) {
            size_t a = L.find("vertex ");
            if (a == std::string::npos) {
                why = "line " + std::to_string(line_no) + " contains \"vertex\" but not \"vertex \"";
                return PT_EINVAL;
            }
            size_t start = a + 7;
            size_t b = L.find("vertex ", start);
            std::string seg = L.substr(start, b == std::string::npos ? std::string::npos : b - start);
            size_t p = 0;
            for (;;) {
                size_t sp = seg.find(' ', p);
                size_t tn = (sp == std::string::npos ? seg.size() : sp) - p;
                out.push_back((float)(js_number(seg.data() + p, tn) * scale));
                if (sp == std::string::npos) break;
                p = sp + 1;
            }
        }
        if (!nl) break;
        pos += n + 1;
    }
    return PT_OK;
}

thread_local std::string g_fmt_error;

}  // namespace

extern "C" int pt_stl_format(const char *stl_text, size_t length, double scale, float **out, uint64_t *n_floats) {
    if (!out || !n_floats || (!stl_text && length)) { g_fmt_error = "null argument"; return PT_EINVAL; }
    try {
        std::vector<float> v;
        int rc = format_text(stl_text, length, scale, v, g_fmt_error);
        if (rc) return rc;
        float *buf = (float *)malloc((v.size() ? v.size() : 1) * sizeof(float));
        if (!buf) return PT_ENOMEM;
        if (!v.empty()) memcpy(buf, v.data(), v.size() * sizeof(float));
        *out = buf;
        *n_floats = v.size();
        return PT_OK;
    } catch (const std::bad_alloc &) {
        return PT_ENOMEM;
    }
}

extern "C" int pt_stl_format_file(const char *stl_path, const char *bin_path, double scale, uint64_t *n_floats) {
    if (!stl_path || !bin_path) { g_fmt_error = "null path"; return PT_EINVAL; }
    FILE *f = fopen(stl_path, "rb");
    if (!f) { g_fmt_error = std::string("cannot read ") + stl_path; return PT_EIO; }
    std::string text;
    char buf[1 << 16];
    size_t got;
    while ((got = fread(buf, 1, sizeof(buf), f)) > 0) text.append(buf, got);
    fclose(f);
    float *out = nullptr;
    uint64_t n = 0;
    int rc = pt_stl_format(text.data(), text.size(), scale, &out, &n);
    if (rc) return rc;
    FILE *o = fopen(bin_path, "wb");
    if (!o) { free(out); g_fmt_error = std::string("cannot write ") + bin_path; return PT_EIO; }
    size_t wrote = n ? fwrite(out, sizeof(float), (size_t)n, o) : 0;
    fclose(o);
    free(out);
    if (wrote != n) { g_fmt_error = std::string("short write to ") + bin_path; return PT_EIO; }
    if (n_floats) *n_floats = n;
    return PT_OK;
}

extern "C" const char *pt_stl_format_error(void) { return g_fmt_error.c_str(); }
