// Two steps before the path: Wireless InSite `*.paths.*.p2m` text -> the float32 ray
// matrices.  Replaces the line-by-line Python parse of
// deepmimo/converter/wireless_insite/p2m_parser.py:48-145 (paths_parser).  Host-only C++ over a file image
// (mmap'ed by the caller); no allocation, outputs are caller-provided [n_rx, max_paths] matrices that are
// NaN-filled here (p2m_parser.py:82-92).
//
// File layout the reference relies on: 21 header lines, a line with the receiver count (p2m_parser.py:36, 80),
// then per receiver "<rx index> <n paths>"; receivers with paths have one summary line, then per path:
//   "<path#> <n interactions> <power dBm> <phase deg> <toa s> <aoa theta> <aoa phi> <aod theta> <aod phi>"
//   "Tx-D-R-Rx"                                  interaction string
//   n interactions + 2 position lines            (Tx, each interaction, Rx)
// Numbers are parsed as the reference does (np.float32(str) = correctly rounded double, then float32).
// One deliberate difference: a receiver with more than max_paths paths has its surplus paths skipped
// properly; the reference stops advancing after the 25th path and mis-parses the rest of such a file.
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "../../include/deepmimo_amd.h"

namespace dmx { void set_error(const char* fmt, ...); }
using dmx::set_error;

namespace {

struct Cursor {
    const char* p;
    const char* end;
    long line = 0;
    // [b, e) of the next line without its terminator; false at end of input
    bool next(const char*& b, const char*& e) {
        if (p >= end) return false;
        b = p;
        const char* nl = (const char*)memchr(p, '\n', (size_t)(end - p));
        e = nl ? nl : end;
        p = nl ? nl + 1 : end;
        if (e > b && e[-1] == '\r') --e;
        ++line;
        return true;
    }
};

// whitespace-separated tokens of [b, e) as doubles; returns how many were read (up to cap)
int read_numbers(const char* b, const char* e, double* out, int cap) {
    int n = 0;
    char buf[64];
    while (b < e && n < cap) {
        while (b < e && (*b == ' ' || *b == '\t')) ++b;
        if (b >= e) break;
        const char* t = b;
        while (b < e && *b != ' ' && *b != '\t') ++b;
        size_t len = (size_t)(b - t);
        if (len >= sizeof(buf)) len = sizeof(buf) - 1;
        memcpy(buf, t, len);
        buf[len] = 0;
        char* endp = nullptr;
        const double v = strtod(buf, &endp);
        if (endp == buf) return -1;
        out[n++] = v;
    }
    return n;
}

// "Tx-D-R-Rx" -> digits per p2m_parser.py:38-46 joined into one number (0 for a direct path); -1 on an unknown code
double interaction_code(const char* b, const char* e) {
    // tokens between the first and the last '-'
    const char* first = (const char*)memchr(b, '-', (size_t)(e - b));
    if (!first) return 0.0;
    const char* last = e;
    while (last > first && last[-1] != '-') --last;
    if (last <= first + 1) return 0.0;                    // "Tx-Rx"
    double code = 0.0;
    const char* t = first + 1;
    const char* stop = last - 1;                           // position of the last '-'
    while (t < stop) {
        const char* u = t;
        while (u < stop && *u != '-') ++u;
        const size_t len = (size_t)(u - t);
        int digit = -1;
        if (len == 1 && *t == 'R') digit = 1;
        else if (len == 1 && *t == 'D') digit = 2;
        else if (len == 2 && t[0] == 'D' && t[1] == 'S') digit = 3;
        else if (len == 1 && (*t == 'T' || *t == 'F' || *t == 'X')) digit = 4;
        if (digit < 0) return -1.0;
        code = code * 10.0 + digit;
        t = u + 1;
    }
    return code;
}

}  // namespace

extern "C" {

int64_t dmx_p2m_count_rx(const char* text, size_t len) {
    if (!text) { set_error("p2m text is NULL"); return -1; }
    Cursor c{text, text + len};
    const char *b, *e;
    for (int i = 0; i < 21; ++i)
        if (!c.next(b, e)) { set_error("p2m file has fewer than 22 lines"); return -1; }
    if (!c.next(b, e)) { set_error("p2m file has fewer than 22 lines"); return -1; }
    double v;
    if (read_numbers(b, e, &v, 1) != 1 || v < 0) { set_error("p2m line 22 is not a receiver count"); return -1; }
    return (int64_t)v;
}

int dmx_p2m_parse_paths(const char* text, size_t len, int32_t max_paths, int32_t max_inter, int64_t n_rx,
                        float* aoa_az, float* aoa_el, float* aod_az, float* aod_el, float* delay, float* power,
                        float* phase, float* inter, float* inter_pos) {
    if (!text || max_paths < 1 || max_inter < 0 || n_rx < 0) { set_error("bad p2m parse arguments"); return DMX_ERR_ARG; }
    float* mats[8] = {aoa_az, aoa_el, aod_az, aod_el, delay, power, phase, inter};
    for (float* m : mats) if (!m && n_rx > 0) { set_error("a p2m output matrix is NULL"); return DMX_ERR_ARG; }
    const float nanf_ = nanf("");
    for (float* m : mats) for (int64_t i = 0; i < n_rx * max_paths; ++i) m[i] = nanf_;
    if (inter_pos) for (int64_t i = 0; i < n_rx * max_paths * max_inter * 3; ++i) inter_pos[i] = nanf_;

    Cursor c{text, text + len};
    const char *b, *e;
    for (int i = 0; i < 22; ++i)
        if (!c.next(b, e)) { set_error("p2m file has fewer than 22 lines"); return DMX_ERR_ARG; }
    double num[9];
    for (int64_t rx = 0; rx < n_rx; ++rx) {
        if (!c.next(b, e) || read_numbers(b, e, num, 2) != 2) { set_error("p2m line %ld: expected '<rx> <n paths>'", c.line); return DMX_ERR_ARG; }
        const long n_paths = (long)num[1];
        if (n_paths == 0) continue;
        if (!c.next(b, e)) { set_error("p2m file truncated at line %ld", c.line); return DMX_ERR_ARG; }   // summary line
        for (long pi = 0; pi < n_paths; ++pi) {
            if (!c.next(b, e) || read_numbers(b, e, num, 9) != 9) { set_error("p2m line %ld: expected 9 path fields", c.line); return DMX_ERR_ARG; }
            const long n_int = (long)num[1];
            const bool keep = pi < max_paths;
            const int64_t o = rx * max_paths + pi;
            if (keep) {
                power[o] = (float)num[2]; phase[o] = (float)num[3]; delay[o] = (float)num[4];
                aoa_el[o] = (float)num[5]; aoa_az[o] = (float)num[6]; aod_el[o] = (float)num[7]; aod_az[o] = (float)num[8];
            }
            if (!c.next(b, e)) { set_error("p2m file truncated at line %ld", c.line); return DMX_ERR_ARG; }
            if (keep) {
                const double code = interaction_code(b, e);
                if (code < 0) { set_error("p2m line %ld: unknown interaction code", c.line); return DMX_ERR_ARG; }
                inter[o] = (float)code;
            }
            if (!c.next(b, e)) { set_error("p2m file truncated at line %ld", c.line); return DMX_ERR_ARG; }   // Tx position
            for (long ii = 0; ii < n_int; ++ii) {
                if (!c.next(b, e)) { set_error("p2m file truncated at line %ld", c.line); return DMX_ERR_ARG; }
                if (keep && inter_pos && ii < max_inter) {
                    double xyz[3];
                    if (read_numbers(b, e, xyz, 3) != 3) { set_error("p2m line %ld: expected an xyz position", c.line); return DMX_ERR_ARG; }
                    float* dst = inter_pos + ((o * max_inter) + ii) * 3;
                    dst[0] = (float)xyz[0]; dst[1] = (float)xyz[1]; dst[2] = (float)xyz[2];
                }
            }
            if (!c.next(b, e)) { set_error("p2m file truncated at line %ld", c.line); return DMX_ERR_ARG; }   // Rx position
        }
    }
    return DMX_OK;
}

}  // extern "C"
