// AddressSanitizer / UBSan harness for the host-side text parser (csrc/p2m_parser.cpp), CPU only.
// Usage: harness <file>   -> parses the file image, then every prefix cut at 97 evenly spread offsets (truncated
// inputs must fail cleanly, never read out of bounds).  The image is copied into an exact-size heap block so
// that ASan sees any overrun.
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../../include/deepmimo_amd.h"

namespace dmx {
void set_error(const char* fmt, ...) { (void)fmt; }
}

static int parse(const char* img, size_t len, bool expect_ok) {
    char* exact = (char*)malloc(len ? len : 1);
    memcpy(exact, img, len);
    const int64_t n = dmx_p2m_count_rx(exact, len);
    int rc = -1;
    if (n >= 0 && n < 100000) {
        const int P = 25, I = 10;
        std::vector<float> m(8 * (size_t)n * P), pos((size_t)n * P * I * 3);
        float* b = m.data();
        const size_t s = (size_t)n * P;
        rc = dmx_p2m_parse_paths(exact, len, P, I, n, b, b + s, b + 2 * s, b + 3 * s, b + 4 * s, b + 5 * s, b + 6 * s, b + 7 * s, pos.data());
    }
    free(exact);
    if (expect_ok && rc != 0) { fprintf(stderr, "full file failed to parse\n"); return 1; }
    return 0;
}

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 2;
    std::vector<char> img;
    char buf[65536];
    size_t r;
    while ((r = fread(buf, 1, sizeof(buf), f)) > 0) img.insert(img.end(), buf, buf + r);
    fclose(f);
    if (parse(img.data(), img.size(), true)) return 1;
    for (int i = 0; i < 97; ++i) parse(img.data(), img.size() * i / 97, false);
    // corrupted copies: digits replaced by letters / dashes at a stride
    for (int stride = 7; stride < 60; stride += 13) {
        std::vector<char> c = img;
        for (size_t k = stride; k < c.size(); k += stride * 11) c[k] = (k & 1) ? 'x' : '-';
        parse(c.data(), c.size(), false);
    }
    printf("p2m asan harness ok\n");
    return 0;
}
