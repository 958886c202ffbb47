// AddressSanitizer / UBSan harness for the MAT-v5 locator (csrc/mat5_parser.cpp), CPU only.
// Usage: harness <file.mat> <var>  -> locates the variable, then retries on every prefix (97 cuts) and on
// byte-corrupted copies (header fields overwritten): must fail cleanly or report offsets inside the image.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../../include/deepmimo_amd.h"

namespace dmx {
void set_error(const char* fmt, ...) { (void)fmt; }
}

static int probe(const char* img, size_t len, const char* var, bool expect_ok) {
    char* exact = (char*)malloc(len ? len : 1);
    memcpy(exact, img, len);
    dmx_mat_info info;
    const int rc = dmx_mat5_find(exact, len, var, &info);
    int bad = 0;
    if (rc == 0) {
        if (info.data_offset < 0 || info.data_bytes < 0 || (size_t)(info.data_offset + info.data_bytes) > len) bad = 1;
        else { volatile char sink = 0; for (int64_t i = 0; i < info.data_bytes; i += 61) sink ^= exact[info.data_offset + i]; (void)sink; }
    }
    free(exact);
    if (bad) { fprintf(stderr, "payload range outside the image\n"); return 1; }
    if (expect_ok && rc != 0) { fprintf(stderr, "intact file failed\n"); return 1; }
    return 0;
}

int main(int argc, char** argv) {
    if (argc < 3) return 2;
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 2;
    std::vector<char> img;
    char buf[65536];
    size_t r;
    while ((r = fread(buf, 1, sizeof(buf), f)) > 0) img.insert(img.end(), buf, buf + r);
    fclose(f);
    if (probe(img.data(), img.size(), argv[2], true)) return 1;
    for (int i = 0; i < 97; ++i) if (probe(img.data(), img.size() * i / 97, argv[2], false)) return 1;
    for (size_t pos = 128; pos < img.size() && pos < 400; pos += 3) {     // clobber tag / size / dims bytes
        for (int v : {0x00, 0x7f, 0xff}) {
            std::vector<char> c = img;
            c[pos] = (char)v;
            if (probe(c.data(), c.size(), argv[2], false)) return 1;
        }
    }
    printf("mat5 asan harness ok\n");
    return 0;
}
