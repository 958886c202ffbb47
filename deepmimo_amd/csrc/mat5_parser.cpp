// Host-side locator of a numeric array inside a MATLAB level-5 MAT-file image (see mat5_loader.hip for the
// device half and the context).  Plain C++ with no HIP dependency so that tests can build it with
// -fsanitize=address,undefined: it parses bytes that come from files.
#include <stdint.h>
#include <string.h>
#include "../../include/deepmimo_amd.h"

namespace dmx { void set_error(const char* fmt, ...); }
using dmx::set_error;

namespace {

// MAT-file data types (MAT-File Format, table 1-1)
enum { miINT8 = 1, miUINT8 = 2, miINT16 = 3, miUINT16 = 4, miINT32 = 5, miUINT32 = 6, miSINGLE = 7, miDOUBLE = 9,
       miINT64 = 12, miUINT64 = 13, miMATRIX = 14, miCOMPRESSED = 15 };

inline uint32_t rd32(const unsigned char* p) { uint32_t v; memcpy(&v, p, 4); return v; }

struct Tag { uint32_t type, nbytes; size_t data_off, next_off; };

// Parse one element tag at `off` (normal or small-element form); false when it does not fit in [0, len).
bool read_tag(const unsigned char* b, size_t len, size_t off, Tag* t) {
    if (off + 8 > len) return false;
    const uint32_t w0 = rd32(b + off);
    if (w0 >> 16) {                                  // small data element: 2-byte type, 2-byte size, 4 data bytes
        t->type = w0 & 0xffffu; t->nbytes = w0 >> 16; t->data_off = off + 4; t->next_off = off + 8;
        return t->nbytes <= 4;
    }
    t->type = w0; t->nbytes = rd32(b + off + 4); t->data_off = off + 8;
    if (t->data_off + (size_t)t->nbytes > len) return false;
    const size_t pad = (t->type == miCOMPRESSED) ? 0 : ((8 - (t->nbytes & 7)) & 7);   // compressed elements are unpadded
    t->next_off = t->data_off + t->nbytes + pad;
    return true;
}

int elem_size(uint32_t type) {
    switch (type) {
        case miINT8: case miUINT8: return 1;
        case miINT16: case miUINT16: return 2;
        case miINT32: case miUINT32: case miSINGLE: return 4;
        case miDOUBLE: case miINT64: case miUINT64: return 8;
        default: return 0;
    }
}


}  // namespace

extern "C" {

int dmx_mat5_find(const void* file_image, size_t len, const char* var_name, dmx_mat_info* info) {
    if (!file_image || !info || len < 128) { set_error("not a MAT-file image (shorter than the 128-byte header)"); return DMX_ERR_ARG; }
    const unsigned char* b = (const unsigned char*)file_image;
    memset(info, 0, sizeof(*info));
    if (!(b[126] == 'I' && b[127] == 'M')) { set_error("MAT-file is not little-endian level 5 (endian tag %c%c)", b[126], b[127]); return DMX_ERR_ARG; }
    size_t off = 128;
    Tag t;
    while (off < len && read_tag(b, len, off, &t)) {
        if (t.type == miCOMPRESSED) {
            // the caller inflates [comp_offset, comp_offset + comp_bytes) with zlib and calls again on the result
            info->compressed = 1; info->comp_offset = (int64_t)t.data_off; info->comp_bytes = (int64_t)t.nbytes;
            set_error("MAT element is zlib-compressed");
            return DMX_ERR_SHAPE;
        }
        if (t.type == miMATRIX) {
            const size_t end = t.data_off + t.nbytes;
            Tag f, d, n, p;
            if (!read_tag(b, end, t.data_off, &f) || f.type != miUINT32 || f.nbytes < 8) break;
            const uint32_t flags = rd32(b + f.data_off);
            if (!read_tag(b, end, f.next_off, &d) || d.type != miINT32) break;
            if (!read_tag(b, end, d.next_off, &n)) break;
            const bool name_ok = !var_name || (strlen(var_name) == n.nbytes && memcmp(b + n.data_off, var_name, n.nbytes) == 0);
            if (name_ok) {
                if (flags & 0x0800u) { set_error("complex MAT arrays are not ray matrices"); return DMX_ERR_SHAPE; }
                const int ndim = (int)(d.nbytes / 4);
                if (ndim < 1 || ndim > 4) { set_error("MAT array with %d dimensions not supported", ndim); return DMX_ERR_SHAPE; }
                if (!read_tag(b, end, n.next_off, &p) || elem_size(p.type) == 0) { set_error("MAT array '%s' has no numeric payload", var_name ? var_name : "?"); return DMX_ERR_SHAPE; }
                info->class_id = (int32_t)(flags & 0xffu);
                info->data_type = (int32_t)p.type;
                info->elem_bytes = elem_size(p.type);
                info->ndim = ndim;
                int64_t count = 1;
                for (int i = 0; i < ndim; ++i) { info->dims[i] = (int32_t)rd32(b + d.data_off + 4 * i); count *= info->dims[i]; }
                info->data_offset = (int64_t)p.data_off;
                info->data_bytes = (int64_t)p.nbytes;
                if (count * info->elem_bytes != info->data_bytes) { set_error("MAT payload size does not match its dimensions"); return DMX_ERR_SHAPE; }
                return DMX_OK;
            }
        }
        off = t.next_off;
    }
    set_error("variable '%s' not found in MAT-file image", var_name ? var_name : "?");
    return DMX_ERR_ARG;
}


}  // extern "C"
