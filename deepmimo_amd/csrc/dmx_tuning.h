// Measurement hooks of the launch paths.  The shipped library is built WITHOUT them: behaviour is a function of the
// arguments of the C-ABI call alone (include/deepmimo_amd.h: no global state besides the thread-local error string), so
// `tuning_int` folds to its default and no launcher reads the process environment.  `make alt ALTFLAGS=-DDMX_TUNING_HOOKS`
// (tools/ab_two_libs.sh, tools/fold_sweep*.sh, tools/lpf_bench.py) builds the A/B library in which the named
// environment variables select the alternatives that DESIGN.md section 6 lists as measured.
#pragma once
#include <stdlib.h>

namespace dmx {

#ifdef DMX_TUNING_HOOKS
inline int tuning_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return v ? atoi(v) : dflt;
}
inline bool tuning_set(const char* name) { return getenv(name) != nullptr; }
#else
constexpr int tuning_int(const char*, int dflt) { return dflt; }
constexpr bool tuning_set(const char*) { return false; }
#endif

}  // namespace dmx
