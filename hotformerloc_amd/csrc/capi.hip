// Library identification entry points of libhotformerloc_hip.so.
#include "hfl_common.h"

extern "C" {

int hfl_version(void) { return 100; }   // 1.00

const char* hfl_arch(void) { return "gfx950"; }

}  // extern "C"
