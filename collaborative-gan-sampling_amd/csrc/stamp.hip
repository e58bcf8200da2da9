// cgs_source_sha(): the sha256 of the kernel sources this library was BUILT from, embedded by the Makefile
// (-DCGS_SOURCE_SHA="..." over the same file list and byte order as cgs_amd/lib.py::source_hash).  The host side
// compares it with the hash of the sources it finds at run time: a prebuilt .so that no longer belongs to the
// tree it ships with is refused instead of being measured under the wrong name (include/cgs_hip.h).
#include "cgs_hip.h"

extern "C" const char* cgs_source_sha(void) { return CGS_SOURCE_SHA; }
