// rpe_build_id(): the first 16 hex digits of the SHA-256 of the library's sources (Makefile: SRC_ID), fixed at build time.
#include "../../include/rpe_hip.h"

#ifndef RPE_SOURCE_ID
#error "build through the Makefile (it defines RPE_SOURCE_ID)"
#endif

extern "C" const char* rpe_build_id(void) { return RPE_SOURCE_ID; }
