// Identity of the build: the first 16 hex digits of the SHA-256 over every source file of the library (Makefile: HASH_SRCS).
#include "../../include/mmsum_hip.h"

static const char kBuildId[] =
#include "build_id.inc"
    ;

extern "C" const char* mmsum_build_id(void) { return kBuildId; }
