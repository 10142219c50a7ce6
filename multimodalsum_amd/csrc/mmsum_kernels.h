// Internal: every translation unit sees the public C ABI (flags, error codes, prototypes).
#pragma once
#include "../../include/mmsum_hip.h"
