// Library identification for the C ABI (include/msfwsi_hip.h).
#include "common.h"
#include "../../include/msfwsi_hip.h"

extern "C" const char* msfwsi_target(void) { return "gfx950"; }
