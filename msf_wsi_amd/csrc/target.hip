// Library identification for the C ABI (include/msfwsi_hip.h).
#include "common.h"
#include "../../include/msfwsi_hip.h"

// MSFWSI_BUILD_ID: 16 hex digits = sha256 over every source the library is built from (csrc/*.hip, csrc/*.h, the public
// header, the Makefile, $(EXTRA)), passed in by the Makefile; the marker string makes the id readable from the FILE without
// loading it (msf_wsi_amd/_lib.py: built_id / source_id decide staleness by this, not by mtimes).
#ifndef MSFWSI_BUILD_ID
#error "MSFWSI_BUILD_ID must be defined by the Makefile"
#endif
static const char kBuildId[] = "MSFWSI_BUILD_ID=" MSFWSI_BUILD_ID;

extern "C" const char* msfwsi_target(void) { return "gfx950"; }
extern "C" const char* msfwsi_build_id(void) { return kBuildId + 16; }
