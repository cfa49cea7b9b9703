// library identification
#include "commu_hip.h"
extern "C" const char* commu_hip_version(void) { return "commu_hip 0.1 (gfx950)"; }
