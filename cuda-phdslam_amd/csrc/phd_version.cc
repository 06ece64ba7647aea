// phd_version(): library name + the build id (hash of the sources, csrc/Makefile).  A translation unit of its own (.cc: outside
// the hash's *.cpp/*.h/*.hip wildcard, so the id does not depend on itself) recompiled whenever the id changes.
#ifndef PHD_BUILD_ID
#define PHD_BUILD_ID "unknown"
#endif
extern "C" const char* phd_version(void) { return "cuda-phdslam_amd 0.3 (gfx950) build " PHD_BUILD_ID; }
