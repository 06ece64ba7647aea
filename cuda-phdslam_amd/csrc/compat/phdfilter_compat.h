// phdfilter_compat.h — the reference's filter interface (src/phdfilter.h:10-34) and RNG C ABI
// (src/rng.h:16-25) over the gfx950 library.  A driver written against the reference links
// against this header + libphdfilter_compat.so instead of phdfilter.cu / rng.cpp.
#ifndef PHDFILTER_COMPAT_H
#define PHDFILTER_COMPAT_H

#include <stdint.h>

#include "slamtypes_compat.h"

// globals shared across the seam (src/main.cpp:57,60; consumed at src/phdfilter.cu:108,111)
extern SlamConfig config;

void initRandomNumberGenerators();                                   // src/phdfilter.h:10
void setDeviceConfig(const SlamConfig& config);                      // :33-34
void phdPredict(SynthSLAM& particles, ...);                          // :16-17 (optional AckermanControl)
SynthSLAM phdUpdateSynth(SynthSLAM& particles, measurementSet measurements); // :23-24
void recoverSlamState(SynthSLAM& particles, ConstantVelocityState& expectedPose,
                      std::vector<REAL>& cn_estimate);               // :26-28 (defined in main.cpp:318)
// src/main.cpp:453-501 (a template there); n_new_particles < 0: keep the count
SynthSLAM resampleParticles(SynthSLAM oldParticles, int n_new_particles = -1);
// src/main.cpp:1281-1284
REAL computeNeff(SynthSLAM& particles);

extern "C" {
double randn();                                                      // src/rng.h:16-17
double randu01();                                                    // :20-21
void phd_compat_seed_rng(uint64_t seed);                             // replaces the wall-clock seed (rng.cpp:10)
}

#endif
