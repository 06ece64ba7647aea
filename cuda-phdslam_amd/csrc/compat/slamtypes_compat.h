// slamtypes_compat.h — the C++ types a driver written against the reference's src/slamtypes.h
// uses on the PHD path, re-declared over the C-ABI PODs of include/phdslam.h (same names, same
// field names, same memory layout; nothing else of the reference header is reproduced).
#ifndef SLAMTYPES_COMPAT_H
#define SLAMTYPES_COMPAT_H

#include <cmath>
#include <vector>

#include "phdslam.h"

#define REAL float
#define PHD_TYPE 0
#define CPHD_TYPE 1
#define CV_MOTION 0
#define ACKERMAN_MOTION 1
#define STATIC_MODEL 0

// identical layouts (static_assert'ed in phdfilter_compat.cpp against the C-ABI types)
typedef phd_pose ConstantVelocityState;        // src/slamtypes.h:44-51
typedef phd_ackerman_control AckermanControl;  // :84-87
typedef phd_ackerman_noise AckermanNoise;      // :90-93
typedef phd_measurement RangeBearingMeasurement; // :96-101
typedef phd_gaussian2d Gaussian2D;             // :123-127
typedef phd_slam_config SlamConfig;            // :142-250

typedef std::vector<Gaussian2D> GaussianMixture;
typedef std::vector<RangeBearingMeasurement> measurementSet;

// src/slamtypes.h:275-337 — the members the PHD path reads and writes
class ParticleSLAM {
public:
    int n_particles;
    std::vector<REAL> weights;                 // log-weights
    std::vector<ConstantVelocityState> states;
    std::vector<int> resample_idx;
    explicit ParticleSLAM(unsigned int n = 100) : n_particles((int)n), weights(n), states(n), resample_idx(n) {}
};

class SynthSLAM : public ParticleSLAM {
public:
    std::vector<std::vector<Gaussian2D> > maps_static;
    std::vector<Gaussian2D> max_map_static;
    std::vector<Gaussian2D> exp_map_static;
    std::vector<std::vector<REAL> > cardinalities;
    std::vector<REAL> variances;
    explicit SynthSLAM(unsigned int n) : ParticleSLAM(n), maps_static(n), cardinalities(n), variances(n) {}

    // copy_particles (src/slamtypes.h:313-333): select, reset weights to -log N, remember the parents
    SynthSLAM copy_particles(const std::vector<int>& indices) const
    {
        SynthSLAM out((unsigned int)indices.size());
        for (size_t k = 0; k < indices.size(); ++k) {
            const int i = indices[k];
            out.maps_static[k] = maps_static[i];
            out.cardinalities[k] = cardinalities[i];
            out.weights[k] = (REAL)(-std::log((double)indices.size()));
            out.states[k] = states[i];
            out.variances[k] = variances[i];
        }
        out.resample_idx = indices;
        return out;
    }
};

#endif
