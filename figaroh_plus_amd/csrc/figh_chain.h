// Types shared by the chain regressor kernel (figh_regressor.hip) and the fused regressor + TSQR kernel (figh_fused.hip).
#pragma once

#include "figh_internal.h"

namespace figh {

// tree constants of a fixed-base serial chain of NJ revolute joints: a by-value kernel argument (SGPRs)
template <int NJ>
struct ChainParams {
    double axis[NJ][3];
    double Rp[NJ][9];
    double pp[NJ][3];
    double g[3];
};

template <int NJ, bool TX40>
struct ChainGeom {
    static constexpr int NC = 14 * NJ + (TX40 ? 3 : 0);
    static constexpr int VEC = (NC % 2 == 0) ? 2 : 1;       // doubles per store
    static constexpr int LDT = (NC % 2 == 0) ? NC + 2 : NC;  // LDS row stride (doubles)
    static constexpr int CH = NC / VEC;                       // store chunks per row
};

// host side: the constants of links 1 .. NJ of a chain model
template <int NJ>
inline ChainParams<NJ> chain_params(const figh_model_s *m) {
    ChainParams<NJ> P;
    const DevModel &h = m->host;
    for (int k = 0; k < NJ; ++k) {
        for (int d = 0; d < 3; ++d) P.axis[k][d] = h.axis[k + 1][d];
        for (int d = 0; d < 9; ++d) P.Rp[k][d] = h.placement[k + 1][d];
        for (int d = 0; d < 3; ++d) P.pp[k][d] = h.placement[k + 1][9 + d];
    }
    for (int d = 0; d < 3; ++d) P.g[d] = h.gravity[d];
    return P;
}

}  // namespace figh
