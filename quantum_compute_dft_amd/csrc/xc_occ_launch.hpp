// Host interface of the occupied-orbital density kernels (xc_occ_kernels.hpp), compiled in their own
// translation unit (xc_occ.hip) so that the two halves of libdft.so build in parallel.
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>

namespace qcdft {

struct OccPlan {
    int nto = 0;       // 16-orbital tiles per pass (1..8)
    int npass = 0;     // passes over the planes (more than 128 occupied orbitals)
    int nch = 0;       // 32-column chunks of the AO index
    int nw = 4;        // waves per workgroup
    bool resident = false; // all of C in LDS, persistent workgroups
    size_t lds_bytes = 0;
    size_t cp_doubles = 0; // packed, zero-padded copy of C the kernel reads
    double mfma_occ = 0.0, mfma_full = 0.0; // fp64 MFMA instructions per 16 grid rows: this path / X = AO . Ds
};

OccPlan occ_plan(int nao, int nocc, bool gga);

// packs C into cp_ws (plan.cp_doubles doubles) and launches the density kernel on `st`
hipError_t launch_rho_occ(hipStream_t st, int num_cu, const OccPlan &plan, bool gga, bool vec16, long ngrid, int nao,
                          int nocc, const double *cocc, double *cp_ws, const double *ao, const double *gx,
                          const double *gy, const double *gz, double *rho, double *grad, double *sigma);

// dm = C C^T
void launch_dm_from_cocc(hipStream_t st, int nao, int nocc, const double *cocc, double *dm);

} // namespace qcdft
