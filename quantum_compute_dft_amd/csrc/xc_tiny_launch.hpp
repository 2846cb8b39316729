// Host interface of the one-pass small-basis sweep kernel (xc_tiny_kernels.hpp), compiled in its own translation
// unit (xc_tiny.hip): that unit is built with -disable-machine-licm (build.py), see the note in xc_tiny.hip.
#pragma once
#include <hip/hip_runtime.h>

namespace qcdft {

constexpr int TINY_MAX_NAO = 32;

// type everywhere: 0 LDA, 1 GGA, 2 B3LYP
// option tiny = -1 (auto): the sizes at which the one-pass kernel beats rho -> xc_points -> vxc -> reduce
bool tiny_pays(int num_cu, int type, int nao, long ngrid);

// workgroups (= Vxc slabs = Exc partials) of a launch over `ngrid` points
int tiny_workgroups(int num_cu, int type, int nao, long ngrid);

// type 0 LDA, 1 GGA, 2 B3LYP (slabs come out as M + M^T); `slabs` holds nwg * nao * nao doubles, `partial` nwg
void launch_sweep_tiny(hipStream_t st, int nwg, int type, long ngrid, int nao, const double *ao, const double *gx,
                       const double *gy, const double *gz, const double *dm, const double *w, double *slabs,
                       double *partial, int quirks);

} // namespace qcdft
