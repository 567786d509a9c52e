// Smooth particle-mesh Ewald reciprocal space (gfx950) — filled in with the PME stage.
#include "remd_internal.h"

int remd_pme_setup(remd_ctx* h) { return remd_fail(h, -4, "PME not built into this libremd_hip.so yet"); }
int remd_pme_destroy(remd_ctx* h) { (void)h; return 0; }
int remd_pme_forces(remd_ctx* h, bool with_energy, double* d_energy) { (void)with_energy; (void)d_energy; return remd_fail(h, -4, "PME not built"); }
