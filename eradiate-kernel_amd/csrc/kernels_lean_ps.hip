// kernels_lean_ps.hip -- kernels_lean_p.hip for the spectral variant (gpu_spectral): `path` as the flat loop, four wavelengths per sample.
#if !defined(MTSAMD_BLOCKSTATS)
#define MTS_SPEC_N 4
#define MTS_LEAN _lean_ps
#define MTS_LEAN_PATH 1
#define MTS_VARIANT_NS v_spectral_lean_p
#define MTS_TRAITS MT_UNIT_P      // dscene.h
#include "kernels.hip"
#endif
