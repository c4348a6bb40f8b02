// kernels_lean_s.hip -- as kernels_lean_b.hip for the spectral variant (gpu_spectral): the 256-path regrouping kernels of `volpath` and
// `volpathmis` for scenes whose media are all heterogeneous with two gridvolume_spectral grids on one geometry and spectral interval, with a
// walked primitive list without spheres, no area emitters and no nested blendphase (rpv and blend-weight grids allowed): the layered
// atmosphere as Eradiate renders it (C5S, C5SB under nbins, C5SM under volpathmis).
#if !defined(MTSAMD_BLOCKSTATS)
#define MTS_SPEC_N 4
#define MTS_LEAN _lean_s
#define MTS_VARIANT_NS v_spectral_lean
#define MTS_TRAITS MT_UNIT_B      // dscene.h
#include "kernels.hip"
#endif
