// kernels_lean_b.hip -- as kernels_lean_a.hip for scenes that keep all promises but two: an rpv BSDF and grids evaluated through
// volume_eval() (the weight volume of a blendphase) are allowed.  The layered atmosphere (C4 / C5) is one.
#if !defined(MTSAMD_BLOCKSTATS)
#define MTS_LEAN _lean_b
#define MTS_VARIANT_NS v_rgb_lean_b
#define MTS_TRAITS MT_UNIT_B      // dscene.h
#include "kernels.hip"
#endif
