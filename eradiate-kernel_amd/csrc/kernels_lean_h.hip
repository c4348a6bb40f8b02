// kernels_lean_h.hip -- as kernels_lean_b.hip for scenes whose media are all HOMOGENEOUS (rgb / mono): no grid behind volume_eval(), a walked
// primitive list without spheres, no area emitters, no nested blendphase; rpv allowed.  A homogeneous atmosphere over a surface (C2) is one:
// the tracking step keeps neither the slab test nor a grid lookup.
#if !defined(MTSAMD_BLOCKSTATS)
#define MTS_LEAN _lean_h
#define MTS_VARIANT_NS v_rgb_lean_h
#define MTS_TRAITS MT_UNIT_H      // dscene.h
#include "kernels.hip"
#endif
