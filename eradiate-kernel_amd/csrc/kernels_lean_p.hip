// kernels_lean_p.hip -- `path` as the flat loop with regeneration (kernels.hip: path_pixel_flat; rgb / mono) for scenes with a walked primitive
// list, no spheres and no rpv BSDF -- the cornell box (C1, C1L) is one; area emitters stay.  Three of the four out-of-line functions of
// the general kernel are not compiled: 35 spilled VGPR dwords instead of 98 at the same four waves per SIMD.
#if !defined(MTSAMD_BLOCKSTATS)
#define MTS_LEAN _lean_p
#define MTS_LEAN_PATH 1
#define MTS_VARIANT_NS v_rgb_lean_p
#define MTS_TRAITS MT_UNIT_P      // dscene.h
#include "kernels.hip"
#endif
