// kernels_lean_c.hip -- as kernels_lean_b.hip for scenes WITH a BVH (more than 40 primitives: mesh canopies, many rectangles) under
// heterogeneous grey media on pair grids: no spheres, no area emitters, no nested blendphase; rpv, blend-weight grids and the BVH allowed.
// The traversal stays the one real function of the kernel (bvh_intersect: its loop keeps its registers out of the blocks that call it).
#if !defined(MTSAMD_BLOCKSTATS)
#define MTS_LEAN _lean_c
#define MTS_VARIANT_NS v_rgb_lean_c
#define MTS_TRAITS MT_UNIT_C      // dscene.h
#include "kernels.hip"
#endif
