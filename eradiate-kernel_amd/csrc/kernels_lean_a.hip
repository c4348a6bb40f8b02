// kernels_lean_a.hip -- the regrouping kernels of `volpath` and `volpathmis` (rgb / mono) compiled for scenes that keep EVERY promise of
// integrator_dev.h's scene traits: heterogeneous grey media on pair grids, a walked primitive list without spheres, no area emitters, no
// nested blendphase, no rpv, no grid evaluated through volume_eval().  The metric scene (C3) is one: a kernel without a single call
// (122 VGPRs, no spilled VGPR dword, 28 B of scratch against 384).  Same source as kernels.hip, same arithmetic: the promised-away branches
// are not compiled.  mts_render (capi.cpp) selects it from HostScene::traits; MTSAMD_LEAN=0 keeps every scene on the general kernels.
#if !defined(MTSAMD_BLOCKSTATS)
#define MTS_LEAN _lean_a
#define MTS_VARIANT_NS v_rgb_lean_a
#define MTS_LEAN_MIS_768 1     // `volpathmis`: 164 VGPRs here (general kernel: 193) -- room for a third wave per SIMD, 768 threads on the 512 paths
#define MTS_TRAITS MT_UNIT_A      // dscene.h
#include "kernels.hip"
#endif
