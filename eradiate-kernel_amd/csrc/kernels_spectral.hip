// kernels_spectral.hip -- the render kernels compiled for the spectral variant (gpu_spectral = the semantics of scalar_spectral):
// Spectrum<Float, 4> (core/spectrum.h:57-73), four wavelengths per camera sample drawn by sample_wavelength (:305-314), the film
// receives spectrum_to_xyz (:210-217).  Same source as kernels.hip, other spectrum type (dmath.h: MTS_SPEC_N); carries the per-lane
// kernels of `path` and `volpath` and the launcher launch_render_spectral.
#define MTS_SPEC_N 4
#include "kernels.hip"
