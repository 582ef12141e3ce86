/* Exports the portable log / cos / pow of path-tracing_amd/csrc/pt_portable_math.h (the functions the HIP kernels
 * compile in their exact-math mode and the oracle in its ORC_SHARED_MATH build) for tests/golden/wgsl_interp.py.
 * Build: gcc -O2 -std=gnu11 -fPIC -ffp-contract=off -fno-fast-math -shared -o pm_shim.so pm_shim.c */
#include <stdint.h>
#include "../../path-tracing_amd/csrc/pt_portable_math.h"

float pm_shim_logf(float x) { return pm_logf(x); }
float pm_shim_cosf(float x) { return pm_cosf(x); }
float pm_shim_powf(float x, float y) { return pm_powf(x, y); }
