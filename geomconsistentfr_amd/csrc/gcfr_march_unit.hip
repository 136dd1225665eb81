// One translation unit of the march: the kernels of ONE (tile width, samples per group) shape and the function that
// launches them.  build.py compiles this file once per shape (-DGCFR_UNIT_TILE_W=.. -DGCFR_UNIT_GROUP=..), in parallel;
// gcfr_shadow.hip selects the unit at run time (march_unit()).  See gcfr_march.hpp.
#include "gcfr_march.hpp"

#if !defined(GCFR_UNIT_TILE_W) || !defined(GCFR_UNIT_GROUP)
#error "compile with -DGCFR_UNIT_TILE_W=<8|16|32|64> -DGCFR_UNIT_GROUP=<1|2|4> (geomconsistentfr_amd/build.py does)"
#endif

namespace gcfr {

void GCFR_MARCH_UNIT_NAME(GCFR_UNIT_TILE_W, GCFR_UNIT_GROUP)(const ShadowQuadArgs &a, bool even_half, bool want_argmin,
                                                             Schedule sch, dim3 grid, hipStream_t st, unsigned lds_bytes)
{
    launch_quad3<GCFR_UNIT_TILE_W, GCFR_UNIT_GROUP>(a, even_half, want_argmin, sch, grid, st, lds_bytes);
}

}  // namespace gcfr
