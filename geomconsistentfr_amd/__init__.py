"""geomconsistentfr_amd -- MI355X-native render block of GeomConsistentFR.

Scope: the differentiable ray-marched soft shadow + Lambertian shading +
compositing that the reference inlines at train_raytracing_relighting_CelebAHQ_DSSIM_8x.py:352-524,
as hand-written HIP kernels for gfx950 behind a C ABI (include/gcfr.h), plus the host-side mirror of
the reference's only callable boundary, RelightNet.forward.
"""
from .block import RenderParams, render, shadow_min_distance, light_prep  # noqa: F401

__version__ = "0.5.0"   # = the library's (gcfr_version(): "gcfr-hip 0.5.0 gfx950")
