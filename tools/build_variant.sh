#!/bin/bash
# build a variant of the library: tools/build_variant.sh <name> [-DMACRO=..]...  -> geomconsistentfr_amd/lib/<name>.so
# (-DGCFR_FAST_BUILD instantiates the default tile shape / group only: ~4x faster to compile, for A/B work)
REPO=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
C=$REPO/geomconsistentfr_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -fno-fast-math -munsafe-fp-atomics -Wall "$@" \
  $C/gcfr_shadow.hip $C/gcfr_shade.hip $C/gcfr_backward.hip $C/gcfr_normals.hip $C/gcfr_postprocess.hip $C/gcfr_dataset.hip -o $REPO/geomconsistentfr_amd/lib/$NAME.so
