#!/bin/bash
# build a variant of the library: tools/build_variant.sh <name> [-DMACRO=..]...  -> geomconsistentfr_amd/lib/<name>.so
# (-DGCFR_FAST_BUILD compiles the default march shape only -- 16 x 4 tiles, groups of four -- for A/B work)
REPO=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
exec python3 "$REPO/geomconsistentfr_amd/build.py" --variant "$NAME" "$@"
