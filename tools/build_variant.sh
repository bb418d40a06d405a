#!/bin/bash
# One-translation-unit variants of libpbc_hip.so for same-box A/B runs and fault reproducers: recompiles ONE unit with
# extra flags and links it with the default objects of the last `make -C pbc_amd` (in /tmp/pbc_hip_build).
#   tools/build_variant.sh <unit: pbc_hip_a|pbc_hip_d|pbc_hip_f|pbc_hip_group|...> <out name> <extra flags ...>
# -> pbc_amd/variants/lib<out>.so  (git-ignored; travels with gpurun snapshots; use with PBC_HIP_LIB=<path>)
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
U=$1; OUT=$2; shift 2
B=/tmp/pbc_hip_build; D=$B/obj_libpbc_hip; V=$B/var_$OUT
[ -f $D/$U.o ] || { echo "no default objects: run make -C pbc_amd first"; exit 1; }
mkdir -p $V $ROOT/pbc_amd/variants
( cd $V && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -Wno-bitwise-instead-of-logical "$@" -save-temps=obj -c $ROOT/pbc_amd/csrc/$U.hip -o $V/$U.o )
OBJS=""; for o in $D/*.o; do b=$(basename $o .o); case $b in *-hip-amdgcn*|*-host-*) continue;; esac; if [ $b = $U ]; then OBJS="$OBJS $V/$U.o"; else OBJS="$OBJS $o"; fi; done
hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o $ROOT/pbc_amd/variants/lib$OUT.so
echo "built pbc_amd/variants/lib$OUT.so"
