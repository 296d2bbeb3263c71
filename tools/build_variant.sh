#!/bin/bash
# build_variant.sh <name> <extra hipcc flags...>: a second build of libgardenia_hip.so with other compile-time
# knobs (VARIANT_FILE=gdn_bfs: the source the knobs are in, default gdn_tc) into gardenia_amd/lib/var_<name>/ (select it with GARDENIA_HIP_LIB=...); measurement scaffolding.
set -e
HERE=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
out=$HERE/gardenia_amd/lib/var_$name
mkdir -p $out/obj
for f in $HERE/gardenia_amd/csrc/*.hip; do
  b=$(basename $f .hip)
  stale=0  # an object older than its source or any header is rebuilt (a stale gdn_bfs.o once made a knob sweep measure old code)
  for h in $f $HERE/gardenia_amd/csrc/*.hpp $HERE/include/*.h; do [ $h -nt $out/obj/$b.o ] && stale=1; done
  if [ "$b" = "${VARIANT_FILE:-gdn_tc}" ] || [ ! -f $out/obj/$b.o ] || [ $stale = 1 ]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$HERE/include "$@" -c $f -o $out/obj/$b.o &
  fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $out/obj/*.o -o $out/libgardenia_hip.so
echo built $out/libgardenia_hip.so
