#!/bin/bash
# Build timing variants of libptmi355.so into tools/exp/_bin/:   tools/exp/p8_variants.sh <name>=<source.hip>[:mask] ...
# (mask = -DP8X bits of make_p8_variant.py: results are WRONG by design when non-zero).  Then on the GPU box:
#   python tools/kbench_p8.py --lib tools/exp/_bin/libptmi355_<name>.so ...
set -e
cd "$(dirname "$0")/../.."
python -m probabilisticteacher_amd.build_ext >/dev/null
OBJS=$(ls probabilisticteacher_amd/_build/*.o | grep -v "/p8.o")
mkdir -p tools/exp/_bin
for spec in "$@"; do
  name=${spec%%=*}; rest=${spec#*=}; src=${rest%%:*}; mask=0
  [[ "$rest" == *:* ]] && mask=${rest##*:}
  cp "$src" probabilisticteacher_amd/csrc/_p8_variant.hip
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -Wno-unused-result ${P8_FLAGS} -DP8X=$mask \
    -c probabilisticteacher_amd/csrc/_p8_variant.hip -o tools/exp/_bin/p8_$name.o
  rm -f probabilisticteacher_amd/csrc/_p8_variant.hip
  hipcc --offload-arch=gfx950 -shared -fPIC -o tools/exp/_bin/libptmi355_$name.so $OBJS tools/exp/_bin/p8_$name.o
done
