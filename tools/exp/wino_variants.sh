#!/bin/bash
# Build timing variants of libptmi355.so into tools/exp/_bin/:   tools/exp/wino_variants.sh <name>=<source.hip>[:mask] ...
# (mask = -DWINO_EXP bits, results are WRONG by design when non-zero).  Then on the GPU box:
#   python tools/exp/wino_bench.py tools/exp/_bin/libptmi355_<name>.so ...
set -e
cd "$(dirname "$0")/../.."
python -m probabilisticteacher_amd.build_ext >/dev/null
OBJS=$(ls probabilisticteacher_amd/_build/*.o | grep -v wino.o)
mkdir -p tools/exp/_bin
for spec in "$@"; do
  name=${spec%%=*}; rest=${spec#*=}; src=${rest%%:*}; mask=0
  [[ "$rest" == *:* ]] && mask=${rest##*:}
  cp "$src" probabilisticteacher_amd/csrc/_wino_variant.hip
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -Wno-unused-result ${WINO_FLAGS} -DWINO_EXP=$mask \
    -c probabilisticteacher_amd/csrc/_wino_variant.hip -o tools/exp/_bin/wino_$name.o
  rm -f probabilisticteacher_amd/csrc/_wino_variant.hip
  hipcc --offload-arch=gfx950 -shared -fPIC -o tools/exp/_bin/libptmi355_$name.so $OBJS tools/exp/_bin/wino_$name.o
done
