#!/bin/bash
# Build timing-experiment variants of libptmi355.so (wino.hip compiled with -DWINO_EXP=<mask>, results are WRONG by design)
# into tools/exp/_bin/ and time conv layers with each:  tools/exp/wino_variants.sh 0 1 2 ...   then on the GPU box
#   python tools/exp/wino_bench.py tools/exp/_bin/libptmi355_exp<mask>.so
set -e
cd "$(dirname "$0")/../.."
python -m probabilisticteacher_amd.build_ext >/dev/null
OBJS=$(ls probabilisticteacher_amd/_build/*.o | grep -v wino.o)
for m in "$@"; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -Wno-unused-result ${WINO_FLAGS} -DWINO_EXP=$m \
    -c probabilisticteacher_amd/csrc/wino.hip -o tools/exp/_bin/wino_exp$m.o
  hipcc --offload-arch=gfx950 -shared -fPIC -o tools/exp/_bin/libptmi355_exp$m.so $OBJS tools/exp/_bin/wino_exp$m.o
done
