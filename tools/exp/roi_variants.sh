#!/bin/bash
# Build variants of libptmi355.so from COPIES of roi_align.hip whose tuning constants are rewritten (the product file carries no
# experiment switches):   tools/exp/roi_variants.sh <name>="RF2_CG8=4 RB3_WAVES=6 ..." ...   ->  tools/exp/_bin/libptmi355_roi_<name>.so
# then on the GPU box:  python tools/exp/roi_bench.py --lib tools/exp/_bin/libptmi355_roi_<name>.so
set -e
cd "$(dirname "$0")/../.."
python -m probabilisticteacher_amd.build_ext >/dev/null
OBJS=$(ls probabilisticteacher_amd/_build/*.o | grep -v "/roi_align.o")
mkdir -p tools/exp/_bin
for spec in "$@"; do
  name=${spec%%=*}; consts=${spec#*=}
  cp probabilisticteacher_amd/csrc/roi_align.hip probabilisticteacher_amd/csrc/_roi_variant.hip
  for kv in $consts; do
    sed -i -E "s/^(constexpr int ${kv%%=*} = )[0-9]+;/\1${kv#*=};/" probabilisticteacher_amd/csrc/_roi_variant.hip
  done
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -Wno-unused-result -Iinclude \
    -c probabilisticteacher_amd/csrc/_roi_variant.hip -o tools/exp/_bin/roi_$name.o
  rm -f probabilisticteacher_amd/csrc/_roi_variant.hip
  hipcc --offload-arch=gfx950 -shared -fPIC -o tools/exp/_bin/libptmi355_roi_$name.so $OBJS tools/exp/_bin/roi_$name.o
done
