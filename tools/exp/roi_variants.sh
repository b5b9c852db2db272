#!/bin/bash
# Build variants of libptmi355.so that differ in roi_align.hip's compile-time switches:
#   tools/exp/roi_variants.sh <name>="<-D flags>" ...   ->  tools/exp/_bin/libptmi355_roi_<name>.so
# then on the GPU box:  python tools/exp/roi_bench.py --lib tools/exp/_bin/libptmi355_roi_<name>.so
set -e
cd "$(dirname "$0")/../.."
python -m probabilisticteacher_amd.build_ext >/dev/null
OBJS=$(ls probabilisticteacher_amd/_build/*.o | grep -v "/roi_align.o")
mkdir -p tools/exp/_bin
for spec in "$@"; do
  name=${spec%%=*}; flags=${spec#*=}
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -Wno-unused-result -Iinclude $flags \
    -c probabilisticteacher_amd/csrc/roi_align.hip -o tools/exp/_bin/roi_$name.o
  hipcc --offload-arch=gfx950 -shared -fPIC -o tools/exp/_bin/libptmi355_roi_$name.so $OBJS tools/exp/_bin/roi_$name.o
done
