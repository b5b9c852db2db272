#!/bin/bash
# Experiment: sample the shader clock and socket power while the conv-dominated train step runs.
# Usage (GPU box): bash tools/exp/clock_under_load.sh > gpurun_out/clock_under_load.txt
python bench.py --steps 12 --warmup 2 > /tmp/bench_clock.log 2>&1 &
BP=$!
sleep 45
for i in $(seq 1 12); do
  date +%s.%N
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|Power|power" | head -8
  sleep 1.5
done
wait $BP
tail -1 /tmp/bench_clock.log | cut -c1-300
echo "--- idle"
sleep 3
rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power|power" | head -6
