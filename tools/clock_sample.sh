#!/bin/bash
# Samples the GPU clocks / power with rocm-smi while the flagship bench runs (is the MFMA-heavy step power-limited?).
# usage: bash tools/clock_sample.sh   (writes gpurun_out/clocks.log)
mkdir -p gpurun_out
python bench.py --steps 3000 --warmup 5 --no-cpu-baseline > gpurun_out/clock_bench.json 2>/dev/null &
BP=$!
sleep 8
for i in 1 2 3 4 5 6; do
  /opt/rocm/bin/rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|fclk|Power|Temperature \(Sensor (edge|junction|hotspot)" >> gpurun_out/clocks.log
  echo "---" >> gpurun_out/clocks.log
  sleep 1
done
wait $BP
tail -c 400 gpurun_out/clock_bench.json | head -c 400 >> gpurun_out/clocks.log
