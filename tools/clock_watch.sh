#!/bin/bash
# tools/clock_watch.sh -- shader clock / power of the card while the ranked argmax loop runs (burst vs sustained state):
# samples `rocm-smi` every 0.1 s beside tools/time_variant.py with CTD_WARM_CALLS=3000 (~1.4 s of continuous work)
mkdir -p gpurun_out/clk
(CTD_WARM_CALLS=3000 python tools/time_variant.py "" > gpurun_out/clk/run.log 2>&1) &
pid=$!
for i in $(seq 1 200); do
  kill -0 $pid 2>/dev/null || break
  echo "t=$i $(rocm-smi --showclocks --showpower 2>/dev/null | grep -E 'sclk|Power' | tr '\n' ' ')"
  sleep 0.1
done > gpurun_out/clk/samples.txt
wait $pid
grep ranked gpurun_out/clk/run.log | cut -c1-120
cut -c1-160 gpurun_out/clk/samples.txt | head -60
