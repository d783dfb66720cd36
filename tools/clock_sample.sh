#!/bin/bash
# GPU clock / power beside sustained launches of the headline mode's dominant kernel (tools/x3_probe.py PROBE_SUSTAIN): is the K loop at
# the MFMA rate of the clock the chip holds under this load?  usage: bash tools/clock_sample.sh <out.log>
out=${1:-clock_sample.log}
( PROBE_ROWS=4000 PROBE_SUSTAIN=6 PROBE_SUSTAIN_SHAPES=w13,wo python tools/x3_probe.py > $out.probe 2>&1 ) &
pid=$!
sleep 25   # model-free probe: import + the first shapes
for i in $(seq 1 24); do
  echo "== sample $i $(date +%s.%N)" >> $out
  rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|power\|mclk\|fclk" >> $out
  kill -0 $pid 2>/dev/null || break
  sleep 1
done
wait $pid
cat $out.probe >> $out; rm -f $out.probe
