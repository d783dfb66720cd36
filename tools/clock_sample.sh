#!/bin/bash
# GPU clock / power beside sustained launches of the headline mode's dominant kernel (tools/x3_probe.py PROBE_SUSTAIN): is the K loop at
# the MFMA rate of the clock the chip holds under this load?  usage: bash tools/clock_sample.sh <out.log>
out=${1:-clock_sample.log}
: > $out
( PROBE_ROWS=4000 PROBE_SUSTAIN=10 PROBE_SUSTAIN_SHAPES=w13,wo python tools/x3_probe.py > $out.probe 2>&1 ) &
pid=$!
for i in $(seq 1 80); do
  echo "== sample $i t=$(date +%s.%N) probe_lines=$(wc -l < $out.probe)" >> $out
  rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|power (W)" >> $out
  kill -0 $pid 2>/dev/null || break
  sleep 0.5
done
wait $pid
echo "== probe output" >> $out
cat $out.probe >> $out; rm -f $out.probe
