#!/bin/bash
# tools/clock_probe_cmd.sh SECONDS_BEFORE_SAMPLING CMD... -- on the GPU box: rocm-smi clock / power samples while CMD runs
D=$1; shift
"$@" > gpurun_out/clock_probe_cmd.out 2>&1 &
BP=$!
sleep $D
for i in 1 2 3 4 5; do
    rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | sed "s/GPU\[0\]\t\t: //g; s/clock level: [01S]: //g; s/Current Socket Graphics Package //" | tr '\n' ';'; echo
    sleep 1
done
wait $BP
tail -2 gpurun_out/clock_probe_cmd.out
