#!/bin/bash
# tools/clock_probe.sh WORKLOAD -- on the GPU box: shader / memory clock and power sampled by rocm-smi while bench.py loops the workload
W=${1:-frames}
python bench.py --workload $W --no-cpu-baseline --no-config4 --steps ${2:-20000} --warmup 20 > gpurun_out/clock_probe_$W.json 2> gpurun_out/clock_probe_$W.err &
BP=$!
sleep 6
for i in 1 2 3 4 5; do
    rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|Power|power" | tr '\n' ';'; echo
    sleep 1
done
wait $BP
cut -c1-200 gpurun_out/clock_probe_$W.json
echo "idle:"; sleep 2; rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power|power" | tr '\n' ';'; echo
