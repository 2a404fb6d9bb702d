#!/bin/bash
# round 3, first GPU call: sensors available to an ordinary user, cold-start clock trace, pk A/B, full GPU tier
set -x
mkdir -p gpurun_out/r3
ls /sys/class/drm/ > gpurun_out/r3/sysfs.txt 2>&1
for d in /sys/class/drm/card*/device; do echo "== $d -> $(readlink -f $d)"; ls $d | tr '\n' ' '; echo; ls $d/hwmon/* 2>/dev/null | tr '\n' ' '; echo; done >> gpurun_out/r3/sysfs.txt 2>&1
python tools/gpu_sensors.py > gpurun_out/r3/sensors.json 2> gpurun_out/r3/sensors.err
python -c "import amdsmi; print('amdsmi ok')" >> gpurun_out/r3/sensors.err 2>&1
python tools/clock_trace.py --seconds 4 > gpurun_out/r3/clock_trace.jsonl 2> gpurun_out/r3/clock_trace.err &&
python tools/ab.py --rounds 12 variants/base.so variants/pk.so variants/base.so variants/pk.so > gpurun_out/r3/ab_pk.txt 2>&1 &&
python -m pytest tests -m gpu -x -q > gpurun_out/r3/tests1.log 2>&1
echo "rc=$?"
tail -5 gpurun_out/r3/ab_pk.txt; tail -3 gpurun_out/r3/tests1.log; cat gpurun_out/r3/sensors.json
