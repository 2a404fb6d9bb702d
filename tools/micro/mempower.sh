#!/bin/bash
# tools/micro/mempower.sh -- on the GPU box: each read pattern of mempower for 7 s with rocm-smi clock / power samples beside it
cd "$(dirname "$0")"
for p in 0 1 2 3 4; do
    ./mempower $p 7 > /tmp/mp_$p.txt &
    BP=$!
    sleep 3
    for i in 1 2 3; do
        rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | sed "s/GPU\[0\]\t\t: //g; s/clock level: [01S]: //g; s/Current Socket Graphics Package //; s/=* Power Consumption =*//" | tr '\n' ' '; echo
        sleep 1
    done
    wait $BP
    cat /tmp/mp_$p.txt
done
