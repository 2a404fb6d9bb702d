#!/bin/bash
# tools/micro/mempower2.sh -- cache-policy sweep of the MFCC kernel's memory walk on noise input; sysfs power / clock beside each pattern
cd "$(dirname "$0")"
sens() {
  for d in /sys/class/drm/card[0-9]*/device; do
    [ -e $d/pp_dpm_sclk ] || continue
    pw=$(cat $d/hwmon/hwmon*/power1_input 2>/dev/null || cat $d/hwmon/hwmon*/power1_average 2>/dev/null)
    c=$(grep '\*' $d/pp_dpm_sclk | head -1)
    [ $((pw / 1000000)) -gt 500 ] && echo -n "[$((pw / 1000000)) W, $c] "
  done
}
for p in 1 100 101 102 103 116 117 118 119 1; do
    ./mempower $p 5 rand > /tmp/mp_$p.txt &
    BP=$!
    sleep 3; sens; sleep 1; sens; echo
    wait $BP
    cat /tmp/mp_$p.txt
done
./mempower 1 5 > /tmp/mp_c.txt & BP=$!; sleep 3; sens; sleep 1; sens; echo; wait $BP; echo "constant input:"; cat /tmp/mp_c.txt
