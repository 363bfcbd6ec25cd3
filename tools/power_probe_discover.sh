#!/bin/bash
# what power / clock telemetry does the GPU box offer?  (development aid, round 4)
echo "== tools"; which amd-smi rocm-smi 2>&1
echo "== amd-smi metric (power, clock)"; timeout 30 amd-smi metric -p -c 2>&1 | head -60
echo "== amd-smi static limit"; timeout 30 amd-smi static -l 2>&1 | head -40
echo "== rocm-smi"; timeout 30 rocm-smi --showpower --showclocks --showmaxpower 2>&1 | head -40
echo "== sysfs"
for d in /sys/class/drm/card*/device; do
  echo "-- $d"; ls $d 2>/dev/null | tr '\n' ' ' | head -c 2000; echo
  for h in $d/hwmon/hwmon*; do echo "-- $h"; ls $h | tr '\n' ' '; echo; for f in power1_average power1_input power1_cap power1_cap_max freq1_input freq2_input temp1_input; do [ -r $h/$f ] && echo "$f = $(cat $h/$f)"; done; done
  for f in pp_dpm_sclk pp_dpm_mclk gpu_busy_percent pp_power_profile_mode; do [ -r $d/$f ] && { echo "-- $f"; cat $d/$f | head -12; }; done
  [ -r $d/gpu_metrics ] && { echo "-- gpu_metrics bytes: $(wc -c < $d/gpu_metrics)"; }
done
