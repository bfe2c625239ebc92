#!/bin/bash
# Which power / clock telemetry an ordinary user can read on the GPU box (round 6).
out=gpurun_out/telemetry_probe.txt
{
echo "== whoami: $(whoami)"
echo "== hwmon"
for h in /sys/class/drm/card*/device/hwmon/hwmon*; do
  echo "-- $h"; ls $h 2>&1 | tr '\n' ' '; echo
  for f in power1_average power1_input power1_cap power1_cap_max freq1_input freq2_input temp1_input in0_input; do
    [ -e $h/$f ] && echo "$f = $(cat $h/$f 2>&1)"
  done
done
echo "== pp_dpm_sclk"; for c in /sys/class/drm/card*/device; do echo "-- $c"; cat $c/pp_dpm_sclk 2>&1 | tail -5; cat $c/gpu_busy_percent 2>&1; done
echo "== gpu_metrics"; ls -la /sys/class/drm/card*/device/gpu_metrics 2>&1
echo "== rocm-smi"; timeout 60 rocm-smi --showpower --showclocks 2>&1 | head -40
echo "== amd-smi"; which amd-smi; timeout 60 amd-smi metric -p -c 2>&1 | head -60
echo "== python amdsmi"; python3 -c "import amdsmi; print('amdsmi ok', amdsmi.__file__)" 2>&1 | tail -1
ls /opt/rocm/lib | grep -i -E "smi|oam" | head
} > $out 2>&1
echo done
