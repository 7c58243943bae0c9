# Round 6, GPU session 3: is the step power-bound?  (hwmon power / shader-clock trace per phase)
set -u
OUT=gpurun_out/r6_s3
mkdir -p $OUT
ls /sys/class/drm/*/device/hwmon/*/ 2>/dev/null | head -40 > $OUT/hwmon_ls.txt
(timeout 600 python tools/power_trace.py 2>&1 | grep -v amdgpu.ids) > $OUT/power_trace.txt
cat $OUT/power_trace.txt | cut -c1-250
