#!/bin/bash
# Identity and power / clock state of the GPU box a gpurun call landed on (the pool's boxes run the same kernel at
# different speeds: DESIGN 5 "box spread"): printed at the start of every evidence run, kept beside its numbers.
echo "host $(hostname)  kernel $(uname -r)  cpus $(nproc)  cgroup cpu.max $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)"
for d in /sys/class/drm/card[0-9]*/device; do
  [ -e $d/pp_dpm_sclk ] || continue
  echo "== $d  unique_id $(cat $d/unique_id 2>/dev/null)  vbios $(cat $d/vbios_version 2>/dev/null)"
  echo "sclk levels: $(tr '\n' ' ' < $d/pp_dpm_sclk)"
  echo "mclk levels: $(tr '\n' ' ' < $d/pp_dpm_mclk 2>/dev/null)"
  echo "perf level: $(cat $d/power_dpm_force_performance_level 2>/dev/null)  xcp/partition: $(cat $d/current_compute_partition 2>/dev/null) / $(cat $d/current_memory_partition 2>/dev/null)"
  for h in $d/hwmon/hwmon*; do
    for f in power1_cap power1_cap_max power1_cap_default power1_average power1_input temp1_input temp2_input temp3_input freq1_input; do
      [ -e $h/$f ] && echo "$f $(cat $h/$f 2>/dev/null)"
    done
  done
done
rocm-smi --showpower --showclocks --showmaxpower --showperflevel --showuniqueid 2>/dev/null | grep -v "^$" | head -40
