#!/bin/bash
# GPU box: clock / power of the part under a sustained bare f16 MFMA stream, smooth vs random operands (rocm-smi sampled beside it).
#   bash tools/mfma_power_probe.sh > gpurun_out/power.txt
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/mfma_f16_ubench.hip -o /tmp/ub || exit 1
smi() { /opt/rocm/bin/rocm-smi --showclocks --showpower --showtemp --csv 2>/dev/null | grep "^card" | head -1 | cut -d, -f1-14; }
echo "idle: $(smi)"
for v in smooth random; do
  /tmp/ub $v 8 > /tmp/ub_$v.txt &
  PID=$!
  sleep 3; echo "$v +3s: $(smi)"; sleep 2; echo "$v +5s: $(smi)"; sleep 2; echo "$v +7s: $(smi)"
  wait $PID; cat /tmp/ub_$v.txt
done
/opt/rocm/bin/rocm-smi --showmaxpower 2>/dev/null | grep -i "max\|power" | head -3
/opt/rocm/bin/rocm-smi --showclocks --showpower --csv 2>/dev/null | head -1 | cut -d, -f1-14
