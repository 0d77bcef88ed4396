#!/bin/bash
# GPU box: the standalone packed-fp32-beside-MFMA reproduction at the default power cap and, where the driver permits, at a
# lowered one (separates a voltage-droop explanation -- mismatches would follow the aggressor's POWER -- from a structural
# hazard -- they follow the aggressor's INSTRUCTION).      bash tools/pk_f32_standalone_repro.sh > gpurun_out/pk_repro.txt
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 tools/pk_f32_standalone_repro.hip -o /tmp/pkrepro || exit 1
smi() { /opt/rocm/bin/rocm-smi --showclocks --showpower --csv 2>/dev/null | grep "^card" | head -1 | cut -d, -f1-14; }
echo "# idle: $(smi)"
/opt/rocm/bin/rocm-smi --showmaxpower 2>/dev/null | grep -i "max" | head -2
echo "## default power cap"
/tmp/pkrepro ${1:-400} &
PID=$!; sleep 4; echo "# during: $(smi)"; wait $PID
for CAP in 750 500; do
  if /opt/rocm/bin/rocm-smi --setpoweroverdrive $CAP --autorespond y > /tmp/cap.txt 2>&1 && ! grep -qi "not supported\|error\|fail\|denied" /tmp/cap.txt; then
    echo "## power cap lowered to $CAP W: $(tail -2 /tmp/cap.txt | tr '\n' ' ')"
    /tmp/pkrepro ${1:-400} &
    PID=$!; sleep 4; echo "# during: $(smi)"; wait $PID
  else
    echo "## power cap $CAP W: rocm-smi refused ($(tail -1 /tmp/cap.txt))"
  fi
done
/opt/rocm/bin/rocm-smi --resetpoweroverdrive --autorespond y > /dev/null 2>&1
echo "# after reset: $(smi)"
