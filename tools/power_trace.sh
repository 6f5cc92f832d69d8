#!/bin/bash
# GPU-box helper: socket power / sclk sampled from the amdgpu hwmon sysfs nodes (10 ms period; rocm-smi itself blocks for seconds
# while the GPU is busy) while the dominant kernel's batch-21 launches run back to back (the D1 residual-block shape of the
# batched plan), so that "power-limited" is evidence and not inference.
# usage: tools/power_trace.sh <out.txt>     (about 15 s)
OUT=${1:-$GRAFT_REPO_ROOT/gpurun_out/power_trace.txt}
cd $GRAFT_REPO_ROOT
HW=$(ls -d /sys/class/drm/card*/device/hwmon/hwmon* 2>/dev/null | head -1)
{
  echo "# hwmon node: $HW"
  echo "# power cap: $(cat $HW/power1_cap 2>/dev/null) uW (max $(cat $HW/power1_cap_max 2>/dev/null), default $(cat $HW/power1_cap_default 2>/dev/null))"
  rocm-smi --showmaxpower --showpower --showclocks 2>/dev/null | grep -E "Power|sclk|mclk" | sed 's/  */ /g'
  echo "# columns: time_s power_W sclk_MHz    (power1_average or power1_input; freq1_input)"
} > $OUT
sample() {
  while true; do
    P=$(cat $HW/power1_average 2>/dev/null || cat $HW/power1_input 2>/dev/null)
    F=$(cat $HW/freq1_input 2>/dev/null)
    echo "$(date +%s.%N) $((P / 1000000)) $((F / 1000000))"
    sleep 0.01
  done
}
sample >> $OUT & SP=$!
sleep 1
echo "# --- load starts $(date +%s.%N): conv_probe c3x3 batch 21 x 300 launches, then c3x3res x 300 (post-ReLU-like data)" >> $OUT
PROBE_B=21 PROBE_DATA=relu python tools/conv_probe.py c3x3 300 2>/dev/null >> $OUT
PROBE_B=21 PROBE_DATA=relu python tools/conv_probe.py c3x3res 300 2>/dev/null >> $OUT
echo "# --- load ends $(date +%s.%N)" >> $OUT
sleep 0.5
kill $SP
python3 - "$OUT" <<'PY'
import sys
rows = []
marks = []
for l in open(sys.argv[1]):
    if l.startswith('# --- load'):
        marks.append(float(l.split()[4].rstrip(':')))
    elif l[0].isdigit():
        a = l.split()
        if len(a) == 3:
            rows.append((float(a[0]), float(a[1]), float(a[2])))
if len(marks) == 2 and rows:
    idle = [r for r in rows if r[0] < marks[0]]
    # the first seconds of the "load" window are python / torch start-up: take the busy samples by power
    load = [r for r in rows if marks[0] < r[0] < marks[1]]
    busy = [r for r in load if r[1] > 0.6 * max(x[1] for x in load)]
    m = lambda xs, i: sum(x[i] for x in xs) / max(1, len(xs))
    print('# summary: idle %.0f W / %.0f MHz (%d samples); under the 64->64 launches %.0f W (max %.0f) / %.0f MHz reported sclk (min %.0f, max %.0f; %d samples)' %
          (m(idle, 1), m(idle, 2), len(idle), m(busy, 1), max(x[1] for x in busy), m(busy, 2), min(x[2] for x in busy), max(x[2] for x in busy), len(busy)))
PY
