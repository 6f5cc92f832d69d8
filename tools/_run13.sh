mkdir -p gpurun_out/r3t
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_e2e.py -q -x 2>&1 | tail -3
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --profile-ops gpurun_out/r3t/per_launch.txt > gpurun_out/r3t/bench.json 2> gpurun_out/r3t/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3t/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['verified']['mismatching_bytes'], d['roofline']['frac'], d['breakdown_ms'])
PY
grep -E "dec3" gpurun_out/r3t/per_launch.txt
export DEMFI_HIP_LIB=$PWD/demfi_amd/csrc/libdemfi_hip_trace.so
for op in Dec_last2 Dec_last2_2 Booster_Module.flow_occ.conv2 Booster_Module.Mixer.conv_blend1 "Refine_Module.dec3#p00f"; do
  python tools/phase_trace.py "op:$op" 2>gpurun_out/r3t/trace.err >> gpurun_out/r3t/trace.txt
done
cat gpurun_out/r3t/trace.txt; grep -v amdgpu gpurun_out/r3t/trace.err | tail -3
