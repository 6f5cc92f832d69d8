mkdir -p gpurun_out/r3k
timeout 2700 python -m pytest tests -m gpu -q > gpurun_out/r3k/tests_gpu.txt 2>&1
tail -12 gpurun_out/r3k/tests_gpu.txt
python bench.py --steps 20 --warmup 5 --with-png --profile-ops gpurun_out/r3k/per_launch.txt > gpurun_out/r3k/bench.json 2> gpurun_out/r3k/bench.err
tail -2 gpurun_out/r3k/bench.err | grep -v amdgpu; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3k/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['verified']['mismatching_bytes'], json.dumps(d['roofline']['variants']), d['roofline']['frac'], d.get('png_pipeline'), d['roofline_hbm']['frac'], d.get('final_frames_only'))
PY
