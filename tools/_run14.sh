mkdir -p gpurun_out/r3u
timeout 2700 python -m pytest tests -m gpu -q > gpurun_out/r3u/tests_gpu.txt 2>&1
tail -5 gpurun_out/r3u/tests_gpu.txt
python bench.py --steps 20 --warmup 5 --profile-ops gpurun_out/r3u/per_launch.txt > gpurun_out/r3u/bench.json 2> gpurun_out/r3u/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3u/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['verified']['mismatching_bytes'], d['roofline']['frac'], d['roofline_hbm']['frac'], d['breakdown_ms'], d.get('psnr'), d.get('cpu_baseline',{}).get('value'))
PY
