#!/bin/bash
# GPU-box helper: same-box A/B of environment-selected variants.  Runs bench.py once per variant and prints the headline plus the
# in-sequence per-launch means (HIP events between consecutive launches of the whole plan) of selected ops.
#   tools/ab_ops.sh <outdir> "<ENV1=.. ENV2=..>" "<...>" ...      (each argument = one variant's environment; "" = product)
# DEMFI_AB_ROUNDS=2 repeats the whole list (alternating order: box drift shows up as a difference between the rounds).
out=$1; shift
mkdir -p "$out"
rounds=${DEMFI_AB_ROUNDS:-1}
for r in $(seq 1 "$rounds"); do
i=0
for v in "$@"; do
  tag=$(echo "r${r}v$i $v" | tr ' =/' '___')
  env $v python bench.py --steps ${DEMFI_AB_STEPS:-6} --warmup 2 --no-cpu-baseline --profile-ops "$out/ops_$tag.txt" > "$out/bench_$tag.json" 2> "$out/err_$tag.txt"
  python - "$out/ops_$tag.txt" "$out/bench_$tag.json" "$v" <<'PY'
import sys, json, collections
ops, bench, tag = sys.argv[1:4]
acc = collections.defaultdict(list)
segsum = collections.defaultdict(float)
try:
    for l in open(ops):
        f = l.split()
        if len(f) >= 4:
            acc[f[2] if f[1] in ('conv', 'resblock', 'gru_r', 'gru_zq') else f[1]].append(float(f[3]) / (float(f[7][3:]) if len(f) > 7 and f[1] == 'warp_fat' else 1.0))
            segsum[f[0]] += float(f[3])
except OSError:
    pass
try:
    d = json.loads(open(bench).read().strip().splitlines()[-1])
    head = '%.2f fps %.2f ms/win verified=%s final_only=%s' % (d['value'], d['ms_per_step'], d.get('verified', {}).get('mismatching_bytes'),
                                                               d.get('final_frames_only', {}).get('value'))
except Exception as e:
    head = 'bench failed: %s' % e
sel = ['warp_fat', 'warp_thin', 'pack', 'cfr', 'fgac', 'Dec_last2', 'Dec_last2_2', 'Booster_Module.flow_occ.conv2', 'Decoder_res.0.conv1',
       'Decoder_res.0.conv2', 'Decoder_res.0', 'Decoder_res_2.0', 'Booster_Module.GB.convzr1', 'Booster_Module.GB.convq1', 'Booster_Module.GB.convr1', 'Booster_Module.GB.step1.convzq', 'Booster_Module.GB.convr2', 'Booster_Module.GB.step2.convzq', 'Ch_Reducer', 'Refine_Module.enc1#t', 'Refine_Module.enc1#aF', 'Refine_Module.enc2', 'Refine_Module.enc3', 'FAC_FB_Module.shared_FGAC.w_gen', 'Dec_first_2#dyn', 'Dec_first_2#rec', 'Dec_first_2', 'Refine_Module.dec0', 'Refine_Module.dec1', 'Refine_Module.dec2', 'FF_RDB_Module.UPNet.2', 'FF_RDB_Module.UPNet.2#F0', 'FF_RDB_Module.UPNet.2#F1', 'FF_RDB_Module.UPNet.2#f']
short = lambda k: k.split('.')[-1] if k.count('.') > 1 else k
print('[%s] %s' % (tag or 'product', head))
print('    ' + '  '.join('%s %.4f x%d' % (short(k), sum(acc[k]) / len(acc[k]), len(acc[k])) for k in sel if acc[k]))
print('    sum of per-launch times %.2f ms (%s), %d launches' % (sum(segsum.values()), ' '.join('%s %.2f' % kv for kv in segsum.items()),
                                                                sum(len(v) for v in acc.values())))
PY
  i=$((i+1))
done
done
