"""GPU-box helper (round 6): HBM bytes, L2 hit rate and texture-addresser duty of every kernel IN THE NETWORK, with the product's
batched launch plan (7 time instants per launch, one trunk set so that nothing overlaps), from rocprofv3 PMC passes over bench.py.

    python tools/pmc_innet.py <outdir> [tag, default r06]       -> <outdir>/<tag>_pmc_innet.json

Separate --pmc passes with --kernel-trace only (the pool refuses / crashes on other combinations): FETCH_SIZE | WRITE_SIZE, TCC_HIT,
TCC_MISS | TA_TA_BUSY.  HBM bytes per launch = 2 x FETCH_SIZE x 1024 + WRITE_SIZE x 1024 (gfx950: FETCH_SIZE counts 64 B per 128-B
request, MI355X_MICROARCH.md "HBM").  The two fat warps of a window (Ft: sources = trunk features shared by the 7 time instants; rF:
sources = the refined features of each time instant) are told apart by dispatch order; bench.py reads the result for its
roofline_hbm block (frac_by_pmc_traffic per half) and for the SepConvGRU launches."""
import collections
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PASSES = {'fetch': 'FETCH_SIZE GRBM_GUI_ACTIVE', 'write': 'WRITE_SIZE TCC_HIT TCC_MISS', 'ta': 'TA_TA_BUSY GRBM_GUI_ACTIVE'}


def short(k):
    return k.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]


def main():
    outdir = os.path.abspath(sys.argv[1])
    tag = sys.argv[2] if len(sys.argv) > 2 else 'r06'
    os.makedirs(outdir, exist_ok=True)
    per_disp = collections.defaultdict(lambda: collections.defaultdict(dict))     # kernel -> dispatch id -> counter -> value
    dur = collections.defaultdict(dict)
    for name, ctrs in PASSES.items():
        d = os.path.join(outdir, 'innet', name)
        os.makedirs(d, exist_ok=True)
        cmd = ['rocprofv3', '--pmc'] + ctrs.split() + ['--kernel-trace', '--output-format', 'csv', '-d', d, '-o', 'r', '--', sys.executable,
                                                        os.path.join(ROOT, 'bench.py'), '--steps', '2', '--warmup', '1', '--no-cpu-baseline', '--no-verify']
        subprocess.run(cmd, cwd='/tmp', env=dict(os.environ, DEMFI_NTRUNK='1', TMPDIR='/tmp'), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=False)
        for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
            for r in csv.DictReader(open(f)):
                k = short(r['Kernel_Name'])
                did = int(r['Dispatch_Id'])
                per_disp[(name, k)][did][r['Counter_Name']] = per_disp[(name, k)][did].get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
        for f in glob.glob(d + '/**/*kernel_trace.csv', recursive=True):
            for r in csv.DictReader(open(f)):
                dur[(name, short(r['Kernel_Name']))].setdefault('ns', []).append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))

    def mean(vals):
        vals = list(vals)
        return sum(vals) / max(1, len(vals))

    def summarise(sel):
        """sel(kernel, ordinal of the dispatch among that kernel's dispatches) -> bool"""
        out = {}
        kernels = sorted({k for (_, k) in per_disp})
        for k in kernels:
            row = {}
            for name in PASSES:
                ds = sorted(per_disp.get((name, k), {}).items())
                ds = [v for i, (_, v) in enumerate(ds) if sel(k, i)]
                if not ds:
                    continue
                for c in ds[0]:
                    row[c] = mean(v.get(c, 0.0) for v in ds)
                row['dispatches'] = len(ds)
            if 'FETCH_SIZE' in row and 'WRITE_SIZE' in row:
                out[k] = {'hbm_bytes_per_launch': 2 * row['FETCH_SIZE'] * 1024 + row['WRITE_SIZE'] * 1024, 'fetch_KiB': row['FETCH_SIZE'], 'write_KiB': row['WRITE_SIZE'],
                          'dispatches': row['dispatches'], 'l2_hit': row.get('TCC_HIT', 0) / max(1.0, row.get('TCC_HIT', 0) + row.get('TCC_MISS', 0)),
                          # TA_TA_BUSY summed over the 256 TAs / (GRBM_GUI_ACTIVE summed over the 8 XCDs x 32 CUs each)
                          'ta_busy_frac': row.get('TA_TA_BUSY', 0) / max(1.0, row.get('GRBM_GUI_ACTIVE', 0) * 32)}
        return out
    allk = summarise(lambda k, i: True)
    out = {'source': 'rocprofv3 --pmc (three separate passes, --kernel-trace only) over bench.py --steps 2 --warmup 1 with DEMFI_NTRUNK=1: the batched '
                     'launch plan (7 time instants per launch) with nothing overlapping; 2 x FETCH_SIZE + WRITE_SIZE (KiB), gfx950 correction; per launch',
           'note': 'FETCH_SIZE / WRITE_SIZE count at the L2s\' fabric port: traffic the 256 MB Infinity Cache serves is included (a tensor re-read '
                   'within a window, like the trunk features the seven Ft warps gather from, counts every time although it does not reach HBM)',
           'kernels': allk}
    wk = [k for k in allk if 'warp_blend_fat' in k]
    if wk:
        k = wk[0]
        out['warp_blend_fat'] = {'Ft': summarise(lambda kk, i: kk == k and i % 2 == 0).get(k), 'rF': summarise(lambda kk, i: kk == k and i % 2 == 1).get(k),
                                 'time_instants_per_launch': 7}
    gk = [k for k in allk if 'gru_sep5' in k]
    for k in gk:
        out.setdefault('gru', {})['r' if '<1>' in k else 'zq'] = allk[k]
    sk = [k for k in allk if 'conv_sep5' in k]
    if sk:
        out.setdefault('gru', {})['round5_kernel'] = allk[sk[0]]
    path = os.path.join(outdir, tag + '_pmc_innet.json')
    json.dump(out, open(path, 'w'), indent=1)
    print('wrote', path)
    for k, v in sorted(allk.items(), key=lambda kv: -kv[1]['hbm_bytes_per_launch'] * kv[1]['dispatches'])[:16]:
        print('%-56s %9.1f MB/launch  L2 hit %.2f  TA busy %.2f  (%d launches)' % (k[:56], v['hbm_bytes_per_launch'] / 1e6, v['l2_hit'], v['ta_busy_frac'], v['dispatches']))
    for h in ('Ft', 'rF'):
        v = (out.get('warp_blend_fat') or {}).get(h)
        if v:
            print('warp_blend_fat %s: %.1f MB per launch of 7 time instants = %.1f MB per time instant (algorithmic 380.6), L2 hit %.2f, TA busy %.2f' %
                  (h, v['hbm_bytes_per_launch'] / 1e6, v['hbm_bytes_per_launch'] / 7e6, v['l2_hit'], v['ta_busy_frac']))


if __name__ == '__main__':
    main()
