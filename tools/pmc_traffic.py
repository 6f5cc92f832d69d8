"""GPU-box helper: rocprofv3 PMC passes (separate runs, --kernel-trace only) of the probes of the two roofline kernels and
a JSON summary for bench.py (profiles/<round>_pmc_traffic.json).

    python tools/pmc_traffic.py <outdir> [round tag, default r04]

HBM bytes per launch = 2 x FETCH_SIZE x 1024 (gfx950: FETCH_SIZE counts 64 B per 128-B request, MI355X_MICROARCH.md "HBM")
+ WRITE_SIZE x 1024, per dispatch of the kernel.  The dominant conv is probed with and without the residual input (the D1
residual blocks alternate conv1 / conv2); warp_blend_fat with white-noise flows (sigma 8 px)."""
import collections
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PASSES = {'fetch': 'FETCH_SIZE GRBM_GUI_ACTIVE', 'write': 'WRITE_SIZE TCC_HIT TCC_MISS',
          'mfma': 'SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA',
          'ta': 'TA_TA_BUSY GRBM_GUI_ACTIVE SQ_WAVES',
          'lds': 'SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM'}


def run(case, outdir, env=None):
    res = {}
    for tag, ctrs in PASSES.items():
        d = os.path.join(outdir, case, tag)
        os.makedirs(d, exist_ok=True)
        cmd = ['rocprofv3', '--pmc'] + ctrs.split() + ['--kernel-trace', '--output-format', 'csv', '-d', d, '-o', 'r', '--',
                                                        sys.executable, os.path.join(ROOT, 'tools', 'conv_probe.py'), case, '3']
        e = dict(os.environ)
        e.update(env or {})
        subprocess.run(cmd, cwd='/tmp', env=e, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=False)
        for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
            acc = collections.defaultdict(lambda: collections.defaultdict(float))
            disp = collections.defaultdict(set)
            for r in csv.DictReader(open(f)):
                k = r['Kernel_Name']
                acc[k][r['Counter_Name']] += float(r['Counter_Value'])
                disp[k].add(r['Dispatch_Id'])
            for k, v in acc.items():
                if 'persist' in k or 'c64_stg' in k or 'warp_blend_fat' in k or 'cfr_' in k or 'resblock3x3' in k:
                    n = len(disp[k])
                    res.setdefault(k, {}).update({c: x / n for c, x in v.items()})
                    res[k]['dispatches'] = n
        for f in glob.glob(d + '/**/*kernel_trace.csv', recursive=True):
            dur = collections.defaultdict(list)
            for r in csv.DictReader(open(f)):
                dur[r['Kernel_Name']].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
            for k, v in dur.items():
                if k in res:
                    res[k].setdefault('avg_us', {})[tag] = sum(v) / len(v) / 1e3
    return res


def run_bench_pmc(outdir):
    """The same two passes over the BENCH itself (sequential: one per-t context, one trunk context): per-kernel HBM bytes with
    the network's own flows / activations and the cache state of the real launch sequence."""
    res = collections.defaultdict(dict)
    for tag in ('fetch', 'write', 'ta'):
        d = os.path.join(outdir, 'bench', tag)
        os.makedirs(d, exist_ok=True)
        cmd = ['rocprofv3', '--pmc'] + PASSES[tag].split() + ['--kernel-trace', '--output-format', 'csv', '-d', d, '-o', 'r', '--',
                                                               sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '2', '--warmup', '1',
                                                               '--no-cpu-baseline']
        e = dict(os.environ, DEMFI_NCTX='1', DEMFI_NTRUNK='1')
        subprocess.run(cmd, cwd='/tmp', env=e, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=False)
        for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
            acc = collections.defaultdict(lambda: collections.defaultdict(float))
            disp = collections.defaultdict(set)
            for r in csv.DictReader(open(f)):
                k = r['Kernel_Name'].split('(anonymous namespace)::')[-1].split('(')[0]
                acc[k][r['Counter_Name']] += float(r['Counter_Value'])
                disp[k].add(r['Dispatch_Id'])
            for k, v in acc.items():
                res[k].update({c: x / len(disp[k]) for c, x in v.items()})
                res[k]['dispatches'] = len(disp[k])
    out = {}
    for k, v in res.items():
        if 'FETCH_SIZE' in v and 'WRITE_SIZE' in v:
            out[k] = {'hbm_bytes_per_launch': 2 * v['FETCH_SIZE'] * 1024 + v['WRITE_SIZE'] * 1024, 'fetch_KiB': v['FETCH_SIZE'],
                      'write_KiB': v['WRITE_SIZE'], 'dispatches': v['dispatches'],
                      'l2_hit': v.get('TCC_HIT', 0) / max(1.0, v.get('TCC_HIT', 0) + v.get('TCC_MISS', 0)),
                      # texture-addresser busy fraction: TA_TA_BUSY summed over the 256 TAs / (GRBM_GUI_ACTIVE summed over 8 XCDs x 32 CUs)
                      'ta_busy_frac': v.get('TA_TA_BUSY', 0) / max(1.0, v.get('GRBM_GUI_ACTIVE', 0) * 32)}
    return out


def main():
    outdir = os.path.abspath(sys.argv[1])                       # rocprofv3 runs with cwd /tmp
    os.makedirs(outdir, exist_ok=True)
    summary = {}
    for case, env in (('c3x3', {'PROBE_DATA': 'relu'}), ('c3x3res', {'PROBE_DATA': 'relu'}), ('warp', {}), ('cfr', {})):
        summary[case] = run(case, outdir, env)

    def hbm(entry):
        return 2 * entry.get('FETCH_SIZE', 0) * 1024 + entry.get('WRITE_SIZE', 0) * 1024
    conv = [v for k, v in summary['c3x3'].items() if 'persist' in k or 'c64_stg' in k]
    convr = [v for k, v in summary['c3x3res'].items() if 'persist' in k or 'c64_stg' in k]
    warp = [v for k, v in summary['warp'].items() if 'warp_blend_fat' in k]
    out = {'source': 'rocprofv3 --pmc (separate passes) of tools/conv_probe.py c3x3 / c3x3res / warp at 736x1280 fp16 batch 3, '
                     'post-ReLU-like activations; 2 x FETCH_SIZE + WRITE_SIZE (KiB), gfx950 correction',
           'raw': summary}
    if conv and convr:
        a, b = hbm(conv[0]), hbm(convr[0])
        out['dominant_traffic_bytes'] = (a + b) / 2                      # the ten D1 launches: five without, five with residual
        out['dominant_traffic_no_res'] = a
        out['dominant_traffic_res'] = b
        out['dominant_algorithmic_bytes'] = (723.5e6 + 1085.2e6) / 2
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs, SQ_VALU_MFMA_BUSY_CYCLES over the 1024 SIMDs
        out['dominant_mfma_busy_frac'] = conv[0].get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / max(1.0, conv[0].get('GRBM_GUI_ACTIVE', 1) * 128)
        out['dominant_clock_GHz'] = conv[0].get('GRBM_GUI_ACTIVE', 0) / 8 / max(1e-9, conv[0].get('avg_us', {}).get('fetch', 0) * 1e3)
    # the batched per-t plan launches the same kernel over batch 3 x 7 time instants = 21 images
    big = {c: run(c, os.path.join(outdir, 'b21'), {'PROBE_DATA': 'relu', 'PROBE_B': '21'}) for c in ('c3x3', 'c3x3res')}
    summary['c3x3_b21'], summary['c3x3res_b21'] = big['c3x3'], big['c3x3res']
    cb = [v for k, v in big['c3x3'].items() if 'persist' in k or 'c64_stg' in k]
    cbr = [v for k, v in big['c3x3res'].items() if 'persist' in k or 'c64_stg' in k]
    if cb and cbr:
        out['dominant_traffic_bytes_b21'] = (hbm(cb[0]) + hbm(cbr[0])) / 2
        out['dominant_algorithmic_bytes_b21'] = 7 * (723.5e6 + 1085.2e6) / 2
        out['dominant_mfma_busy_frac_b21'] = cb[0].get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / max(1.0, cb[0].get('GRBM_GUI_ACTIVE', 1) * 128)
        out['dominant_clock_GHz_b21'] = cb[0].get('GRBM_GUI_ACTIVE', 0) / 8 / max(1e-9, cb[0].get('avg_us', {}).get('fetch', 0) * 1e3)
    # round 5: the fused residual block (conv1 -> ReLU -> conv2 + identity in one launch) at batch 21: what the D1 blocks of the
    # batched plan launch.  Algorithmic bytes: input + output only.
    rb = run('resblock', os.path.join(outdir, 'rb21'), {'PROBE_DATA': 'relu', 'PROBE_B': '21'})
    summary['resblock_b21'] = rb
    rbk = [v for k, v in rb.items() if 'resblock3x3' in k]
    if rbk:
        out['resblock_traffic_bytes_b21'] = hbm(rbk[0])
        out['resblock_algorithmic_bytes_b21'] = 2.0 * 736 * 1280 * 64 * 2 * 21
        out['resblock_mfma_busy_frac_b21'] = rbk[0].get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / max(1.0, rbk[0].get('GRBM_GUI_ACTIVE', 1) * 128)
        out['resblock_clock_GHz_b21'] = rbk[0].get('GRBM_GUI_ACTIVE', 0) / 8 / max(1e-9, rbk[0].get('avg_us', {}).get('fetch', 0) * 1e3)
        pair = [v for k, v in rb.items() if 'c64_stg' in k]
        if pair:
            out['two_launch_pair_traffic_bytes_b21'] = sum(hbm(v) for v in pair)      # conv1 (no residual) + conv2 (residual) instantiations
    if warp:
        out['warp_traffic_bytes'] = hbm(warp[0])
    innet = run_bench_pmc(outdir)
    out['in_network'] = innet
    wk = [v for k, v in innet.items() if 'warp_blend_fat' in k]
    if wk:
        out['warp_traffic_bytes_probe_white_noise'] = out.get('warp_traffic_bytes')
        out['warp_traffic_bytes'] = wk[0]['hbm_bytes_per_launch']             # the network's own flows, in sequence
    json.dump(out, open(os.path.join(outdir, (sys.argv[2] if len(sys.argv) > 2 else 'r05') + '_pmc_traffic.json'), 'w'), indent=1)
    print(json.dumps({k: v for k, v in out.items() if k not in ('raw', 'in_network')}, indent=1))
    for k, v in sorted(innet.items(), key=lambda kv: -kv[1]['hbm_bytes_per_launch'])[:14]:
        print('%-52s %8.1f MB/launch  L2 hit %.2f  (%d launches)' % (k[:52], v['hbm_bytes_per_launch'] / 1e6, v['l2_hit'], v['dispatches']))


if __name__ == '__main__':
    main()
