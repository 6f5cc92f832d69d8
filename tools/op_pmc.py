"""GPU-box tool: HBM bytes of one named op of the batched 720p plan (rocprofv3 PMC, separate passes for FETCH_SIZE / WRITE_SIZE).

    python tools/op_pmc.py <outdir> <kernel-name substring> <op name> [<op name> ...]

Runs tools/op_time.py <op> under rocprofv3 and averages the counters over the LAST 6 dispatches of the kernel (= the timed,
in-sequence launches of that op).  HBM bytes = 2 x FETCH_SIZE x 1024 (gfx950 correction, MI355X_MICROARCH.md) + WRITE_SIZE x 1024."""
import collections
import csv
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def mfma_pass(outdir, kname, op):
    """matrix-pipe busy fraction and effective clock of the op's kernel: SQ_VALU_MFMA_BUSY_CYCLES (summed over the 1 024 SIMDs) /
    (GRBM_GUI_ACTIVE (summed over the 8 XCDs) x 128); clock = GRBM_GUI_ACTIVE / 8 / duration."""
    d = os.path.join(outdir, op.replace('.', '_'), 'mfma')
    os.makedirs(d, exist_ok=True)
    cmd = ['rocprofv3', '--pmc', 'SQ_VALU_MFMA_BUSY_CYCLES', 'GRBM_GUI_ACTIVE', 'SQ_INSTS_VALU_MFMA_MOPS_F16', '--kernel-trace', '--output-format', 'csv',
           '-d', d, '-o', 'r', '--', sys.executable, os.path.join(ROOT, 'tools', 'op_time.py'), op]
    subprocess.run(cmd, cwd='/tmp', stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=False)
    acc = collections.defaultdict(dict)
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if kname in r['Kernel_Name']:
                a = acc[int(r['Dispatch_Id'])]
                a[r['Counter_Name']] = a.get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
    dur = {}
    for f in glob.glob(d + '/**/*kernel_trace.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if kname in r['Kernel_Name']:
                dur[int(r['Dispatch_Id'])] = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    ids = sorted(acc)[-6:]
    if not ids:
        print('%-34s no dispatches of %s found' % (op, kname))
        return
    busy = sum(acc[i].get('SQ_VALU_MFMA_BUSY_CYCLES', 0) for i in ids)
    gui = sum(acc[i].get('GRBM_GUI_ACTIVE', 0) for i in ids)
    ns = sum(dur.get(i, 0) for i in ids)
    print('%-34s matrix pipes busy %.3f of the cycles, effective clock %.2f GHz (profiling mode), %.3f ms per launch' %
          (op, busy / max(1.0, gui * 128), gui / 8 / max(1.0, ns), ns / len(ids) / 1e6))


def main():
    outdir, kname = sys.argv[1], sys.argv[2]
    if sys.argv[3] == '--mfma':
        for op in sys.argv[4:]:
            mfma_pass(outdir, kname, op)
        return
    for op in sys.argv[3:]:
        vals = {}
        for ctr in ('FETCH_SIZE', 'WRITE_SIZE'):
            d = os.path.join(outdir, op.replace('.', '_'), ctr)
            os.makedirs(d, exist_ok=True)
            cmd = ['rocprofv3', '--pmc', ctr, '--kernel-trace', '--output-format', 'csv', '-d', d, '-o', 'r', '--',
                   sys.executable, os.path.join(ROOT, 'tools', 'op_time.py'), op]
            subprocess.run(cmd, cwd='/tmp', stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=False)
            rows = []
            for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
                per = collections.OrderedDict()
                for r in csv.DictReader(open(f)):
                    if kname in r['Kernel_Name'] and r['Counter_Name'] == ctr:
                        per[int(r['Dispatch_Id'])] = per.get(int(r['Dispatch_Id']), 0.0) + float(r['Counter_Value'])
                rows = [per[k] for k in sorted(per)]
            vals[ctr] = sum(rows[-6:]) / max(1, len(rows[-6:]))
        hbm = 2 * vals['FETCH_SIZE'] * 1024 + vals['WRITE_SIZE'] * 1024
        px = 7 * 736 * 1280
        print('%-34s fetch %.1f MB  write %.1f MB  total %.1f MB = %.0f B/px' % (op, 2 * vals['FETCH_SIZE'] * 1024 / 1e6, vals['WRITE_SIZE'] * 1024 / 1e6,
                                                                              hbm / 1e6, hbm / px))


if __name__ == '__main__':
    main()
