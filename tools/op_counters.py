"""GPU-box tool: SQ instruction / wait counters of the kernel behind named ops of the batched 720p plan (rocprofv3 --pmc, two passes).

    python tools/op_counters.py <outdir> <kernel-name substring> <op name> [<op name> ...]

Averages over the last 6 dispatches of the kernel (the timed, in-sequence launches of tools/op_time.py).  Printed per op: instructions
per wave by class, and the share of wave-cycles spent waiting -- "is the wave executing a lot of instructions or waiting for something?"."""
import collections
import csv
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PASSES = ['SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU',
          'SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_MFMA']


def main():
    outdir, kname, ops = os.path.abspath(sys.argv[1]), sys.argv[2], sys.argv[3:]
    for op in ops:
        acc = collections.defaultdict(float)
        for i, ctrs in enumerate(PASSES):
            d = os.path.join(outdir, op.replace('.', '_'), 'p%d' % i)
            os.makedirs(d, exist_ok=True)
            cmd = ['rocprofv3', '--pmc'] + ctrs.split() + ['--kernel-trace', '--output-format', 'csv', '-d', d, '-o', 'r', '--', sys.executable,
                   os.path.join(ROOT, 'tools', 'op_time.py'), op]
            subprocess.run(cmd, cwd='/tmp', stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=False)
            per = collections.defaultdict(dict)
            for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
                for r in csv.DictReader(open(f)):
                    if kname in r['Kernel_Name']:
                        a = per[int(r['Dispatch_Id'])]
                        a[r['Counter_Name']] = a.get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
            ids = sorted(per)[-6:]
            for f in glob.glob(d + '/**/*kernel_trace.csv', recursive=True):
                dur = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-3 for r in csv.DictReader(open(f)) if kname in r['Kernel_Name']]
                if dur:
                    acc['us'] = sum(dur[-6:]) / len(dur[-6:])
            for k in {c for i_ in ids for c in per[i_]}:
                acc[k] = sum(per[i_].get(k, 0.0) for i_ in ids) / max(1, len(ids))
        w = max(1.0, acc['SQ_WAVES'])
        print('%s  (%s, %d waves per launch)' % (op, kname, w))
        print('   instructions per wave: VALU %.0f  SALU %.0f  VMEM %.0f  LDS %.0f  SMEM %.0f  MFMA %.0f' %
              tuple(acc[k] / w for k in ('SQ_INSTS_VALU', 'SQ_INSTS_SALU', 'SQ_INSTS_VMEM', 'SQ_INSTS_LDS', 'SQ_INSTS_SMEM', 'SQ_INSTS_MFMA')))
        if acc['us']:
            # SQ_WAVE_CYCLES counts in units of 4 cycles; every wave of these persistent kernels lives for the whole launch
            print('   launch %.1f us under the counters; wave-cycles per wave x 4 / launch time = %.2f GHz shader clock; SQ busy cycles / time = %.2f GHz' %
                  (acc['us'], 4 * acc['SQ_WAVE_CYCLES'] / w / acc['us'] * 1e-3, acc['SQ_BUSY_CYCLES'] / acc['us'] * 1e-3))
        wc = max(1.0, acc['SQ_WAVE_CYCLES'])
        print('   of the wave-cycles: waiting (any) %.2f, waiting for an instruction to issue %.2f, an instruction active %.2f '
              '(VALU %.2f VMEM %.2f LDS %.2f scalar %.2f)' % tuple(acc[k] / wc for k in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_ACTIVE_INST_VALU',
                                                                                           'SQ_ACTIVE_INST_VMEM', 'SQ_ACTIVE_INST_LDS', 'SQ_ACTIVE_INST_SCA')))


if __name__ == '__main__':
    main()
