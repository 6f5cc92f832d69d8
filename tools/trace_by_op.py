"""Map a rocprofv3 kernel trace of a SEQUENTIAL bench run (DEMFI_NTRUNK=1: the trunk of the next window does not overlap; plan
order) back to the ops of the launch plan and average the true kernel durations per op.

    python tools/trace_by_op.py <kernel_trace.csv> <ops.txt written by bench.py --profile-ops> [out.md] [passes per window] [roofline.json]

The trace holds, per window, the trunk ops followed by `passes` per-t sequences: 1 for the batched plan (every launch
covers the 7 time instants of the window; the default runner), 7 for one graph per time instant (DEMFI_TB=0 DEMFI_NCTX=1).  Output: per op (plan order) the kernel name, calls and mean / min duration --
the in-sequence ground truth the per-launch HIP-event numbers of bench.py are compared with."""
import csv
import sys
from collections import defaultdict


def main():
    trace, ops_path = sys.argv[1], sys.argv[2]
    out = open(sys.argv[3], 'w') if len(sys.argv) > 3 else sys.stdout
    passes = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    rows = []
    for r in csv.DictReader(open(trace)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']) - int(r['Start_Timestamp']), r['Kernel_Name']))
    rows.sort()
    ops = [l.split() for l in open(ops_path) if l.strip()]
    n_trunk = sum(1 for o in ops if o[0] == 'trunk')
    n_t = len(ops) - n_trunk
    names = [r[2] for r in rows]
    # kernels per plan op: cfr = 2 kernels (far + tile), every other op 1
    per = [2 if o[1] == 'cfr' else 1 for o in ops]
    k_trunk, k_t = sum(per[:n_trunk]), sum(per[n_trunk:])
    # find windows: a window = k_trunk + 7 * k_t kernels (+ ingest / egress kernels of the runner, which carry other names)
    plan_kernels = [n for n in names if 'u8_' not in n and 'frame_to_u8' not in n and 'reflect_pad' not in n and 'Memcpy' not in n
                    and 'fill' not in n.lower() and 'copy' not in n.lower() and 'elementwise' not in n.lower()]
    keep = set(plan_kernels)
    idx = [i for i, n in enumerate(names) if n in keep]
    # the runner warms every context eagerly before capturing its graphs (full trunk + one per-t pass per context); the uint8
    # pipeline then replays, per window, the trunk BODY (ops 2.. : s2d / overlay are done by the fused ingest kernel) and 7
    # per-t passes
    prefix = k_trunk + k_t
    body0 = 2
    k_body = sum(per[body0:n_trunk])
    win = k_body + passes * k_t
    nwin = (len(idx) - prefix) // win
    acc = defaultdict(list)
    first = None
    for wi in range(nwin):
        base = prefix + wi * win
        # the per-op profile passes bench.py runs after the timed region (full trunks, one per-t pass each) follow the windows:
        # stop at the first "window" whose kernel sequence differs from window 0
        sig = [rows[idx[base + j]][2].split('<')[0] for j in range(win)]
        if first is None:
            first = sig
        elif sig != first:
            break
        pos = 0
        for oi in range(body0, n_trunk):
            d = sum(rows[idx[base + pos + j]][1] for j in range(per[oi]))
            acc[oi].append((d, rows[idx[base + pos]][2]))
            pos += per[oi]
        for t in range(passes):
            for oi in range(n_trunk, len(ops)):
                d = sum(rows[idx[base + pos + j]][1] for j in range(per[oi]))
                acc[oi].append((d, rows[idx[base + pos]][2]))
                pos += per[oi]
    # sanity: the kernel family of every op must be the same in all its samples
    for oi, v in acc.items():
        assert len({k.split('<')[0] for _, k in v}) == 1, (ops[oi], {k for _, k in v})
    out.write('| segment | op | name | kernel | calls | mean us | min us | bench events us |\n|---|---|---|---|---|---|---|---|\n')
    for oi, o in enumerate(ops):
        ds = [d for d, _ in acc[oi]]
        if not ds:
            continue
        kn = acc[oi][0][1]
        kn = kn.split('(anonymous namespace)::')[-1].split('(')[0][:44]
        out.write('| %s | %s | %s | %s | %d | %.1f | %.1f | %.1f |\n' % (o[0], o[1], o[2], kn, len(ds), sum(ds) / len(ds) / 1e3, min(ds) / 1e3,
                                                                  float(o[3]) * 1e3))


    # the ten launches of bench.py's roofline object (D1 residual blocks): mean kernel duration in this trace, for roofline.frac_rocprof
    if len(sys.argv) > 5:
        import json
        grp = [oi for oi, o in enumerate(ops) if o[1] in ('conv', 'resblock') and o[2].startswith('Decoder_res.') and acc[oi]]
        if grp:
            means = [sum(d for d, _ in acc[oi]) / len(acc[oi]) / 1e6 for oi in grp]
            json.dump({'avg_launch_ms': round(sum(means) / len(means), 5), 'launches': len(grp), 'samples_per_launch': len(acc[grp[0]]),
                       'fused': all(ops[oi][1] == 'resblock' for oi in grp),      # round 5: one launch per residual block (2 convolutions)
                       'no_residual_ms': round(sum(m for oi, m in zip(grp, means) if ops[oi][2].endswith('.conv1')) / max(1, sum(1 for oi in grp if ops[oi][2].endswith('.conv1'))), 5),
                       'residual_ms': round(sum(m for oi, m in zip(grp, means) if ops[oi][2].endswith('.conv2')) / max(1, sum(1 for oi in grp if ops[oi][2].endswith('.conv2'))), 5),
                       'source': 'rocprofv3 --kernel-trace of a sequential bench run (DEMFI_NTRUNK=1), tools/trace_by_op.py; another box than the live HIP-event numbers'},
                      open(sys.argv[5], 'w'))


if __name__ == '__main__':
    main()
