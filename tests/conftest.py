import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run on the GPU box with -m gpu)')


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN


@pytest.fixture(scope='session')
def synthetic_sd():
    from demfi_amd.weights import synthetic_state_dict
    return synthetic_state_dict(0)


def record_fp16_margin(test, frame, psnr_vs_oracle, dpsnr_vs_gt, gate_db=44.0, **extra):
    """fp16-vs-fp32-oracle acceptance points at full size: every gate appends what it measured to
    gpurun_out/fp16_margins.json (merged back from the GPU box; tools/profile_bench.sh copies it to profiles/<tag>_fp16_margins.json)
    so that the margins over the gate are a committed record, not a print swallowed by ``pytest -q``."""
    import json
    out = os.path.join(os.environ.get('GRAFT_REPO_ROOT', ROOT), 'gpurun_out')
    os.makedirs(out, exist_ok=True)
    path = os.path.join(out, 'fp16_margins.json')
    try:
        with open(path) as f:
            rows = json.load(f)
    except (OSError, ValueError):
        rows = []
    row = {'test': test, 'frame': frame, 'psnr_vs_fp32_oracle_db': round(float(psnr_vs_oracle), 3), 'gate_db': gate_db,
           'margin_db': round(float(psnr_vs_oracle) - gate_db, 3), 'dpsnr_vs_pseudo_gt_db': round(float(dpsnr_vs_gt), 5),
           'dpsnr_gate_db': 5e-3, 'weights': 'synthetic_state_dict(0) (xavier-scaled random init: no checkpoint offline)'}
    row.update(extra)
    rows = [r for r in rows if (r['test'], r['frame']) != (test, frame)] + [row]
    with open(path, 'w') as f:
        json.dump(rows, f, indent=1)
