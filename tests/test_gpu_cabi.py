"""The C ABI driven by a NON-Python host: tests/c/forward_golden.c (plain C99, gcc) creates a context, loads the
state_dict, binds a hipMalloc'ed workspace and runs a full 64x96 forward; this wrapper only writes its two input
files (synthetic weights, one frozen reference case) and runs the binary."""
import os
import struct
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, 'tests', 'c', 'forward_golden')


def _write_inputs(tmp, golden_dir, name):
    from demfi_amd import synthetic_state_dict, synthetic_window
    sd = synthetic_state_dict(0)
    wpath, cpath = os.path.join(tmp, 'weights.bin'), os.path.join(tmp, 'case.bin')
    with open(wpath, 'wb') as f:
        f.write(struct.pack('<i', len(sd)))
        for k, v in sd.items():
            a = np.ascontiguousarray(v.numpy(), np.float32)
            kb = k.encode()
            f.write(struct.pack('<i', len(kb)) + kb + struct.pack('<i', a.ndim) + struct.pack('<%dq' % a.ndim, *a.shape))
            f.write(a.tobytes())
    g = np.load(os.path.join(golden_dir, name + '.npz'))
    H, W, N = int(g['H']), int(g['W']), int(g['N'])
    x = synthetic_window(H, W, int(g['seed']))[0].numpy()                    # [3,4,H,W]
    with open(cpath, 'wb') as f:
        f.write(struct.pack('<iiif', H, W, N, float(g['t'])))
        f.write(np.ascontiguousarray(x, np.float32).tobytes())
        f.write(np.ascontiguousarray(g['finals'][N - 1, 2], np.float32).tobytes())
        f.write(np.ascontiguousarray(g['flows'][N], np.float32).tobytes())
    return wpath, cpath


@pytest.mark.parametrize('name', ['e2e_64x96_t0500_n3', 'e2e_32x64_t0375_n5'])
def test_forward_from_plain_c(tmp_path, golden_dir, name):
    subprocess.check_call(['bash', os.path.join(ROOT, 'tests', 'c', 'build.sh')])      # < 1 s; never run a stale binary
    w, c = _write_inputs(str(tmp_path), golden_dir, name)
    r = subprocess.run([BIN, w, c], capture_output=True, text=True, timeout=300)
    print(r.stdout, r.stderr)
    assert r.returncode == 0, r.stderr
    assert 'C-ABI forward OK' in r.stdout
