"""Round 2, UKF investigation: after a chosen launch pattern, copy the UKF's HBM scratch (D, DZ, Tc, K) and the state (X, P) of
filter 0 to an .npz, for a variant library.  The scratch pointers are not part of the ABI: they are found by scanning the
context object for the UkfView {int MP; double *D, *DZ, *Tc, *K} pattern (diagnostic tool, fragile by design).

    python tests/manual/ukf_scratch_dump.py <variant|base> <out.npz> <chunk list, e.g. 8,2>      (repository root, GPU box)
    python tests/manual/ukf_scratch_dump.py --compare a.npz b.npz
"""
import ctypes
import re
import sys

import numpy as np

sys.path.insert(0, '.')
sys.path.insert(0, 'tests')


def hip_lib():
    for line in open('/proc/self/maps'):
        m = re.search(r'(/\S*libamdhip64\.so\S*)', line)
        if m:
            return ctypes.CDLL(m.group(1))
    raise RuntimeError('HIP runtime not loaded')


def dump(k, out, chunks, L=5):
    import torch
    import awesomeslam_amd.core as ac
    if k != 'base':
        ac._CORE = ac._CORE.replace('libaslam_core.so', 'libaslam_core_v%s.so' % k)
    import awesomeslam_amd.trace as tg
    from awesomeslam_amd.core import Core
    T = sum(chunks)
    tr = tg.make_traces(L, max(T, 150), B=2, seed=21)
    core = Core('ukf', tg.dim_cap(L), batch=2, max_obs=tr.max_obs, max_wait=512)
    core.set_trace(tr)
    t0 = 0
    for c in chunks:
        p = torch.zeros((2, c, 3), dtype=torch.float64, device='cuda')
        core.replay(t0, c, p.data_ptr(), None)
        torch.cuda.synchronize()
        t0 += c
    NP, _ = core.layout()
    MP = 2 * NP + 16
    raw = (ctypes.c_uint64 * 160).from_address(core._h.value)
    words = [int(x) for x in raw]
    hit = None
    for i in range(len(words) - 5):
        if (words[i] & 0xffffffff) == MP and all(w > (1 << 32) and w % 8 == 0 for w in words[i + 1:i + 5]) and len(set(words[i + 1:i + 5])) == 4:
            hit = i
            break
    if hit is None:
        raise SystemExit('UkfView not found in the context object')
    hip = hip_lib()
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]

    def grab(ptr, shape):
        a = np.empty(shape)
        rc = hip.hipMemcpy(a.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(ptr), a.nbytes, 2)
        assert rc == 0, rc
        return a

    D = grab(words[hit + 1], (NP, MP))
    DZ = grab(words[hit + 2], (NP, MP))
    Tc = grab(words[hit + 3], (NP, NP))
    K = grab(words[hit + 4], (NP, NP))
    X, Z, P = core.state(0)
    np.savez(out, D=D, DZ=DZ, Tc=Tc, K=K, X=X, Z=Z, P=P, n=len(X))
    print('variant', k, 'chunks', chunks, 'N', len(X), 'status', core.status(0))


def compare(a, b):
    A, B = np.load(a), np.load(b)
    n = int(A['n'])
    np.set_printoptions(linewidth=250, precision=1)
    for key in ('X', 'Z', 'P', 'D', 'DZ', 'Tc', 'K'):
        x, y = A[key], B[key]
        scale = max(np.abs(y).max(), 1e-300)
        e = np.abs(x - y) / scale
        print(f'{key:3s} shape {x.shape}: max rel diff {e.max():.2e}', end='')
        if e.max() > 1e-13:
            idx = np.argwhere(e > 1e-13)
            rows = sorted(set(int(i[0]) for i in idx))
            cols = sorted(set(int(i[-1]) for i in idx)) if x.ndim == 2 else []
            print(f'  differing rows {rows[:24]} cols {cols[:40]}', end='')
        print()
    print('n =', n)


if __name__ == '__main__':
    if sys.argv[1] == '--compare':
        compare(sys.argv[2], sys.argv[3])
    else:
        dump(sys.argv[1], sys.argv[2], [int(x) for x in sys.argv[3].split(',')])
