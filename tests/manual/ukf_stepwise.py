"""Round 2, UKF chol(P) investigation: replay ONE callback per launch with a variant library and dump (N, X, P, status) after
every callback, so that two variants can be compared callback by callback (which callback diverges first, and in which entries).

    python tests/manual/ukf_stepwise.py <variant|base> <out.npz> [L] [T]      (run from the repository root, on the GPU box)
    python tests/manual/ukf_stepwise.py --compare a.npz b.npz
"""
import sys

import numpy as np

sys.path.insert(0, '.')
sys.path.insert(0, 'tests')


def dump(k, out, L, T):
    import torch
    import awesomeslam_amd.core as ac
    if k != 'base':
        ac._CORE = ac._CORE.replace('libaslam_core.so', 'libaslam_core_v%s.so' % k)
    import awesomeslam_amd.trace as tg
    from awesomeslam_amd.core import Core
    tr = tg.make_traces(L, T, B=2, seed=21)
    core = Core('ukf', tg.dim_cap(L), batch=2, max_obs=tr.max_obs, max_wait=512)
    core.set_trace(tr)
    NP = 16 * ((tg.dim_cap(L) - 1 + 15) // 16)
    Xs, Ps, Ns, St = np.zeros((T, NP)), np.zeros((T, NP, NP)), np.zeros(T, int), np.zeros(T, int)
    p = torch.zeros((2, 1, 3), dtype=torch.float64, device='cuda')
    for t in range(T):
        core.replay(t, 1, p.data_ptr(), None)
        torch.cuda.synchronize()
        X, Z, P = core.state(0)
        n = len(X)
        Ns[t], St[t] = n, core.status(0)
        Xs[t, :n], Ps[t, :n, :n] = X, P
    np.savez(out, X=Xs, P=Ps, N=Ns, status=St)
    print('variant', k, 'dumped', T, 'callbacks, final N', Ns[-1], 'status', St[-1])


def dump_poses(k, out, L, T, chunk):
    """replay in launches of `chunk` callbacks; record the pose and the dimension after every callback (poses_out / dims_out)"""
    import torch
    import awesomeslam_amd.core as ac
    if k != 'base':
        ac._CORE = ac._CORE.replace('libaslam_core.so', 'libaslam_core_v%s.so' % k)
    import awesomeslam_amd.trace as tg
    from awesomeslam_amd.core import Core
    tr = tg.make_traces(L, T, B=2, seed=21)
    core = Core('ukf', tg.dim_cap(L), batch=2, max_obs=tr.max_obs, max_wait=512)
    core.set_trace(tr)
    poses = np.zeros((T, 3))
    dims = np.zeros(T, int)
    if isinstance(chunk, int):
        chunk = [chunk] * ((T + chunk - 1) // chunk)
    t0 = 0
    for c in chunk:
        c = min(c, T - t0)
        if c <= 0:
            break
        p = torch.zeros((2, c, 3), dtype=torch.float64, device='cuda')
        dm = torch.zeros((2, c), dtype=torch.int32, device='cuda')
        core.replay(t0, c, p.data_ptr(), dm.data_ptr())
        torch.cuda.synchronize()
        poses[t0:t0 + c] = p[0].cpu().numpy()
        dims[t0:t0 + c] = dm[0].cpu().numpy()
        t0 += c
    X, Z, P = core.state(0)
    np.savez(out, poses=poses, dims=dims, X=X, P=P)
    print('variant', k, 'chunk', chunk, 'final N', dims[-1])


def compare_poses(a, b):
    A, B = np.load(a), np.load(b)
    T = len(A['dims'])
    e = np.abs(A['poses'] - B['poses']).max(axis=1) / np.abs(B['poses']).max()
    bad = np.nonzero(e > 1e-12)[0]
    print('pose streams: first callback above 1e-12:', (int(bad[0]), int(A['dims'][bad[0]])) if len(bad) else None, ' max', e.max(),
          ' final P rel diff', np.abs(A['P'] - B['P']).max() / np.abs(B['P']).max())
    grow = [int(t) for t in range(1, T) if A['dims'][t] != A['dims'][t - 1]]
    print('dimension changes at callbacks', grow, '->', [int(A['dims'][t]) for t in grow])
    for t in list(range(0, T, 10)):
        print(f'  t={t:4d} N={int(A["dims"][t]):3d} pose rel diff {e[t]:.2e}')


def compare(a, b):
    A, B = np.load(a), np.load(b)
    T = len(A['N'])
    first = None
    for t in range(T):
        n = int(A['N'][t])
        assert n == int(B['N'][t])
        Pa, Pb = A['P'][t, :n, :n], B['P'][t, :n, :n]
        Xa, Xb = A['X'][t, :n], B['X'][t, :n]
        ep = np.abs(Pa - Pb).max() / max(np.abs(Pb).max(), 1e-300)
        ex = np.abs(Xa - Xb).max() / max(np.abs(Xb).max(), 1e-300)
        if first is None and max(ep, ex) > 1e-11:
            first = t
            D = np.abs(Pa - Pb) / np.abs(Pb).max()
            i, j = np.unravel_index(np.argmax(D), D.shape)
            print(f'first divergence at callback {t}: N={n} rel err P {ep:.2e} X {ex:.2e}; worst P entry ({i},{j}); status {A["status"][t]} / {B["status"][t]}')
            np.set_printoptions(linewidth=220, precision=1)
            print('|dP| / max|P| (rows/cols 0..n-1):')
            print(D)
        if t in (0, 1, 2, 5, 10, 20, 50, 100, T - 1):
            print(f't={t:4d} N={n:3d} rel err P {ep:.2e} X {ex:.2e}')
    if first is None:
        print('no divergence above 1e-11')


if __name__ == '__main__':
    if sys.argv[1] == '--compare':
        compare(sys.argv[2], sys.argv[3])
    elif sys.argv[1] == '--compare-poses':
        compare_poses(sys.argv[2], sys.argv[3])
    elif sys.argv[1] == '--poses':
        dump_poses(sys.argv[2], sys.argv[3], 5, 150, [int(x) for x in sys.argv[4].split(',')] if ',' in sys.argv[4] else int(sys.argv[4]))
    else:
        dump(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 5, int(sys.argv[4]) if len(sys.argv) > 4 else 150)
