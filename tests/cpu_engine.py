"""CPU stand-in for ionotomo_amd.engine.RayEngine, built on the ORACLE (tests only).

Lets the sharding / all-reduce / solver logic (ionotomo_amd/parallel.py, solvers.py) run under
gloo on CPU with world_size > 1, where no HIP kernel can execute.  It is never used by the
product: RayEngine raises without a GPU."""
import numpy as np
import torch

from oracle import oracle as O
from oracle import oracle_c as OC


class OracleEngine(object):
    device = torch.device("cpu")

    def __init__(self, xvec, yvec, zvec):
        self.xv, self.yv, self.zv = (np.asarray(v, dtype=np.float64) for v in (xvec, yvec, zvec))
        self.shape = (len(self.xv), len(self.yv), len(self.zv))
        self.M = np.zeros(self.shape)

    def set_values(self, M_t):
        self.M = M_t.detach().cpu().numpy().reshape(self.shape).copy()

    def set_log_model(self, m_t, scale):
        self.M = scale * np.exp(m_t.detach().cpu().numpy().reshape(self.shape))

    def forward(self, o, d, tmax, Ns, out=None, order=None):
        tec = OC.forward_tec_straight(self.xv, self.yv, self.zv, self.M, o.numpy(), d.numpy(), tmax, Ns, 2)
        return torch.from_numpy(tec.reshape(-1))

    def adjoint(self, o, d, w, tmax, Ns, out=None, accum=None, order=None):
        g = OC.adjoint_straight(self.xv, self.yv, self.zv, o.numpy(), d.numpy(), w.numpy(), tmax, Ns)
        return torch.from_numpy(g)

    def adjoint_residual(self, o, d, tec, dobs, cdct, Na, i0, tmax, Ns, out=None, accum=None, order=None):
        t2 = tec.view(Na, -1)
        dd = ((t2 - t2[i0:i0 + 1]).reshape(-1) - dobs) / (cdct + 1e-15)
        w = torch.from_numpy(O.differential_weights(dd.view(Na, -1).numpy(), i0)).reshape(-1)
        return self.adjoint(o, d, w, tmax, Ns)

    def smooth(self, v, kx, ky, kz, out=None, work=None):
        from scipy.ndimage import convolve1d
        a = v.numpy().reshape(self.shape)
        for ax, k in enumerate((kx, ky, kz)):
            a = convolve1d(a, np.asarray(k), axis=ax, mode='nearest')
        return torch.from_numpy(a.copy())

    def axpby_(self, y, x, a_num=None, a_den=None, a_sign=1.0, b_num=None, b_den=None):
        one = torch.ones((), dtype=torch.float64)
        a = a_sign * (one if a_num is None else a_num) / (one if a_den is None else a_den)
        b = (one if b_num is None else b_num) / (one if b_den is None else b_den)
        y.mul_(b).add_(a * x)
        return y
