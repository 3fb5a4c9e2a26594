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

    NPART = 8          # partial sums per "dot-producing pass" (the product uses 512 workgroups)

    def __init__(self, xvec, yvec, zvec):
        self.xv, self.yv, self.zv = (np.asarray(v, dtype=np.float64) for v in (xvec, yvec, zvec))
        self.shape = (len(self.xv), len(self.yv), len(self.zv))
        self.ncells = int(np.prod(self.shape))
        self.M = np.zeros(self.shape)
        self._bound = None

    def set_values(self, M_t):
        self.M = M_t.detach().cpu().numpy().reshape(self.shape).copy()

    def set_log_model(self, m_t, scale):
        self.M = scale * np.exp(m_t.detach().cpu().numpy().reshape(self.shape))

    def _values(self):
        if self._bound is not None:
            return self._bound[:self.ncells].numpy().reshape(self.shape)
        return self.M

    def forward(self, o, d, tmax, Ns, out=None, order=None):
        tec = OC.forward_tec_straight(self.xv, self.yv, self.zv, self._values(), o.numpy(), d.numpy(), tmax, Ns, 2)
        return torch.from_numpy(tec.reshape(-1))

    # -- the fused solver passes of RayEngine, restated with torch (same interface, same semantics) -------------
    def new_grid_buffer(self):
        padded = torch.zeros(self.ncells + self.shape[1] * self.shape[2] + self.shape[2] + 2, dtype=torch.float64)
        return padded, padded[:self.ncells].view(self.shape)

    def bind_values(self, padded):
        self._bound = padded

    def values_changed(self):
        pass

    def _partials(self, v):
        """sum(v) split over NPART chunks, like one partial per workgroup"""
        out = torch.zeros(self.NPART, dtype=torch.float64)
        for b, chunk in enumerate(torch.chunk(v.reshape(-1), self.NPART)):
            out[b] = chunk.sum()
        return out

    @staticmethod
    def _val(t):
        return 1.0 if t is None else t.sum()

    def adjoint_differential(self, o, d, v, scale, Na, i0, tmax, Ns, out=None, accum=None, order=None):
        y = (v if scale is None else v * scale).view(Na, -1)
        w = torch.from_numpy(O.differential_weights(y.numpy(), i0)).reshape(-1)
        g = self.adjoint(o, d, w, tmax, Ns)
        if out is None:
            return g
        out += g
        return out

    def rays_combine(self, tec, Na, i0, a, b, dobs=None, s1=None, s2=None, out=None, want_dot=True):
        t2 = tec.view(Na, -1)
        v = a * (t2 - t2[i0:i0 + 1]).reshape(-1)
        if dobs is not None:
            v = v + b * dobs
        if s1 is not None:
            v = v * s1
        return v, (self._partials(v * v if s2 is None else v * v * s2) if want_dot else None)

    def axpby_dot_(self, y, x, an=None, ad=None, a_sign=1.0, bn=None, bd=None, want_dot=True):
        alpha = a_sign * self._val(an) / self._val(ad)
        beta = self._val(bn) / self._val(bd)
        y.copy_(alpha * x + beta * y)
        return self._partials(y * y) if want_dot else None

    def compact_gather(self, full, idx, out=None, zero=True, want_dot=True):
        flat = full.view(-1)
        v = flat.index_select(0, idx.long())
        if zero:
            flat.index_fill_(0, idx.long(), 0.0)
        if out is not None:
            out.copy_(v)
            v = out
        return v, (self._partials(v * v) if want_dot else None)

    def compact_scatter(self, full, idx, src):
        full.view(-1).index_copy_(0, idx.long(), src)

    def compact_cg_update(self, x, p, s, idx, full_p, an, ad, bn, bd):
        alpha, beta = self._val(an) / self._val(ad), self._val(bn) / self._val(bd)
        x.add_(alpha * p)
        p.copy_(s + beta * p)
        full_p.view(-1).index_copy_(0, idx.long(), p)

    def compact_sirt_update(self, x, C, full_s, idx, full_x, relax=1.0, nonneg=False, want_max=False):
        flat = full_s.view(-1)
        s = flat.index_select(0, idx.long())
        flat.index_fill_(0, idx.long(), 0.0)
        xn = x + relax * C * s
        if nonneg:
            xn = xn.clamp(min=0)
        step = (xn - x).abs()
        x.copy_(xn)
        full_x.view(-1).index_copy_(0, idx.long(), xn)
        return step.max().reshape(1) if want_max else None

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
