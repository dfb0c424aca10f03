"""Drop-in for `sample.edm_sampler` (sample.py:30-66): EDM 2nd-order Heun sampler, Karras
rho=7 schedule, fp64 state, fp32 network I/O, optional classifier-free guidance.

For the shipped setting (S_churn = 0 => gamma = 0, x_hat = x_cur, t_hat = t_cur) with a
maskdit_amd EDMPrecond, one whole Heun step -- input scaling, the CFG-doubled network
evaluation, the EDM output blend, Euler update, second evaluation, Heun average, step-counter
bump -- is captured ONCE into a hipGraph on a side stream; every step replays that graph,
reading its scalars (t_i, t_{i+1}) from a device-side fp64 schedule indexed by a device-side
counter.  The last step (no 2nd-order correction, sample.py:61) replays a second, shorter graph.
`use_graph=False` issues the SAME launch sequence directly on the stream (no capture): same kernels, same bits.
Stochastic churn (S_churn > 0; sample.py:51-53; no shipped configuration) runs every network evaluation on the HIP
forward plan with the step's fp64 state algebra in torch (`_churn_steps`), checked against the reference fixture
`s2_sampler_churn.npz`.
"""
from __future__ import annotations

import ctypes as C
import weakref
from typing import Dict, Tuple

import torch

from . import _lib
from ._lib import call
from .engine import plan_key
from .loss import unwrap_model
from .precond import EDMPrecond


def edm_t_steps(num_steps, sigma_min, sigma_max, rho, device) -> torch.Tensor:
    """sample.py:40-43: fp64 schedule with t_N = 0 appended."""
    i = torch.arange(num_steps, dtype=torch.float64, device=device)
    t = (sigma_max ** (1 / rho) + i / (num_steps - 1) * (sigma_min ** (1 / rho) - sigma_max ** (1 / rho))) ** rho
    return torch.cat([t, torch.zeros_like(t[:1])])


class _GraphedHeun:
    """Captured graphs + persistent state buffers for one (net, batch, cfg?) shape."""

    def __init__(self, net: EDMPrecond, B: int, use_cfg: bool, precision: str = 'bf16', max_steps: int = 1024):
        # neither the network nor its engine is OWNED by the graph cache (a module-level dict): a cached graph must not
        # keep a deleted model's arenas and plans alive
        self.B, self.use_cfg = B, use_cfg
        self.sigma_data = float(net.sigma_data)
        sp = net.spec
        dev = next(net.parameters()).device
        self.chw = sp.C * sp.R * sp.R
        self.dup = 2 if use_cfg else 1
        self._eng_ref = weakref.ref(net.engine())
        self.precision = precision
        self.pl = self.eng.plan(B * self.dup, False, False, None, precision)
        f64 = dict(device=dev, dtype=torch.float64)
        self.x_hat = torch.zeros(B, self.chw, **f64)
        self.x_next = torch.zeros(B, self.chw, **f64)
        self.d_cur = torch.zeros(B, self.chw, **f64)
        self.t_steps = torch.zeros(max_steps + 1, **f64)
        self.step_idx = torch.zeros(1, device=dev, dtype=torch.int32)
        self.sig = torch.zeros(B * self.dup, device=dev, dtype=torch.float32)
        self.cfg_scale = torch.zeros(1)  # host copy of the captured value
        self.stream = torch.cuda.Stream(device=dev)
        self.graph_full = self.graph_last = None
        self.captured_cfg = None

    @property
    def eng(self):
        return self._eng_ref()

    def _eval(self, st, src, which):
        """network evaluation at t_{i+which} of fp64 state `src` -> plan buffer F"""
        pl, sd = self.pl, self.sigma_data
        call('mdt_sampler_prep', src.data_ptr(), self.t_steps.data_ptr(), self.step_idx.data_ptr(), which,
             pl.buf['xin'].data_ptr(), self.sig.data_ptr(), self.B, self.chw, self.dup, sd, st)
        call('mdt_precond_coef', self.sig.data_ptr(), pl.buf['coef'].data_ptr(), self.B * self.dup, sd, st)
        pl.fwd.run(st)

    def _record(self, st, cfg_scale, last):
        sd = self.sigma_data
        Fp = self.pl.buf['F'].data_ptr()
        self._eval(st, self.x_hat, 0)
        call('mdt_sampler_euler', self.x_hat.data_ptr(), Fp, self.t_steps.data_ptr(), self.step_idx.data_ptr(), cfg_scale,
             int(self.use_cfg), self.x_next.data_ptr(), self.d_cur.data_ptr(), self.B, self.chw, sd, st)
        if not last:
            self._eval(st, self.x_next, 1)
            call('mdt_sampler_heun', self.x_hat.data_ptr(), self.x_next.data_ptr(), Fp, self.d_cur.data_ptr(),
                 self.t_steps.data_ptr(), self.step_idx.data_ptr(), cfg_scale, int(self.use_cfg), self.B, self.chw, sd, st)
        call('mdt_sampler_advance', self.step_idx.data_ptr(), st)

    def capture(self, cfg_scale: float):
        L = _lib.lib()
        self.destroy()
        if self.eng.shadows_dirty and self.precision != 'fp32':
            self.eng.refresh_shadows()
        torch.cuda.synchronize()
        graphs = []
        with torch.cuda.stream(self.stream):
            st = self.stream.cuda_stream
            for last in (False, True):
                _lib.check(L.mdt_graph_begin(st), 'mdt_graph_begin')
                try:
                    self._record(st, cfg_scale, last)
                finally:
                    g = C.c_void_p()
                    rc = L.mdt_graph_end(st, C.byref(g))
                _lib.check(rc, 'mdt_graph_end')
                graphs.append(g)
        self.graph_full, self.graph_last = graphs
        self.captured_cfg = cfg_scale

    def destroy(self):
        L = _lib.lib()
        for g in (self.graph_full, self.graph_last):
            if g is not None:
                L.mdt_graph_destroy(g)
        self.graph_full = self.graph_last = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


_CACHE: Dict[Tuple[int, int, bool, str], _GraphedHeun] = {}


def release_graphs():
    """Destroy every cached sampler graph (and with it the references to the inference plans they replay)."""
    while _CACHE:
        _CACHE.popitem()[1].destroy()


def _graphed(net: EDMPrecond, B: int, use_cfg: bool, precision: str = 'bf16') -> _GraphedHeun:
    for k in [k for k, v in _CACHE.items() if v.eng is None]:  # graphs of models that no longer exist
        _CACHE.pop(k).destroy()
    key = (id(net.engine()), B, use_cfg, precision)
    g = _CACHE.get(key)
    if g is None or g.eng is not net.engine() or g.pl is not net.engine()._plans.get(plan_key(B * g.dup, False, False, None, precision)):
        if len(_CACHE) >= 4:
            _CACHE.pop(next(iter(_CACHE))).destroy()
        g = _GraphedHeun(net, B, use_cfg, precision)
        _CACHE[key] = g

        def dropped(key=key, ref=weakref.ref(g)):  # the plan cache evicted the plan this graph replays
            if _CACHE.get(key) is ref() and ref() is not None:
                _CACHE.pop(key).destroy()
        g.pl.evict_hooks.append(dropped)
    return g


def _churn_steps(net, x, t_steps, class_labels, cfg_scale, randn_like, num_steps, S_churn, S_min, S_max, S_noise):
    """The S_churn > 0 branch of sample.py:46-64.  Not graph-captured: gamma depends on the host-side schedule."""
    gamma_max = min(S_churn / num_steps, 2.0 ** 0.5 - 1.0)
    ts = [float(v) for v in t_steps.tolist()]

    def denoise(state, sigma):
        sig = torch.tensor(sigma, dtype=torch.float64, device=state.device)
        return net(state.float(), sig, class_labels, cfg_scale)['x'].to(torch.float64)

    for i in range(num_steps):
        t_cur, t_next = ts[i], ts[i + 1]
        t_hat = t_cur * (1.0 + (gamma_max if S_min <= t_cur <= S_max else 0.0))   # round_sigma = identity (maskdit.py:775)
        x_hat = x + ((t_hat * t_hat - t_cur * t_cur) ** 0.5 * S_noise) * randn_like(x)
        slope = (x_hat - denoise(x_hat, t_hat)) / t_hat
        x = x_hat + (t_next - t_hat) * slope                                       # Euler
        if i + 1 < num_steps:                                                      # Heun average (not on the last step)
            slope2 = (x - denoise(x, t_next)) / t_next
            x = x_hat + (t_next - t_hat) * 0.5 * (slope + slope2)
    return x


@torch.no_grad()
def edm_sampler(net, latents, class_labels=None, cfg_scale=None, feat=None, randn_like=torch.randn_like, num_steps=18,
                sigma_min=0.002, sigma_max=80, rho=7, S_churn=0, S_min=0, S_max=float('inf'), S_noise=1, use_graph=True,
                precision=None):
    """Same signature and result (fp64 [N, C, H, W]) as sample.py:30-66.  `precision` (an addition): arithmetic of the
    network evaluations -- 'bf16' (bf16 MFMA operands, fp32 residual stream: the training kernels; 18-19 samples/s for XL/2,
    drift 7e-4 of the latent range against the reference's fp32 network after 50 steps) or 'fp32' (exact fp32 weights,
    activations and matrix instructions: what sample.py:56 itself runs, agreeing with the reference fixture to fp32
    rounding); None = the network's `eval_precision` (default 'bf16')."""
    raw = unwrap_model(net)
    if not isinstance(raw, EDMPrecond):
        raise TypeError(f'maskdit_amd.edm_sampler expects a maskdit_amd EDMPrecond, got {type(raw).__name__}')
    if feat is not None:
        raise NotImplementedError('feat conditioning is outside the shipped configurations')
    if not latents.is_cuda:
        raise _lib.MaskDiTLibError('maskdit_amd: latents are not on a HIP device; there is no CPU path')
    if raw.training:
        raise RuntimeError('edm_sampler needs net.eval() (generate.py:41)')
    sigma_min = max(sigma_min, raw.sigma_min)
    sigma_max = min(sigma_max, raw.sigma_max)
    t_steps = edm_t_steps(num_steps, sigma_min, sigma_max, rho, latents.device)
    B = latents.shape[0]
    labels = raw._labels(class_labels, B, latents.device)
    x_next = latents.to(torch.float64) * t_steps[0]  # sample.py:46
    precision = raw.eval_precision if precision is None else precision
    if precision not in ('bf16', 'fp32'):
        raise ValueError(f"precision must be 'bf16' or 'fp32', got {precision!r}")
    if S_churn != 0:
        # stochastic churn (sample.py:51-53; no shipped config uses it): every network evaluation is the HIP forward
        # plan, the fp64 state algebra of the step -- which now carries a per-step noise level t_hat != t_i -- is torch
        prev, raw.eval_precision = raw.eval_precision, precision
        try:
            return _churn_steps(raw, x_next, t_steps, class_labels, cfg_scale, randn_like, num_steps, S_churn, S_min, S_max, S_noise)
        finally:
            raw.eval_precision = prev

    use_cfg = cfg_scale is not None
    g = _graphed(raw, B, use_cfg, precision)
    if num_steps + 1 > g.t_steps.numel():
        raise ValueError('num_steps exceeds the captured schedule capacity (1024)')
    s = float(cfg_scale) if use_cfg else 0.0
    # (an fp32 plan reads the fp32 master arena: stale bf16 shadows do not concern its captured graphs)
    stale = raw.engine().shadows_dirty and precision != 'fp32'
    if use_graph and (g.graph_full is None or g.captured_cfg != s or stale):
        g.capture(s)
    if not use_graph and stale:
        raw.engine().refresh_shadows()
    L = _lib.lib()
    cur = torch.cuda.current_stream()
    g.t_steps[:num_steps + 1].copy_(t_steps)
    g.step_idx.zero_()
    g.x_hat.copy_(x_next.reshape(B, -1))
    lab = g.pl.buf['labels']
    lab[:B].copy_(labels)
    if use_cfg:
        lab[B:].zero_()  # models/maskdit.py:566-567: y_null
    g.stream.wait_stream(cur)
    with torch.cuda.stream(g.stream):
        st = g.stream.cuda_stream
        for i in range(num_steps):
            # sample.py:53 draws randn_like(x_cur) even when gamma = 0 (its coefficient is then 0):
            # keep the caller's generator in the same state as the reference would leave it
            randn_like(g.x_hat.view_as(latents))
            last = i == num_steps - 1
            if use_graph:
                _lib.check(L.mdt_graph_launch(g.graph_last if last else g.graph_full, st), 'mdt_graph_launch')
            else:
                g._record(st, s, last)
            if not last:
                g.x_hat.copy_(g.x_next)  # x_cur <- x_next (sample.py:48); on the same stream, after the graph
        out = g.x_next.clone().view_as(x_next)
    cur.wait_stream(g.stream)
    return out
