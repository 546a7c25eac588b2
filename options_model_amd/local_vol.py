"""Local-volatility paths through an implied-vol network:
drop-in for `IVModel` (options_model_3/options_model_3.py:263-298) and
`simulate_local_vol_paths_antithetic` (:300-333) on the MI355X.

Per time step the reference calls the IV network on ALL paths (2 -> 64 -> 4 x [Linear 64x64,
LayerNorm, GELU, residual] -> 1; NN_training_stock_iv.py:109-155) through numpy <-> torch-CPU
round trips.  Here the whole simulation is ONE kernel of the library (`omc_localvol_paths_f32`,
csrc/omc_mlp.hip): a wave carries 32 columns through all time steps and evaluates the network on
the matrix cores in float32, weights in LDS, activations in registers; normals come from the
library's Philox generator (`omc_gbm_normals_f32`, the GBM kernel's counter layout).  Networks of
another width go through PyTorch-ROCm (batched GEMMs per step; backend="torch" forces that
path, which the tests use as a cross-check).  The resulting [step][path] float32 matrix feeds the
same HIP backward-induction kernels as every other model (`omc_lsm_poly` on the tensor's device
pointer).
"""
from __future__ import annotations

import math

from . import _ffi


_warned = set()


def _torch():
    import torch
    if not torch.cuda.is_available():
        raise _ffi.OmcError("local-vol paths need PyTorch-ROCm with a visible GPU")
    return torch


def make_iv_network(hidden_dim=64, num_hidden_layers=4, epsilon=1e-4, dropout=0.1):
    """Architecture of the reference's ImprovedIVNetwork (NN_training_stock_iv.py:109-155) with the
    same parameter names, so a `state_dict` saved by the reference's trainer (:529-538) loads as is:
    input_proj Linear(2,H)+GELU, L x [h += Dropout(GELU(LayerNorm(Linear(h))))], output Linear(H,1),
    clamped at epsilon.  Attach a `scaler` (m_scale, tau_scale) before wrapping it in IVModel."""
    torch = _torch()
    nn, F = torch.nn, torch.nn.functional

    class IVNetwork(nn.Module):
        def __init__(self):
            super().__init__()
            self.epsilon = float(epsilon)
            self.scaler = None
            self.input_proj = nn.Linear(2, hidden_dim)
            self.layers = nn.ModuleList(
                nn.Sequential(nn.Linear(hidden_dim, hidden_dim), nn.LayerNorm(hidden_dim), nn.GELU(),
                              nn.Dropout(dropout) if dropout > 0 else nn.Identity())
                for _ in range(num_hidden_layers))
            self.output = nn.Linear(hidden_dim, 1)

        def forward(self, x):
            h = F.gelu(self.input_proj(x))
            for layer in self.layers:
                h = h + layer(h)
            return self.output(h).clamp(min=self.epsilon)

    return IVNetwork()


class IVModel:
    """Same contract as the reference wrapper: takes the trained network object, which must
    carry a fitted `scaler` with `m_scale` / `tau_scale`; evaluates sigma(K, S, tau)."""

    def __init__(self, nn_model, device: int = 0):
        torch = _torch()
        if not hasattr(nn_model, "scaler") or nn_model.scaler is None:
            raise ValueError("Model does not have a fitted scaler")
        self.device = torch.device("cuda", device)
        self.model = nn_model.eval().to(self.device)
        self.m_scale = float(nn_model.scaler.m_scale)
        self.tau_scale = float(nn_model.scaler.tau_scale)

    def sigma_tensor(self, K: float, S, tau: float):
        """S: float32 device tensor [n] -> sigma float32 [n] (clamped at 1e-6 like :293)."""
        torch = _torch()
        tau = max(float(tau), 1e-6)
        m = torch.log(max(K, 1e-8) / torch.clamp(S.double(), min=1e-8))
        X = torch.stack([m / self.m_scale, torch.full_like(m, tau / self.tau_scale)], dim=1).float()
        with torch.no_grad():
            return self.model(X).squeeze(1).clamp_min(1e-6)

    def get_volatility_batch(self, K: float, S_batch, tau: float):
        """numpy in / numpy float64 out, with the reference's input checks (:276-283)."""
        import numpy as np
        torch = _torch()
        S_batch = np.asarray(S_batch, dtype=np.float64)
        if K <= 0:
            raise ValueError(f"K must be positive, got {K}")
        if np.any(S_batch <= 0):
            raise ValueError("All S_batch values must be positive")
        s = torch.from_numpy(S_batch).to(self.device)
        return self.sigma_tensor(K, s, tau).double().cpu().numpy()


def _flatten_iv_network(net):
    """-> (flat float32 parameters in omc_localvol_paths_f32's layout, hidden layers) or None if the
    network is not the ImprovedIVNetwork shape with hidden_dim 64."""
    torch = _torch()
    try:
        lin_in, layers, out = net.input_proj, list(net.layers), net.output
        if lin_in.weight.shape != (64, 2) or out.weight.shape != (1, 64) or not 1 <= len(layers) <= 8:
            return None
        parts = [torch.cat([lin_in.weight, lin_in.bias[:, None], torch.zeros_like(lin_in.bias[:, None])],
                           dim=1).reshape(-1)]
        for blk in layers:
            lin, ln, act = blk[0], blk[1], blk[2]
            if (lin.weight.shape != (64, 64) or not isinstance(ln, torch.nn.LayerNorm) or abs(ln.eps - 1e-5) > 0
                    or not isinstance(act, torch.nn.GELU) or getattr(act, "approximate", "none") != "none"):
                return None
            parts += [lin.weight.reshape(-1), lin.bias, ln.weight, ln.bias]
        parts += [out.weight.reshape(-1), out.bias]
        with torch.no_grad():
            return torch.cat([p.detach() for p in parts]).float().contiguous(), len(layers)
    except (AttributeError, IndexError, TypeError):
        return None


def simulate_local_vol_paths(S0, r, T, num_simulations, num_time_steps, iv_model: IVModel, K, seed,
                             stream=0, z_half=None, backend="auto"):
    """-> float32 device tensor S [num_time_steps+1, M], M = num_simulations // 2 * 2, antithetic
    partner of column j is j + M/2 (:306-307).  z_half (optional, [N][M/2]) injects normals.
    backend: "hip" = the library's kernel (network evaluated inside the path loop; hidden_dim 64),
    "torch" = per-step batched GEMMs through PyTorch-ROCm, "auto" = hip where the shape allows."""
    torch = _torch()
    if backend not in ("auto", "hip", "torch"):
        raise ValueError("backend must be 'auto', 'hip' or 'torch'")
    dev = iv_model.device
    N = int(num_time_steps)
    M = int(num_simulations) // 2 * 2
    P = M // 2
    dt = T / N
    with torch.cuda.device(dev):
        if z_half is None:
            Z = torch.empty((N, P), dtype=torch.float32, device=dev)
            torch.cuda.synchronize(dev)
            ctx = _ffi.default_context(dev.index or 0)
            _ffi._check(ctx.lib, ctx.lib.omc_gbm_normals_f32(ctx.handle, Z.data_ptr(), P, P, N, int(seed),
                                                             int(stream), 0))
        else:
            Z = torch.as_tensor(z_half, dtype=torch.float32, device=dev)
            assert Z.shape == (N, P)
        S = torch.empty((N + 1, M), dtype=torch.float32, device=dev)
        flat = _flatten_iv_network(iv_model.model) if backend != "torch" else None
        if backend == "hip" and flat is None:
            raise ValueError("backend='hip' needs the ImprovedIVNetwork shape with hidden_dim 64")
        if flat is not None:
            params, layers = flat
            params = params.to(dev)
            ctx = _ffi.default_context(dev.index or 0)
            torch.cuda.synchronize(dev)
            _ffi._check(ctx.lib, ctx.lib.omc_localvol_paths_f32(
                ctx.handle, S.data_ptr(), M, M, N, float(S0), float(r), float(T), float(K), 64, layers,
                params.data_ptr(), iv_model.m_scale, iv_model.tau_scale, float(iv_model.model.epsilon),
                Z.contiguous().data_ptr()))
            return S
        if backend == "auto" and "localvol" not in _warned:
            _warned.add("localvol")
            import warnings
            warnings.warn("options_model_amd: this implied-vol network is not the shape localvol_paths_kernel covers "
                          "(ImprovedIVNetwork with hidden_dim 64, LayerNorm eps 1e-5, exact GELU): simulating through "
                          "PyTorch-ROCm, one batched network evaluation per time step (several times slower)",
                          RuntimeWarning, stacklevel=2)
        S[0] = S0
        sq = math.sqrt(dt)
        for t in range(1, N + 1):
            prev = S[t - 1]
            tau_t = max(T - (t - 1) * dt, 1e-6)
            sig = iv_model.sigma_tensor(K, prev, tau_t)
            z = torch.cat([Z[t - 1], -Z[t - 1]])
            S[t] = prev * torch.exp((r - 0.5 * sig * sig) * dt + sig * sq * z)
    return S


def price_american_on_paths(ctx, S, K, r, T, is_put, semantics="two_pass"):
    """Backward induction by the HIP kernels directly on the torch tensor's memory."""
    torch = _torch()
    torch.cuda.synchronize(S.device)
    sem = {"two_pass": "two_pass", "per_step": "reference", "textbook": "textbook"}[semantics]
    return ctx.lsm_poly(S, K, r, T, is_put, sem)
