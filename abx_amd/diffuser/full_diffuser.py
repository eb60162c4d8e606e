"""Drop-in for the reference's `diffuser.full_diffuser.FullDiffuser` (diffuser/full_diffuser.py:28-290) on MI355X.

Hot-path methods run on libabx_hip kernels:
  reverse()            -> abx_reverse_step  (SO(3) geodesic step + R^3 VP-SDE step + token tau-leaping + mask merge, float64)
  IGSO(3) tables       -> abx_igso3_tables  (so3_diffuser.py:153-166), cached as .npy under `so3.cache_dir` like the reference
  rot/trans score      -> fused into the score network (abx_scores); calc_quat_score / calc_trans_score are kept for API parity
Schedule scalars (sigma index, score scalings) are evaluated on the device from `t` without host synchronisation.
Init-time sampling (sample_ref / forward_marginal, once per sample) is tensor plumbing on the caller's device.

Noise: `noise=` injects recorded draws (parity mode: z_rot, z_trans (B,L,3) f32 and jumps (B,L,20) f32, the reference's
draw order so3 randn -> r3 randn -> Poisson).  Without it the kernel draws Philox4x32-10 noise keyed by
(seed, sample id, step), which makes results independent of batching and of the rank a sample runs on.
"""
import os

import numpy as np
import torch

from abx_amd import ops

diffuser_obj_dict = {}


def _sin_half_over(angles, half):
    small = torch.abs(angles) < 1e-6
    safe = torch.where(small, torch.ones_like(angles), angles)
    return torch.where(small, 0.5 - angles * angles / 48, torch.sin(half) / safe)


def quat_to_rotvec(q):
    flip = (q[..., :1] < 0).to(q.dtype)
    q = (-1.0 * q) * flip + (1.0 - flip) * q
    norms = torch.norm(q[..., 1:], p=2, dim=-1, keepdim=True)
    half = torch.atan2(norms, q[..., :1])
    return q[..., 1:] / _sin_half_over(2 * half, half)


def rotvec_to_quat(v):
    angles = torch.norm(v, p=2, dim=-1, keepdim=True)
    half = angles * 0.5
    return torch.cat([torch.cos(half), v * _sin_half_over(angles, half)], dim=-1)


def quat_multiply(q1, q2):
    a1, b1, c1, d1 = q1.unbind(-1)
    a2, b2, c2, d2 = q2.unbind(-1)
    return torch.stack([a1 * a2 - b1 * b2 - c1 * c2 - d1 * d2, a1 * b2 + b1 * a2 + c1 * d2 - d1 * c2,
                        a1 * c2 - b1 * d2 + c1 * a2 + d1 * b2, a1 * d2 + b1 * c2 - c1 * b2 + d1 * a2], dim=-1)


def _interp_rows(x_new, x, y):
    """Batched linear interpolation y(x_new), rows of x ascending (reference abx/utils.py:31-59)."""
    b = torch.sum((x.unsqueeze(2) < x_new.unsqueeze(1)), dim=1)
    b = torch.clamp(b, 0, x.shape[1] - 2)
    xl, xh = torch.gather(x, -1, b), torch.gather(x, -1, b + 1)
    yl, yh = torch.gather(y, -1, b), torch.gather(y, -1, b + 1)
    w = (x_new - xl) / (xh - xl + 1e-8)
    w = torch.where(x_new > x[:, -1:], torch.ones_like(w), w)
    w = torch.where(x_new < x[:, :1], torch.zeros_like(w), w)
    return yl * (1 - w) + yh * w


class FullDiffuser:
    def __init__(self, diff_conf):
        self._diff_conf = diff_conf
        so3, r3, seq = diff_conf['so3'], diff_conf['r3'], diff_conf['seq']
        self.min_sigma, self.max_sigma = so3['min_sigma'], so3['max_sigma']
        self.num_sigma, self.num_omega = int(so3['num_sigma']), int(so3['num_omega'])
        self.cache_dir = so3['cache_dir']
        # fp32 0-dim constants of the reference, as exact python floats
        self.exp_max_sigma = float(torch.exp(torch.tensor(self.max_sigma)))
        self.exp_min_sigma = float(torch.exp(torch.tensor(self.min_sigma)))
        self.min_b_f32 = float(torch.tensor(r3['min_b']))
        self.bdiff_f32 = float(torch.tensor(r3['max_b'] - r3['min_b']))
        self.coord_scale_f32 = float(torch.tensor(r3['coordinate_scaling']))
        self.rate_const = float(torch.tensor(seq['rate_const'] if isinstance(seq, dict) else seq.rate_const))
        self.discrete_omega = torch.linspace(0, np.pi, self.num_omega + 1)[1:]
        lin = torch.linspace(0.0, 1.0, self.num_sigma)
        self.discrete_sigma = torch.log(lin * torch.exp(torch.tensor(self.max_sigma)) + (1 - lin) * torch.exp(torch.tensor(self.min_sigma)))
        self.device = None
        self.seed = 0
        self._step_counter = 0

    @staticmethod
    def get(diff_conf):
        if 'diffuser' not in diffuser_obj_dict:
            diffuser_obj_dict['diffuser'] = FullDiffuser(diff_conf)
        return diffuser_obj_dict['diffuser']

    # ---- tables ------------------------------------------------------------------------------------------------------
    def _cache_path(self):
        rp = lambda x: str(x).replace('.', '_')
        return os.path.join(self.cache_dir, f'eps_{self.num_sigma}_omega_{self.num_omega}_min_sigma_{rp(self.min_sigma)}'
                                            f'_max_sigma_{rp(self.max_sigma)}_schedule_logarithmic')

    def to(self, device):
        device = torch.device(device)
        if self.device == device:
            return self
        if device.type != 'cuda':
            raise RuntimeError('abx_amd FullDiffuser runs on an MI355X only (no CPU path)')
        d = self._cache_path()
        names = [os.path.join(d, n) for n in ('pdf_vals.npy', 'cdf_vals.npy', 'score_norms.npy')]
        self.discrete_sigma_dev = self.discrete_sigma.to(device).contiguous()
        self.discrete_omega_dev = self.discrete_omega.to(device).contiguous()
        if all(os.path.exists(n) for n in names):
            self._pdf, self._cdf, self.score_norms = (torch.from_numpy(np.load(n)).to(device).contiguous() for n in names)
        else:
            shape = (self.num_sigma, self.num_omega)
            self._pdf = torch.empty(shape, device=device)
            self._cdf = torch.empty(shape, device=device)
            self.score_norms = torch.empty(shape, device=device)
            ops.igso3_tables(self.discrete_sigma_dev, self.discrete_omega_dev, self._pdf, self._cdf, self.score_norms)
            try:
                os.makedirs(d, exist_ok=True)
                for n, t in zip(names, (self._pdf, self._cdf, self.score_norms)):
                    np.save(n, t.cpu().numpy())
            except OSError:
                pass
        self._score_scaling = torch.sqrt(torch.abs(torch.sum(self.score_norms ** 2 * self._pdf, dim=-1) /
                                                   torch.sum(self._pdf, dim=-1))) / float(np.sqrt(3))
        self.device = device
        return self

    def set_tables(self, pdf, cdf, score_norms, device):
        """Parity hook: use externally supplied IGSO(3) tables (same role as the reference's .npy cache)."""
        device = torch.device(device)
        self.discrete_sigma_dev = self.discrete_sigma.to(device).contiguous()
        self.discrete_omega_dev = self.discrete_omega.to(device).contiguous()
        self._pdf, self._cdf, self.score_norms = (torch.as_tensor(x).to(device).float().contiguous() for x in (pdf, cdf, score_norms))
        self._score_scaling = torch.sqrt(torch.abs(torch.sum(self.score_norms ** 2 * self._pdf, dim=-1) /
                                                   torch.sum(self._pdf, dim=-1))) / float(np.sqrt(3))
        self.device = device
        return self

    # ---- schedules (device, no host sync) ---------------------------------------------------------------------------------
    def _sigma(self, t):
        return torch.log(t * self.exp_max_sigma + (1 - t) * self.exp_min_sigma)

    def _sigma_idx(self, t):
        s = self._sigma(t)
        ds = self.discrete_sigma_dev
        return torch.clamp(torch.sum(ds[None, :] <= s[:, None] + 1e-5, -1) - 1, 0, self.num_sigma - 1)

    def _marginal_b_t(self, t):
        return t * self.min_b_f32 + (1 / 2) * (t ** 2) * self.bdiff_f32

    def score_scaling(self, t):
        self.to(t.device)
        rot = self._score_scaling[self._sigma_idx(t)]
        trans = 1 / torch.sqrt(1 - torch.exp(-self._marginal_b_t(t)))
        return rot, trans

    def calc_trans_score(self, trans_t, trans_0, t, scale=True):
        if scale:
            trans_t, trans_0 = trans_t * self.coord_scale_f32, trans_0 * self.coord_scale_f32
        t = t[:, None, None]
        mb = self._marginal_b_t(t)
        return -(trans_t - torch.exp(-1 / 2 * mb) * trans_0) / (1 - torch.exp(-mb))

    def calc_quat_score(self, quat_t, quat_0, t):
        self.to(quat_t.device)
        inv = torch.cat([quat_0[..., :1], -quat_0[..., 1:]], dim=-1) / torch.sqrt(torch.sum(quat_0 ** 2, dim=-1, keepdim=True))
        vec = quat_to_rotvec(quat_multiply(inv, quat_t))
        omega = torch.linalg.norm(vec, dim=-1) + 1e-6
        sn = self.score_norms[self._sigma_idx(t)]
        oi = torch.bucketize(omega, self.discrete_omega_dev[:-1])
        return torch.gather(sn, 1, oi)[..., None] * vec / (omega[..., None] + 1e-6)

    # ---- reverse step -------------------------------------------------------------------------------------------------------
    def reverse(self, rigid_t, seq_t, rot_score, trans_score, logits_t, t, dt, diffuse_mask=None, center=True,
                noise_scale=1.0, noise=None, sample_ids=None, step=None, step_dev=None, rates_out=None, jumps_out=None):
        """full_diffuser.py:174-227.  Extension kwargs (not in the reference): `noise` = recorded draws {'z_rot','z_trans'[,'jumps']}
        or {'u_jumps'} (B,L,20) uniforms that drive the Poisson inverse cdf; `sample_ids` / `step` / `step_dev` key the device
        Philox streams; `rates_out` / `jumps_out` (B,L,20) fp32 receive the Poisson rates * dt and the applied jump counts."""
        dev = rigid_t.device
        self.to(dev)
        B, L = rigid_t.shape[:2]
        if diffuse_mask is None:
            diffuse_mask = torch.ones(B, L, dtype=torch.int32, device=dev)
        if step is None:
            step = self._step_counter
            self._step_counter += 1
        rigid_in = rigid_t.contiguous()
        if rigid_in.dtype not in (torch.float32, torch.float64):
            rigid_in = rigid_in.float()
        ts = trans_score.contiguous()
        if ts.dtype not in (torch.float32, torch.float64):
            ts = ts.float()
        out_r = torch.empty(B, L, 7, dtype=torch.float64, device=dev)
        out_s = torch.empty(B, L, dtype=torch.int64, device=dev)
        # the reference's loop hands dt over as a 0-dim device tensor (inference.py:198-199): read it on the device, no host sync
        dt_dev = dt.to(torch.float32).reshape(1).contiguous() if (torch.is_tensor(dt) and dt.is_cuda) else None
        kw = dict(rigid_in=rigid_in, rigid_is_f64=int(rigid_in.dtype == torch.float64), seq_in=seq_t.to(torch.int64).contiguous(),
                  rot_score=rot_score.float().contiguous(), trans_score=ts, ts_is_f32=int(ts.dtype == torch.float32),
                  logits=logits_t.float().contiguous(), diffuse_mask=diffuse_mask.to(torch.int32).contiguous(),
                  t=t.to(torch.float64).contiguous(), dt=0.0 if dt_dev is not None else float(dt), seed=int(self.seed), step=int(step),
                  exp_max_sigma=self.exp_max_sigma, exp_min_sigma=self.exp_min_sigma, min_b=self.min_b_f32,
                  bdiff=self.bdiff_f32, coord_scale=self.coord_scale_f32, rate_const=self.rate_const,
                  noise_scale=float(noise_scale), center=int(bool(center)), rigid_out=out_r, seq_out=out_s, B=B, L=L)
        if noise is not None:
            if noise.get('z_rot') is not None:
                kw.update(z_rot=noise['z_rot'].float().contiguous(), z_trans=noise['z_trans'].float().contiguous())
            if noise.get('jumps') is not None:
                kw.update(jumps=noise['jumps'].float().contiguous())
            if noise.get('u_jumps') is not None:
                kw.update(u_jumps=noise['u_jumps'].float().contiguous())
        for name, buf in (('rates_out', rates_out), ('jumps_out', jumps_out)):
            if buf is not None:
                assert buf.dtype == torch.float32 and buf.is_contiguous() and buf.shape == (B, L, 20)
                kw[name] = buf
        if sample_ids is not None:
            kw.update(sample_ids=sample_ids.to(torch.int64).contiguous())
        if dt_dev is not None:
            kw.update(dt_dev=dt_dev)
        if step_dev is not None:
            assert step_dev.dtype == torch.int32 and step_dev.is_cuda
            kw.update(step_dev=step_dev)
        ops.reverse_step(**kw)
        return out_r, out_s

    # ---- init-time sampling (once per sample) ------------------------------------------------------------------------------
    def _apply_mask(self, x_diff, x_fixed, m):
        return m * x_diff + (1 - m) * x_fixed

    def _sample_igso3(self, t, shape, noise, dev):
        """Axis * angle with the angle from the inverse cdf at sigma(t) (so3_diffuser.py:222-258)."""
        x = noise['rot_axis'].to(dev) if noise else torch.randn((*shape, 3), device=dev)
        u = noise['rot_u'].to(dev) if noise else torch.rand(shape, device=dev)
        x = x / torch.linalg.norm(x, dim=-1, keepdim=True)
        om = self.discrete_omega_dev[None, :].expand(t.shape[0], -1)
        return x * _interp_rows(u, self._cdf[self._sigma_idx(t)], om)[..., None]

    def sample_ref(self, n_samples, impute_rigids=None, impute_seq=None, diffuse_mask=None, noise=None):
        """full_diffuser.py:229-290.  Draw order: randn(B,L,3), rand(B,L), randn(B,L,3), randint(B,L)."""
        dev = impute_rigids.device
        self.to(dev)
        B, L = n_samples
        trans_imp = impute_rigids[..., 4:] * self.coord_scale_f32
        rot_imp = quat_to_rotvec(impute_rigids[..., :4])
        rot_ref = self._sample_igso3(torch.ones(B, device=dev), (B, L), noise, dev)
        trans_ref = noise['trans_z'].to(dev) if noise else torch.randn((B, L, 3), device=dev)
        seq_ref = noise['seq'].to(dev) if noise else torch.randint(low=0, high=20, size=(B, L), device=dev)
        if diffuse_mask is not None:
            rot_ref = self._apply_mask(rot_ref, rot_imp, diffuse_mask[..., None])
            trans_ref = self._apply_mask(trans_ref, trans_imp, diffuse_mask[..., None])
            seq_ref = self._apply_mask(seq_ref, impute_seq, diffuse_mask)
        trans_ref = trans_ref / self.coord_scale_f32
        return {'rigids_t': torch.cat([rotvec_to_quat(rot_ref), trans_ref], dim=-1), 'seq_t': seq_ref}

    def forward_marginal(self, rigids_0, seq_0, t, diffuse_mask=None, noise=None):
        """full_diffuser.py:57-126 (optimize mode: noise the ground truth to time t)."""
        dev = rigids_0.device
        self.to(dev)
        B, L = rigids_0.shape[:2]
        trans_0, rot_0 = rigids_0[..., 4:], quat_to_rotvec(rigids_0[..., :4])
        sampled = self._sample_igso3(t, (B, L), noise, dev)
        rot_score = self.calc_rotvec_score(sampled, t)
        rot_t = quat_to_rotvec(quat_multiply(rotvec_to_quat(rot_0), rotvec_to_quat(sampled)))
        x0 = trans_0 * self.coord_scale_f32
        lmc = (-0.5 * self._marginal_b_t(t)).view(-1, 1, 1)
        z = noise['trans_z'].to(dev) if noise else torch.randn(x0.shape, device=dev)
        x_t = torch.exp(lmc) * x0 + torch.sqrt(1.0 - torch.exp(2.0 * lmc)) * z
        trans_score = self.calc_trans_score(x_t, x0, t, scale=False)
        trans_t = x_t / self.coord_scale_f32
        # tokens (discrete_diffuser.py:72-127): x_t ~ Categorical(q_t0[x_0]); then ONE extra jump per sample ("x_tilde"): the
        # position ~ Categorical(sum of off-diagonal rates) (uniform here), the new token ~ Categorical(rate row of x_t)
        e = torch.exp(-20 * self.rate_const * t.float())[:, None, None]
        q = e * torch.eye(20, device=dev)[None] + (1 - e) / 20
        q = torch.where(q < 1e-8, torch.zeros_like(q), q)
        x0c = torch.clamp(seq_0, 0, 19).long()
        rows = torch.gather(q, 1, x0c[..., None].expand(B, L, 20))
        def cat_draw(probs, key, ukey):
            # recorded draw (parity) | inverse-cdf of a supplied uniform (per-sample keyed noise) | torch's global generator
            if noise and key in noise:
                return noise[key].to(dev)
            if noise and ukey in noise:
                cdf = torch.cumsum(probs, dim=-1)
                u = noise[ukey].to(dev).reshape(probs.shape[:-1])[..., None] * cdf[..., -1:]
                return torch.clamp(torch.sum(cdf <= u, dim=-1), max=probs.shape[-1] - 1)
            return torch.distributions.Categorical(probs).sample()
        x_t = cat_draw(rows, 'seq_xt', 'u_xt')
        rate_rows = self.rate_const * (1.0 - torch.nn.functional.one_hot(x_t.long(), 20).float())      # (B,L,20), diagonal zeroed
        dims = cat_draw(rate_rows.sum(-1), 'seq_dim', 'u_dim')
        bidx = torch.arange(B, device=dev)
        newv = cat_draw(rate_rows[bidx, dims], 'seq_new', 'u_new')
        seq_t = x_t.clone()
        seq_t[bidx, dims] = newv
        if diffuse_mask is not None:
            m = diffuse_mask
            rot_t = self._apply_mask(rot_t, rot_0, m[..., None])
            trans_t = self._apply_mask(trans_t, trans_0, m[..., None])
            trans_score = self._apply_mask(trans_score, torch.zeros_like(trans_score), m[..., None])
            rot_score = self._apply_mask(rot_score, torch.zeros_like(rot_score), m[..., None])
            seq_t = self._apply_mask(seq_t, seq_0, m)
        rs, tsc = self.score_scaling(t)
        return {'rigids_t': torch.cat([rotvec_to_quat(rot_t), trans_t], dim=-1), 'trans_score': trans_score,
                'rot_score': rot_score, 'trans_score_scaling': tsc, 'rot_score_scaling': rs, 'seq_t': seq_t}

    def calc_rotvec_score(self, vec, t):
        omega = torch.linalg.norm(vec, dim=-1) + 1e-6
        sn = self.score_norms[self._sigma_idx(t)]
        oi = torch.bucketize(omega, self.discrete_omega_dev[:-1])
        return torch.gather(sn, 1, oi)[..., None] * vec / (omega[..., None] + 1e-6)
