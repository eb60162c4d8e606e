"""Per-sample feature construction run once before the reverse loop (host plumbing on torch tensors, any device).

Mirrors the seven transforms of the reference's config/config_data_feature.json as far as the sampling path reads
their outputs:
  make_restype_atom_constants  abx/model/features.py:52-66
  make_gt_frames               abx/model/features.py:88-96  -> abx/common/geometry.py:9-63
  make_torsion_angles          abx/model/features.py:107-115 -> abx/common/geometry.py:115-211
  make_pseudo_beta             abx/model/features.py:78-86
  make_diffuser_features       abx/model/features.py:130-212  (mask logic, t, noise initialisation)
Loss-only outputs (alt positions, calpha3 frames) are not produced.
"""
import torch
import torch.nn.functional as F

from abx_amd import residue_constants as rc


def _t(a, device, dtype=None):
    return torch.as_tensor(a, device=device) if dtype is None else torch.as_tensor(a, device=device).to(dtype)


def gather_rows(params, idx):
    """params (B,L,A,...) , idx (B,L,K) -> params[b,l,idx[b,l,k],...]  (reference batched_select, batch_dims=2)."""
    B, L, K = idx.shape
    tail = params.shape[3:]
    ix = idx.long().reshape(B, L, K, *([1] * len(tail))).expand(B, L, K, *tail)
    return torch.gather(params, 2, ix)


def make_restype_atom_constants(batch):
    dev = batch['seq'].device
    seq = batch['seq'].long()
    batch['atom14_atom_exists'] = _t(rc.restype_atom14_mask, dev)[seq]
    if 'residx_atom37_to_atom14' not in batch:
        batch['residx_atom37_to_atom14'] = _t(rc.restype_atom37_to_atom14, dev)[seq]
    if 'atom37_atom_exists' not in batch:
        batch['atom37_atom_exists'] = _t(rc.restype_atom37_mask, dev)[seq]
    return batch


def make_atom37_positions(batch):
    batch['atom37_gt_positions'] = gather_rows(batch['atom14_gt_positions'], batch['residx_atom37_to_atom14'])
    batch['atom37_gt_exists'] = torch.logical_and(
        gather_rows(batch['atom14_gt_exists'], batch['residx_atom37_to_atom14']), batch['atom37_atom_exists'])
    return batch


def _robust_normalize(v, eps=1e-8):
    return v / torch.sqrt(torch.sum(v * v, dim=-1, keepdim=True) + eps)


def rigids_from_3_points(p_neg_x, origin, p_xy):
    """Gram-Schmidt frame, columns (e0,e1,e2) (reference abx/model/r3.py:89-109)."""
    e0 = _robust_normalize(origin - p_neg_x)
    e1u = p_xy - origin
    c = torch.sum(e1u * e0, dim=-1, keepdim=True)
    e1 = _robust_normalize(e1u - c * e0)
    e2 = torch.stack([e0[..., 1] * e1[..., 2] - e0[..., 2] * e1[..., 1],
                      e0[..., 2] * e1[..., 0] - e0[..., 0] * e1[..., 2],
                      e0[..., 0] * e1[..., 1] - e0[..., 1] * e1[..., 0]], dim=-1)
    return torch.stack((e0, e1, e2), dim=-1), origin


def atom37_to_frames(aatype, pos37, mask37):
    dev = aatype.device
    aa = aatype.long()
    base_idx = _t(rc.restype_rigidgroup_base_atom37_idx, dev).long()[aa]          # (B,L,8,3)
    B, L = aa.shape
    flat = base_idx.reshape(B, L, 24)
    base = gather_rows(pos37, flat).reshape(B, L, 8, 3, 3)
    rots, trans = rigids_from_3_points(base[..., 0, :], base[..., 1, :], base[..., 2, :])
    group_exists = _t(rc.restype_rigidgroup_mask, dev)[aa]
    atoms_exist = gather_rows(mask37, flat).reshape(B, L, 8, 3)
    gt_exists = torch.logical_and(torch.all(atoms_exist, dim=-1), group_exists)
    flip = torch.eye(3, dtype=rots.dtype, device=dev).repeat(8, 1, 1)
    flip[0, 0, 0] = -1
    flip[0, 2, 2] = -1
    rots = torch.einsum('...rd,...dm->...rm', rots, flip)
    return {'rigidgroups_gt_frames': (rots, trans), 'rigidgroups_gt_exists': gt_exists,
            'rigidgroups_group_exists': group_exists}


# The seven torsions of a residue as ONE table of (residue offset, atom37 index) quadruples: rows 0-2 are the backbone torsions
# pre-omega (CA, C of the previous residue; N, CA), phi (C of the previous residue; N, CA, C) and psi (N, CA, C, O), rows 3-6 the side
# chain chi angles of the residue type.  A quadruple (a, b, c, d) measures the rotation of d about the b -> c axis, from the a side.
_BACKBONE_TORSIONS = (((-1, 'CA'), (-1, 'C'), (0, 'N'), (0, 'CA')),
                      ((-1, 'C'), (0, 'N'), (0, 'CA'), (0, 'C')),
                      ((0, 'N'), (0, 'CA'), (0, 'C'), (0, 'O')))
_PSI_SIGN = (1.0, 1.0, -1.0, 1.0, 1.0, 1.0, 1.0)       # the reference stores psi mirrored (geometry.py:196-198)


def _torsion_tables(device):
    """-> atom (21, 7, 4) int64 atom37 slots, prev (7, 4) bool (atom taken from residue l - 1), defined (21, 7) bool, mirror (21, 7)."""
    key = str(device)
    if key not in _torsion_tables.cache:
        chi = torch.as_tensor(rc.chi_angles_atom_indices).long()                                  # (21, 4, 4)
        bb = torch.tensor([[rc.atom_order[name] for _, name in quad] for quad in _BACKBONE_TORSIONS])
        atom = torch.cat([bb[None].expand(chi.shape[0], 3, 4), chi], dim=1)
        prev = torch.tensor([[off < 0 for off, _ in quad] for quad in _BACKBONE_TORSIONS] + [[False] * 4] * 4)
        defined = torch.cat([torch.ones(chi.shape[0], 3, dtype=torch.bool), torch.as_tensor(rc.chi_angles_mask) > 0], dim=1)
        mirror = torch.cat([torch.ones(chi.shape[0], 3), 1.0 - 2.0 * torch.as_tensor(rc.chi_pi_periodic).float()], dim=1)
        _torsion_tables.cache[key] = tuple(t.to(device) for t in (atom, prev, defined, mirror))
    return _torsion_tables.cache[key]


_torsion_tables.cache = {}


def dihedral_sin_cos(a, b, c, d):
    """(sin, cos) of the rotation of d about the axis b -> c, measured from a (..., 3 each): d is expressed in the frame whose
    origin is c, whose x axis continues b -> c and whose xy plane holds a; its (z, y) coordinates, normalised with the reference's
    1e-8 under the root, are (sin, cos).  The coordinates are taken as R^T d - R^T c - two rotations, then the difference - because
    that is the rounding the reference's frame algebra has (geometry.py:176-190) and the goldens are compared at 2e-6."""
    frame, origin = rigids_from_3_points(b, c, a)                                                 # columns e0, e1, e2
    local = torch.einsum('...dr,...d->...r', frame, d) - torch.einsum('...dr,...d->...r', frame, origin)
    zy = local[..., [2, 1]]
    return zy / torch.sqrt((zy * zy).sum(-1, keepdim=True) + 1e-8)


def atom37_to_torsion_angles(aatype, pos, mask):
    """make_torsion_angles (abx/model/features.py:107-115 -> abx/common/geometry.py:115-211): torsion_angles_sin_cos (B, L, 7, 2),
    the pi-periodic alternative and the mask (all four atoms present and the torsion defined for the residue type; residue 0 has
    no predecessor: its pre-omega / phi are masked and evaluated on zero coordinates, like the reference's zero padding)."""
    B, L = aatype.shape
    atom, prev, defined, mirror = _torsion_tables(aatype.device)
    aa = aatype.long()
    slot = atom[aa].reshape(B, L, 28)                                                             # atom37 slot of every torsion atom
    here = (gather_rows(pos, slot), gather_rows(mask.bool(), slot))
    before = tuple(F.pad(t[:, :-1], [0, 0] * (t.dim() - 2) + [1, 0]) for t in here)               # the same slots of residue l - 1
    take_prev = prev.reshape(1, 1, 28)
    xyz = torch.where(take_prev[..., None], before[0], here[0]).reshape(B, L, 7, 4, 3)
    present = torch.where(take_prev, before[1], here[1]).reshape(B, L, 7, 4)
    sc = dihedral_sin_cos(xyz[..., 0, :], xyz[..., 1, :], xyz[..., 2, :], xyz[..., 3, :])
    sc = sc * torch.tensor(_PSI_SIGN, device=sc.device)[:, None]
    return {'torsion_angles_sin_cos': sc, 'alt_torsion_angles_sin_cos': sc * mirror[aa][..., None],
            'torsion_angles_mask': present.all(-1) & defined[aa]}


def pseudo_beta_fn(aatype, pos37, mask37):
    is_gly = aatype == rc.restype_order['G']
    ca, cb = rc.atom_order['CA'], rc.atom_order['CB']
    pb = torch.where(is_gly[..., None], pos37[..., ca, :], pos37[..., cb, :])
    pm = torch.where(is_gly, mask37[..., ca].float(), mask37[..., cb].float())
    return pb, pm


def _sqrt_pos(x):
    return torch.sqrt(torch.clamp(x, min=0.0)) * (x > 0)


def rot_to_quat(m):
    """Best-conditioned-candidate rotation->quaternion (reference quat_affine.py:181-231)."""
    b = m.shape[:-2]
    m00, m01, m02, m10, m11, m12, m20, m21, m22 = torch.unbind(m.reshape(b + (9,)), dim=-1)
    q_abs = _sqrt_pos(torch.stack([1.0 + m00 + m11 + m22, 1.0 + m00 - m11 - m22,
                                   1.0 - m00 + m11 - m22, 1.0 - m00 - m11 + m22], dim=-1))
    cand = torch.stack([
        torch.stack([q_abs[..., 0] ** 2, m21 - m12, m02 - m20, m10 - m01], dim=-1),
        torch.stack([m21 - m12, q_abs[..., 1] ** 2, m10 + m01, m02 + m20], dim=-1),
        torch.stack([m02 - m20, m10 + m01, q_abs[..., 2] ** 2, m12 + m21], dim=-1),
        torch.stack([m10 - m01, m20 + m02, m21 + m12, q_abs[..., 3] ** 2], dim=-1)], dim=-2)
    cand = cand / (2.0 * torch.clamp(q_abs[..., None], min=0.1))
    best = q_abs.argmax(dim=-1)
    return torch.gather(cand, -2, best[..., None, None].expand(b + (1, 4))).squeeze(-2)


def make_diffuser_features(batch, generate_area, diffuser, diff_conf=None, opt_step=None, noise=None):
    """Inference branch of the reference's make_diffuser_features (features.py:130-212, is_training=False).

    diffused residues of CDR c with anchors at a<b:  [a+1, b-1)  — the last CDR residue stays fixed (features.py:166).
    design/trajectory (opt_step None): t=1, FullDiffuser.sample_ref ; optimize: t=opt_step/inference_step,
    FullDiffuser.forward_marginal.  `noise`: optional recorded draws (parity mode), see FullDiffuser.
    """
    dev = batch['seq'].device
    anchor_flag = batch['anchor_flag'].int()
    Lab = anchor_flag.shape[1]
    B = batch['seq'].shape[0]
    rots, trans = batch['rigidgroups_gt_frames']
    rigids_0 = torch.cat([rot_to_quat(rots[:, :, 0]), trans[:, :, 0]], dim=-1)
    seq_0 = batch['seq']
    if generate_area == 'cdr':
        cdrs = sorted(set(anchor_flag[anchor_flag > 0].tolist()))
    else:
        cdrs = [rc.cdr_str_to_enum[generate_area]]
    # Anchor positions of a CDR are paired in ROW-MAJOR order over the whole batch (features.py:159-161 walk a flat nonzero() list two
    # entries at a time): entries 2p, 2p + 1 are the (opening, closing) pair p, applied to the row of the OPENING anchor; an unpaired
    # last entry is dropped, and an odd count in one row of a batch makes the pairing run across rows - kept, it is what the
    # reference feeds its network.  Diffused: columns opening + 1 ... closing - 2 (the last residue before the closing anchor stays
    # fixed, :166); structure-loss window: opening - 1 ... closing, its end clipped with the TOTAL length of the complex (:167), so
    # the last antibody position is left out only when no antigen follows it.
    Lab_, Ltot = anchor_flag.shape[1], batch['mask'].shape[1]
    diffused = torch.zeros_like(batch['mask'], dtype=torch.int32)
    ab_loss_mask = torch.zeros_like(anchor_flag, dtype=torch.int32)
    struc_loss_mask = batch['mask'].to(torch.int32).clone()
    cols = torch.arange(Lab_, device=dev)[None, :]
    for c in cdrs:
        pos = torch.nonzero(anchor_flag == c)                       # (n, 2) row-major
        npair = pos.shape[0] // 2
        if npair == 0:
            continue
        opening, closing = pos[0:2 * npair:2], pos[1:2 * npair:2]
        rows, lo, hi = opening[:, 0], opening[:, 1:2], closing[:, 1:2]
        dsel = (cols >= lo + 1) & (cols < hi - 1)
        lsel = (cols >= torch.clamp(lo - 1, min=0)) & (cols < torch.clamp(hi + 1, max=Ltot - 1))
        diffused[:, :Lab_].index_put_((rows,), dsel.int(), accumulate=True)
        ab_loss_mask.index_put_((rows,), lsel.int(), accumulate=True)
    diffused.clamp_(max=1)
    ab_loss_mask.clamp_(max=1)
    struc_loss_mask[:, :Lab] = ab_loss_mask
    fixed_mask = 1 - diffused
    if opt_step is None:
        t = torch.ones((B,), device=dev, dtype=torch.float32)
        feats = diffuser.sample_ref(n_samples=rigids_0.shape[:2], impute_rigids=rigids_0, impute_seq=seq_0,
                                    diffuse_mask=diffused, noise=noise)
    else:
        step = (diff_conf or diffuser._diff_conf)['inference_step']
        t = torch.full((B,), fill_value=opt_step / step, device=dev, dtype=torch.float32)
        feats = diffuser.forward_marginal(rigids_0=rigids_0, seq_0=seq_0, t=t, diffuse_mask=diffused, noise=noise)
    batch.update(feats)
    batch.update(t=t, struc_loss_mask=struc_loss_mask, fixed_mask=fixed_mask, rigids_0=rigids_0)
    return batch


def per_sample_init_noise(sample_ids, L, seed, device=None):
    """Init-time noise keyed by (seed, sample id): one CPU generator per sample, drawn with shape (1, L, ...) in the reference's
    order (SURVEY.md §7 hard part 2: randn(B,L,3) axis, rand(B,L), randn(B,L,3), randint(B,L)), so a sample's starting point does
    not depend on the batch or on the rank it runs in.  The extra uniforms feed forward_marginal's categorical draws (optimize mode).
    Returns the `noise=` dict of FullDiffuser.sample_ref / forward_marginal."""
    ids = [int(i) for i in (sample_ids.tolist() if torch.is_tensor(sample_ids) else sample_ids)]
    cols = {k: [] for k in ('rot_axis', 'rot_u', 'trans_z', 'seq', 'u_xt', 'u_dim', 'u_new')}
    for sid in ids:
        g = torch.Generator().manual_seed((int(seed) * 0x9E3779B1 + sid * 0x85EBCA77 + 0x165667B1) & 0x7FFFFFFFFFFFFFFF)
        cols['rot_axis'].append(torch.randn(1, L, 3, generator=g))
        cols['rot_u'].append(torch.rand(1, L, generator=g))
        cols['trans_z'].append(torch.randn(1, L, 3, generator=g))
        cols['seq'].append(torch.randint(low=0, high=20, size=(1, L), generator=g))
        cols['u_xt'].append(torch.rand(1, L, generator=g))
        cols['u_dim'].append(torch.rand(1, generator=g))
        cols['u_new'].append(torch.rand(1, generator=g))
    if not ids:
        return None
    out = {k: torch.cat(v, dim=0) for k, v in cols.items()}
    return {k: v.to(device) for k, v in out.items()} if device is not None else out


def build_features(batch, diffuser, generate_area='H3', opt_step=None, noise=None, device=None):
    """The whole inference feature pipeline on a collated batch (config_data_feature.json order)."""
    if device is not None:
        batch = {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in batch.items()}
    batch = make_restype_atom_constants(batch)
    batch = make_atom37_positions(batch)
    batch.update(atom37_to_frames(batch['seq'], batch['atom37_gt_positions'], batch['atom37_gt_exists']))
    batch.update(atom37_to_torsion_angles(batch['seq'], batch['atom37_gt_positions'], batch['atom37_gt_exists']))
    batch['pseudo_beta'], batch['pseudo_beta_mask'] = pseudo_beta_fn(
        batch['seq'], batch['atom37_gt_positions'], batch['atom37_gt_exists'])
    return make_diffuser_features(batch, generate_area, diffuser, opt_step=opt_step, noise=noise)
