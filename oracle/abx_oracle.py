"""ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product path.

A plain PyTorch-CPU (fp32 / fp64 exactly where the reference is) restatement of the AbX reverse-diffusion
sampling hot path, written from the behaviour documented in SURVEY.md §8(a).  Only `tests/`,
`__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import this file; the product
(`abx_amd/`) never does and fails loudly when its HIP library is missing.

Pinning: the reference ships no tests / golden vectors for this path, so the oracle is pinned
against outputs of the reference itself, generated in the build container by `tests/golden/make_golden.py`
(imports /root/reference with in-memory stand-ins) and committed under `tests/golden/*.npz`;
`tests/test_oracle_golden.py` checks every function below against them.

Reference sites restated (file:line under /root/reference):
  score network   abx/model/abx.py:17-104, abx/model/seqformer.py:49-630, abx/model/encoder.py:123-269,
                  abx/model/score_network.py:83-196, abx/model/folding.py:47-132, abx/model/head.py:143-226,
                  abx/model/sidechain.py:28-91, abx/model/atom.py:9-76, abx/model/quat_affine.py:53-238,
                  abx/model/r3.py:9-59, abx/model/common_modules.py:62-120, abx/model/utils.py:12-14,158-171
  diffuser        diffuser/full_diffuser.py:12-290, diffuser/so3_diffuser.py:15-361, diffuser/r3_diffuser.py:10-164,
                  diffuser/discrete_diffuser.py:9-190, abx/utils.py:31-59
  sampling loop   inference.py:166-273
Functions take a flat `params` dict keyed like the reference state_dict (190 tensors, ESM disabled).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from abx_amd import residue_constants as rc

P_SEQF = 'impl.seqformer.'
P_BLK = 'impl.seqformer.seqformer.blocks.0.'
P_IPA = 'impl.diffusion_module.ScoreNetwork.'


# --------------------------------------------------------------------------------------------------
# small helpers
# --------------------------------------------------------------------------------------------------
def lin(p, name, x, bias=True):
    return F.linear(x, p[name + '.weight'], p[name + '.bias'] if bias and (name + '.bias') in p else None)


def lnorm(p, name, x):
    w = p[name + '.weight']
    return F.layer_norm(x, (w.shape[0],), w, p[name + '.bias'], 1e-5)


def _tbl(a, like):
    return torch.as_tensor(a, device=like.device)


# ---- quaternion / rigid algebra (quat_affine.py, r3.py) -------------------------------------------
def quat_to_rot(q):
    """quat_affine.py:53-60 (table contraction written out)."""
    a, b, c, d = q.unbind(-1)
    r = torch.stack([
        a * a + b * b - c * c - d * d, 2 * (b * c - a * d), 2 * (b * d + a * c),
        2 * (b * c + a * d), a * a - b * b + c * c - d * d, 2 * (c * d - a * b),
        2 * (b * d - a * c), 2 * (c * d + a * b), a * a - b * b - c * c + d * d], dim=-1)
    return r.reshape(q.shape[:-1] + (3, 3))


def quat_multiply(q1, q2):
    """Hamilton product, quat_affine.py:69-75."""
    a1, b1, c1, d1 = q1.unbind(-1)
    a2, b2, c2, d2 = q2.unbind(-1)
    return torch.stack([a1 * a2 - b1 * b2 - c1 * c2 - d1 * d2,
                        a1 * b2 + b1 * a2 + c1 * d2 - d1 * c2,
                        a1 * c2 - b1 * d2 + c1 * a2 + d1 * b2,
                        a1 * d2 + b1 * c2 - c1 * b2 + d1 * a2], dim=-1)


def quat_multiply_by_vec(q, v):
    """q (x) (0,v), quat_affine.py:62-67."""
    a, b, c, d = q.unbind(-1)
    x, y, z = v.unbind(-1)
    return torch.stack([-b * x - c * y - d * z,
                        a * x + c * z - d * y,
                        a * y - b * z + d * x,
                        a * z + b * y - c * x], dim=-1)


def l2_normalize(v, eps=1e-12):
    return v / torch.sqrt(torch.sum(v * v, dim=-1, keepdim=True) + eps)


def quat_precompose_vec(q, v):
    """quat_affine.py:77-85."""
    return l2_normalize(q + quat_multiply_by_vec(q, v))


def _sin_half_over(angles, half):
    small = torch.abs(angles) < 1e-6
    safe = torch.where(small, torch.ones_like(angles), angles)
    return torch.where(small, 0.5 - angles * angles / 48, torch.sin(half) / safe)


def quat_to_rotvec(q):
    """quat_affine.py:113-131: flip to w>=0, angle=2 atan2(|v|,w), small-angle series."""
    flip = (q[..., :1] < 0).to(q.dtype)
    q = (-1.0 * q) * flip + (1.0 - flip) * q
    norms = torch.norm(q[..., 1:], p=2, dim=-1, keepdim=True)
    half = torch.atan2(norms, q[..., :1])
    return q[..., 1:] / _sin_half_over(2 * half, half)


def rotvec_to_quat(v):
    """quat_affine.py:133-150."""
    angles = torch.norm(v, p=2, dim=-1, keepdim=True)
    half = angles * 0.5
    return torch.cat([torch.cos(half), v * _sin_half_over(angles, half)], dim=-1)


def invert_quat(q):
    """quat_affine.py:234-238."""
    qp = torch.cat([q[..., :1], -q[..., 1:]], dim=-1)
    return qp / torch.sqrt(torch.sum(q ** 2, dim=-1, keepdim=True))


def rigid_apply(rots, trans, pts):
    """R p + t on (..., L, M, 3) points (r3.py:9-16)."""
    return trans[..., None, :] + torch.einsum('...lrd,...lmd->...lmr', rots, pts)


def rigid_invert_apply(rots, trans, pts):
    """R^T p - R^T t (r3.py:54-59 then 9-16)."""
    inv_t = -torch.einsum('...ldr,...ld->...lr', rots, trans)
    return inv_t[..., None, :] + torch.einsum('...ldr,...lmd->...lmr', rots, pts)


# ---- atoms from torsions (atom.py) ------------------------------------------------------------------
def torsion_angles_to_frames(aatype, rots, trans, sincos):
    """atom.py:9-56. returns (R (B,L,8,3,3), t (B,L,8,3)) global frames."""
    m = _tbl(rc.restype_rigid_group_default_frame, sincos)[aatype.long()]       # (B,L,8,4,4)
    dR, dt = m[..., :3, :3], m[..., :3, 3]
    sin = F.pad(sincos[..., 0], (1, 0), value=0.)
    cos = F.pad(sincos[..., 1], (1, 0), value=1.)
    z, o = torch.zeros_like(sin), torch.ones_like(sin)
    rx = torch.stack([o, z, z, z, cos, -sin, z, sin, cos], dim=-1).reshape(sin.shape + (3, 3))
    fR = torch.einsum('...rd,...dm->...rm', dR, rx)
    ft = dt

    def compose(Ra, ta, Rb, tb):
        return torch.einsum('...rd,...dm->...rm', Ra, Rb), torch.einsum('...rd,...d->...r', Ra, tb) + ta

    R4, t4 = fR[:, :, 4], ft[:, :, 4]
    R5, t5 = compose(R4, t4, fR[:, :, 5], ft[:, :, 5])
    R6, t6 = compose(R5, t5, fR[:, :, 6], ft[:, :, 6])
    R7, t7 = compose(R6, t6, fR[:, :, 7], ft[:, :, 7])
    bR = torch.cat([fR[:, :, 0:5], R5[:, :, None], R6[:, :, None], R7[:, :, None]], dim=2)
    bt = torch.cat([ft[:, :, 0:5], t5[:, :, None], t6[:, :, None], t7[:, :, None]], dim=2)
    gR = rots[:, :, None].expand(-1, -1, 8, -1, -1)
    gt = trans[:, :, None].expand(-1, -1, 8, -1)
    return compose(gR, gt, bR, bt)


def frames_to_atom14(aatype, fR, ft):
    """atom.py:58-76."""
    aa = aatype.long()
    grp = _tbl(rc.restype_atom14_to_rigid_group, ft).long()[aa]                 # (B,L,14)
    lit = _tbl(rc.restype_atom14_rigid_group_positions, ft)[aa]                 # (B,L,14,3)
    R = torch.gather(fR, 2, grp[..., None, None].expand(-1, -1, -1, 3, 3))
    t = torch.gather(ft, 2, grp[..., None].expand(-1, -1, -1, 3))
    return t + torch.einsum('...rd,...d->...r', R, lit)


def atom14_to_atom37(atom14, residx_atom37_to_atom14):
    idx = residx_atom37_to_atom14.long()
    return torch.gather(atom14, 2, idx[..., None].expand(-1, -1, -1, 3))


def pseudo_beta_v2(pos):
    """common_modules.py:62-83 (N, CA, C are atoms 0,1,2 in both atom14 and atom37)."""
    N, CA, C = pos[..., 0, :], pos[..., 1, :], pos[..., 2, :]
    b = CA - N
    c = C - CA
    a = torch.cross(b, c, dim=-1)
    return -0.58273431 * a + 0.56802827 * b - 0.54067466 * c + CA


def dgram_from_positions(pos, num_bins, min_bin, max_bin):
    """common_modules.py:107-120 -> int64 bins (B,L,L)."""
    breaks = torch.linspace(min_bin, max_bin, steps=num_bins - 1, device=pos.device)
    sq = torch.square(breaks)
    d2 = torch.sum(torch.square(pos[:, :, None, :] - pos[:, None, :, :]), dim=-1, keepdim=True)
    return torch.sum(d2 > sq, dim=-1).long()


def timestep_embedding(t, dim, max_positions=10000):
    """seqformer.py:49-65 (note: t*max_positions is done in t's dtype, THEN cast to float)."""
    t = t * max_positions
    half = dim // 2
    e = math.log(max_positions) / (half - 1)
    e = torch.exp(torch.arange(half, dtype=torch.float32, device=t.device) * -e)
    e = t.float()[:, None] * e[None, :]
    return torch.cat([torch.sin(e), torch.cos(e)], dim=1)


# --------------------------------------------------------------------------------------------------
# trajectory-invariant encoders (encoder.py)
# --------------------------------------------------------------------------------------------------
def residue_embedding(p, batch):
    """encoder.py:123-174."""
    pre = P_SEQF + 'encode_residue_emb.'
    mask = torch.logical_and(batch['mask'], batch['fixed_mask'])
    B, L = mask.shape
    aa = batch['seq_t'].long()
    aa_feat = p[pre + 'aatype_embed.weight'][aa] * mask[:, :, None]
    cdr_feat = p[pre + 'cdr_embed.weight'][batch['cdr_def'].long()]
    x = torch.cat([batch['atom14_gt_positions'].reshape(B, L, -1),
                   batch['torsion_angles_sin_cos'].reshape(B, L, -1)], dim=-1)
    x = lin(p, pre + 'coordinate_embed.2', F.relu(lin(p, pre + 'coordinate_embed.0', x)))
    h = torch.cat([aa_feat, batch['chain_id'][..., None].float(), batch['residx'][..., None].float(), cdr_feat, x],
                  dim=-1)
    h = F.relu(lin(p, pre + 'mlp.0', h))
    h = F.relu(lin(p, pre + 'mlp.2', h))
    h = F.relu(lin(p, pre + 'mlp.4', h))
    h = lin(p, pre + 'mlp.6', h)
    return h * mask[:, :, None]


def pair_embedding(p, batch, cfg_prev_pos):
    """encoder.py:178-269."""
    pre = P_SEQF + 'encode_pair_emb.'
    mask = torch.logical_and(batch['mask'], batch['fixed_mask'])
    mask_pair = mask[:, :, None] * mask[:, None, :]
    B, L = mask.shape
    aa = batch['seq_t'].long()
    coords = batch['atom14_gt_positions']
    mask_atoms = batch['atom14_gt_exists'][..., 1]
    aa_pair = aa[:, :, None] * 23 + aa[:, None, :]
    f_aapair = p[pre + 'aa_pair_embed.weight'][aa_pair]
    same_chain = batch['chain_id'][:, :, None] == batch['chain_id'][:, None, :]
    relpos = torch.clamp(batch['residx'][:, :, None] - batch['residx'][:, None, :], min=-32, max=32)
    f_relpos = p[pre + 'relpos_embed.weight'][(relpos + 32).long()] * same_chain[:, :, :, None]
    dist = (torch.linalg.norm(coords[:, :, None, :, None] - coords[:, None, :, None, :], dim=-1, ord=2) / 10
            ).reshape(B, L, L, -1)
    coef = F.softplus(p[pre + 'aapair_to_distcoef.weight'][aa_pair])
    d_gauss = torch.exp(-1 * coef * dist ** 2)
    mask_atom_pair = mask_atoms[:, :, None, None] * mask_atoms[:, None, :, None]
    f_dist = F.relu(lin(p, pre + 'distance_embed.2', F.relu(lin(p, pre + 'distance_embed.0', d_gauss * mask_atom_pair))))
    pb = pseudo_beta_v2(coords)
    bins = dgram_from_positions(pb, **cfg_prev_pos)
    f_dgram = p[pre + 'dgram_embed.weight'][bins]
    h = torch.cat([f_aapair, f_relpos, f_dist, f_dgram], dim=-1)
    h = F.relu(lin(p, pre + 'out_mlp.0', h))
    h = F.relu(lin(p, pre + 'out_mlp.2', h))
    h = lin(p, pre + 'out_mlp.4', h)
    return h * mask_pair[:, :, :, None]


def relpos_block(p, residx, max_rel=32):
    off = residx[:, None, :] - residx[:, :, None]
    rel = torch.clip(off + max_rel, min=0, max=2 * max_rel) + 1
    return p[P_SEQF + 'proj_rel_pos.weight'][rel.long()]


def static_embeddings(p, batch, cfg):
    """Everything in EmbeddingAndSeqformer.forward that does not change along the trajectory
    (seqformer.py:177-212 minus the antibody token embedding): returns (seq_static (B,L,512) WITHOUT
    proj_aa_type[seq_t] for the antibody, pair_static (B,L,L,128))."""
    c = cfg.model.embeddings_and_seqformer
    Lab = batch['anchor_flag'].shape[1]
    B, L = batch['seq'].shape
    ag = p[P_SEQF + 'proj_aa_type.weight'][batch['seq'][:, Lab:].long()]
    ag = lnorm(p, P_SEQF + 'aa_proj.0', ag)
    ag = lin(p, P_SEQF + 'aa_proj.3', F.relu(lin(p, P_SEQF + 'aa_proj.1', ag)))
    seq_static = residue_embedding(p, batch).clone()
    seq_static[:, Lab:] = seq_static[:, Lab:] + ag
    pair = torch.zeros(B, L, L, c.pair_channel)
    pair[:, :Lab, :Lab] = relpos_block(p, batch['residx'][:, :Lab], c.max_relative_feature)
    pair[:, Lab:, Lab:] = relpos_block(p, batch['residx'][:, Lab:], c.max_relative_feature)
    pair = pair + pair_embedding(p, batch, dict(c.prev_pos))
    return seq_static, pair


# --------------------------------------------------------------------------------------------------
# Seqformer block (seqformer.py:228-606)
# --------------------------------------------------------------------------------------------------
def _attention(q, k, v, bias, k_mask, key_dim):
    """q,k,v (b,s,h,l,d); bias (b,h,q,k); k_mask (b,s|1,k) bool."""
    q = q * key_dim ** (-0.5)
    logits = torch.einsum('bshqd,bshkd->bshqk', q, k)
    if bias is not None:
        logits = logits + bias[:, None]
    if k_mask is not None:
        logits = logits.masked_fill(~k_mask.bool()[:, :, None, None, :], torch.finfo(logits.dtype).min)
    w = F.softmax(logits, dim=-1)
    o = torch.einsum('bshqk,bshkd->bshqd', w, v)
    b, s, h, l, d = o.shape
    return o.permute(0, 1, 3, 2, 4).reshape(b, s, l, h * d)


def seq_attention(p, seq, pair, mask):
    pre = P_BLK + 'seq_attn.'
    H = 32
    x = lnorm(p, pre + 'seq_norm', seq)
    z = lnorm(p, pre + 'pair_norm', pair)
    bias = lin(p, pre + 'proj_pair', z, bias=False).permute(0, 3, 1, 2)
    B, L, C = x.shape
    t = lin(p, pre + 'attn.proj_in', x, bias=False).reshape(B, L, H, 3 * C // H).permute(0, 2, 1, 3)[:, None]
    q, k, v = torch.chunk(t, 3, dim=-1)
    o = _attention(q, k, v, bias, mask[:, None, :], C // H)[:, 0]
    o = o * torch.sigmoid(lin(p, pre + 'attn.gate', x))
    return lin(p, pre + 'attn.proj_out', o)


def transition(p, pre, x):
    return lin(p, pre + '.transition.3', F.relu(lin(p, pre + '.transition.1', lnorm(p, pre + '.transition.0', x))))


def outer_product_mean(p, seq, mask):
    pre = P_BLK + 'outer_product_mean.'
    m = mask[:, :, None]
    x = lnorm(p, pre + 'norm', seq)
    left = m * lin(p, pre + 'left_proj', x)
    right = m * lin(p, pre + 'right_proj', x)
    prod = left[:, None, :, :] * right[:, :, None, :]
    diff = left[:, None, :, :] - right[:, :, None, :]
    return lin(p, pre + 'out_proj', torch.cat([prod, diff], dim=-1))


def triangle_multiplication(p, name, pair, mask, per_row):
    pre = P_BLK + name + '.'
    pm = (mask[:, :, None, None] * mask[:, None, :, None]).to(pair.dtype)
    x = lnorm(p, pre + 'norm', pair)
    left = pm * lin(p, pre + 'left_proj', x) * torch.sigmoid(lin(p, pre + 'left_gate', x))
    right = pm * lin(p, pre + 'right_proj', x) * torch.sigmoid(lin(p, pre + 'right_gate', x))
    if per_row:
        t = torch.einsum('bikc,bjkc->bijc', left, right)
    else:
        t = torch.einsum('bkic,bkjc->bijc', left, right)
    t = lin(p, pre + 'proj_out', lnorm(p, pre + 'final_norm', t))
    return t * torch.sigmoid(lin(p, pre + 'final_gate', x))


def triangle_attention(p, name, pair, mask, per_row):
    pre = P_BLK + name + '.'
    H = 4
    if not per_row:
        pair = pair.transpose(1, 2)
    x = lnorm(p, pre + 'norm', pair)
    bias = lin(p, pre + 'proj_pair', x, bias=False).permute(0, 3, 1, 2)
    B, S, L, C = x.shape

    def heads(t):
        return t.reshape(B, S, L, H, C // H).permute(0, 1, 3, 2, 4)

    q = heads(lin(p, pre + 'attn.proj_q', x, bias=False))
    k = heads(lin(p, pre + 'attn.proj_k', x, bias=False))
    v = heads(lin(p, pre + 'attn.proj_v', x, bias=False))
    o = _attention(q, k, v, bias, mask[:, None, :], C // H)
    o = o * torch.sigmoid(lin(p, pre + 'attn.gate', x))
    o = lin(p, pre + 'attn.proj_out', o)
    if not per_row:
        o = o.transpose(1, 2)
    return o


def seqformer_block(p, seq, pair, mask):
    """seqformer.py:569-606 (eval: dropout = identity)."""
    seq = seq + seq_attention(p, seq, pair, mask)
    seq = seq + transition(p, P_BLK + 'seq_transition', seq)
    pair = pair + outer_product_mean(p, seq, mask)
    pair = pair + triangle_multiplication(p, 'triangle_multiplication_outgoing', pair, mask, True)
    pair = pair + triangle_multiplication(p, 'triangle_multiplication_incoming', pair, mask, False)
    pair = pair + triangle_attention(p, 'triangle_attention_starting_node', pair, mask, True)
    pair = pair + triangle_attention(p, 'triangle_attention_ending_node', pair, mask, False)
    pair = pair + transition(p, P_BLK + 'pair_transition', pair)
    return seq, pair


def embed_and_seqformer(p, batch, cfg, static=None):
    """EmbeddingAndSeqformer.forward (seqformer.py:170-226)."""
    c = cfg.model.embeddings_and_seqformer
    if static is None:
        static = static_embeddings(p, batch, cfg)
    seq_static, pair_static = static
    Lab = batch['anchor_flag'].shape[1]
    B, L = batch['seq'].shape
    seq_act = seq_static.clone()
    seq_act[:, :Lab] = seq_act[:, :Lab] + p[P_SEQF + 'proj_aa_type.weight'][batch['seq_t'][:, :Lab].long()]
    if c.esm.enabled:
        # seqformer.py:185-191: layer-softmax mix of the supplied ESM2 representations (B, Lab, C, layers) + projection MLP
        wl = torch.softmax(p[P_SEQF + 'esm_embed_weights'], dim=-1)
        e = torch.einsum('blcn,n->blc', batch['esm_embed'].to(wl.dtype), wl)
        e = lin(p, P_SEQF + 'proj_esm_embed.3', torch.relu(lin(p, P_SEQF + 'proj_esm_embed.1', lnorm(p, P_SEQF + 'proj_esm_embed.0', e))))
        seq_act[:, :Lab] = seq_act[:, :Lab] + e
    temb = timestep_embedding(batch['t'], c.index_embed_size)                      # (B,32)
    seq_act = torch.cat([seq_act, temb[:, None, :].expand(B, L, -1)], dim=-1).float()
    tp = temb[:, None, None, :].expand(B, L, L, -1)
    pair_act = torch.cat([pair_static, tp, tp], dim=-1).float()
    if 'prev_seq' in batch:
        seq_act = seq_act + lnorm(p, P_SEQF + 'prev_seq_norm', batch['prev_seq'])
    if 'prev_pair' in batch:
        pair_act = pair_act + lnorm(p, P_SEQF + 'prev_pair_norm', batch['prev_pair'])
    if 'prev_pos' in batch:
        pair_act = pair_act + p[P_SEQF + 'proj_prev_pos.weight'][batch['prev_pos']]
    return seqformer_block(p, seq_act, pair_act, batch['mask'])


# --------------------------------------------------------------------------------------------------
# IpaScore + heads
# --------------------------------------------------------------------------------------------------
def ipa_attention(p, s, z, mask, rots, trans, ipa_cfg):
    """InvariantPointAttention.forward (folding.py:47-132)."""
    pre = P_IPA + 'attention_module.'
    c = ipa_cfg
    H, sqk, sv, pqk, pv = c.num_head, c.num_scalar_qk, c.num_scalar_v, c.num_point_qk, c.num_point_v
    B, L, _ = s.shape
    w_s = np.sqrt(1.0 / (3 * max(sqk, 1) * 1.))
    w_p = np.sqrt(1.0 / (3 * max(pqk, 1) * 9. / 2))
    w_2d = np.sqrt(1.0 / 3)
    q_s = lin(p, pre + 'proj_q_scalar', s).reshape(B, L, H, sqk).permute(0, 2, 1, 3)
    kv_s = lin(p, pre + 'proj_kv_scalar', s).reshape(B, L, H, sqk + sv).permute(0, 2, 1, 3)
    k_s, v_s = kv_s[..., :sqk], kv_s[..., sqk:]
    logits = torch.einsum('bhic,bhjc->bhij', q_s * w_s, k_s)
    q_p = lin(p, pre + 'proj_q_point_local', s).reshape(B, L, 3, H * pqk).transpose(-1, -2)      # (B,L,n,3)
    kv_p = lin(p, pre + 'proj_kv_point_local', s).reshape(B, L, 3, H * (pqk + pv)).transpose(-1, -2)
    q_g = rigid_apply(rots, trans, q_p).reshape(B, L, H, pqk, 3)
    kv_g = rigid_apply(rots, trans, kv_p).reshape(B, L, H, pqk + pv, 3)
    k_g, v_g = kv_g[:, :, :, :pqk], kv_g[:, :, :, pqk:]
    d2 = torch.sum(torch.square(q_g[:, :, None] - k_g[:, None]), dim=[-1, -2])                   # (B,i,j,H)
    pw = -0.5 * w_p * F.softplus(p[pre + 'trainable_point_weights'])
    logits = logits + (pw * d2).permute(0, 3, 1, 2)
    logits = logits + w_2d * lin(p, pre + 'proj_pair', z).permute(0, 3, 1, 2)
    m2 = (mask[:, :, None] * mask[:, None, :])[:, None]
    logits = logits.masked_fill(~m2.bool(), torch.finfo(logits.dtype).min)
    a = F.softmax(logits, dim=-1)
    o_s = torch.matmul(a, v_s).permute(0, 2, 1, 3).reshape(B, L, H * sv)
    o_pg = torch.einsum('bhij,bjhnr->bhinr', a, v_g).permute(0, 2, 1, 3, 4).reshape(B, L, H * pv, 3)
    o_pl = rigid_invert_apply(rots, trans, o_pg)
    o_pts = o_pl.transpose(-1, -2).reshape(B, L, 3 * H * pv)                                     # '(r n)'
    o_norm = torch.sqrt(torch.sum(torch.square(o_pl), dim=-1) + 1e-8)
    o_2d = torch.einsum('bhij,bijc->bhic', a, z).permute(0, 2, 1, 3).reshape(B, L, -1)
    return lin(p, pre + 'final_proj', torch.cat([o_s, o_pts, o_norm, o_2d], dim=-1))


def torsion_module(p, act, init_act):
    pre = P_IPA + 'sidechain_module.torsion_module.'
    a = lin(p, pre + 'proj_act.1', F.relu(act)) + lin(p, pre + 'proj_init_act.1', F.relu(init_act))
    for b in range(2):
        a = a + lin(p, pre + f'blocks.{b}.net.3', F.relu(lin(p, pre + f'blocks.{b}.net.1', F.relu(a))))
    out = lin(p, pre + 'projection', F.relu(a))
    return out.reshape(out.shape[:-1] + (7, 2))


def ipa_score(p, seq_act, pair_act, batch, cfg, diffuser):
    """IpaScore.forward (score_network.py:83-196)."""
    c = cfg.model.heads.diffusion_module.IPA
    node_mask = batch['mask'].float()
    fixed = batch['fixed_mask']
    init_rigids = batch['rigids_t'].float()
    init_q, init_t = init_rigids[..., :4], init_rigids[..., 4:]
    B, L = batch['seq_t'].shape
    delta_q = torch.zeros(B, L, 4)
    delta_q[..., 0] = 1.0
    cur_q = init_q
    cur_t = init_t / c.position_scale
    cur_R = quat_to_rot(cur_q)
    s = lnorm(p, P_IPA + 'init_seq_layer_norm', lin(p, P_IPA + 'proj_init_seq_act', seq_act))
    z = lnorm(p, P_IPA + 'init_pair_layer_norm', lin(p, P_IPA + 'proj_init_pair_act', pair_act))
    s0 = s
    s = lin(p, P_IPA + 'proj_seq', s)
    dm = (1 - fixed[..., None])
    for it in range(c.num_layer):
        s = s + ipa_attention(p, s, z, node_mask, cur_R, cur_t, c)
        s = lnorm(p, P_IPA + 'attention_layer_norm', s)
        h = F.relu(lin(p, P_IPA + 'transition_module.0', s))
        h = F.relu(lin(p, P_IPA + 'transition_module.2', h))
        s = s + lin(p, P_IPA + 'transition_module.4', h)
        s = lnorm(p, P_IPA + 'transition_layer_norm', s)
        upd = lin(p, P_IPA + 'affine_update', s)
        qu, tu = upd[..., :3], upd[..., 3:]
        delta_q = quat_precompose_vec(delta_q, qu)
        cur_q = quat_precompose_vec(cur_q, qu)
        cur_t = cur_t + torch.einsum('...rd,...d->...r', cur_R, tu)
        cur_q = dm * cur_q + (1 - dm) * init_q
        cur_t = dm * cur_t + (1 - dm) * (init_t / c.position_scale)
        cur_R = quat_to_rot(cur_q)
    un = torsion_module(p, s, s0)
    ang = l2_normalize(un)
    fm = fixed[..., None, None].bool()
    ang = torch.where(fm, batch['torsion_angles_sin_cos'], ang)
    q_fin = quat_multiply(init_q, delta_q)
    q_fin = dm * q_fin + (1 - dm) * init_q
    rot_score = diffuser.calc_quat_score(init_q, q_fin, batch['t'])
    trans_score = diffuser.calc_trans_score(init_t, cur_t * c.position_scale, batch['t'])
    rigids = torch.cat([q_fin, cur_t * c.position_scale], dim=-1)
    return dict(rot_score=rot_score, trans_score=trans_score, rigids=rigids, structure_module=s, angles=ang)


def _mlp_head(p, pre, x):
    h = F.relu(lin(p, pre + 'net.1', lnorm(p, pre + 'net.0', x)))
    h = F.relu(lin(p, pre + 'net.3', h))
    return lin(p, pre + 'net.5', h)


def sequence_head(p, fold, batch):
    """SequenceHead.forward (head.py:162-201): logits, seq_0 and the atom14/atom37 rebuilt with seq_0."""
    logits = _mlp_head(p, 'impl.sequence_module.', fold['structure_module'])
    seq_0 = torch.max(F.softmax(logits, dim=-1), dim=-1)[1]
    fixed = batch['fixed_mask']
    seq_0 = seq_0 * (1 - fixed) + batch['seq_t'] * fixed
    rig = fold['rigids']
    fR, ft = torsion_angles_to_frames(seq_0, quat_to_rot(rig[..., :4]), rig[..., 4:], fold['angles'])
    atom14 = frames_to_atom14(seq_0, fR, ft)
    atom37 = atom14_to_atom37(atom14, batch['residx_atom37_to_atom14'])
    return logits, seq_0, atom14, atom37


def plddt_head(p, fold):
    logits = _mlp_head(p, 'impl.predicted_lddt.', fold['structure_module'])
    n = logits.shape[-1]
    w = 1.0 / n
    centers = torch.arange(start=0.5 * w, end=1.0, step=w)
    return torch.sum(F.softmax(logits, dim=-1) * centers, dim=-1) * 100


def network_pass(p, batch, cfg, diffuser, static=None, final=True):
    """ScoreNetworkIteration.forward (abx.py:42-63); distogram/metric/tmscore heads are out of scope."""
    seq_act, pair_act = embed_and_seqformer(p, batch, cfg, static)
    fold = ipa_score(p, seq_act, pair_act, batch, cfg, diffuser)
    logits, seq_0, atom14, atom37 = sequence_head(p, fold, batch)
    ret = {'representations': {'seq': seq_act, 'pair': pair_act},
           'heads': {'folding': {'rot_score': fold['rot_score'], 'trans_score': fold['trans_score'],
                                 'rigids': fold['rigids'], 'final_atom14_positions': atom14,
                                 'final_atom_positions': atom37,
                                 'representations': {'structure_module': fold['structure_module']},
                                 'angles_sin_cos': fold['angles']},
                     'sequence_module': {'logits': logits, 'seq_0': seq_0}}}
    if final:
        ret['heads']['predicted_lddt'] = {'pLDDT': plddt_head(p, fold)}
    return ret


def get_prev(batch, value, cfg):
    """abx.py:17-26."""
    pb = pseudo_beta_v2(value['heads']['folding']['final_atom_positions'])
    bins = dgram_from_positions(pb, **dict(cfg.model.embeddings_and_seqformer.prev_pos))
    return {'prev_pos': bins, 'prev_seq': value['representations']['seq'], 'prev_pair': value['representations']['pair']}


def score_network(p, batch, cfg, diffuser, static=None):
    """ScoreNetwork.forward, eval mode (abx.py:75-104): mutates `batch` exactly like the reference."""
    c = cfg.model.embeddings_and_seqformer
    B, L = batch['seq'].shape
    if 'prev_seq' not in batch:
        batch.update(prev_pos=torch.zeros(B, L, L, dtype=torch.int64),
                     prev_seq=torch.zeros(B, L, c.seq_channel + c.index_embed_size),
                     prev_pair=torch.zeros(B, L, L, c.pair_channel + 2 * c.index_embed_size))
    with torch.no_grad():
        batch.update(is_recycling=True)
        for _ in range(cfg.model.num_recycle):
            ret = network_pass(p, batch, cfg, diffuser, static, final=False)
            prev = get_prev(batch, ret, cfg)
            batch.update(seq_t=ret['heads']['sequence_module']['seq_0'])
            batch.update(prev)
        batch.update(is_recycling=False)
        return network_pass(p, batch, cfg, diffuser, static, final=True)


# --------------------------------------------------------------------------------------------------
# Diffuser (diffuser/*.py)
# --------------------------------------------------------------------------------------------------
def igso3_expansion(omega, eps, L=1000):
    """so3_diffuser.py:15-49 for 1-D omega and scalar eps."""
    ls = torch.arange(L)[None]
    om = omega[..., None]
    return ((2 * ls + 1) * torch.exp(-ls * (ls + 1) * eps ** 2 / 2) * torch.sin(om * (ls + 1 / 2)) / torch.sin(om / 2)
            ).sum(dim=-1)


def igso3_score(exp, omega, eps, L=1000):
    """so3_diffuser.py:72-112."""
    ls = torch.arange(L)[None]
    om = omega[..., None]
    hi = torch.sin(om * (ls + 1 / 2))
    dhi = (ls + 1 / 2) * torch.cos(om * (ls + 1 / 2))
    lo = torch.sin(om / 2)
    dlo = 1 / 2 * torch.cos(om / 2)
    ds = ((2 * ls + 1) * torch.exp(-ls * (ls + 1) * eps ** 2 / 2) * (lo * dhi - hi * dlo) / lo ** 2).sum(dim=-1)
    return ds / (exp + 1e-4)


def torch_interp(x_new, x, y):
    """abx/utils.py:31-59."""
    idx = x.argsort(dim=1)
    x = torch.gather(x, -1, idx)
    y = torch.gather(y, -1, idx)
    b = torch.sum((x.unsqueeze(2) < x_new.unsqueeze(1)), dim=1)
    b = torch.clamp(b, 0, x.shape[1] - 2)
    xl, xh = torch.gather(x, -1, b), torch.gather(x, -1, b + 1)
    yl, yh = torch.gather(y, -1, b), torch.gather(y, -1, b + 1)
    w = (x_new - xl) / (xh - xl + 1e-8)
    w[x_new > x[:, -1].unsqueeze(1)] = 1.0
    w[x_new < x[:, 0].unsqueeze(1)] = 0
    return yl * (1 - w) + yh * w


class OracleSO3:
    def __init__(self, conf, tables=None):
        self.min_sigma, self.max_sigma = conf['min_sigma'], conf['max_sigma']
        self.num_sigma, self.num_omega = conf['num_sigma'], conf['num_omega']
        self.discrete_omega = torch.linspace(0, np.pi, self.num_omega + 1)[1:]
        if tables is None:
            ds = self.discrete_sigma
            ex = torch.stack([igso3_expansion(self.discrete_omega, s, 1000) for s in ds])
            pdf = ex * (1 - torch.cos(self.discrete_omega)) / torch.tensor(np.pi)
            cdf = torch.stack([torch.cumsum(r, dim=0) / self.num_omega * torch.tensor(np.pi) for r in pdf])
            sn = torch.stack([igso3_score(ex[i], self.discrete_omega, s) for i, s in enumerate(ds)])
            tables = dict(pdf=pdf, cdf=cdf, score_norms=sn)
        self._pdf, self._cdf, self._score_norms = (torch.as_tensor(tables[k]) for k in ('pdf', 'cdf', 'score_norms'))
        self._score_scaling = torch.sqrt(torch.abs(
            torch.sum(self._score_norms ** 2 * self._pdf, dim=-1) / torch.sum(self._pdf, dim=-1))) / torch.tensor(np.sqrt(3))

    @property
    def discrete_sigma(self):
        return self.sigma(torch.linspace(0.0, 1.0, self.num_sigma))

    def sigma(self, t):
        return torch.log(t * torch.exp(torch.tensor(self.max_sigma)) + (1 - t) * torch.exp(torch.tensor(self.min_sigma)))

    def sigma_idx(self, sigma):
        return torch.sum(self.discrete_sigma[None, ...] <= sigma[..., None] + 1e-5, -1) - 1

    def t_to_idx(self, t):
        return self.sigma_idx(self.sigma(t)).tolist()

    def diffusion_coef(self, t):
        s = self.sigma(t)
        return torch.sqrt(2 * (torch.exp(torch.tensor(self.max_sigma)) - torch.exp(torch.tensor(self.min_sigma))) * s / torch.exp(s))

    def score(self, vec, t, eps=1e-6):
        omega = torch.linalg.norm(vec, dim=-1) + eps
        sn = self._score_norms[self.t_to_idx(t)]
        oi = torch.bucketize(omega, self.discrete_omega[:-1])
        return torch.gather(sn, 1, oi)[..., None] * vec / (omega[..., None] + eps)

    def score_scaling(self, t):
        return self._score_scaling[self.t_to_idx(t)]

    def sample(self, t, n_samples, noise=None):
        x = noise['rot_axis'] if noise else torch.randn((*n_samples, 3))
        u = noise['rot_u'] if noise else torch.rand(n_samples)
        x = x / torch.linalg.norm(x, dim=-1, keepdim=True)
        om = self.discrete_omega[None, ...].expand(t.shape[0], -1)
        return x * torch_interp(u, self._cdf[self.t_to_idx(t)], om)[..., None]

    def reverse(self, rot_t, score_t, t, dt, z):
        g = self.diffusion_coef(t)[:, None, None]
        perturb = (g ** 2) * score_t * dt + g * torch.sqrt(dt) * z
        return quat_to_rotvec(quat_multiply(rotvec_to_quat(rot_t), rotvec_to_quat(perturb)))

    def forward_marginal(self, rot_0, t, noise=None):
        sampled = self.sample(t, rot_0.shape[:-1], noise)
        sc = self.score(sampled, t).reshape(rot_0.shape)
        return quat_to_rotvec(quat_multiply(rotvec_to_quat(rot_0), rotvec_to_quat(sampled))), sc


class OracleR3:
    def __init__(self, conf):
        self.min_b, self.max_b, self.cs = conf['min_b'], conf['max_b'], conf['coordinate_scaling']

    def scale(self, x):
        return x * torch.tensor(self.cs)

    def unscale(self, x):
        return x / torch.tensor(self.cs)

    def b_t(self, t):
        return torch.tensor(self.min_b) + t * torch.tensor(self.max_b - self.min_b)

    def marginal_b_t(self, t):
        return t * torch.tensor(self.min_b) + (1 / 2) * (t ** 2) * torch.tensor(self.max_b - self.min_b)

    def conditional_var(self, t):
        return 1 - torch.exp(-self.marginal_b_t(t))

    def score(self, x_t, x_0, t, scale=False):
        if scale:
            x_t, x_0 = self.scale(x_t), self.scale(x_0)
        t = t[:, None, None]
        return -(x_t - torch.exp(-1 / 2 * self.marginal_b_t(t)) * x_0) / self.conditional_var(t)

    def score_scaling(self, t):
        return 1 / torch.sqrt(self.conditional_var(t))

    def reverse(self, x_t, score_t, t, dt, z, center=True):
        x_t = self.scale(x_t)
        g = torch.sqrt(self.b_t(t))[:, None, None]
        f = -1 / 2 * self.b_t(t)[:, None, None] * x_t
        perturb = (f - g ** 2 * score_t) * dt + g * dt * z
        x1 = x_t - perturb
        if center:
            mask = torch.ones(x_t.shape[:-1])
            com = torch.sum(x1, dim=-2) / torch.sum(mask, dim=-1, keepdim=True)
            x1 = x1 - com[..., None, :]
        return self.unscale(x1)

    def forward_marginal(self, x_0, t, noise=None):
        x_0 = self.scale(x_0)
        lmc = (-0.5 * self.marginal_b_t(t)).view(-1, *([1] * (x_0.dim() - 1)))
        mean = torch.exp(lmc) * x_0
        std = torch.sqrt(1.0 - torch.exp(2.0 * lmc))
        z = noise['trans_z'] if noise else torch.randn(x_0.shape)
        x_t = mean + std * z
        return self.unscale(x_t), self.score(x_t, x_0, t)


def poisson_icdf(lam, u, cap=64):
    """Poisson(lam) count as a pure function of one uniform u in (0,1) - restatement of abx_amd/csrc/diffuser.hip::poisson_icdf
    (the build's device sampler for discrete_diffuser.py:182-183; the reference draws torch.poisson, whose stream cannot be
    reproduced): smallest k with u <= cdf(k), fp32 pmf recurrence p_k = p_{k-1} * (lam / k) from p_0 = float32(exp(-float64(lam))),
    stop when the fp32 cdf no longer grows past the mode, cap at 64.  lam, u: float32 arrays of one shape -> float32 counts."""
    lam = np.asarray(lam, dtype=np.float32)
    u = np.asarray(u, dtype=np.float32)
    pk = np.exp(-lam.astype(np.float64)).astype(np.float32)
    cdf = pk.copy()
    kk = np.zeros(lam.shape, dtype=np.int32)
    active = u > cdf
    k = 0
    while active.any() and k < cap:
        k += 1
        kk = np.where(active, k, kk)
        pk = np.where(active, pk * (lam / np.float32(k)), pk).astype(np.float32)
        nc = (cdf + pk).astype(np.float32)
        stalled = active & (nc == cdf) & (np.float32(k) > lam)
        cdf = np.where(active & ~stalled, nc, cdf)
        active = active & ~stalled & (u > cdf)
    return kk.astype(np.float32)


class OracleSeq:
    def __init__(self, conf, eigh=False):
        """eigh=True follows the reference's route to q_t0 literally (fp32 eigh factors of the rate matrix,
        discrete_diffuser.py:15-26,53-67): used to pin `reverse_rates` against the reference-recorded Poisson rates at 1e-6.  The
        default is the closed form the HIP kernel evaluates; the reference's fp32 eigh factors carry up to 3.3e-5 relative error
        on the off-diagonal entries at small t (measured against fp64: tests/test_oracle_golden.py), the closed form 1.4e-7."""
        self.K = 20
        self.rate_const = conf['rate_const']
        r = self.rate_const * torch.ones(self.K, self.K)
        r = r - torch.diag(torch.diag(r))
        r = r - torch.diag(torch.sum(r, dim=1))
        self.rate_matrix = r.float()
        self.eigh = eigh
        if eigh:
            ev, evec = torch.linalg.eigh(r)
            self.eigvals, self.eigvecs = ev.float(), evec.float()

    def transition(self, t):
        """Closed form of V exp(lambda t) V^T for the uniform-rate generator (discrete_diffuser.py:53-67):
        exp(-K r t) I + (1 - exp(-K r t))/K, entries < 1e-8 -> 0."""
        t = t.float()
        if self.eigh:
            K = self.K
            q = self.eigvecs.reshape(1, K, K) @ torch.diag_embed(torch.exp(self.eigvals.reshape(1, K) * t.reshape(-1, 1))) @ \
                self.eigvecs.T.reshape(1, K, K)
        else:
            e = torch.exp(-self.K * self.rate_const * t)[:, None, None]
            q = e * torch.eye(self.K)[None] + (1 - e) / self.K
        q = torch.where(q < 1e-8, torch.zeros_like(q), q)
        return q

    def reverse_rates(self, x_t, logits, t):
        B, L = x_t.shape
        x_t = torch.clamp(x_t, 0, self.K - 1).long()
        p0t = F.softmax(logits, dim=2)
        qt0 = self.transition(t * torch.ones((B,)))
        denom = torch.gather(qt0.transpose(1, 2), 1, x_t[..., None].expand(B, L, self.K)) + torch.tensor(1e-9)
        fwd = self.rate_matrix.t()[x_t]                                           # rate[s, x_t]
        rates = fwd * ((p0t / denom) @ qt0)
        return rates.scatter(2, x_t[..., None], 0.0), x_t

    def reverse(self, x_t, logits, t, dt, jumps=None, u_jumps=None):
        """jumps: recorded Poisson draws; u_jumps: uniforms for the inverse-cdf sampler the HIP kernel uses (poisson_icdf)."""
        rates, x_t = self.reverse_rates(x_t, logits, t)
        if jumps is None and u_jumps is not None:
            jumps = torch.from_numpy(poisson_icdf((rates * dt).numpy(), u_jumps.numpy()))
        if jumps is None:
            jumps = torch.poisson(rates * dt)
        diffs = torch.arange(self.K).view(1, 1, self.K) - x_t[..., None]
        xp = x_t + torch.sum(jumps * diffs, dim=2)
        return torch.clamp(xp, 0, self.K - 1).to(torch.int32)

    def forward_marginal(self, x_0, t, noise=None):
        """discrete_diffuser.py:72-127: x_t ~ Categorical(q_t0[x_0]) plus ONE extra jump per sample (x_tilde)."""
        B, L = x_0.shape
        qt0 = self.transition(t)
        x_0 = torch.clamp(x_0, 0, self.K - 1).long()
        rows = torch.gather(qt0, 1, x_0[..., None].expand(B, L, self.K))
        x_t = noise['seq_xt'] if noise else torch.distributions.Categorical(rows).sample()
        rate = self.rate_matrix[x_t.long()].clone()                                    # rate[x_t, :]
        rate.scatter_(2, x_t.long()[..., None], 0.0)
        dims = noise['seq_dim'] if noise else torch.distributions.Categorical(rate.sum(-1)).sample()
        bi = torch.arange(B)
        newv = noise['seq_new'] if noise else torch.distributions.Categorical(rate[bi, dims]).sample()
        x_tilde = x_t.clone()
        x_tilde[bi, dims] = newv
        return x_tilde


def _mask_merge(x_diff, x_fixed, m):
    return m * x_diff + (1 - m) * x_fixed


class OracleDiffuser:
    """FullDiffuser facade (full_diffuser.py:28-290).  `noise` dicts inject recorded draws (parity mode)."""

    def __init__(self, diff_conf, tables=None):
        self._diff_conf = diff_conf
        self.so3 = OracleSO3(diff_conf['so3'], tables)
        self.r3 = OracleR3(diff_conf['r3'])
        self.seq = OracleSeq(diff_conf['seq'])

    def score_scaling(self, t):
        return self.so3.score_scaling(t), self.r3.score_scaling(t)

    def calc_quat_score(self, quat_t, quat_0, t):
        return self.so3.score(quat_to_rotvec(quat_multiply(invert_quat(quat_0), quat_t)), t)

    def calc_trans_score(self, trans_t, trans_0, t):
        return self.r3.score(trans_t, trans_0, t, scale=True)

    def reverse(self, rigid_t, seq_t, rot_score, trans_score, logits_t, t, dt, diffuse_mask, noise=None,
                center=True, noise_scale=1.0):
        """noise: {'z_rot','z_trans' (B,L,3) f32, 'jumps' (B,L,20) f32 | 'u_jumps' (B,L,20) uniforms}; drawn in that order if absent."""
        trans_t, rot_t = rigid_t[..., 4:], quat_to_rotvec(rigid_t[..., :4])
        z_rot = noise['z_rot'] if noise else torch.randn(rot_score.shape)
        rot1 = self.so3.reverse(rot_t, rot_score, t, dt, noise_scale * z_rot)
        z_tr = noise['z_trans'] if noise else torch.randn(trans_score.shape)
        tr1 = self.r3.reverse(trans_t, trans_score, t, dt, noise_scale * z_tr, center)
        seq1 = self.seq.reverse(seq_t, logits_t, t, dt, noise.get('jumps') if noise else None, noise.get('u_jumps') if noise else None)
        m = diffuse_mask
        tr1 = _mask_merge(tr1, trans_t, m[..., None])
        rot1 = _mask_merge(rot1, rot_t, m[..., None])
        seq1 = _mask_merge(seq1, seq_t, m)
        return torch.cat([rotvec_to_quat(rot1), tr1], dim=-1), seq1

    def forward_marginal(self, rigids_0, seq_0, t, diffuse_mask=None, noise=None):
        """full_diffuser.py:57-126 (optimize mode: noise the ground truth to time t)."""
        trans_0, rot_0 = rigids_0[..., 4:], quat_to_rotvec(rigids_0[..., :4])
        rot_t, rot_score = self.so3.forward_marginal(rot_0, t, noise)
        trans_t, trans_score = self.r3.forward_marginal(trans_0, t, noise)
        seq_t = self.seq.forward_marginal(seq_0, t, noise)
        if diffuse_mask is not None:
            m = diffuse_mask
            rot_t = _mask_merge(rot_t, rot_0, m[..., None])
            trans_t = _mask_merge(trans_t, trans_0, m[..., None])
            trans_score = _mask_merge(trans_score, torch.zeros_like(trans_score), m[..., None])
            rot_score = _mask_merge(rot_score, torch.zeros_like(rot_score), m[..., None])
            seq_t = _mask_merge(seq_t, seq_0, m)
        rs, ts = self.score_scaling(t)
        return {'rigids_t': torch.cat([rotvec_to_quat(rot_t), trans_t], dim=-1), 'trans_score': trans_score,
                'rot_score': rot_score, 'trans_score_scaling': ts, 'rot_score_scaling': rs, 'seq_t': seq_t}

    def sample_ref(self, n_samples, impute_rigids, impute_seq, diffuse_mask, noise=None):
        """full_diffuser.py:229-290; draw order randn(B,L,3), rand(B,L), randn(B,L,3), randint(B,L)."""
        B, L = n_samples
        tr_imp = self.r3.scale(impute_rigids[..., 4:])
        rot_imp = quat_to_rotvec(impute_rigids[..., :4])
        if noise is None:
            noise = dict(rot_axis=torch.randn(B, L, 3), rot_u=torch.rand(B, L), trans_z=torch.randn(B, L, 3),
                         seq=torch.randint(0, 20, (B, L)))
        rot_ref = self.so3.sample(torch.ones(B), (B, L), noise)
        m = diffuse_mask
        rot_ref = _mask_merge(rot_ref, rot_imp, m[..., None])
        tr_ref = self.r3.unscale(_mask_merge(noise['trans_z'], tr_imp, m[..., None]))
        seq_ref = _mask_merge(noise['seq'], impute_seq, m)
        return {'rigids_t': torch.cat([rotvec_to_quat(rot_ref), tr_ref], dim=-1), 'seq_t': seq_ref}


# --------------------------------------------------------------------------------------------------
# sampling loop (inference.py:166-273)
# --------------------------------------------------------------------------------------------------
def set_t_feats(batch, diffuser, t, ones):
    batch['t'] = t * ones
    rs, ts = diffuser.score_scaling(batch['t'])
    batch['rot_score_scaling'] = rs * ones
    batch['trans_score_scaling'] = ts * ones
    return batch


def sample_fn(p, data_init, cfg, diffuser, mode='design', num_t=100, min_t=0.01, noise_fn=None, static=None,
              eps=1e-8, record=None):
    """Reverse-time driver.  noise_fn(step_index, shape_info) -> recorded noise dict or None.
    Returns the list of per-step dicts (all steps; caller keeps the last unless mode == 'trajectory')."""
    batch = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in data_init.items()}
    diffuse_mask = (1 - batch['fixed_mask']) * batch['atom14_gt_exists'][..., 0]
    Lab = batch['anchor_flag'].shape[1]
    B = batch['rigids_t'].shape[0]
    ones = torch.ones(B, dtype=torch.float32)
    steps = np.linspace(min_t, 1.0, num_t)[::-1]
    dt = torch.tensor(1 / num_t)
    if mode == 'optimize':
        opt = batch['t'][0].cpu().numpy()
        if opt < 1.0:
            steps = steps[steps <= opt + eps]
    traj = []
    if static is None:
        static = static_embeddings(p, batch, cfg)
    if len(steps) > 0:
        batch = set_t_feats(batch, diffuser, steps[0], ones)
        out = score_network(p, batch, cfg, diffuser, static)
        batch.update(get_prev(batch, out, cfg))
    for k, t in enumerate(steps):
        if t > min_t:
            t_ = torch.tile(torch.tensor(t), (B,))
            batch = set_t_feats(batch, diffuser, t_, ones)
            out = score_network(p, batch, cfg, diffuser, static)
            f = out['heads']['folding']
            batch.update(get_prev(batch, out, cfg))
            rigids_t, seq_t = diffuser.reverse(
                rigid_t=batch['rigids_t'], seq_t=batch['seq_t'], rot_score=f['rot_score'],
                trans_score=f['trans_score'], logits_t=out['heads']['sequence_module']['logits'],
                diffuse_mask=diffuse_mask, t=t_, dt=dt, noise=noise_fn(k) if noise_fn else None)
        else:
            out = score_network(p, batch, cfg, diffuser, static)
            rigids_t = out['heads']['folding']['rigids']
            seq_t = out['heads']['sequence_module']['seq_0']
        batch['rigids_t'] = rigids_t
        batch['seq_t'] = seq_t
        pl = out['heads']['predicted_lddt']['pLDDT']
        pl = torch.sum(pl * diffuse_mask, dim=1) / torch.sum(diffuse_mask, dim=1)
        traj.append({'seq': torch.clamp(seq_t[:, :Lab], min=0, max=19).long(),
                     'atom14_results': out['heads']['folding']['final_atom14_positions'][:, :Lab],
                     'pLDDT': torch.tile(pl[:, None], (1, Lab)), 'time': t,
                     'rigids_t': rigids_t, 'seq_t': seq_t})
        if record is not None:
            record(k, t, batch, out)
    return traj


# ---------------------------------------------------------------------------------------------------------------------
# Guidance terms: NOT part of the reference's sampler; restated from the violation material it ships
# (abx/common/residue_constants.py:381-386 van_der_waals_radius, :475-476 between_res_bond_length_c_n / stddev,
# config/config_model.json:116,213-214 clash_overlap_tolerance 1.5 / between_chain_factor 0.2 / violation_tolerance_factor 12,
# eval/metric_scripts/cal_vio.py:29-74 C-N bond term).  Differentiable torch (use float64 + autograd as the checker of the
# analytic gradients of csrc/guidance.hip).
# ---------------------------------------------------------------------------------------------------------------------
_VDW = {'C': 1.7, 'N': 1.55, 'O': 1.52, 'S': 1.8}


def vdw_radius_table():
    from abx_amd import residue_constants as rc
    t = torch.zeros(21, 14, dtype=torch.float64)
    for i, r in enumerate(rc.restypes):
        for j, name in enumerate(rc.restype_name_to_atom14_names[rc.restype_1to3[r]]):
            if name:
                t[i, j] = _VDW[name[0]]
    return t


def peptide_violation_terms(atom14, atom_mask, aatype, chain_id, residx=None, tolerance_factor=12.0):
    """Restatement of eval/metric_scripts/cal_vio.py:29-110 (between_residue_bond_loss): per residue pair (i, i+1) the flat-bottom
    violations of the C-N bond length and of cos(CA_i,C_i,N_i+1), cos(C_i,N_i+1,CA_i+1), their masks and hard-violation masks.
    residx=None is the reference's rule (neighbours are linked when they share a chain id); with residue numbers a link also needs
    residx[i+1] == residx[i] + 1 (the build's guidance default).  Pinned by tests/golden/vio_pdb.npz."""
    x = atom14
    dt = x.dtype
    m = atom_mask.to(dt)
    ca, c, n, ca2 = x[:, :-1, 1], x[:, :-1, 2], x[:, 1:, 0], x[:, 1:, 1]
    m_ca, m_c, m_n, m_ca2 = m[:, :-1, 1], m[:, :-1, 2], m[:, 1:, 0], m[:, 1:, 1]
    link = chain_id[:, 1:] == chain_id[:, :-1]
    if residx is not None:
        link = link & (residx[:, 1:] == residx[:, :-1] + 1)
    link = link.to(dt)
    pro = (aatype[:, 1:] == 14).to(dt)
    l0 = (1 - pro) * 1.329 + pro * 1.341
    sd = (1 - pro) * 0.014 + pro * 0.016
    dist = torch.sqrt(1e-6 + ((c - n) ** 2).sum(-1))
    err_b = torch.sqrt(1e-6 + (dist - l0) ** 2)
    unit = lambda v: v / torch.sqrt(torch.clamp((v ** 2).sum(-1, keepdim=True), min=1e-12))      # abx/model/utils.py:12-14
    c_ca, c_n, n_ca = unit(ca - c), unit(n - c), unit(ca2 - n)
    err_a1 = torch.sqrt(1e-6 + ((c_ca * c_n).sum(-1) - (-0.4473)) ** 2)                           # residue_constants.py:480
    err_a2 = torch.sqrt(1e-6 + (((-c_n) * n_ca).sum(-1) - (-0.5203)) ** 2)                        # residue_constants.py:479
    t = tolerance_factor
    return {
        'c_n_loss_per_residue': torch.relu(err_b - t * sd), 'c_n_mask': m_c * m_n * link,
        'ca_c_n_loss_per_residue': torch.relu(err_a1 - t * 0.0311), 'ca_c_n_mask': m_ca * m_c * m_n * link,
        'c_n_ca_loss_per_residue': torch.relu(err_a2 - t * 0.0353), 'c_n_ca_mask': m_c * m_n * m_ca2 * link,
        'c_n_violation_mask': m_c * m_n * link * (err_b > t * sd),
        'ca_c_n_violation_mask': m_ca * m_c * m_n * link * (err_a1 > t * 0.0311),
        'c_n_ca_violation_mask': m_c * m_n * m_ca2 * link * (err_a2 > t * 0.0353),
        'has_no_gap_mask': link,
    }


def violation_energy(atom14, atom_mask, aatype, chain_id, overlap_tolerance=1.5, between_chain_factor=0.2,
                     bond_tolerance_factor=12.0, w_clash=1.0, w_bond=1.0, w_angle=1.0, residx=None):
    """atom14 (B,L,14,3), atom_mask (B,L,14) bool, aatype (B,L), chain_id (B,L) -> (E_clash, E_bond, E_angle), each (B,): the
    un-normalised sums that abx_clash_grad returns (csrc/guidance.hip)."""
    B, L = aatype.shape
    x = atom14.reshape(B, L * 14, 3)
    m = atom_mask.reshape(B, L * 14).bool()
    aa = torch.clamp(aatype.long(), 0, 20)
    rad = vdw_radius_table().to(x.dtype)[aa].reshape(B, L * 14)
    res = torch.arange(L).repeat_interleave(14)[None].expand(B, -1)
    slot = torch.arange(14).repeat(L)[None].expand(B, -1)
    ch = chain_id.long().repeat_interleave(14, dim=1)
    sg = ((aa == 4).repeat_interleave(14, dim=1)) & (slot == 5)
    d = torch.sqrt(1e-10 + ((x[:, :, None] - x[:, None]) ** 2).sum(-1))
    pair = m[:, :, None] & m[:, None] & (res[:, :, None] < res[:, None])              # every pair of different residues once
    same_chain = ch[:, :, None] == ch[:, None]
    terms = peptide_violation_terms(atom14, atom_mask, aatype, chain_id, residx, bond_tolerance_factor)
    linked = torch.zeros(B, L, dtype=torch.bool)                                      # residue is linked to its array predecessor
    linked[:, 1:] = terms['has_no_gap_mask'].bool()
    lk = linked.repeat_interleave(14, dim=1)
    bonded = lk[:, None] & (res[:, None] == res[:, :, None] + 1) & (slot[:, :, None] == 2) & (slot[:, None] == 0)
    pair = pair & ~bonded & ~(sg[:, :, None] & sg[:, None])
    w = torch.where(same_chain, torch.ones_like(d), torch.full_like(d, between_chain_factor))
    ov = torch.relu(rad[:, :, None] + rad[:, None] - overlap_tolerance - d)
    e_clash = w_clash * (w * ov * pair).sum((1, 2))
    e_bond = w_bond * (terms['c_n_loss_per_residue'] * terms['c_n_mask']).sum(1)
    e_angle = w_angle * ((terms['ca_c_n_loss_per_residue'] * terms['ca_c_n_mask']).sum(1) +
                         (terms['c_n_ca_loss_per_residue'] * terms['c_n_ca_mask']).sum(1))
    return e_clash, e_bond, e_angle
