"""Opt-in structural-violation guidance of the reverse process (BASELINE config 4 "guidance-gradient terms on"; SURVEY.md §8a
row G, §8f-4).  The reference samples WITHOUT guidance (its loop runs under no_grad), so this is an extension:
default off, and with `guidance=None` the sampler executes exactly the un-guided code path (bit-identical, tested).

Energy: the clash + peptide-bond + bond-angle violation terms of csrc/guidance.hip (`abx_clash_grad`), evaluated on the network's predicted
structure x0_hat (`final_atom14_positions`, frames = predicted rigids).  Reconstruction guidance: the scores handed to
`FullDiffuser.reverse` become
    trans_score -= scale_trans * dE/dt_i / coordinate_scaling         (the R^3 process runs on 0.1 x coordinates, r3_diffuser.py:27-40)
    rot_score   -= scale_rot   * R_i^T (sum_a (x_a - t_i) x dE/dx_a)  (the SO(3) step right-multiplies: body-frame tangent vector)
for diffused residues only (fixed residues are restored by the mask merge of `reverse` anyway)."""
import torch

from abx_amd import ops


def quat_to_rot(q):
    """(…,4) unit quaternion (w,x,y,z) -> (…,3,3)."""
    w, x, y, z = q.unbind(-1)
    return torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
                        2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                        2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], dim=-1).reshape(q.shape[:-1] + (3, 3))


class ViolationGuidance:
    def __init__(self, scale_trans=1.0, scale_rot=1.0, w_clash=1.0, w_bond=1.0, w_angle=1.0, overlap_tolerance=1.5,
                 between_chain_factor=0.2, bond_tolerance_factor=12.0, coordinate_scaling=0.1, link_by_residx=True):
        """link_by_residx: array neighbours are peptide-bonded only when they share a chain id AND have consecutive residue numbers
        (a cropped antigen patch keeps one chain id across its gaps); False = the chain-only rule of cal_vio.py:51."""
        self.scale_trans, self.scale_rot = float(scale_trans), float(scale_rot)
        self.kw = dict(w_clash=w_clash, w_bond=w_bond, w_angle=w_angle, overlap_tolerance=overlap_tolerance,
                       between_chain_factor=between_chain_factor, bond_tolerance_factor=bond_tolerance_factor)
        self.coordinate_scaling = float(coordinate_scaling)
        self.link_by_residx = bool(link_by_residx)
        self.last_energy = None             # (B, 3) [clash, bond, angle] of the most recent call (device tensor)

    def energy_and_grads(self, batch, out):
        f = out['heads']['folding']
        seq0 = out['heads']['sequence_module']['seq_0']
        exists = ops.atom14_mask_table(seq0.device)[torch.clamp(seq0, 0, 20)] & batch['mask'][..., None].bool()
        return ops.clash_grad(f['final_atom14_positions'], exists, seq0, batch['chain_id'], f['rigids'][..., 4:],
                              residx=batch['residx'] if self.link_by_residx else None, **self.kw)

    def __call__(self, batch, out, rot_score, trans_score, diffuse_mask):
        energy, _, g_t, g_r = self.energy_and_grads(batch, out)
        self.last_energy = energy
        R = quat_to_rot(out['heads']['folding']['rigids'][..., :4])
        body = torch.einsum('...ji,...j->...i', R, g_r)                  # R^T tau
        m = diffuse_mask.to(g_t.dtype)[..., None]
        rot = rot_score - (self.scale_rot * body * m).to(rot_score.dtype)
        trans = trans_score - (self.scale_trans / self.coordinate_scaling * g_t * m).to(trans_score.dtype)
        return rot, trans
