"""Design-quality metrics of the reference's offline evaluation (abx/common/ab_utils.py:124-167 `calc_ab_metrics`,
abx/utils.py:444-465 `kabsch_numpy`): Kabsch-aligned C-alpha RMSD and amino-acid recovery per CDR.  numpy, host side."""
from collections import OrderedDict

import numpy as np

_SCHEMA = {'cdr1': 1, 'cdr2': 3, 'cdr3': 5}


def kabsch(X, Y):
    """Kabsch alignment of X onto Y, both (3, N): returns the centred + rotated X and the centred Y (abx/utils.py:444-465)."""
    X_ = X - X.mean(axis=-1, keepdims=True)
    Y_ = Y - Y.mean(axis=-1, keepdims=True)
    C = np.dot(X_, Y_.transpose())
    V, S, W = np.linalg.svd(C)
    if (np.linalg.det(V) * np.linalg.det(W)) < 0.0:
        S[-1] = -S[-1]
        V[:, -1] = -V[:, -1]
    U = np.dot(V, W)
    return np.dot(X_.T, U).T, Y_


def _rmsd(A, B):
    return float(np.sqrt(np.mean(np.sum(np.square(A - B), axis=0))))


def calc_ab_metrics(gt_coord, pred_coord, cdr_def, gt_str_seq=None, pred_str_seq=None):
    """gt_coord / pred_coord (N, 3) C-alpha coordinates of the antibody, cdr_def (N,) IMGT region codes (H: 0..6, L: 7..13).
    -> OrderedDict {heavy|light}_cdr{1,2,3}_{AAR,RMSD} (+ *_cdr3_Loop_* on residues [4:-2] of CDR-H3), as ab_utils.py:124-167."""
    gt_al, pred_al = kabsch(np.transpose(gt_coord, [1, 0]), np.transpose(pred_coord, [1, 0]))
    cdr_def = np.asarray(cdr_def)
    names = {v: 'heavy_' + k for k, v in _SCHEMA.items()}
    names.update({v + 7: 'light_' + k for k, v in _SCHEMA.items()})
    ret = OrderedDict()
    for k, v in names.items():
        idx = cdr_def == k
        gt, pred = gt_al[:, idx], pred_al[:, idx]
        if gt_str_seq is not None:
            gs = ''.join(c for c, keep in zip(gt_str_seq, idx) if keep)
            ps = ''.join(c for c, keep in zip(pred_str_seq, idx) if keep)
            ret[v + '_AAR'] = float(np.mean([a == b for a, b in zip(gs, ps)]))
            if k == 5:
                ret[v + '_Loop_AAR'] = float(np.mean([a == b for a, b in zip(gs[4:-2], ps[4:-2])]))
        ret[v + '_RMSD'] = _rmsd(gt, pred)
        if k == 5:
            ret[v + '_Loop_RMSD'] = _rmsd(gt[:, 4:-2], pred[:, 4:-2])
    return ret
