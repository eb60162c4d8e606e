"""Plain-text PDB reader for the input side of the sampling path.

Replaces, without Biopython, what the reference gets from `Bio.PDB.PDBParser().get_structure(...)[0]` and
`make_chain_feature` (abx/preprocess/make_ab_data_from_mmcif.py:49-74): per chain, the residues with a standard amino-acid name
as a one-letter sequence plus atom14 coordinates and mask.  Behaviour of the Biopython parser that matters here and is kept:
first MODEL only; chains in order of first appearance; a residue is identified by (hetero flag, number, insertion code);
alternate locations: the atom with the highest occupancy wins (the first one on ties); a repeated atom name inside a residue
keeps its first occurrence; hydrogens and atoms outside the residue type's atom14 set are ignored by the feature builder.
PARITY NOTE: Biopython is not available in this image, so this reader is pinned by the two example complexes the reference
ships (sequence lengths, known CDR-H3 strings, atom counts) and not by a run of the reference's parser.
"""
from collections import OrderedDict

import numpy as np

from .. import residue_constants as rc

_THREE_TO_ONE = {v: k for k, v in rc.restype_1to3.items()}


class Residue:
    __slots__ = ('resname', 'resseq', 'icode', 'het', 'atoms', '_occ', '_alt', '_last_name')

    def __init__(self, resname, resseq, icode, het):
        self.resname, self.resseq, self.icode, self.het = resname, resseq, icode, het
        self.atoms = OrderedDict()          # name -> xyz
        self._occ = {}                      # name -> occupancy of the selected alternate location
        self._alt = None                    # other residue NAMES at this position (point-mutation disorder): name -> Residue
        self._last_name = resname           # residue name of the last atom line read for this position


class PdbFormatError(ValueError):
    pass


def read_pdb(path):
    """-> OrderedDict chain_id -> list[Residue] (first model)."""
    chains = OrderedDict()
    index = {}
    with open(path) as f:
        for lineno, line in enumerate(f, 1):
            rec = line[:6]
            if rec == 'ENDMDL':
                break
            if rec not in ('ATOM  ', 'HETATM'):
                continue
            try:
                name = line[12:16].strip()
                altloc = line[16]
                resname = line[17:20].strip()
                chain_id = line[21]
                resseq = int(line[22:26])
                icode = line[26]
                xyz = (float(line[30:38]), float(line[38:46]), float(line[46:54]))
            except (ValueError, IndexError) as e:
                raise PdbFormatError(f'{path}:{lineno}: malformed {rec.strip()} record ({e}): {line.rstrip()!r}') from None
            try:
                occ = float(line[54:60])
            except ValueError:
                occ = 1.0
            het = ' ' if rec == 'ATOM  ' else ('W' if resname in ('HOH', 'WAT') else 'H_' + resname)
            key = (chain_id, het, resseq, icode)
            res = index.get(key)
            if res is None:
                res = Residue(resname, resseq, icode, het)
                index[key] = res
                chains.setdefault(chain_id, []).append(res)
            elif resname != res.resname:
                # point-mutation microheterogeneity: two residue NAMES share (het, number, insertion code).  Biopython keeps both in a
                # DisorderedResidue and every atom line SELECTS the child of its residue name (StructureBuilder.init_residue ->
                # disordered_add / disordered_select), so the name of the LAST atom line of the residue is the one its iterators
                # return; the alternatives are collected on the side and the loop below keeps that one
                if res._alt is None:
                    res._alt = OrderedDict()
                alt = res._alt.get(resname)
                if alt is None:
                    alt = res._alt[resname] = Residue(resname, resseq, icode, het)
                index[key]._last_name = resname
                res = alt
            else:
                res._last_name = resname
            if name in res.atoms:
                # alternate location of a known atom: keep the higher occupancy; a plain duplicate keeps the first
                if altloc != ' ' and occ > res._occ[name]:
                    res.atoms[name] = xyz
                    res._occ[name] = occ
                continue
            res.atoms[name] = xyz
            res._occ[name] = occ
    for cid, residues in chains.items():
        for i, r in enumerate(residues):
            if r._alt:
                best = r if r._last_name == r.resname else r._alt[r._last_name]
                best._alt = None
                residues[i] = best
    return chains


def chain_feature(residues):
    """make_chain_feature (make_ab_data_from_mmcif.py:49-74): residues with one of the 20 standard names ->
    dict(str_seq, coords (N,14,3) f32, coord_mask (N,14) bool)."""
    keep = [r for r in residues if r.resname in _THREE_TO_ONE]
    n = len(keep)
    coords = np.zeros((n, 14, 3), dtype=np.float32)
    mask = np.zeros((n, 14), dtype=bool)
    for i, r in enumerate(keep):
        names = rc.restype_name_to_atom14_names[r.resname]
        for aname, xyz in r.atoms.items():
            if aname == '' or aname not in names:
                continue
            j = names.index(aname)
            coords[i, j] = xyz
            mask[i, j] = True
    return dict(str_seq=''.join(_THREE_TO_ONE[r.resname] for r in keep), coords=coords, coord_mask=mask)
