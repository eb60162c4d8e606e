"""PDB output of the sampling path: the reference's `postprocess_trajectory` / `postprocess_one`
(inference.py:127-161) and `save_pdb` / `make_chain` (abx/data/utils.py:200-263) without Biopython, plus an asynchronous
writer so that a per-step trajectory dump (BASELINE config 5) never stalls the GPU.

The reference builds Bio.PDB Atom/Residue/Chain objects and calls `PDBIO.save`; Biopython is not available in this image, so
the record layout below restates PDBIO's ATOM / TER / END format (Bio/PDB/PDBIO.py `_ATOM_FORMAT_STRING`,
`_TER_FORMAT_STRING`: atoms renumbered from 1, the TER record takes the next serial WITHOUT consuming it, element = first
letter of the atom name, occupancy 1.00, B-factor = pLDDT).  PINNED on the two example complexes the reference ships
(test_data/*.pdb, themselves written by PDBIO: tests/golden/pdb/): every ATOM / TER / END record of those files is reproduced
character for character from its parsed fields (tests/test_pdb_writer.py::test_record_layout_matches_the_shipped_pdbio_files).

This is host-side I/O: plain Python / numpy, no kernel involved."""
import os
import queue
import threading

import numpy as np
import torch

from .. import residue_constants as rc

_ATOM_FMT = "%s%5i %-4s%c%3s %c%4i%c   %8.3f%8.3f%8.3f%s%6.2f      %4s%2s%2s\n"
_TER_FMT = "TER   %5i      %3s %c%4i%c" + " " * 54 + "\n"          # PDBIO's TER record is 81 columns wide (pinned on test_data/*.pdb)


def index_to_str_seq(idx):
    """abx/data/utils.py index_to_str_seq: restype index -> one-letter string ('X' beyond the 20 standard types)."""
    return ''.join(rc.restypes_with_x[int(i)] if 0 <= int(i) < len(rc.restypes_with_x) else 'X' for i in idx)


def _atom_line(serial, name, resname, chain_id, resseq, xyz, bfactor):
    element = name[:1].upper()
    # PDBIO._get_atom_line: names shorter than 4 characters whose element has one letter start in column 14
    full = name if len(name) == 4 else ' ' + name
    return _ATOM_FMT % ('ATOM  ', serial, full, ' ', resname, chain_id, resseq, ' ', xyz[0], xyz[1], xyz[2], '%6.2f' % 1.0,
                        bfactor, '', element.rjust(2), '  ')


def _chain_lines(lines, serial, str_seq, coords, chain_id, bfactors, mask=None):
    """make_chain (abx/data/utils.py:200-232): residues numbered from 1, atom14 order of the residue type, '' slots skipped."""
    last = None
    for i, aa in enumerate(str_seq):
        if mask is not None and not mask[i]:
            continue
        resname = rc.restype_1to3.get(aa, 'UNK')
        names = rc.restype_name_to_atom14_names[resname]
        for j, atom_name in enumerate(names):
            if atom_name == '':
                continue
            lines.append(_atom_line(serial, atom_name, resname, chain_id, i + 1, coords[i, j], float(bfactors[i, j])))
            serial += 1
        last = (resname, i + 1)
    if last is not None:
        lines.append(_TER_FMT % (serial, last[0], chain_id, last[1], ' '))
    return serial


def format_pdb(str_heavy_seq, heavy_chain, str_light_seq, light_chain, coord, pLDDT, antigen_data=None):
    """save_pdb (abx/data/utils.py:234-263) as text.  coord (Lab,14,3), pLDDT (Lab,), antigen_data = dict(antigen_str_seq,
    antigen_coords (Lag,14,3), antigen_coord_mask (Lag,14), antigen_chain_ids (Lag,) with values 2.., antigen_chains)."""
    coord = np.asarray(coord, dtype=np.float64)
    pLDDT = np.asarray(pLDDT, dtype=np.float64)
    nh, nl = len(str_heavy_seq), len(str_light_seq)
    assert nh + nl == coord.shape[0]
    bf = np.repeat(pLDDT[..., None], rc.atom_type_num, axis=-1)
    lines, serial = [], 1
    serial = _chain_lines(lines, serial, str_heavy_seq, coord[:nh], heavy_chain, bf[:nh])
    serial = _chain_lines(lines, serial, str_light_seq, coord[nh:], light_chain, bf[nh:])
    if antigen_data is not None:
        ids = np.asarray(antigen_data['antigen_chain_ids'])
        seq = antigen_data['antigen_str_seq']
        ac = np.asarray(antigen_data['antigen_coords'], dtype=np.float64)
        am = np.asarray(antigen_data['antigen_coord_mask'])
        start = 0
        for i, chain in enumerate(antigen_data['antigen_chains']):
            n = int((ids == i + 2).sum())
            bfa = np.full((n, rc.atom_type_num), pLDDT[0])           # the reference tags antigen atoms with pLDDT[0]
            serial = _chain_lines(lines, serial, seq[start:start + n], ac[start:start + n], chain, bfa,
                                  am[start:start + n, rc.atom_order['CA']])
            start += n
    lines.append('END   \n')
    return ''.join(lines)


def save_pdb(str_heavy_seq, heavy_chain, str_light_seq, light_chain, coord, pdb_path, pLDDT, antigen_data=None):
    with open(pdb_path, 'w') as f:
        f.write(format_pdb(str_heavy_seq, heavy_chain, str_light_seq, light_chain, coord, pLDDT, antigen_data))


def _one_record_files(meta, rec, output_dir, multi):
    """postprocess_trajectory / postprocess_one (inference.py:127-161) for one trajectory element already on the host."""
    files = []
    time = rec['time'] if multi else None
    for i, name in enumerate(meta['name']):
        nh, nl = len(meta['str_heavy_seq'][i]), len(meta['str_light_seq'][i])
        seq = rec['seq'][i]
        parts = name.split('_')
        antigen = None
        if meta.get('antigen_origin_str_seq') is not None:
            antigen = {'antigen_str_seq': meta['antigen_origin_str_seq'][i],
                       'antigen_coords': meta['antigen_origin_atom14_gt_positions'][i],
                       'antigen_coord_mask': meta['antigen_origin_atom14_gt_exists'][i],
                       'antigen_chain_ids': meta['antigen_origin_chain_ids'][i],
                       # design.py:152 splits the antigen chain ids on '|' ('6qd7_X_Z_F|E'); inference.py:149 takes the characters
                       'antigen_chains': parts[-1].split('|') if '|' in parts[-1] else list(parts[-1])}
        out_i = output_dir
        if meta.get('subdir') is not None:              # inference.py's layout: one directory per sample (<k:04d>/<name>.pdb)
            out_i = os.path.join(output_dir, meta['subdir'][i])
            os.makedirs(out_i, exist_ok=True)
        path = f'{out_i}/{name}@{time:.4f}.pdb' if time else f'{out_i}/{name}.pdb'
        save_pdb(index_to_str_seq(seq[:nh]), parts[1], index_to_str_seq(seq[nh:nh + nl]), parts[2],
                 rec['atom14_results'][i, :nh + nl], path, rec['pLDDT'][i], antigen)
        files.append(path)
    return files


def postprocess_trajectory(meta, traj, output_dir):
    """Synchronous form: traj = list of records as returned by sampler.sample_fn.  meta: name / str_heavy_seq / str_light_seq
    lists (one entry per sample) and optionally the antigen_origin_* fields of the reference batch."""
    os.makedirs(output_dir, exist_ok=True)
    files = []
    for rec in traj:
        host = {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else v) for k, v in rec.items()
                if k in ('seq', 'atom14_results', 'pLDDT', 'time')}
        files += _one_record_files(meta, host, output_dir, len(traj) > 1)
    return files


class TrajectoryWriter:
    """Asynchronous PDB dump.  `submit(rec)` enqueues device->host copies of the three result tensors on a side stream into
    pinned buffers and returns immediately; a worker thread waits for the copy event, formats and writes the files.  Use as the
    `on_record` sink of a sampling loop (trajectory mode writes `{name}@{t:.4f}.pdb` for every step)."""

    def __init__(self, meta, output_dir, multi=True, max_pending=8):
        self.meta, self.output_dir, self.multi = meta, output_dir, multi
        os.makedirs(output_dir, exist_ok=True)
        self.q = queue.Queue(maxsize=max_pending)
        self.files, self.error = [], None
        self.stream = torch.cuda.Stream() if torch.cuda.is_available() else None
        # The worker formats PDB text in Python and holds the GIL while it does; the sampling thread gives the GIL up at every host
        # synchronisation (the range word of a network pass, model/abx.py) and would then wait a full switch interval (5 ms) to get it
        # back with the GPU idle: a short interval while a writer is alive (ABX_WRITER_SWITCH_INTERVAL seconds; 0 keeps the default)
        import sys
        self._switch = sys.getswitchinterval()
        want = float(os.environ.get('ABX_WRITER_SWITCH_INTERVAL', '2e-4'))
        if want > 0:
            sys.setswitchinterval(min(self._switch, want))
        self.worker = threading.Thread(target=self._run, daemon=True)
        self.worker.start()

    def submit(self, rec):
        if self.error is not None:
            raise self.error
        host, event = {'time': rec['time']}, None
        if self.stream is not None and rec['seq'].is_cuda:
            self.stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.stream):
                for k in ('seq', 'atom14_results', 'pLDDT'):
                    src = rec[k]
                    src.record_stream(self.stream)
                    buf = torch.empty(src.shape, dtype=src.dtype, pin_memory=True)
                    buf.copy_(src, non_blocking=True)
                    host[k] = buf
                event = torch.cuda.Event()
                event.record(self.stream)
        else:
            for k in ('seq', 'atom14_results', 'pLDDT'):
                host[k] = rec[k].detach().cpu().clone()
        self.q.put((host, event))           # blocks only when max_pending records are waiting for the disk

    def _run(self):
        while True:
            item = self.q.get()
            if item is None:
                return
            try:
                host, event = item
                if event is not None:
                    event.synchronize()
                rec = {k: (v.numpy() if torch.is_tensor(v) else v) for k, v in host.items()}
                self.files += _one_record_files(self.meta, rec, self.output_dir, self.multi)
            except Exception as e:          # surfaced by the next submit() / close()
                self.error = e

    def close(self):
        self.q.put(None)
        self.worker.join()
        import sys
        sys.setswitchinterval(self._switch)
        if self.error is not None:
            raise self.error
        return self.files
