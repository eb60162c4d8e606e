"""Drop-in for the reference's `abx.model.abx` (abx/model/abx.py:17-104): `ScoreNetwork` and `get_prev` with the same
constructor, call signature, batch keys read / mutated in place, output dictionary and state_dict layout — computed by the
HIP kernels of libabx_hip.so (no PyTorch/CPU fallback: a missing library raises).

Extensions (all optional, default = reference behaviour):
  batch['_shared_context'] = True   the B samples are copies of ONE complex -> trajectory-invariant embeddings are built once
  ScoreNetwork.max_chunk            samples per launch through the pair stack (workspace size); None = fit 50 % of the free HBM

ESM2 hook (seqformer.py:185-191, encoder.py:72-121).  With `esm.enabled` the per-layer ESM2 representations of
the CURRENT antibody sequence are an INPUT of every network pass: a (B, Lab, embed_channel, num_layers + 1) tensor taken from
`ScoreNetwork.esm_provider(batch)` (a callable running ESM2 on the host PyTorch stack, called once per pass because the recycles
change `seq_t`) or, if no provider is set, from `batch['esm_embed']` (a fixed tensor).  The layer-softmax mix and the
LayerNorm -> Linear -> ReLU -> Linear projection run here; the ESM2 weights (`impl.seqformer.encode_esm_emb.*` keys of a
reference checkpoint) are skipped by `load_state_dict`.

Lifetime of what a call returns: `representations` and `_prev_pos` are views of two internal ping-pong buffers (a (B,L,L,192)
tensor is 9.5 GB at B = 100, L = 352; three since round 4: the input buffer of a call stays pinned until its range word is read).  A pass never writes the buffer that `batch['prev_pair']` currently points to, so the
self-conditioning input of the next call is always intact; the representations returned by call n are overwritten during call
n + 1.  Callers that keep them longer set `ScoreNetwork.clone_outputs = True` (fresh tensors, as the reference returns).
The trajectory-invariant embeddings travel in `batch['_static']`: their lifetime is that of the batch dict they were built from
(a new dict = a new complex = recomputed), never that of the module; the entry is keyed by the storage address and version counter of
`seq`, `fixed_mask` and `atom14_gt_positions`, so refilling a dict with another complex recomputes them too.
"""
import torch
from torch import nn

from abx_amd import _lib
from abx_amd.model.modules import ScoreNetworkIteration
from abx_amd.model.forward import Engine, Packed


def get_prev(batch, value, config):
    """abx.py:17-26.  The distogram of the virtual C-beta atoms is produced by the network pass itself
    (abx_prev_pos kernel) and travels in `value['_prev_pos']`; representations are returned by reference, as upstream."""
    return {
        'prev_pos': value['_prev_pos'],
        'prev_seq': value['representations']['seq'],
        'prev_pair': value['representations']['pair'],
    }


# per-sample workspace of a pair-stack pass: w384 + the op-group workspaces of csrc/blocks.hip (attention: q|k|v|gate, bias copies, output; tri-mul: operand images, product) ~ 7.3 KB per pair position
_WORKSPACE_BYTES_PER_PAIR = 7300      # (w384 1536 + attention block 3872 + tri-mul block 1540 + seq-attn bias 128 + masks: 7.1 KB measured)
_MAX_CHUNK = 8192


class ScoreNetwork(nn.Module):
    def __init__(self, model_conf, diffuser):
        super().__init__()
        self._model_conf = model_conf
        c = model_conf.embeddings_and_seqformer
        self.num_in_seq_channel = c.seq_channel
        self.num_in_pair_channel = c.pair_channel
        self.index_embed_size = c.index_embed_size
        self.impl = ScoreNetworkIteration(model_conf)
        self.diffuser = diffuser
        self._auto_chunks = {}
        self.max_chunk = None            # None: as many samples per pair-stack launch as fit in 50 % of the free HBM
        self.clone_outputs = False
        self.esm_provider = None         # callable(batch) -> (B, Lab, embed_channel, num_layers + 1) when esm.enabled
        self._engine = None
        self._engine_key = None
        self._engine_serial = 0
        self._bufs = {}
        self.range_log = []              # calls that left the split-f16 operand ranges and were repeated with the flagged op class exact
        self.range_sticky_after = 2      # flagged calls after which the flagged classes STAY on the exact kernels (no more repeats)
        self._sticky_tags = 0            # op classes (ops.RANGE_TAGS bits) that run exact from now on - for the CURRENT complex
        self._flagged_calls = 0          # flagged calls on the current complex
        self._complex_sig = None         # what `_sticky_tags` / `_flagged_calls` belong to (see _scope_range_state)
        self.forced_exact_ops = ()       # op classes (ops.RANGE_TAGS names) pinned to the exact kernels for the module's life (bench.py
                                         # --exact-class: what a checkpoint whose activations leave the split range would cost)
        self._pinned = set()
        self._n_calls = 0

    def load_state_dict(self, state_dict, strict=True, **kw):
        """The ESM2 module of a reference checkpoint belongs to the external embedding provider, not to this module."""
        sd = {k: v for k, v in state_dict.items() if not k.startswith('impl.seqformer.encode_esm_emb.')}
        return super().load_state_dict(sd, strict=strict, **kw)

    # ---- engine / packing ----------------------------------------------------------------------------------------
    def _get_engine(self, device):
        key = (str(device), tuple(p._version for p in self.parameters()), tuple(p.data_ptr() for p in self.parameters()))
        if self._engine is None or self._engine_key != key:
            _lib.load()                                    # raises if the HIP library is missing
            self._engine = Engine(self._model_conf, Packed(self.state_dict(), device), device)
            self._engine_key = key
            self._engine_serial += 1
            self._auto_chunks = {}
        return self._engine

    def _auto_chunk(self, B, L, device):
        """Samples per pair-stack launch.  Decided from the free HBM when the workspace for this problem size does not exist yet
        (the workspace a previous decision allocated counts as available: it is what gets reused or replaced), and clamped to
        the grid limits of the kernels (abx_transpose_last2 / abx_tri_attn_fwd: 4 * chunk, chunk <= 65535)."""
        key = (B, L)
        held = 0
        if self._engine is not None:        # what a previous decision allocated: the named scratch buffers and the op-group workspaces
            held = sum(b.numel() * b.element_size() for b in list(self._engine.ws.bufs.values()) + [v[0] for v in self._engine._blk_ws.values()])
        need = _WORKSPACE_BYTES_PER_PAIR * L * L
        hit = self._auto_chunks.get(key)
        if hit is not None and hit * need <= held * 1.05:
            return hit
        free, _ = torch.cuda.mem_get_info(device)
        chunk = max(1, min(B, _MAX_CHUNK, int(0.5 * (free + held) / need)))
        chunk = -(-B // -(-B // chunk))          # equal shares: 100 samples that do not fit one launch run as 50 + 50, not 99 + 1
        self._auto_chunks[key] = chunk
        return chunk

    def _buf(self, name, shape, dtype, device):
        b = self._bufs.get(name)
        if b is None or tuple(b.shape) != tuple(shape) or b.dtype != dtype or b.device != device:
            b = torch.empty(shape, dtype=dtype, device=device)
            self._bufs[name] = b
        return b

    def _scope_range_state(self, batch, L):
        """Sticky exact classes and the flagged-call count belong to ONE complex: a flag history of an earlier complex, sample
        block or rank must not change the low-order bits of a later one.  Called where the trajectory-invariant embeddings are rebuilt (a
        new batch dict): the complex is identified by the CONTENT of sample 0's context (length, sequence, fixed mask, coordinates - four
        small reductions, one read-back next to the one the range word of the embeddings costs anyway), so a fresh dict of the same
        complex keeps its history and another complex starts clean."""
        seq0 = batch['seq'][0].to(torch.float64)
        pos = torch.arange(1, L + 1, device=seq0.device, dtype=torch.float64)
        sig = (L,) + tuple(torch.stack([(seq0 * pos).sum(), (batch['fixed_mask'][0].to(torch.float64) * pos).sum(),
                                        batch['atom14_gt_positions'][0].to(torch.float64).sum()]).tolist())
        if sig != self._complex_sig:
            self._complex_sig = sig
            self._sticky_tags = self._forced_tags()
            self._flagged_calls = 0

    def _forced_tags(self):
        from abx_amd import ops
        bits = 0
        for n in self.forced_exact_ops:
            bits |= ops.RANGE_TAGS[n]
        return bits

    @property
    def range_sticky_ops(self):
        """Names of the op classes that run on the exact kernels for the current complex."""
        from abx_amd import ops
        return ops.range_names(self._sticky_tags)

    # ---- forward ----------------------------------------------------------------------------------------------------
    def forward(self, input_feats, compute_loss=True):
        batch = input_feats
        device = batch['seq'].device
        if device.type != 'cuda':
            raise RuntimeError('abx_amd.ScoreNetwork runs on an MI355X only (no CPU path); use oracle/ for CPU checks')
        B, L = batch['seq'].shape[:2]
        c = self._model_conf.embeddings_and_seqformer
        WS_, WZ = c.seq_channel + c.index_embed_size, c.pair_channel + 2 * c.index_embed_size
        eng = self._get_engine(device)
        if 'prev_seq' not in batch:
            batch.update(prev_pos=torch.zeros([B, L, L], device=device, dtype=torch.int64),
                         prev_seq=torch.zeros([B, L, WS_], device=device),
                         prev_pair=torch.zeros([B, L, L, WZ], device=device))
        shared = bool(batch.get('_shared_context', False)) or B == 1
        # Residue/PairEmbedding depend on the fixed context only: built once per batch dict and carried in it
        # (keyed by the identity AND version of the context tensors: a caller that refills one dict with another complex of the same
        # shape, in place or by replacing the tensors, gets fresh embeddings)
        ctx = tuple((batch[k].data_ptr(), batch[k]._version) for k in ('seq', 'fixed_mask', 'atom14_gt_positions'))
        from abx_amd import ops
        # Range safety of the split-f16 kernels (include/abx_hip.h, AbxGemm.range_flag): the kernels OR the bit of their op class into the
        # device's range word when a value they store is not finite - what an activation beyond their operand ranges becomes (the
        # reference's plain fp32 contractions have no such range: seqformer.py:260-312, 443-504).
        word = ops.range_words(device) if ops.RANGE_CHECK and not ops.GEMM_EXACT else None
        capturing = torch.cuda.is_current_stream_capturing()
        P = eng.P
        self._sticky_tags |= self._forced_tags()
        P.exact_tags = self._sticky_tags
        skey = (id(self), self._engine_serial, B, L, shared, ctx)
        hit = batch.get('_static')
        if hit is None or hit[0] != skey:
            if not capturing:
                self._scope_range_state(batch, L)
                P.exact_tags = self._sticky_tags
            # the cached embeddings are built by range-tagged GEMMs too: their own clean word, and a rebuild on the exact kernels when
            # they set it (one host synchronisation per complex)
            if word is not None and not capturing:
                word.zero_()
            ops.RANGE_SLOT = 0
            emb = eng.static_embeddings(batch, shared)
            if word is not None and not capturing and any(word.tolist()):
                self._sticky_tags |= ops.RANGE_TAGS['gemm']
                P.exact_tags = self._sticky_tags
                self.range_log.append({'call': self._n_calls + 1, 'ops': ['gemm'], 'where': 'static embeddings', 'L': L, 'B': B, 'sticky': True,
                                       'sticky_set': ops.range_names(self._sticky_tags)})
                emb = eng.static_embeddings(batch, shared)
            hit = (skey, emb)
            batch['_static'] = hit
        self._static = hit[1]
        num_recycle = self._model_conf.num_recycle
        self._n_calls += 1

        def passes():
            # (every pass reports to its own range word: a pass that left the range hands NaN rows to ALL classes of the next one
            # through the recycled representations, so only the first flagged pass says which class it was)
            try:
                batch.update(is_recycling=True)
                for i in range(num_recycle):
                    ops.RANGE_SLOT = min(i, ops.RANGE_SLOTS - 1)
                    r = self._pass(eng, batch, final=False)
                    prev = get_prev(batch, r, self._model_conf)
                    batch.update(seq_t=r['heads']['sequence_module']['seq_0'])
                    batch.update(prev)
                batch.update(is_recycling=False)
                ops.RANGE_SLOT = min(num_recycle, ops.RANGE_SLOTS - 1)
                return self._pass(eng, batch, final=bool(compute_loss))
            finally:
                ops.RANGE_SLOT = 0

        # The word is read ONCE per call (one host synchronisation; a read per pass cost 18 % of a 32-sample trajectory step, whose kernels
        # are short enough for the launch thread to be exposed after every drain); the call's inputs - the self-conditioning tensors of
        # the previous call, the tokens - stay intact until then (the representation buffers rotate three ways and the call's input buffer
        # is pinned).  A flagged call is repeated with the FIRST flagged op class of the pass (ops.RANGE_ORDER: every class behind it only
        # saw its NaN rows) on the exact fp32-MFMA kernels and everything else still on the split-f16 ones; if the repeat flags a later
        # class, that one joins, and so on (at most one repeat per class).  After `range_sticky_after` flagged calls the classes found
        # so far stay exact for the rest of THIS COMPLEX (a checkpoint whose activations sit beyond the range would otherwise pay split
        # + exact on every call; _scope_range_state starts the next complex clean, so that its bits do not depend on what ran before it on
        # this rank), logged once and carried in the trajectory's metadata (`range_fallbacks` / `range_sticky_ops`).  A caller never sees the contract: results are the reference's either way.  All samples of
        # a call share the arithmetic switch, so a sample's low-order bits depend on its batch mates IN A FLAGGED CALL ONLY.  Inside a
        # hipGraph capture the word only accumulates (abx_amd.graph checks it after each replay).
        start = {k: batch.get(k) for k in ('seq_t', 'prev_pos', 'prev_seq', 'prev_pair')}
        self._pinned = {v.data_ptr() for k, v in start.items() if k != 'seq_t' and torch.is_tensor(v)}
        try:
            with torch.no_grad():
                tags, entry = self._sticky_tags, None
                for _attempt in range(len(ops.RANGE_ORDER) + 1):
                    P.exact_tags = tags
                    if word is not None and not capturing:
                        word.zero_()
                    ret = passes()
                    if word is None or capturing:
                        break
                    per_pass = word.tolist()
                    bits = 0
                    for v in per_pass:
                        bits |= v
                    if not bits:
                        break
                    # the first flagged pass names the class (a class that is exact already does not report: `skip`)
                    first = ops.first_range_tag(next(v for v in per_pass if v), skip=tags)
                    if entry is None:
                        entry = {'call': self._n_calls, 'ops': ops.range_names(bits), 'exact_ops': [], 'L': L, 'B': B, 'repeats': 0}
                        self.range_log.append(entry)
                    if not first:       # every flagged class is exact already: the values are not finite for another reason (inputs, weights)
                        break
                    tags |= first
                    entry['exact_ops'] += ops.range_names(first)
                    entry['repeats'] += 1
                    batch.update(start)
                if entry is not None:
                    self._flagged_calls += 1
                    if self._flagged_calls >= self.range_sticky_after and (tags & ~self._sticky_tags):
                        self._sticky_tags |= tags
                        entry['sticky'] = True
                        entry['sticky_set'] = ops.range_names(self._sticky_tags)
                        import logging
                        logging.getLogger('abx_amd').warning(
                            'split-f16 operand ranges left in %d calls: the op classes %s run on the exact fp32-MFMA kernels from now on',
                            self._flagged_calls, ops.range_names(self._sticky_tags))
        finally:
            self._pinned = set()
            P.exact_tags = self._sticky_tags
        return ret

    def _esm_embed(self, batch):
        c = self._model_conf.embeddings_and_seqformer
        if not c.esm.enabled:
            return None
        e = self.esm_provider(batch) if self.esm_provider is not None else batch.get('esm_embed')
        if e is None:
            raise RuntimeError('esm.enabled: set ScoreNetwork.esm_provider (callable(batch) -> (B, Lab, C, layers) tensor) or '
                               "batch['esm_embed']; the ESM2 model itself runs outside this module")
        B, Lab = batch['seq'].shape[0], batch['anchor_flag'].shape[1]
        assert tuple(e.shape) == (B, Lab, c.esm.embed_channel, c.esm.num_layers + 1), tuple(e.shape)
        return e.to(device=batch['seq'].device)

    def _pass(self, eng, batch, final):
        device = batch['seq'].device
        B, L = batch['seq'].shape[:2]
        c = self._model_conf.embeddings_and_seqformer
        WS_, WZ = c.seq_channel + c.index_embed_size, c.pair_channel + 2 * c.index_embed_size
        NC = self._model_conf.heads.diffusion_module.IPA.num_channel
        # ping-pong representation buffers: write the one that batch['prev_*'] does NOT point to (the self-conditioning input of
        # this pass); foreign prev_* tensors (first call, reference-style drivers) leave both free
        f32, i64 = torch.float32, torch.int64
        busy = {batch[k].data_ptr() for k in ('prev_seq', 'prev_pair', 'prev_pos') if torch.is_tensor(batch.get(k))} | self._pinned
        tag = '0'
        for cand in ('0', '1', '2'):
            mine = [self._bufs.get(n + cand) for n in ('rep_seq', 'rep_pair', 'prev_pos')]
            if not any(b is not None and b.data_ptr() in busy for b in mine):
                tag = cand
                break
        t = batch['t']
        t_is_f32 = t.dtype != torch.float64
        st = dict(
            L=L, Lab=batch['anchor_flag'].shape[1], diffuser=self.diffuser, static=self._static,
            seq_t=batch['seq_t'].to(i64).contiguous(), mask_f=batch['mask'].to(f32).contiguous(),
            fixed_i32=batch['fixed_mask'].to(torch.int32).contiguous(), rigids_t=batch['rigids_t'],
            torsion_gt=batch['torsion_angles_sin_cos'].to(f32), a37to14=batch['residx_atom37_to_atom14'].to(i64),
            prev_seq=batch.get('prev_seq'), prev_pair=batch.get('prev_pair'), prev_pos=batch.get('prev_pos'),
            esm_embed=self._esm_embed(batch),
            t64=t.to(torch.float64).contiguous(), t_is_f32=t_is_f32,
            rep_seq_out=self._buf('rep_seq' + tag, (B, L, WS_), f32, device),
            rep_pair_out=self._buf('rep_pair' + tag, (B, L, L, WZ), f32, device),
            prev_pos_out=self._buf('prev_pos' + tag, (B, L, L), i64, device),
            structure_module=torch.empty(B, L, NC, device=device),
            angles=torch.empty(B, L, 7, 2, device=device),
            rot_score=torch.empty(B, L, 3, device=device),
            trans_score=torch.empty(B, L, 3, device=device, dtype=f32 if t_is_f32 else torch.float64),
            rigids=torch.empty(B, L, 7, device=device),
            logits=torch.empty(B, L, 20, device=device), seq_0=torch.empty(B, L, device=device, dtype=i64),
            atom14=torch.empty(B, L, 14, 3, device=device), atom37=torch.empty(B, L, 37, 3, device=device),
            pLDDT=torch.empty(B, L, device=device),
        )
        for k in ('prev_seq', 'prev_pair', 'prev_pos'):
            if st[k] is not None:
                st[k] = st[k].contiguous()
                assert st[k].data_ptr() != st['rep_seq_out'].data_ptr() and st[k].data_ptr() != st['rep_pair_out'].data_ptr() \
                    and st[k].data_ptr() != st['prev_pos_out'].data_ptr(), 'self-conditioning buffer aliasing'
        st['temb'] = torch.empty(B, c.index_embed_size, device=device)
        from abx_amd import ops
        ops.timestep_embedding(st['t64'], c.index_embed_size, st['temb'])
        chunk = max(1, min(self.max_chunk, B)) if self.max_chunk else self._auto_chunk(B, L, device)
        for b0 in range(0, B, chunk):
            eng.run_chunk(st, b0, min(B, b0 + chunk), final)
        folding = {
            'rot_score': st['rot_score'], 'trans_score': st['trans_score'], 'rigids': st['rigids'],
            'final_atom14_positions': st['atom14'], 'final_atom_positions': st['atom37'],
            'representations': {'structure_module': st['structure_module']},
            'sidechains': [{'angles_sin_cos': st['angles']}],
        }
        ret = {'representations': {'seq': st['rep_seq_out'], 'pair': st['rep_pair_out']},
               'heads': {'folding': folding, 'sequence_module': {'logits': st['logits'], 'seq_0': st['seq_0']}},
               '_prev_pos': st['prev_pos_out']}
        if final:
            ret['heads']['predicted_lddt'] = {'pLDDT': st['pLDDT']}
            if self.clone_outputs:
                ret['representations'] = {k: v.clone() for k, v in ret['representations'].items()}
                ret['_prev_pos'] = ret['_prev_pos'].clone()
        return ret
