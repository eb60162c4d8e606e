"""Score-network forward on libabx_hip kernels (host orchestration only: buffer views, strides, launch order).

Follows the reference's ScoreNetworkIteration.forward (abx/model/abx.py:42-63): EmbeddingAndSeqformer
(abx/model/seqformer.py:170-226, block :569-606) -> IpaScore (abx/model/score_network.py:83-196) -> SequenceHead
(abx/model/head.py:162-201) -> PredictedLDDTHead (:222-226), plus get_prev's distogram (abx/model/abx.py:17-26).
Distogram / metric / TM-score heads are not computed (unused by sampling).

Memory plan for B samples of one complex (fp32): two ping-pong (B,L,L,192) pair representations (the previous one is the
self-conditioning input), and a per-chunk workspace of L^2 * 1187 floats per sample that is reused by every stage.
"""
import math

import numpy as np
import torch

from abx_amd import ops
from abx_amd import residue_constants as rc

# round-6 experiment, OFF unless ABX_ASSEMBLE_BIAS is set: abx_assemble_pair_bias (assembly + the sequence attention's pair bias in one pass) is correct
# and SLOWER than the two launches (7.14 vs 5.99 ms at 100 samples: its fragment-layout loads and stores touch 32 cache lines per instruction,
# profiles/r06q_kb_assemble_bias.txt)
_NO_ASSEMBLE_BIAS = not bool(__import__('os').environ.get('ABX_ASSEMBLE_BIAS'))
_NO_OPM_FUSED = bool(__import__('os').environ.get('ABX_NO_OPM_FUSED'))      # (A / B runs: the feature tensor + GEMM form of rounds 1 - 5)
P_SEQF = 'impl.seqformer.'
P_BLK = 'impl.seqformer.seqformer.blocks.0.'
P_IPA = 'impl.diffusion_module.ScoreNetwork.'


class Packed:
    """Kernel-ready copies of the parameters: Linear weights transposed to [K][N] (n-contiguous B operand), fused
    projection groups concatenated, embeddings as gather tables."""

    def __init__(self, sd, device):
        f = lambda k: sd[k].detach().to(device=device, dtype=torch.float32).contiguous()
        self.sd = {k: f(k) for k in sd}
        g = self.sd
        self.wt, self.b = {}, {}
        self._ln_cache = {}
        self._split_cache = {}
        self.gemm_mode = 0           # set per pass by Engine.run_chunk (ops.gemm_mode(L))
        # op classes (bits of ops.RANGE_TAGS) that run on the exact fp32-MFMA kernels although the complex is large enough for the split-f16
        # ones: set by ScoreNetwork when such a class left the split ranges (per class, not per call: everything else stays split-f16)
        self.exact_tags = 0
        self.plain_class = 'gemm'    # range class of the plain GEMMs being issued: 'gemm' in front of the pair stack, 'gemm_late' behind it

        def lin(name, key=None):
            key = key or name
            self.wt[key] = g[name + '.weight'].t().contiguous()
            self.b[key] = g.get(name + '.bias')

        def fused(key, names):
            self.wt[key] = torch.cat([g[n + '.weight'] for n in names], dim=0).t().contiguous()
            bs = [g.get(n + '.bias') for n in names]
            if any(b is not None for b in bs):
                self.b[key] = torch.cat([b if b is not None else torch.zeros(g[n + '.weight'].shape[0], device=device)
                                         for b, n in zip(bs, names)]).contiguous()
            else:
                self.b[key] = None

        for name in list(g):
            if name.endswith('.weight') and g[name].dim() == 2 and not any(e in name for e in (
                    'embed.weight', 'proj_aa_type', 'proj_rel_pos', 'proj_prev_pos', 'aapair_to_distcoef')):
                lin(name[:-7])
        for tm in ('triangle_multiplication_outgoing', 'triangle_multiplication_incoming'):
            fused(P_BLK + tm + '.lr_gates', [P_BLK + tm + s for s in ('.left_gate', '.right_gate')])
            fused(P_BLK + tm + '.lr_proj', [P_BLK + tm + s for s in ('.left_proj', '.right_proj')])
            # gated projections in one GEMM: (value, gate) column pairs of the same 256 channels (ops.pack_glu_weights)
            self.wt[P_BLK + tm + '.lr_glu'], self.b[P_BLK + tm + '.lr_glu'] = ops.pack_glu_weights(
                self.wt[P_BLK + tm + '.lr_proj'], self.wt[P_BLK + tm + '.lr_gates'],
                self.b.get(P_BLK + tm + '.lr_proj'), self.b.get(P_BLK + tm + '.lr_gates'))
        for ta in ('triangle_attention_starting_node', 'triangle_attention_ending_node'):
            fused(P_BLK + ta + '.qkv', [P_BLK + ta + s for s in ('.attn.proj_q', '.attn.proj_k', '.attn.proj_v')])
        fused(P_BLK + 'outer_product_mean.lr', [P_BLK + 'outer_product_mean.left_proj', P_BLK + 'outer_product_mean.right_proj'])
        fused(P_IPA + 'attention_module.proj', [P_IPA + 'attention_module.' + s for s in (
            'proj_q_scalar', 'proj_kv_scalar', 'proj_q_point_local', 'proj_kv_point_local')])
        # IPA point weights: -0.5 * sqrt(1/(3*4*9/2)) * softplus(w)   (folding.py:59-66,96)
        w_p = float(np.sqrt(1.0 / (3 * 4 * 9. / 2)))
        self.ipa_pw = (-0.5 * w_p * torch.nn.functional.softplus(g[P_IPA + 'attention_module.trainable_point_weights'])).contiguous()
        self.ipa_ws = float(np.sqrt(1.0 / (3 * 16 * 1.)))
        self.ipa_w2d = float(np.sqrt(1.0 / 3))
        # residue tables
        self.default_frames = torch.as_tensor(rc.restype_rigid_group_default_frame, device=device).contiguous()
        self.group_idx = torch.as_tensor(rc.restype_atom14_to_rigid_group, device=device).to(torch.int32).contiguous()
        self.lit_pos = torch.as_tensor(rc.restype_atom14_rigid_group_positions, device=device).contiguous()

    def ln(self, name):
        return self.sd[name + '.weight'], self.sd[name + '.bias']

    def exact(self, *classes):
        """True when the ops of (one of) the named range classes must run on the exact kernels in this pass."""
        return ops.GEMM_EXACT or self.gemm_mode != 2 or any(self.exact_tags & ops.RANGE_TAGS[c] for c in classes)

    def ln_linear(self, key, ln_name):
        """LayerNorm folded into the following Linear for abx_gemm's algebraic-LN epilogue:
        returns (Wt' = gamma[:,None] * Wt, csum = colsum(Wt'), bias' = beta @ Wt + b), built once in float64."""
        ck = (key, ln_name)
        hit = self._ln_cache.get(ck)
        if hit is None:
            gamma, beta = self.sd[ln_name + '.weight'].double(), self.sd[ln_name + '.bias'].double()
            wt = self.wt[key].double()
            wts = gamma[:, None] * wt
            bias = beta @ wt
            if self.b.get(key) is not None:
                bias = bias + self.b[key].double()
            w32 = wts.float().contiguous()
            hit = (w32, wts.sum(0).float().contiguous(), bias.float().contiguous(), self._split(w32))
            self._ln_cache[ck] = hit
        return hit

    @staticmethod
    def _split(wt):
        # float16 weight planes for the split-f16 GEMM kernels (only problems with N > 64 ever take that path)
        return ops.split_weights(wt) if (wt.shape[1] > 64 and wt.is_cuda) else None

    def split(self, key):
        if key not in self._split_cache:
            self._split_cache[key] = self._split(self.wt[key])
        return self._split_cache[key]

    def mlp_second(self, key):
        """(planes, bias) of the second layer of a fused transition (AbxGemm.mlp): k order permuted inside every 16-tile."""
        ck = ('mlp2', key)
        if ck not in self._split_cache:
            self._split_cache[ck] = ops.split_weights(ops.permute_k16(self.wt[key]))
        return self._split_cache[ck], self.b.get(key)

    def block_packs(self):
        """Weights of the three pair-stack op groups packed by abx_pack_linear (csrc/blocks.hip) for the op-group entry points of the
        C ABI: {'pair_transition': (l1, l2), '<tri-mul name>': AbxTriMulPack, '<tri-attn name>': AbxTriAttnPack}; built once."""
        if getattr(self, '_blocks', None) is None:
            g = self.sd
            W = lambda n: g[n + '.weight']
            Bv = lambda n: g.get(n + '.bias')
            ln = lambda n: (g[n + '.weight'], g[n + '.bias'])
            blk = {}
            pre = P_BLK + 'pair_transition.transition.'
            blk['pair_transition'] = (ops.LinearPack([(W(pre + '1'), Bv(pre + '1'), 0)], 192, ln=ln(pre + '0')),
                                      ops.LinearPack([(W(pre + '3'), Bv(pre + '3'), 0)], W(pre + '3').shape[1], permute_k16=True))
            for tm in ('triangle_multiplication_outgoing', 'triangle_multiplication_incoming'):
                pre = P_BLK + tm + '.'
                glu = ops.LinearPack([(W(pre + 'left_proj'), Bv(pre + 'left_proj'), 1), (W(pre + 'right_proj'), Bv(pre + 'right_proj'), 1),
                                      (W(pre + 'left_gate'), Bv(pre + 'left_gate'), 2), (W(pre + 'right_gate'), Bv(pre + 'right_gate'), 2)],
                                     192, ln=ln(pre + 'norm'))
                out = ops.LinearPack([(W(pre + 'proj_out'), Bv(pre + 'proj_out'), 0)], 128, ln=ln(pre + 'final_norm'))
                gate = ops.LinearPack([(W(pre + 'final_gate'), Bv(pre + 'final_gate'), 0)], 192, ln=ln(pre + 'norm'))
                blk[tm] = ops.tri_mul_pack(glu, out, gate)
            for ta in ('triangle_attention_starting_node', 'triangle_attention_ending_node'):
                pre = P_BLK + ta + '.'
                qkv = ops.LinearPack([(W(pre + 'attn.proj_q'), Bv(pre + 'attn.proj_q'), 0), (W(pre + 'attn.proj_k'), Bv(pre + 'attn.proj_k'), 0),
                                      (W(pre + 'attn.proj_v'), Bv(pre + 'attn.proj_v'), 0)], 192, ln=ln(pre + 'norm'))
                gate = ops.LinearPack([(W(pre + 'attn.gate'), Bv(pre + 'attn.gate'), 0)], 192, ln=ln(pre + 'norm'))
                pair = ops.LinearPack([(W(pre + 'proj_pair'), Bv(pre + 'proj_pair'), 0)], 192, ln=ln(pre + 'norm'))
                # (k-permuted planes: the gated tail feeds the output projection from the registers of the gate projection)
                out = ops.LinearPack([(W(pre + 'attn.proj_out'), Bv(pre + 'attn.proj_out'), 0)], 192, permute_k16=True)
                blk[ta] = ops.tri_attn_pack(qkv, gate, pair, out)
            self._blocks = blk
            # ONE packing per weight: the descriptor-level path (the _ln_lin / _lin calls of Engine.run_chunk) reads the same buffers,
            # so the two ways of issuing the op groups give the same bits
            l1, l2 = blk['pair_transition']
            pre = P_BLK + 'pair_transition.transition.'
            self._ln_cache[(pre + '1', pre + '0')] = (l1.Wt, l1.csum, l1.bias, l1.planes)
            self._split_cache[('mlp2', pre + '3')] = l2.planes
            self.wt[pre + '3'], self.b[pre + '3'] = l2.Wt, l2.bias
            self._split_cache[pre + '3'] = None                       # (its planes are k-permuted: never the operand of a plain GEMM)
            for tm in ('triangle_multiplication_outgoing', 'triangle_multiplication_incoming'):
                pre = P_BLK + tm + '.'
                glu, out, gate = blk[tm]._keep
                self._ln_cache[(pre + 'lr_glu', pre + 'norm')] = (glu.Wt, glu.csum, glu.bias, glu.planes)
                self._ln_cache[(pre + 'proj_out', pre + 'final_norm')] = (out.Wt, out.csum, out.bias, out.planes)
                self._ln_cache[(pre + 'final_gate', pre + 'norm')] = (gate.Wt, gate.csum, gate.bias, gate.planes)
            for ta in ('triangle_attention_starting_node', 'triangle_attention_ending_node'):
                pre = P_BLK + ta + '.'
                qkv, gate, pair, out = blk[ta]._keep
                self._ln_cache[(pre + 'qkv', pre + 'norm')] = (qkv.Wt, qkv.csum, qkv.bias, qkv.planes)
                self._ln_cache[(pre + 'attn.gate', pre + 'norm')] = (gate.Wt, gate.csum, gate.bias, gate.planes)
                self._ln_cache[(pre + 'proj_pair', pre + 'norm')] = (pair.Wt, pair.csum, pair.bias, pair.planes)
                self.wt[pre + 'attn.proj_out'], self.b[pre + 'attn.proj_out'] = out.Wt, out.bias
                self._split_cache[('mlp2', pre + 'attn.proj_out')] = out.planes
                self._split_cache[pre + 'attn.proj_out'] = None             # (its planes are k-permuted: never the operand of a plain GEMM)
        return self._blocks

    def heads_pack(self):
        """Operands of ops.heads_tail (torsion ResNet + SequenceHead + PredictedLDDTHead in one launch), or None when the checkpoint's
        widths are not the 256 -> 128 ones the kernel is built for; built once."""
        if not hasattr(self, '_heads'):
            self._heads = None
            pre = P_IPA + 'sidechain_module.torsion_module.'
            names = [pre + 'proj_act.1', pre + 'proj_init_act.1', pre + 'blocks.0.net.1', pre + 'blocks.0.net.3', pre + 'blocks.1.net.1',
                     pre + 'blocks.1.net.3', pre + 'projection']
            shapes = [(256, 128), (256, 128), (128, 128), (128, 128), (128, 128), (128, 128), (128, 14)]
            heads = (('impl.sequence_module.net.', 20), ('impl.predicted_lddt.net.', 50))
            ok = all(n in self.wt and tuple(self.wt[n].shape) == sh for n, sh in zip(names, shapes)) and (pre + 'blocks.2.net.1') not in self.wt
            for hp, nout in heads:
                ok = ok and all((hp + i) in self.wt for i in '135') and tuple(self.wt[hp + '1'].shape) == (256, 128) and \
                    tuple(self.wt[hp + '3'].shape) == (128, 128) and tuple(self.wt[hp + '5'].shape) == (128, nout)
            if ok and self.wt[names[0]].is_cuda:
                zb = lambda n: self.b[n] if self.b.get(n) is not None else torch.zeros(128, device=self.wt[n].device)
                full = lambda n: (ops.split_weights(self.wt[n]), zb(n))
                tors = [full(n) for n in names[:-1]] + [ops.pad_planes_128(self.wt[names[-1]], self.b.get(names[-1]))]
                hd = []
                for hp, _ in heads:
                    hd.append((self.ln(hp + '0'), full(hp + '1'), full(hp + '3'), ops.pad_planes_128(self.wt[hp + '5'], self.b.get(hp + '5'))))
                self._heads = (tors, hd[0], hd[1])
        return self._heads

    def split_narrow(self, wt, key):
        """float16 weight planes of a skinny (N <= 32) weight matrix: the pair-stack bias projections stream their 9.5 GB A operand
        through the 128 x 32 tile of the split-f16 GEMM (DMA pipeline) instead of the exact kernel's register-staged loads."""
        ck = ('narrow', key)
        if ck not in self._split_cache:
            self._split_cache[ck] = ops.split_weights(wt) if (wt.is_cuda and wt.shape[0] % 16 == 0) else None
        return self._split_cache[ck]


class Workspace:
    def __init__(self, device):
        self.device = device
        self.bufs = {}
        self.shapes = {}

    def get(self, name, shape, dtype=torch.float32, zero=False):
        """zero=True: the view is zero-filled whenever the buffer is new or was last handed out with another shape
        (padding regions that the kernels never write must read as 0)."""
        n = int(np.prod(shape))
        buf = self.bufs.get(name)
        fresh = buf is None or buf.numel() < n or buf.dtype != dtype
        if fresh:
            buf = torch.empty(max(n, 1), device=self.device, dtype=dtype)
            self.bufs[name] = buf
        if zero and (fresh or self.shapes.get(name) != tuple(shape)):
            buf[:n].zero_()
        self.shapes[name] = tuple(shape)
        return buf[:n].view(*shape)


def _lin(P, name, x, out, narrow=False, **kw):
    kw.setdefault('exact', 1 if P.exact(P.plain_class) else 2)
    kw.setdefault('range_class', P.plain_class)
    w3 = P.split(name)
    if narrow and w3 is None and kw['exact'] == 2:
        w3 = P.split_narrow(P.wt[name], name)
    return ops.gemm(x, P.wt[name], out, bias=P.b.get(name), B3=w3, **kw)


def _ln_lin(P, name, ln_name, stats, x, out, narrow=False, **kw):
    """out = epi(LN(x) @ W^T + b) with the LayerNorm folded into the GEMM epilogue."""
    wt, csum, bias, w3 = P.ln_linear(name, ln_name)
    kw.setdefault('exact', 1 if P.exact(P.plain_class) else 2)
    kw.setdefault('range_class', P.plain_class)
    if narrow and w3 is None and kw['exact'] == 2:
        w3 = P.split_narrow(wt, (name, ln_name))
    return ops.gemm(x, wt, out, bias=bias, ln=(stats, csum), B3=w3, **kw)


class Engine:
    def __init__(self, cfg_model, packed, device):
        self.cfg = cfg_model
        self.P = packed
        self.dev = device
        self.ws = Workspace(device)
        # True: the three pair-stack op groups (triangle multiplication x 2, triangle attention x 2, pair transition) run through the
        # op-group entry points of the C ABI (abx_tri_mul_fwd / abx_tri_attn_block_fwd / abx_transition_fwd, csrc/blocks.hip) - the same
        # kernels in the same order as the descriptor-level path below, which stays as the per-kernel view (bench.py's op profile, the
        # exact-arithmetic tri-mul)
        self.block_api = not bool(__import__('os').environ.get('ABX_NO_BLOCK_API'))
        self._blk_ws = {}
        # True: torsion ResNet + sequence / pLDDT head MLPs of a pass in one launch (abx_heads_tail) instead of 13 - 18 small ones
        self.fused_heads = not bool(__import__('os').environ.get('ABX_NO_FUSED_HEADS'))
        # K-slices of the IPA final_proj GEMM in front of abx_ipa_tail (0: the tail walks K itself)
        self.ipa_splitk = int(__import__('os').environ.get('ABX_IPA_SPLITK', '11'))
        c = cfg_model.embeddings_and_seqformer
        pp = c.prev_pos
        # squared distogram breaks exactly as torch computes them on the host (common_modules.py:108-109)
        self.sq_breaks = torch.square(torch.linspace(pp.min_bin, pp.max_bin, steps=pp.num_bins - 1)).to(device)

    def _block_workspace(self, group, Bc, L):
        """Workspace of an op group ('attn' / 'mul') for a chunk of Bc samples: one buffer per (group, L), sized for the largest chunk
        asked for so far.  The tri-mul workspace's operand-image region must read zero in its pad k-tiles and its layout depends on the
        chunk size: switching sizes re-zeroes it (only when ceil4(L) % 16 != 0: otherwise there are no pad k-tiles)."""
        key = (group, L)
        hit = self._blk_ws.get(key)
        if hit is None or hit[1] < Bc:
            self._blk_ws = {k: v for k, v in self._blk_ws.items() if k[0] != group}       # another L (another complex): release first
            buf = ops.tri_attn_block_workspace(Bc, L, self.dev) if group == 'attn' else ops.tri_mul_workspace(Bc, L, self.dev)
            hit = [buf, Bc, Bc]
            self._blk_ws[key] = hit
        elif group == 'mul' and hit[2] != Bc:
            if ((L + 3) // 4 * 4) % 16 != 0:
                ops.tri_mul_workspace_init(hit[0], Bc, L)
            hit[2] = Bc
        return hit[0]

    # ------------------------------------------------------------------------------------------------------------
    # trajectory-invariant encoders (encoder.py:123-269 + seqformer.py:177-206)
    # ------------------------------------------------------------------------------------------------------------
    def static_embeddings(self, batch, shared):
        P, ws = self.P, self.ws
        sl = slice(0, 1) if shared else slice(None)
        seq_t = batch['seq_t'][sl].long().contiguous()
        seq = batch['seq'][sl].long().contiguous()
        B, L = seq.shape
        P.gemm_mode = ops.gemm_mode(L)
        P.plain_class = 'gemm'
        Lab = batch['anchor_flag'].shape[1]
        mask = torch.logical_and(batch['mask'][sl], batch['fixed_mask'][sl].bool())
        mask_f = mask.float().contiguous()
        chain = batch['chain_id'][sl].to(torch.int32).contiguous()
        residx = batch['residx'][sl].to(torch.int32).contiguous()
        atom14 = batch['atom14_gt_positions'][sl].float().contiguous()
        exists = batch['atom14_gt_exists'][sl].to(torch.uint8).contiguous()
        M1, M2 = B * L, B * L * L
        dev = self.dev
        # ---- ResidueEmbedding
        pre = P_SEQF + 'encode_residue_emb.'
        h = torch.empty(M1, 1538, device=dev)
        ops.gather_rows(P.sd[pre + 'aatype_embed.weight'], seq_t, h[:, 0:512], rowscale=mask_f.reshape(-1))
        h[:, 512] = chain.reshape(-1).float()
        h[:, 513] = residx.reshape(-1).float()
        ops.gather_rows(P.sd[pre + 'cdr_embed.weight'], batch['cdr_def'][sl].long().contiguous(), h[:, 514:1026])
        cx = torch.cat([atom14.reshape(M1, 42), batch['torsion_angles_sin_cos'][sl].float().reshape(M1, 14)], dim=-1).contiguous()
        c1 = torch.empty(M1, 512, device=dev)
        _lin(P, pre + 'coordinate_embed.0', cx, c1, act=1)
        _lin(P, pre + 'coordinate_embed.2', c1, h[:, 1026:1538])
        a1 = torch.empty(M1, 1024, device=dev)
        _lin(P, pre + 'mlp.0', h, a1, act=1)
        a2 = torch.empty(M1, 512, device=dev)
        _lin(P, pre + 'mlp.2', a1, a2, act=1)
        a3 = torch.empty(M1, 512, device=dev)
        _lin(P, pre + 'mlp.4', a2, a3, act=1)
        seq_static = torch.empty(M1, 512, device=dev)
        _lin(P, pre + 'mlp.6', a3, seq_static, rowscale=mask_f.reshape(-1))
        # antigen rows: + aa_proj(proj_aa_type[seq])   (seqformer.py:199-201)
        if L > Lab:
            ag_idx = seq[:, Lab:].contiguous()
            n_ag = ag_idx.numel()
            e = torch.empty(n_ag, 512, device=dev)
            ops.gather_rows(P.sd[P_SEQF + 'proj_aa_type.weight'], ag_idx, e)
            e1 = torch.empty(n_ag, 512, device=dev)
            _ln_lin(P, P_SEQF + 'aa_proj.1', P_SEQF + 'aa_proj.0', None, e, e1, act=1)
            ss = seq_static.view(B, L, 512)
            # rows of the antigen are not contiguous over the batch: one GEMM per sample keeps the C ABI simple
            for b in range(B):
                tgt = ss[b, Lab:]
                _lin(P, P_SEQF + 'aa_proj.3', e1[b * (L - Lab):(b + 1) * (L - Lab)], tgt, resid=tgt)
        # ---- PairEmbedding
        pre = P_SEQF + 'encode_pair_emb.'
        feat = torch.empty(M2, 512, device=dev)
        dist = torch.empty(M2, 196, device=dev)
        ops.pair_embed_features(seq_t, chain, residx, atom14, exists, P.sd[pre + 'aa_pair_embed.weight'],
                                P.sd[pre + 'relpos_embed.weight'], P.sd[pre + 'aapair_to_distcoef.weight'],
                                P.sd[pre + 'dgram_embed.weight'], self.sq_breaks, feat, dist, B, L)
        d1 = torch.empty(M2, 128, device=dev)
        _lin(P, pre + 'distance_embed.0', dist, d1, act=1)
        _lin(P, pre + 'distance_embed.2', d1, feat[:, 256:384], act=1)
        o1 = torch.empty(M2, 128, device=dev)
        _lin(P, pre + 'out_mlp.0', feat, o1, act=1)
        _lin(P, pre + 'out_mlp.2', o1, d1, act=1)
        pmask = torch.empty(M2, device=dev)
        ops.pair_mask(mask_f, pmask, B, L)
        rel = torch.empty(M2, 128, device=dev)
        ops.relpos_block(residx, P.sd[P_SEQF + 'proj_rel_pos.weight'], rel, B, L, Lab, self.cfg.embeddings_and_seqformer.max_relative_feature)
        pair_static = torch.empty(M2, 128, device=dev)
        _lin(P, pre + 'out_mlp.4', d1, pair_static, rowscale=pmask, resid=rel)
        return seq_static.view(B, L, 512), pair_static.view(B, L, L, 128)

    # ------------------------------------------------------------------------------------------------------------
    # one network pass over samples [b0, b1)
    # ------------------------------------------------------------------------------------------------------------
    def run_chunk(self, st, b0, b1, final):
        """st: per-call state (full-batch tensors).  Writes outputs for samples b0:b1 in place."""
        P, ws, cfg = self.P, self.ws, self.cfg
        c = cfg.embeddings_and_seqformer
        Bc = b1 - b0
        L, Lab = st['L'], st['Lab']
        P.gemm_mode = ops.gemm_mode(L)
        P.plain_class = 'gemm'
        packs = P.block_packs()          # (built once; both ways of issuing the op groups below read these buffers)
        blocks = packs if self.block_api else None
        M1, M2, LL = Bc * L, Bc * L * L, L * L
        CS, CZ, E = c.seq_channel, c.pair_channel, c.index_embed_size
        WS_, WZ = CS + E, CZ + 2 * E
        seq_t = st['seq_t'][b0:b1]
        mask_f = st['mask_f'][b0:b1]
        fixed = st['fixed_i32'][b0:b1]
        sstat, pstat = st['static']
        if sstat.shape[0] != 1:
            sstat, pstat = sstat[b0:b1], pstat[b0:b1]
        temb = st['temb'][b0:b1]
        seq_act = st['rep_seq_out'][b0:b1]                   # (Bc,L,544) view, contiguous
        pair_act = st['rep_pair_out'][b0:b1]                 # (Bc,L,L,192)
        prev_seq = st['prev_seq'][b0:b1] if st['prev_seq'] is not None else None
        prev_pair = st['prev_pair'][b0:b1] if st['prev_pair'] is not None else None
        prev_pos = st['prev_pos'][b0:b1] if st['prev_pos'] is not None else None
        ops.assemble_seq(sstat, P.sd[P_SEQF + 'proj_aa_type.weight'], seq_t, Lab, temb, prev_seq,
                         *P.ln(P_SEQF + 'prev_seq_norm'), seq_act, Bc, L, CS, E)
        H = c.seqformer.seq_attention_with_pair_bias.num_head
        biasT = ws.get('biasT', (Bc, H, LL))
        pre = P_BLK + 'seq_attn.'
        fused_bias = False
        if not P.exact(P.plain_class) and CZ == 128 and E == 32 and H == 32 and not _NO_ASSEMBLE_BIAS:
            # round 6: the assembly and the sequence attention's pair bias (proj_pair(pair_norm(z0)), seqformer.py:324-333) from ONE pass over the
            # pair rows - the rows are in registers in the fragment layout of the matrix cores anyway; the projection's 9.5 GB read of z0 is gone
            bw, bcs, bbias, bw3 = P.ln_linear(pre + 'proj_pair', pre + 'pair_norm')
            if bw3 is None:
                bw3 = P.split_narrow(bw, (pre + 'proj_pair', pre + 'pair_norm'))
            if bw3 is not None and tuple(bw3.shape) == (12, 2, 32, 16):
                ops.assemble_pair_bias(pstat, temb, prev_pair, *P.ln(P_SEQF + 'prev_pair_norm'), prev_pos, P.sd[P_SEQF + 'proj_prev_pos.weight'],
                                       pair_act, bw3, bcs, bbias, biasT, Bc, L, range_class=P.plain_class)
                fused_bias = True
        if not fused_bias:
            ops.assemble_pair(pstat, temb, prev_pair, *P.ln(P_SEQF + 'prev_pair_norm'), prev_pos,
                              P.sd[P_SEQF + 'proj_prev_pos.weight'], pair_act, Bc, L, CZ, E)
        if st.get('esm_embed') is not None:
            # seqformer.py:185-191: softmax layer mix of the ESM2 representations (host-side tensor plumbing of the embedding
            # hook), then LayerNorm -> Linear -> ReLU -> Linear, added to the antibody rows' aa-type embedding
            e = st['esm_embed'][b0:b1]
            wl = torch.softmax(P.sd[P_SEQF + 'esm_embed_weights'], dim=-1)
            mixed = torch.einsum('blcn,n->blc', e.to(wl.dtype), wl).reshape(Bc * Lab, -1).contiguous()
            h1 = ws.get('esm_h', (Bc * Lab, CS))
            _ln_lin(P, P_SEQF + 'proj_esm_embed.1', P_SEQF + 'proj_esm_embed.0', None, mixed, h1, act=1)
            tgt = seq_act[:, :Lab, :CS]                        # (Bc, Lab, 512) window of the (Bc, L, 544) rows
            _lin(P, P_SEQF + 'proj_esm_embed.3', h1.view(Bc, Lab, CS), tgt, resid=tgt)
        s2 = seq_act.view(M1, WS_)
        z2 = pair_act.view(M2, WZ)
        z3 = pair_act.view(Bc, LL, WZ)
        if self.block_api:
            # the 768-wide scratch IS the q | k | v (+ exact-path hidden) region at the head of the triangle-attention block workspace (no
            # second copy).  One workspace per (group, L), sized for the LARGEST chunk seen and reused by smaller ones (the last chunk of a
            # batch that the chunk size does not divide: keyed by the chunk size the multi-GB buffers were freed and re-allocated twice per pass)
            attn_ws = self._block_workspace('attn', Bc, L)
            w768 = attn_ws[:M2 * 768 * 4].view(torch.float32).view(M2, 768)
        else:
            w768 = ws.get('w768', (M2, 768))
        w384 = ws.get('w384', (M2 * 384,))
        pmask = ws.get('pmask', (M2,))
        ops.pair_mask(mask_f, pmask, Bc, L)

        # ---------------- seq attention with pair bias (seqformer.py:314-356)
        pre = P_BLK + 'seq_attn.'
        if not fused_bias:
            _ln_lin(P, pre + 'proj_pair', pre + 'pair_norm', None, z3, biasT.transpose(1, 2), narrow=True)
        qkv = ws.get('s_a', (M1, 3 * WS_))
        sgate = ws.get('s_b', (M1, WS_))
        so = ws.get('s_c', (M1, WS_))
        _ln_lin(P, pre + 'attn.proj_in', pre + 'seq_norm', None, s2, qkv)
        _ln_lin(P, pre + 'attn.gate', pre + 'seq_norm', None, s2, sgate)
        ops.seq_attn(qkv, biasT, mask_f, sgate, so, Bc, L, H, WS_ // H)
        _lin(P, pre + 'attn.proj_out', so, s2, resid=s2)
        # ---------------- seq transition
        pre = P_BLK + 'seq_transition.transition.'
        hid = ws.get('s_a', (M1, 4 * WS_))
        _ln_lin(P, pre + '1', pre + '0', None, s2, hid, act=1)
        _lin(P, pre + '3', hid, s2, resid=s2)
        # ---------------- outer product mean (seqformer.py:395-411)
        pre = P_BLK + 'outer_product_mean.'
        lr = ws.get('s_b', (M1, 128))
        _ln_lin(P, pre + 'lr', pre + 'norm', None, s2, lr, rowscale=mask_f.reshape(-1))
        if not P.exact(P.plain_class) and tuple(P.wt[pre + 'out_proj'].shape) == (128, 192) and not _NO_OPM_FUSED:
            # round 6: one kernel, the (Bc, L, L, 128) feature tensor never exists - z += l_j . (diag(r_i) W1 + W2) + (b - r_i . W2), a workgroup
            # per (b, i) row of the pair tensor; HBM traffic = z read + z written (19 GB instead of 31.7 at 100 samples of L = 352)
            ops.opm_out(lr, P.wt[pre + 'out_proj'], P.b.get(pre + 'out_proj'), z2, Bc, L, range_class=P.plain_class)
        else:
            feat = w384[:M2 * 128].view(M2, 128)
            ops.opm_features(lr, feat, Bc, L, 64)
            _lin(P, pre + 'out_proj', feat, z2, resid=z2)
        # ---------------- triangle multiplication (seqformer.py:443-504)
        # Large problems run the contraction on the split-f16 kernels: the projections write left/right directly as the
        # k-tiled f16 operand images of the contraction (C_split), and the incoming variant reads z pair-transposed
        # (a_pair_transpose) so that both einsums become the same row-major 'ik,jk->ij' product.
        # Any residue count: inside the triangle multiplication the pair positions are indexed with a row stride Lp = L rounded
        # up to 4 (m' = i*Lp + j), so that plane rows, float4 stores and the channel-major product stay 16-byte aligned; the
        # projections read / the output projection writes the unpadded pair tensor through the row maps a_pair / c_pair, and
        # the padded row scale zeroes the pad columns of the contraction operands.
        planes = not P.exact('plane_projection', 'contraction', 'tri_mul_tail')
        Lp = (L + 3) // 4 * 4
        LLp = L * Lp
        pad = (L, Lp) if Lp != L else None
        for name, outgoing in (('triangle_multiplication_outgoing', True), ('triangle_multiplication_incoming', False)):
            pre = P_BLK + name + '.'
            if blocks is not None and planes:
                # one call of the C ABI per module: out-of-place (z3 -> the 768-wide workspace -> z3 for the two variants)
                zb = w768.view(-1)[:Bc * LL * 192].view(Bc, LL, 192)
                zin, zout = (z3, zb) if outgoing else (zb, z3)
                ops.tri_mul_fwd(blocks[name], zin, zout, mask_f, Bc, L, outgoing, self._block_workspace('mul', Bc, L))
                continue
            # sigmoid(left_gate | right_gate) channel-major like the projections they gate; sigmoid(final_gate) row-major
            GT = w768.view(-1)[:Bc * 256 * LL].view(Bc, 256, LL)
            Gf = w768.view(-1)[Bc * 256 * LL:Bc * 448 * LL].view(Bc, LL, 192)
            if planes:
                # The dual output GEMM reads z (gate operand + residual) for ALL 192 input channels while other tiles of the same
                # rows write their output columns, so it must not run in place: the outgoing variant writes into the free
                # 768-wide workspace, the incoming one reads that and writes the pair tensor again.
                zb = w768.view(-1)[:Bc * LL * 192].view(Bc, LL, 192)
                zin, zout = (z3, zb) if outgoing else (zb, z3)
                # one GEMM: [left | right] projections * sigmoid(their gates) * pair mask -> plane operands of the contraction
                KT = (Lp + 15) // 16
                lrp = ws.get('tm_lr', (Bc, 256, KT, 2, L, 16), torch.int16, zero=(Lp % 16 != 0))
                if pad is None:
                    pm = pmask
                else:
                    pm = ws.get('pmask_p', (Bc * LLp,))
                    ops.pair_mask(mask_f, pm, Bc, L, Lp)
                # (GEMM rows in (8 i x 16 k) block order: 64 contiguous plane bytes per store instruction - half the HBM write bytes)
                _ln_lin(P, pre + 'lr_glu', pre + 'norm', None, zin, lrp, rowscale=pm, glu=True, c_split_nA=128, c_split_tile=True,
                        a_pair_transpose=0 if outgoing else L, pair=(L, Lp), a_pair=True)
                tt = w384[:Bc * 128 * LLp].view(Bc, 128, LLp)      # channel-major product, padded pair rows (pads: never stored)
                tz = tt.as_strided((Bc * 128, L, L), (LLp, Lp, 1))
                ops.gemm(lrp[:, 0:128], lrp[:, 128:256], tz, exact=2)
                # output projection and final gate in ONE dual GEMM: proj_out(final_norm(product)) * sigmoid(final_gate(norm(z))) + z;
                # the gate never exists in HBM and z is read once for the gate and the residual
                gw, gcs, gb, gw3 = P.ln_linear(pre + 'final_gate', pre + 'norm')
                _ln_lin(P, pre + 'proj_out', pre + 'final_norm', None, tt.transpose(1, 2), zout, resid=zin, pair=pad, c_pair=pad is not None,
                        dual=(zin, gw3, gcs, gb))
            else:
                _ln_lin(P, pre + 'final_gate', pre + 'norm', None, z3, Gf, act=2)
                tt = w384[2 * Bc * 128 * LL:3 * Bc * 128 * LL].view(Bc, 128, LL)
                tz = tt.view(Bc * 128, L, L)
                _ln_lin(P, pre + 'lr_gates', pre + 'norm', None, z3, GT.transpose(1, 2), act=2)
                left = w384[0:Bc * 128 * LL].view(Bc, 128, LL)
                right = w384[Bc * 128 * LL:2 * Bc * 128 * LL].view(Bc, 128, LL)
                _ln_lin(P, pre + 'left_proj', pre + 'norm', None, z3, left.transpose(1, 2), rowscale=pmask,
                        gate=GT[:, 0:128].transpose(1, 2), gate_sigmoid=False)
                _ln_lin(P, pre + 'right_proj', pre + 'norm', None, z3, right.transpose(1, 2), rowscale=pmask,
                        gate=GT[:, 128:256].transpose(1, 2), gate_sigmoid=False)
                lz = left.view(Bc * 128, L, L)
                rz = right.view(Bc * 128, L, L)
                if outgoing:      # 'bikc,bjkc->bijc'
                    ops.gemm(lz, rz.transpose(1, 2), tz, exact=1)
                else:             # 'bkic,bkjc->bijc'
                    ops.gemm(lz.transpose(1, 2), rz, tz, exact=1)
                tcm = tt.transpose(1, 2)                                   # (Bc, LL, 128) logical, channel-major storage
                _ln_lin(P, pre + 'proj_out', pre + 'final_norm', None, tcm, z3, gate=Gf, gate_sigmoid=False, resid=z3)
        # ---------------- triangle attention (seqformer.py:506-550)
        for name, per_row in (('triangle_attention_starting_node', True), ('triangle_attention_ending_node', False)):
            pre = P_BLK + name + '.'
            ax = P.exact('tri_attn')                       # (small complexes: exact GEMMs - too few rows for the split tiles - and the
            if blocks is not None:                          # split-f16 attention, which has no size limit; a flagged class: all exact)
                flagged = bool(P.exact_tags & ops.RANGE_TAGS['tri_attn'])
                ops.tri_attn_block_fwd(blocks[name], z2, mask_f, Bc, L, per_row, attn_ws, exact=ax, attn_exact=ops.GEMM_EXACT or flagged)
                continue
            am = 1 if ax else 2
            ae = True if (P.exact_tags & ops.RANGE_TAGS['tri_attn']) else None
            # q | k | v and the pair bias (b, h, i, j) read the same LayerNorm(z) rows: one launch, the bias columns in the free half of the
            # projection's last column tile (ops.gemm_side; two launches when the pair does not qualify - exact arithmetic, small problems)
            bT = ws.get('biasT', (Bc, 4, LL))
            qkv = w768.view(-1)[:M2 * 576].view(M2, 576)
            # (the split-f16 attention takes the bias in its accumulator units: the factor rides in the projection's epilogue)
            bl2 = not (ops.GEMM_EXACT if ae is None else ae)
            # (round-6 experiment, as abx_tri_attn_block_fwd, OFF unless ABX_KV_PLANES is set: the k | v columns leave the projection as the
            # operand images the attention stages by DMA - bit-identical to the fp32 route, measured slower: profiles/r06h_kb_kvplanes.txt)
            kvp = ops.KV_PLANES and am == 2 and bl2 and ops.kv_planes_ok(M2)
            # (range class 'tri_attn' as abx_tri_attn_block_fwd tags them: the tag of a launch follows the switch that sets its arithmetic)
            ops.gemm_side(_ln_lin(P, pre + 'qkv', pre + 'norm', None, z2, qkv, defer=True, exact=am, range_class='tri_attn',
                                  c_plane_cols=(192, 48) if kvp else None),
                          _ln_lin(P, pre + 'proj_pair', pre + 'norm', None, z3, bT.transpose(1, 2), narrow=True, defer=True, exact=am,
                                  alpha=ops.TRI_BIAS_LOG2 if bl2 else 1.0, range_class='tri_attn'))
            o = w384[:M2 * 192].view(M2, 192)
            # bias[b,h,q,k] key-contiguous in rows of Lp floats (16-byte loads for any L).  Ending node: bias[b,h,q,k] = P[b,k,q,h],
            # i.e. the transpose (2 MB per sample); starting node: a padded copy only when L % 4 != 0
            if not per_row or Lp != L:
                bT2 = ws.get('biasT2', (Bc, 4, L, Lp))
                ops.transpose_last2(bT.view(Bc * 4, L, L), bT2.view(Bc * 4, L, Lp), transpose=not per_row)
                bT = bT2
            else:
                bT = bT.view(Bc, 4, L, L)
            ops.tri_attn(qkv, bT, mask_f, o, Bc, L, per_row, bias_is_qk=True, exact=ae, bias_log2=bl2, kv_planes=kvp)       # (no gate: the tail applies it)
            if not ax:
                # gate projection, sigmoid, * attention output, output projection, + residual in ONE kernel (AbxGemm.mlp = 2)
                _ln_lin(P, pre + 'attn.gate', pre + 'norm', None, z2, z2, act=2, gate=o, resid=z2, mlp=P.mlp_second(pre + 'attn.proj_out'), exact=2)
            else:
                hid = w768.view(-1)[M2 * 576:M2 * 768].view(M2, 192)
                _ln_lin(P, pre + 'attn.gate', pre + 'norm', None, z2, hid, act=2, gate=o, gate_sigmoid=False, exact=1)
                _lin(P, pre + 'attn.proj_out', hid, z2, resid=z2, exact=1)
        # ---------------- pair transition
        pre = P_BLK + 'pair_transition.transition.'
        tx = P.exact('pair_transition')
        if blocks is not None:
            ops.transition_fwd(*blocks['pair_transition'], z2, exact=tx, workspace=w768)
        elif not tx:
            # LayerNorm -> Linear -> ReLU -> Linear + residual in ONE kernel: the 768-wide hidden never travels through HBM
            _ln_lin(P, pre + '1', pre + '0', None, z2, z2, act=1, resid=z2, mlp=P.mlp_second(pre + '3'), exact=2)
        else:
            _ln_lin(P, pre + '1', pre + '0', None, z2, w768, act=1, exact=1)
            _lin(P, pre + '3', w768, z2, resid=z2, exact=1)

        # ================= IpaScore (score_network.py:83-196)
        P.plain_class = 'gemm_late'
        ic = cfg.heads.diffusion_module.IPA
        NC = ic.num_channel
        s_pre = ws.get('i_a', (M1, NC))
        _lin(P, P_IPA + 'proj_init_seq_act', s2, s_pre)
        s0 = ws.get('i_s0', (M1, NC))
        ops.layernorm(s_pre, *P.ln(P_IPA + 'init_seq_layer_norm'), out=s0)
        s = ws.get('i_s', (M1, NC))
        _lin(P, P_IPA + 'proj_seq', s0, s)
        zi = w384[:M2 * 128].view(M2, 128)
        if not P.exact('ipa_pair_init'):
            # Linear -> LayerNorm in one kernel (the 128 output channels of a row sit in one wave tile of the split-f16 GEMM)
            _lin(P, P_IPA + 'proj_init_pair_act', z2, zi, out_ln=P.ln(P_IPA + 'init_pair_layer_norm'), exact=2)
        else:
            _lin(P, P_IPA + 'proj_init_pair_act', z2, zi, exact=1)
            ops.layernorm(zi, *P.ln(P_IPA + 'init_pair_layer_norm'), out=zi)
        bias2d = w384[M2 * 128:M2 * 140].view(M2, 12)
        _lin(P, P_IPA + 'attention_module.proj_pair', zi, bias2d, alpha=P.ipa_w2d, narrow=True)
        attn_ws = w384[M2 * 140:M2 * 152].view(M2, 12)
        init_q = ws.get('f_iq', (M1, 4)); init_t = ws.get('f_it', (M1, 3))
        cur_q = ws.get('f_q', (M1, 4)); cur_t = ws.get('f_t', (M1, 3)); cur_R = ws.get('f_R', (M1, 9))
        delta_q = ws.get('f_dq', (M1, 4))
        rig_in = st['rigids_t'][b0:b1].contiguous()
        ops.frames_init(rig_in, init_q, init_t, cur_q, cur_t, cur_R, delta_q, M1, ic.position_scale)
        proj = ws.get('i_proj', (M1, 1152))
        qpack = ws.get('i_qp', (ops.ipa_qpack_numel(Bc, L),)); kpack = ws.get('i_kp', (M1 * 12 * 28,)); vpack = ws.get('i_vp', (M1 * 12 * 40,))
        ifeat = ws.get('i_feat', (M1, 2112))
        h1 = ws.get('i_h1', (M1, NC)); h2 = ws.get('i_h2', (M1, NC))
        upd = ws.get('i_upd', (M1, 6))
        tail = None
        if not P.exact('ipa_tail', 'gemm_late') and NC == 256:
            wb = lambda n: (P.split(P_IPA + n), P.b[P_IPA + n])
            tail = (wb('attention_module.final_proj'), P.ln(P_IPA + 'attention_layer_norm'), wb('transition_module.0'),
                    wb('transition_module.2'), wb('transition_module.4'), P.ln(P_IPA + 'transition_layer_norm'))
        for _ in range(ic.num_layer):
            _lin(P, P_IPA + 'attention_module.proj', s, proj)
            ops.ipa_pack(proj, cur_R, cur_t, qpack, kpack, vpack, Bc, L, P.ipa_ws)
            ops.ipa_weights(qpack, kpack, vpack, bias2d, mask_f, cur_R, cur_t, P.ipa_pw, attn_ws, ifeat, Bc, L)
            ops.ipa_pair(attn_ws, zi, ifeat, Bc, L)
            if tail is not None:
                # final_proj + residual + LayerNorm + the three-layer transition + residual + LayerNorm + affine_update + frame update:
                # one launch, the 256-wide activations stay on the CU (eight launches otherwise).  final_proj (K = 2 112) runs in front of
                # it as a split-K GEMM of 11 K-slices (one launch, batch = slice) whose products the tail adds in slice order: the
                # tail's 32-row blocks would otherwise walk 132 k-steps each - at every batch size, so that a sample's numbers do not
                # depend on how many samples share the launch
                part = None
                if self.ipa_splitk and ifeat.shape[1] % (16 * self.ipa_splitk) == 0:
                    part = ops.gemm_splitk(ifeat, tail[0][0], ws.get('i_part', (self.ipa_splitk, M1, NC)), range_class='gemm_late')
                ops.ipa_tail(ifeat, s, *tail, affine=(P.wt[P_IPA + 'affine_update'], P.b[P_IPA + 'affine_update']),
                             rigid=(fixed.reshape(-1), init_q, init_t, cur_q, cur_t, cur_R, delta_q, ic.position_scale), partial=part)
                continue
            else:
                _lin(P, P_IPA + 'attention_module.final_proj', ifeat, s, resid=s)
                ops.layernorm(s, *P.ln(P_IPA + 'attention_layer_norm'), out=s)
                _lin(P, P_IPA + 'transition_module.0', s, h1, act=1)
                _lin(P, P_IPA + 'transition_module.2', h1, h2, act=1)
                _lin(P, P_IPA + 'transition_module.4', h2, s, resid=s)
                ops.layernorm(s, *P.ln(P_IPA + 'transition_layer_norm'), out=s)
            _lin(P, P_IPA + 'affine_update', s, upd)
            ops.rigid_update(upd, fixed.reshape(-1), init_q, init_t, cur_q, cur_t, cur_R, delta_q, M1, ic.position_scale)
        # torsions (sidechain.py:28-72)
        pre = P_IPA + 'sidechain_module.torsion_module.'
        ta = ws.get('t_a', (M1, 128)); tb = ws.get('t_b', (M1, 128))
        un = ws.get('t_un', (M1, 14))
        logits = st['logits'][b0:b1]
        pl = ws.get('h_pl', (M1, 50))
        # torsion ResNet + SequenceHead MLP (+ PredictedLDDTHead MLP on the last pass) in ONE launch: 13 (18) launches otherwise
        heads = P.heads_pack() if (self.fused_heads and not P.exact('heads_tail', 'gemm_late') and NC == 256 and ic.torsion.num_residual_block == 2 and
                                   logits.is_contiguous()) else None
        if heads is not None:
            ops.heads_tail(s, s0, heads[0], heads[1], heads[2], un, logits.view(M1, 20), pl if final else None)
        else:
            _lin(P, pre + 'proj_act.1', s, ta, a_relu=True)
            _lin(P, pre + 'proj_init_act.1', s0, ta, a_relu=True, resid=ta)
            for blk in range(ic.torsion.num_residual_block):
                _lin(P, pre + f'blocks.{blk}.net.1', ta, tb, a_relu=True)
                _lin(P, pre + f'blocks.{blk}.net.3', tb, ta, a_relu=True, resid=ta)
            _lin(P, pre + 'projection', ta, un, a_relu=True)
        angles = st['angles'][b0:b1]
        ops.torsion_finalize(un, st['torsion_gt'][b0:b1].contiguous(), fixed.reshape(-1), angles, M1)
        sm = st['structure_module'][b0:b1]
        sm.view(M1, NC).copy_(s)
        t64 = st['t64'][b0:b1].contiguous()
        D = st['diffuser']
        ops.scores(init_q=init_q, init_t=init_t, delta_q=delta_q, cur_t=cur_t, fixed_mask=fixed.reshape(-1), t=t64,
                   t_is_f32=int(st['t_is_f32']), score_norms=D.score_norms, num_sigma=D.num_sigma, num_omega=D.num_omega,
                   discrete_sigma=D.discrete_sigma_dev, discrete_omega=D.discrete_omega_dev,
                   exp_max_sigma=D.exp_max_sigma, exp_min_sigma=D.exp_min_sigma, min_b=D.min_b_f32, bdiff=D.bdiff_f32,
                   coord_scale=D.coord_scale_f32, position_scale=float(ic.position_scale),
                   rot_score=st['rot_score'][b0:b1], trans_score=st['trans_score'][b0:b1], rigids=st['rigids'][b0:b1],
                   B=Bc, L=L)
        # ---------------- sequence head (head.py:162-201)
        pre = 'impl.sequence_module.net.'
        hx = ws.get('h_x', (M1, NC))
        if heads is None:
            ops.layernorm(s, *P.ln(pre + '0'), out=hx)
            _lin(P, pre + '1', hx, ta, act=1)
            _lin(P, pre + '3', ta, tb, act=1)
            _lin(P, pre + '5', tb, logits.view(M1, 20))
        ops.seq_head_atoms(logits, fixed.reshape(-1), seq_t.contiguous(), st['rigids'][b0:b1], angles,
                           st['a37to14'][b0:b1].contiguous(), P.default_frames, P.group_idx, P.lit_pos,
                           st['seq_0'][b0:b1], st['atom14'][b0:b1], st['atom37'][b0:b1], M1)
        if final:
            if heads is None:
                pre = 'impl.predicted_lddt.net.'
                ops.layernorm(s, *P.ln(pre + '0'), out=hx)
                _lin(P, pre + '1', hx, ta, act=1)
                _lin(P, pre + '3', ta, tb, act=1)
                _lin(P, pre + '5', tb, pl)
            ops.plddt(pl, st['pLDDT'][b0:b1], M1, 50)
        # ---------------- self-conditioning distogram for the next call (abx.py:17-26)
        ops.prev_pos(st['atom37'][b0:b1], self.sq_breaks, st['prev_pos_out'][b0:b1], Bc, L)
