"""Parameter containers with the reference's module tree, so that `state_dict()` has exactly the reference's 190
keys/shapes (ESM disabled) and `load_state_dict(ckpt['model_state_dict'], strict=True)` works unchanged
(SURVEY.md §8b "checkpoint contract"; reference constructors: abx/model/seqformer.py:123-168,228-258,314-606,
abx/model/encoder.py:123-229, abx/model/score_network.py:42-78, abx/model/folding.py:23-45,
abx/model/sidechain.py:16-53, abx/model/head.py:26-36,146-160,205-220).

These modules hold parameters only; the computation is in abx_amd/model/forward.py on HIP kernels.
"""
import math

import torch
from torch import nn


def _lin(i, o, bias=True):
    return nn.Linear(i, o, bias=bias)


def _mlp(*layers):
    return nn.Sequential(*layers)


class _Holder(nn.Module):
    """A module that only names sub-modules / parameters."""

    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError('parameter container: use abx_amd.model.abx.ScoreNetwork')


class ResidueEmbedding(_Holder):
    def __init__(self, feat=512):
        super().__init__()
        self.aatype_embed = nn.Embedding(23, feat)
        self.cdr_embed = nn.Embedding(15, feat)
        self.coordinate_embed = _mlp(_lin(56, feat), nn.ReLU(), _lin(feat, feat))
        self.mlp = _mlp(_lin(feat * 3 + 2, feat * 2), nn.ReLU(), _lin(feat * 2, feat), nn.ReLU(), _lin(feat, feat), nn.ReLU(),
                        _lin(feat, feat))


class PairEmbedding(_Holder):
    def __init__(self, feat=128, num_bins=15):
        super().__init__()
        self.aa_pair_embed = nn.Embedding(529, feat)
        self.relpos_embed = nn.Embedding(65, feat)
        self.aapair_to_distcoef = nn.Embedding(529, 196)
        self.distance_embed = _mlp(_lin(196, feat), nn.ReLU(), _lin(feat, feat), nn.ReLU())
        self.dgram_embed = nn.Embedding(num_bins, feat)
        self.out_mlp = _mlp(_lin(feat * 4, feat), nn.ReLU(), _lin(feat, feat), nn.ReLU(), _lin(feat, feat))


class Attention(_Holder):
    def __init__(self, dim, split_first):
        super().__init__()
        if split_first:
            self.proj_q = _lin(dim, dim, False)
            self.proj_k = _lin(dim, dim, False)
            self.proj_v = _lin(dim, dim, False)
        else:
            self.proj_in = _lin(dim, dim * 3, False)
        self.gate = _lin(dim, dim)
        self.proj_out = _lin(dim, dim)


class SeqAttentionWithPairBias(_Holder):
    def __init__(self, cs, cz, heads):
        super().__init__()
        self.seq_norm = nn.LayerNorm(cs)
        self.pair_norm = nn.LayerNorm(cz)
        self.proj_pair = _lin(cz, heads, False)
        self.attn = Attention(cs, split_first=False)


class Transition(_Holder):
    def __init__(self, c, factor):
        super().__init__()
        self.transition = _mlp(nn.LayerNorm(c), _lin(c, c * factor), nn.ReLU(), _lin(c * factor, c))


class OuterProductMean(_Holder):
    def __init__(self, cs, cz, co):
        super().__init__()
        self.norm = nn.LayerNorm(cs)
        self.left_proj = _lin(cs, co)
        self.right_proj = _lin(cs, co)
        self.out_proj = _lin(2 * co, cz)


class TriangleMultiplication(_Holder):
    def __init__(self, cz, ci):
        super().__init__()
        self.norm = nn.LayerNorm(cz)
        self.left_proj = _lin(cz, ci)
        self.right_proj = _lin(cz, ci)
        self.final_norm = nn.LayerNorm(ci)
        self.left_gate = _lin(cz, ci)
        self.right_gate = _lin(cz, ci)
        self.final_gate = _lin(cz, cz)
        self.proj_out = _lin(ci, cz)


class TriangleAttention(_Holder):
    def __init__(self, cz, heads):
        super().__init__()
        self.norm = nn.LayerNorm(cz)
        self.proj_pair = _lin(cz, heads, False)
        self.attn = Attention(cz, split_first=True)


class SeqformerIteration(_Holder):
    def __init__(self, c, cs, cz):
        super().__init__()
        self.seq_attn = SeqAttentionWithPairBias(cs, cz, c.seq_attention_with_pair_bias.num_head)
        self.seq_transition = Transition(cs, c.seq_transition.num_intermediate_factor)
        self.outer_product_mean = OuterProductMean(cs, cz, c.outer_product_mean.num_outer_channel)
        self.triangle_multiplication_outgoing = TriangleMultiplication(cz, c.triangle_multiplication_outgoing.num_intermediate_channel)
        self.triangle_multiplication_incoming = TriangleMultiplication(cz, c.triangle_multiplication_incoming.num_intermediate_channel)
        self.triangle_attention_starting_node = TriangleAttention(cz, c.triangle_attention_starting_node.num_head)
        self.triangle_attention_ending_node = TriangleAttention(cz, c.triangle_attention_ending_node.num_head)
        self.pair_transition = Transition(cz, c.pair_transition.num_intermediate_factor)


class Seqformer(_Holder):
    def __init__(self, c):
        super().__init__()
        cs, cz = c.seq_channel + c.index_embed_size, c.pair_channel + 2 * c.index_embed_size
        self.blocks = nn.ModuleList([SeqformerIteration(c.seqformer, cs, cz) for _ in range(c.seqformer_num_block)])


class EmbeddingAndSeqformer(_Holder):
    def __init__(self, c):
        super().__init__()
        self.proj_aa_type = nn.Embedding(23, c.seq_channel, padding_idx=20)
        self.encode_residue_emb = ResidueEmbedding(c.seq_channel)
        self.encode_pair_emb = PairEmbedding(c.pair_channel, c.prev_pos.num_bins)
        self.aa_proj = _mlp(nn.LayerNorm(c.seq_channel), _lin(c.seq_channel, c.seq_channel), nn.ReLU(),
                            _lin(c.seq_channel, c.seq_channel))
        if c.esm.enabled:
            # seqformer.py:143-154.  The ESM2-3B module itself (`encode_esm_emb.model.*`) is NOT instantiated: its per-layer
            # representations arrive through the embedding hook (ScoreNetwork.esm_provider / batch['esm_embed'])
            self.esm_embed_weights = nn.Parameter(torch.zeros(c.esm.num_layers + 1))
            self.proj_esm_embed = _mlp(nn.LayerNorm(c.esm.embed_channel), _lin(c.esm.embed_channel, c.seq_channel), nn.ReLU(),
                                       _lin(c.seq_channel, c.seq_channel))
        self.proj_rel_pos = nn.Embedding(c.max_relative_feature * 2 + 2, c.pair_channel)
        self.prev_seq_norm = nn.LayerNorm(c.seq_channel + c.index_embed_size)
        self.prev_pair_norm = nn.LayerNorm(c.pair_channel + 2 * c.index_embed_size)
        self.proj_prev_pos = nn.Embedding(c.prev_pos.num_bins, c.pair_channel + 2 * c.index_embed_size)
        self.seqformer = Seqformer(c)


class InvariantPointAttention(_Holder):
    def __init__(self, c, cz):
        super().__init__()
        H = c.num_head
        self.trainable_point_weights = nn.Parameter(torch.full((H,), math.log(math.e - 1.0)))
        self.proj_q_scalar = _lin(c.num_channel, H * c.num_scalar_qk)
        self.proj_kv_scalar = _lin(c.num_channel, H * (c.num_scalar_v + c.num_scalar_qk))
        self.proj_q_point_local = _lin(c.num_channel, 3 * H * c.num_point_qk)
        self.proj_kv_point_local = _lin(c.num_channel, 3 * H * (c.num_point_v + c.num_point_qk))
        self.proj_pair = _lin(cz, H)
        self.final_proj = _lin(H * (c.num_scalar_v + cz + c.num_point_v * 4), c.num_channel)


class ResNetBlock(_Holder):
    def __init__(self, dim):
        super().__init__()
        self.net = _mlp(nn.ReLU(), _lin(dim, dim), nn.ReLU(), _lin(dim, dim))


class TorsionModule(_Holder):
    def __init__(self, c, cin):
        super().__init__()
        self.proj_act = _mlp(nn.ReLU(), _lin(cin, c.num_channel))
        self.proj_init_act = _mlp(nn.ReLU(), _lin(cin, c.num_channel))
        self.blocks = nn.Sequential(*[ResNetBlock(c.num_channel) for _ in range(c.num_residual_block)])
        self.projection = _lin(c.num_channel, 14)


class MultiRigidSidechain(_Holder):
    def __init__(self, c):
        super().__init__()
        self.torsion_module = TorsionModule(c.torsion, c.num_channel)


class IpaScore(_Holder):
    def __init__(self, config, cs, cz):
        super().__init__()
        c = config.IPA
        e = config.embed.index_embed_size
        self.proj_init_seq_act = _lin(cs + e, c.num_channel)
        self.proj_init_pair_act = _lin(cz + 2 * e, cz)
        self.init_seq_layer_norm = nn.LayerNorm(c.num_channel)
        self.init_pair_layer_norm = nn.LayerNorm(cz)
        self.proj_seq = _lin(c.num_channel, c.num_channel)
        self.attention_module = InvariantPointAttention(c, cz)
        self.attention_layer_norm = nn.LayerNorm(c.num_channel)
        layers = []
        for k in range(c.num_layer_in_transition):
            layers.append(_lin(c.num_channel, c.num_channel))
            if k != c.num_layer_in_transition - 1:
                layers.append(nn.ReLU())
        self.transition_module = nn.Sequential(*layers)
        self.transition_layer_norm = nn.LayerNorm(c.num_channel)
        self.affine_update = _lin(c.num_channel, 6)
        self.sidechain_module = MultiRigidSidechain(c)


class DiffusionHead(_Holder):
    def __init__(self, config, cs, cz):
        super().__init__()
        self.ScoreNetwork = IpaScore(config, cs, cz)


class MlpHead(_Holder):
    def __init__(self, c, nout):
        super().__init__()
        d, h = c.num_channel, c.num_hidden_channel
        self.net = _mlp(nn.LayerNorm(d), _lin(d, h), nn.ReLU(), _lin(h, h), nn.ReLU(), _lin(h, nout))


class DistogramHead(_Holder):
    def __init__(self, c, cz):
        super().__init__()
        self.proj = _lin(cz + 2 * c.index_embed_size, c.num_bins)


class ScoreNetworkIteration(_Holder):
    def __init__(self, model_conf):
        super().__init__()
        c = model_conf.embeddings_and_seqformer
        self.seqformer = EmbeddingAndSeqformer(c)
        h = model_conf.heads
        # registration order of the reference's HeaderBuilder (head.py:230-237)
        if 'diffusion_module' in h:
            self.diffusion_module = DiffusionHead(h.diffusion_module, c.seq_channel, c.pair_channel)
        if 'sequence_module' in h:
            self.sequence_module = MlpHead(h.sequence_module, 20)
        if 'distogram' in h:
            self.distogram = DistogramHead(h.distogram, c.pair_channel)
        if 'predicted_lddt' in h:
            self.predicted_lddt = MlpHead(h.predicted_lddt, 50)
