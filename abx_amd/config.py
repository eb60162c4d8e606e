"""Configuration for the hot path.

`load_config(path)` accepts the reference's `config/config_model.json` unchanged (only the `model` and `diffuser`
sections are read; reference inference.py:92-99).  `default_config()` is the build's own statement of the
hyper-parameters the path uses (values as in the reference file, `loss` section omitted: it is training-only).
"""
import copy
import json


class AttrDict(dict):
    """Recursive attribute-access dict: stands in for ml_collections.ConfigDict (absent in this image).
    Supports attribute access, cfg[name], `name in cfg`, .get and ** expansion."""

    def __init__(self, d=None, **kw):
        super().__init__()
        for k, v in dict(d or {}, **kw).items():
            self[k] = AttrDict(v) if isinstance(v, dict) and not isinstance(v, AttrDict) else v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    def __deepcopy__(self, memo):
        return AttrDict({k: copy.deepcopy(v, memo) for k, v in self.items()})


def _tri(orientation, **kw):
    d = dict(orientation=orientation, inp_kernels=[], dropout_rate=0.1, shared_dropout=False)
    d.update(kw)
    return d


_DEFAULT = {
    'model': {
        'num_atom': 5,
        'num_recycle': 2,
        'embeddings_and_seqformer': {
            'seqformer_num_block': 1, 'seq_channel': 512, 'pair_channel': 128, 'max_relative_feature': 32,
            'index_embed_size': 32,
            'esm': {'enabled': False, 'embed_channel': 2560, 'num_layers': 36, 'dropout_rate': 0.1, 'norm': True},
            'recycle_features': True, 'recycle_pos': True,
            'prev_pos': {'min_bin': 3.375, 'num_bins': 15, 'max_bin': 21.375},
            'seqformer': {
                'seq_attention_with_pair_bias': dict(orientation='per_row', num_head=32, inp_kernels=[],
                                                     dropout_rate=0.1, shared_dropout=True),
                'seq_transition': dict(orientation='per_row', num_intermediate_factor=4, dropout_rate=0,
                                       shared_dropout=True),
                'outer_product_mean': dict(orientation='per_row', num_outer_channel=64, dropout_rate=0,
                                           shared_dropout=True),
                'triangle_multiplication_outgoing': _tri('per_row', num_intermediate_channel=128, gating=True,
                                                         num_head=4),
                'triangle_multiplication_incoming': _tri('per_column', num_intermediate_channel=128, gating=True,
                                                         num_head=4),
                'triangle_attention_starting_node': _tri('per_row', num_head=4, gating=True),
                'triangle_attention_ending_node': _tri('per_column', num_head=4, gating=True),
                'pair_transition': dict(orientation='per_row', num_intermediate_factor=4, dropout_rate=0,
                                        shared_dropout=True),
            },
        },
        'heads': {
            'diffusion_module': {
                'Path_score': False, 'coordinate_scaling': 0.1, 'num_blocks': 4, 'node_embed_size': 256,
                'edge_embed_size': 128,
                'embed': {'index_embed_size': 32, 'num_bins': 22, 'min_bin': 1e-5, 'max_bin': 20.0,
                          'embed_self_conditioning': True},
                'IPA': {'num_layer': 8, 'position_scale': 10,
                        'torsion': {'num_residual_block': 2, 'atom_clamp_distance': 10, 'num_channel': 128},
                        'num_layer_in_transition': 3, 'clash_overlap_tolerance': 1.5, 'num_head': 12,
                        'num_channel': 256, 'num_scalar_qk': 16, 'num_scalar_v': 16, 'num_point_qk': 4,
                        'num_point_v': 8, 'dropout': 0.1},
            },
            'predicted_lddt': {'num_channel': 256, 'num_hidden_channel': 128, 'index_embed_size': 32},
            'sequence_module': {'num_channel': 256, 'num_hidden_channel': 128, 'index_embed_size': 32},
            'distogram': {'first_break': 2.3125, 'last_break': 21.6875, 'num_bins': 64, 'index_embed_size': 32},
            'tmscore': {'num_atom': 5},
            'metric': {},
        },
    },
    'diffuser': {
        'inference_step': 100,
        'diffuse': {'diffuse_trans': True, 'diffuse_rot': True, 'diffuse_seq': True},
        'r3': {'min_b': 0.1, 'max_b': 20.0, 'coordinate_scaling': 0.1},
        'so3': {'num_omega': 1000, 'num_sigma': 1000, 'min_sigma': 0.1, 'max_sigma': 1.5,
                'schedule': 'logarithmic', 'cache_dir': '.cache/', 'use_cached_score': True},
        'seq': {'rate_const': 0.3},
    },
}


def default_config():
    return AttrDict(copy.deepcopy(_DEFAULT))


def load_config(path, esm_enabled=False):
    """Read the reference's config_model.json; force so3.use_cached_score=True as inference.py:99 does."""
    with open(path, 'r', encoding='utf-8') as f:
        cfg = json.load(f)
    cfg = AttrDict({k: cfg[k] for k in ('model', 'diffuser')})
    cfg.diffuser.so3.use_cached_score = True
    if not esm_enabled:
        cfg.model.embeddings_and_seqformer.esm.enabled = False
    return cfg
