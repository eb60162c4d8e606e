"""ctypes binding of libabx_hip.so (C ABI declared in include/abx_hip.h).

The product path has NO fallback: if the library is missing or a call fails, an exception is raised.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# ABX_HIP_LIB: an alternative build of the same C ABI (kernel experiments under tools/probes/); the default is the in-tree library
LIB_PATH = os.environ.get('ABX_HIP_LIB') or os.path.join(_HERE, 'csrc', 'libabx_hip.so')

c_f = C.c_void_p          # device pointers travel as integers
LL = C.c_longlong
I = C.c_int
F = C.c_float


class AbxGemm(C.Structure):
    _fields_ = [
        ('A', c_f), ('sAb', LL), ('sAm', LL), ('sAk', LL),
        ('B', c_f), ('sBb', LL), ('sBk', LL), ('sBn', LL),
        ('C', c_f), ('sCb', LL), ('sCm', LL),
        ('M', I), ('N', I), ('K', I), ('batch', I),
        ('c_transposed', I),
        ('ln_stats', c_f), ('sSb', LL),
        ('ln_csum', c_f),
        ('ln_eps', F),
        ('a_relu', I),
        ('bias', c_f),
        ('alpha', F),
        ('act', I),
        ('rowscale', c_f), ('sRSb', LL),
        ('gate', c_f), ('sGb', LL), ('sGm', LL), ('gate_sigmoid', I),
        ('resid', c_f), ('sRb', LL), ('sRm', LL),
        ('B_split', C.c_void_p), ('sB3p', LL), ('sB3n', LL), ('sB3k', LL), ('sB3b', LL),
        ('A_split', C.c_void_p), ('sA3p', LL), ('sA3m', LL), ('sA3k', LL), ('sA3b', LL),
        ('batch_inner', I), ('sA3i', LL), ('sB3i', LL),
        ('c_split_tile', I),
        ('c_split_nA', I),
        ('C_split', C.c_void_p), ('sCp', LL), ('sCk', LL), ('c_split_L', I),
        ('glu', I),
        ('a_pair_transpose', I),
        ('pair_L', I), ('pair_Lp', I), ('a_pair', I), ('c_pair', I),
        ('A2', c_f), ('sA2b', LL), ('sA2m', LL), ('K2', I),
        ('B2_split', C.c_void_p), ('sB23p', LL), ('sB23n', LL), ('sB23k', LL),
        ('ln2_csum', c_f), ('bias2', c_f),
        ('mlp', I), ('N2', I),
        ('out_ln_w', c_f), ('out_ln_b', c_f), ('out_ln_eps', F),
        ('exact', I),
        ('b_f16', I), ('b_exp', I), ('b2_exp', I),
        ('tune', I),
        ('c_planes_from', I), ('c_planes_group', I),
        ('range_flag', c_f), ('range_tag', I),
        ('clock_probe', c_f),
        ('a_vec_ok', I), ('b_vec_ok', I), ('fast_ok', I),
        ('c_vec_ok', I), ('g_vec_ok', I), ('r_vec_ok', I), ('rs_vec_ok', I),
    ]


class AbxIpaTail(C.Structure):
    _fields_ = [
        ('feat', c_f), ('s_feat', LL),
        ('s', c_f), ('s_s', LL),
        ('M', I), ('K1', I), ('C', I),
        ('W_final', C.c_void_p), ('e_final', I), ('b_final', c_f),
        ('ln1_w', c_f), ('ln1_b', c_f),
        ('W_t0', C.c_void_p), ('e_t0', I), ('b_t0', c_f),
        ('W_t2', C.c_void_p), ('e_t2', I), ('b_t2', c_f),
        ('W_t4', C.c_void_p), ('e_t4', I), ('b_t4', c_f),
        ('ln2_w', c_f), ('ln2_b', c_f),
        ('ln_eps', F),
        ('W_aff', c_f), ('b_aff', c_f),
        ('fixed', c_f), ('init_q', c_f), ('init_t', c_f),
        ('cur_q', c_f), ('cur_t', c_f), ('cur_R', c_f), ('delta_q', c_f), ('pscale', F),
        ('range_flag', c_f), ('range_tag', I),
        ('partial', c_f), ('n_partial', I), ('s_partial', LL),
    ]


class AbxHeadsTail(C.Structure):
    _fields_ = [
        ('s', c_f), ('s_s', LL), ('s0', c_f), ('s_s0', LL), ('M', I),
        ('W_act', C.c_void_p), ('e_act', I), ('b_act', c_f),
        ('W_init', C.c_void_p), ('e_init', I), ('b_init', c_f),
        ('W_r0', C.c_void_p), ('e_r0', I), ('b_r0', c_f),
        ('W_r1', C.c_void_p), ('e_r1', I), ('b_r1', c_f),
        ('W_r2', C.c_void_p), ('e_r2', I), ('b_r2', c_f),
        ('W_r3', C.c_void_p), ('e_r3', I), ('b_r3', c_f),
        ('W_proj', C.c_void_p), ('e_proj', I), ('b_proj', c_f),
        ('un', c_f),
        ('lns_w', c_f), ('lns_b', c_f),
        ('W_s1', C.c_void_p), ('e_s1', I), ('b_s1', c_f),
        ('W_s3', C.c_void_p), ('e_s3', I), ('b_s3', c_f),
        ('W_s5', C.c_void_p), ('e_s5', I), ('b_s5', c_f),
        ('logits', c_f),
        ('lnp_w', c_f), ('lnp_b', c_f),
        ('W_p1', C.c_void_p), ('e_p1', I), ('b_p1', c_f),
        ('W_p3', C.c_void_p), ('e_p3', I), ('b_p3', c_f),
        ('W_p5', C.c_void_p), ('e_p5', I), ('b_p5', c_f),
        ('pl', c_f),
        ('ln_eps', F),
        ('range_flag', c_f), ('range_tag', I),
    ]


class AbxTriAttn(C.Structure):
    _fields_ = [
        ('q', c_f), ('k', c_f), ('v', c_f), ('gate', c_f),
        ('sb', LL), ('ss', LL), ('sl', LL),
        ('bias', c_f), ('bias_sb', LL), ('bias_sh', LL), ('bias_sq', LL), ('bias_sk', LL),
        ('keymask', c_f), ('km_sb', LL),
        ('out', c_f), ('ob', LL), ('os', LL), ('ol', LL),
        ('B', I), ('S', I), ('L', I), ('H', I), ('D', I),
        ('scale', F),
        ('exact', I),
        ('clock_probe', c_f),
        ('range_flag', c_f), ('range_tag', I),
        ('tune', I),
        ('bias_log2', I),
        ('kv_planes', I),
        ('q_parts', I), ('row_groups', I),
    ]


class AbxLinearPack(C.Structure):
    _fields_ = [('Wt', c_f), ('csum', c_f), ('bias', c_f), ('planes', C.c_void_p), ('b_exp', I), ('K', I), ('N', I)]


class AbxLinearSrc(C.Structure):
    _fields_ = [('W', c_f), ('b', c_f), ('rows', I), ('glu', I)]


class AbxTriMulPack(C.Structure):
    _fields_ = [('glu', AbxLinearPack), ('out', AbxLinearPack), ('gate', AbxLinearPack)]


class AbxTriAttnPack(C.Structure):
    _fields_ = [('qkv', AbxLinearPack), ('gate', AbxLinearPack), ('pair', AbxLinearPack), ('out', AbxLinearPack)]


class AbxScoreArgs(C.Structure):
    _fields_ = [
        ('init_q', c_f), ('init_t', c_f), ('delta_q', c_f), ('cur_t', c_f), ('fixed_mask', c_f),
        ('t', c_f), ('t_is_f32', I),
        ('score_norms', c_f), ('num_sigma', I), ('num_omega', I), ('discrete_sigma', c_f), ('discrete_omega', c_f),
        ('exp_max_sigma', F), ('exp_min_sigma', F),
        ('min_b', F), ('bdiff', F),
        ('coord_scale', F),
        ('position_scale', F),
        ('rot_score', c_f), ('trans_score', c_f), ('rigids', c_f),
        ('B', I), ('L', I),
    ]


class AbxReverseArgs(C.Structure):
    _fields_ = [
        ('rigid_in', c_f), ('rigid_is_f64', I),
        ('seq_in', c_f),
        ('rot_score', c_f), ('trans_score', c_f), ('ts_is_f32', I), ('logits', c_f),
        ('diffuse_mask', c_f), ('t', c_f), ('dt', F),
        ('z_rot', c_f), ('z_trans', c_f), ('jumps', c_f),
        ('u_jumps', c_f),
        ('dt_dev', c_f),
        ('seed', C.c_ulonglong), ('sample_ids', c_f), ('step', I),
        ('step_dev', c_f),
        ('exp_max_sigma', F), ('exp_min_sigma', F), ('min_b', F), ('bdiff', F), ('coord_scale', F), ('rate_const', F),
        ('noise_scale', F), ('center', I),
        ('rigid_out', c_f), ('seq_out', c_f), ('rates_out', c_f), ('jumps_out', c_f),
        ('B', I), ('L', I),
    ]


class AbxGuidanceArgs(C.Structure):
    _fields_ = [
        ('atom14', c_f), ('atom_mask', c_f), ('aatype', c_f), ('chain_id', c_f),
        ('residx', c_f),
        ('radius', c_f), ('frame_trans', c_f),
        ('overlap_tolerance', F), ('between_chain_factor', F), ('bond_tolerance_factor', F), ('w_clash', F), ('w_bond', F), ('w_angle', F),
        ('energy', c_f), ('grad_atom', c_f), ('grad_trans', c_f), ('grad_rot', c_f),
        ('B', I), ('L', I),
    ]


_S = c_f   # hipStream_t

_PROTOS = {
    'abx_version': (I, []),
    'abx_last_error_string': (C.c_char_p, []),
    'abx_init': (I, [I]),
    'abx_gemm': (I, [C.POINTER(AbxGemm), _S]),
    'abx_gemm_side': (I, [C.POINTER(AbxGemm), C.POINTER(AbxGemm), _S]),
    'abx_gemm_check_modes': (I, [C.POINTER(AbxGemm)]),
    'abx_gemm_planes_ok': (I, [LL]),
    'abx_split_weights_f16': (I, [c_f, LL, LL, I, I, I, C.c_void_p, _S]),
    'abx_ipa_tail': (I, [C.POINTER(AbxIpaTail), _S]),
    'abx_heads_tail': (I, [C.POINTER(AbxHeadsTail), _S]),
    'abx_gemm3_occupancy': (I, [I]),
    'abx_row_stats': (I, [c_f, LL, LL, LL, I, I, I, F, c_f, _S]),
    'abx_layernorm': (I, [c_f, LL, LL, I, c_f, c_f, F, c_f, LL, c_f, LL, _S]),
    'abx_tri_attn_fwd': (I, [C.POINTER(AbxTriAttn), _S]),
    'abx_seq_attn_fwd': (I, [c_f, c_f, c_f, c_f, c_f, I, I, I, I, F, _S]),
    'abx_ipa_pack': (I, [c_f, c_f, c_f, c_f, c_f, c_f, I, I, F, _S]),
    'abx_ipa_attn': (I, [c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, I, I, _S]),
    'abx_debug_poison_lds': (I, [C.c_uint, _S]),
    'abx_ipa_weights': (I, [c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, I, I, _S]),
    'abx_ipa_pair': (I, [c_f, c_f, c_f, I, I, _S]),
    'abx_ipa_attn_workspace_bytes': (C.c_longlong, [I, I]),
    'abx_ipa_qpack_bytes': (C.c_longlong, [I, I]),
    'abx_timestep_embedding': (I, [c_f, c_f, I, I, c_f, _S]),
    'abx_assemble_seq': (I, [c_f, LL, c_f, c_f, I, c_f, c_f, c_f, c_f, c_f, I, I, I, I, _S]),
    'abx_assemble_pair': (I, [c_f, LL, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, I, I, I, I, _S]),
    'abx_opm_features': (I, [c_f, c_f, LL, c_f, I, I, I, _S]),
    'abx_opm_out_fwd': (I, [c_f, LL, c_f, c_f, c_f, I, I, c_f, I, _S]),
    'abx_assemble_pair_bias': (I, [c_f, LL, c_f, c_f, c_f, c_f, c_f, c_f, c_f, C.c_void_p, I, c_f, c_f, F, c_f, I, I, c_f, I, _S]),
    'abx_pair_mask': (I, [c_f, c_f, I, I, I, _S]),
    'abx_transpose_last2': (I, [c_f, c_f, I, I, I, I, _S]),
    'abx_pair_embed_features': (I, [c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, I, I, _S]),
    'abx_relpos_block': (I, [c_f, c_f, c_f, I, I, I, I, I, _S]),
    'abx_gather_rows': (I, [c_f, c_f, c_f, c_f, LL, LL, I, _S]),
    'abx_frames_init': (I, [c_f, I, c_f, c_f, c_f, c_f, c_f, c_f, I, F, _S]),
    'abx_rigid_update': (I, [c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, I, F, _S]),
    'abx_scores': (I, [C.POINTER(AbxScoreArgs), _S]),
    'abx_torsion_finalize': (I, [c_f, c_f, c_f, c_f, I, _S]),
    'abx_seq_head_atoms': (I, [c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, I, _S]),
    'abx_prev_pos': (I, [c_f, c_f, I, c_f, I, I, _S]),
    'abx_plddt': (I, [c_f, c_f, I, I, _S]),
    'abx_igso3_tables': (I, [c_f, c_f, I, I, I, c_f, c_f, c_f, _S]),
    'abx_reverse_step': (I, [C.POINTER(AbxReverseArgs), _S]),
    'abx_clash_grad_workspace_bytes': (LL, [I, I]),
    'abx_clash_grad': (I, [C.POINTER(AbxGuidanceArgs), c_f, _S]),
    'abx_pack_linear_bytes': (LL, [I, I]),
    'abx_pack_linear': (I, [C.POINTER(AbxLinearSrc), I, I, c_f, c_f, I, c_f, C.POINTER(AbxLinearPack), _S]),
    'abx_transition_workspace_bytes': (LL, [LL, I, I]),
    'abx_transition_fwd': (I, [C.POINTER(AbxLinearPack), C.POINTER(AbxLinearPack), c_f, LL, I, c_f, c_f, I, _S]),
    'abx_tri_mul_workspace_bytes': (LL, [I, I]),
    'abx_tri_mul_workspace_init': (I, [c_f, I, I, _S]),
    'abx_tri_mul_fwd': (I, [C.POINTER(AbxTriMulPack), c_f, c_f, c_f, I, I, I, c_f, c_f, I, _S]),
    'abx_tri_attn_block_workspace_bytes': (LL, [I, I]),
    'abx_tri_attn_block_fwd': (I, [C.POINTER(AbxTriAttnPack), c_f, c_f, I, I, I, I, c_f, c_f, I, _S]),
}

EXPORTED = tuple(_PROTOS)
_lib = None


class AbxHipError(RuntimeError):
    pass


def library_path():
    """Path of the shared object that load() binds (ABX_HIP_LIB or the in-tree build)."""
    return LIB_PATH


def load():
    """Load the library (raises if it has not been built: run `python -c 'import __graft_entry__ as g; g.build()'`)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise AbxHipError(f'{LIB_PATH} not found: build it with `make -C abx_amd/csrc` (no CPU fallback exists)')
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _PROTOS.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if lib.abx_version() != 1:
        raise AbxHipError('libabx_hip ABI version mismatch')
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().abx_last_error_string()
        raise AbxHipError(f'{what} failed (rc={rc}): {msg.decode() if msg else ""}')
