"""ctypes binding of the C ABI declared in ``include/lemo_hip.h`` (``liblemo_hip.so``).

There is NO CPU fallback: :func:`get_lib` raises if the gfx950 library has not been built
(``python -c 'import __graft_entry__ as g; g.build()'`` or ``make -C lemo_amd/csrc``).
``HipLib(path)`` can also open the host-emulated build of the *same* sources
(``liblemo_emu.so``, tests/hipemu) -- that is done by the CPU test-suite only, explicitly, to check
kernel index arithmetic without a GPU; product code never does.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

_CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'csrc')
LIB_PATH = os.environ.get('LEMO_HIP_LIB') or os.path.join(_CSRC, 'liblemo_hip.so')   # override: A/B builds (tools/ab_build.sh)
EMU_LIB_PATH = os.path.join(_CSRC, 'liblemo_emu.so')

fp = C.POINTER(C.c_float)
ip = C.POINTER(C.c_int)
vp = C.c_void_p


class VPoserW(C.Structure):
    _fields_ = [(n, vp) for n in ('w1', 'w1t', 'b1', 'w2', 'w2t', 'b2', 'w3', 'w3t', 'b3')]


class BodyConst(C.Structure):
    _fields_ = [('nj', C.c_int), ('nshape', C.c_int), ('ncomp', C.c_int), ('nlev', C.c_int)] + \
        [(n, vp) for n in ('parents', 'level_start', 'level_joints', 'child_start', 'child_list',
                           'J_template', 'J_dirs', 'pose_mean', 'lh_comp', 'rh_comp')]


class PoseIn(C.Structure):
    _fields_ = [(n, vp) for n in ('global_orient', 'body_pose', 'jaw', 'leye', 'reye', 'lh', 'rh')] + \
        [('hand_stride', C.c_int), ('betas', vp), ('betas_stride', C.c_int), ('expr', vp),
         ('rot6d', vp), ('vposer_o', vp), ('go_out', vp), ('zero_f64', vp), ('n_zero', C.c_int),
         ('step_ctr', vp), ('step_cur', vp), ('nonfinite', vp)]


class PoseWs(C.Structure):
    _fields_ = [(n, vp) for n in ('full_pose', 'R', 'J', 'T', 'A', 'Jtr', 'Xg')] + [('Bp', C.c_int), ('XgS', vp), ('xgs_f16', C.c_int)]


class PoseGradIn(C.Structure):
    _fields_ = [(n, vp) for n in ('dA', 'dJtr', 'dX', 'd_full_pose')]


class PoseGradOut(C.Structure):
    _fields_ = [(n, vp) for n in ('d_global_orient', 'd_body_pose', 'd_jaw', 'd_leye', 'd_reye', 'd_lh', 'd_rh')] + \
        [('hand_stride', C.c_int), ('d_betas', vp), ('d_expr', vp),
         ('rot6d', vp), ('d_rot6d', vp), ('vposer_o', vp), ('d_vposer_o', vp)]


WGRAD_MAX_JOBS = 24


class WgradJob(C.Structure):
    _fields_ = [('partial', vp), ('dy', vp), ('dw', vp), ('db', vp), ('nslab', C.c_int), ('cin', C.c_int), ('cout', C.c_int),
                ('cin_real', C.c_int), ('cout_real', C.c_int), ('H', C.c_int), ('W', C.c_int)]


class AeDesc(C.Structure):
    """lemo_ae_desc"""
    _fields_ = [('H', C.c_int), ('W', C.c_int), ('lr', C.c_float), ('ws', vp), ('ws_floats', C.c_longlong), ('clips', C.c_int)]


class SkinConst(C.Structure):
    _fields_ = [('V', C.c_int), ('NC', C.c_int), ('KW', C.c_int), ('blend_fp32', C.c_int)] + \
        [(n, vp) for n in ('Dg', 'v_template', 'w_idx', 'w_val', 'DgH')] + [('dgh_inv', C.c_float)]


class VertexSetBwd(C.Structure):
    _fields_ = [('n', C.c_int), ('NCs', C.c_int)] + \
        [(n, vp) for n in ('ids', 'vp_row', 'Dk', 'DkT', 'jcsr_start', 'jcsr_u', 'jcsr_w', 'jcsr_chunk', 'jc_u', 'jc_w', 'part')] + \
        [('part_frames', C.c_int), ('gemm_slabs', C.c_int), ('gemm_part', vp), ('DkG', vp)]


class FitConst(C.Structure):
    _fields_ = [('n', C.c_int), ('n67', C.c_int), ('n81', C.c_int)] + \
        [(n, vp) for n in ('row67', 'row81', 'foot_start', 'foot_row', 'u_row', 'u_m67', 'u_m81',
                           'u_foot_mask', 'Xstd', 'Xmean', 'cam2world')]


CHAIN_MAX = 8


class FitDesc(C.Structure):
    _fields_ = [
        ('B', C.c_int), ('Bp', C.c_int), ('V', C.c_int), ('nrows', C.c_int), ('full_vertices', C.c_int),
        ('conv_variant', C.c_int),
        ('vposer', VPoserW), ('body', BodyConst), ('skin', SkinConst), ('uset', VertexSetBwd), ('fit', FitConst),
        ('fwd_ids', vp),
        ('enc_ch', C.c_int * 11), ('enc_w', vp * 10), ('enc_b', vp * 10), ('enc_wbwd', vp * 10),
        ('enc_w2', vp * 10), ('enc_wbwd2', vp * 10), ('enc_w3', vp * 10), ('enc_wbwd3', vp * 10), ('enc_w3_inv', C.c_float * 10), ('enc_wbwd3_inv', C.c_float * 10),
        ('target', vp), ('contact', vp), ('weights', vp), ('weights_host', C.c_float * 6),
        ('transl', vp), ('rot6d', vp), ('other', vp), ('shape', vp),
        ('adam_m', vp * 3), ('adam_v', vp * 3), ('step_ctr', vp),
        ('lr0', C.c_float), ('lr1', C.c_float), ('lr_switch', C.c_int),
        ('go_aa', vp), ('body_aa', vp), ('h1', vp), ('h2', vp), ('vo', vp), ('vp_scratch', vp),
        ('pose', PoseWs),
        ('verts', vp), ('v_posed', vp), ('x0', vp), ('canon', vp),
        ('act', vp * 11), ('dact', vp * 2),
        ('dx0', vp), ('spartial', vp), ('vpartial', vp), ('losses', vp), ('dverts', vp), ('dvp', vp),
        ('dA', vp), ('dX', vp), ('loss_acc', vp), ('step_cur', vp),
        ('g_transl', vp), ('g_rot6d', vp), ('g_other', vp), ('g_go', vp), ('g_body', vp),
        ('snap', vp), ('nonfinite', vp), ('per_frame', C.c_int), ('lr2', C.c_float), ('lr_switch2', C.c_int),
        ('verts_side', vp), ('transl_side', vp),
    ]


class ProxConst(C.Structure):
    """lemo_prox_const"""
    _fields_ = [('n_op', C.c_int), ('n_sj', C.c_int), ('joint_map', vp), ('jm_start', vp), ('jm_list', vp),
                ('n_extra', C.c_int), ('n_lmk', C.c_int), ('extra_rows', vp), ('lmk_rows', vp), ('lmk_bary', vp),
                ('n_s', C.c_int), ('s_vid', vp), ('s_m67', vp), ('s_m81', vp), ('s_foot_mask', vp), ('s_fric', vp),
                ('s_jstart', vp), ('s_jidx', vp), ('s_jw', vp), ('n_fric', C.c_int), ('fric_vid', vp), ('n67', C.c_int),
                ('m67_vid', vp), ('foot_start', vp), ('foot_vid', vp)]


PROX_NW = 13


class ProxDesc(C.Structure):
    """lemo_prox_desc"""
    _fields_ = [
        ('B', C.c_int), ('Bp', C.c_int), ('V', C.c_int), ('conv_variant', C.c_int), ('first_batch_flag', C.c_int),
        ('use_infill', C.c_int), ('T', C.c_int),
        ('vposer', VPoserW), ('body', BodyConst), ('skin', SkinConst), ('uset', VertexSetBwd), ('fit', FitConst), ('pc', ProxConst),
        ('enc_ch', C.c_int * 11), ('enc_w', vp * 10), ('enc_b', vp * 10), ('enc_wbwd', vp * 10), ('enc_w2', vp * 10),
        ('enc_wbwd2', vp * 10), ('enc_w3', vp * 10), ('enc_wbwd3', vp * 10), ('enc_w3_inv', C.c_float * 10), ('enc_wbwd3_inv', C.c_float * 10),
        ('sdf', vp), ('sdf_dim', C.c_int * 3), ('grid_min', C.c_float * 3), ('grid_max', C.c_float * 3),
        ('cam2world', C.c_float * 12), ('cam', C.c_float * 4),
        ('gt_joints', vp), ('w2', vp), ('marker_mask', vp), ('body_markers_rec', vp), ('contact_lbl_rec', vp),
        ('weights', vp), ('weights_host', C.c_float * PROX_NW),
        ('global_orient', vp), ('transl', vp), ('left_hand_pose', vp), ('right_hand_pose', vp), ('jaw_pose', vp),
        ('leye_pose', vp), ('reye_pose', vp), ('expression', vp), ('pose_embedding', vp), ('betas', vp),
        ('adam_m', vp), ('adam_v', vp), ('step_ctr', vp), ('step_cur', vp), ('nonfinite', vp), ('lr', C.c_float),
        ('h1', vp), ('h2', vp), ('vo', vp), ('vp_scratch', vp), ('pose', PoseWs),
        ('verts', vp), ('v_posed', vp), ('dverts', vp), ('x0', vp), ('canon', vp), ('dx0', vp),
        ('act', vp * 11), ('dact', vp * 2),
        ('dJtr', vp), ('dJv', vp), ('dtr_j', vp), ('gp', vp), ('dfp_add', vp),
        ('dvp', vp), ('dA', vp), ('dtr_v', vp), ('dX', vp),
        ('g_go', vp), ('g_lh', vp), ('g_rh', vp), ('g_jaw', vp), ('g_leye', vp), ('g_reye', vp), ('g_expr', vp), ('g_pe', vp),
        ('loss_acc', vp), ('losses', vp),
    ]


class FitState(C.Structure):
    """lemo_fit_state"""
    _fields_ = [('transl', vp), ('rot6d', vp), ('other', vp), ('adam_m', vp * 3), ('adam_v', vp * 3), ('step', vp)]


class ProxState(C.Structure):
    """lemo_prox_state"""
    _fields_ = [(n, vp) for n in ('global_orient', 'transl', 'left_hand_pose', 'right_hand_pose', 'jaw_pose', 'leye_pose', 'reye_pose',
                                  'expression', 'pose_embedding', 'adam_m', 'adam_v', 'step')]


def ptr(t: Optional[torch.Tensor]):
    """raw device pointer of a contiguous tensor (None -> NULL)."""
    if t is None:
        return None
    assert t.is_contiguous(), 'C ABI takes contiguous buffers'
    return t.data_ptr()


class LemoHipError(RuntimeError):
    pass


_SIGS = {
    'lemo_abi_version': (C.c_int, []),
    'lemo_build_flags': (C.c_int, []),
    'lemo_conv3x3_mfma': (C.c_int, [vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    'lemo_conv3x3_mfma_splitk': (C.c_int, [vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    'lemo_conv3x3_mfma_lds': (C.c_int, [vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    'lemo_lbs_verts_fwd_active': (C.c_int, [C.POINTER(SkinConst), C.POINTER(VertexSetBwd), vp, C.c_int, vp, C.c_int, vp, C.c_int, vp, vp, vp, vp]),
    'lemo_capture_begin': (C.c_int, [vp]),
    'lemo_capture_end': (C.c_int, [vp, C.POINTER(C.c_void_p)]),
    'lemo_graph_launch': (C.c_int, [vp, vp]),
    'lemo_graph_destroy': (C.c_int, [vp]),
    'lemo_reconstruct_global_body': (C.c_int, [vp, C.c_int, C.c_int, C.c_double, vp, vp]),
    'lemo_reconstruct_global_body_dev': (C.c_int, [vp, C.c_int, C.c_int, vp, vp, vp]),
    'lemo_local_markers_4chan': (C.c_int, [vp, vp, C.c_int, C.c_int, vp, vp, vp]),
    'lemo_decode_clip': (C.c_int, [vp, vp, vp, vp, vp, C.c_int, C.c_int, vp, vp, vp]),
    'lemo_conv3x3_split_supported': (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int]),
    'lemo_conv3x3_mfma_split': (C.c_int, [vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    'lemo_conv3x3_mfma_split_census': (C.c_int, [vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp]),
    'lemo_conv3x3_mfma_split_f16': (C.c_int, [vp, vp, C.c_float, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    'lemo_conv3x3_pair_supported': (C.c_int, [C.c_int] * 5),
    'lemo_conv3x3_wino_supported': (C.c_int, [C.c_int] * 4),
    'lemo_conv3x3_wino_f16': (C.c_int, [vp, vp, C.c_float, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, vp, vp]),
    'lemo_conv3x3_pair_f16': (C.c_int, [vp, vp, C.c_float, vp, vp, vp, vp, C.c_float, vp, vp, vp, C.c_int, C.c_int, C.c_int, vp, vp]),
    'lemo_enc_head': (C.c_int, [vp, vp, C.c_int, vp, C.c_int, vp, C.c_int, vp, vp, vp, C.c_float, vp, vp, vp, vp, vp, vp]),
    'lemo_enc_tail': (C.c_int, [vp, vp, C.c_float, vp, vp, vp, C.c_int, C.c_int, vp]),
    'lemo_enc_tail3': (C.c_int, [vp, vp, C.c_float, vp, vp, C.c_float, vp, vp, vp, C.c_int, C.c_int, vp]),
    'lemo_conv3x3_mfma_split_census2': (C.c_int, [vp, vp, C.c_float, C.c_int, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp]),
    'lemo_conv3x3_mfma_lds_census': (C.c_int, [vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp]),
    'lemo_conv3x3_c1': (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, vp]),
    'lemo_conv3x3_c1_bwd': (C.c_int, [vp, vp, vp, C.c_int, C.c_int, C.c_int, vp]),
    'lemo_smooth_loss_blocks': (C.c_int, [C.c_int, C.c_int, C.c_int]),
    'lemo_smooth_loss': (C.c_int, [vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_float, vp]),
    'lemo_vposer_decode_fwd': (C.c_int, [C.POINTER(VPoserW), vp, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp]),
    'lemo_vposer_decode_bwd': (C.c_int, [C.POINTER(VPoserW), vp, vp, vp, vp, vp, C.c_int, vp, C.c_int, vp, vp]),
    'lemo_vposer_mlp_bwd': (C.c_int, [C.POINTER(VPoserW), vp, vp, C.c_int, vp, C.c_int, vp, vp]),
    'lemo_gemm_nt16': (C.c_int, [vp, C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int, vp, vp, C.c_int, C.c_int, vp]),
    'lemo_gemm_nt16_splitk_part_floats': (C.c_int, [C.c_int, C.c_int]),
    'lemo_gemm_nt16_splitk': (C.c_int, [vp, C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int, vp, C.c_int, vp, vp]),
    'lemo_rot6d_to_aa_fwd': (C.c_int, [vp, C.c_int, C.c_int, vp, vp]),
    'lemo_rot6d_to_aa_bwd': (C.c_int, [vp, C.c_int, vp, C.c_int, vp, vp]),
    'lemo_smplx_pose_fwd': (C.c_int, [C.POINTER(BodyConst), C.POINTER(PoseIn), C.POINTER(PoseWs), C.c_int, vp]),
    'lemo_smplx_pose_bwd': (C.c_int, [C.POINTER(BodyConst), C.POINTER(PoseWs), C.POINTER(PoseGradIn),
                                      C.POINTER(PoseGradOut), C.c_int, vp]),
    'lemo_lbs_verts_fwd': (C.c_int, [C.POINTER(SkinConst), vp, C.c_int, vp, C.c_int, vp, vp, C.c_int, C.c_int, vp, vp, vp]),
    'lemo_lbs_verts_fwd_xs': (C.c_int, [C.POINTER(SkinConst), vp, vp, C.c_int, vp, C.c_int, vp, vp, C.c_int, C.c_int, vp, vp, vp]),
    'lemo_lbs_verts_fwd_census': (C.c_int, [C.POINTER(SkinConst), vp, C.c_int, vp, C.c_int, vp, C.c_int, C.c_int, vp, vp, vp, vp, vp]),
    'lemo_lbs_verts_bwd': (C.c_int, [C.POINTER(SkinConst), C.POINTER(VertexSetBwd), vp, C.c_int, vp, C.c_int, vp,
                                     C.c_int, C.c_int, vp, vp, vp, vp, vp]),
    'lemo_joints_assemble': (C.c_int, [vp, C.c_int, vp, C.c_int, vp, C.c_int, vp, vp, C.c_int, vp, C.c_int, vp, vp]),
    'lemo_maxpool3s2_fwd': (C.c_int, [vp, C.c_int, C.c_int, vp, vp, C.c_int, vp]),
    'lemo_maxpool3s2_bwd': (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, vp]),
    'lemo_stuff2_fwd': (C.c_int, [vp, C.c_int, C.c_int, vp, C.c_int, C.c_int, C.c_int, vp]),
    'lemo_stuff2_bwd': (C.c_int, [vp, C.c_int, C.c_int, vp, vp, C.c_int, C.c_int, C.c_int, vp]),
    'lemo_conv3x3_wgrad_nslab': (C.c_int, [C.c_int, C.c_int]),
    'lemo_conv3x3_wgrad': (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp]),
    'lemo_conv3x3_wgrad_partial': (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp]),
    'lemo_conv3x3_wgrad_reduce_multi': (C.c_int, [C.POINTER(WgradJob), C.c_int, vp]),
    'lemo_adam_flat': (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_float, C.c_int, vp]),
    'lemo_adam_flat_ctr': (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_float, vp, vp]),
    'lemo_ae_ws_floats': (C.c_longlong, [C.c_int, C.c_int]),
    'lemo_ae_n_param': (C.c_int, []),
    'lemo_ae_create': (vp, [C.POINTER(AeDesc)]),
    'lemo_ae_destroy': (None, [vp]),
    'lemo_ae_load': (C.c_int, [vp, vp, vp, vp, vp]),
    'lemo_ae_step': (C.c_int, [vp, C.c_int, C.c_int, vp]),
    'lemo_ae_forward': (C.c_int, [vp, vp, vp, vp]),
    'lemo_ae_params': (C.c_int, [vp, vp, vp]),
    'lemo_ae_load_clip': (C.c_int, [vp, C.c_int, vp, vp, vp, vp]),
    'lemo_ae_forward_clip': (C.c_int, [vp, C.c_int, vp, vp, vp]),
    'lemo_ae_params_clip': (C.c_int, [vp, C.c_int, vp, vp]),
    'lemo_ae_wgrad_probe': (C.c_int, [vp, C.c_int, vp]),
    'lemo_ae_conv': (C.c_int, [vp, vp, vp, vp, vp] + [C.c_int] * 12 + [vp]),
    'lemo_ae_conv_f16': (C.c_int, [vp, vp, vp, vp, vp] + [C.c_int] * 12 + [vp, C.c_float, vp, vp, vp]),
    'lemo_sdf_sample': (C.c_int, [vp, C.c_int, C.c_int, C.c_int, vp, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float), vp, vp, vp]),
    'lemo_fit_create': (vp, [C.POINTER(FitDesc)]),
    'lemo_fit_destroy': (None, [vp]),
    'lemo_fit_forward': (C.c_int, [vp, vp]),
    'lemo_fit_backward': (C.c_int, [vp, vp]),
    'lemo_fit_step': (C.c_int, [vp, C.c_int, C.c_int, vp]),
    'lemo_fit_prepare': (C.c_int, [vp, C.c_int, vp]),
    'lemo_fit_census': (C.c_int, [vp, C.c_int, C.POINTER(C.c_float), vp]),
    'lemo_fit_load_state': (C.c_int, [vp, C.POINTER(FitState), vp]),
    'lemo_fit_save_state': (C.c_int, [vp, C.POINTER(FitState), vp]),
    'lemo_prox_load_state': (C.c_int, [vp, C.POINTER(ProxState), vp]),
    'lemo_prox_save_state': (C.c_int, [vp, C.POINTER(ProxState), vp]),
    'lemo_prox_create': (vp, [C.POINTER(ProxDesc)]),
    'lemo_prox_destroy': (None, [vp]),
    'lemo_prox_closure': (C.c_int, [vp, vp]),
    'lemo_prox_step': (C.c_int, [vp, C.c_int, C.c_int, vp]),
}
EXPORTED_SYMBOLS = tuple(_SIGS)


class HipLib:
    """typed handle on liblemo_hip.so (or, tests only, liblemo_emu.so)."""

    def __init__(self, path: str, is_emu: bool = False):
        if not os.path.exists(path):
            raise LemoHipError(
                f'{path} not found: the HIP extension is not built. Build it with '
                f'`make -C {_CSRC}` (hipcc --offload-arch=gfx950); there is no CPU fallback.')
        self.path, self.is_emu = path, is_emu
        self._dll = C.CDLL(path)
        for name, (res, args) in _SIGS.items():
            fn = getattr(self._dll, name)
            fn.restype, fn.argtypes = res, args
            setattr(self, name[len('lemo_'):], fn)
        if not self.build_flags() & 1:
            # auto-vectorised packed fp32 gave a wrong rotation entry in ~1 of 100 runs when another kernel shared the
            # CU (root cause open; the flags remove every v_pk_*_f32).  A library built some other way is not the product.
            raise LemoHipError(f'{path} was built without -fno-slp-vectorize -fno-vectorize (-DLEMO_NO_PACKED_FP32): rebuild it with '
                               f'`make -C {_CSRC}`')

    def check(self, rc: int, what: str = ''):
        if rc != 0:
            raise LemoHipError(f'liblemo_hip: {what} failed with code {rc}')

    def stream(self, device) -> Optional[int]:
        """current torch stream handle on ``device`` (NULL for the host-emulated library)."""
        if self.is_emu:
            return None
        return torch.cuda.current_stream(device).cuda_stream


_DEFERRED: list = []


def quiesce(device, lib=None) -> bool:
    """wait for everything queued on `device` (destructors of objects whose buffers are in use on non-default streams call
    this before they release them).  Returns False -- and does nothing -- while the current stream is being captured: a device
    synchronisation would invalidate the capture (the garbage collector can run a destructor in the middle of somebody
    else's capture).  Never raises: it also runs at interpreter shutdown."""
    try:
        dev = torch.device(device)
        if dev.type == 'cuda' and not (lib is not None and lib.is_emu) and torch.cuda.is_available():
            if torch.cuda.is_current_stream_capturing():
                return False
            torch.cuda.synchronize(dev)
    except Exception:
        pass
    return True


def flush_deferred() -> None:
    """run the destructors that were parked during a stream capture (called from `release` and on every engine launch, so
    parked handles do not wait for some later object to die)"""
    if not _DEFERRED:
        return
    try:
        if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
            return
    except Exception:
        return
    pending, _DEFERRED[:] = list(_DEFERRED), []
    for fn in pending:
        try:
            fn()
        except Exception:
            pass


def release(device, lib, destroy, event=None) -> None:
    """run `destroy()` (a native handle's destructor) once the engine's work is done; during a stream capture it is parked
    and run later (`flush_deferred`).  `event`: the engine's last-launch event (`StreamOrdered._run_ev`) -- only THAT is
    waited for, so dropping one engine does not stall the device under its peers; without it (an engine that
    does not track its launches) the whole device is synchronised."""
    def wait():
        if event is None:
            return quiesce(device, lib)
        try:
            if torch.cuda.is_current_stream_capturing():
                return False
            event.synchronize()
        except Exception:
            pass
        return True

    def run():
        if event is not None:          # parked during a capture: the event may not have completed yet when it is flushed
            try:
                event.synchronize()
            except Exception:
                pass
        destroy()

    if wait():
        flush_deferred()
        try:
            destroy()
        except Exception:
            pass
    else:
        _DEFERRED.append(run)


_LIB: Optional[HipLib] = None


def get_lib() -> HipLib:
    """The product library.  Raises LemoHipError when it is missing (never falls back)."""
    global _LIB
    if _LIB is None:
        _LIB = HipLib(LIB_PATH, is_emu=False)
    return _LIB


def check_device(lib: HipLib, t: torch.Tensor):
    if lib.is_emu:
        if t.is_cuda:
            raise LemoHipError('the host-emulated test library only takes CPU tensors')
    elif not t.is_cuda:
        raise LemoHipError('lemo_amd compute entry points need tensors on a HIP device (no CPU fallback)')


class StreamOrdered:
    """Ordering between an engine's host-side state writes (torch ops on whatever stream is current) and its launches
    (possibly on another, non-blocking stream), owned by the engine: each side records an event that the other side's
    stream waits for.  Without it a clip's first Adam update could overtake the zeroing of its own moments."""
    _gpu = False
    _setup_ev = _run_ev = None

    def _init_order(self, device, lib):
        # LEMO_UNORDERED=1 (diagnostics only, tools/concurrent_clips.py): run without the events, to reproduce what the
        # ordering fixes
        self._gpu = torch.device(device).type == 'cuda' and not lib.is_emu and os.environ.get('LEMO_UNORDERED', '0') != '1'
        self._odev = torch.device(device)
        self._setup_ev = self._run_ev = None

    def _cur(self):
        return torch.cuda.current_stream(self._odev)

    def _before_write(self):
        if self._gpu and self._run_ev is not None:
            self._cur().wait_event(self._run_ev)

    def _after_write(self):
        if self._gpu:
            if self._setup_ev is None:
                self._setup_ev = torch.cuda.Event()
            self._setup_ev.record(self._cur())

    def _before_run(self):
        if self._gpu:
            flush_deferred()
            for ev in (self._setup_ev, self._run_ev):
                if ev is not None:
                    self._cur().wait_event(ev)

    def _after_run(self):
        if self._gpu:
            if self._run_ev is None:
                self._run_ev = torch.cuda.Event()
            self._run_ev.record(self._cur())

    _before_read = _before_write
