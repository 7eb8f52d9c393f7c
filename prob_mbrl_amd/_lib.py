"""ctypes binding of libpmbrl_hip.so (C ABI declared in include/pmbrl.h).

There is NO fallback: importing this module without the built library, or
calling into it without a GPU, raises.  Build with `python __graft_entry__.py`
(or `make -C prob_mbrl_amd/csrc`).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('PMBRL_LIB_PATH', os.path.join(_HERE, 'csrc', 'libpmbrl_hip.so'))

MAX_LAYERS = 8
MAX_ANGLE = 8
MAX_TIP = 8
MAX_DIM = 64
FLAG_MM_STATES, FLAG_MM_REWARDS, FLAG_INFER_NS, FLAG_ZMM_PER_STEP = 1, 2, 4, 8
FLAG_FORCE_GENERIC = 16
FLAG_NO_SHAPED = 32
FLAG_POL_MASKS_PER_STEP, FLAG_DYN_MASKS_PER_STEP = 64, 128
FLAG_GMM_EXACT_NOISE_GRAD = 256
MAX_COMP = 8
REWARD_EXP, REWARD_NEG = 0, 1
PREC_F32, PREC_SPLIT, PREC_SPLIT_F16 = 0, 1, 2
INFO_COUNT = 21
TIMER_COUNT = 8
TIMER_NAMES = ['pack', 'fwd', 'bwd', 'dw', 'dw_reduce', 'reward']


class MLP(C.Structure):
    _fields_ = [('n_layers', C.c_int32),
                ('dims', C.c_int32 * (MAX_LAYERS + 1)),
                ('keep', C.c_float * MAX_LAYERS)]


class MlpCall(C.Structure):
    _fields_ = [('B', C.c_int32), ('net', MLP), ('max_log_std', C.c_float)]


class BnnConfig(C.Structure):
    _fields_ = [('M', C.c_int32), ('N', C.c_int32), ('net', MLP), ('max_log_std', C.c_float),
                ('temperature', C.c_float * MAX_LAYERS), ('reg_scale', C.c_float * MAX_LAYERS),
                ('drop_reg', C.c_float * MAX_LAYERS), ('reg_weight', C.c_float), ('loss_kind', C.c_int32),
                ('n_components', C.c_int32)]


class Reward(C.Structure):
    _fields_ = [('kind', C.c_int32), ('expand', C.c_int32),
                ('n_angle', C.c_int32), ('angle_dims', C.c_int32 * MAX_ANGLE),
                ('k', C.c_int32), ('C', C.c_float * (MAX_TIP * MAX_DIM)),
                ('tip_target', C.c_float * MAX_TIP), ('norm', C.c_float),
                ('w', C.c_float), ('Q', C.c_float * (MAX_TIP * MAX_TIP)),
                ('R', C.c_float * (MAX_DIM * MAX_DIM))]


class Config(C.Structure):
    _fields_ = [('B', C.c_int32), ('D', C.c_int32), ('U', C.c_int32),
                ('H', C.c_int32), ('B_global', C.c_int32),
                ('row_offset', C.c_int32), ('flags', C.c_int32),
                ('mm_groups', C.c_int32), ('max_log_std_pol', C.c_float),
                ('max_log_std_dyn', C.c_float), ('pol', MLP), ('dyn', MLP),
                ('reward', Reward), ('rows_per_wg_hint', C.c_int32), ('precision', C.c_int32),
                ('n_pol_angle', C.c_int32), ('pol_angle_dims', C.c_int32 * MAX_ANGLE),
                ('n_dyn_angle', C.c_int32), ('dyn_angle_dims', C.c_int32 * MAX_ANGLE),
                ('dyn_components', C.c_int32),
                ('mm_span_rows', C.c_int32), ('mm_span_offset', C.c_int32),
                ('mm_span_ranks', C.c_int32), ('mm_span_rank', C.c_int32)]


# int fn(void* ctx, void* stream, double* buf_d, int64_t n): pmbrl_plan_set_collective
COLLECTIVE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64)


class Inputs(C.Structure):
    _fields_ = [('x0', C.c_void_p), ('pol_params', C.c_void_p),
                ('dyn_params', C.c_void_p), ('mx', C.c_void_p),
                ('iSx', C.c_void_p), ('my', C.c_void_p), ('Sy', C.c_void_p),
                ('pol_scale', C.c_void_p), ('pol_bias', C.c_void_p),
                ('pol_mask_bits', C.c_void_p * MAX_LAYERS),
                ('dyn_mask_bits', C.c_void_p * MAX_LAYERS),
                ('z_pol', C.c_void_p), ('z_dyn', C.c_void_p),
                ('z_pol_step_stride', C.c_int64), ('z_dyn_step_stride', C.c_int64),
                ('z_mm', C.c_void_p), ('z_rr', C.c_void_p),
                ('z_pi', C.c_void_p), ('u_cat', C.c_void_p)]


EXPORTS = [
    'pmbrl_last_error', 'pmbrl_version', 'pmbrl_build_id', 'pmbrl_plan_create',
    'pmbrl_graph_capture_begin', 'pmbrl_graph_capture_end', 'pmbrl_graph_launch', 'pmbrl_graph_num_nodes', 'pmbrl_graph_destroy',
    'pmbrl_plan_destroy', 'pmbrl_plan_workspace_bytes', 'pmbrl_plan_info',
    'pmbrl_pack_mask', 'pmbrl_draw_masks', 'pmbrl_rollout_fwd', 'pmbrl_rollout_bwd', 'pmbrl_rollout_bwd_adam',
    'pmbrl_plan_set_loss',
    'pmbrl_weighted_sum', 'pmbrl_weighted_sum_steps', 'pmbrl_clip_adam', 'pmbrl_clip_adam_guarded', 'pmbrl_debug_linear',
    'pmbrl_plan_set_timing', 'pmbrl_plan_read_timing', 'pmbrl_plan_set_prof', 'pmbrl_plan_set_replay', 'pmbrl_plan_replay_count',
    'pmbrl_mlp_workspace_bytes', 'pmbrl_mlp_forward', 'pmbrl_mlp_grad_input',
    'pmbrl_bnn_plan_create', 'pmbrl_bnn_plan_destroy', 'pmbrl_bnn_plan_workspace_bytes',
    'pmbrl_bnn_plan_n_params', 'pmbrl_bnn_loss_grad', 'pmbrl_bnn_loss_grad_ex', 'pmbrl_bnn_train_steps',
    'pmbrl_comm_unique_id', 'pmbrl_comm_init', 'pmbrl_allreduce_sum', 'pmbrl_comm_count', 'pmbrl_comm_destroy',
    'pmbrl_plan_set_comm', 'pmbrl_plan_set_collective',
    'pmbrl_p2p_create', 'pmbrl_p2p_handle', 'pmbrl_p2p_open', 'pmbrl_p2p_allreduce_f32', 'pmbrl_p2p_allreduce_f64',
    'pmbrl_plan_set_p2p', 'pmbrl_p2p_error', 'pmbrl_p2p_destroy',
]

class Adam(C.Structure):
    """pmbrl_adam (include/pmbrl.h): the optimiser state pmbrl_rollout_bwd_adam updates."""
    _fields_ = [('params_d', C.c_void_p), ('exp_avg_d', C.c_void_p), ('exp_avg_sq_d', C.c_void_p),
                ('step_d', C.c_void_p), ('lr', C.c_double), ('beta1', C.c_double), ('beta2', C.c_double),
                ('eps', C.c_double), ('max_norm', C.c_double), ('norm_out_d', C.c_void_p), ('expect_steps', C.c_int32),
                ('loss_out_d', C.c_void_p)]


_lib = None


def load():
    """Load (once) and return the ctypes handle; raises if the library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            'prob_mbrl_amd: %s is missing -- build the HIP extension first '
            '(python -c "import __graft_entry__ as g; g.build()"). There is no '
            'CPU fallback for the rollout hot path.' % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    vp, i32, i64, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float
    lib.pmbrl_last_error.restype = C.c_char_p
    lib.pmbrl_last_error.argtypes = []
    lib.pmbrl_version.restype = C.c_int
    lib.pmbrl_build_id.restype = C.c_char_p
    lib.pmbrl_build_id.argtypes = []
    lib.pmbrl_graph_capture_begin.restype = C.c_int
    lib.pmbrl_graph_capture_begin.argtypes = [vp]
    lib.pmbrl_graph_capture_end.restype = C.c_int
    lib.pmbrl_graph_capture_end.argtypes = [vp, C.POINTER(vp)]
    lib.pmbrl_graph_launch.restype = C.c_int
    lib.pmbrl_graph_launch.argtypes = [vp, vp]
    lib.pmbrl_graph_num_nodes.restype = C.c_int
    lib.pmbrl_graph_num_nodes.argtypes = [vp, C.POINTER(C.c_int64)]
    lib.pmbrl_graph_destroy.restype = None
    lib.pmbrl_graph_destroy.argtypes = [vp]
    lib.pmbrl_plan_create.restype = C.c_int
    lib.pmbrl_plan_create.argtypes = [C.POINTER(Config), C.c_int, C.POINTER(vp)]
    lib.pmbrl_plan_destroy.restype = None
    lib.pmbrl_plan_destroy.argtypes = [vp]
    lib.pmbrl_plan_workspace_bytes.restype = C.c_size_t
    lib.pmbrl_plan_workspace_bytes.argtypes = [vp]
    lib.pmbrl_plan_info.restype = C.c_int
    lib.pmbrl_plan_info.argtypes = [vp, C.POINTER(i32)]
    lib.pmbrl_pack_mask.restype = C.c_int
    lib.pmbrl_pack_mask.argtypes = [vp, vp, i32, i32, i32, vp]
    lib.pmbrl_rollout_fwd.restype = C.c_int
    lib.pmbrl_rollout_fwd.argtypes = [vp, vp, vp, C.POINTER(Inputs), vp, vp, vp, vp]
    lib.pmbrl_rollout_bwd.restype = C.c_int
    lib.pmbrl_rollout_bwd.argtypes = [vp, vp, vp, C.POINTER(Inputs), vp, vp, vp,
                                      vp, vp, vp, vp, vp, vp, vp]
    lib.pmbrl_weighted_sum.restype = C.c_int
    lib.pmbrl_weighted_sum.argtypes = [vp, vp, vp, i64, vp]
    lib.pmbrl_weighted_sum_steps.restype = C.c_int
    lib.pmbrl_weighted_sum_steps.argtypes = [vp, vp, vp, i64, i32, vp, vp]
    lib.pmbrl_clip_adam.restype = C.c_int
    f64 = C.c_double
    lib.pmbrl_clip_adam.argtypes = [vp, vp, vp, vp, vp, i64, i64, f64, f64, f64,
                                    f64, f64, vp]
    lib.pmbrl_clip_adam_guarded.restype = C.c_int
    lib.pmbrl_clip_adam_guarded.argtypes = [vp, vp, vp, vp, vp, i64, vp, f64, f64, f64, f64, f64, vp, vp, i32]
    lib.pmbrl_mlp_workspace_bytes.restype = C.c_size_t
    lib.pmbrl_mlp_workspace_bytes.argtypes = [C.POINTER(MlpCall)]
    lib.pmbrl_mlp_forward.restype = C.c_int
    lib.pmbrl_mlp_forward.argtypes = [vp, C.POINTER(MlpCall), vp, vp, vp, C.POINTER(vp)] + [vp] * 10
    lib.pmbrl_mlp_grad_input.restype = C.c_int
    lib.pmbrl_mlp_grad_input.argtypes = [vp, C.POINTER(MlpCall), vp, vp, vp, C.POINTER(vp)] + [vp] * 11
    lib.pmbrl_bnn_plan_create.restype = C.c_int
    lib.pmbrl_bnn_plan_create.argtypes = [C.POINTER(BnnConfig), C.c_int, C.POINTER(vp)]
    lib.pmbrl_bnn_plan_destroy.restype = None
    lib.pmbrl_bnn_plan_destroy.argtypes = [vp]
    lib.pmbrl_bnn_plan_workspace_bytes.restype = C.c_size_t
    lib.pmbrl_bnn_plan_workspace_bytes.argtypes = [vp]
    lib.pmbrl_bnn_plan_n_params.restype = C.c_int64
    lib.pmbrl_bnn_plan_n_params.argtypes = [vp]
    lib.pmbrl_bnn_loss_grad.restype = C.c_int
    lib.pmbrl_bnn_loss_grad.argtypes = [vp] * 11
    lib.pmbrl_bnn_loss_grad_ex.restype = C.c_int
    lib.pmbrl_bnn_loss_grad_ex.argtypes = [vp] * 13 + [i32]
    lib.pmbrl_bnn_train_steps.restype = C.c_int
    lib.pmbrl_bnn_train_steps.argtypes = ([vp] * 6 + [i32] + [vp] * 4 + [C.c_double] * 4 + [C.c_uint64] * 2 + [vp] * 4)
    lib.pmbrl_debug_linear.restype = C.c_int
    lib.pmbrl_debug_linear.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, vp, vp]
    lib.pmbrl_plan_set_replay.restype = C.c_int
    lib.pmbrl_plan_set_replay.argtypes = [vp, C.c_int]
    lib.pmbrl_plan_replay_count.restype = C.c_int
    lib.pmbrl_plan_replay_count.argtypes = [vp, C.POINTER(C.c_int64)]
    lib.pmbrl_plan_set_timing.restype = C.c_int
    lib.pmbrl_plan_set_timing.argtypes = [vp, C.c_int]
    lib.pmbrl_plan_read_timing.restype = C.c_int
    lib.pmbrl_plan_read_timing.argtypes = [vp, C.POINTER(C.c_float)]
    lib.pmbrl_plan_set_prof.restype = C.c_int
    lib.pmbrl_plan_set_prof.argtypes = [vp, vp, vp]
    lib.pmbrl_comm_unique_id.restype = C.c_int
    lib.pmbrl_comm_unique_id.argtypes = [vp]
    lib.pmbrl_comm_init.restype = C.c_int
    lib.pmbrl_comm_init.argtypes = [vp, i32, i32, i32, C.POINTER(vp)]
    lib.pmbrl_allreduce_sum.restype = C.c_int
    lib.pmbrl_allreduce_sum.argtypes = [vp, vp, vp, i64]
    lib.pmbrl_plan_set_loss.restype = C.c_int
    lib.pmbrl_plan_set_loss.argtypes = [vp, vp, vp]
    lib.pmbrl_rollout_bwd_adam.restype = C.c_int
    lib.pmbrl_rollout_bwd_adam.argtypes = [vp, vp, vp, C.POINTER(Inputs)] + [vp] * 10 + [C.POINTER(Adam)]
    lib.pmbrl_draw_masks.restype = C.c_int
    lib.pmbrl_draw_masks.argtypes = [vp, i32, C.c_uint64, C.c_uint64, vp, i32, C.c_float, i32, i32, vp, vp, vp, i32, i32, vp, vp, vp]
    lib.pmbrl_comm_count.restype = C.c_int
    lib.pmbrl_comm_count.argtypes = [vp, C.POINTER(i32)]
    lib.pmbrl_comm_destroy.restype = None
    lib.pmbrl_comm_destroy.argtypes = [vp]
    lib.pmbrl_plan_set_comm.restype = C.c_int
    lib.pmbrl_plan_set_comm.argtypes = [vp, vp]
    lib.pmbrl_p2p_create.restype = C.c_int
    lib.pmbrl_p2p_create.argtypes = [i32, i32, i32, i64, C.POINTER(vp)]
    lib.pmbrl_p2p_handle.restype = C.c_int
    lib.pmbrl_p2p_handle.argtypes = [vp, vp]
    lib.pmbrl_p2p_open.restype = C.c_int
    lib.pmbrl_p2p_open.argtypes = [vp, i32, vp]
    for fn in (lib.pmbrl_p2p_allreduce_f32, lib.pmbrl_p2p_allreduce_f64):
        fn.restype = C.c_int
        fn.argtypes = [vp, vp, vp, i64]
    lib.pmbrl_plan_set_p2p.restype = C.c_int
    lib.pmbrl_plan_set_p2p.argtypes = [vp, vp]
    lib.pmbrl_p2p_error.restype = C.c_int
    lib.pmbrl_p2p_error.argtypes = [vp, C.POINTER(i32)]
    lib.pmbrl_p2p_destroy.restype = None
    lib.pmbrl_p2p_destroy.argtypes = [vp]
    lib.pmbrl_plan_set_collective.restype = C.c_int
    lib.pmbrl_plan_set_collective.argtypes = [vp, COLLECTIVE_FN, vp]
    _lib = lib
    return lib


class PmbrlError(Exception):
    """A C-ABI call failed: bad configuration, unsupported shape or a HIP error.  Deliberately NOT a
    RuntimeError: the reference's control flow treats RuntimeError as "this rollout failed
    numerically, resample and carry on" (algorithms/mc_pilco.py:122-131, utils/rollout.py:154-157);
    a usage error must surface instead of being retried for opt_iters iterations."""


def check(rc, what):
    if rc != 0:
        msg = load().pmbrl_last_error()
        raise PmbrlError('%s failed (%d): %s' %
                         (what, rc, msg.decode() if msg else ''))
