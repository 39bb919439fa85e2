"""ctypes binding of libgator_hip.so (C ABI: include/gator_hip.h).  No fallback: if the library is missing or
does not load, importing the HIP path raises -- the product never routes through a CPU implementation."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# GATOR_AMD_LIB selects another build of the same ABI (the diagnostic library of `python -m gator_amd.build --diag`)
LIB_PATH = os.environ.get('GATOR_AMD_LIB') or os.path.join(_HERE, 'lib', 'libgator_hip.so')

GATOR_F32, GATOR_I64, GATOR_I32 = 0, 1, 2
IMPL_FUSED, IMPL_BASIC = 0, 1
PART_GAT, PART_MDR = 1, 2


class GatorTensor(ctypes.Structure):
    _fields_ = [('name', ctypes.c_char_p), ('data', ctypes.c_void_p), ('dtype', ctypes.c_int32), ('ndim', ctypes.c_int32),
                ('shape', ctypes.c_int64 * 4), ('is_host', ctypes.c_int32), ('reserved', ctypes.c_int32)]


class GemmProblem(ctypes.Structure):            # include/gator_train.h: gator_gemm_problem
    _fields_ = [('A', ctypes.c_void_p), ('B', ctypes.c_void_p), ('C', ctypes.c_void_p), ('a_rowsum', ctypes.c_void_p),
                ('M', ctypes.c_int32), ('N', ctypes.c_int32), ('K', ctypes.c_int32), ('ksplit', ctypes.c_int32),
                ('stride_a', ctypes.c_int64 * 2), ('stride_b', ctypes.c_int64 * 2), ('stride_c', ctypes.c_int64 * 2),
                ('alpha', ctypes.c_float), ('accumulate', ctypes.c_int32), ('wg_begin', ctypes.c_int32), ('fin_begin', ctypes.c_int32),
                ('ws_off', ctypes.c_int64), ('total_wgs', ctypes.c_int32), ('total_fin', ctypes.c_int32), ('bias', ctypes.c_void_p)]


ABI_VERSION = 2                                 # include/gator_hip.h: GATOR_ABI_VERSION this binding was written against
EDEVICE, EDEVICE_DEFERRED = -7, -8
REASON_PERSIST_INCOMPLETE, REASON_NONFINITE = 1, 2      # gator_status_reason


class GatorConfig(ctypes.Structure):
    _fields_ = [('struct_size', ctypes.c_int32), ('num_joint', ctypes.c_int32), ('alpha', ctypes.c_int32), ('impl', ctypes.c_int32),
                ('max_batch', ctypes.c_int32), ('parts', ctypes.c_int32), ('subbatch_streams', ctypes.c_int32), ('arithmetic', ctypes.c_int32)]


# every symbol include/gator_hip.h and include/gator_train.h declare: name -> (restype, argtypes)
_P, _I, _L = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64
SIGNATURES = {
    'gator_create': (_I, [ctypes.POINTER(GatorTensor), _I, ctypes.POINTER(GatorConfig), ctypes.POINTER(_P)]),
    'gator_destroy': (_I, [_P]),
    'gator_device_status': (_I, [_P, _I]),
    'gator_status_reason': (_I, [_P]),
    'gator_c3_state': (_I, [_P, _P]),
    'gator_abi_version': (_I, []),
    'gator_forward_f32': (_I, [_P, _P, _I, _P, _P, _P]),
    'gator_forward_bf16': (_I, [_P, _P, _I, _P, _P, _P]),
    'gator_upsample_bf16': (_I, [_P, _P, _I, _P, _P]),
    'gator_gat_forward_f32': (_I, [_P, _P, _I, _P, _P, _P]),
    'gator_mdr_forward_f32': (_I, [_P, _P, _I, _P, _P]),
    'gator_upsample_f32': (_I, [_P, _P, _I, _P, _P]),
    'gator_get_tap': (_I, [_P, ctypes.c_char_p, _P, _L, ctypes.POINTER(_L), _P]),
    'gator_enable_block_taps': (_I, [_P, _I]),
    'gator_set_encoder': (_I, [_P, _I]),
    'gator_encoder_for_batch': (_I, [_P, _I]),
    'gator_set_graph_replay': (_I, [_P, _I]),
    'gator_profile_enable': (_I, [_P, _I]),
    'gator_profile_read': (_I, [_P, ctypes.c_char_p, _L, _P, _P, _I, _P]),
    'gator_regress_joints_f32': (_I, [_P, _I, _P, _P, _P, _I, _I, _P, _P]),
    'gator_set_joint_regressor': (_I, [_P, _P, _P, _P, _I, _I]),
    'gator_forward_joints_f32': (_I, [_P, _P, _I, _P, _P, _P, _P]),
    'gator_preprocess_pose2d_f32': (_I, [_P, _I, _I, _I, _I, _P, _P]),
    'gator_preprocess_chain_f32': (_I, [_P, _I, _I, _I, _I, _P, _P, _P, _I, _I, _I, _P, _P, _P]),
    'gator_joint_errors_f32': (_I, [_P, _P, _I, _I, _P, _I, _I, ctypes.c_float, _P, _P]),
    'gator_rigid_align_f32': (_I, [_P, _P, _I, _I, _P, _P]),
    'gator_comm_unique_id': (_I, [_P]),
    'gator_comm_create': (_I, [_P, _I, _I, ctypes.POINTER(_P)]),
    'gator_comm_destroy': (_I, [_P]),
    'gator_allgather_verts': (_I, [_P, _P, _P, _I, _I, _P, _P, _P]),
    'gator_emulate_gather_traffic': (_I, [_P, _P, _L, _I, _I, _P]),
    'gator_floyd_warshall': (_I, [_P, _I, _P, _P]),
    'gator_gen_edge_input': (_I, [_P, _P, _I, _I, _P]),
    'gator_verts_joints_relation': (_I, [_P, _I, _P, _I, _P]),
    # include/gator_train.h
    'gator_t_binary': (_I, [_I, _P, _P, _P, _P, _P, _P, _P, _P]),
    'gator_t_unary': (_I, [_I, _P, _P, _P, _P, _P, ctypes.c_float, ctypes.c_float, _P]),
    'gator_t_reduce_ws_bytes': (_L, [_P, _P]),
    'gator_t_reduce_sum': (_I, [_P, _P, _P, _P, _P, _I, _P, _P]),
    'gator_t_gemm': (_I, [_P, _P, _P, _I, _I, _I, _P, _P, _P, _I, _I, _P, _P, _P, _P, ctypes.c_float, _I, _I, _P, _P, _P]),
    'gator_t_gemm_grouped_prepare': (_L, [_P, _I]),
    'gator_t_gemm_grouped': (_I, [_P, _I, _P, _P, _P]),
    'gator_t_attn_fwd': (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, ctypes.c_float, ctypes.c_float, ctypes.c_uint64, ctypes.c_uint64, _P, _P]),
    'gator_t_attn_bwd': (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, ctypes.c_float, ctypes.c_float, ctypes.c_uint64, ctypes.c_uint64, _P, _I, _P]),
    'gator_t_attn_small_fwd': (_I, [_P, _P, _P, _P, _I, _I, _I, _I, ctypes.c_float, ctypes.c_float, ctypes.c_uint64, ctypes.c_uint64, _P, _P]),
    'gator_t_attn_small_bwd': (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, ctypes.c_float, ctypes.c_float, ctypes.c_uint64, ctypes.c_uint64, _P, _P]),
    'gator_t_mgcn_fwd': (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    'gator_t_mgcn_bwd': (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    'gator_t_batchnorm_fwd': (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, ctypes.c_float, ctypes.c_float, _P]),
    'gator_t_batchnorm_bwd': (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    'gator_t_struct_size': (_L, [_I]),
    'gator_t_layernorm_fwd': (_I, [_P, _L, _I, _P, _P, ctypes.c_float, _I, _P, _P, _P, _P]),
    'gator_t_layernorm_bwd': (_I, [_P, _P, _P, _P, _P, _L, _I, ctypes.c_float, _I, _P, _P, _P, _P]),
    'gator_t_add_n': (_I, [_P, _P, _P, _P, _P, _L, _P]),
    'gator_t_softmax_fwd': (_I, [_P, _L, _I, _P, _P]),
    'gator_t_softmax_bwd': (_I, [_P, _P, _L, _I, _P, _P]),
    'gator_t_dropout': (_I, [_P, _L, ctypes.c_float, ctypes.c_uint64, ctypes.c_uint64, _P, _P, _P, _P]),
    'gator_t_step_advance': (_I, [_P, _P]),
    'gator_t_drop_fused': (_I, [_P, _P, _L, _L, _I, ctypes.c_float, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_float, ctypes.c_uint64, _P, _P, _P, _P, _P]),
    'gator_t_drop_fused_bwd': (_I, [_P, _P, _P, _P, _L, _L, _I, ctypes.c_float, _P, _P]),
    'gator_t_mask_scale': (_I, [_P, _P, _L, ctypes.c_float, _P, _P]),
    'gator_t_adam': (_I, [_P, _P, _P, _P, _L, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_double, _I, _P, _P]),
    'gator_t_loss_ws_bytes': (_L, [_L, _L]),
    'gator_t_coord_loss': (_I, [_P, _P, _P, _P, _P, ctypes.c_float, _P, _P, _P, _P]),
    'gator_t_normal_loss': (_I, [_P, _P, _P, _P, _P, _L, _L, _L, ctypes.c_float, _P, _P, _P, _P]),
    'gator_t_edge_loss': (_I, [_P, _P, _P, _P, _P, _L, _L, _L, ctypes.c_float, _P, _P, _P, _P]),
    'gator_last_error': (ctypes.c_char_p, []),
    'gator_version': (ctypes.c_char_p, []),
}

_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError('libgator_hip.so not built (%s); run `python -m gator_amd.build` -- there is no CPU fallback'
                               % LIB_PATH)
        # torch first: it ships its own libamdhip64, and the HIP runtime that is loaded FIRST is the one every later library binds to.  Loaded
        # before torch, this library would pull in the system's copy and the process would hold two runtimes -- tensors in one, gator_create
        # ("no HIP device available") in the other (seen with __graft_entry__.build() followed by smoke() in one process).
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        if lib.gator_abi_version() != ABI_VERSION:
            raise RuntimeError('%s is ABI %d, this binding is ABI %d: rebuild with `python -m gator_amd.build`' % (LIB_PATH, lib.gator_abi_version(), ABI_VERSION))
        _lib = lib
    return _lib


class DeviceStatusError(RuntimeError):
    """GATOR_EDEVICE: a kernel of an EARLIER call on the ctx flagged its result as invalid (include/gator_hip.h: gator_device_status)."""
    code = EDEVICE
    reason = 0          # gator_status_reason, filled in by the caller that knows the ctx
    outputs = None


class DeferredDeviceStatus(DeviceStatusError):
    """GATOR_EDEVICE_DEFERRED: the same report carried by a forward that was queued normally.  `outputs` holds the tensors that forward
    is writing (valid unless the next call reports again), so a caller that must stay in step with other ranks can go on with them."""
    code = EDEVICE_DEFERRED


def check(rc, what):
    if rc != 0:
        cls = {EDEVICE: DeviceStatusError, EDEVICE_DEFERRED: DeferredDeviceStatus}.get(rc, RuntimeError)
        raise cls('%s failed (%d): %s' % (what, rc, load().gator_last_error().decode()))
