"""ctypes binding of include/cor_asv_ann_hip.h (the C ABI of the HIP hot path).

There is no fallback: if the shared library is missing or a call fails, this raises.
"""
import ctypes
import os
from ctypes import (POINTER, Structure, c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_void_p)

import numpy as np

LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'lib', 'libcor_asv_ann_hip.so')
if os.environ.get('CASV_LIB_PATH'):          # measurement aid: A/B of two builds of the library on one box
    LIB_PATH = os.environ['CASV_LIB_PATH']

CASV_ERR_NAN = -5


class NativeError(RuntimeError):
    def __init__(self, code, message):
        super().__init__('cor_asv_ann_hip error %d: %s' % (code, message))
        self.code = code


class Config(Structure):
    _fields_ = [('depth', c_int32), ('width', c_int32), ('voc_size', c_int32), ('window_width', c_int32),
                ('residual_connections', c_int32), ('deep_bidirectional_encoder', c_int32),
                ('bridge_dense', c_int32), ('lm', c_int32), ('stateful', c_int32)]


class AdamParams(Structure):
    _fields_ = [('lr', c_float), ('beta1', c_float), ('beta2', c_float), ('epsilon', c_float), ('clipnorm', c_float)]


class BeamParams(Structure):
    _fields_ = [('batch_size', c_int32), ('beam_width_in', c_int32), ('beam_width_out', c_int32),
                ('max_results', c_int32), ('beam_threshold_in', c_double), ('rejection_threshold', c_double),
                ('cost0', c_double)]


# name -> (restype, argtypes); every symbol declared in the header is listed here and checked at load
SIGNATURES = {
    'casv_last_error': (c_char_p, []),
    'casv_device_count': (c_int, []),
    'casv_version': (c_char_p, []),
    'casv_model_create': (c_int, [POINTER(Config), c_int, POINTER(c_void_p)]),
    'casv_model_destroy': (None, [c_void_p]),
    'casv_set_weight': (c_int, [c_void_p, c_char_p, c_void_p, c_int64]),
    'casv_get_weight': (c_int, [c_void_p, c_char_p, c_void_p, c_int64]),
    'casv_commit_weights': (c_int, [c_void_p]),
    'casv_encode': (c_int, [c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p]),
    'casv_set_encoder_outputs': (c_int, [c_void_p, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p]),
    'casv_get_encoder_outputs': (c_int, [c_void_p, c_void_p, c_void_p]),
    'casv_decoder_step': (c_int, [c_void_p, c_int32] + [c_void_p] * 7),
    'casv_decode_greedy': (c_int, [c_void_p, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p]),
    'casv_decode_beam': (c_int, [c_void_p, POINTER(BeamParams), c_int32] + [c_void_p] * 8),
    'casv_train_begin': (c_int, [c_void_p, POINTER(AdamParams), c_char_p]),
    'casv_train_step': (c_int, [c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32] + [c_void_p] * 8 + [POINTER(c_double), POINTER(c_double)]),
    'casv_train_get_gradient': (c_int, [c_void_p, c_char_p, c_void_p, c_int64]),
    'casv_train_sync_weights': (c_int, [c_void_p]),
    'casv_train_end': (c_int, [c_void_p]),
    'casv_profile': (c_int, [c_void_p, c_int32]),
    'casv_profile_read': (c_int, [c_void_p, c_char_p, POINTER(c_int64), POINTER(c_double), POINTER(c_double),
                                  POINTER(c_double)]),
    'casv_debug_gemm': (c_int, [c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, POINTER(c_double)]),
    'casv_debug_contract': (c_int, [c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p]),
    'casv_set_option': (c_int, [c_void_p, c_char_p, c_int64]),
    'casv_get_alignments_sparse': (c_int, [c_void_p, c_int32, c_void_p, c_void_p]),
    'casv_comm_unique_id': (c_int, [c_void_p]),
    'casv_comm_init': (c_int, [c_void_p, c_int32, c_int32, c_void_p]),
    'casv_comm_all_gather': (c_int, [c_void_p, c_void_p, c_void_p, c_int64]),
    'casv_comm_all_reduce_max': (c_int, [c_void_p, POINTER(c_double)]),
    'casv_comm_destroy': (c_int, [c_void_p]),
    'casv_records_reset': (c_int, [c_void_p, c_int32, c_int32]),
    'casv_records_append': (c_int, [c_void_p, c_int32]),
    'casv_records_read': (c_int, [c_void_p, c_void_p]),
    'casv_records_device_ptr': (c_int, [c_void_p, POINTER(c_void_p), POINTER(c_int64)]),
    'casv_comm_all_gather_records': (c_int, [c_void_p, c_void_p]),
    'casv_get_stat': (c_int, [c_void_p, c_char_p, POINTER(c_int64)]),
    'casv_realign_path': (c_int, [c_int32, c_int32, c_int32, c_void_p, c_void_p, c_int32, c_int32, c_float, c_void_p, POINTER(c_double)]),
    'casv_synchronize': (c_int, [c_void_p]),
}

_lib = None


def load():
    """Load the HIP library (once).  Raises if it has not been built (`python -c 'import
    __graft_entry__ as g; g.build()'` or `make -C cor_asv_ann_amd/csrc`)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError('HIP library %s not found: build it with `make -C cor_asv_ann_amd/csrc` '
                           '(there is no CPU fallback)' % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (restype, argtypes) in SIGNATURES.items():
        if os.environ.get('CASV_LIB_PATH') and name.startswith('casv_debug_') and not hasattr(lib, name):
            continue                     # (A/B against an older build of the library: it may lack a test-support entry)
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def check(code):
    if code != 0:
        raise NativeError(code, load().casv_last_error().decode('utf-8', 'replace'))


def ptr(a):
    return None if a is None else a.ctypes.data_as(c_void_p)


def carray(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)
