"""ctypes binding of libamfwd.so (C ABI: include/amfwd.h).

The library is the product's only compute path.  If it is missing or cannot be loaded this module
raises -- there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libamfwd.so")

AM_OK = 0
AM_NET_FF, AM_NET_LSTM = 0, 1
AM_CELL_LSTM, AM_CELL_GRU, AM_CELL_MGRU, AM_CELL_PEEPHOLE = 0, 1, 2, 3
AM_ACT_RELU, AM_ACT_SIGMOID, AM_ACT_TANH = 0, 1, 2
AM_COMBINE_LOGITS, AM_COMBINE_LOGPOST = 0, 1
AM_PRECISION_F32, AM_PRECISION_SPLIT_F16 = 0, 1
AM_KERNEL_ID_MAX = 16

c_float_p = C.POINTER(C.c_float)
c_int_p = C.POINTER(C.c_int)
c_int64_p = C.POINTER(C.c_int64)
c_void_pp = C.POINTER(C.c_void_p)

# name -> (restype, argtypes); every symbol declared in include/amfwd.h
SIGNATURES = {
    "am_device_count": (C.c_int, []),
    "am_device_pci_bus_id": (C.c_int, [C.c_int, C.c_char_p, C.c_int]),
    "am_pack_stream_order": (C.c_int, [C.POINTER(C.c_float), C.c_int, C.c_int, C.POINTER(C.c_float)]),
    "am_last_error": (C.c_char_p, []),
    "am_version": (C.c_char_p, []),
    "am_engine_create": (C.c_int, [C.c_int, c_void_pp]),
    "am_engine_destroy": (C.c_int, [C.c_void_p]),
    "am_engine_synchronize": (C.c_int, [C.c_void_p]),
    "am_engine_device_name": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
    "am_engine_set_precision": (C.c_int, [C.c_void_p, C.c_int]),
    "am_pipeline_overflowed": (C.c_int, [C.c_void_p, c_int_p]),
    "am_event_record": (C.c_int, [C.c_void_p, C.c_int]),
    "am_event_elapsed_ms": (C.c_int, [C.c_void_p, C.c_int, C.c_int, c_float_p]),
    "am_decode_elapsed_ms": (C.c_int, [C.c_void_p, c_float_p]),
    "am_profile_enable": (C.c_int, [C.c_void_p, C.c_int]),
    "am_profile_reset": (C.c_int, [C.c_void_p]),
    "am_profile_read": (C.c_int, [C.c_void_p, c_int64_p, C.POINTER(C.c_double)]),
    "am_kernel_name": (C.c_char_p, [C.c_int]),
    "am_host_alloc": (C.c_int, [C.c_size_t, c_void_pp]),
    "am_host_free": (C.c_int, [C.c_void_p]),
    "am_model_create": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, c_int_p, c_int_p,
                                  C.POINTER(c_float_p), C.POINTER(c_float_p), C.POINTER(c_float_p), c_void_pp]),
    "am_model_create_tdnn": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, c_int_p, c_int_p, C.POINTER(c_float_p), C.POINTER(c_float_p),
                                       c_float_p, c_float_p, C.c_int, c_void_pp]),
    "am_model_create_rnn": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                      C.POINTER(c_float_p), c_float_p, c_float_p, c_void_pp]),
    "am_model_destroy": (C.c_int, [C.c_void_p]),
    "am_pipeline_create": (C.c_int, [C.c_void_p, c_void_pp]),
    "am_pipeline_destroy": (C.c_int, [C.c_void_p]),
    "am_pipeline_set_frontend": (C.c_int, [C.c_void_p, C.c_int, C.c_int, c_int_p, c_float_p, c_float_p, C.c_int]),
    "am_pipeline_set_members": (C.c_int, [C.c_void_p, C.c_int, c_void_pp, C.c_int, C.c_int]),
    "am_pipeline_set_rpl4": (C.c_int, [C.c_void_p, c_float_p, c_float_p, c_float_p]),
    "am_pipeline_set_prior": (C.c_int, [C.c_void_p, c_float_p]),
    "am_pipeline_set_normalise": (C.c_int, [C.c_void_p, C.c_int]),
    "am_ff_upload": (C.c_int, [C.c_void_p, c_float_p, c_float_p, C.c_int64, C.c_int, C.c_int]),
    "am_ff_run": (C.c_int, [C.c_void_p]),
    "am_ff_download": (C.c_int, [C.c_void_p, c_float_p]),
    "am_forward_ff": (C.c_int, [C.c_void_p, c_float_p, c_float_p, C.c_int64, C.c_int, C.c_int, c_float_p]),
    "am_lstm_upload": (C.c_int, [C.c_void_p, c_float_p, c_float_p, c_int64_p, C.c_int, C.c_int]),
    "am_lstm_run": (C.c_int, [C.c_void_p]),
    "am_lstm_download": (C.c_int, [C.c_void_p, c_float_p]),
    "am_forward_lstm": (C.c_int, [C.c_void_p, c_float_p, c_float_p, c_int64_p, C.c_int, C.c_int, c_float_p]),
    "am_lstm_stamps": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, c_int_p, c_int_p]),
    "am_ff_stack_stamps": (C.c_int, [C.c_void_p, C.c_void_p]),
    "am_lstm_fallback_count": (C.c_int, []),
    "am_lstm_persist_enabled": (C.c_int, [C.c_int]),
    "am_lstm_persist_reenable": (C.c_int, [C.c_int]),
    "am_pipeline_recurrent_form": (C.c_int, [C.c_void_p, c_int_p, c_int_p]),
    "am_pipeline_synchronize": (C.c_int, [C.c_void_p]),
    "am_pipeline_output_shape": (C.c_int, [C.c_void_p, c_int64_p, c_int_p]),
    "am_frontend_apply": (C.c_int, [C.c_void_p, c_float_p, c_float_p, C.c_int64, C.c_int, C.c_int, c_float_p]),
    "am_lognormalise": (C.c_int, [C.c_void_p, c_float_p, C.c_int64, C.c_int, c_float_p]),
    "am_decoder_create": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, c_int_p, c_int_p, c_int_p, c_int_p, c_float_p, c_float_p,
                                    c_int_p, c_int_p, c_float_p, C.c_float, C.c_float, c_void_pp]),
    "am_decoder_destroy": (C.c_int, [C.c_void_p]),
    "am_decoder_set_form": (C.c_int, [C.c_void_p, C.c_int]),
    "am_decode_host": (C.c_int, [C.c_void_p, c_float_p, c_int64_p, C.c_int, c_int_p, c_int_p, c_int_p, c_int_p]),
    "am_decode_pipeline": (C.c_int, [C.c_void_p, C.c_void_p, c_int64_p, C.c_int, c_int_p, c_int_p, c_int_p, c_int_p]),
    "am_linear": (C.c_int, [C.c_void_p, c_float_p, C.c_int64, C.c_int, c_float_p, c_float_p, C.c_int, C.c_int, c_float_p]),
}


class AmError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("libamfwd error %d: %s" % (code, message))
        self.code = code


_lib = None


def load():
    """Load libamfwd.so once; raise loudly when it is absent (run ``python -c 'import __graft_entry__ as g; g.build()'``)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise ImportError(
            "%s not found: the HIP extension has not been built (make -C nnacousticmodeling_amd/csrc). "
            "There is no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if a declared symbol is not exported
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def check(code):
    if code != AM_OK:
        raise AmError(code, load().am_last_error().decode("utf-8", "replace"))
