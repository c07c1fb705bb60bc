"""ctypes wrapper of oracle/eval_oracle.c (TEST INFRASTRUCTURE ONLY, see that file)."""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "_build", "libzkir_oracle.so")


def build():
    subprocess.check_call(["make", "-s", "-C", HERE])
    return LIB


def load():
    if not os.path.exists(LIB):
        build()
    lib = ctypes.CDLL(LIB)
    lib.zkir_eval_cpu.restype = ctypes.c_int
    lib.zkir_eval_cpu.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p,
                                  ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint32),
                                  ctypes.POINTER(ctypes.c_uint64), ctypes.c_int]
    lib.zkir_oracle_op.restype = ctypes.c_int
    lib.zkir_oracle_op.argtypes = [ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    return lib


def eval_cpu(hdr_bytes, blob, inputs_soa, n, n_outputs, want_outputs=True, threads=1):
    """Returns (outputs_soa or None, unsat, first_unsat or None, faults)."""
    lib = load()
    out = np.zeros((n_outputs, 8, n), dtype=np.uint32) if want_outputs else None
    unsat, faults, first = ctypes.c_uint64(), ctypes.c_uint64(), (ctypes.c_uint32 * 2)()
    inp = np.ascontiguousarray(inputs_soa, dtype=np.uint32) if inputs_soa is not None else None
    rc = lib.zkir_eval_cpu(hdr_bytes, blob, inp.ctypes.data if inp is not None and inp.size else None, n,
                           out.ctypes.data if out is not None else None, ctypes.byref(unsat), first,
                           ctypes.byref(faults), threads)
    assert rc == 0
    fu = None if first[0] == 0xFFFFFFFF else (first[0], first[1])
    return out, unsat.value, fu, faults.value


def op(code, a, b=0):
    """One operator on canonical integers; returns (result, fault)."""
    lib = load()
    la = (ctypes.c_uint32 * 8)(*[(a >> (32 * i)) & 0xFFFFFFFF for i in range(8)])
    lb = (ctypes.c_uint32 * 8)(*[(b >> (32 * i)) & 0xFFFFFFFF for i in range(8)])
    lr = (ctypes.c_uint32 * 8)()
    fault = lib.zkir_oracle_op(code, la, lb, lr)
    return sum(int(v) << (32 * i) for i, v in enumerate(lr)), fault
