"""ctypes binding of include/zkir_hip.h and a thin Python mirror of the reference's
interface for this path (Rng / Generator / CircomTranspiler of razn-v/zkir, plus the
new Lowerer and the GPU evaluator).

Nothing here computes: every call goes to libzkir_hip.so.  If the library is
missing, importing this module raises; if no gfx950 device is present, creating a
Context raises.  There is no CPU fallback.
"""
import ctypes
import os
from ctypes import POINTER, byref, c_char_p, c_double, c_int32, c_size_t, c_uint32, c_uint64, c_void_p

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ZKIR_LIB") or os.path.join(HERE, "libzkir_hip.so")  # ZKIR_LIB: another build of the library (A/B measurements)

OK, E_INVALID, E_NOMEM, E_HIP, E_NODEVICE, E_LOWER, E_BUFSIZE, E_LIMIT, E_INTERNAL, E_UNSUPPORTED, E_BUSY = 0, -1, -2, -3, -4, -5, -6, -7, -8, -9, -10
LOWER_CLASSES = {0: "OK", 1: "DoubleAssign", 2: "OOBIndex", 3: "DivByZero", 4: "NonQuadratic", 5: "LoopCap",
                 6: "UnknownControl", 7: "UninitSignal", 8: "BadAst"}


class ZkirError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("zkir error %d: %s" % (code, msg))
        self.code = code


class LowerRejected(ZkirError):
    """lower refused the circuit; status_class is one of LOWER_CLASSES."""

    def __init__(self, status_class, msg):
        ZkirError.__init__(self, E_LOWER, "%s: %s" % (LOWER_CLASSES.get(status_class, status_class), msg))
        self.status_class = status_class


class TapeHdr(ctypes.Structure):
    _fields_ = [(n, c_uint32) for n in ("magic", "version", "n_inputs", "n_outputs", "n_consts", "n_ops", "n_gslots",
                                        "n_lslots", "n_constraints", "n_mul_ops", "flags")] + [("strand_ops", c_uint32 * 8), ("n_xslots", c_uint32), ("n_strands", c_uint32),
                                                                                                ("n_scale_ops", c_uint32)]


class GeneratorConfig(ctypes.Structure):
    """Defaults are the literals of the reference's src/generator.rs (see the header)."""
    _fields_ = [(n, c_uint32) for n in ("min_instructions", "max_instructions", "statement_budget", "max_array_len",
                                        "max_expr_depth", "loop_init_max", "loop_bound_max", "loop_body_max",
                                        "name_len", "inputs_readable", "arithmetic_only", "quadratic_aware",
                                        "forced_inputs", "no_loops", "safe_array_index", "signal_window", "enable_if", "max_linear_terms", "signal_factors")] + [("reserved", c_uint32 * 1)]


class LowerOptions(ctypes.Structure):
    _fields_ = [(n, c_uint32) for n in ("fold_constants", "n_lslots", "reuse_slots", "max_ops", "bnot_variant",
                                        "use_acc", "max_iterations", "window_lifetime", "strands", "max_xslots", "reorder", "window_cache", "order_ops", "scaled_values")]


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError("%s is missing: build it with `python zkir_amd/build.py` (needs hipcc)" % LIB_PATH)
    # One HIP runtime per process.  PyTorch wheels carry their own libamdhip64 / libhsa-runtime64;
    # if the system copies were loaded first and torch's second, the second runtime would find no
    # device.  Importing torch first makes libzkir_hip.so bind to the runtime torch already loaded
    # (same SONAME), so device pointers and streams of torch tensors can be passed straight in.
    # Without torch installed the library binds to the system ROCm runtime.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = ctypes.CDLL(LIB_PATH)
    P = POINTER
    sig = {
        "zkir_generator_config_default": [P(GeneratorConfig)],
        "zkir_generate": [c_uint64, P(GeneratorConfig), P(c_void_p)],
        "zkir_rng_stream": [c_uint64, c_uint32, P(c_uint64)],
        "zkir_circuit_generation_counts": [c_void_p, P(c_uint64), P(c_uint64)],
        "zkir_circuit_parse": [c_char_p, P(c_void_p)],
        "zkir_circuit_print": [c_void_p, c_char_p, c_size_t, P(c_size_t)],
        "zkir_circuit_transpile_circom": [c_void_p, c_char_p, c_size_t, P(c_size_t)],
        "zkir_circuit_transpile_noir": [c_void_p, c_char_p, c_size_t, P(c_size_t)],
        "zkir_circuit_debug": [c_void_p, c_char_p, c_size_t, P(c_size_t)],
        "zkir_gadget_poseidon_style": [c_uint32, c_uint32, c_uint64, P(c_void_p)],
        "zkir_lower_options_default": [P(LowerOptions)],
        "zkir_lower": [c_void_p, P(LowerOptions), P(c_void_p), P(c_int32)],
        "zkir_generate_lower_many": [P(c_uint64), P(GeneratorConfig), c_uint32, P(LowerOptions), c_uint32, P(c_void_p),
                                     P(c_int32)],
        "zkir_host_phase_seconds": [P(c_double), P(c_double)],
        "zkir_tape_header": [c_void_p, P(TapeHdr)],
        "zkir_tape_blob": [c_void_p, P(c_void_p), P(c_size_t)],
        "zkir_tape_io_map": [c_void_p, c_int32, P(c_uint64), c_size_t, P(c_size_t)],
        "zkir_tape_from_blob": [P(TapeHdr), c_void_p, c_size_t, P(c_void_p)],
        "zkir_ctx_create": [c_int32, P(c_void_p)],
        "zkir_ctx_destroy": [c_void_p],
        "zkir_tape_upload": [c_void_p, P(TapeHdr), c_void_p, c_size_t, P(c_uint32)],
        "zkir_tape_release": [c_void_p, c_uint32],
        "zkir_tape_upload_many": [c_void_p, c_uint32, P(TapeHdr), P(c_void_p), P(c_size_t), P(c_uint32)],
        "zkir_tape_upload_handles": [c_void_p, c_uint32, P(c_void_p), P(c_uint32)],
        "zkir_eval_batch": [c_void_p, c_uint32, c_void_p, c_uint64, c_void_p, P(c_uint64), P(c_uint32)],
        "zkir_eval_batch_dev": [c_void_p, c_uint32, c_void_p, c_uint64, c_void_p, c_void_p, c_void_p],
        "zkir_eval_multi": [c_void_p, P(c_uint32), c_uint32, c_void_p, c_uint64, P(c_uint64)],
        "zkir_eval_batch_dev_noalloc": [c_void_p, c_uint32, c_void_p, c_uint64, c_void_p, c_void_p, c_void_p],
        "zkir_eval_multi_dev": [c_void_p, P(c_uint32), c_uint32, c_void_p, c_uint64, c_void_p, c_void_p, c_void_p],
        "zkir_eval_multi_dev_faults": [c_void_p, P(c_uint32), c_uint32, c_void_p, c_uint64, c_void_p, c_void_p, c_void_p, c_void_p],
        "zkir_last_faults": [c_void_p, P(c_uint64)],
        "zkir_reducer_create": [P(c_int32), c_uint32, c_uint32, c_uint32, P(c_void_p)],
        "zkir_reducer_reduce": [c_void_p, P(c_uint64), P(c_uint32), P(c_uint32), P(c_uint64), P(c_uint32), P(c_uint32)],
        "zkir_reducer_destroy": [c_void_p],
        "zkir_measure_int_mul_rate": [c_void_p, c_int32, P(c_double)],
        "zkir_measure_fr_mul_rate": [c_void_p, P(c_double)],
        "zkir_measure_fr_mul_rate29": [c_void_p, P(c_double)],
    }
    for name, args in sig.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = c_int32
    lib.zkir_circuit_free.argtypes = [c_void_p]
    lib.zkir_circuit_free.restype = None
    lib.zkir_tape_free.argtypes = [c_void_p]
    lib.zkir_tape_free.restype = None
    lib.zkir_last_error.argtypes = [c_void_p]
    lib.zkir_last_error.restype = c_char_p
    lib.zkir_reducer_last_error.argtypes = [c_void_p]
    lib.zkir_reducer_last_error.restype = c_char_p
    lib.zkir_host_last_error.argtypes = []
    lib.zkir_host_last_error.restype = c_char_p
    return lib


lib = _load()


def _host_check(rc):
    if rc != OK:
        raise ZkirError(rc, lib.zkir_host_last_error().decode())


def _text(fn, handle):
    need = c_size_t()
    rc = fn(handle, None, 0, byref(need))
    if rc not in (OK, E_BUFSIZE):
        _host_check(rc)
    buf = ctypes.create_string_buffer(need.value)
    _host_check(fn(handle, buf, need.value, byref(need)))
    return buf.value.decode()


def rng_stream(seed, n):
    """First n values of Rng::next for `seed` (reference src/rng.rs:18-24)."""
    out = (c_uint64 * n)()
    _host_check(lib.zkir_rng_stream(seed, n, out))
    return list(out)


def default_generator_config():
    c = GeneratorConfig()
    _host_check(lib.zkir_generator_config_default(byref(c)))
    return c


def default_lower_options():
    o = LowerOptions()
    _host_check(lib.zkir_lower_options_default(byref(o)))
    return o


class Circuit:
    """ast::Circuit (reference src/ast.rs:113-132), held by the library."""

    def __init__(self, handle):
        self._h = c_void_p(handle) if not isinstance(handle, c_void_p) else handle

    def __del__(self):
        if getattr(self, "_h", None) and lib is not None:  # lib is gone when the interpreter shuts down
            lib.zkir_circuit_free(self._h)
            self._h = None

    @staticmethod
    def parse(text):
        h = c_void_p()
        _host_check(lib.zkir_circuit_parse(text.encode(), byref(h)))
        return Circuit(h)

    def text(self):
        return _text(lib.zkir_circuit_print, self._h)

    def generation_counts(self):
        """(instructions asked for, top-level instructions produced): reference src/generator.rs:148,158."""
        a, b = c_uint64(), c_uint64()
        _host_check(lib.zkir_circuit_generation_counts(self._h, byref(a), byref(b)))
        return a.value, b.value

    def debug(self):
        """`{:#?}` of circuit.instructions (what reference src/main.rs:91 prints)."""
        return _text(lib.zkir_circuit_debug, self._h)


class Generator:
    """Rng + Generator of the reference (src/main.rs:81-83) with an explicit seed."""

    def __init__(self, seed, config=None):
        self.seed, self.config = seed, config

    def generate(self):
        h = c_void_p()
        _host_check(lib.zkir_generate(self.seed, byref(self.config) if self.config is not None else None, byref(h)))
        return Circuit(h)


class CircomTranspiler:
    """CircomTranspiler::new(&circuit) / transpile_circuit (reference src/circom.rs:10-15, :412-429)."""

    def __init__(self, circuit):
        self.circuit = circuit

    def transpile_circuit(self):
        return _text(lib.zkir_circuit_transpile_circom, self.circuit._h)


class NoirTranspiler:
    """The second backend the reference names as a goal (README.md:5,10): same call shape as
    CircomTranspiler.  Raises ZkirError with code E_UNSUPPORTED for circuits that use an operator
    or loop Noir's Field type cannot express."""

    def __init__(self, circuit):
        self.circuit = circuit

    def transpile_circuit(self):
        return _text(lib.zkir_circuit_transpile_noir, self.circuit._h)


def poseidon_style_gadget(width, rounds, seed):
    h = c_void_p()
    _host_check(lib.zkir_gadget_poseidon_style(width, rounds, seed, byref(h)))
    return Circuit(h)


class Tape:
    def __init__(self, handle):
        self._h = handle
        self.hdr = TapeHdr()
        _host_check(lib.zkir_tape_header(self._h, byref(self.hdr)))

    def __del__(self):
        if getattr(self, "_h", None) and lib is not None:
            lib.zkir_tape_free(self._h)
            self._h = None

    def blob(self):
        p, n = c_void_p(), c_size_t()
        _host_check(lib.zkir_tape_blob(self._h, byref(p), byref(n)))
        return ctypes.string_at(p, n.value)

    def ops(self):
        """[(code, dst, a, b)]"""
        a = np.frombuffer(self.blob(), dtype=np.uint32, count=self.hdr.n_ops * 4).reshape(-1, 4)
        return [tuple(int(x) for x in r) for r in a]

    def consts(self):
        off = self.hdr.n_ops * 16
        a = np.frombuffer(self.blob(), dtype=np.uint32, count=self.hdr.n_consts * 8, offset=off).reshape(-1, 8)
        return [sum(int(v) << (32 * i) for i, v in enumerate(r)) for r in a]

    def triples(self):
        off = self.hdr.n_ops * 16 + self.hdr.n_consts * 32
        a = np.frombuffer(self.blob(), dtype=np.uint32, count=self.hdr.n_constraints * 3, offset=off).reshape(-1, 3)
        return [tuple(int(x) for x in r) for r in a]

    def io_map(self, which):
        """[(variable id, first element, element count)] for inputs (0) or outputs (1)."""
        n = c_size_t()
        lib.zkir_tape_io_map(self._h, which, None, 0, byref(n))
        buf = (c_uint64 * (3 * max(1, n.value)))()
        _host_check(lib.zkir_tape_io_map(self._h, which, buf, n.value, byref(n)))
        return [(buf[3 * i], buf[3 * i + 1], buf[3 * i + 2]) for i in range(n.value)]

    @staticmethod
    def from_parts(hdr, blob):
        h = c_void_p()
        _host_check(lib.zkir_tape_from_blob(byref(hdr), blob, len(blob), byref(h)))
        return Tape(h)


class Lowerer:
    """Lowerer::new(&circuit).lower(): same shape as the reference's backend structs."""

    def __init__(self, circuit, options=None):
        self.circuit, self.options = circuit, options

    def lower(self):
        h, cls = c_void_p(), c_int32()
        rc = lib.zkir_lower(self.circuit._h, byref(self.options) if self.options is not None else None, byref(h),
                            byref(cls))
        if rc == E_LOWER:
            raise LowerRejected(cls.value, lib.zkir_host_last_error().decode())
        _host_check(rc)
        return Tape(h)


def generate_lower_many(seeds, configs=None, options=None, threads=1):
    """zkir_generate_lower_many: circuit k from (seeds[k], configs[k]) generated and lowered on `threads` host
    threads inside the library.  Returns [(status class, Tape or None)]."""
    n = len(seeds)
    if n == 0:
        return []
    if configs is not None and len(configs) != n:
        raise ValueError("one generator config per seed")
    c_seeds = (c_uint64 * n)(*seeds)
    c_cfgs = (GeneratorConfig * n)(*configs) if configs is not None else None
    tapes, classes = (c_void_p * n)(), (c_int32 * n)()
    _host_check(lib.zkir_generate_lower_many(c_seeds, c_cfgs, n, byref(options) if options is not None else None,
                                             threads, tapes, classes))
    return [(classes[k], Tape(c_void_p(tapes[k])) if tapes[k] else None) for k in range(n)]


class Reducer:
    """zkir_reducer: the campaign's exchange step over RCCL for one process that drives several GPUs."""

    def __init__(self, devices, n_counters, id_capacity):
        self.n, self.n_counters, self.capacity = len(devices), n_counters, id_capacity
        h = c_void_p()
        rc = lib.zkir_reducer_create((c_int32 * self.n)(*devices), self.n, n_counters, id_capacity, byref(h))
        if rc != OK:
            raise ZkirError(rc, "zkir_reducer_create")
        self._h = h

    def reduce(self, counters, id_lists):
        """counters: one list of n_counters ints per device; id_lists: one list of divergent ids per device.
        Returns (totals, sorted union of the ids)."""
        flat = (c_uint64 * (self.n * self.n_counters))(*[v for row in counters for v in row])
        ids = (c_uint32 * max(1, self.n * self.capacity))()
        n_ids = (c_uint32 * self.n)(*[len(l) for l in id_lists])
        for k, l in enumerate(id_lists):
            for j, v in enumerate(l[:self.capacity]):
                ids[k * self.capacity + j] = v
        totals = (c_uint64 * self.n_counters)()
        out = (c_uint32 * max(1, self.n * self.capacity))()
        n_out = c_uint32()
        rc = lib.zkir_reducer_reduce(self._h, flat, ids, n_ids, totals, out, byref(n_out))
        if rc != OK:
            raise ZkirError(rc, lib.zkir_reducer_last_error(self._h).decode())
        return list(totals), list(out[:n_out.value])

    def close(self):
        if self._h:
            lib.zkir_reducer_destroy(self._h)
            self._h = None


def host_phase_seconds():
    """(generate, lower): thread-seconds generate_lower_many has spent in each so far, in this process."""
    g, l = c_double(), c_double()
    _host_check(lib.zkir_host_phase_seconds(byref(g), byref(l)))
    return g.value, l.value


def pack_inputs(vectors, n_inputs):
    """list of per-vector lists of ints -> limb-major SoA uint32 [n_inputs][8][N]."""
    n = len(vectors)
    a = np.zeros((n_inputs, 8, n), dtype=np.uint32)
    for t, vec in enumerate(vectors):
        for s, x in enumerate(vec):
            for l in range(8):
                a[s, l, t] = (x >> (32 * l)) & 0xFFFFFFFF
    return a


def unpack_outputs(a):
    """SoA uint32 [n_outputs][8][N] -> list of per-vector lists of ints."""
    n_out, _, n = a.shape
    return [[sum(int(a[s, l, t]) << (32 * l) for l in range(8)) for s in range(n_out)] for t in range(n)]


class Context:
    """One per device.  Raises if there is no gfx950 device: nothing falls back to the CPU."""

    def __init__(self, device=0):
        self._h = c_void_p()
        rc = lib.zkir_ctx_create(device, byref(self._h))
        if rc != OK:
            self._h = None
            raise ZkirError(rc, "zkir_ctx_create failed (no gfx950 device?)")
        self._tapes = {}

    def close(self):
        if getattr(self, "_h", None) and lib is not None:
            lib.zkir_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def _check(self, rc):
        if rc != OK:
            raise ZkirError(rc, lib.zkir_last_error(self._h).decode())

    def upload(self, tape):
        tid = c_uint32()
        blob = tape.blob()
        self._check(lib.zkir_tape_upload(self._h, byref(tape.hdr), blob, len(blob), byref(tid)))
        self._tapes[tid.value] = (tape.hdr.n_inputs, tape.hdr.n_outputs)
        return tid.value

    def upload_many(self, tapes):
        """zkir_tape_upload_many: one allocation, one copy and one conversion kernel for all of `tapes`."""
        n = len(tapes)
        if n == 0:
            return []
        handles = (c_void_p * n)(*[t._h for t in tapes])  # the library's own tapes: validated when they were made
        ids = (c_uint32 * n)()
        self._check(lib.zkir_tape_upload_handles(self._h, n, handles, ids))
        for k, t in enumerate(tapes):
            self._tapes[ids[k]] = (t.hdr.n_inputs, t.hdr.n_outputs)
        return list(ids)

    def upload_many_parts(self, parts):
        """zkir_tape_upload_many for (header, blob bytes) pairs from anywhere: every op is validated before
        anything is uploaded."""
        n = len(parts)
        hdrs = (TapeHdr * n)(*[h for h, _ in parts])
        keep = [ctypes.create_string_buffer(b, len(b)) for _, b in parts]
        blobs = (c_void_p * n)(*[ctypes.addressof(k) for k in keep])
        sizes = (c_size_t * n)(*[len(b) for _, b in parts])
        ids = (c_uint32 * n)()
        self._check(lib.zkir_tape_upload_many(self._h, n, hdrs, blobs, sizes, ids))
        for k, (h, _) in enumerate(parts):
            self._tapes[ids[k]] = (h.n_inputs, h.n_outputs)
        return list(ids)

    def release(self, tid):
        self._check(lib.zkir_tape_release(self._h, tid))
        self._tapes.pop(tid, None)

    def eval_batch(self, tid, inputs_soa, n, want_outputs=True):
        """Host arrays in, host arrays out: (outputs_soa or None, unsat_count, first_unsat, faults)."""
        n_in, n_out = self._tapes[tid]
        if n_in:
            inputs_soa = np.ascontiguousarray(inputs_soa, dtype=np.uint32)
            assert inputs_soa.shape == (n_in, 8, n), (inputs_soa.shape, (n_in, 8, n))
        out = np.zeros((n_out, 8, n), dtype=np.uint32) if want_outputs else None
        unsat, first = c_uint64(), (c_uint32 * 2)()
        self._check(lib.zkir_eval_batch(self._h, tid, inputs_soa.ctypes.data if n_in else None, n,
                                        out.ctypes.data if out is not None else None, byref(unsat), first))
        faults = c_uint64()
        self._check(lib.zkir_last_faults(self._h, byref(faults)))
        fu = None if first[0] == 0xFFFFFFFF else (first[0], first[1])
        return out, unsat.value, fu, faults.value

    def eval_batch_dev(self, tid, inputs_ptr, n, outputs_ptr, result_ptr, stream=None):
        """Device pointers (ints), asynchronous on `stream` (a hipStream_t as int, or None)."""
        self._check(lib.zkir_eval_batch_dev(self._h, tid, inputs_ptr, n, outputs_ptr, result_ptr, stream))

    def eval_multi(self, tids, inputs_concat, n_per_tape):
        ids = (c_uint32 * len(tids))(*tids)
        unsat = (c_uint64 * max(1, len(tids)))()
        ptr = None
        if inputs_concat is not None and inputs_concat.size:
            inputs_concat = np.ascontiguousarray(inputs_concat, dtype=np.uint32)
            ptr = inputs_concat.ctypes.data
        self._check(lib.zkir_eval_multi(self._h, ids, len(tids), ptr, n_per_tape, unsat))
        faults = c_uint64()
        self._check(lib.zkir_last_faults(self._h, byref(faults)))
        return [unsat[i] for i in range(len(tids))], faults.value

    def eval_batch_dev_noalloc(self, tid, inputs_ptr, n, outputs_ptr, result_ptr, stream=None):
        """eval_batch_dev on the library's fixed scratch array: the call allocates nothing."""
        self._check(lib.zkir_eval_batch_dev_noalloc(self._h, tid, inputs_ptr, n, outputs_ptr, result_ptr, stream))

    def eval_multi_dev(self, tids, inputs_ptr, n_per_tape, unsat_ptr, stream=None, result_ptr=None, faults_ptr=None):
        """result_ptr: None, or 4 uint64 of device memory for {0, first unsat pair, faults, reserved}.
        faults_ptr: None, or one uint64 per tape of device memory for the vectors of that tape that faulted."""
        ids = (c_uint32 * len(tids))(*tids)
        if faults_ptr is None:
            self._check(lib.zkir_eval_multi_dev(self._h, ids, len(tids), inputs_ptr, n_per_tape, unsat_ptr, result_ptr, stream))
        else:
            self._check(lib.zkir_eval_multi_dev_faults(self._h, ids, len(tids), inputs_ptr, n_per_tape, unsat_ptr, faults_ptr,
                                                       result_ptr, stream))

    def int_mul_rate(self, which):
        d = c_double()
        self._check(lib.zkir_measure_int_mul_rate(self._h, which, byref(d)))
        return d.value

    def fr_mul_rate(self, limbs29=False):
        """Montgomery products per second in registers: the 8 x 32-bit multiplier, or the 9 x 29-bit one."""
        d = c_double()
        self._check((lib.zkir_measure_fr_mul_rate29 if limbs29 else lib.zkir_measure_fr_mul_rate)(self._h, byref(d)))
        return d.value


def lib_noalloc_slot_rows():
    """ZKIR_NOALLOC_SLOT_ROWS of include/zkir_hip.h: witness-slot rows in the fixed scratch of
    zkir_eval_batch_dev_noalloc (64 MiB of 9 KiB rows, the first row holds the work counter)."""
    return (64 << 20) // 9216 - 1
