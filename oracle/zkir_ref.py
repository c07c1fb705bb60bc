"""CPU restatement of the reference (razn-v/zkir) in plain Python, plus a direct
big-integer interpreter of its AST.

TEST INFRASTRUCTURE ONLY: nothing under oracle/ is shipped or imported by the
product; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it.

What follows the reference line by line (and is pinned by the known-answer vectors
of tests/golden/, which come from the definitions in src/rng.rs):
    Rng                 src/rng.rs:11-53
    Generator           src/generator.rs:117-602  (same order of random draws)
    transpile_circuit   src/circom.rs:21-429      (same text)
What the reference does NOT have and is defined here from circom's operator
semantics (see fr_ref.py: PARITY UNPINNED against circom itself):
    Interpreter         meaning of a Circuit for one input vector, and the class of
                        circuits circom would refuse to compile.

The AST is kept as plain tuples, shaped like src/ast.rs:
    expr  := ("const", "decimal") | ("var", id, idx_or_None) | (op, e, e) | ("bnot", e)
    instr := ("decl", var_id, default_or_None) | ("assign", (id, idx), e)
           | ("aassign", (id, idx), e_index, e) | ("constrain", (id, idx), e)
           | ("if", cond, [then], [else] or None) | ("for", init, cond, step, [body])
           | ("while", cond, [body])
    var   := dict(id, name, array_len or None, role in {"local", "input", "output"}, init [bool])
"""
import re

from . import fr_ref as F

M64 = (1 << 64) - 1
CHARSET = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz"  # rng.rs:37


class Rng:
    def __init__(self, seed):
        assert seed != 0
        self.s = seed & M64

    def next(self):  # rng.rs:18-24: returns the state before the update
        v = self.s
        self.s ^= (self.s << 13) & M64
        self.s ^= self.s >> 17
        self.s ^= (self.s << 43) & M64
        return v

    def rand(self, lo, hi):  # rng.rs:26-34: no draw when lo == hi
        assert lo <= hi
        if lo == hi:
            return lo
        return lo + self.next() % (hi - lo + 1)

    def rand_string(self, n):  # rng.rs:36-45
        return "".join(CHARSET[self.rand(0, len(CHARSET) - 1)] for _ in range(n))

    def rand_vec(self, v):  # rng.rs:47-52
        if not v:
            return None
        return v[self.rand(0, len(v) - 1)]


BIN_OPS = ["add", "sub", "mul", "pow", "div", "idiv", "rem", "lt", "le", "gt", "ge", "eq", "and", "or", "band",
           "bor", "bnot", "bxor", "shr", "shl"]  # order of generator.rs:509-530 (bnot is unary)
TOKENS = {"add": "+", "sub": "-", "mul": "*", "pow": "**", "div": "/", "idiv": "\\", "rem": "%", "lt": "<",
          "le": "<=", "gt": ">", "ge": ">=", "eq": "==", "and": "&&", "or": "||", "band": "&", "bor": "|",
          "bxor": "^", "shr": ">>", "shl": "<<"}  # circom.rs:35-157


class Generator:
    """generator.rs with its literals as attributes (defaults = the literals)."""

    def __init__(self, rng, **cfg):
        self.rng = rng
        self.min_instructions, self.max_instructions = 15, 30  # :147
        self.statement_budget = 10  # :167
        self.max_array_len = 10  # :224
        self.max_expr_depth = 1  # :486
        self.loop_init_max, self.loop_bound_max, self.loop_body_max = 30, 60, 5  # :406 :416 :436
        self.name_len = 8  # :264
        self.inputs_readable = self.arithmetic_only = self.quadratic_aware = self.no_loops = 0
        self.forced_inputs = self.safe_array_index = self.signal_window = self.enable_if = self.max_linear_terms = 0
        self.signal_factors = 0
        for k, v in cfg.items():
            assert hasattr(self, k), k
            setattr(self, k, v)
        self.n_instr = 0
        self.stack, self.all_vars, self.next_id = [], [], 0
        self.pending = []

    # ScopeStack, generator.rs:48-115
    def add_var(self, name, array_len, role, init):
        v = dict(id=self.next_id, name=name, array_len=array_len, role=role, init=list(init))
        self.stack[-1].append(v)
        self.all_vars.append(dict(v, init=list(init)))  # a copy: later marks do not reach it (:84)
        self.next_id += 1
        return v

    def scope_vars(self):
        return [v for s in self.stack for v in s]

    def mark(self, vid, idx):
        for v in self.scope_vars():
            if v["id"] == vid:
                if idx is not None:
                    v["init"][idx] = True
                else:
                    v["init"] = [True]
                return

    def _push(self, out, ins):
        if ins is not None:
            out.extend(self.pending)
            out.append(ins)
        self.pending = []

    def generate(self):  # :142-164
        instrs = []
        self.stack.append([])
        n = self.rng.rand(self.min_instructions, self.max_instructions)
        self.requested = n  # "Generating {} instructions..." (:148)
        for _ in range(self.forced_inputs):
            instrs.append(self.gen_var_decl(("field", None), "input", None))
        for _ in range(n):
            ins = self.generate_instr()
            self._push(instrs, ins)
            if ins is not None:
                self.n_instr = 0
        self.stack.pop()
        assert not self.stack
        self.generated = len(instrs)  # "Generated {}/{} instructions" (:158)
        return dict(vars=self.all_vars, instrs=instrs)

    def generate_instr(self):  # :166-208
        if self.n_instr >= self.statement_budget:
            return None
        self.n_instr += 1
        sv = self.scope_vars()
        has_vars = any(v["role"] == "local" for v in sv)
        has_arrays = any(v["role"] == "local" and v["array_len"] is not None for v in sv)
        has_out = any(v["role"] == "output" for v in sv)
        funcs = [lambda: self.gen_var_decl(None, None, None)]
        if self.enable_if:  # extension: the entry the reference keeps commented out (:190)
            funcs.append(self.gen_if)
        if not self.no_loops:
            funcs.append(self.gen_for)
        if has_vars:
            funcs.append(self.gen_assign)
        if has_arrays:
            funcs.append(self.gen_array_assign)
        if has_out or self.quadratic_aware:
            funcs.append(self.gen_constraint_quadratic if self.quadratic_aware else self.gen_constraint)
        return funcs[self.rng.rand(0, len(funcs) - 1)]()

    def gen_var_decl(self, type_kind, role_kind, default):  # :214-267
        if type_kind is not None:
            array_len = type_kind[1]
        elif self.rng.rand(0, 1) == 0:
            array_len = None
        else:
            array_len = self.rng.rand(1, self.max_array_len)
        wants_signal = role_kind in ("input", "output")
        if default is None and array_len is None and not wants_signal and self.rng.rand(1, 10) != 1:
            default = self.generate_expr(0)
            if default is None:
                return None
        if role_kind is not None:
            role = role_kind
        elif default is not None:
            role = "local"
        elif self.rng.rand(0, 1) == 0:
            role = "input" if self.rng.rand(0, 1) == 0 else "output"
            if self.quadratic_aware and role == "output":
                role = "local"
            if self.forced_inputs and role == "input":
                role = "local"
        else:
            role = "local"
        readable = bool(self.inputs_readable and role == "input")
        init = [readable] * (array_len if array_len is not None else 1)
        v = self.add_var(self.rng.rand_string(self.name_len), array_len, role, init)
        return ("decl", v["id"], default)

    def gen_assign(self):  # :269-289
        vs = [v["id"] for v in self.scope_vars() if v["role"] == "local" and v["array_len"] is None]
        vid = self.rng.rand_vec(vs)
        if vid is None:
            return None
        e = self.generate_expr(0)
        if e is None:
            return None
        self.mark(vid, None)
        return ("assign", (vid, None), e)

    def gen_array_assign(self):  # :291-327
        vs = [(v["id"], v["array_len"]) for v in self.scope_vars() if v["role"] == "local" and v["array_len"] is not None]
        pick = self.rng.rand_vec(vs)
        if pick is not None:
            e = self.generate_expr(0)
            if e is not None:
                idx = self.rng.rand(0, pick[1] - 1)
                self.mark(pick[0], idx)
                return ("aassign", (pick[0], None), ("const", str(idx)), e)
        return None

    def gen_constraint(self):  # :329-353
        vs = [v["id"] for v in self.scope_vars()
              if v["role"] == "output" and v["array_len"] is None and not all(v["init"])]
        vid = self.rng.rand_vec(vs)
        if vid is not None:
            self.mark(vid, None)  # before the right-hand side is generated (:344)
            e = self.generate_expr(0)
            if e is not None:
                return ("constrain", (vid, None), e)
        return None

    def gen_if(self):  # :355-399
        if self.enable_if == 2:  # extension: a condition lower can decide
            k = ("lt", "eq", "gt")[self.rng.rand(0, 2)]
            cond = (k, ("const", str(self.rng.next())), ("const", str(self.rng.next())))
        else:
            cond = self.generate_expr(0)
        if cond is None:
            return None
        self.stack.append([])
        then = []
        for _ in range(self.rng.rand(1, self.loop_body_max)):
            self._push(then, self.generate_instr())
        self.stack.pop()
        if not then:
            return None
        self.stack.append([])
        other = None
        if self.rng.rand(0, 1) == 0:
            instrs = []
            for _ in range(self.rng.rand(1, self.loop_body_max)):
                self._push(instrs, self.generate_instr())
            if instrs:
                other = instrs
        self.stack.pop()
        return ("if", cond, then, other)

    def gen_for(self):  # :401-456
        self.stack.append([])
        start = self.rng.rand(0, self.loop_init_max)
        init = self.gen_var_decl(("field", None), "local", ("const", str(start)))
        bound = self.rng.rand(start, self.loop_bound_max)
        vid = init[1]
        cond = ("le", ("var", vid, None), ("const", str(bound)))
        step = ("assign", (vid, None), ("add", ("var", vid, None), ("const", "1")))
        body = []
        for _ in range(self.rng.rand(1, self.loop_body_max)):
            self._push(body, self.generate_instr())
        self.stack.pop()
        if not body:
            return None
        return ("for", init, cond, step, body)

    def generate_expr(self, depth):  # :484-543
        if depth >= self.max_expr_depth:
            return ("const", str(self.rng.next()))
        if depth >= 1 and self.rng.rand(1, 3) == 1:  # :490-493, dead at the reference's limit
            return ("const", str(self.rng.next()))
        sv = self.scope_vars()
        has_field = any(v["array_len"] is None and all(v["init"]) for v in sv)
        has_array = any(v["array_len"] is not None and any(v["init"]) for v in sv)
        fixed = ["const"] + (["add", "sub", "mul"] if self.arithmetic_only else BIN_OPS)
        n = len(fixed) + (1 if has_field else 0) + (1 if has_array else 0)
        i = self.rng.rand(0, n - 1)
        if i < len(fixed):
            f = fixed[i]
            if f == "const":
                return ("const", str(self.rng.next()))  # :568-570
            if f == "bnot":
                e = self.generate_expr(depth + 1)
                return None if e is None else ("bnot", e)
            l = self.generate_expr(depth + 1)
            r = self.generate_expr(depth + 1)
            if l is None or r is None:
                return None
            return (f, l, r)
        if has_field and i == len(fixed):
            return self.gen_var_ref()
        return self.gen_array_index()

    def gen_var_ref(self):  # :545-566
        vs = [v["id"] for v in self.scope_vars() if all(v["init"]) and v["array_len"] is None]
        vid = self.rng.rand_vec(vs)
        return None if vid is None else ("var", vid, None)

    def gen_array_index(self):  # :572-601
        vs = []
        for v in self.scope_vars():
            if v["array_len"] is None or not any(v["init"]):
                continue
            written = [i for i, b in enumerate(v["init"]) if b]
            idx = self.rng.rand(0, len(written))  # a position in 0..=len (:584)
            if self.safe_array_index:
                idx = written[idx % len(written)]
            vs.append((v["id"], idx))
        pick = self.rng.rand_vec(vs)
        return None if pick is None else ("var", pick[0], pick[1])

    # ---- extensions, mirrored from the product so the same seed gives the same circuit ----
    def gen_linear(self, factor=False):
        sigs = [v["id"] for v in self.scope_vars()
                if v["role"] in ("input", "output") and v["array_len"] is None and all(v["init"])]

        def term():
            if sigs and ((factor and self.signal_factors) or self.rng.rand(0, 3) != 0):
                k = self.rng.rand(0, len(sigs) - 1)
                keep = self.forced_inputs + self.signal_window
                if self.signal_window and len(sigs) > keep and self.forced_inputs <= k < len(sigs) - self.signal_window:
                    k = k % keep if k % keep < self.forced_inputs else len(sigs) - keep + k % keep
                return ("var", sigs[k], None)
            return ("const", str(self.rng.next()))

        t = term()
        if self.max_linear_terms == 1:
            return t
        shape = self.rng.rand(0, 2)
        if shape == 0:
            return t
        return ("add" if shape == 1 else "sub", t, term())

    def gen_constraint_quadratic(self):
        a, b, c = self.gen_linear(True), self.gen_linear(True), self.gen_linear()
        decl = self.gen_var_decl(("field", None), "output", None)
        self.pending.append(decl)
        self.mark(decl[1], None)
        return ("constrain", (decl[1], None), ("add", ("mul", a, b), c))


# ---- src/circom.rs ----------------------------------------------------------
def transpile_circuit(circ):
    by_id = {v["id"]: v for v in circ["vars"]}

    def ex(e):
        if e[0] == "const":
            return e[1]
        if e[0] == "var":
            name = by_id[e[1]]["name"]
            return "(%s[%d])" % (name, e[2]) if e[2] is not None else name  # circom.rs:23-33
        if e[0] == "bnot":
            return "(~%s)" % ex(e[1])
        return "(%s%s%s)" % (ex(e[1]), TOKENS[e[0]], ex(e[2]))

    def block(body, lvl):
        return "\n".join(ins(i, lvl) for i in body)

    def ins(i, lvl):
        ind = "   " * lvl
        k = i[0]
        if k == "assign":
            return "%s%s=%s;" % (ind, by_id[i[1][0]]["name"], ex(i[2]))
        if k == "aassign":
            return "%s%s[%s]=%s;" % (ind, by_id[i[1][0]]["name"], ex(i[2]), ex(i[3]))
        if k == "constrain":
            return "%s%s<==%s;" % (ind, by_id[i[1][0]]["name"], ex(i[2]))
        if k == "decl":
            v = by_id[i[1]]
            prefix = {"local": "var", "input": "signal input", "output": "signal output"}[v["role"]]
            s = "%s%s %s" % (ind, prefix, v["name"])
            if v["array_len"] is not None:
                s += "[%d]" % v["array_len"]
            if i[2] is not None:
                s += " = " + ex(i[2])
            return s + ";"
        if k == "if":
            s = "%sif (%s) {\n%s\n" % (ind, ex(i[1]), block(i[2], lvl + 1))
            if i[3] is not None:
                s += "%s} else {\n%s\n" % (ind, block(i[3], lvl + 1))
            return s + ind + "}"
        if k == "for":
            init = ins(i[1], lvl).strip()
            step = ins(i[3], lvl).replace(";", "").strip()
            return "%sfor (%s %s; %s) {\n%s\n%s}" % (ind, init, ex(i[2]), step, block(i[4], lvl + 1), ind)
        if k == "while":
            return "%swhile (%s) {\n%s\n%s}" % (ind, ex(i[1]), block(i[2], lvl + 1), ind)
        raise ValueError(k)

    body = "".join(ins(i, 1) + "\n" for i in circ["instrs"])
    return "\ntemplate Program() {\n" + body + "}\n\ncomponent main = Program();\n        "  # circom.rs:421-428


# ---- the s-expression text form used at the C ABI ----------------------------
def parse_text(text):
    toks = re.findall(r'\(|\)|"(?:[^"\\]|\\.)*"|[^\s()]+', text)
    pos = [0]

    def rd():
        t = toks[pos[0]]
        pos[0] += 1
        if t == "(":
            out = []
            while toks[pos[0]] != ")":
                out.append(rd())
            pos[0] += 1
            return out
        if t.startswith('"'):
            return ("str", re.sub(r"\\(.)", r"\1", t[1:-1]))
        return t

    root = rd()
    assert root[0] == "circuit"

    def ref(n):
        return (int(n[0]), int(n[1])) if isinstance(n, list) else (int(n), None)

    def ex(n):
        h = n[0]
        if h == "const":
            return ("const", n[1][1])
        if h == "var":
            r = ref(n[1])
            return ("var", r[0], r[1])
        if h == "bnot":
            return ("bnot", ex(n[1]))
        return (h, ex(n[1]), ex(n[2]))

    def ins(n):
        h = n[0]
        if h == "decl":
            return ("decl", int(n[1]), ex(n[2]) if len(n) > 2 else None)
        if h == "assign":
            return ("assign", ref(n[1]), ex(n[2]))
        if h == "aassign":
            return ("aassign", ref(n[1]), ex(n[2]), ex(n[3]))
        if h == "constrain":
            return ("constrain", ref(n[1]), ex(n[2]))
        if h == "if":
            return ("if", ex(n[1]), [ins(x) for x in n[2][1:]], [ins(x) for x in n[3][1:]] if len(n) > 3 else None)
        if h == "for":
            return ("for", ins(n[1]), ex(n[2]), ins(n[3]), [ins(x) for x in n[4][1:]])
        if h == "while":
            return ("while", ex(n[1]), [ins(x) for x in n[2][1:]])
        raise ValueError(h)

    vs = []
    for n in root[1][1:]:
        vs.append(dict(id=int(n[1]), name=n[2][1], array_len=int(n[3][1]) if isinstance(n[3], list) else None,
                       role=n[4], init=[x == "1" for x in n[5][1:]]))
    return dict(vars=vs, instrs=[ins(x) for x in root[2][1:]])


def print_text(circ):
    def q(s):
        return '"' + s.replace("\\", "\\\\").replace('"', '\\"') + '"'

    def ref(r):
        return "(%d %d)" % r if r[1] is not None else str(r[0])

    def ex(e):
        if e[0] == "const":
            return "(const %s)" % q(e[1])
        if e[0] == "var":
            return "(var %s)" % ref((e[1], e[2]))
        return "(" + " ".join([e[0]] + [ex(x) for x in e[1:]]) + ")"

    def blk(name, body):
        return "(" + " ".join([name] + [ins(i) for i in body]) + ")"

    def ins(i):
        k = i[0]
        if k == "decl":
            return "(decl %d%s)" % (i[1], " " + ex(i[2]) if i[2] is not None else "")
        if k == "assign":
            return "(assign %s %s)" % (ref(i[1]), ex(i[2]))
        if k == "aassign":
            return "(aassign %s %s %s)" % (ref(i[1]), ex(i[2]), ex(i[3]))
        if k == "constrain":
            return "(constrain %s %s)" % (ref(i[1]), ex(i[2]))
        if k == "if":
            return "(if %s %s%s)" % (ex(i[1]), blk("then", i[2]), " " + blk("else", i[3]) if i[3] is not None else "")
        if k == "for":
            return "(for %s %s %s %s)" % (ins(i[1]), ex(i[2]), ins(i[3]), blk("body", i[4]))
        if k == "while":
            return "(while %s %s)" % (ex(i[1]), blk("body", i[2]))
        raise ValueError(k)

    o = "(circuit\n (vars"
    for v in circ["vars"]:
        t = "(array %d)" % v["array_len"] if v["array_len"] is not None else "field"
        o += "\n  (var %d %s %s %s (init%s))" % (v["id"], q(v["name"]), t, v["role"],
                                                "".join(" 1" if b else " 0" for b in v["init"]))
    o += ")\n (instrs"
    for i in circ["instrs"]:
        o += "\n  " + ins(i)
    return o + "))\n"


# ---- meaning of a Circuit -----------------------------------------------------
class Rejected(Exception):
    """The circuit is one circom would refuse; .cls is the class name."""

    def __init__(self, cls, msg=""):
        Exception.__init__(self, cls + ": " + msg)
        self.cls = cls


KNOWN, LINEAR, QUADRATIC, NONQUAD = 0, 1, 2, 3


class Interpreter:
    """Runs a Circuit for ONE input vector with big integers.

    A value is (int, degree): degree tracks how the value depends on the signals, the
    way circom's compiler does, because `<==` is only legal on A*B + C with A, B, C of
    degree <= 1, and conditions and indices must not depend on signals at all.
    Statement meanings are those of the Circom text src/circom.rs:161-410 prints.
    """

    def __init__(self, circ, inputs, bnot_variant=0, max_iterations=1 << 20):
        self.circ = circ
        self.vars = {}
        self.n_in = self.n_out = 0
        self.in_map, self.out_map = {}, {}
        for v in circ["vars"]:
            if v["id"] in self.vars:
                raise Rejected("BadAst", "duplicate id")
            n = v["array_len"] if v["array_len"] is not None else 1
            self.vars[v["id"]] = dict(v, cells=None, declared=0)
            if v["role"] == "input":
                self.in_map[v["id"]] = self.n_in
                self.n_in += n
            elif v["role"] == "output":
                self.out_map[v["id"]] = self.n_out
                self.n_out += n
        assert len(inputs) == self.n_in, (len(inputs), self.n_in)
        self.inputs = inputs
        self.outputs = [0] * self.n_out
        self.constraints = []  # (A, B, C) values: A*B == C must hold
        self.faults = 0
        self.iterations, self.max_iterations = 0, max_iterations
        self.bnot = F.bnot256 if bnot_variant else F.bnot

    def run(self):
        self.block(self.circ["instrs"])
        return self

    def unsat(self):
        return [k for k, (a, b, c) in enumerate(self.constraints) if F.mul(a, b) != c]

    # expressions: returns (value, degree, quad, from_signal) with quad = (A, B, [(coef, value)]) for degree 2;
    # from_signal: the value derives from a signal, so circom does not know it at compile time even where this
    # interpreter does (a signal constrained to a constant): it may not decide a condition or an index
    def ev(self, e):
        k = e[0]
        if k == "const":
            if not re.fullmatch(r"[0-9]+", e[1]):
                raise Rejected("BadAst", "constant")
            return (int(e[1]) % F.P, KNOWN, None, False)
        if k == "var":
            v = self.vars.get(e[1])
            if v is None or not v["declared"]:
                raise Rejected("BadAst", "unknown or undeclared variable")
            i = 0
            if v["array_len"] is not None:
                if e[2] is None:
                    raise Rejected("BadAst", "array without index")
                if e[2] >= v["array_len"]:
                    raise Rejected("OOBIndex", v["name"])
                i = e[2]
            elif e[2] is not None:
                raise Rejected("BadAst", "index on scalar")
            c = v["cells"][i]
            if c is None:
                raise Rejected("UninitSignal", v["name"])
            return c
        a = self.ev(e[1])
        b = self.ev(e[2]) if k != "bnot" else (0, KNOWN, None, False)
        both_known = a[1] == KNOWN and b[1] == KNOWN
        sig = a[3] or b[3]
        if k in ("div", "idiv", "rem") and b[1] == KNOWN and b[0] == 0:
            raise Rejected("DivByZero")
        try:
            val = self.bnot(a[0]) if k == "bnot" else F.BINARY[OPCODE[k]](a[0], b[0])
        except F.Fault:
            val = 0
            self.faults = 1
        if both_known:
            return (val, KNOWN, None, sig)
        da, db = a[1], b[1]
        deg, quad = NONQUAD, None
        if k in ("add", "sub"):
            if da <= LINEAR and db <= LINEAR:
                deg = LINEAR
            elif (da == QUADRATIC and db <= LINEAR) or (da <= LINEAR and db == QUADRATIC):
                deg = QUADRATIC
                q, l, q_is_b = (a, b, False) if da == QUADRATIC else (b, a, True)
                s, A, B, terms = q[2]
                lt = [(1, l[0])]
                if k == "sub":
                    if q_is_b:
                        s, terms = F.neg(s), [(F.neg(c), x) for c, x in terms]
                    else:
                        lt = [(F.neg(1), l[0])]
                quad = (s, A, B, terms + lt)
        elif k == "mul":
            if da == KNOWN or db == KNOWN:
                kn, ot = (a, b) if da == KNOWN else (b, a)
                if ot[1] <= LINEAR:
                    deg = LINEAR
                elif ot[1] == QUADRATIC:
                    deg = QUADRATIC
                    s, A, B, terms = ot[2]
                    quad = (F.mul(s, kn[0]), A, B, [(F.mul(c, kn[0]), x) for c, x in terms])
            elif da == LINEAR and db == LINEAR:
                deg, quad = QUADRATIC, (1, a[0], b[0], [])
        elif k == "div" and db == KNOWN:
            inv = pow(b[0], -1, F.P)
            if da <= LINEAR:
                deg = LINEAR
            elif da == QUADRATIC:
                deg = QUADRATIC
                s, A, B, terms = a[2]
                quad = (F.mul(s, inv), A, B, [(F.mul(c, inv), x) for c, x in terms])
        return (val, deg, quad, sig)

    def truth(self, e):
        v = self.ev(e)
        if v[1] != KNOWN or v[3]:
            raise Rejected("UnknownControl")
        return v[0] != 0

    def tick(self):
        self.iterations += 1
        if self.iterations > self.max_iterations:
            raise Rejected("LoopCap")

    def block(self, body):
        for i in body:
            self.instr(i)

    def instr(self, i):
        k = i[0]
        if k == "decl":
            v = self.vars.get(i[1])
            if v is None:
                raise Rejected("BadAst", "decl of unknown id")
            if v["role"] != "local" and v["declared"]:
                raise Rejected("DoubleAssign", "signal declared again by a loop")
            v["declared"] += 1
            n = v["array_len"] if v["array_len"] is not None else 1
            if v["role"] == "local":
                v["cells"] = [(0, KNOWN, None, False)] * n  # circom variables start at 0
            elif v["role"] == "input":
                base = self.in_map[v["id"]]
                v["cells"] = [(self.inputs[base + j], LINEAR, None, True) for j in range(n)]
            else:
                v["cells"] = [None] * n
            if i[2] is not None:
                if v["role"] != "local" or v["array_len"] is not None:
                    raise Rejected("BadAst", "default on non-local or array")
                v["cells"][0] = self.ev(i[2])
        elif k == "assign":
            v = self.vars.get(i[1][0])
            if v is None or v["role"] != "local" or v["array_len"] is not None or not v["declared"]:
                raise Rejected("BadAst", "assign target")
            v["cells"][0] = self.ev(i[2])
        elif k == "aassign":
            v = self.vars.get(i[1][0])
            if v is None or v["role"] != "local" or v["array_len"] is None or not v["declared"]:
                raise Rejected("BadAst", "array assign target")
            idx = self.ev(i[2])
            if idx[1] != KNOWN or idx[3]:
                raise Rejected("UnknownControl", "index")
            if idx[0] >= v["array_len"]:
                raise Rejected("OOBIndex", v["name"])
            v["cells"][idx[0]] = self.ev(i[3])
        elif k == "constrain":
            v = self.vars.get(i[1][0])
            if v is None or v["role"] != "output" or not v["declared"]:
                raise Rejected("BadAst", "constraint target")
            j = 0
            if v["array_len"] is not None:
                if i[1][1] is None:
                    raise Rejected("BadAst", "array signal without index")
                if i[1][1] >= v["array_len"]:
                    raise Rejected("OOBIndex", v["name"])
                j = i[1][1]
            if v["cells"][j] is not None:
                raise Rejected("DoubleAssign", v["name"])
            val = self.ev(i[2])
            if val[1] == NONQUAD:
                raise Rejected("NonQuadratic", v["name"])
            self.outputs[self.out_map[v["id"]] + j] = val[0]
            if val[1] == QUADRATIC:
                s, A, B, terms = val[2]
                lin = 0
                for c, x in terms:
                    lin = F.add(lin, F.mul(c, x))
                self.constraints.append((F.mul(s, A), B, F.sub(val[0], lin)))
            else:
                self.constraints.append((val[0], 1, val[0]))
            v["cells"][j] = (val[0], KNOWN if val[1] == KNOWN else LINEAR, None, True)
        elif k == "if":
            if self.truth(i[1]):
                self.block(i[2])
            elif i[3] is not None:
                self.block(i[3])
        elif k == "for":
            self.instr(i[1])
            while self.truth(i[2]):
                self.tick()
                self.block(i[4])
                self.instr(i[3])
        elif k == "while":
            while self.truth(i[1]):
                self.tick()
                self.block(i[2])
        else:
            raise Rejected("BadAst", k)


OPCODE = {"add": F.OP_ADD, "sub": F.OP_SUB, "mul": F.OP_MUL, "pow": F.OP_POW, "div": F.OP_DIV, "idiv": F.OP_IDIV,
          "rem": F.OP_REM, "lt": F.OP_LT, "le": F.OP_LE, "gt": F.OP_GT, "ge": F.OP_GE, "eq": F.OP_EQ,
          "and": F.OP_AND, "or": F.OP_OR, "band": F.OP_BAND, "bor": F.OP_BOR, "bxor": F.OP_BXOR, "shr": F.OP_SHR,
          "shl": F.OP_SHL}
