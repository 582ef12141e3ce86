"""A small interpreter for the subset of WGSL the reference's fragment shader is written in.

TEST INFRASTRUCTURE ONLY.  tests/golden/make_wgsl_golden.py feeds it the TEXT of the reference's
src/shaders/shader.wgsl (read from /root/reference where that exists; the text is never stored in this
repository) and records what the shader's own functions return for given inputs; those numbers are the
golden vectors of tests/test_wgsl_golden.py.  That pins the CPU oracle (and through it the HIP kernels)
by executing the reference's source, statement by statement, instead of by somebody's reading of it.

What is interpreted: module-scope `const`, `struct`, storage `var` and `fn` declarations; `let` / `var`,
assignment and compound assignment, `++` / `--`, `if` / `else`, `for`, `while`, `break`, `continue`,
`return`; f32 / u32 / i32 / bool scalars, AbstractInt / AbstractFloat literals and constant expressions
(evaluated in i64 / f64 and converted when they meet a concrete type, as the WGSL specification says),
vecN<f32>, fixed-size arrays, structs with value semantics.

Arithmetic: every f32 operation is one IEEE binary32 operation (numpy float32 scalars: round to nearest
even, no contraction), u32 / i32 wrap.  Built-in functions whose result WGSL leaves to the implementation
are defined by a `Builtins` object; the default one states the same choices as the header of
oracle/pt_oracle.h (dot summed left to right, normalize = v * (1 / sqrt(dot)), reflect, minNum / maxNum,
smoothstep, truncating conversions) and takes log / cos / pow from a math provider.  sin, cos and tan of
the camera angles (`rotate_vector`, `vs_main`) are evaluated in double and rounded once, the rule this
build fixes for them.
"""
import math
import re

import numpy as np

F32 = np.float32


class AInt(int):
    """AbstractInt"""


class AFloat(float):
    """AbstractFloat"""


class U32(int):
    pass


class I32(int):
    pass


class Vec:
    __slots__ = ("c",)

    def __init__(self, comps):
        self.c = tuple(comps)


class Arr:
    __slots__ = ("v",)

    def __init__(self, items):
        self.v = list(items)


class Struct:
    __slots__ = ("name", "f")

    def __init__(self, name, fields):
        self.name, self.f = name, fields


class NpArr:
    """A storage buffer backed by a numpy float32 array; `vec_last`: the last axis is a vecN<f32>."""
    __slots__ = ("a", "vec_last")

    def __init__(self, a, vec_last=False):
        self.a, self.vec_last = a, vec_last

    def index(self, i):
        if self.a.ndim == 1:
            return F32(self.a[i])
        if self.a.ndim == 2 and self.vec_last:
            return Vec(F32(x) for x in self.a[i])
        return NpArr(self.a[i], self.vec_last)

    def store(self, i, v):
        if self.a.ndim == 2 and self.vec_last:
            self.a[i, :] = [float(x) for x in v.c]
        elif self.a.ndim == 1:
            self.a[i] = float(v)
        else:
            raise TypeError("storing a whole row of a storage array is not supported")


def copyval(v):
    if isinstance(v, Arr):
        return Arr(copyval(x) for x in v.v)
    if isinstance(v, Struct):
        return Struct(v.name, {k: copyval(x) for k, x in v.f.items()})
    if isinstance(v, NpArr):
        if v.a.ndim == 1:
            return Arr(F32(x) for x in v.a)
        return Arr(copyval(v.index(i)) for i in range(v.a.shape[0]))
    return v


# ----------------------------------------------------------------------------- tokens
_TOKEN = re.compile(r"""
    (?P<ws>\s+|//[^\n]*)
  | (?P<num>(?:\d+\.\d*|\.\d+)(?:[eE][+-]?\d+)?[fh]?|\d+[eE][+-]?\d+[fh]?|\d+[fiu]?)
  | (?P<id>[A-Za-z_][A-Za-z_0-9]*)
  | (?P<op>->|\+\+|--|&&|\|\||==|!=|<=|>=|<<|>>|\+=|-=|\*=|/=|[-+*/%<>=!&|^(){}\[\],;:.@])
""", re.X)


def tokenize(src):
    out, pos = [], 0
    while pos < len(src):
        m = _TOKEN.match(src, pos)
        if not m:
            raise SyntaxError("cannot tokenize at offset %d: %r" % (pos, src[pos:pos + 20]))
        pos = m.end()
        if m.lastgroup != "ws":
            out.append((m.lastgroup, m.group()))
    out.append(("eof", ""))
    return out


# ----------------------------------------------------------------------------- parser
class Parser:
    VEC = {"vec2": 2, "vec3": 3, "vec4": 4}

    def __init__(self, src):
        self.t = tokenize(src)
        self.i = 0

    def peek(self, k=0):
        return self.t[self.i + k]

    def at(self, text):
        return self.t[self.i][1] == text and self.t[self.i][0] != "eof"

    def eat(self, text=None):
        tok = self.t[self.i]
        if text is not None and tok[1] != text:
            raise SyntaxError("expected %r, found %r (token %d)" % (text, tok[1], self.i))
        self.i += 1
        return tok

    def accept(self, text):
        if self.at(text):
            self.i += 1
            return True
        return False

    def close_angle(self):
        """`>` that closes a template list; splits a `>>` or `>=` token if the lexer glued it."""
        kind, text = self.t[self.i]
        if text == ">":
            self.i += 1
        elif text in (">>", ">="):
            self.t[self.i] = (kind, text[1:])
        else:
            raise SyntaxError("expected '>', found %r" % text)

    def attributes(self):
        while self.accept("@"):
            self.eat()
            if self.accept("("):
                depth = 1
                while depth:
                    tok = self.eat()[1]
                    depth += (tok == "(") - (tok == ")")

    def parse_type(self):
        name = self.eat()[1]
        if name in self.VEC:
            self.eat("<")
            elem = self.parse_type()
            self.close_angle()
            return ("vec", self.VEC[name], elem)
        if name == "array":
            self.eat("<")
            elem = self.parse_type()
            count = None
            if self.accept(","):
                count = self.additive()      # no relational operators inside a template list
            self.close_angle()
            return ("array", elem, count)
        return ("named", name)

    # ---- module
    def module(self):
        decls = []
        while self.peek()[0] != "eof":
            self.attributes()
            if self.accept("struct"):
                name = self.eat()[1]
                self.eat("{")
                fields = []
                while not self.at("}"):
                    self.attributes()
                    fname = self.eat()[1]
                    self.eat(":")
                    fields.append((fname, self.parse_type()))
                    self.accept(",")
                self.eat("}")
                self.accept(";")
                decls.append(("struct", name, fields))
            elif self.accept("const"):
                name = self.eat()[1]
                typ = self.parse_type() if self.accept(":") else None
                self.eat("=")
                decls.append(("const", name, typ, self.expr()))
                self.eat(";")
            elif self.accept("var"):
                if self.accept("<"):
                    while not self.at(">"):
                        self.eat()
                    self.eat(">")
                name = self.eat()[1]
                while not self.at(";"):          # the type of a storage variable is not needed: the host binds the buffer
                    self.eat()
                self.eat(";")
                decls.append(("storage", name))
            elif self.accept("fn"):
                name = self.eat()[1]
                self.eat("(")
                params = []
                while not self.at(")"):
                    self.attributes()
                    pname = self.eat()[1]
                    self.eat(":")
                    params.append((pname, self.parse_type()))
                    self.accept(",")
                self.eat(")")
                ret = None
                if self.accept("->"):
                    self.attributes()
                    ret = self.parse_type()
                decls.append(("fn", name, params, ret, self.block()))
            else:
                raise SyntaxError("unexpected token %r at module scope" % (self.peek()[1],))
        return decls

    # ---- statements
    def block(self):
        self.eat("{")
        body = []
        while not self.at("}"):
            body.append(self.statement())
        self.eat("}")
        return ("block", body)

    def simple_statement(self):
        """declaration, assignment, increment or call, without the closing `;` (for-headers use it too)"""
        if self.at("var") or self.at("let") or self.at("const"):
            kind = self.eat()[1]
            name = self.eat()[1]
            typ = self.parse_type() if self.accept(":") else None
            init = self.expr() if self.accept("=") else None
            return ("decl", kind, name, typ, init)
        lhs = self.unary()
        for op in ("=", "+=", "-=", "*=", "/="):
            if self.accept(op):
                return ("assign", op, lhs, self.expr())
        if self.accept("++"):
            return ("assign", "+=", lhs, ("num", AInt(1)))
        if self.accept("--"):
            return ("assign", "-=", lhs, ("num", AInt(1)))
        return ("expr", lhs)

    def statement(self):
        if self.at("{"):
            return self.block()
        if self.accept("if"):
            cond = self.expr()
            then = self.block()
            other = None
            if self.accept("else"):
                other = self.statement() if self.at("if") else self.block()
            return ("if", cond, then, other)
        if self.accept("for"):
            self.eat("(")
            init = None if self.at(";") else self.simple_statement()
            self.eat(";")
            cond = None if self.at(";") else self.expr()
            self.eat(";")
            step = None if self.at(")") else self.simple_statement()
            self.eat(")")
            return ("for", init, cond, step, self.block())
        if self.accept("while"):
            cond = self.expr()
            return ("while", cond, self.block())
        if self.accept("break"):
            self.eat(";")
            return ("break",)
        if self.accept("continue"):
            self.eat(";")
            return ("continue",)
        if self.accept("return"):
            value = None if self.at(";") else self.expr()
            self.eat(";")
            return ("return", value)
        s = self.simple_statement()
        self.eat(";")
        return s

    # ---- expressions (usual precedence; the shader parenthesises where WGSL demands it)
    def expr(self):
        return self.binary(0)

    LEVELS = [("||",), ("&&",), ("|",), ("^",), ("&",), ("==", "!="), ("<", ">", "<=", ">="), ("<<", ">>"), ("+", "-"), ("*", "/", "%")]

    def binary(self, level):
        if level == len(self.LEVELS):
            return self.unary()
        left = self.binary(level + 1)
        while self.peek()[0] == "op" and self.peek()[1] in self.LEVELS[level]:
            op = self.eat()[1]
            left = ("bin", op, left, self.binary(level + 1))
        return left

    def additive(self):
        return self.binary(8)

    def unary(self):
        if self.accept("-"):
            return ("neg", self.unary())
        if self.accept("!"):
            return ("not", self.unary())
        return self.postfix(self.primary())

    def postfix(self, e):
        while True:
            if self.accept("."):
                e = ("member", e, self.eat()[1])
            elif self.accept("["):
                e = ("index", e, self.expr())
                self.eat("]")
            else:
                return e

    def args(self):
        self.eat("(")
        out = []
        while not self.at(")"):
            out.append(self.expr())
            self.accept(",")
        self.eat(")")
        return out

    def primary(self):
        kind, text = self.peek()
        if kind == "num":
            self.eat()
            return ("num", parse_number(text))
        if text == "(":
            self.eat("(")
            e = self.expr()
            self.eat(")")
            return e
        if kind == "id":
            if text in ("true", "false"):
                self.eat()
                return ("num", text == "true")
            if (text in self.VEC or text == "array") and self.peek(1)[1] == "<":
                typ = self.parse_type()
                return ("construct", typ, self.args())
            self.eat()
            if self.at("("):
                return ("call", text, self.args())
            return ("var", text)
        raise SyntaxError("unexpected token %r in an expression" % (text,))


def parse_number(text):
    if text[-1] == "u":
        return U32(int(text[:-1]))
    if text[-1] == "i":
        return I32(int(text[:-1]))
    if text[-1] in "fh" or "." in text or "e" in text or "E" in text:
        body = text.rstrip("fh")
        return F32(body) if text[-1] == "f" else AFloat(float(body))
    return AInt(int(text))


# ----------------------------------------------------------------------------- arithmetic
def kind_of(v):
    if isinstance(v, (bool, np.bool_)):
        return "bool"
    if isinstance(v, np.float32):
        return "f32"
    if isinstance(v, AFloat):
        return "af"
    if isinstance(v, U32):
        return "u32"
    if isinstance(v, I32):
        return "i32"
    if isinstance(v, AInt):
        return "ai"
    raise TypeError("not a scalar: %r" % (v,))


def to_kind(v, k):
    kv = kind_of(v)
    if kv == k:
        return v
    if k == "f32" and kv in ("af", "ai"):
        return F32(v)
    if k == "af" and kv == "ai":
        return AFloat(float(v))
    if k == "u32" and kv == "ai":
        return U32(int(v) & 0xFFFFFFFF)
    if k == "i32" and kv == "ai":
        return I32(wrap_i32(int(v)))
    raise TypeError("no implicit conversion from %s to %s" % (kv, k))


def unify(a, b):
    ka, kb = kind_of(a), kind_of(b)
    if ka == kb:
        return a, b, ka
    rank = {"ai": 0, "af": 1}
    if ka in rank and kb in rank:
        k = "af"
    elif ka in rank:
        k = kb
    elif kb in rank:
        k = ka
    else:
        raise TypeError("operands of different concrete types: %s, %s" % (ka, kb))
    return to_kind(a, k), to_kind(b, k), k


def wrap_i32(x):
    x &= 0xFFFFFFFF
    return x - (1 << 32) if x & 0x80000000 else x


def int_div(a, b):
    q = abs(a) // abs(b)
    return q if (a >= 0) == (b >= 0) else -q


def scalar_binary(op, a, b):
    a, b, k = unify(a, b)
    if op in ("==", "!=", "<", ">", "<=", ">="):
        r = {"==": a == b, "!=": a != b, "<": a < b, ">": a > b, "<=": a <= b, ">=": a >= b}[op]
        return bool(r)
    if k == "bool":
        if op in ("&&", "&"):
            return bool(a) and bool(b)
        if op in ("||", "|"):
            return bool(a) or bool(b)
        raise TypeError("operator %s on bool" % op)
    if k == "f32":
        with np.errstate(all="ignore"):
            if op == "+":
                return a + b
            if op == "-":
                return a - b
            if op == "*":
                return a * b
            if op == "/":
                return a / b
        raise TypeError("operator %s on f32" % op)
    if k == "af":
        x, y = float(a), float(b)
        if op == "/":
            return AFloat(x / y) if y != 0.0 else AFloat(math.copysign(math.inf, x) if x != 0.0 else math.nan)
        return AFloat({"+": x + y, "-": x - y, "*": x * y}[op])
    x, y = int(a), int(b)
    if op == "+":
        r = x + y
    elif op == "-":
        r = x - y
    elif op == "*":
        r = x * y
    elif op == "/":
        r = int_div(x, y)
    elif op == "%":
        r = x - int_div(x, y) * y
    elif op == "^":
        r = x ^ y
    elif op == "&":
        r = x & y
    elif op == "|":
        r = x | y
    elif op == "<<":
        r = x << (y & 31 if k != "ai" else y)
    elif op == ">>":
        r = x >> (y & 31 if k != "ai" else y)
    else:
        raise TypeError("operator %s on integers" % op)
    if k == "u32":
        return U32(r & 0xFFFFFFFF)
    if k == "i32":
        return I32(wrap_i32(r))
    return AInt(r)


def binary(op, a, b):
    if op in ("<<", ">>") and not isinstance(a, Vec):
        # the shift amount is u32 whatever the shifted type is
        kb = kind_of(b)
        amount = int(b)
        ka = kind_of(a)
        if ka == "u32":
            r = (int(a) << (amount & 31)) if op == "<<" else (int(a) >> (amount & 31))
            return U32(r & 0xFFFFFFFF)
        if ka == "i32":
            r = (int(a) << (amount & 31)) if op == "<<" else (int(a) >> (amount & 31))
            return I32(wrap_i32(r))
        del kb
        return AInt(int(a) << amount if op == "<<" else int(a) >> amount)
    av, bv = isinstance(a, Vec), isinstance(b, Vec)
    if av or bv:
        n = len(a.c) if av else len(b.c)
        ac = a.c if av else (a,) * n
        bc = b.c if bv else (b,) * n
        comps = [scalar_binary(op, x, y) for x, y in zip(ac, bc)]
        return Vec(comps)
    return scalar_binary(op, a, b)


def negate(v):
    if isinstance(v, Vec):
        return Vec(negate(x) for x in v.c)
    k = kind_of(v)
    if k == "f32":
        return F32(-v)
    if k == "af":
        return AFloat(-float(v))
    if k == "ai":
        return AInt(-int(v))
    if k == "i32":
        return I32(wrap_i32(-int(v)))
    raise TypeError("negation of %s" % k)


def concretize(v):
    """what a `let` / `var` without a type makes of an abstract value"""
    if isinstance(v, Vec):
        return Vec(concretize(x) for x in v.c)
    if isinstance(v, (bool, np.bool_, np.float32, U32, I32)):
        return v
    if isinstance(v, AFloat):
        return F32(v)
    if isinstance(v, AInt):
        return I32(wrap_i32(int(v)))
    return v


def f32_to_u32(x):
    x = float(x)
    if x != x or x <= 0.0:
        return U32(0)
    return U32(min(int(x), 0xFFFFFFFF))


def f32_to_i32(x):
    x = float(x)
    if x != x:
        return I32(0)
    return I32(max(-(1 << 31), min(int(x), (1 << 31) - 1)))


# ----------------------------------------------------------------------------- built-in functions
class LibmMath:
    """log / cos / pow of the C library, in binary32"""

    def __init__(self):
        import ctypes
        import ctypes.util
        m = ctypes.CDLL(ctypes.util.find_library("m") or "libm.so.6")
        for name in ("logf", "cosf", "sinf"):
            getattr(m, name).restype = ctypes.c_float
            getattr(m, name).argtypes = [ctypes.c_float]
        m.powf.restype = ctypes.c_float
        m.powf.argtypes = [ctypes.c_float, ctypes.c_float]
        self.m = m

    def log(self, x):
        return F32(self.m.logf(float(x)))

    def cos(self, x):
        return F32(self.m.cosf(float(x)))

    def sin(self, x):
        return F32(self.m.sinf(float(x)))

    def pow(self, x, y):
        return F32(self.m.powf(float(x), float(y)))


class SharedMath:
    """log / cos / pow of path-tracing_amd/csrc/pt_portable_math.h through a C shim (the functions the HIP
    kernels compile in their exact-math mode and the oracle in its ORC_SHARED_MATH build)"""

    def __init__(self, shim_path):
        import ctypes
        m = ctypes.CDLL(shim_path)
        for name in ("pm_shim_logf", "pm_shim_cosf"):
            getattr(m, name).restype = ctypes.c_float
            getattr(m, name).argtypes = [ctypes.c_float]
        m.pm_shim_powf.restype = ctypes.c_float
        m.pm_shim_powf.argtypes = [ctypes.c_float, ctypes.c_float]
        self.m = m

    def log(self, x):
        return F32(self.m.pm_shim_logf(float(x)))

    def cos(self, x):
        return F32(self.m.pm_shim_cosf(float(x)))

    def sin(self, x):
        raise NotImplementedError("sin is only used on the camera angles")

    def pow(self, x, y):
        return F32(self.m.pm_shim_powf(float(x), float(y)))


CAMERA_FUNCTIONS = ("rotate_vector", "vs_main")


def _f(v):
    return to_kind(v, "f32") if not isinstance(v, Vec) else Vec(to_kind(x, "f32") for x in v.c)


class Builtins:
    """The built-in functions.  WGSL leaves the rounding of several of them to the implementation; the default choices
    are the ones the oracle and the kernels pin.  `choices` swaps single ones for another legal evaluation, to measure
    what hangs on them: normalize="div" (v / length(v)), dot="fma" (a chain of fused multiply-adds), minmax="nan"
    (a NaN operand is handed on), camera="f32" (sin / cos / tan of the camera angles in binary32)."""

    def __init__(self, math_provider, choices=None):
        self.mp = math_provider
        self.choices = dict(choices or {})
        unknown = set(self.choices) - {"normalize", "dot", "minmax", "camera"}
        if unknown:
            raise ValueError("unknown built-in choice: %s" % sorted(unknown))
        self._fmaf = None

    def fma(self, a, b, c):
        if self._fmaf is None:
            import ctypes
            import ctypes.util
            m = ctypes.CDLL(ctypes.util.find_library("m") or "libm.so.6")
            m.fmaf.restype = ctypes.c_float
            m.fmaf.argtypes = [ctypes.c_float] * 3
            m.tanf.restype = ctypes.c_float
            m.tanf.argtypes = [ctypes.c_float]
            self._libm = m
            self._fmaf = m.fmaf
        return F32(self._fmaf(float(a), float(b), float(c)))

    def dot(self, a, b):
        acc = a.c[0] * b.c[0]
        for x, y in zip(a.c[1:], b.c[1:]):
            acc = self.fma(x, y, acc) if self.choices.get("dot") == "fma" else acc + x * y
        return acc

    def minmax(self, name, x, y):
        if self.choices.get("minmax") == "nan" and (x != x or y != y):
            return F32(np.nan)
        return F32((np.fmin if name == "min" else np.fmax)(x, y))        # IEEE minNum / maxNum

    def call(self, name, args, where):
        with np.errstate(all="ignore"):
            return self._call(name, args, where)

    def _call(self, name, args, where):
        if name == "dot":
            return self.dot(_f(args[0]), _f(args[1]))
        if name == "cross":
            a, b = _f(args[0]).c, _f(args[1]).c
            return Vec((a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]))
        if name == "normalize":
            v = _f(args[0])
            if self.choices.get("normalize") == "div":
                length = F32(np.sqrt(self.dot(v, v)))
                return Vec(x / length for x in v.c)
            inv = F32(1.0) / np.sqrt(self.dot(v, v))
            return Vec(x * inv for x in v.c)
        if name == "reflect":
            i, n = _f(args[0]), _f(args[1])
            k = F32(2.0) * self.dot(n, i)
            return Vec(x - y * k for x, y in zip(i.c, n.c))
        if name in ("min", "max"):
            a, b = _f(args[0]), _f(args[1])
            if isinstance(a, Vec):
                return Vec(self.minmax(name, x, y) for x, y in zip(a.c, b.c))
            return self.minmax(name, a, b)
        if name == "smoothstep":
            e0, e1, x = (_f(v) for v in args)
            t = (x - e0) / (e1 - e0)
            t = self.minmax("min", self.minmax("max", t, F32(0.0)), F32(1.0))
            return t * t * (F32(3.0) - F32(2.0) * t)
        if name == "sqrt":
            return F32(np.sqrt(_f(args[0])))
        if name == "sign":
            x = _f(args[0])
            return F32(1.0) if x > 0 else (F32(-1.0) if x < 0 else F32(0.0))
        if name == "abs":
            return F32(abs(_f(args[0])))
        if name in ("cos", "sin", "tan"):
            x = _f(args[0])
            if where in CAMERA_FUNCTIONS:            # camera angles: evaluated in double, rounded once
                if self.choices.get("camera") == "f32":
                    if name == "tan":
                        self.fma(F32(0.0), F32(0.0), F32(0.0))        # loads libm
                        return F32(self._libm.tanf(float(x)))
                    return getattr(LibmMath(), name)(x)
                return F32(getattr(math, name)(float(x)))
            if name == "tan":
                raise NotImplementedError("tan outside the camera set-up")
            return getattr(self.mp, name)(x)
        if name == "log":
            return self.mp.log(_f(args[0]))
        if name == "pow":
            return self.mp.pow(_f(args[0]), _f(args[1]))
        raise NameError("built-in function %s is not implemented" % name)


# ----------------------------------------------------------------------------- evaluation
class _Break(Exception):
    pass


class _Continue(Exception):
    pass


class _Return(Exception):
    def __init__(self, value):
        self.value = value


class Interpreter:
    def __init__(self, source, builtins, overrides=None):
        """overrides: {name of a module-scope const: python value (number, bool, or tuple for a vector)} applied
        instead of the initialiser in the source - what a user of the reference does by editing those lines."""
        self.builtins = builtins
        self.structs, self.functions, self.globals = {}, {}, {}
        self.storage_names = []
        self.scopes = []
        self.where = []
        overrides = dict(overrides or {})
        for d in Parser(source).module():
            if d[0] == "struct":
                self.structs[d[1]] = d[2]
            elif d[0] == "fn":
                self.functions[d[1]] = d
            elif d[0] == "storage":
                self.storage_names.append(d[1])
            elif d[0] == "const":
                _, name, typ, init = d
                if name in overrides:
                    o = overrides.pop(name)
                    if isinstance(o, tuple):
                        value = Vec(AFloat(float(x)) for x in o)
                    elif isinstance(o, bool):
                        value = o
                    elif isinstance(o, int):
                        value = AInt(o)
                    else:
                        value = AFloat(float(o))
                else:
                    value = self.eval(init)
                self.globals[name] = self.convert(typ, value) if typ else value
        if overrides:
            raise KeyError("no such module-scope const: %s" % ", ".join(overrides))

    def bind(self, name, value):
        if name not in self.storage_names:
            raise KeyError("the shader declares no storage variable %s" % name)
        self.globals[name] = value

    # ---- types
    def zero(self, typ):
        if typ[0] == "vec":
            return Vec(self.zero(typ[2]) for _ in range(typ[1]))
        if typ[0] == "array":
            n = int(self.eval(typ[2]))
            return Arr(self.zero(typ[1]) for _ in range(n))
        name = typ[1]
        if name == "f32":
            return F32(0.0)
        if name == "u32":
            return U32(0)
        if name == "i32":
            return I32(0)
        if name == "bool":
            return False
        return Struct(name, {f: self.zero(t) for f, t in self.structs[name]})

    def convert(self, typ, v):
        if typ[0] == "vec":
            return Vec(self.convert(typ[2], x) for x in v.c)
        if typ[0] == "array":
            return v if not isinstance(v, Arr) else Arr(self.convert(typ[1], x) for x in v.v)
        name = typ[1]
        if name in ("f32", "u32", "i32"):
            return to_kind(v, name)
        return v

    # ---- variables
    def lookup(self, name):
        for s in reversed(self.scopes):
            if name in s:
                return s[name]
        if name in self.globals:
            return self.globals[name]
        raise NameError(name)

    def set_var(self, name, value):
        for s in reversed(self.scopes):
            if name in s:
                s[name] = value
                return
        raise NameError("assignment to %s, which is not a local variable" % name)

    # ---- expressions
    def eval(self, e):
        tag = e[0]
        if tag == "num":
            return e[1]
        if tag == "var":
            return self.lookup(e[1])
        if tag == "bin":
            op = e[1]
            if op == "&&":
                return bool(self.eval(e[2])) and bool(self.eval(e[3]))
            if op == "||":
                return bool(self.eval(e[2])) or bool(self.eval(e[3]))
            return binary(op, self.eval(e[2]), self.eval(e[3]))
        if tag == "neg":
            return negate(self.eval(e[1]))
        if tag == "not":
            return not bool(self.eval(e[1]))
        if tag == "member":
            base = self.eval(e[1])
            if isinstance(base, Struct):
                return base.f[e[2]]
            if isinstance(base, Vec):
                if len(e[2]) != 1:
                    return Vec(base.c["xyzw".index(ch)] for ch in e[2])
                return base.c["xyzw".index(e[2])]
            raise TypeError("member access on %r" % (base,))
        if tag == "index":
            base, idx = self.eval(e[1]), int(self.eval(e[2]))
            if isinstance(base, NpArr):
                return base.index(idx)
            if isinstance(base, Arr):
                return base.v[idx]
            if isinstance(base, Vec):
                return base.c[idx]
            raise TypeError("indexing %r" % (base,))
        if tag == "construct":
            typ, args = e[1], [self.eval(a) for a in e[2]]
            if typ[0] == "vec":
                flat = []
                for a in args:
                    flat.extend(a.c if isinstance(a, Vec) else [a])
                if len(flat) == 1:
                    flat = flat * typ[1]
                if len(flat) != typ[1]:
                    raise TypeError("vec%d constructed from %d components" % (typ[1], len(flat)))
                return Vec(self.convert(typ[2], x) for x in flat)
            if not args:
                return self.zero(typ)
            return Arr(copyval(self.convert(typ[1], a)) for a in args)
        if tag == "call":
            return self.call(e[1], [self.eval(a) for a in e[2]])
        raise TypeError("cannot evaluate %r" % (e,))

    def call(self, name, args):
        if name == "f32":
            v = args[0]
            return F32(float(v)) if kind_of(v) == "af" else F32(int(v)) if kind_of(v) in ("u32", "i32", "ai") else F32(v)
        if name == "u32":
            v = args[0]
            k = kind_of(v)
            return f32_to_u32(v) if k in ("f32", "af") else U32(int(v) & 0xFFFFFFFF)
        if name == "i32":
            v = args[0]
            k = kind_of(v)
            return f32_to_i32(v) if k in ("f32", "af") else I32(wrap_i32(int(v)))
        if name in self.functions:
            _, _, params, _, body = self.functions[name]
            frame = {p: copyval(self.convert(t, a)) for (p, t), a in zip(params, args)}
            saved, self.scopes = self.scopes, [frame]
            self.where.append(name)
            try:
                self.exec_block(body)
                result = None
            except _Return as r:
                result = r.value
            finally:
                self.scopes = saved
                self.where.pop()
            return result
        return self.builtins.call(name, args, self.where[-1] if self.where else "")

    # ---- statements
    def exec_block(self, block):
        self.scopes.append({})
        try:
            for s in block[1]:
                self.exec(s)
        finally:
            self.scopes.pop()

    def assign(self, target, value):
        if target[0] == "var":
            self.set_var(target[1], value)
            return
        if target[0] == "member":
            base = self.eval(target[1])
            if not isinstance(base, Struct):
                raise TypeError("assignment to a component of %r" % (base,))
            base.f[target[2]] = value
            return
        if target[0] == "index":
            base, idx = self.eval(target[1]), int(self.eval(target[2]))
            if isinstance(base, NpArr):
                base.store(idx, value)
            elif isinstance(base, Arr):
                base.v[idx] = value
            else:
                raise TypeError("indexed assignment into %r" % (base,))
            return
        raise TypeError("cannot assign to %r" % (target,))

    def exec(self, s):
        tag = s[0]
        if tag == "block":
            self.exec_block(s)
        elif tag == "decl":
            _, _, name, typ, init = s
            if init is None:
                value = self.zero(typ)
            else:
                value = copyval(self.eval(init))
                value = self.convert(typ, value) if typ else concretize(value)
            self.scopes[-1][name] = value
        elif tag == "assign":
            _, op, target, rhs = s
            value = self.eval(rhs)
            if op != "=":
                value = binary(op[0], self.eval(target), value)
            else:
                cur = self.eval(target)          # an abstract right-hand side takes the type of the target
                if not isinstance(cur, (Vec, Arr, Struct, NpArr)) and not isinstance(value, (Vec, Arr, Struct, NpArr)):
                    if kind_of(value) in ("ai", "af"):
                        value = to_kind(value, kind_of(cur))
            self.assign(target, copyval(value))
        elif tag == "expr":
            self.eval(s[1])
        elif tag == "if":
            if bool(self.eval(s[1])):
                self.exec_block(s[2])
            elif s[3] is not None:
                self.exec(s[3])
        elif tag == "for":
            _, init, cond, step, body = s
            self.scopes.append({})
            try:
                if init is not None:
                    self.exec(init)
                while cond is None or bool(self.eval(cond)):
                    try:
                        self.exec_block(body)
                    except _Break:
                        break
                    except _Continue:
                        pass
                    if step is not None:
                        self.exec(step)
            finally:
                self.scopes.pop()
        elif tag == "while":
            while bool(self.eval(s[1])):
                try:
                    self.exec_block(s[2])
                except _Break:
                    break
                except _Continue:
                    pass
        elif tag == "break":
            raise _Break()
        elif tag == "continue":
            raise _Continue()
        elif tag == "return":
            raise _Return(copyval(self.eval(s[1])) if s[1] is not None else None)
        else:
            raise TypeError("cannot execute %r" % (s,))

    # ---- helpers for the host side of a test
    def struct(self, name, **fields):
        s = self.zero(("named", name))
        for k, v in fields.items():
            if k not in s.f:
                raise KeyError(k)
            s.f[k] = v
        return s


def vec3(x, y, z):
    return Vec((F32(x), F32(y), F32(z)))
