"""A small front end for the GLSL ES 3.00 subset the reference's two shaders use, and an f32 runtime for it.

Used ONLY at fixture-generation time (tests/golden/make_golden_shader.py): the reference's shader text is read where it
lies, translated to Python statement by statement, and executed on sample inputs; inputs and outputs go into
tests/golden/shader_golden.json.  No line of the shader is stored anywhere in this repository.

Semantics chosen where GLSL leaves them to the implementation (the same choices oracle/oracle.c states):
every float operation is one IEEE binary32 operation (numpy.float32), no contraction, operands evaluated left to right;
matrix * vector and matrix * matrix accumulate over the inner index in ascending order starting from its first product;
dot(a, b) = a.x*b.x + a.y*b.y (+ ...) left to right; normalize(v) = v / sqrt(dot(v, v)); min(x, y) = y < x ? y : x;
max(x, y) = x < y ? y : x; clamp(x, a, b) = min(max(x, a), b); inverse(mat4) is computed in f64 and rounded.
"""
import re
import struct

import numpy as np

F = np.float32


# ---------------------------------------------------------------------------------------------------------------------
# runtime
# ---------------------------------------------------------------------------------------------------------------------
class Discard(Exception):
    pass


_SW = {"x": 0, "y": 1, "z": 2, "w": 3, "r": 0, "g": 1, "b": 2, "a": 3}


def _flat(args):
    out = []
    for a in args:
        if isinstance(a, (Vec, UVec)):
            out.extend(a.c)
        elif isinstance(a, Mat):
            for col in a.cols:
                out.extend(col.c)
        else:
            out.append(a)
    return out


class _VecBase:
    __array_ufunc__ = None      # numpy scalars must not try to broadcast over these

    def __len__(self):
        return len(self.c)

    def __getitem__(self, i):
        return self.c[int(i)]

    def __setitem__(self, i, v):
        self.c[int(i)] = self._conv(v)

    def __getattr__(self, name):
        if name and all(ch in _SW for ch in name):
            if len(name) == 1:
                return self.c[_SW[name]]
            return type(self)([self.c[_SW[ch]] for ch in name])
        raise AttributeError(name)

    def __setattr__(self, name, value):
        if name != "c" and all(ch in _SW for ch in name):
            if len(name) == 1:
                self.c[_SW[name]] = self._conv(value)
            else:
                for ch, v in zip(name, value.c):
                    self.c[_SW[ch]] = self._conv(v)
        else:
            object.__setattr__(self, name, value)


class Vec(_VecBase):
    def __init__(self, comps):
        self.c = [F(x) for x in comps]

    @staticmethod
    def _conv(v):
        return F(v)

    def _zip(self, o, f):
        if isinstance(o, Vec):
            assert len(o.c) == len(self.c)
            return Vec([f(a, b) for a, b in zip(self.c, o.c)])
        return Vec([f(a, F(o)) for a in self.c])

    def _rzip(self, o, f):
        return Vec([f(F(o), a) for a in self.c])

    def __add__(self, o): return self._zip(o, lambda a, b: a + b)
    def __sub__(self, o): return self._zip(o, lambda a, b: a - b)
    def __mul__(self, o):
        if isinstance(o, Mat):
            return NotImplemented
        return self._zip(o, lambda a, b: a * b)
    def __truediv__(self, o): return self._zip(o, lambda a, b: a / b)
    def __radd__(self, o): return self._rzip(o, lambda a, b: a + b)
    def __rsub__(self, o): return self._rzip(o, lambda a, b: a - b)
    def __rmul__(self, o): return self._rzip(o, lambda a, b: a * b)
    def __rtruediv__(self, o): return self._rzip(o, lambda a, b: a / b)
    def __neg__(self): return Vec([-a for a in self.c])


class UVec(_VecBase):
    def __init__(self, comps):
        self.c = [int(x) & 0xffffffff for x in comps]

    @staticmethod
    def _conv(v):
        return int(v) & 0xffffffff


class IVec(_VecBase):
    def __init__(self, comps):
        self.c = [int(x) for x in comps]

    @staticmethod
    def _conv(v):
        return int(v)


class Mat:
    """n x n, column major: m[c] is column c (a Vec, so m[c][r] reads and writes)."""
    __array_ufunc__ = None

    def __init__(self, cols):
        self.cols = cols
        self.n = len(cols)

    def __getitem__(self, c):
        return self.cols[int(c)]

    def __mul__(self, o):
        n = self.n
        if isinstance(o, Vec):
            out = []
            for r in range(n):
                s = self.cols[0].c[r] * o.c[0]
                for k in range(1, n):
                    s = s + self.cols[k].c[r] * o.c[k]
                out.append(s)
            return Vec(out)
        if isinstance(o, Mat):
            cols = []
            for c in range(n):
                col = []
                for r in range(n):
                    s = self.cols[0].c[r] * o.cols[c].c[0]
                    for k in range(1, n):
                        s = s + self.cols[k].c[r] * o.cols[c].c[k]
                    col.append(s)
                cols.append(Vec(col))
            return Mat(cols)
        return Mat([col * o for col in self.cols])


def _vecn(n):
    def ctor(*args):
        f = _flat(args)
        if len(f) == 1:
            f = f * n
        assert len(f) >= n, "vec%d from %d components" % (n, len(f))
        return Vec(f[:n])
    return ctor


def _uvecn(n, cls):
    def ctor(*args):
        f = _flat(args)
        if len(f) == 1:
            f = f * n
        return cls(f[:n])
    return ctor


def _matn(n):
    def ctor(*args):
        if len(args) == 1 and isinstance(args[0], Mat):     # mat3(mat4): the upper-left block
            src = args[0]
            return Mat([Vec(src.cols[c].c[:n]) for c in range(n)])
        f = _flat(args)
        assert len(f) == n * n
        return Mat([Vec(f[c * n:(c + 1) * n]) for c in range(n)])
    return ctor


class Texture:
    """usampler2D over a flat array of 32-bit words, RGBA32UI texels, `width` texels per row."""
    def __init__(self, words, width):
        self.words, self.width = words, width

    def fetch(self, x, y):
        o = (int(y) * self.width + int(x)) * 4
        return UVec([int(w) for w in self.words[o:o + 4]])


def _dot(a, b):
    s = a.c[0] * b.c[0]
    for k in range(1, len(a.c)):
        s = s + a.c[k] * b.c[k]
    return s


def _min(x, y):
    return y if y < x else x


def _max(x, y):
    return y if x < y else x


def _bits_to_float(u):
    conv = lambda w: np.frombuffer(struct.pack("<I", int(w) & 0xffffffff), dtype=np.float32)[0]
    return Vec([conv(w) for w in u.c]) if isinstance(u, UVec) else conv(u)


def _unpack_half(u):
    h = np.frombuffer(struct.pack("<I", int(u) & 0xffffffff), dtype=np.float16)
    return Vec([F(h[0]), F(h[1])])


def _inverse(m):
    a = np.array([[float(m.cols[c].c[r]) for c in range(m.n)] for r in range(m.n)], dtype=np.float64)
    inv = np.linalg.inv(a)
    return Mat([Vec([inv[r][c] for r in range(m.n)]) for c in range(m.n)])


def runtime():
    return {
        "F": F, "Discard": Discard,
        "vec2": _vecn(2), "vec3": _vecn(3), "vec4": _vecn(4),
        "uvec2": _uvecn(2, UVec), "uvec3": _uvecn(3, UVec), "uvec4": _uvecn(4, UVec),
        "ivec2": _uvecn(2, IVec), "ivec3": _uvecn(3, IVec), "ivec4": _uvecn(4, IVec),
        "mat3": _matn(3), "mat4": _matn(4),
        "float": lambda x: F(x), "int": lambda x: int(x), "uint": lambda x: int(x) & 0xffffffff, "bool": bool,
        "texelFetch": lambda tex, p, lod: tex.fetch(p.c[0], p.c[1]),
        "uintBitsToFloat": _bits_to_float, "unpackHalf2x16": _unpack_half,
        "transpose": lambda m: Mat([Vec([m.cols[r].c[c] for r in range(m.n)]) for c in range(m.n)]),
        "inverse": _inverse,
        "dot": _dot, "normalize": lambda v: v / np.sqrt(_dot(v, v)),
        "sqrt": lambda x: np.sqrt(F(x)), "exp": lambda x: np.exp(F(x)), "ceil": lambda x: np.ceil(F(x)),
        "min": _min, "max": _max, "clamp": lambda x, a, b: _min(_max(x, a), b),
    }


# ---------------------------------------------------------------------------------------------------------------------
# front end: GLSL -> Python source
# ---------------------------------------------------------------------------------------------------------------------
_TOKEN = re.compile(r"""
    (?P<ws>\s+|//[^\n]*|/\*.*?\*/)
  | (?P<num>0[xX][0-9a-fA-F]+[uU]?|(?:\d+\.\d*|\.\d+|\d+)(?:[eE][-+]?\d+)?[fFuU]?)
  | (?P<id>[A-Za-z_]\w*)
  | (?P<op>\+\+|--|\+=|-=|\*=|/=|==|!=|<=|>=|&&|\|\||<<|>>|[-+*/%<>=!&|^?:,;()\[\]{}.])
""", re.X | re.S)

_TYPES = {"float", "int", "uint", "bool", "vec2", "vec3", "vec4", "uvec2", "uvec3", "uvec4", "ivec2", "ivec3", "ivec4",
          "mat3", "mat4", "void", "usampler2D", "sampler2D"}
_QUAL = {"const", "in", "out", "inout", "highp", "mediump", "lowp", "uniform", "precision"}
_PYKW = {"in", "is", "def", "lambda", "pass", "from", "import", "global", "with", "as", "del", "and", "or", "not", "None", "class", "try", "end"}


def tokenize(src):
    out, pos = [], 0
    while pos < len(src):
        m = _TOKEN.match(src, pos)
        if not m:
            raise SyntaxError("cannot tokenize at %r" % src[pos:pos + 30])
        pos = m.end()
        if m.lastgroup != "ws":
            out.append((m.lastgroup, m.group()))
    out.append(("eof", ""))
    return out


class Translator:
    def __init__(self, src):
        self.t = tokenize(src)
        self.p = 0
        self.globals_assigned = ["gl_Position", "vColor", "vPosition", "fragColor"]
        self.in_main = False      # main() returns its locals, so that the caller can read the shader's intermediate values

    # -- token helpers
    def peek(self, k=0): return self.t[self.p + k]
    def next(self):
        tok = self.t[self.p]; self.p += 1; return tok
    def accept(self, v):
        if self.peek()[1] == v:
            self.p += 1; return True
        return False
    def expect(self, v):
        tok = self.next()
        if tok[1] != v:
            raise SyntaxError("expected %r, got %r (token %d)" % (v, tok[1], self.p))

    @staticmethod
    def name(n):
        return n + "_" if n in _PYKW else n

    # -- expressions
    def number(self, s):
        if s.lower().startswith("0x"):
            return str(int(s.rstrip("uU"), 16))
        if s[-1] in "uU":
            return s[:-1]
        if s[-1] in "fF" or "." in s or "e" in s.lower():
            v = s.rstrip("fF")
            return "F(%s)" % (v + "0" if v.endswith(".") else v)
        return s

    def primary(self):
        kind, v = self.next()
        if kind == "num":
            return self.number(v)
        if kind == "id":
            if v == "float" and self.peek()[1] == "[":       # float[](a, b, ...)
                self.expect("["); self.expect("]"); self.expect("(")
                items = self.args()
                return "[" + ", ".join(items) + "]"
            return self.name(v)
        if v == "(":
            e = self.expr()
            self.expect(")")
            return "(" + e + ")"
        raise SyntaxError("unexpected %r" % v)

    def args(self):
        items = []
        if not self.accept(")"):
            while True:
                items.append(self.assign())
                if self.accept(")"):
                    break
                self.expect(",")
        return items

    def postfix(self):
        e = self.primary()
        while True:
            v = self.peek()[1]
            if v == "(":
                self.next()
                e = e + "(" + ", ".join(self.args()) + ")"
            elif v == "[":
                self.next()
                i = self.expr()
                self.expect("]")
                e = e + "[" + i + "]"
            elif v == ".":
                self.next()
                e = e + "." + self.next()[1]
            elif v in ("++", "--"):
                self.next()
                e = ("__post_inc__" if v == "++" else "__post_dec__") + " " + e
            else:
                return e

    def unary(self):
        v = self.peek()[1]
        if v == "-":
            self.next(); return "(-" + self.unary() + ")"
        if v == "+":
            self.next(); return self.unary()
        if v == "!":
            self.next(); return "(not " + self.unary() + ")"
        return self.postfix()

    def binary(self, level=0):
        levels = [["||"], ["&&"], ["|"], ["^"], ["&"], ["==", "!="], ["<", ">", "<=", ">="], ["<<", ">>"], ["+", "-"], ["*", "/", "%"]]
        if level == len(levels):
            return self.unary()
        e = self.binary(level + 1)
        while self.peek()[1] in levels[level]:
            op = self.next()[1]
            r = self.binary(level + 1)
            e = "(" + e + " " + {"||": "or", "&&": "and"}.get(op, op) + " " + r + ")"
        return e

    def ternary(self):
        c = self.binary()
        if self.accept("?"):
            a = self.assign()
            self.expect(":")
            b = self.assign()
            return "(" + a + " if " + c + " else " + b + ")"
        return c

    def assign(self):
        lhs = self.ternary()
        v = self.peek()[1]
        if v in ("=", "+=", "-=", "*=", "/="):
            self.next()
            return lhs + " " + v + " " + self.assign()
        return lhs

    def expr(self):
        return self.assign()

    # -- statements
    def skip_qualifiers(self):
        while self.peek()[1] in _QUAL:
            self.next()

    def declaration(self, out, ind):
        """type declarator (= init)? (, declarator (= init)?)* ;"""
        ty = self.next()[1]
        while True:
            nm = self.name(self.next()[1])
            size = None
            if self.accept("["):
                size = self.expr()
                self.expect("]")
            if self.accept("="):
                out.append(ind + nm + " = " + self.assign())
            elif size is not None:
                out.append(ind + nm + " = [F(0)] * " + size)
            else:
                out.append(ind + nm + " = None")
            if self.accept(";"):
                return
            self.expect(",")

    def statement(self, out, ind):
        self.skip_qualifiers()
        kind, v = self.peek()
        if v == "{":
            self.next()
            n0 = len(out)
            while not self.accept("}"):
                self.statement(out, ind)
            if len(out) == n0:
                out.append(ind + "pass")
            return
        if v == "if":
            self.next(); self.expect("(")
            c = self.expr()
            self.expect(")")
            out.append(ind + "if " + c + ":")
            self.block(out, ind)
            if self.accept("else"):
                out.append(ind + "else:")
                self.block(out, ind)
            return
        if v == "for":
            self.next(); self.expect("(")
            self.statement(out, ind)                 # init (a declaration or an expression statement, ends with ';')
            c = self.expr(); self.expect(";")
            inc = self.expr(); self.expect(")")
            out.append(ind + "while " + c + ":")
            body = []
            self.block(body, ind)
            out.extend(body)
            out.append(ind + "    " + self.fix_incdec(inc))
            return
        if v == "return":
            self.next()
            if self.accept(";"):
                out.append(ind + ("return locals()" if self.in_main else "return"))
            else:
                out.append(ind + "return " + self.expr())
                self.expect(";")
            return
        if v == "discard":
            self.next(); self.expect(";")
            out.append(ind + "raise Discard()")
            return
        if kind == "id" and v in _TYPES and self.peek(1)[0] == "id":
            self.declaration(out, ind)
            return
        e = self.expr()
        self.expect(";")
        out.append(ind + self.fix_incdec(e))

    @staticmethod
    def fix_incdec(e):
        if e.startswith("__post_inc__ "):
            return e[len("__post_inc__ "):] + " += 1"
        if e.startswith("__post_dec__ "):
            return e[len("__post_dec__ "):] + " -= 1"
        return e

    def block(self, out, ind):
        n0 = len(out)
        self.statement(out, ind + "    ")
        if len(out) == n0:
            out.append(ind + "    pass")

    # -- top level
    def translate(self):
        out = []
        while self.peek()[0] != "eof":
            kind, v = self.peek()
            if v == "#":                                   # (never reached: '#' lines are removed before tokenizing)
                raise SyntaxError("preprocessor line")
            if v == "precision":
                while self.next()[1] != ";":
                    pass
                continue
            if v in ("uniform", "in", "out"):              # interface: provided by the caller as module globals
                while self.next()[1] != ";":
                    pass
                continue
            self.skip_qualifiers()
            # function definition or global constant
            if self.peek()[0] == "id" and self.peek()[1] in _TYPES and self.peek(1)[0] == "id" and self.peek(2)[1] == "(":
                self.next()
                fname = self.name(self.next()[1])
                self.expect("(")
                params = []
                if not self.accept(")"):
                    while True:
                        self.skip_qualifiers()
                        self.next()                         # type
                        params.append(self.name(self.next()[1]))
                        if self.accept("["):
                            self.expr(); self.expect("]")
                        if self.accept(")"):
                            break
                        self.expect(",")
                out.append("def %s(%s):" % (fname, ", ".join(params)))
                out.append("    global " + ", ".join(self.globals_assigned))
                body = []
                self.in_main = fname == "main"
                self.statement(body, "    ")
                out.extend(body or ["    pass"])
                if self.in_main:
                    out.append("    return locals()")
                self.in_main = False
                out.append("")
            else:
                self.declaration(out, "")
        return "\n".join(out) + "\n"


def shader_source(ts_text):
    """The GLSL inside the template literal of one of the reference's *.glsl.ts modules, without its '#' lines."""
    body = ts_text[ts_text.index("`") + 1:ts_text.rindex("`")]
    return "\n".join(l for l in body.split("\n") if not l.lstrip().startswith("#"))


def compile_shader(ts_text):
    """-> (namespace, python_source): call namespace['main']() after setting the shader's inputs in the namespace."""
    py = Translator(shader_source(ts_text)).translate()
    ns = runtime()
    ns["true"], ns["false"] = True, False
    for g in ("gl_Position", "vColor", "vPosition", "fragColor"):
        ns[g] = None
    exec(compile(py, "<shader>", "exec"), ns)
    return ns, py
