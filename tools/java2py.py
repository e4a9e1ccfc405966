"""java2py -- a mechanical translator for the subset of Java the reference's operator classes are written in.

TEST INFRASTRUCTURE (fixture generation only; never imported by reflexiv_amd/).  The reference is Java and no JVM exists
in the build container, so its operator classes cannot be run as they are.  Their bodies, however, use only `long` /
`int` arithmetic, arrays, `if` / `for` / `while`, `ArrayList<Row>` and Spark's `Row` / `Seq` accessors.  This module
parses such a class from the reference's source TEXT (read from /root/reference at fixture-generation time -- the
text is never stored in this repository), translates it statement by statement into Python and executes it on
Java's integer semantics:

  * `int` / `long` values are `J` objects (two's complement, 32 or 64 bits wide; binary numeric promotion; shift
    counts masked with 31 / 63; `>>>` logical; `/` and `%` truncate towards zero);
  * `Long.numberOfLeadingZeros`, `Long.SIZE`, `Long.parseLong`, `Integer.parseInt`, `Math.*`;
  * `Row` (`getLong/getInt/getSeq/getString/get`), Scala `Seq` (`apply/length/size`), `RowFactory.create`, the
    `JavaConverters...toSeq()` wrappers (= the sequence of the wrapped array's elements), `ArrayList`, `Iterator`,
    `Tuple2`, `String` methods, `HashMap` -- as small shims (`RT` below).

tests/golden/make_reference_vectors.py uses it to run the reference's own operator classes on seeded inputs and
commits inputs + outputs as fixtures: a pin of the oracle to the reference's code that does not go through anybody's
reading of it (SURVEY.md D.4 did the same by hand).

The translator is deliberately strict: any construct it does not know raises, nothing is guessed.
"""
from __future__ import annotations

import re

# ------------------------------------------------------------------------------------------------ lexer
_TOKEN = re.compile(r"""
    (?P<ws>\s+)
  | (?P<lc>//[^\n]*)
  | (?P<bc>/\*.*?\*/)
  | (?P<fnum>\d+\.\d*(?:[eE][-+]?\d+)?[fFdD]?|\d+[fFdD])
  | (?P<num>0[xX][0-9a-fA-F]+[lL]?|\d+[lL]?)
  | (?P<str>"(?:\\.|[^"\\])*")
  | (?P<chr>'(?:\\.|[^'\\])')
  | (?P<id>[A-Za-z_$][A-Za-z_0-9$]*)
  | (?P<op>>>>=|<<=|>>=|>>>|\+\+|--|&&|\|\||==|!=|<=|>=|\+=|-=|\*=|/=|%=|&=|\|=|\^=|<<|>>|->|[-+*/%&|^~!<>=?:;,.(){}\[\]@])
""", re.X | re.S)


def lex(src):
    out = []
    pos = 0
    n = len(src)
    while pos < n:
        m = _TOKEN.match(src, pos)
        if not m:
            raise SyntaxError(f"java2py: cannot lex at {src[pos:pos + 40]!r}")
        pos = m.end()
        k = m.lastgroup
        if k in ("ws", "lc", "bc"):
            continue
        out.append((k, m.group()))
    # `>>` and `>>>` inside generics are split on demand by the parser (it only ever skips generics)
    out.append(("eof", ""))
    return out


PRIMS = {"int", "long", "boolean", "char", "short", "byte", "double", "float", "void"}
KNOWN_TYPES = PRIMS | {"Long", "Integer", "String", "Row", "Seq", "Object", "List", "ArrayList", "Iterator", "Tuple2",
                       "Boolean", "Character", "HashMap", "Map", "StringBuilder", "Iterable", "Tuple3", "Tuple4"}
MODIFIERS = {"public", "private", "protected", "static", "final", "transient", "volatile", "synchronized", "abstract"}


class Parser:
    def __init__(self, toks, class_names=()):
        self.t = toks
        self.i = 0
        self.class_names = set(class_names)

    # -- token helpers
    def peek(self, o=0):
        return self.t[self.i + o]

    def at(self, v, o=0):
        return self.t[self.i + o][1] == v and self.t[self.i + o][0] in ("op", "id")

    def eat(self, v=None):
        k, s = self.t[self.i]
        if v is not None and s != v:
            ctx = " ".join(x[1] for x in self.t[max(0, self.i - 12): self.i + 6])
            raise SyntaxError(f"java2py: expected {v!r}, got {s!r} near: {ctx}")
        self.i += 1
        return s

    def is_type_name(self, s):
        return s in KNOWN_TYPES or s in self.class_names or (s[:1].isupper() and s not in ("Long", "Integer") or s in ("Long", "Integer"))

    # -- types:  Ident(.Ident)* [<...>] ([])*
    def try_type(self, j):
        """if a type starts at token j: (type string, next index) else None"""
        k, s = self.t[j]
        if k != "id":
            return None
        if s not in PRIMS and not s[:1].isupper():
            return None
        name = s
        j += 1
        while self.t[j][1] == "." and self.t[j + 1][0] == "id" and self.t[j + 1][1][:1].isupper():
            name = self.t[j + 1][1]
            j += 2
        if self.t[j][1] == "<":
            depth = 0
            while True:
                v = self.t[j][1]
                if v == "<":
                    depth += 1
                elif v == ">":
                    depth -= 1
                elif v == ">>":
                    depth -= 2
                elif v == ">>>":
                    depth -= 3
                elif v in (";", "{", "}", "(", ")", "=") or self.t[j][0] == "eof":
                    return None
                j += 1
                if depth <= 0:
                    break
        dims = 0
        while self.t[j][1] == "[" and self.t[j + 1][1] == "]":
            dims += 1
            j += 2
        return (name + "[]" * dims, j)

    # ------------------------------------------------------------------------------------- class level
    def parse_class_body(self):
        """after the opening '{' of a class: -> dict(fields=[(type,name,init_expr|None)], methods={name: (params, body)},
        inner={name: ...})"""
        fields, methods, inner = [], {}, {}
        while not self.at("}"):
            if self.at(";"):
                self.eat()
                continue
            while self.peek()[1] in MODIFIERS or self.at("@"):
                if self.at("@"):
                    self.eat()
                    self.eat()
                    if self.at("("):
                        self.skip_parens()
                else:
                    self.eat()
            if self.at("class") or self.at("interface"):
                self.eat()
                name = self.eat()
                while not self.at("{"):
                    self.eat()
                self.eat("{")
                self.class_names.add(name)
                inner[name] = self.parse_class_body()
                self.eat("}")
                continue
            if self.at("{"):                       # initialiser block: not used by the operator classes
                raise SyntaxError("java2py: initialiser block")
            # constructor?
            if self.peek()[0] == "id" and self.peek(1)[1] == "(":
                name = self.eat()
                params = self.parse_params()
                while not self.at("{"):
                    self.eat()
                body = self.parse_block()
                methods["<init>"] = (params, body, "void")
                continue
            ty = self.try_type(self.i)
            if ty is None:
                raise SyntaxError(f"java2py: member? {self.peek()} {self.peek(1)}")
            tname, self.i = ty
            name = self.eat()
            if self.at("("):
                params = self.parse_params()
                while not self.at("{") and not self.at(";"):
                    self.eat()                      # throws ...
                if self.at(";"):
                    self.eat()
                    continue
                body = self.parse_block()
                methods[name] = (params, body, tname)
            else:
                while True:
                    dims = 0
                    while self.at("["):
                        self.eat("[")
                        self.eat("]")
                        dims += 1
                    init = None
                    if self.at("="):
                        self.eat()
                        init = self.parse_expr()
                    fields.append((tname + "[]" * dims, name, init))
                    if self.at(","):
                        self.eat()
                        name = self.eat()
                        continue
                    break
                self.eat(";")
        return dict(fields=fields, methods=methods, inner=inner)

    def skip_parens(self):
        depth = 0
        while True:
            v = self.eat()
            if v == "(":
                depth += 1
            elif v == ")":
                depth -= 1
                if depth == 0:
                    return

    def parse_params(self):
        self.eat("(")
        ps = []
        while not self.at(")"):
            while self.peek()[1] in MODIFIERS:
                self.eat()
            ty = self.try_type(self.i)
            tname, self.i = ty
            if self.at("."):                        # varargs "..."
                while self.at("."):
                    self.eat()
                tname += "[]"
            name = self.eat()
            while self.at("["):
                self.eat("[")
                self.eat("]")
                tname += "[]"
            ps.append((tname, name))
            if self.at(","):
                self.eat()
        self.eat(")")
        return ps

    # ------------------------------------------------------------------------------------- statements
    def parse_block(self):
        self.eat("{")
        out = []
        while not self.at("}"):
            out.append(self.parse_stmt())
        self.eat("}")
        return ("block", out)

    def parse_stmt(self):
        k, s = self.peek()
        if s == "{" and k == "op":
            return self.parse_block()
        if s == ";" and k == "op":
            self.eat()
            return ("empty",)
        if k == "id":
            if s == "if":
                self.eat()
                self.eat("(")
                c = self.parse_expr()
                self.eat(")")
                a = self.parse_stmt()
                b = None
                if self.at("else"):
                    self.eat()
                    b = self.parse_stmt()
                return ("if", c, a, b)
            if s == "while":
                self.eat()
                self.eat("(")
                c = self.parse_expr()
                self.eat(")")
                return ("while", c, self.parse_stmt())
            if s == "do":
                self.eat()
                body = self.parse_stmt()
                self.eat("while")
                self.eat("(")
                c = self.parse_expr()
                self.eat(")")
                self.eat(";")
                return ("dowhile", c, body)
            if s == "for":
                self.eat()
                self.eat("(")
                # for-each?
                save = self.i
                while self.peek()[1] in MODIFIERS:
                    self.eat()
                ty = self.try_type(self.i)
                if ty is not None and self.t[ty[1]][0] == "id" and self.t[ty[1] + 1][1] == ":":
                    self.i = ty[1]
                    name = self.eat()
                    self.eat(":")
                    coll = self.parse_expr()
                    self.eat(")")
                    return ("foreach", ty[0], name, coll, self.parse_stmt())
                self.i = save
                init = []
                if not self.at(";"):
                    init.append(self.parse_simple_stmt())
                    while self.at(","):
                        self.eat()
                        init.append(("expr", self.parse_expr()))
                self.eat(";")
                cond = None if self.at(";") else self.parse_expr()
                self.eat(";")
                upd = []
                while not self.at(")"):
                    upd.append(("expr", self.parse_expr()))
                    if self.at(","):
                        self.eat()
                self.eat(")")
                return ("for", init, cond, upd, self.parse_stmt())
            if s == "return":
                self.eat()
                e = None if self.at(";") else self.parse_expr()
                self.eat(";")
                return ("return", e)
            if s == "break":
                self.eat()
                self.eat(";")
                return ("break",)
            if s == "continue":
                self.eat()
                self.eat(";")
                return ("continue",)
            if s == "throw":
                self.eat()
                e = self.parse_expr()
                self.eat(";")
                return ("throw", e)
            if s == "try":
                self.eat()
                body = self.parse_block()
                handlers = []
                while self.at("catch"):
                    self.eat()
                    self.skip_parens()
                    handlers.append(self.parse_block())
                fin = None
                if self.at("finally"):
                    self.eat()
                    fin = self.parse_block()
                return ("try", body, handlers, fin)
            if s in ("switch", "synchronized", "class"):
                raise SyntaxError(f"java2py: statement {s}")
        st = self.parse_simple_stmt()
        self.eat(";")
        return st

    def parse_simple_stmt(self):
        """local declaration or expression (no trailing ';')"""
        while self.peek()[1] == "final":
            self.eat()
        ty = self.try_type(self.i)
        if ty is not None and self.t[ty[1]][0] == "id" and self.t[ty[1] + 1][1] in ("=", ";", ",", "["):
            tname, self.i = ty
            decls = []
            while True:
                name = self.eat()
                dims = 0
                while self.at("["):
                    self.eat("[")
                    self.eat("]")
                    dims += 1
                init = None
                if self.at("="):
                    self.eat()
                    init = self.parse_array_init() if self.at("{") else self.parse_expr()
                decls.append((tname + "[]" * dims, name, init))
                if self.at(","):
                    self.eat()
                    continue
                break
            return ("decl", decls)
        return ("expr", self.parse_expr())

    def parse_array_init(self):
        self.eat("{")
        items = []
        while not self.at("}"):
            items.append(self.parse_array_init() if self.at("{") else self.parse_expr())
            if self.at(","):
                self.eat()
        self.eat("}")
        return ("arrayinit", items)

    # ------------------------------------------------------------------------------------- expressions
    ASSIGN = {"=", "+=", "-=", "*=", "/=", "%=", "&=", "|=", "^=", "<<=", ">>=", ">>>="}
    BIN = [["||"], ["&&"], ["|"], ["^"], ["&"], ["==", "!="], ["<", ">", "<=", ">=", "instanceof"], ["<<", ">>", ">>>"],
           ["+", "-"], ["*", "/", "%"]]

    def parse_expr(self):
        lhs = self.parse_ternary()
        if self.peek()[0] == "op" and self.peek()[1] in self.ASSIGN:
            op = self.eat()
            rhs = self.parse_expr()
            return ("assign", op, lhs, rhs)
        return lhs

    def parse_ternary(self):
        c = self.parse_bin(0)
        if self.at("?"):
            self.eat()
            a = self.parse_expr()
            self.eat(":")
            b = self.parse_ternary()
            return ("cond", c, a, b)
        return c

    def parse_bin(self, lvl):
        if lvl == len(self.BIN):
            return self.parse_unary()
        lhs = self.parse_bin(lvl + 1)
        while (self.peek()[0] in ("op", "id")) and self.peek()[1] in self.BIN[lvl]:
            op = self.eat()
            if op == "instanceof":
                ty = self.try_type(self.i)
                self.i = ty[1]
                lhs = ("instanceof", lhs, ty[0])
                continue
            rhs = self.parse_bin(lvl + 1)
            lhs = ("bin", op, lhs, rhs)
        return lhs

    def parse_unary(self):
        k, s = self.peek()
        if k == "op":
            if s in ("+", "-", "~", "!"):
                self.eat()
                return ("un", s, self.parse_unary())
            if s in ("++", "--"):
                self.eat()
                return ("preinc", s, self.parse_unary())
            if s == "(":
                # cast?
                ty = self.try_type(self.i + 1)
                if ty is not None and self.t[ty[1]][1] == ")":
                    nxt = self.t[ty[1] + 1]
                    tn = ty[0]
                    base = tn.rstrip("[]")
                    is_prim = base in PRIMS
                    starts_unary_not_pm = nxt[0] in ("id", "num", "fnum", "str", "chr") or nxt[1] in ("(", "!", "~")
                    if (is_prim and (starts_unary_not_pm or nxt[1] in ("+", "-"))) or \
                            (not is_prim and starts_unary_not_pm and self.is_type_name(base) and nxt[1] not in ("instanceof",)):
                        self.i = ty[1] + 1
                        return ("cast", tn, self.parse_unary())
        return self.parse_postfix()

    def parse_postfix(self):
        e = self.parse_primary()
        while True:
            if self.at("."):
                self.eat()
                if self.at("<"):                       # explicit generic method call  .<T>foo()
                    ty = self.try_type(self.i - 1)
                name = self.eat()
                if self.at("("):
                    e = ("call", e, name, self.parse_args())
                else:
                    e = ("field", e, name)
            elif self.at("["):
                self.eat()
                idx = self.parse_expr()
                self.eat("]")
                e = ("index", e, idx)
            elif self.at("++") or self.at("--"):
                e = ("postinc", self.eat(), e)
            else:
                return e

    def parse_args(self):
        self.eat("(")
        a = []
        while not self.at(")"):
            a.append(self.parse_expr())
            if self.at(","):
                self.eat()
        self.eat(")")
        return a

    def parse_primary(self):
        k, s = self.peek()
        if k == "num":
            self.eat()
            return ("num", s)
        if k == "fnum":
            self.eat()
            return ("fnum", s)
        if k == "str":
            self.eat()
            return ("str", s)
        if k == "chr":
            self.eat()
            return ("chr", s)
        if k == "op" and s == "(":
            self.eat()
            e = self.parse_expr()
            self.eat(")")
            return ("paren", e)
        if k == "id":
            if s == "new":
                self.eat()
                ty = self.try_type(self.i)
                # array creation: new T[n] / new T[]{...}
                base_end = self.i
                # re-scan: type without dims
                j = self.i
                name = self.t[j][1]
                j += 1
                while self.t[j][1] == "." and self.t[j + 1][0] == "id":
                    name = self.t[j + 1][1]
                    j += 2
                if self.t[j][1] == "<":
                    depth = 0
                    while True:
                        v = self.t[j][1]
                        depth += {"<": 1, ">": -1, ">>": -2, ">>>": -3}.get(v, 0)
                        j += 1
                        if depth <= 0:
                            break
                self.i = j
                if self.at("["):
                    dims = []
                    nd = 0
                    while self.at("["):
                        self.eat("[")
                        if self.at("]"):
                            dims.append(None)
                        else:
                            dims.append(self.parse_expr())
                        self.eat("]")
                        nd += 1
                    if self.at("{"):
                        return ("newarray_init", name, self.parse_array_init())
                    return ("newarray", name, dims)
                args = self.parse_args()
                if self.at("{"):
                    raise SyntaxError("java2py: anonymous class")
                return ("new", name, args)
            if s in ("true", "false", "null"):
                self.eat()
                return ("const", s)
            if s == "this":
                self.eat()
                return ("this",)
            self.eat()
            if self.at("("):
                return ("call", None, s, self.parse_args())
            return ("name", s)
        raise SyntaxError(f"java2py: primary? {k} {s!r} near " + " ".join(x[1] for x in self.t[max(0, self.i - 10): self.i + 5]))


# ------------------------------------------------------------------------------------------------ code generator
def _elem_conv(tname):
    """conversion applied when a value is stored into a variable / array element of this declared type"""
    base = tname
    if base in ("long", "Long"):
        return "_L"
    if base in ("int", "Integer", "short", "byte"):
        return "_I"
    return None


class Gen:
    def __init__(self, cls_name, cls, outer_fields=(), sibling_classes=()):
        self.cls_name = cls_name
        self.cls = cls
        self.field_types = {n: t for t, n, _ in cls["fields"]}
        self.methods = set(cls["methods"])
        self.outer = set(outer_fields)
        self.siblings = set(sibling_classes)
        self.lines = []
        self.scopes = []
        self.tmp = 0

    def emit(self, ind, s):
        self.lines.append("    " * ind + s)

    # -- names
    def lookup_local(self, name):
        for sc in reversed(self.scopes):
            if name in sc:
                return sc[name]
        return None

    def var_type(self, e):
        """declared type of an lvalue expression (for the store conversion), or None"""
        if e[0] == "name":
            t = self.lookup_local(e[1])
            if t is not None:
                return t
            return self.field_types.get(e[1])
        if e[0] == "index":
            t = self.var_type(e[1])
            if t and t.endswith("[]"):
                return t[:-2]
            return None
        if e[0] == "field" and e[1][0] == "this":
            return self.field_types.get(e[2])
        if e[0] == "paren":
            return self.var_type(e[1])
        return None

    def name(self, n):
        if self.lookup_local(n) is not None:
            return n
        if n in self.field_types:
            return "self." + n
        if n in self.outer:
            return "self._outer." + n
        return n                                      # class names (Long, Math, RowFactory ...) -> runtime shims

    # -- expressions
    def ex(self, e):
        k = e[0]
        if k == "num":
            s = e[1]
            if s[-1] in "lL":
                return f"_L({int(s[:-1], 0)})"
            return f"_I({int(s, 0) if not (s.startswith('0') and len(s) > 1 and s[1] not in 'xX') else int(s, 8)})"
        if k == "fnum":
            return repr(float(e[1].rstrip("fFdD")))
        if k == "str":
            return e[1]
        if k == "chr":
            return f"_C({e[1]})"
        if k == "const":
            return {"true": "True", "false": "False", "null": "None"}[e[1]]
        if k == "this":
            return "self"
        if k == "paren":
            return "(" + self.ex(e[1]) + ")"
        if k == "name":
            return self.name(e[1])
        if k == "field":
            if e[1][0] == "this":
                return "self." + e[2]
            if e[2] == "length":
                return f"_len({self.ex(e[1])})"
            if re.fullmatch(r"_\d", e[2]):                # scala tuple field  s._1
                return f"_call({self.ex(e[1])}, {e[2]!r})"
            return f"{self.ex(e[1])}.{e[2]}"
        if k == "index":
            return f"{self.ex(e[1])}[{self.ex(e[2])}]"
        if k == "un":
            op = e[1]
            if op == "!":
                return f"(not {self.ex(e[2])})"
            if op == "+":
                return self.ex(e[2])
            return f"({op}{self.ex(e[2])})"
        if k == "bin":
            op, a, b = e[1], self.ex(e[2]), self.ex(e[3])
            if op == "&&":
                return f"({a} and {b})"
            if op == "||":
                return f"({a} or {b})"
            if op == ">>>":
                return f"_ushr({a}, {b})"
            if op in ("==", "!="):
                return f"_eq({a}, {b})" if op == "==" else f"(not _eq({a}, {b}))"
            if op == "+":
                return f"_add({a}, {b})"
            return f"({a} {op} {b})"
        if k == "cond":
            return f"({self.ex(e[2])} if {self.ex(e[1])} else {self.ex(e[3])})"
        if k == "instanceof":
            return f"_instanceof({self.ex(e[1])}, {e[2]!r})"
        if k == "cast":
            t = e[1]
            v = self.ex(e[2])
            if t == "int":
                return f"_cast_int({v})"
            if t == "long":
                return f"_cast_long({v})"
            if t == "char":
                return f"_cast_char({v})"
            if t in ("double", "float"):
                return f"_cast_double({v})"
            if t in ("short", "byte"):
                raise SyntaxError(f"java2py: cast to {t}")
            if t == "Long":
                return f"_box_long({v})"
            if t == "Integer":
                return f"_box_int({v})"
            return v                                   # reference casts: (Row), (String), (Seq) ...
        if k == "call":
            tgt, name, args = e[1], e[2], [self.ex(a) for a in e[3]]
            if tgt is None:
                if name in self.methods:
                    return f"self.{name}({', '.join(args)})"
                return f"self._outer.{name}({', '.join(args)})"
            if tgt[0] == "this":
                return f"self.{name}({', '.join(args)})"
            return f"_call({self.ex(tgt)}, {name!r}{''.join(', ' + a for a in args)})"
        if k == "new":
            name, args = e[1], [self.ex(a) for a in e[2]]
            if name in self.siblings:
                return f"self._outer._new({name!r}{''.join(', ' + a for a in args)})"
            return f"_new({name!r}{''.join(', ' + a for a in args)})"
        if k == "newarray":
            name, dims = e[1], e[2]
            if len(dims) == 1:
                return f"_newarr({name!r}, {self.ex(dims[0])})"
            if len(dims) == 2 and dims[1] is not None:
                return f"[_newarr({name!r}, {self.ex(dims[1])}) for _ in range(int({self.ex(dims[0])}))]"
            if len(dims) == 2:
                return f"[None] * int({self.ex(dims[0])})"
            raise SyntaxError("java2py: array dims")
        if k == "newarray_init":
            conv = _elem_conv(e[1])
            items = [self.ex(x) for x in e[2][1]]
            if conv:
                items = [f"{conv}({x})" for x in items]
            return "[" + ", ".join(items) + "]"
        if k == "arrayinit":
            return "[" + ", ".join(self.ex(x) for x in e[1]) + "]"
        if k == "assign":
            raise SyntaxError("java2py: assignment inside an expression")
        if k in ("postinc", "preinc"):
            raise SyntaxError("java2py: ++/-- inside an expression")
        raise SyntaxError(f"java2py: expr {k}")

    # -- statements
    def store(self, ind, lhs, rhs_code, compound=False):
        """plain assignment: the value must already fit the declared type (_I raises on a long, as javac rejects it);
        compound assignment `E1 op= E2` is `E1 = (T)((E1) op (E2))` with T the type of E1 (JLS 15.26.2): an implicit
        NARROWING cast, likewise ++ / -- (15.14.2, 15.15.1)"""
        t = self.var_type(lhs)
        conv = _elem_conv(t) if t else None
        if conv and compound:
            rhs_code = f"{'_cast_int' if conv == '_I' else '_cast_long'}({rhs_code})"
        elif conv:
            rhs_code = f"{conv}({rhs_code})"
        elif t == "char":
            rhs_code = f"_cast_char({rhs_code})"
        self.emit(ind, f"{self.ex(lhs)} = {rhs_code}")

    def expr_stmt(self, ind, e):
        k = e[0]
        if k == "paren":
            return self.expr_stmt(ind, e[1])
        if k == "assign":
            op, lhs, rhs = e[1], e[2], e[3]
            if rhs[0] == "assign":                        # a = b = c
                self.expr_stmt(ind, rhs)
                rhs = rhs[2]
            if op == "=":
                if rhs[0] == "arrayinit":
                    return self.store(ind, lhs, self.ex(rhs))
                return self.store(ind, lhs, self.ex(rhs))
            bop = op[:-1]
            if self.var_type(lhs) == "String" and bop != "+":
                raise SyntaxError("java2py: String " + op)
            return self.store(ind, lhs, self.ex(("bin", bop, lhs, ("paren", rhs))), compound=self.var_type(lhs) != "String")
        if k in ("postinc", "preinc"):
            lhs = e[2]
            bop = "+" if e[1] == "++" else "-"
            return self.store(ind, lhs, self.ex(("bin", bop, lhs, ("num", "1"))), compound=True)
        if k in ("call", "new"):
            return self.emit(ind, self.ex(e))
        raise SyntaxError(f"java2py: expression statement {k}")

    def stmt(self, ind, s, loop_upd=None):
        k = s[0]
        if k == "block":
            self.scopes.append({})
            if not s[1]:
                self.emit(ind, "pass")
            for x in s[1]:
                self.stmt(ind, x, loop_upd)
            self.scopes.pop()
        elif k == "empty":
            self.emit(ind, "pass")
        elif k == "decl":
            for t, n, init in s[1]:
                self.scopes[-1][n] = t
                if init is None:
                    dflt = {"long": "_L(0)", "int": "_I(0)", "boolean": "False", "char": "_C('\\0')"}.get(t, "None")
                    self.emit(ind, f"{n} = {dflt}")
                else:
                    if init[0] == "arrayinit":
                        conv = _elem_conv(t[:-2]) if t.endswith("[]") else None
                        items = [self.ex(x) for x in init[1]]
                        if conv:
                            items = [f"{conv}({x})" for x in items]
                        self.emit(ind, f"{n} = [" + ", ".join(items) + "]")
                    else:
                        self.store(ind, ("name", n), self.ex(init))
        elif k == "expr":
            self.expr_stmt(ind, s[1])
        elif k == "if":
            self.emit(ind, f"if {self.ex(s[1])}:")
            self.body(ind + 1, s[2], loop_upd)
            b = s[3]
            while b is not None and b[0] == "if":
                self.emit(ind, f"elif {self.ex(b[1])}:")
                self.body(ind + 1, b[2], loop_upd)
                b = b[3]
            if b is not None:
                self.emit(ind, "else:")
                self.body(ind + 1, b, loop_upd)
        elif k == "while":
            self.emit(ind, f"while {self.ex(s[1])}:")
            self.body(ind + 1, s[2], [])
        elif k == "dowhile":
            self.emit(ind, "while True:")
            self.body(ind + 1, s[2], [("__dowhile__", s[1])])
            self.emit(ind + 1, f"if not ({self.ex(s[1])}):")
            self.emit(ind + 2, "break")
        elif k == "for":
            self.scopes.append({})
            for x in s[1]:
                self.stmt(ind, x)
            self.emit(ind, f"while {self.ex(s[2]) if s[2] is not None else 'True'}:")
            self.body(ind + 1, s[4], s[3])
            for u in s[3]:
                self.stmt(ind + 1, u)
            self.scopes.pop()
        elif k == "foreach":
            self.scopes.append({s[2]: s[1]})
            self.emit(ind, f"for {s[2]} in _iterate({self.ex(s[3])}):")
            self.body(ind + 1, s[4], [])
            self.scopes.pop()
        elif k == "return":
            if s[1] is None:
                self.emit(ind, "return")
            else:
                conv = _elem_conv(self.cur_rtype) if getattr(self, "cur_rtype", None) in ("int", "long", "short", "byte") else None
                v = self.ex(s[1])
                self.emit(ind, f"return {conv}({v})" if conv else f"return {v}")
        elif k == "break":
            self.emit(ind, "break")
        elif k == "continue":
            for u in (loop_upd or []):
                if u[0] == "__dowhile__":
                    self.emit(ind, f"if not ({self.ex(u[1])}):")
                    self.emit(ind + 1, "break")
                else:
                    self.stmt(ind, u)
            self.emit(ind, "continue")
        elif k == "throw":
            self.emit(ind, f"raise _JavaThrow({self.ex(s[1])})")
        elif k == "try":
            self.emit(ind, "try:")
            self.body(ind + 1, s[1], loop_upd)
            self.emit(ind, "except _JavaThrow:")
            if s[2]:
                self.body(ind + 1, s[2][0], loop_upd)
            else:
                self.emit(ind + 1, "raise")
            if s[3]:
                self.emit(ind, "finally:")
                self.body(ind + 1, s[3], loop_upd)
        else:
            raise SyntaxError(f"java2py: stmt {k}")

    def body(self, ind, s, loop_upd):
        n0 = len(self.lines)
        self.stmt(ind, s, loop_upd)
        if len(self.lines) == n0:
            self.emit(ind, "pass")

    def gen_class(self):
        self.emit(0, f"class {self.cls_name}:")
        self.emit(1, "def __init__(self, _outer):")
        self.emit(2, "self._outer = _outer")
        self.scopes = [{}]
        for t, n, init in self.cls["fields"]:
            if init is None:
                dflt = {"long": "_L(0)", "int": "_I(0)", "boolean": "False"}.get(t, "None")
                self.emit(2, f"self.{n} = {dflt}")
            else:
                if init[0] == "arrayinit":
                    self.emit(2, f"self.{n} = {self.ex(init)}")
                else:
                    self.store(2, ("name", n), self.ex(init))
        for name, (params, body, rtype) in self.cls["methods"].items():
            if name == "<init>":
                continue
            self.scopes = [{n: t for t, n in params}]
            self.cur_rtype = rtype
            self.emit(1, f"def {name}(self{''.join(', ' + n for _, n in params)}):")
            n0 = len(self.lines)
            # parameters of primitive type arrive converted (callers may pass a narrower type)
            for t, n in params:
                conv = _elem_conv(t)
                if conv:
                    self.emit(2, f"{n} = {conv}({n})")
            self.stmt(2, body)
            if len(self.lines) == n0:
                self.emit(2, "pass")
        return "\n".join(self.lines) + "\n"


# ------------------------------------------------------------------------------------------------ runtime
class _JavaThrow(Exception):
    pass


class J:
    """a Java int (w = 32) or long (w = 64), value kept signed"""
    __slots__ = ("v", "w")

    def __init__(self, v, w):
        m = (1 << w) - 1
        v &= m
        if v >> (w - 1):
            v -= 1 << w
        self.v = v
        self.w = w

    def __repr__(self):
        return f"{self.v}{'L' if self.w == 64 else ''}"

    def __str__(self):
        return str(self.v)

    def __index__(self):
        return self.v

    def __int__(self):
        return self.v

    def __hash__(self):
        return hash(self.v)

    def __bool__(self):
        raise TypeError("java2py: int used as boolean")

    @staticmethod
    def _o(o):
        if isinstance(o, J):
            return o
        if isinstance(o, bool):
            raise TypeError("java2py: boolean in arithmetic")
        if isinstance(o, int):
            return J(o, 32)
        if isinstance(o, Ch):
            return J(ord(o.c), 32)
        raise TypeError(f"java2py: arithmetic on {type(o).__name__}")

    def _bin(self, o, f):
        o = J._o(o)
        return J(f(self.v, o.v), max(self.w, o.w, 32))

    def __add__(self, o): return self._bin(o, lambda a, b: a + b)
    def __radd__(self, o): return J._o(o)._bin(self, lambda a, b: a + b)
    def __sub__(self, o): return self._bin(o, lambda a, b: a - b)
    def __rsub__(self, o): return J._o(o)._bin(self, lambda a, b: a - b)
    def __mul__(self, o): return self._bin(o, lambda a, b: a * b)
    def __rmul__(self, o): return J._o(o)._bin(self, lambda a, b: a * b)
    def __and__(self, o): return self._bin(o, lambda a, b: a & b)
    def __rand__(self, o): return J._o(o)._bin(self, lambda a, b: a & b)
    def __or__(self, o): return self._bin(o, lambda a, b: a | b)
    def __ror__(self, o): return J._o(o)._bin(self, lambda a, b: a | b)
    def __xor__(self, o): return self._bin(o, lambda a, b: a ^ b)
    def __rxor__(self, o): return J._o(o)._bin(self, lambda a, b: a ^ b)

    @staticmethod
    def _jdiv(a, b):
        if b == 0:
            raise _JavaThrow("ArithmeticException: / by zero")
        q = abs(a) // abs(b)
        return q if (a < 0) == (b < 0) else -q

    def __truediv__(self, o):
        if isinstance(o, float):
            return float(self.v) / o
        return self._bin(o, J._jdiv)

    def __rtruediv__(self, o):
        if isinstance(o, float):
            return o / float(self.v) if self.v else (float("inf") if o > 0 else float("-inf") if o < 0 else float("nan"))
        return J._o(o)._bin(self, J._jdiv)
    def __mod__(self, o): return self._bin(o, lambda a, b: a - b * J._jdiv(a, b))
    def __rmod__(self, o): return J._o(o)._bin(self, lambda a, b: a - b * J._jdiv(a, b))

    def __lshift__(self, o):
        o = J._o(o)
        return J(self.v << (o.v & (self.w - 1)), self.w)

    def __rlshift__(self, o): return J._o(o).__lshift__(self)

    def __rshift__(self, o):
        o = J._o(o)
        return J(self.v >> (o.v & (self.w - 1)), self.w)

    def __rrshift__(self, o): return J._o(o).__rshift__(self)

    def ushr(self, o):
        o = J._o(o)
        return J((self.v & ((1 << self.w) - 1)) >> (o.v & (self.w - 1)), self.w)

    def __neg__(self): return J(-self.v, self.w)
    def __invert__(self): return J(~self.v, self.w)
    def __lt__(self, o): return self.v < J._o(o).v
    def __le__(self, o): return self.v <= J._o(o).v
    def __gt__(self, o): return self.v > J._o(o).v
    def __ge__(self, o): return self.v >= J._o(o).v
    def __eq__(self, o): return o is not None and not isinstance(o, str) and self.v == J._o(o).v
    def __ne__(self, o): return not self.__eq__(o)


class Ch:
    """a Java char"""
    __slots__ = ("c",)

    def __init__(self, c):
        self.c = c

    def __eq__(self, o):
        if isinstance(o, Ch):
            return self.c == o.c
        if isinstance(o, J):
            return ord(self.c) == o.v
        return False

    def __hash__(self):
        return hash(self.c)

    def __str__(self):
        return self.c

    def __repr__(self):
        return repr(self.c)

    # binary numeric promotion (JLS 5.6.2): a char operand is an int
    def _j(self): return J(ord(self.c), 32)
    def __add__(self, o): return self._j() + o
    def __radd__(self, o): return J._o(o) + self._j()
    def __sub__(self, o): return self._j() - o
    def __rsub__(self, o): return J._o(o) - self._j()
    def __mul__(self, o): return self._j() * o
    def __rmul__(self, o): return J._o(o) * self._j()
    def __and__(self, o): return self._j() & o
    def __or__(self, o): return self._j() | o
    def __xor__(self, o): return self._j() ^ o
    def __lshift__(self, o): return self._j() << o
    def __rshift__(self, o): return self._j() >> o
    def __truediv__(self, o): return self._j() / o
    def __mod__(self, o): return self._j() % o
    def __index__(self): return ord(self.c)
    def __lt__(self, o): return ord(self.c) < _ordv(o)
    def __le__(self, o): return ord(self.c) <= _ordv(o)
    def __gt__(self, o): return ord(self.c) > _ordv(o)
    def __ge__(self, o): return ord(self.c) >= _ordv(o)


def _ordv(o):
    return ord(o.c) if isinstance(o, Ch) else J._o(o).v


class Boxed(J):
    """a java.lang.Long / Integer OBJECT: `==` between two of them compares references (Java caches -128..127 only)"""
    __slots__ = ()


class Row:
    def __init__(self, vals):
        self.vals = list(vals)

    def getLong(self, i): return _L(self.vals[int(i)])
    def getInt(self, i): return _I(self.vals[int(i)])
    def getSeq(self, i):
        v = self.vals[int(i)]
        return Seq(v) if isinstance(v, list) else v      # a Java array put into a Row reads back as a Seq
    def getString(self, i): return self.vals[int(i)]
    def get(self, i): return self.vals[int(i)]
    def getList(self, i): return JList(self.vals[int(i)].items)
    def length(self): return _I(len(self.vals))
    def size(self): return _I(len(self.vals))
    def __repr__(self): return f"Row{self.vals}"


class Seq:
    def __init__(self, items):
        self.items = list(items)

    def apply(self, i):
        i = int(i)
        if i < 0 or i >= len(self.items):
            raise _JavaThrow(f"IndexOutOfBoundsException: {i}")
        return self.items[i]

    def length(self): return _I(len(self.items))
    def size(self): return _I(len(self.items))
    def __repr__(self): return f"Seq{self.items}"


class JIter:
    def __init__(self, items):
        self.it = list(items)
        self.p = 0

    def hasNext(self): return self.p < len(self.it)

    def next(self):
        if self.p >= len(self.it):
            raise _JavaThrow("NoSuchElementException")
        self.p += 1
        return self.it[self.p - 1]

    def asScala(self): return self
    def toSeq(self): return Seq(self.it[self.p:])


class JList:
    def __init__(self, items=()):
        self.items = list(items)

    def add(self, *a):
        if len(a) == 1:
            self.items.append(a[0])
        else:
            self.items.insert(int(a[0]), a[1])
        return True

    def addAll(self, o):
        self.items.extend(o.items)
        return True

    def get(self, i):
        i = int(i)
        if i < 0 or i >= len(self.items):
            raise _JavaThrow(f"IndexOutOfBoundsException: {i} of {len(self.items)}")
        return self.items[i]

    def set(self, i, v):
        self.items[int(i)] = v

    def remove(self, i):
        if isinstance(i, J) and i.w == 32 and not isinstance(i, Boxed):
            return self.items.pop(int(i))
        self.items.remove(i)
        return True

    def size(self): return _I(len(self.items))
    def isEmpty(self): return len(self.items) == 0
    def clear(self): self.items = []
    def iterator(self): return JIter(self.items)
    def asScala(self): return self
    def toSeq(self): return Seq(self.items)
    def seq(self): return Seq(self.items)
    def toList(self): return self
    def contains(self, x): return x in self.items


class JTuple:
    """scala.Tuple2 .. Tuple4: `t._1` and `t._1()` both read element 1"""
    def __init__(self, *items):
        self.items = list(items)

    def __repr__(self):
        return f"Tuple{tuple(self.items)}"


class JMap:
    """java.util.HashMap; a Java ARRAY used as a key hashes by identity (never equal to a boxed Integer key)"""
    def __init__(self): self.d = {}
    @staticmethod
    def _k(k): return ("@array", id(k)) if isinstance(k, list) else k
    def put(self, k, v): self.d[JMap._k(k)] = v
    def get(self, k): return self.d.get(JMap._k(k))
    def containsKey(self, k): return JMap._k(k) in self.d
    def size(self): return _I(len(self.d))
    def remove(self, k): return self.d.pop(k, None)


class JStringBuilder:
    def __init__(self, s=""): self.s = str(s)
    def append(self, x):
        self.s += _tostr(x)
        return self
    def toString(self): return self.s
    def length(self): return _I(len(self.s))
    def reverse(self):
        self.s = self.s[::-1]
        return self


def _I(x):
    if isinstance(x, J):
        if x.w != 32:
            raise TypeError("java2py: long stored into an int without a cast")
        return J(x.v, 32)
    if isinstance(x, Ch):
        return J(ord(x.c), 32)
    if isinstance(x, bool) or x is None:
        raise TypeError(f"java2py: {x!r} stored into an int")
    return J(int(x), 32)


def _L(x):
    if isinstance(x, J):
        return J(x.v, 64)
    if isinstance(x, Ch):
        return J(ord(x.c), 64)
    if x is None:
        return None                                   # a Long variable may hold null
    if isinstance(x, bool):
        raise TypeError("java2py: boolean stored into a long")
    return J(int(x), 64)


def _C(s):
    if isinstance(s, Ch):
        return s
    return Ch(s.encode().decode("unicode_escape") if len(s) > 1 else s)


def _cast_int(x):
    x = J._o(x)
    return J(x.v, 32)


def _cast_long(x):
    x = J._o(x)
    return J(x.v, 64)


def _cast_double(x):
    return float(x) if isinstance(x, float) else float(J._o(x).v)


def _cast_char(x):
    if isinstance(x, Ch):
        return x
    return Ch(chr(J._o(x).v & 0xFFFF))


def _box_long(x):
    if x is None:
        return None
    if not isinstance(x, J) or x.w != 64:
        raise _JavaThrow(f"ClassCastException: {type(x).__name__} {x!r} to Long")
    return x


def _box_int(x):
    if x is None:
        return None
    if not isinstance(x, J) or x.w != 32:
        raise _JavaThrow(f"ClassCastException: {x!r} to Integer")
    return x


def _ushr(a, b):
    return J._o(a).ushr(b)


def _eq(a, b):
    if a is None or b is None:
        return a is b
    if isinstance(a, str) or isinstance(b, str):
        raise TypeError("java2py: String == (reference comparison) is not modelled")
    if isinstance(a, (J, Ch)) or isinstance(b, (J, Ch)):
        return a == b
    if isinstance(a, bool) and isinstance(b, bool):
        return a == b
    return a is b


def _tostr(x):
    if x is None:
        return "null"
    if isinstance(x, bool):
        return "true" if x else "false"
    return str(x)


def _add(a, b):
    if isinstance(a, str) or isinstance(b, str):
        return _tostr(a) + _tostr(b)
    if isinstance(a, Ch):
        a = J(ord(a.c), 32)
    if isinstance(b, Ch):
        b = J(ord(b.c), 32)
    return a + b


def _len(a):
    return J(len(a), 32)


def _newarr(tname, n):
    n = int(n)
    if n < 0:
        raise _JavaThrow("NegativeArraySizeException")
    if tname == "long":
        return [J(0, 64) for _ in range(n)]
    if tname == "int":
        return [J(0, 32) for _ in range(n)]
    if tname == "boolean":
        return [False] * n
    if tname == "char":
        return [Ch("\0")] * n
    return [None] * n


def _iterate(c):
    if isinstance(c, (JList, Seq)):
        return list(c.items)
    if isinstance(c, list):
        return list(c)
    raise TypeError(f"java2py: for-each over {type(c).__name__}")


def _instanceof(x, t):
    if t == "Long":
        return isinstance(x, J) and x.w == 64
    if t == "Integer":
        return isinstance(x, J) and x.w == 32
    if t == "String":
        return isinstance(x, str)
    if t in ("Seq", "WrappedArray"):
        return isinstance(x, Seq)
    raise TypeError(f"java2py: instanceof {t}")


class _Long:
    SIZE = J(64, 32)
    MAX_VALUE = J((1 << 63) - 1, 64)
    MIN_VALUE = J(-(1 << 63), 64)

    @staticmethod
    def numberOfLeadingZeros(x):
        x = J._o(x)
        if x.w != 64:
            raise TypeError("Long.numberOfLeadingZeros(int)")   # would widen; flag it rather than guess
        u = x.v & ((1 << 64) - 1)
        return J(64 - u.bit_length(), 32)

    @staticmethod
    def numberOfTrailingZeros(x):
        u = J._o(x).v & ((1 << 64) - 1)
        return J(64 if u == 0 else (u & -u).bit_length() - 1, 32)

    @staticmethod
    def parseLong(s):
        try:
            return J(int(s, 10), 64)
        except ValueError:
            raise _JavaThrow(f"NumberFormatException: {s!r}")

    @staticmethod
    def valueOf(x):
        return _L(x) if not isinstance(x, str) else _Long.parseLong(x)

    @staticmethod
    def toBinaryString(x):
        return bin(J._o(x).v & ((1 << 64) - 1))[2:]


class _Integer:
    SIZE = J(32, 32)
    MAX_VALUE = J((1 << 31) - 1, 32)
    MIN_VALUE = J(-(1 << 31), 32)

    @staticmethod
    def parseInt(s):
        try:
            return J(int(s, 10), 32)
        except ValueError:
            raise _JavaThrow(f"NumberFormatException: {s!r}")

    @staticmethod
    def valueOf(x):
        return _I(x) if not isinstance(x, str) else _Integer.parseInt(x)

    @staticmethod
    def numberOfLeadingZeros(x):
        u = J._o(x).v & 0xFFFFFFFF
        return J(32 - u.bit_length(), 32)


class _Math:
    @staticmethod
    def abs(x):
        x = J._o(x)
        return J(abs(x.v), x.w)

    @staticmethod
    def min(a, b):
        a, b = J._o(a), J._o(b)
        return J(min(a.v, b.v), max(a.w, b.w))

    @staticmethod
    def max(a, b):
        a, b = J._o(a), J._o(b)
        return J(max(a.v, b.v), max(a.w, b.w))


class _RowFactory:
    @staticmethod
    def create(*vals):
        return Row(vals)


class _Arrays:
    @staticmethod
    def equals(a, b):
        if a is None or b is None:
            return a is b
        return len(a) == len(b) and all(_eq(x, y) if not (isinstance(x, J) and isinstance(y, J)) else x.v == y.v for x, y in zip(a, b))

    @staticmethod
    def asList(*a):
        # Arrays.asList(T... a): one array argument -> a list VIEW of its elements.  (For a primitive long[] javac
        # makes a one-element List<long[]>; the operator classes then wrap that into a Scala Seq and read it back
        # element-wise, i.e. they mean the elements -- modelled as the elements; noted in DESIGN.md.)
        if len(a) == 1 and isinstance(a[0], list):
            return JList(a[0])
        return JList(a)


class _Conv:
    def __init__(self, x): self.x = x
    def asScala(self): return self
    def toSeq(self): return Seq(self.x.items if isinstance(self.x, (JList, Seq)) else self.x.it[self.x.p:])
    def seq(self): return self.toSeq()
    def toList(self): return self
    def iterator(self): return JIter(self.toSeq().items)


class _JavaConverters:
    collectionAsScalaIterableConverter = staticmethod(lambda x: _Conv(x))
    asScalaIteratorConverter = staticmethod(lambda x: _Conv(x))
    asScalaBufferConverter = staticmethod(lambda x: _Conv(x))
    iterableAsScalaIterableConverter = staticmethod(lambda x: _Conv(x))
    seqAsJavaListConverter = staticmethod(lambda x: type("X", (), {"asJava": lambda s: JList(x.items)})())


class _Sys:
    class out:
        @staticmethod
        def println(*a): pass
        @staticmethod
        def print(*a): pass


_STR_METHODS = {
    "length": lambda s: J(len(s), 32),
    "charAt": lambda s, i: Ch(s[int(i)]) if 0 <= int(i) < len(s) else (_ for _ in ()).throw(_JavaThrow("StringIndexOutOfBounds")),
    "equals": lambda s, o: isinstance(o, str) and s == o,
    "startsWith": lambda s, p: s.startswith(p),
    "endsWith": lambda s, p: s.endswith(p),
    "substring": lambda s, a, b=None: s[int(a):] if b is None else s[int(a):int(b)],
    "isEmpty": lambda s: len(s) == 0,
    "trim": lambda s: s.strip(),
    "toString": lambda s: s,
    "indexOf": lambda s, p: J(s.find(str(p)), 32),
    "contains": lambda s, p: str(p) in s,
    "toCharArray": lambda s: [Ch(c) for c in s],
    "hashCode": lambda s: J(_jhash(s), 32),
    "replace": lambda s, a, b: s.replace(str(a), str(b)),
    "replaceAll": lambda s, a, b: re.sub(a, b, s),
    "split": lambda s, p, lim=None: _jsplit(s, p),
    "compareTo": lambda s, o: J((s > o) - (s < o), 32),
}


def _jhash(s):
    h = 0
    for c in s:
        h = (31 * h + ord(c)) & 0xFFFFFFFF
    return h


def _jsplit(s, p):
    parts = re.split(p, s)
    while parts and parts[-1] == "":                  # Java drops trailing empty strings
        parts.pop()
    if not parts:
        parts = [""] if s == "" else parts
    return parts


def _call(obj, name, *args):
    if isinstance(obj, str):
        f = _STR_METHODS.get(name)
        if f is None:
            raise TypeError(f"java2py: String.{name}")
        return f(obj, *args)
    if isinstance(obj, J):
        if name == "equals":
            o = args[0]
            return isinstance(o, J) and o.w == obj.w and o.v == obj.v
        if name in ("longValue",):
            return J(obj.v, 64)
        if name in ("intValue",):
            return J(obj.v, 32)
        if name == "toString":
            return str(obj.v)
        if name == "hashCode":
            return J(obj.v ^ (obj.v >> 32), 32) if obj.w == 64 else J(obj.v, 32)
        if name == "compareTo":
            o = args[0]
            return J((obj.v > o.v) - (obj.v < o.v), 32)
        raise TypeError(f"java2py: Long.{name}")
    if isinstance(obj, Ch) and name == "equals":
        return obj == args[0]
    if obj is None:
        raise _JavaThrow(f"NullPointerException: .{name}()")
    if isinstance(obj, JTuple) and len(name) == 2 and name[0] == "_":
        return obj.items[int(name[1]) - 1]
    if isinstance(obj, list) and name == "clone":
        return list(obj)
    return getattr(obj, name)(*args)


def _new(name, *args):
    if name in ("ArrayList", "LinkedList"):
        if args and isinstance(args[0], (JList,)):
            return JList(args[0].items)
        return JList()
    if name in ("Tuple2", "Tuple3", "Tuple4", "Tuple5"):
        return JTuple(*args)
    if name in ("HashMap", "Hashtable", "TreeMap"):
        return JMap()
    if name in ("StringBuilder", "StringBuffer"):
        return JStringBuilder(*args)
    if name == "String":
        return "".join(str(c) for c in args[0]) if args else ""
    if name == "Long":
        return _L(args[0])
    if name == "Integer":
        return _I(args[0])
    raise TypeError(f"java2py: new {name}")


class _Collections:
    @staticmethod
    def sort(lst):
        lst.items.sort(key=lambda x: x.v if isinstance(x, J) else x)


class _StringUtils:
    @staticmethod
    def chop(s):
        return s[:-1]


RUNTIME = dict(Collections=_Collections, StringUtils=_StringUtils, _cast_double=_cast_double, _L=_L, _I=_I, _C=_C, _cast_int=_cast_int, _cast_long=_cast_long, _cast_char=_cast_char, _box_long=_box_long,
               _box_int=_box_int, _ushr=_ushr, _eq=_eq, _add=_add, _len=_len, _newarr=_newarr, _iterate=_iterate,
               _instanceof=_instanceof, _call=_call, _new=_new, _JavaThrow=_JavaThrow, Long=_Long, Integer=_Integer,
               Math=_Math, RowFactory=_RowFactory, Arrays=_Arrays, JavaConverters=_JavaConverters, System=_Sys)


# ------------------------------------------------------------------------------------------------ driver
def find_class_span(src, cls_name):
    """text of `class <cls_name> ... { ... }` (brace matched on the token stream, comments / strings respected)"""
    m = re.search(r"\bclass\s+" + re.escape(cls_name) + r"\b", src)
    if not m:
        raise KeyError(cls_name)
    return m.start()


class Outer:
    """stands for the enclosing pipeline object: `param`, `info`, and `new Inner()`"""

    def __init__(self, param, classes):
        self.param = param
        self._classes = classes

        class _Info:
            def readMessage(self, *a): pass
            def screenDump(self, *a): pass
            def readParagraphedMessages(self, *a): pass
        self.info = _Info()

    def _new(self, name, *args):
        return self._classes[name](self)


def translate_classes(java_path, class_names, extra_outer_fields=("param", "info"), dump=None):
    """-> {class name: python class}; instantiate with Outer(param, classes)"""
    src = open(java_path, encoding="utf-8", errors="replace").read()
    classes = {}
    ns = dict(RUNTIME)
    for cn in class_names:
        start = find_class_span(src, cn)
        toks = lex(src[start:])
        p = Parser(toks, class_names=class_names)
        p.eat("class")
        p.eat(cn)
        while not p.at("{"):
            p.eat()
        p.eat("{")
        body = p.parse_class_body()
        p.eat("}")
        g = Gen(cn, body, outer_fields=extra_outer_fields, sibling_classes=class_names)
        code = g.gen_class()
        if dump is not None:
            dump[cn] = code
        exec(compile(code, f"<java2py:{cn}>", "exec"), ns)
        classes[cn] = ns[cn]
    return classes


def translate_plain_class(java_path, cls_name, dump=None):
    """a top-level class without an enclosing object (U/DefaultParam.java): -> python class taking no outer"""
    src = open(java_path, encoding="utf-8", errors="replace").read()
    start = find_class_span(src, cls_name)
    toks = lex(src[start:])
    p = Parser(toks, class_names=[cls_name])
    p.eat("class")
    p.eat(cls_name)
    while not p.at("{"):
        p.eat()
    p.eat("{")
    body = p.parse_class_body()
    g = Gen(cls_name, body)
    code = g.gen_class()
    if dump is not None:
        dump[cls_name] = code
    ns = dict(RUNTIME)
    exec(compile(code, f"<java2py:{cls_name}>", "exec"), ns)
    return ns[cls_name]
