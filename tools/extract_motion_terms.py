#!/usr/bin/env python3
"""Writes pbrt-v3_amd/data/motion_terms.bin: the coefficients of the motion-derivative function of AnimatedTransform
(core/transform.cpp:451-1099, c1[3] .. c5[3] = 15 DerivativeTerms of 4 polynomials each) as postfix programs over float.

The 60 polynomials are machine-generated symbolic data of the reference (the derivative of Translate(lerp T) * Slerp(R0, R1) * lerp S
applied to a point, in the 33 entries of the two decompositions and theta: book section 2.9.4); the bounds of a rotating shape
(AnimatedTransform::BoundPointMotion, transform.cpp:1226-1247) come out identical to the reference's only when every one of them is
evaluated with the reference's own association, so -- like the Sobol' matrices and the CIE tables -- they are extracted as data rather
than retyped: this script parses the C expressions of the reference's source into trees and writes them in postfix form;
host/motion_bounds.cpp runs them on a small float stack.  Runs in the build container only (needs /root/reference); the .bin is committed.

Layout (little endian): int32 magic 'MOTN', int32 nPrograms (60), int32 nConsts, float consts[nConsts]; then per program (order
c1[0].kc, .kx, .ky, .kz, c2[0].kc, ..., c5[0].kz, c1[1].kc, ... c5[2].kz -- i.e. component-major, term, coefficient) uint16 length and
`length` opcode bytes: 0 .. 33 push variable (VARS below), 64 + k push consts[k], 128 negate, 129 add, 130 subtract, 131 multiply."""
import os
import re
import struct
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = "/root/reference/src/core/transform.cpp"
OUT = os.path.join(ROOT, "pbrt-v3_amd", "data", "motion_terms.bin")
VARS = ["t0x", "t0y", "t0z", "t1x", "t1y", "t1z", "q0x", "q0y", "q0z", "q0w", "qperpx", "qperpy", "qperpz", "qperpw",
        "s000", "s001", "s002", "s010", "s011", "s012", "s020", "s021", "s022",
        "s100", "s101", "s102", "s110", "s111", "s112", "s120", "s121", "s122", "theta"]
OP_CONST, OP_NEG, OP_ADD, OP_SUB, OP_MUL = 64, 128, 129, 130, 131


class Parser:
    """C's grammar for + - * and unary minus over identifiers and numeric literals: left-associative, * binds tighter"""

    def __init__(self, text, consts):
        self.toks = re.findall(r"[A-Za-z_][A-Za-z_0-9]*|\d+\.?\d*|[-+*()]", text)
        assert "".join(self.toks) == re.sub(r"\s+", "", text), "unexpected character in " + text[:60]
        self.i = 0
        self.consts = consts

    def peek(self):
        return self.toks[self.i] if self.i < len(self.toks) else None

    def take(self):
        self.i += 1
        return self.toks[self.i - 1]

    def expr(self):
        code, isconst = self.term()
        while self.peek() in ("+", "-"):
            op = self.take()
            rhs, rc = self.term()
            assert not (isconst and rc), "integer arithmetic between two literals would not be float arithmetic"
            code = code + rhs + [OP_ADD if op == "+" else OP_SUB]
            isconst = False
        return code, isconst

    def term(self):
        code, isconst = self.unary()
        while self.peek() == "*":
            self.take()
            rhs, rc = self.unary()
            assert not (isconst and rc), "integer arithmetic between two literals would not be float arithmetic"
            code = code + rhs + [OP_MUL]
            isconst = False
        return code, isconst

    def unary(self):
        if self.peek() == "-":
            self.take()
            code, isconst = self.unary()
            if isconst:  # -1, -2: a negative literal (the conversion to float is exact either way)
                v = -self.consts[code[0] - OP_CONST]
                return [self.const(v)], True
            return code + [OP_NEG], False
        return self.primary()

    def const(self, v):
        if v not in self.consts:
            self.consts.append(v)
        assert len(self.consts) <= 64
        return OP_CONST + self.consts.index(v)

    def primary(self):
        t = self.take()
        if t == "(":
            code, isconst = self.expr()
            assert self.take() == ")"
            return code, isconst
        if t in VARS:
            return [VARS.index(t)], False
        v = float(t)
        assert v == int(v) and abs(v) < 2 ** 24, t
        return [self.const(v)], True


def split_args(body):
    args, depth, cur = [], 0, ""
    for ch in body:
        if ch == "(":
            depth += 1
        elif ch == ")":
            depth -= 1
        if ch == "," and depth == 0:
            args.append(cur)
            cur = ""
        else:
            cur += ch
    args.append(cur)
    return args


def main():
    if not os.path.exists(SRC):
        sys.exit("needs the reference's source (build container only)")
    text = open(SRC).read()
    consts = []
    programs = {}
    for m in re.finditer(r"\bc([1-5])\[([0-2])\]\s*=\s*DerivativeTerm\(", text):
        start = m.end()
        depth, j = 1, start
        while depth:
            depth += {"(": 1, ")": -1}.get(text[j], 0)
            j += 1
        args = split_args(text[start:j - 1])
        assert len(args) == 4 and text[j] == ";"
        for k, a in enumerate(args):
            p = Parser(a, consts)
            code, _ = p.expr()
            assert p.peek() is None
            programs[(int(m.group(2)), int(m.group(1)) - 1, k)] = code
    assert len(programs) == 60, len(programs)
    out = struct.pack("<iii", 0x4E544F4D, 60, len(consts)) + struct.pack("<%df" % len(consts), *consts)
    for c in range(3):
        for term in range(5):
            for k in range(4):
                code = programs[(c, term, k)]
                assert len(code) < 65536 and max(code) <= OP_MUL
                out += struct.pack("<H", len(code)) + bytes(code)
    open(OUT, "wb").write(out)
    print("%s: %d bytes, %d constants %s, longest program %d ops" % (OUT, len(out), len(consts), consts, max(len(p) for p in programs.values())))


if __name__ == "__main__":
    main()
