"""Developer script: keep Python sources within 160 columns.  Lines are only broken INSIDE brackets (after a comma, or by splitting a string literal at a
space into two adjacent literals), so the program is unchanged; the script proves it by comparing the syntax tree before and after and refuses to write
the file otherwise.

    python3 tools/wrap_py.py bench.py tests/*.py
"""
import ast
import sys

LIMIT = 160


def scan(line, depth, in_triple):
    """walk one physical line; returns (events, depth, in_triple) where events are (col, kind, info):
    kind 'comma' = a break is allowed after this column (inside brackets, outside strings), 'space' = a space inside a one-line string literal
    (info = (prefix, quote)) outside any f-string replacement field."""
    ev, i, n = [], 0, len(line)
    while i < n:
        if in_triple:
            j = line.find(in_triple, i)
            if j < 0:
                return ev, depth, in_triple
            i, in_triple = j + 3, None
            continue
        c = line[i]
        if c == "#":
            break
        if c in "([{":
            depth += 1
        elif c in ")]}":
            depth -= 1
        elif c in "\"'":
            # prefix letters before the quote
            k = i
            while k > 0 and line[k - 1].isalpha():
                k -= 1
            prefix = line[k:i]
            if line[i:i + 3] in ('"""', "'''"):
                in_triple = line[i:i + 3]
                i += 3
                continue
            q, j, braces = c, i + 1, 0
            is_f = "f" in prefix.lower()
            raw = "r" in prefix.lower()
            while j < n and (line[j] != q or braces > 0 and False):
                ch = line[j]
                if ch == "\\" and not raw:
                    j += 2
                    continue
                if is_f and ch == "{":
                    if line[j + 1:j + 2] == "{":
                        j += 2
                        continue
                    braces += 1
                elif is_f and ch == "}":
                    if braces == 0 and line[j + 1:j + 2] == "}":
                        j += 2
                        continue
                    braces = max(0, braces - 1)
                elif ch == " " and braces == 0 and depth > 0 and "b" not in prefix.lower():
                    ev.append((j, "space", (prefix, q)))
                j += 1
            i = j + 1
            continue
        elif c == "," and depth > 0 and line[i + 1:i + 2] == " ":
            ev.append((i + 1, "comma", None))
        i += 1
    return ev, depth, in_triple


def wrap_line(line, depth0):
    """break one over-long line; depth0 = bracket depth at its start"""
    out = []
    indent = line[:len(line) - len(line.lstrip())]
    cont = indent + ("    " if depth0 == 0 else "")
    cur, depth = line, depth0
    guard = 0
    while len(cur) > LIMIT and guard < 50:
        guard += 1
        ev, _, _ = scan(cur, depth, None)
        lo = len(cur) - len(cur.lstrip()) + 30
        commas = [e for e in ev if e[1] == "comma" and lo < e[0] < LIMIT - 1]
        spaces = [e for e in ev if e[1] == "space" and lo < e[0] < LIMIT - 2]
        pick = None
        if commas and (not spaces or commas[-1][0] >= LIMIT - 50 or commas[-1][0] >= spaces[-1][0] - 30):
            pick = commas[-1]
        elif spaces:
            pick = spaces[-1]
        elif commas:
            pick = commas[-1]
        if pick is None:
            break
        col, kind, info = pick
        if kind == "comma":
            head, tail = cur[:col], cur[col:].lstrip()
        else:
            prefix, q = info
            head, tail = cur[:col + 1] + q, prefix + q + cur[col + 1:]
        _, depth, _ = scan(head, depth, None)
        out.append(head.rstrip() if kind == "comma" else head)
        cur = cont + tail
    out.append(cur)
    return out


def process(path):
    src = open(path).read()
    lines = src.split("\n")
    out, depth, in_triple, changed = [], 0, None, False
    for line in lines:
        d0, t0 = depth, in_triple
        _, depth, in_triple = scan(line, depth, in_triple)
        if len(line) > LIMIT and t0 is None and in_triple is None and not line.lstrip().startswith("#"):
            got = wrap_line(line, d0)
            if len(got) > 1:
                changed = True
            out.extend(got)
        else:
            out.append(line)
    new = "\n".join(out)
    left = sum(1 for l in out if len(l) > LIMIT)
    if changed:
        if ast.dump(ast.parse(src)) != ast.dump(ast.parse(new)):
            print(f"{path}: the wrapped file parses differently -- left alone")
            return sum(1 for l in lines if len(l) > LIMIT)
        open(path, "w").write(new)
    for n, l in enumerate(out, 1):
        if len(l) > LIMIT:
            print(f"{path}:{n}: still {len(l)} columns")
    return left


if __name__ == "__main__":
    total = sum(process(p) for p in sys.argv[1:])
    print(f"{total} lines left over {LIMIT} columns")
