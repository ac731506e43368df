"""Developer script: wrap source lines longer than LIMIT columns in jmcodec_amd/csrc.  Conservative and purely textual:
  1. a line that is only a `//` comment is re-flowed at word boundaries (same indent, same `// ` lead; tables / ASCII art are left alone);
  2. a code line with a trailing `//` comment gets the comment moved onto its own line(s) in front of it;
  3. a code line that is still too long and holds several statements is split after top-level `;` (outside parentheses, brackets, strings) --
     continuation statements keep the line's indent; `for (...)` headers are never split; lines inside macros (ending in backslash) are left alone.
  4. any other over-long code line is broken at a token boundary outside string literals -- by preference after `;` `{` `,` `&&` `||` `?` `:` `=` --
     and continued one indent level deeper (C++ does not care where white space falls); preprocessor lines are left alone.
What remains too long is listed for hand editing.  Usage: python tools/wrap_lines.py [--check] files..."""
import re
import sys

LIMIT = 160


def split_comment(line):
    """(code, comment) with `//` found outside string / char literals; comment is None when there is none."""
    in_s = None
    i = 0
    while i < len(line) - 1:
        c = line[i]
        if in_s:
            if c == "\\":
                i += 2
                continue
            if c == in_s:
                in_s = None
        elif c in "\"'":
            in_s = c
        elif c == "/" and line[i + 1] == "/":
            return line[:i].rstrip(), line[i:]
        i += 1
    return line, None


def reflow_comment(indent, text, lead="// "):
    words = text.split()
    out, cur = [], ""
    width = LIMIT - len(indent) - len(lead)
    for w in words:
        if cur and len(cur) + 1 + len(w) > width:
            out.append(indent + lead + cur)
            cur = w
        else:
            cur = (cur + " " + w) if cur else w
    if cur:
        out.append(indent + lead + cur)
    return out


def split_statements(code):
    """split after top-level ';' -- returns None when the line should not be touched"""
    indent = re.match(r"\s*", code).group(0)
    body = code[len(indent):]
    parts, depth, in_s, start, i = [], 0, None, 0, 0
    brace = 0
    while i < len(body):
        c = body[i]
        if in_s:
            if c == "\\":
                i += 2
                continue
            if c == in_s:
                in_s = None
        elif c in "\"'":
            in_s = c
        elif c in "([":
            depth += 1
        elif c in ")]":
            depth -= 1
        elif c == "{":
            brace += 1
        elif c == "}":
            brace -= 1
        elif c == ";" and depth == 0 and brace == 0:
            parts.append(body[start:i + 1].strip())
            start = i + 1
        i += 1
    tail = body[start:].strip()
    if tail:
        parts.append(tail)
    if len(parts) < 2 or depth != 0 or brace != 0:
        return None
    # greedy regroup so that each output line is as full as allowed
    out, cur = [], ""
    for p in parts:
        if cur and len(indent) + len(cur) + 1 + len(p) > LIMIT:
            out.append(indent + cur)
            cur = p
        else:
            cur = (cur + " " + p) if cur else p
    if cur:
        out.append(indent + cur)
    return out if all(len(x) <= LIMIT for x in out) else (out if len(out) > 1 else None)


def break_tokens(line):
    """break one over-long code line at token boundaries; returns a list of lines.  Inside parentheses / brackets only after `,` `&&` `||` (never
    inside a `for (;;)` header's clauses), outside them also after `;` `{` `}` and, failing all that, at any space."""
    indent = re.match(r"\s*", line).group(0)
    cont = indent + "    "
    out = []
    cur = line
    while len(cur) > LIMIT:
        in_s, best, best_rank, i, depth = None, -1, -1, 0, 0
        while i < min(len(cur), LIMIT):
            c = cur[i]
            if in_s:
                if c == "\\":
                    i += 2
                    continue
                if c == in_s:
                    in_s = None
            elif c in "\"'":
                in_s = c
            elif c in "([":
                depth += 1
            elif c in ")]":
                depth -= 1
            elif c == " " and i > len(indent) + 24:
                prev = cur[:i].rstrip()
                rank = 0
                if depth == 0:
                    if prev.endswith((";", "{", "}")):
                        rank = 6
                    elif prev.endswith(","):
                        rank = 4
                    elif prev.endswith(("&&", "||")):
                        rank = 4
                    elif prev.endswith(("?", ":", "=")) and not prev.endswith("::"):
                        rank = 2
                    else:
                        rank = 1
                else:
                    if prev.endswith(("&&", "||")):
                        rank = 3
                    elif prev.endswith(","):
                        rank = 3
                if rank and rank >= best_rank:
                    best, best_rank = i, rank
                elif rank and rank >= 3 and best < 80 and i > 120:
                    best, best_rank = i, rank
            i += 1
        if best < 0:
            break
        out.append(cur[:best].rstrip())
        cur = cont + cur[best:].lstrip()
    out.append(cur)
    return out


def process(path, check):
    src = open(path).read().split("\n")
    out, left = [], []
    for ln, line in enumerate(src, 1):
        if len(line) <= LIMIT or line.rstrip().endswith("\\"):
            out.append(line)
            continue
        indent = re.match(r"\s*", line).group(0)
        stripped = line.strip()
        if stripped.startswith("//"):
            text = stripped[2:].strip()
            if re.search(r"\|.*\||-{6,}|={6,}", text):          # tables, rulers
                out.append(line); left.append((ln, len(line)))
                continue
            out.extend(reflow_comment(indent, text))
            continue
        code, comment = split_comment(line)
        new = []
        if comment is not None and code.strip():
            new.extend(reflow_comment(indent, comment[2:].strip()))
            line2 = code
        else:
            line2 = line
        if len(line2) > LIMIT:
            parts = split_statements(line2)
            if parts:
                new.extend(parts)
            else:
                new.append(line2)
            if any(len(x) > LIMIT for x in new) and not stripped.startswith("#"):
                new2 = []
                for x in new:
                    new2.extend(break_tokens(x) if len(x) > LIMIT and not x.strip().startswith("//") else [x])
                new = new2
        else:
            new.append(line2)
        for x in new:
            if len(x) > LIMIT:
                left.append((ln, len(x)))
        out.extend(new)
    if not check:
        open(path, "w").write("\n".join(out))
    return left


if __name__ == "__main__":
    check = "--check" in sys.argv
    total = 0
    for p in [a for a in sys.argv[1:] if not a.startswith("--")]:
        left = process(p, check)
        total += len(left)
        for ln, n in left:
            print(f"{p}:{ln}: still {n} columns")
    print(total, "lines left over", LIMIT, "columns")
