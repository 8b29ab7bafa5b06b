#!/usr/bin/env python3
"""tools/unifdef.py FILE MACRO...: rewrite FILE as if every MACRO were undefined (#ifdef M / #ifndef M / #else / #endif
blocks only, no nesting of the named macros inside each other).  Used once to take the exploration switches out of the
shipped kernels; kept because the exploration builds of a later round will want to do the same."""
import re
import sys


def strip(text, macros):
    out, stack = [], []          # stack entries: None (foreign conditional) or [keeping: bool]
    for line in text.split("\n"):
        m = re.match(r"\s*#\s*(ifdef|ifndef)\s+(\w+)", line)
        if m:
            if m.group(2) in macros:
                stack.append([m.group(1) == "ifndef"])
                continue
            stack.append(None)
        elif re.match(r"\s*#\s*if\b", line):
            stack.append(None)
        elif re.match(r"\s*#\s*else\b", line) and stack and stack[-1] is not None:
            stack[-1][0] = not stack[-1][0]
            continue
        elif re.match(r"\s*#\s*endif\b", line):
            top = stack.pop()
            if top is not None:
                continue
        if all(s is None or s[0] for s in stack):
            out.append(line)
    return "\n".join(out)


if __name__ == "__main__":
    path, macros = sys.argv[1], set(sys.argv[2:])
    src = open(path).read()
    open(path, "w").write(strip(src, macros))
