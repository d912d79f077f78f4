#!/usr/bin/env python3
"""Re-wrap the prose of a Markdown file to a maximum line width (default 120) without touching tables, code fences, headings or
the structure of lists.  usage: reflow_md.py FILE [WIDTH]"""
import re
import sys
import textwrap

BULLET = re.compile(r"^(\s*)((?:[*\-+]|\d+\.|\([a-z0-9]\))\s+)?")


def reflow(text, width):
    out, para, in_code = [], [], False

    def flush():
        if not para:
            return
        m = BULLET.match(para[0])
        first = m.group(0)
        rest = " " * len(first)
        body = " ".join(line.strip() for line in para)
        body = body[len(first.strip()) + (1 if first.strip() else 0):] if first.strip() else body
        body = " ".join(body.split(" "))
        out.extend(textwrap.fill(body.strip(), width=width, initial_indent=first, subsequent_indent=rest,
                                 break_long_words=False, break_on_hyphens=False).split("\n"))
        para.clear()

    for line in text.split("\n"):
        stripped = line.strip()
        if stripped.startswith("```"):
            flush(); in_code = not in_code; out.append(line); continue
        if in_code or not stripped or stripped.startswith("|") or stripped.startswith("#") or stripped.startswith("{"):
            flush(); out.append(line); continue
        m = BULLET.match(line)
        if para and m.group(2):                     # a new list item ends the paragraph before it
            flush()
        elif para and len(m.group(1)) < len(BULLET.match(para[0]).group(1)):      # dedent: a new paragraph
            flush()
        para.append(line)
    flush()
    return "\n".join(out)


if __name__ == "__main__":
    path = sys.argv[1]
    width = int(sys.argv[2]) if len(sys.argv) > 2 else 120
    src = open(path).read()
    open(path, "w").write(reflow(src, width))
