#!/usr/bin/env python3
"""Reflow the prose of a markdown file to a column limit (tables, code fences, headings and list markers are kept: a table row is one line by
definition and stays as long as it is).   usage: python tools/reflow_md.py in.md out.md [columns=160]"""
import re
import sys
import textwrap


def reflow(text, width=160):
    out, para, fence = [], [], False

    def flush():
        if not para:
            return
        first = para[0]
        m = re.match(r"^(\s*(?:[*+-]|\d+\.)\s+)", first)
        indent = m.group(1) if m else re.match(r"^(\s*)", first).group(1)
        body = " ".join(l.strip() for l in para)
        if m:
            body = body[len(m.group(1).strip()) + 1:].lstrip() if body.startswith(m.group(1).strip()) else body
            out.extend(textwrap.wrap(body, width, initial_indent=indent, subsequent_indent=" " * len(indent), break_long_words=False, break_on_hyphens=False))
        else:
            out.extend(textwrap.wrap(body, width, initial_indent=indent, subsequent_indent=indent, break_long_words=False, break_on_hyphens=False))
        para.clear()

    for line in text.split("\n"):
        if line.lstrip().startswith("```"):
            flush(); fence = not fence; out.append(line); continue
        if fence or line.startswith("|") or line.startswith("#") or not line.strip():
            flush(); out.append(line); continue
        if re.match(r"^\s*(?:[*+-]|\d+\.)\s+", line):
            flush()
        para.append(line)
    flush()
    return "\n".join(out)


if __name__ == "__main__":
    w = int(sys.argv[3]) if len(sys.argv) > 3 else 160
    open(sys.argv[2], "w").write(reflow(open(sys.argv[1]).read(), w))
