#!/usr/bin/env python3
"""Re-wrap the prose of a Markdown file at 118 columns (tables, headings, code fences and blank lines are left alone; bullet
items keep their marker and a two-space hanging indent).  usage: tools/wrap_md.py FILE [width]"""
import re
import sys
import textwrap


def main(path, width=118):
    out, para, indent, first = [], [], "", ""
    fence = False

    def flush():
        nonlocal para, indent, first
        if para:
            text = " ".join(s.strip() for s in para)
            out.extend(textwrap.wrap(text, width=width, initial_indent=first, subsequent_indent=indent,
                                     break_long_words=False, break_on_hyphens=False))
            para = []

    for line in open(path).read().split("\n"):
        if line.strip().startswith("```"):
            flush()
            fence = not fence
            out.append(line)
            continue
        if fence or not line.strip() or line.lstrip().startswith(("|", "#")) or line.strip() == "---":
            flush()
            out.append(line)
            continue
        m = re.match(r"^(\s*)([*-]|\d+\.)\s+", line)
        if m:                                       # a new list item
            flush()
            first = m.group(0)
            indent = " " * len(m.group(0))
            para = [line[len(m.group(0)):]]
        elif para and line.startswith(" ") and indent:   # continuation of a list item
            para.append(line)
        elif para and not line.startswith(" ") and indent:
            flush()
            first = indent = ""
            para = [line]
        else:
            if not para:
                first = indent = ""
            para.append(line)
    flush()
    open(path, "w").write("\n".join(out))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 118)
