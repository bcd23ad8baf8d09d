"""Re-wrap the prose of a markdown file to <= 118 columns (tables, headings and code blocks are left alone).  usage: python tools/wrap_md.py DESIGN.md"""
import re
import sys
import textwrap


def wrap(text, width=118):
    out, buf, in_code = [], [], False

    def flush():
        if not buf:
            return
        first = buf[0]
        m = re.match(r'^(\s*)([*-] |\d+\. )', first)
        if m:
            ind = m.group(1) + ' ' * len(m.group(2))
            body = [first[len(m.group(0)):]] + [l.strip() for l in buf[1:]]
            out.extend(textwrap.wrap(' '.join(b.strip() for b in body), width=width, initial_indent=m.group(0), subsequent_indent=ind,
                                     break_long_words=False, break_on_hyphens=False))
        else:
            ind = re.match(r'^(\s*)', first).group(1)
            out.extend(textwrap.wrap(' '.join(l.strip() for l in buf), width=width, initial_indent=ind, subsequent_indent=ind,
                                     break_long_words=False, break_on_hyphens=False))
        buf.clear()
    for l in text.split('\n'):
        if l.startswith('```'):
            flush(); in_code = not in_code; out.append(l); continue
        if in_code or l.startswith('|') or l.startswith('#') or l.strip() == '':
            flush(); out.append(l); continue
        if re.match(r'^\s*([*-] |\d+\. )', l):
            flush(); buf.append(l); continue
        buf.append(l)
    flush()
    return '\n'.join(out)


if __name__ == "__main__":
    for path in sys.argv[1:]:
        s = open(path).read()
        s = re.sub(r'([^\n])\n(#{2,3} )', r'\1\n\n\2', s)       # a blank line in front of every heading
        open(path, 'w').write(wrap(s).rstrip('\n') + '\n')
        long = [(n + 1, len(l)) for n, l in enumerate(open(path).read().split('\n')) if len(l) > 120 and not l.startswith('|')]
        print(path, "prose lines > 120:", long)
