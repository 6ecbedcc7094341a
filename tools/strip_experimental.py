"""Remove the EXPERIMENTAL-only blocks from a kernel source before it is embedded into a RELEASE libelemhip.so
(elementary_amd/csrc/Makefile -> build/spec_text.inc): every measurement hook — several render wrong samples by design — sits
behind `#if defined(ELEMHIP_EXPERIMENTAL) && ...` / `#elif defined(ELEMHIP_EXPERIMENTAL) && ...` / `#ifdef ELEMHIP_EXPERIMENTAL`,
and a release library's run-time compiler never sees their text (tests/test_host_logic.py greps the library for it).
Usage: python tools/strip_experimental.py file > stripped"""
import re
import sys

FALSE_IF = re.compile(r'^\s*#\s*(if\s+defined\(ELEMHIP_EXPERIMENTAL\)\s*&&|ifdef\s+ELEMHIP_EXPERIMENTAL\b)')
FALSE_ELIF = re.compile(r'^\s*#\s*elif\s+defined\(ELEMHIP_EXPERIMENTAL\)\s*&&')
ANY_IF = re.compile(r'^\s*#\s*(if|ifdef|ifndef)\b')
ELIF = re.compile(r'^\s*#\s*elif\b')
ELSE = re.compile(r'^\s*#\s*else\b')
ENDIF = re.compile(r'^\s*#\s*endif\b')


def strip(lines):
    out = []
    # stack entries: ['ours', state] with state in {'skip' (a false experimental branch), 'keep' (the #else of one), 'pass'
    # (an #elif chain that turned into an ordinary conditional)} or ['other']
    stack = []

    def emitting():
        return all(not (e[0] == 'ours' and e[1] == 'skip') for e in stack)
    for line in lines:
        if FALSE_IF.match(line):
            stack.append(['ours', 'skip'])
            continue
        if ANY_IF.match(line):
            stack.append(['other'])
            if emitting():
                out.append(line)
            continue
        if stack and stack[-1][0] == 'ours':
            top = stack[-1]
            if FALSE_ELIF.match(line):
                if top[1] != 'pass':
                    top[1] = 'skip'
                    continue
            elif ELIF.match(line):
                if top[1] == 'skip':          # the first live branch of the chain opens an ordinary conditional
                    top[1] = 'pass'
                    stack.pop(); stack.append(['ours', 'pass'])
                    if emitting():
                        out.append(re.sub(r'#(\s*)elif', r'#\1if', line, count=1))
                    continue
            elif ELSE.match(line):
                if top[1] == 'skip':
                    top[1] = 'keep'
                    continue
            elif ENDIF.match(line):
                kind = top[1]
                stack.pop()
                if kind == 'pass' and emitting():
                    out.append(line)
                continue
        elif stack and ENDIF.match(line):
            stack.pop()
            if emitting():
                out.append(line)
            continue
        if emitting():
            out.append(line)
    assert not stack, "unbalanced conditionals"
    return out


if __name__ == "__main__":
    sys.stdout.write("".join(strip(open(sys.argv[1]).readlines())))
