"""Every `raise X("...")` of the reference modules the mirror covers, as (file, line, exception, message template) -- out of the syntax
tree, placeholders normalised to {} ("same ... error behaviour").  tests/golden/reference_error_messages.json;
tests/test_signatures.py requires each template in the mirror's source unless the raise is listed there as outside the hot path.
usage: python tests/golden/make_golden_error_messages.py"""
import ast
import json
import os
import re

ROOT = '/root/reference/openmmtools/'
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'reference_error_messages.json')
FILES = ['multistate/multistatesampler.py', 'multistate/replicaexchange.py', 'multistate/paralleltempering.py', 'multistate/sams.py',
         'mcmc.py', 'states.py', 'integrators.py', 'alchemy/alchemy.py', 'multistate/multistatereporter.py']


def literal(node):
    if isinstance(node, ast.Constant) and isinstance(node.value, str):
        return node.value
    if isinstance(node, ast.JoinedStr):
        return ''.join(v.value if isinstance(v, ast.Constant) else '{}' for v in node.values)
    if isinstance(node, ast.Call) and isinstance(node.func, ast.Attribute) and node.func.attr == 'format':
        return literal(node.func.value)
    if isinstance(node, ast.BinOp) and isinstance(node.op, ast.Add):
        l, r = literal(node.left), literal(node.right)
        return (l or '{}') + (r or '{}') if (l or r) else None
    if isinstance(node, ast.BinOp) and isinstance(node.op, ast.Mod):
        return literal(node.left)
    return None


def normalise(text):
    return re.sub(r'\s+', ' ', re.sub(r'\{[^}]*\}|%[sdfr]|%\.\d+f', '{}', text)).strip()


def messages(path):
    out = []
    for n in ast.walk(ast.parse(open(path).read())):
        if isinstance(n, ast.Raise) and isinstance(n.exc, ast.Call) and n.exc.args:
            t = literal(n.exc.args[0])
            if t and len(normalise(t)) >= 12:
                out.append(dict(line=n.lineno, exception=getattr(n.exc.func, 'id', getattr(n.exc.func, 'attr', '?')), template=normalise(t)))
    return sorted(out, key=lambda d: d['line'])


if __name__ == '__main__':
    out = {f: messages(ROOT + f) for f in FILES}
    with open(OUT, 'w') as fh:
        json.dump(out, fh, indent=1)
    print({f: len(v) for f, v in out.items()})
