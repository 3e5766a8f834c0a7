"""
Generates tests/golden/shuosher_ssprk33.json by EXECUTING the reference's own pure-numpy function
``butcher_to_shuosher_form`` (thetis/rungekutta.py:13-87) on the SSPRK33 tableau (rungekutta.py:342-346).

``import thetis`` is impossible here (needs firedrake), so the function and the tableau class body are pulled out of
the reference file by AST and executed stand-alone.  Run in the build container only (needs /root/reference):

    python tests/golden/make_shuosher_golden.py
"""
import ast
import json
import os

import numpy

REF = '/root/reference/thetis/rungekutta.py'
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'shuosher_ssprk33.json')


def main():
    src = open(REF).read()
    tree = ast.parse(src)
    ns = {'numpy': numpy}
    tableau = {}
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name == 'butcher_to_shuosher_form':
            exec(compile(ast.Module(body=[node], type_ignores=[]), REF, 'exec'), ns)
        if isinstance(node, ast.ClassDef) and node.name == 'SSPRK33Abstract':
            for stmt in node.body:
                if isinstance(stmt, ast.Assign):
                    exec(compile(ast.Module(body=[stmt], type_ignores=[]), REF, 'exec'), tableau)
    a = numpy.array(tableau['a'], dtype=float)
    b = numpy.array(tableau['b'], dtype=float)
    alpha, beta = ns['butcher_to_shuosher_form'](a, b)
    data = {
        'source': 'thetis/rungekutta.py:13-87 executed on SSPRK33Abstract (rungekutta.py:342-346)',
        'a': a.tolist(), 'b': b.tolist(), 'c': list(tableau['c']), 'cfl_coeff': tableau['cfl_coeff'],
        'alpha': alpha.tolist(), 'beta': beta.tolist(),
        'alpha_hex': [[float(x).hex() for x in row] for row in alpha],
        'beta_hex': [[float(x).hex() for x in row] for row in beta],
    }
    with open(OUT, 'w') as f:
        json.dump(data, f, indent=1)
    print(json.dumps(data, indent=1))
    # every EXPLICIT tableau the reference defines (zero diagonal of a): pins a general re-implementation of the conversion
    allx = {}
    for node in tree.body:
        if isinstance(node, ast.ClassDef) and node.name.endswith('Abstract') and node.name != 'AbstractRKScheme':
            tab = {}
            for stmt in node.body:
                if isinstance(stmt, ast.Assign):
                    try:
                        exec(compile(ast.Module(body=[stmt], type_ignores=[]), REF, 'exec'), {'numpy': numpy, 'np': numpy}, tab)
                    except Exception:
                        pass
            if 'a' not in tab or 'b' not in tab:
                continue
            a = numpy.array(tab['a'], dtype=float)
            b = numpy.array(tab['b'], dtype=float)
            if a.ndim != 2 or numpy.diag(a).any():
                continue
            al, be = ns['butcher_to_shuosher_form'](a, b)
            allx[node.name] = {'a': a.tolist(), 'b': b.tolist(), 'alpha_hex': [[float(x).hex() for x in r] for r in al],
                               'beta_hex': [[float(x).hex() for x in r] for r in be]}
    with open(os.path.join(os.path.dirname(OUT), 'shuosher_explicit.json'), 'w') as f:
        json.dump({'source': 'thetis/rungekutta.py:13-87 executed on every explicit *Abstract tableau of the file', 'schemes': allx},
                  f, indent=1)
    print(sorted(allx))


if __name__ == '__main__':
    main()
