"""
Generates tests/golden/reference_surface.json from the reference's OWN source files - the host surface of the path pinned to data
taken from the reference rather than to this build's reading of it:

* ``options``: every option of ``CommonModelOptions``, ``ModelOptions2d`` and the explicit time-stepper option classes
  (thetis/options.py) with its trait type and DEFAULT value.  ``import thetis`` is impossible here (firedrake, traitlets), so the
  class bodies are read by AST: the default is the trait's first positional argument or ``default_value=``, evaluated as a literal
  (``Constant(x)`` -> x);
* ``paired``: the time-stepper names of the ``attach_paired_options`` decorators of ``ModelOptions2d`` and their defaults;
* ``field_metadata``: thetis/field_defs.py is plain Python and is EXECUTED; the 2D entries are kept;
* ``physical_constants``: thetis/physical_constants.py, ``Constant(x)`` -> x.

Run in the build container only (needs /root/reference):   python tests/golden/make_surface_golden.py
"""
import ast
import json
import os

REF = '/root/reference/thetis'
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'reference_surface.json')
CLASSES = ('TimeStepperOptions', 'ExplicitTimeStepperOptions', 'ExplicitTimeStepperOptions2d', 'ExplicitSWETimeStepperOptions2d',
           'ExplicitTracerTimeStepperOptions2d', 'TracerFieldOptions', 'CommonModelOptions', 'ModelOptions2d')


def literal(node):
    """literal value of a default expression; Constant(x) -> x; anything else -> its source text, tagged"""
    if isinstance(node, ast.Call) and getattr(node.func, 'id', None) == 'Constant' and node.args:
        return literal(node.args[0])
    try:
        return ast.literal_eval(node)
    except (ValueError, SyntaxError):
        return {'expr': ast.unparse(node)}


def trait_call(node):
    """the innermost call of ``Trait(...).tag(config=True)`` chains"""
    while isinstance(node, ast.Call) and isinstance(node.func, ast.Attribute):
        node = node.func.value
    return node if isinstance(node, ast.Call) else None


def main():
    tree = ast.parse(open(os.path.join(REF, 'options.py')).read())
    options, paired = {}, {}
    for node in tree.body:
        if not isinstance(node, ast.ClassDef) or node.name not in CLASSES:
            continue
        opts = {}
        for stmt in node.body:
            if not isinstance(stmt, ast.Assign) or len(stmt.targets) != 1 or not isinstance(stmt.targets[0], ast.Name):
                continue
            call = trait_call(stmt.value)
            if call is None or not isinstance(call.func, ast.Name):
                continue
            default = {'expr': 'no default'}
            kw = {k.arg: k.value for k in call.keywords}
            if 'default_value' in kw:
                default = literal(kw['default_value'])
            elif call.args:
                default = literal(call.args[0])
            opts[stmt.targets[0].id] = {'trait': call.func.id, 'default': default, 'allow_none': bool(literal(kw['allow_none'])) if 'allow_none' in kw else False}
        options[node.name] = {'bases': [ast.unparse(b) for b in node.bases], 'options': opts}
        if node.name == 'ModelOptions2d':
            for dec in node.decorator_list:
                if isinstance(dec, ast.Call) and getattr(dec.func, 'id', '') == 'attach_paired_options':
                    name = literal(dec.args[0])
                    enum = dec.args[1]
                    while isinstance(enum, ast.Call) and isinstance(enum.func, ast.Attribute):
                        enum = enum.func.value
                    pairs = [(literal(t.elts[0]), ast.unparse(t.elts[1])) for t in enum.args[0].elts]
                    kw = {k.arg: k.value for k in enum.keywords}
                    paired[name] = {'choices': pairs, 'slave': literal(enum.args[1]), 'default': literal(kw['default_value'])}
    ns = {}
    exec(compile(open(os.path.join(REF, 'field_defs.py')).read(), 'field_defs.py', 'exec'), ns)
    fields = {k: v for k, v in ns['field_metadata'].items() if k.endswith('_2d')}
    pc = {}
    for node in ast.parse(open(os.path.join(REF, 'physical_constants.py')).read()).body:
        if isinstance(node, ast.Assign) and isinstance(node.value, ast.Dict):
            for k, v in zip(node.value.keys, node.value.values):
                pc[literal(k)] = literal(v)
    data = {'source': 'thetis/options.py (AST: trait defaults), thetis/field_defs.py (executed), thetis/physical_constants.py (AST)',
            'options': options, 'paired': paired, 'field_metadata': fields, 'physical_constants': pc}
    with open(OUT, 'w') as f:
        json.dump(data, f, indent=1, sort_keys=True)
    print('{:d} classes, {:d} options, {:d} 2D fields'.format(len(options), sum(len(c['options']) for c in options.values()), len(fields)))


if __name__ == '__main__':
    main()
