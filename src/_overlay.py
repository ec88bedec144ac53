"""Overlay plumbing for the repo-root ``src`` package.

The reference's scripts import everything through ``src.*`` (scripts/face_swap.py:15-27,
scripts/optimization.py:18-29, src/training/coach.py:20-30).  Only the hot-path modules are replaced by
this repo (`src.models.networks`, `src.models.stylegan2.model`, `src.models.stylegan2.op`,
`src.models.encoders.{psp_encoders,helpers}`, `src.utils.torch_utils.accumulate`); every other
``src.*`` module (options, datasets, criteria, pretrained models, morphology, alignment ...) must keep
resolving to the reference checkout.  The reference's ``src`` is a namespace package (no __init__.py), so
a regular package here would shadow ALL of it; instead each overlay package appends the matching
directory of ``$E4S_REFERENCE_ROOT/src`` to its ``__path__``: names that exist here win, everything else
falls through to the reference.  Without E4S_REFERENCE_ROOT (or /root/reference) only the replaced
modules are importable -- enough for this repo's own tests and bench.
"""
import os


def reference_src():
    root = os.environ.get("E4S_REFERENCE_ROOT")
    cands = [root] if root else ["/root/reference"]
    for c in cands:
        if c and os.path.isfile(os.path.join(c, "src", "models", "networks.py")):
            return os.path.join(c, "src")
    return None


def extend(path_list, rel):
    """Append <reference>/src/<rel> to a package's __path__ (idempotent)."""
    ref = reference_src()
    if ref is None:
        return
    d = os.path.join(ref, rel) if rel else ref
    if os.path.isdir(d) and d not in path_list:
        path_list.append(d)


def exec_reference_module(rel_file, namespace):
    """Run <reference>/src/<rel_file> inside `namespace` (a module's globals()); returns False when the reference (or
    one of its third-party imports) is unavailable.  Used by overlay modules that replace only a few names."""
    ref = reference_src()
    if ref is None:
        return False
    path = os.path.join(ref, rel_file)
    if not os.path.isfile(path):
        return False
    with open(path, "r", encoding="utf-8") as fh:
        code = compile(fh.read(), path, "exec")
    try:
        exec(code, namespace)
    except ImportError:
        return False
    return True
