"""Execute pieces of the REFERENCE's orchestrators in the build container, straight from their source files.

``ALBEF_attack/adv_attack.py`` and ``vlmo/modules/vlmo_module.py`` cannot be imported here (tensorflow, tensorflow_hub,
nltk, timm, pytorch_lightning, sacred and real checkpoints are missing), but the methods on the hot path are
self-contained function bodies.  This module parses the reference file with ``ast``, picks whole methods (or a
line range of statements inside a method) and compiles exactly those nodes -- nothing is rewritten -- into a
namespace whose globals (``torch``, ``np``, ``copy``, ``F``, ``nn``, ``pgd``, ...) are the real libraries and whose
``self`` is a stub object the caller provides.  The results are stored as fixtures (data only): no reference source or
bytecode is ever written into the repository, and nothing here runs on the GPU box.

Used by ``tests/golden/make_text_golden.py``; build container only.
"""
import ast
import copy
import types

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

REF = "/root/reference"
ALBEF_ATTACK = REF + "/ALBEF_VQAttack/ALBEF_attack/adv_attack.py"
ALBEF_XBERT = REF + "/ALBEF_VQAttack/ALBEF_attack/models/xbert.py"
ALBEF_VQA_MODEL = REF + "/ALBEF_VQAttack/ALBEF_attack/models/model_vqa.py"
ALBEF_FILTER = REF + "/ALBEF_VQAttack/ALBEF_attack/filter_words.py"
VLMO_MODULE = REF + "/VLMO_VQAttack/vlmo/modules/vlmo_module.py"
VLMO_OBJECTIVES = REF + "/VLMO_VQAttack/vlmo/modules/objectives.py"

BASE_GLOBALS = dict(torch=torch, np=np, copy=copy, F=F, nn=nn, math=__import__("math"), os=__import__("os"),
                    json=__import__("json"))


def _tree(path):
    with open(path) as fh:
        return ast.parse(fh.read(), filename=path)


def _find_class(tree, name):
    for node in tree.body:
        if isinstance(node, ast.ClassDef) and node.name == name:
            return node
    raise KeyError(name)


def class_methods(path, class_name, names, extra_globals=None):
    """{name: function} for whole methods of ``class_name`` in the reference file, decorators dropped."""
    cls = _find_class(_tree(path), class_name)
    picked = []
    for node in cls.body:
        if isinstance(node, ast.FunctionDef) and node.name in names:
            node = copy.deepcopy(node)
            node.decorator_list = []
            picked.append(node)
    missing = set(names) - {n.name for n in picked}
    if missing:
        raise KeyError("methods not found in {}::{}: {}".format(path, class_name, sorted(missing)))
    ns = dict(BASE_GLOBALS)
    ns.update(extra_globals or {})
    exec(compile(ast.Module(body=picked, type_ignores=[]), path, "exec"), ns)
    return {n.name: ns[n.name] for n in picked}, ns


def module_items(path, names, extra_globals=None):
    """Top-level classes / functions of a reference module by name (e.g. ``BertEmbeddings`` of ``models/xbert.py``)."""
    tree = _tree(path)
    picked = [n for n in tree.body if isinstance(n, (ast.ClassDef, ast.FunctionDef)) and n.name in names]
    missing = set(names) - {n.name for n in picked}
    if missing:
        raise KeyError("not found in {}: {}".format(path, sorted(missing)))
    ns = dict(BASE_GLOBALS)
    ns.update(extra_globals or {})
    exec(compile(ast.Module(body=picked, type_ignores=[]), path, "exec"), ns)
    return {n.name: ns[n.name] for n in picked}


def _stmts_in_range(body, first, last):
    """Statements of a (nested) body whose line span lies inside [first, last]; descends into compound statements
    that straddle the range start (the reference's attack code sits inside ``for`` / ``if`` bodies)."""
    inside = [s for s in body if s.lineno >= first and s.end_lineno <= last]
    if inside:
        return inside
    for s in body:
        if s.lineno <= first and s.end_lineno >= last:
            for field in ("body", "orelse", "finalbody"):
                found = _stmts_in_range(getattr(s, field, []) or [], first, last)
                if found:
                    return found
    return []


def method_block(path, class_name, method, first, last, args, starts_with, extra_globals=None, returns=None):
    """Compile the statements on lines [first, last] of ``class_name.method`` into ``block(self, *args) -> dict``.

    ``starts_with``: text the first selected statement's source must start with (guards against picking the wrong
    range if the reference ever changes).  The block returns ``locals()`` (or just the names in ``returns``)."""
    with open(path) as fh:
        src = fh.read()
    cls = _find_class(ast.parse(src, filename=path), class_name)
    fn = next(n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name == method)
    stmts = _stmts_in_range(fn.body, first, last)
    if not stmts:
        raise KeyError("no statements on lines {}-{} of {}.{}".format(first, last, class_name, method))
    seg = ast.get_source_segment(src, stmts[0]) or ""
    if not seg.lstrip().startswith(starts_with):
        raise AssertionError("lines {}-{} of {} start with {!r}, expected {!r}".format(first, last, path, seg[:60],
                                                                                         starts_with))
    ret = ast.parse("return locals()").body[0] if returns is None else \
        ast.parse("return {" + ", ".join("'{0}': {0}".format(n) for n in returns) + "}").body[0]
    wrapper = ast.parse("def block(self, {}):\n    pass".format(", ".join(args))).body[0]
    wrapper.body = [copy.deepcopy(s) for s in stmts] + [ret]
    mod = ast.fix_missing_locations(ast.Module(body=[wrapper], type_ignores=[]))
    ns = dict(BASE_GLOBALS)
    ns.update(extra_globals or {})
    exec(compile(mod, path, "exec"), ns)
    return ns["block"]


def make_stub(methods, **attrs):
    """An object whose bound methods are the given reference functions and whose attributes are ``attrs``."""
    cls = type("ReferenceStub", (), dict(methods))
    obj = cls()
    for k, v in attrs.items():
        setattr(obj, k, v)
    return obj


class cpu_as_cuda:
    """The reference calls ``.cuda()`` / ``.get_device()`` on everything; in the GPU-less build container those become
    identity / 'cpu' for the duration of a run (patched on torch.Tensor, restored on exit)."""

    def __enter__(self):
        self._cuda, self._get = torch.Tensor.cuda, torch.Tensor.get_device
        torch.Tensor.cuda = lambda t, *a, **k: t
        torch.Tensor.get_device = lambda t: torch.device("cpu")
        return self

    def __exit__(self, *exc):
        torch.Tensor.cuda, torch.Tensor.get_device = self._cuda, self._get
        return False


def load_filter_words():
    """``filter_words`` as ``adv_attack.py:23-27`` builds it, minus NLTK's English stop-word list (nltk is not
    installed; the list actually used is stored in the fixture so oracle and product consume the same one)."""
    ns = {}
    with open(ALBEF_FILTER) as fh:
        exec(compile(fh.read(), ALBEF_FILTER, "exec"), ns)
    return list(ns["filter_words"]) + ["?", "."]


def namespace(**kw):
    return types.SimpleNamespace(**kw)
