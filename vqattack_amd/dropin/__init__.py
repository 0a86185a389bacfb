"""Drop-in ``cleverhans`` packages for the two attack drivers of the reference.

The reference's drivers do ``sys.path.append('../cleverhans')`` and then
``import cleverhans.torch.attacks.projected_gradient_descent as pgd`` / ``..._vl as pgd_vl``
(``ALBEF_attack/adv_attack.py:41-43``, ``vlmo/modules/vlmo_module.py:27-29``).  To switch a driver to the MI355X
path, point that ``sys.path`` entry at ``vqattack_amd/dropin/albef`` or ``vqattack_amd/dropin/vlmo`` instead
(see INTEGRATION.md).  Both directories hold a package named ``cleverhans`` with the reference's module layout;
they differ only in the ``flavor`` they bind (loss definition / ``y`` slicing of the respective copy).

``load(flavor)`` gives programmatic access without touching ``sys.path`` (used by tests and the orchestrators).
"""
import functools
import types

from .. import attacks, utils

_cache = {}


def load(flavor):
    """Namespace with the reference's module names as attributes, bound to ``flavor``."""
    if flavor not in (attacks.ALBEF, attacks.VLMO):
        raise ValueError("flavor must be 'albef' or 'vlmo'")
    ns = _cache.get(flavor)
    if ns is None:
        def mod(name, **fns):
            m = types.SimpleNamespace(__name__=name, **fns)
            return m
        bind = lambda f: functools.wraps(f)(functools.partial(f, flavor=flavor))  # noqa: E731
        ns = types.SimpleNamespace(
            utils=utils,
            fast_gradient_method=mod("fast_gradient_method",
                                     fast_gradient_method=bind(attacks.fast_gradient_method)),
            projected_gradient_descent=mod("projected_gradient_descent",
                                           projected_gradient_descent=bind(attacks.projected_gradient_descent)),
            fast_gradient_method_vl=mod("fast_gradient_method_vl",
                                        fast_gradient_method=bind(attacks.fast_gradient_method_vl)),
            projected_gradient_descent_vl=mod("projected_gradient_descent_vl",
                                              projected_gradient_descent=bind(attacks.projected_gradient_descent_vl)),
        )
        _cache[flavor] = ns
    return ns
