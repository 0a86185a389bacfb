"""Implementation adapters for ``tests/golden/cases.py::run_case`` (oracle and HIP product)."""
import functools


class OracleImpl:
    """CPU restatement under ``oracle/`` (test infrastructure)."""
    accepts_init_eta = True

    def __init__(self):
        from oracle import cleverhans_cpu as o
        self.o = o

    def clip_eta(self, eta, norm, eps):
        return self.o.clip_eta(eta, norm, eps)

    def optimize_linear(self, grad, eps, norm):
        return self.o.optimize_linear(grad, eps, norm)

    def zero_out_clipped_grads(self, grad, x, clip_min, clip_max):
        return self.o.zero_out_clipped_grads(grad, x, clip_min, clip_max)

    def fgm(self, flavor):
        return functools.partial(self.o.fast_gradient_method, flavor=flavor)

    def pgd(self, flavor):
        return functools.partial(self.o.projected_gradient_descent, flavor=flavor)

    def fgm_vl(self, flavor):
        return functools.partial(self.o.fast_gradient_method_vl, flavor=flavor)

    def pgd_vl(self, flavor):
        return functools.partial(self.o.projected_gradient_descent_vl, flavor=flavor)


class ProductImpl:
    """The shipped HIP path, reached through the drop-in ``cleverhans`` packages."""
    accepts_init_eta = True

    def __init__(self):
        from vqattack_amd import dropin
        self._load = dropin.load

    def clip_eta(self, eta, norm, eps):
        return self._load("albef").utils.clip_eta(eta, norm, eps)

    def optimize_linear(self, grad, eps, norm):
        return self._load("albef").utils.optimize_linear(grad, eps, norm)

    def zero_out_clipped_grads(self, grad, x, clip_min, clip_max):
        return self._load("albef").utils.zero_out_clipped_grads(grad, x, clip_min, clip_max)

    def fgm(self, flavor):
        return self._load(flavor).fast_gradient_method.fast_gradient_method

    def pgd(self, flavor):
        return self._load(flavor).projected_gradient_descent.projected_gradient_descent

    def fgm_vl(self, flavor):
        return self._load(flavor).fast_gradient_method_vl.fast_gradient_method

    def pgd_vl(self, flavor):
        return self._load(flavor).projected_gradient_descent_vl.projected_gradient_descent


def reference_closures(flavor, model, batch):
    """The reference's own batch-1 ``model_fn`` members (``pgd_attack`` ... reading ``self.batch``, torch.stack / cat
    packed plain tensors, ``[0]`` indexing) over a bundled white box -- NOT the oracle's adapters: this is the form a
    user of the reference drives the drop-in with (``vqattack_amd/whitebox/reference_style.py``)."""
    from vqattack_amd.whitebox import reference_style
    cls = reference_style.VlmoReferenceClosures if flavor == "vlmo" else reference_style.AlbefReferenceClosures
    return cls(model, batch)
