"""svol_amd — MI355X-native (gfx950) implementation of the SVOL hot path: the sketch<->video
cross-modal DETR head and the Hungarian / GIoU set-matching loss, behind the reference's
``build_model`` / ``build_loss`` / ``configs`` surface.

Importing the package is cheap and works without a GPU; the HIP kernel library
(``libsvol_hip.so``, built by ``python -m svol_amd.build``) is loaded on first use and there is
NO CPU fallback.
"""
__version__ = '0.2.0'

# reference module -> module of this build.  ONLY the hot path (SURVEY.md §8 rows a / f) is replaced; every other
# ``lib.*`` module the reference's drivers import (train.py:24-33, test.py:24-34: lib.dataset.*, lib.utils.comm,
# lib.utils.misc, lib.utils.model_utils, lib.utils.logger, lib.utils.tensor_utils, lib.evaluate.utils ...) stays the
# reference's own file.
_OVERRIDES = {
    'lib.configs': 'svol_amd.configs',
    'lib.modeling': 'svol_amd.modeling',
    'lib.modeling.model': 'svol_amd.modeling.model',
    'lib.modeling.backbone': 'svol_amd.modeling.backbone',
    'lib.modeling.svanet': 'svol_amd.modeling.svanet',
    'lib.modeling.cross_modal_transformer': 'svol_amd.modeling.cross_modal_transformer',
    'lib.modeling.position_encoding': 'svol_amd.modeling.position_encoding',
    'lib.modeling.transformer': 'svol_amd.modeling.transformer',
    'lib.modeling.svanet_variants': 'svol_amd.modeling.svanet_variants',
    'lib.modeling.sketch_detr': 'svol_amd.modeling.sketch_detr',
    'lib.modeling.matcher': 'svol_amd.modeling.matcher',
    'lib.modeling.loss': 'svol_amd.modeling.loss',
    'lib.utils.box_utils': 'svol_amd.utils.box_utils',
    'lib.evaluate.eval': 'svol_amd.evaluate.eval',
}
# packages of the reference that keep their own files: imported for real when a real ``lib`` is on sys.path,
# synthesised (empty, no search path) when there is none
_KEPT_PACKAGES = ('lib', 'lib.utils', 'lib.evaluate')


def install_as_lib():
    """Register this build under the reference's import paths (``lib.configs``, ``lib.modeling.model`` ...), so an
    SVOL-style ``train.py`` / ``test.py`` that does ``from lib.modeling.model import build_model`` picks up the
    MI355X build unchanged.

    When the reference's own ``lib`` package is importable (the normal case: the two lines are added to the
    reference's train.py, whose directory is on ``sys.path``), it is imported for real and keeps its ``__path__``:
    only the hot-path submodules listed in ``_OVERRIDES`` are replaced, so ``lib.dataset.*``, ``lib.utils.comm`` /
    ``misc`` / ``model_utils`` / ``logger`` and ``lib.evaluate.utils`` still resolve to the reference's files.
    Without a real ``lib`` on the path a bare namespace is synthesised that holds the overrides only.
    Idempotent; returns the list of module names it bound."""
    import importlib
    import importlib.util
    import sys
    import types

    def _real_package(name):
        mod = sys.modules.get(name)
        if mod is not None and not getattr(mod, '_svol_amd_synthetic', False):
            return mod if hasattr(mod, '__path__') else None
        if mod is not None:
            return None
        try:
            spec = importlib.util.find_spec(name)
        except (ImportError, ValueError, AttributeError):
            spec = None
        if spec is None or spec.submodule_search_locations is None:
            return None
        return importlib.import_module(name)

    for pkg in _KEPT_PACKAGES:
        if _real_package(pkg) is None and pkg not in sys.modules:
            m = types.ModuleType(pkg)
            m.__path__ = []          # a package with nothing of its own to find
            m._svol_amd_synthetic = True
            sys.modules[pkg] = m
        parent, _, child = pkg.rpartition('.')
        if parent:
            setattr(sys.modules[parent], child, sys.modules[pkg])

    bound = []
    for ref, ours in _OVERRIDES.items():
        mod = importlib.import_module(ours)
        sys.modules[ref] = mod
        parent, _, child = ref.rpartition('.')
        setattr(sys.modules[parent], child, mod)   # ``import lib.modeling.model as m`` walks attributes
        bound.append(ref)
    return bound
