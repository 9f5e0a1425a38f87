"""svol_amd — MI355X-native (gfx950) implementation of the SVOL hot path: the sketch<->video
cross-modal DETR head and the Hungarian / GIoU set-matching loss, behind the reference's
``build_model`` / ``build_loss`` / ``configs`` surface.

Importing the package is cheap and works without a GPU; the HIP kernel library
(``libsvol_hip.so``, built by ``python -m svol_amd.build``) is loaded on first use and there is
NO CPU fallback.
"""
__version__ = '0.1.0'


def install_as_lib():
    """Register this package under the reference's import paths (``lib.configs``,
    ``lib.modeling.model`` ...), so an SVOL-style ``train.py`` / ``test.py`` that does
    ``from lib.modeling.model import build_model`` picks up the MI355X build unchanged."""
    import importlib
    import sys
    import types
    root = sys.modules.setdefault('lib', types.ModuleType('lib'))
    root.__path__ = []
    for ref, ours in {
        'lib.configs': 'svol_amd.configs',
        'lib.modeling': 'svol_amd.modeling',
        'lib.modeling.model': 'svol_amd.modeling.model',
        'lib.modeling.svanet': 'svol_amd.modeling.svanet',
        'lib.modeling.cross_modal_transformer': 'svol_amd.modeling.cross_modal_transformer',
        'lib.modeling.position_encoding': 'svol_amd.modeling.position_encoding',
        'lib.modeling.transformer': 'svol_amd.modeling.transformer',
        'lib.modeling.svanet_variants': 'svol_amd.modeling.svanet_variants',
        'lib.modeling.sketch_detr': 'svol_amd.modeling.sketch_detr',
        'lib.modeling.matcher': 'svol_amd.modeling.matcher',
        'lib.modeling.loss': 'svol_amd.modeling.loss',
        'lib.utils': 'svol_amd.utils',
        'lib.utils.box_utils': 'svol_amd.utils.box_utils',
        'lib.evaluate': 'svol_amd.evaluate',
        'lib.evaluate.eval': 'svol_amd.evaluate.eval',
    }.items():
        sys.modules[ref] = importlib.import_module(ours)
