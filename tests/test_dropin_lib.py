"""install_as_lib() must coexist with the reference's own ``lib`` package: the reference
drivers import lib.dataset.*, lib.utils.{comm,misc,model_utils,logger} next to the hot-path modules
(train.py:24-33, test.py:24-34).  The tree built here is a STUB written by this test — not the reference's
files — whose hot-path modules raise on import, so a passing run proves they were overridden and that
everything else still resolves to the tree on sys.path."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

BOOM = "raise ImportError('hot-path module of the stub tree was imported: install_as_lib() did not override it')\n"


def _write(path, text=''):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, 'w') as f:
        f.write(text)


def _stub_tree(base):
    lib = os.path.join(base, 'lib')
    for pkg in ('', 'dataset', 'dataset/sampler', 'utils', 'evaluate', 'modeling'):
        _write(os.path.join(lib, pkg, '__init__.py'))
    # modules the drivers import that are NOT on the hot path: they must keep working, including their own
    # intra-package imports of an overridden module (svol_dataset -> lib.utils.box_utils)
    _write(os.path.join(lib, 'dataset', 'svol_dataset.py'), textwrap.dedent('''
        from lib.utils.tensor_utils import pad_sequences_1d
        from lib.utils.box_utils import box_xyxy_to_cxcywh
        def prepare_batch_inputs(batch):
            return {'stub': 'prepare_batch_inputs', 'pad': pad_sequences_1d(), 'box_fn': box_xyxy_to_cxcywh.__module__}
    '''))
    _write(os.path.join(lib, 'dataset', 'svol_dataloader.py'), textwrap.dedent('''
        from lib.dataset.svol_dataset import prepare_batch_inputs
        from lib.dataset.sampler import STUB_SAMPLER
        from lib.utils.comm import get_world_size, get_rank
        def build_dataloader(*a, **k):
            return ('stub-loader', get_world_size(), get_rank(), STUB_SAMPLER)
    '''))
    _write(os.path.join(lib, 'dataset', 'sampler', '__init__.py'), "STUB_SAMPLER = 'stub-sampler'\n")
    _write(os.path.join(lib, 'utils', 'comm.py'), textwrap.dedent('''
        def get_rank(): return 0
        def get_world_size(): return 1
        def reduce_tensor(t): return t
    '''))
    _write(os.path.join(lib, 'utils', 'misc.py'), textwrap.dedent('''
        def cur_time(): return 'now'
        def save_jsonl(*a): pass
        def save_json(*a): pass
        class AverageMeter: pass
    '''))
    _write(os.path.join(lib, 'utils', 'model_utils.py'), "def count_parameters(m): return 7\n")
    _write(os.path.join(lib, 'utils', 'logger.py'), "def setup_logger(*a, **k): return 'stub-logger'\n")
    _write(os.path.join(lib, 'utils', 'tensor_utils.py'), "def pad_sequences_1d(*a, **k): return 'stub-pad'\n")
    _write(os.path.join(lib, 'evaluate', 'utils.py'), "STUB_EVAL_UTILS = 1\n")
    # hot-path modules: importing the stub's copy is an error
    for rel in ('configs.py', 'utils/box_utils.py', 'evaluate/eval.py', 'modeling/model.py', 'modeling/loss.py',
                'modeling/matcher.py', 'modeling/svanet.py', 'modeling/cross_modal_transformer.py',
                'modeling/position_encoding.py', 'modeling/transformer.py', 'modeling/backbone.py',
                'modeling/sketch_detr.py', 'modeling/svanet_variants.py'):
        _write(os.path.join(lib, rel), BOOM)
    return lib


DRIVER = textwrap.dedent('''
    import sys
    sys.argv = ['train.py', '--num_layers', '6', '--matcher', 'video_matcher', '--num_queries', '100']
    import svol_amd; svol_amd.install_as_lib()
    # what train.py:24-33 and test.py:24-34 import from lib.*
    from lib.modeling.model import build_model
    from lib.modeling.loss import build_loss
    from lib.dataset.svol_dataset import prepare_batch_inputs
    from lib.dataset.svol_dataloader import build_dataloader
    from lib.utils.comm import get_rank, get_world_size, reduce_tensor
    from lib.utils.misc import cur_time, save_jsonl, save_json, AverageMeter
    from lib.utils.model_utils import count_parameters
    from lib.utils.logger import setup_logger
    from lib.utils.box_utils import box_cxcywh_to_xyxy
    from lib.evaluate.eval import eval_results
    from lib.configs import args
    import lib.evaluate.utils, lib.modeling.matcher, lib.modeling.svanet
    import lib
    assert build_model.__module__ == 'svol_amd.modeling.model', build_model.__module__
    assert build_loss.__module__ == 'svol_amd.modeling.loss'
    assert box_cxcywh_to_xyxy.__module__ == 'svol_amd.utils.box_utils'
    assert eval_results.__module__ == 'svol_amd.evaluate.eval'
    assert args.num_layers == 6 and args.matcher == 'video_matcher' and args.num_queries == 100
    assert lib.__file__.startswith(sys.argv_stub_base), lib.__file__      # the REAL package, not a synthetic one
    assert lib.utils.comm.__file__.startswith(sys.argv_stub_base)
    assert prepare_batch_inputs(None) == {'stub': 'prepare_batch_inputs', 'pad': 'stub-pad',
                                          'box_fn': 'svol_amd.utils.box_utils'}
    assert build_dataloader() == ('stub-loader', 1, 0, 'stub-sampler')
    assert count_parameters(None) == 7 and setup_logger() == 'stub-logger' and cur_time() == 'now'
    assert lib.evaluate.utils.STUB_EVAL_UTILS == 1
    assert lib.modeling.model is sys.modules['svol_amd.modeling.model']
    assert lib.utils.box_utils is sys.modules['svol_amd.utils.box_utils']
    svol_amd.install_as_lib()                                              # idempotent
    assert sys.modules['lib'].__file__.startswith(sys.argv_stub_base)
    print('DROPIN-OK')
''')


def _run(code, cwd, extra_path):
    env = dict(os.environ)
    env['PYTHONPATH'] = os.pathsep.join([extra_path, ROOT] + ([env['PYTHONPATH']] if env.get('PYTHONPATH') else []))
    return subprocess.run([sys.executable, '-c', code], cwd=cwd, env=env, capture_output=True, text=True, timeout=300)


def test_install_as_lib_keeps_the_reference_packages_importable(tmp_path):
    base = str(tmp_path)
    _stub_tree(base)
    code = f'import sys; sys.argv_stub_base = {base!r}\n' + DRIVER
    r = _run(code, base, base)
    assert r.returncode == 0 and 'DROPIN-OK' in r.stdout, r.stdout + r.stderr


def test_install_as_lib_without_a_reference_tree(tmp_path):
    """no ``lib`` on the path: the overrides live in a synthesised namespace; anything else raises ModuleNotFoundError."""
    code = textwrap.dedent('''
        import sys
        sys.argv = ['x']
        import svol_amd; svol_amd.install_as_lib()
        from lib.modeling.model import build_model
        from lib.modeling.loss import build_loss
        from lib.utils.box_utils import box_cxcywh_to_xyxy
        from lib.configs import args
        assert args.hidden_dim == 256
        try:
            import lib.utils.comm
        except ModuleNotFoundError:
            print('DROPIN-OK')
    ''')
    r = _run(code, str(tmp_path), str(tmp_path))
    assert r.returncode == 0 and 'DROPIN-OK' in r.stdout, r.stdout + r.stderr
