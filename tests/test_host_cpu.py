"""CPU-side checks (no GPU): the C-ABI library loads and exports every declared symbol, the
module tree matches the reference's state_dict keys, the cfg table, codebook sampling and surgery,
and the product path refuses CPU tensors instead of falling back."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

import golden_util as gu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from mcgen_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib = _lib.load()                                   # binds every name in SYMBOLS (AttributeError if missing)
    header = open(os.path.join(ROOT, 'include', 'mcgen_hip.h')).read()
    declared = set(re.findall(r'\b(mcgen_[a-z0-9_]+)\s*\(', header))
    assert declared, 'no declarations found'
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(raw, name), name
    assert lib.mcgen_abi_version() == 9
    # struct sizes agree with the C side: weight image size query is pure host code
    assert lib.mcgen_weight_image_elems(128, 3, 3, 0) == 1 * 9 * 128 * 32
    assert lib.mcgen_weight_image_elems(128, 3, 3, 1) == 4 * 9 * 16 * 32
    assert lib.mcgen_weight_image_elems(4096, 128, 1, 0) == 4 * 4096 * 32


def test_graft_entry_build_runs_clean():
    """The driver's "does it build" entry point: compiles what changed (nothing, normally) and checks the ABI version it
    expects against the library -- this is the call that breaks when the version is bumped in one place only."""
    import __graft_entry__
    __graft_entry__.build()


def test_bad_arguments_are_rejected_before_launch():
    from mcgen_amd import _lib
    lib = _lib.load()
    p = _lib.Conv()
    p.nseg = 3
    assert lib.mcgen_conv_fused(ctypes.byref(p), _lib.F32, None) != 0
    assert b'nseg' in lib.mcgen_last_error()
    p.nseg = 1; p.N = 1; p.H = 6; p.W = 6
    assert lib.mcgen_conv_fused(ctypes.byref(p), _lib.F32, None) != 0
    assert b'powers of two' in lib.mcgen_last_error()


def _cfg_for(data_name, classes=None):
    from mcgen_amd.config import cfg, process_control
    cfg.update(data_name=data_name, model_name='mcgan', device='cpu')
    cfg['control'] = {'controller_rate': '0.5'}
    cfg.pop('classes_size', None)
    process_control()
    if classes:
        cfg['classes_size'] = classes
    return cfg


def test_process_control_table():
    cfg = _cfg_for('CIFAR10')
    assert cfg['data_shape'] == [3, 32, 32] and cfg['classes_size'] == 10
    assert cfg['gan']['generator_hidden_size'] == [256] * 4 and cfg['gan']['discriminator_hidden_size'] == [128] * 4
    assert cfg['gan']['latent_size'] == 128 and cfg['batch_size'] == {'train': 128, 'test': 512}
    assert cfg['controller_rate'] == 0.5
    cfg = _cfg_for('COIL100')
    assert cfg['gan']['generator_hidden_size'] == [512, 256, 128, 64] and cfg['classes_size'] == 100
    cfg = _cfg_for('Omniglot')
    assert cfg['data_shape'] == [1, 32, 32] and cfg['classes_size'] == 1623
    from mcgen_amd.config import cfg as c2, process_control
    c2['data_name'] = 'nope'
    with pytest.raises(ValueError):
        process_control()
    _cfg_for('CIFAR10')


@pytest.mark.parametrize('fixture,g,d,classes,name', [
    ('mcgan_small.npz', [32] * 4, [16] * 4, 10, 'CIFAR10'),
    ('mcgan_coil_small.npz', [64, 32, 16, 8], [8, 16, 32, 64], 20, 'COIL100'),
])
def test_module_tree_loads_reference_state_dict(fixture, g, d, classes, name):
    from mcgen_amd import models
    cfg = _cfg_for(name, classes)
    cfg['gan']['generator_hidden_size'], cfg['gan']['discriminator_hidden_size'] = g, d
    m = models.mcgan()
    sd = gu.state_from_npz(gu.load_npz(fixture))
    assert set(m.state_dict().keys()) == set(sd.keys())
    m.load_state_dict(sd)
    assert gu.mcgan_shapes(g, d, classes, cifar_layout=(name == 'CIFAR10')) == {k: tuple(v.shape) for k, v in sd.items()}
    # the shared mc_1 instance shows up under aliased keys
    blk = m.generator.blocks[0]
    assert blk.conv[3] is blk.mc_1 and blk.shortcut[1] is blk.mc_1 and blk.conv[7] is blk.mc_2
    n_params = sum(p.numel() for p in m.parameters())
    assert n_params == sum(int(np.prod(v.shape)) for k, v in sd.items() if k.endswith(('.weight', '.bias', '.weight_orig')))
    _cfg_for('CIFAR10')


def test_full_size_parameter_count():
    from mcgen_amd import models
    _cfg_for('CIFAR10')
    m = models.mcgan()
    assert sum(p.numel() for p in m.parameters()) == 5330564
    assert sum(p.numel() for p in m.generator.parameters()) == 4276739
    # init_param: xavier on the SN-wrapped weights too (it runs after make_SpectralNormalization)
    w = m.discriminator.blocks[1].conv[2].module.weight_orig
    assert abs(float(w.abs().max()) - (6 / 2304) ** 0.5) < 1e-3


def test_codebook_sampling_and_surgery():
    from mcgen_amd.modules import MultimodalController, sample_codebook
    from mcgen_amd.models import utils as mu
    from mcgen_amd.config import cfg
    torch.manual_seed(0)
    cb = sample_codebook(100, 24, 0.5)
    assert cb.shape == (100, 24) and set(cb.unique().tolist()) <= {0.0, 1.0}
    assert len({tuple(r) for r in cb.tolist()}) == 100
    assert torch.equal(sample_codebook(7, 5, 1), torch.ones(7, 5))
    mc = MultimodalController(16, 10, 0.5)
    assert 'codebook' in dict(mc.named_buffers()) and mc.codebook.shape == (10, 16)
    cfg['device'] = 'cpu'
    orig = mc.codebook.clone()
    out = mu.transit_codebook(mc.codebook, root=3, alpha=0.25)
    cross = int(round(0.75 * 16))
    assert torch.equal(out[3], mc.codebook[3])
    assert all(torch.equal(out[i, :cross], mc.codebook[3, :cross]) for i in range(10))
    assert all(torch.equal(out[i, cross:], mc.codebook[i, cross:]) for i in range(10))
    holder = torch.nn.Sequential(mc)
    mu.transit(holder, 3, 0.25)
    assert torch.equal(mc.codebook_orig, orig) and torch.equal(mc.codebook, out)
    cfg['classes_size'] = 6
    mu.create(holder)
    assert mc.codebook.shape == (6, 16)
    cfg['classes_size'] = 10


def test_no_cpu_fallback():
    from mcgen_amd import _lib, models, ops
    cfg = _cfg_for('CIFAR10')
    cfg['gan']['generator_hidden_size'], cfg['gan']['discriminator_hidden_size'] = [16] * 4, [16] * 4
    m = models.mcgan()
    with pytest.raises(_lib.McgenError):
        m.generate(torch.zeros(2, dtype=torch.int64), torch.zeros(2, 128))
    with pytest.raises(_lib.McgenError):
        ops.mc_code(torch.zeros(2, 10), torch.zeros(10, 16))
    _cfg_for('CIFAR10')


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, 'multimodal-controller-for-generative-models_amd')
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', src, re.M), os.path.join(dp, f)
    # tools/ is not test infrastructure either: scripts that run the oracle live under tests/diag/
    for f in os.listdir(os.path.join(ROOT, 'tools')):
        if f.endswith('.py'):
            src = open(os.path.join(ROOT, 'tools', f)).read()
            assert not re.search(r'^\s*(from|import)\s+oracle\b', src, re.M), f


@pytest.mark.parametrize('fixture, model_name, extra', [
    ('mcglow_small.npz', 'mcglow', {'classes_size': 12, 'data_shape': [1, 32, 32],
                                    'glow': {'hidden_size': 32, 'K': 2, 'L': 3, 'affine': True, 'conv_lu': True}}),
    ('mcpixelcnn_small.npz', 'mcpixelcnn', {'classes_size': 10, 'pixelcnn': {'num_layer': 4, 'hidden_size': 16, 'num_embedding': 32}}),
    ('mcvae_small.npz', 'mcvae', {'classes_size': 10, 'data_shape': [3, 32, 32],
                                  'vae': {'hidden_size': [8, 16, 32], 'latent_size': 16, 'num_res_block': 2, 'embedding_size': 32}}),
    ('vqvae_small.npz', 'vqvae', {'data_shape': [3, 32, 32],
                                  'vqvae': {'hidden_size': [16, 16], 'num_res_block': 2, 'embedding_size': 8, 'num_embedding': 64,
                                            'vq_commit': 0.25}}),
])
def test_other_model_trees_load_reference_state_dicts(fixture, model_name, extra):
    """MCGlow / MCPixelCNN / MCVAE / VQ-VAE module trees carry exactly the reference's state_dict keys and shapes
    (the fixtures store state dicts saved by the reference itself), so reference checkpoints load unchanged."""
    from mcgen_amd import models
    from mcgen_amd.config import cfg
    cfg.update(model_name=model_name, device='cpu', controller_rate=0.5)
    cfg.update(extra)
    np.random.seed(0)
    m = getattr(models, model_name)()
    ref = gu.state_from_npz(gu.load_npz(fixture))
    own = m.state_dict()
    assert set(own) == set(ref), set(own) ^ set(ref)
    for k, v in ref.items():
        assert tuple(own[k].shape) == tuple(v.shape) and own[k].dtype == v.dtype, k
    m.load_state_dict(ref)                                   # strict
    # and the fused paths refuse CPU tensors instead of silently running something else
    from mcgen_amd._lib import McgenError
    with pytest.raises((McgenError, RuntimeError, NotImplementedError)):
        if model_name == 'mcglow':
            m({'img': torch.zeros(2, 1, 32, 32), 'label': torch.zeros(2, dtype=torch.int64)})
        elif model_name == 'mcpixelcnn':
            m({'img': torch.zeros(2, 8, 8, dtype=torch.int64), 'label': torch.zeros(2, dtype=torch.int64)})
        elif model_name == 'mcvae':
            m({'img': torch.zeros(2, 3, 32, 32), 'label': torch.zeros(2, dtype=torch.int64)})
        else:
            m.train(False).encode(torch.zeros(2, 3, 32, 32))


def test_process_control_fills_vqvae_and_rederives_classes():
    """utils.py:127-137,183: `ae_name == 'vqvae'` (the default) fills cfg['vqvae'], cfg['classifier'] is always set;
    classes_size follows data_name across calls unless the caller pinned another value by hand."""
    from mcgen_amd import models
    from mcgen_amd.config import cfg, process_control
    cfg.update(data_name='CIFAR10', model_name='mcpixelcnn', ae_name='vqvae', device='cpu')
    cfg.pop('classes_size', None); cfg.pop('vqvae', None)
    process_control()
    assert cfg['vqvae'] == {'hidden_size': [128, 128], 'num_res_block': 2, 'embedding_size': 64, 'num_embedding': 512,
                            'vq_commit': 0.25}
    assert cfg['classifier'] == {'hidden_size': [8, 16, 32, 64]} and cfg['classes_size'] == 10
    ae = models.vqvae()                                   # the shipped config builds the frozen auto-encoder
    assert sum(p.numel() for p in ae.parameters()) == 1868355          # VQ-VAE parameter count
    assert ae.quantizer.embedding.shape == (64, 512)
    cfg['data_name'] = 'Omniglot'
    process_control()
    assert cfg['classes_size'] == 1623                    # derived value follows the dataset
    cfg['classes_size'] = 12                              # pinned by hand: kept
    cfg['data_name'] = 'COIL100'
    process_control()
    assert cfg['classes_size'] == 12
    _cfg_for('CIFAR10')


def test_library_loader_imports_torch_first():
    """_lib.load() in a fresh interpreter that has not imported torch: afterwards torch IS loaded (one HIP runtime per
    process, torch's; loading libmcgen_hip.so first made every later launch fail on the GPU box)."""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r); import mcgen_amd._lib as L; assert 'torch' not in sys.modules; "
            "L.load(); assert 'torch' in sys.modules; print('ok')" % ROOT)
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and 'ok' in r.stdout, r.stderr[-2000:]


def test_shipped_library_reads_no_environment():
    """DESIGN 4.1 'no environment-dependent dispatch': every getenv() in csrc/ sits in the first branch of an
    `#ifdef MCGEN_TUNING` (tuning builds only); the shipped build never compiles one."""
    csrc = os.path.join(ROOT, 'multimodal-controller-for-generative-models_amd', 'csrc')
    for f in sorted(os.listdir(csrc)):
        if not f.endswith(('.hip', '.h')):
            continue
        stack = []                                   # per open #if: True while inside `#ifdef MCGEN_TUNING`'s first branch
        for ln, line in enumerate(open(os.path.join(csrc, f)), 1):
            t = line.strip()
            if t.startswith(('#ifdef', '#ifndef', '#if ')):
                stack.append(t.split()[:2] == ['#ifdef', 'MCGEN_TUNING'])
            elif t.startswith(('#else', '#elif')) and stack:
                stack[-1] = False
            elif t.startswith('#endif') and stack:
                stack.pop()
            elif 'getenv' in t and not t.startswith('//'):
                assert any(stack), f'{f}:{ln}: getenv outside #ifdef MCGEN_TUNING: {t}'


def test_tuning_switches_need_opt_in():
    """The Python-side MCGEN_* A/B switches are honoured only under MCGEN_TUNING=1 (mcgen_amd/_tuning.py): a variable left
    in the driver's environment cannot change what bench.py times, and an honoured one shows up in `tuning_switches`."""
    code = ('import sys; sys.path.insert(0, %r)\n'
            'from mcgen_amd import gan_engine as GE, trainer as T, ops, _tuning\n'
            'print(GE._GK, T._PAIR_D, ops._WG_MAX_SPLITS, sorted(_tuning.ACTIVE.items()))\n') % ROOT
    env = dict(os.environ, MCGEN_GK='0', MCGEN_PAIR_D='0', MCGEN_WGRAD_MAX_SPLITS='7')
    env.pop('MCGEN_TUNING', None)
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout.split('\n')[-2] == 'True True 128 []', r.stdout
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, env=dict(env, MCGEN_TUNING='1'), timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout.split('\n')[-2] == "False False 7 [('MCGEN_GK', '0'), ('MCGEN_PAIR_D', '0'), ('MCGEN_WGRAD_MAX_SPLITS', '7')]", r.stdout
