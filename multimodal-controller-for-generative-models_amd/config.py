"""Global configuration dict, the same role as the reference's ``config.py`` + ``config.yml``
(src/config.py:1-5) and the hyper-parameter table of ``process_control`` (src/utils.py:104-192).

``cfg`` starts from the reference's defaults; ``process_control()`` derives the model sizes
from ``data_name`` / ``model_name`` / ``control`` exactly as the reference table does.
"""
from __future__ import annotations

cfg = {
    'control': {'controller_rate': '0.5'},
    'data_name': 'CIFAR10', 'subset': 'label',
    'batch_size': {'train': 128, 'test': 128},
    'shuffle': {'train': True, 'test': False},
    'num_workers': 0,
    'model_name': 'mcgan',
    'optimizer_name': 'Adam', 'lr': 1.0e-3, 'momentum': 0, 'weight_decay': 0,
    'scheduler_name': 'None',
    'init_seed': 0, 'num_experiments': 1, 'num_epochs': 200, 'log_interval': 0.25,
    'device': 'cuda', 'world_size': 1, 'resume_mode': 0,
    'ae_name': 'vqvae',
    # mcgen_amd extension: arithmetic type of the fused kernels ('float32' parity / 'bfloat16' throughput)
    'compute_dtype': 'float32',
}

_CLASSES = {'MNIST': 10, 'FashionMNIST': 10, 'SVHN': 10, 'CIFAR10': 10, 'CIFAR100': 100,
            'COIL100': 100, 'Omniglot': 1623}


def process_control():
    """utils.py:104-192 restated as a table: data shape, per-model sizes, batch sizes."""
    if 'controller_rate' in cfg['control']:
        cfg['controller_rate'] = float(cfg['control']['controller_rate'])
    name = cfg['data_name']
    shapes = {'MNIST': ([1, 32, 32], 1000), 'FashionMNIST': ([1, 32, 32], 1000), 'Omniglot': ([1, 32, 32], 20),
              'SVHN': ([3, 32, 32], 1000), 'CIFAR10': ([3, 32, 32], 1000), 'COIL100': ([3, 32, 32], 100),
              'ImageNet32': ([3, 32, 32], 20), 'CelebA-HQ': ([3, 128, 128], 20), 'ImageNet': ([3, 128, 128], 20)}
    if name not in shapes:
        raise ValueError('Not valid dataset')
    cfg['data_shape'], cfg['generate_per_mode'] = list(shapes[name][0]), shapes[name][1]
    if name in _CLASSES:
        # the reference takes it from the dataset object (utils.py:99-101, process_dataset).  Derived here on every call:
        # a value this function derived for ANOTHER data_name is replaced, a value the caller set by hand is kept.
        derived = cfg.get('_classes_size_derived')
        stale = derived is not None and derived[0] != name and cfg.get('classes_size') == derived[1]
        if 'classes_size' not in cfg or stale:
            cfg['classes_size'] = _CLASSES[name]
            cfg['_classes_size_derived'] = (name, _CLASSES[name])
    side = cfg['data_shape'][1]
    if side not in (32, 128):
        raise ValueError('Not valid data shape')
    if cfg.get('ae_name') in ('vqvae',):                     # utils.py:127-137
        cfg['vqvae'] = {'hidden_size': [128, 128] if side == 32 else [128, 128, 128, 128], 'num_res_block': 2,
                        'embedding_size': 64, 'num_embedding': 512, 'vq_commit': 0.25}
    cfg['classifier'] = {'hidden_size': [8, 16, 32, 64]}      # utils.py:183
    model = cfg['model_name']
    if model in ('cgan', 'mcgan'):
        gan = {'latent_size': 128, 'embedding_size': 32}
        if side == 32 and name in ('CIFAR10',):
            gan['generator_hidden_size'], gan['discriminator_hidden_size'] = [256] * 4, [128] * 4
        elif side == 32:
            gan['generator_hidden_size'], gan['discriminator_hidden_size'] = [512, 256, 128, 64], [64, 128, 256, 512]
        else:
            gan['generator_hidden_size'] = [1024, 512, 256, 128, 64]
            gan['discriminator_hidden_size'] = [64, 128, 256, 512, 1024]
        cfg['gan'] = gan
    elif model in ('cvae', 'mcvae'):
        cfg['vae'] = ({'hidden_size': [64, 128, 256], 'latent_size': 128} if side == 32 else
                      {'hidden_size': [64, 128, 256, 512, 512], 'latent_size': 256})
        cfg['vae'].update(num_res_block=2, embedding_size=32)
    elif model in ('cglow', 'mcglow'):
        cfg['glow'] = {'hidden_size': 512, 'K': 16, 'L': 3 if side == 32 else 5, 'affine': True, 'conv_lu': True}
    elif model in ('cpixelcnn', 'mcpixelcnn'):
        cfg['pixelcnn'] = {'num_layer': 15, 'hidden_size': 128, 'num_embedding': 512}
    cfg['batch_size'] = {'train': 128, 'test': 512} if side == 32 else {'train': 32, 'test': 128}
    return
