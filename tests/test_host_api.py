"""CPU-side host logic: API surface, state_dict compatibility with the reference (keys + OIHW
shapes recorded in the golden files), symbol export of the C ABI, loud failure without a GPU."""
import os

import numpy as np
import pytest
import torch

from oracle import ref_cpu


def test_c_abi_exports_every_declared_symbol():
    from objectdetection.pytorch_amd import _lib
    names = _lib.declared_symbols()
    assert 'mi355_conv_gather' in names and 'mi355_focal_loss_fwd' in names and len(names) >= 25
    lib = _lib.lib()
    for n in names:
        assert hasattr(lib, n), 'libmi355det.so does not export %s' % n
    assert lib.mi355_version() >= 1


@pytest.mark.parametrize('tag,depth', [('r18_256', 18), ('r50_128x160', 50), ('r34_64x96', 34), ('r101_64x96', 101)])
def test_state_dict_matches_reference(golden_dir, tag, depth):
    from objectdetection.pytorch_amd.architectures import retinanet
    g = np.load(os.path.join(golden_dir, 'e2e_%s.npz' % tag))
    nc = int(g['num_classes'])
    model = retinanet.retinanet(depth, num_classes=nc)
    sd = model.state_dict()
    assert list(sd.keys()) == [str(k) for k in g['state_keys']]
    assert [','.join(map(str, v.shape)) for v in sd.values()] == [str(s) for s in g['state_shapes']]
    # reference init values (retinanet.py:240-246)
    assert float(sd['classificationModel.output.weight'].abs().max()) == 0.0
    np.testing.assert_allclose(sd['classificationModel.output.bias'].numpy(), -np.log(99.0), rtol=1e-6)
    assert all(not m.training for m in model.modules() if isinstance(m, torch.nn.BatchNorm2d))
    # a reference-layout (contiguous OIHW) checkpoint loads, masters stay KRSC in memory
    model.load_state_dict({k: v.contiguous() for k, v in sd.items()})
    assert model.layer1[0].conv1.weight.stride(1) == 1


@pytest.mark.parametrize('tag,ctor', [('tiny_64x96', 'retinanet_tiny'), ('p45p6_192x288', 'retinanet_p45p6')])
def test_three_level_variant_state_dict_and_anchor_config(golden_dir, tag, ctor):
    """retinanet_tiny / retinanet_p45p6 (retinanet.py:672-679, :921-923): state_dict keys + shapes and the
    anchor configuration (levels, strides, sizes, ratios) of the reference."""
    from objectdetection.pytorch_amd.architectures import retinanet
    g = np.load(os.path.join(golden_dir, 'e2e_%s.npz' % tag))
    model = getattr(retinanet, ctor)(int(g['num_classes']))
    sd = model.state_dict()
    assert list(sd.keys()) == [str(k) for k in g['state_keys']]
    assert [','.join(map(str, v.shape)) for v in sd.values()] == [str(s) for s in g['state_shapes']]
    a = model.anchors
    if ctor == 'retinanet_tiny':
        assert [list(map(float, v)) for v in (a.pyramid_levels, a.strides, a.sizes, a.ratios)] == \
            [[2, 3, 4], [4, 8, 16], [16, 32, 64], [1, 1.5, 2]]
        assert not hasattr(model, 'maxpool') and not hasattr(model, 'layer4')
    else:
        assert [list(map(float, v)) for v in (a.pyramid_levels, a.strides, a.sizes, a.ratios)] == \
            [[4, 5, 6], [16, 32, 96], [32, 64, 160], [1, 1.5, 2]]
        assert model.fpn.P6.kernel_size == (5, 5) and model.fpn.P6.stride == (3, 3)
    with pytest.raises(RuntimeError):
        getattr(retinanet, ctor)(3, pretrained=True)


def test_factories_and_attrs():
    from objectdetection.pytorch_amd.architectures import retinanet, anchors, losses, utils
    m = retinanet.build_retinanet('retinanet-res18', num_classes=3)
    assert isinstance(m.anchors, anchors.Anchors) and isinstance(m.focalLoss, losses.FocalLoss)
    assert isinstance(m.regressBoxes, utils.BBoxTransform) and isinstance(m.clipBoxes, utils.ClipBoxes)
    assert m.anchors.strides == [8, 16, 32, 64, 128] and m.anchors.sizes == [32, 64, 128, 256, 512]
    assert sum(p.numel() for p in retinanet.retinanet18(80).parameters()) == 21410100
    with pytest.raises(RuntimeError):
        retinanet.retinanet50(80, pretrained=True)           # must never fetch
    m.train()
    m.freeze_bn()
    assert not m.bn1.training and m.training


def test_no_cpu_fallback():
    """The product path must fail loudly off-GPU (never route through the oracle)."""
    from objectdetection.pytorch_amd.architectures import retinanet, losses, utils
    m = retinanet.retinanet18(4)
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 3, 64, 64), torch.zeros(1, 1, 5))
    with pytest.raises(RuntimeError):
        m.layer1[0](torch.zeros(1, 64, 8, 8))
    with pytest.raises(RuntimeError):
        losses.FocalLoss()(torch.rand(1, 9, 2), torch.rand(1, 9, 4), torch.rand(1, 9, 4), -torch.ones(1, 1, 5))
    with pytest.raises(RuntimeError):
        utils.BBoxTransform()(torch.rand(1, 9, 4), torch.rand(1, 9, 4))
    import objectdetection.pytorch_amd as pkg
    src = open(os.path.join(os.path.dirname(pkg.__file__), 'engine.py')).read()
    assert 'oracle' not in src


def test_oracle_state_shapes_helper():
    s = ref_cpu.retinanet_state_shapes(50, 80)
    assert len(s) == 354


@pytest.mark.parametrize('tag,coef', [('effdet_d0_128', 0), ('effdet_d2_128x256', 2), ('effdet_d6_128', 6)])
def test_efficientdet_state_dict_matches_reference(golden_dir, tag, coef):
    from objectdetection.pytorch_amd.architectures import efficientdet
    g = np.load(os.path.join(golden_dir, 'e2e_%s.npz' % tag))
    model = efficientdet.build_efficientdet('efficientdet-d%d' % coef, num_classes=int(g['num_classes']))
    sd = model.state_dict()
    assert list(sd.keys()) == [str(k) for k in g['state_keys']]           # 784 (D0) / 1076 (D2) entries
    assert [','.join(map(str, v.shape)) for v in sd.values()] == [str(s) for s in g['state_shapes']]
    assert model.anchors.num_anchors_per_cell == 6 and list(model.anchors.ratios) == [1, 2]
    assert all(not m.training for m in model.modules() if isinstance(m, torch.nn.BatchNorm2d))
    with pytest.raises(RuntimeError):
        model(torch.zeros(1, 3, 128, 128), torch.zeros(1, 1, 5))          # no CPU fallback
    with pytest.raises(RuntimeError):
        efficientdet.EfficientDet(num_classes=3, compound_coef=0, load_weights=True)   # never fetch


@pytest.mark.parametrize('coef', range(8))
def test_every_efficientdet_variant_matches_the_reference_tables(golden_dir, coef):
    """efficientdet.py:442-495 (compound-coefficient tables) and :588: D0 .. D7 built by the mirrored constructor have the
    reference's state_dict keys and shapes, parameter count, anchor configuration and attention / plain-sum BiFPN choice
    (fixture made by building every variant with the reference: tools/make_golden.py:gen_effdet_variants)."""
    from objectdetection.pytorch_amd.architectures import efficientdet
    g = np.load(os.path.join(golden_dir, 'effdet_variants.npz'))
    model = efficientdet.build_efficientdet('efficientdet-d%d' % coef, num_classes=7)
    sd = model.state_dict()
    pre = 'd%d/' % coef
    assert list(sd.keys()) == [str(k) for k in g[pre + 'keys']]
    assert [','.join(map(str, v.shape)) for v in sd.values()] == [str(x) for x in g[pre + 'shapes']]
    assert sum(p.numel() for p in model.parameters()) == int(g[pre + 'nparams'])
    a = model.anchors
    assert list(a.pyramid_levels) == list(g[pre + 'anchor_levels'])
    np.testing.assert_allclose(np.array(a.strides, dtype=np.float64), g[pre + 'anchor_strides'])
    np.testing.assert_allclose(np.array(a.sizes, dtype=np.float64), g[pre + 'anchor_sizes'])
    np.testing.assert_allclose(np.array(a.ratios, dtype=np.float64), g[pre + 'anchor_ratios'])
    np.testing.assert_allclose(np.array(a.scales, dtype=np.float64), g[pre + 'anchor_scales'])
    assert [bool(getattr(c, 'attention', True)) for c in model.bifpn] == [bool(x) for x in g[pre + 'attention']]


def test_wgrad_split_rows_are_whole_granules():
    """conv_wgrad takes pixel chunks in multiples of 64 rows whatever the layer size (the relaxed floor for
    launches that would leave CUs without a block included)."""
    from objectdetection.pytorch_amd.ops_conv import pick_split_rows
    for total in (1, 63, 64, 1000, 1024, 4096, 16 * 40 * 40, 16 * 64 * 64 + 7, 8 * 96 * 96, 16 * 256 * 256):
        for tiles in (1, 2, 3, 5, 16, 24, 72, 144):
            for floor in (512, 1024):
                rows = pick_split_rows(total, tiles, 512, floor, 64)
                assert rows % 64 == 0 and rows >= 256, (total, tiles, floor, rows)
    assert pick_split_rows(16 * 64 * 64, 1, 512, 1024, 64) == 512        # one small tile: a block per two CUs (128 blocks)
    assert pick_split_rows(16 * 32 * 32, 16, 512, 1024, 64) == 1024       # enough tiles: the floor stands


def test_package_import_raises_hw_queue_count():
    """The two compute streams of a plan need their own hardware queues beside RCCL's streams: importing the
    package sets GPU_MAX_HW_QUEUES (read when the HIP runtime starts) unless the user already chose a value."""
    import subprocess
    import sys
    code = ("import os; os.environ.pop('GPU_MAX_HW_QUEUES', None); import objectdetection.pytorch_amd; "
            "print(os.environ['GPU_MAX_HW_QUEUES'])")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, '-c', code], cwd=root, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    assert int(out.stdout.strip()) >= 8
    code = ("import os; os.environ['GPU_MAX_HW_QUEUES'] = '5'; import objectdetection.pytorch_amd; "
            "print(os.environ['GPU_MAX_HW_QUEUES'])")
    out = subprocess.run([sys.executable, '-c', code], cwd=root, capture_output=True, text=True, timeout=300)
    assert out.stdout.strip() == '5'


@pytest.mark.parametrize('tag,neck', [('kseven_r18_fpn_128x160', 'fpn'), ('kseven_r18_panet_128', 'panet-fpn'),
                                      ('kseven_r18_bifpn_128x256', 'bifpn'), ('kseven_r18_bifpn2_128', 'bifpn2')])
def test_ksevendet_state_dict_and_config_contract(golden_dir, tag, neck):
    """`KSevenDet(cfg, num_classes, pretrained=False)` as train_ksevendet.py:85-89, :207 builds it from
    the YAML model config; state_dict keys / shapes equal those of the reference modules the golden was made with
    (backbone trunk, neck/fpn.py PANetFPN, head/simple_head.py Regressor / Classifier)."""
    import yaml
    from objectdetection.pytorch_amd.ksevendet.architecture import ksevendet
    from objectdetection.pytorch_amd.ksevendet.architecture.backbone import registry
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    bifpn = neck.startswith('bifpn')
    cfg = yaml.safe_load(open(os.path.join(root, 'objectdetection', 'pytorch_amd', 'config', 'model',
                                           'ksevendet_resnet_bifpn.yaml' if bifpn else 'ksevendet_resnet.yaml')))
    assert cfg.pop('architecture') == 'ksevendet'
    if neck == 'bifpn2':                      # two BiFPN repeats, heads on all five neck levels (as the golden)
        cfg['neck_config']['bifpn_repeats'] = 2
        cfg['head_feature_pyramid_levels'] = [3, 4, 5, 6, 7]
    elif not bifpn:
        cfg['neck'] = neck
    g = np.load(os.path.join(golden_dir, 'e2e_%s.npz' % tag))
    model = ksevendet.KSevenDet(cfg, num_classes=int(g['num_classes']), pretrained=False)
    sd = model.state_dict()
    assert list(sd.keys()) == [str(k) for k in g['state_keys']]
    assert [','.join(map(str, v.shape)) for v in sd.values()] == [str(s) for s in g['state_shapes']]
    assert model.anchors.num_anchors_per_cell == 6
    assert list(model.anchors.sizes) == [32, 64, 128, 256, 512][:len(cfg.get('head_feature_pyramid_levels', [3, 4, 5, 6, 7]))]
    assert all(not m.training for m in model.modules() if isinstance(m, torch.nn.BatchNorm2d))    # ksevendet.py:164
    assert registry.is_model('resnet50') and 'resnet18' in registry._module_to_models['resnet']
    with pytest.raises(RuntimeError):
        model(torch.zeros(1, 3, 128, 128), torch.zeros(1, 1, 5), return_loss=True)                # no CPU fallback
    with pytest.raises(NotImplementedError):   # a stack of TFDBiFPN cells (ksevendet.py:87-104): not built
        ksevendet.KSevenDet(dict(cfg, neck='tfdbifpn', neck_config=dict(cfg['neck_config'], bifpn_repeats=1, bifpn_attention=False)),
                            num_classes=3)
    if bifpn:                                  # the reference's attention path cannot run either (neck/bifpn.py:372)
        with pytest.raises(NotImplementedError):
            ksevendet.KSevenDet(dict(cfg, neck_config=dict(cfg['neck_config'], bifpn_attention=True)), num_classes=3)
    with pytest.raises(NotImplementedError):
        ksevendet.KSevenDet(dict(cfg, head_config=dict(cfg['head_config'], act_type='swish')), num_classes=3)
