"""GPU: the device-side input path (mi355_collate_images / mi355_collate_annots through
objectdetection.pytorch_amd.datasettool.dataloader) against the reference's own Normalizer + collater
output (tests/golden/input.npz) and the numpy oracle.  fp32 results of float64 arithmetic: bit-exact."""
import os

import numpy as np
import pytest
import torch

from oracle import ref_input

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'input.npz')


def _samples(g, n=4):
    return [{'img': g['u8_%d' % i], 'annot': g['ann_%d' % i], 'scale': 1.0} for i in range(n)]


def test_raw_uint8_batch_matches_reference_golden():
    from objectdetection.pytorch_amd.datasettool.dataloader import DeviceCollater, Normalizer
    g = np.load(GOLD)
    out = DeviceCollater(Normalizer())(_samples(g))
    assert out['img'].is_cuda and out['img'].dtype == torch.float32
    assert np.array_equal(out['img'].cpu().numpy(), g['img'])
    assert np.array_equal(out['annot'].cpu().numpy(), g['annot'])
    assert out['scale'] == [1.0] * 4


def test_collater_dropin_on_normalised_samples():
    """The reference's call shape: float32 HWC samples the host already normalised -> pad + permute only."""
    from objectdetection.pytorch_amd.datasettool.dataloader import collater
    g = np.load(GOLD)
    data = [{'img': torch.from_numpy(ref_input.normalize(ref_input.load_scale(g['u8_%d' % i])).astype(np.float32)),
             'annot': torch.from_numpy(g['ann_%d' % i]), 'scale': 1.0} for i in range(4)]
    out = collater(data)
    assert np.array_equal(out['img'].cpu().numpy(), g['img'])
    assert np.array_equal(out['annot'].cpu().numpy(), g['annot'])


def test_all_empty_annotations_give_one_pad_row():
    from objectdetection.pytorch_amd.datasettool.dataloader import DeviceCollater, Normalizer
    g = np.load(GOLD)
    data = [{'img': g['u8_%d' % i], 'annot': np.zeros((0, 5)), 'scale': 1.0} for i in range(2)]
    out = DeviceCollater(Normalizer())(data)
    assert np.array_equal(out['annot'].cpu().numpy(), g['annot_all_empty'])
    assert out['annot'].shape == (2, 1, 5)


def test_float_images_in_unit_range_and_resizer_canvas_and_scaled_boxes():
    from objectdetection.pytorch_amd.datasettool.dataloader import DeviceCollater, Normalizer
    rng = np.random.default_rng(5)
    imgs = [rng.random((h, w, 3)).astype(np.float32) for h, w in ((50, 70), (60, 33), (17, 90))]
    annots = [np.concatenate([rng.random((m, 4)) * 97.3, rng.integers(0, 20, (m, 1)).astype(np.float64)], axis=1)
              for m in (2, 7, 0)]
    scales = [0.731, 1.0 / 3.0, 1.9]
    col = DeviceCollater(Normalizer(), height=60, width=90, base_size=32, scale_annots=True)
    out = col([{'img': i, 'annot': a, 'scale': s} for i, a, s in zip(imgs, annots, scales)])
    want_img, want_ann = ref_input.collate([ref_input.canvas(ref_input.normalize(i), 60, 90, 32) for i in imgs],
                                           annots, scales)
    assert out['img'].shape == (3, 3, 64, 96)
    assert np.array_equal(out['img'].cpu().numpy(), want_img)
    assert np.array_equal(out['annot'].cpu().numpy(), want_ann)
    with pytest.raises(ValueError):
        col([{'img': np.zeros((70, 10, 3), np.float32), 'annot': np.zeros((0, 5))}])
    # no canvas: batch maximum 60 x 90 (a width that is not a multiple of 4: the scalar store path)
    out = DeviceCollater(Normalizer())([{'img': i, 'annot': a} for i, a in zip(imgs, annots)])
    want_img, want_ann = ref_input.collate([ref_input.normalize(i) for i in imgs], annots)
    assert out['img'].shape == (3, 3, 60, 90)
    assert np.array_equal(out['img'].cpu().numpy(), want_img)
    assert np.array_equal(out['annot'].cpu().numpy(), want_ann)


def test_full_size_batch_and_staging_reuse():
    """BASELINE configs[1] input: 16 x 512 x 512 decoded images, several batches through the two pinned buffers."""
    from objectdetection.pytorch_amd.datasettool.dataloader import DeviceCollater, Normalizer
    col = DeviceCollater(Normalizer())
    rng = np.random.default_rng(9)
    for it in range(3):
        imgs = [rng.integers(0, 256, size=(512 - 32 * (b % 3 == it), 512, 3), dtype=np.uint8) for b in range(16)]
        annots = [np.concatenate([rng.random((8, 4)) * 500.0, rng.integers(0, 80, (8, 1)).astype(np.float64)], axis=1)
                  for _ in range(16)]
        out = col([{'img': i, 'annot': a} for i, a in zip(imgs, annots)])
        want_img, want_ann = ref_input.collate([ref_input.normalize(ref_input.load_scale(i)) for i in imgs], annots)
        assert np.array_equal(out['img'].cpu().numpy(), want_img)
        assert np.array_equal(out['annot'].cpu().numpy(), want_ann)


def test_rejects_mixed_or_wrong_inputs():
    from objectdetection.pytorch_amd.datasettool.dataloader import DeviceCollater
    col = DeviceCollater()
    with pytest.raises(TypeError):
        col([{'img': np.zeros((4, 4, 3), np.uint8), 'annot': np.zeros((0, 5))},
             {'img': np.zeros((4, 4, 3), np.float32), 'annot': np.zeros((0, 5))}])
    with pytest.raises(ValueError):
        col([{'img': np.zeros((4, 4), np.uint8), 'annot': np.zeros((0, 5))}])
    with pytest.raises(ValueError):
        col([])
